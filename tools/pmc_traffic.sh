#!/bin/bash
# HBM-side traffic of one conv / GEMM shape as MI355X_MICROARCH.md prescribes: FETCH_SIZE and WRITE_SIZE in SEPARATE
# rocprofv3 --pmc passes (they need 3 + 2 of the 4 TCC slots), kernel-trace only.
# usage: tools/pmc_traffic.sh <tag> <pmc_one args...>      -> gpurun_out/pmct_<tag>.txt
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
i=0
for C in FETCH_SIZE WRITE_SIZE; do
  i=$((i+1)); rm -rf /tmp/pmct_${tag}_$i
  timeout 240 rocprofv3 --kernel-trace --pmc $C --output-format csv -d /tmp/pmct_${tag}_$i -o run -- python $GRAFT_REPO_ROOT/tools/pmc_one.py "$@" > /tmp/pmct_${tag}_$i.log 2>&1
  echo "$C rc=$?"
done
python $GRAFT_REPO_ROOT/tools/pmc_parse.py /tmp/pmct_${tag}_ 2 > $GRAFT_REPO_ROOT/gpurun_out/pmct_${tag}.txt
cat $GRAFT_REPO_ROOT/gpurun_out/pmct_${tag}.txt
