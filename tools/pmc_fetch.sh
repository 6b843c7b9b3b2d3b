#!/bin/bash
# HBM-side traffic of one conv / GEMM shape: FETCH_SIZE / WRITE_SIZE only, own pass (other TCC/TCP counter groups hang
# rocprofv3 on this image).  usage: tools/pmc_fetch.sh <tag> <pmc_one args...>
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pmcf_$tag
timeout 240 rocprofv3 --kernel-trace --pmc FETCH_SIZE WRITE_SIZE --output-format csv -d /tmp/pmcf_${tag}_1 -o run -- python $GRAFT_REPO_ROOT/tools/pmc_one.py "$@" > /tmp/pmcf_$tag.log 2>&1
echo "rc=$?"
python $GRAFT_REPO_ROOT/tools/pmc_parse.py /tmp/pmcf_${tag}_ 1 > $GRAFT_REPO_ROOT/gpurun_out/pmcf_${tag}.txt
cat $GRAFT_REPO_ROOT/gpurun_out/pmcf_${tag}.txt
