#!/bin/bash
# SQ counter pass (one rocprofv3 --pmc group, kernel-trace only) over a workload script, summarised per kernel name.
# PMC_GROUP="FETCH_SIZE" (or "WRITE_SIZE") switches to an HBM-traffic pass (one of them per pass: TCC slots).
# usage: tools/pmc_kernels.sh <tag> <script path relative to the repo root> [args...]     -> gpurun_out/pmck_<tag>.txt
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
G="${PMC_GROUP:-SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU GRBM_GUI_ACTIVE}"
rm -rf /tmp/pmck_$tag
timeout 600 rocprofv3 --kernel-trace --pmc $G --output-format csv -d /tmp/pmck_$tag -o run -- env PYTHONPATH=$GRAFT_REPO_ROOT python $GRAFT_REPO_ROOT/"$@" > /tmp/pmck_$tag.log 2>&1
echo "rc=$?"
python $GRAFT_REPO_ROOT/tools/pmc_kernels_parse.py /tmp/pmck_$tag > $GRAFT_REPO_ROOT/gpurun_out/pmck_$tag.txt
cat $GRAFT_REPO_ROOT/gpurun_out/pmck_$tag.txt
