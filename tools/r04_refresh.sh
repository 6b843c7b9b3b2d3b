#!/bin/bash
# End-of-round refresh of the judged artefacts (round 4): kernel traces + families of config 2 and config 3 under rocprofv3
# --kernel-trace, the PMC traffic passes of the roofline kernel (FETCH_SIZE / WRITE_SIZE in separate passes) and the MfmaUtil pass.
# Everything lands in gpurun_out/r04/; the summaries that are judged are copied into profiles/r04_* by hand.
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04; mkdir -p $O
set -x
tools/insitu_profile.sh r04final > $O/insitu_head.txt 2>&1
BENCH_EXTRA="--config 3" FAM_BATCH=8 FAM_RANK=320 tools/insitu_profile.sh r04final_c3 >> $O/insitu_head.txt 2>&1
cp gpurun_out/insitu_r04final* $O/
bash tools/pmc_traffic.sh geglu geglu 32768 1280 320 > /dev/null 2>&1; cp gpurun_out/pmct_geglu.txt $O/
bash tools/pmc_traffic.sh conv8 conv 8 64 320 320 > /dev/null 2>&1; cp gpurun_out/pmct_conv8.txt $O/
bash tools/r03_pmc_mfma.sh > $O/pmc_mfma_util.txt 2>&1
