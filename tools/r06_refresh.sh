#!/bin/bash
# End-of-round refresh of the judged artefacts (round 6): kernel traces + families of config 2 and config 3, the configs-4 / 5 / VAE
# kernel summaries that round 4 left in gpurun_out/ only, the PMC passes (traffic of the roofline kernel and of a chain launch, MfmaUtil).
# Everything lands in gpurun_out/r06/; the summaries that are judged are copied into profiles/r06_* afterwards.
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06; mkdir -p $O
set -x
tools/insitu_profile.sh r06final > $O/insitu_head.txt 2>&1
BENCH_EXTRA="--config 3" FAM_BATCH=8 FAM_RANK=320 tools/insitu_profile.sh r06final_c3 >> $O/insitu_head.txt 2>&1
cp gpurun_out/insitu_r06final* $O/
tools/prof_robft.sh r06 > /dev/null 2>&1; cp gpurun_out/robft_r06_stats.txt gpurun_out/robft_r06_line.txt $O/
tools/prof_extract.sh 1 > /dev/null 2>&1; cp gpurun_out/extract_b1_stats.txt $O/
tools/prof_extract.sh 16 > /dev/null 2>&1; cp gpurun_out/extract_b16_stats.txt $O/
tools/prof_vae.sh > /dev/null 2>&1; cp gpurun_out/prof_vae.txt $O/
tools/prof_infer.sh r06 > /dev/null 2>&1; cp gpurun_out/infer_r06_stats.txt gpurun_out/infer_r06_line.txt $O/
bash tools/pmc_traffic.sh geglu geglu 32768 1280 320 > /dev/null 2>&1; cp gpurun_out/pmct_geglu.txt $O/
bash tools/pmc_traffic.sh conv8 conv 8 64 320 320 > /dev/null 2>&1; cp gpurun_out/pmct_conv8.txt $O/
bash tools/pmc_traffic.sh chain chain 32768 > /dev/null 2>&1; cp gpurun_out/pmct_chain.txt $O/
bash tools/r03_pmc_mfma.sh > $O/pmc_mfma_util.txt 2>&1
ls -la $O
bash tools/pmc_robft.sh > /dev/null 2>&1; cp gpurun_out/pmc_robft.json $O/
python tools/cmp_vendor.py > $O/cmp_vendor.txt 2>&1
tools/prof_infer_seq.sh r06 > /dev/null 2>&1; cp gpurun_out/infer_r06_sequence.txt $O/
python tools/time_defer.py > $O/time_defer.txt 2>&1
tools/micro/atomic_dq > $O/atomic_dq.txt 2>&1
ls -la $O
