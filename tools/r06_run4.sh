cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r06
AQL_DEFER_FINALIZE=1 python -m pytest tests/test_gpu_parity.py -x -q -k "network_alpha or tiny" > gpurun_out/r06/t4a.log 2>&1; tail -3 gpurun_out/r06/t4a.log | cut -c1-300
python -m pytest tests -m gpu -x -q > gpurun_out/r06/gpu_tests_mid.txt 2>&1; tail -4 gpurun_out/r06/gpu_tests_mid.txt | cut -c1-300
BENCH_EXTRA="--config 3" FAM_BATCH=8 FAM_RANK=320 tools/insitu_profile.sh r06mid_c3 > gpurun_out/r06/insitu_mid_c3.txt 2>&1
cp gpurun_out/insitu_r06mid_c3* gpurun_out/r06/ 2>/dev/null
head -45 gpurun_out/r06/insitu_r06mid_c3_summary.txt
