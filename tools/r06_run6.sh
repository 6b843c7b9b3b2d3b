cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r06
python -m pytest tests/test_gpu_kernels.py -x -q -k "rank320 or transpose_read" > gpurun_out/r06/t6.log 2>&1; tail -3 gpurun_out/r06/t6.log | cut -c1-300
ROUNDS=2 BENCH_ARGS="--config 3 --steps 15 --warmup 3 --no-cpu-baseline --no-extras" tools/ab_bench.sh gpurun_out/r06/ab_tps_c3.txt "AQL_TN_TPS=32" "AQL_TN_TPS=64" "AQL_TN_TPS=128" "AQL_TN_TPS=16"
python -m pytest tests/test_full_size.py tests/test_gpu_parity.py -x -q -k "320" > gpurun_out/r06/t6b.log 2>&1; tail -3 gpurun_out/r06/t6b.log | cut -c1-300
