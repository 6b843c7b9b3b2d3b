cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r06
python -m pytest tests/test_gpu_kernels.py -x -q -k "rank320_grouped or splitk_finalize_inside" -s > gpurun_out/r06/t3.log 2>&1; tail -15 gpurun_out/r06/t3.log | cut -c1-400
ROUNDS=2 BENCH_ARGS="--config 3 --steps 15 --warmup 3 --no-cpu-baseline --no-extras" tools/ab_bench.sh gpurun_out/r06/ab_wide_c3.txt "AQL_GROUPED_WIDE=0 AQL_DEFER_FINALIZE=0" "AQL_GROUPED_WIDE=1 AQL_DEFER_FINALIZE=0"
python -m pytest tests/test_full_size.py tests/test_gpu_parity.py -x -q -k "320" > gpurun_out/r06/t3b.log 2>&1; tail -5 gpurun_out/r06/t3b.log | cut -c1-300
