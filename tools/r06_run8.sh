cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r06
ROUNDS=2 BENCH_ARGS="--config 3 --steps 15 --warmup 3 --no-cpu-baseline --no-extras" tools/ab_bench.sh gpurun_out/r06/ab_tntr_ring_c3.txt "AQL_TNTR160=0 AQL_TNTR_NST=0" "AQL_TNTR160=0 AQL_TNTR_NST=2" "AQL_TNTR160=0 AQL_TNTR_NST=3" "AQL_TNTR160=1"
