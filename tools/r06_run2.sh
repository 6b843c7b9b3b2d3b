cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r06
python tools/probe_defer.py > gpurun_out/r06/probe_defer.txt 2>&1; tail -12 gpurun_out/r06/probe_defer.txt
ROUNDS=2 tools/ab_bench.sh gpurun_out/r06/ab_defer_c2.txt "AQL_DEFER_FINALIZE=0" "AQL_DEFER_FINALIZE=1"
ROUNDS=2 BENCH_ARGS="--config 3 --steps 15 --warmup 3 --no-cpu-baseline --no-extras" tools/ab_bench.sh gpurun_out/r06/ab_defer_c3.txt "AQL_DEFER_FINALIZE=0" "AQL_DEFER_FINALIZE=1"
for v in 0 1 0 1; do AQL_DEFER_FINALIZE=$v python bench.py --mode infer --steps 10 2>/dev/null | grep '^{' | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('infer DEFER=$v', d.get('sampling_only'), d.get('bits_equal_to_oracle'))" | tee -a gpurun_out/r06/ab_defer_infer.txt; done
