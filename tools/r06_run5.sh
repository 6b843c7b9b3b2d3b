cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r06
python tools/probe_tntr.py > gpurun_out/r06/probe_tntr.txt 2>&1; grep -E "FAIL|ALL PASS|SOME|tn_tr160|time M|problems on" gpurun_out/r06/probe_tntr.txt | tail -30
ROUNDS=2 BENCH_ARGS="--config 3 --steps 15 --warmup 3 --no-cpu-baseline --no-extras" tools/ab_bench.sh gpurun_out/r06/ab_tntr160_c3.txt "AQL_TNTR160=0" "AQL_TNTR160=1"
