#!/bin/bash
# A/B the captured train step on ONE box (box-to-box spread is +-0.5 ms, larger than most single optimisations):
#   tools/ab_bench.sh OUT "ENV_A" "ENV_B" [...]    e.g.  tools/ab_bench.sh gpurun_out/ab.txt "AQL_GEGLU_FUSED=0" "AQL_GEGLU_FUSED=1"
# Runs every variant ROUNDS times, interleaved, and prints the HIP-event median / min step time of each run.
out=$1; shift
rounds=${ROUNDS:-2}
args=${BENCH_ARGS:---steps 30 --warmup 5 --no-cpu-baseline --no-extras}
: > "$out"
for r in $(seq 1 $rounds); do
  for v in "$@"; do
    line=$(env $v python bench.py $args 2>/dev/null | grep '^{' | tail -1)
    echo "$v round $r: $(echo "$line" | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print("median %.3f ms  min %.3f ms  wall %.3f ms  loss %.6f" % (d["ms_per_step_hip_event_median"], d["ms_per_step_hip_event_min"], d["ms_per_step"], d["config"]["loss"]))')" | tee -a "$out"
  done
done
