#!/bin/bash
# Kernel totals of BASELINE config 5's training part (bench.py --mode robft: distortion + SecretDecoder train step, batch 16) under
# rocprofv3 --kernel-trace --stats.   usage (GPU box): tools/prof_robft.sh <tag> [ENV=VAL ...]  -> gpurun_out/robft_<tag>_stats.txt
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prob_$tag
env "$@" rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prob_$tag -o run -- python $GRAFT_REPO_ROOT/bench.py --mode robft --steps 4 --warmup 1 > /tmp/prob_$tag.log 2>&1
cd $GRAFT_REPO_ROOT
tail -1 /tmp/prob_$tag.log | cut -c1-500 > gpurun_out/robft_${tag}_line.txt
F=$(find /tmp/prob_$tag -name "*kernel_stats.csv" | head -1)
python - "$F" > gpurun_out/robft_${tag}_stats.txt <<'PY'
import csv, sys, re
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
calls = sum(int(r["Calls"]) for r in rows)
print(f"total kernel time {tot / 1e6:.1f} ms in {calls} launches (5 steps: 1 warm-up + 4 timed)")
print(f"{'%':>6} {'ms':>9} {'calls':>7} {'avg us':>8}  kernel")
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:40]:
    name = re.sub(r"\(anonymous namespace\)::", "", r["Name"])[:120]
    print(f"{100 * float(r['TotalDurationNs']) / tot:6.2f} {float(r['TotalDurationNs']) / 1e6:9.2f} {int(r['Calls']):7d} {float(r['AverageNs']) / 1e3:8.1f}  {name}")
PY
T=$(find /tmp/prob_$tag -name "*kernel_trace.csv" | head -1)
python - "$T" >> gpurun_out/robft_${tag}_stats.txt <<'PY'
import csv, sys, re, collections
agg = collections.defaultdict(lambda: [0, 0.0])
for r in csv.DictReader(open(sys.argv[1])):
    n = re.sub(r"\(anonymous namespace\)::|void ", "", r["Kernel_Name"]); n = re.sub(r"\(.*", "", n)[:40]
    if not any(t in n for t in ("gemm_f32", "chan_reduce", "dw_kernel", "bn_apply", "stem")): continue
    k = (n, int(r["Grid_Size_X"]) // 256, int(r["Grid_Size_Y"]), int(r["Grid_Size_Z"]))
    agg[k][0] += 1; agg[k][1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
print("\nper (kernel, grid in workgroups), all 5 steps:")
for k, (c, us) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:40]:
    print(f"{us / 1e3:8.2f} ms {c:5d}x {us / c:8.1f} us  {k[0]} grid=({k[1]},{k[2]},{k[3]})")
PY
