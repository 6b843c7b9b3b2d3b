#!/bin/bash
# Kernel trace of BASELINE config 4's sampling loop (batch 1 = U-Net batch 2 under CFG): per-kernel totals of one bench --mode infer run.
# usage (GPU box): tools/prof_infer.sh <tag> [ENV=VAL ...]  -> gpurun_out/infer_<tag>_stats.txt
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pinf_$tag
env "$@" rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pinf_$tag -o run -- python $GRAFT_REPO_ROOT/bench.py --mode infer --steps 5 > /tmp/pinf_$tag.log 2>&1
cd $GRAFT_REPO_ROOT
tail -2 /tmp/pinf_$tag.log | cut -c1-600 > gpurun_out/infer_${tag}_line.txt
F=$(find /tmp/pinf_$tag -name "*kernel_stats.csv" | head -1)
python - "$F" > gpurun_out/infer_${tag}_stats.txt <<'PY'
import csv, sys, re
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
calls = sum(int(r["Calls"]) for r in rows)
print(f"total kernel time {tot / 1e6:.1f} ms in {calls} launches (warm-up pipeline + 1 sampling run + 1 pipeline run: 150 U-Net forwards of batch 2)")
print(f"{'%':>6} {'ms':>9} {'calls':>7} {'avg us':>8}  kernel")
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:45]:
    name = re.sub(r"\(anonymous namespace\)::", "", r["Name"])[:110]
    print(f"{100 * float(r['TotalDurationNs']) / tot:6.2f} {float(r['TotalDurationNs']) / 1e6:9.2f} {int(r['Calls']):7d} {float(r['AverageNs']) / 1e3:8.1f}  {name}")
PY
