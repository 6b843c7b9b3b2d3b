#!/bin/bash
# Kernel totals of the SecretDecoder's eval forward (bit extraction, utils_eval.py:131-140) at batch $1 (default 1), 20 forwards.
B=${1:-1}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pext
cat > /tmp/pext.py <<PY
import sys, torch
sys.path.insert(0, "$GRAFT_REPO_ROOT")
import bench
dec = bench.synthetic_decoder(48, "cuda").eval()
x = torch.rand($B, 3, 512, 512, device="cuda") * 2 - 1
for _ in range(21): dec(x)
torch.cuda.synchronize()
PY
AQL_DECODER_GRAPH=0 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pext -o run -- python /tmp/pext.py > /tmp/pext.log 2>&1
cd $GRAFT_REPO_ROOT
T=$(find /tmp/pext -name "*kernel_trace.csv" | head -1)
python - "$T" > gpurun_out/extract_b${B}_stats.txt <<'PY'
import csv, sys, re, collections
agg = collections.defaultdict(lambda: [0, 0.0])
for r in csv.DictReader(open(sys.argv[1])):
    n = re.sub(r"\(anonymous namespace\)::|void ", "", r["Kernel_Name"]); n = re.sub(r"\(.*", "", n)[:40]
    if "at::native" in n: continue
    k = (n, int(r["Grid_Size_X"]) // 256, int(r["Grid_Size_Y"]), int(r["Grid_Size_Z"]))
    agg[k][0] += 1; agg[k][1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
tot = sum(v[1] for v in agg.values())
print(f"kernel time per forward {tot / 21 / 1e3:.3f} ms")
for k, (c, us) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:30]:
    print(f"{us / 21:8.1f} us/fwd {c // 21:4d}x {us / c:8.1f} us  {k[0]} grid=({k[1]},{k[2]},{k[3]})")
PY
head -32 gpurun_out/extract_b${B}_stats.txt
