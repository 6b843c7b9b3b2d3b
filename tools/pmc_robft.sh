#!/bin/bash
# HBM-side traffic of BASELINE config 5's training step (bench.py --mode robft: distortion + SecretDecoder train step, batch 16, fp32) as
# MI355X_MICROARCH.md prescribes: FETCH_SIZE and WRITE_SIZE in SEPARATE rocprofv3 --pmc passes, kernel trace only.  Every dispatch of the
# process is summed and divided by the iterations run (1 warm-up + 3 timed): set-up kernels (weights, 50 MB of synthetic images) are < 1 %.
# usage (GPU box): tools/pmc_robft.sh   -> gpurun_out/pmc_robft.json
cd /tmp && export TMPDIR=/tmp
for C in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmcr_$C
  timeout 600 rocprofv3 --kernel-trace --pmc $C --output-format csv -d /tmp/pmcr_$C -o run -- python $GRAFT_REPO_ROOT/bench.py --mode robft --steps 3 --warmup 1 > /tmp/pmcr_$C.log 2>&1
  echo "$C rc=$?"
done
python - > $GRAFT_REPO_ROOT/gpurun_out/pmc_robft.json <<'PY'
import csv, glob, json, collections, re
out = {"_how": "tools/pmc_robft.sh on MI355X: rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE (separate passes) over `bench.py --mode robft "
               "--steps 3 --warmup 1`, all dispatches summed / 4 iterations; counters in KiB; traffic = (2 x FETCH_SIZE + WRITE_SIZE) x 1024 bytes "
               "(the 2x read correction of MI355X_MICROARCH.md, calibrated in profiles/r05_fetch_calib.txt; Infinity-Cache hits are counted: an "
               "upper bound on HBM bytes)", "iterations": 4}
per = {}
for C in ("FETCH_SIZE", "WRITE_SIZE"):
    files = glob.glob(f"/tmp/pmcr_{C}/**/*counter_collection.csv", recursive=True)
    tot, byk = 0.0, collections.Counter()
    for row in csv.DictReader(open(files[0])):
        if row["Counter_Name"] != C:
            continue
        v = float(row["Counter_Value"])
        tot += v
        byk[re.sub(r"\(.*", "", re.sub(r"^void |\(anonymous namespace\)::", "", row["Kernel_Name"]))[:60]] += v
    out[C + "_KiB_per_step"] = tot / 4
    per[C] = byk
out["traffic_bytes_per_step"] = int((2 * out["FETCH_SIZE_KiB_per_step"] + out["WRITE_SIZE_KiB_per_step"]) * 1024)
top = collections.Counter()
for k in set(per["FETCH_SIZE"]) | set(per["WRITE_SIZE"]):
    top[k] = (2 * per["FETCH_SIZE"].get(k, 0) + per["WRITE_SIZE"].get(k, 0)) * 1024 / 4
out["top_kernels_bytes_per_step"] = {k: int(v) for k, v in top.most_common(12)}
print(json.dumps(out, indent=1))
PY
cat $GRAFT_REPO_ROOT/gpurun_out/pmc_robft.json | head -30
