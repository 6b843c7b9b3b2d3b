#!/bin/bash
# FETCH_SIZE of known-bytes reads in the library's access patterns (tools/micro/fetch_calib.hip).  -> gpurun_out/r05_fetch_calib.txt
cd $GRAFT_REPO_ROOT
hipcc --offload-arch=gfx950 -O3 tools/micro/fetch_calib.hip -o /tmp/fetch_calib || exit 1
out=gpurun_out/r05_fetch_calib.txt
: > $out
cd /tmp && export TMPDIR=/tmp
for m in 0 1 2 3 4 5; do
  rm -rf /tmp/fc_$m
  timeout 120 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/fc_$m -o run -- /tmp/fetch_calib $m > /tmp/fc_$m.log 2>&1
  grep "^mode" /tmp/fc_$m.log >> $GRAFT_REPO_ROOT/$out
  F=$(find /tmp/fc_$m -name "*counter_collection.csv" | head -1)
  python - "$F" >> $GRAFT_REPO_ROOT/$out <<'PY'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "reader" in r.get("Kernel_Name", "")]
per = {}
for r in rows:
    if r["Counter_Name"] == "FETCH_SIZE":
        per[r["Dispatch_Id"]] = per.get(r["Dispatch_Id"], 0.0) + float(r["Counter_Value"])
tot = sum(per.values())
print(f"        FETCH_SIZE = {tot:.0f} KiB = {tot / 1024:.1f} MiB as counted" + (f"   per launch: {[round(v / 1024, 1) for v in per.values()]} MiB" if len(per) > 1 else ""))
PY
done
cat $GRAFT_REPO_ROOT/$out
