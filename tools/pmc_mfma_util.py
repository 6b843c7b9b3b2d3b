"""MfmaUtil per kernel from a rocprofv3 --pmc csv pass that carries SQ_VALU_MFMA_BUSY_CYCLES and GRBM_GUI_ACTIVE (tools/pmc_kernels.sh):
the gfx94x derived-counter formula  100 * SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE * 256 CUs * 4 SIMDs), i.e. the fraction of
SIMD-cycles in which the matrix pipe was busy (padding MFMAs count as busy: it is a pipe utilisation, not a useful-FLOP fraction).
GRBM_GUI_ACTIVE is reported summed over the 8 XCDs on gfx950 (checked against start/end timestamps x clock below)."""
import collections, csv, glob, re, sys
d = sys.argv[1]
cc = glob.glob(d + "/**/*counter_collection.csv", recursive=True)[0]
kt = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)
dur = {}
if kt:
    for row in csv.DictReader(open(kt[0])):
        dur[row["Dispatch_Id"]] = (int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e3
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for row in csv.DictReader(open(cc)):
    k = re.sub(r"\(anonymous namespace\)::|aqlgemm::|void ", "", row["Kernel_Name"]); k = re.sub(r"\(.*", "", k)[:60]
    key = (k, row["Grid_Size"])
    agg[key][row["Counter_Name"]].append(float(row["Counter_Value"]))
    if row["Dispatch_Id"] in dur:
        agg[key]["_us"].append(dur[row["Dispatch_Id"]])
print(f"{'kernel':60s} {'grid':>9s} {'n':>3s} {'us':>8s} {'GRBM_GUI_ACTIVE':>15s} {'GHz(/8 XCD)':>11s} {'MFMA_BUSY':>12s} {'MfmaUtil %':>10s}")
for (k, grid), cs in sorted(agg.items(), key=lambda kv: -sum(kv[1].get("SQ_VALU_MFMA_BUSY_CYCLES", [0]))):
    if "SQ_VALU_MFMA_BUSY_CYCLES" not in cs or "GRBM_GUI_ACTIVE" not in cs or "at::native" in k:
        continue
    m = lambda c: sum(cs[c]) / len(cs[c])  # noqa: E731
    gui, busy = m("GRBM_GUI_ACTIVE"), m("SQ_VALU_MFMA_BUSY_CYCLES")
    us = m("_us") if cs.get("_us") else float("nan")
    print(f"{k:60s} {grid:>9s} {len(cs['GRBM_GUI_ACTIVE']):3d} {us:8.1f} {gui:15.0f} {gui / 8 / us / 1e3:11.2f} {busy:12.0f} {100 * busy / (gui / 8 * 256 * 4):10.1f}")
