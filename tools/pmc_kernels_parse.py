"""Per-kernel averages of a rocprofv3 --pmc csv (tools/pmc_kernels.sh): raw SQ counters per launch and two ratios inside
the SQ counter domain -- SQ_VALU_MFMA_BUSY_CYCLES / SQ_BUSY_CYCLES (how much of the time the SQs were busy the matrix pipe
was busy) and the split of SQ_WAVE_CYCLES into wait-any / wait-inst / active-inst.  The absolute scale of
SQ_VALU_MFMA_BUSY_CYCLES is NOT calibrated on gfx950 (it comes out at 8-40 "cycles" per 16x16x32 MFMA depending on how many
SIMDs of a CU issue concurrently), so no utilisation against the 2.5 PFLOP/s peak is derived from it; the time-based
algorithmic TFLOP/s of bench.py / tools/tune_*.py are the figures to compare with the peak."""
import collections
import csv
import glob
import re
import sys

files = glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for row in csv.DictReader(open(files[0])):
    k = re.sub(r"\(anonymous namespace\)::|aqlgemm::|void ", "", row["Kernel_Name"])
    k = re.sub(r"\(.*", "", k)[:70]
    agg[(k, row["Grid_Size"])][row["Counter_Name"]].append(float(row["Counter_Value"]))
if not any("SQ_BUSY_CYCLES" in cs for cs in agg.values()):   # an HBM-traffic pass: just list the counters per kernel
    for (k, grid), cs in sorted(agg.items(), key=lambda kv: -max(sum(v) for v in kv[1].values())):
        if "at::native" in k or "elementwise" in k or "rocclr" in k:
            continue
        print(f"{k:70s} {grid:>9s} " + "  ".join(f"{c}={sum(v) / len(v):12.1f} (n={len(v)})" for c, v in cs.items()))
    sys.exit(0)
print(f"{'kernel':70s} {'grid':>9s} {'n':>4s} {'SQ_BUSY':>10s} {'MFMA_BUSY':>10s} {'mfma/busy':>9s} {'WAVE_CYC':>10s} {'wait_any':>8s} {'wait_inst':>9s} {'active':>7s} {'VALU insts':>10s}")
for (k, grid), cs in sorted(agg.items(), key=lambda kv: -sum(kv[1].get("SQ_BUSY_CYCLES", [0]))):
    if "at::native" in k or "elementwise" in k or "rocclr" in k:
        continue
    m = lambda c: (sum(cs[c]) / len(cs[c])) if c in cs and cs[c] else 0.0  # noqa: E731
    busy, mf, wc = m("SQ_BUSY_CYCLES"), m("SQ_VALU_MFMA_BUSY_CYCLES"), m("SQ_WAVE_CYCLES")
    print(f"{k:70s} {grid:>9s} {len(cs.get('SQ_BUSY_CYCLES', [])):4d} {busy:10.3g} {mf:10.3g} {(mf / busy if busy else 0):9.2f} {wc:10.3g} "
          f"{(m('SQ_WAIT_ANY') / wc if wc else 0):8.2f} {(m('SQ_WAIT_INST_ANY') / wc if wc else 0):9.2f} "
          f"{(m('SQ_ACTIVE_INST_ANY') / wc if wc else 0):7.2f} {m('SQ_INSTS_VALU'):10.3g}")
