"""Summarise a rocprofv3 rocpd sqlite db: per-kernel totals over the LAST n steps (delimited by adamw launches)."""
import re, sqlite3, sys
db, nsteps = sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 3
c = sqlite3.connect(db)
rows = c.execute("select name, start, end from kernels order by start").fetchall()
marks = [i for i, r in enumerate(rows) if "adamw_kernel" in r[0]]
# two adamw launches per step (LoRA group + mapper group)
marks = marks[1::2]
lo = marks[-nsteps - 1] + 1 if len(marks) > nsteps else 0
sel = rows[lo:marks[-1] + 1]
wall = (sel[-1][2] - sel[0][1]) / 1e6 / nsteps
agg = {}
for n, s, e in sel:
    n = re.sub(r"\(anonymous namespace\)::|aqlgemm::|void ", "", n)
    n = re.sub(r"\(.*", "", n)[:90]
    a = agg.setdefault(n, [0, 0.0]); a[0] += 1; a[1] += (e - s) / 1e3
tot = sum(v[1] for v in agg.values())
print(f"steps analysed: {nsteps}; GPU span per step {wall:.2f} ms; kernel-busy per step {tot / 1e3 / nsteps:.2f} ms; launches per step {sum(v[0] for v in agg.values()) / nsteps:.0f}")
print(f"{'%':>6} {'ms/step':>8} {'calls/step':>10} {'avg us':>8}  kernel")
for n, (cnt, us) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:int(sys.argv[3]) if len(sys.argv) > 3 else 30]:
    print(f"{100 * us / tot:6.2f} {us / 1e3 / nsteps:8.3f} {cnt / nsteps:10.1f} {us / cnt:8.1f}  {n}")

# launches AFTER the last step: bench.py's isolated timing of the dominant kernels (roofline.achieved comes from these)
tail = rows[marks[-1] + 1:]
tagg = {}
for n, s_, e in tail:
    n = re.sub(r"\(anonymous namespace\)::|aqlgemm::|void ", "", n)
    n = re.sub(r"\(.*", "", n)[:90]
    a = tagg.setdefault(n, [0, 0.0]); a[0] += 1; a[1] += (e - s_) / 1e3
if tagg:
    print("\nisolated launches after the last step (bench.py dominant_kernels):")
    for n, (cnt, us) in sorted(tagg.items(), key=lambda kv: -kv[1][1])[:6]:
        print(f"   {cnt:5d} launches  avg {us / cnt:8.1f} us  {n}")
