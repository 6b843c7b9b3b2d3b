"""Per (kernel, grid) totals over the last n steps of a rocprofv3 rocpd db: spots pathological shapes."""
import re, sqlite3, sys
db, nsteps = sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 3
c = sqlite3.connect(db)
rows = c.execute("select name, start, end, grid_x, grid_y, grid_z from kernels order by start").fetchall()
marks = [i for i, r in enumerate(rows) if "adamw_kernel" in r[0]][1::2]
sel = rows[marks[-nsteps - 1] + 1:marks[-1] + 1]
agg = {}
for n, s, e, gx, gy, gz in sel:
    n = re.sub(r"\(anonymous namespace\)::|aqlgemm::|void ", "", n); n = re.sub(r"\(.*", "", n)[:70]
    a = agg.setdefault((n, gx // 256 if gx else 0, gy, gz), [0, 0.0]); a[0] += 1; a[1] += (e - s) / 1e3
tot = sum(v[1] for v in agg.values())
for (n, gx, gy, gz), (cnt, us) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:int(sys.argv[3]) if len(sys.argv) > 3 else 40]:
    print(f"{100*us/tot:5.2f}% {us/1e3/nsteps:7.3f} ms/step {cnt/nsteps:6.1f}x {us/cnt:8.1f} us  blocks=({gx},{gy},{gz})  {n}")
