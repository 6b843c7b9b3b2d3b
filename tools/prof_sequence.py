"""One captured step of a rocprofv3 rocpd db as a launch sequence: index, start offset, duration, gap to the previous kernel,
grid, kernel name -- for reading in-step durations against the isolated timings of the same shapes."""
import re, sqlite3, sys
db = sys.argv[1]
c = sqlite3.connect(db)
rows = c.execute("select name, start, end, grid_x, grid_y, grid_z from kernels order by start").fetchall()
marks = [i for i, r in enumerate(rows) if "adamw_kernel" in r[0]][1::2]
sel = rows[marks[-2] + 1:marks[-1] + 1]
t0, prev = sel[0][1], sel[0][1]
for i, (n, s, e, gx, gy, gz) in enumerate(sel):
    n = re.sub(r"\(anonymous namespace\)::|aqlgemm::|void ", "", n); n = re.sub(r"\(.*", "", n)[:80]
    print(f"{i:4d} t={(s - t0) / 1e3:9.1f} dur={(e - s) / 1e3:7.1f} gap={(s - prev) / 1e3:6.1f} grid=({gx // 256 if gx else 0},{gy},{gz}) {n}")
    prev = e
