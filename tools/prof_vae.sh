#!/bin/bash
# rocprofv3 kernel trace of the frozen VAE encode / decode (bench.py --mode vae): per-kernel totals of the whole run
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_vae
rocprofv3 --kernel-trace --output-format rocpd -d /tmp/prof_vae -o run -- python $GRAFT_REPO_ROOT/bench.py --mode vae --steps 6 > /tmp/prof_vae.log 2>&1
cd $GRAFT_REPO_ROOT
DB=$(find /tmp/prof_vae -name "*.db" | head -1)
python - "$DB" > gpurun_out/prof_vae.txt <<'PY'
import re, sqlite3, sys
c = sqlite3.connect(sys.argv[1])
rows = c.execute("select name, start, end, grid_x, grid_y, grid_z from kernels order by start").fetchall()
agg = {}
for n, s, e, gx, gy, gz in rows:
    n = re.sub(r"\(anonymous namespace\)::|aqlgemm::|void ", "", n); n = re.sub(r"\(.*", "", n)[:80]
    a = agg.setdefault((n, gx // 256 if gx else 0, gy, gz), [0, 0.0]); a[0] += 1; a[1] += (e - s) / 1e3
tot = sum(v[1] for v in agg.values())
print("total kernel ms", tot / 1e3)
for (n, gx, gy, gz), (cnt, us) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:45]:
    print(f"{100*us/tot:5.2f}% {us/1e3:8.3f} ms {cnt:5d}x {us/cnt:8.1f} us  blocks=({gx},{gy},{gz})  {n}")
PY
tail -1 /tmp/prof_vae.log | cut -c1-400
