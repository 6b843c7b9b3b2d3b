#!/bin/bash
# One U-Net forward of BASELINE config 4's sampling loop (CFG batch 2) as a launch sequence: start, duration, gap, grid, kernel.
# usage (GPU box): tools/prof_infer_seq.sh <tag> [ENV=VAL ...]  -> gpurun_out/infer_<tag>_sequence.txt
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pseq_$tag
env "$@" rocprofv3 --kernel-trace --output-format rocpd -d /tmp/pseq_$tag -o run -- python $GRAFT_REPO_ROOT/bench.py --mode infer --steps 5 > /tmp/pseq_$tag.log 2>&1
cd $GRAFT_REPO_ROOT
DB=$(find /tmp/pseq_$tag -name "*.db" | head -1)
python - "$DB" > gpurun_out/infer_${tag}_sequence.txt <<'PY'
import re, sqlite3, sys
c = sqlite3.connect(sys.argv[1])
rows = c.execute("select name, start, end, grid_x, grid_y, grid_z from kernels order by start").fetchall()
marks = [i for i, r in enumerate(rows) if "ddim_step_kernel" in r[0]]
a, b = marks[-12], marks[-11]
sel = rows[a + 1:b + 1]
t0, prev = sel[0][1], sel[0][1]
busy = sum(e - s for _, s, e, *_ in sel)
print(f"one forward of CFG batch 2 + the DDIM update: {len(sel)} launches, wall {(sel[-1][2] - t0) / 1e3:.1f} us, kernel time {busy / 1e3:.1f} us")
for i, (n, s, e, gx, gy, gz) in enumerate(sel):
    n = re.sub(r"\(anonymous namespace\)::|aqlgemm::|void ", "", n); n = re.sub(r"\(.*", "", n)[:90]
    print(f"{i:4d} t={(s - t0) / 1e3:9.1f} dur={(e - s) / 1e3:7.1f} gap={(s - prev) / 1e3:6.1f} grid=({gx // 256 if gx else 0},{gy},{gz}) {n}")
    prev = e
PY
head -1 gpurun_out/infer_${tag}_sequence.txt
tail -2 /tmp/pseq_$tag.log | cut -c1-300
