cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_r320
rocprofv3 --kernel-trace --output-format rocpd -d /tmp/prof_r320 -o run -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 2 --rank 320 --batch 8 --no-cpu-baseline > /tmp/prof_r320.log 2>&1
cd $GRAFT_REPO_ROOT
DB=$(find /tmp/prof_r320 -name "*.db" | head -1)
python tools/prof_summary.py $DB 2 40 > gpurun_out/insitu_r320_summary.txt
python tools/prof_shapes.py $DB 2 60 > gpurun_out/insitu_r320_shapes.txt
tail -1 /tmp/prof_r320.log | cut -c1-200
