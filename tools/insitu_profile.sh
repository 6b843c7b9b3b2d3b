#!/bin/bash
# In-situ kernel durations of the captured train step under rocprofv3 (weights cold in L2, unlike the micro probes).
# usage: tools/insitu_profile.sh <tag> [ENV=VAL ...]   -> gpurun_out/insitu_<tag>_{summary,shapes}.txt
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_$tag
env "$@" rocprofv3 --kernel-trace --output-format rocpd -d /tmp/prof_$tag -o run -- python $GRAFT_REPO_ROOT/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-extras $BENCH_EXTRA > /tmp/prof_$tag.log 2>&1
cd $GRAFT_REPO_ROOT
DB=$(find /tmp/prof_$tag -name "*.db" | head -1)
python tools/prof_summary.py $DB 3 > gpurun_out/insitu_${tag}_summary.txt
python tools/prof_shapes.py $DB 3 400 > gpurun_out/insitu_${tag}_shapes.txt
python tools/prof_families.py $DB 3 ${FAM_BATCH:-4} ${FAM_RANK:-32} > gpurun_out/insitu_${tag}_families.json
head -1 gpurun_out/insitu_${tag}_summary.txt
python tools/prof_sequence.py $DB > gpurun_out/insitu_${tag}_sequence.txt
