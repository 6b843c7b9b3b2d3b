#!/bin/bash
# round-3 run B: PMC traffic of the time-dominant kernel (LoRA + GEGLU launch), SQ counters of it, config-3 in-situ profile
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03b; mkdir -p $O
bash tools/pmc_traffic.sh geglu geglu 32768 1280 320; cp gpurun_out/pmct_geglu.txt $O/
PMC_ONLY="1 2" bash tools/pmc_run.sh geglu_sq geglu 32768 1280 320; cp gpurun_out/pmc_geglu_sq.txt $O/
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/prof_c3
rocprofv3 --kernel-trace --output-format rocpd -d /tmp/prof_c3 -o run -- python $GRAFT_REPO_ROOT/bench.py --config 3 --steps 4 --warmup 2 --no-cpu-baseline --no-extras > /tmp/prof_c3.log 2>&1
cd $GRAFT_REPO_ROOT
DB=$(find /tmp/prof_c3 -name "*.db" | head -1)
python tools/prof_summary.py $DB 3 40 > $O/insitu_config3_summary.txt
python tools/prof_shapes.py $DB 3 80 > $O/insitu_config3_shapes.txt
python tools/prof_families.py $DB 3 8 320 > $O/families_config3.json
head -2 $O/insitu_config3_summary.txt
