#!/bin/bash
# End-of-round refresh of the judged artefacts: default bench line, kernel traces + families of config 2 and config 3.
set -x
python bench.py > gpurun_out/r03_bench_line.json 2> gpurun_out/r03_bench_line.err
tools/insitu_profile.sh r03final
BENCH_EXTRA="--config 3" FAM_BATCH=8 FAM_RANK=320 tools/insitu_profile.sh r03final_c3
python -m pytest tests -m gpu -x -q 2>&1 | tail -3 > gpurun_out/r03_gpu_tests.txt
