#!/bin/bash
# round-3 validation run A: full GPU test suite, default bench line, one-graph A/B, in-situ profile (families)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03a; mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -q -rA --durations=15 > $O/pytest.txt 2>&1; echo "pytest rc=$?" >> $O/pytest.txt
tail -5 $O/pytest.txt
timeout 900 python bench.py > $O/bench.txt 2> $O/bench.err; echo "bench rc=$?"
ROUNDS=2 tools/ab_bench.sh $O/ab_one_graph.txt "AQL_ONE_GRAPH=0" "AQL_ONE_GRAPH=1"
tools/insitu_profile.sh r03a
cp gpurun_out/insitu_r03a_* $O/ 2>/dev/null
