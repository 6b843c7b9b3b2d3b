cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r06
ROUNDS=2 tools/ab_bench.sh gpurun_out/r06/bound_gn_stats.txt "AQL_LIB=altlib/gnexp.so AQL_EXP_GN_SKIPSTATS=0" "AQL_LIB=altlib/gnexp.so AQL_EXP_GN_SKIPSTATS=1" "AQL_LIB=altlib/gnexp.so AQL_EXP_GN_SKIPSTATS=2"
