#!/bin/bash
cd $GRAFT_REPO_ROOT
echo "== 16x16x32"; AQL_ATTN32=0 python tools/time_attn.py 2>&1 | grep "B="
echo "== 32x32x16 full"; python tools/time_attn.py 2>&1 | grep "B="
for a in "$@"; do echo "== ablation $a"; AQL_LIB=altlib/a32abl$a.so python tools/time_attn.py 2>&1 | grep "B=4 N=4096"; done
