#!/bin/bash
# An alternative build of the library with extra defines on ONE source file (ablation / trace builds), selected at run time
# with AQL_LIB=altlib/<name>.so (aqualora_amd/_lib.py).   usage: tools/build_alt.sh <name> <file.hip> [-DX=1 ...]
set -e
cd "$(dirname "$0")/../aqualora_amd/csrc"
name=$1; src=$2; shift 2
mkdir -p ../../altlib
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -ffp-contract=fast -munsafe-fp-atomics "$@" -c $src -o ../../altlib/${name}_${src%.hip}.o
objs=$(ls aql_*.o | grep -v "^${src%.hip}.o$")
hipcc --offload-arch=gfx950 -shared -fPIC $objs ../../altlib/${name}_${src%.hip}.o -ldl -o ../../altlib/${name}.so
echo built altlib/${name}.so
