#!/usr/bin/env bash
# A variant build of the library for same-box A/B timing: tools/build_variant.sh <name> <unit.hip> <-Dflags ...>  -> tools/_bin/libmulls_<name>.so
# (the named translation unit recompiled with the flags, every other object taken from the in-tree build; select it with MULLS_HIP_LIB)
set -eu
cd "$(dirname "$0")/.."; mkdir -p tools/_bin
N=$1; U=$2; shift 2
F="--offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -fopenmp -Wall -Wno-unused-function"
/opt/rocm/bin/hipcc $F "$@" -c mulls_amd/csrc/$U -o tools/_bin/$N.$U.o
OBJS=$(ls mulls_amd/csrc/*.o | grep -v "/$U.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -fopenmp -o tools/_bin/libmulls_$N.so $OBJS tools/_bin/$N.$U.o
echo tools/_bin/libmulls_$N.so
