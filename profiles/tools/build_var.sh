#!/bin/bash
# build_var.sh <name> <file.hip> [extra hipcc flags...]: a variant of ONE kernel file linked with
# the library's other objects -> pydrobert-pytorch_amd/csrc/build/variants/<name>/lib.so (git-ignored,
# travels to the GPU box).  Run it with PDT_AMD_LIB=<that path>.  Diagnostic builds:
#   -DPDT_STATS (event counters, profiles/tools/stats_ctc.py), -DPDT_STAMPS (per-phase cycles,
#   profiles/tools/stamps_ctc.py).
set -e
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
NAME=$1; SRC=$2; shift 2
CS=$ROOT/pydrobert-pytorch_amd/csrc
D=$CS/build/variants/$NAME
mkdir -p "$D"
cd "$CS"
make -s -j8 > /dev/null
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -ffp-contract=off"
BASE=$(basename "$SRC" .hip)
[ "$BASE" = ctc_search ] && FLAGS="$FLAGS -fno-slp-vectorize"  # (as the Makefile builds it)
/opt/rocm/bin/hipcc $FLAGS "$@" -c "$SRC" -o "$D/$BASE.o"
OBJS=$(ls build/*.o | grep -v "build/$BASE.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$D/lib.so" "$D/$BASE.o" $OBJS
echo "$D/lib.so"
