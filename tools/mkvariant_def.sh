#!/bin/bash
# usage: tools/mkvariant_def.sh <name> <source.hip> <object-it-replaces> [-DX=Y ...]
# builds scratch/variants/lib_<name>.so = in-tree objects with ONE translation unit recompiled under extra defines
# (A/B switches like OCT_REGTAB, OCT_REAL2_REGTAB; for tools/ab.sh)
name=$1; src=$2; obj=$3; shift 3
root=$(cd $(dirname $0)/.. && pwd); cs=$root/octproz_amd/csrc
mkdir -p $root/scratch/variants /tmp/var_$name
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-value -Wno-pass-failed "$@" -c $cs/$src -o /tmp/var_$name/v.o || exit 1
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC /tmp/var_$name/v.o $(ls $cs/build/*.o | grep -v "/$obj\$") -o $root/scratch/variants/lib_$name.so -pthread -ldl
