#!/bin/bash
# usage: [RS=2] [VFLAGS='-mllvm ...'] tools/mkvariant.sh <log2n> name...   (RS: the resampling mode whose object is replaced, fused_<log2n>_rs<RS>.o)
# usage: tools/mkvariant.sh <log2n> name...   builds /tmp/abl_<name>/{kernels.h,...} into scratch/variants/lib_<name>.so
# (the variant directory holds edited copies of the csrc headers + fused_inst.hip; all other objects come from csrc/build)
L=$1; shift
root=$(cd $(dirname $0)/.. && pwd)
mkdir -p $root/scratch/variants
for v in "$@"; do
 ( cd /tmp/abl_$v && rm -f v.o && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-value -Wno-pass-failed -DOCT_LOG2N=$L -DOCT_FUSED_RS=${RS:-2} -Wno-inline-asm $VFLAGS -c fused_inst.hip -o v.o 2>&1 | grep -E "error|static_assert"
   [ -f v.o ] && /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC v.o $(ls $root/octproz_amd/csrc/build/*.o | grep -v fused_${L}_rs${RS:-2}.o) -o $root/scratch/variants/lib_$v.so -pthread ) &
done
wait
