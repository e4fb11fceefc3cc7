#!/bin/bash
# Builds a variant of libohgpu.so HERE (no GPU needed): the workgroup matrix kernel (and what else is named in VARIANT_SOURCES)
# recompiled with the given flags, every other object taken from the tree's build.  The result goes to
# ohpipeline_amd/lib/variants/libohgpu.<tag>.so and travels to the GPU box with the snapshot; tools/exp_prebuilt.sh times them there,
# turn and turn about, without spending box time on hipcc.
# Usage: bash tools/build_variant.sh <tag> [flags...]       e.g.  bash tools/build_variant.sh early -DMF_WG_EARLY_LOADS
set -e
cd "$(dirname "$0")/.."
TAG=$1; shift
bash tools/check_diag_flags.sh "$@"
OBJ=ohpipeline_amd/build/obj
python3 ohpipeline_amd/build.py > /dev/null                      # (the tree's objects are current)
CC="/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wno-inline-asm -I include"
SRCS=${VARIANT_SOURCES:-src_mfma_wg_kernel.hip}
mkdir -p /tmp/variant_$TAG ohpipeline_amd/lib/variants
excl=""
extra=""
for f in $SRCS; do
  $CC "$@" -x hip -c ohpipeline_amd/csrc/$f -o /tmp/variant_$TAG/$f.o
  excl="$excl -e /$f\\."
  extra="$extra /tmp/variant_$TAG/$f.o"
done
[ -f $OBJ/linked.txt ] || python3 ohpipeline_amd/build.py --force > /dev/null      # (the list of objects the tree's library is linked from)
objs=$(grep -v $excl $OBJ/linked.txt)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ohpipeline_amd/lib/variants/libohgpu.$TAG.so $objs $extra
ls -la ohpipeline_amd/lib/variants/libohgpu.$TAG.so
