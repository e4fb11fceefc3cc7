#!/bin/bash
# Compiles csrc/src_block_kernel.hip to assembly and prints register/spill/instruction statistics of the
# S24LE->S24BE stereo instantiation (the headline kernel).  Usage: bash tools/inspect_kernel.sh
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
B=$R/ohpipeline_amd/build
mkdir -p "$B"
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -I $R/include"
/opt/rocm/bin/hipcc $FLAGS -S --cuda-device-only "$R/ohpipeline_amd/csrc/src_block_kernel.hip" -o "$B/sbk.s" 2>&1 | grep -E "error|warning: loop" || true
awk '/^_ZN5ohgpu16src_block_kernelILi32ELi2ELi3ELb1ELi3ELb0ELb0EE/{f=1} f{print} /s_endpgm/{if(f)exit}' "$B/sbk.s" > "$B/k.s"
echo "lines $(wc -l < "$B/k.s")"
awk '/^_ZN5ohgpu16src_block_kernelILi32ELi2ELi3ELb1ELi3ELb0ELb0EE/{f=1} f&&/\.(sgpr_count|vgpr_count|sgpr_spill_count|vgpr_spill_count|private_segment_fixed_size|group_segment_fixed_size):/{print}' "$B/sbk.s" | head -0
grep -A40 "\.name:.*src_block_kernelILi32ELi2ELi3ELb1ELi3ELb0ELb0EE" "$B/sbk.s" | grep -E "sgpr_count|vgpr_count|spill_count|private_segment_fixed" | head -6
for pat in v_fmac_f64 v_fmac_f64_dpp v_readlane v_writelane ds_read ds_write scratch_ flat_load s_barrier global_load_lds global_store; do
  echo "$pat $(grep -c "$pat" "$B/k.s")"
done
