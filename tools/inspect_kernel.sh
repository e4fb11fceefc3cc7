#!/bin/bash
# Compiles csrc/src_block_kernel.hip to assembly, prints register/spill/instruction statistics of the
# S24LE->S24BE stereo instantiation (the headline kernel) and checks the invariant its hand-counted LDS waits
# rely on: no scalar memory instruction between the first and the last tap of the main loop (scalar loads share
# lgkmcnt with LDS operations and complete out of order).  Usage: bash tools/inspect_kernel.sh
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
B=$R/ohpipeline_amd/build
mkdir -p "$B"
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -I $R/include"
/opt/rocm/bin/hipcc $FLAGS -S --cuda-device-only "$R/ohpipeline_amd/csrc/src_block_kernel.hip" -o "$B/sbk.s" 2>&1 | grep -E "error|warning: loop" || true
K=_ZN5ohgpu16src_block_kernelILi32ELi2ELi3ELb1ELi3ELb0ELb0EE
awk -v k="$K" 'index($0, k) == 1 {f=1} f{print} /s_endpgm/{if(f)exit}' "$B/sbk.s" > "$B/k.s"
echo "lines $(wc -l < "$B/k.s")"
grep -A40 "\.name:.*src_block_kernelILi32ELi2ELi3ELb1ELi3ELb0ELb0EE" "$B/sbk.s" | grep -E "sgpr_count|vgpr_count|spill_count|private_segment_fixed|group_segment_fixed" | head -6
for pat in v_fmac_f64_dpp v_readlane v_writelane "ds_read" "ds_write" scratch_ flat_load s_barrier global_load_lds global_store "s_nop" "lgkmcnt(0)" "lgkmcnt(1)" "lgkmcnt(2)"; do
  echo "$pat $(grep -c "$pat" "$B/k.s" || true)"
done
first=$(grep -n "v_fmac_f64_dpp" "$B/k.s" | head -1 | cut -d: -f1)
last=$(grep -n "v_fmac_f64_dpp" "$B/k.s" | tail -1 | cut -d: -f1)
bad=$(awk -v a="$first" -v b="$last" 'NR>=a && NR<=b && /^\s*(s_load|s_buffer_load|s_memtime|s_memrealtime|s_dcache|s_scratch_load|s_store|s_atomic)/' "$B/k.s" | wc -l)
echo "scalar memory instructions inside the main loop (lines $first..$last): $bad"
[ "$bad" = 0 ] || { echo "FAIL: counted lgkmcnt waits are unsafe with scalar memory traffic in the loop"; exit 1; }
