#!/bin/bash
# Profiles bench.py on the GPU box: one kernel-trace/stats run plus separate --pmc passes (never combined with
# tracing).  Usage (inside gpurun):  bash tools/profile_bench.sh <tag> [bench args...]
# Results land in gpurun_out/prof_<tag>/; tools/summarize_prof.py turns them into profiles/<tag>_*.{md,json}.
set -u
TAG=${1:-run}; shift || true
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
TARGET=${OHGPU_PROFILE_TARGET:-$R/bench.py}                      # e.g. tools/bench_pcm.py (then pass its own arguments)
if [ "$TARGET" = "$R/bench.py" ]; then BENCH_ARGS="--steps 10 --warmup 3 --no-cpu $*"; PMC_ARGS="--sustain 0.1"; else BENCH_ARGS="$*"; PMC_ARGS=""; fi
# (the kernel-trace pass runs behind the full sustain phase: its average is the steady-state figure; the counter passes count
# events per launch, which do not depend on the chip's temperature, and a second of launches each would be 50 MB of rows)
echo "== kernel trace" | tee "$OUT/log.txt"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 "$TARGET" $BENCH_ARGS >> "$OUT/log.txt" 2>&1 || echo "trace run failed" | tee -a "$OUT/log.txt"
if [ -n "${OHGPU_PROFILE_TRACE_ONLY:-}" ]; then echo done; exit 0; fi    # (kernel times only: the line kernels' steady-state summaries)
i=0
while IFS= read -r SET; do
  [ -z "$SET" ] && continue
  i=$((i+1))
  echo "== pmc$i: $SET" | tee -a "$OUT/log.txt"
  timeout -k 10 300 rocprofv3 --pmc $SET --output-format csv -d "$OUT/pmc$i" -- python3 "$TARGET" $BENCH_ARGS $PMC_ARGS >> "$OUT/log.txt" 2>&1 || echo "pmc$i failed" | tee -a "$OUT/log.txt"
done <<'SETS'
SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA
SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_VALU_FMA_F64 SQ_IFETCH SQ_INST_LEVEL_SMEM
SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES
SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INST_CYCLES_SMEM SQ_ACTIVE_INST_VMEM SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS
FETCH_SIZE
WRITE_SIZE
GRBM_GUI_ACTIVE GRBM_COUNT
SETS
find "$OUT" -name "*.csv" | head -40 >> "$OUT/log.txt"
echo done
