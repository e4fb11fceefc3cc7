#!/bin/bash
# Memory-system PMC passes for bench.py (separate from tracing).  Usage: bash tools/profile_mem.sh <tag> [bench args]
TAG=${1:-mem}; shift || true
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
TARGET=${OHGPU_PROFILE_TARGET:-$R/bench.py}                      # e.g. tools/bench_pcm.py (then pass its own arguments)
if [ "$TARGET" = "$R/bench.py" ]; then BENCH_ARGS="--steps 5 --warmup 2 --no-cpu $*"; else BENCH_ARGS="$*"; fi
i=0
while IFS= read -r SET; do
  [ -z "$SET" ] && continue
  i=$((i+1))
  echo "== pmc$i: $SET" | tee -a "$OUT/log.txt"
  timeout -k 10 300 rocprofv3 --pmc $SET --output-format csv -d "$OUT/pmc$i" -- python3 "$TARGET" $BENCH_ARGS >> "$OUT/log.txt" 2>&1 || echo "pmc$i failed" | tee -a "$OUT/log.txt"
done <<'SETS'
SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_VMEM_RD SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES
TCP_TOTAL_ACCESSES TCP_TCC_READ_REQ TCP_TCC_WRITE_REQ TCP_PENDING_STALL_CYCLES
TCP_TOTAL_READ TCP_TOTAL_WRITE TCP_TCP_TA_DATA_STALL_CYCLES TCP_TA_TCP_STATE_READ
TCC_REQ TCC_HIT TCC_MISS TCC_TAG_STALL
TCC_EA0_WRREQ TCC_EA0_WRREQ_64B TCC_EA0_WRREQ_STALL TCC_EA0_RDREQ
TCC_EA0_RDREQ_32B TCC_READ TCC_WRITE TCC_WRITEBACK
GRBM_TA_BUSY GRBM_GUI_ACTIVE
SETS
echo done
