#!/bin/bash
# Wave-stall attribution PMC passes for bench.py (instruction cache, scalar cache, LDS, issue).  Counters only, no tracing.
# Usage: bash tools/profile_stall.sh <tag> [bench args]
TAG=${1:-stall}; shift || true
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
SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH SQ_WAVE_CYCLES SQ_BUSY_CYCLES
SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES SQC_DCACHE_MISSES_DUPLICATE SQC_TC_INST_REQ SQC_TC_DATA_READ_REQ SQC_TC_STALL
SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES
SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_BRANCH SQ_INSTS_VMEM SQ_INST_LEVEL_SMEM SQ_INST_LEVEL_LDS
SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_LDS_IDX_ACTIVE SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_IFETCH_LEVEL
SETS
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
tot = collections.defaultdict(float); cnt = collections.defaultdict(int)
for f in glob.glob(out + "/pmc*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if not any(k in row["Kernel_Name"] for k in ("src_lean_kernel", "src_block_kernel", "pcm_line_kernel", "pcm_msg_kernel")): continue
        tot[row["Counter_Name"]] += float(row["Counter_Value"]); cnt[row["Counter_Name"]] += 1
with open(out + "/summary.txt", "w") as o:
    for k in sorted(tot):
        line = "%-32s per launch %.6g  (%d launches)" % (k, tot[k] / cnt[k], cnt[k])
        print(line); o.write(line + "\n")
PY
