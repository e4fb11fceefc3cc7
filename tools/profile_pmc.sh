#!/bin/bash
# PMC passes (counters only, one set per run) for bench.py's headline launch, summarised per launch for one kernel.
# Usage: bash tools/profile_pmc.sh <tag> <kernel name substring> [bench args]
TAG=${1:-pmc}; shift || true
KERNEL=${1:-src_mfma_kernel}; shift || true
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
BENCH_ARGS="--steps 5 --warmup 2 --no-cpu --no-extra-configs --sustain 0.05 $*"
i=0
while IFS= read -r SET; do
  [ -z "$SET" ] && continue
  i=$((i+1))
  if [ -n "$PMC_ONLY" ] && ! echo " $PMC_ONLY " | grep -q " $i "; then continue; fi
  echo "== pmc$i: $SET" >> "$OUT/log.txt"
  timeout -k 10 300 rocprofv3 --pmc $SET --output-format csv -d "$OUT/pmc$i" -- python3 "$R/bench.py" $BENCH_ARGS >> "$OUT/log.txt" 2>&1 || echo "pmc$i failed" | tee -a "$OUT/log.txt"
done <<'SETS'
SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_VMEM_RD SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES
TCP_TOTAL_ACCESSES TCP_TCC_READ_REQ TCP_TCC_WRITE_REQ TCP_PENDING_STALL_CYCLES
TCP_TOTAL_READ TCP_TOTAL_WRITE TCP_TCP_TA_DATA_STALL_CYCLES TCP_TA_TCP_STATE_READ
TCC_REQ TCC_HIT TCC_MISS TCC_TAG_STALL
TCC_EA0_WRREQ TCC_EA0_WRREQ_64B TCC_EA0_WRREQ_STALL TCC_EA0_RDREQ
TCC_EA0_RDREQ_32B TCC_READ TCC_WRITE TCC_WRITEBACK
GRBM_TA_BUSY GRBM_GUI_ACTIVE TA_TA_BUSY TA_BUSY_CYCLES
TA_ADDR_STALLED_BY_TC_CYCLES TA_DATA_STALLED_BY_TC_CYCLES TA_ADDR_STALLED_BY_TD_CYCLES TA_FLAT_READ_WAVEFRONTS TA_FLAT_WRITE_WAVEFRONTS
SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES
SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_BRANCH SQ_INSTS_VMEM SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES
SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_LDS_IDX_ACTIVE SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_BUSY_CYCLES
SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_FLAT SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_WAIT_INST_VMEM SQ_INSTS_VALU_MFMA_I8 SQ_WAVES
SETS
python3 - "$OUT" "$KERNEL" <<'PY'
import csv, glob, sys, collections
out, kern = sys.argv[1], sys.argv[2]
tot = collections.defaultdict(float); cnt = collections.defaultdict(int)
for f in glob.glob(out + "/pmc*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if kern not in row["Kernel_Name"]: continue
        tot[row["Counter_Name"]] += float(row["Counter_Value"]); cnt[row["Counter_Name"]] += 1
with open(out + "/summary.txt", "w") as o:
    for k in sorted(tot):
        line = "%-36s per launch %.6g  (%d launches)" % (k, tot[k] / cnt[k], cnt[k])
        print(line); o.write(line + "\n")
PY
