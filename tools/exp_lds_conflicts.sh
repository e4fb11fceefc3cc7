#!/bin/bash
# Which LDS access of the lean kernel pays the bank conflicts: SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE per launch for the full
# kernel and with one access class taken out at a time (diagnostic one-kernel builds; wrong audio in the ablated ones, counters
# only).  Output: gpurun_out/r3/lds_conflicts.txt.  The product build is restored at the end.
cd "${GRAFT_REPO_ROOT:-$(pwd)}"
trap 'python3 ohpipeline_amd/build.py --force > /dev/null 2>&1' EXIT     # an interrupted experiment must not leave a diagnostic library behind
R=$(pwd)
OUT=$R/gpurun_out/r3/lds_conflicts.txt
: > "$OUT"
for spec in "full|" "no_sample_reads|-DOHGPU_DIAG_NO_X" "no_ring_stores|-DOHGPU_DIAG_NO_RING" "no_drain|-DOHGPU_DIAG_NO_DRAIN" "no_staging_dma|-DOHGPU_DIAG_NO_DMA"; do
  IFS='|' read -r label flags <<< "$spec"
  OHGPU_EXTRA_FLAGS="-DOHGPU_DIAG -DOHGPU_DIAG_ONE_KERNEL $flags" python3 ohpipeline_amd/build.py --force > /dev/null 2>&1 || { echo "$label: build failed" >> "$OUT"; continue; }
  D=$R/gpurun_out/r3/pmc_lds_$label
  rm -rf "$D"
  (cd /tmp && TMPDIR=/tmp timeout -k 10 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS --output-format csv -d "$D" -- python3 "$R/bench.py" --steps 10 --warmup 3 --no-cpu --sustain 0.2 > /dev/null 2>&1)
  python3 - "$D" "$label" >> "$OUT" <<'PY'
import csv, glob, sys
from collections import defaultdict
acc = defaultdict(list)
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "src_lean_kernel" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
print(sys.argv[2], {k: round(sum(v) / len(v) / 1e6, 2) for k, v in sorted(acc.items())}, "(M per launch)")
PY
done
python3 ohpipeline_amd/build.py --force > /dev/null 2>&1
cat "$OUT"
