#!/bin/bash
# Twelve waves with the rotated rings (the tree's kernel) against eleven waves (ring pad 4), alternating on one box; then the
# LDS counters of the tree's kernel.
cd "${GRAFT_REPO_ROOT:-$(pwd)}"
trap 'python3 ohpipeline_amd/build.py --force > /dev/null 2>&1' EXIT     # an interrupted experiment must not leave a diagnostic library behind
R=$(pwd)
for pad in 0 4 0 4 0 4; do
  OHGPU_EXTRA_FLAGS="-DOHGPU_DIAG -DOHGPU_DIAG_ONE_KERNEL -DOHGPU_LEAN_RING_PAD=$pad" python3 ohpipeline_amd/build.py --force > /dev/null 2>&1 || { echo "pad $pad: build failed"; continue; }
  echo -n "pad $pad: "
  timeout -k 10 120 python3 bench.py --steps 300 --warmup 20 --no-cpu | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['roofline']['kernel_avg_ms'], d['roofline']['frac'])"
done
OHGPU_EXTRA_FLAGS="-DOHGPU_DIAG -DOHGPU_DIAG_ONE_KERNEL" python3 ohpipeline_amd/build.py --force > /dev/null 2>&1
D=$R/gpurun_out/r3/pmc_rot
rm -rf "$D"
(cd /tmp && TMPDIR=/tmp timeout -k 10 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d "$D" -- python3 "$R/bench.py" --steps 10 --warmup 3 --no-cpu --sustain 0.2 > /dev/null 2>&1)
python3 - "$D" <<'PY'
import csv, glob, sys
from collections import defaultdict
acc = defaultdict(list)
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "src_lean_kernel" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
print("rotated rings, pad 0:", {k: round(sum(v) / len(v) / 1e6, 2) for k, v in sorted(acc.items())})
PY
rm -rf "$D"
python3 ohpipeline_amd/build.py --force > /dev/null 2>&1
