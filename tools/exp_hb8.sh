#!/bin/bash
# Same-box A/B on config 4: the eight-channel half-band kernel at 16 waves per CU (the tree) against 12.
cd "${GRAFT_REPO_ROOT:-$(pwd)}"
trap 'python3 ohpipeline_amd/build.py --force > /dev/null 2>&1' EXIT     # an interrupted experiment must not leave a diagnostic library behind
for w in 16 12 16 12; do
  OHGPU_EXTRA_FLAGS="-DOHGPU_DIAG -DOHGPU_DIAG_HB8_WAVES=$w" python3 ohpipeline_amd/build.py --force > /dev/null 2>&1 || { echo "w $w: build failed"; continue; }
  echo -n "hb8 waves $w: "
  timeout -k 10 300 python3 bench.py --config 4 --steps 20 --warmup 5 --no-cpu | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print(d['roofline']['kernel_avg_ms'], d['roofline']['frac'], [g['kernel_ms'] for g in d['config']['groups']])"
done
python3 ohpipeline_amd/build.py --force > /dev/null 2>&1
