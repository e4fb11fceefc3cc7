#!/bin/bash
# Ring pad 4 (eleven waves per CU) against 0 (twelve, but the ring stores then collide in the banks), both under the
# one-long-unit-per-wave schedule, alternating on one box.
cd "${GRAFT_REPO_ROOT:-$(pwd)}"
trap 'python3 ohpipeline_amd/build.py --force > /dev/null 2>&1' EXIT     # an interrupted experiment must not leave a diagnostic library behind
for pad in 4 0 4 0 4 0 4 0; do
  OHGPU_EXTRA_FLAGS="-DOHGPU_DIAG -DOHGPU_DIAG_ONE_KERNEL -DOHGPU_LEAN_RING_PAD=$pad" python3 ohpipeline_amd/build.py --force > /dev/null 2>&1 || { echo "pad $pad: build failed"; continue; }
  echo -n "pad $pad: "
  timeout -k 10 120 python3 bench.py --steps 300 --warmup 20 --no-cpu | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['roofline']['kernel_avg_ms'], d['roofline']['frac'])"
done
python3 ohpipeline_amd/build.py --force > /dev/null 2>&1
