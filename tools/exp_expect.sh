#!/bin/bash
# Same-box A/B, alternating: the advance-without-output block out of line (the tree) against in line (round 2's layout).
cd "${GRAFT_REPO_ROOT:-$(pwd)}"
trap 'python3 ohpipeline_amd/build.py --force > /dev/null 2>&1' EXIT     # an interrupted experiment must not leave a diagnostic library behind
for flags in "" "-DOHGPU_DIAG_NO_EXPECT" "" "-DOHGPU_DIAG_NO_EXPECT" "" "-DOHGPU_DIAG_NO_EXPECT"; do
  OHGPU_EXTRA_FLAGS="-DOHGPU_DIAG -DOHGPU_DIAG_ONE_KERNEL $flags" python3 ohpipeline_amd/build.py --force > /dev/null 2>&1 || { echo "[$flags]: build failed"; continue; }
  echo -n "[$flags]: "
  timeout -k 10 120 python3 bench.py --steps 300 --warmup 20 --no-cpu | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['roofline']['kernel_avg_ms'], d['roofline']['frac'])"
done
python3 ohpipeline_amd/build.py --force > /dev/null 2>&1
