#!/bin/bash
# Same-box A/B of the lean kernel's unit schedule (blocks per row of the long units, rounds of one-block units kept for the end):
# diagnostic build, one instantiation, the planner's OHGPU_DIAG_KB_MAX / OHGPU_DIAG_TAIL_ROUNDS hooks; then the per-phase stamps.
# Usage (inside gpurun): bash tools/exp_units.sh    The product build is restored at the end.
cd "${GRAFT_REPO_ROOT:-$(pwd)}"
trap 'python3 ohpipeline_amd/build.py --force > /dev/null 2>&1' EXIT     # an interrupted experiment must not leave a diagnostic library behind
OHGPU_EXTRA_FLAGS="-DOHGPU_DIAG -DOHGPU_DIAG_ONE_KERNEL" python3 ohpipeline_amd/build.py --force > /dev/null 2>&1 || { echo "build failed"; exit 1; }
for envs in "OHGPU_DIAG_KB_MAX=1" "OHGPU_DIAG_KB_MAX=2" "OHGPU_DIAG_KB_MAX=3" "OHGPU_DIAG_KB_MAX=8 OHGPU_DIAG_TAIL_ROUNDS=1.5" "OHGPU_DIAG_KB_MAX=8 OHGPU_DIAG_TAIL_ROUNDS=0.6" "OHGPU_DIAG_KB_MAX=1" "OHGPU_DIAG_KB_MAX=3"; do
  echo -n "$envs: "
  env $envs timeout -k 10 120 python3 bench.py --steps 300 --warmup 20 --no-cpu | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['roofline']['kernel_avg_ms'], d['roofline']['frac'], d['plan_ms'])"
done
OHGPU_EXTRA_FLAGS="-DOHGPU_DIAG -DOHGPU_DIAG_ONE_KERNEL -DOHGPU_DIAG_STAMP" python3 ohpipeline_amd/build.py --force > /dev/null 2>&1 || { echo "stamp build failed"; exit 1; }
for envs in "OHGPU_DIAG_KB_MAX=1" "OHGPU_DIAG_KB_MAX=3" "OHGPU_DIAG_KB_MAX=8 OHGPU_DIAG_TAIL_ROUNDS=0.6"; do
  echo "== $envs"
  env $envs OHGPU_DIAG_STAMP_FILE=/tmp/stamp.txt timeout -k 10 120 python3 bench.py --steps 5 --warmup 2 --no-cpu > /dev/null 2>&1
  cat /tmp/stamp.txt
done
python3 ohpipeline_amd/build.py --force > /dev/null 2>&1
