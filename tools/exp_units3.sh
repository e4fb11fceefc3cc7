#!/bin/bash
# Unit schedules against each other on ONE box, alternating (box-to-box differences are larger than the effect).
cd "${GRAFT_REPO_ROOT:-$(pwd)}"
trap 'python3 ohpipeline_amd/build.py --force > /dev/null 2>&1' EXIT     # an interrupted experiment must not leave a diagnostic library behind
OHGPU_EXTRA_FLAGS="-DOHGPU_DIAG -DOHGPU_DIAG_ONE_KERNEL" python3 ohpipeline_amd/build.py --force > /dev/null 2>&1 || { echo "build failed"; exit 1; }
for rep in 1 2 3; do
for envs in "OHGPU_DIAG_KB_MAX=1" "OHGPU_DIAG_TAIL_ROUNDS=1.5" "OHGPU_DIAG_LONG_ROUNDS=1 OHGPU_DIAG_TAIL_ROUNDS=1.0" "OHGPU_DIAG_LONG_ROUNDS=1 OHGPU_DIAG_TAIL_ROUNDS=2.0" "OHGPU_DIAG_KB_MAX=2 OHGPU_DIAG_LONG_ROUNDS=3 OHGPU_DIAG_TAIL_ROUNDS=2.0"; do
  echo -n "$envs: "
  env $envs timeout -k 10 120 python3 bench.py --steps 200 --warmup 20 --no-cpu | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['roofline']['kernel_avg_ms'], d['roofline']['frac'])"
done
done
python3 ohpipeline_amd/build.py --force > /dev/null 2>&1
