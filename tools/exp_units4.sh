#!/bin/bash
# Unit schedules at twelve waves per CU (ring pad 0), alternating on one box.
cd "${GRAFT_REPO_ROOT:-$(pwd)}"
trap 'python3 ohpipeline_amd/build.py --force > /dev/null 2>&1' EXIT     # an interrupted experiment must not leave a diagnostic library behind
OHGPU_EXTRA_FLAGS="-DOHGPU_DIAG -DOHGPU_DIAG_ONE_KERNEL" python3 ohpipeline_amd/build.py --force > /dev/null 2>&1 || { echo "build failed"; exit 1; }
for rep in 1 2; do
for envs in "X=1" "OHGPU_DIAG_KB_MAX=1" "OHGPU_DIAG_KB_MAX=6" "OHGPU_DIAG_TAIL_ROUNDS=0.5" "OHGPU_DIAG_TAIL_ROUNDS=1.5" "OHGPU_DIAG_MAX_WAVES=11"; do
  echo -n "$envs: "
  env $envs timeout -k 10 120 python3 bench.py --steps 200 --warmup 20 --no-cpu | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['roofline']['kernel_avg_ms'], d['roofline']['frac'])"
done
done
python3 ohpipeline_amd/build.py --force > /dev/null 2>&1
