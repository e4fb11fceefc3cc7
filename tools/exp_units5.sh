#!/bin/bash
# The "one long unit per wave" rule and its neighbours at twelve waves per CU, alternating on one box; then config 4.
cd "${GRAFT_REPO_ROOT:-$(pwd)}"
trap 'python3 ohpipeline_amd/build.py --force > /dev/null 2>&1' EXIT     # an interrupted experiment must not leave a diagnostic library behind
OHGPU_EXTRA_FLAGS="-DOHGPU_DIAG -DOHGPU_DIAG_ONE_KERNEL" python3 ohpipeline_amd/build.py --force > /dev/null 2>&1 || { echo "build failed"; exit 1; }
for rep in 1 2; do
for envs in "X=1" "OHGPU_DIAG_KB_MAX=1" "OHGPU_DIAG_TAIL_ROUNDS=0.5" "OHGPU_DIAG_LONG_ROUNDS=2" "OHGPU_DIAG_KB_MAX=6" "OHGPU_DIAG_MAX_WAVES=11"; do
  echo -n "$envs: "
  env $envs timeout -k 10 120 python3 bench.py --steps 200 --warmup 20 --no-cpu | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['roofline']['kernel_avg_ms'], d['roofline']['frac'])"
done
done
python3 ohpipeline_amd/build.py --force > /dev/null 2>&1
for i in 1 2; do python3 bench.py --config 4 --no-cpu --steps 20 --warmup 5 | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('config 4:', d['roofline']['kernel_avg_ms'], d['roofline']['frac'], [g['kernel_ms'] for g in d['config']['groups']])"; done
