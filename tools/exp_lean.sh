#!/bin/bash
# Same-box A/B of diagnostic builds of the lean block kernel (-DOHGPU_DIAG: one instantiation, environment hooks).
# Usage (inside gpurun): bash tools/exp_lean.sh "<label>|<extra -D flags>|<env assignments for the run>|<env assignments for the build>" ...
# The product build is restored at the end.
cd "${GRAFT_REPO_ROOT:-$(pwd)}"
trap 'python3 ohpipeline_amd/build.py --force > /dev/null 2>&1' EXIT     # an interrupted experiment must not leave a diagnostic library behind
for spec in "$@"; do
  IFS='|' read -r label flags envs benvs <<< "$spec"
  env $benvs OHGPU_EXTRA_FLAGS="-DOHGPU_DIAG -DOHGPU_DIAG_ONE_KERNEL $flags" python3 ohpipeline_amd/build.py --force > /dev/null 2>&1 || { echo "$label: build failed"; continue; }
  for rep in 1 2; do
    echo -n "$label: "
    env $envs timeout -k 10 120 python3 bench.py --steps 300 --warmup 20 --no-cpu | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['roofline']['kernel_avg_ms'], d['roofline']['frac'])"
  done
done
python3 ohpipeline_amd/build.py --force > /dev/null 2>&1
