#!/bin/bash
# Per-phase shader-clock stamps of the lean block kernel (diagnostic build -DOHGPU_DIAG_STAMP; the product build never stamps).
# Usage (inside gpurun): bash tools/exp_stamp.sh [env assignments...]   The product build is restored at the end.
cd "${GRAFT_REPO_ROOT:-$(pwd)}"
trap 'python3 ohpipeline_amd/build.py --force > /dev/null 2>&1' EXIT     # an interrupted experiment must not leave a diagnostic library behind
OHGPU_EXTRA_FLAGS="-DOHGPU_DIAG -DOHGPU_DIAG_ONE_KERNEL -DOHGPU_DIAG_STAMP" python3 ohpipeline_amd/build.py --force > /dev/null 2>&1 || { echo "build failed"; exit 1; }
for envs in "X=1" "$@"; do
  echo "== $envs"
  env $envs OHGPU_DIAG_STAMP_FILE=/tmp/stamp.txt timeout -k 10 120 python3 bench.py --steps 5 --warmup 2 --no-cpu > /dev/null 2>&1
  cat /tmp/stamp.txt
done
python3 ohpipeline_amd/build.py --force > /dev/null 2>&1
