#!/bin/bash
# Timing experiments on the GPU box: rebuilds libohgpu.so with one experiment macro at a time and times bench.py.
# (Experiment builds produce wrong audio on purpose; the product build is restored at the end.)
cd "${GRAFT_REPO_ROOT:-$(pwd)}"
for v in BASE "$@"; do
  if [ "$v" = BASE ]; then F="-DOHGPU_EXP_ONE_KERNEL"; else F="-DOHGPU_EXP_ONE_KERNEL -DOHGPU_EXP_$v"; fi
  OHGPU_EXTRA_FLAGS="$F" python3 ohpipeline_amd/build.py --force > /dev/null 2>&1 || { echo "$v build failed"; continue; }
  echo -n "$v: "
  timeout -k 10 120 python3 bench.py --steps 5 --warmup 2 --no-cpu | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['roofline']['kernel_avg_ms'])"
done
python3 ohpipeline_amd/build.py --force > /dev/null 2>&1
