#!/bin/bash
# Cumulative ablations of the block kernel on the GPU box: each line removes one more part of the per-output work
# (builds produce wrong audio on purpose; the product build is restored at the end).  Prints kernel ms per launch.
cd "${GRAFT_REPO_ROOT:-$(pwd)}"
ACC="-DOHGPU_EXP_ONE_KERNEL"
run() {
  OHGPU_EXTRA_FLAGS="$ACC" python3 ohpipeline_amd/build.py --force > /dev/null 2>&1 || { echo "$1 build failed"; return; }
  echo -n "$1: "
  timeout -k 10 120 python3 bench.py --steps 5 --warmup 2 --no-cpu | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['roofline']['kernel_avg_ms'])"
}
run BASE
for v in "$@"; do ACC="$ACC -DOHGPU_EXP_$v"; run "+$v"; done
python3 ohpipeline_amd/build.py --force > /dev/null 2>&1
