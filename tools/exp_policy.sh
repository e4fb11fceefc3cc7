#!/bin/bash
# Same-box A/B: cache policy bits on the lean kernel's staging loads (global_load_lds_dwordx4 [nt | sc1 | sc0 sc1 | sc0]).
cd "${GRAFT_REPO_ROOT:-$(pwd)}"
trap 'python3 ohpipeline_amd/build.py --force > /dev/null 2>&1' EXIT     # an interrupted experiment must not leave a diagnostic library behind
for id in 0 1 2 3 4 0 1; do
  OHGPU_EXTRA_FLAGS="-DOHGPU_DIAG -DOHGPU_DIAG_ONE_KERNEL -DOHGPU_DIAG_DMA_POLICY_ID=$id" python3 ohpipeline_amd/build.py --force > /dev/null 2>&1 || { echo "policy $id: build failed"; continue; }
  echo -n "policy $id: "
  timeout -k 10 120 python3 bench.py --steps 200 --warmup 20 --no-cpu | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['roofline']['kernel_avg_ms'], d['roofline']['frac'])"
done
python3 ohpipeline_amd/build.py --force > /dev/null 2>&1
