#!/bin/bash
# Cumulative ablation of the lean kernel (diagnostic one-kernel builds, same box): what is left of the launch when the taps,
# the coefficient reads, the sample reads, the ring stores, the write-back and the staging DMA are taken out one after the other.
# (Wrong audio from the second line on: timing only.)  The product build is restored at the end.
cd "${GRAFT_REPO_ROOT:-$(pwd)}"
trap 'python3 ohpipeline_amd/build.py --force > /dev/null 2>&1' EXIT     # an interrupted experiment must not leave a diagnostic library behind
acc=""
# (round 3 ran this with a coefficient-read switch between the taps and the sample reads: those builds faulted on the device --
# the counted waits of the output body rely on the reloads being issued -- and the switch is gone)
for step in "" "-DOHGPU_DIAG_NO_TAPS" "-DOHGPU_DIAG_NO_RING -DOHGPU_DIAG_NO_DRAIN" "-DOHGPU_DIAG_NO_DMA"; do
  acc="$acc $step"
  OHGPU_EXTRA_FLAGS="-DOHGPU_DIAG -DOHGPU_DIAG_ONE_KERNEL $acc" python3 ohpipeline_amd/build.py --force > /dev/null 2>&1 || { echo "[$acc]: build failed"; continue; }
  echo -n "[$acc]: "
  OHGPU_DIAG_KB_MAX=8 OHGPU_DIAG_TAIL_ROUNDS=1.0 timeout -k 10 120 python3 bench.py --steps 300 --warmup 20 --no-cpu | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['roofline']['kernel_avg_ms'], d['roofline']['frac'])"
done
# and single ablations on top of the full kernel
for step in "-DOHGPU_DIAG_NO_DMA" "-DOHGPU_DIAG_NO_DRAIN" "-DOHGPU_DIAG_NO_DMA -DOHGPU_DIAG_NO_DRAIN" "-DOHGPU_DIAG_NO_STAGE_WAIT"; do
  OHGPU_EXTRA_FLAGS="-DOHGPU_DIAG -DOHGPU_DIAG_ONE_KERNEL $step" python3 ohpipeline_amd/build.py --force > /dev/null 2>&1 || { echo "[$step]: build failed"; continue; }
  echo -n "only [$step]: "
  OHGPU_DIAG_KB_MAX=8 OHGPU_DIAG_TAIL_ROUNDS=1.0 timeout -k 10 120 python3 bench.py --steps 300 --warmup 20 --no-cpu | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['roofline']['kernel_avg_ms'], d['roofline']['frac'])"
done
python3 ohpipeline_amd/build.py --force > /dev/null 2>&1
