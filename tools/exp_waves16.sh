#!/bin/bash
# Same-box A/B: the stereo lean kernel with 8-frame stages and up to 16 waves per CU (four per SIMD) against the product shape
# (16-frame stages, 11 waves).  Diagnostic one-kernel builds; the product build is restored at the end.
cd "${GRAFT_REPO_ROOT:-$(pwd)}"
trap 'python3 ohpipeline_amd/build.py --force > /dev/null 2>&1' EXIT     # an interrupted experiment must not leave a diagnostic library behind
run() {
  label="$1"; flags="$2"; shift 2
  OHGPU_EXTRA_FLAGS="-DOHGPU_DIAG -DOHGPU_DIAG_ONE_KERNEL $flags" python3 ohpipeline_amd/build.py --force > /dev/null 2>&1 || { echo "$label: build failed"; return; }
  for envs in "$@"; do
    echo -n "$label [$envs]: "
    env $envs timeout -k 10 120 python3 bench.py --steps 300 --warmup 20 --no-cpu | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['roofline']['kernel_avg_ms'], d['roofline']['frac'])"
  done
}
run "sf16 w11" "" "X=1" "OHGPU_DIAG_KB_MAX=8 OHGPU_DIAG_TAIL_ROUNDS=1.0"
run "sf8 w16" "-DOHGPU_LEAN_STAGE_FRAMES=8 -DOHGPU_LEAN_MAX_WAVES_T32=16" "X=1" "OHGPU_DIAG_KB_MAX=1" "OHGPU_DIAG_KB_MAX=8 OHGPU_DIAG_TAIL_ROUNDS=1.0" "OHGPU_DIAG_MAX_WAVES=14" "OHGPU_DIAG_MAX_WAVES=12"
run "sf8 w12" "-DOHGPU_LEAN_STAGE_FRAMES=8" "X=1"
run "sf16 w11" "" "X=1"
python3 ohpipeline_amd/build.py --force > /dev/null 2>&1
