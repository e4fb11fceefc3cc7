#!/bin/bash
# The shader clock the chip holds under the lean kernel, by waves per CU (stamped diagnostic builds: s_memtime against the
# 100 MHz s_memrealtime over every wave's life).  The product build is restored at the end.
cd "${GRAFT_REPO_ROOT:-$(pwd)}"
trap 'python3 ohpipeline_amd/build.py --force > /dev/null 2>&1' EXIT     # an interrupted experiment must not leave a diagnostic library behind
run() {
  label="$1"; flags="$2"; shift 2
  OHGPU_EXTRA_FLAGS="-DOHGPU_DIAG -DOHGPU_DIAG_ONE_KERNEL -DOHGPU_DIAG_STAMP $flags" python3 ohpipeline_amd/build.py --force > /dev/null 2>&1 || { echo "$label: build failed"; return; }
  for envs in "$@"; do
    echo "== $label [$envs]"
    env $envs OHGPU_DIAG_STAMP_FILE=/tmp/stamp.txt timeout -k 10 120 python3 bench.py --steps 5 --warmup 2 --no-cpu > /dev/null 2>&1
    cat /tmp/stamp.txt
  done
}
run "sf16" "" "X=1" "OHGPU_DIAG_MAX_WAVES=8" "OHGPU_DIAG_MAX_WAVES=4"
run "sf8 w16" "-DOHGPU_LEAN_STAGE_FRAMES=8 -DOHGPU_LEAN_MAX_WAVES_T32=16" "X=1" "OHGPU_DIAG_MAX_WAVES=12" "OHGPU_DIAG_MAX_WAVES=8"
python3 ohpipeline_amd/build.py --force > /dev/null 2>&1
