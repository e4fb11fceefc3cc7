#!/bin/bash
# Same-box ablations of src_mfma_kernel: recompiles that one source (and the planner) per flag set -- the other objects of the tree's
# build are reused --, relinks libohgpu.so and times the headline launch.
# Usage: bash tools/exp_mfma.sh "<flags A>" "<flags B>" ...   (a flag set may start with ENV=VALUE words; restores the tree's build at the end)
cd "${GRAFT_REPO_ROOT:-$(pwd)}"
trap 'python3 ohpipeline_amd/build.py --force > /dev/null 2>&1' EXIT
OBJ=ohpipeline_amd/build/obj
STEPS=${MF_EXP_STEPS:-200}
CC="/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -I include"
for set in "$@"; do
  bash tools/check_diag_flags.sh $set || { echo "[$set]: refused"; continue; }
  envs=""; flags=""
  for w in $set; do case "$w" in -*) flags="$flags $w";; *=*) envs="$envs $w";; esac; done
  $CC $flags -x hip -c ohpipeline_amd/csrc/src_mfma_kernel.hip -o /tmp/mf_exp.o 2> /tmp/mf_exp.err || { echo "[$set]: build failed"; tail -5 /tmp/mf_exp.err; continue; }
  $CC $flags -x hip -c ohpipeline_amd/csrc/src_plan.cpp -o /tmp/mf_plan.o 2> /tmp/mf_exp.err || { echo "[$set]: build failed"; tail -5 /tmp/mf_exp.err; continue; }
  $CC $flags -x hip -c ohpipeline_amd/csrc/src_mfma_wg_kernel.hip -o /tmp/mf_wg.o 2> /tmp/mf_exp.err || { echo "[$set]: build failed"; tail -5 /tmp/mf_exp.err; continue; }
  TAG=$(ls -t $OBJ/ohgpu_api.hip.*.o | head -1 | sed 's/.*ohgpu_api\.hip\.\([0-9a-f]*\)\..*/\1/')      # (one build's objects: the restore at exit adds a second set)
  objs=$(ls $OBJ/*.$TAG.*.o | grep -v -e src_mfma_kernel -e src_mfma_wg_kernel -e src_plan)
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ohpipeline_amd/lib/libohgpu.so $objs /tmp/mf_exp.o /tmp/mf_plan.o /tmp/mf_wg.o || { echo "[$set]: link failed"; continue; }
  echo -n "[$set]: "
  env $envs timeout -k 10 120 python3 bench.py --steps $STEPS --warmup 20 --no-cpu --no-extra-configs ${MF_EXP_ARGS} | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['roofline']['kernel_avg_ms'], d['roofline']['frac'])"
done
