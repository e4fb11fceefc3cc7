#!/bin/bash
# Where ohgpu_src_batch_create spends its time on the headline's 512 000 messages: rebuilds the API and the planner with
# -DOHGPU_PLAN_TIMING (stage timings on stderr), relinks, runs a short bench.py (whose plan_ms is that call); restores the tree's build at the end.
cd "${GRAFT_REPO_ROOT:-$(pwd)}"
trap 'python3 ohpipeline_amd/build.py --force > /dev/null 2>&1' EXIT
OBJ=ohpipeline_amd/build/obj
CC="/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -I include -DOHGPU_PLAN_TIMING $*"
$CC -x hip -c ohpipeline_amd/csrc/ohgpu_api.hip -o /tmp/pt_api.o || exit 1
$CC -x hip -c ohpipeline_amd/csrc/src_plan.cpp -o /tmp/pt_plan.o || exit 1
TAG=$(ls -t $OBJ/ohgpu_api.hip.*.o | head -1 | sed 's/.*ohgpu_api\.hip\.\([0-9a-f]*\)\..*/\1/')
objs=$(ls $OBJ/*.$TAG.*.o | grep -v -e ohgpu_api -e src_plan)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ohpipeline_amd/lib/libohgpu.so $objs /tmp/pt_api.o /tmp/pt_plan.o || exit 1
python3 bench.py --steps 3 --warmup 1 --no-cpu --no-extra-configs --sustain 0.05 2>&1 | grep -o "\[plan timing\].*\|\"plan_ms\": [0-9.]*" | head -20
