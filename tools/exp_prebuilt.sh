#!/bin/bash
# Times prebuilt variants of libohgpu.so (tools/build_variant.sh) on the GPU box, turn and turn about: ROUNDS passes over the tags,
# each pass one bench.py run per tag (the headline launch unless EXP_ARGS says otherwise).  The tree's own library is put back at the end.
# Usage (inside gpurun): bash tools/exp_prebuilt.sh base early ...        ("base" = the tree's library)
cd "${GRAFT_REPO_ROOT:-$(pwd)}"
LIB=ohpipeline_amd/lib/libohgpu.so
cp $LIB /tmp/libohgpu.base.so
trap 'cp /tmp/libohgpu.base.so $LIB' EXIT
STEPS=${EXP_STEPS:-200}
for r in $(seq 1 ${ROUNDS:-2}); do
  for tag in "$@"; do
    if [ "$tag" = base ]; then cp /tmp/libohgpu.base.so $LIB; else cp ohpipeline_amd/lib/variants/libohgpu.$tag.so $LIB || continue; fi
    echo -n "[$tag] "
    timeout -k 10 150 python3 bench.py --steps $STEPS --warmup 20 --no-cpu --no-extra-configs ${EXP_ARGS} 2> /tmp/exp_err.txt | python3 -c "
import json,sys
t=sys.stdin.read()
try:
    d=json.loads(t); r=d['roofline']; print(r['kernel_avg_ms'], r.get('kernel_median_ms'), r.get('kernel_min_ms'), r['frac'], d.get('check'), r.get('shader_clock_mhz'), 'plan_ms', d.get('plan_ms'), [(g.get('rate_in'), g.get('channels'), g.get('kernel_ms')) for g in d.get('config', {}).get('groups', [])])
except Exception as e:
    print('FAILED', e, t[-300:])
" || { tail -5 /tmp/exp_err.txt; }
  done
done
