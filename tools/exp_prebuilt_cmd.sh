#!/bin/bash
# As tools/exp_prebuilt.sh, for any of the tools' benchmarks that print one JSON line with "ms_avg": prebuilt variants of libohgpu.so,
# turn and turn about.  Usage (inside gpurun): EXP_CMD="python3 tools/bench_pcm.py --steps 200 --warmup 5" bash tools/exp_prebuilt_cmd.sh base line2 ...
cd "${GRAFT_REPO_ROOT:-$(pwd)}"
LIB=ohpipeline_amd/lib/libohgpu.so
cp $LIB /tmp/libohgpu.base.so
trap 'cp /tmp/libohgpu.base.so $LIB' EXIT
for r in $(seq 1 ${ROUNDS:-2}); do
  for tag in "$@"; do
    if [ "$tag" = base ]; then cp /tmp/libohgpu.base.so $LIB; else cp ohpipeline_amd/lib/variants/libohgpu.$tag.so $LIB || continue; fi
    echo -n "[$tag] "
    timeout -k 10 150 $EXP_CMD 2> /tmp/exp_err.txt | python3 -c "
import json,sys
for line in sys.stdin:
    line=line.strip()
    if not line.startswith('{'): continue
    d=json.loads(line)
    print(d.get('ms_avg'), d.get('ms_min'), d.get('frac_of_8TBps'), d.get('check'), d.get('case', ''), end='; ')
print()
" || tail -3 /tmp/exp_err.txt
  done
done
