#!/bin/bash
# Steady-state kernel-trace summaries of the line kernels (round 3: every summary over >= 200 launches behind a 1 s sustain
# phase; round 2's were 13 launches on a cool chip).  Usage (inside gpurun): [PROFILE_ROUND=r05] bash tools/profile_lines.sh
cd "${GRAFT_REPO_ROOT:-$(pwd)}"
R=$(pwd)
P=${PROFILE_ROUND:-r05}
export OHGPU_PROFILE_TRACE_ONLY=1
OHGPU_PROFILE_TARGET=$R/tools/bench_pcm.py bash tools/profile_round.sh ${P}_pcm "tools/bench_pcm.py --steps 200 --warmup 5 (sustain 1 s)" -- --steps 200 --warmup 5
OHGPU_PROFILE_TARGET=$R/tools/bench_pcm.py bash tools/profile_round.sh ${P}_pcm_ramped "tools/bench_pcm.py --steps 200 --warmup 5 --all-ramped (sustain 1 s)" -- --steps 200 --warmup 5 --all-ramped
OHGPU_PROFILE_TARGET=$R/tools/bench_pcm.py bash tools/profile_round.sh ${P}_pcm_mix "tools/bench_pcm.py --steps 200 --warmup 5 --mix --all-ramped (sustain 1 s)" -- --steps 200 --warmup 5 --mix --all-ramped
OHGPU_PROFILE_TARGET=$R/tools/bench_ohm.py bash tools/profile_round.sh ${P}_ohm "tools/bench_ohm.py --steps 200 (sustain 1 s per case)" -- --steps 200
OHGPU_PROFILE_TARGET=$R/tools/bench_fmt.py bash tools/profile_round.sh ${P}_fmt "tools/bench_fmt.py --steps 200 (sustain 1 s per case)" -- --steps 200
grep -h "ms_avg" gpurun_out/profiles_new/${P}_pcm*_log.txt gpurun_out/profiles_new/${P}_ohm_log.txt gpurun_out/profiles_new/${P}_fmt_log.txt | cut -c1-260
