#!/bin/bash
# Steady-state kernel-trace summaries of the line kernels (round 3: every summary over >= 200 launches behind a 1 s sustain
# phase; round 2's were 13 launches on a cool chip).  Usage (inside gpurun): bash tools/profile_lines.sh
cd "${GRAFT_REPO_ROOT:-$(pwd)}"
R=$(pwd)
export OHGPU_PROFILE_TRACE_ONLY=1
OHGPU_PROFILE_TARGET=$R/tools/bench_pcm.py bash tools/profile_round.sh r03_pcm "tools/bench_pcm.py --steps 200 --warmup 5 (sustain 1 s)" -- --steps 200 --warmup 5
OHGPU_PROFILE_TARGET=$R/tools/bench_pcm.py bash tools/profile_round.sh r03_pcm_ramped "tools/bench_pcm.py --steps 200 --warmup 5 --all-ramped (sustain 1 s)" -- --steps 200 --warmup 5 --all-ramped
OHGPU_PROFILE_TARGET=$R/tools/bench_pcm.py bash tools/profile_round.sh r03_pcm_mix "tools/bench_pcm.py --steps 200 --warmup 5 --mix --all-ramped (sustain 1 s)" -- --steps 200 --warmup 5 --mix --all-ramped
OHGPU_PROFILE_TARGET=$R/tools/bench_ohm.py bash tools/profile_round.sh r03_ohm "tools/bench_ohm.py --steps 200 (sustain 1 s per case)" -- --steps 200
OHGPU_PROFILE_TARGET=$R/tools/bench_fmt.py bash tools/profile_round.sh r03_fmt "tools/bench_fmt.py --steps 200 (sustain 1 s per case)" -- --steps 200
grep -h "ms_avg" gpurun_out/profiles_new/r03_pcm*_log.txt gpurun_out/profiles_new/r03_ohm_log.txt gpurun_out/profiles_new/r03_fmt_log.txt | cut -c1-260
