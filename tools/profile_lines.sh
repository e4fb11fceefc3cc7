#!/bin/bash
# Steady-state kernel-trace summaries of the line kernels (round 3: every summary over >= 200 launches behind a 1 s sustain
# phase; round 2's were 13 launches on a cool chip).  Usage (inside gpurun): bash tools/profile_lines.sh
cd "${GRAFT_REPO_ROOT:-$(pwd)}"
R=$(pwd)
export OHGPU_PROFILE_TRACE_ONLY=1
OHGPU_PROFILE_TARGET=$R/tools/bench_pcm.py bash tools/profile_bench.sh r03_pcm --steps 200 --warmup 5 > gpurun_out/r3/prof_pcm.log 2>&1
OHGPU_PROFILE_TARGET=$R/tools/bench_pcm.py bash tools/profile_bench.sh r03_pcm_ramped --steps 200 --warmup 5 --all-ramped > gpurun_out/r3/prof_pcm_ramped.log 2>&1
OHGPU_PROFILE_TARGET=$R/tools/bench_pcm.py bash tools/profile_bench.sh r03_pcm_mix --steps 200 --warmup 5 --mix > gpurun_out/r3/prof_pcm_mix.log 2>&1
OHGPU_PROFILE_TARGET=$R/tools/bench_ohm.py bash tools/profile_bench.sh r03_ohm --steps 200 > gpurun_out/r3/prof_ohm.log 2>&1
OHGPU_PROFILE_TARGET=$R/tools/bench_fmt.py bash tools/profile_bench.sh r03_fmt --steps 200 > gpurun_out/r3/prof_fmt.log 2>&1
grep -h "ms_avg" gpurun_out/prof_r03_pcm*/log.txt gpurun_out/prof_r03_ohm/log.txt gpurun_out/prof_r03_fmt/log.txt | cut -c1-260
