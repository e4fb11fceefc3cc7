#!/bin/bash
# One profiling session on the GPU box: profile, summarise THERE (tools/summarize_prof.py), keep the summaries under
# gpurun_out/profiles_new/ and drop the raw rocprofv3 directories (gpurun merges at most 64 MiB back).
# Usage (inside gpurun): bash tools/profile_round.sh <tag> "<what the summary should say was profiled>" [--latest] -- [bench args...]
cd "${GRAFT_REPO_ROOT:-$(pwd)}"
TAG=$1; WHAT=$2; shift 2
LATEST=""
if [ "${1:-}" = "--latest" ]; then LATEST="--latest"; shift; fi
[ "${1:-}" = "--" ] && shift
bash tools/profile_bench.sh "$TAG" "$@" > "gpurun_out/prof_$TAG.log" 2>&1
python3 tools/summarize_prof.py "$TAG" "$WHAT" $LATEST > /dev/null 2>> "gpurun_out/prof_$TAG.log"
mkdir -p gpurun_out/profiles_new
cp profiles/${TAG}_summary.md profiles/${TAG}_summary.json gpurun_out/profiles_new/ 2>/dev/null
[ -n "$LATEST" ] && cp profiles/pmc_latest.json gpurun_out/profiles_new/
cp "gpurun_out/prof_$TAG/log.txt" "gpurun_out/profiles_new/${TAG}_log.txt" 2>/dev/null
rm -rf "gpurun_out/prof_$TAG"
echo "$TAG done"
