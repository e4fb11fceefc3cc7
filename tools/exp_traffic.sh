#!/bin/bash
# Kernel time and L2<->fabric traffic (FETCH_SIZE / WRITE_SIZE, KiB) of experiment builds.  Usage: bash tools/exp_traffic.sh STORE_SC1 ...
cd "${GRAFT_REPO_ROOT:-$(pwd)}"
R=$(pwd)
for v in BASE "$@"; do
  if [ "$v" = BASE ]; then F="-DOHGPU_EXP_ONE_KERNEL"; else F="-DOHGPU_EXP_ONE_KERNEL -DOHGPU_EXP_$v"; fi
  OHGPU_EXTRA_FLAGS="$F" python3 ohpipeline_amd/build.py --force > /dev/null 2>&1 || { echo "$v build failed"; continue; }
  ms=$(timeout -k 10 120 python3 bench.py --steps 5 --warmup 2 --no-cpu | python3 -c "import json,sys; print(json.loads(sys.stdin.read())['roofline']['kernel_avg_ms'])")
  for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf /tmp/trf_$c && (cd /tmp && TMPDIR=/tmp timeout -k 10 200 rocprofv3 --pmc $c --output-format csv -d /tmp/trf_$c -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu > /dev/null 2>&1)
  done
  python3 - "$v" "$ms" <<'PY'
import csv, glob, sys, collections
tot = collections.defaultdict(float); cnt = collections.defaultdict(int)
for f in glob.glob("/tmp/trf_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "src_block_kernel" in r["Kernel_Name"]:
            tot[r["Counter_Name"]] += float(r["Counter_Value"]); cnt[r["Counter_Name"]] += 1
fs = tot["FETCH_SIZE"] / max(cnt["FETCH_SIZE"], 1); ws = tot["WRITE_SIZE"] / max(cnt["WRITE_SIZE"], 1)
print("%-10s %s ms  read %.0f MB (2 x FETCH_SIZE)  written %.0f MB  total %.0f MB" % (sys.argv[1], sys.argv[2], 2 * fs * 1024 / 1e6, ws * 1024 / 1e6, (2 * fs + ws) * 1024 / 1e6))
PY
done
python3 ohpipeline_amd/build.py --force > /dev/null 2>&1
