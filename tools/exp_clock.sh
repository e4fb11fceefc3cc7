#!/bin/bash
# Does the shader clock change between experiment builds?  For each variant: kernel time (HIP events, bench.py) and
# SQ_BUSY_CYCLES / GRBM_GUI_ACTIVE per launch (rocprofv3 --pmc, separate run).  Usage: bash tools/exp_clock.sh NODMA NOFMA
cd "${GRAFT_REPO_ROOT:-$(pwd)}"
R=$(pwd)
for v in BASE "$@"; do
  if [ "$v" = BASE ]; then F="-DOHGPU_EXP_ONE_KERNEL"; else F="-DOHGPU_EXP_ONE_KERNEL -DOHGPU_EXP_$v"; fi
  OHGPU_EXTRA_FLAGS="$F" python3 ohpipeline_amd/build.py --force > /dev/null 2>&1 || { echo "$v build failed"; continue; }
  ms=$(timeout -k 10 120 python3 bench.py --steps 5 --warmup 2 --no-cpu | python3 -c "import json,sys; print(json.loads(sys.stdin.read())['roofline']['kernel_avg_ms'])")
  rm -rf /tmp/clk && (cd /tmp && TMPDIR=/tmp timeout -k 10 200 rocprofv3 --pmc SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d /tmp/clk -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu > /dev/null 2>&1)
  python3 - "$v" "$ms" <<'PY'
import csv, glob, sys, collections
tot = collections.defaultdict(float); cnt = collections.defaultdict(int)
for f in glob.glob("/tmp/clk/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "src_block_kernel" in r["Kernel_Name"]:
            tot[r["Counter_Name"]] += float(r["Counter_Value"]); cnt[r["Counter_Name"]] += 1
ms = float(sys.argv[2])
sq = tot["SQ_BUSY_CYCLES"] / max(cnt["SQ_BUSY_CYCLES"], 1) / 32.0       # summed over 32 shader engines
gui = tot["GRBM_GUI_ACTIVE"] / max(cnt["GRBM_GUI_ACTIVE"], 1) / 8.0     # summed over 8 XCDs
print("%-8s %.4f ms  SQ_BUSY/SE %.0f cycles (%.3f GHz)  GUI_ACTIVE/XCD %.0f (%.3f GHz)" % (sys.argv[1], ms, sq, sq / ms / 1e6, gui, gui / ms / 1e6))
PY
done
python3 ohpipeline_amd/build.py --force > /dev/null 2>&1
