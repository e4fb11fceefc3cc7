#!/bin/bash
# Shader clock and launch time against the number of waves per CU (OHGPU_EXP_MAX_WAVES): is the flat part of the
# occupancy curve the clock giving way?  Product build; SQ_BUSY_CYCLES / GRBM_GUI_ACTIVE from a separate --pmc run each.
cd "${GRAFT_REPO_ROOT:-$(pwd)}"
R=$(pwd)
for w in 4 8 10 12; do
  ms=$(OHGPU_EXP_MAX_WAVES=$w timeout -k 10 120 python3 bench.py --steps 5 --warmup 2 --no-cpu | python3 -c "import json,sys; print(json.loads(sys.stdin.read())['roofline']['kernel_avg_ms'])")
  rm -rf /tmp/clk && (cd /tmp && OHGPU_EXP_MAX_WAVES=$w TMPDIR=/tmp timeout -k 10 200 rocprofv3 --pmc SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VALU --output-format csv -d /tmp/clk -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu > /dev/null 2>&1)
  python3 - "$w" "$ms" <<'PY'
import csv, glob, sys, collections
tot = collections.defaultdict(float); cnt = collections.defaultdict(int)
for f in glob.glob("/tmp/clk/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "src_block_kernel" in r["Kernel_Name"]:
            tot[r["Counter_Name"]] += float(r["Counter_Value"]); cnt[r["Counter_Name"]] += 1
ms = float(sys.argv[2])
gui = tot["GRBM_GUI_ACTIVE"] / max(cnt["GRBM_GUI_ACTIVE"], 1) / 8.0     # summed over 8 XCDs
valu = tot["SQ_ACTIVE_INST_VALU"] / max(cnt["SQ_ACTIVE_INST_VALU"], 1) * 4 / 1024.0     # quad-cycles, 1024 SIMDs
print("%2s waves/CU  %.4f ms  %.3f GHz  vector pipe busy %.0f %%" % (sys.argv[1], ms, gui / ms / 1e6, 100.0 * valu / gui))
PY
done
