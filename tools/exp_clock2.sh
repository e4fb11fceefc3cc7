#!/bin/bash
# Clock held under the block kernels: kernel time by HIP events (bench.py) and GRBM_GUI_ACTIVE / SQ_BUSY_CYCLES per launch
# (rocprofv3 --pmc, separate run, counters only).  Usage (inside gpurun): bash tools/exp_clock2.sh <variant> [<variant> ...]
cd "${GRAFT_REPO_ROOT:-$(pwd)}"
R=$(pwd)
for v in "$@"; do
  ms=$(timeout -k 10 120 python3 bench.py --steps 200 --warmup 20 --no-cpu --variant $v | python3 -c "import json,sys; print(json.loads(sys.stdin.read())['roofline']['kernel_avg_ms'])")
  rm -rf /tmp/clk && (cd /tmp && TMPDIR=/tmp timeout -k 10 200 rocprofv3 --pmc SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_BRANCH --output-format csv -d /tmp/clk -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu --variant $v > /dev/null 2>&1)
  python3 - "$v" "$ms" <<'PY'
import csv, glob, sys, collections
tot = collections.defaultdict(float); cnt = collections.defaultdict(int)
for f in glob.glob("/tmp/clk/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "src_" in r["Kernel_Name"] and "kernel" in r["Kernel_Name"]:
            tot[r["Counter_Name"]] += float(r["Counter_Value"]); cnt[r["Counter_Name"]] += 1
ms = float(sys.argv[2])
m = {k: tot[k] / max(cnt[k], 1) for k in tot}
print("variant %s: %.4f ms | GUI_ACTIVE/XCD %.0f cycles (%.3f GHz) | SQ_BUSY/SE %.0f | VALU %.1fM (active %.1fM quad-cycles) SALU %.1fM branches %.1fM waves %.0f"
      % (sys.argv[1], ms, m.get("GRBM_GUI_ACTIVE", 0) / 8, m.get("GRBM_GUI_ACTIVE", 0) / 8 / ms / 1e6, m.get("SQ_BUSY_CYCLES", 0) / 32,
         m.get("SQ_INSTS_VALU", 0) / 1e6, m.get("SQ_ACTIVE_INST_VALU", 0) / 1e6, m.get("SQ_INSTS_SALU", 0) / 1e6, m.get("SQ_INSTS_BRANCH", 0) / 1e6, m.get("SQ_WAVES", 0)))
PY
done
