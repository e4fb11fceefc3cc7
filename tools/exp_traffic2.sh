#!/bin/bash
# Kernel time and L2<->fabric traffic (FETCH_SIZE / WRITE_SIZE, KiB; TCC request counters) of diagnostic builds of the lean kernel.
# Usage (inside gpurun): bash tools/exp_traffic2.sh "<label>|<extra -D flags>" ...
cd "${GRAFT_REPO_ROOT:-$(pwd)}"
trap 'python3 ohpipeline_amd/build.py --force > /dev/null 2>&1' EXIT     # an interrupted experiment must not leave a diagnostic library behind
R=$(pwd)
for spec in "$@"; do
  IFS='|' read -r label flags <<< "$spec"
  OHGPU_EXTRA_FLAGS="-DOHGPU_DIAG -DOHGPU_DIAG_ONE_KERNEL $flags" python3 ohpipeline_amd/build.py --force > /dev/null 2>&1 || { echo "$label build failed"; continue; }
  ms=$(timeout -k 10 120 python3 bench.py --steps 200 --warmup 20 --no-cpu | python3 -c "import json,sys; print(json.loads(sys.stdin.read())['roofline']['kernel_avg_ms'])")
  i=0
  for c in "FETCH_SIZE" "WRITE_SIZE" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum"; do
    i=$((i+1)); rm -rf /tmp/trf_$i && (cd /tmp && TMPDIR=/tmp timeout -k 10 200 rocprofv3 --pmc $c --output-format csv -d /tmp/trf_$i -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu > /dev/null 2>&1)
  done
  python3 - "$label" "$ms" <<'PY'
import csv, glob, sys, collections
tot = collections.defaultdict(float); cnt = collections.defaultdict(int)
for f in glob.glob("/tmp/trf_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "src_" in r["Kernel_Name"] and "kernel" in r["Kernel_Name"]:
            tot[r["Counter_Name"]] += float(r["Counter_Value"]); cnt[r["Counter_Name"]] += 1
m = {k: tot[k] / max(cnt[k], 1) for k in tot}
fs = m.get("FETCH_SIZE", 0); ws = m.get("WRITE_SIZE", 0)
print("%-14s %s ms  read %.0f MB (2 x FETCH_SIZE)  written %.0f MB  total %.0f MB | %s" % (sys.argv[1], sys.argv[2], 2 * fs * 1024 / 1e6, ws * 1024 / 1e6, (2 * fs + ws) * 1024 / 1e6,
      "  ".join("%s %.3gM" % (k.replace("TCC_", "").replace("_sum", ""), v / 1e6) for k, v in sorted(m.items()) if k.startswith("TCC"))))
PY
done
python3 ohpipeline_amd/build.py --force > /dev/null 2>&1
