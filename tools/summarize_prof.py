#!/usr/bin/env python3
"""Summarises gpurun_out/prof_<tag>/ (written by tools/profile_bench.sh) into profiles/<tag>_summary.md/json:
per-kernel time statistics from the kernel trace and per-kernel, per-dispatch means of every PMC counter."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def short(name):
    name = name.split("(")[0]
    for k in ("src_mfma_wg_kernel", "src_mfma_kernel", "src_lean_kernel"):      # keep the instantiation (lean: <T, channels, src bytes, src LE, dst bytes, dst LE>)
        if k in name:
            return k + name[name.index(k) + len(k):].replace(" ", "")
    for k in ("src_block_kernel", "src_msg_kernel_v1", "pcm_line_kernel", "pcm_msg_kernel_v1", "fmt_line_kernel", "fmt_kernel_v1", "flywheel_kernel", "ohm_header_kernel", "ohm_wide_kernel", "ohm_select_ramp_kernel",
              "unpack_stereo_kernel", "flac_stereo_kernel"):
        if k in name:
            return k
    return name[-60:]


def newest(pattern):
    """gpurun merges every call's files into gpurun_out/: of several runs of one pass keep the latest."""
    by_dir = {}
    for f in glob.glob(pattern, recursive=True):
        d = os.path.dirname(f)
        if d not in by_dir or os.path.getmtime(f) > os.path.getmtime(by_dir[d]):
            by_dir[d] = f
    return sorted(by_dir.values())


def main():
    tag = sys.argv[1]
    what = sys.argv[2] if len(sys.argv) > 2 else "bench.py --steps 10 --warmup 3 --no-cpu"   # the profiled command
    latest = len(sys.argv) > 3 and sys.argv[3] == "--latest"           # this run is bench.py's default workload: feed roofline.traffic
    src = os.path.join(ROOT, "gpurun_out", f"prof_{tag}")
    out = {"tag": tag, "kernels": {}, "counters": {}}
    for f in newest(os.path.join(src, "trace", "**", "*kernel_trace.csv")):
        per = defaultdict(list)
        for r in csv.DictReader(open(f)):
            per[short(r["Kernel_Name"])].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
        for k, v in per.items():
            v.sort()
            out["kernels"][k] = {"calls": len(v), "avg_us": sum(v) / len(v), "min_us": v[0], "max_us": v[-1],
                                 "median_us": v[len(v) // 2]}
    for f in newest(os.path.join(src, "pmc*", "**", "*counter_collection.csv")):
        acc = defaultdict(lambda: defaultdict(list))
        for r in csv.DictReader(open(f)):
            acc[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, cs in acc.items():
            for c, v in cs.items():
                out["counters"].setdefault(k, {})[c] = sum(v) / len(v)
    stats = newest(os.path.join(src, "trace", "**", "*kernel_stats.csv"))
    os.makedirs(os.path.join(ROOT, "profiles"), exist_ok=True)
    json.dump(out, open(os.path.join(ROOT, "profiles", f"{tag}_summary.json"), "w"), indent=1, sort_keys=True)
    with open(os.path.join(ROOT, "profiles", f"{tag}_summary.md"), "w") as md:
        md.write(f"# rocprofv3 summary `{tag}`\n\nCommand: `rocprofv3 --kernel-trace --stats -- python3 {what}` "
                 "plus one `--pmc` pass per counter set (tools/profile_bench.sh).\n\n## Kernel time (kernel trace)\n\n")
        md.write("| kernel | calls | avg us | median us | min us | max us |\n|---|---|---|---|---|---|\n")
        for k, v in sorted(out["kernels"].items(), key=lambda kv: -kv[1]["avg_us"] * kv[1]["calls"]):
            md.write(f"| {k} | {v['calls']} | {v['avg_us']:.1f} | {v['median_us']:.1f} | {v['min_us']:.1f} | {v['max_us']:.1f} |\n")
        if stats:
            md.write("\n## rocprofv3 --stats (verbatim)\n\n```\n" + open(stats[0]).read() + "```\n")
        # A hand-written reading of the numbers, kept next to them -- and only next to the run it reads: the notes name the trace
        # average they were written from (`<!-- trace-avg-us: 438.8 -->`), and notes of another run are left out, loudly.
        notes = os.path.join(ROOT, "profiles", f"{tag}_notes.md")
        if os.path.exists(notes) and out["kernels"]:
            import re
            text = open(notes).read()
            m = re.search(r"<!--\s*trace-avg-us:\s*([0-9.]+)\s*-->", text)
            dom_avg = max(out["kernels"].values(), key=lambda v: v["avg_us"] * v["calls"])["avg_us"]
            if m and abs(float(m.group(1)) - dom_avg) <= 0.005 * dom_avg:
                md.write("\n" + text.rstrip() + "\n")
            else:
                sys.stderr.write(f"summarize_prof: profiles/{tag}_notes.md was written for a trace average of {m.group(1) if m else '?'} us, this run's is "
                                 f"{dom_avg:.1f} us: notes NOT included (re-read the numbers, then update the marker)\n")
        md.write("\n## PMC counters (mean per dispatch)\n\n")
        for k, cs in out["counters"].items():
            md.write(f"### {k}\n\n| counter | value |\n|---|---|\n")
            for c, v in sorted(cs.items()):
                md.write(f"| {c} | {v:.6g} |\n")
            md.write("\n")
    # HBM traffic of the dominant kernel, corrected as MI355X_MICROARCH.md (HBM section) prescribes: FETCH_SIZE and
    # WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE reads half the bytes of a wide (16 B/lane) streaming read -> x2.
    dom = max(out["kernels"].items(), key=lambda kv: kv[1]["avg_us"] * kv[1]["calls"])[0] if out["kernels"] else None
    if dom and dom in out["counters"] and "FETCH_SIZE" in out["counters"][dom] and "WRITE_SIZE" in out["counters"][dom]:
        c = out["counters"][dom]
        traffic = (2.0 * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024.0
        pmc = {"kernel": dom, "hbm_bytes_per_launch": int(traffic), "fetch_size_kib": c["FETCH_SIZE"],
               "write_size_kib": c["WRITE_SIZE"], "kernel_avg_us_trace": out["kernels"][dom]["avg_us"],
               "source": f"profiles/{tag}_summary.md: (2*FETCH_SIZE + WRITE_SIZE)*1024, separate --pmc passes",
               "streams_per_gpu": 256, "frames_per_stream": 441000, "kernel_variant": 0}
        sys.path.insert(0, ROOT)
        import bench                                                   # (the kernel sources this profile was taken from: bench.py quotes
        pmc["source_sha16"] = bench.source_fingerprint()               #  `traffic` only while they are unchanged)
        if latest:                                                     # bench.py reads this one for roofline.traffic
            json.dump(pmc, open(os.path.join(ROOT, "profiles", "pmc_latest.json"), "w"), indent=1, sort_keys=True)
        out["hbm_traffic"] = pmc
    print(json.dumps(out, indent=1, sort_keys=True))


if __name__ == "__main__":
    main()
