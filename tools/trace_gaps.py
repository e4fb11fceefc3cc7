#!/usr/bin/env python3
"""Gaps between consecutive dispatches of a kernel in a rocprofv3 --kernel-trace CSV (end of one to start of the next, on the
device's clock): what a back-to-back launch sequence pays between kernels.  Usage: python3 tools/trace_gaps.py <dir> [kernel substring]"""
import csv
import glob
import sys

import numpy as np

d, name = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else "src_mfma_wg_kernel")
rows = []
for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
gaps, durs = [], []
for (s0, e0, n0), (s1, e1, n1) in zip(rows, rows[1:]):
    if name in n0 and name in n1:
        gaps.append(s1 - e0)
        durs.append(e0 - s0)
g, du = np.array(gaps) / 1e3, np.array(durs) / 1e3
back = g[g < 50.0]                                          # (back-to-back: the queue was not empty)
print("pairs %d, back-to-back %d: gap us median %.2f mean %.2f p10 %.2f p90 %.2f; kernel us median %.1f" %
      (g.size, back.size, np.median(back), back.mean(), np.percentile(back, 10), np.percentile(back, 90), np.median(du)))
