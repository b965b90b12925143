#!/usr/bin/env python3
"""GPU box: kernel timeline of a rocprofv3 --kernel-trace directory (csv): start (us), duration (us), short kernel name of the last N dispatches.
usage: timeline.py <dir> [N]"""
import csv, glob, re, sys
d = sys.argv[1]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 200
rows = []
for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    rows += list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
rows = rows[-n:]
t0 = int(rows[0]["Start_Timestamp"])
prev_end = t0
for r in rows:
    m = re.search(r"(walk_\w+|mlp_\w+|rerank_\w+|knn_\w+|\w+_kernel)", r["Kernel_Name"])
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print("%10.1f  dur %8.1f  gap %7.1f  %s" % ((s - t0) / 1e3, (e - s) / 1e3, (s - prev_end) / 1e3, m.group(1) if m else r["Kernel_Name"][:40]))
    prev_end = max(prev_end, e)
