#!/usr/bin/env python3
"""summarise a rocprofv3 --kernel-trace csv as a timeline: every kernel between the (k+1)-th last and the last launch of the
anchor kernel (default sd_sift), i.e. the last k bench steps when nothing else runs after them
   python tools/trace_gaps.py <kernel_trace.csv> [k_steps] [anchor]"""
import csv
import sys
rows = list(csv.DictReader(open(sys.argv[1])))
k = int(sys.argv[2]) if len(sys.argv) > 2 else 2
anchor = sys.argv[3] if len(sys.argv) > 3 else "sd_sift"
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0][-40:], "%s/%s" % (r.get("Stream_Id", "?"), r.get("Queue_Id", "?"))) for r in rows))
idx = [i for i, e in enumerate(ev) if anchor in e[2]]
lo = idx[-(k + 1)] if len(idx) > k else 0
ev = ev[lo:idx[-1] + 12]
t0 = ev[0][0]
for s, e, name, q in ev:
    print("%9.1f us  +%7.1f us  ->%9.1f  s/q %-8s %s" % ((s - t0) / 1e3, (e - s) / 1e3, (e - t0) / 1e3, q, name))
