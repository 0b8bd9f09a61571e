#!/usr/bin/env python3
"""summarise a rocprofv3 --kernel-trace csv: per-kernel totals and, for the last `steps` bench steps, the timeline of one step
   python tools/trace_gaps.py <kernel_trace.csv> [n_last_kernels]"""
import csv
import sys
rows = list(csv.DictReader(open(sys.argv[1])))
n_last = int(sys.argv[2]) if len(sys.argv) > 2 else 80
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][-40:], r.get("Stream_Id", r.get("Queue_Id", "?"))) for r in rows))
ev = ev[-n_last:]
t0 = ev[0][0]
for s, e, k, q in ev:
    print("%9.1f us  +%7.1f us  q%-6s %s" % ((s - t0) / 1e3, (e - s) / 1e3, q, k))
