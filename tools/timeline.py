#!/usr/bin/env python3
"""Text timeline of a rocprofv3 kernel trace: every kernel of a window in the middle of the run, one line each, with the queue it ran on,
its start relative to the window and its duration (us). tools/timeline.py kernel_trace.csv [window_ms]"""
import csv
import sys

rows = []
for r in csv.DictReader(open(sys.argv[1])):
    name = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("ah::", "").split("<")[0]
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), name, r.get("Queue_Id", "?"), r.get("Stream_Id", "?")))
rows.sort()
win = float(sys.argv[2]) * 1e6 if len(sys.argv) > 2 else 6e6
t0 = rows[len(rows) // 2][0]
for s, e, k, q, st in rows:
    if t0 <= s < t0 + win:
        print("%9.1f %8.1f  q%-3s s%-3s %s" % ((s - t0) / 1e3, (e - s) / 1e3, q, st, k))
