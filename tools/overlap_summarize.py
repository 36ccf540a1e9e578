#!/usr/bin/env python3
"""Summary of a rocprofv3 kernel trace of the pipelined bench: the steady-state window (middle half of the threshold
launches), per kernel the launches and mean duration inside it, the time covered by at least one kernel, by a
threshold kernel, and the mean number of kernels running at once."""
import collections
import csv
import sys

rows = []
for r in csv.DictReader(open(sys.argv[1])):
    name = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("ah::", "").split("<")[0]
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), name))
rows.sort()
thr = [r for r in rows if "threshold" in r[2]]
n = len(thr)
lo, hi = thr[n // 4][0], thr[(3 * n) // 4][0]
nb = (3 * n) // 4 - n // 4
win = [(max(s, lo), min(e, hi), k) for s, e, k in rows if e > lo and s < hi]
per = collections.defaultdict(lambda: [0, 0])
for s, e, k in win:
    per[k][0] += 1
    per[k][1] += e - s
wall = hi - lo
print("window %.3f ms, %d batches, %.3f ms per batch" % (wall / 1e6, nb, wall / 1e6 / nb))
print("%-28s %8s %12s %12s" % ("kernel", "launches", "ms/batch", "mean us"))
for k, (c, t) in sorted(per.items(), key=lambda kv: -kv[1][1]):
    print("%-28s %8d %12.3f %12.1f" % (k, c, t / 1e6 / nb, t / 1e3 / c))
tot = sum(t for c, t in per.values())
# coverage
ev = []
for s, e, k in win:
    ev.append((s, 1, k)), ev.append((e, -1, k))
ev.sort()
depth = 0; tdepth = 0; last = lo; cov = 0; tcov = 0; hist = collections.Counter()
for t, d, k in ev:
    if depth > 0:
        cov += t - last
    if tdepth > 0:
        tcov += t - last
    hist[min(depth, 6)] += t - last
    last = t
    depth += d
    if "threshold" in k:
        tdepth += d
print("sum of kernel time %.3f ms per batch; mean kernels at once %.2f; some kernel running %.1f%%; a threshold kernel running %.1f%%"
      % (tot / 1e6 / nb, tot / wall, 100.0 * cov / wall, 100.0 * tcov / wall))
print("time share by number of kernels running: " + ", ".join("%d: %.1f%%" % (d, 100.0 * t / wall) for d, t in sorted(hist.items())))
