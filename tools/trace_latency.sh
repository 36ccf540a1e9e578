#!/bin/bash
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf /tmp/p_lat
timeout -k 10 300 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d /tmp/p_lat -- python3 tools/latency_trace.py ${1:-single} ${2:-} ${3:-} > gpurun_out/lat_run.log 2>&1
python3 - <<'PY'
import csv, glob
k = glob.glob("/tmp/p_lat/*/*kernel_trace.csv")[0]
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void ", "").replace("ah::", "")[:36]) for r in csv.DictReader(open(k))]
try:
    m = glob.glob("/tmp/p_lat/*/*memory_copy_trace.csv")[0]
    rows += [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "COPY " + r.get("Direction", "")) for r in csv.DictReader(open(m))]
except Exception as e:
    print("no copy trace", e)
rows.sort()
# last call: find the last threshold kernel
idx = max(i for i, r in enumerate(rows) if "threshold" in r[2])
start = idx
while start > 0 and rows[start][0] - rows[start - 1][1] < 30000: start -= 1
t0 = rows[start][0]
busy = 0
for s, e, n in rows[start:]:
    print("%-40s start %8.1f us dur %7.1f us" % (n, (s - t0) / 1e3, (e - s) / 1e3))
    busy += e - s
print("span %.1f us, busy %.1f us, %d ops" % ((rows[-1][1] - t0) / 1e3, busy / 1e3, len(rows) - start))
PY
tail -2 gpurun_out/lat_run.log
