#!/bin/bash
# Per-dispatch kernel trace of tools/kbench.py (one batch at a time, no result checks): every launch of the last batch.
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf /tmp/p_kb
timeout -k 10 300 rocprofv3 --kernel-trace --kernel-include-regex "ah::" --output-format csv -d /tmp/p_kb -- python3 tools/kbench.py --frames ${FRAMES:-1024} --steps 3 > /tmp/p_kb.log 2>&1
python3 - /tmp/p_kb/*/*kernel_trace.csv <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = max(i for i, r in enumerate(rows) if "threshold" in r["Kernel_Name"])
t0 = int(rows[idx]["Start_Timestamp"])
for r in rows[idx:]:
    name = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("ah::", "")[:40]
    print("%-42s start %9.1f us  dur %8.1f us" % (name, (int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3))
PY
