#!/bin/bash
# Per-dispatch kernel trace of a short bench run (depth 1): every launch of one batch with its start and duration.
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/trace_${1:-x}
mkdir -p $OUT
rm -rf /tmp/p_tr
timeout -k 10 400 rocprofv3 --kernel-trace --kernel-include-regex "ah::" --output-format csv -d /tmp/p_tr -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-latency --depth 1 ${BENCH_ARGS:-} > $OUT/run.log 2>&1
python3 - /tmp/p_tr/*/*kernel_trace.csv <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# last batch = from the last threshold launch on
idx = max(i for i, r in enumerate(rows) if "threshold" in r["Kernel_Name"])
t0 = int(rows[idx]["Start_Timestamp"])
for r in rows[idx:]:
    name = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("ah::", "")[:40]
    print("%-42s start %9.1f us  dur %8.1f us  grid %s" % (name, (int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, r.get("Grid_Size", "")))
PY
