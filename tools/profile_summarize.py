#!/usr/bin/env python3
"""Turns gpurun_out/prof_<tag>/ (tools/profile.sh) into the committed summaries under profiles/:
   <tag>_kernel_stats.csv, <tag>_pmc.csv and profiles/hbm_traffic.json (read by bench.py for roofline.traffic)."""
import csv, json, os, sys, collections

tag = sys.argv[1]
frames = int(sys.argv[2]) if len(sys.argv) > 2 else 256
src = os.path.join("gpurun_out", "prof_" + tag)
# the profiled command's own line (bench.py --profile-run): which build, how many batches in flight, frames per launch
run = {}
for log in ("fetch.log", "trace.log"):
    try:
        for line in open(os.path.join(src, log)):
            if line.startswith('{"profile_run"'):
                run = json.loads(line)
    except OSError:
        pass
    if run:
        break
frames = run.get("frames_per_launch", frames)
os.makedirs("profiles", exist_ok=True)
rows = list(csv.reader(open(os.path.join(src, "kernel_stats.csv"))))
rows = [rows[0]] + [r for r in rows[1:] if "ah::" in r[0]]   # this library's kernels only (torch's frame generator is noise)
with open(os.path.join("profiles", tag + "_kernel_stats.csv"), "w", newline="") as f:
    csv.writer(f).writerows(rows)
if not os.path.exists(os.path.join(src, "fetch_counters.csv")):   # PASSES=trace: kernel stats only
    for r in rows[:14]:
        print(r[0][:60], r[1:4])
    sys.exit(0)
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for name in ("fetch_counters.csv", "write_counters.csv"):
    for r in csv.DictReader(open(os.path.join(src, name))):
        acc[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {"tag": tag, "frames_per_launch": frames, "build": run.get("build"), "batches_in_flight": run.get("batches_in_flight"),
       "command": "rocprofv3 --pmc FETCH_SIZE | WRITE_SIZE (separate passes) -- python3 bench.py --profile-run --steps %s --warmup %s" % (run.get("steps"), run.get("warmup")),
       "unit_note": "FETCH_SIZE / WRITE_SIZE are reported in KiB by rocprofv3 (MI355X_MICROARCH.md: hbm_bytes = (FETCH_SIZE + WRITE_SIZE) * 1024); "
       "the guide's x2 correction for FETCH_SIZE applies to 16 B/lane streaming reads: applied to the wide threshold kernel only; "
       "raw and x2-corrected values are both listed", "kernels": {}}
with open(os.path.join("profiles", tag + "_pmc.csv"), "w", newline="") as f:
    w = csv.writer(f)
    w.writerow(["kernel", "dispatches", "FETCH_SIZE_KiB_per_batch", "WRITE_SIZE_KiB_per_batch", "hbm_MB_per_batch_raw", "hbm_MB_per_batch_fetch_x2"])
    # batches in the profiled run = dispatches of the threshold kernel; a kernel launched twice per batch (the two walker
    # passes) is summed per batch so that the numbers line up with bench.py's per-batch event intervals
    nbatch = max(len(c.get("FETCH_SIZE", [])) for k, c in acc.items() if "threshold" in k)
    for k, c in acc.items():
        fe = sum(c.get("FETCH_SIZE", [0])) / nbatch
        wr = sum(c.get("WRITE_SIZE", [0])) / nbatch
        w.writerow([k, len(c.get("FETCH_SIZE", [])), round(fe, 1), round(wr, 1), round((fe + wr) * 1024 / 1e6, 2), round((2 * fe + wr) * 1024 / 1e6, 2)])
        short = k.split("(")[0].replace("void ", "").replace("ah::", "")
        short = short.split("<")[0]
        # MI355X_MICROARCH.md: FETCH_SIZE reports half the bytes of a 16-byte-per-lane streaming read; the wide threshold kernel
        # reads that way (its raw FETCH_SIZE is 0.58 x W*H per frame), every other kernel here reads 4 or 8 bytes per lane
        wide = short in ("threshold_wide_kernel", "threshold_eo_kernel")
        short = {"threshold_strip_kernel": "threshold_kernel", "threshold_wide_kernel": "threshold_kernel", "threshold_eo_kernel": "threshold_kernel",
                 "candidates_sparse_kernel": "candidates_kernel"}.get(short, short)
        out["kernels"][short] = {"fetch_bytes_per_launch": fe * 1024 * (2 if wide else 1), "write_bytes_per_launch": wr * 1024,
                                 "fetch_correction": "x2 (16 B / lane streaming read)" if wide else "none",
                                 "hbm_bytes_per_frame": ((2 if wide else 1) * fe + wr) * 1024 / frames,
                                 "hbm_bytes_per_frame_raw": (fe + wr) * 1024 / frames, "hbm_bytes_per_frame_fetch_x2": (2 * fe + wr) * 1024 / frames}
json.dump(out, open(os.path.join("profiles", "hbm_traffic.json"), "w"), indent=1)
print(open(os.path.join("profiles", tag + "_pmc.csv")).read())
for r in rows[:12]:
    print(r[0][:60], r[1:4])
