#!/bin/bash
# Are 1-spp frames of a light scene bound by kernels or by launches?  Kernel trace of tools/frame_time.py: busy time vs span.
WL=${1:-lion_boulder_1080p}
export TMPDIR=/tmp
rm -rf gpurun_out/gaps
timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/gaps -- python3 tools/frame_time.py $WL 64 > gpurun_out/gaps.json 2> gpurun_out/gaps.log || exit 1
python3 - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/gaps/*/*_kernel_trace.csv")[0]
rows = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(f))), key=lambda r: r[0])
# the last 64 x 50 launches or so: take the final third of the trace
rows = rows[len(rows) * 2 // 3:]
span = rows[-1][1] - rows[0][0]
busy = 0; cur_s, cur_e = rows[0][0], rows[0][1]
for s, e, _ in rows[1:]:
    if s > cur_e: busy += cur_e - cur_s; cur_s, cur_e = s, e
    else: cur_e = max(cur_e, e)
busy += cur_e - cur_s
dur = sorted(e - s for s, e, _ in rows)
print("launches", len(rows), "span_ms", round(span / 1e6, 3), "gpu_busy_ms(union)", round(busy / 1e6, 3), "idle_frac", round(1 - busy / span, 3),
      "median_kernel_us", dur[len(dur) // 2] / 1e3, "per_launch_span_us", round(span / len(rows) / 1e3, 2))
PY
python3 - <<'PY'
import csv, glob, re, collections
f = glob.glob("gpurun_out/gaps/*/*_kernel_trace.csv")[0]
rows = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Grid_Size", "")) for r in csv.DictReader(open(f))), key=lambda r: r[0])
rows = rows[len(rows) * 2 // 3:]
by = collections.defaultdict(list)
for s, e, n, g in rows:
    m = re.search(r"(k_\w+(<[\w, ]+>)?)", n)
    by[(m.group(1) if m else n[:30], g)].append((e - s) / 1e3)
for k, v in sorted(by.items(), key=lambda kv: -sum(kv[1])):
    v.sort()
    print("%-28s grid %-9s n %4d  total %8.1f us  median %6.1f  min %6.1f  max %7.1f" % (k[0], k[1], len(v), sum(v), v[len(v) // 2], v[0], v[-1]))
PY
tail -1 gpurun_out/gaps.json | cut -c1-200
