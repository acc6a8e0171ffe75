"""Kernel timeline of the last bench steps with both chunks in flight (rocprofv3 kernel trace CSV directory as argv[1]):
start offset, duration, queue of every launch from the third-last k_raygen on -- who runs beside whom."""
import csv, glob, re, sys
f = glob.glob(sys.argv[1] + "/**/*_kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
out = []
for r in rows:
    m = re.search(r"(k_\w+)", r["Kernel_Name"])
    out.append((m.group(1) if m else r["Kernel_Name"][:24], int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Queue_Id", "?"), r.get("Grid_Size", "?")))
idx = [i for i, o in enumerate(out) if o[0] == "k_raygen"]
i0 = idx[-3] if len(idx) >= 3 else idx[0]
t0 = out[i0][1]
for k, s, e, q, g in out[i0:]:
    print("%-18s q%-3s grid %-9s start %9.1f  end %9.1f  dur %8.1f us" % (k, q, g, (s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3))
