"""Prints the kernels of the timed bench step (between the last two k_raygen launches) with start offsets and durations."""
import csv, glob, re, sys
f = glob.glob(sys.argv[1] + "/**/*_kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
out = []
for r in rows:
    m = re.search(r"(k_\w+(<[\w, ]+>)?)", r["Kernel_Name"])
    out.append((m.group(1) if m else r["Kernel_Name"][:24], (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, int(r["Start_Timestamp"])))
idx = [i for i, o in enumerate(out) if o[0] == "k_raygen"]
i0, i1 = idx[-2], idx[-1]
t0 = out[i0][2]
for k, d, s in out[i0:i1]:
    print("%-28s start %9.1f us  dur %9.1f us" % (k, (s - t0) / 1e3, d))
