"""Summarise a rocprofv3 --kernel-trace CSV of `bench.py`: per kernel the call count / total / average, and for the
traversal kernel the launches of the timed region (the last steps*bounces uninstrumented launches) so that the
figure can be compared with bench.py's own HIP-event average."""
import csv, json, re, sys
trace, steps, bounces = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
rows = list(csv.DictReader(open(trace)))
per = {}
for r in rows:
    m = re.search(r"(k_\w+(<[\w, ]+>)?|__amd\w+)", r["Kernel_Name"])
    k = m.group(1) if m else r["Kernel_Name"]
    per.setdefault(k, []).append((int(r["Start_Timestamp"]), (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6,
                                  int(r["VGPR_Count"]), int(r["LDS_Block_Size"]), int(r["Scratch_Size"])))
out = {"source": trace, "kernels": {}}
for k, v in per.items():
    v.sort()
    d = [x[1] for x in v]
    out["kernels"][k] = {"calls": len(d), "total_ms": round(sum(d), 3), "avg_ms": round(sum(d) / len(d), 4),
                         "vgprs": v[0][2], "lds_bytes": v[0][3], "scratch_bytes": v[0][4]}
ext = [x[1] for x in per.get("k_extend<false, false>", [])]
n = steps * bounces
if len(ext) >= n:
    timed = ext[-n:]
    out["k_extend_timed_region"] = {"launches": n, "avg_ms": round(sum(timed) / n, 4), "total_ms": round(sum(timed), 3),
                                     "per_launch_ms": [round(x, 3) for x in timed],
                                     "note": "the %d launches before them are warm-up and the host mirror's one-sample settle frames" % (len(ext) - n)}
print(json.dumps(out, indent=1))
