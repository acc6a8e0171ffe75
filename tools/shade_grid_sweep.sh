#!/bin/bash
# k_shade time by rocprofv3 kernel trace for several PTX_SHADE_GRID caps (run on the GPU box): tools/shade_grid_sweep.sh [workload]
WL=${1:-sponza_1080p}
export TMPDIR=/tmp
for g in 1000000 32768 16384 8192; do
  export PTX_SHADE_GRID=$g
  rm -rf gpurun_out/sgs_$g
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/sgs_$g -- python3 bench.py --workload $WL --steps 2 --warmup 1 --cpu-baseline-spp 0 > gpurun_out/sgs_$g.json 2> gpurun_out/sgs_$g.log || exit 1
  python3 - "$g" <<'PY'
import csv, glob, json, re, sys
g = sys.argv[1]
f = glob.glob("gpurun_out/sgs_%s/*/*_kernel_stats.csv" % g)[0]
rows = {}
for r in csv.DictReader(open(f)):
    m = re.search(r"(k_\w+(<[\w, ]+>)?)", r["Name"])
    if m:
        rows[m.group(1)] = r
d = json.loads(open("gpurun_out/sgs_%s.json" % g).read())
print(g, "Msamples/s", round(d["value"] / 1e6, 1), " ".join("%s %.2f ms" % (k, float(v["TotalDurationNs"]) / 1e6) for k, v in rows.items() if k.startswith(("k_shade", "k_extend<false"))))
PY
done
