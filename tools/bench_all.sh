#!/bin/bash
# Every bench workload once, lines kept under gpurun_out/<tag>_bench_<workload>.json (copy the keepers into profiles/).
#   gpurun -- tools/bench_all.sh <tag>
TAG=${1:-r03}
for w in sponza_1080p sponza_4k sponza_rot30_1080p lion_boulder_room_1080p; do
  timeout -k 10 400 python3 bench.py --workload $w > gpurun_out/${TAG}_bench_$w.json 2> gpurun_out/${TAG}_bench_$w.err || echo "bench $w failed"
done
for w in cornell_512 spheres_720p lion_boulder_1080p; do
  timeout -k 10 400 python3 bench.py --workload $w --steps 20 > gpurun_out/${TAG}_bench_$w.json 2> gpurun_out/${TAG}_bench_$w.err || echo "bench $w failed"
done
timeout -k 10 400 python3 bench.py --integrator lit > gpurun_out/${TAG}_bench_sponza_1080p_lit.json 2> gpurun_out/${TAG}_bench_sponza_1080p_lit.err || echo "bench lit failed"
timeout -k 10 400 python3 bench.py --total-spp 64 --cpu-baseline-spp 0 > gpurun_out/${TAG}_bench_sponza_1080p_fixed64.json 2> /dev/null || echo "bench fixed failed"
python3 - $TAG <<'PY'
import json, sys, glob
for f in sorted(glob.glob("gpurun_out/%s_bench_*.json" % sys.argv[1])):
    try:
        d = json.load(open(f)); r = d["roofline"]
        print("%-46s %8.1f Msamples/s %8.1f Mrays/s  rays/sample %6.3f  nodes/ray %5.2f tris/ray %5.2f  extend %.3f ms  parity %s  stale %s" % (
            f.split("_bench_")[1][:-5], d["value"] / 1e6, d["mrays_per_s"], d["rays_per_sample"], r["nodes_per_ray"], r["tris_per_ray"], r["avg_launch_ms"],
            (d.get("parity") or {}).get("albedo_bit_exact"), r.get("profile_stale")))
    except Exception as e:
        print(f, "unreadable", e)
PY
