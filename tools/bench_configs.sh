#!/bin/bash
# BASELINE.json's fixed-budget configs on the GPUs at hand (here: one): config 3 (Sponza stand-in 1920x1080, 1024 spp; reference
# integrator and lit = "textured PBR + NEE") and config 5 (3840x2160, 4096 spp), time to the finished image.
#   gpurun -- tools/bench_configs.sh <tag>
TAG=${1:-r03}
timeout -k 10 300 python3 bench.py --total-spp 1024 --cpu-baseline-spp 0 > gpurun_out/${TAG}_bench_config3_1024spp.json 2>/dev/null || echo "config 3 failed"
timeout -k 10 300 python3 bench.py --total-spp 1024 --integrator lit --cpu-baseline-spp 0 > gpurun_out/${TAG}_bench_config3_1024spp_lit.json 2>/dev/null || echo "config 3 lit failed"
timeout -k 10 900 python3 bench.py --workload sponza_4k --total-spp 4096 --cpu-baseline-spp 0 > gpurun_out/${TAG}_bench_config5_4k_4096spp.json 2>/dev/null || echo "config 5 failed"
python3 - $TAG <<'PY'
import json, glob, sys
for f in sorted(glob.glob("gpurun_out/%s_bench_config*.json" % sys.argv[1])):
    d = json.loads(open(f).read().strip().splitlines()[-1])
    print(f.split("_bench_")[1][:-5], round(d["value"] / 1e6, 1), "Msamples/s", d["mrays_per_s"], "Mrays/s  time to image", round(d["time_to_image_ms"] / 1e3, 2), "s  spp", d["total_spp"], d["scaling"])
PY
