#!/bin/bash
# A/B of library builds on ONE box (box-to-box variance on the pool is a few percent: variants are only comparable inside
# one gpurun call).  Build the variants on the dev box:  python -m path_tracer_amd.build --variant <name> [-DFLAG ...]
# (into probe_libs/<name>/, the product libraries are never touched), then on the GPU box:
#   tools/ab.sh [-w workload] [-i integrator] [-r rounds] <name> [<name> ...]      ("base" = the product libraries)
# Runs bench.py for every variant, round-robin `rounds` times, and prints Msamples/s, Mrays/s and the traversal launch time.
WL=sponza_1080p; INTEG=reference; ROUNDS=2; STEPS=8
while getopts "w:i:r:s:" o; do case $o in w) WL=$OPTARG;; i) INTEG=$OPTARG;; r) ROUNDS=$OPTARG;; s) STEPS=$OPTARG;; esac; done
shift $((OPTIND - 1))
for r in $(seq $ROUNDS); do
  for v in "$@"; do
    if [ "$v" = base ]; then unset PTX_LIB_DIR; else export PTX_LIB_DIR=$PWD/probe_libs/$v; fi
    echo -n "$v round $r: "
    timeout -k 10 300 python3 bench.py --workload $WL --integrator $INTEG --steps $STEPS --warmup 2 --cpu-baseline-spp 0 \
      | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print(round(d['value']/1e6,1), 'Msamples/s', d['mrays_per_s'], 'Mrays/s  extend', r['avg_launch_ms'], 'ms  shade', r['second_kernel']['ms_per_step'], 'ms/step  nodes/ray', r['nodes_per_ray'], 'tris/ray', r['tris_per_ray'])" || exit 1
  done
done
