#!/bin/bash
# Developer experiment: the traversal kernel's refill / triangle-block thresholds (PTX_REFILL, PTX_TRI_MIN: read once per
# process) on several workloads, round-robin on ONE box.   gpurun -- tools/lab/threshold_ab.sh "14 36" "18 32"
# Round 3, 64 Mi-path chunks: 18 / 32 against the default 14 / 36: Sponza stand-in +0.7 %, rotated +1.5 %, room +0.3 %, lit -0.6 %: left alone.
CFGS=("$@")
for r in 1 2; do for cfg in "${CFGS[@]}"; do read -r REFILL TRIMIN <<< "$cfg"
  for w in "--workload sponza_1080p" "--workload sponza_rot30_1080p" "--workload lion_boulder_room_1080p" "--integrator lit"; do
    echo -n "round $r refill $REFILL trimin $TRIMIN $w: "
    PTX_REFILL=$REFILL PTX_TRI_MIN=$TRIMIN timeout -k 10 300 python3 bench.py $w --steps 4 --warmup 1 --cpu-baseline-spp 0 | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print(round(d['value']/1e6,1), d['mrays_per_s'], r['avg_launch_ms'])"
  done
done; done
