#!/bin/bash
# Developer experiment: chunks in flight (PTX_FRAMES_IN_FLIGHT) x chunk size for both integrators, one box, two rounds.
for r in 1 2; do for cfg in "2 67108864 32" "3 67108864 32" "4 67108864 32" "4 33554432 32" "3 44739242 32"; do read -r NF TP SPP <<< "$cfg"
  for w in "--integrator reference" "--integrator lit"; do
    echo -n "round $r in-flight $NF paths $TP spp $SPP $w: "
    PTX_FRAMES_IN_FLIGHT=$NF PTX_TARGET_PATHS=$TP timeout -k 10 300 python3 bench.py $w --spp-per-step $SPP --steps 4 --warmup 1 --cpu-baseline-spp 0 | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value']/1e6,1), d['mrays_per_s'])"
  done
done; done
