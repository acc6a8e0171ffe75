#!/bin/bash
# Developer experiment: k_extend's persistent grid capped below the resident-wave count (8192 = 8 waves per SIMD), so
# that the other chunk's shading launches find free wave slots while a traversal kernel runs.  Run on the GPU box.
for w in 8192 7168 6144 5120 4096; do
  echo -n "PTX_EXTEND_WAVES=$w: "
  PTX_EXTEND_WAVES=$w timeout -k 10 200 python3 bench.py --steps 8 --warmup 2 --cpu-baseline-spp 0 | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value']/1e6,1), 'Msamples/s; k_extend', d['roofline']['avg_launch_ms'], 'ms; step', d['ms_per_step'])" || exit 1
done
