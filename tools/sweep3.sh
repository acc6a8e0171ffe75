#!/bin/bash
# usage: SPP=16 tools/sweep3.sh "A=1 B=2" ...  (Sponza probe once per env set, SPP samples per batch)
for v in "$@"; do
  echo "== $v"
  env $v timeout -k 10 120 python -u tools/gpu_probe.py --scene ${SCENE:-sponza} --spp ${SPP:-16} --reps 2 2>&1 | grep "profiling=0" || exit 1
done
