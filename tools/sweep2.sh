#!/bin/bash
# usage: tools/sweep2.sh "A=1 B=2" "A=3 B=4" ...  (runs the Sponza probe once per env set)
for v in "$@"; do
  echo "== $v"
  env $v timeout -k 10 120 python -u tools/gpu_probe.py --scene ${SCENE:-sponza} --spp 8 --reps 2 2>&1 | grep "profiling=0" || exit 1
done
