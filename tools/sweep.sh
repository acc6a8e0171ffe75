#!/bin/bash
# usage: tools/sweep.sh VAR v1 v2 ...   (runs the Sponza probe once per value)
VAR=$1; shift
for v in "$@"; do
  echo "== $VAR=$v"
  env $VAR=$v timeout -k 10 120 python -u tools/gpu_probe.py --scene sponza --spp 8 --reps 2 2>&1 | grep "profiling=0" || exit 1
done
