#!/bin/bash
# like sweep4.sh, with the traversal statistics line
for v in "$@"; do
  echo "== $v"
  env $v timeout -k 10 160 python -u tools/gpu_probe.py --scene ${SCENE:-sponza} --spp ${SPP:-16} --reps 2 2>&1 | grep -E "profiling=1|nodes/ray|bvh" || exit 1
done
