#!/bin/bash
# like sweep3.sh but prints the profiled extend / shade split
for v in "$@"; do
  echo "== $v"
  env $v timeout -k 10 120 python -u tools/gpu_probe.py --scene ${SCENE:-sponza} --spp ${SPP:-16} --reps 2 2>&1 | grep "profiling=" || exit 1
done
