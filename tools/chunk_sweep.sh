#!/bin/bash
# Ray-fetch chunk sweep (run on the GPU box): tools/chunk_sweep.sh "<PTX_CHUNK values>" "<PTX_CHUNK_GROW values>" workload...
CH=${1:-"64 128 256"}; GR=${2:-"1 2 4"}; shift 2
for w in "$@"; do for c in $CH; do for g in $GR; do
  echo -n "$w chunk=$c grow=$g: "
  PTX_CHUNK=$c PTX_CHUNK_GROW=$g timeout -k 10 200 python bench.py --workload $w --cpu-baseline-spp 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value']/1e6,1))" || exit 1
done; done; done
