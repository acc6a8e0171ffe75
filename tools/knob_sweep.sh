#!/bin/bash
# Traversal scheduling knobs on the headline workload (run on the GPU box): "PTX_TRI_MIN PTX_REFILL" pairs
one() { python bench.py --steps 8 --cpu-baseline-spp 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value']/1e6,1))"; }
for pair in "64 8" "48 8" "48 12" "40 12" "56 8" "56 12" "64 8"; do set -- $pair; echo -n "tri_min=$1 refill=$2: "; PTX_TRI_MIN=$1 PTX_REFILL=$2 one; done
