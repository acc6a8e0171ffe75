#!/bin/bash
# Round-2 re-sweep of the traversal scheduling knobs on the headline workload after the node test was rebalanced
# (run on the GPU box): PTX_TRI_MIN / PTX_REFILL / PTX_FLUSH_MIN / PTX_CHUNK.  One bench run each, 8 steps.
one() { timeout -k 10 120 python bench.py --steps 8 --warmup 2 --cpu-baseline-spp 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value']/1e6,1), d['roofline']['avg_launch_ms'])"; }
echo -n "default: "; one
for pair in "48 8" "48 16" "48 20" "40 12" "32 12" "56 12" "64 12" "40 16" "32 16"; do set -- $pair; echo -n "tri_min=$1 refill=$2: "; PTX_TRI_MIN=$1 PTX_REFILL=$2 one; done
for f in 4 16 24; do echo -n "flush_min=$f: "; PTX_FLUSH_MIN=$f one; done
for c in 64 256; do echo -n "chunk=$c: "; PTX_CHUNK=$c one; done
echo -n "default again: "; one
