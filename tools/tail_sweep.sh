#!/bin/bash
# Sweep of the fused tail's hand-over rule (run on the GPU box): PTX_TAIL_RAYS x PTX_TAIL_SHIFT -> batch Msamples/s of four workloads, 1-spp frame ms of two
one() { python bench.py "$@" --cpu-baseline-spp 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value']/1e6,1), end=' ')"; }
for t in ${1:-65536}; do for sh in ${2:-31 8 7 6 5}; do
echo -n "tail_rays=$t shift=$sh: "
for w in lion_boulder_1080p spheres_720p cornell_512 sponza_1080p; do PTX_TAIL_RAYS=$t PTX_TAIL_SHIFT=$sh one --workload $w; done
for w in lion_boulder_1080p cornell_512; do PTX_TAIL_RAYS=$t PTX_TAIL_SHIFT=$sh python tools/frame_time.py $w 128 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['plain']['ms_per_frame'], end=' ')"; done
echo
done; done
