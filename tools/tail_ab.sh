#!/bin/bash
# Fused tail launch on / off (run on the GPU box): batch throughput and 1-spp frame times
one() { python bench.py "$@" --cpu-baseline-spp 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value']/1e6,1), d['roofline']['extend_ms'], d['roofline']['shade_ms'])"; }
for w in sponza_1080p lion_boulder_1080p spheres_720p cornell_512; do
echo -n "$w tail: "; one --workload $w
echo -n "$w off:  "; PTX_TAIL_RAYS=0 one --workload $w
done
for w in lion_boulder_1080p cornell_512 sponza_1080p; do
echo -n "frames $w tail: "; python tools/frame_time.py $w 128 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['plain']['ms_per_frame'], d['moving_camera']['ms_per_frame'])"
echo -n "frames $w off:  "; PTX_TAIL_RAYS=0 python tools/frame_time.py $w 128 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['plain']['ms_per_frame'], d['moving_camera']['ms_per_frame'])"
done
