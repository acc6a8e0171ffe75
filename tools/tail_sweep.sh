one() { python bench.py "$@" --cpu-baseline-spp 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value']/1e6,1), end=' ')"; }
for t in 4096 16384 65536 262144 1048576; do
echo -n "tail_rays=$t: "
for w in lion_boulder_1080p spheres_720p cornell_512 sponza_1080p; do PTX_TAIL_RAYS=$t one --workload $w; done
for w in lion_boulder_1080p cornell_512; do PTX_TAIL_RAYS=$t python tools/frame_time.py $w 128 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['plain']['ms_per_frame'], end=' ')"; done
echo
done
