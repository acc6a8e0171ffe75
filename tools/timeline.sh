#!/bin/bash
# Kernel timeline of one timed bench step of a workload (rocprofv3 kernel trace): tools/timeline.sh <workload> [spp]
W=${1:-lion_boulder_1080p}
SPP=${2:-16}
OUT=gpurun_out/timeline_$W
mkdir -p $OUT
export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $OUT -- python3 bench.py --workload $W --steps 1 --warmup 1 --spp-per-step $SPP --cpu-baseline-spp 0 > $OUT/bench.json 2> $OUT/log.txt || exit 1
python3 tools/timeline.py $OUT
