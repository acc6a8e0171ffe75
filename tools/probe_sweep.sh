#!/bin/bash
# Developer experiment: which resource does k_extend lean on?  Builds the library with 40 extra instructions of ONE kind
# per node visit (PTX_PROBE = 1 full-rate vector, 2 half-rate vector, 3 scalar, 4 s_nop) into probe_libs/ (git-ignored) and
# benches each (results are wrong-by-design only in speed: the probe's output is folded away).  Run the build part on
# the dev box (hipcc), the bench part on the GPU box:  tools/probe_sweep.sh build | tools/probe_sweep.sh run
set -e
if [ "$1" = build ]; then
  mkdir -p probe_libs
  for p in 0 1 2 3 4; do
    if [ $p = 0 ]; then PTX_EXTRA_FLAGS="" python -m path_tracer_amd.build --force > /dev/null 2>&1; else PTX_EXTRA_FLAGS="-DPTX_PROBE=$p" python -m path_tracer_amd.build --force > /dev/null 2>&1; fi
    cp path_tracer_amd/libptx_amd.so probe_libs/libptx_probe$p.so
  done
  PTX_EXTRA_FLAGS="" python -m path_tracer_amd.build --force > /dev/null 2>&1
else
  for p in 0 1 2 3 4; do
    cp probe_libs/libptx_probe$p.so path_tracer_amd/libptx_amd.so
    echo -n "probe $p: "; timeout -k 10 300 python3 bench.py --steps 8 --warmup 2 --cpu-baseline-spp 0 | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value']/1e6,1), d['roofline']['avg_launch_ms'])"
  done
fi
