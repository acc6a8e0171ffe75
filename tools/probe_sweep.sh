#!/bin/bash
# Developer experiment: which resource does k_extend lean on?  Builds the library with 40 extra instructions of ONE kind
# per node visit (PTX_PROBE = 1 full-rate vector, 2 half-rate vector, 3 scalar, 4 s_nop) into probe_libs/probe<N>/
# (git-ignored; the product libraries are never touched) and benches each through PTX_LIB_DIR (results are
# wrong-by-design only in speed: the probe's output is folded into a dead store).  Run the build part on the dev box
# (hipcc), the bench part on the GPU box:  tools/probe_sweep.sh build ; gpurun -- tools/probe_sweep.sh run
set -e
if [ "$1" = build ]; then
  for p in 1 2 3 4; do python -m path_tracer_amd.build --variant probe$p -DPTX_PROBE=$p > /dev/null; done
else
  exec tools/ab.sh -r 1 base probe1 probe2 probe3 probe4
fi
