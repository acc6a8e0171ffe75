#!/bin/bash
# usage: tools/variant.sh NAME "-DFOO=1 ..."  -> path_tracer_amd/_variants/libptx_NAME.so (experiments; use with PTX_LIB=)
set -e
cd "$(dirname "$0")/.."
mkdir -p path_tracer_amd/_variants /tmp/ptxvar_$1
objs=""
for f in ptx_api.cpp ptx_kernels.hip bvh_build.cpp ptx_rccl.cpp; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-fast-math -fno-slp-vectorize -fPIC -Wno-unused-function $2 -x hip -c path_tracer_amd/csrc/$f -o /tmp/ptxvar_$1/$f.o &
  objs="$objs /tmp/ptxvar_$1/$f.o"
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o path_tracer_amd/_variants/libptx_$1.so $objs -ldl -lpthread
echo path_tracer_amd/_variants/libptx_$1.so
