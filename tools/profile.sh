#!/bin/bash
# rocprofv3 passes for one round; run on the GPU box via gpurun.  Usage: tools/profile.sh <tag>
set -e
TAG=${1:-r01}
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
python3 -c "import sys; sys.path.insert(0,'.'); from path_tracer_amd import standins; standins.ensure_standins('assets')"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 bench.py --steps 2 --warmup 1 --cpu-baseline-spp 0 > $OUT/bench_under_rocprof.json 2> $OUT/trace.log
for pass in "FETCH_SIZE" "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum" "SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_INSTS_SALU" "SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SMEM SQ_LEVEL_WAVES SQ_INSTS_VALU"; do
  name=$(echo $pass | tr ' ' '_' | cut -c1-40)
  timeout -s KILL 120 rocprofv3 --pmc $pass --output-format csv -d $OUT/pmc_$name -- python3 tools/prof_run.py 2 > $OUT/pmc_$name.log 2>&1 || echo "pass $name failed" >> $OUT/errors.log
done
find $OUT -name '*.csv' | head -50
