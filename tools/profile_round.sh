#!/bin/bash
# One round's rocprofv3 evidence, run on the GPU box via gpurun:  tools/profile_round.sh <tag>
#   1. kernel trace + stats of `bench.py --steps 2 --warmup 1` (the bench command, shortened)
#   2. PMC passes, one rocprofv3 run each (never combined with tracing domains other than the kernel trace):
#      FETCH_SIZE | WRITE_SIZE + TCC hit/miss | SQ issue counters | SQ memory counters   on tools/prof_run.py 2
#      SQ VALU-utilisation counters, VALU instruction mix by type                       on tools/prof_run.py 32
#   3. summaries -> gpurun_out/prof_<tag>/*.json (copy the ones to keep into profiles/)
TAG=${1:-r01}
INTEG=${2:-reference}      # "lit": the same passes over bench.py --integrator lit (k_shade_lit, the any-hit traversal)
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
python3 -c "import sys; sys.path.insert(0,'.'); from path_tracer_amd import standins; standins.ensure_standins('assets')"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 bench.py --steps 2 --warmup 1 --cpu-baseline-spp 0 --integrator $INTEG > $OUT/bench_under_rocprof.json 2> $OUT/trace.log || { echo "trace failed"; exit 1; }
for pass in "FETCH_SIZE" "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum" "SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_INSTS_SALU" "SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SMEM SQ_LEVEL_WAVES"; do
  name=$(echo $pass | tr ' ' '_' | cut -c1-40)
  timeout -s KILL 120 rocprofv3 --pmc $pass --kernel-trace --output-format csv -d $OUT/pmc_$name -- python3 tools/prof_run.py 2 $INTEG > $OUT/pmc_$name.log 2>&1 || { echo "pass $name failed" >> $OUT/errors.log; exit 1; }
done
mkdir -p $OUT/valu
timeout -s KILL 200 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY --kernel-trace --output-format csv -d $OUT/valu/pmc_a -- python3 tools/prof_run.py 32 $INTEG > $OUT/valu/a.log 2>&1 || { echo "valu pass failed" >> $OUT/errors.log; exit 1; }
timeout -s KILL 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_INT64 --kernel-trace --output-format csv -d $OUT/valu/pmc_b -- python3 tools/prof_run.py 32 $INTEG > $OUT/valu/b.log 2>&1 || { echo "valu mix pass failed" >> $OUT/errors.log; exit 1; }
python3 tools/profile_round_summary.py $OUT $TAG $INTEG
