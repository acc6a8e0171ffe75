#!/bin/bash
# usage: tools/pmc.sh <tag>  -- SQ + cache PMC passes over tools/prof_run.py (2 spp Sponza 1080p)
TAG=$1
OUT=gpurun_out/pmc_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
python3 -c "import sys; sys.path.insert(0,'.'); from path_tracer_amd import standins; standins.ensure_standins('assets')"
for pass in "SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_INSTS_SALU" "SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SMEM SQ_INST_CYCLES_VMEM_RD SQ_INSTS_VALU" "FETCH_SIZE" "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum" "GRBM_GUI_ACTIVE TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum"; do
  name=$(echo $pass | tr ' ' '_' | cut -c1-40)
  timeout -s KILL 120 rocprofv3 --pmc $pass --kernel-trace --output-format csv -d $OUT/pmc_$name -- python3 tools/prof_run.py 2 > $OUT/pmc_$name.log 2>&1 || echo "pass $name failed" >> $OUT/errors.log
done
python3 tools/pmc_summary.py $OUT > $OUT/summary.json
python3 - <<PY
import json
r=json.load(open("$OUT/summary.json"))
for k,v in r.items():
    if k.startswith("k_extend") or k.startswith("k_shade"):
        print(k)
        for c,d in sorted(v.items()): print("   %-28s %.5g (%d)"%(c,d["sum"],d["dispatches"]))
PY
cat $OUT/errors.log 2>/dev/null
