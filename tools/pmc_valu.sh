#!/bin/bash
# usage: tools/pmc_valu.sh <tag>  -- one SQ counter pass at the bench's launch size (16 spp Sponza 1080p) -> VALU issue utilisation
TAG=$1
OUT=gpurun_out/pmc_valu_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
python3 -c "import sys; sys.path.insert(0,'.'); from path_tracer_amd import standins; standins.ensure_standins('assets')"
timeout -s KILL 200 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU --kernel-trace --output-format csv -d $OUT/pmc_a -- python3 tools/prof_run.py 16 > $OUT/a.log 2>&1 || echo "pass failed"
python3 tools/pmc_summary.py $OUT > $OUT/summary.json
python3 - <<PY
import json
r = json.load(open("$OUT/summary.json"))
out = {"workload": "sponza_1080p", "source": "rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU, tools/prof_run.py 16 (one 16-spp batch, the bench's launch size)",
       "definitions": "cycles = GRBM_GUI_ACTIVE / 8 XCDs; valu_busy = SQ_ACTIVE_INST_VALU x 4 / (1024 SIMDs x cycles): fraction of SIMD issue slots spent on VALU; residency = SQ_WAVE_CYCLES x 4 / (waves_per_simd x 1024 x cycles); lane_util = SQ_THREAD_CYCLES_VALU / (64 x SQ_ACTIVE_INST_VALU)", "kernels": {}}
for k, waves in (("k_extend<false>", 7), ("k_shade", 6)):
    v = {c: d["sum"] for c, d in r[k].items()}
    cyc = v["GRBM_GUI_ACTIVE"] / 8.0
    out["kernels"][k] = {"counters": v, "cycles": cyc, "valu_busy": v["SQ_ACTIVE_INST_VALU"] * 4 / (1024 * cyc),
                         "residency": v["SQ_WAVE_CYCLES"] * 4 / (waves * 1024 * cyc),
                         "lane_util": v["SQ_THREAD_CYCLES_VALU"] / (64 * v["SQ_ACTIVE_INST_VALU"]), "waves_per_simd": waves}
json.dump(out, open("$OUT/valu.json", "w"), indent=1)
print(json.dumps({k: {x: round(y, 4) for x, y in d.items() if x != "counters"} for k, d in out["kernels"].items()}))
PY
