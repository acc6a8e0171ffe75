#!/bin/bash
# Vector-L1 (TCP) / texture-addresser view of the traversal kernel: hit rate, average latency of an L1 access and of an
# L1 -> L2 read, stall cycles.  Two rocprofv3 --pmc passes over tools/prof_run.py 2 (never combined with other trace
# domains); summary printed as JSON.   gpurun -- tools/l1_probe.sh <tag> [integrator]
TAG=${1:-l1}; INTEG=${2:-reference}
OUT=gpurun_out/l1_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
timeout -s KILL 120 rocprofv3 --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCP_LATENCY_sum GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/pmc_a -- python3 tools/prof_run.py 2 $INTEG > $OUT/pmc_a.log 2>&1 || { echo "pass a failed"; tail -5 $OUT/pmc_a.log; exit 1; }
timeout -s KILL 120 rocprofv3 --pmc TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_GATE_EN1_sum TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TD_TD_BUSY_sum TD_TC_STALL_sum --kernel-trace --output-format csv -d $OUT/pmc_b -- python3 tools/prof_run.py 2 $INTEG > $OUT/pmc_b.log 2>&1 || { echo "pass b failed"; tail -5 $OUT/pmc_b.log; exit 1; }
python3 tools/pmc_summary.py $OUT > $OUT/summary.json && python3 - $OUT/summary.json <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
for k, v in d.items():
    if not k.startswith("k_"): continue
    c = {n: x["sum"] for n, x in v.items()}
    acc, req = c.get("TCP_TOTAL_CACHE_ACCESSES_sum", 0), c.get("TCP_TCC_READ_REQ_sum", 0)
    if not acc: continue
    print(json.dumps({"kernel": k, "l1_accesses": acc, "l1_to_l2_reads": req, "l1_hit_rate": round(1 - req / acc, 4),
                      "avg_l1_latency_cycles": round(c.get("TCP_TCP_LATENCY_sum", 0) / acc, 1),
                      "avg_l2_read_latency_cycles": round(c.get("TCP_TCC_READ_REQ_LATENCY_sum", 0) / max(1, req), 1),
                      "tcp_busy_cycles": c.get("TCP_GATE_EN1_sum"), "tcp_pending_stall": c.get("TCP_PENDING_STALL_CYCLES_sum"),
                      "tcr_stall": c.get("TCP_TCR_TCP_STALL_CYCLES_sum"), "tagconflict_stall": c.get("TCP_READ_TAGCONFLICT_STALL_CYCLES_sum"),
                      "ta_busy": c.get("TA_TA_BUSY_sum"), "ta_stalled_by_tc": c.get("TA_ADDR_STALLED_BY_TC_CYCLES_sum"),
                      "td_busy": c.get("TD_TD_BUSY_sum"), "td_tc_stall": c.get("TD_TC_STALL_sum"), "gui_active": c.get("GRBM_GUI_ACTIVE")}))
PY
