#!/bin/bash
# Developer experiment: how much could ANY better culling / traversal order buy k_extend?  A PTX_LAB_SEED build runs every
# closest-hit launch twice over the same queue: an ordinary pass, then a pass in which each ray starts with the distance of
# its closest hit as its limit (perfect knowledge: every box behind the hit is culled from the first visit on).  The second
# pass finds the same hits.  Kernel trace: launch times alternate ordinary / seeded; instrumented step: visits of both.
#   dev box:  python -m path_tracer_amd.build --variant seed -DPTX_LAB_SEED
#   GPU box:  tools/lab/seed_bound.sh
export TMPDIR=/tmp PTX_LIB_DIR=$PWD/probe_libs/seed PTX_LAB_SEED=1
OUT=gpurun_out/seed_bound; mkdir -p $OUT
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -- python3 tools/prof_run.py 16 > $OUT/run.log 2>&1 || { tail -5 $OUT/run.log; exit 1; }
python3 - $OUT <<'PY'
import csv, glob, json, sys
f = sorted(glob.glob(sys.argv[1] + "/trace/**/*_kernel_trace.csv", recursive=True))[0]
rows = [r for r in csv.DictReader(open(f)) if "k_extend" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6 for r in rows]
d = d[-32:]                                   # the 16-spp batch: 16 bounces x (ordinary, seeded)
plain, seeded = d[0::2], d[1::2]
print(json.dumps({"launch_pairs": len(plain), "ordinary_ms": round(sum(plain), 3), "seeded_ms": round(sum(seeded), 3),
                  "seeded_over_ordinary": round(sum(seeded) / sum(plain), 4),
                  "per_bounce_ordinary_ms": [round(x, 3) for x in plain], "per_bounce_seeded_ms": [round(x, 3) for x in seeded]}))
PY
# visits: the instrumented step counts both passes against one pass's rays -> reported = ordinary + seeded
timeout -k 10 300 python3 bench.py --steps 1 --warmup 1 --cpu-baseline-spp 0 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); r = d['roofline']
print(json.dumps({'nodes_per_ray_both_passes': r['nodes_per_ray'], 'tris_per_ray_both_passes': r['tris_per_ray']}))"
