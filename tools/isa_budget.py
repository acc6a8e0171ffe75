"""Dynamic instruction budget of the traversal kernel k_extend<false, false>, per phase.

Static half: the kernel is compiled to gfx950 assembly with line tables; every instruction is assigned to a phase through
its `.loc` -- lines of ptx_kernels.hip belong to the latest "[phase: X]" comment above them inside the kernel, lines of
traverse.h to the function they are in (node_eval / load_node: node, tri_pick: append-loop, ray set-up: refill), helper
lines (pt_math.h, the ballot / scan helpers) to the phase of the instruction before them -- and counted by issue class
(tools/isa_cost.py: full-rate / half-rate / transcendental VALU, SALU incl. waits and branches, VMEM, LDS).

Dynamic half: how often each phase runs, from the counters of the instrumented build that bench.py prints as
roofline.wave_counters (one untimed step of the same workload): node / tri-decision / retire once per wave iteration,
append per iteration that queued triangles, append-loop per pass of the per-lane write loop, tri-block per triangle
block, refill per ray-fetch pass.

  python tools/isa_budget.py <bench line .json> > profiles/r03_isa_budget.json        (no GPU needed)"""
import json, os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
sys.path.insert(0, ROOT)
import isa_cost
from path_tracer_amd import build as B

bench = json.load(open(sys.argv[1]))
wc = bench["roofline"]["wave_counters"]
src = os.path.join(B.CSRC, "ptx_kernels.hip")
text_src = open(src).read().splitlines()
k_lo = next(i for i, l in enumerate(text_src) if l.startswith("k_extend(SceneDev sc")) + 1
k_hi = next(i for i, l in enumerate(text_src) if i > k_lo and l.startswith("// ---------------------------------------------------------------- shade")) + 1
markers = [(i + 1, m.group(1)) for i, l in enumerate(text_src) for m in [re.search(r"\[phase: ([\w-]+)\]", l)] if m and k_lo <= i + 1 <= k_hi]
trav = open(os.path.join(B.CSRC, "traverse.h")).read().splitlines()
def fn_range(name):
    a = next(i for i, l in enumerate(trav) if re.search(r"\b%s\(" % name, l) and ("PT_HD" in l or "__device__" in l)) + 1
    b = next(i for i in range(a, len(trav)) if trav[i].startswith("    }") or trav[i].startswith("}")) + 1
    return a, b
TRAV_FN = {"node_eval": "node", "load_node": "node", "tri_pick": "append-loop", "ray_slab_setup_rcp": "refill", "init_prepared": "refill"}
trav_ranges = [(fn_range(n), ph) for n, ph in TRAV_FN.items()]
aw_lo = next(i for i, l in enumerate(text_src) if "uint32_t advance_walk(" in l) + 1
aw_hi = next(i for i in range(aw_lo, len(text_src)) if text_src[i].startswith("}")) + 1

with tempfile.TemporaryDirectory() as td:
    out = os.path.join(td, "k.s")
    flags = [f for f in B.FLAGS if f != "-fPIC"]
    subprocess.check_call([B.HIPCC] + flags + ["-gline-tables-only", "-S", "--cuda-device-only", src, "-o", out], stderr=subprocess.DEVNULL)
    asm = open(out).read().splitlines()
files = {}
for l in asm:
    m = re.match(r'\s*\.file\s+(\d+)\s+(?:"([^"]*)"\s+)?"([^"]+)"', l)
    if m:
        files[int(m.group(1))] = os.path.basename(m.group(3))
start = next(i for i, l in enumerate(asm) if re.match(r"^_ZN3ptx.*k_extendILb0ELb0E.*:", l))
end = next(i for i in range(start, len(asm)) if "s_endpgm" in asm[i])
phase, cur = "prologue", None
per = {}
for l in asm[start:end + 1]:
    m = re.match(r"\s*\.loc\s+(\d+)\s+(\d+)", l)
    if m:
        f, ln = files.get(int(m.group(1)), "?"), int(m.group(2))
        if f == "ptx_kernels.hip":
            if k_lo <= ln <= k_hi:
                before = [p for (ml, p) in markers if ml <= ln]
                phase = before[-1] if before else "prologue"
            elif aw_lo <= ln <= aw_hi:
                phase = "node"
        elif f == "traverse.h":
            for (a, b), ph in trav_ranges:
                if a <= ln <= b:
                    phase = ph
        continue
    c = isa_cost.classify(l)
    if not c:
        continue
    cls = c[0]
    if cls == "other" or (cls == "salu" and c[1] in ("s_waitcnt", "s_nop")):
        cls = "wait_nop"
    d = per.setdefault(phase, {"full": 0, "half": 0, "trans": 0, "salu": 0, "vmem": 0, "lds": 0, "wait_nop": 0})
    d[cls] += 1

execs = {"prologue": wc["waves"], "refill": wc["refill_passes"], "node": wc["iterations"], "append": wc["append_blocks"],
         "append-loop": wc["append_passes"], "tri-decision": wc["iterations"], "tri-block": wc["tri_blocks"], "retire": wc["iterations"]}
rays, iters = wc["rays"], wc["iterations"]
rows, tot = {}, {"valu": 0.0, "salu": 0.0, "mem": 0.0, "wait_nop": 0.0}
for ph, d in per.items():
    n = execs.get(ph, 0)
    valu = d["full"] + d["half"] + d["trans"]
    rows[ph] = {"static": d, "runs_per_wave_iteration": round(n / iters, 4),
                "valu_per_wave_iteration": round(valu * n / iters, 1), "salu_per_wave_iteration": round(d["salu"] * n / iters, 1),
                "vmem_lds_per_wave_iteration": round((d["vmem"] + d["lds"]) * n / iters, 1),
                "wait_nop_per_wave_iteration": round(d["wait_nop"] * n / iters, 1)}
    tot["valu"] += valu * n; tot["salu"] += d["salu"] * n; tot["mem"] += (d["vmem"] + d["lds"]) * n; tot["wait_nop"] += d["wait_nop"] * n
issued = tot["valu"] + tot["salu"] + tot["mem"]
print(json.dumps({
    "kernel": "k_extend<false, false>", "workload": bench["config"]["workload"], "csrc_sha16": B.csrc_hash(),
    "source": "static: hipcc -S -gline-tables-only + the [phase: ...] comments of ptx_kernels.hip; dynamic: roofline.wave_counters of " + os.path.basename(sys.argv[1]),
    "wave_counters": wc,
    "wave_iterations_per_ray_lane": round(iters * 64.0 / rays, 2),
    "phases": rows,
    "per_wave_iteration": {k: round(v / iters, 1) for k, v in tot.items()},
    "issued_per_ray": {"wave_instructions_per_ray": round(issued / rays, 1), "lane_slots_per_ray": round(issued * 64.0 / rays, 0),
                       "valu_lane_slots_per_ray": round(tot["valu"] * 64.0 / rays, 0), "salu_share": round(tot["salu"] / issued, 4),
                       "algorithmic_lane_ops_per_ray": bench["roofline"]["algorithmic_lane_ops_per_ray"],
                       "note": "lane_slots = wave instructions x 64 lanes / rays: what one ray's lane sits through; the algorithmic figure is SURVEY 8d's 112 per node visit + 60 per triangle test"}}, indent=1))
