"""Summaries of one tools/profile_round.sh run: kernel-trace summary (to compare with bench.py's HIP-event average),
PMC sums per kernel, HBM traffic per ray (traffic.json) and VALU utilisation of the traversal / shading kernels (valu.json)."""
import glob, json, os, subprocess, sys
out, tag = sys.argv[1], sys.argv[2]
integrator = sys.argv[3] if len(sys.argv) > 3 else "reference"
here = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(here))
from path_tracer_amd import build as B
SRC_HASH = B.csrc_hash()
WORKLOAD = "sponza_1080p" + ("+lit" if integrator == "lit" else "")
trace = sorted(glob.glob(out + "/trace/**/*_kernel_trace.csv", recursive=True))[0]
ts = json.loads(subprocess.check_output([sys.executable, os.path.join(here, "trace_summary.py"), trace, "2", "16"]))
json.dump(ts, open(out + "/%s_kernel_trace_summary.json" % tag, "w"), indent=1)
stats = sorted(glob.glob(out + "/trace/**/*_kernel_stats.csv", recursive=True))[0]
open(out + "/%s_kernel_stats.csv" % tag, "w").write(open(stats).read())
pmc = json.loads(subprocess.check_output([sys.executable, os.path.join(here, "pmc_summary.py"), out]))
json.dump(pmc, open(out + "/%s_pmc_summary.json" % tag, "w"), indent=1)
EXT, SHD = "k_extend<false, false>", ("k_shade_lit" if integrator == "lit" else "k_shade")
# rays of the 2-spp PMC workload: printed by tools/prof_run.py into every pass log
rays = None
for f in glob.glob(out + "/pmc_*.log"):
    for line in open(f):
        if line.startswith("rays "):
            rays = int(line.split()[1])
e = pmc[EXT]
fetch = e["FETCH_SIZE"]["sum"] * 1024 * 2
write = e["WRITE_SIZE"]["sum"] * 1024
sh = pmc.get(SHD, {})
shade_hbm = ((sh["FETCH_SIZE"]["sum"] * 1024 * 2 + sh["WRITE_SIZE"]["sum"] * 1024) / rays) if "FETCH_SIZE" in sh and "WRITE_SIZE" in sh else None
traffic = {"workload": WORKLOAD, "csrc_sha16": SRC_HASH,
           "shade_kernel": SHD, "shade_hbm_bytes_per_ray": shade_hbm,
           "source": "profiles/%s_pmc_summary.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes, tools/prof_run.py 2 spp = %d rays over %d k_extend launches)" % (tag, rays, e["FETCH_SIZE"]["dispatches"]),
           "fetch_bytes_per_ray": fetch / rays, "write_bytes_per_ray": write / rays, "hbm_bytes_per_ray": (fetch + write) / rays,
           "tcc_hit_rate": e["TCC_HIT_sum"]["sum"] / max(1.0, e["TCC_HIT_sum"]["sum"] + e["TCC_MISS_sum"]["sum"]),
           "corrections": "FETCH_SIZE KiB x1024 x2 (gfx950 counts 128-B requests as 64 B for 16-B-per-lane loads; divergent gathers are outside the calibrated pattern, so treat as an upper-bound estimate); WRITE_SIZE KiB x1024"}
json.dump(traffic, open(out + "/traffic.json", "w"), indent=1)
vp = json.loads(subprocess.check_output([sys.executable, os.path.join(here, "pmc_summary.py"), out + "/valu"]))
# static instruction mix by issue class (tools/isa_mix.py, made where the library is built) and the measured issue costs
mix_path = os.path.join(os.path.dirname(here), "profiles", "r03_isa_mix.json")
static_mix = json.load(open(mix_path))["kernels"] if os.path.exists(mix_path) else {}
valu = {"workload": WORKLOAD, "csrc_sha16": SRC_HASH,
        "source": "rocprofv3 --pmc, two passes over tools/prof_run.py 32 (one 32-spp batch, the bench's launch size): SQ_WAVES SQ_WAVE_CYCLES "
                  "SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY | SQ_INSTS_VALU "
                  "and its _FMA_F32 _ADD_F32 _MUL_F32 _CVT _INT32 _TRANS_F32 _INT64 breakdown; round tag %s" % tag,
        "definitions": "cycles = GRBM_GUI_ACTIVE / 8 XCDs.  Issue model measured on this chip (profiles/r02_valu_calibration.json): a SIMD has two vector "
                       "pipes; a wave64 instruction holds one for 4 cycles; full-rate instructions (fp32 fma / add / mul, bitwise, integer add, right shifts, "
                       "with VGPR / literal operands) can use either pipe, all others only one; transcendentals hold it for 8.  SQ_ACTIVE_INST_VALU counts 1 "
                       "(transcendental: 2) per instruction in units of 4 cycles, whatever the class -- round 1's `valu_busy` = that x 4 / SIMD cycles "
                       "therefore reads up to 2.0 on full-rate code and is kept only as `active_x4_per_simd_cycle`.  "
                       "two_pipe_busy = SQ_INSTS_VALU x 4 / (2 pipes x 1024 SIMDs x cycles): share of ALL vector issue capacity in use (<= 1).  "
                       "half_rate_pipe_busy = half-rate instructions x 4 / (1024 x cycles) (<= 1): `_static_mix` prices SQ_INSTS_VALU with the "
                       "kernel's static half-rate share (profiles/r03_isa_mix.json), `_lo` counts only conversions and transcendentals (x 2), `_hi` "
                       "counts everything that is not an fp32 fma / add / mul.  residency = SQ_WAVE_CYCLES x 4 / (waves_per_simd x 1024 x cycles); "
                       "lane_util = SQ_THREAD_CYCLES_VALU / (64 x SQ_ACTIVE_INST_VALU); wait_mem / wait_issue = SQ_WAIT_ANY / SQ_WAIT_INST_ANY over SQ_WAVE_CYCLES",
        "kernels": {}}
# resident waves per SIMD: k_extend 64 VGPRs, pinned to 8 by amdgpu_waves_per_eu; k_shade ~100 VGPRs, k_shade_lit ~124, in 512-thread blocks: two blocks per CU = 4
for k, waves in ((EXT, 8), (SHD, 4), ("k_extend<false, true>", 8)):
    if k not in vp:
        continue
    v = {c: d["sum"] for c, d in vp[k].items()}
    cyc = v["GRBM_GUI_ACTIVE"] / 8.0
    simd_cyc = 1024.0 * cyc
    insts = v["SQ_INSTS_VALU"]
    share = static_mix.get(k, {}).get("half_rate_share")
    fast_fp = v.get("SQ_INSTS_VALU_FMA_F32", 0.0) + v.get("SQ_INSTS_VALU_ADD_F32", 0.0) + v.get("SQ_INSTS_VALU_MUL_F32", 0.0)
    e = {"counters": v, "cycles": cyc, "waves_per_simd": waves,
         "insts_per_simd_cycle": insts / simd_cyc,
         "two_pipe_busy": insts * 4.0 / (2.0 * simd_cyc),
         "half_rate_pipe_busy_static_mix": (insts * share * 4.0 / simd_cyc) if share is not None else None,
         "half_rate_pipe_busy_lo": (v.get("SQ_INSTS_VALU_CVT", 0.0) + 2.0 * v.get("SQ_INSTS_VALU_TRANS_F32", 0.0)) * 4.0 / simd_cyc,
         "half_rate_pipe_busy_hi": (insts - fast_fp) * 4.0 / simd_cyc if fast_fp else None,
         "static_half_rate_share": share,
         "active_x4_per_simd_cycle": v["SQ_ACTIVE_INST_VALU"] * 4.0 / simd_cyc,
         "residency": v["SQ_WAVE_CYCLES"] * 4.0 / (waves * simd_cyc),
         "lane_util": v["SQ_THREAD_CYCLES_VALU"] / (64.0 * v["SQ_ACTIVE_INST_VALU"]),
         "wait_mem": v.get("SQ_WAIT_ANY", 0.0) / v["SQ_WAVE_CYCLES"], "wait_issue": v.get("SQ_WAIT_INST_ANY", 0.0) / v["SQ_WAVE_CYCLES"]}
    valu["kernels"][k] = e
json.dump(valu, open(out + "/valu.json", "w"), indent=1)
print(json.dumps({"trace_timed_region": ts.get("k_extend_timed_region", {}).get("avg_ms"), "hbm_bytes_per_ray": traffic["hbm_bytes_per_ray"],
                  "valu": {k: {x: round(y, 4) for x, y in d.items() if x != "counters"} for k, d in valu["kernels"].items()}}))
