"""Summaries of one tools/profile_round.sh run: kernel-trace summary (to compare with bench.py's HIP-event average),
PMC sums per kernel, HBM traffic per ray (traffic.json) and VALU utilisation of the traversal / shading kernels (valu.json)."""
import glob, json, os, subprocess, sys
out, tag = sys.argv[1], sys.argv[2]
here = os.path.dirname(os.path.abspath(__file__))
trace = sorted(glob.glob(out + "/trace/**/*_kernel_trace.csv", recursive=True))[0]
ts = json.loads(subprocess.check_output([sys.executable, os.path.join(here, "trace_summary.py"), trace, "2", "16"]))
json.dump(ts, open(out + "/%s_kernel_trace_summary.json" % tag, "w"), indent=1)
stats = sorted(glob.glob(out + "/trace/**/*_kernel_stats.csv", recursive=True))[0]
open(out + "/%s_kernel_stats.csv" % tag, "w").write(open(stats).read())
pmc = json.loads(subprocess.check_output([sys.executable, os.path.join(here, "pmc_summary.py"), out]))
json.dump(pmc, open(out + "/%s_pmc_summary.json" % tag, "w"), indent=1)
EXT, SHD = "k_extend<false, false>", "k_shade"
# rays of the 2-spp PMC workload: printed by tools/prof_run.py into every pass log
rays = None
for f in glob.glob(out + "/pmc_*.log"):
    for line in open(f):
        if line.startswith("rays "):
            rays = int(line.split()[1])
e = pmc[EXT]
fetch = e["FETCH_SIZE"]["sum"] * 1024 * 2
write = e["WRITE_SIZE"]["sum"] * 1024
traffic = {"workload": "sponza_1080p",
           "source": "profiles/%s_pmc_summary.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes, tools/prof_run.py 2 spp = %d rays over %d k_extend launches)" % (tag, rays, e["FETCH_SIZE"]["dispatches"]),
           "fetch_bytes_per_ray": fetch / rays, "write_bytes_per_ray": write / rays, "hbm_bytes_per_ray": (fetch + write) / rays,
           "tcc_hit_rate": e["TCC_HIT_sum"]["sum"] / max(1.0, e["TCC_HIT_sum"]["sum"] + e["TCC_MISS_sum"]["sum"]),
           "corrections": "FETCH_SIZE KiB x1024 x2 (gfx950 counts 128-B requests as 64 B for 16-B-per-lane loads; divergent gathers are outside the calibrated pattern, so treat as an upper-bound estimate); WRITE_SIZE KiB x1024"}
json.dump(traffic, open(out + "/traffic.json", "w"), indent=1)
vp = json.loads(subprocess.check_output([sys.executable, os.path.join(here, "pmc_summary.py"), out + "/valu"]))
valu = {"workload": "sponza_1080p",
        "source": "rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU, tools/prof_run.py 16 (one 16-spp batch, the bench's launch size), round tag %s" % tag,
        "definitions": "cycles = GRBM_GUI_ACTIVE / 8 XCDs; valu_busy = SQ_ACTIVE_INST_VALU x 4 / (1024 SIMDs x cycles): fraction of SIMD issue slots spent on VALU; residency = SQ_WAVE_CYCLES x 4 / (waves_per_simd x 1024 x cycles); lane_util = SQ_THREAD_CYCLES_VALU / (64 x SQ_ACTIVE_INST_VALU)",
        "kernels": {}}
for k, waves in ((EXT, 7), (SHD, 8)):   # resident waves per SIMD: 66-70 VGPRs -> 7 (pinned by amdgpu_waves_per_eu); 56 VGPRs, 512-thread blocks -> 8
    v = {c: d["sum"] for c, d in vp[k].items()}
    cyc = v["GRBM_GUI_ACTIVE"] / 8.0
    valu["kernels"][k] = {"counters": v, "cycles": cyc, "valu_busy": v["SQ_ACTIVE_INST_VALU"] * 4 / (1024 * cyc),
                          "residency": v["SQ_WAVE_CYCLES"] * 4 / (waves * 1024 * cyc),
                          "lane_util": v["SQ_THREAD_CYCLES_VALU"] / (64 * v["SQ_ACTIVE_INST_VALU"]), "waves_per_simd": waves}
json.dump(valu, open(out + "/valu.json", "w"), indent=1)
print(json.dumps({"trace_timed_region": ts.get("k_extend_timed_region", {}).get("avg_ms"), "hbm_bytes_per_ray": traffic["hbm_bytes_per_ray"],
                  "valu": {k: {x: round(y, 4) for x, y in d.items() if x != "counters"} for k, d in valu["kernels"].items()}}))
