"""Joins tools/calib/valu_calib's own JSON (s_memtime cycles per instruction) with the rocprofv3 --pmc pass over the
same program (counter value per dispatch) into one calibration table: cycles per wave64 VALU instruction per SIMD at
1 .. 8 waves per SIMD, and what each SQ counter reads per instruction.   summarize.py <calib.json> <pmc dir> > out.json"""
import collections, csv, glob, json, re, sys
calib = json.load(open(sys.argv[1]))
rows = collections.defaultdict(dict)        # dispatch id -> {counter: value, kernel, grid}
for f in glob.glob(sys.argv[2] + "/**/*_counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        d = rows[int(r["Dispatch_Id"])]
        d[r["Counter_Name"]] = d.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
        d["kernel"] = re.sub(r"\(.*", "", r["Kernel_Name"])
        d["grid"] = int(r["Grid_Size"])
# the program launches every (instruction, waves) pair twice (warm-up, timed): keep the second dispatch of each pair
seen = collections.defaultdict(list)
for did in sorted(rows):
    d = rows[did]
    seen[(d["kernel"], d["grid"])].append(d)
out = []
for res in calib["results"]:
    key = ("calib_" + res["instr"], res["blocks"] * 256)
    if key not in seen:
        out.append(res); continue
    d = seen[key][-1]
    n_inst = res["valu_instr_per_wave"] * res["blocks"] * 4
    cyc = d.get("GRBM_GUI_ACTIVE", 0.0) / 8.0
    r = dict(res)
    r["pmc"] = {k: v for k, v in d.items() if k not in ("kernel", "grid")}
    r["per_instr"] = {k: round(v / n_inst, 4) for k, v in d.items() if k.startswith("SQ_") and k != "SQ_WAVES"}
    if cyc:
        simds = calib["cus"] * 4
        r["kernel_cycles_grbm"] = cyc
        r["cycles_per_instr_per_simd_grbm"] = round(cyc * simds / n_inst, 4)     # includes launch ramp and tail
        if "SQ_ACTIVE_INST_VALU" in d:
            r["active_inst_valu_x4_over_simd_cycles"] = round(d["SQ_ACTIVE_INST_VALU"] * 4 / (simds * cyc), 4)
        if "SQ_INSTS_VALU" in d:
            r["insts_valu_over_simd_cycles"] = round(d["SQ_INSTS_VALU"] / (simds * cyc), 4)
    out.append(r)
calib["results"] = out
print(json.dumps(calib, indent=1))
