"""Static instruction mix of the HIP kernels by issue class (tools/isa_cost.py's classes, measured by
tools/calib/valu_calib): compiles csrc/ptx_kernels.hip to gfx950 assembly with the library's flags and counts, per
kernel, full-rate / half-rate / transcendental vector instructions, scalar, vector-memory and LDS instructions.
Runs without a GPU.   python tools/isa_mix.py > profiles/r02_isa_mix.json"""
import json, os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
sys.path.insert(0, ROOT)
import isa_cost
from path_tracer_amd import build as B
src = os.path.join(B.CSRC, "ptx_kernels.hip")
with tempfile.TemporaryDirectory() as td:
    out = os.path.join(td, "k.s")
    flags = [f for f in B.FLAGS if f != "-fPIC"]
    subprocess.check_call([B.HIPCC] + flags + ["-S", "--cuda-device-only", src, "-o", out], stderr=subprocess.DEVNULL)
    text = open(out).read().splitlines()
kernels = {}
cur = None
for ln in text:
    m = re.match(r"^(_ZN3ptx\S+):\s", ln)
    if m:
        cur = m.group(1); kernels[cur] = []
    elif cur is not None:
        kernels[cur].append(ln)
        if "s_endpgm" in ln:
            cur = None
res = {}
for name, lines in kernels.items():
    cnt, cyc, hp, both, ops = isa_cost.cost(lines)
    short = subprocess.check_output(["c++filt", name]).decode().strip().replace("ptx::(anonymous namespace)::", "").replace("void ", "").split("(")[0]
    valu = cnt["full"] + cnt["half"] + cnt["trans"]
    res[short] = dict(cnt, valu=valu, half_rate_share=round((cnt["half"] + cnt["trans"]) / max(1, valu), 4),
                      model_cycles_per_pass=cyc)
print(json.dumps({"source": "static counts over each kernel's whole body (every basic block once); classes of tools/isa_cost.py",
                  "flags": " ".join(B.FLAGS), "kernels": res}, indent=1))
