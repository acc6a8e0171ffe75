set -e
cd $GRAFT_REPO_ROOT
python3 -c "import sys; sys.path.insert(0,'.'); from path_tracer_amd import standins; standins.ensure_standins('assets')"
rm -f gpurun_out/soak_*
timeout -k 10 200 path_tracer_amd/ptx-render assets/sponza/Sponza.gltf --lit --res 1920x1080 --spp 512 --camera -8,1.6,0 --yaw -90 --out gpurun_out/soak_lit
timeout -k 10 200 path_tracer_amd/ptx-render assets/sponza/Sponza.gltf --no-ref-quirks --res 1920x1080 --spp 128 --camera -8,1.6,0 --yaw -90 --out gpurun_out/soak_nrq
timeout -k 10 200 path_tracer_amd/ptx-render assets/lion_head_2k/lion_head_2k.gltf@2,0,0 assets/boulder_01_2k/boulder_01_2k.gltf@-2,0,0 --lit --res 1920x1080 --spp 512 --out gpurun_out/soak_lion
python3 - <<'PY'
import numpy as np, glob
for f in sorted(glob.glob("gpurun_out/soak_*_albedo.pfm")):
    with open(f, "rb") as fh:
        assert fh.readline().strip() == b"PF"; w, h = map(int, fh.readline().split()); fh.readline()
        a = np.frombuffer(fh.read(), "<f4").reshape(h, w, 3)
    print(f, "finite", bool(np.isfinite(a).all()), "mean", a.mean(axis=(0, 1)).round(4), "max", float(a.max()))
PY
rm -f gpurun_out/soak_*.pfm
