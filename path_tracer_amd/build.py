"""Builds path_tracer_amd/libptx_amd.so (HIP kernels + C ABI) for gfx950 with hipcc.

hipcc cross-compiles without a GPU.  The library is built in-tree so that it travels with the
repository snapshot to the GPU box.
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libptx_amd.so")
SOURCES = ["ptx_api.cpp", "ptx_kernels.hip", "ptx_refit.hip", "bvh_build.cpp", "ptx_rccl.cpp"]
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
# -ffp-contract=off: shading / triangle-test arithmetic must round exactly like the CPU oracle's.
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-fast-math", "-fno-slp-vectorize", "-fPIC",
         "-Wall", "-Wno-unused-function", "-x", "hip"] + os.environ.get("PTX_EXTRA_FLAGS", "").split()


HOST_DIR = os.path.join(HERE, "host")
HOST_LIB = os.path.join(HERE, "libpt_host.so")
HOST_SOURCES = ["pth_capi.cpp", "pth_renderer.cpp", "pth_scene.cpp", "pth_image.cpp"]
CXX = os.environ.get("CXX", "g++")


def build_host(force=False, verbose=False):
    """The C++ host mirror (Renderer / Scene / glTF / image codecs): plain g++, talks to the
    kernels only through the C ABI of include/ptx.h."""
    deps = [os.path.join(HOST_DIR, f) for f in os.listdir(HOST_DIR)] + [os.path.join(HERE, "..", "include", "ptx.h"), __file__]
    if (not force and os.path.exists(HOST_LIB) and os.path.exists(LIB)
            and all(os.path.getmtime(d) <= os.path.getmtime(HOST_LIB) for d in deps)):
        return HOST_LIB
    cmd = [CXX, "-O2", "-std=c++17", "-Wall", "-Wextra", "-ffp-contract=off", "-fPIC", "-shared", "-o", HOST_LIB]
    cmd += [os.path.join(HOST_DIR, f) for f in HOST_SOURCES]
    cmd += ["-L" + HERE, "-lptx_amd", "-Wl,-rpath,$ORIGIN", "-lz"]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return HOST_LIB


CLI = os.path.join(HERE, "ptx-render")


def build_cli(force=False, verbose=False):
    """The headless driver (host/ptx_render_main.cpp) over libpt_host.so + libptx_amd.so."""
    src = os.path.join(HOST_DIR, "ptx_render_main.cpp")
    if (not force and os.path.exists(CLI) and os.path.exists(HOST_LIB)
            and os.path.getmtime(CLI) >= max(os.path.getmtime(src), os.path.getmtime(HOST_LIB))):
        return CLI
    cmd = [CXX, "-O2", "-std=c++17", "-Wall", "-Wextra", "-o", CLI, src, "-L" + HERE, "-lpt_host", "-lptx_amd",
           "-Wl,-rpath,$ORIGIN"]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return CLI


def csrc_hash():
    """sha256 (first 16 hex digits) over the kernel / ABI sources: stamped into profiles/traffic.json and valu.json by
    tools/profile_round_summary.py and recomputed by bench.py, which drops counter-derived side fields that were
    measured on other sources."""
    import hashlib
    h = hashlib.sha256()
    files = sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC)) + [os.path.join(HERE, "..", "include", "ptx.h")]
    for f in files:
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


def needs_build():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)] + [os.path.join(HERE, "..", "include", "ptx.h"), __file__]
    return any(os.path.getmtime(d) > t for d in deps)


def build_library(force=False, verbose=False):
    if not force and not needs_build():
        return LIB
    objs = []
    procs = []
    os.makedirs(os.path.join(HERE, "_build"), exist_ok=True)
    for src in SOURCES:
        obj = os.path.join(HERE, "_build", src + ".o")
        cmd = [HIPCC] + FLAGS + ["-c", os.path.join(CSRC, src), "-o", obj]
        if verbose:
            print(" ".join(cmd))
        procs.append((cmd, subprocess.Popen(cmd)))
        objs.append(obj)
    for cmd, p in procs:
        if p.wait() != 0:
            raise RuntimeError("hipcc failed: " + " ".join(cmd))
    link = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs + ["-ldl", "-lpthread"]
    if verbose:
        print(" ".join(link))
    subprocess.check_call(link)
    return LIB


def build_variant(name, extra_flags):
    """A developer build of both libraries with extra compiler flags into probe_libs/<name>/ (git-ignored; travels to the
    GPU box), loaded with PTX_LIB_DIR=probe_libs/<name>.  The product libraries are left alone."""
    root = os.path.dirname(HERE)
    out = os.path.join(root, "probe_libs", name)
    bdir = os.path.join(HERE, "_build", "variant_" + name)
    os.makedirs(out, exist_ok=True)
    os.makedirs(bdir, exist_ok=True)
    procs, objs = [], []
    for src in SOURCES:
        obj = os.path.join(bdir, src + ".o")
        cmd = [HIPCC] + FLAGS + list(extra_flags) + ["-c", os.path.join(CSRC, src), "-o", obj]
        procs.append((cmd, subprocess.Popen(cmd)))
        objs.append(obj)
    for cmd, p in procs:
        if p.wait() != 0:
            raise RuntimeError("hipcc failed: " + " ".join(cmd))
    lib = os.path.join(out, "libptx_amd.so")
    subprocess.check_call([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib] + objs + ["-ldl", "-lpthread"])
    cmd = [CXX, "-O2", "-std=c++17", "-Wall", "-Wextra", "-ffp-contract=off", "-fPIC", "-shared", "-o", os.path.join(out, "libpt_host.so")]
    cmd += [os.path.join(HOST_DIR, f) for f in HOST_SOURCES] + ["-L" + out, "-lptx_amd", "-Wl,-rpath,$ORIGIN", "-lz"]
    subprocess.check_call(cmd)
    return out


if __name__ == "__main__":
    if len(sys.argv) >= 3 and sys.argv[1] == "--variant":
        print(build_variant(sys.argv[2], sys.argv[3:]))
        sys.exit(0)
    build_library(force="--force" in sys.argv, verbose=True)
    build_host(force="--force" in sys.argv, verbose=True)
    build_cli(force="--force" in sys.argv, verbose=True)
    print(LIB, HOST_LIB, CLI)
