"""ptx_reduce_accumulation's communicator cache and call sequence, executed on ONE GPU against a call-recording stand-in
for librccl (tests/native/rccl_stub.cpp, loaded through PTX_RCCL_LIB).  RCCL itself refuses two ranks on one device and a
builder's box has one GPU, so without this the N > 1 branch of path_tracer_amd/csrc/ptx_rccl.cpp first runs in the
driver's 8-GPU job.  What the stub cannot show is RCCL's own behaviour over xGMI: that stays "unverified on hardware"
(DESIGN.md section 6).  Runs in a child process: the library keeps the collective library it loaded for the life of the
process."""
import os
import subprocess
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


def _child_real_one_rank():
    """One context, PTX_RCCL_SINGLE_RANK set: the real librccl, loaded by dlopen, driven through the prototypes and enum
    values csrc/ptx_rccl.cpp declares by hand -- ncclCommInitAll for [0], a group of one ncclReduce (fp32 sum, in place),
    ncclCommDestroy with the context.  The image must come back unchanged (mean of one mean), twice (cached communicator)."""
    import numpy as np
    import torch  # noqa: F401
    sys.path.insert(0, os.path.dirname(HERE))
    sys.path.insert(0, HERE)
    from helpers import cornell, gpu_render, upload
    from oracle import pyoracle as oracle
    from path_tracer_amd import capi
    cam = oracle.camera_matrices([0, 1.0, 3.9], oracle.yaw_quat(0), np.deg2rad(np.float32(40)), 1.0, 0.1, 1000.0)
    a = capi.Context(0)
    upload(a, cornell(), oracle)
    want = gpu_render(a, oracle, cam, 96, 96, 4, params=capi.default_params(max_depth=6))["albedo"]
    for _ in range(2):
        capi.reduce_accumulation([a], root=0)
        got = a.read_image(capi.IMAGE_ALBEDO)
        assert np.array_equal(got.view(np.uint32), want.view(np.uint32))
        assert torch.cuda.current_device() == 0
    a.close()
    b = capi.Context(0)                      # a fresh context after the communicator went with the last one
    upload(b, cornell(), oracle)
    want = gpu_render(b, oracle, cam, 96, 96, 2, params=capi.default_params(max_depth=6))["albedo"]
    capi.reduce_accumulation([b], root=0)
    assert np.array_equal(b.read_image(capi.IMAGE_ALBEDO).view(np.uint32), want.view(np.uint32))
    b.close()


def _child(with_stub):
    import ctypes as C

    import numpy as np
    import torch  # noqa: F401  (first: its HIP runtime is the one the process shares, as in conftest.py)
    sys.path.insert(0, os.path.dirname(HERE))
    sys.path.insert(0, HERE)
    from helpers import RCCL_STUB, SEED, cornell, gpu_render, upload
    from oracle import pyoracle as oracle
    from path_tracer_amd import capi

    ing = cornell()
    cam = oracle.camera_matrices([0, 1.0, 3.9], oracle.yaw_quat(0), np.deg2rad(np.float32(40)), 1.0, 0.1, 1000.0)

    def make():
        c = capi.Context(0)
        upload(c, ing, oracle)
        return c

    def parts(ctxs, spp=4):
        return [gpu_render(c, oracle, cam, 96, 96, spp, params=capi.default_params(max_depth=6, rng_sample_offset=spp * g))["albedo"]
                for g, c in enumerate(ctxs)]

    a, b = make(), make()
    if not with_stub:
        parts([a, b])
        with pytest.raises(capi.PtxError) as e:
            capi.reduce_accumulation([a, b], root=0)
        assert e.value.code == capi.PTX_ERR_INVALID and "two contexts on one device" in str(e.value)
        assert torch.cuda.current_device() == 0
        return

    stub = C.CDLL(RCCL_STUB)

    def counts():
        out = (C.c_int * 6)()
        stub.ptx_rccl_stub_counts(out)
        return dict(zip(("init", "destroy", "group_start", "group_end", "reduce", "live"), list(out)))

    for round_, root in enumerate((0, 1, 0)):
        p = parts([a, b])
        capi.reduce_accumulation([a, b], root=root)
        assert torch.cuda.current_device() == 0
        got = (a, b)[root].read_image(capi.IMAGE_ALBEDO)
        want = (p[0][..., :3] + p[1][..., :3]) * np.float32(0.5)            # the stub sums in rank order in fp32, then x 1/n
        assert np.array_equal(got[..., :3].view(np.uint32), want.view(np.uint32)), round_
        assert np.all(got[..., 3] == 1.0)
        k = counts()
        # ONE ncclCommInitAll for the device list [0, 0], reused by every later reduce; one group of two ncclReduce per call
        assert (k["init"], k["live"], k["destroy"]) == (1, 2, 0), k
        assert (k["group_start"], k["group_end"], k["reduce"]) == (round_ + 1, round_ + 1, 2 * (round_ + 1)), k
    # mismatched extents: refused before any collective call, device untouched
    b.resize(64, 64)
    with pytest.raises(capi.PtxError) as e:
        capi.reduce_accumulation([a, b], root=0)
    assert e.value.code == capi.PTX_ERR_INVALID and counts()["reduce"] == 6
    # the communicators live as long as a context is alive on each of their devices
    c3 = make()
    a.close()
    assert counts()["destroy"] == 0
    b.close(); c3.close()
    k = counts()
    assert (k["destroy"], k["live"]) == (2, 0), k
    # fresh contexts: a fresh ncclCommInitAll
    a, b = make(), make()
    p = parts([a, b])
    capi.reduce_accumulation([a, b], root=0)
    k = counts()
    assert (k["init"], k["live"]) == (2, 2), k
    got = a.read_image(capi.IMAGE_ALBEDO)
    assert np.array_equal(got[..., :3].view(np.uint32), ((p[0][..., :3] + p[1][..., :3]) * np.float32(0.5)).view(np.uint32))
    a.close(); b.close()
    assert counts()["live"] == 0


@pytest.mark.gpu
@pytest.mark.parametrize("with_stub", [True, False])
def test_reduce_accumulation_communicator_cache_against_stub(with_stub):
    sys.path.insert(0, HERE)
    from helpers import RCCL_STUB, build_rccl_stub
    build_rccl_stub()
    env = dict(os.environ)
    env.pop("PTX_RCCL_LIB", None)
    if with_stub:
        env["PTX_RCCL_LIB"] = RCCL_STUB
    p = subprocess.run([sys.executable, os.path.abspath(__file__), "stub" if with_stub else "nostub"], env=env,
                       capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-4000:]
    assert "rccl stub check ok" in p.stdout


@pytest.mark.gpu
def test_reduce_accumulation_one_rank_through_real_rccl():
    """What the stand-in cannot show, as far as one GPU goes: librccl itself behind ptx_reduce_accumulation (one rank)."""
    env = dict(os.environ)
    env.pop("PTX_RCCL_LIB", None)
    env["PTX_RCCL_SINGLE_RANK"] = "1"
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    p = subprocess.run([sys.executable, os.path.abspath(__file__), "real1"], env=env, capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-4000:]
    assert "rccl stub check ok" in p.stdout


if __name__ == "__main__":
    if sys.argv[1] == "real1":
        _child_real_one_rank()
    else:
        _child(sys.argv[1] == "stub")
    print("rccl stub check ok")
