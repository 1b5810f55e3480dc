"""CPU tests of the boundary: libdlco.so loads and exports every symbol include/dlco.h declares,
fails loudly without a GPU (no CPU fallback), and the host-side pair indexing of the product
(csrc/pair_index.hpp) is bit-exact against the oracle.  No compute calls without a GPU."""
import ctypes as C
import os
import re
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol(dlco):
    lib = dlco.load()
    names = dlco.exported_symbols()
    assert len(names) >= 35 and "dlco_step" in names and "dlco_grad_rda" in names
    missing = [n for n in names if not hasattr(lib, n)]
    assert not missing, missing
    assert lib.dlco_version().decode().startswith("dlco-mi355x")


def test_header_cites_reference_lines():
    text = open(os.path.join(ROOT, "include", "dlco.h")).read()
    assert len(re.findall(r"src/pj-learn\.cpp:\d+", text)) >= 10
    assert "src/misc.cpp:266-333" in text and "src/kernelop-opencv.cu:49-80" in text


def test_cfg_defaults_are_the_reference_constants(dlco):
    lib = dlco.load()
    cfg = dlco.Cfg()
    lib.dlco_cfg_default(C.byref(cfg))
    # src/pj-learn.cpp:89-93,225
    assert cfg.B == 200 and abs(cfg.mu - 0.001) < 1e-9 and abs(cfg.gamma - 0.5) < 1e-9 and cfg.seed == 2215
    assert cfg.world == 1 and cfg.rank == 0


def test_no_cpu_fallback_without_gpu(dlco):
    """On a machine without a usable gfx950 device the context must refuse to exist."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present: covered by the gpu-marked suite")
    with pytest.raises(dlco.DlcoError) as e:
        dlco.Context(64, 100, B=4)
    assert e.value.code in (dlco.ERR_NODEVICE, dlco.ERR_HIP)


def test_invalid_configs_are_rejected_before_touching_the_device(dlco):
    lib = dlco.load()
    for F, N, B, world, rank in ((0, 100, 4, 1, 0), (-8, 100, 4, 1, 0), (32, 1, 4, 1, 0), (32, 100, 0, 1, 0), (32, 100, 5, 2, 0), (32, 100, 4, 2, 2)):
        cfg = dlco.Cfg()
        lib.dlco_cfg_default(C.byref(cfg))
        cfg.F, cfg.N, cfg.B, cfg.world, cfg.rank = F, N, B, world, rank
        h = C.c_void_p()
        assert lib.dlco_ctx_create(C.byref(h), C.byref(cfg)) == dlco.ERR_INVALID
        assert h.value is None and lib.dlco_last_error(None)
    assert lib.dlco_ctx_create(None, None) == dlco.ERR_INVALID


@pytest.fixture(scope="module")
def shim(tmp_path_factory):
    out = tmp_path_factory.mktemp("shim") / "libhostshim.so"
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-shared", "-fPIC", "-o", str(out),
                           os.path.join(ROOT, "tests", "shim", "host_logic_shim.cpp")])
    L = C.CDLL(str(out))
    i32p, u8p = C.POINTER(C.c_int32), C.POINTER(C.c_uint8)
    L.shim_build_index.argtypes = [u8p, C.c_int, i32p, C.POINTER(C.c_int), C.POINTER(C.c_int), i32p, C.POINTER(C.c_int), C.POINTER(C.c_int)]
    L.shim_sample.argtypes = [C.c_uint64, C.c_int, C.c_int, C.c_int, C.c_int, i32p, i32p]
    L.shim_split.argtypes = [C.c_uint64]
    L.shim_rng_next.argtypes = [C.POINTER(C.c_uint64)]
    L.shim_rng_next.restype = C.c_uint32
    L.shim_rda_coeffs.argtypes = [C.c_uint32, C.c_uint32, C.POINTER(C.c_float), C.POINTER(C.c_float)]
    return L


@pytest.mark.parametrize("N,mode", [(500000, "alt"), (4099, "ragged"), (3, "alt"), (50, "allpos")])
def test_product_pair_index_bit_exact_vs_oracle(shim, ref, N, mode):
    rng = np.random.default_rng(N)
    if mode == "alt":
        lab = (np.arange(N) % 2 == 0).astype(np.uint8)
    elif mode == "allpos":
        lab = np.ones(N, np.uint8)
    else:
        lab = (rng.random(N) < 0.3).astype(np.uint8)
        lab[rng.integers(0, N, 60)] = 9
    pos, neg = np.empty(N, np.int32), np.empty(N, np.int32)
    a, b, c_, d = C.c_int(), C.c_int(), C.c_int(), C.c_int()
    i32p, u8p = C.POINTER(C.c_int32), C.POINTER(C.c_uint8)
    shim.shim_build_index(lab.ctypes.data_as(u8p), N, pos.ctypes.data_as(i32p), C.byref(a), C.byref(b),
                          neg.ctypes.data_as(i32p), C.byref(c_), C.byref(d))
    rp, rn = ref.build_index(lab)
    assert np.array_equal(pos[:a.value], rp) and np.array_equal(neg[:c_.value], rn)
    assert b.value == ref.split(rp.size) and d.value == ref.split(rn.size)


def test_product_sampler_and_rng_bit_exact_vs_oracle(shim, ref):
    B, steps = 200, 25
    ip, ineg = np.empty(B * steps, np.int32), np.empty(B * steps, np.int32)
    i32p = C.POINTER(C.c_int32)
    shim.shim_sample(2215, 200000, 199999, B, steps, ip.ctypes.data_as(i32p), ineg.ctypes.data_as(i32p))
    r = ref.Rng(2215)
    for s in range(steps):
        a, b = r.sample(200000, 199999, B)
        assert np.array_equal(ip[s * B:(s + 1) * B], a) and np.array_equal(ineg[s * B:(s + 1) * B], b)
    st = C.c_uint64(0xFFFFFFFF)
    r2 = ref.Rng(0xFFFFFFFF)
    assert [shim.shim_rng_next(C.byref(st)) for _ in range(100)] == [r2.next() for _ in range(100)]
    assert [shim.shim_split(n) for n in (250000, 1999, 7, 0)] == [ref.split(n) for n in (250000, 1999, 7, 0)]


def test_rda_coefficients_do_not_wrap_at_multi_gpu_batch_sizes(shim, ref):
    """src/pj-learn.cpp:422 forms szBatch*szBatch*(t+1) in 32-bit unsigned arithmetic; with the
    GLOBAL batch of an 8-GPU run (B = 1600) that product passes 2^32 at t = 1677.  Product and
    oracle form it in 64 bits: same bits as the reference wherever the reference does not wrap,
    monotonically decreasing beyond."""
    def coeffs(B, t):
        a, b = C.c_float(), C.c_float()
        shim.shim_rda_coeffs(B, t, C.byref(a), C.byref(b))
        return a.value, b.value
    # the reference's own range: identical to its 32-bit expression
    for B, t in ((200, 0), (200, 1), (200, 49999), (200, 50000), (50, 7), (1, 0)):
        a, b = coeffs(B, t)
        assert a == float(np.float32(1.0) / np.float32(np.uint32(B * B * (t + 1))))
        assert b == float(np.float32(np.float64(t) / np.float64(t + 1)))
    # across the old wrap points: strictly decreasing, never inf / nan
    for B, ts in ((1600, range(1670, 1685)), (400, range(26840, 26850)), (294, range(49680, 49700))):
        al = [coeffs(B, t)[0] for t in ts]
        assert all(np.isfinite(al)) and all(x > y for x, y in zip(al, al[1:]))
        assert abs(al[0] * (B * B * (ts[0] + 1)) - 1.0) < 1e-6
    # the oracle applies the same coefficients
    df = np.full((2, 2), 3.0, np.float32)
    dl = np.full((2, 2), 5.0e9, np.float32)
    for B, t in ((1600, 1676), (1600, 1677), (1600, 1678), (200, 11)):
        a, b = coeffs(B, t)
        want = df * np.float32(b) + dl * np.float32(a)
        assert np.array_equal(ref.rda_update(df, dl, t, B), want)
