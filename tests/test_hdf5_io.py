"""Row L1 of the hot-path table: the HDF5 side of the command-line tools (cli/dlco_io.hpp).

The reference reads "Distance" f32 [N,F] and "Label" u8 [N,1] in 128-row hyperslabs
(src/pj-learn.cpp:173-212) from a file its producer writes chunked {128,1} with gzip level 9
(src/comp-uprjdists.cpp:254,289-290), and writes "W" / "A" (src/pj-learn.cpp:592-597).
h5py is not available, so the files are written / read through tests/shim/io_shim.cpp (the
product's own reader + libhdf5 via dlopen)."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from util import golden, relmax, synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_H5 = "/root/reference/workspace/pj-learn/liberty-liberty-0.035-0.250-pr#7-0.0010-0.100-pj.h5"
f32p, u8p = C.POINTER(C.c_float), C.POINTER(C.c_uint8)


@pytest.fixture(scope="module")
def io(tmp_path_factory):
    out = tmp_path_factory.mktemp("ioshim") / "libioshim.so"
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-shared", "-fPIC", "-o", str(out),
                           os.path.join(ROOT, "tests", "shim", "io_shim.cpp"), "-ldl"])
    L = C.CDLL(str(out))
    L.shim_read_f32.argtypes = [C.c_char_p, C.c_char_p, f32p, C.c_size_t, C.POINTER(C.c_size_t)]
    L.shim_write_unproj.argtypes = [C.c_char_p, f32p, u8p, C.c_size_t, C.c_size_t, C.c_size_t, C.c_size_t, C.c_int]
    L.shim_stream_unproj.argtypes = [C.c_char_p, f32p, u8p, C.c_size_t, C.c_size_t, C.c_size_t, C.c_size_t, C.c_int, C.c_size_t]
    L.shim_write_nd.argtypes = [C.c_char_p, C.c_int, C.c_char_p, C.c_int, C.c_void_p, C.c_int, C.POINTER(C.c_size_t)]
    L.shim_read_i32.argtypes = [C.c_char_p, C.c_char_p, C.POINTER(C.c_int32), C.c_size_t, C.POINTER(C.c_size_t)]
    L.shim_read_u8.argtypes = [C.c_char_p, C.c_char_p, u8p, C.c_size_t, C.POINTER(C.c_size_t)]
    if not L.shim_hdf5_available():
        pytest.skip("no libhdf5 >= 1.10 on this machine")
    return L


def read_f32(io, path, name, cap):
    out = np.empty(cap, np.float32)
    sh = (C.c_size_t * 4)()
    nd = io.shim_read_f32(path.encode(), name.encode(), out.ctypes.data_as(f32p), cap, sh)
    assert nd >= 0, "read of %s failed" % name
    shape = tuple(sh[i] for i in range(nd))
    return out[:int(np.prod(shape))].reshape(shape)


def write_unproj(io, path, D, L, chunk=(128, 1), gzip=9):
    D = np.ascontiguousarray(D, np.float32)
    L = np.ascontiguousarray(L, np.uint8).ravel()
    rc = io.shim_write_unproj(path.encode(), D.ctypes.data_as(f32p), L.ctypes.data_as(u8p), D.shape[0], D.shape[1],
                              chunk[0], chunk[1], gzip)
    assert rc == 0, rc


def test_reads_the_reference_result_file(io):
    """/W and /A of a result file the reference ships, read through the product's reader, against
    the golden copy of the same file (tests/golden/ref_results.npz, made by make_golden.py)."""
    if not os.path.exists(REF_H5):
        pytest.skip("/root/reference is not present on this machine")
    z = golden("ref_results.npz")
    assert str(z["r0_name"]) in REF_H5
    W = read_f32(io, REF_H5, "W", 544 * 544)
    A = read_f32(io, REF_H5, "A", 544 * 544)
    assert W.shape == (67, 544) and A.shape == (544, 544)
    assert np.array_equal(W, z["r0_W"]) and np.array_equal(A, z["r0_A"])
    assert relmax(W.T.astype(np.float64) @ W.astype(np.float64), A) <= 1e-6          # A == W^T W


@pytest.mark.parametrize("chunk,gzip", [((128, 1), 9), ((64, 16), 0), ((4096, 4096), 4)])
def test_chunked_deflate_input_round_trip(io, tmp_path, chunk, gzip):
    """The producer's layout ({128,1} chunks, gzip 9) and two others: every byte comes back."""
    N, F = 700, 48
    D, L = synth(N, F, k=6, seed=3)
    path = str(tmp_path / "x-unproj.h5")
    write_unproj(io, path, D, L, chunk, gzip)
    got = read_f32(io, path, "Distance", N * F)
    assert got.shape == (N, F) and np.array_equal(got, D)


@pytest.mark.gpu
def test_pj_learn_trains_from_a_producer_style_h5_file(io, tmp_path):
    """pj-learn on an HDF5 input written like comp-uprjdists writes it, against the same run from
    the .npy directory: identical stdout (up to the timing fields) and identical W / A."""
    cli = os.path.join(ROOT, "opencv-dlco_amd", "cli")
    subprocess.check_call(["make", "-s", "-C", cli])
    pj = os.path.join(cli, "pj-learn")
    N, F = 4000, 64
    D, L = synth(N, F, k=10, seed=41, sp=0.7, noise=0.2)
    h5 = str(tmp_path / "liberty-unproj.h5")
    write_unproj(io, h5, D, L.reshape(-1, 1), (128, 1), 9)
    npy = tmp_path / "unproj_npy"
    npy.mkdir()
    np.save(npy / "Distance.npy", D)
    np.save(npy / "Label.npy", L.reshape(-1, 1))
    outs = []
    for src, dst in ((h5, str(tmp_path / "out.h5")), (str(npy), str(tmp_path / "out_npy"))):
        p = subprocess.run([pj, src, dst, "-mu", "0.004", "-iters", "200", "-batch", "50"], capture_output=True, text=True, timeout=600)
        assert p.returncode == 0, p.stderr
        outs.append(p.stdout)
    import re
    cut = lambda s: [re.sub(r"Ttime: \S+ Vtime: \S+", "", l) for l in s.splitlines()]
    assert cut(outs[0]) == cut(outs[1])
    assert "Load Distances: 4000 x 64" in outs[0]
    W1 = read_f32(io, str(tmp_path / "out.h5"), "W", F * F)
    A1 = read_f32(io, str(tmp_path / "out.h5"), "A", F * F)
    assert np.array_equal(W1, np.load(tmp_path / "out_npy" / "W.npy")) and np.array_equal(A1, np.load(tmp_path / "out_npy" / "A.npy"))
    assert W1.shape[1] == F and A1.shape == (F, F)


def write_nd(io, path, name, arr, create=False):
    arr = np.ascontiguousarray(arr)
    code = {np.dtype(np.float32): 0, np.dtype(np.uint8): 1, np.dtype(np.int32): 2}[arr.dtype]
    sh = (C.c_size_t * 4)(*arr.shape)
    rc = io.shim_write_nd(path.encode(), 1 if create else 0, name.encode(), code, arr.ctypes.data_as(C.c_void_p), arr.ndim, sh)
    assert rc == 0, rc


def test_image_set_datasets_round_trip(io, tmp_path):
    """The inputs of comp-uprjdists / comp-fulldists: "Patches" u8 [n,64,64], "Indices" i32 [pairs,4], "PRFilters"
    f32 [rows,64,64] (src/comp-uprjdists.cpp:144-218): rank-3 datasets and the int32 type through the product's reader."""
    rng = np.random.default_rng(8)
    patches = rng.integers(0, 256, (9, 64, 64)).astype(np.uint8)
    pairs = rng.integers(-5, 100000, (11, 4)).astype(np.int32)
    filt = rng.random((8, 64, 64)).astype(np.float32)
    path = str(tmp_path / "imageset.h5")
    write_nd(io, path, "Patches", patches, create=True)
    write_nd(io, path, "Indices", pairs)
    write_nd(io, path, "PRFilters", filt)
    sh = (C.c_size_t * 4)()
    got_p = np.empty(patches.size, np.uint8)
    assert io.shim_read_u8(path.encode(), b"Patches", got_p.ctypes.data_as(u8p), got_p.size, sh) == 3 and tuple(sh[:3]) == (9, 64, 64)
    assert np.array_equal(got_p.reshape(patches.shape), patches)
    got_i = np.empty(pairs.size, np.int32)
    assert io.shim_read_i32(path.encode(), b"Indices", got_i.ctypes.data_as(C.POINTER(C.c_int32)), got_i.size, sh) == 2 and tuple(sh[:2]) == (11, 4)
    assert np.array_equal(got_i.reshape(pairs.shape), pairs)
    got_f = read_f32(io, path, "PRFilters", filt.size)
    assert got_f.shape == (8, 64, 64) and np.array_equal(got_f, filt)
    assert io.shim_read_i32(path.encode(), b"Missing", got_i.ctypes.data_as(C.POINTER(C.c_int32)), got_i.size, sh) == -1


@pytest.mark.gpu
def test_comp_uprjdists_reads_hdf5_inputs(io, tmp_path):
    """comp-uprjdists on .h5 inputs (the reference's own container) gives the bytes of the .npy run."""
    from test_descriptors import make_filters, make_patches
    cli = os.path.join(ROOT, "opencv-dlco_amd", "cli")
    subprocess.check_call(["make", "-s", "-C", cli])
    n = 60
    rng = np.random.default_rng(2)
    patches = make_patches(n, seed=5)
    PR = np.zeros((16, 4096), np.float32)
    PR[:12] = make_filters(12, seed=4, scale=25.0)
    w = np.array([[0.5, 0.25]], np.float32)
    pairs = np.stack([rng.integers(0, n, 200), rng.integers(0, 7, 200), rng.integers(0, n, 200), rng.integers(0, 7, 200)], 1).astype(np.int32)
    for sub in ("flt", "img", "prj"):
        (tmp_path / sub).mkdir()
    np.save(tmp_path / "flt" / "PRFilters.npy", PR.reshape(-1, 64, 64))
    np.save(tmp_path / "img" / "Patches.npy", patches)
    np.save(tmp_path / "img" / "Indices.npy", pairs)
    np.save(tmp_path / "prj" / "w.npy", w)
    write_nd(io, str(tmp_path / "flt.h5"), "PRFilters", PR.reshape(-1, 64, 64), create=True)
    write_nd(io, str(tmp_path / "img.h5"), "Patches", patches, create=True)
    write_nd(io, str(tmp_path / "img.h5"), "Indices", pairs)
    write_nd(io, str(tmp_path / "prj.h5"), "w", w, create=True)
    cu = os.path.join(cli, "comp-uprjdists")
    a = subprocess.run([cu, str(tmp_path / "flt"), str(tmp_path / "img"), "-prj", str(tmp_path / "prj"), "-id", "0", "-out", str(tmp_path / "out_npy")],
                       capture_output=True, text=True, timeout=600)
    b = subprocess.run([cu, str(tmp_path / "flt.h5"), str(tmp_path / "img.h5"), "-prj", str(tmp_path / "prj.h5"), "-id", "0", "-out", str(tmp_path / "out.h5")],
                       capture_output=True, text=True, timeout=600)
    assert a.returncode == 0 and b.returncode == 0, a.stderr + b.stderr
    D = np.load(tmp_path / "out_npy" / "Distance.npy")
    got = read_f32(io, str(tmp_path / "out.h5"), "Distance", D.size)
    assert got.shape == D.shape and np.array_equal(got, D)
    lab = np.empty(200, np.uint8)
    sh = (C.c_size_t * 4)()
    assert io.shim_read_u8(str(tmp_path / "out.h5").encode(), b"Label", lab.ctypes.data_as(u8p), 200, sh) == 2
    assert np.array_equal(lab, np.load(tmp_path / "out_npy" / "Label.npy").ravel())


def read_u8(io, path, name, cap):
    out = np.empty(cap, np.uint8)
    sh = (C.c_size_t * 4)()
    nd = io.shim_read_u8(path.encode(), name.encode(), out.ctypes.data_as(u8p), cap, sh)
    assert nd >= 0
    shape = tuple(sh[i] for i in range(nd))
    return out[:int(np.prod(shape))].reshape(shape)


@pytest.mark.parametrize("form", ["h5", "npy"])
def test_row_stream_writes_the_producer_layout_block_by_block(io, tmp_path, form):
    """comp-uprjdists / comp-fulldists stream their rows chunk by chunk into {128,1} gzip-9 datasets
    (src/comp-uprjdists.cpp:254-256,289-290,337-339) instead of holding the matrix: the product's RowStream
    against the whole-matrix writer, ragged last block, and an interrupted run that keeps what it wrote."""
    D, L = synth(300, 24, k=4, seed=5)
    path = str(tmp_path / ("s.h5" if form == "h5" else "sdir"))
    rc = io.shim_stream_unproj(path.encode(), D.ctypes.data_as(f32p), np.ascontiguousarray(L).ctypes.data_as(u8p), 300, 24, 128, 128, 9, 0)
    assert rc == 0
    got = read_f32(io, path, "Distance", D.size)
    assert got.shape == (300, 24) and np.array_equal(got, D)
    lab = read_u8(io, path, "Label", 300)
    assert lab.shape == (300, 1) and np.array_equal(lab.ravel(), L)
    if form == "h5":
        out = subprocess.run(["/opt/conda/bin/h5dump", "-H", "-p", path], capture_output=True, text=True)
        if out.returncode == 0:
            assert "CHUNKED ( 128, 1 )" in out.stdout and "DEFLATE { LEVEL 9 }" in out.stdout
        # stopped after the first block: the rows of that block are in the file, the rest is fill
        path2 = str(tmp_path / "partial.h5")
        assert io.shim_stream_unproj(path2.encode(), D.ctypes.data_as(f32p), np.ascontiguousarray(L).ctypes.data_as(u8p), 300, 24, 128, 128, 9, 128) == 0
        part = read_f32(io, path2, "Distance", D.size)
        assert np.array_equal(part[:128], D[:128]) and not part[128:].any()
