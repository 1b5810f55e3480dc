"""The widths the reference itself ran: FeatDim comes from the input file (src/pj-learn.cpp:176-179) and every log the
reference ships says `Load Distances: 500000 x 480` or `x 544` (e.g.
workspace/pj-learn/logging/liberty-liberty-0.035-0.250-pr#7-0.0010-0.100-pj.log:3); the authors' originals are 608 wide.
None is a multiple of the 128-column tile of the fused kernels: the library keeps every row at the next multiple (zero
columns, see include/dlco.h `dlco_device_width`) and converts at the ABI.  These tests run the HIP path at those widths
with the reference's batch (200 + 200) and its usual flags (mu 0.001, gamma 0.1) against the CPU oracle:
teacher-forced steps (ids bit-exact, distances 2e-5, dual average 5e-6 * 4, A+ 1e-4), a free run inside the metric's
bands, the operators at odd widths, and the pj-learn program from a producer-style {128,1} gzip-9 HDF5 file."""
import os
import re
import subprocess

import numpy as np
import pytest

from util import relmax, synth

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TOL_DIST, TOL_GRAD, TOL_A = 2e-5, 5e-6, 1e-4        # the gates of tests/test_gpu_parity.py


def _data(N, F, seed):
    # 64 latent directions, matches at 0.8 of the non-matches' spread, noise 0.25: the classes overlap (FPR@95 of a few per cent)
    return synth(N, F, k=64, seed=seed, sp=0.8, noise=0.25)


@pytest.mark.parametrize("F", [544, 480, 608])
def test_teacher_forced_steps_at_the_reference_widths(dlco, ref, F):
    N, B, mu, gamma = 20000, 200, 0.001, 0.1
    D, L = _data(N, F, seed=F)
    tr = ref.Trainer(D, L, B=B, mu=mu, gamma=gamma, grad_order=1)
    ctx = dlco.Context(F, N, B=B, mu=mu, gamma=gamma)
    assert ctx.device_width() == (F + 127) // 128 * 128 and ctx.device_width() != F
    ctx.set_data(D, L)
    assert np.array_equal(ctx.get_rows(17, 3), D[17:20])                  # the pad columns never leave the device
    nstep, checked, worst = 24, 0, 0.0
    for s in range(nstep):
        before = tr.state()
        tr.step()
        after = tr.state()
        ctx.set_state(s, before["dfavg"], before["W"] if s else None)
        ctx.step()
        b = ctx.batch()
        pr, nr = tr.batch_ids()
        assert np.array_equal(b["pos_rows"], pr) and np.array_equal(b["neg_rows"], nr)      # R3: bit-exact
        pd, nd = tr.batch_dists()
        scale = max(pd.max(), nd.max(), 1e-30)
        assert np.abs(b["pd"] - pd).max() <= TOL_DIST * scale and np.abs(b["nd"] - nd).max() <= TOL_DIST * scale
        rho, kap = ref.viol_counts(b["pd"], b["nd"])                      # exact given the GPU's own distances
        assert np.array_equal(b["rho"], rho) and np.array_equal(b["kappa"], kap)
        rho_o, kap_o = ref.viol_counts(pd, nd)
        if np.array_equal(rho, rho_o) and np.array_equal(kap, kap_o):    # (a distance within 1 ulp of a margin may flip a count)
            G = ctx.dfavg()
            assert G.shape == (F, F)
            assert relmax(G, after["dfavg"]) <= TOL_GRAD * 4
            W = ctx.W()
            assert W.shape[1] == F and abs(W.shape[0] - after["W"].shape[0]) <= 1
            e = relmax(ctx.A(), after["A"])
            worst = max(worst, e)
            assert e <= TOL_A, (s, e)
            checked += 1
    assert checked >= nstep - 4, checked
    assert ctx.counters()["nonconverged"] == 0
    print("F=%d: %d of %d steps checked, worst A+ error %.2e, rank %d" % (F, checked, nstep, worst, ctx.W().shape[0]))
    ctx.close()
    tr.close()


def test_free_run_band_at_544(dlco, ref):
    """300 free-running steps of both trainers at the reference's own shape (544 wide, batch 200 + 200, mu 0.001,
    gamma 0.1): objective, rank, FPR@95 and AUC inside the bands two equally valid chaotic trajectories allow, and the
    GPU's final W scored by both sides within the metric's +-0.1 %."""
    N, F, B, mu, gamma = 20000, 544, 200, 0.001, 0.1
    D, L = _data(N, F, seed=5440)
    tr = ref.Trainer(D, L, B=B, mu=mu, gamma=gamma, grad_order=1)
    ctx = dlco.Context(F, N, B=B, mu=mu, gamma=gamma)
    ctx.set_data(D, L)
    for _ in range(300):
        tr.step()
    ctx.steps(300)
    assert ctx.counters()["nonconverged"] == 0
    lo, rg, rank = ctx.validate()
    lo_r, rg_r = tr.validate()
    dim, f95, auc = ctx.stats()
    dim_r, f95_r, auc_r = tr.stats()
    assert abs(lo - lo_r) <= 0.05 * max(lo_r, 1e-6) + 1e-4 and abs(rg - rg_r) <= 0.05 * max(rg_r, 1e-6) + 1e-4
    assert abs(rank - dim_r) <= max(2, dim_r // 20)
    n_neg = int((L == 0).sum())
    se = float(np.sqrt(max(f95_r * (1.0 - f95_r), 1e-6) / n_neg))
    assert abs(f95 - f95_r) <= max(1e-3, 3.0 * se) and abs(auc - auc_r) <= 3e-3
    # one model, both scorers: the +-0.1 % statement
    W = ctx.W()
    assert W.shape == (rank, F)
    d_all = ctx.project_sqdist(np.arange(N, dtype=np.int32), W)
    d_ref = ((D.astype(np.float64) @ W.T.astype(np.float64)) ** 2).sum(1)
    assert np.abs(d_all - d_ref).max() <= TOL_DIST * np.abs(d_ref).max()
    f_o, a_o = ref.roc_stats(d_all, L)
    assert f_o == f95 and abs(a_o - auc) <= 1e-12
    e = ctx.log_step()
    assert e.is_best == 1 and e.saved == 1 and e.dim == rank
    Ws, As = ctx.saved()
    assert Ws.shape == (rank, F) and As.shape == (F, F)
    assert relmax(Ws.T.astype(np.float64) @ Ws.astype(np.float64), As) <= 1e-5
    print("F=544 free run: rank %d (oracle %d), FPR95 %.4f (%.4f), AUC %.5f (%.5f)" % (rank, dim_r, f95, f95_r, auc, auc_r))
    ctx.close()
    tr.close()


def test_full_size_reference_shape_500000_x_544(dlco, ref):
    """The reference's run at its own size - `Load Distances: 500000 x 544`, 200 000 + 200 000 training rows, 50 000 + 50 000
    validation rows, batch 200 + 200, mu 0.001, gamma 0.1: the six count lines, teacher-forced steps against the oracle, then
    150 free-running steps of both trainers inside the metric's bands and the LogStep block over all 500 000 rows with the
    GPU's W scored by both sides."""
    N, F, B, mu, gamma = 500000, 544, 200, 0.001, 0.1
    rng = np.random.default_rng(544)
    U = np.linalg.qr(rng.standard_normal((F, 64)))[0].T.astype(np.float32)
    L = (np.arange(N) % 2 == 0).astype(np.uint8)
    D = np.empty((N, F), np.float32)
    for r0 in range(0, N, 50000):                                        # (in slices: the float64 temporaries stay small)
        r1 = r0 + 50000
        z = rng.standard_normal((r1 - r0, 64)).astype(np.float32) * np.where(L[r0:r1, None] == 1, 0.8, 1.0).astype(np.float32)
        D[r0:r1] = np.clip(z @ U + 0.25 * rng.standard_normal((r1 - r0, F)).astype(np.float32), -1, 1)
    tr = ref.Trainer(D, L, B=B, mu=mu, gamma=gamma, grad_order=1)
    ctx = dlco.Context(F, N, B=B, mu=mu, gamma=gamma)
    ctx.set_data(D, L)
    idx = ctx.index()
    assert (idx["pos"].size, idx["neg"].size, idx["n_pos_trn"], idx["n_neg_trn"]) == (250000, 250000, 200000, 200000)   # ...pj.log:5-10
    ipos, ineg = ref.build_index(L)
    assert np.array_equal(idx["pos"], ipos) and np.array_equal(idx["neg"], ineg)       # R1 at full size, bit-exact
    checked = 0
    for s in range(6):
        before = tr.state()
        tr.step()
        after = tr.state()
        ctx.set_state(s, before["dfavg"], before["W"] if s else None)
        ctx.step()
        b = ctx.batch()
        pr, nr = tr.batch_ids()
        assert np.array_equal(b["pos_rows"], pr) and np.array_equal(b["neg_rows"], nr)
        pd, nd = tr.batch_dists()
        scale = max(pd.max(), nd.max(), 1e-30)
        assert np.abs(b["pd"] - pd).max() <= TOL_DIST * scale and np.abs(b["nd"] - nd).max() <= TOL_DIST * scale
        rho, kap = ref.viol_counts(pd, nd)
        if np.array_equal(b["rho"], rho) and np.array_equal(b["kappa"], kap):
            assert relmax(ctx.dfavg(), after["dfavg"]) <= TOL_GRAD * 4
            assert relmax(ctx.A(), after["A"]) <= TOL_A
            checked += 1
    assert checked >= 5
    st = tr.state()
    ctx.set_state(st["t"], st["dfavg"], st["W"])                             # both continue from the oracle's state, freely
    for _ in range(150):
        tr.step()
    ctx.steps(150)
    assert ctx.counters()["nonconverged"] == 0
    lo, rg, rank = ctx.validate()
    lo_r, rg_r = tr.validate()
    dim, f95, auc = ctx.stats()
    dim_r, f95_r, auc_r = tr.stats()
    assert abs(lo - lo_r) <= 0.05 * max(lo_r, 1e-6) + 1e-4 and abs(rg - rg_r) <= 0.05 * max(rg_r, 1e-6) + 1e-4
    assert abs(rank - dim_r) <= max(2, dim_r // 20)
    se = float(np.sqrt(max(f95_r * (1.0 - f95_r), 1e-6) / 250000))
    assert abs(f95 - f95_r) <= max(1e-3, 3.0 * se) and abs(auc - auc_r) <= 3e-3
    W = ctx.W()
    d_all = ctx.project_sqdist(np.arange(N, dtype=np.int32), W)
    f_o, a_o = ref.roc_stats(d_all, L)                                        # one model, both scorers: the +-0.1 % statement
    assert f_o == f95 and abs(a_o - auc) <= 1e-12
    print("500000 x 544: rank %d (oracle %d), FPR95 %.4f (%.4f), AUC %.5f (%.5f)" % (rank, dim_r, f95, f95_r, auc, auc_r))
    ctx.close()
    tr.close()


@pytest.mark.parametrize("F,B", [(544, 200), (100, 33), (8, 4), (3, 2), (130, 16)])
def test_operators_at_odd_widths(dlco, ref, F, B):
    """The single operators of the ABI at widths that are not tile multiples: every host-side array has the caller's
    width, and the results equal the oracle's on the unpadded data."""
    N = 1200
    D, L = synth(N, F, k=min(F, 12), seed=300 + F, sp=0.6, noise=0.15)
    ctx = dlco.Context(F, N, B=B, mu=0.002, gamma=0.5)
    ctx.set_data(D, L)
    rng = np.random.default_rng(F)
    r = min(F, 7)
    W = rng.standard_normal((r, F)).astype(np.float32) * 0.3
    ids = rng.integers(0, N, 300).astype(np.int32)
    d = ctx.project_sqdist(ids, W)
    want = ((D[ids].astype(np.float64) @ W.T.astype(np.float64)) ** 2).sum(1)
    assert np.abs(d - want).max() <= TOL_DIST * max(np.abs(want).max(), 1e-30)
    # gradient + dual average on a random batch with random counts
    pr, nr = rng.integers(0, N, B).astype(np.int32), rng.integers(0, N, B).astype(np.int32)
    rho, kap = rng.integers(0, 5, B).astype(np.int32), rng.integers(0, 5, B).astype(np.int32)
    G0 = rng.standard_normal((F, F)).astype(np.float32)
    G0 = (G0 + G0.T) * 0.5
    alpha, beta = 0.37, 0.81
    G1 = ctx.grad_rda(pr, nr, rho, kap, alpha, beta, G0)
    P, Nn = D[pr].astype(np.float64), D[nr].astype(np.float64)
    want = beta * G0 + alpha * ((P * rho[:, None]).T @ P - (Nn * kap[:, None]).T @ Nn)
    assert G1.shape == (F, F) and relmax(G1, want) <= TOL_GRAD * 4
    # PSD projection of an indefinite dual average: about a third of the directions end up with a positive eigenvalue of A
    Q = np.linalg.qr(rng.standard_normal((F, F)))[0]
    npos = max(1, F // 3)
    ev = np.concatenate([-rng.random(npos) * 0.5 - 0.02, rng.random(F - npos) * 0.2 - 0.001])
    G = ((Q * ev) @ Q.T).astype(np.float32)
    G = (G + G.T) * np.float32(0.5)
    Ao, Wo, _ = ref.psd_project(ref.dual_to_primal(G, 0.002, 0.5, 3))
    Wp, A = ctx.psd_project(G, 3, want_A=True)
    assert A.shape == (F, F) and Wp.shape[1] == F and abs(Wp.shape[0] - Wo.shape[0]) <= 1
    assert relmax(A, Ao) <= TOL_A
    ctx.close()


def test_width_beyond_the_packed_layout(dlco, ref):
    """F = 8330 (66 tiles after padding to 8448: more than the 64 the packed layout takes): full F x F dual average, the
    row-streaming product kernel with a K split that divides the tile count.  Gradient against float64, then three training
    steps from W = 0 whose PSD projection is cheap to check exactly (the dual average has rank <= 3 * 2B = 48: A+ is computed
    here in float64 from the eigen-decomposition of the matrix restricted to its range)."""
    N, F, B = 600, 8330, 8
    D, L = synth(N, F, k=16, seed=8330, sp=0.6, noise=0.15)
    mu, gamma = 0.002, 0.5
    ctx = dlco.Context(F, N, B=B, mu=mu, gamma=gamma)
    assert ctx.device_width() == 8448
    ctx.set_data(D, L)
    rng = np.random.default_rng(1)
    pr, nr = rng.integers(0, N, B).astype(np.int32), rng.integers(0, N, B).astype(np.int32)
    rho, kap = rng.integers(0, 5, B).astype(np.int32), rng.integers(0, 5, B).astype(np.int32)
    G1 = ctx.grad_rda(pr, nr, rho, kap, 0.37, 0.0, None)
    P, Nn = D[pr].astype(np.float64), D[nr].astype(np.float64)
    want = 0.37 * ((P * rho[:, None]).T @ P - (Nn * kap[:, None]).T @ Nn)
    assert G1.shape == (F, F) and relmax(G1, want) <= TOL_GRAD * 4
    del G1, want
    for t in range(3):
        ctx.step()
        assert ctx.counters()["nonconverged"] == 0
        G = ctx.dfavg().astype(np.float64)
        # exact A+ = c (-G - mu I)_+ from the range of G: G = R^T S R with R = the (at most 2B (t+1)) rows that entered it
        c = np.sqrt(t + 1.0) / gamma
        Om = np.random.default_rng(t).standard_normal((F, 64))
        R = np.linalg.qr(G @ Om)[0]                                        # an orthonormal basis of the range (rank <= 48)
        w, Vs = np.linalg.eigh(-(R.T @ G @ R))
        V = R @ Vs
        keep = w > mu
        Aplus = (V[:, keep] * (c * (w[keep] - mu))) @ V[:, keep].T
        A = ctx.A().astype(np.float64)
        assert np.abs(A - Aplus).max() <= TOL_A * np.abs(Aplus).max(), t
        W = ctx.W()
        assert W.shape == (int(keep.sum()), F) or abs(W.shape[0] - int(keep.sum())) <= 1
    ctx.close()


def test_device_resident_inputs_of_odd_width_and_shared_data(dlco):
    """`dlco_set_data_device` with a matrix of the caller's row stride (544 floats: not a tile multiple, so the library copies
    it into its padded layout) and `dlco_set_data_shared` (a second trainer on the first one's resident matrix) against
    `dlco_set_data` from the host: bit-identical trainers."""
    import torch
    N, F, B = 6000, 544, 200
    D, L = _data(N, F, seed=11)
    a = dlco.Context(F, N, B=B, mu=0.001, gamma=0.1)
    a.set_data(D, L)
    t = torch.from_numpy(D).to("cuda")
    torch.cuda.synchronize()
    b = dlco.Context(F, N, B=B, mu=0.001, gamma=0.1)
    b.set_data_device(t.data_ptr(), L)
    c = dlco.Context(F, N, B=B, mu=0.001, gamma=0.1)
    c.set_data_shared(a)
    for ctx in (a, b, c):
        ctx.steps(12)
    Wa = a.W()
    assert np.array_equal(Wa, b.W()) and np.array_equal(Wa, c.W())
    assert np.array_equal(a.dfavg(), b.dfavg()) and np.array_equal(a.dfavg(), c.dfavg())
    assert np.array_equal(b.get_rows(100, 5), D[100:105])
    for ctx in (c, b, a):
        ctx.close()


def test_pj_learn_program_from_a_544_wide_gzip9_file(tmp_path):
    """The program a user of the reference runs (workspace/08-pjlearn.sh), on a file written like comp-uprjdists writes
    it ({128,1} chunks, gzip 9, src/comp-uprjdists.cpp:254,289-290) at the reference's own width."""
    from test_hdf5_io import io as _io_fixture, read_f32, write_unproj  # noqa: F401  (helpers; the fixture is rebuilt below)
    import ctypes as C
    out = tmp_path / "libioshim.so"
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-shared", "-fPIC", "-o", str(out), os.path.join(ROOT, "tests", "shim", "io_shim.cpp"), "-ldl"])
    io = C.CDLL(str(out))
    f32p, u8p = C.POINTER(C.c_float), C.POINTER(C.c_uint8)
    io.shim_read_f32.argtypes = [C.c_char_p, C.c_char_p, f32p, C.c_size_t, C.POINTER(C.c_size_t)]
    io.shim_write_unproj.argtypes = [C.c_char_p, f32p, u8p, C.c_size_t, C.c_size_t, C.c_size_t, C.c_size_t, C.c_int]
    if not io.shim_hdf5_available():
        pytest.skip("no libhdf5 >= 1.10 on this machine")
    cli = os.path.join(ROOT, "opencv-dlco_amd", "cli")
    subprocess.check_call(["make", "-s", "-C", cli])
    N, F = 6000, 544
    D, L = _data(N, F, seed=77)
    h5 = str(tmp_path / "liberty-liberty-unproj.h5")
    write_unproj(io, h5, D, L.reshape(-1, 1), (128, 1), 9)
    dst = str(tmp_path / "liberty-liberty-pj.h5")
    p = subprocess.run([os.path.join(cli, "pj-learn"), h5, dst, "-mu", "0.001", "-gamma", "0.1", "-iters", "300"],
                       capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr
    assert "Load Distances: %d x %d" % (N, F) in p.stdout
    saved = [l for l in p.stdout.splitlines() if "saved" in l]
    assert saved, p.stdout
    # workspace/09-pjstats.sh:28 reads Dim, AUC_best, FPR95_best from the last [saved] line
    fields = re.sub(r"[:()]", " ", saved[-1]).split()
    dim = int(fields[2].strip("[]"))
    W = read_f32(io, dst, "W", F * F)
    A = read_f32(io, dst, "A", F * F)
    assert W.shape == (dim, F) and A.shape == (F, F)
    assert relmax(W.T.astype(np.float64) @ W.astype(np.float64), A) <= 1e-5
    from oracle import ref
    d = ((D.astype(np.float64) @ W.T.astype(np.float64)) ** 2).sum(1).astype(np.float32)
    f95, auc = ref.roc_stats(d, L)
    m = re.search(r"AUC: (\S+) \((\S+)\) FPR95: (\S+) \((\S+)\)", saved[-1])
    assert abs(float(m.group(1)) - auc) <= 2e-4 and abs(float(m.group(3)) - 100.0 * f95) <= 0.1 + 1e-9
