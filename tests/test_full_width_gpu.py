"""Parity at the TIMED configuration: the workloads bench.py measures (BASELINE configs[1] and
configs[2]: 500 000 pair-rows x PR-dim 8192, batch 200+200, learned rank ~64 / ~128) compared
with the CPU oracle at full width.

The reference's step ends in LAPACKE_ssyevr on the full F x F matrix followed by
W = rows sqrt(e)*v (src/pj-learn.cpp:434-490); the HIP path replaces that with a subspace
tracker (split-bf16 Chebyshev filter, three-way-split Rayleigh-Ritz product, block of ~96-160
rows).  Here the GPU runs to the very step count bench.py times (its burn-in + warm-up from W = 0:
320 steps for c2, 520 for c3, where the rank has come down to ~128), its dual average is handed to
the oracle, and ONE ssyevr at n = 8192 (~60 s on the box's host cores) gives the reference's W for
the same matrix.  The c2 case also carries the BASELINE configs[4] leg: a second trainer on the same
resident data with cfg.grad_bf16 (bf16 MFMA + fp32 accumulate in the gradient, projection and
statistics GEMMs), checked the same way against its own ssyevr and, for the metric's gate, against
the fp32 trainer's FPR@95 over all 500 000 rows.

Tolerances (fp32, relative to the largest magnitude of the reference quantity):
  rank                    +-1   (an eigenvalue within fp32 noise of mu may fall on either side,
                                 as between two LAPACK builds)
  A+ = W^T W              1e-4  (SURVEY 8(d)'s gate; the measured error is printed: ~1e-5)
  kept eigenvalues        1e-5 * lambda_max
  per-pair distances      2e-5 * max  between the two W over 16 384 rows, and vs the oracle's
                                 own projection on 256 rows
  FPR@95 of the two W     1e-3 absolute (+-0.1 %, the metric's band) over all 500 000 rows
  dfAvg after one teacher-forced fused SYRK + dual average   5e-6 vs fp64
  bf16 variant: the same W scored through the bf16 statistics pass    FPR@95 +-1e-3 (the metric's band), distances
                                 2e-2 * max (two operands rounded to 8 significant bits); the bf16-trained model vs
                                 the fp32-trained one after the same number of steps: FPR@95 within 5e-3, rank +-6
                                 (two free-running trajectories: the band is the run-to-run one, not the metric's)
"""
import os
import sys

import numpy as np
import pytest

from util import relmax

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

pytestmark = pytest.mark.gpu

TOL_A = 1e-4
TOL_EVAL = 1e-5
TOL_DIST = 2e-5
TOL_GRAD = 5e-6
TOL_FPR = 1e-3


def _oracle_W(ref, dfavg, mu, gamma, t_last):
    """The reference's W for the dual average `dfavg` of iteration t_last (src/pj-learn.cpp:426-490)."""
    A = ref.dual_to_primal(dfavg, mu, gamma, t_last)
    ref.set_threads(len(os.sched_getaffinity(0)))         # ssyevr at n = 8192: every core (the suite's default is one thread)
    W, ev = ref.psd_factor(A)
    ref.set_threads(1)
    del A
    return W, ev


def _gram64(W):
    W64 = W.astype(np.float64)
    return W64.T @ W64


def _steady_state_case(dlco, ref, name, steps, ctx=None, label="", score=None):
    """`score`: the context whose (fp32) projection / statistics passes compare the two W; default: the trainer itself."""
    import bench
    wl = bench.WORKLOADS[name]
    if ctx is None:
        ctx = bench.build_context(dlco, wl)
    score = score or ctx
    ctx.steps(steps - 60)
    early = ctx.counters()["nonconverged"]
    ctx.steps(60)
    cn = ctx.counters()
    # the start-up transient (rank of several hundred collapsing to the regime's) may cost a step or two;
    # the steady state that the bench times must not miss the tolerance at all
    assert early <= 3 and cn["nonconverged"] == early, "tracker missed its tolerance: %d early, %d total" % (early, cn["nonconverged"])
    t = ctx.t()
    assert t == steps
    W_gpu = ctx.W()
    dfavg = ctx.dfavg()
    assert np.array_equal(dfavg, dfavg.T)
    W_ref, ev = _oracle_W(ref, dfavg, wl["mu"], wl["gamma"], t - 1)
    r_gpu, r_ref = W_gpu.shape[0], W_ref.shape[0]
    lo, hi = wl["rank_band"]
    assert lo <= r_ref <= hi, "workload drifted out of the regime the config names: rank %d" % r_ref
    assert abs(r_gpu - r_ref) <= 1, (r_gpu, r_ref)

    # A+ (E2): W^T W on both sides
    Ag, Ar = _gram64(W_gpu), _gram64(W_ref)
    err_A = np.abs(Ag - Ar).max() / np.abs(Ar).max()
    del Ag, Ar
    print("%s%s after %d steps: rank %d (oracle ssyevr %d), err_A = %.3e (gate %.0e)" % (name, label, steps, r_gpu, r_ref, err_A, TOL_A))
    assert err_A <= TOL_A, err_A

    # kept eigenvalues = squared row norms of W (ascending, like LAPACK's)
    e_gpu = (W_gpu.astype(np.float64) ** 2).sum(1)
    e_ref = (W_ref.astype(np.float64) ** 2).sum(1)
    assert (np.diff(e_gpu) >= -1e-6 * e_gpu.max()).all()
    k = min(r_gpu, r_ref)
    assert np.abs(e_gpu[-k:] - e_ref[-k:]).max() <= TOL_EVAL * e_ref.max()
    # the oracle's eigenvalues of A are those of W's rows
    pos = ev[ev > 0].astype(np.float64)
    assert np.abs(np.sort(pos)[-k:] - e_ref[-k:]).max() <= 1e-5 * e_ref.max()
    # rows of the GPU's W are mutually orthogonal
    G = W_gpu.astype(np.float64) @ W_gpu.T.astype(np.float64)
    assert np.abs(G - np.diag(np.diag(G))).max() <= 1e-4 * e_gpu.max()

    # per-pair distances under the two W (P1+P2 at F = 8192)
    rng = np.random.default_rng(7)
    ids = rng.integers(0, score.N, 16384).astype(np.int32)
    d_gpu = score.project_sqdist(ids, W_gpu)
    d_ref = score.project_sqdist(ids, W_ref)
    assert np.abs(d_gpu - d_ref).max() <= TOL_DIST * d_ref.max(), np.abs(d_gpu - d_ref).max() / d_ref.max()
    rows = score.get_rows(1000, 256)
    d_or = ref.project_sqdist(W_ref, rows)
    d_hip = score.project_sqdist(np.arange(1000, 1256, dtype=np.int32), W_ref)
    assert np.abs(d_hip - d_or).max() <= TOL_DIST * d_or.max()

    # FPR@95 / AUC of the two models on all N rows (S1-S4), and the oracle's ROC on the GPU's distances
    L = (np.arange(score.N) % 2 == 0).astype(np.uint8)
    dim_g, f_g, a_g = score.stats(W_gpu)
    dim_r, f_r, a_r = score.stats(W_ref)
    assert dim_g == r_gpu and dim_r == r_ref
    assert abs(f_g - f_r) <= TOL_FPR and abs(a_g - a_r) <= 1e-4, (f_g, f_r, a_g, a_r)
    d_all = score.project_sqdist(np.arange(score.N, dtype=np.int32), W_gpu)
    f_o, a_o = ref.roc_stats(d_all, L)
    assert f_o == f_g and abs(a_o - a_g) <= 1e-12
    flo, fhi = wl["fpr95_band"]
    assert flo <= f_r <= fhi, "the workload's operating point moved: FPR95 %.4f" % f_r
    return ctx, dfavg, t


def test_config1_steady_state_vs_oracle_ssyevr(dlco, ref):
    """configs[1]: rank ~64 at the bench's timed state (burn-in 300 + warm-up 20 steps).  Also the teacher-forced
    fused SYRK + dual average at F = 8192, and the configs[4] leg (bf16 MFMA + fp32 accumulate) on the same data."""
    import bench
    ctx, dfavg, t = _steady_state_case(dlco, ref, "c2", 320)
    # Q1+U1 at full width: replay the last batch's gradient on the state before it
    b = ctx.batch()
    K = int((b["rho"] > 0).sum() + (b["kappa"] > 0).sum())
    assert 100 <= K <= 400
    P = np.concatenate([ctx.get_rows(int(r), 1) for r in b["pos_rows"]])
    Ng = np.concatenate([ctx.get_rows(int(r), 1) for r in b["neg_rows"]])
    B = ctx.B
    alpha = np.float32(1.0) / np.float32(B * B * (t + 1))
    beta = np.float32(np.float64(t) / np.float64(t + 1))
    got = ctx.grad_rda(b["pos_rows"], b["neg_rows"], b["rho"], b["kappa"], float(alpha), float(beta), dfavg)
    want = ref.grad_reform(P, Ng, b["rho"], b["kappa"], f64=True)
    want *= np.float64(alpha)
    want += np.float64(beta) * dfavg
    assert relmax(got, want) <= TOL_GRAD
    assert np.array_equal(got, got.T)
    del got, want, dfavg

    # ---- BASELINE configs[4]: bf16 MFMA + fp32 accumulate in every GEMM over the resident matrix ---------------------
    wl = bench.WORKLOADS["c2"]
    W32 = ctx.W()
    _, f32_32, a32_32 = ctx.stats(W32)                       # the fp32 model through the fp32 statistics pass
    cb = bench.build_context(dlco, wl, data_from=ctx, grad_bf16=1)
    # (i) one model, both arithmetic variants of the statistics pass: the metric's own band
    dim_b, f32_b, a32_b = cb.stats(W32)
    rng = np.random.default_rng(3)
    ids = rng.integers(0, ctx.N, 16384).astype(np.int32)
    d32, d16 = ctx.project_sqdist(ids, W32), cb.project_sqdist(ids, W32)
    e16 = np.abs(d16 - d32).max() / d32.max()
    print("c2 bf16 statistics pass on the fp32 model: FPR95 %.4f vs %.4f, AUC %.6f vs %.6f, distances %.2e * max" % (f32_b, f32_32, a32_b, a32_32, e16))
    assert dim_b == W32.shape[0] and abs(f32_b - f32_32) <= TOL_FPR and abs(a32_b - a32_32) <= 1e-3
    assert 1e-6 < e16 <= 2e-2                                # really the bf16 path, and inside its budget
    # (ii) a trainer that runs every such GEMM in bf16 from W = 0 to the same step count, against its own ssyevr ...
    cb, _, _ = _steady_state_case(dlco, ref, "c2", 320, ctx=cb, label=" [bf16 MFMA + fp32 accumulate]", score=ctx)
    # (iii) ... and against the fp32 trainer, both scored by the fp32 statistics pass over all 500 000 rows
    W16 = cb.W()
    _, f16_32, a16_32 = ctx.stats(W16)
    print("c2 bf16-trained model vs fp32-trained model (fp32 scoring): FPR95 %.4f vs %.4f, AUC %.6f vs %.6f, rank %d vs %d"
          % (f16_32, f32_32, a16_32, a32_32, W16.shape[0], W32.shape[0]))
    assert abs(f16_32 - f32_32) <= 5e-3 and abs(a16_32 - a32_32) <= 2e-3 and abs(W16.shape[0] - W32.shape[0]) <= 6
    cb.close()
    ctx.close()


def test_config2_rank128_steady_state_vs_oracle_ssyevr(dlco, ref):
    """configs[2]: the rank ~128 regime (denser spectrum around mu, taller tracker block) at the state bench.py --config c3
    times: its burn-in of 500 steps + 20 warm-up steps."""
    ctx, _, _ = _steady_state_case(dlco, ref, "c3", 520)
    ctx.close()
