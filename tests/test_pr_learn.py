"""pr-learn: the pooling-region stage, src/pr-learn.cpp:229-434 + ComputePRStats
(src/misc.cpp:171-264).

The reference holds no vectors for this stage either (its inputs are not in the repository), so the
oracle's restatement is frozen against an independent numpy restatement of the same lines (CPU test)
and the HIP path is compared with the oracle (GPU tests): the iteration is sequential and its
arithmetic is restated operation for operation, so w and dfAvg must agree BIT FOR BIT after any number
of iterations; the validation / statistics products carry fp32 tolerances.  Parity against a live
OpenCV build: unpinned (MatExpr / scaleAdd / gemm semantics are restated, see oracle/dlco_ref.c)."""
import os
import re

import numpy as np
import pytest

from util import golden, synth

REF_LOG = "/root/reference/workspace/pr-learn/logging/liberty-0.035-0.250-pr.log"


def pr_data(N, F, seed, nsig=12):
    """Squared-difference style features: positives small on the informative columns."""
    rng = np.random.default_rng(seed)
    L = (np.arange(N) % 2 == 0).astype(np.uint8)
    D = rng.random((N, F)).astype(np.float32)
    D[:, :nsig] *= np.where(L[:, None] == 1, 0.25, 1.0).astype(np.float32)
    return D, L


def numpy_pr_steps(D, L, mu, gamma, n, ref):
    """Independent restatement of src/pr-learn.cpp:302-329 in numpy float32 / float64."""
    pos, neg = ref.build_index(L)
    npt, nnt = ref.split(pos.size), ref.split(neg.size)
    rng = ref.Rng(2215)
    F = D.shape[1]
    w = np.zeros(F, np.float32)
    df = np.zeros(F, np.float32)
    for t in range(n):
        ip = rng.uniform(0, npt)
        ineg = rng.uniform(0, nnt)
        diff = (D[pos[ip]] - D[neg[ineg]]).astype(np.float32)
        f = np.float32(np.dot(w.astype(np.float64), diff.astype(np.float64)))
        df = (df * np.float32(np.float64(t) * (np.float64(1.0) / np.float64(t + 1)))).astype(np.float32)
        if f > np.float32(-1.0):
            df = ((diff * np.float32(np.float64(1.0) / np.float64(t + 1))).astype(np.float32) + df).astype(np.float32)
        a = -np.sqrt(np.float64(t + 1)) / np.float64(np.float32(gamma))
        w = ((df * np.float32(a)).astype(np.float32) + np.float32(np.float64(np.float32(mu)) * a)).astype(np.float32)
        w = np.maximum(w, np.float32(0.0))
    return w, df


def test_oracle_matches_an_independent_numpy_restatement(ref):
    N, F = 600, 64
    D, L = pr_data(N, F, 3)
    mu, gamma = 0.02, 0.25
    tr = ref.PrTrainer(D, L, mu=mu, gamma=gamma)
    for n in (1, 7, 60, 400):
        tr.set_state(0, np.zeros(F, np.float32), np.zeros(F, np.float32))
        tr2 = ref.PrTrainer(D, L, mu=mu, gamma=gamma)          # fresh sampler
        tr2.steps(n)
        st = tr2.state()
        w, df = numpy_pr_steps(D, L, mu, gamma, n, ref)
        assert st["t"] == n
        assert np.array_equal(st["w"], w) and np.array_equal(st["dfavg"], df)
        tr2.close()
    tr.close()


def test_oracle_statistics_count_pooling_regions_like_the_reference(ref):
    """nzDim = selected rows with a non-zero entry, nPR = nzDim - (rows that have a twin) / 2, Dim = nPR * 8
    (src/misc.cpp:183-217), on a table with one duplicated region and one all-zero row."""
    F = 6
    P = np.zeros((8 * F, 3), np.float32)
    rng = np.random.default_rng(1)
    P[:] = rng.integers(1, 9, P.shape)
    P[8 * 2 + 3] = P[8 * 0 + 1]               # a twin pair between regions 0 and 2
    P[8 * 4 + 7] = 0                          # an all-zero row of region 4
    D, L = pr_data(200, F, 5, nsig=2)
    w = np.array([0.5, 0.0, 0.1, 0.0, 2.0, -1.0], np.float32)      # regions 0, 2, 4 selected
    tr = ref.PrTrainer(D, L)
    s = tr.stats(P, w)
    assert s["nzdim"] == 8 + 8 + 7 and s["nPR"] == 23 - 1 and s["dim"] == 22 * 8
    assert 0.0 <= s["fpr95"] <= 1.0 and 0.0 <= s["auc"] <= 1.0
    s2 = tr.stats(P, w, max_dim=100)          # Dim > MaxDim: returns before the ROC pass
    assert s2["dim"] == 176 and s2["fpr95"] == -1.0
    tr.close()


def test_reference_log_grammar_of_the_pr_stage():
    """The lines the CLI must reproduce (src/pr-learn.cpp:366-403); scraped by workspace/05-prstats.sh."""
    if not os.path.exists(REF_LOG):
        pytest.skip("/root/reference is not present on this machine")
    lines = open(REF_LOG).read().splitlines()
    assert lines[0].startswith("mu: 0.035 gamma: 0.25")
    assert lines[1] == "Load PRParams." and lines[2] == "Load RingParams."
    best = re.compile(r"^Best: \d+  Loss: \d+\.\d{6} Regul: \d+\.\d{6} Obj: \d+\.\d{6} \(\d+\.\d{6}\)  NNZ: \d+ \(\d+\)  Ttime: \d+\.\d{4} Vtime: \d+\.\d{4}$")
    stat = re.compile(r"^Stat: nPR #\d+ \(#\d+\) Dim/MaxDim \[\d+/\d+\] AUC: \d\.\d{6} FPR95: \d+\.\d{2}( \[saved\])?$")
    step = re.compile(r"^Step: \d+  Loss: .* NNZ: \d+ \(\d+\)  Ttime: \d+\.\d{4} Vtime: \d+\.\d{4}$")
    body = [l for l in lines if l.startswith(("Best:", "Stat:", "Step:"))]
    assert len(body) > 100
    for l in body:
        assert best.match(l) or stat.match(l) or step.match(l), l


def test_oracle_regulariser_and_nnz_against_the_reference_result_files(ref):
    """Known answers the reference itself holds for this stage: every "[saved]" line of a pr-learn log has a row in the
    "w" dataset of the matching result file (src/pr-learn.cpp:385-400), and the "Best:" line above it prints
    Regul = mu * sum|w| (%.6f) and NNZ = countNonZero(w) of that very w (:358,366-369); the Stat line's second count
    is nzDim = 8 rows per positive weight (src/misc.cpp:183-193).  Fixture tests/golden/pr_saved.npz (three runs,
    24 rows).  Pins the regulariser / NNZ half of dlco_ref_pr_validate and the selection count of dlco_ref_pr_stats."""
    z = golden("pr_saved.npz")
    N = 40
    rng = np.random.default_rng(0)
    checked = 0
    for i in range(3):
        w_rows, log, mu = z["p%d_w" % i], z["p%d_log" % i], float(z["p%d_mu" % i])
        F = w_rows.shape[1]
        assert F == 5120 and len(log) == w_rows.shape[0] >= 5
        D = rng.random((N, F)).astype(np.float32)
        L = (np.arange(N) % 2 == 0).astype(np.uint8)
        tr = ref.PrTrainer(D, L, mu=mu, gamma=0.25)
        P = np.ones((8 * F, 3), np.float32)                # every pooling-region row non-zero and distinct enough for the count
        P[:, 0] = np.arange(8 * F)
        for w, (t, regul, nnz, nzdim, npr, dim) in zip(w_rows, log):
            assert (w >= 0).all()                          # w = max(w, 0), :326
            tr.set_state(int(t), w=w)
            _, rg, nz = tr.validate()
            assert nz == int(nnz), (str(z["p%d_name" % i]), t)
            assert abs(rg - regul) <= 5.1e-7, (rg, regul)  # the log prints six decimals
            st = tr.stats(P, w=w, max_dim=0)               # returns before the ROC pass (Dim > MaxDim)
            assert st["nzdim"] == int(nzdim) == 8 * int(nnz)
            checked += 1
        tr.close()
    assert checked == 24


@pytest.mark.gpu
def test_gpu_trajectory_is_the_oracles_bit_for_bit(dlco, ref):
    N, F = 4000, 5120                          # the reference's FeatDim
    D, L = pr_data(N, F, 11, nsig=40)
    mu, gamma = 0.035, 0.25
    tr = ref.PrTrainer(D, L, mu=mu, gamma=gamma)
    ctx = dlco.PrContext(F, N, mu=mu, gamma=gamma)
    ctx.set_data(D, L)
    ix = ctx.index()
    assert ix["n_pos_trn"] == ref.split(N // 2) and ix["n_neg_trn"] == ref.split(N // 2)
    import time
    done = 0
    for n in (1, 2, 97, 3000, 20000):          # windows of different lengths, one launch each
        t0 = time.perf_counter()
        tr.steps(n)
        t1 = time.perf_counter()
        ctx.steps(n)
        ctx.state()
        t2 = time.perf_counter()
        if n == 20000:
            print("pr-learn, %d iterations at F = %d: oracle (one host core) %.1f k it/s, GPU %.1f k it/s (incl. the read-back)"
                  % (n, F, n / (t1 - t0) / 1e3, n / (t2 - t1) / 1e3))
        done += n
        a, b = tr.state(), ctx.state()
        assert a["t"] == b["t"] == done
        assert np.array_equal(a["dfavg"], b["dfavg"]), "dfAvg differs after %d iterations" % done
        assert np.array_equal(a["w"], b["w"]), "w differs after %d iterations" % done
    assert 0 < int((b["w"] != 0).sum()) < F    # the L1 term has switched most regions off, not all
    ctx.close()
    tr.close()


@pytest.mark.gpu
@pytest.mark.parametrize("F", [64, 1024, 2052, 8192])
def test_gpu_trajectory_other_widths(dlco, ref, F):
    N = 1200
    D, L = pr_data(N, F, F, nsig=min(F // 4, 30))
    tr = ref.PrTrainer(D, L, mu=0.03, gamma=0.2)
    ctx = dlco.PrContext(F, N, mu=0.03, gamma=0.2)
    ctx.set_data(D, L)
    tr.steps(2500)
    ctx.steps(2500)
    a, b = tr.state(), ctx.state()
    assert np.array_equal(a["w"], b["w"]) and np.array_equal(a["dfavg"], b["dfavg"])
    ctx.close()
    tr.close()


@pytest.mark.gpu
def test_gpu_validation_and_statistics(dlco, ref):
    N, F = 6000, 512
    D, L = pr_data(N, F, 21, nsig=30)
    mu, gamma = 0.03, 0.25
    tr = ref.PrTrainer(D, L, mu=mu, gamma=gamma)
    ctx = dlco.PrContext(F, N, mu=mu, gamma=gamma)
    ctx.set_data(D, L)
    tr.steps(30000)
    ctx.steps(30000)
    lo, rg, nz = ctx.validate()
    lo_r, rg_r, nz_r = tr.validate()
    assert nz == nz_r and nz > 0
    assert abs(rg - rg_r) <= 1e-6 * max(rg_r, 1e-12)             # mu * sum|w|, same w
    assert abs(lo - lo_r) <= 1e-5 * max(lo_r, 1e-12)             # 600 x 600 hinge terms, float dot products
    rng = np.random.default_rng(2)
    P = rng.integers(0, 5, (8 * F, 3)).astype(np.float32)
    P[8 * 5 + 1] = P[8 * 9 + 2]
    s, s_r = ctx.stats(P), tr.stats(P)
    assert (s["nPR"], s["dim"], s["nzdim"]) == (s_r["nPR"], s_r["dim"], s_r["nzdim"])
    assert abs(s["fpr95"] - s_r["fpr95"]) <= 1e-3 and abs(s["auc"] - s_r["auc"]) <= 1e-4
    w = tr.state()["w"]
    s2 = ctx.stats(P, w=w, max_dim=8)
    assert s2["dim"] == s_r["dim"] and s2["fpr95"] == -1.0       # early return above MaxDim
    ctx.close()
    tr.close()
