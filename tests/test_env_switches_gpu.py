"""The library's developer switches are read once per process (getenv), so each one gets a child process: the code
behind a switch - the fp32-MFMA gradient kernel `syrk_rda_kernel8` (the k-ordered v_mfma_f32_32x32x2_f32 chain that
north_star literally names), the fp32 filter / Rayleigh-Ritz products, the unpacked dual average, the recovery from a
grid-barrier time-out of the multi-workgroup Jacobi - is held to the same oracle gates as the default path."""
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _child(case):
    """Runs in the child: same bodies and gates as tests/test_gpu_parity.py."""
    import importlib
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch  # noqa: F401  (first HIP runtime in the process, see conftest.py)
    dlco = importlib.import_module("opencv-dlco_amd")
    from oracle import ref
    from util import relmax, synth
    ref.lib()
    ref.set_threads(1)                                    # small problems: see tests/conftest.py
    TOL_DIST, TOL_GRAD, TOL_A = 2e-5, 5e-6, 1e-4

    def grad_case(F, B, zero_frac):
        N = 1500
        D, L = synth(N, F, k=12, seed=F + B)
        ctx = dlco.Context(F, N, B=B, mu=0.004, gamma=0.5)
        ctx.set_data(D, L)
        rng = np.random.default_rng(B)
        pr, nr = rng.integers(0, N, B).astype(np.int32), rng.integers(0, N, B).astype(np.int32)
        rho, kap = rng.integers(0, 6, B).astype(np.int32), rng.integers(0, 6, B).astype(np.int32)
        rho[rng.random(B) < zero_frac] = 0
        kap[rng.random(B) < zero_frac] = 0
        G0 = rng.standard_normal((F, F)).astype(np.float32)
        G0 = (G0 + G0.T) * np.float32(0.5)
        got = ctx.grad_rda(pr, nr, rho, kap, 0.3, 0.9, G0)
        P, Nn = D[pr].astype(np.float64), D[nr].astype(np.float64)
        want = 0.9 * G0 + 0.3 * ((P * rho[:, None]).T @ P - (Nn * kap[:, None]).T @ Nn)
        e = relmax(got, want)
        assert e <= TOL_GRAD, (F, B, e)
        ctx.close()
        return e

    def teacher_forced(F, B, nstep, N=4000, expect_timeouts=False):
        D, L = synth(N, F, k=20, seed=9)
        mu, gamma = 0.004, 0.5
        tr = ref.Trainer(D, L, B=B, mu=mu, gamma=gamma, grad_order=1)
        ctx = dlco.Context(F, N, B=B, mu=mu, gamma=gamma)
        ctx.set_data(D, L)
        checked, worst = 0, 0.0
        for s in range(nstep):
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
                e = relmax(ctx.A(), after["A"])
                worst = max(worst, e)
                assert e <= TOL_A, (s, e)
                checked += 1
        assert checked >= nstep - 2, checked
        cn = ctx.counters()
        assert cn["nonconverged"] == 0
        if expect_timeouts:
            assert cn["jacobi_barrier_timeouts"] >= 1, cn
        else:
            assert cn["jacobi_barrier_timeouts"] == 0, cn
        ctx.close()
        tr.close()
        return worst

    def free_run(F, nstep, expect_passes, grad_bf16=0):
        N, B, mu, gamma = 4000, 200, 0.004, 0.5
        D, L = synth(N, F, k=20, seed=9)
        ctx = dlco.Context(F, N, B=B, mu=mu, gamma=gamma, grad_bf16=grad_bf16)
        ctx.set_data(D, L)
        worst = 0.0
        for s in range(nstep):
            ctx.step()
            Ap, _, _ = ref.psd_project(ref.dual_to_primal(ctx.dfavg(), mu, gamma, s))
            e = relmax(ctx.A(), Ap)
            worst = max(worst, e)
            assert e <= TOL_A, (s, e)
        cn = ctx.counters()
        assert cn["nonconverged"] == 0
        if expect_passes:
            assert cn["rank_update_passes"] >= nstep - 8, cn
        elif expect_passes is not None:
            assert cn["rank_update_passes"] == 0, cn
        ctx.close()
        return worst, cn

    if case == "rank_update_check":
        # the shortcut term beside the product it replaces, every step (DLCO_RANK_UPDATE_CHECK): the product is a
        # two-way split one (~1e-5 of its largest entry), the shortcut is the more exact of the two
        worst, cn = free_run(256, 30, True)
        print("free run with the check: A+ %.2e, largest deviation of a first term %.2e" % (worst, cn["rank_update_check"]))
        assert 0.0 < cn["rank_update_check"] <= 1e-4, cn
        # the bf16-once variant (BASELINE configs[4]): its gradient is formed from operands rounded to bf16 once, and so are
        # the projections and planes the shortcut reads - the term agrees with the product to the 2^-9 of that arithmetic in
        # its update part (largest in the first steps, where the update is a large share of the matrix); the result of the
        # step is still held to the fp32 gate against ssyevr on the variant's own dual average
        worst, cn = free_run(256, 30, True, grad_bf16=1)
        print("bf16-once variant: A+ %.2e, largest deviation of a first term %.2e" % (worst, cn["rank_update_check"]))
        assert 0.0 < cn["rank_update_check"] <= 1e-3, cn
    elif case == "no_locking":
        # the start-up transient without locking converged pairs out of the filter: more passes, the same gates
        worst, cn = free_run(256, 16, True)
        assert cn["locked_passes"] == 0, cn
        print("free run without locking: A+ %.2e" % worst)
    elif case == "tol_pass1":
        # the first pass held to the plain tolerance (the library's factor is 0.85): the F = 256 run stays inside the gate
        worst, cn = free_run(256, 30, True)
        print("free run, first pass to the plain tolerance: A+ %.2e" % worst)
    elif case.startswith("knob_"):
        # the remaining developer knobs (tolerances, thresholds, tile and read-back choices): each alternative path is held to
        # the same oracle gates on a teacher-forced run and on a short free run
        print("teacher forced F=256: %.2e" % teacher_forced(256, 200, 6))
        worst, cn = free_run(256, 12, None)
        print("free run: A+ %.2e" % worst)
    elif case == "no_rank_update":
        worst, cn = free_run(256, 30, False)
        print("free run without the shortcut: A+ %.2e" % worst)
    elif case == "syrk_fp32":
        for F, B, z in ((128, 8, 0.0), (256, 200, 0.3), (384, 33, 0.9), (544, 200, 0.2)):
            print("grad_rda F=%d B=%d: %.2e" % (F, B, grad_case(F, B, z)))
        print("teacher forced: %.2e" % teacher_forced(256, 200, 8))
    elif case in ("fp32_products", "no_packed"):
        print("teacher forced F=512: %.2e" % teacher_forced(512, 200, 10))
        print("teacher forced F=544: %.2e" % teacher_forced(544, 200, 6))
    elif case == "jmw_timeout":
        # step 0 of a batch of 200 + 200 seeds the block with 400 rows: the multi-workgroup Jacobi's size class
        print("teacher forced with forced barrier time-outs: %.2e" % teacher_forced(512, 200, 3, expect_timeouts=True))
    else:
        raise SystemExit("unknown case " + case)
    print("child ok")


CASES = {
    "syrk_fp32": {"DLCO_SYRK_FP32": "1"},
    "fp32_products": {"DLCO_FP32_FILTER": "1", "DLCO_FP32_RR": "1"},
    "no_packed": {"DLCO_NO_PACKED": "1"},
    "jmw_timeout": {"DLCO_TEST_JMW_TIMEOUT": "1"},
    "rank_update_check": {"DLCO_RANK_UPDATE_CHECK": "1"},
    "no_rank_update": {"DLCO_NO_RANK_UPDATE": "1"},
    "no_locking": {"DLCO_NO_LOCKING": "1"},
    "tol_pass1": {"DLCO_EIG_TOL_PASS1": "1.0"},
    "knob_uniform_crit": {"DLCO_EIG_UNIFORM_CRIT": "1"},
    "knob_eig_tol": {"DLCO_EIG_TOL": "1e-4"},
    "knob_gemm_big_tiles": {"DLCO_GEMM_BIG_TILES": "1"},
    "knob_jacobi_all_pairs": {"DLCO_JACOBI_ALL_PAIRS": "1"},
    "knob_jacobi_rot": {"DLCO_JACOBI_ROT": "1e-5"},
    "knob_jacobi_stop": {"DLCO_JACOBI_STOP": "3e-4"},
    "knob_no_cheap_pass": {"DLCO_NO_CHEAP_PASS": "1"},
    "knob_no_xcdmap": {"DLCO_NO_XCDMAP": "1"},
    "knob_panel_amp": {"DLCO_PANEL_AMP": "1e3"},
    "knob_sync_readback": {"DLCO_SYNC_READBACK": "1"},
}


@pytest.mark.gpu
@pytest.mark.parametrize("case", sorted(CASES))
def test_switch_in_a_child_process(case):
    env = dict(os.environ)
    env.update(CASES[case])
    p = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", case], capture_output=True, text=True, timeout=600, env=env)
    print(p.stdout)
    assert p.returncode == 0 and "child ok" in p.stdout, p.stdout[-3000:] + p.stderr[-3000:]


if __name__ == "__main__":
    if len(sys.argv) == 3 and sys.argv[1] == "--child":
        _child(sys.argv[2])
    else:
        raise SystemExit("usage: test_env_switches_gpu.py --child <case>")
