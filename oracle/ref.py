"""ctypes loader for the CPU oracle (oracle/libdlco_ref.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py.  Nothing in the product package uses it.
See oracle/dlco_ref.h for scope, reference citations and pin status.
"""
import ctypes as C
import glob
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

c_f32p = C.POINTER(C.c_float)
c_f64p = C.POINTER(C.c_double)
c_i32p = C.POINTER(C.c_int32)
c_u8p = C.POINTER(C.c_uint8)


def build():
    """Compile the oracle (gcc; a few seconds)."""
    subprocess.check_call(["make", "-s", "-C", _HERE])


def find_openblas():
    """Path of an OpenBLAS shared object with cblas + LAPACKE, or None."""
    env = os.environ.get("DLCO_OPENBLAS")
    if env and os.path.exists(env):
        return env
    try:
        import scipy  # noqa: F401
        base = os.path.join(os.path.dirname(os.path.dirname(scipy.__file__)), "scipy.libs")
        hits = sorted(glob.glob(os.path.join(base, "libscipy_openblas*.so")))
        if hits:
            return hits[0]
    except Exception:
        pass
    for pat in ("/usr/lib/x86_64-linux-gnu/libopenblas.so*", "/usr/lib64/libopenblas.so*"):
        hits = sorted(glob.glob(pat))
        if hits:
            return hits[0]
    return None


def _p(a, typ):
    return a.ctypes.data_as(typ) if a is not None else None


def lib():
    global _LIB
    if _LIB is not None:
        return _LIB
    path = os.path.join(_HERE, "libdlco_ref.so")
    if not os.path.exists(path):
        build()
    L = C.CDLL(path)
    L.dlco_ref_load_blas.argtypes = [C.c_char_p]
    L.dlco_ref_load_blas.restype = C.c_int
    L.dlco_ref_blas_kind.restype = C.c_int
    L.dlco_ref_set_threads.argtypes = [C.c_int]
    L.dlco_ref_rng_next.argtypes = [C.POINTER(C.c_uint64)]
    L.dlco_ref_rng_next.restype = C.c_uint32
    L.dlco_ref_rng_uniform.argtypes = [C.POINTER(C.c_uint64), C.c_int, C.c_int]
    L.dlco_ref_rng_uniform.restype = C.c_int
    L.dlco_ref_rand_shuffle_i32.argtypes = [c_i32p, C.c_uint32, C.POINTER(C.c_uint64)]
    L.dlco_ref_build_index.argtypes = [c_u8p, C.c_int, c_i32p, C.POINTER(C.c_int), c_i32p, C.POINTER(C.c_int)]
    L.dlco_ref_split.argtypes = [C.c_size_t]
    L.dlco_ref_split.restype = C.c_size_t
    L.dlco_ref_sample.argtypes = [C.POINTER(C.c_uint64), C.c_uint, C.c_uint, C.c_int, c_i32p, c_i32p]
    L.dlco_ref_project_sqdist.argtypes = [c_f32p, C.c_int, C.c_int, c_f32p, C.c_int, c_f32p]
    L.dlco_ref_project_sqdist_ids.argtypes = [c_f32p, C.c_int, C.c_int, c_f32p, c_i32p, C.c_int, c_f32p]
    L.dlco_ref_viol_counts.argtypes = [c_f32p, c_f32p, C.c_int, c_i32p, c_i32p]
    L.dlco_ref_grad_reforder.argtypes = [c_f32p, c_f32p, c_f32p, c_f32p, C.c_int, C.c_int, c_f32p]
    L.dlco_ref_grad_reform.argtypes = [c_f32p, c_f32p, c_i32p, c_i32p, C.c_int, C.c_int, c_f32p]
    L.dlco_ref_grad_reform_f64.argtypes = [c_f32p, c_f32p, c_i32p, c_i32p, C.c_int, C.c_int, c_f64p]
    L.dlco_ref_rda_update.argtypes = [c_f32p, c_f32p, C.c_uint, C.c_uint, C.c_int]
    L.dlco_ref_dual_to_primal.argtypes = [c_f32p, C.c_float, C.c_float, C.c_uint, C.c_int, c_f32p]
    L.dlco_ref_psd_project.argtypes = [c_f32p, C.c_int, c_f32p, C.POINTER(C.c_int), c_f32p]
    L.dlco_ref_psd_project.restype = C.c_int
    L.dlco_ref_psd_factor.argtypes = [c_f32p, C.c_int, c_f32p, C.POINTER(C.c_int), c_f32p]
    L.dlco_ref_psd_factor.restype = C.c_int
    L.dlco_ref_get_timers.argtypes = [C.c_void_p, c_f64p]
    L.dlco_ref_hinge_sum.argtypes = [c_f32p, C.c_int, c_f32p, C.c_int]
    L.dlco_ref_hinge_sum.restype = C.c_double
    L.dlco_ref_trace.argtypes = [c_f32p, C.c_int]
    L.dlco_ref_trace.restype = C.c_double
    L.dlco_ref_nonzero_rows.argtypes = [c_f32p, C.c_int, C.c_int, c_f32p]
    L.dlco_ref_nonzero_rows.restype = C.c_int
    L.dlco_ref_roc_stats.argtypes = [c_f32p, c_u8p, C.c_int, C.POINTER(C.c_float), C.POINTER(C.c_double)]
    L.dlco_ref_create.argtypes = [c_f32p, c_u8p, C.c_int, C.c_int, C.c_int, C.c_float, C.c_float]
    L.dlco_ref_create.restype = C.c_void_p
    L.dlco_ref_destroy.argtypes = [C.c_void_p]
    L.dlco_ref_set_grad_order.argtypes = [C.c_void_p, C.c_int]
    L.dlco_ref_step.argtypes = [C.c_void_p]
    L.dlco_ref_step.restype = C.c_int
    L.dlco_ref_get_batch_ids.argtypes = [C.c_void_p, c_i32p, c_i32p]
    L.dlco_ref_get_batch_dists.argtypes = [C.c_void_p, c_f32p, c_f32p]
    L.dlco_ref_get_state.argtypes = [C.c_void_p, C.POINTER(C.c_uint), C.POINTER(C.c_int), c_f32p, c_f32p, c_f32p, c_f32p]
    L.dlco_ref_set_state.argtypes = [C.c_void_p, C.c_uint, c_f32p, c_f32p, C.c_int]
    L.dlco_ref_get_index.argtypes = [C.c_void_p, c_i32p, C.POINTER(C.c_int), C.POINTER(C.c_int),
                                     c_i32p, C.POINTER(C.c_int), C.POINTER(C.c_int)]
    L.dlco_ref_validate.argtypes = [C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_float)]
    L.dlco_ref_stats.argtypes = [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_float), C.POINTER(C.c_double)]
    L.dlco_ref_pr_create.argtypes = [c_f32p, c_u8p, C.c_int, C.c_int, C.c_float, C.c_float]
    L.dlco_ref_pr_create.restype = C.c_void_p
    L.dlco_ref_pr_destroy.argtypes = [C.c_void_p]
    L.dlco_ref_pr_steps.argtypes = [C.c_void_p, C.c_uint]
    L.dlco_ref_pr_get.argtypes = [C.c_void_p, C.POINTER(C.c_uint), c_f32p, c_f32p, c_i32p, c_i32p, c_f32p]
    L.dlco_ref_pr_set.argtypes = [C.c_void_p, C.c_uint, c_f32p, c_f32p]
    L.dlco_ref_pr_validate.argtypes = [C.c_void_p, c_f32p, c_f32p, c_i32p]
    L.dlco_ref_pr_stats.argtypes = [c_f32p, C.c_int, c_f32p, c_u8p, C.c_int, C.c_int, c_f32p, C.c_int, C.c_int,
                                    c_i32p, c_i32p, c_i32p, c_f32p, c_f64p]
    L.dlco_ref_get_desc.argtypes = [c_u8p, C.c_int, C.c_float, C.c_int, c_f32p]
    L.dlco_ref_patch_descriptor.argtypes = [c_u8p, c_f32p, C.c_int, c_f32p]
    ob = find_openblas()
    if ob:
        L.dlco_ref_load_blas(ob.encode())
    _LIB = L
    return L


def blas_kind():
    return "openblas" if lib().dlco_ref_blas_kind() else "builtin"


def set_threads(n):
    lib().dlco_ref_set_threads(int(n))


# --------------------------------------------------------------------------
# function wrappers (numpy in / numpy out)
# --------------------------------------------------------------------------
class Rng:
    """cv::RNG restatement (state is a uint64)."""

    def __init__(self, state=0xFFFFFFFF):
        self.state = C.c_uint64(state if state else 0xFFFFFFFF)

    def next(self):
        return int(lib().dlco_ref_rng_next(C.byref(self.state)))

    def uniform(self, a, b):
        return int(lib().dlco_ref_rng_uniform(C.byref(self.state), a, b))

    def shuffle(self, arr):
        arr = np.ascontiguousarray(arr, dtype=np.int32)
        lib().dlco_ref_rand_shuffle_i32(_p(arr, c_i32p), arr.size, C.byref(self.state))
        return arr

    def sample(self, n_pos_trn, n_neg_trn, B):
        ip = np.empty(B, np.int32)
        ineg = np.empty(B, np.int32)
        lib().dlco_ref_sample(C.byref(self.state), n_pos_trn, n_neg_trn, B, _p(ip, c_i32p), _p(ineg, c_i32p))
        return ip, ineg


def build_index(labels):
    labels = np.ascontiguousarray(labels, dtype=np.uint8).ravel()
    n = labels.size
    pos = np.empty(max(n, 1), np.int32)
    neg = np.empty(max(n, 1), np.int32)
    npos, nneg = C.c_int(), C.c_int()
    lib().dlco_ref_build_index(_p(labels, c_u8p), n, _p(pos, c_i32p), C.byref(npos), _p(neg, c_i32p), C.byref(nneg))
    return pos[:npos.value].copy(), neg[:nneg.value].copy()


def split(n):
    return int(lib().dlco_ref_split(n))


def project_sqdist(W, X):
    W = np.ascontiguousarray(W, np.float32)
    X = np.ascontiguousarray(X, np.float32)
    r, F = W.shape if W.size else (0, X.shape[1])
    out = np.empty(X.shape[0], np.float32)
    lib().dlco_ref_project_sqdist(_p(W, c_f32p), r, F, _p(X, c_f32p), X.shape[0], _p(out, c_f32p))
    return out


def project_sqdist_ids(W, D, ids):
    W = np.ascontiguousarray(W, np.float32)
    ids = np.ascontiguousarray(ids, np.int32)
    assert D.dtype == np.float32 and D.flags.c_contiguous
    out = np.empty(ids.size, np.float32)
    lib().dlco_ref_project_sqdist_ids(_p(W, c_f32p), W.shape[0], D.shape[1], _p(D, c_f32p), _p(ids, c_i32p), ids.size, _p(out, c_f32p))
    return out


def viol_counts(pd, nd):
    pd = np.ascontiguousarray(pd, np.float32)
    nd = np.ascontiguousarray(nd, np.float32)
    B = pd.size
    rho = np.empty(B, np.int32)
    kap = np.empty(B, np.int32)
    lib().dlco_ref_viol_counts(_p(pd, c_f32p), _p(nd, c_f32p), B, _p(rho, c_i32p), _p(kap, c_i32p))
    return rho, kap


def grad_reforder(P, Ng, pd, nd):
    P = np.ascontiguousarray(P, np.float32)
    Ng = np.ascontiguousarray(Ng, np.float32)
    pd = np.ascontiguousarray(pd, np.float32)
    nd = np.ascontiguousarray(nd, np.float32)
    B, F = P.shape
    out = np.empty((F, F), np.float32)
    lib().dlco_ref_grad_reforder(_p(P, c_f32p), _p(Ng, c_f32p), _p(pd, c_f32p), _p(nd, c_f32p), B, F, _p(out, c_f32p))
    return out


def grad_reform(P, Ng, rho, kappa, f64=False):
    P = np.ascontiguousarray(P, np.float32)
    Ng = np.ascontiguousarray(Ng, np.float32)
    rho = np.ascontiguousarray(rho, np.int32)
    kappa = np.ascontiguousarray(kappa, np.int32)
    B, F = P.shape
    if f64:
        out = np.empty((F, F), np.float64)
        lib().dlco_ref_grad_reform_f64(_p(P, c_f32p), _p(Ng, c_f32p), _p(rho, c_i32p), _p(kappa, c_i32p), B, F, _p(out, c_f64p))
    else:
        out = np.empty((F, F), np.float32)
        lib().dlco_ref_grad_reform(_p(P, c_f32p), _p(Ng, c_f32p), _p(rho, c_i32p), _p(kappa, c_i32p), B, F, _p(out, c_f32p))
    return out


def rda_update(dfavg, dloss, t, B):
    dfavg = np.array(dfavg, np.float32, order="C", copy=True)
    dloss = np.ascontiguousarray(dloss, np.float32)
    lib().dlco_ref_rda_update(_p(dfavg, c_f32p), _p(dloss, c_f32p), t, B, dfavg.shape[0])
    return dfavg


def dual_to_primal(dfavg, mu, gamma, t):
    dfavg = np.ascontiguousarray(dfavg, np.float32)
    A = np.empty_like(dfavg)
    lib().dlco_ref_dual_to_primal(_p(dfavg, c_f32p), mu, gamma, t, dfavg.shape[0], _p(A, c_f32p))
    return A


def psd_project(A):
    """returns (A_plus, W[r,F], evals[F])"""
    A = np.array(A, np.float32, order="C", copy=True)
    F = A.shape[0]
    W = np.empty((F, F), np.float32)
    ev = np.empty(F, np.float32)
    r = C.c_int()
    rc = lib().dlco_ref_psd_project(_p(A, c_f32p), F, _p(W, c_f32p), C.byref(r), _p(ev, c_f32p))
    if rc != 0:
        raise RuntimeError("dlco_ref_psd_project failed: %d" % rc)
    return A, W[:r.value].copy(), ev


def psd_factor(A):
    """E1 + the W half of E2 (ssyevr, W = rows sqrt(e)*v ascending) without the F^3 product: (W[r,F], evals[F])"""
    A = np.ascontiguousarray(A, np.float32)
    F = A.shape[0]
    W = np.empty((F, F), np.float32)
    ev = np.empty(F, np.float32)
    r = C.c_int()
    rc = lib().dlco_ref_psd_factor(_p(A, c_f32p), F, _p(W, c_f32p), C.byref(r), _p(ev, c_f32p))
    if rc != 0:
        raise RuntimeError("dlco_ref_psd_factor failed: %d" % rc)
    return W[:r.value].copy(), ev


def hinge_sum(pos, neg):
    pos = np.ascontiguousarray(pos, np.float32)
    neg = np.ascontiguousarray(neg, np.float32)
    return float(lib().dlco_ref_hinge_sum(_p(pos, c_f32p), pos.size, _p(neg, c_f32p), neg.size))


def nonzero_rows(W):
    W = np.ascontiguousarray(W, np.float32)
    out = np.empty_like(W)
    n = lib().dlco_ref_nonzero_rows(_p(W, c_f32p), W.shape[0], W.shape[1], _p(out, c_f32p))
    return out[:n].copy()


def roc_stats(dist, labels):
    dist = np.ascontiguousarray(dist, np.float32)
    labels = np.ascontiguousarray(labels, np.uint8).ravel()
    f = C.c_float()
    a = C.c_double()
    lib().dlco_ref_roc_stats(_p(dist, c_f32p), _p(labels, c_u8p), dist.size, C.byref(f), C.byref(a))
    return float(f.value), float(a.value)


class Trainer:
    """The reference training loop (src/pj-learn.cpp:214-587), one step at a time."""

    def __init__(self, dists, labels, B=200, mu=0.001, gamma=0.5, grad_order=0):
        self.dists = np.ascontiguousarray(dists, np.float32)
        self.labels = np.ascontiguousarray(labels, np.uint8).ravel()
        self.N, self.F = self.dists.shape
        self.B, self.mu, self.gamma = B, mu, gamma
        self._h = lib().dlco_ref_create(_p(self.dists, c_f32p), _p(self.labels, c_u8p), self.N, self.F, B, mu, gamma)
        lib().dlco_ref_set_grad_order(self._h, grad_order)

    def close(self):
        if self._h:
            lib().dlco_ref_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def step(self):
        rc = lib().dlco_ref_step(self._h)
        if rc != 0:
            raise RuntimeError("dlco_ref_step failed: %d" % rc)

    def timers(self):
        """seconds since creation: sample+project, gradient, RDA + dual->primal, PSD projection"""
        out = np.zeros(4, np.float64)
        lib().dlco_ref_get_timers(self._h, _p(out, c_f64p))
        return dict(project=out[0], grad=out[1], rda=out[2], eig=out[3])

    def batch_ids(self):
        p = np.empty(self.B, np.int32)
        n = np.empty(self.B, np.int32)
        lib().dlco_ref_get_batch_ids(self._h, _p(p, c_i32p), _p(n, c_i32p))
        return p, n

    def batch_dists(self):
        p = np.empty(self.B, np.float32)
        n = np.empty(self.B, np.float32)
        lib().dlco_ref_get_batch_dists(self._h, _p(p, c_f32p), _p(n, c_f32p))
        return p, n

    def state(self):
        F = self.F
        t, r = C.c_uint(), C.c_int()
        W = np.empty((F, F), np.float32)
        A = np.empty((F, F), np.float32)
        df = np.empty((F, F), np.float32)
        dl = np.empty((F, F), np.float32)
        lib().dlco_ref_get_state(self._h, C.byref(t), C.byref(r), _p(W, c_f32p), _p(A, c_f32p), _p(df, c_f32p), _p(dl, c_f32p))
        return dict(t=t.value, r=r.value, W=W[:r.value].copy(), A=A, dfavg=df, dloss=dl)

    def set_state(self, t, dfavg=None, W=None):
        dfavg = None if dfavg is None else np.ascontiguousarray(dfavg, np.float32)
        W = None if W is None else np.ascontiguousarray(W, np.float32)
        lib().dlco_ref_set_state(self._h, t, _p(dfavg, c_f32p), _p(W, c_f32p), 0 if W is None else W.shape[0])

    def index(self):
        pos = np.empty(self.N, np.int32)
        neg = np.empty(self.N, np.int32)
        a, b, c_, d = C.c_int(), C.c_int(), C.c_int(), C.c_int()
        lib().dlco_ref_get_index(self._h, _p(pos, c_i32p), C.byref(a), C.byref(b), _p(neg, c_i32p), C.byref(c_), C.byref(d))
        return dict(pos=pos[:a.value].copy(), n_pos_trn=b.value, neg=neg[:c_.value].copy(), n_neg_trn=d.value)

    def validate(self):
        lo, rg = C.c_float(), C.c_float()
        lib().dlco_ref_validate(self._h, C.byref(lo), C.byref(rg))
        return float(lo.value), float(rg.value)

    def stats(self):
        d, f, a = C.c_int(), C.c_float(), C.c_double()
        lib().dlco_ref_stats(self._h, C.byref(d), C.byref(f), C.byref(a))
        return d.value, float(f.value), float(a.value)


class PrTrainer:
    """The reference's pr-learn loop (src/pr-learn.cpp:229-434), single-thread order."""

    def __init__(self, dists, labels, mu=0.025, gamma=0.10):
        self.dists = np.ascontiguousarray(dists, np.float32)
        self.labels = np.ascontiguousarray(labels, np.uint8).ravel()
        self.N, self.F = self.dists.shape
        self._h = lib().dlco_ref_pr_create(_p(self.dists, c_f32p), _p(self.labels, c_u8p), self.N, self.F, mu, gamma)

    def close(self):
        if self._h:
            lib().dlco_ref_pr_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def steps(self, n):
        lib().dlco_ref_pr_steps(self._h, n)

    def state(self):
        t = C.c_uint()
        w, df = np.empty(self.F, np.float32), np.empty(self.F, np.float32)
        lp, ln, lf = C.c_int32(), C.c_int32(), C.c_float()
        lib().dlco_ref_pr_get(self._h, C.byref(t), _p(w, c_f32p), _p(df, c_f32p), C.byref(lp), C.byref(ln), C.byref(lf))
        return dict(t=t.value, w=w, dfavg=df, last_pos=lp.value, last_neg=ln.value, last_f=lf.value)

    def set_state(self, t, w=None, dfavg=None):
        w = None if w is None else np.ascontiguousarray(w, np.float32)
        d = None if dfavg is None else np.ascontiguousarray(dfavg, np.float32)
        lib().dlco_ref_pr_set(self._h, t, _p(w, c_f32p), _p(d, c_f32p))

    def validate(self):
        lo, rg, nz = C.c_float(), C.c_float(), C.c_int32()
        lib().dlco_ref_pr_validate(self._h, C.byref(lo), C.byref(rg), C.byref(nz))
        return float(lo.value), float(rg.value), nz.value

    def stats(self, prparams, w=None, nchannels=8, max_dim=-1):
        p = np.ascontiguousarray(prparams, np.float32)
        w = self.state()["w"] if w is None else np.ascontiguousarray(w, np.float32)
        npr, dim, nz, f, a = C.c_int32(), C.c_int32(), C.c_int32(), C.c_float(-1.0), C.c_double(0.0)
        lib().dlco_ref_pr_stats(_p(p, c_f32p), p.shape[1], _p(self.dists, c_f32p), _p(self.labels, c_u8p), self.N, self.F,
                                _p(w, c_f32p), nchannels, max_dim, C.byref(npr), C.byref(dim), C.byref(nz), C.byref(f), C.byref(a))
        return dict(nPR=npr.value, dim=dim.value, nzdim=nz.value, fpr95=float(f.value), auc=float(a.value))


def get_desc(patch, n_angle_bins=8, init_sigma=1.4, norm=True):
    """get_desc (src/vgg-desc.cpp:41-152): 64x64 u8 patch -> PatchTrans [4096, n_angle_bins]."""
    p = np.ascontiguousarray(patch, np.uint8)
    assert p.shape == (64, 64)
    out = np.empty((4096, n_angle_bins), np.float32)
    lib().dlco_ref_get_desc(_p(p, c_u8p), n_angle_bins, init_sigma, 1 if norm else 0, _p(out, c_f32p))
    return out


def patch_descriptor(patch, sPR):
    """min(sPRFilters * get_desc(patch), 1) flattened row-major [nsel*8] (src/comp-uprjdists.cpp:317-325)."""
    p = np.ascontiguousarray(patch, np.uint8)
    f = np.ascontiguousarray(sPR, np.float32)
    assert p.shape == (64, 64) and f.shape[1] == 4096
    out = np.empty(f.shape[0] * 8, np.float32)
    lib().dlco_ref_patch_descriptor(_p(p, c_u8p), _p(f, c_f32p), f.shape[0], _p(out, c_f32p))
    return out


def select_pr_filters(pr_filters, w):
    """SelectPRFilters restated loop for loop (src/misc.cpp:78-168): keep rows whose w entry is positive and
    that have a non-zero, drop repeats (first occurrence stays), then the reference's insertion sort."""
    f = np.ascontiguousarray(pr_filters, np.float32)
    wv = np.asarray(w, np.float32).ravel()
    assert wv.size * 8 == f.shape[0]
    kept = [f[i * 8 + j] for i in range(wv.size) for j in range(8)
            if wv[i] > 0.0 and np.count_nonzero(f[i * 8 + j]) != 0]                 # :89-100
    uniq = []
    for r in kept:                                                                    # :104-122
        if not any(np.array_equal(r, u) for u in uniq):
            uniq.append(r)
    if not uniq:
        return np.empty((0, f.shape[1]), np.float32)
    s = np.zeros((len(uniq), f.shape[1]), np.float32)
    s[0] = uniq[0]
    for i in range(1, len(uniq)):                                                     # :127-165
        idx = i
        while idx > 0:
            cmp = False
            for j in range(f.shape[1]):
                if uniq[i][j] == s[idx - 1][j]:
                    continue
                if uniq[i][j] < s[idx - 1][j]:
                    cmp = True
                    s[idx] = s[idx - 1]
                else:
                    cmp = False
                    s[idx] = uniq[i]
                break
            if cmp:
                idx -= 1
                if idx == 0:
                    s[idx] = uniq[i]
                continue
            break
    return s


def full_dists(patch1, patch2, pr_filters):
    """One row of comp-fulldists' "Distance" (src/comp-fulldists.cpp:318-343): PRFilters [8*n_regions, 4096],
    dist[g] = sum over the region's 8 rows and 8 bins of (Desc2 - Desc1)^2, row sums first (cuda::reduce twice).
    The reference forms Desc with cuda::gemm (fp32, order unspecified); here with the double accumulation of
    patch_descriptor — the comparison tolerance covers the difference."""
    f = np.ascontiguousarray(pr_filters, np.float32)
    assert f.shape[0] % 8 == 0
    d1 = patch_descriptor(patch1, f).reshape(-1, 8)
    d2 = patch_descriptor(patch2, f).reshape(-1, 8)
    l2 = (d2 - d1) ** np.float32(2)
    rows = l2.sum(axis=1, dtype=np.float32)
    return rows.reshape(-1, 8).sum(axis=1, dtype=np.float32)
