"""ctypes binding of libdlco.so — the MI355X (gfx950) pj-learn hot path.

The package directory name carries a hyphen (it mirrors the upstream project name), so it
is loaded with importlib:  ``dlco = importlib.import_module("opencv-dlco_amd")``.

This module is plumbing only: every computation happens in the HIP kernels of
``libdlco.so`` (sources under ``csrc/``).  There is no CPU fallback — a missing library or
a missing gfx950 device raises.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libdlco.so")
HEADER_PATH = os.path.join(os.path.dirname(_HERE), "include", "dlco.h")

OK, ERR_INVALID, ERR_HIP, ERR_NODEVICE, ERR_NOCONV, ERR_COMM = 0, -2, -3, -4, -5, -6
BUF_DIST, BUF_GRAD, BUF_DFAVG, BUF_W, BUF_GATHER, BUF_DATA = 1, 2, 3, 4, 5, 6
ALLGATHER_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int32, C.c_size_t)


class DlcoError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("libdlco error %d: %s" % (code, msg))
        self.code = code


class Cfg(C.Structure):
    _fields_ = [
        ("F", C.c_int32), ("N", C.c_int32), ("B", C.c_int32),
        ("mu", C.c_float), ("gamma", C.c_float),
        ("seed", C.c_uint64),
        ("device", C.c_int32), ("rank", C.c_int32), ("world", C.c_int32),
        ("eig_tol", C.c_float), ("eig_guard", C.c_int32), ("eig_max_iter", C.c_int32),
        ("shard", C.c_int32), ("strict_conv", C.c_int32), ("grad_bf16", C.c_int32),
        ("reserved", C.c_int32 * 5),
    ]


class LogEntry(C.Structure):
    _fields_ = [
        ("t", C.c_uint32), ("is_best", C.c_int32), ("saved", C.c_int32),
        ("loss_val", C.c_float), ("regul", C.c_float), ("obj", C.c_float), ("obj_best", C.c_float),
        ("rank", C.c_int32), ("rank_best", C.c_int32), ("dim", C.c_int32),
        ("auc", C.c_double), ("auc_best", C.c_double),
        ("fpr95", C.c_float), ("fpr95_best", C.c_float),
        ("vtime", C.c_double),
        ("nonconv", C.c_int32), ("reserved", C.c_int32),
    ]


def build(verbose=False):
    """Compile libdlco.so in-tree with hipcc for gfx950 (cross-compiles without a GPU)."""
    cmd = ["make", "-C", os.path.join(_HERE, "csrc"), "-j8"]
    if not verbose:
        cmd.insert(1, "-s")
    subprocess.check_call(cmd)


_lib = None
f32p, i32p, u8p = C.POINTER(C.c_float), C.POINTER(C.c_int32), C.POINTER(C.c_uint8)


def load():
    """dlopen libdlco.so; raises if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise DlcoError(ERR_NODEVICE, "%s is missing: run __graft_entry__.build() (no CPU fallback exists)" % LIB_PATH)
    L = C.CDLL(LIB_PATH)
    vp = C.c_void_p
    L.dlco_version.restype = C.c_char_p
    L.dlco_last_error.restype = C.c_char_p
    L.dlco_last_error.argtypes = [vp]
    L.dlco_cfg_default.argtypes = [C.POINTER(Cfg)]
    L.dlco_cfg_default.restype = None
    L.dlco_ctx_create.argtypes = [C.POINTER(vp), C.POINTER(Cfg)]
    L.dlco_ctx_destroy.argtypes = [vp]
    L.dlco_ctx_destroy.restype = None
    L.dlco_device_name.argtypes = [vp, C.c_char_p, C.c_size_t, C.POINTER(C.c_int), C.POINTER(C.c_int)]
    L.dlco_set_data.argtypes = [vp, f32p, u8p]
    L.dlco_set_data_device.argtypes = [vp, vp, u8p]
    L.dlco_set_data_shared.argtypes = [vp, vp]
    L.dlco_device_width.argtypes = [vp]
    L.dlco_set_pairs.argtypes = [vp, f32p, C.c_int32, i32p]
    L.dlco_synth_data.argtypes = [vp, f32p, C.c_int32, C.c_uint64, C.c_float, C.c_float, C.c_float, C.c_float]
    L.dlco_get_rows.argtypes = [vp, C.c_int32, C.c_int32, f32p]
    L.dlco_get_index.argtypes = [vp, i32p, i32p, i32p, i32p, i32p, i32p]
    for name in ("dlco_step", "dlco_step_begin", "dlco_step_grad", "dlco_step_finish", "dlco_sync"):
        getattr(L, name).argtypes = [vp]
    L.dlco_steps.argtypes = [vp, C.c_int32]
    L.dlco_dev_buffer.argtypes = [vp, C.c_int32, C.POINTER(vp), C.POINTER(C.c_size_t)]
    L.dlco_bind_buffer.argtypes = [vp, C.c_int32, vp, C.c_size_t]
    L.dlco_stream.argtypes = [vp, C.POINTER(vp)]
    L.dlco_set_allgather.argtypes = [vp, ALLGATHER_FN, vp]
    L.dlco_comm_unique_id.argtypes = [vp, C.c_size_t, C.c_char_p]
    L.dlco_comm_init.argtypes = [vp, vp, C.c_size_t, C.c_char_p]
    L.dlco_comm_destroy.argtypes = [vp]
    L.dlco_get_batch.argtypes = [vp, i32p, i32p, f32p, f32p, i32p, i32p]
    L.dlco_get_t.argtypes = [vp, C.POINTER(C.c_uint32)]
    L.dlco_get_W.argtypes = [vp, f32p, i32p]
    L.dlco_get_A.argtypes = [vp, f32p]
    L.dlco_get_dfavg.argtypes = [vp, f32p]
    L.dlco_set_state.argtypes = [vp, C.c_uint32, f32p, f32p, C.c_int32]
    L.dlco_validate.argtypes = [vp, f32p, f32p, i32p]
    L.dlco_stats.argtypes = [vp, f32p, C.c_int32, i32p, f32p, C.POINTER(C.c_double)]
    L.dlco_project_sqdist.argtypes = [vp, i32p, C.c_int32, f32p, C.c_int32, f32p]
    L.dlco_viol_counts.argtypes = [vp, f32p, f32p, C.c_int32, i32p, i32p]
    L.dlco_grad_rda.argtypes = [vp, i32p, i32p, i32p, i32p, C.c_int32, C.c_float, C.c_float, f32p, f32p]
    L.dlco_psd_project.argtypes = [vp, f32p, C.c_uint32, f32p, i32p, f32p]
    L.dlco_hinge_sum.argtypes = [vp, f32p, C.c_int32, f32p, C.c_int32, C.POINTER(C.c_double)]
    L.dlco_sym_product.argtypes = [vp, f32p, C.c_int32, f32p, C.c_int32, f32p]
    L.dlco_roc_stats.argtypes = [vp, f32p, u8p, C.c_int32, f32p, C.POINTER(C.c_double)]
    L.dlco_log_step.argtypes = [vp, C.POINTER(LogEntry)]
    L.dlco_get_saved.argtypes = [vp, f32p, i32p, f32p]
    L.dlco_profile_enable.argtypes = [vp, C.c_int32]
    L.dlco_profile_read.argtypes = [vp, C.c_char_p, C.POINTER(C.c_int64), C.POINTER(C.c_double)]
    L.dlco_eig_stats.argtypes = [vp, C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.POINTER(C.c_int64), i32p]
    L.dlco_counters.argtypes = [vp, C.POINTER(C.c_int64)]
    i64 = C.c_int64
    L.dlco_set_pairs_device.argtypes = [vp, vp, C.c_int32, i32p]
    L.dlco_desc_last_error.restype = C.c_char_p
    L.dlco_desc_last_error.argtypes = [vp]
    L.dlco_desc_create.argtypes = [C.POINTER(vp), C.c_float, C.c_int32, C.c_int32, C.c_int32]
    L.dlco_desc_destroy.argtypes = [vp]
    L.dlco_desc_destroy.restype = None
    L.dlco_desc_select_filters.argtypes = [f32p, C.c_int32, C.c_int32, f32p, C.c_int32, f32p, C.POINTER(C.c_int32)]
    L.dlco_desc_set_filters.argtypes = [vp, f32p, C.c_int32]
    L.dlco_desc_size.argtypes = [vp]
    L.dlco_desc_size.restype = C.c_int32
    L.dlco_desc_transform.argtypes = [vp, u8p, f32p]
    L.dlco_desc_compute.argtypes = [vp, u8p, i64, f32p]
    L.dlco_desc_compute_device.argtypes = [vp, u8p, i64, vp, i64]
    L.dlco_desc_pair_dists.argtypes = [vp, u8p, i64, i32p, i64, f32p, u8p]
    L.dlco_desc_full_dists.argtypes = [vp, u8p, i64, i32p, i64, f32p, u8p]
    L.dlco_desc_last_kernel_ms.argtypes = [vp]
    L.dlco_desc_last_kernel_ms.restype = C.c_double
    L.dlco_pr_last_error.restype = C.c_char_p
    L.dlco_pr_last_error.argtypes = [vp]
    L.dlco_pr_create.argtypes = [C.POINTER(vp), C.c_int32, C.c_int32, C.c_float, C.c_float, C.c_uint64, C.c_int32]
    L.dlco_pr_destroy.argtypes = [vp]
    L.dlco_pr_destroy.restype = None
    L.dlco_pr_device_name.argtypes = [vp, C.c_char_p, C.c_size_t, C.POINTER(C.c_int), C.POINTER(C.c_int)]
    L.dlco_pr_set_data.argtypes = [vp, f32p, u8p]
    L.dlco_pr_get_index.argtypes = [vp, i32p, i32p, i32p, i32p]
    L.dlco_pr_steps.argtypes = [vp, C.c_uint32]
    L.dlco_pr_get_state.argtypes = [vp, C.POINTER(C.c_uint32), f32p, f32p]
    L.dlco_pr_set_state.argtypes = [vp, C.c_uint32, f32p, f32p]
    L.dlco_pr_validate.argtypes = [vp, f32p, f32p, i32p]
    L.dlco_pr_stats.argtypes = [vp, f32p, f32p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, i32p, i32p, i32p, f32p, C.POINTER(C.c_double)]
    _lib = L
    return L


def comm_unique_id(rccl_path=None):
    """128-byte ncclUniqueId for dlco_comm_init (rank 0 creates it, every rank receives the same bytes)."""
    L = load()
    buf = C.create_string_buffer(128)
    rc = L.dlco_comm_unique_id(buf, 128, rccl_path.encode() if rccl_path else None)
    if rc != OK:
        raise DlcoError(rc, L.dlco_last_error(None).decode())
    return buf.raw


def exported_symbols():
    """Names declared in include/dlco.h (used by the ABI test)."""
    import re
    text = open(HEADER_PATH).read()
    return sorted(set(re.findall(r"\b(dlco_[a-z0-9_]+)\s*\(", text)))


def _p(a, t):
    return None if a is None else a.ctypes.data_as(t)


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _i32(a):
    return np.ascontiguousarray(a, dtype=np.int32)


class Context:
    """One pj-learn trainer on one GPU (thin wrapper over dlco_ctx)."""

    def __init__(self, F, N, B=200, mu=0.001, gamma=0.5, seed=2215, device=0, rank=0, world=1,
                 eig_tol=None, eig_guard=None, eig_max_iter=None, shard=0, strict_conv=0, grad_bf16=0):
        self.L = load()
        cfg = Cfg()
        self.L.dlco_cfg_default(C.byref(cfg))
        cfg.F, cfg.N, cfg.B, cfg.mu, cfg.gamma, cfg.seed = F, N, B, mu, gamma, seed
        cfg.device, cfg.rank, cfg.world, cfg.shard = device, rank, world, shard
        cfg.strict_conv = strict_conv
        cfg.grad_bf16 = grad_bf16
        if eig_tol is not None:
            cfg.eig_tol = eig_tol
        if eig_guard is not None:
            cfg.eig_guard = eig_guard
        if eig_max_iter is not None:
            cfg.eig_max_iter = eig_max_iter
        self.F, self.N, self.B, self.world, self.rank = F, N, B, world, rank
        h = C.c_void_p()
        rc = self.L.dlco_ctx_create(C.byref(h), C.byref(cfg))
        if rc != OK:
            raise DlcoError(rc, self.L.dlco_last_error(None).decode())
        self.h = h

    def close(self):
        if getattr(self, "h", None):
            self.L.dlco_ctx_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _ck(self, rc):
        if rc != OK:
            raise DlcoError(rc, self.L.dlco_last_error(self.h).decode())

    # ---- data -------------------------------------------------------------------------------
    def device_name(self):
        buf = C.create_string_buffer(256)
        a, b = C.c_int(), C.c_int()
        self._ck(self.L.dlco_device_name(self.h, buf, 256, C.byref(a), C.byref(b)))
        return buf.value.decode(), a.value, b.value

    def set_data(self, dists, labels):
        d, l = _f32(dists), np.ascontiguousarray(labels, np.uint8).ravel()
        assert d.shape == (self.N, self.F) and l.size == self.N
        self._ck(self.L.dlco_set_data(self.h, _p(d, f32p), _p(l, u8p)))

    def set_data_device(self, dev_ptr, labels):
        l = np.ascontiguousarray(labels, np.uint8).ravel()
        self._ck(self.L.dlco_set_data_device(self.h, C.c_void_p(dev_ptr), _p(l, u8p)))

    def set_data_shared(self, other):
        """Train on the resident matrix of another context (same device, F, N): shared, not copied."""
        self._ck(self.L.dlco_set_data_shared(self.h, other.h))

    def device_width(self):
        """Row width on the device: F rounded up to whole 128-column tiles (see dlco_device_width)."""
        return int(self.L.dlco_device_width(self.h))

    def set_pairs(self, desc, pairs):
        """Pair mode: per-patch descriptors [P,F] + the [N,4] Indices table (see dlco_set_pairs)."""
        d, q = _f32(desc), _i32(pairs)
        assert d.ndim == 2 and d.shape[1] == self.F and q.shape == (self.N, 4)
        self._ck(self.L.dlco_set_pairs(self.h, _p(d, f32p), d.shape[0], _p(q, i32p)))

    def set_pairs_device(self, desc_dev_ptr, P, pairs):
        """Pair mode on a descriptor table that already lives in device memory (dlco_set_pairs_device)."""
        q = _i32(pairs)
        assert q.shape == (self.N, 4)
        self._ck(self.L.dlco_set_pairs_device(self.h, C.c_void_p(desc_dev_ptr), P, _p(q, i32p)))

    def synth_data(self, U, seed, sigma_pos, sigma_neg, noise, scale_jitter=0.0):
        U = _f32(U)
        assert U.shape[1] == self.F
        self._ck(self.L.dlco_synth_data(self.h, _p(U, f32p), U.shape[0], seed, sigma_pos, sigma_neg, noise, scale_jitter))

    def get_rows(self, row0, n):
        out = np.empty((n, self.F), np.float32)
        self._ck(self.L.dlco_get_rows(self.h, row0, n, _p(out, f32p)))
        return out

    def index(self):
        a, b, c_, d = C.c_int32(), C.c_int32(), C.c_int32(), C.c_int32()
        self._ck(self.L.dlco_get_index(self.h, None, C.byref(a), C.byref(b), None, C.byref(c_), C.byref(d)))
        pos, neg = np.empty(max(a.value, 1), np.int32), np.empty(max(c_.value, 1), np.int32)
        self._ck(self.L.dlco_get_index(self.h, _p(pos, i32p), None, None, _p(neg, i32p), None, None))
        return dict(pos=pos[:a.value], n_pos_trn=b.value, neg=neg[:c_.value], n_neg_trn=d.value)

    # ---- training ---------------------------------------------------------------------------
    def step(self):
        self._ck(self.L.dlco_step(self.h))

    def steps(self, n):
        self._ck(self.L.dlco_steps(self.h, n))

    def step_begin(self):
        self._ck(self.L.dlco_step_begin(self.h))

    def step_grad(self):
        self._ck(self.L.dlco_step_grad(self.h))

    def step_finish(self):
        self._ck(self.L.dlco_step_finish(self.h))

    def sync(self):
        self._ck(self.L.dlco_sync(self.h))

    def dev_buffer(self, which):
        p, n = C.c_void_p(), C.c_size_t()
        self._ck(self.L.dlco_dev_buffer(self.h, which, C.byref(p), C.byref(n)))
        return p.value, n.value

    def bind_buffer(self, which, dev_ptr, nbytes):
        self._ck(self.L.dlco_bind_buffer(self.h, which, C.c_void_p(dev_ptr), nbytes))

    def stream(self):
        p = C.c_void_p()
        self._ck(self.L.dlco_stream(self.h, C.byref(p)))
        return p.value

    def comm_init(self, id_bytes, rccl_path=None):
        """Collective: creates the library's own RCCL communicator from a 128-byte ncclUniqueId."""
        buf = C.create_string_buffer(bytes(id_bytes), 128)
        self._ck(self.L.dlco_comm_init(self.h, buf, 128, rccl_path.encode() if rccl_path else None))

    def comm_destroy(self):
        self._ck(self.L.dlco_comm_destroy(self.h))

    def set_allgather(self, fn):
        """fn(which, bytes_per_rank) -> 0 on success; kept alive by the context (see dlco_set_allgather)."""
        def tramp(_user, which, nbytes):
            try:
                return int(fn(int(which), int(nbytes)) or 0)
            except Exception:                    # an exception must not unwind through the C frames
                import traceback
                traceback.print_exc()
                return -1
        self._ag = ALLGATHER_FN(tramp)
        self._ck(self.L.dlco_set_allgather(self.h, self._ag, None))

    def batch(self):
        B = self.B
        pr, nr = np.empty(B, np.int32), np.empty(B, np.int32)
        pd, nd = np.empty(B, np.float32), np.empty(B, np.float32)
        rho, kap = np.empty(B, np.int32), np.empty(B, np.int32)
        self._ck(self.L.dlco_get_batch(self.h, _p(pr, i32p), _p(nr, i32p), _p(pd, f32p), _p(nd, f32p), _p(rho, i32p), _p(kap, i32p)))
        return dict(pos_rows=pr, neg_rows=nr, pd=pd, nd=nd, rho=rho, kappa=kap)

    def t(self):
        v = C.c_uint32()
        self._ck(self.L.dlco_get_t(self.h, C.byref(v)))
        return v.value

    def W(self):
        W = np.empty((self.F, self.F), np.float32)
        r = C.c_int32()
        self._ck(self.L.dlco_get_W(self.h, _p(W, f32p), C.byref(r)))
        return W[:r.value].copy()

    def A(self):
        A = np.empty((self.F, self.F), np.float32)
        self._ck(self.L.dlco_get_A(self.h, _p(A, f32p)))
        return A

    def dfavg(self):
        A = np.empty((self.F, self.F), np.float32)
        self._ck(self.L.dlco_get_dfavg(self.h, _p(A, f32p)))
        return A

    def set_state(self, t, dfavg=None, W=None):
        d = None if dfavg is None else _f32(dfavg)
        w = None if W is None or len(W) == 0 else _f32(W)
        self._ck(self.L.dlco_set_state(self.h, t, _p(d, f32p), _p(w, f32p), 0 if w is None else w.shape[0]))

    def validate(self):
        lo, rg, rk = C.c_float(), C.c_float(), C.c_int32()
        self._ck(self.L.dlco_validate(self.h, C.byref(lo), C.byref(rg), C.byref(rk)))
        return lo.value, rg.value, rk.value

    def stats(self, W=None):
        dim, f, a = C.c_int32(), C.c_float(), C.c_double()
        w = None if W is None else _f32(W)
        self._ck(self.L.dlco_stats(self.h, _p(w, f32p), 0 if w is None else w.shape[0], C.byref(dim), C.byref(f), C.byref(a)))
        return dim.value, f.value, a.value

    def log_step(self):
        e = LogEntry()
        self._ck(self.L.dlco_log_step(self.h, C.byref(e)))
        return e

    def saved(self):
        r = C.c_int32()
        self._ck(self.L.dlco_get_saved(self.h, None, C.byref(r), None))
        if r.value == 0:
            return None, None
        W, A = np.empty((r.value, self.F), np.float32), np.empty((self.F, self.F), np.float32)
        self._ck(self.L.dlco_get_saved(self.h, _p(W, f32p), C.byref(r), _p(A, f32p)))
        return W, A

    # ---- single operators -------------------------------------------------------------------
    def project_sqdist(self, row_ids, W):
        ids, w = _i32(row_ids), _f32(W)
        out = np.empty(ids.size, np.float32)
        self._ck(self.L.dlco_project_sqdist(self.h, _p(ids, i32p), ids.size, _p(w, f32p), w.shape[0] if w.size else 0, _p(out, f32p)))
        return out

    def viol_counts(self, pd, nd):
        pd, nd = _f32(pd), _f32(nd)
        rho, kap = np.empty(pd.size, np.int32), np.empty(pd.size, np.int32)
        self._ck(self.L.dlco_viol_counts(self.h, _p(pd, f32p), _p(nd, f32p), pd.size, _p(rho, i32p), _p(kap, i32p)))
        return rho, kap

    def grad_rda(self, pos_rows, neg_rows, rho, kappa, alpha, beta, dfavg_in=None):
        pr, nr, rho, kap = _i32(pos_rows), _i32(neg_rows), _i32(rho), _i32(kappa)
        d = None if dfavg_in is None else _f32(dfavg_in)
        out = np.empty((self.F, self.F), np.float32)
        self._ck(self.L.dlco_grad_rda(self.h, _p(pr, i32p), _p(nr, i32p), _p(rho, i32p), _p(kap, i32p), pr.size, alpha, beta, _p(d, f32p), _p(out, f32p)))
        return out

    def psd_project(self, dfavg, t, want_A=True):
        d = _f32(dfavg)
        W = np.empty((self.F, self.F), np.float32)
        A = np.empty((self.F, self.F), np.float32) if want_A else None
        r = C.c_int32()
        self._ck(self.L.dlco_psd_project(self.h, _p(d, f32p), t, _p(W, f32p), C.byref(r), _p(A, f32p)))
        return W[:r.value].copy(), A

    def sym_product(self, X, G, mode=0):
        x, g = _f32(X), _f32(G)
        out = np.empty_like(x)
        self._ck(self.L.dlco_sym_product(self.h, _p(x, f32p), x.shape[0], _p(g, f32p), mode, _p(out, f32p)))
        return out

    def hinge_sum(self, pos, neg):
        p, n = _f32(pos), _f32(neg)
        out = C.c_double()
        self._ck(self.L.dlco_hinge_sum(self.h, _p(p, f32p), p.size, _p(n, f32p), n.size, C.byref(out)))
        return out.value

    def roc_stats(self, dist, labels):
        d, l = _f32(dist), np.ascontiguousarray(labels, np.uint8).ravel()
        f, a = C.c_float(), C.c_double()
        self._ck(self.L.dlco_roc_stats(self.h, _p(d, f32p), _p(l, u8p), d.size, C.byref(f), C.byref(a)))
        return f.value, a.value

    # ---- measurement ------------------------------------------------------------------------
    def profile_enable(self, on=True):
        self._ck(self.L.dlco_profile_enable(self.h, int(on)))          # True / 1: every kernel group, 2: the gradient SYRK only

    def profile_read(self, kernel="grad_syrk"):
        n, ms = C.c_int64(), C.c_double()
        self._ck(self.L.dlco_profile_read(self.h, kernel.encode(), C.byref(n), C.byref(ms)))
        return n.value, ms.value

    def counters(self):
        out = (C.c_int64 * 8)()
        self._ck(self.L.dlco_counters(self.h, out))
        return dict(steps=out[0], active_rows=out[1], nonconverged=out[2], jacobi_barrier_timeouts=out[3],
                    rank_update_passes=out[4], rank_update_check=out[5] * 1e-9, locked_passes=out[6], locked_rows=out[7])

    def eig_stats(self):
        a, b, c_, d = C.c_int64(), C.c_int64(), C.c_int64(), C.c_int32()
        self._ck(self.L.dlco_eig_stats(self.h, C.byref(a), C.byref(b), C.byref(c_), C.byref(d)))
        return dict(iters=a.value, product_rows=b.value, jacobi_sweeps=c_.value, block_rows=d.value)


class PrContext:
    """pr-learn: L1-regularised dual averaging on the pooling-region weights (thin wrapper over dlco_pr_ctx)."""

    def __init__(self, F, N, mu=0.025, gamma=0.10, seed=2215, device=0):
        self.L = load()
        self.F, self.N = F, N
        h = C.c_void_p()
        rc = self.L.dlco_pr_create(C.byref(h), F, N, mu, gamma, seed, device)
        if rc != OK:
            raise DlcoError(rc, self.L.dlco_pr_last_error(None).decode())
        self.h = h

    def close(self):
        if getattr(self, "h", None):
            self.L.dlco_pr_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _ck(self, rc):
        if rc != OK:
            raise DlcoError(rc, self.L.dlco_pr_last_error(self.h).decode())

    def set_data(self, dists, labels):
        d, l = _f32(dists), np.ascontiguousarray(labels, np.uint8).ravel()
        assert d.shape == (self.N, self.F) and l.size == self.N
        self._ck(self.L.dlco_pr_set_data(self.h, _p(d, f32p), _p(l, u8p)))

    def index(self):
        a, b, c_, d = C.c_int32(), C.c_int32(), C.c_int32(), C.c_int32()
        self._ck(self.L.dlco_pr_get_index(self.h, C.byref(a), C.byref(b), C.byref(c_), C.byref(d)))
        return dict(n_pos=a.value, n_pos_trn=b.value, n_neg=c_.value, n_neg_trn=d.value)

    def steps(self, n):
        self._ck(self.L.dlco_pr_steps(self.h, n))

    def state(self):
        t = C.c_uint32()
        w, df = np.empty(self.F, np.float32), np.empty(self.F, np.float32)
        self._ck(self.L.dlco_pr_get_state(self.h, C.byref(t), _p(w, f32p), _p(df, f32p)))
        return dict(t=t.value, w=w, dfavg=df)

    def set_state(self, t, w=None, dfavg=None):
        w = None if w is None else _f32(w)
        d = None if dfavg is None else _f32(dfavg)
        self._ck(self.L.dlco_pr_set_state(self.h, t, _p(w, f32p), _p(d, f32p)))

    def validate(self):
        lo, rg, nz = C.c_float(), C.c_float(), C.c_int32()
        self._ck(self.L.dlco_pr_validate(self.h, C.byref(lo), C.byref(rg), C.byref(nz)))
        return lo.value, rg.value, nz.value

    def stats(self, prparams, w=None, nchannels=8, max_dim=-1):
        p = _f32(prparams)
        wv = None if w is None else _f32(w)
        npr, dim, nz, f, a = C.c_int32(), C.c_int32(), C.c_int32(), C.c_float(-1.0), C.c_double(0.0)
        self._ck(self.L.dlco_pr_stats(self.h, _p(wv, f32p), _p(p, f32p), p.shape[0], p.shape[1], nchannels, max_dim,
                                      C.byref(npr), C.byref(dim), C.byref(nz), C.byref(f), C.byref(a)))
        return dict(nPR=npr.value, dim=dim.value, nzdim=nz.value, fpr95=f.value, auc=a.value)


def select_filters(pr_filters, w):
    """SelectPRFilters (src/misc.cpp:78-168): host logic of the library, no device needed."""
    L = load()
    f, wv = _f32(pr_filters), _f32(w).ravel()
    assert f.ndim == 2 and f.shape[0] == 8 * wv.size
    n = C.c_int32()
    rc = L.dlco_desc_select_filters(_p(f, f32p), f.shape[0], f.shape[1], _p(wv, f32p), wv.size, None, C.byref(n))
    if rc != OK:
        raise DlcoError(rc, "dlco_desc_select_filters")
    out = np.empty((n.value, f.shape[1]), np.float32)
    if n.value:
        L.dlco_desc_select_filters(_p(f, f32p), f.shape[0], f.shape[1], _p(wv, f32p), wv.size, _p(out, f32p), C.byref(n))
    return out


class DescContext:
    """comp-uprjdists on the GPU: get_desc + pooling product per patch (thin wrapper over dlco_desc_ctx)."""

    def __init__(self, init_sigma=1.4, n_angle_bins=8, norm=True, device=0):
        self.L = load()
        h = C.c_void_p()
        rc = self.L.dlco_desc_create(C.byref(h), init_sigma, n_angle_bins, 1 if norm else 0, device)
        if rc != OK:
            raise DlcoError(rc, self.L.dlco_desc_last_error(None).decode())
        self.h = h

    def close(self):
        if getattr(self, "h", None):
            self.L.dlco_desc_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _ck(self, rc):
        if rc != OK:
            raise DlcoError(rc, self.L.dlco_desc_last_error(self.h).decode())

    def set_filters(self, sPR):
        f = _f32(sPR)
        assert f.ndim == 2 and f.shape[1] == 4096
        self._ck(self.L.dlco_desc_set_filters(self.h, _p(f, f32p), f.shape[0]))

    @property
    def size(self):
        return self.L.dlco_desc_size(self.h)

    def transform(self, patch):
        p = np.ascontiguousarray(patch, np.uint8)
        assert p.shape == (64, 64)
        out = np.empty((4096, 8), np.float32)
        self._ck(self.L.dlco_desc_transform(self.h, _p(p, u8p), _p(out, f32p)))
        return out

    def compute(self, patches):
        p = np.ascontiguousarray(patches, np.uint8)
        assert p.ndim == 3 and p.shape[1:] == (64, 64)
        out = np.empty((p.shape[0], self.size), np.float32)
        self._ck(self.L.dlco_desc_compute(self.h, _p(p, u8p), p.shape[0], _p(out, f32p)))
        return out

    def compute_device(self, patches, dev_ptr, ld=None):
        p = np.ascontiguousarray(patches, np.uint8)
        assert p.ndim == 3 and p.shape[1:] == (64, 64)
        self._ck(self.L.dlco_desc_compute_device(self.h, _p(p, u8p), p.shape[0], C.c_void_p(dev_ptr), ld or self.size))

    def pair_dists(self, patches, pairs):
        p, q = np.ascontiguousarray(patches, np.uint8), _i32(pairs)
        assert p.ndim == 3 and p.shape[1:] == (64, 64) and q.ndim == 2 and q.shape[1] == 4
        dist = np.empty((q.shape[0], self.size), np.float32)
        lab = np.empty(q.shape[0], np.uint8)
        self._ck(self.L.dlco_desc_pair_dists(self.h, _p(p, u8p), p.shape[0], _p(q, i32p), q.shape[0], _p(dist, f32p), _p(lab, u8p)))
        return dist, lab

    def full_dists(self, patches, pairs):
        """comp-fulldists: per-region squared descriptor distances [n_pairs, rows/8] and labels (pr-learn's input)."""
        p, q = np.ascontiguousarray(patches, np.uint8), _i32(pairs)
        assert p.ndim == 3 and p.shape[1:] == (64, 64) and q.ndim == 2 and q.shape[1] == 4
        dist = np.empty((q.shape[0], max(1, self.size // 64)), np.float32)
        lab = np.empty(q.shape[0], np.uint8)
        self._ck(self.L.dlco_desc_full_dists(self.h, _p(p, u8p), p.shape[0], _p(q, i32p), q.shape[0], _p(dist, f32p), _p(lab, u8p)))
        return dist, lab

    def last_kernel_ms(self):
        return self.L.dlco_desc_last_kernel_ms(self.h)
