#!/usr/bin/env python3
"""Generates the fixtures under tests/golden/.  Run in the build container
(where /root/reference is mounted); the GPU box only ever sees the outputs.

  ref_results.npz   W (and one A) read from the reference's committed result
                    files workspace/pj-learn/*.h5, plus the numbers its logs
                    print for the last "[saved]" entry.  These are DATA the
                    reference holds (outputs of its own runs), converted from
                    HDF5 because h5py is not available on the test box.
  ref_log_head.txt  first lines of one reference log (stdout grammar fixture).
  export_NN.npz, vgg_generated_NN.i.gz  (NN = 48, 64, 80, 120)
                    the four "vgg_generated_XX.i" headers the reference ships, each with
                    the W (from its result file) and pooling-region filters (recovered
                    from the header itself) it was written from: golden vectors of
                    the export format (see make_export_fixture).
  pr_saved.npz      rows of "w" of three pr-learn result files with the Regul / NNZ /
                    nzDim its logs print for them (see make_pr_saved_fixture).
  oracle_*.npz      seeded input/output vectors produced by the CPU oracle
                    (oracle/dlco_ref.c).  They pin nothing against the
                    reference by themselves; they freeze the restatement so a
                    change in it is noticed, and give the GPU tests fixed
                    inputs.
"""
import ctypes as C
import os
import re
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
REF = "/root/reference/workspace/pj-learn"


def h5_reader():
    h5 = C.CDLL("/opt/conda/lib/libhdf5.so")
    h5.H5open()
    hid = C.c_int64
    h5.H5Fopen.restype = hid
    h5.H5Fopen.argtypes = [C.c_char_p, C.c_uint, hid]
    h5.H5Dopen2.restype = hid
    h5.H5Dopen2.argtypes = [hid, C.c_char_p, hid]
    h5.H5Dget_space.restype = hid
    h5.H5Dget_space.argtypes = [hid]
    h5.H5Sget_simple_extent_ndims.argtypes = [hid]
    h5.H5Sget_simple_extent_dims.argtypes = [hid, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
    h5.H5Dread.argtypes = [hid, hid, hid, hid, hid, C.c_void_p]
    h5.H5Dclose.argtypes = [hid]
    h5.H5Fclose.argtypes = [hid]
    h5.H5Sclose.argtypes = [hid]
    h5.H5Lexists.argtypes = [hid, C.c_char_p, hid]
    native_float = hid.in_dll(h5, "H5T_NATIVE_FLOAT_g").value

    def read(path, name):
        f = h5.H5Fopen(path.encode(), 0, 0)
        assert f >= 0, path
        if h5.H5Lexists(f, name.encode(), 0) <= 0:
            h5.H5Fclose(f)
            return None
        d = h5.H5Dopen2(f, name.encode(), 0)
        s = h5.H5Dget_space(d)
        nd = h5.H5Sget_simple_extent_ndims(s)
        dims = (C.c_uint64 * nd)()
        h5.H5Sget_simple_extent_dims(s, dims, None)
        arr = np.empty(tuple(dims), np.float32)
        assert h5.H5Dread(d, native_float, 0, 0, 0, arr.ctypes.data) >= 0
        h5.H5Sclose(s)
        h5.H5Dclose(d)
        h5.H5Fclose(f)
        return arr

    return read


def last_saved(logpath):
    """(mu, gamma, step, Loss, Regul, Rank, Dim, AUC, FPR95) of the last [saved] entry."""
    lines = open(logpath).read().splitlines()
    m = re.match(r"mu: (\S+) gamma: (\S+)", lines[0])
    mu, gamma = float(m.group(1)), float(m.group(2))
    for i in range(len(lines) - 1, -1, -1):
        if "[saved]" in lines[i]:
            s = re.match(r"Stat: Dim \[(\d+)\] AUC: (\S+) \((\S+)\) FPR95: (\S+) \((\S+)\)", lines[i])
            b = re.match(r"Best: (\d+)  Loss: (\S+) Regul: (\S+) Obj: (\S+) \((\S+)\) Rank: (\d+) \((\d+)\)", lines[i - 1])
            return dict(mu=mu, gamma=gamma, step=int(b.group(1)), loss=float(b.group(2)), regul=float(b.group(3)),
                        rank=int(b.group(6)), dim=int(s.group(1)), auc=float(s.group(2)), fpr95=float(s.group(4)))
    raise RuntimeError("no saved entry in " + logpath)


def make_ref_results():
    read = h5_reader()
    picks = [
        ("liberty-liberty-0.035-0.250-pr#7-0.0010-0.100-pj", True),
        ("notredame-notredame-0.003-0.040-pr#7-0.0010-0.250-pj", False),
        ("yosemite-yosemite-0.025-0.075-pr#7-0.0020-0.500-pj", False),
    ]
    avail = sorted(f[:-3] for f in os.listdir(REF) if f.endswith("-pj.h5"))
    out = {}
    names = []
    for want, keep_a in picks:
        name = want if want in avail else next(a for a in avail if a.startswith(want.split("-")[0]) and a not in names)
        names.append(name)
        W = read(os.path.join(REF, name + ".h5"), "W")
        A = read(os.path.join(REF, name + ".h5"), "A")
        info = last_saved(os.path.join(REF, "logging", name + ".log"))
        # invariants checked at generation time on the full-precision data
        rel = np.abs(A - W.T.astype(np.float64) @ W.astype(np.float64)).max() / np.abs(A).max()
        assert rel < 1e-6, rel
        assert W.shape[0] == info["dim"] == info["rank"]
        key = "r%d" % (len(names) - 1)
        out[key + "_W"] = W
        if keep_a:
            out[key + "_A"] = A
        out[key + "_traceA"] = np.float64(np.trace(A.astype(np.float64)))
        out[key + "_info"] = np.array([info[k] for k in ("mu", "gamma", "step", "loss", "regul", "rank", "dim", "auc", "fpr95")], np.float64)
        out[key + "_name"] = np.array(name)
    orig = read(os.path.join(REF, "originals", "liberty-rank_m0.002_g1.h5"), "W")
    out["orig_W"] = orig
    np.savez_compressed(os.path.join(HERE, "ref_results.npz"), **out)
    with open(os.path.join(REF, "logging", names[0] + ".log")) as f:
        head = f.read().splitlines()[:20]
    with open(os.path.join(HERE, "ref_log_head.txt"), "w") as f:
        f.write("\n".join(head) + "\n")
    print("ref_results:", names, "orig", orig.shape)


def synth(N, F, k=16, seed=2215, sp=0.6, sn=1.0, noise=0.15):
    """Synthetic stand-in for a *-unproj.h5: d = U^T z + eps, clipped to [-1,1]."""
    rng = np.random.default_rng(seed)
    U = np.linalg.qr(rng.standard_normal((F, k)))[0].T.astype(np.float32)
    labels = (np.arange(N) % 2 == 0).astype(np.uint8)
    z = rng.standard_normal((N, k)).astype(np.float32)
    z *= np.where(labels[:, None] == 1, sp, sn).astype(np.float32)
    d = z @ U + noise * rng.standard_normal((N, F)).astype(np.float32)
    return np.clip(d, -1, 1).astype(np.float32), labels


def make_oracle_vectors():
    from oracle import ref

    out = {}
    # RNG / index fixtures (R1-R3)
    for seed in (2215, 0xFFFFFFFF):
        r = ref.Rng(seed)
        out["rng_next_%d" % seed] = np.array([r.next() for _ in range(64)], np.uint64)
    r = ref.Rng(2215)
    ip, ineg = r.sample(200000, 200000, 200)
    out["sample_200k_pos"], out["sample_200k_neg"] = ip, ineg
    ip, ineg = r.sample(2000, 1999, 200)
    out["sample_2k_pos"], out["sample_2k_neg"] = ip, ineg
    out["shuffle16"] = ref.Rng(0xFFFFFFFF).shuffle(np.arange(16, dtype=np.int32))
    labels = (np.arange(500000) % 2 == 0).astype(np.uint8)
    pos, neg = ref.build_index(labels)
    out["idx500k_pos_head"], out["idx500k_pos_tail"] = pos[:8], pos[-8:]
    out["idx500k_neg_head"], out["idx500k_neg_tail"] = neg[:8], neg[-8:]
    out["idx500k_pos_sum"] = np.int64((pos.astype(np.int64) * (np.arange(pos.size) % 977)).sum())
    out["idx500k_neg_sum"] = np.int64((neg.astype(np.int64) * (np.arange(neg.size) % 977)).sum())
    out["split"] = np.array([ref.split(n) for n in (250000, 2500, 1, 0, 7, 1999)], np.int64)
    np.savez_compressed(os.path.join(HERE, "oracle_rng.npz"), **out)

    # step vectors at two shapes, both gradient orders
    for (N, F, B, k, mu, gamma, nstep) in ((600, 32, 8, 6, 0.01, 0.5, 4), (2000, 64, 40, 12, 0.005, 0.5, 6)):
        D, L = synth(N, F, k=k, seed=2215 + F)
        tr = ref.Trainer(D, L, B=B, mu=mu, gamma=gamma, grad_order=0)
        rec = dict(D=D, L=L, cfg=np.array([N, F, B, mu, gamma, nstep], np.float64))
        for s in range(nstep):
            before = tr.state()
            tr.step()
            after = tr.state()
            pr, nr = tr.batch_ids()
            pd, nd = tr.batch_dists()
            rho, kap = ref.viol_counts(pd, nd)
            rec["s%d_W_in" % s] = before["W"]
            rec["s%d_dfavg_in" % s] = before["dfavg"]
            rec["s%d_pos_rows" % s], rec["s%d_neg_rows" % s] = pr, nr
            rec["s%d_pd" % s], rec["s%d_nd" % s] = pd, nd
            rec["s%d_rho" % s], rec["s%d_kappa" % s] = rho, kap
            rec["s%d_dloss" % s] = after["dloss"]
            rec["s%d_dfavg" % s] = after["dfavg"]
            rec["s%d_A" % s] = after["A"]
            rec["s%d_W" % s] = after["W"]
        lo, rg = tr.validate()
        dim, f95, auc = tr.stats()
        rec["final"] = np.array([lo, rg, dim, f95, auc], np.float64)
        np.savez_compressed(os.path.join(HERE, "oracle_step_F%d_B%d.npz" % (F, B)), **rec)
    print("oracle vectors written (blas: %s)" % ref.blas_kind())


EXPORTS = {
    # dims: (pr-learn result file, row of w, pj-learn result file) as workspace/11-opencv-export.sh:9-27 calls the tool
    48: ("pr-learn/olderbest/yosemite-0.025-0.075-pr.h5", 7, "pj-learn/notredame-yosemite-0.025-0.075-pr#7-0.0020-0.200-pj.h5"),
    64: ("pr-learn/liberty-0.035-0.250-pr.h5", 7, "pj-learn/notredame-liberty-0.035-0.250-pr#7-0.0010-0.100-pj.h5"),
    80: ("pr-learn/liberty-0.035-0.250-pr.h5", 7, "pj-learn/notredame-liberty-0.035-0.250-pr#7-0.0005-0.100-pj.h5"),
    120: ("pr-learn/liberty-0.035-0.250-pr.h5", 7, "pj-learn/notredame-liberty-0.035-0.250-pr#7-0.0001-0.025-pj.h5"),
}


def make_export_fixture(dims=(48, 64, 80, 120)):
    """export_NN.npz + vgg_generated_NN.i.gz: the four headers the reference ships
    (workspace/opencv/vgg_generated_{48,64,80,120}.i, written by its export-opencv as workspace/11-opencv-export.sh
    calls it) together with their inputs: W read from the pj-learn result file named in the header, and the selected
    pooling-region filters recovered from the header's own sparse PR arrays (the filters.h5 they were made from is
    not in the repository).  Output file + inputs = a golden vector for the export format."""
    import gzip
    read = h5_reader()
    for dim in dims:
        prg, widx, prj = EXPORTS[dim]
        src = "/root/reference/workspace/opencv/vgg_generated_%d.i" % dim
        text = open(src, "rb").read()
        body = text.decode()

        def ints(name):
            m = re.search(r"static const unsigned int %s\[\] =\n\{(.*?)\};" % name, body, re.S)
            return [int(t, 16) for t in re.findall(r"0x[0-9a-fA-F]+", m.group(1))]
        assert ("// PR: [%s]#%d" % (prg, widx)) in body and ("// PJ: [%s]" % prj) in body
        rows = int(re.search(r"PRrows = (\d+);", body).group(1))
        cols = int(re.search(r"PRcols = (\d+);", body).group(1))
        idx, vals = ints("PRidx"), np.array(ints("PR"), np.uint32).view(np.float32)
        PR = np.zeros(rows * cols, np.float32)
        k = 0
        for start, count in zip(idx[0::2], idx[1::2]):
            PR[start:start + count] = vals[k:k + count]
            k += count
        assert k == vals.size
        W = read(os.path.join("/root/reference/workspace", prj), "W")
        assert W.shape == (dim, rows * 8)
        np.savez_compressed(os.path.join(HERE, "export_%d.npz" % dim), PR=PR.reshape(rows, cols), W=W,
                            prg=np.array(prg), widx=np.array(widx), prj=np.array(prj))
        with gzip.GzipFile(os.path.join(HERE, "vgg_generated_%d.i.gz" % dim), "wb", mtime=0) as f:
            f.write(text)
        print("export fixture %d: PR" % dim, PR.reshape(rows, cols).shape, "W", W.shape, "header bytes", len(text))


def make_pr_saved_fixture():
    """pr_saved.npz: the PR stage's own record of its runs.  pr-learn appends one row to the "w" dataset of its result
    file at every "[saved]" log line (src/pr-learn.cpp:385-400), and the "Best:" line before it prints
    Regul = mu * sum|w| and NNZ = countNonZero(w) of that w (:358,366-369).  For three runs: the rows of "w" and, per
    row, (t, Regul, NNZ, nzDim) as logged - a known-answer test for the regulariser / NNZ part of the oracle's
    dlco_ref_pr_validate and the nzDim count of ComputePRStats (nzDim = 8 * NNZ, src/misc.cpp:183-193,215)."""
    read = h5_reader()
    base = "/root/reference/workspace/pr-learn"
    out = {}
    names = ["liberty-0.035-0.250-pr", "notredame-0.003-0.040-pr", "yosemite-0.025-0.075-pr"]
    have = sorted(f[:-3] for f in os.listdir(base) if f.endswith("-pr.h5"))
    for i, want in enumerate(names):
        name = want if want in have else next(h for h in have if h.startswith(want.split("-")[0]))
        lines = open(os.path.join(base, "logging", name + ".log")).read().splitlines()
        mu = float(re.match(r"mu: (\S+) gamma: (\S+)", lines[0]).group(1))
        rec = []
        for j, l in enumerate(lines):
            if l.endswith("[saved]"):
                b = re.match(r"Best: (\d+)  Loss: (\S+) Regul: (\S+) Obj: (\S+) \((\S+)\)  NNZ: (\d+) \((\d+)\)", lines[j - 1])
                st = re.match(r"Stat: nPR #(\d+) \(#(\d+)\) Dim/MaxDim \[(\d+)/(\d+)\]", l)
                rec.append([int(b.group(1)), float(b.group(3)), int(b.group(6)), int(st.group(2)), int(st.group(1)), int(st.group(3))])
        w = read(os.path.join(base, name + ".h5"), "w")
        assert w.shape[0] == len(rec), (name, w.shape, len(rec))
        out["p%d_name" % i] = np.array(name)
        out["p%d_mu" % i] = np.float64(mu)
        out["p%d_w" % i] = w
        out["p%d_log" % i] = np.array(rec, np.float64)          # t, Regul, NNZ, nzDim, nPR, Dim
        print("pr fixture:", name, w.shape, "mu", mu)
    np.savez_compressed(os.path.join(HERE, "pr_saved.npz"), **out)


if __name__ == "__main__":
    only = sys.argv[1:]                       # e.g. `make_golden.py export pr` regenerates just those fixtures
    have_ref = os.path.isdir(REF)
    if not have_ref:
        print("reference not mounted: keeping the fixtures made from its files")
    if have_ref and (not only or "results" in only):
        make_ref_results()
    if have_ref and (not only or "export" in only):
        make_export_fixture()
    if have_ref and (not only or "pr" in only):
        make_pr_saved_fixture()
    if not only or "oracle" in only:
        make_oracle_vectors()
