#!/usr/bin/env python3
"""Generates the fixtures under tests/golden/.  Run in the build container
(where /root/reference is mounted); the GPU box only ever sees the outputs.

  ref_results.npz   W (and one A) read from the reference's committed result
                    files workspace/pj-learn/*.h5, plus the numbers its logs
                    print for the last "[saved]" entry.  These are DATA the
                    reference holds (outputs of its own runs), converted from
                    HDF5 because h5py is not available on the test box.
  ref_log_head.txt  first lines of one reference log (stdout grammar fixture).
  export_48.npz, vgg_generated_48.i.gz
                    one "vgg_generated_XX.i" header the reference ships, with the W
                    (from its result file) and pooling-region filters (recovered
                    from the header itself) it was written from: golden vector of
                    the export format (see make_export_fixture).
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


def make_export_fixture():
    """export_48.npz + vgg_generated_48.i.gz: a header the reference ships
    (workspace/opencv/vgg_generated_48.i, written by its export-opencv from the W of
    workspace/pj-learn/notredame-yosemite-0.025-0.075-pr#7-0.0020-0.200-pj.h5) together with its
    inputs: W read from that result file, and the selected pooling-region filters recovered from
    the header's own sparse PR arrays (the filters.h5 it was made from is not in the repository).
    Output file + inputs = a golden vector for the export format."""
    import gzip
    src = "/root/reference/workspace/opencv/vgg_generated_48.i"
    text = open(src, "rb").read()
    body = text.decode()
    def ints(name):
        m = re.search(r"static const unsigned int %s\[\] =\n\{(.*?)\};" % name, body, re.S)
        return [int(t, 16) for t in re.findall(r"0x[0-9a-fA-F]+", m.group(1))]
    rows = int(re.search(r"PRrows = (\d+);", body).group(1))
    cols = int(re.search(r"PRcols = (\d+);", body).group(1))
    idx, vals = ints("PRidx"), np.array(ints("PR"), np.uint32).view(np.float32)
    PR = np.zeros(rows * cols, np.float32)
    k = 0
    for start, count in zip(idx[0::2], idx[1::2]):
        PR[start:start + count] = vals[k:k + count]
        k += count
    assert k == vals.size
    read = h5_reader()
    W = read(os.path.join(REF, "notredame-yosemite-0.025-0.075-pr#7-0.0020-0.200-pj.h5"), "W")
    np.savez_compressed(os.path.join(HERE, "export_48.npz"), PR=PR.reshape(rows, cols), W=W,
                        prg=np.array("pr-learn/olderbest/yosemite-0.025-0.075-pr.h5"), widx=np.array(7),
                        prj=np.array("pj-learn/notredame-yosemite-0.025-0.075-pr#7-0.0020-0.200-pj.h5"))
    with gzip.GzipFile(os.path.join(HERE, "vgg_generated_48.i.gz"), "wb", mtime=0) as f:
        f.write(text)
    print("export fixture: PR", PR.reshape(rows, cols).shape, "W", W.shape, "header bytes", len(text))


if __name__ == "__main__":
    if os.path.isdir(REF):
        make_ref_results()
        make_export_fixture()
    else:
        print("reference not mounted: keeping existing ref_results.npz")
    make_oracle_vectors()
