"""World-size-2 test of the data-parallel protocol over gloo (CPU).  The product's
DataParallelTrainer (opencv-dlco_amd/dist.py) is driven by an oracle-backed engine that
implements the same three phases as libdlco.so's dlco_step_begin / _grad / _finish, so the
sharding, the all-gather layout and the all-reduce are exercised without a GPU; the result
must equal a single-rank run on the same global batch."""
import importlib
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class OracleEngine:
    """Same contract as HipEngine: `dist` [world][2*B/world] and `grad` [F*F] exchange tensors."""

    def __init__(self, D, L, B, mu, gamma, rank, world):
        from oracle import ref
        self.ref, self.D, self.L = ref, D, L
        self.N, self.F = D.shape
        self.B, self.Bl, self.rank, self.world = B, B // world, rank, world
        self.mu, self.gamma = mu, gamma
        self.pos, self.neg = ref.build_index(L)
        self.npt, self.nnt = ref.split(self.pos.size), ref.split(self.neg.size)
        self.rng = ref.Rng(2215)
        self.t = 0
        self.W = np.zeros((0, self.F), np.float32)
        self.dfavg = np.zeros((self.F, self.F), np.float32)
        self.dist = torch.zeros(2 * B, dtype=torch.float32)
        self.grad = torch.zeros(self.F * self.F, dtype=torch.float32)

    def begin(self):
        ip, ineg = self.rng.sample(self.npt, self.nnt, self.B)            # identical on every rank
        self.pos_rows, self.neg_rows = self.pos[ip], self.neg[ineg]
        lo, hi = self.rank * self.Bl, (self.rank + 1) * self.Bl
        pd = self.ref.project_sqdist_ids(self.W, self.D, self.pos_rows[lo:hi]) if len(self.W) else np.zeros(self.Bl, np.float32)
        nd = self.ref.project_sqdist_ids(self.W, self.D, self.neg_rows[lo:hi]) if len(self.W) else np.zeros(self.Bl, np.float32)
        base = self.rank * 2 * self.Bl
        self.dist[base:base + self.Bl] = torch.from_numpy(pd)
        self.dist[base + self.Bl:base + 2 * self.Bl] = torch.from_numpy(nd)

    def grad_phase(self):
        d = self.dist.numpy().reshape(self.world, 2, self.Bl)
        self.pd, self.nd = d[:, 0, :].reshape(-1).copy(), d[:, 1, :].reshape(-1).copy()
        rho, kap = self.ref.viol_counts(self.pd, self.nd)
        lo, hi = self.rank * self.Bl, (self.rank + 1) * self.Bl
        g = self.ref.grad_reform(self.D[self.pos_rows[lo:hi]], self.D[self.neg_rows[lo:hi]], rho[lo:hi], kap[lo:hi])
        self.grad.copy_(torch.from_numpy(g.reshape(-1)))

    def finish(self):
        g = self.grad.numpy().reshape(self.F, self.F)
        self.dfavg = self.ref.rda_update(self.dfavg, g, self.t, self.B)
        A = self.ref.dual_to_primal(self.dfavg, self.mu, self.gamma, self.t)
        _, self.W, _ = self.ref.psd_project(A)
        self.t += 1


def _worker(rank, world, port, steps, out_path):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from util import synth
    ddist = importlib.import_module("opencv-dlco_amd.dist")
    D, L = synth(1200, 32, k=6, seed=17)
    eng = OracleEngine(D, L, B=20, mu=0.01, gamma=0.5, rank=rank, world=world)
    tr = ddist.DataParallelTrainer(eng)
    rows = []
    for _ in range(steps):
        tr.step()
        rows.append(np.concatenate([eng.pos_rows, eng.neg_rows]))
    if rank == 0:
        np.savez(out_path, dfavg=eng.dfavg, W=eng.W, rows=np.stack(rows), pd=eng.pd, nd=eng.nd)
    dist.barrier()
    dist.destroy_process_group()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_two_ranks_equal_one_rank_on_the_same_global_batch(ref, tmp_path):
    steps = 5
    out = str(tmp_path / "dp2.npz")
    mp.spawn(_worker, args=(2, _free_port(), steps, out), nprocs=2, join=True)
    got = np.load(out)

    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from util import relmax, synth
    D, L = synth(1200, 32, k=6, seed=17)
    single = OracleEngine(D, L, B=20, mu=0.01, gamma=0.5, rank=0, world=1)
    ddist = importlib.import_module("opencv-dlco_amd.dist")
    tr1 = ddist.DataParallelTrainer(single)
    for s in range(steps):
        tr1.step()
        assert np.array_equal(got["rows"][s], np.concatenate([single.pos_rows, single.neg_rows]))   # bit-exact pair indexing
    # W depends on dfAvg, whose fp32 sums are grouped differently across ranks: tolerance, not bits
    scale = max(single.pd.max(), single.nd.max())
    assert np.abs(got["pd"] - single.pd).max() <= 1e-4 * scale and np.abs(got["nd"] - single.nd).max() <= 1e-4 * scale
    assert relmax(got["dfavg"], single.dfavg) <= 2e-6          # only the grouping of the fp32 sums differs
    assert got["W"].shape == single.W.shape

    # and both equal the plain reference loop with szBatch = global B
    tr = ref.Trainer(D, L, B=20, mu=0.01, gamma=0.5, grad_order=1)
    for _ in range(steps):
        tr.step()
    assert relmax(single.dfavg, tr.state()["dfavg"]) <= 2e-6
    tr.close()


def test_trainer_without_process_group_is_single_rank(ref):
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from util import synth
    D, L = synth(400, 16, k=4, seed=3)
    eng = OracleEngine(D, L, B=8, mu=0.01, gamma=0.5, rank=0, world=1)
    ddist = importlib.import_module("opencv-dlco_amd.dist")
    t = ddist.DataParallelTrainer(eng)
    assert t.world == 1 and t.rank == 0
    t.steps(3)
    assert eng.t == 3 and eng.dfavg.any()


# ---- column-sharded dual average (cfg.shard = 1) --------------------------------------------------
class OracleShardEngine:
    """Same contract as HipShardEngine (dist.py): the step runs inside the engine and calls back for
    in-place all-gathers of `dist` [world][2*B/world] and `gather` [world][chunk].  Rank g keeps only
    the columns [g*F/world, (g+1)*F/world) of the dual average and computes that slab of the
    gradient over the whole global batch; the oracle's dense PSD projection stands in for the
    tracker's products, so its exchange is the all-gather of the F x F/world slabs themselves."""
    BUF_DIST, BUF_GATHER = 1, 5

    def __init__(self, D, L, B, mu, gamma, rank, world):
        import contextlib
        from oracle import ref
        self.ref, self.D, self.L = ref, D, L
        self.N, self.F = D.shape
        self.B, self.Bl, self.rank, self.world = B, B // world, rank, world
        self.cw = self.F // world
        self.c0 = rank * self.cw
        self.mu, self.gamma = mu, gamma
        self.pos, self.neg = ref.build_index(L)
        self.npt, self.nnt = ref.split(self.pos.size), ref.split(self.neg.size)
        self.rng = ref.Rng(2215)
        self.t = 0
        self.W = np.zeros((0, self.F), np.float32)
        self.slab = np.zeros((self.F, self.cw), np.float32)           # this rank's columns of dfAvg
        self.own = np.zeros((self.F, self.F), np.float32)             # F x F scratch, valid inside the slab only
        self.dist = torch.zeros(2 * B, dtype=torch.float32)
        self.gather = torch.zeros(self.F * self.F, dtype=torch.float32)
        self.cb = None
        self.stream_guard = contextlib.nullcontext

    def set_allgather(self, fn):
        self.cb = fn

    def step(self):
        ip, ineg = self.rng.sample(self.npt, self.nnt, self.B)            # identical on every rank
        self.pos_rows, self.neg_rows = self.pos[ip], self.neg[ineg]
        lo, hi = self.rank * self.Bl, (self.rank + 1) * self.Bl
        pd = self.ref.project_sqdist_ids(self.W, self.D, self.pos_rows[lo:hi]) if len(self.W) else np.zeros(self.Bl, np.float32)
        nd = self.ref.project_sqdist_ids(self.W, self.D, self.neg_rows[lo:hi]) if len(self.W) else np.zeros(self.Bl, np.float32)
        base = self.rank * 2 * self.Bl
        self.dist[base:base + self.Bl] = torch.from_numpy(pd)
        self.dist[base + self.Bl:base + 2 * self.Bl] = torch.from_numpy(nd)
        assert self.cb(self.BUF_DIST, 2 * self.Bl * 4) == 0
        d = self.dist.numpy().reshape(self.world, 2, self.Bl)
        self.pd, self.nd = d[:, 0, :].reshape(-1).copy(), d[:, 1, :].reshape(-1).copy()
        rho, kap = self.ref.viol_counts(self.pd, self.nd)
        g = self.ref.grad_reform(self.D[self.pos_rows], self.D[self.neg_rows], rho, kap)   # whole global batch
        # the oracle's dual-average update is elementwise on F x F arrays: run it on a copy whose
        # columns outside the slab are zero and keep the slab
        gm = np.zeros_like(g)
        gm[:, self.c0:self.c0 + self.cw] = g[:, self.c0:self.c0 + self.cw]
        self.own = self.ref.rda_update(self.own, gm, self.t, self.B)
        self.slab = np.ascontiguousarray(self.own[:, self.c0:self.c0 + self.cw])
        n = self.F * self.cw
        self.gather[self.rank * n:(self.rank + 1) * n] = torch.from_numpy(self.slab.reshape(-1))
        assert self.cb(self.BUF_GATHER, n * 4) == 0
        chunks = self.gather.numpy().reshape(self.world, self.F, self.cw)
        self.dfavg = np.ascontiguousarray(np.concatenate(list(chunks), axis=1))
        A = self.ref.dual_to_primal(self.dfavg, self.mu, self.gamma, self.t)
        _, self.W, _ = self.ref.psd_project(A)
        self.t += 1

    def steps(self, n):
        for _ in range(n):
            self.step()


def _shard_worker(rank, world, port, steps, out_path):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from util import synth
    ddist = importlib.import_module("opencv-dlco_amd.dist")
    D, L = synth(1200, 32, k=6, seed=17)
    eng = OracleShardEngine(D, L, B=20, mu=0.01, gamma=0.5, rank=rank, world=world)
    tr = ddist.ShardedTrainer(eng)
    rows = []
    for _ in range(steps):
        tr.step()
        rows.append(np.concatenate([eng.pos_rows, eng.neg_rows]))
    np.savez(out_path % rank, dfavg=eng.dfavg, slab=eng.slab, W=eng.W, rows=np.stack(rows), pd=eng.pd, nd=eng.nd)
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_two_ranks_equal_one_rank(ref, tmp_path):
    steps = 5
    out = str(tmp_path / "shard2_r%d.npz")
    mp.spawn(_shard_worker, args=(2, _free_port(), steps, out), nprocs=2, join=True)
    r0, r1 = np.load(out % 0), np.load(out % 1)
    assert np.array_equal(r0["dfavg"], r1["dfavg"]) and np.array_equal(r0["W"], r1["W"])   # replicas agree bit for bit
    F = r0["dfavg"].shape[0]
    assert np.array_equal(r0["slab"], r0["dfavg"][:, :F // 2]) and np.array_equal(r1["slab"], r1["dfavg"][:, F // 2:])

    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from util import relmax, synth
    D, L = synth(1200, 32, k=6, seed=17)
    tr = ref.Trainer(D, L, B=20, mu=0.01, gamma=0.5, grad_order=1)
    for s in range(steps):
        tr.step()
        pr, nr = tr.batch_ids()
        assert np.array_equal(r0["rows"][s], np.concatenate([pr, nr]))                      # bit-exact pair indexing
    assert relmax(r0["dfavg"], tr.state()["dfavg"]) <= 2e-6
    assert r0["W"].shape == tr.state()["W"].shape
    tr.close()
