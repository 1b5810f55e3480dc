"""The data-parallel protocol of the HIP path on real hardware.  A GPU box of this pool has one
GPU, so two ranks (world = 2) are emulated as two libdlco contexts on the same device and the two
collectives are performed by hand on the bound torch tensors — exactly the byte movement RCCL's
all-gather / all-reduce would do.  The result must equal a single context with the same global
batch (the N > 1 collectives themselves are covered by tests/test_distributed_cpu.py over gloo)."""
import importlib

import numpy as np
import pytest

from util import relmax, synth

pytestmark = pytest.mark.gpu


def test_two_rank_protocol_equals_single_rank(dlco):
    import torch
    ddist = importlib.import_module("opencv-dlco_amd.dist")
    N, F, B = 3000, 256, 40
    D, L = synth(N, F, k=16, seed=51)
    mu, gamma = 0.004, 0.5
    dev = torch.device("cuda", 0)
    single = dlco.Context(F, N, B=B, mu=mu, gamma=gamma)
    single.set_data(D, L)
    ranks = []
    for r in range(2):
        c = dlco.Context(F, N, B=B, mu=mu, gamma=gamma, rank=r, world=2)
        c.set_data(D, L)
        ranks.append(ddist.HipEngine(dlco, c, dev))
    per = 2 * B // 2
    for step in range(6):
        single.step()
        for e in ranks:
            e.begin()
        torch.cuda.synchronize()
        # all-gather: every rank receives the other's slice
        ranks[0].dist[per:2 * per].copy_(ranks[1].dist[per:2 * per])
        ranks[1].dist[0:per].copy_(ranks[0].dist[0:per])
        torch.cuda.synchronize()
        for e in ranks:
            e.grad_phase()
        torch.cuda.synchronize()
        total = ranks[0].grad + ranks[1].grad            # all-reduce (sum)
        ranks[0].grad.copy_(total)
        ranks[1].grad.copy_(total)
        torch.cuda.synchronize()
        for e in ranks:
            e.finish()
        b0, b1, bs = ranks[0].ctx.batch(), ranks[1].ctx.batch(), single.batch()
        assert np.array_equal(b0["pos_rows"], bs["pos_rows"]) and np.array_equal(b1["neg_rows"], bs["neg_rows"])
        assert np.array_equal(b0["rho"], b1["rho"]) and np.array_equal(b0["kappa"], b1["kappa"])
    d0, d1, ds = ranks[0].ctx.dfavg(), ranks[1].ctx.dfavg(), single.dfavg()
    assert np.array_equal(d0, d1)                          # replicas stay bit-identical
    assert relmax(d0, ds) <= 5e-6                          # vs one rank: only the fp32 sum grouping differs
    assert relmax(ranks[0].ctx.A(), single.A()) <= 5e-4
    for e in ranks:
        e.ctx.close()
    single.close()


def test_single_rank_trainer_with_bound_buffers(dlco):
    import torch
    ddist = importlib.import_module("opencv-dlco_amd.dist")
    N, F, B = 2000, 128, 20
    D, L = synth(N, F, k=8, seed=52)
    a = dlco.Context(F, N, B=B, mu=0.004)
    b = dlco.Context(F, N, B=B, mu=0.004)
    a.set_data(D, L)
    b.set_data(D, L)
    tr = ddist.DataParallelTrainer(ddist.HipEngine(dlco, b, torch.device("cuda", 0)))
    assert tr.world == 1
    a.steps(5)
    tr.steps(5)
    assert np.array_equal(a.dfavg(), b.dfavg())
    assert np.array_equal(a.W(), b.W())
    a.close()
    b.close()


def _run_sharded_ranks(dlco, D, L, F, N, B, mu, gamma, world, steps, pairs=None):
    """`world` sharded contexts on the one GPU of the box, one host thread each; the all-gather
    callback moves the peers' chunks by hand (what RCCL's all-gather does) between two barriers."""
    import threading

    import torch
    dev = torch.device("cuda", 0)
    ctxs, bufs = [], []
    for r in range(world):
        c = dlco.Context(F, N, B=B, mu=mu, gamma=gamma, rank=r, world=world, shard=1)
        if pairs is None:
            c.set_data(D, L)
        else:
            c.set_pairs(*pairs)                              # descriptors + Indices table (pair mode)
        _, nbytes = c.dev_buffer(dlco.BUF_GATHER)
        gather = torch.zeros(nbytes // 4, dtype=torch.float32, device=dev)
        dist_t = torch.zeros(2 * B, dtype=torch.float32, device=dev)
        torch.cuda.synchronize()
        c.bind_buffer(dlco.BUF_GATHER, gather.data_ptr(), nbytes)
        c.bind_buffer(dlco.BUF_DIST, dist_t.data_ptr(), dist_t.numel() * 4)
        ctxs.append(c)
        bufs.append({dlco.BUF_GATHER: gather, dlco.BUF_DIST: dist_t})
    barrier = threading.Barrier(world, timeout=120)
    calls = [0] * world

    def make_cb(r):
        def cb(which, nbytes):
            n = nbytes // 4
            ctxs[r].sync()                                   # my chunk has been written
            barrier.wait()
            for g in range(world):
                if g != r:
                    bufs[r][which][g * n:(g + 1) * n].copy_(bufs[g][which][g * n:(g + 1) * n])
            torch.cuda.synchronize()
            barrier.wait()                                   # nobody overwrites a chunk a peer still reads
            calls[r] += 1
            return 0
        return cb

    for r in range(world):
        ctxs[r].set_allgather(make_cb(r))
    errors = []

    def worker(r):
        try:
            for _ in range(steps):
                ctxs[r].step()
        except Exception as e:                               # noqa: BLE001
            errors.append((r, e))
            barrier.abort()

    threads = [threading.Thread(target=worker, args=(r,)) for r in range(world)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=300)
    assert not errors, errors
    assert all(not t.is_alive() for t in threads)
    return ctxs, calls


@pytest.mark.parametrize("F,N,B", [(256, 3000, 40), (1024, 2000, 24)])   # generic products / split-bf16 slab products
def test_sharded_dual_average_equals_single_rank(dlco, F, N, B):
    """cfg.shard = 1: rank g owns columns [g*F/2, (g+1)*F/2) of the dual average; the gradient slab
    is computed over the whole global batch and the tracker's products all-gather their slabs.
    Assembled, the result must equal one context with the same global batch."""
    D, L = synth(N, F, k=16, seed=53)
    mu, gamma, world, steps = 0.004, 0.5, 2, 6
    single = dlco.Context(F, N, B=B, mu=mu, gamma=gamma)
    single.set_data(D, L)
    single.steps(steps)
    ctxs, calls = _run_sharded_ranks(dlco, D, L, F, N, B, mu, gamma, world, steps)
    assert calls[0] == calls[1] and calls[0] >= 2 * steps       # distances + at least one product per step
    bs = single.batch()
    for c in ctxs:
        b = c.batch()
        for k in ("pos_rows", "neg_rows", "rho", "kappa"):
            assert np.array_equal(b[k], bs[k]), k
    W0, W1 = ctxs[0].W(), ctxs[1].W()
    assert np.array_equal(W0, W1)                              # the replicated tracker stays bit-identical
    cw = F // world
    df = np.concatenate([ctxs[g].dfavg()[:, g * cw:(g + 1) * cw] for g in range(world)], axis=1)
    ds = single.dfavg()
    assert relmax(df, ds) <= 5e-6
    assert relmax(ctxs[0].A(), single.A()) <= 5e-4
    assert ctxs[0].validate()[2] == single.validate()[2]        # same rank
    for c in ctxs:
        c.close()
    single.close()


def test_sharded_trainer_single_rank_path(dlco, monkeypatch):
    """ShardedTrainer end to end with world = 1 (DLCO_FORCE_SHARD exercises the slab kernels, the
    pack / callback / unpack path and the torch ExternalStream plumbing on one GPU)."""
    import torch
    ddist = importlib.import_module("opencv-dlco_amd.dist")
    monkeypatch.setenv("DLCO_FORCE_SHARD", "1")
    N, F, B = 2000, 512, 20
    D, L = synth(N, F, k=8, seed=54)
    a = dlco.Context(F, N, B=B, mu=0.004)
    b = dlco.Context(F, N, B=B, mu=0.004, shard=1)
    a.set_data(D, L)
    b.set_data(D, L)
    tr = ddist.ShardedTrainer(ddist.HipShardEngine(dlco, b, torch.device("cuda", 0)))
    a.steps(5)
    tr.steps(5)
    assert relmax(b.dfavg(), a.dfavg()) <= 5e-6
    assert relmax(b.A(), a.A()) <= 5e-4
    tr.close()                                            # hand torch its own stream back before the context dies
    torch.zeros(8, device="cuda").sum().item()            # torch is usable afterwards
    a.close()
    b.close()


def test_sharded_pair_mode_equals_single_rank_row_mode(dlco):
    """Both extensions at once: two sharded ranks fed with per-patch descriptors + the Indices table
    against one rank fed with the materialised differences."""
    from test_gpu_parity import _pair_case
    N, F, B, P = 3000, 256, 40, 900
    desc, pairs, D, L = _pair_case(N, F, P, seed=7)
    mu, gamma, steps = 0.004, 0.5, 6
    single = dlco.Context(F, N, B=B, mu=mu, gamma=gamma)
    single.set_data(D, L)
    single.steps(steps)
    ctxs, _ = _run_sharded_ranks(dlco, None, None, F, N, B, mu, gamma, 2, steps, pairs=(desc, pairs))
    bs = single.batch()
    for c in ctxs:
        b = c.batch()
        for k in ("pos_rows", "neg_rows", "rho", "kappa"):
            assert np.array_equal(b[k], bs[k]), k
    assert np.array_equal(ctxs[0].W(), ctxs[1].W())
    cw = F // 2
    df = np.concatenate([ctxs[g].dfavg()[:, g * cw:(g + 1) * cw] for g in range(2)], axis=1)
    assert relmax(df, single.dfavg()) <= 5e-6
    assert relmax(ctxs[0].A(), single.A()) <= 5e-4
    for c in ctxs:
        c.close()
    single.close()


def test_comm_init_argument_checks(dlco):
    """dlco_comm_init is only meaningful on a context with other ranks to talk to and needs a full 128-byte id."""
    N, F, B = 500, 128, 8
    D, L = synth(N, F, k=4, seed=60)
    ctx = dlco.Context(F, N, B=B)
    ctx.set_data(D, L)
    with pytest.raises(dlco.DlcoError) as e:
        ctx.comm_init(b"\0" * 128)
    assert e.value.code == dlco.ERR_INVALID and "single-rank" in str(e.value)
    assert ctx.L.dlco_comm_init(ctx.h, None, 128, None) == dlco.ERR_INVALID
    ctx.step()                                            # the context is still usable
    ctx.close()
