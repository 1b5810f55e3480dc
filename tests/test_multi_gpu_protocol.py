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
