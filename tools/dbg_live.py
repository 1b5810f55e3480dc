import sys, importlib
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
import numpy as np
from oracle import ref
from util import synth, relmax
dlco = importlib.import_module('opencv-dlco_amd')
N, F, B = 4000, 128, 200
D, L = synth(N, F, k=20, seed=9)
mu, gamma = 0.004, 0.5
tr = ref.Trainer(D, L, B=B, mu=mu, gamma=gamma, grad_order=1)
ctx = dlco.Context(F, N, B=B, mu=mu, gamma=gamma)
ctx.set_data(D, L)
for s in range(30):
    before = tr.state(); tr.step(); after = tr.state()
    ctx.set_state(s, before["dfavg"], before["W"] if s else None)
    print("---- step", s, file=sys.stderr, flush=True)
    ctx.step()
    es = ctx.eig_stats()
    print("step", s, "rank", ctx.W().shape[0], after["r"], "A err %.2e" % relmax(ctx.A(), after["A"]), es, file=sys.stderr, flush=True)
