"""Developer aid: free run, per-step A+ error against the oracle's ssyevr on the GPU's own dual average, tracker passes,
and (DLCO_RANK_UPDATE_CHECK=1) the deviation of the rank-update first term from the product it replaces."""
import importlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa
dlco = importlib.import_module("opencv-dlco_amd")
from oracle import ref
from util import relmax, synth
ref.lib(); ref.set_threads(1)
F = int(sys.argv[1]) if len(sys.argv) > 1 else 256
nstep = int(sys.argv[2]) if len(sys.argv) > 2 else 30
mu = float(sys.argv[3]) if len(sys.argv) > 3 else 0.004
N, B, gamma = 4000, 200, 0.5
D, L = synth(N, F, k=20, seed=9)
ctx = dlco.Context(F, N, B=B, mu=mu, gamma=gamma, grad_bf16=int(os.environ.get("RU_DEBUG_BF16", "0")))
ctx.set_data(D, L)
prev = ctx.eig_stats()
for s in range(nstep):
    Wp = ctx.W()
    ctx.step()
    b = ctx.batch()                        # the step's distances against |W_prev x|^2 in float64
    pd = ((D[b["pos_rows"]].astype(np.float64) @ Wp.T.astype(np.float64)) ** 2).sum(1)
    nd = ((D[b["neg_rows"]].astype(np.float64) @ Wp.T.astype(np.float64)) ** 2).sum(1)
    derr = max(np.abs(b["pd"] - pd).max(), np.abs(b["nd"] - nd).max()) / max(pd.max(), nd.max(), 1e-30)
    Ap, W, _ = ref.psd_project(ref.dual_to_primal(ctx.dfavg(), mu, gamma, s))
    e = relmax(ctx.A(), Ap)
    st, cn = ctx.eig_stats(), ctx.counters()
    print("step %2d rank %3d/%3d err %.2e passes %d rows %d block %d ru %d chk %.2e locked %d/%d dist %.1e" % (s, ctx.W().shape[0], W.shape[0], e, st["iters"] - prev["iters"],
          st["product_rows"] - prev["product_rows"], st["block_rows"], cn["rank_update_passes"], cn["rank_update_check"], cn["locked_passes"], cn["locked_rows"], derr))
    prev = st
