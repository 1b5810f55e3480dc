"""Developer check: how the tracker's convergence criterion relates to the error of A+ at full width.  Runs a bench workload to
its regime, then steps one at a time; after every step the GPU's A+ is compared with the oracle's ssyevr on the GPU's own dual
average, and the criterion the step stopped at is read from the tracker's debug trace (DLCO_EIG_DEBUG=1 must be set; stderr
goes to the file named by argv[3]).
    DLCO_EIG_DEBUG=1 python3 tools/full_width_crit_vs_error.py c3 520 8 2> trace.txt"""
import importlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from oracle import ref  # noqa: E402

dlco = importlib.import_module("opencv-dlco_amd")
name = sys.argv[1] if len(sys.argv) > 1 else "c3"
start = int(sys.argv[2]) if len(sys.argv) > 2 else 520
nchk = int(sys.argv[3]) if len(sys.argv) > 3 else 6
wl = bench.WORKLOADS[name]
ctx = bench.build_context(dlco, wl)
ref.lib()
ctx.steps(start)
for s in range(nchk):
    it0 = ctx.eig_stats()["iters"]
    ctx.step()
    t = ctx.t()
    W = ctx.W().astype(np.float64)
    A = ref.dual_to_primal(ctx.dfavg(), wl["mu"], wl["gamma"], t - 1)
    ref.set_threads(len(os.sched_getaffinity(0)))
    Wr, _ = ref.psd_factor(A)
    ref.set_threads(1)
    del A
    Wr = Wr.astype(np.float64)
    Ag, Ar = W.T @ W, Wr.T @ Wr
    print("step %d: rank %d / %d, passes %d, err_A %.3e" % (t, W.shape[0], Wr.shape[0], ctx.eig_stats()["iters"] - it0,
                                                          np.abs(Ag - Ar).max() / np.abs(Ar).max()), flush=True)
    sys.stderr.write("[check] step %d done\n" % t)
    sys.stderr.flush()
ctx.close()
