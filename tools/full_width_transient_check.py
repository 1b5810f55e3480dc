"""Developer check (GPU box, ~5 min of host time): the start-up transient at FULL width against the oracle's ssyevr.
The suite's full-width tests compare the steady state (steps 320 / 520); the transient - blocks of several hundred rows,
locked passes, the multi-workgroup Jacobi - is compared step by step only at F = 256 / 544.  Here the c2 workload
(500 000 x 8192) is stopped after a few early steps and the GPU's A+ = W^T W is compared with the oracle's projection of
the GPU's own dual average (src/pj-learn.cpp:426-490), one ssyevr at n = 8192 per stop.

    python3 tools/full_width_transient_check.py [stop steps ...]      default: 3 20 60"""
import importlib
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from oracle import ref  # noqa: E402

dlco = importlib.import_module("opencv-dlco_amd")
stops = [int(a) for a in sys.argv[1:]] or [3, 20, 60]
wl = bench.WORKLOADS["c2"]
ctx = bench.build_context(dlco, wl)
ref.lib()
done = 0
for stop in stops:
    ctx.steps(stop - done)
    done = stop
    cn, es = ctx.counters(), ctx.eig_stats()
    W = ctx.W().astype(np.float64)
    dfavg = ctx.dfavg()
    t0 = time.perf_counter()
    A = ref.dual_to_primal(dfavg, wl["mu"], wl["gamma"], stop - 1)
    ref.set_threads(len(os.sched_getaffinity(0)))
    Wr, ev = ref.psd_factor(A)
    ref.set_threads(1)
    del A
    Wr = Wr.astype(np.float64)
    Ag, Ar = W.T @ W, Wr.T @ Wr
    err = np.abs(Ag - Ar).max() / np.abs(Ar).max()
    print("after step %3d: rank %d (oracle ssyevr %d), block %d rows, err_A = %.3e (gate 1e-4), tracker passes so far %d, locked passes %d, "
          "non-converged %d   [oracle %.0f s]" % (stop, W.shape[0], Wr.shape[0], es["block_rows"], err, es["iters"], cn["locked_passes"],
                                                  cn["nonconverged"], time.perf_counter() - t0), flush=True)
    del Ag, Ar
ctx.close()
