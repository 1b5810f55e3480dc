"""Developer check: a long free run (default 8000 steps of c2), no non-converged step allowed, then the GPU's A+ against the
oracle's ssyevr on the GPU's own dual average at the end (drift of the carried block, the rank-update chain and Y = Q H over
thousands of steps would show here).
    python3 tools/long_soak_check.py [config] [steps]"""
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
name = sys.argv[1] if len(sys.argv) > 1 else "c2"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 8000
wl = bench.WORKLOADS[name]
ctx = bench.build_context(dlco, wl)
ref.lib()
t0 = time.perf_counter()
ctx.steps(steps)
ctx.sync()
dt = time.perf_counter() - t0
cn, es = ctx.counters(), ctx.eig_stats()
W = ctx.W().astype(np.float64)
A = ref.dual_to_primal(ctx.dfavg(), wl["mu"], wl["gamma"], steps - 1)
ref.set_threads(len(os.sched_getaffinity(0)))
Wr, _ = ref.psd_factor(A)
del A
Wr = Wr.astype(np.float64)
Ag, Ar = W.T @ W, Wr.T @ Wr
print("%s: %d steps in %.2f s (%.0f pair-rows/s incl. the start-up transient), non-converged %d, tracker passes %d, rank-update passes %d, "
      "locked passes %d; after the last step rank %d (oracle ssyevr %d), err_A = %.3e (gate 1e-4)"
      % (name, steps, dt, 2.0 * wl["batch"] * steps / dt, cn["nonconverged"], es["iters"], cn["rank_update_passes"], cn["locked_passes"],
         W.shape[0], Wr.shape[0], np.abs(Ag - Ar).max() / np.abs(Ar).max()))
ctx.close()
