"""Developer check: cost of the reference's LogStep block (validation + ComputePJStats,
src/pj-learn.cpp:492-587) at the bench shape, through dlco_log_step."""
import importlib
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

dlco = importlib.import_module("opencv-dlco_amd")
F, N, B = 8192, 500000, 200
ctx = dlco.Context(F, N, B=B, mu=0.002, gamma=0.5)
ctx.synth_data(bench.make_U(F, 96, 2216), 2216, 0.35, 1.0, 0.05)
ctx.steps(320)
ctx.sync()
for i in range(3):
    t0 = time.perf_counter()
    e = ctx.log_step()
    ctx.sync()
    dt = time.perf_counter() - t0
    print("log_step %d: %.1f ms  (vtime field %.4f s)  rank %d dim %d fpr95 %.4f auc %.6f loss %.5f" %
          (i, dt * 1e3, e.vtime, e.rank, e.dim, e.fpr95, e.auc, e.loss_val))
    ctx.steps(100)
