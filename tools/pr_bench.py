"""Times the pr-learn stage at the reference's shape: F = 5120 pooling regions, windows of
100 000 sequential iterations per launch (the reference's LogStep).  Prints iterations/s of the GPU path
(the CPU restatement is timed by tests/test_pr_learn.py::test_gpu_trajectory_is_the_oracles_bit_for_bit)."""
import importlib, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
dlco = importlib.import_module("opencv-dlco_amd")

F, N = 5120, int(sys.argv[1]) if len(sys.argv) > 1 else 100000
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 300000
rng = np.random.default_rng(0)
L = (np.arange(N) % 2 == 0).astype(np.uint8)
D = rng.random((N, F), dtype=np.float32)
D[L == 1, :512] *= 0.5
ctx = dlco.PrContext(F, N, mu=0.025, gamma=0.10)
ctx.set_data(D, L)
ctx.steps(1000)
t0 = time.time()
ctx.steps(iters)
st = ctx.state()
dt = time.time() - t0
print(f"pr-learn GPU: {iters} iterations in {dt:.3f} s = {iters/dt/1e3:.1f} k it/s ({1e6*dt/iters:.2f} us per iteration, "
      f"2 rows x {F*4} B each), nnz(w) = {int((st['w'] != 0).sum())}")
