"""Times descriptor generation at the size pj-learn's configs[1] implies: 1024 pooling
regions x 8 bins = 8192 floats per patch.  Prints kernel time per patch (HIP events around the transform +
pooling launches, inputs resident) and the f64-MFMA rate of the pooling product."""
import importlib, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
dlco = importlib.import_module("opencv-dlco_amd")
import torch

n = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
nsel = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
rng = np.random.default_rng(0)
patches = rng.integers(0, 256, (n, 64, 64)).astype(np.uint8)
F = (rng.random((nsel, 4096)) < 0.05).astype(np.float32) * rng.random((nsel, 4096)).astype(np.float32) * 0.01
ctx = dlco.DescContext()
ctx.set_filters(F)
table = torch.empty((n, nsel * 8), dtype=torch.float32, device="cuda:0")
ctx.compute_device(patches[:2048], table.data_ptr())
t0 = time.time()
ctx.compute_device(patches, table.data_ptr())
wall = time.time() - t0
ms = ctx.last_kernel_ms()
flops = 2.0 * nsel * 4096 * 8 * n
print(f"descriptors: {n} patches x {nsel*8}: kernels {ms:.2f} ms = {1e3*ms/n:.3f} us/patch, "
      f"{flops/ms/1e9:.1f} TFLOP/s f64 in the pooling product; wall incl. upload {wall*1e3:.1f} ms")
