import sys, importlib
sys.path.insert(0, '/root/repo')
import numpy as np
dlco = importlib.import_module('opencv-dlco_amd')
F, rows = 8192, 96
rng = np.random.default_rng(0)
G = rng.standard_normal((F, F)).astype(np.float32); G = ((G + G.T) * 0.5).astype(np.float32)
X = rng.standard_normal((rows, F)).astype(np.float32)
ctx = dlco.Context(F, 16, B=4)
for mode in (0, 1, 0, 1, 1):
    out = ctx.sym_product(X, G, mode=mode)
print("done", float(np.abs(out).max()))
