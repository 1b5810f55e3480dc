import sys, importlib
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
import numpy as np
from oracle import ref
dlco = importlib.import_module('opencv-dlco_amd')
from test_gpu_parity import _psd_case
F, rh = 256, 40
mu, gamma, t = 0.004, 0.5, 17
G, Ap, Wref = _psd_case(ref, F, F+rh, t, mu, gamma, rh)
ctx = dlco.Context(F, 16, B=4, mu=mu, gamma=gamma)
try:
    W, A = ctx.psd_project(G, t)
    print("ok", W.shape, np.abs(A-Ap).max()/np.abs(Ap).max())
except Exception as e:
    print("ERR", e)
