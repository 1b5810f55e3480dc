import sys, importlib
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
import numpy as np
from oracle import ref
from util import golden, relmax
dlco = importlib.import_module('opencv-dlco_amd')
z = golden(sys.argv[1] if len(sys.argv) > 1 else "oracle_step_F32_B8.npz")
N, F, B, mu, gamma, nstep = z["cfg"]; N, F, B, nstep = int(N), int(F), int(B), int(nstep)
ctx = dlco.Context(F, N, B=B, mu=float(mu), gamma=float(gamma))
ctx.set_data(z["D"], z["L"])
for s in range(nstep):
    W_in = z["s%d_W_in" % s]
    if s == 0: W_in = W_in[:0]
    ctx.set_state(s, z["s%d_dfavg_in" % s], W_in)
    ctx.step()
    A = ctx.A(); W = ctx.W()
    print("step", s, "rank gpu", W.shape[0], "ref", z["s%d_W" % s].shape[0], "dfavg err", relmax(ctx.dfavg(), z["s%d_dfavg" % s]), "A err", relmax(A, z["s%d_A" % s]), flush=True)
