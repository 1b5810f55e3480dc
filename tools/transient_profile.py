"""Developer check: the start-up transient of pj-learn on the bench workload - the first iterations from W = 0, where
the rank overshoots to several hundred and the tracker works on blocks of up to 1024 rows.  Run under
`rocprofv3 --kernel-trace --stats` to see which kernels the time goes to (profiles/r3_transient_*).

    python3 tools/transient_profile.py [steps] [config]"""
import importlib
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

dlco = importlib.import_module("opencv-dlco_amd")
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 100
wl = bench.WORKLOADS[sys.argv[2] if len(sys.argv) > 2 else "c2"]
ctx = bench.build_context(dlco, wl)
ctx.sync()
marks = []
t0 = time.perf_counter()
done = 0
for n in (1, 4, 5, 10, 10, 20, 50, 100, 100):
    if done >= steps:
        break
    n = min(n, steps - done)
    ctx.steps(n)
    ctx.sync()
    done += n
    es = ctx.eig_stats()
    marks.append((done, time.perf_counter() - t0, ctx.W().shape[0], es["block_rows"], es["iters"], es["jacobi_sweeps"]))
for m in marks:
    print("t=%4d  %.3f s  rank %4d  block %4d  tracker passes so far %d  jacobi sweeps %d" % m)
print("nonconverged", ctx.counters()["nonconverged"])
ctx.close()
