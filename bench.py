#!/usr/bin/env python3
"""bench.py — pj-learn patch-pairs/sec on the BASELINE workload (configs[1]):
Liberty-shaped 500k labelled pair-rows x PR-dim 8192, batch 200 positives + 200 negatives
per GPU, fp32, learned rank ~64.  Synthetic data of that shape is generated in HBM (the
Brown/Winder sets are not redistributable and there is no network).

A "step" is one full iteration of the reference's training loop (src/pj-learn.cpp:305-490):
sample the batch, project + squared distances, violation counts, fused weighted-SYRK
gradient + dual average, PSD projection.  Nothing is skipped inside the timed region.

    python bench.py --gpus 1 --steps 200 --warmup 20
    python bench.py --config c3                      (configs[2]: the rank ~128 regime)
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...   (N > 1)

The configuration BASELINE quotes is a trainer whose learned rank has settled near 64; from
the reference's start (W = 0, every pair violates, rank of a few hundred) that takes about 300
iterations.  Reaching that state is part of building the workload, like generating the data:
`--burn-in` full steps (default 300; 500 for --config c3, whose rank passes 128 only then; reported
in config) run before the W warm-up steps, so
that whatever --warmup / --steps the caller passes, the K timed steps are steps of the named
configuration and not of the start-up transient.

N > 1 (one process per GPU, RCCL): the headline `value` is the reference's own optimiser on N GPUs - the
same global batch of 200 + 200 rows, split over the ranks (`"scaling": "strong"`) - with the
column-sharded dual average; the same invocation then measures, outside the headline's timed region and
reported under `other_modes`, the literal contract of BASELINE configs[3] (replicated dual average + F x F
all-reduce per step, same global batch) and weak scaling (per-GPU batch fixed at 200 + 200, global batch
N*200 + N*200: a different optimiser, so its rate is not comparable with the single-GPU number).

Prints ONE JSON line on rank 0.
"""
import argparse
import importlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

PEAK_F32_MFMA_TFLOPS = 157.3      # /opt/skills/guides/MI355X_MICROARCH.md: fp32 matrix peak (dense)
PEAK_HBM_TBS = 8.0                # same guide: HBM3E spec peak (6.29 TB/s measured copy)

# The two single-GPU workloads BASELINE.json names.  The generator (dlco_synth_data) draws
# d = U^T z + noise*eps with a per-row log-normal scale, so that matches and non-matches overlap the
# way real patch pairs do: FPR@95 of a few per cent at rank ~64-128 (the reference's liberty run
# shows rank 71 / FPR95 9.19 % at step 500 and 67 / 5.67 % at the end:
# workspace/pj-learn/logging/liberty-liberty-0.035-0.250-pr#7-0.0010-0.100-pj.log:23-24,423-424).
# tests/test_full_width_gpu.py compares exactly these workloads with the oracle's ssyevr.
WORKLOADS = {
    # tools/sweep_synth.py picked these (gpurun_out/sweep2.log): rank 64 / FPR95 6.1 % and rank 127 / FPR95 5.6 % at step 520
    "c2": dict(name="configs[1] Liberty-shaped, rank ~64", F=8192, N=500000, batch=200, mu=0.0025, gamma=0.5,
               latent=96, sigma_pos=0.35, sigma_neg=1.0, noise=0.05, jitter=0.3, seed=2216,
               rank_band=(48, 88), fpr95_band=(0.02, 0.15), burn_in=300),
    "c3": dict(name="configs[2] NotreDame-shaped, rank ~128", F=8192, N=500000, batch=200, mu=0.001, gamma=0.5,
               latent=192, sigma_pos=0.35, sigma_neg=1.0, noise=0.05, jitter=0.3, seed=2216,
               # from W = 0 the rank overshoots (789 at step 50) and decays: 166 at step 320, 127 at step 520, 116 at
               # step 640 (DLCO_EIG_DEBUG trace, profiles/r2_c3_rank_trajectory.txt): the named regime, rank ~128, is reached
               # after ~500 steps, and before that every step still takes 2-3 tracker passes on a block of 192+ rows
               rank_band=(100, 160), fpr95_band=(0.02, 0.15), burn_in=500),
    # The shape the reference itself ran (every committed log: 500000 x 480 or x 544, batch 200 + 200; this one is
    # workspace/pj-learn/logging/liberty-liberty-0.035-0.250-pr#7-0.0010-0.100-pj.log: mu 0.001, gamma 0.1, rank 71 at
    # step 500 and 67 at the end, FPR95 9.19 % -> 5.67 %).  544 is not a multiple of the kernels' 128-column tile: the
    # library keeps the rows at 640 floats (include/dlco.h, dlco_device_width).  The CPU port runs FULL steps here
    # (ssyevr at n = 544 is ~0.1 s), and the reference's own logs give its Ttime.
    "ref544": dict(name="reference-real shape (liberty 544-wide run)", F=544, N=500000, batch=200, mu=0.001, gamma=0.1,
                   latent=96, sigma_pos=0.35, sigma_neg=1.0, noise=0.05, jitter=0.3, seed=2216,
                   rank_band=(40, 100), fpr95_band=(0.02, 0.15), burn_in=300, cpu_steps=50, cpu_rows=20000,
                   reference_logged={"Ttime_s_per_100_steps": {"mean": 7.46, "p05": 6.35, "p95": 8.62}, "ms_per_step": 74.6,
                                     "pair_rows_per_s": 5.4e3, "hardware": "unknown x86 host (OpenBLAS + OpenMP), GTX 970 for validation only",
                                     "source": "BASELINE.md section 1: Ttime of the 135 reference logs with `Load Distances: 500000 x 544`",
                                     "note": "context only: another machine, the real Liberty data"}),
}


def make_U(F, k, seed):
    """Latent directions with decaying strength so that the trace-norm threshold mu selects a rank."""
    rng = np.random.default_rng(seed)
    U = rng.standard_normal((k, F)).astype(np.float32)
    U /= np.linalg.norm(U, axis=1, keepdims=True)
    decay = (1.0 / (1.0 + np.arange(k) / 24.0)).astype(np.float32)
    return (U * decay[:, None]).astype(np.float32)


def build_context(dlco, wl, B=None, device=0, rank=0, world=1, shard=0, data_from=None, **kw):
    """A context on workload `wl` with its synthetic data resident in HBM.  `data_from` (another
    context on the same device and workload) shares that context's Distance matrix instead of
    generating a second copy."""
    ctx = dlco.Context(wl["F"], wl["N"], B=B or wl["batch"], mu=wl["mu"], gamma=wl["gamma"], device=device, rank=rank,
                       world=world, shard=shard, **kw)
    if data_from is not None:
        ctx.set_data_shared(data_from)
    else:
        ctx.synth_data(make_U(wl["F"], wl["latent"], wl["seed"]), wl["seed"], wl["sigma_pos"], wl["sigma_neg"],
                       wl["noise"], wl["jitter"])
    return ctx


def make_pair_data(F, N, P, U, seed, sigma_pos=0.35, sigma_neg=1.0, noise=0.05, jitter=0.0):
    """Pair mode stand-in for the reference's producer (src/comp-uprjdists.cpp:260-327): P per-patch
    descriptors + the [N,4] Indices table.  Three patches per 3-D point; a descriptor is
    U^T (c_point + s * delta_patch) + noise, so that a matching pair's difference has the latent
    spread sigma_pos and a non-matching one about sigma_neg, like the row-mode generator; `jitter` is that
    generator's log-normal scale, drawn per 3-D point (a pair's rows cannot be scaled one by one: they are
    differences of shared descriptors), which is what makes matches and non-matches overlap."""
    rng = np.random.default_rng(seed)
    k = U.shape[0]
    npts = P // 3
    P = 3 * npts
    point = (np.arange(P) % npts).astype(np.int32)
    centre = (rng.standard_normal((npts, k)) * (sigma_neg / np.sqrt(2.0))).astype(np.float32)
    scale = np.exp(jitter * rng.standard_normal(npts)).astype(np.float32) if jitter > 0 else np.ones(npts, np.float32)
    centre *= scale[:, None]
    desc = np.empty((P, F), np.float32)
    for r0 in range(0, P, 8192):
        r1 = min(P, r0 + 8192)
        z = centre[point[r0:r1]] + (sigma_pos / np.sqrt(2.0)) * scale[point[r0:r1], None] * rng.standard_normal((r1 - r0, k)).astype(np.float32)
        desc[r0:r1] = z @ U + (noise / np.sqrt(2.0)) * rng.standard_normal((r1 - r0, F)).astype(np.float32)
    np.clip(desc, -1.0, 1.0, out=desc)
    a = rng.integers(0, P, N).astype(np.int64)
    b = rng.integers(0, P, N).astype(np.int64)
    match = (np.arange(N) % 2 == 0)
    b[match] = (a[match] + npts * rng.integers(1, 3, int(match.sum()))) % P      # another patch of the same point
    pairs = np.stack([a, point[a], b, point[b]], axis=1).astype(np.int32)
    return desc, pairs


def committed_profile(name):
    try:
        return json.load(open(os.path.join(ROOT, "profiles", name)))
    except Exception:
        return None


CURRENT_ROUND = 4


def committed_pmc(stem):
    """A committed rocprofv3 --pmc summary (profiles/rN_<stem>.json), newest round first, with where it came from: PMC
    counters cannot be read inside a timed run, so these figures are NOT measured by this invocation; a file of an earlier
    round than the code is marked stale."""
    for rnd in range(CURRENT_ROUND, 0, -1):
        name = "r%d_%s.json" % (rnd, stem)
        d = committed_profile(name)
        if d:
            return d, {"file": "profiles/" + name, "round": rnd, "stale": rnd != CURRENT_ROUND,
                       "measured_in_this_run": False}
    return None, None


def pmc_traffic(F, bl):
    """HBM bytes per SYRK launch from the committed rocprofv3 PMC passes (separate FETCH_SIZE /
    WRITE_SIZE runs of this command, gfx950 read correction applied); only reported for the
    configuration it was measured on."""
    d, src = committed_pmc("pmc_syrk")
    if d and F == 8192 and bl == 200:
        return d.get("hbm_bytes_per_launch_corrected"), src
    return None, None


def ncores():
    """Threads the CPU baseline may use: the affinity mask, cut to the cgroup's CPU quota when one is set
    (more threads than the quota only get throttled) and to DLCO_CPU_THREADS when given."""
    try:
        n = len(os.sched_getaffinity(0))
    except Exception:
        n = os.cpu_count() or 1
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                quota, period = txt[0], float(txt[1])
            else:
                quota, period = txt[0], float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if quota not in ("max", "-1") and float(quota) > 0:
                n = max(1, min(n, int(-(-float(quota) // period))))
            break
        except Exception:
            continue
    env = os.environ.get("DLCO_CPU_THREADS")
    if env:
        n = max(1, min(n, int(env)))
    return n


def cpu_baseline(ctx, wl, rows, steps, t0_step):
    """The oracle (reference loop order, OpenBLAS sgemm/ssyevr) timed on this box's host cores on a
    bounded sample: `steps` full training steps teacher-forced from the GPU's steady state (its
    dual average, W and iteration counter after the timed run) on a `rows`-row subset of the same
    synthetic data.  The "no-eigen" figure leaves out E1/E2 (ssyevr + the F^3 back-multiplication),
    i.e. it is the reference's projection + gradient + dual average only."""
    from oracle import ref

    nc = ncores()
    ref.lib()
    F, B, mu, gamma = wl["F"], wl["batch"], wl["mu"], wl["gamma"]
    if ref.blas_kind() != "openblas":
        return {"value": None, "unit": "pair-rows/s", "cores": nc, "kind": "port",
                "sample": "skipped: no OpenBLAS found for the oracle's ssyevr at F=%d" % F}
    ref.set_threads(nc)
    D = ctx.get_rows(0, rows)
    L = (np.arange(rows) % 2 == 0).astype(np.uint8)
    out = {"unit": "pair-rows/s", "cores": nc, "kind": "port"}
    tr = ref.Trainer(D, L, B=B, mu=mu, gamma=gamma, grad_order=0)
    tr.set_state(ctx.t(), ctx.dfavg(), ctx.W())
    w0 = time.perf_counter()
    for _ in range(steps):
        tr.step()
    dt = time.perf_counter() - w0
    tm = tr.timers()
    tr.close()
    no_eig = tm["project"] + tm["grad"] + tm["rda"]
    out["value"] = 2.0 * B * steps / dt
    out["no_eigen_value"] = 2.0 * B * steps / max(no_eig, 1e-9)
    out["seconds_per_step"] = {"total": dt / steps, "project": tm["project"] / steps, "gradient": tm["grad"] / steps,
                               "dual_average": tm["rda"] / steps, "eigen_ssyevr_and_backmultiply": tm["eig"] / steps}
    out["sample"] = ("%d full steps (reference loop order, OpenBLAS sgemm + ssyevr) teacher-forced from the GPU's steady "
                     "state (t=%d, its dfAvg and W) at F=%d B=%d on a %d-row subset of the same synthetic data, %.1f s; "
                     "no_eigen_value = the same steps without E1/E2" % (steps, ctx.t(), F, B, rows, dt))
    if t0_step:
        tr = ref.Trainer(D, L, B=B, mu=mu, gamma=gamma, grad_order=0)
        w0 = time.perf_counter()
        tr.step()
        dt0 = time.perf_counter() - w0
        tm0 = tr.timers()
        tr.close()
        out["worst_case_t0"] = {"value": 2.0 * B / dt0, "seconds": dt0, "gradient_seconds": tm0["grad"],
                                "note": "first iteration from W = 0: every one of the B*B pairs violates (200 sgemms of F x F x 200)"}
    return out


def quality(ctx, wl, with_oracle):
    """Where the trainer stands after the timed steps (outside the timed region): the reference's
    LogStep block (src/pj-learn.cpp:492-587) on the GPU, and the SAME W scored by the oracle's
    ROC sweep (src/misc.cpp:297-332) on the distances of all N rows."""
    e = ctx.log_step()
    q = {"val_loss": e.loss_val, "regul": e.regul, "fpr95": e.fpr95, "auc": e.auc, "dim": e.dim}
    if with_oracle:
        from oracle import ref
        W = ctx.W()
        d_all = ctx.project_sqdist(np.arange(wl["N"], dtype=np.int32), W)
        f_o, a_o = ref.roc_stats(d_all, (np.arange(wl["N"]) % 2 == 0).astype(np.uint8))
        q["oracle_on_same_W"] = {"fpr95": f_o, "auc": a_o, "abs_fpr95_diff": abs(f_o - e.fpr95),
                                 "note": "oracle ROC sweep on the GPU's distances of all N rows; the +-0.1 % gate is "
                                         "abs_fpr95_diff <= 1e-3 (tests/test_full_width_gpu.py also scores the oracle's "
                                         "own ssyevr W)"}
    return q


def reference_run(dlco, wl, data_ctx, iters, logstep, kw):
    """What the pj-learn program does with its time (src/pj-learn.cpp:305-589), as opposed to the steady-state step
    the headline times: iterations t = 0..iters from W = 0 (the start-up transient, rank of several hundred, is
    inside), with the LogStep block - validation over the 50 000 + 50 000 held-out rows and, on a new best objective,
    ComputePJStats over all N rows - at t = logstep, 2*logstep, ... like the reference.  Runs on a second context
    that shares the resident Distance matrix.  Ttime / Vtime are the fields of the reference's own log lines."""
    c2 = build_context(dlco, wl, data_from=data_ctx, **kw)
    c2.sync()
    windows = []
    w0 = time.perf_counter()
    t_train = t_log = 0.0
    done = 0                                     # iterations run: t = 0 .. done-1
    while done <= iters:
        upto = min(done + logstep + (1 if done == 0 else 0), iters + 1)      # the first log comes inside iteration t = logstep
        a = time.perf_counter()
        c2.steps(upto - done)
        c2.sync()
        b = time.perf_counter()
        t_train += b - a
        done = upto
        if (done - 1) % logstep != 0 or done - 1 == 0:
            break                                # iters is not a multiple of logstep: the tail has no log line
        e = c2.log_step()
        c = time.perf_counter()
        t_log += c - b
        windows.append({"t": done - 1, "Ttime": b - a, "Vtime": c - b, "rank": e.rank, "best": bool(e.is_best)})
    total = time.perf_counter() - w0
    B = wl["batch"]
    cn = c2.counters()
    c2.close()
    return {"value": 2.0 * B * done / total, "unit": "pair-rows/s", "iterations": done, "log_step": logstep, "seconds": total,
            "train_seconds": t_train, "log_step_seconds": t_log, "nonconverged_steps": cn["nonconverged"],
            "windows": windows,
            "note": "pj-learn's own loop from W = 0: every iteration and every LogStep block inside the clock (the headline "
                    "`value` times steady-state steps only, like the reference's Ttime per 100 steps divided out)"}


def self_launch(args, argv):
    """`python bench.py --gpus N` with N > 1 and no WORLD_SIZE in the environment: this process becomes the launcher.  It
    has not imported torch.cuda, the product library or anything else that touches HIP (and never will): it starts
    `python -m torch.distributed.run --nproc-per-node N bench.py <same flags>` as a CHILD process (no exec of a process
    that has initialised the GPU - nothing here has), passes rank 0's single JSON line through to its own stdout and
    exits with the child's code."""
    import socket
    import subprocess
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env["DLCO_BENCH_LAUNCHED"] = "1"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + argv
    p = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, text=True)
    out, _ = p.communicate()
    lines = [l for l in out.splitlines() if l.lstrip().startswith("{")]
    if lines:
        print(lines[-1], flush=True)
    elif p.returncode == 0:
        print("bench.py: the ranks exited 0 without a JSON line", file=sys.stderr)
        return 1
    return p.returncode


def dry_launch(args):
    """`--dry-launch`: the launcher path without a GPU - every rank joins a gloo process group, the step is the CPU oracle's
    (test infrastructure, a tiny problem), the barrier / max-over-ranks timing and the one JSON line are the real
    code's.  What tests/test_bench_launcher.py runs in the container."""
    import torch
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29519")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import ref
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from util import synth
    D, L = synth(600, 32, k=6, seed=5)
    tr = ref.Trainer(D, L, B=8, mu=0.004, gamma=0.5, grad_order=1)
    for _ in range(args.warmup):
        tr.step()
    dist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        tr.step()
    dist.barrier()
    tmax = torch.tensor([time.perf_counter() - t0], dtype=torch.float64)
    dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    ranks = torch.tensor([1], dtype=torch.int64)
    dist.all_reduce(ranks)
    if rank == 0:
        print(json.dumps({"metric": "pj-learn patch-pairs/sec", "dry_launch": True, "value": 2.0 * 8 * args.steps * world / float(tmax.item()),
                          "unit": "pair-rows/s", "n_gpus": world, "ranks_joined": int(ranks.item()), "steps": args.steps, "warmup": args.warmup,
                          "data": "synthetic (CPU oracle, launcher rehearsal: not a measurement)"}), flush=True)
    tr.close()
    dist.destroy_process_group()
    return 0


def side_run(dlco, wl, kw, steps, warmup, data_from=None):
    """A short timed run of another BASELINE configuration in the same invocation (so that the driver's record carries a
    number for it): burn-in to the named rank regime, warm-up, `steps` timed full steps, then where the trainer stands."""
    c = build_context(dlco, wl, data_from=data_from, **kw)
    c.steps(wl.get("burn_in", 300))
    c.steps(warmup)
    c.sync()
    t0 = time.perf_counter()
    c.steps(steps)
    c.sync()
    dt = time.perf_counter() - t0
    e = c.log_step()
    cn = c.counters()
    r = {"workload": wl["name"], "value": 2.0 * wl["batch"] * steps / dt, "unit": "pair-rows/s", "ms_per_step": dt / steps * 1e3, "steps": steps,
         "warmup": warmup, "burn_in_steps": wl.get("burn_in", 300), "rank": int(e.rank), "fpr95": e.fpr95, "auc": e.auc,
         "nonconverged_steps": cn["nonconverged"], "dtype": "bf16 MFMA inputs + fp32 accumulate in every GEMM over the resident matrix"
         if kw.get("grad_bf16") else "f32"}
    c.close()
    return r


class Runner:
    """One trainer (context + optional distributed wrapper) and its timed run."""

    def __init__(self, dlco, ctx, trainer, use_dist, torch=None, dist=None):
        self.dlco, self.ctx, self.trainer, self.use_dist, self.torch, self.dist = dlco, ctx, trainer, use_dist, torch, dist

    def run(self, n):
        if self.trainer is None:
            self.ctx.steps(n)
        else:
            self.trainer.steps(n)

    def barrier(self):
        self.ctx.sync()
        if self.use_dist:
            self.torch.cuda.synchronize()
            self.dist.barrier()
            self.torch.cuda.synchronize()

    def timed(self, steps):
        self.barrier()
        t0 = time.perf_counter()
        self.run(steps)
        self.barrier()
        dt = time.perf_counter() - t0
        if self.use_dist:
            tmax = self.torch.tensor([dt], dtype=self.torch.float64, device="cuda")
            self.dist.all_reduce(tmax, op=self.dist.ReduceOp.MAX)
            dt = float(tmax.item())
        return dt

    def close(self):
        if self.trainer is not None and hasattr(self.trainer, "close"):
            self.trainer.close()                          # torch's current stream must not outlive the context
        self.ctx.close()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--config", choices=sorted(WORKLOADS), default="c2",
                    help="c2 = BASELINE configs[1] (rank ~64, the metric's configuration); c3 = configs[2] (rank ~128); "
                         "ref544 = the shape the reference itself ran (500000 x 544, mu 0.001, gamma 0.1)")
    ap.add_argument("--burn-in", type=int, default=None,
                    help="full training steps run while the workload is set up, to reach the rank regime the "
                         "BASELINE configuration names (default: 300 for c2, 500 for c3; 0 = time the start-up transient)")
    ap.add_argument("--F", type=int, default=None)
    ap.add_argument("--N", type=int, default=None)
    ap.add_argument("--batch", type=int, default=None, help="pair-rows per class PER GPU")
    ap.add_argument("--mu", type=float, default=None)
    ap.add_argument("--gamma", type=float, default=None)
    ap.add_argument("--latent", type=int, default=None)
    ap.add_argument("--jitter", type=float, default=None)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-rows", type=int, default=None, help="rows of the CPU baseline's subset (default 4096; 20000 for ref544)")
    ap.add_argument("--cpu-steps", type=int, default=None, help="full steps of the CPU baseline (default 3; 50 for ref544)")
    ap.add_argument("--cpu-t0", action="store_true", help="also time the worst-case first iteration (W = 0) on the CPU")
    ap.add_argument("--pair-mode", action="store_true",
                    help="train from per-patch descriptors + the Indices table (dlco_set_pairs), differences formed "
                         "inside the kernels, instead of the materialised Distance matrix")
    ap.add_argument("--patches", type=int, default=65536, help="pair mode: number of patch descriptors")
    ap.add_argument("--force-dist", action="store_true",
                    help="developer check: run the N > 1 code path (process group, trainer, callbacks) with one rank")
    ap.add_argument("--dp-mode", choices=["shard", "allreduce"], default="shard",
                    help="N > 1 headline mode: column-sharded dual average (all-gathers of a few MB) or replicated "
                         "dual average with an F x F all-reduce per step")
    ap.add_argument("--scaling", choices=["weak", "strong"], default="strong",
                    help="N > 1 headline mode: global batch fixed at --batch, i.e. the reference's optimiser (strong, default), "
                         "or per-GPU batch fixed (weak: a different optimiser, reported under other_modes)")
    ap.add_argument("--no-other-modes", action="store_true", help="N > 1: measure the headline mode only")
    ap.add_argument("--bf16", action="store_true",
                    help="BASELINE configs[4] variant: the gradient SYRK on the bf16 matrix cores with fp32 accumulation "
                         "(cfg.grad_bf16; not the reference's arithmetic, gated on the FPR@95 band)")
    ap.add_argument("--reference-iters", type=int, default=1000,
                    help="single GPU: also time pj-learn's own loop from W = 0 for this many iterations with the LogStep block "
                         "every --reference-logstep (reported as reference_run; 0 = skip)")
    ap.add_argument("--reference-logstep", type=int, default=100)
    ap.add_argument("--no-profile", action="store_true",
                    help="developer check: no HIP events inside the timed region (the roofline / breakdown fields are then empty)")
    ap.add_argument("--guard", type=int, default=None, help="tracker guard vectors (library default 32)")
    ap.add_argument("--eig-tol", type=float, default=None, help="tracker tolerance (library default 2e-4)")
    ap.add_argument("--dry-launch", action="store_true",
                    help="rehearse the N-rank launch on the CPU (gloo + the oracle's step on a toy problem): no GPU is touched")
    ap.add_argument("--no-other-configs", action="store_true",
                    help="single GPU, default workload: skip the short configs[2] and configs[4] runs reported as other_configs")
    args = ap.parse_args()

    # N > 1 without a launcher around us: become the launcher BEFORE anything touches the GPU (see self_launch)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args, sys.argv[1:]))
    if args.dry_launch:
        sys.exit(dry_launch(args))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if rank == 0:
            print("bench.py: --gpus %d but WORLD_SIZE=%d" % (args.gpus, world), file=sys.stderr)
        sys.exit(2)

    dlco = importlib.import_module("opencv-dlco_amd")
    wl = dict(WORKLOADS[args.config])
    for k_, v_ in (("F", args.F), ("N", args.N), ("batch", args.batch), ("mu", args.mu), ("gamma", args.gamma),
                   ("latent", args.latent), ("jitter", args.jitter)):
        if v_ is not None:
            wl[k_] = v_
    F, N, Bl = wl["F"], wl["N"], wl["batch"]
    if args.burn_in is None:
        args.burn_in = wl.get("burn_in", 300)
    if args.cpu_steps is None:
        args.cpu_steps = wl.get("cpu_steps", 3)
    if args.cpu_rows is None:
        args.cpu_rows = wl.get("cpu_rows", 4096)

    use_dist = world > 1 or args.force_dist
    if args.force_dist:
        os.environ.setdefault("MASTER_PORT", "29517")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        os.environ["DLCO_FORCE_SHARD"] = "1"
    # RCCL prints a version banner on stdout when a communicator is created: keep fd 1 clean so
    # that the only thing ever written to the real stdout is the one JSON line of rank 0
    real_stdout = os.dup(1)
    sys.stdout.flush()
    os.dup2(2, 1)
    torch = dist = ddist = None
    if use_dist:
        import torch
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        ddist = importlib.import_module("opencv-dlco_amd.dist")
    kw = dict(eig_guard=args.guard, eig_tol=args.eig_tol, grad_bf16=1 if args.bf16 else 0)

    def make_runner(dp_mode, scaling, data_from=None):
        """weak: global batch world*Bl; strong: global batch Bl (must divide by world)."""
        Bg = Bl * world if scaling == "weak" else Bl
        if Bg % world != 0:
            return None
        shard = use_dist and dp_mode == "shard" and F % (128 * world) == 0
        if args.pair_mode:
            ctx = dlco.Context(F, N, B=Bg, mu=wl["mu"], gamma=wl["gamma"], device=local_rank, rank=rank, world=world,
                               shard=1 if shard else 0, **kw)
            desc, pairs = make_pair_data(F, N, args.patches, make_U(F, wl["latent"], wl["seed"]), wl["seed"], wl["sigma_pos"],
                                         wl["sigma_neg"], wl["noise"], wl["jitter"])
            ctx.set_pairs(desc, pairs)                  # identical bytes on every rank (dataset replicated)
        else:
            ctx = build_context(dlco, wl, B=Bg, device=local_rank, rank=rank, world=world, shard=1 if shard else 0,
                                data_from=data_from, **kw)
        trainer = None
        if use_dist:
            dev = torch.device("cuda", local_rank)
            trainer = (ddist.ShardedTrainer(ddist.HipShardEngine(dlco, ctx, dev)) if shard
                       else ddist.DataParallelTrainer(ddist.HipEngine(dlco, ctx, dev)))
        r = Runner(dlco, ctx, trainer, use_dist, torch, dist)
        r.shard, r.Bg, r.dp_mode, r.scaling = shard, Bg, ("shard" if shard else "allreduce"), scaling
        return r

    R = make_runner(args.dp_mode, args.scaling)
    ctx = R.ctx
    dev_name, _, _ = ctx.device_name()
    B = R.Bg

    R.run(args.burn_in)
    R.run(args.warmup)
    # timed region: HIP events around the dominant kernel only (two records per step); the other kernel groups are timed
    # over a second, untimed stretch of steps right after it - every event record costs queue time (all four groups: ~3 %
    # of the step), and `value` should not pay for the breakdown
    ctx.profile_enable(0 if args.no_profile else 2)
    es0 = ctx.eig_stats()
    cn0 = ctx.counters()
    dt = R.timed(args.steps)
    es1 = ctx.eig_stats()
    cn1 = ctx.counters()
    n_syrk, ms_syrk = ctx.profile_read("grad_syrk")
    ctx.profile_enable(0)
    bsteps = 0 if args.no_profile else max(1, min(args.steps, 100))
    n_prod = n_jac = n_prj = n_ru = 0
    ms_prod = ms_jac = ms_prj = ms_ru = 0.0
    if bsteps:
        ctx.profile_enable(1)
        R.run(bsteps)
        ctx.sync()
        n_prod, ms_prod = ctx.profile_read("eig_product")
        n_jac, ms_jac = ctx.profile_read("jacobi")
        n_prj, ms_prj = ctx.profile_read("project")
        n_ru, ms_ru = ctx.profile_read("rank_update")
        ctx.profile_enable(0)
    bdiv = max(bsteps, 1)

    rank_now = ctx.W().shape[0]
    q = None
    if not R.shard:                                       # a sharded context validates the same replicated W
        q = quality(ctx, wl, with_oracle=(rank == 0 and world == 1 and not args.pair_mode))
    value = 2.0 * B * args.steps / dt
    shard = R.shard
    # dominant kernel of the hot path: the fused weighted-SYRK gradient + dual average.
    # algorithmic flops per launch (dense, no symmetry credit): 2 * K * F^2 where K is the
    # measured mean number of rows with a non-zero violation count (the reference skips the others
    # too, src/pj-learn.cpp:378), K <= 2*B.
    k_mean = (cn1["active_rows"] - cn0["active_rows"]) / max(cn1["steps"] - cn0["steps"], 1)
    flops_launch = 2.0 * k_mean * F * F
    if shard:                      # the rank's launch covers its F x F/world column slab over the whole global batch
        flops_launch /= world
    t_syrk = ms_syrk / max(n_syrk, 1) * 1e-3
    ach = flops_launch / t_syrk / 1e12 if n_syrk else None
    exec_flops = flops_launch if shard else flops_launch * (F // 128 + 1) / (2.0 * (F // 128))
    ach_exec = exec_flops / t_syrk / 1e12 if n_syrk else None
    traffic, traffic_src = pmc_traffic(F, Bl) if world == 1 and not args.pair_mode else (None, None)
    sq, sq_src = committed_pmc("pmc_sq")
    mfma_busy = None
    try:
        if F == 8192:
            mfma_busy = sq["kernels"]["syrk_planes_kernel"]["mfma_util"]   # committed --pmc pass of this command (tools/collect_sq.sh)
    except Exception:
        pass
    # what the matrix pipe in use is asked to do: the fp32 result comes from bf16 MFMAs on three-way split operands, six per
    # product term (one per term with --bf16: operands rounded once)
    Fd = ctx.device_width()
    ntile = Fd // 128
    packed = world == 1 and not shard and ntile <= 64
    issued_mult = 1.0 if args.bf16 else 6.0
    exec_flops_dev = (flops_launch if shard else 2.0 * k_mean * Fd * Fd * (ntile + 1) / (2.0 * ntile))   # on the padded width the kernels run at
    issued = issued_mult * exec_flops_dev
    ach_issued = issued / t_syrk / 1e12 if n_syrk else None
    dfavg_bytes = (ntile * (ntile + 1) // 2) * 65536.0 if packed else 4.0 * Fd * Fd / (world if shard else 1)
    # per pair-row 2F^2 + 2Fr flops (projection + gradient, dense)
    flops_pair_row = 2.0 * F * F + 2.0 * F * rank_now
    t_kernel_path = (ms_syrk / max(args.steps, 1) + ms_prj / bdiv) * 1e-3          # P1+P2 + Q1+U1 per step, HIP events
    t_prod = ms_prod / max(n_prod, 1) * 1e-3
    out = {
        "metric": "pj-learn patch-pairs/sec",
        "value": value,
        "unit": "pair-rows/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": dt / args.steps * 1e3,
        "higher_is_better": True,
        "scaling": R.scaling,
        "vs_baseline": None,
        "dtype": "bf16 (gradient MFMA inputs; f32 accumulate, f32 elsewhere)" if args.bf16 else "f32",
        "data": "synthetic",
        "config": {
            "workload": "pj-learn %s: %d pair-rows x PR-dim %d, batch %d+%d per GPU (global %d+%d), mu=%g gamma=%g, %s, rank %d after %d steps"
                        % (wl["name"], N, F, B // world, B // world, B, B, wl["mu"], wl["gamma"],
                           "bf16 MFMA + fp32 accumulate in the gradient, projection and statistics GEMMs (configs[4] variant), fp32 elsewhere"
                           if args.bf16 else "fp32", rank_now, args.burn_in + args.warmup + args.steps + bsteps)
                        + (" [pair mode: %d patch descriptors + Indices, differences formed in the kernels]" % args.patches
                           if args.pair_mode else ""),
            "generator": {k_: wl[k_] for k_ in ("latent", "sigma_pos", "sigma_neg", "noise", "jitter", "seed")},
            "device": dev_name,
            "burn_in_steps": args.burn_in,
            "state_after_run": q,
            "parallelism": ("dp%d (replicated data, batch slots sharded; dual average sharded by columns: all-gather of the "
                            "2B distances and of the tracker products' column slabs, no F x F exchange; collectives issued by %s)"
                            % (world, "the library through RCCL" if getattr(R.trainer, "native", False) else "a torch.distributed callback")) if shard else
                           ("dp%d (replicated data, batch slots sharded, all-gather dists + all-reduce gradient)" % world),
            "combinations_per_s": float(B) * B * args.steps / dt,
        },
        "roofline": {
            "bound": "mfma",
            "pipe": ("bf16 MFMA (v_mfma_f32_32x32x16_bf16), operands rounded to bf16 once, fp32 accumulation" if args.bf16 else
                     "bf16 MFMA (v_mfma_f32_32x32x16_bf16) producing fp32 results: operands split three ways, six bf16 MFMAs per fp32 product term"),
            "kernel": "grad_syrk_rda (fused weighted SYRK + dual average: syrk_split_rows_kernel + syrk_planes_kernel, both inside the timed launch)",
            "achieved": ach_issued,
            "peak": 2500.0,
            "unit": "TFLOP/s",
            "frac": (ach_issued / 2500.0) if ach_issued else None,
            "note": "frac is a utilisation of the pipe in use: MFMA flops ISSUED by the launch (%g x the fp32 flops of the tiles it computes - "
                    "those on or above the diagonal, at the device width %d, K = rows with a non-zero violation count) / the launch's HIP-event "
                    "time / the dense bf16 MFMA peak.  fp32_equivalent_* prices the executed fp32 flops against the fp32 MFMA peak the contract "
                    "names for dtype f32 (it can pass 1.0 because the work runs on the faster pipe: NOT a utilisation); algorithmic_* is SURVEY "
                    "8(d)'s dense accounting (2*K*F^2, no symmetry credit, caller's width) over the same time, also not a utilisation.  "
                    "mfma_busy_frac_pmc and traffic come from committed PMC passes (see their *_source)" % (issued_mult, Fd),
            "issued_mfma_flops_per_launch": issued,
            "fp32_equivalent_achieved": ach_exec,
            "fp32_equivalent_frac": (ach_exec / PEAK_F32_MFMA_TFLOPS) if ach_exec else None,
            "algorithmic_achieved": ach,
            "algorithmic_frac": (ach / PEAK_F32_MFMA_TFLOPS) if ach else None,
            "traffic": traffic,
            "traffic_source": traffic_src,
            "traffic_unit": "HBM bytes per launch (2*FETCH_SIZE + WRITE_SIZE); algorithmic bytes 8*F*F + 4*K*F = %d; the packed layout needs %d"
                            % (int(8 * F * F + 4 * k_mean * F), int(2 * dfavg_bytes + 12 * k_mean * Fd)),
            "avg_launch_ms": t_syrk * 1e3,
            "launches": n_syrk,
            "algorithmic_flops_per_launch": flops_launch,
            "mean_active_rows_per_launch": k_mean,
            "executed_flops_per_launch": exec_flops,
            "mfma_busy_frac_pmc": mfma_busy,
            "mfma_busy_source": sq_src if mfma_busy is not None else None,
            # kernel path only (P1+P2+V1+Q1+U1), (ii) end to end incl. the PSD projection
            "flops_per_pair_row": flops_pair_row,
            "kernel_path_frac": (2.0 * B / world * flops_pair_row / t_kernel_path / 1e12 / PEAK_F32_MFMA_TFLOPS) if t_kernel_path > 0 else None,
            "end_to_end_frac": value / world * flops_pair_row / 1e12 / PEAK_F32_MFMA_TFLOPS,
            "tracker_product": {
                "kernel": "one pass of the eigen tracker over dfAvg (split-bf16 MFMA), HBM-bound",
                "avg_launch_ms": t_prod * 1e3 if n_prod else None,
                "launches_per_step": n_prod / bdiv,
                "necessary_bytes_per_launch": dfavg_bytes,
                "layout": "packed upper 128 x 128 tiles, each fetched once per pass" if packed else "full matrix (column slab of a sharded rank)",
                "hbm_frac": (dfavg_bytes / t_prod / 1e12 / PEAK_HBM_TBS) if n_prod else None,
            },
            "tracker_nonconverged_steps": cn1["nonconverged"] - cn0["nonconverged"],
        },
        # one entry per kernel group of breakdown_ms_per_step, each priced on what it has to do: issued MFMA flops or the
        # bytes the pass cannot avoid, its time per step, the fraction of that unit's peak
        "step_roofline": [
            {"group": "grad_syrk", "bound": "mfma (bf16 pipe)", "ms_per_step": ms_syrk / args.steps, "launches_per_step": n_syrk / max(args.steps, 1),
             "work_per_launch": issued, "work_unit": "bf16 MFMA flops issued", "achieved": ach_issued, "peak": 2500.0, "unit": "TFLOP/s",
             "frac": (ach_issued / 2500.0) if ach_issued else None},
            {"group": "eig_products", "bound": "hbm", "ms_per_step": ms_prod / bdiv, "launches_per_step": n_prod / bdiv,
             "work_per_launch": dfavg_bytes, "work_unit": "bytes of the dual average one pass must read",
             "achieved": (dfavg_bytes / t_prod / 1e9) if n_prod else None, "peak": PEAK_HBM_TBS * 1e3, "unit": "GB/s",
             "frac": (dfavg_bytes / t_prod / 1e12 / PEAK_HBM_TBS) if n_prod else None},
            # the step's first filter term from its own rank update (kernels_rankupd.hip): in place of one pass over dfAvg it
            # reads the x planes of the active rows, Y and Q of the block, and writes the term and its two-way planes
            {"group": "eig_rank_update", "bound": "latency / L2 (two short launches: coefficient fragments, then 256 workgroups of a few hundred MFMAs)",
             "ms_per_step": ms_ru / bdiv, "launches_per_step": n_ru / bdiv,
             "work_per_launch": 6.0 * k_mean * Fd + 4.0 * 4.0 * es1["block_rows"] * Fd, "work_unit": "bytes (x planes in, Y and Q in, term and planes out)",
             "achieved": ((6.0 * k_mean * Fd + 16.0 * es1["block_rows"] * Fd) / (ms_ru / n_ru * 1e-3) / 1e9) if n_ru else None,
             "peak": PEAK_HBM_TBS * 1e3, "unit": "GB/s",
             "frac": ((6.0 * k_mean * Fd + 16.0 * es1["block_rows"] * Fd) / (ms_ru / n_ru * 1e-3) / 1e12 / PEAK_HBM_TBS) if n_ru else None},
            {"group": "eig_jacobi", "bound": "latency (m x m problem on one workgroup; a chain of dependent rotation rounds)",
             "ms_per_step": ms_jac / bdiv, "launches_per_step": n_jac / bdiv, "frac": None},
            {"group": "project", "bound": "latency (2B gathered rows x F x 4 bytes = %d per step)" % int(8 * B * Fd / world),
             "ms_per_step": ms_prj / bdiv, "launches_per_step": n_prj / bdiv, "frac": None},
            {"group": "other", "bound": "orthonormalisation (Gram, L^-1, L^-1 Z), rotation GEMMs, reductions, violation counts, launch gaps",
             "ms_per_step": dt / args.steps * 1e3 - ms_syrk / args.steps - (ms_prod + ms_jac + ms_prj + ms_ru) / bdiv, "frac": None},
        ],
        "breakdown_ms_per_step": {
            "grad_syrk": ms_syrk / args.steps,
            "eig_products": ms_prod / bdiv,
            "eig_rank_update": ms_ru / bdiv,
            "eig_jacobi": ms_jac / bdiv,
            "project": ms_prj / bdiv,
            "note": "grad_syrk: HIP events inside the timed region; the other groups: over the %d steps that follow it" % bsteps,
            "eig_iters_per_step": (es1["iters"] - es0["iters"]) / args.steps,
            "eig_jacobi_sweeps_per_step": (es1["jacobi_sweeps"] - es0["jacobi_sweeps"]) / args.steps,
            "eig_product_rows_per_step": (es1["product_rows"] - es0["product_rows"]) / args.steps,
            "eig_block_rows": es1["block_rows"],
            "eig_rank_update_passes_per_step": (cn1["rank_update_passes"] - cn0["rank_update_passes"]) / args.steps,
            "eig_locked_passes_per_step": (cn1["locked_passes"] - cn0["locked_passes"]) / args.steps,
        },
    }
    # ---- N > 1: the other two modes SURVEY 8(d)/(e) and BASELINE configs[3] name, measured after the headline
    if world > 1 and not args.no_other_modes and not args.pair_mode:
        others = {}
        for dp_mode, scaling in (("allreduce", "strong"), ("shard", "weak")):
            if dp_mode == R.dp_mode and scaling == R.scaling:
                dp_mode, scaling = "shard", "strong"
            r2 = make_runner(dp_mode, scaling, data_from=ctx)
            if r2 is None:
                continue
            r2.run(args.burn_in)
            r2.run(args.warmup)
            dt2 = r2.timed(args.steps)
            others["%s_%s" % (r2.dp_mode, scaling)] = {
                "value": 2.0 * r2.Bg * args.steps / dt2, "unit": "pair-rows/s", "ms_per_step": dt2 / args.steps * 1e3,
                "global_batch": "%d+%d" % (r2.Bg, r2.Bg), "scaling": scaling, "dp_mode": r2.dp_mode,
                "rank": int(r2.ctx.W().shape[0]), "nonconverged": r2.ctx.counters()["nonconverged"]}
            r2.close()
        out["other_modes"] = others
    if world == 1 and not args.pair_mode and args.reference_iters > 0:
        out["reference_run"] = reference_run(dlco, wl, ctx, args.reference_iters, args.reference_logstep, kw)
    if wl.get("reference_logged"):
        out["reference_logged"] = wl["reference_logged"]
    # the other single-GPU configurations BASELINE names, timed by the same invocation: configs[2] (rank ~128) and the
    # configs[4] variant (bf16 MFMA + fp32 accumulate) on the headline workload
    if (world == 1 and args.config == "c2" and not args.pair_mode and not args.bf16 and not args.no_other_configs
            and all(v is None for v in (args.F, args.N, args.batch, args.mu, args.gamma, args.latent, args.jitter))):
        n_side = max(20, min(args.steps, 100))
        kw_bf = dict(kw, grad_bf16=1)
        out["other_configs"] = {
            "configs[4] bf16 variant on configs[1]": side_run(dlco, wl, kw_bf, n_side, args.warmup, data_from=ctx),
            "configs[2]": side_run(dlco, dict(WORKLOADS["c3"]), kw, n_side, args.warmup),
        }
    if rank == 0:
        if world == 1 and not args.no_cpu_baseline and not args.pair_mode:
            out["cpu_baseline"] = cpu_baseline(ctx, wl, args.cpu_rows, args.cpu_steps, args.cpu_t0)
        else:
            out["cpu_baseline"] = None
        sys.stdout.flush()
        os.write(real_stdout, (json.dumps(out) + "\n").encode())
    R.close()
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
