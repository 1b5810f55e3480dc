"""Data-parallel pj-learn over torch.distributed (backend "nccl" = RCCL over xGMI on ROCm).

One process per GPU.  The global batch of B positive and B negative pair-rows is sampled
identically on every rank (same cv::RNG stream); rank g owns the batch slots
[g*B/G, (g+1)*B/G).  Per step there are exactly two exchanges:

    begin  : project the rank's slots                     -> its slice of DIST  [G][2B/G] f32
    all_gather(DIST)                                       (1.6 kB at B = 200)
    grad   : global violation counts, the rank's partial   -> GRAD [F,F] f32
             gradient P^T diag(rho) P - N^T diag(kappa) N
    all_reduce(GRAD, SUM)                                  (268 MB at F = 8192)
    finish : dual average + PSD projection, replicated

That protocol moves F*F floats per step and stops scaling once the all-reduce (>= 0.44 ms at
F = 8192 even on seven xGMI links) outweighs the 0.2 ms of gradient MFMA time.  ShardedTrainer
is the scalable layout (SURVEY 8e "output sharding"): the dual average is sharded by columns,
rank g computes its F x F/G slab of the gradient over the whole global batch and keeps it, and
only the 2B distances and the m x F/G column slabs of the eigen tracker's ~6 products per step
(m ~ 100 rows) are all-gathered — a few MB instead of 268 MB.

The trainer below is backend-agnostic: `engine` is any object with the three phase methods
and two torch tensors (`dist`, `grad`) that alias its exchange buffers.  The product engine
is HipEngine (libdlco.so); the CPU tests drive the same class with an oracle-backed engine
over gloo.
"""
import torch
import torch.distributed as dist


class HipEngine:
    """libdlco.so context whose exchange buffers are torch tensors (so RCCL can move them)."""

    def __init__(self, dlco, ctx, device):
        self.dlco, self.ctx = dlco, ctx
        B = ctx.B
        _, grad_bytes = ctx.dev_buffer(dlco.BUF_GRAD)       # (device width)^2 floats: F padded to whole tiles
        self.dist = torch.zeros(2 * B, dtype=torch.float32, device=device)
        self.grad = torch.zeros(grad_bytes // 4, dtype=torch.float32, device=device)
        torch.cuda.synchronize(device)
        ctx.bind_buffer(dlco.BUF_DIST, self.dist.data_ptr(), self.dist.numel() * 4)
        ctx.bind_buffer(dlco.BUF_GRAD, self.grad.data_ptr(), self.grad.numel() * 4)
        self.device = device

    def begin(self):
        self.ctx.step_begin()
        self.ctx.sync()                 # library stream -> visible to torch's stream

    def grad_phase(self):
        torch.cuda.current_stream(self.device).synchronize()
        self.ctx.step_grad()
        self.ctx.sync()

    def finish(self):
        torch.cuda.current_stream(self.device).synchronize()
        self.ctx.step_finish()


class DataParallelTrainer:
    def __init__(self, engine, group=None):
        self.e = engine
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0

    def step(self):
        e = self.e
        e.begin()
        if self.world > 1:
            per = e.dist.numel() // self.world
            mine = e.dist[self.rank * per:(self.rank + 1) * per]
            dist.all_gather_into_tensor(e.dist, mine.clone(), group=self.group)
        e.grad_phase()
        if self.world > 1:
            dist.all_reduce(e.grad, op=dist.ReduceOp.SUM, group=self.group)
        e.finish()

    def steps(self, n):
        for _ in range(n):
            self.step()


class HipShardEngine:
    """A sharded libdlco.so context (cfg.shard = 1) whose exchange buffers are torch tensors."""

    def __init__(self, dlco, ctx, device):
        self.dlco, self.ctx, self.device = dlco, ctx, device
        self.BUF_DIST, self.BUF_GATHER = dlco.BUF_DIST, dlco.BUF_GATHER
        _, nbytes = ctx.dev_buffer(dlco.BUF_GATHER)
        self.gather = torch.empty(nbytes // 4, dtype=torch.float32, device=device)
        self.dist = torch.zeros(2 * ctx.B, dtype=torch.float32, device=device)
        torch.cuda.synchronize(device)
        ctx.bind_buffer(dlco.BUF_GATHER, self.gather.data_ptr(), nbytes)
        ctx.bind_buffer(dlco.BUF_DIST, self.dist.data_ptr(), self.dist.numel() * 4)
        self.stream = torch.cuda.ExternalStream(ctx.stream(), device=device)

    def stream_guard(self):
        return torch.cuda.stream(self.stream)             # collectives are ordered on the library's stream

    def enable_native_rccl(self, group=None):
        """Collective over `group`: gives the library its own RCCL communicator (dlco_comm_init), so
        that the step's all-gathers are ncclAllGather calls issued from C++ on the library's stream
        with no Python in between.  Returns True when every rank succeeded; otherwise the
        communicators are dropped again and the caller keeps the torch.distributed callback."""
        import os
        rank = dist.get_rank(group)
        rccl_path = os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so")
        if not os.path.exists(rccl_path):
            rccl_path = None
        ident = [None]
        if rank == 0:
            try:
                ident[0] = self.dlco.comm_unique_id(rccl_path)
            except Exception:                             # noqa: BLE001 - no usable librccl: everyone falls back
                ident[0] = None
        dist.broadcast_object_list(ident, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
        ok = 0
        if ident[0] is not None:
            try:
                self.ctx.comm_init(ident[0], rccl_path)
                ok = 1
            except Exception:                             # noqa: BLE001
                ok = 0
        flag = torch.tensor([ok], dtype=torch.int32, device=self.device)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=group)
        if int(flag.item()) != 1:
            self.ctx.comm_destroy()
            return False
        return True

    def set_allgather(self, fn):
        self.ctx.set_allgather(fn)

    def step(self):
        self.ctx.step()

    def steps(self, n):
        self.ctx.steps(n)


class ShardedTrainer:
    """Column-sharded dual average (cfg.shard = 1): the whole step runs inside the engine (for
    HipShardEngine: dlco_step) and calls back for its all-gathers, which are issued on the engine's
    own stream (no host sync).  `engine` provides the exchange tensors `dist` and `gather`, the ids
    BUF_DIST / BUF_GATHER, `stream_guard()`, `set_allgather(fn)`, `step()`; the CPU tests drive the
    same class with an oracle-backed engine over gloo."""

    def __init__(self, engine, group=None, native=None):
        self.e, self.group = engine, group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self._views = {}
        # Preferred: the library talks to RCCL itself (no Python on the step's critical path).  The
        # torch.distributed callback below stays registered as the fallback.
        import os
        if native is None:
            native = os.environ.get("DLCO_NATIVE_RCCL", "1") != "0"
        self.native = bool(native and dist.is_initialized() and hasattr(engine, "enable_native_rccl")
                           and engine.enable_native_rccl(group))
        # The callback runs about six times per step on the critical path, so it is kept to one
        # torch call: the engine's stream becomes this thread's current stream once (instead of a
        # context manager per call), and the all-gather is in place (send buffer = the rank's own
        # chunk of the receive buffer, the layout NCCL/RCCL define as in-place) when the backend
        # accepts it — probed here with a collective every rank takes part in.
        self._inplace = False
        if dist.is_initialized():
            with engine.stream_guard():
                probe = torch.zeros(self.world * 4, dtype=torch.float32, device=engine.dist.device)
                probe[self.rank * 4:(self.rank + 1) * 4] = float(self.rank + 1)
                try:
                    dist.all_gather_into_tensor(probe, probe[self.rank * 4:(self.rank + 1) * 4], group=group)
                    want = torch.arange(1, self.world + 1, dtype=torch.float32).repeat_interleave(4)
                    ok = bool(torch.equal(probe.cpu(), want))
                except Exception:                         # noqa: BLE001 - backend refuses aliased buffers
                    ok = False
                flag = torch.tensor([1 if ok else 0], dtype=torch.int32, device=engine.dist.device)
                dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=group)    # all ranks must agree on the mode
                self._inplace = bool(flag.item() == 1)
        stream = getattr(engine, "stream", None)
        self._restore = None
        if stream is not None:
            self._restore = torch.cuda.current_stream(engine.device)
            torch.cuda.set_stream(stream)                 # collectives are ordered on the library's stream
        engine.set_allgather(self._allgather)

    def close(self):
        """Gives this thread its previous torch stream back.  Call before the context is destroyed:
        the library's stream dies with it, and torch must not be left pointing at it."""
        if self._restore is not None:
            torch.cuda.synchronize(self.e.device)
            torch.cuda.set_stream(self._restore)
            self._restore = None

    def _allgather(self, which, nbytes):
        if not dist.is_initialized():
            return 0                                      # single rank without a process group: nothing to move
        views = self._views.get((which, nbytes))
        if views is None:                                 # a handful of distinct sizes per run
            buf = self.e.dist if which == self.e.BUF_DIST else self.e.gather
            n = nbytes // 4
            full = buf[: n * self.world]
            mine = full[self.rank * n:(self.rank + 1) * n]
            views = self._views[(which, nbytes)] = (full, mine, None if self._inplace else torch.empty_like(mine))
        full, mine, send = views
        if send is None:
            dist.all_gather_into_tensor(full, mine, group=self.group)
        else:
            send.copy_(mine)
            dist.all_gather_into_tensor(full, send, group=self.group)
        return 0

    def step(self):
        self.e.step()

    def steps(self, n):
        self.e.steps(n)
