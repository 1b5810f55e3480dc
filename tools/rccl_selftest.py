"""Developer check: ShardedTrainer over a one-rank RCCL process group (a GPU box of this pool has
one GPU).  Exercises the real torch.distributed all_gather_into_tensor issued from inside the
library's callback on the library's own stream; the result must match a plain context."""
import importlib
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from util import relmax, synth  # noqa: E402

os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29533")
os.environ["DLCO_FORCE_SHARD"] = "1"
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
dlco = importlib.import_module("opencv-dlco_amd")
ddist = importlib.import_module("opencv-dlco_amd.dist")
N, F, B = 3000, 1024, 40
D, L = synth(N, F, k=16, seed=61)
a = dlco.Context(F, N, B=B, mu=0.004)
b = dlco.Context(F, N, B=B, mu=0.004, shard=1)
a.set_data(D, L)
b.set_data(D, L)
tr = ddist.ShardedTrainer(ddist.HipShardEngine(dlco, b, torch.device("cuda", 0)), native=(os.environ.get("DLCO_NATIVE_RCCL", "1") != "0"))
print("native RCCL communicator:", tr.native)
a.steps(20)
tr.steps(20)
b.sync()
print("dfavg relmax", relmax(b.dfavg(), a.dfavg()), "A relmax", relmax(b.A(), a.A()), "rank", a.W().shape[0], b.W().shape[0])
assert relmax(b.dfavg(), a.dfavg()) <= 5e-6 and relmax(b.A(), a.A()) <= 5e-4
tr.close()
dist.destroy_process_group()
print("rccl selftest ok")
