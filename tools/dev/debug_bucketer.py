"""Developer check (GPU box): which parameter notifies the bucketed all-reduce twice in one step, and from where."""
import os
import socket
import sys
import traceback

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from hfa_gp_amd.synthetic import make_frame_set  # noqa: E402
from hfa_gp_amd.trainer import BucketedAllReduce, FlatGrads, Trainer  # noqa: E402
from tests.test_gpu_round2 import FitArgs  # noqa: E402

dev = torch.device("cuda:0")
s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
dist.init_process_group("gloo", rank=0, world_size=1)
torch.manual_seed(0)
tr = Trainer(FitArgs(), dev, mode="3dmm", lpips="none")
tr.optimizer = torch.optim.SGD(tr.gen.parameters(), lr=0.0)
tr.tune_generator()
tr.force_collective = True
tr._flat = FlatGrads(tr.shared_parameters(), bucket_bytes=64 << 10)
tr._bucketer = None
names = {id(p): n for n, p in tr.gen.named_parameters()}
log = []
orig = BucketedAllReduce.on_grad


def on_grad(self, p):
    src = "sink" if any("sink" in f.name for f in traceback.extract_stack()[-4:]) else "hook"
    b = self.bucket_of.get(id(p))
    launched = b is not None and self.works[b] is not None
    log.append((names.get(id(p)), src, b, launched, id(p) in self.seen))
    if launched:
        return
    return orig(self, p)


BucketedAllReduce.on_grad = on_grad
data = make_frame_set(tr.gen, 2, size=FitArgs.size, seed=42, params_len=76)
tr.gen_update(data["real"], data["label"].clone(), data["params"])
seen = {}
for n, src, b, launched, was_seen in log:
    seen.setdefault(n, []).append((src, b, launched, was_seen))
for n, ev in seen.items():
    if len(ev) > 1 or ev[0][2]:
        print(n, ev)
print("events", len(log), "params", len(seen))
dist.destroy_process_group()
