#!/usr/bin/env python3
"""Per-step cost of the all-reduce path on ONE GPU (1-rank RCCL communicator / torch hook): a floor for the
multi-GPU step overhead (kernel launch + protocol of the collective, publish kernel), measured with host-synchronous
steps driven from C++ (sba_problem_eval_steps)."""
import os, socket, sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np  # noqa: E402
from spherical_bundle_adjuster_amd import api, synthetic  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
c = synthetic.full_rt(n)
def run(p, label):
    p.eval_steps(api.MODE_RT, c.rot_init, c.tran_init, depth_mode=api.DEPTH_PER_MATCH, steps=10)
    best = 1e9
    for _ in range(3):
        _, sec = p.eval_steps(api.MODE_RT, c.rot_init, c.tran_init, depth_mode=api.DEPTH_PER_MATCH, steps=100)
        best = min(best, sec / 100)
    print(f"{label:28s} {best*1e6:8.1f} us per step", flush=True)
with api.Problem(0) as p:
    p.upload(c.x1, c.x2, c.d12)
    run(p, "no collective")
    p.comm_init_rank(1, 0, api.comm_unique_id())
    run(p, "1-rank ncclAllReduce")
import torch, torch.distributed as dist
from spherical_bundle_adjuster_amd import distributed
with socket.socket() as s:
    s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
with api.Problem(0, stream=torch.cuda.current_stream().cuda_stream) as p:
    p.upload(c.x1, c.x2, c.d12)
    distributed.attach(p, prefer_native=False, force=True)
    run(p, "torch.distributed hook")
dist.destroy_process_group()
