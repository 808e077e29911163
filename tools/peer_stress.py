#!/usr/bin/env python3
"""Stress of the direct peer exchange: W processes on one GPU, R rounds of the self-test (known sums checked every
round), then R sharded sweeps whose packs must stay bit-identical across ranks.  python tools/peer_stress.py [W] [R]"""
import os, socket, sys, time
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))


def worker(rank, world, port, rounds, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import numpy as np, torch, torch.distributed as dist
    from spherical_bundle_adjuster_amd import api, distributed, synthetic
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    c = synthetic.full_rt(40000, seed=5)
    lo, hi = synthetic.shard_range(40000, rank, world)
    with api.Problem(0) as p:
        p.upload(c.x1[lo:hi], c.x2[lo:hi], c.d12[lo:hi])
        used = distributed.attach(p, transport="peer")
        t0 = time.perf_counter(); ok = p.peer_selftest(rounds); t1 = time.perf_counter()
        first, same = None, True
        pk, sec = p.eval_steps(api.MODE_RT, c.rot_init, c.tran_init, depth_mode=api.DEPTH_PER_MATCH, steps=rounds)
        for k in range(50):
            pk2 = p.eval_pack(api.MODE_RT, c.rot_init, c.tran_init, depth_mode=api.DEPTH_PER_MATCH)
            same &= bool(np.array_equal(pk, pk2))
        dist.barrier()
        p.peer_disable()
    q.put((rank, used, ok, (t1 - t0) / rounds * 1e6, sec / rounds * 1e6, same, pk.tobytes()))
    dist.destroy_process_group()


if __name__ == "__main__":
    import torch.multiprocessing as mp
    world = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 20000
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]
    ctx = mp.get_context("spawn"); q = ctx.Queue()
    procs = [ctx.Process(target=worker, args=(r, world, port, rounds, q)) for r in range(world)]
    [p.start() for p in procs]
    res = sorted(q.get(timeout=900) for _ in procs)
    [p.join(60) for p in procs]
    for r in res:
        print(f"rank {r[0]}: {r[1]} selftest_ok={r[2]} {r[3]:.1f} us/exchange-round {r[4]:.1f} us/sharded-step repeat_identical={r[5]}")
    print("packs bit-identical across ranks:", all(r[6] == res[0][6] for r in res))
