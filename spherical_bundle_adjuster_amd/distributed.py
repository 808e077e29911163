"""Multi-GPU glue: one process (and one ``Problem``) per GPU, correspondences sharded contiguously,
ONE all-reduce of the 24-double normal-equation pack per sweep.

Two transports for that all-reduce:
  * native  -- the C++ shim calls ``ncclAllReduce`` (RCCL) itself on the problem's stream; the
               128-byte unique id is shipped between ranks with ``torch.distributed`` (host plumbing).
  * hook    -- ``torch.distributed.all_reduce`` on a tensor aliasing the shim's device pack (needs
               the problem to launch on torch's current stream).
The payload is 192 bytes, so either way the exchange is latency-bound; xGMI bandwidth is irrelevant.
"""
from __future__ import annotations

from . import api


class _DevicePack:
    """Exposes the shim's device pack through __cuda_array_interface__ so torch can alias it."""

    def __init__(self, ptr: int, n: int):
        self.__cuda_array_interface__ = {"shape": (n,), "typestr": "<f8", "data": (ptr, False), "version": 3}


def attach(problem: api.Problem, prefer_native: bool = True, force: bool = False) -> str:
    """Install the per-sweep all-reduce on `problem` for the current torch.distributed world.
    Returns the transport used: "none" (world size 1, unless `force`), "rccl-native" or "torch-hook".
    The torch-hook transport requires the problem to have been created on torch's current stream."""
    import torch
    import torch.distributed as dist

    world, rank = dist.get_world_size(), dist.get_rank()
    if world == 1 and not force:
        return "none"
    dev = torch.device("cuda", torch.cuda.current_device())
    if prefer_native:
        uid, ok = bytes(128), 1
        if rank == 0:
            try:
                uid = api.comm_unique_id()
            except api.SbaError:
                ok = 0
        t = torch.tensor([ok] + list(uid), dtype=torch.int32, device=dev)
        dist.broadcast(t, 0)
        vals = t.cpu().tolist()
        if vals[0] == 1:
            problem.comm_init_rank(world, rank, bytes(vals[1:]))
            return "rccl-native"
    pack = torch.as_tensor(_DevicePack(problem.pack_device_ptr, 24), device=dev)

    def hook(_ptr, _count, _stream):
        dist.all_reduce(pack)
        return 0
    problem.set_allreduce(hook)
    problem._torch_pack_alias = pack
    return "torch-hook"
