"""Multi-GPU glue: one process (and one ``Problem``) per GPU, correspondences sharded contiguously,
ONE exchange of the 24-double normal-equation pack per sweep.  The payload is 192 bytes,
so the exchange is latency-bound; xGMI bandwidth is irrelevant, the number of launches and hops is what counts.

Transports for that exchange (``attach(..., transport=...)``):
  * "peer"   -- direct peer exchange: every rank's 4 KiB inbox is mapped into all peers with HIP IPC; one wave per
                rank stores its pack into every inbox over xGMI, polls its own and sums in rank order (bit-identical on
                all ranks), then publishes to the host.  No collective library on the data path.
  * "rccl"   -- the C++ shim calls ``ncclAllReduce`` (RCCL) itself on the problem's stream; the 128-byte unique id is
                shipped between ranks with ``torch.distributed``.
  * "hook"   -- ``torch.distributed.all_reduce`` on a tensor aliasing the shim's device pack, issued with the problem's
                stream made torch's current stream (so it is ordered against the shim's kernels whichever stream the
                problem was created on).
  * "auto"   -- "rccl" (the collective BASELINE's north star names) if every rank can load RCCL and joins the
                communicator, else "hook".  "peer" is opt-in (``transport="peer"`` / ``SBA_TRANSPORT=peer``): its stores
                into another DEVICE's IPC-mapped inbox have so far only run between processes on one GPU, so it is not
                a default until it has one cross-device run behind it.
``torch.distributed`` is host plumbing here (shipping handles / ids, agreeing on the fallback), never the data path
of "peer" and "rccl".  Every rank issues the same sequence of control collectives whatever fails locally, and all ranks
end on the same transport.
"""
from __future__ import annotations

import os

from . import api


class _DevicePack:
    """Exposes the shim's device pack through __cuda_array_interface__ so torch can alias it."""

    def __init__(self, ptr: int, n: int):
        self.__cuda_array_interface__ = {"shape": (n,), "typestr": "<f8", "data": (ptr, False), "version": 3}


def _ctrl_device(torch, dist):
    """Device for the few control messages: the process group's own (cuda for nccl, cpu for gloo)."""
    return torch.device("cuda", torch.cuda.current_device()) if dist.get_backend() == "nccl" else torch.device("cpu")


def _all_agree(torch, dist, ok: bool) -> bool:
    t = torch.tensor([1 if ok else 0], dtype=torch.int32, device=_ctrl_device(torch, dist))
    dist.all_reduce(t, op=dist.ReduceOp.MIN)
    return bool(int(t.item()) == 1)


def _try_peer(problem: api.Problem, torch, dist) -> bool:
    world, rank = dist.get_world_size(), dist.get_rank()
    dev = _ctrl_device(torch, dist)
    handle, ok = bytes(64), world <= 8
    if ok:
        try:
            handle = problem.peer_export(world, rank)
        except api.SbaError:
            ok = False
    mine = torch.tensor(list(handle), dtype=torch.uint8, device=dev)
    gathered = [torch.zeros(64, dtype=torch.uint8, device=dev) for _ in range(world)]
    dist.all_gather(gathered, mine)
    if _all_agree(torch, dist, ok):
        try:
            problem.peer_connect(b"".join(bytes(g.cpu().tolist()) for g in gathered))
        except api.SbaError:
            ok = False
    else:
        ok = False
    if _all_agree(torch, dist, ok):
        try:
            ok = problem.peer_selftest(8)
        except api.SbaError:
            ok = False
    else:
        ok = False
    if _all_agree(torch, dist, ok):
        return True
    problem.peer_disable()
    return False


def attach_peer_local_ranks(problems, ranks, world: int, selftest_rounds: int = 8) -> bool:
    """The direct peer exchange for a process that hosts SEVERAL ranks -- one ``Problem`` (and, while sweeping, one host
    thread) per rank: a host that drives all GPUs of a node from one process, or a rehearsal of the node's full width
    of 8 ranks inside the few processes a one-GPU box admits.  ``ranks[i]`` is the global rank of ``problems[i]``; every
    process of the torch.distributed group must host the same number of ranks, ``world`` in all.  Ranks of one process
    reach each other's inbox by device pointer (the shim recognises its own handles), all others through HIP IPC.
    Control collectives are issued from the calling thread only; the self-test exchanges run one thread per rank (an
    exchange blocks until every rank of the world has arrived).  Returns True when every rank of the world is
    connected and passed the self-test; otherwise every problem is detached again and False is returned."""
    import threading

    import torch
    import torch.distributed as dist

    ok = world <= 8 and len(problems) == len(ranks) and len(ranks) * dist.get_world_size() == world
    mine = []
    if ok:
        try:
            mine = [(int(r), p.peer_export(world, int(r))) for p, r in zip(problems, ranks)]
        except api.SbaError:
            ok = False
    gathered = [None] * dist.get_world_size()
    dist.all_gather_object(gathered, mine)
    table = dict(kv for part in gathered for kv in part)
    ok = ok and sorted(table) == list(range(world))
    if _all_agree(torch, dist, ok):
        blob = b"".join(table[r] for r in range(world))
        try:
            for p in problems:
                p.peer_connect(blob)
        except api.SbaError:
            ok = False
    else:
        ok = False
    if _all_agree(torch, dist, ok):
        passed = [False] * len(problems)

        def selftest(i):
            try:
                passed[i] = bool(problems[i].peer_selftest(selftest_rounds))
            except api.SbaError:
                passed[i] = False
        threads = [threading.Thread(target=selftest, args=(i,)) for i in range(len(problems))]
        [t.start() for t in threads]
        [t.join() for t in threads]
        ok = all(passed)
    else:
        ok = False
    if _all_agree(torch, dist, ok):
        return True
    for p in problems:
        p.peer_disable()
    return False


def _try_rccl(problem: api.Problem, torch, dist) -> bool:
    world, rank = dist.get_world_size(), dist.get_rank()
    # ncclCommInitRank is itself a collective: a rank that cannot even load librccl would leave the others blocked
    # inside it.  So: agree that everybody CAN call it before anybody does.
    try:
        can = bool(api.rccl_available())
    except Exception:
        can = False
    if not _all_agree(torch, dist, can):
        return False
    uid, ok = bytes(128), 1
    if rank == 0:
        try:
            uid = api.comm_unique_id()
        except api.SbaError:
            ok = 0
    t = torch.tensor([ok] + list(uid), dtype=torch.int32, device=_ctrl_device(torch, dist))
    dist.broadcast(t, 0)
    vals = t.cpu().tolist()
    if vals[0] != 1:
        return False
    joined = True
    try:
        problem.comm_init_rank(world, rank, bytes(vals[1:]))
    except api.SbaError:
        joined = False
    # ... and agree afterwards: if any rank came back with an error, every rank drops its communicator and all fall
    # back together instead of some all-reducing over RCCL and others not.
    if _all_agree(torch, dist, joined):
        return True
    try:
        problem.comm_destroy()
    except api.SbaError:
        pass
    return False


def attach(problem: api.Problem, transport: str = "auto", force: bool = False, prefer_native: bool | None = None) -> str:
    """Install the per-sweep exchange on `problem` for the current torch.distributed world.
    Returns the transport used: "none" (world size 1, unless `force`), "xgmi-peer", "rccl-native" or "torch-hook".
    `transport`: "auto" (rccl, else hook) | "peer" | "rccl" | "hook" (env SBA_TRANSPORT overrides "auto")."""
    import torch
    import torch.distributed as dist

    if prefer_native is not None:            # older call sites: True -> rccl, False -> hook
        transport = "rccl" if prefer_native else "hook"
    if transport == "auto":
        transport = os.environ.get("SBA_TRANSPORT", "auto")
    if dist.get_world_size() == 1 and not force:
        return "none"
    if transport == "peer":
        if _try_peer(problem, torch, dist):
            return "xgmi-peer"
        raise RuntimeError("direct peer exchange could not be set up on every rank")
    if transport in ("auto", "rccl"):
        if _try_rccl(problem, torch, dist):
            return "rccl-native"
        if transport == "rccl":
            raise RuntimeError("RCCL communicator could not be created on every rank")
    return _install_hook(problem, torch, dist)


def _install_hook(problem: api.Problem, torch, dist) -> str:
    dev = torch.device("cuda", torch.cuda.current_device())
    pack_ptr = problem.pack_device_ptr
    pack = torch.as_tensor(_DevicePack(pack_ptr, 24), device=dev)
    aliases = {(pack_ptr, 24): pack}

    streams = {}

    def hook(ptr, count, stream):
        # the 24-double pack on every sweep; other device buffers (the 64 x 45 group moments of the initial guess)
        # get their own alias the first time they appear
        t = aliases.get((ptr, count))
        if t is None:
            t = aliases[(ptr, count)] = torch.as_tensor(_DevicePack(ptr, count), device=dev)
            if len(aliases) > 64:          # scratch buffers come and go: keep the table small
                for k in [k for k in aliases if k != (pack_ptr, 24) and k != (ptr, count)]:
                    del aliases[k]
        # The collective must be ordered after the kernels the shim has just enqueued on ITS stream and before the
        # ones that follow: torch orders a collective against the current stream, so make the shim's stream current
        # (a problem created on torch's own current stream needs nothing).
        if stream and stream != torch.cuda.current_stream(dev).cuda_stream:
            ext = streams.get(stream)
            if ext is None:
                ext = streams[stream] = torch.cuda.ExternalStream(stream, device=dev)
            with torch.cuda.stream(ext):
                dist.all_reduce(t)
        else:
            dist.all_reduce(t)
        return 0
    problem.set_allreduce(hook)
    problem.set_shard(dist.get_rank(), dist.get_world_size())
    problem._torch_pack_alias = pack
    return "torch-hook"
