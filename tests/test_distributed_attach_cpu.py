"""CPU: the transport agreement of distributed.attach() with a 2-rank gloo group and a MOCK problem.  Every rank must
issue the same sequence of control collectives whatever fails locally (a mismatch would deadlock a multi-GPU job), and
all ranks must end on the same transport: peer -> rccl -> hook."""
import os
import socket
import sys

import pytest

from helpers import ROOT

SCENARIOS = {
    # name: (failure injected on which rank / where, expected transport on every rank)
    "all_ok": (None, "xgmi-peer"),
    "export_fails_rank1": ((1, "export"), "rccl-native"),
    "connect_fails_rank0": ((0, "connect"), "rccl-native"),
    "selftest_false_rank1": ((1, "selftest"), "rccl-native"),
    "selftest_raises_rank0": ((0, "selftest_raise"), "rccl-native"),
    "peer_and_uid_fail": ((0, "export+uid"), "torch-hook"),
    "forced_rccl": (None, "rccl-native"),
}


def _worker(rank, world, port, scenario, q):
    sys.path.insert(0, str(ROOT))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import torch.distributed as dist
    from spherical_bundle_adjuster_amd import api, distributed

    fail, _ = SCENARIOS[scenario]
    where = fail[1] if fail and fail[0] == rank else ""

    class MockProblem:
        def __init__(self):
            self.calls = []

        def peer_export(self, nranks, rank_):
            self.calls.append("export")
            if "export" in where:
                raise api.SbaError(-5, "injected")
            return bytes([rank_ + 1] * 64)

        def peer_connect(self, handles):
            self.calls.append("connect")
            assert handles == bytes([1] * 64) + bytes([2] * 64)         # all-gathered in rank order
            if where == "connect":
                raise api.SbaError(-5, "injected")

        def peer_selftest(self, rounds):
            self.calls.append("selftest")
            if where == "selftest_raise":
                raise api.SbaError(-5, "injected")
            return where != "selftest"

        def peer_disable(self):
            self.calls.append("disable")

        def comm_init_rank(self, nranks, rank_, uid):
            self.calls.append("comm_init")
            assert uid == bytes(range(128))

    def fake_uid():
        if "uid" in where:
            raise api.SbaError(-5, "injected")
        return bytes(range(128))

    api.comm_unique_id = fake_uid
    distributed._install_hook = lambda problem, torch, dist_: (problem.calls.append("hook") or "torch-hook")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        p = MockProblem()
        used = distributed.attach(p, transport="rccl" if scenario == "forced_rccl" else "auto")
        q.put((rank, used, p.calls))
    except Exception:
        import traceback
        q.put((rank, "ERROR", traceback.format_exc()))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("scenario", sorted(SCENARIOS))
def test_attach_agreement(scenario):
    import torch.multiprocessing as mp
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, scenario, q)) for r in range(2)]
    [p.start() for p in procs]
    res = sorted([q.get(timeout=120) for _ in procs])        # a deadlock shows up as a timeout here
    [p.join(30) for p in procs]
    expected = SCENARIOS[scenario][1]
    for rank, used, calls in res:
        assert used == expected, calls
        if expected != "xgmi-peer" and scenario != "forced_rccl":
            assert calls.count("disable") == 1                 # every rank tears its peer state down
        if expected == "xgmi-peer":
            assert calls == ["export", "connect", "selftest"]
    assert res[0][1] == res[1][1]
