"""CPU: the transport agreement of distributed.attach() with a 2-rank gloo group and a MOCK problem.  Every rank must
issue the same sequence of control collectives whatever fails locally (a mismatch would deadlock a multi-GPU job), and
all ranks must end on the same transport: auto = rccl -> hook; peer only on request."""
import os
import socket
import sys

import pytest

from helpers import ROOT

SCENARIOS = {
    # name: (transport asked for, failure injected on which rank / where, expected transport on every rank or "RAISES")
    "auto_all_ok": ("auto", None, "rccl-native"),
    "auto_rccl_missing_rank1": ("auto", (1, "norccl"), "torch-hook"),        # nobody may enter ncclCommInitRank
    "auto_uid_fails": ("auto", (0, "uid"), "torch-hook"),
    "auto_comm_init_fails_rank1": ("auto", (1, "comm_init"), "torch-hook"),  # rank 0 joined: it must drop its communicator
    "auto_comm_init_fails_rank0": ("auto", (0, "comm_init"), "torch-hook"),
    "forced_rccl": ("rccl", None, "rccl-native"),
    "forced_rccl_comm_init_fails_rank1": ("rccl", (1, "comm_init"), "RAISES"),
    "peer_all_ok": ("peer", None, "xgmi-peer"),
    "peer_export_fails_rank1": ("peer", (1, "export"), "RAISES"),
    "peer_connect_fails_rank0": ("peer", (0, "connect"), "RAISES"),
    "peer_selftest_false_rank1": ("peer", (1, "selftest"), "RAISES"),
    "peer_selftest_raises_rank0": ("peer", (0, "selftest_raise"), "RAISES"),
}


def _worker(rank, world, port, scenario, q):
    sys.path.insert(0, str(ROOT))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import torch.distributed as dist
    from spherical_bundle_adjuster_amd import api, distributed

    asked, fail, _ = SCENARIOS[scenario]
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
            if where == "comm_init":
                raise api.SbaError(-5, "injected")

        def comm_destroy(self):
            self.calls.append("comm_destroy")

    def fake_uid():
        if "uid" in where:
            raise api.SbaError(-5, "injected")
        return bytes(range(128))

    api.comm_unique_id = fake_uid
    api.rccl_available = lambda: where != "norccl"
    distributed._install_hook = lambda problem, torch, dist_: (problem.calls.append("hook") or "torch-hook")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        p = MockProblem()
        try:
            used = distributed.attach(p, transport=asked)
        except RuntimeError:
            used = "RAISES"
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
    expected = SCENARIOS[scenario][2]
    for rank, used, calls in res:
        assert used == expected, calls
        if scenario.startswith("peer") and expected == "RAISES":
            assert calls.count("disable") == 1                 # every rank tears its peer state down
        if expected == "xgmi-peer":
            assert calls == ["export", "connect", "selftest"]
        if "rccl_missing" in scenario or "uid_fails" in scenario:
            assert "comm_init" not in calls                    # nobody entered the (collective) communicator set-up
        if "comm_init_fails" in scenario:
            assert calls.count("comm_init") == 1 and calls.count("comm_destroy") == 1   # joined ranks drop theirs too
        if expected == "torch-hook":
            assert calls[-1] == "hook"
    assert res[0][1] == res[1][1]


# ---- bench.py's second measurement over the peer exchange (N > 1): same rule -- no rank may be left in a collective ----
TRIAL_SCENARIOS = {
    # name: (failure injected on which rank / where, expected (ok, failed_phase))
    "trial_all_ok": (None, (True, None)),
    "trial_detach_fails_rank1": ((1, "detach"), (False, "detach")),
    "trial_export_fails_rank0": ((0, "export"), (False, "attach")),
    "trial_selftest_false_rank1": ((1, "selftest"), (False, "attach")),
    "trial_warmup_raises_rank1": ((1, "warmup"), (False, "warmup")),
    "trial_steps_raise_rank0": ((0, "steps"), (False, "steps")),
}


def _trial_worker(rank, world, port, scenario, q):
    sys.path.insert(0, str(ROOT))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import numpy as np
    import torch
    import torch.distributed as dist
    from spherical_bundle_adjuster_amd import api
    import bench

    fail, _ = TRIAL_SCENARIOS[scenario]
    where = fail[1] if fail and fail[0] == rank else ""

    class MockProblem:
        calls = []

        def comm_destroy(self):
            self.calls.append("comm_destroy")
            if where == "detach":
                raise api.SbaError(-5, "injected")

        def peer_export(self, nranks, rank_):
            self.calls.append("export")
            if where == "export":
                raise api.SbaError(-5, "injected")
            return bytes([rank_ + 1] * 64)

        def peer_connect(self, handles):
            self.calls.append("connect")

        def peer_selftest(self, rounds):
            self.calls.append("selftest")
            return where != "selftest"

        def peer_disable(self):
            self.calls.append("disable")

    ref = np.arange(24, dtype=np.float64)
    state = {"calls": 0}

    def run_steps(k):
        state["calls"] += 1
        if (where == "warmup" and k is not None) or (where == "steps" and k is None):
            raise api.SbaError(-5, "injected")
        return ref * (1.0 + 1e-16), (7 if k is None else k)

    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        res = bench.peer_trial(MockProblem(), "rccl-native", dist, torch, run_steps, dist.barrier, ref, world, 1000)
        q.put((rank, res, MockProblem.calls))
    except Exception:
        import traceback
        q.put((rank, "ERROR", traceback.format_exc()))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("scenario", sorted(TRIAL_SCENARIOS))
def test_bench_peer_trial_agreement(scenario):
    import torch.multiprocessing as mp
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_trial_worker, args=(r, 2, port, scenario, q)) for r in range(2)]
    [p.start() for p in procs]
    res = sorted([q.get(timeout=120) for _ in procs], key=lambda x: x[0])   # a deadlock shows up as a timeout here
    [p.join(30) for p in procs]
    ok, phase = TRIAL_SCENARIOS[scenario][1]
    for rank, r, calls in res:
        assert r != "ERROR", calls
        assert r["ok"] is ok, (r, calls)
        assert r.get("failed_phase") == phase, r
        if ok:
            assert r["ms_per_step"] > 0 and r["max_rel_diff_to_quoted_pack"] < 1e-12
            assert abs(r["value"] - 1000 * 2 * 7 / (r["ms_per_step"] * 1e-3 * 7)) <= 1e-6 * r["value"]
    assert res[0][1]["ok"] == res[1][1]["ok"]
