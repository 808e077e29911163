"""GPU (-m gpu): the direct peer exchange with SEVERAL PROCESSES ON ONE GPU (each rank opens the others' inboxes through
HIP IPC; on a multi-GPU node the same stores travel over xGMI).  Control messages go through a gloo process group (an
NCCL group cannot place several ranks on one device).  Checks: set-up + self-test, sharded sweeps == single-process
sweep, every rank obtains the bit-identical pack, the LM runs in lock-step to the single-process answer."""
import os
import socket
import sys

import numpy as np
import pytest

from helpers import REL_TOL_F64, ROOT
from spherical_bundle_adjuster_amd import api, synthetic

pytestmark = pytest.mark.gpu

N = 60001
LS_N = 500


def _worker(proc, nproc, local, port, q):
    """One process hosting `local` ranks (global ranks proc * local ... + local - 1), one Problem and one thread each.
    local = 1 is the production shape (one process per rank, distributed.attach); local = 2 puts the node's full width of
    8 ranks into the 4 processes a one-GPU box admits (at most 6 may use the card at once)."""
    sys.path.insert(0, str(ROOT))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import threading
    import traceback
    import torch
    import torch.distributed as dist
    import numpy as np
    from spherical_bundle_adjuster_amd import api as A, distributed, synthetic as syn
    dist.init_process_group("gloo", rank=proc, world_size=nproc)
    world = nproc * local
    ranks = [proc * local + i for i in range(local)]
    out = [None] * local

    def attach(problems):
        if local == 1:
            return distributed.attach(problems[0], transport="peer")
        return "xgmi-peer" if distributed.attach_peer_local_ranks(problems, ranks, world) else "FAILED"

    def in_threads(fn):
        errs = []

        def run(i):
            try:
                fn(i)
            except Exception:
                errs.append(traceback.format_exc())
        ts = [threading.Thread(target=run, args=(i,)) for i in range(local)]
        [t.start() for t in ts]
        [t.join() for t in ts]
        if errs:
            raise RuntimeError(errs[0])
    try:
        torch.cuda.set_device(0)
        c = syn.full_rt(N, seed=77)
        spans = [syn.shard_range(N, r, world) for r in ranks]
        problems = [A.Problem(0) for _ in ranks]
        for p, (lo, hi) in zip(problems, spans):
            p.upload(c.x1[lo:hi], c.x2[lo:hi], c.d12[lo:hi])
        used = attach(problems)
        res1 = [None] * local

        def body1(i):
            p = problems[i]
            packs = [p.eval_pack(A.MODE_RT, c.rot_init, c.tran_init, depth_mode=A.DEPTH_PER_MATCH) for _ in range(5)]
            packs.append(p.eval_steps(A.MODE_RT, c.rot_init, c.tran_init, depth_mode=A.DEPTH_PER_MATCH, steps=20)[0])
            r, t, s = p.solve(A.MODE_RT, c.rot_init, c.tran_init, depth_mode=A.DEPTH_PER_MATCH)
            tr, _, _ = p.solve(A.MODE_ROT, c.rot_init, c.tran_init, 1.1, 0.9)
            # d-only stage on the sharded problem: nine global reductions exchanged per pass (the two max-norms in pack
            # slots 8 + rank and 16 + rank: rank 7 reaches slots 15 and 23), identical step logic on every rank
            d, sd = p.solve_depths(c.rot_true, c.tran_true)
            # 8-point initial guess on the sharded problem: 64 x 45 group moments all-reduced (120 back-to-back
            # exchanges), same guess on every rank
            gm = p.epipolar_moments()
            eul, tg, ncand = p.initial_guess(80, 0.25, 5)
            res1[i] = (packs, r, t, s.num_iterations, tr, d, sd, (gm, eul, tg, ncand))
        in_threads(body1)
        dist.barrier()
        for p in problems:
            p.peer_disable()
            p.close()
        # ... and a small problem whose first full step fails Armijo (500 matches, start d = 1: one contraction to
        # a = 0.49): the line search's extra passes are sharded and all-reduced like every other pass
        c2 = syn.full_rt(LS_N, seed=6)
        spans2 = [syn.shard_range(LS_N, r, world) for r in ranks]
        problems2 = [A.Problem(0) for _ in ranks]
        for p, (lo2, hi2) in zip(problems2, spans2):
            p.upload(c2.x1[lo2:hi2], c2.x2[lo2:hi2], np.ones((hi2 - lo2, 2)))
        used2 = attach(problems2)
        res2 = [None] * local

        def body2(i):
            res2[i] = problems2[i].solve_depths(c2.rot_init, c2.tran_init)
        in_threads(body2)
        dist.barrier()
        for p in problems2:
            p.peer_disable()
            p.close()
        for i, rank in enumerate(ranks):
            packs, r, t, iters, tr, d, sd, guess = res1[i]
            d_ls, sd_ls = res2[i]
            (lo, hi), (lo2, hi2) = spans[i], spans2[i]
            q.put((rank, used if used == used2 else "FAILED", packs, r, t, iters, tr,
                   (lo, hi, d, sd.num_iterations, sd.termination, sd.final_cost, d_ls, sd_ls.num_iterations,
                    sd_ls.num_line_search_steps, sd_ls.final_cost, lo2, hi2), guess))
    except Exception:      # surface the failure in the parent instead of a silent timeout
        for rank in ranks:
            q.put((rank, "ERROR", traceback.format_exc(), None, None, None, None, None, None))
    finally:
        dist.destroy_process_group()


def _shard_moments(c, world):
    """Group moments of every shard, each computed by a plain single-process problem."""
    for rank in range(world):
        lo, hi = synthetic.shard_range(N, rank, world)
        with api.Problem(0) as p:
            p.upload(c.x1[lo:hi], c.x2[lo:hi], c.d12[lo:hi])
            yield p.epipolar_moments()


@pytest.mark.parametrize("nproc,local", [(2, 1), (4, 1), (4, 2)])
def test_peer_exchange_processes_on_one_gpu(nproc, local):
    """world = nproc x local ranks on ONE GPU.  (4, 2) is the node's real width: 8 ranks -- kMaxPeers reached exactly,
    the sharded d-only stage's max-norm slots 8 + 7 and 16 + 7, the 120-exchange moment all-reduce against 8 pollers --
    in 4 processes of 2 ranks each, because a box of this pool admits at most 6 processes on its card."""
    import torch.multiprocessing as mp
    world = nproc * local
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, nproc, local, port, q)) for r in range(nproc)]
    [p.start() for p in procs]
    res = sorted([q.get(timeout=240) for _ in range(world)], key=lambda x: x[0])
    [p.join(60) for p in procs]
    assert [r[0] for r in res] == list(range(world))
    for r in res:
        assert r[1] == "xgmi-peer", r[2] if r[1] == "ERROR" else r[1]
    c = synthetic.full_rt(N, seed=77)
    with api.Problem(0) as p:
        p.upload(c.x1, c.x2, c.d12)
        single = p.eval_pack(api.MODE_RT, c.rot_init, c.tran_init, depth_mode=api.DEPTH_PER_MATCH)
        r1, t1, s1 = p.solve(api.MODE_RT, c.rot_init, c.tran_init, depth_mode=api.DEPTH_PER_MATCH)
        tr1, _, _ = p.solve(api.MODE_ROT, c.rot_init, c.tran_init, 1.1, 0.9)
        d1, sd1 = p.solve_depths(c.rot_true, c.tran_true)
    c2 = synthetic.full_rt(LS_N, seed=6)
    with api.Problem(0) as p:
        p.upload(c2.x1, c2.x2, np.ones((LS_N, 2)))
        d1_ls, sd1_ls = p.solve_depths(c2.rot_init, c2.tran_init)
    assert sd1_ls.num_line_search_steps >= 1                 # the case really contracts a step
    total_moments = sum(m for m in _shard_moments(c, world))
    for rank, used, packs, r, t, iters, tr, depth, guess in res:
        for pk in packs:
            assert np.array_equal(pk, res[0][2][0])                         # bit-identical on every rank, every repeat
        assert np.abs(packs[0] - single).max() <= REL_TOL_F64 * np.abs(single).max()
        assert np.array_equal(r, res[0][3]) and np.array_equal(t, res[0][4])   # lock-step LM
        assert iters == s1.num_iterations and np.abs(r - r1).max() <= 1e-11 and np.abs(t - t1).max() <= 1e-11
        assert np.abs(tr - tr1).max() <= 1e-11
        lo, hi, d, d_iters, d_term, d_cost, d_ls, ls_iters, ls_steps, ls_cost, lo2, hi2 = depth
        assert (d_iters, d_term) == (sd1.num_iterations, sd1.termination)
        assert d_cost == res[0][7][5]                                            # the same reduced numbers on every rank
        assert abs(d_cost - sd1.final_cost) <= 1e-12 * sd1.final_cost
        assert np.abs(d - d1[lo:hi]).max() <= 1e-9
        assert (ls_iters, ls_steps) == (sd1_ls.num_iterations, sd1_ls.num_line_search_steps)
        assert ls_cost == res[0][7][9] and abs(ls_cost - sd1_ls.final_cost) <= 1e-10 * sd1_ls.final_cost
        assert np.abs(d_ls - d1_ls[lo2:hi2]).max() <= 1e-7 * max(1.0, np.abs(d1_ls).max())
        gm, eul, tg, ncand = guess
        assert np.array_equal(gm, res[0][8][0]) and np.array_equal(eul, res[0][8][1]) and np.array_equal(tg, res[0][8][2])
        assert np.abs(gm - total_moments).max() <= 1e-12 * np.abs(total_moments).max()
        assert ncand > 0


def test_bench_multi_rank_rehearsal(tmp_path):
    """bench.py exactly as the driver launches it for N = 2 (torch.distributed.run, one process per rank), in its
    one-GPU rehearsal mode: the JSON contract, the transport agreement and the max-over-ranks timing must all work."""
    import json
    import subprocess
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, SBA_BENCH_ONE_GPU="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
           "127.0.0.1", "--master-port", str(port), str(ROOT / "bench.py"), "--gpus", "2", "--steps", "10", "--warmup",
           "2", "--matches", "200000", "--transport", "peer",    # RCCL (the default) cannot place two ranks on ONE device
           "--peer-trial"]                                        # the un-quoted second measurement is opt-in
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=300, env=env, cwd=tmp_path)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1                                    # rank 0 prints ONE JSON line
    b = json.loads(lines[0])
    assert b["n_gpus"] == 2 and b["steps"] == 10 and b["warmup"] == 2 and b["scaling"] == "weak"
    assert b["config"]["allreduce"] == "xgmi-peer" and b["unit"] == "evals/s" and b["higher_is_better"] is True
    assert abs(b["value"] - 2 * 200000 * 10 / (b["ms_per_step"] * 1e-3 * 10)) <= 1e-6 * b["value"]
    assert b["lm"]["termination"].startswith("CONVERGENCE") and "cpu_baseline" not in b
    for key in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert key in b["roofline"]
    assert b["cold"]["value"] > 0 and b["preconditioning"]["launches_before_the_timed_region"] >= 2 + 2 + 10
    # the second, un-quoted measurement over the peer exchange (detach -> attach -> self-test -> K steps) went through,
    # and the quoted figure was on record (stderr) before it started
    t = b["peer_trial"]
    assert t["ok"] and t["ms_per_step"] > 0 and t["max_rel_diff_to_quoted_pack"] <= 1e-12, t
    pre = [ln for ln in r.stderr.splitlines() if ln.startswith("bench.py: quoted measurement before the peer trial: ")]
    assert len(pre) == 1 and json.loads(pre[0].split(": ", 2)[2])["value"] == b["value"]


def test_bench_four_rank_rehearsal_default_flags(tmp_path):
    """The driver's own N = 4 command line (no transport / trial flags) in the one-GPU rehearsal mode, except that the
    transport must be named: RCCL cannot place several ranks on one device.  Four ranks + this process = 5 processes on
    the card (the pool admits 6), so the node's width of 8 cannot be rehearsed as processes -- the 8-rank exchange is
    covered by test_peer_exchange_processes_on_one_gpu[4-2].  No peer trial unless asked for."""
    import json
    import subprocess
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, SBA_BENCH_ONE_GPU="1", SBA_TRANSPORT="peer")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "4", "--master-addr",
           "127.0.0.1", "--master-port", str(port), str(ROOT / "bench.py"), "--gpus", "4", "--steps", "10", "--warmup",
           "2", "--matches", "100000"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=300, env=env, cwd=tmp_path)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    b = json.loads(lines[0])
    assert b["n_gpus"] == 4 and b["config"]["allreduce"] == "xgmi-peer" and b["peer_trial"] is None
    assert abs(b["value"] - 4 * 100000 * 10 / (b["ms_per_step"] * 1e-3 * 10)) <= 1e-6 * b["value"]
    assert b["lm"]["termination"].startswith("CONVERGENCE")


def test_bench_c5_two_rank_rehearsal(tmp_path):
    """bench.py --workload c5 for N = 2 in the one-GPU rehearsal mode: pairs are independent (no collective), every rank
    holds its own pairs, rank 0 prints ONE line with the totals."""
    import json
    import subprocess
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, SBA_BENCH_ONE_GPU="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
           "127.0.0.1", "--master-port", str(port), str(ROOT / "bench.py"), "--gpus", "2", "--steps", "5", "--warmup",
           "1", "--workload", "c5", "--pairs", "256", "--pair-matches", "2000", "--frames", "2", "--precondition-ms", "5"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=300, env=env, cwd=tmp_path)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    b = json.loads(lines[0])
    assert b["n_gpus"] == 2 and b["steps"] == 5 and b["scaling"] == "weak" and b["config"]["pairs_per_gpu"] == 256
    assert abs(b["value"] - 2 * 256 * 2000 * 5 / (b["ms_per_step"] * 1e-3 * 5)) <= 1e-6 * b["value"]
    assert b["lm"]["all_converged"] and "cpu_baseline" not in b and b["equi2cube"]["frames"] == 2
