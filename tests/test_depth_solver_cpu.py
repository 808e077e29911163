"""CPU: the product's d-only step logic (csrc/sba_depth_solver.hpp -- the state machine sba_problem_solve_depths runs
between device passes) driven by an EMULATED device pass.

The emulation (numpy, `EmulatedShard`) does what depth_step_kernel does per match -- step of the damped 2x2 system at the
current depths with the stored Jacobi scaling / LM diagonal, projected candidate P(d + alpha delta), the nine
reductions -- on top of the independently written residual / derivative code of tests/ref_depth_numpy.py.  Checked:
  * single process: same termination, iteration / accepted-step / contraction counts and depths as the oracle;
  * world size 2 over gloo: every rank holds a shard, the seven sums are all-reduced with SUM and the two max-norms with
    MAX, every rank replays the identical logic -- same result as the single process (SURVEY section 8e for this stage).
"""
import ctypes as C
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

import ref_depth_numpy as rd
from helpers import ROOT
from spherical_bundle_adjuster_amd import _cabi as cabi
from spherical_bundle_adjuster_amd import synthetic

TERM = {1: "function", 2: "gradient", 3: "parameter", 4: "no_convergence"}
_h = None


def harness():
    global _h
    if _h is None:
        so = ROOT / "tests" / "harness" / "libdepth_harness.so"
        src = ROOT / "tests" / "harness" / "depth_harness.cpp"
        hdrs = [ROOT / "spherical_bundle_adjuster_amd" / "csrc" / f for f in ("sba_depth_solver.hpp", "sba_line_search.hpp", "sba_lm.hpp")]
        if not so.exists() or so.stat().st_mtime < max(f.stat().st_mtime for f in [src] + hdrs):
            subprocess.run(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-o", str(so), str(src)], check=True)
        _h = C.CDLL(str(so))
        _h.depth_harness_create.restype = C.c_void_p
    return _h


class EmulatedShard:
    """What the device holds for one shard: current depths, candidate, Jacobi scaling, LM diagonal."""

    def __init__(self, x1, x2, rot, tran, d0, lam=1.0, c=1.0, min_diag=1e-6, max_diag=1e32):
        self.P = rd.DepthProblem(x1, x2, rot, tran, lam, c)
        self.d = np.array(d0, dtype=np.float64).reshape(-1, 2).copy()
        self.cand = self.d.copy()
        self.scale = None
        self.diag = None
        self.min_diag, self.max_diag = min_diag, max_diag

    def run_pass(self, alpha, keep_diagonal, first, radius):
        """-> (seven sums, two maxima)"""
        P, d = self.P, self.d
        g = P.gradient(d)
        h11, h12, h22 = P.hessian_blocks(d)
        if first:
            self.scale = np.stack([1.0 / (1.0 + np.sqrt(h11)), 1.0 / (1.0 + np.sqrt(h22))], axis=1)
        s = self.scale
        H11, H12, H22 = s[:, 0] ** 2 * h11, s[:, 0] * s[:, 1] * h12, s[:, 1] ** 2 * h22
        G = s * g
        if not keep_diagonal:
            self.diag = np.stack([np.clip(H11, self.min_diag, self.max_diag), np.clip(H22, self.min_diag, self.max_diag)], axis=1)
        D = self.diag
        A11, A22 = H11 + D[:, 0] / radius, H22 + D[:, 1] / radius
        det = A11 * A22 - H12 * H12
        y = np.stack([(-G[:, 0] * A22 + G[:, 1] * H12) / det, (-G[:, 1] * A11 + G[:, 0] * H12) / det], axis=1)
        delta = s * y
        self.cand = np.maximum(d + alpha * delta, 0.0)
        sums = np.array([P.cost(d),
                         -np.sum(G * y) - 0.5 * np.sum(H11 * y[:, 0] ** 2 + 2 * H12 * y[:, 0] * y[:, 1] + H22 * y[:, 1] ** 2),
                         P.cost(self.cand), np.sum((self.cand - d) ** 2), np.sum(d * d), np.sum(g * delta),
                         np.sum(P.gradient(self.cand) * delta)])
        maxima = np.array([np.abs(d - np.maximum(d - g, 0.0)).max(initial=0.0), np.abs(delta).max(initial=0.0)])
        return sums, maxima

    def take_candidate(self):
        self.d = self.cand.copy()


def drive(shard, reduce_fn=None, **opt):
    """Run the product's state machine over `shard`; reduce_fn(sums, maxima) all-reduces on a sharded problem."""
    h = harness()
    o = cabi.LmOptions()
    h.depth_harness_default_options(C.byref(o))
    for k, v in opt.items():
        setattr(o, k, v)
    s = C.c_void_p(h.depth_harness_create(C.byref(o)))
    rq = (C.c_double * 4)()
    passes = 0
    while not h.depth_harness_done(s):
        h.depth_harness_request(s, rq)
        sums, maxima = shard.run_pass(rq[0], rq[1] != 0.0, rq[2] != 0.0, rq[3])
        if reduce_fn is not None:
            sums, maxima = reduce_fn(sums, maxima)
        out9 = np.ascontiguousarray(np.concatenate([sums, maxima]))
        h.depth_harness_feed(s, out9.ctypes.data_as(C.c_void_p))
        passes += 1
        if h.depth_harness_take_candidate(s):
            shard.take_candidate()
        assert passes < 10000
    summ = cabi.LmSummary()
    h.depth_harness_summary(s, C.byref(summ))
    status = h.depth_harness_status(s)
    h.depth_harness_destroy(s)
    return shard.d, summ, status, passes


CASES = [(500, 6, 1.0, 1.0, 1.0), (400, 9, 0.05, 1.0, 1.0), (64, 23, 0.01, 20.0, 4.0), (300, 12, 2.0, 1.0, 1.0), (1, 3, 3.0, 1.0, 1.0)]


@pytest.mark.parametrize("n,seed,d0,lam,c", CASES)
@pytest.mark.parametrize("ls", [20, 0], ids=["ceres_default", "no_line_search"])
def test_depth_state_machine_matches_oracle(oracle, n, seed, d0, lam, c, ls):
    cs = synthetic.full_rt(n, seed=seed)
    start = np.full((n, 2), d0)
    dref, sref, rc = oracle.depth_solve(cs.x1, cs.x2, cs.rot_init, cs.tran_init, start, lam=lam, c=c,
                                        options=oracle.default_options(max_num_line_search_step_size_iterations=ls))
    assert rc == 0
    d, s, status, passes = drive(EmulatedShard(cs.x1, cs.x2, cs.rot_init, cs.tran_init, start, lam, c),
                                 max_num_line_search_step_size_iterations=ls)
    assert status == 0
    assert (TERM[s.termination], s.num_iterations, s.num_successful_steps, s.num_line_search_steps) == \
        (TERM[sref.termination], sref.num_iterations, sref.num_successful_steps, sref.num_line_search_steps)
    assert np.abs(d - dref).max() <= 1e-7 * max(1.0, np.abs(dref).max())
    assert abs(s.final_cost - sref.final_cost) <= 1e-10 * sref.final_cost
    # a step that passes Armijo at once costs ONE pass; every contraction one more
    assert passes == s.num_evaluations and passes >= s.num_iterations + s.num_line_search_steps


def test_depth_state_machine_limits_and_failures():
    cs = synthetic.full_rt(50, seed=4)
    mk = lambda: EmulatedShard(cs.x1, cs.x2, cs.rot_init, cs.tran_init, np.full((50, 2), 3.0))
    d, s, status, passes = drive(mk(), max_num_iterations=0)
    assert TERM[s.termination] == "no_convergence" and s.num_iterations == 0 and passes == 1
    d, s, status, passes = drive(mk(), max_num_iterations=3)
    assert TERM[s.termination] == "no_convergence" and s.num_iterations == 3
    d, s, status, passes = drive(mk(), gradient_tolerance=1e30)
    assert TERM[s.termination] == "gradient" and s.num_iterations == 0
    d, s, status, passes = drive(mk(), initial_trust_region_radius=1e-40)
    assert s.termination == 5                                     # SBA_TERM_MIN_RADIUS
    nan_shard = mk()
    nan_shard.d[0, 0] = np.nan
    d, s, status, passes = drive(nan_shard)
    assert s.termination == 6 and status == cabi.SBA_ERR_NUMERIC  # non-finite cost: failure, not a spin


def _rank(rank, world, port, q, case):
    sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import torch
    import torch.distributed as dist
    import test_depth_solver_cpu as T
    from spherical_bundle_adjuster_amd import synthetic as syn
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        n, seed, d0, lam, c = case
        cs = syn.full_rt(n, seed=seed)
        lo, hi = syn.shard_range(n, rank, world)
        exchanges = []

        def reduce_fn(sums, maxima):
            # ONE exchange per pass, in the product's own wire format (csrc/sba_depth.hip depth_finalize_kernel with
            # gather_slot = rank, csrc/sba_stages.cpp): a 24-double pack for a SUM all-reduce -- the seven sums in slots
            # 0..6, and the two max-norms, which a SUM cannot carry, as one slot per rank each (8 + rank, 16 + rank; zeros
            # elsewhere), whose maxima every rank takes on the host.  At most 8 ranks: rank 7 fills slots 15 and 23.
            assert world <= 8 and len(sums) == 7 and len(maxima) == 2
            pack = np.zeros(24)
            pack[:7] = sums
            pack[8 + rank], pack[16 + rank] = maxima
            t = torch.from_numpy(pack)
            dist.all_reduce(t, op=dist.ReduceOp.SUM)
            exchanges.append(1)
            return pack[:7].copy(), np.array([pack[8:8 + world].max(), pack[16:16 + world].max()])
        shard = T.EmulatedShard(cs.x1[lo:hi], cs.x2[lo:hi], cs.rot_init, cs.tran_init, np.full((hi - lo, 2), d0), lam, c)
        d, s, status, passes = T.drive(shard, reduce_fn)
        q.put((rank, lo, hi, d, s.termination, s.num_iterations, s.num_successful_steps, s.num_line_search_steps, s.final_cost,
               status, passes, len(exchanges)))
    except Exception:
        import traceback
        q.put((rank, "ERROR", traceback.format_exc()))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 8])
@pytest.mark.parametrize("case", [(500, 6, 1.0, 1.0, 1.0), (300, 12, 2.0, 1.0, 1.0)], ids=["with_contraction", "plain"])
def test_depth_stage_sharded_gloo_matches_single(oracle, case, world):
    """world = 8 is the node's width (BASELINE config C4): the last rank deposits its max-norms in pack slots 15 and 23."""
    import torch.multiprocessing as mp
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_rank, args=(r, world, port, q, case)) for r in range(world)]
    [p.start() for p in procs]
    res = sorted([q.get(timeout=180) for _ in procs], key=lambda x: x[0])
    [p.join(30) for p in procs]
    for r in res:
        assert r[1] != "ERROR", r[2]
    n, seed, d0, lam, c = case
    cs = synthetic.full_rt(n, seed=seed)
    dref, sref, rc = oracle.depth_solve(cs.x1, cs.x2, cs.rot_init, cs.tran_init, np.full((n, 2), d0), lam=lam, c=c)
    for rank, lo, hi, d, term, it, succ, lsteps, cost, status, passes, exchanges in res:
        assert status == 0 and (term, it, succ, lsteps) == (sref.termination, sref.num_iterations, sref.num_successful_steps,
                                                             sref.num_line_search_steps)
        assert np.abs(d - dref[lo:hi]).max() <= 1e-7 * max(1.0, np.abs(dref).max())
        assert abs(cost - sref.final_cost) <= 1e-10 * sref.final_cost
        assert exchanges == passes                         # exactly one exchange per pass
    assert len(res) == world and all(r[8] == res[0][8] and r[10] == res[0][10] for r in res)   # bit-identical reduced cost, lock-step passes
