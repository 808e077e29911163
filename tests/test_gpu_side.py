"""GPU (-m gpu): callers / data formats either side of the path -- pixel->sphere, equi2cube, RCCL hook."""
import ctypes as C

import numpy as np
import pytest

from spherical_bundle_adjuster_amd import _cabi as cabi
from spherical_bundle_adjuster_amd import api, synthetic

pytestmark = pytest.mark.gpu


def test_keypoints_to_sphere(oracle):
    rng = np.random.default_rng(3)
    n, W, H = 20001, 3840, 1920
    kp = np.zeros((n, 7), dtype=np.float32)            # cv::KeyPoint: 28-byte records, pt.x pt.y first
    kp[:, 0] = rng.uniform(0, W, n)
    kp[:, 1] = rng.uniform(0, H, n)
    kp[:4, :2] = [[0, 0], [W, H], [W / 2, H / 2], [0, H]]
    got = api.keypoints_to_sphere(kp, W, H)
    ref = oracle.keypoints_to_sphere(kp, W, H)
    # device sin/cos vs glibc: a few ulp
    assert np.abs(got - ref).max() <= 4e-16
    assert np.abs(np.linalg.norm(got, axis=1) - 1).max() <= 4e-16
    assert api.keypoints_to_sphere(kp[:0], W, H).shape == (0, 3)


@pytest.mark.parametrize("store", [api.STORE_F64, api.STORE_F32])
def test_upload_keypoints_fused(oracle, store):
    """Matched key-point records -> device planes in one kernel: same sweep result as going through the host arrays."""
    from helpers import REL_TOL_F64, REL_TOL_F32, assert_normal_eq_close
    rng = np.random.default_rng(17)
    n, W, H = 5003, 3840, 1920
    kl = np.zeros((n, 7), dtype=np.float32)
    kl[:, 0] = rng.uniform(0, W, n)
    kl[:, 1] = rng.uniform(0.05 * H, 0.95 * H, n)
    kr = kl.copy()
    kr[:, :2] += rng.normal(0, 3.0, (n, 2)).astype(np.float32)
    d12 = rng.uniform(0.5, 2.0, (n, 2))
    rot, tran = [0.02, -0.01, 0.03], [0.1, 0.05, -0.2]
    x1, x2 = oracle.keypoints_to_sphere(kl, W, H), oracle.keypoints_to_sphere(kr, W, H)
    tol = REL_TOL_F64 if store == api.STORE_F64 else REL_TOL_F32
    with api.Problem() as p:
        p.upload_keypoints(kl, kr, W, H, d12=d12, store=store)
        assert p.size == n
        for mode in (api.MODE_ROT, api.MODE_TRAN, api.MODE_RT):
            got = p.eval(mode, rot, tran, depth_mode=api.DEPTH_PER_MATCH)
            assert_normal_eq_close(got, oracle.evaluate(mode, x1, x2, rot, tran, d12=d12), tol)
        # identical bits to the two-step route (device pixel->sphere, then upload): same device arithmetic
        fused = p.eval_pack(api.MODE_RT, rot, tran, depth_mode=api.DEPTH_PER_MATCH)
        p.upload(api.keypoints_to_sphere(kl, W, H), api.keypoints_to_sphere(kr, W, H), d12=d12, store=store)
        two_step = p.eval_pack(api.MODE_RT, rot, tran, depth_mode=api.DEPTH_PER_MATCH)
        assert np.array_equal(fused, two_step)
        p.upload_keypoints(kl[:0], kr[:0], W, H)
        assert p.size == 0 and p.eval(api.MODE_RT, rot, tran).cost == 0.0
        with pytest.raises(api.SbaError):
            p.upload_keypoints(kl, kr, 0, H)


@pytest.mark.parametrize("H,W,S", [(64, 128, 16), (480, 960, 150), (1920, 3840, 600), (100, 200, 33), (1080, 2160, 512),
                                   (1000, 2000, 250), (960, 1920, 300)])
def test_equi2cube_bit_exact(oracle, H, W, S):
    """Byte-exact against the oracle (integer index work): every output pixel identical."""
    rng = np.random.default_rng(H + S)
    im = rng.integers(0, 256, (H, W, 3), dtype=np.uint8)
    got = api.equi2cube(im, S)
    ref, clamped = oracle.equi2cube(im, S, clamp=True)
    # The reference's south-pole overrun (equi2cube.cpp:47-50): even S puts the bottom-face centre at
    # theta = pi exactly; whether int(H * pi / pi) lands on H (out of bounds) depends on H's rounding.
    overrun = S % 2 == 0 and int(H * np.arccos(-1.0) / np.pi) >= H
    assert clamped == (1 if overrun else 0)
    mism = int((got != ref).any(axis=2).sum())
    assert mism == 0, f"{mism} of {S * 6 * S} pixels differ"


def test_rccl_single_rank_and_hook():
    """nranks = 1 native RCCL all-reduce and the user hook both leave the pack unchanged / scaled."""
    c = synthetic.rotation_only(10000, seed=4)
    with api.Problem(0) as p:
        p.upload(c.x1, c.x2)
        base = p.eval_pack(api.MODE_ROT, c.rot_init, c.tran_init)
        calls = []

        def hook(ptr, count, stream):
            calls.append((ptr, count))
            return 0
        p.set_allreduce(hook)
        assert np.array_equal(p.eval_pack(api.MODE_ROT, c.rot_init, c.tran_init), base)
        assert calls and calls[0] == (p.pack_device_ptr, 24)
        p.set_allreduce(lambda *a: 3)
        with pytest.raises(api.SbaError) as ei:
            p.eval_pack(api.MODE_ROT, c.rot_init, c.tran_init)
        assert ei.value.code == -5
        p.set_allreduce(None)
        p.comm_init_rank(1, 0, api.comm_unique_id())
        assert np.array_equal(p.eval_pack(api.MODE_ROT, c.rot_init, c.tran_init), base)
        r, t, s = p.solve(api.MODE_ROT, c.rot_init, c.tran_init)
        assert s.termination.startswith("CONVERGENCE")


@pytest.mark.parametrize("n,store", [(1, api.STORE_F64), (255, api.STORE_F64), (256, api.STORE_F64), (257, api.STORE_F64),
                                     (100003, api.STORE_F64), (100003, api.STORE_F32)])
def test_epipolar_moments_vs_numpy(n, store):
    """Device pass of the 8-point initial guess: per-group A^T A of the kron(left, right) rows (.cpp:53-68)."""
    from test_initial_guess_cpu import group_moments
    c = synthetic.full_rt(n, seed=800 + n)
    x1, x2 = (c.x1, c.x2) if store == api.STORE_F64 else (c.x1.astype(np.float32).astype(np.float64),
                                                          c.x2.astype(np.float32).astype(np.float64))
    ref, _, _ = group_moments(x1, x2)
    with api.Problem(0) as p:
        p.upload(c.x1, c.x2, store=store)
        got = p.epipolar_moments()
        assert np.abs(got - ref).max() <= 1e-12 * max(np.abs(ref).max(), 1e-300)
        assert np.array_equal(got, p.epipolar_moments())            # fixed fold order: bit-identical
        if n >= 100000:
            e, t, ncand = p.initial_guess(80, 0.25, 3)
            e2, t2, n2 = api.initial_guess_from_moments(got, 80, 0.25, 3)
            assert np.array_equal(e, e2) and np.array_equal(t, t2) and ncand == n2


@pytest.mark.parametrize("n,store", [(20, api.STORE_F64), (40, api.STORE_F64), (2048, api.STORE_F64), (2048, api.STORE_F32),
                                     (50001, api.STORE_F64)])
def test_initial_guess_from_the_reference_subsets(oracle, n, store, monkeypatch):
    """SURVEY 8 f-1 with the reference's OWN sampling: 80 x random_array(n) -- std::iota +
    std::random_shuffle on the process's rand() stream (reference .hpp:182-211), the first int(n * 0.25) entries each
    (.cpp:130-141).  The oracle draws the lists with the real libstdc++ std::random_shuffle; the product must (a) draw
    the same lists from the same rand() state, (b) accumulate each list's A^T A on the device (vs numpy, 1e-12), (c) reach
    the consensus R_vec_out / T_vec_out of the reference recipe in numpy (explicit A + LAPACK SVD) on those lists --
    at the float32 resolution the reference keeps them in (cv::Vec3f).  BASELINE config C1 is n ~ 2 k."""
    from test_initial_guess_cpu import _recipe_on_subsets, subset_moments
    c = synthetic.full_rt(n, seed=900 + n, sigma=2e-4, outlier_fraction=0.0)
    x1, x2 = (c.x1, c.x2) if store == api.STORE_F64 else (c.x1.astype(np.float32).astype(np.float64),
                                                          c.x2.astype(np.float32).astype(np.float64))
    subsets = oracle.reference_trial_subsets(n, 80, reseed=True)
    api.reference_rand_seed(1)
    assert np.array_equal(api.reference_trial_subsets(n, 80, 0.25), subsets)
    e_ref, t_ref, nc_ref = _recipe_on_subsets(x1, x2, subsets)
    with api.Problem(0) as p:
        p.upload(c.x1, c.x2, store=store)
        mom = p.epipolar_subset_moments(subsets)
        ref = subset_moments(x1, x2, subsets)
        assert np.abs(mom - ref).max() <= 1e-12 * np.abs(ref).max()
        assert np.array_equal(mom, p.epipolar_subset_moments(subsets))              # fixed fold order
        api.reference_rand_seed(1)
        e, t, nc = p.initial_guess_reference(80, 0.25)
        after = api.reference_rand_next()
        # SBA_GUESS_SAMPLING=reference routes the seeded entry point to the same path
        monkeypatch.setenv("SBA_GUESS_SAMPLING", "reference")
        api.reference_rand_seed(1)
        e2, t2, nc2 = p.initial_guess(80, 0.25, 12345)
        monkeypatch.delenv("SBA_GUESS_SAMPLING")
        assert np.array_equal(e, e2) and np.array_equal(t, t2) and nc == nc2
        with pytest.raises(api.SbaError):
            p.epipolar_subset_moments(np.full((2, 3), n, dtype=np.int32))            # index out of range
    oracle.reference_trial_subsets(n, 80, reseed=True)
    assert after == oracle.c_rand()                                                  # consumed exactly the reference's draws
    loose = 1e-4 if n == 20 else 0.0           # 5 rows per trial: E is the 5th singular vector of a 5 x 9 matrix, ill-conditioned
    assert nc == nc_ref
    assert np.abs(e - e_ref).max() < 2e-6 * max(1.0, np.abs(e_ref).max()) + loose, (e, e_ref)
    assert min(np.abs(t - t_ref).max(), np.abs(t + t_ref).max()) < 1e-5 + 10 * loose


def test_set_depths(oracle):
    c = synthetic.full_rt(3000, seed=77)
    with api.Problem(0) as p:
        p.upload(c.x1, c.x2)                       # coordinates only
        p.set_depths(c.d12)
        got = p.eval(api.MODE_RT, c.rot_init, c.tran_init, depth_mode=api.DEPTH_PER_MATCH)
        ref = oracle.evaluate(2, c.x1, c.x2, c.rot_init, c.tran_init, d12=c.d12)
        assert np.abs(got.H - ref.H).max() <= 1e-12 * np.abs(ref.H).max()
        p.set_depths(2 * c.d12)                    # replace
        got2 = p.eval(api.MODE_RT, c.rot_init, c.tran_init, depth_mode=api.DEPTH_PER_MATCH)
        ref2 = oracle.evaluate(2, c.x1, c.x2, c.rot_init, c.tran_init, d12=2 * c.d12)
        assert np.abs(got2.H - ref2.H).max() <= 1e-12 * np.abs(ref2.H).max()


def _all_band_pixels(H, W):
    rows = np.arange(H // 4)
    kp = np.zeros((len(rows) * W, 7), dtype=np.float32)
    rr, cc = np.meshgrid(rows, np.arange(W), indexing="ij")
    kp[:, 0], kp[:, 1] = cc.ravel(), rr.ravel()
    return kp


# The bands of spherical_surf::do_all are +45, -45 and -90 (spherical_surf.cpp:137-153, 181-193; the 0 band never goes
# through rotate_pixel); 0, 30 and 90 ride along because they are the hard cases: at pitch 0 EVERY pixel sits exactly on
# an integer boundary of `height * acos(..) / M_PI`, at +-90 whole lines do.
@pytest.mark.parametrize("H,W", [(1920, 3840), (480, 960), (1080, 2160), (1000, 2000)])
def test_rotate_keypoints_and_crop_bit_exact_every_pitch(oracle, H, W):
    """Integer pixel maps (rotate_keypoint, crop_rotated_image: spherical_surf.cpp:48-123): ZERO differing results against
    the oracle over every pixel of the band, for every pitch -- decisions within 1e-6 pixel of an integer are taken on the
    host with the reference's own C library (csrc/sba_maps.hip)."""
    lib = cabi.load_library()
    allkp = _all_band_pixels(H, W)
    rng = np.random.default_rng(H)
    im = rng.integers(0, 256, (H, W, 3), dtype=np.uint8)
    pitches = (45.0, -45.0, -90.0, 0.0, 30.0, 90.0) if H <= 1080 else (45.0, -45.0, -90.0)
    for pitch in pitches:
        got = api.rotate_keypoints(allkp, pitch, W, H)
        ref = oracle.rotate_keypoints(allkp, pitch, W, H)
        assert np.array_equal(got[:, 2:], allkp[:, 2:])
        bad = (got[:, :2] != ref[:, :2]).any(axis=1)
        assert not bad.any(), f"pitch {pitch}: {int(bad.sum())} of {len(allkp)} key-points differ"
        g2 = api.crop_rotated_image(im, pitch)
        r2 = oracle.crop_rotated_image(im, pitch)
        assert np.array_equal(g2, r2), f"pitch {pitch}: {int((g2 != r2).any(axis=2).sum())} pixels differ"
        bits = int(np.float32(pitch).view(np.int32))
        decided = lib.sba_map_table_host_decided(0, 1, bits, H, W)
        assert decided >= 0                                   # the table exists (and is reused by the next call)
        if pitch in (45.0, -45.0, 30.0):
            assert decided <= 2 * (H // 4 + W)                # ties lie along a few lines: the host finishes a sliver
        if pitch == 0.0:
            assert decided == (H // 4) * W                    # every pixel is an exact tie: all decided by the host


def test_coordinate_maps_golden_fixtures():
    """The HIP maps against the COMMITTED fixtures (tests/golden/maps.npz): every result bit-identical."""
    from helpers import GOLDEN
    m = np.load(GOLDEN / "maps.npz", allow_pickle=False)
    H, W, S = (int(v) for v in m["geometry"])
    for pitch in (45.0, -45.0, -90.0, 0.0):
        tag = f"p{int(pitch)}".replace("-", "m")
        assert np.array_equal(api.rotate_keypoints(m["kp"], pitch, W, H)[:, :2], m[f"rotate_{tag}"]), pitch
        assert np.array_equal(api.crop_rotated_image(m["im"], pitch), m[f"crop_{tag}"]), pitch
    got = api.cube2equi_keypoints(m["cube_kp"], S, W, H)[:, :2]
    assert np.array_equal(got.view(np.uint32), m["cube2equi"].view(np.uint32))
    assert np.array_equal(api.equi2cube(m["im"], S), m["equi2cube"])


def test_matcher_coordinate_maps(oracle):
    """rotate_keypoint on real-valued key-points, cube2equi_pixel (equi2cube_surf.cpp:19-76; float32 results, compared
    bit for bit), device-resident variants, batched crop."""
    torch = pytest.importorskip("torch")
    lib = cabi.load_library()
    rng = np.random.default_rng(21)
    H, W, S, n = 1920, 3840, 600, 50001
    kp = np.zeros((n, 7), dtype=np.float32)
    kp[:, 0] = rng.uniform(0, W - 1, n); kp[:, 1] = rng.uniform(0, H // 4 - 1, n)
    kp[:, 2:] = rng.standard_normal((n, 5))                       # size/angle/response/octave/class_id ride along untouched
    stream = torch.cuda.current_stream().cuda_stream
    for pitch in (45.0, -45.0, -90.0):
        got = api.rotate_keypoints(kp, pitch, W, H)
        ref = oracle.rotate_keypoints(kp, pitch, W, H)
        assert np.array_equal(got, ref)
        dev = torch.from_numpy(kp.copy()).cuda()                  # device-resident records, in place
        cabi.check(lib, lib.sba_rotate_keypoints_device(0, C.c_void_p(stream), C.c_void_p(dev.data_ptr()), n, 28,
                                                        C.c_float(pitch), W, H))
        torch.cuda.synchronize()
        assert np.array_equal(dev.cpu().numpy(), ref)
    cube = np.zeros((n, 7), dtype=np.float32)
    cube[:, 0] = rng.uniform(0, 6 * S, n); cube[:, 1] = rng.uniform(0, S, n)
    cube[:6, 0] = [S / 2 + k * S for k in range(6)]; cube[:6, 1] = S / 2          # face centres: the two poles among them
    cube[6:12, 0] = [k * S for k in range(6)]; cube[6:12, 1] = 0.0                # face corners
    got = api.cube2equi_keypoints(cube, S, W, H)
    ref = oracle.cube2equi_keypoints(cube, S, W, H)
    assert np.array_equal(got[:, :2].view(np.uint32), ref[:, :2].view(np.uint32))  # bit-identical float32 pixels
    dev = torch.from_numpy(cube.copy()).cuda()
    cabi.check(lib, lib.sba_cube2equi_keypoints_device(0, C.c_void_p(stream), C.c_void_p(dev.data_ptr()), n, 28, S, W, H))
    torch.cuda.synchronize()
    assert np.array_equal(dev.cpu().numpy().view(np.uint32), ref.view(np.uint32))
    # batched, device-resident crop (3 frames)
    ims = rng.integers(0, 256, (3, 480, 960, 3), dtype=np.uint8)
    src = torch.from_numpy(ims).cuda()
    dst = torch.zeros((3, 120, 960, 3), dtype=torch.uint8, device="cuda")
    cabi.check(lib, lib.sba_crop_rotated_image_device(0, C.c_void_p(stream), C.c_void_p(src.data_ptr()), 480, 960,
                                                      C.c_float(-90.0), 3, C.c_void_p(dst.data_ptr())))
    torch.cuda.synchronize()
    for k in range(3):
        assert np.array_equal(dst[k].cpu().numpy(), oracle.crop_rotated_image(ims[k], -90.0))
    assert api.rotate_keypoints(kp[:0], 45.0, W, H).shape == (0, 7)
    # odd sizes take the one-pixel-per-lane gather
    im = rng.integers(0, 256, (50, 101, 3), dtype=np.uint8)
    assert np.array_equal(api.crop_rotated_image(im, -45.0), oracle.crop_rotated_image(im, -45.0))


def test_distributed_attach_one_rank_process_group(oracle):
    """distributed.attach() on a 1-rank NCCL(=RCCL) process group: the native transport (unique id shipped through
    torch.distributed, ncclCommInitRank inside the shim) and the torch hook (all_reduce on a tensor aliasing the
    shim's device pack) must both leave single-rank results unchanged and keep the LM working."""
    import os
    import socket
    torch = pytest.importorskip("torch")
    import torch.distributed as dist
    from spherical_bundle_adjuster_amd import distributed
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        c = synthetic.full_rt(20000, seed=31)
        with api.Problem(0) as p:
            p.upload(c.x1, c.x2, c.d12)
            base = p.eval_pack(api.MODE_RT, c.rot_init, c.tran_init, depth_mode=api.DEPTH_PER_MATCH)
            gm_plain = p.epipolar_moments()
            d_plain, sd_plain = p.solve_depths(c.rot_true, c.tran_true)          # no transport attached
            p.set_depths(c.d12)
            assert distributed.attach(p) == "none"
            assert distributed.attach(p, prefer_native=True, force=True) == "rccl-native"
            assert np.array_equal(p.eval_pack(api.MODE_RT, c.rot_init, c.tran_init, depth_mode=api.DEPTH_PER_MATCH), base)
            r, t, s_ = p.solve(api.MODE_RT, c.rot_init, c.tran_init, depth_mode=api.DEPTH_PER_MATCH)
            assert s_.termination.startswith("CONVERGENCE")
            # the d-only stage over the transport (sums all-reduced, the max gathered through one slot per rank)
            d_rccl, sd_rccl = p.solve_depths(c.rot_true, c.tran_true)
            assert (sd_rccl.num_iterations, sd_rccl.termination) == (sd_plain.num_iterations, sd_plain.termination)
            assert np.abs(d_rccl - d_plain).max() <= 1e-12
            assert np.array_equal(p.epipolar_moments(), gm_plain)                # 64 x 45 group moments all-reduced
        with api.Problem(0, stream=torch.cuda.current_stream().cuda_stream) as p:
            p.upload(c.x1, c.x2, c.d12)
            assert distributed.attach(p, prefer_native=False, force=True) == "torch-hook"
            got = p.eval_pack(api.MODE_RT, c.rot_init, c.tran_init, depth_mode=api.DEPTH_PER_MATCH)
            assert np.array_equal(got, base)
            # the alias really is the shim's device pack: explicit kernel -> device pack == SBA pack layout
            p.set_kernel(api.KERNEL_EXPLICIT)
            got = p.eval_pack(api.MODE_RT, c.rot_init, c.tran_init, depth_mode=api.DEPTH_PER_MATCH)
            torch.cuda.synchronize()
            assert np.array_equal(p._torch_pack_alias.cpu().numpy(), got)
            r2, t2, s2 = p.solve(api.MODE_RT, c.rot_init, c.tran_init, depth_mode=api.DEPTH_PER_MATCH)
            assert np.abs(r2 - r).max() <= 1e-12 and s2.num_iterations == s_.num_iterations
            d_hook, sd_hook = p.solve_depths(c.rot_true, c.tran_true)
            assert (sd_hook.num_iterations, sd_hook.termination) == (sd_plain.num_iterations, sd_plain.termination)
            assert np.abs(d_hook - d_plain).max() <= 1e-12
            assert np.array_equal(p.epipolar_moments(), gm_plain)                # the hook aliases that buffer too
        # the hook on a problem with its OWN stream (what "auto" falls back to when RCCL cannot be set up): the
        # collective is issued with that stream current
        with api.Problem(0) as p:
            p.upload(c.x1, c.x2, c.d12)
            assert distributed.attach(p, transport="hook", force=True) == "torch-hook"
            assert np.array_equal(p.eval_pack(api.MODE_RT, c.rot_init, c.tran_init, depth_mode=api.DEPTH_PER_MATCH), base)
            r3, t3, s3 = p.solve(api.MODE_RT, c.rot_init, c.tran_init, depth_mode=api.DEPTH_PER_MATCH)
            assert np.abs(r3 - r).max() <= 1e-12 and s3.num_iterations == s_.num_iterations
    finally:
        dist.destroy_process_group()


def test_upload_from_device_arrays(oracle):
    """sba_problem_upload_device: cv::Point3d-layout arrays already resident on the GPU (here torch tensors)."""
    torch = pytest.importorskip("torch")
    c = synthetic.full_rt(30001, seed=55)
    x1 = torch.from_numpy(c.x1).cuda(); x2 = torch.from_numpy(c.x2).cuda(); d12 = torch.from_numpy(c.d12).cuda()
    torch.cuda.synchronize()
    ref = oracle.evaluate(2, c.x1, c.x2, c.rot_init, c.tran_init, d12=c.d12)
    with api.Problem(0) as p:
        p.upload_device(x1.data_ptr(), x2.data_ptr(), d12.data_ptr(), 30001)
        got = p.eval(api.MODE_RT, c.rot_init, c.tran_init, depth_mode=api.DEPTH_PER_MATCH)
        assert np.abs(got.H - ref.H).max() <= 1e-12 * np.abs(ref.H).max() and got.n_outlier == ref.n_outlier
        p.upload_device(x1.data_ptr(), x2.data_ptr(), None, 30001, store=api.STORE_F32)     # no depths, f32 planes
        got = p.eval(api.MODE_ROT, c.rot_init, c.tran_init, 1.2, 0.8)
        ref2 = oracle.evaluate(0, c.x1.astype(np.float32).astype(np.float64), c.x2.astype(np.float32).astype(np.float64),
                               c.rot_init, c.tran_init, 1.2, 0.8)
        assert np.abs(got.H - ref2.H).max() <= 1e-12 * np.abs(ref2.H).max()


@pytest.mark.parametrize("H,W,S,F", [(96, 192, 24, 3), (480, 960, 150, 5), (90, 182, 22, 3), (1920, 3840, 600, 2)])
def test_gather_paths_agree_and_tiles_are_staged(oracle, H, W, S, F):
    """The remap runs through one of two kernels: 32 x 32 output tiles staged through LDS (frames on 16-byte boundaries),
    or one pixel per lane straight from global memory (any alignment; SBA_GATHER_TILED=0).  Both must give the oracle's
    bytes for every frame of a batch -- also when the source pointer or the frame size breaks the 16-byte alignment the
    tiled kernel needs (then the other kernel runs), and for the crop map with its outside pixels."""
    import os
    torch = pytest.importorskip("torch")
    lib = cabi.load_library()
    rng = np.random.default_rng(H * 7 + S)
    frames = rng.integers(0, 256, (F, H, W, 3), dtype=np.uint8)
    want = np.stack([oracle.equi2cube(frames[f], S, clamp=True)[0] for f in range(F)])
    stream = torch.cuda.current_stream().cuda_stream

    def run(src_t):
        dst = torch.zeros((F, S, 6 * S, 3), dtype=torch.uint8, device="cuda")
        cabi.check(lib, lib.sba_equi2cube_device(0, C.c_void_p(stream), C.c_void_p(src_t.data_ptr()), H, W, S, F,
                                                 C.c_void_p(dst.data_ptr())))
        torch.cuda.synchronize()
        return dst.cpu().numpy()

    src = torch.from_numpy(frames).cuda()
    assert np.array_equal(run(src), want)
    tiles, staged, lds = C.c_int(0), C.c_int(0), C.c_int(0)
    assert lib.sba_map_table_tiles(0, 0, S, H, W, C.byref(tiles), C.byref(staged), C.byref(lds)) == 0
    assert tiles.value == -(-S // 32) * -(-6 * S // 32) and 0 < staged.value <= tiles.value and 0 < lds.value <= 8192
    if (H, W) == (1920, 3840):
        assert staged.value >= 0.6 * tiles.value           # the C5 geometry: most tiles fit the LDS budget
    os.environ["SBA_GATHER_TILED"] = "0"
    try:
        assert np.array_equal(run(src), want)
    finally:
        del os.environ["SBA_GATHER_TILED"]
    # round 3 option: over-budget 32 x 32 tiles split into their four 16 x 16 quarters (at most 2 x 256 chunks each: they
    # always fit), so that EVERY tile is staged -- except one whose last chunk would reach past the end of the frame.
    # Measured 3.6 % slower than letting those tiles gather from global memory, hence off by default; byte-exact all the
    # same.  (The table cache is keyed by geometry: a cube size one pixel smaller gets a table of its own.)
    os.environ["SBA_GATHER_SUBTILES"] = "1"
    try:
        S2 = S - 1
        want2 = np.stack([oracle.equi2cube(frames[f], S2, clamp=True)[0] for f in range(F)])
        dst2 = torch.zeros((F, S2, 6 * S2, 3), dtype=torch.uint8, device="cuda")
        cabi.check(lib, lib.sba_equi2cube_device(0, C.c_void_p(stream), C.c_void_p(src.data_ptr()), H, W, S2, F,
                                                 C.c_void_p(dst2.data_ptr())))
        torch.cuda.synchronize()
        assert np.array_equal(dst2.cpu().numpy(), want2)
        assert lib.sba_map_table_tiles(0, 0, S2, H, W, C.byref(tiles), C.byref(staged), C.byref(lds)) == 0
        assert staged.value >= tiles.value - 4, (staged.value, tiles.value)
    finally:
        del os.environ["SBA_GATHER_SUBTILES"]
    # wider tiles (64 x 16, 128 x 8 output pixels; SBA_GATHER_TILE_W, read when a table is built): byte-exact all the same
    for k, tw in enumerate(("64", "128")):
        os.environ["SBA_GATHER_TILE_W"] = tw
        try:
            S3 = S - 2 - k
            want3 = np.stack([oracle.equi2cube(frames[f], S3, clamp=True)[0] for f in range(F)])
            dst3 = torch.zeros((F, S3, 6 * S3, 3), dtype=torch.uint8, device="cuda")
            cabi.check(lib, lib.sba_equi2cube_device(0, C.c_void_p(stream), C.c_void_p(src.data_ptr()), H, W, S3, F,
                                                     C.c_void_p(dst3.data_ptr())))
            torch.cuda.synchronize()
            assert np.array_equal(dst3.cpu().numpy(), want3), tw
            assert lib.sba_map_table_tiles(0, 0, S3, H, W, C.byref(tiles), C.byref(staged), C.byref(lds)) == 0
            assert tiles.value == -(-S3 // (1024 // int(tw))) * -(-6 * S3 // int(tw)) and staged.value > 0
        finally:
            del os.environ["SBA_GATHER_TILE_W"]
    # a source that does not start on a 16-byte boundary
    raw = torch.zeros(frames.size + 64, dtype=torch.uint8, device="cuda")
    shifted = raw[4:4 + frames.size]
    shifted.copy_(src.reshape(-1))
    assert shifted.data_ptr() % 16 != 0
    assert np.array_equal(run(shifted), want)
    # the crop map (outside pixels -> 0) for a batch, both kernels
    for pitch in (-45.0, -90.0, 30.0):
        wantc = np.stack([oracle.crop_rotated_image(frames[f], pitch) for f in range(F)])
        for env in (None, "0"):
            if env is not None:
                os.environ["SBA_GATHER_TILED"] = env
            try:
                dst = torch.zeros((F, H // 4, W, 3), dtype=torch.uint8, device="cuda")
                cabi.check(lib, lib.sba_crop_rotated_image_device(0, C.c_void_p(stream), C.c_void_p(src.data_ptr()), H, W,
                                                                  C.c_float(pitch), F, C.c_void_p(dst.data_ptr())))
                torch.cuda.synchronize()
                assert np.array_equal(dst.cpu().numpy(), wantc), (pitch, env)
            finally:
                os.environ.pop("SBA_GATHER_TILED", None)
