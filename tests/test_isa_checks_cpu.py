"""CPU: shape checks on the generated gfx950 code of the one-launch per-pair LM kernel (no GPU needed: hipcc cross-compiles).

batch_lm_kernel is a block-wide loop -- thread 0 prepares, barrier, all threads sweep, barrier, fold, barrier -- whose
correctness needs every barrier to be executed by all 256 threads of the block in the same trip of the SAME loop.  The
compiler once threaded a thread-0 region at the bottom of the loop into the one at its top, which left the other 255
threads in a private inner loop: their barrier ran without lane 0 (lane 0 waits for the inner loop to end under the
structured-control-flow lowering) and the kernel never finished.  The source now rules that shape out by construction
(one thread-0 region per trip, bracketed by two barriers of the same trip; a trip counter every thread keeps; the exit
decision read from LDS after a barrier) -- this file is the second line of defence: in every instantiation, each
s_barrier sits in loop depth exactly 1, there are exactly four of them, and nothing of the kernel spills beyond the
solver's scratch frame.  The one-launch step kernel (no loop around its barriers) must have its barriers at depth 0.
The flags are the Makefile's own (`make print-flags`), and the profiling builds (-DSBA_LM_PROFILE, -DSBA_STEP_PROFILE),
which add clock reads to the thread-0 regions, are held to the same shape."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "spherical_bundle_adjuster_amd", "csrc")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")

pytestmark = pytest.mark.skipif(not (os.path.exists(HIPCC) or shutil.which("hipcc")), reason="hipcc not available")


def _makefile_flags():
    """The device-compile flags libsba_hip.so is built with, straight from the Makefile."""
    r = subprocess.run(["make", "-s", "-C", CSRC, "print-flags"], check=True, capture_output=True, text=True)
    flags = r.stdout.split()
    assert any(f.startswith("--offload-arch=") for f in flags) and "-O3" in flags, flags
    return flags


def _compile_asm(tmp_path_factory, extra, source="sba_batch_kernels.hip"):
    out = tmp_path_factory.mktemp("isa") / (os.path.splitext(source)[0] + ".s")
    hipcc = HIPCC if os.path.exists(HIPCC) else shutil.which("hipcc")
    subprocess.run([hipcc, *_makefile_flags(), *extra, "-S", "--cuda-device-only",
                    os.path.join(CSRC, source), "-o", str(out)], check=True, capture_output=True, cwd=CSRC)
    return out.read_text()


@pytest.fixture(scope="module")
def depth_asm(tmp_path_factory):
    return _compile_asm(tmp_path_factory, [], "sba_depth.hip")


@pytest.fixture(scope="module")
def batch_asm(tmp_path_factory):
    return _compile_asm(tmp_path_factory, [])


@pytest.fixture(scope="module")
def batch_asm_profile(tmp_path_factory):
    return _compile_asm(tmp_path_factory, ["-DSBA_LM_PROFILE", "-DSBA_STEP_PROFILE"])


def _functions(asm):
    """name -> list of body lines (label line to the function's .Lfunc_end marker; a kernel may hold several s_endpgm)."""
    funcs, name, body = {}, None, []
    for line in asm.splitlines():
        m = re.match(r"^(_Z\w+):\s*(;.*)?$", line)
        if m and name is None:
            name, body = m.group(1), []
            continue
        if name is not None:
            body.append(line)
            if line.startswith(".Lfunc_end"):
                funcs[name] = body
                name = None
    return funcs


def _barrier_depths(body):
    """Loop depth of each s_barrier: the depth LLVM's asm printer annotates on the enclosing basic block."""
    depths, depth, in_header = [], 0, False
    for line in body:
        s = line.strip()
        is_label = re.match(r"^(\.LBB\d+_\d+:|; %bb\.\d+:)", s) is not None
        if is_label:
            m = re.search(r"Depth=(\d+)", s)
            depth = int(m.group(1)) if m else 0
            in_header = True
            continue
        if in_header and s.startswith(";"):            # continuation of a loop-header comment: innermost depth wins
            m = re.search(r"Depth=(\d+)", s)
            if m:
                depth = max(depth, int(m.group(1)))
            continue
        in_header = False
        if s.startswith("s_barrier"):
            depths.append(depth)
    return depths


def test_lm_kernel_barriers_are_block_uniform(batch_asm):
    funcs = _functions(batch_asm)
    lm = {k: v for k, v in funcs.items() if "batch_lm_kernel" in k}
    assert len(lm) >= 20, len(lm)            # modes x depth x store x kind x loss instantiations
    for name, body in lm.items():
        d = _barrier_depths(body)
        assert d == [1, 1, 1, 1], (name, d)


def test_profile_builds_keep_the_shape(batch_asm_profile):
    funcs = _functions(batch_asm_profile)
    lm = {k: v for k, v in funcs.items() if "batch_lm_kernel" in k}
    st = {k: v for k, v in funcs.items() if "batch_step_kernel" in k}
    assert len(lm) >= 20 and len(st) >= 20, (len(lm), len(st))
    for name, body in lm.items():
        d = _barrier_depths(body)
        assert d == [1, 1, 1, 1], (name, d)
    for name, body in st.items():
        d = _barrier_depths(body)
        assert d and all(x == 0 for x in d), (name, d)


def test_resident_kernel_barriers_are_block_uniform(batch_asm):
    """resident_sweep_kernel (csrc/sba_resident.hpp: one block stays resident for a whole solve stage and polls a host
    record) has the same loop shape by construction: two barriers around the one wave-0 region of a trip, the sweep's two,
    all at loop depth 1, and one barrier after the loop at depth 0."""
    funcs = _functions(batch_asm)
    rs = {k: v for k, v in funcs.items() if "resident_sweep_kernel" in k}
    assert len(rs) >= 20, len(rs)
    for name, body in rs.items():
        d = sorted(_barrier_depths(body))
        assert d == [0, 1, 1, 1, 1], (name, d)


def test_step_kernel_barriers_outside_loops(batch_asm):
    funcs = _functions(batch_asm)
    st = {k: v for k, v in funcs.items() if "batch_step_kernel" in k}
    assert len(st) >= 20, len(st)
    for name, body in st.items():
        d = _barrier_depths(body)
        assert d and all(x == 0 for x in d), (name, d)


def test_lm_kernel_scratch_is_the_solver_frame_only(batch_asm):
    sizes = {}
    for m in re.finditer(r"\.name:\s+(\S*batch_lm_kernel\S*)\n(?:.*\n){0,40}?\s+\.private_segment_fixed_size:\s+(\d+)", batch_asm):
        sizes[m.group(1)] = int(m.group(2))
    assert sizes, "no kernel metadata found"
    assert max(sizes.values()) <= 512, sizes


def test_depth_solve_kernels_barriers_are_block_uniform(depth_asm):
    """batch_depth_solve_kernel (one DepthStageSolver per pair on the device) and resident_depth_kernel have the loop shape
    of batch_lm_kernel: B0, B1 and the fold's barrier at loop depth 1 -- the solver's own loops (line-search bisection)
    sit in the thread-0 region and hold no barrier -- and one barrier after the loop at depth 0."""
    funcs = _functions(depth_asm)
    for pat, count, want in (("batch_depth_solve_kernel", 2, [0, 1, 1, 1]), ("resident_depth_kernel", 2, [0, 1, 1, 1]),
                             ("batch_depth_dyn_kernel", 2, [1, 1])):       # the dynamic-share pass: fold + end-of-item barrier per item
        ks = {k: v for k, v in funcs.items() if pat in k}
        assert len(ks) == count, (pat, sorted(ks))
        for name, body in ks.items():
            d = sorted(_barrier_depths(body))
            assert d == want, (name, d)


def test_depth_solve_kernel_keeps_the_stream_out_of_scratch(depth_asm):
    """The solver (thread 0) may use a scratch frame for its polynomial work; the budget here is what was measured when the
    kernel was written -- a jump means the per-match loop started to spill."""
    sizes = {}
    for m in re.finditer(r"\.name:\s+(\S*batch_depth_solve_kernel\S*)\n(?:.*\n){0,40}?\s+\.private_segment_fixed_size:\s+(\d+)", depth_asm):
        sizes[m.group(1)] = int(m.group(2))
    assert len(sizes) == 2, sizes
    assert max(sizes.values()) <= 1024, sizes
