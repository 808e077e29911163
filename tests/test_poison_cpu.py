"""CPU: a wedged device must be survivable end to end (VERDICT r2 item 2 / ADVICE r2).

Round 2's logs show what was not: gpurun_out/r2_gputest10.log -- a kernel never finished, `wait_for_sequence` gave up
after SBA_WAIT_TIMEOUT_S, and the caller's next move, destroying the handle, blocked for ever in hipStreamSynchronize.
Now a wait that gives up (or a device error) POISONS the handle: every later entry point on it returns SBA_ERR_HIP at once,
and destroy leaks the device resources instead of calling anything that waits for the device, returning non-zero.

The product's host translation units are compiled here with g++ against a stand-in for <hip/hip_runtime.h>
(tests/harness/fake_hip) and linked with a mock device whose stream can be wedged (tests/harness/wedge_harness.cpp counts
every runtime call that would block on a real wedged GPU).  No GPU, no HIP runtime, a handful of one-second time-outs.

The same harness drives the host side of the RESIDENT EVALUATOR (csrc/sba_resident.hpp: small problems are served by one
resident kernel per solve stage, commanded through a mapped record) against a host thread that speaks the device side of
the protocol: 200 commands through one kernel, a kernel that ended itself after its idle time-out and one that used up
its trip budget are restarted on the pending command, QUIT drains the stream, and a kernel that never answers poisons
the handle after SBA_WAIT_TIMEOUT_S."""
import os
import subprocess

from helpers import ROOT

CSRC = ROOT / "spherical_bundle_adjuster_amd" / "csrc"


def test_wedged_device_poisons_the_handle_and_destroy_returns(tmp_path):
    exe = tmp_path / "wedge_harness"
    srcs = [ROOT / "tests" / "harness" / "wedge_harness.cpp"] + [CSRC / f for f in
            ("sba_shim.cpp", "sba_transport.cpp", "sba_stages.cpp", "sba_batch.cpp", "sba_resident.cpp")]
    subprocess.run(["g++", "-O1", "-g", "-std=c++17", "-Wall", "-I", str(ROOT / "tests" / "harness" / "fake_hip"), "-o", str(exe),
                    *map(str, srcs), "-ldl", "-lpthread"], check=True)
    r = subprocess.run([str(exe)], capture_output=True, text=True, timeout=120, env=dict(os.environ, SBA_WAIT_TIMEOUT_S="1"))
    assert r.returncode == 0 and "wedge_harness: ok" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]
