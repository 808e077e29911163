"""CPU: the product's HIP-free host code (LmSolver, ArmijoSearch, DepthStageSolver, 8-point host math, rotation helpers)
compiled with g++ -fsanitize=address,undefined and run once (tests/harness/sanitize_main.cpp).  GPU sanitizers are not
available on this pool; this is the part of the product that can run under one."""
import subprocess

from helpers import ROOT


def test_host_code_under_asan_and_ubsan(tmp_path):
    exe = tmp_path / "sanitize_main"
    src = ROOT / "tests" / "harness" / "sanitize_main.cpp"
    subprocess.run(["g++", "-O1", "-g", "-std=c++17", "-fsanitize=address,undefined", "-fno-sanitize-recover=all",
                    "-fno-omit-frame-pointer", "-o", str(exe), str(src)], check=True)
    r = subprocess.run([str(exe)], capture_output=True, text=True, timeout=120,
                       env={"ASAN_OPTIONS": "detect_leaks=1:abort_on_error=0", "UBSAN_OPTIONS": "print_stacktrace=1"})
    assert r.returncode == 0 and "sanitize_main: ok" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]
