"""The batched equi2cube remap alone (F frames 3840x1920 -> S = 600 strips, `repeat` calls) -- the workload of the PMC
passes that look at gather_kernel's memory pipeline (tools/profile_gather.sh)."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch  # noqa: E402

from spherical_bundle_adjuster_amd import _cabi as cabi  # noqa: E402

F = int(sys.argv[1]) if len(sys.argv) > 1 else 64
repeat = int(sys.argv[2]) if len(sys.argv) > 2 else 3
H, W, S = 1920, 3840, 600
lib = cabi.load_library()
src = torch.randint(0, 256, (F, H, W, 3), dtype=torch.uint8, device="cuda")
dst = torch.zeros((F, S, 6 * S, 3), dtype=torch.uint8, device="cuda")
st = torch.cuda.current_stream().cuda_stream
for _ in range(repeat):
    cabi.check(lib, lib.sba_equi2cube_device(0, C.c_void_p(st), C.c_void_p(src.data_ptr()), H, W, S, F, C.c_void_p(dst.data_ptr())))
torch.cuda.synchronize()
print("done", F, repeat)
