set -o pipefail
mkdir -p gpurun_out
timeout -k 10 300 python -m pytest tests/test_gpu_pipeline.py -m gpu -q --durations=8 > gpurun_out/r3_gputest3.log 2>&1; echo "pytest exit $?" >> gpurun_out/r3_gputest3.log; tail -30 gpurun_out/r3_gputest3.log
grep -q "pytest exit 0" gpurun_out/r3_gputest3.log || exit 1
timeout -k 10 100 python - > gpurun_out/r3_hsa_rand.log 2>&1 <<'PY'
import sys
sys.path.insert(0, ".")
from oracle import oracle_py as orc
orc.c_srand(1)
from spherical_bundle_adjuster_amd import api
import numpy as np
with api.Problem(0) as p:
    p.upload(np.eye(3), np.eye(3))
    p.eval_pack(0, [0.1, 0, 0], [0, 0, 0])
v = [orc.c_rand() for _ in range(3)]
print("first rand() values after srand(1) + HIP initialisation + one sweep:", v, "untouched stream would give [1804289383, 846930886, 1681692777]")
PY
cat gpurun_out/r3_hsa_rand.log
timeout -k 10 200 python - > gpurun_out/r3_c1.json 2> gpurun_out/r3_c1.err <<'PY'
import json, sys
sys.path.insert(0, ".")
import bench
for n in (512, 2048, 4096, 8192):
    print(json.dumps({"n": n, **bench.c1_leg(0, n=n, reps=20)}), flush=True)
PY
cat gpurun_out/r3_c1.json | cut -c1-1500
timeout -k 10 600 python -m pytest tests -m gpu -q --durations=8 > gpurun_out/r3_gputest3b.log 2>&1; echo "pytest exit $?" >> gpurun_out/r3_gputest3b.log; tail -15 gpurun_out/r3_gputest3b.log
