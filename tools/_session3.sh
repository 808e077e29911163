set -o pipefail
mkdir -p gpurun_out
timeout -k 10 300 python -m pytest tests/test_gpu_pipeline.py -m gpu -q -x --durations=8 > gpurun_out/r3_gputest3.log 2>&1; echo "pytest exit $?" >> gpurun_out/r3_gputest3.log; tail -30 gpurun_out/r3_gputest3.log
grep -q "pytest exit 0" gpurun_out/r3_gputest3.log || exit 1
timeout -k 10 200 python - > gpurun_out/r3_c1.json 2> gpurun_out/r3_c1.err <<'PY'
import json, sys
sys.path.insert(0, ".")
import bench
for n in (512, 2048, 4096, 8192):
    print(json.dumps({"n": n, **bench.c1_leg(0, n=n, reps=20)}), flush=True)
PY
cat gpurun_out/r3_c1.json | cut -c1-1500
timeout -k 10 600 python -m pytest tests -m gpu -q --durations=8 > gpurun_out/r3_gputest3b.log 2>&1; echo "pytest exit $?" >> gpurun_out/r3_gputest3b.log; tail -15 gpurun_out/r3_gputest3b.log
