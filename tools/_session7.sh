set -o pipefail
mkdir -p gpurun_out
timeout -k 10 700 python -m pytest tests -m gpu -q --durations=5 > gpurun_out/r3_gputest7.log 2>&1; echo "pytest exit $?" >> gpurun_out/r3_gputest7.log; tail -10 gpurun_out/r3_gputest7.log
grep -q "pytest exit 0" gpurun_out/r3_gputest7.log || exit 1
timeout -k 10 100 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
timeout -k 10 500 python bench.py > gpurun_out/r3_bench2.json 2> gpurun_out/r3_bench2.err && python - <<'PY'
import json
d=json.load(open("gpurun_out/r3_bench2.json"))
print("value %.4g step %.2f us kernel %.2f frac %.3f cold %.2f us" % (d["value"], d["ms_per_step"]*1e3, d["roofline"]["kernel_ms"]*1e3, d["roofline"]["frac"], d["cold"]["ms_per_step"]*1e3))
print("scaling_reference", d["scaling_reference"]["ms_per_step"], d["scaling_reference"]["frac"])
print("c1", {k: round(v) for k, v in d["c1"]["gpu_resident"].items()}, {k: round(v) for k, v in d["c1"]["gpu_launch_per_sweep"].items()}, d["c1"]["iterations"])
print("c1 cpu", d["c1"]["cpu_oracle"])
print("c2", d["c2"])
print("c5", d["c5"]["ms_per_step"], d["c5"]["roofline"]["frac"], d["c5"]["equi2cube"])
print("stages", d["stages"])
PY
