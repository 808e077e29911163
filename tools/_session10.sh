set -o pipefail
mkdir -p gpurun_out
timeout -k 10 700 python -m pytest tests -m gpu -q > gpurun_out/r3_gputest10.log 2>&1; echo "pytest exit $?" >> gpurun_out/r3_gputest10.log; tail -4 gpurun_out/r3_gputest10.log
grep -q "pytest exit 0" gpurun_out/r3_gputest10.log || exit 1
bash tools/profile_stages.sh r03_stages 512 > gpurun_out/r3_profile_stages.log 2>&1 && echo "stage profile done"
bash tools/profile_round.sh r03 50 > gpurun_out/r3_profile_round.log 2>&1 && echo "round profile done"
timeout -k 10 500 python bench.py > gpurun_out/r3_bench3.json 2> gpurun_out/r3_bench3.err && python - <<'PY'
import json
d=json.load(open("gpurun_out/r3_bench3.json"))
print("value %.4g step %.2f us kernel %.2f frac %.3f cold %.2f us lm %s" % (d["value"], d["ms_per_step"]*1e3, d["roofline"]["kernel_ms"]*1e3, d["roofline"]["frac"], d["cold"]["ms_per_step"]*1e3, d["lm"]))
print("c1", {k: round(v) for k, v in d["c1"]["gpu_resident"].items()}, d["c1"]["iterations"]["equal"], {k: (round(v) if isinstance(v,float) else v) for k,v in d["c1"]["cpu_oracle"].items() if k!="what"})
print("c2", d["c2"]["ms_per_step"], d["c2"]["roofline"]["frac"], d["c2"]["lm"])
print("c5", d["c5"]["ms_per_step"], d["c5"]["roofline"]["frac"], d["c5"]["lm"]["seconds_inside_the_library"])
PY
