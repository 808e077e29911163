set -o pipefail
mkdir -p gpurun_out
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "depth" > gpurun_out/r3_gputest12.log 2>&1; echo "pytest exit $?" >> gpurun_out/r3_gputest12.log; tail -3 gpurun_out/r3_gputest12.log
grep -q "pytest exit 0" gpurun_out/r3_gputest12.log || exit 1
for rep in 1 2 3; do for A in 1 2; do for CAP in 2; do
  SBA_DEPTH_AHEAD=$A timeout -k 10 120 python tools/depth_workload.py | sed "s/^/ahead=$A rep=$rep /" | tee -a gpurun_out/r3_depth_ahead.log
done; done; done
export TMPDIR=/tmp
for A in 1 2; do
  export SBA_DEPTH_AHEAD=$A
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_depth_ahead$A -o t -- python3 tools/depth_workload.py > /dev/null 2> gpurun_out/r3_depth_ahead_trace$A.err || exit 2
  python3 - <<PY
import csv,glob
for f in glob.glob("gpurun_out/prof_depth_ahead$A/*/*kernel_stats.csv")+glob.glob("gpurun_out/prof_depth_ahead$A/*kernel_stats.csv"):
    for r in csv.DictReader(open(f)):
        if "depth_step" in r["Name"]: print("ahead=$A rocprofv3:", r["Calls"], "calls avg %.2f us min %.2f us" % (float(r["AverageNs"])/1e3, float(r["MinNs"])/1e3))
PY
done 2>&1 | tee -a gpurun_out/r3_depth_ahead.log
