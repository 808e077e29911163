set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -q --durations=15 > gpurun_out/r3_gputest2.log 2>&1; echo "pytest exit $?" >> gpurun_out/r3_gputest2.log; tail -25 gpurun_out/r3_gputest2.log
grep -q "pytest exit 0" gpurun_out/r3_gputest2.log || exit 1
for OCC in 2 3 4; do for CAP in 2 3 4; do [ $CAP -le $OCC ] && SBA_DEPTH_OCC=$OCC SBA_DEPTH_BLOCKS_PER_CU=$CAP timeout -k 10 120 python tools/depth_workload.py | sed "s/^/occ=$OCC cap=$CAP /" >> gpurun_out/r3_depth_tune.log; done; done
cat gpurun_out/r3_depth_tune.log
timeout -k 10 400 python bench.py > gpurun_out/r3_bench1.json 2> gpurun_out/r3_bench1.err && tail -c 3000 gpurun_out/r3_bench1.json
