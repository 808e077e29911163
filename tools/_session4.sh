set -o pipefail
mkdir -p gpurun_out
timeout -k 10 700 python -m pytest tests -m gpu -q --durations=6 > gpurun_out/r3_gputest4.log 2>&1; echo "pytest exit $?" >> gpurun_out/r3_gputest4.log; tail -12 gpurun_out/r3_gputest4.log
grep -q "pytest exit 0" gpurun_out/r3_gputest4.log || exit 1
WT=$PWD/spherical_bundle_adjuster_amd/libsba_hip_wt.so
SBA_LIBRARY_PATH=$WT timeout -k 10 400 python -m pytest tests/test_gpu_parity.py tests/test_gpu_batch.py tests/test_gpu_pipeline.py -m gpu -q > gpurun_out/r3_gputest4_wt.log 2>&1; echo "pytest exit $?" >> gpurun_out/r3_gputest4_wt.log; tail -4 gpurun_out/r3_gputest4_wt.log
grep -q "pytest exit 0" gpurun_out/r3_gputest4_wt.log || exit 1
for rep in 1 2; do
  for V in default wt; do
    if [ $V = wt ]; then export SBA_LIBRARY_PATH=$WT; else unset SBA_LIBRARY_PATH; fi
    timeout -k 10 200 python bench.py --steps 50 --no-cpu-baseline --no-c5-leg --no-stage-leg --no-scaling-reference > gpurun_out/r3_ab_${V}_$rep.json 2> gpurun_out/r3_ab_${V}_$rep.err || exit 2
    timeout -k 10 200 python bench.py --workload c5 --steps 50 --frames 0 --no-cpu-baseline > gpurun_out/r3_ab_c5_${V}_$rep.json 2>> gpurun_out/r3_ab_${V}_$rep.err || exit 3
    python - <<PY
import json
d=json.load(open("gpurun_out/r3_ab_${V}_$rep.json")); c=json.load(open("gpurun_out/r3_ab_c5_${V}_$rep.json"))
c1=d["c1"]
print("$V rep $rep: step %.2f us kernel %.2f us | cold %.2f | c1 resident %s launch %s | c5 step %.2f us kernel %.2f" % (d["ms_per_step"]*1e3, d["roofline"]["kernel_ms"]*1e3, d["cold"]["ms_per_step"]*1e3, {k:round(v) for k,v in c1["gpu_resident"].items()}, {k:round(v) for k,v in c1["gpu_launch_per_sweep"].items()}, c["ms_per_step"]*1e3, c["roofline"]["kernel_ms"]*1e3))
PY
  done
done
unset SBA_LIBRARY_PATH
for SUB in 1 0; do
  SBA_GATHER_SUBTILES=$SUB timeout -k 10 200 python bench.py --workload c5 --steps 5 --pairs 256 --pair-matches 2000 --frames 512 --no-cpu-baseline > gpurun_out/r3_gather_sub$SUB.json 2> gpurun_out/r3_gather_sub$SUB.err || exit 4
  python -c "
import json; d=json.load(open('gpurun_out/r3_gather_sub$SUB.json')); print('subtiles=$SUB', d['equi2cube'])"
done
