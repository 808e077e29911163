set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_batch.py -m gpu -q --durations=5 > gpurun_out/r3_gputest8.log 2>&1; echo "pytest exit $?" >> gpurun_out/r3_gputest8.log; tail -8 gpurun_out/r3_gputest8.log
grep -q "pytest exit 0" gpurun_out/r3_gputest8.log || exit 1
for rep in 1 2 3; do for IL in 0 1; do
  SBA_BATCH_INTERLEAVE=$IL timeout -k 10 200 python bench.py --workload c5 --steps 50 --frames 0 --no-cpu-baseline > gpurun_out/r3_il.json 2> gpurun_out/r3_il.err || exit 3
  python -c "
import json; d=json.load(open('gpurun_out/r3_il.json')); r=d['roofline']; print('interleave=$IL rep=$rep step %.2f us kernel %.2f us (min %.2f) frac %.3f sweep_alone %.2f us lm %.1f us' % (d['ms_per_step']*1e3, r['kernel_ms']*1e3, r['kernel_ms_min']*1e3, r['frac'], r['sweep_kernel_alone']['kernel_ms']*1e3, d['lm']['seconds_inside_the_library']*1e6))" | tee -a gpurun_out/r3_interleave.log
done; done
