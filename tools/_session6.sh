set -o pipefail
mkdir -p gpurun_out
timeout -k 10 300 python -m pytest tests/test_gpu_side.py -m gpu -q -k "gather or equi2cube or crop" > gpurun_out/r3_gputest6.log 2>&1; echo "pytest exit $?" >> gpurun_out/r3_gputest6.log; tail -5 gpurun_out/r3_gputest6.log
grep -q "pytest exit 0" gpurun_out/r3_gputest6.log || exit 1
for rep in 1 2; do for TW in 32 64 128; do for FPB in 2 4; do
  SBA_GATHER_TILE_W=$TW SBA_GATHER_FPB=$FPB timeout -k 10 200 python bench.py --workload c5 --steps 5 --pairs 256 --pair-matches 2000 --frames 512 --no-cpu-baseline > gpurun_out/r3_tw.json 2> gpurun_out/r3_tw.err || exit 4
  python -c "
import json; d=json.load(open('gpurun_out/r3_tw.json'))['equi2cube']; print('tile_w=$TW fpb=$FPB rep=$rep ms_per_batch %.3f us_per_frame %.2f GBps %.0f' % (d['ms_per_batch'], d['ms_per_batch']/512*1e3, d['algorithmic_GBps']))" | tee -a gpurun_out/r3_tile_width.log
done; done; done
SBA_LIBRARY_PATH=$PWD/spherical_bundle_adjuster_amd/libsba_hip_prof.so timeout -k 10 300 python tools/step_profile.py > gpurun_out/r3_step_profile.log 2>&1; cat gpurun_out/r3_step_profile.log
