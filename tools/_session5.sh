set -o pipefail
mkdir -p gpurun_out
timeout -k 10 400 python -m pytest tests/test_gpu_pipeline.py tests/test_gpu_side.py -m gpu -q > gpurun_out/r3_gputest5.log 2>&1; echo "pytest exit $?" >> gpurun_out/r3_gputest5.log; tail -5 gpurun_out/r3_gputest5.log
grep -q "pytest exit 0" gpurun_out/r3_gputest5.log || exit 1
timeout -k 10 120 spherical_bundle_adjuster_amd/csrc/build/chunk_probe 8 5 > gpurun_out/r3_chunk_probe.log 2>&1 && cat gpurun_out/r3_chunk_probe.log
bash tools/profile_stages.sh r03_stages 512 > gpurun_out/r3_profile_stages.log 2>&1 && echo "stage profile done"
bash tools/profile_round.sh r03 50 > gpurun_out/r3_profile_round.log 2>&1 && echo "round profile done"
