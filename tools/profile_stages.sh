#!/bin/bash
# Run ON THE GPU BOX (through gpurun): rocprofv3 kernel trace + the two HBM PMC passes of tools/stage_workload.py
# (d-only stage, 8-point moments, config C5 batched step + LM, equi2cube of 512 frames).
# Outputs land under gpurun_out/prof_<tag>/ ; tools/summarize_profiles.py --stages condenses them into profiles/.
set -o pipefail
TAG=${1:-r02_stages}
FRAMES=${2:-512}
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
ARGS="tools/stage_workload.py --frames $FRAMES"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o trace -- python3 $ARGS > $OUT/workload_trace.json 2> $OUT/trace.err || exit 1
echo "trace done" >&2
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o fetch -- python3 $ARGS > $OUT/workload_fetch.json 2> $OUT/fetch.err || exit 2
echo "fetch done" >&2
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o write -- python3 $ARGS > $OUT/workload_write.json 2> $OUT/write.err || exit 3
find $OUT -name "*.csv" | head -20
