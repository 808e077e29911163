#!/bin/bash
# Run ON THE GPU BOX (through gpurun): rocprofv3 kernel trace + the two HBM PMC passes of bench.py.
# Outputs land under gpurun_out/prof_<tag>/ ; copy the summaries you want judged into profiles/.
set -o pipefail
TAG=${1:-r01}
STEPS=${2:-50}
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
ARGS="bench.py --steps $STEPS --warmup 5 --no-cpu-baseline --no-c5-leg --no-stage-leg --no-c1-leg --no-scaling-reference --no-cold-leg"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o trace -- python3 $ARGS > $OUT/bench_trace.json 2> $OUT/trace.err || exit 1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o fetch -- python3 $ARGS > $OUT/bench_fetch.json 2> $OUT/fetch.err || exit 2
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o write -- python3 $ARGS > $OUT/bench_write.json 2> $OUT/write.err || exit 3
find $OUT -name "*.csv" | head -20
