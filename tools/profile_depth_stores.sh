#!/bin/bash
# Run ON THE GPU BOX (through gpurun): A/B of depth_step_kernel's candidate stores, non-temporal (SBA_DEPTH_NT_STORES=1)
# against plain (=0): kernel trace for the durations, then WRITE_SIZE / FETCH_SIZE and the raw write-request counters.
set -o pipefail
OUT=gpurun_out/prof_depth_stores
mkdir -p $OUT
export TMPDIR=/tmp
for NT in 1 0; do
  export SBA_DEPTH_NT_STORES=$NT
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_nt$NT -o t -- python3 tools/depth_workload.py > $OUT/trace_nt$NT.json 2> $OUT/trace_nt$NT.err || { echo "trace nt=$NT failed" >&2; exit 1; }
  echo "trace nt=$NT done" >&2
  i=0
  for SET in "WRITE_SIZE" "FETCH_SIZE" "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_WR_UNCACHED_32B_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum"; do
    i=$((i+1))
    timeout -k 10 200 rocprofv3 --pmc $SET --output-format csv -d $OUT/pmc_nt${NT}_$i -o p -- python3 tools/depth_workload.py > $OUT/pmc_nt${NT}_$i.log 2>&1 || { echo "pmc nt=$NT set $i failed (continuing)" >&2; }
    echo "pmc nt=$NT set $i done" >&2
  done
done
python3 - <<PY
import csv, glob, collections, statistics
for f in sorted(glob.glob("$OUT/pmc_*/*counter_collection.csv")):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "depth_step_kernel" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in acc.items():
        print(f.split("/")[-2], k, "launches", len(v), "median %.6g" % statistics.median(v), "min %.6g" % min(v), "max %.6g" % max(v))
for f in sorted(glob.glob("$OUT/trace_*/*kernel_stats.csv")):
    for r in csv.DictReader(open(f)):
        if "depth_step" in r["Name"] or "depth_final" in r["Name"]:
            print(f.split("/")[-2], r["Name"][:40], "calls", r["Calls"], "avg ns", r["AverageNs"], "min", r["MinNs"], "max", r["MaxNs"])
PY
