#!/bin/bash
# Run ON THE GPU BOX (through gpurun): PMC passes over tools/gather_workload.py -- what bounds the remap's gather kernel (gather_tiled_kernel by default)?
# (A pass with TA_BUSY_avr / TA_*_STALLED_* / TCP_PENDING_STALL_CYCLES aborted inside rocprofv3 on this pool and then sat
# silent until gpurun killed it: those counters are left out.)  Every pass prints a line so a slow one is not taken for hung.
set -o pipefail
OUT=gpurun_out/prof_gather
mkdir -p $OUT
export TMPDIR=/tmp
ARGS="tools/gather_workload.py ${1:-64} 3"
i=0
for SET in "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum" \
           "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_DRAM_sum TCC_REQ_sum" \
           "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_TCC_READ_REQ_LATENCY_sum GRBM_GUI_ACTIVE" \
           "SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_VMEM_RD SQ_WAIT_INST_ANY" \
           "TCC_BUSY_avr TCC_TAG_STALL_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCC_EA0_RDREQ_LEVEL_sum"; do
  i=$((i+1))
  timeout -k 10 150 rocprofv3 --pmc $SET --output-format csv -d $OUT/p$i -o p -- python3 $ARGS > $OUT/p$i.log 2>&1 || { echo "pass $i failed: stopping" >&2; break; }
  echo "pass $i done" >&2
done
python3 - <<PY
import csv, glob, collections
for f in sorted(glob.glob("$OUT/p*/*counter_collection.csv")):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "gather" in r["Kernel_Name"]:      # gather_tiled_kernel (default) or gather_kernel
            acc[(r["Kernel_Name"].split("(")[0][:32], r["Counter_Name"])].append(float(r["Counter_Value"]))
    for (kn, k), v in acc.items():
        print(f.split("/")[-2], kn, k, "launches", len(v), "mean %.6g" % (sum(v) / len(v)))
PY
