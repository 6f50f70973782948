#!/bin/bash
# Counter passes over tools/pmc_grid.py (run through gpurun from the repo root)
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pmc_grid
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for set in "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum" \
           "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum" \
           "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE" \
           "TCC_TAG_STALL_sum TCC_EA0_RDREQ_32B_sum TCC_BUBBLE_sum TCC_EA0_RD_UNCACHED_32B_sum"; do
    i=$((i + 1))
    timeout -s KILL 170 rocprofv3 --pmc $set --output-format csv -d $OUT/p$i -- python3 $ROOT/tools/pmc_grid.py > $OUT/run$i.log 2> $OUT/run$i.err || { echo "pass $i failed"; tail -5 $OUT/run$i.err; exit 1; }
    echo "pass $i ok"
done
python3 - <<PY
import csv, glob, collections
for d in sorted(glob.glob("$OUT/p*")):
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        rows = [r for r in csv.DictReader(open(f)) if "trace_primary" in r["Kernel_Name"]]
        by = collections.OrderedDict()
        for r in rows:
            by.setdefault(r["Dispatch_Id"], {})[r["Counter_Name"]] = float(r["Counter_Value"])
        for k, v in by.items():
            print(d.split("/")[-1], k, " ".join(f"{n}={x:.4g}" for n, x in v.items()))
PY
