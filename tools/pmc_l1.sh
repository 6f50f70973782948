#!/bin/bash
# L1 / L2 request counts of the glass frame's kernels (development aid)
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_l1
mkdir -p $OUT
timeout -s KILL 120 rocprofv3 --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_ACCESSES_sum TCP_GATE_EN1_sum TD_TD_BUSY_sum TD_TC_STALL_sum GRBM_GUI_ACTIVE --output-format csv -d $OUT/p1 -- python3 $GRAFT_REPO_ROOT/tools/pmc_run.py c3 > $OUT/p1.out 2> $OUT/p1.err || exit 1
timeout -s KILL 120 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum --output-format csv -d $OUT/p2 -- python3 $GRAFT_REPO_ROOT/tools/pmc_run.py c3 > $OUT/p2.out 2> $OUT/p2.err || exit 1
python3 - "$OUT" <<'PY'
import csv, glob, collections, sys
agg = collections.defaultdict(list)
for d in sorted(glob.glob(sys.argv[1] + "/p*/*/*counter_collection.csv")):
    for r in csv.DictReader(open(d)):
        agg[(r["Kernel_Name"].split("(")[0][-50:], r["Counter_Name"])].append(float(r["Counter_Value"]))
for (k, c), v in sorted(agg.items()):
    if "rocclr" in k: continue
    print(f"{k:52s} {c:32s} {len(v):3d} {sum(v)/len(v):.6g}")
PY
