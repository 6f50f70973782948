#!/bin/bash
# PMC passes (one rocprofv3 run each, never combined with tracing) over an arbitrary command:
#   bash profiles/run_pmc.sh <tag> <command...>
set -u
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
while IFS= read -r pass; do
  [ -z "$pass" ] && continue
  i=$((i+1))
  timeout -s KILL 120 rocprofv3 --pmc $pass --output-format csv -d $OUT/p$i -- "$@" > $OUT/p$i.out 2> $OUT/p$i.err || echo "pass $i failed: $pass" >> $OUT/failed.txt
done <<'PASSES'
SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU
SQ_INSTS_VMEM_RD SQ_INST_CYCLES_VMEM_RD SQ_INSTS_SALU SQ_INSTS_BRANCH SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_INSTS_LDS SQ_ACTIVE_INST_LDS
TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TD_TD_BUSY_sum TD_TC_STALL_sum GRBM_GUI_ACTIVE GRBM_TA_BUSY
SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_SALU
PASSES
python3 - "$OUT" <<'PY'
import csv, glob, collections, sys
out = sys.argv[1]
agg = collections.defaultdict(list)
for d in sorted(glob.glob(out + "/p*/*/*counter_collection.csv")):
    for r in csv.DictReader(open(d)):
        agg[(r["Kernel_Name"].split("(")[0][-60:], r["Counter_Name"])].append(float(r["Counter_Value"]))
with open(out + "/summary.csv", "w") as f:
    f.write("kernel,counter,launches,mean\n")
    for (k, c), v in sorted(agg.items()):
        if "rocclr" in k: continue
        f.write(f"\"{k}\",{c},{len(v)},{sum(v)/len(v):.6g}\n")
print(open(out + "/summary.csv").read())
PY
