#!/bin/bash
# Kernel times and hardware counters of one bench configuration, on the GPU box (through gpurun, from the repo root):
#   bash profiles/profile_config.sh <tag> <c2|c3|c4|c5|sss>
# 1. rocprofv3 --kernel-trace --stats over tools/pmc_run.py <cfg>; 2. one rocprofv3 --pmc run per counter group (never
# combined with tracing), the program directly after `--`.  Results under gpurun_out/prof_<tag>_<cfg>/; condense them
# into profiles/ with `python3 profiles/digest_config.py <tag> <cfg>` afterwards.
set -u
TAG=$1
CFG=$2
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_${TAG}_${CFG}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
PROG="python3 $ROOT/tools/pmc_run.py $CFG"
LIMIT=${PASS_SECONDS:-150}
timeout -k 10 $LIMIT rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- $PROG > $OUT/trace.out 2> $OUT/trace.err || { echo "kernel trace of $CFG failed"; tail -5 $OUT/trace.err; exit 1; }
i=0
while IFS= read -r pass; do
  [ -z "$pass" ] && continue
  i=$((i+1))
  echo "pass $i: $pass"
  timeout -s KILL $LIMIT rocprofv3 --pmc $pass --output-format csv -d $OUT/p$i -- $PROG > $OUT/p$i.out 2> $OUT/p$i.err || { echo "pass $i failed: $pass" | tee -a $OUT/failed.txt; exit 1; }
done <<'PASSES'
SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU
SQ_INSTS_VMEM_RD SQ_INST_CYCLES_VMEM_RD SQ_INSTS_SALU SQ_INSTS_BRANCH SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_SMEM SQ_ACTIVE_INST_ANY
TA_TA_BUSY_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TD_TD_BUSY_sum GRBM_GUI_ACTIVE
FETCH_SIZE
WRITE_SIZE
TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_REQ_sum
SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_WAIT_INST_LDS SQ_INST_CYCLES_SALU SQ_INSTS_VALU_TRANS SQ_ACTIVE_INST_SCA SQ_INSTS_WAVE32_LDS
PASSES
echo "profile of $CFG done: $OUT"
