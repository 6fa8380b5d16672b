#!/bin/bash
# usage (on the GPU box): tools/pmc_c5.sh <tag> -- counter passes of BASELINE configuration 5 (table read through L2):
# L2 hits / misses, fetched bytes, LDS conflicts, VALU lane utilisation.  One rocprofv3 run per pass, summarised.
TAG=$1
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/pmc_c5_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
pass() {
    local name=$1; shift
    timeout -s KILL 120 rocprofv3 --pmc "$@" -d $OUT/$name -o $name -- python3 $REPO/tools/c5_run.py 8 > $OUT/$name.stdout 2> $OUT/$name.stderr || { echo "pass $name failed"; tail -5 $OUT/$name.stderr; exit 1; }
}
pass l2 TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum SQ_INSTS_VMEM_RD SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES || exit 1
pass fetch FETCH_SIZE || exit 1
pass lanes SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_ANY SQ_INSTS_SALU || exit 1
cat $OUT/l2.stdout
python3 $REPO/tools/rocpd_summary.py $OUT | grep -E "^==|trace_kernel"
