#!/bin/bash
# usage (on the GPU box): tools/pmc_waits.sh <tag> [lib.so] -- where a wave's cycles go (tools/quick.py 100, RTX_FILTER from the environment):
# executing, waiting on an s_waitcnt, or ready and not issued
TAG=$1; LIB=$2
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/pmcw_$TAG
mkdir -p $OUT
if [ -n "$LIB" ]; then export RTX_LIB_PATH=$REPO/$LIB; fi
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_BRANCH -d $OUT/p -o pmc -- python3 $REPO/tools/quick.py 100 > $OUT/stdout.txt 2> $OUT/stderr.txt || { echo "pmc run failed"; tail -5 $OUT/stderr.txt; exit 1; }
cat $OUT/stdout.txt
python3 $REPO/tools/rocpd_summary.py $OUT | grep trace_kernel || true
timeout -s KILL 120 rocprofv3 --pmc SQ_INST_CYCLES_SALU SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_IFETCH SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_MISC -d $OUT/q -o pmc -- python3 $REPO/tools/quick.py 100 > $OUT/stdout2.txt 2> $OUT/stderr2.txt || { echo "pmc run 2 failed"; tail -5 $OUT/stderr2.txt; exit 1; }
python3 $REPO/tools/rocpd_summary.py $OUT | grep trace_kernel | grep "SALU\|VMEM\|LDS\|IFETCH\|SMEM\|MISC" || true
