#!/bin/bash
# usage (on the GPU box): tools/pmc_quick.sh <tag> [lib.so] -- one SQ instruction-mix PMC pass of tools/quick.py 100, summarised
TAG=$1; LIB=$2
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/pmcq_$TAG
mkdir -p $OUT
if [ -n "$LIB" ]; then export RTX_LIB_PATH=$REPO/$LIB; fi
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_MFMA -d $OUT/p -o pmc -- python3 $REPO/tools/quick.py 100 > $OUT/stdout.txt 2> $OUT/stderr.txt || { echo "pmc run failed"; tail -5 $OUT/stderr.txt; exit 1; }
cat $OUT/stdout.txt
python3 $REPO/tools/rocpd_summary.py $OUT | grep trace_kernel || true
