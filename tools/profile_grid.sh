#!/bin/bash
# usage (on the GPU box): tools/profile_grid.sh -- the grid filter (RTX_FILTER=grid) on BASELINE configuration 5 (20 164 spheres,
# 100 spp): a kernel trace, one SQ pass (instruction counts, lane utilisation, waits) and one TCC pass (L2 hit rate); summary on stdout.
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/prof_grid
mkdir -p $OUT
export RTX_FILTER=grid
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 120 rocprofv3 --kernel-trace --stats -d $OUT/trace -o stats -- python3 $REPO/tools/c5_run.py 100 > $OUT/stdout_trace.txt 2> $OUT/stderr_trace.txt || { echo "trace run failed"; tail -5 $OUT/stderr_trace.txt; exit 1; }
cat $OUT/stdout_trace.txt
timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_WAIT_INST_ANY -d $OUT/sq -o p -- python3 $REPO/tools/c5_run.py 100 > $OUT/stdout_sq.txt 2> $OUT/stderr_sq.txt || { echo "sq pass failed"; tail -5 $OUT/stderr_sq.txt; exit 1; }
timeout -s KILL 120 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum -d $OUT/tcc -o p -- python3 $REPO/tools/c5_run.py 100 > $OUT/stdout_tcc.txt 2> $OUT/stderr_tcc.txt || { echo "tcc pass failed"; tail -5 $OUT/stderr_tcc.txt; exit 1; }
python3 $REPO/tools/rocpd_summary.py $OUT
