#!/bin/bash
# Runs on the GPU box: the calibration loops and the bench workload under the same PMC counters, then tools/valu_issue.py.
# usage: tools/valu_issue.sh <tag>   -> gpurun_out/valu_<tag>/{calib,bench}, gpurun_out/valu_<tag>/valu_issue.json
set -o pipefail
TAG=${1:-r02}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/valu_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
PMC="SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE"
timeout -s KILL 120 rocprofv3 --pmc $PMC -d $OUT/calib -o calib -- $REPO/tools/microbench/bin/valu_issue_calib > $OUT/calib.txt 2> $OUT/calib.err || { echo "calibration pass failed"; tail -3 $OUT/calib.err; exit 1; }
timeout -s KILL 240 rocprofv3 --pmc $PMC -d $OUT/bench -o bench -- python3 $REPO/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras > $OUT/bench.json 2> $OUT/bench.err || { echo "bench pass failed"; tail -3 $OUT/bench.err; exit 1; }
cat $OUT/calib.txt
python3 $REPO/tools/valu_issue.py $OUT $OUT/valu_issue.json
