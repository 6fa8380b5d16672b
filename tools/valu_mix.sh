#!/bin/bash
# Runs on the GPU box: dynamic VALU instruction mix of the bench workload by class (two PMC passes), for tools/valu_issue.py.
# usage: tools/valu_mix.sh <tag>   -> gpurun_out/valu_<tag>/mix{1,2}
set -o pipefail
TAG=${1:-r02}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/valu_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
BENCH="python3 $REPO/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras"
P1="SQ_INSTS_VALU SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT"
P2="SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MFMA_F16 SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE"
P3="SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES"
i=1
for P in "$P1" "$P2" "$P3"; do
  timeout -s KILL 240 rocprofv3 --pmc $P -d $OUT/mix$i -o mix$i -- $BENCH > $OUT/mix$i.json 2> $OUT/mix$i.err || { echo "pass $i failed"; tail -3 $OUT/mix$i.err; exit 1; }
  i=$((i+1))
done
python3 $REPO/tools/rocpd_summary.py $OUT | grep "trace_kernel"
python3 $REPO/tools/valu_issue.py $OUT $OUT/valu_issue.json > /dev/null    # with the mix passes present the class model is complete
