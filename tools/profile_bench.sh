#!/bin/bash
# Runs on the GPU box (via gpurun): kernel-trace stats + separate PMC passes of the bench workload.
# Outputs land in gpurun_out/prof_<tag>/; copy the summaries you want to keep into profiles/.
set -o pipefail
TAG=${1:-r01}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
BENCH="python3 $REPO/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras"
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $OUT/stats -o stats -- $BENCH > $OUT/bench_stats.json 2> $OUT/stats.err || exit 1
PMC1="SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU"
PMC2="SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_CYCLES_SALU"
PMC3="GRBM_GUI_ACTIVE GRBM_COUNT TCC_HIT_sum TCC_MISS_sum"
PMC4="FETCH_SIZE"
PMC5="WRITE_SIZE"
i=1
for P in "$PMC1" "$PMC2" "$PMC3" "$PMC4" "$PMC5"; do
  timeout -s KILL 240 rocprofv3 --pmc $P -d $OUT/pmc$i -o pmc$i -- $BENCH > $OUT/bench_pmc$i.json 2> $OUT/pmc$i.err || { echo "pmc pass $i failed"; tail -3 $OUT/pmc$i.err; exit 1; }
  i=$((i+1))
done
find $OUT -name "*.csv" | head -30
