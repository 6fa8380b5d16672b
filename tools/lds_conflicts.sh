#!/bin/bash
# usage (on the GPU box): tools/lds_conflicts.sh <tag> [lib.so] -- LDS bank-conflict share of trace_kernel on tools/quick.py 100
TAG=$1; LIB=$2
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/lds_$TAG
mkdir -p $OUT
if [ -n "$LIB" ]; then export RTX_LIB_PATH=$REPO/$LIB; fi
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 120 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES -d $OUT -o pmc -- python3 $REPO/tools/quick.py 100 > $OUT/stdout.txt 2> $OUT/stderr.txt || { echo "pmc run failed"; tail -5 $OUT/stderr.txt; exit 1; }
cat $OUT/stdout.txt
python3 - "$OUT" <<'PY'
import glob, os, sqlite3, sys
v = {}
for db in glob.glob(os.path.join(sys.argv[1], "**", "*_results.db"), recursive=True):
    c = sqlite3.connect(db)
    cols = [r[1] for r in c.execute("pragma table_info(counters_collection)")]
    nc = "kernel_name" if "kernel_name" in cols else "name"
    for name, counter, avg in c.execute("select %s, counter_name, max(value) from counters_collection group by 1, 2" % nc):
        if "trace_kernel" in name:
            v[counter] = avg
print({k: int(x) for k, x in v.items()})
print("LDS bank-conflict share of LDS-array cycles: %.2f %%" % (100.0 * v["SQ_LDS_BANK_CONFLICT"] / v["SQ_LDS_IDX_ACTIVE"]))
PY
