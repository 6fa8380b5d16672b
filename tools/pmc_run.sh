#!/bin/bash
# usage: pmc_run.sh <tag> "<counters>" <python script> [args...]   (runs on the GPU box)
TAG=$1; PMC=$2; shift 2
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 240 rocprofv3 --pmc $PMC -d $OUT -o pmc -- python3 $REPO/$1 "${@:2}" > $OUT/stdout.txt 2> $OUT/stderr.txt || { echo "pmc run failed"; tail -5 $OUT/stderr.txt; exit 1; }
cat $OUT/stdout.txt
