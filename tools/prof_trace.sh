#!/bin/bash
# Kernel trace of a short bench.py run (run on the GPU box from the repo root).
# Usage: tools/prof_trace.sh <tag> [extra bench.py args]   -> gpurun_out/trace_<tag>/
set -e
TAG=$1
shift
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/trace_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 120 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-bruteforce-pass "$@" > $OUT/run.log 2>&1
find $OUT -name "*kernel_trace.csv" | head -1
