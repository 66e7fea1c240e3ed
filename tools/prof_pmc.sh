#!/bin/bash
# One rocprofv3 counter pass over a short bench.py run (run on the GPU box from the repo root).
# Usage: tools/prof_pmc.sh <tag> "<counters>" [extra bench.py args]   -> gpurun_out/pmc_<tag>/
set -e
TAG=$1
COUNTERS=$2
shift 2
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 400 rocprofv3 --pmc $COUNTERS --output-format csv -d $OUT -- python3 $R/bench.py --steps 1 --warmup 1 --spp-per-step 512 --no-cpu-baseline --no-bruteforce-pass "$@" > $OUT/run.log 2>&1
find $OUT -name "*counter_collection.csv" | head -2
