#!/bin/bash
# rocprofv3 passes over a short bench.py run (run on the GPU box from the repo root).
# Usage: tools/prof_bench.sh <tag> [extra bench.py args, e.g. --workload c3]     -> gpurun_out/prof_<tag>/
# Counter passes are separate runs (FETCH_SIZE and WRITE_SIZE do not fit one pass).
set -e
TAG=${1:-bench}
shift || true
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="$R/bench.py --steps 2 --warmup 1 --spp-per-step 512 --no-cpu-baseline --no-bruteforce-pass --lane-utilisation 0 $@"
echo "python bench.py --steps 2 --warmup 1 --spp-per-step 512 --no-cpu-baseline --no-bruteforce-pass --lane-utilisation 0 $@" > $OUT/command.txt
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $ARGS > $OUT/trace.log 2>&1
timeout -s KILL 300 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_sq -- python3 $ARGS > $OUT/pmc_sq.log 2>&1
timeout -s KILL 300 rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_WAIT_ANY SQ_INSTS_FLAT SQ_LDS_BANK_CONFLICT TA_BUSY_avr --output-format csv -d $OUT/pmc_sq2 -- python3 $ARGS > $OUT/pmc_sq2.log 2>&1
timeout -s KILL 300 rocprofv3 --pmc FETCH_SIZE GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_fetch -- python3 $ARGS > $OUT/pmc_fetch.log 2>&1
timeout -s KILL 300 rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/pmc_write -- python3 $ARGS > $OUT/pmc_write.log 2>&1
find $OUT -name "*.csv" | sort
