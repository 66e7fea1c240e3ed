#!/bin/bash
# Counter passes over the closest-hit microbenchmark (run on the GPU box from the repo root).
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof_closest
mkdir -p $OUT
timeout -s KILL 120 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $R/tools/microbench_closest.py cubes > $OUT/trace.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_BUSY_CYCLES SQ_INSTS_SMEM --output-format csv -d $OUT/pmc1 -- python3 $R/tools/microbench_closest.py cubes > $OUT/pmc1.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM SQ_THREAD_CYCLES_VALU GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc2 -- python3 $R/tools/microbench_closest.py cubes > $OUT/pmc2.log 2>&1
find $OUT -name "*.csv" | head -20
