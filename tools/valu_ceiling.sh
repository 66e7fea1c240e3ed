#!/bin/bash
# Vector-instruction issue ceiling of gfx950, measured: builds tools/microbench_valu.hip, lists what the timed loop of
# every kernel holds according to the compiler's assembly listing (so the instruction count is read, not assumed),
# and runs it.  Run on the GPU box from the repo root:  tools/valu_ceiling.sh > gpurun_out/valu_ceiling.txt
set -e
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$R"
hipcc --offload-arch=gfx950 -O3 -o tools/microbench_valu tools/microbench_valu.hip
hipcc --offload-arch=gfx950 -O3 --cuda-device-only -S -o /tmp/microbench_valu.s tools/microbench_valu.hip 2>/dev/null
echo "# timed loop of each kernel (assembly listing: instructions between the loop label and its backward branch)"
awk '
/^_Z6k_valuILi[0-9]+E/ { kernel = $1; sub(/^_Z6k_valuILi/, "", kernel); sub(/E.*/, "", kernel); next }
/Inner Loop Header/ { inloop = 1; delete c; next }
inloop && /^\t?s_cbranch_scc0/ { line = "#   kind " kernel ":"; for (k in c) line = line " " c[k] " x " k; print line " + s_cbranch_scc0"; inloop = 0; next }
inloop && /^[ \t]*(v|s|ds)_[a-z0-9_]+/ { m = $1; c[m]++ }
' /tmp/microbench_valu.s
./tools/microbench_valu "$@"
