# usage (on the GPU box, from the repository root): bash tools/ab_variants.sh <outdir> "<variants>" "<workloads>" [extra bench args]   (A/B timing of build/variants/*.so)
O=gpurun_out/$1; mkdir -p $O
V=$2; WL=$3; shift 3
B="--no-cpu-baseline --no-bruteforce-pass --lane-utilisation 0"
for w in $WL; do
  for lib in $V; do
    GPURT_LIB=build/variants/$lib.so timeout -k 10 400 python bench.py $B --workload $w "$@" 2>$O/${w}_$lib.err > $O/${w}_$lib.json || { echo "FAILED $w $lib"; tail -5 $O/${w}_$lib.err; exit 1; }
    python -c "import json;d=json.load(open('$O/${w}_$lib.json'));r=d['roofline'];s=r['launches'];p=d['ms_per_step']*d['steps']/s;print('$w $lib %.0f Mrays/s  pass %.2f ms  first %.3f ms  deeper extends %.2f probes %.2f rest %.2f ms/pass'%(d['value'],p,r['avg_launch_ms'],r['extend_kernels_ms']/s-r['avg_launch_ms'],r['probe_kernels_ms']/s, p-r['extend_kernels_ms']/s-r['probe_kernels_ms']/s))"
  done
done
