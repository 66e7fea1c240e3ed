# usage (GPU box, repository root): bash tools/ab_flags.sh <outdir> "<workloads>" "<flag sets separated by ;>" [common bench args]
O=gpurun_out/$1; mkdir -p $O
WL=$2; SETS=$3; shift 3
B="--no-cpu-baseline --no-bruteforce-pass --lane-utilisation 0"
for w in $WL; do
  i=0
  echo "$SETS" | tr ';' '\n' | while read -r F; do
    i=$((i+1))
    timeout -k 10 400 python bench.py $B --workload $w $F "$@" 2>$O/${w}_$i.err > $O/${w}_$i.json || { echo "FAILED $w $F"; tail -5 $O/${w}_$i.err; exit 1; }
    python -c "import json;d=json.load(open('$O/${w}_$i.json'));r=d['roofline'];s=r['launches'];p=d['ms_per_step']*d['steps']/s;print('$w [$F] %.0f Mrays/s  pass %.2f ms  first %.3f ms  deeper extends %.2f probes %.2f rest %.2f ms/pass'%(d['value'],p,r['avg_launch_ms'],r['extend_kernels_ms']/s-r['avg_launch_ms'],r['probe_kernels_ms']/s, p-r['extend_kernels_ms']/s-r['probe_kernels_ms']/s))"
  done
done
