B="--no-cpu-baseline --no-bruteforce-pass --lane-utilisation 0 --steps 2"
for w in cubes obiwan; do for r in 0 1; do
  timeout -k 10 300 python bench.py $B --workload $w --refill $r 2>gpurun_out/abr_${w}_$r.err > gpurun_out/abr_${w}_$r.json || exit 1
  python -c "import json;d=json.load(open('gpurun_out/abr_${w}_$r.json'));r=d['roofline'];print('$w refill $r %.0f Mrays/s first %.2f ms extend %.1f probe %.1f shade %.1f'%(d['value'],r['avg_launch_ms'],r['extend_kernels_ms'],r['probe_kernels_ms'],r['shade_kernels_ms']))"
done; done
