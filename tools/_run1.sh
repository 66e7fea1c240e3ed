set -e
mkdir -p gpurun_out/r02j
timeout -k 10 600 python -u -m pytest tests -m gpu -x -q --timeout 120 --timeout-method thread > gpurun_out/r02j/pytest.log 2>&1
tail -3 gpurun_out/r02j/pytest.log
timeout -k 10 300 python bench.py > gpurun_out/r02j/bench.json 2> gpurun_out/r02j/bench.err
timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --total-spp 256 --width 3840 --height 2160 --no-cpu-baseline --no-bruteforce-pass > gpurun_out/r02j/bench_strong4k.json 2> gpurun_out/r02j/bench_strong4k.err
bash tools/prof_bench.sh r02_cubes
