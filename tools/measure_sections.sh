# Clock split and search histograms of the pipeline's kernels, from measurement builds of the library.
#   here (build container):  bash tools/measure_sections.sh build       compiles build/variants/{clk1,clk3,histo1,histo2,histo3}.so
#   on the GPU box:          bash tools/measure_sections.sh run <out>   runs the tools and writes <out>
set -e
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -fno-gpu-flush-denormals-to-zero -fno-slp-vectorize -shared -fPIC"
SRC=gpuraytracer_amd/csrc/rt_api.hip
if [ "$1" = build ]; then
  mkdir -p build/variants
  for v in "clk1 -DRT_SECTION_CLOCKS=1" "clk3 -DRT_SECTION_CLOCKS=3" "histo1 -DRT_PROBE_HISTO=1" "histo2 -DRT_PROBE_HISTO=2" "histo3 -DRT_PROBE_HISTO=3"; do
    set -- $v
    hipcc $FLAGS $2 -o build/variants/$1.so $SRC &
  done
  wait
  exit 0
fi
OUT=$2
{
  echo "# clocks of the depth-0 kernel by section (-DRT_SECTION_CLOCKS=1, RECURSION 1, 32 spp at 1920x1080)"
  for w in cubes obiwan; do GPURT_LIB=build/variants/clk1.so timeout -k 10 200 python tools/section_clocks.py $w || exit 1; done
  echo
  echo "# clocks of the extend stage by section (-DRT_SECTION_CLOCKS=3, one pass of 512 spp)"
  for w in cubes obiwan; do GPURT_LIB=build/variants/clk3.so timeout -k 10 200 python tools/section_clocks.py $w extend | tail -n +6 || exit 1; done
  echo
  echo "# searches of the depth-0 kernel by the number of lanes they run with (-DRT_PROBE_HISTO=1 shadow, =2 glossy)"
  for w in cubes obiwan; do for k in 1 2; do GPURT_LIB=build/variants/histo$k.so timeout -k 10 200 python tools/probe_histogram.py $w 512 || exit 1; done; done
  echo
  echo "# camera-ray chunks by the length of their tile's candidate list (-DRT_PROBE_HISTO=3)"
  for w in cubes obiwan; do GPURT_LIB=build/variants/histo3.so timeout -k 10 200 python tools/probe_histogram.py $w 512 lists || exit 1; done
} > $OUT 2>gpurun_out/measure_sections.err
