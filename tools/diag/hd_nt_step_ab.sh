set -x
mkdir -p gpurun_out/r03ah
FL="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -Wall -Wno-unused-function"
for rep in 1 2; do
  for nt in 1 0; do
    touch mipnerf360_amd/csrc/m360_linear.hip
    make -C mipnerf360_amd/csrc CXXFLAGS="$FL -DM360_HD_NT_STORES=$nt" > gpurun_out/r03ah/make_$nt.log 2>&1
    timeout 600 python bench.py --cpu-rays 0 --frame-steps 0 > gpurun_out/r03ah/bench_nt${nt}_$rep.json 2> gpurun_out/r03ah/bench_nt${nt}_$rep.err
    python -c "
import json; d=json.loads(open('gpurun_out/r03ah/bench_nt${nt}_$rep.json').read().strip().splitlines()[-1]); print('nt=$nt rep=$rep', d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'])"
  done
done
