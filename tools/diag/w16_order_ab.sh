# A/B of a generator flag of the ring kernel on ONE box: rebuilds libm360_diag.so both ways and alternates tools/linear_bench.py
# usage: bash tools/diag/w16_order_ab.sh FLAG   (FLAG: a module-level boolean of tools/gen_w16_slab.py, e.g. JB_OUTER)
flag=${1:-JB_OUTER}; out=gpurun_out/w16_ab_$flag; mkdir -p $out
for rep in 1 2; do
  for val in True False; do
    sed -i "s/^$flag = \(True\|False\)/$flag = $val/" tools/gen_w16_slab.py
    python tools/gen_w16_slab.py > /dev/null
    make -C mipnerf360_amd/csrc diag > $out/make.log 2>&1
    timeout 200 python tools/linear_bench.py --dtype bf16 --variant 100 --no-check --rounds 7 > $out/run_${val}_$rep.log 2>&1
    echo "$flag=$val rep=$rep $(tail -1 $out/run_${val}_$rep.log | cut -c1-330)"
  done
done
