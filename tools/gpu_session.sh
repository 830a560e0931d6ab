#!/bin/bash
# One GPU-box session: tests, bench lines, clock/ceiling evidence.  Usage: tools/gpu_session.sh <tag> [steps...]
# Everything lands under gpurun_out/<tag>/ (merged back by gpurun).
tag=${1:-r02}; shift
out=gpurun_out/$tag; mkdir -p $out
steps=${@:-tests bench bf16 c5 ceiling clock}
for s in $steps; do
  case $s in
    tests)   timeout 2400 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider > $out/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $out/pytest_gpu.log; tail -5 $out/pytest_gpu.log ;;
    newtests) timeout 1800 python -m pytest tests/test_gpu_configs.py -m gpu -q --tb=short -p no:cacheprovider > $out/pytest_new.log 2>&1; echo "pytest rc=$?" >> $out/pytest_new.log; tail -15 $out/pytest_new.log ;;
    bench)   timeout 900 python bench.py > $out/bench.json 2> $out/bench.err; tail -c 3000 $out/bench.json ;;
    bf16)    timeout 900 python bench.py --mlp-dtype bf16 --cpu-rays 0 > $out/bench_bf16.json 2> $out/bench_bf16.err; tail -c 2500 $out/bench_bf16.json ;;
    c5)      timeout 900 python bench.py --config c5 > $out/bench_c5.json 2> $out/bench_c5.err; tail -c 2500 $out/bench_c5.json ;;
    ceiling) timeout 300 tools/mfma_ceiling.bin > $out/mfma_ceiling.jsonl 2>&1; cat $out/mfma_ceiling.jsonl ;;
    clock)   timeout 600 python tools/linear_bench.py --dtype fp32 --clock --json $out/linear_clock.jsonl > $out/linear_fp32.log 2>&1; tail -3 $out/linear_fp32.log
             timeout 600 python tools/linear_bench.py --dtype bf16 --clock --json $out/linear_clock.jsonl > $out/linear_bf16.log 2>&1; tail -3 $out/linear_bf16.log ;;
    smoke)   timeout 600 python __graft_entry__.py smoke > $out/smoke.log 2>&1; tail -2 $out/smoke.log ;;
    *) echo "unknown step $s" ;;
  esac
done
