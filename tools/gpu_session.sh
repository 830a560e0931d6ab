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
    x3)      timeout 900 python bench.py --mlp-dtype bf16x3 > $out/bench_bf16x3.json 2> $out/bench_bf16x3.err; tail -c 1500 $out/bench_bf16x3.json ;;
    x3c5)    timeout 900 python bench.py --config c5 --mlp-dtype bf16x3 > $out/bench_c5_bf16x3.json 2> $out/bench_c5_bf16x3.err; tail -c 900 $out/bench_c5_bf16x3.json ;;
    x3prof)  R=$PWD; O=$R/$out; ( cd /tmp && export TMPDIR=/tmp; timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_x3 -- python3 $R/bench.py --mlp-dtype bf16x3 --steps 5 --warmup 2 --cpu-rays 0 > $O/bench_x3_under_rocprof.log 2>&1 ); find $O/stats_x3 -name "*.db" -delete 2>/dev/null; cp $O/stats_x3/*/*_kernel_stats.csv $O/rocprofv3_kernel_stats_bench_bf16x3.csv 2>/dev/null; head -12 $O/rocprofv3_kernel_stats_bench_bf16x3.csv | cut -c1-160 ;;
    b16prof) R=$PWD; O=$R/$out; ( cd /tmp && export TMPDIR=/tmp; timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_b16 -- python3 $R/bench.py --mlp-dtype bf16 --steps 5 --warmup 2 --cpu-rays 0 > $O/bench_b16_under_rocprof.log 2>&1 ); find $O/stats_b16 -name "*.db" -delete 2>/dev/null; cp $O/stats_b16/*/*_kernel_stats.csv $O/rocprofv3_kernel_stats_bench_bf16.csv 2>/dev/null; head -12 $O/rocprofv3_kernel_stats_bench_bf16.csv | cut -c1-160 ;;
    gloo2)   timeout 900 python bench.py --gpus 2 --backend gloo --steps 5 --warmup 2 --frame-size 400x300 > $out/bench_gloo2.json 2> $out/bench_gloo2.err; tail -c 1200 $out/bench_gloo2.json ;;
    c5)      timeout 900 python bench.py --config c5 > $out/bench_c5.json 2> $out/bench_c5.err; tail -c 2500 $out/bench_c5.json ;;
    ceiling) timeout 300 tools/mfma_ceiling.bin > $out/mfma_ceiling.jsonl 2>&1; cat $out/mfma_ceiling.jsonl ;;
    clock)   timeout 600 python tools/linear_bench.py --dtype fp32 --clock --json $out/linear_clock.jsonl > $out/linear_fp32.log 2>&1; tail -3 $out/linear_fp32.log
             timeout 600 python tools/linear_bench.py --dtype bf16 --clock --json $out/linear_clock.jsonl > $out/linear_bf16.log 2>&1; tail -3 $out/linear_bf16.log ;;
    sp16)    for v in 3 2; do timeout 600 python tools/linear_bench.py --dtype bf16 --variant $v --clock --json $out/linear_bf16_variants.jsonl > $out/linear_bf16_v$v.log 2>&1; tail -4 $out/linear_bf16_v$v.log; done
             timeout 300 python tools/linear_bench.py --dtype bf16 --variant 2 --m 65536 --n 256 --k 256 > $out/linear_bf16_v2_small.log 2>&1; tail -2 $out/linear_bf16_v2_small.log
             timeout 300 python tools/linear_bench.py --dtype bf16 --variant 2 --m 77056 --n 768 --k 192 > $out/linear_bf16_v2_odd.log 2>&1; tail -2 $out/linear_bf16_v2_odd.log ;;
    hd)      timeout 900 python tools/hd_check.py --ablate --out $out/hd_check.jsonl > $out/hd_check.log 2>&1; tail -25 $out/hd_check.log | cut -c1-400 ;;
    ablate)  for v in 4 5 6 7; do timeout 300 python tools/linear_bench.py --dtype bf16 --variant $v --no-check --rounds 5 --json $out/linear_bf16_ablate.jsonl > $out/linear_bf16_abl$v.log 2>&1; tail -1 $out/linear_bf16_abl$v.log; done ;;
    ldpad)   for pad in 0 32 64 128; do timeout 300 python tools/linear_bench.py --dtype bf16 --variant 2 --ld-pad $pad --rounds 5 --json $out/linear_bf16_ldpad.jsonl > $out/linear_bf16_pad$pad.log 2>&1; tail -1 $out/linear_bf16_pad$pad.log; done ;;
    rg16)    timeout 600 python tools/linear_bench.py --dtype bf16 --variant 9 --clock --json $out/linear_bf16_rg.jsonl > $out/linear_bf16_v9.log 2>&1; tail -4 $out/linear_bf16_v9.log
             for v in 10 11 12 13; do timeout 300 python tools/linear_bench.py --dtype bf16 --variant $v --no-check --rounds 5 --json $out/linear_bf16_rg.jsonl > $out/linear_bf16_v$v.log 2>&1; tail -1 $out/linear_bf16_v$v.log; done
             timeout 300 python tools/linear_bench.py --dtype bf16 --variant 9 --m 77056 --n 768 --k 192 > $out/linear_bf16_v9_odd.log 2>&1; tail -2 $out/linear_bf16_v9_odd.log
             timeout 300 python tools/linear_bench.py --dtype bf16 --variant 9 --m 65536 --n 256 --k 128 > $out/linear_bf16_v9_k128.log 2>&1; tail -2 $out/linear_bf16_v9_k128.log ;;
    prof)    R=$PWD; O=$R/$out; PMC="--kernel-trace --output-format csv"; B="python3 $R/bench.py --steps 3 --warmup 1 --cpu-rays 0 --frame-steps 0 --no-named"
             ( cd /tmp && export TMPDIR=/tmp
               timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $R/bench.py --steps 10 --warmup 3 --cpu-rays 0 --frame-steps 0 --no-named > $O/bench_under_rocprof.log 2>&1
               timeout 600 rocprofv3 $PMC --pmc FETCH_SIZE -d $O/pmc_fetch -- $B > $O/pmc_fetch.log 2>&1
               timeout 600 rocprofv3 $PMC --pmc WRITE_SIZE -d $O/pmc_write -- $B > $O/pmc_write.log 2>&1
               timeout 600 rocprofv3 $PMC --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_INSTS_MFMA SQ_LDS_BANK_CONFLICT -d $O/pmc_sq -- $B > $O/pmc_sq.log 2>&1 )
             # the sources these counters were measured on, stamped HERE (the summariser copies it, it never recomputes it)
             python3 -c "import bench, json; print(json.dumps({'fp32': bench.kernel_source_sha(), 'bf16': bench.kernel_source_sha(bench.TRAFFIC_SOURCES_BF16)}))" > $O/kernel_source_sha256_at_measurement.json
             # raw traces are large: keep the csv files the summariser needs, drop the rest
             find $O -name "*.db" -delete 2>/dev/null; du -sh $O; tail -2 $O/bench_under_rocprof.log | cut -c1-300 ;;
    finpmc)  # round 5: what the HBM-bound kernels' microseconds are made of - SQ counters of the finishers / the encoder / the prologue, fp32 and bf16 steps
             R=$PWD; O=$R/$out; PMC="--kernel-trace --output-format csv"
             ( cd /tmp && export TMPDIR=/tmp
               timeout 600 rocprofv3 $PMC --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU GRBM_GUI_ACTIVE -d $O/pmc_fin_a -- python3 $R/bench.py --steps 3 --warmup 1 --cpu-rays 0 --frame-steps 0 --no-named > $O/pmc_fin_a.log 2>&1
               timeout 600 rocprofv3 $PMC --pmc SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM SQ_LDS_BANK_CONFLICT -d $O/pmc_fin_b -- python3 $R/bench.py --steps 3 --warmup 1 --cpu-rays 0 --frame-steps 0 --no-named > $O/pmc_fin_b.log 2>&1
               timeout 600 rocprofv3 $PMC --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU GRBM_GUI_ACTIVE -d $O/pmc_fin_a16 -- python3 $R/bench.py --mlp-dtype bf16 --steps 3 --warmup 1 --cpu-rays 0 --frame-steps 0 > $O/pmc_fin_a16.log 2>&1 )
             find $O -name "*.db" -delete 2>/dev/null
             python3 tools/summarize_profiles.py --hbm-kernels $tag | cut -c1-400 ;;
    train16) # round 5: the bf16 training iteration - timings, then its kernels under rocprofv3 (per-kernel time of the iteration alone)
             timeout 600 python tools/train_step_bench.py --mlp-dtype bf16 --iters 5 > $out/train_step_bf16.json 2> $out/train_step_bf16.err; cat $out/train_step_bf16.json
             timeout 600 python tools/train_step_bench.py --iters 2 > $out/train_step_fp32.json 2> $out/train_step_fp32.err; cat $out/train_step_fp32.json
             R=$PWD; O=$R/$out; ( cd /tmp && export TMPDIR=/tmp; timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_train16 -- python3 $R/tools/train_step_bench.py --mlp-dtype bf16 --iters 5 --no-gemms > $O/train_step_bf16_under_rocprof.log 2>&1 ); find $O/stats_train16 -name "*.db" -delete 2>/dev/null; cp $O/stats_train16/*/*_kernel_stats.csv $O/rocprofv3_kernel_stats_train_bf16.csv 2>/dev/null; head -24 $O/rocprofv3_kernel_stats_train_bf16.csv | cut -c1-200 ;;
    r3new)   timeout 2400 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -k "span or one_chunk or bench_ or two_ranks or c2_full or rccl or g18 or nan_rows" > $out/pytest_r3new.log 2>&1; echo "pytest rc=$?" >> $out/pytest_r3new.log; tail -25 $out/pytest_r3new.log ;;
    nccl2)   # two RCCL ranks on a 1-GPU box: must fail with RCCL's own error (not a SystemExit of bench.py), and must not hang
             timeout 300 python bench.py --gpus 2 --steps 2 --warmup 1 --frame-steps 0 > $out/nccl2.out 2> $out/nccl2.err; echo "rc=$?" >> $out/nccl2.out; tail -3 $out/nccl2.out; grep -i "nccl\|rccl\|duplicate\|error" $out/nccl2.err | head -12 ;;
    c4)      timeout 900 python bench.py --config c4 --steps 2 --warmup 1 > $out/bench_c4.json 2> $out/bench_c4.err; tail -c 2500 $out/bench_c4.json ;;
    lintests) timeout 1800 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -k "linear or soak or bit_identical or one_chunk or g8 or g7 or fused" > $out/pytest_linear.log 2>&1; echo "pytest rc=$?" >> $out/pytest_linear.log; tail -6 $out/pytest_linear.log ;;
    wide)    timeout 300 tools/wide_wave_probe.bin > $out/bf16_wide_wave_probe.jsonl 2>&1; cat $out/bf16_wide_wave_probe.jsonl | cut -c1-330
             timeout 300 tools/loader_wave_probe.bin > $out/bf16_loader_wave_probe_same_box.jsonl 2>&1; grep '"mode": [14]' $out/bf16_loader_wave_probe_same_box.jsonl | cut -c1-330 ;;
    k64)     for n in 1024 256; do m=$((n == 1024 ? 524288 : 262144))
               timeout 300 python tools/linear_bench.py --dtype bf16 --m $m --n $n --k 64 --rounds 5 --json $out/linear_bf16_k64.jsonl > $out/linear_bf16_k64_prod_$n.log 2>&1; tail -2 $out/linear_bf16_k64_prod_$n.log
               timeout 300 python tools/linear_bench.py --dtype bf16 --variant 3 --m $m --n $n --k 64 --rounds 5 --json $out/linear_bf16_k64.jsonl > $out/linear_bf16_k64_pp_$n.log 2>&1; tail -2 $out/linear_bf16_k64_pp_$n.log; done ;;
    w16)     timeout 300 python tools/linear_bench.py --dtype bf16 --variant 200 --rounds 5 --json $out/linear_bf16_w16.jsonl > $out/linear_bf16_w16_200.log 2>&1; tail -3 $out/linear_bf16_w16_200.log
             timeout 300 python tools/linear_bench.py --dtype bf16 --variant 100 --rounds 5 --json $out/linear_bf16_w16.jsonl > $out/linear_bf16_w16_100.log 2>&1; tail -2 $out/linear_bf16_w16_100.log
             for v in 101 102 104 107 116 132 139; do timeout 300 python tools/linear_bench.py --dtype bf16 --variant $v --no-check --rounds 5 --json $out/linear_bf16_w16.jsonl > $out/linear_bf16_w16_$v.log 2>&1; tail -2 $out/linear_bf16_w16_$v.log; done
             timeout 300 python tools/linear_bench.py --dtype bf16 --variant 3 --rounds 5 --json $out/linear_bf16_w16.jsonl > $out/linear_bf16_pp_same_box.log 2>&1; tail -1 $out/linear_bf16_pp_same_box.log
             timeout 300 python tools/linear_bench.py --dtype bf16 --variant 200 --m 65536 --n 256 --k 1024 --json $out/linear_bf16_w16.jsonl > $out/linear_bf16_w16_small.log 2>&1; tail -2 $out/linear_bf16_w16_small.log
             timeout 300 python tools/linear_bench.py --dtype bf16 --variant 200 --m 77056 --n 768 --k 1152 --json $out/linear_bf16_w16.jsonl > $out/linear_bf16_w16_odd.log 2>&1; tail -2 $out/linear_bf16_w16_odd.log ;;
    w16dbg)  timeout 300 python tools/diag/w16_debug.py --k 1024 > $out/w16_debug_small.log 2>&1; cat $out/w16_debug_small.log | grep -v amdgpu.ids
             timeout 300 python tools/diag/w16_debug.py --m 4096 --n 256 --k 1024 > $out/w16_debug_tiny.log 2>&1; cat $out/w16_debug_tiny.log | grep -v amdgpu.ids
             timeout 300 python tools/diag/w16_debug.py --m 77056 --n 768 --k 1152 --reps 2 > $out/w16_debug_odd.log 2>&1; cat $out/w16_debug_odd.log | grep -v amdgpu.ids ;;
    k256)    for v in 3 200; do timeout 300 python tools/linear_bench.py --dtype bf16 --variant $v --m 524288 --n 256 --k 256 --rounds 5 --json $out/linear_bf16_k256.jsonl > $out/linear_bf16_k256_$v.log 2>&1; tail -1 $out/linear_bf16_k256_$v.log; done ;;
    b16tests) timeout 1800 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -k "bf16 or c5" > $out/pytest_bf16.log 2>&1; echo "pytest rc=$?" >> $out/pytest_bf16.log; tail -6 $out/pytest_bf16.log ;;
    x3tests) timeout 1800 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -x -k "x3" > $out/pytest_x3.log 2>&1; echo "pytest rc=$?" >> $out/pytest_x3.log; tail -12 $out/pytest_x3.log ;;
    b16pmc)  R=$PWD; O=$R/$out; PMC="--kernel-trace --output-format csv"; B="python3 $R/bench.py --mlp-dtype bf16 --steps 3 --warmup 1 --cpu-rays 0 --frame-steps 0"
             ( cd /tmp && export TMPDIR=/tmp
               timeout 600 rocprofv3 $PMC --pmc FETCH_SIZE -d $O/pmc_b16_fetch -- $B > $O/pmc_b16_fetch.log 2>&1
               timeout 600 rocprofv3 $PMC --pmc WRITE_SIZE -d $O/pmc_b16_write -- $B > $O/pmc_b16_write.log 2>&1
               timeout 600 rocprofv3 $PMC --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_MFMA SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS -d $O/pmc_b16_sq -- $B > $O/pmc_b16_sq.log 2>&1 )
             python3 -c "import bench, json; print(json.dumps({'fp32': bench.kernel_source_sha(), 'bf16': bench.kernel_source_sha(bench.TRAFFIC_SOURCES_BF16)}))" > $O/kernel_source_sha256_at_measurement_b16.json
             find $O -name "*.db" -delete 2>/dev/null; ls $O/pmc_b16_sq/*/ | head -3 ;;
    c4b16)   timeout 900 python bench.py --config c4 --mlp-dtype bf16 --steps 3 --warmup 1 > $out/bench_c4_bf16.json 2> $out/bench_c4_bf16.err; tail -c 700 $out/bench_c4_bf16.json
             timeout 900 python bench.py --config c4 --mlp-dtype bf16x3 --steps 3 --warmup 1 > $out/bench_c4_bf16x3.json 2> $out/bench_c4_bf16x3.err; tail -c 700 $out/bench_c4_bf16x3.json ;;
    r4new)   timeout 2400 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -s -k "g19 or x6 or psnr or structured or bf16" > $out/pytest_r4new.log 2>&1; echo "pytest rc=$?" >> $out/pytest_r4new.log; grep -a "^G19\|^c2 structured\|passed\|failed\|FAILED\|Error" $out/pytest_r4new.log | tail -60 ;;
    r4fin)   timeout 1800 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -k "finish or forward_vs_oracle or g7 or g9 or g14 or one_chunk or grouped or fused or c5_full_size_fp32 or x6 or checkpoint" > $out/pytest_r4fin.log 2>&1; echo "pytest rc=$?" >> $out/pytest_r4fin.log; tail -8 $out/pytest_r4fin.log | cut -c1-300 ;;
    nowait)  for v in 100 116 131 133 134 135 100; do timeout 300 python tools/linear_bench.py --dtype bf16 --variant $v --no-check --rounds 5 --json $out/bf16_w16_no_wait_ablation.jsonl > $out/linear_bf16_w16_$v.log 2>&1; tail -2 $out/linear_bf16_w16_$v.log | cut -c1-400; done ;;
    k384)    for v in 100 128 100 128; do timeout 300 python tools/linear_bench.py --dtype bf16 --variant $v --k 384 --no-check --rounds 5 --json $out/bf16_w16_k384.jsonl > $out/linear_bf16_k384_$v.log 2>&1; tail -1 $out/linear_bf16_k384_$v.log | cut -c1-400; done
             for v in 100 116; do timeout 300 python tools/linear_bench.py --dtype bf16 --variant $v --k 384 --n 256 --no-check --rounds 5 --json $out/bf16_w16_k384.jsonl > $out/linear_bf16_k384_n256_$v.log 2>&1; tail -1 $out/linear_bf16_k384_n256_$v.log | cut -c1-400; done ;;
    g20)     timeout 900 python tools/train_demo.py --steps 500 --rays 1024 --samples 32 --hidden 32 64 --lr 3e-3 --kind lego --teacher structured --white-bkgd --save $out/g20_trained_student.pt > $out/g20_train.json 2> $out/g20_train.err; cat $out/g20_train.json | cut -c1-1500 ;;
    ldsepi)  for v in 136 100 136 100 137 116; do timeout 300 python tools/linear_bench.py --dtype bf16 --variant $v --rounds 5 --json $out/bf16_w16_lds_epilogue.jsonl $( [ $v = 137 -o $v = 116 ] && echo --no-check ) > $out/linear_bf16_ldsepi_$v.log 2>&1; tail -2 $out/linear_bf16_ldsepi_$v.log | cut -c1-300; done
             for v in 136 100; do timeout 300 python tools/linear_bench.py --dtype bf16 --variant $v --k 384 --rounds 5 --json $out/bf16_w16_lds_epilogue.jsonl > $out/linear_bf16_ldsepi_k384_$v.log 2>&1; tail -1 $out/linear_bf16_ldsepi_k384_$v.log | cut -c1-300; done ;;
    r4nan)   timeout 1800 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -k "g18 or nan or bf16 or g19 or g20 or finish" > $out/pytest_r4nan.log 2>&1; echo "pytest rc=$?" >> $out/pytest_r4nan.log; grep -a "^FAILED\|passed\|failed" $out/pytest_r4nan.log | tail -12 ;;
    valu)    /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 tools/valu_issue_probe.hip -o /tmp/valu_issue_probe.bin 2> $out/valu_issue_probe.err && timeout 300 /tmp/valu_issue_probe.bin > $out/valu_issue_probe.jsonl 2>> $out/valu_issue_probe.err; cut -c1-260 $out/valu_issue_probe.jsonl ;;
    epi)     for v in 100 116 100; do timeout 300 python tools/linear_bench.py --dtype bf16 --variant $v $( [ $v = 116 ] && echo --no-check ) --rounds 5 --json $out/bf16_w16_epilogue.jsonl > $out/linear_bf16_epi_$v.log 2>&1; tail -2 $out/linear_bf16_epi_$v.log | cut -c1-400; done ;;
    pairtests) timeout 2400 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -k "paired or soak or bit_identical or g19 or x6 or bf16 or c5 or fused" > $out/pytest_paired.log 2>&1; echo "pytest rc=$?" >> $out/pytest_paired.log; tail -12 $out/pytest_paired.log | cut -c1-300 ;;
    pairab)  for r in 1 2; do for f in "" "--plain-rows"; do for d in bf16 bf16x3; do timeout 600 python bench.py --mlp-dtype $d --cpu-rays 0 --frame-steps 0 $f >> $out/paired_rows_ab.jsonl 2>> $out/paired_rows_ab.err; done; done; done
             python3 - <<PYEOF
import json
for l in open("$out/paired_rows_ab.jsonl"):
    d = json.loads(l); print(d["dtype"], "plain" if d["config"].get("plain_rows") else "paired", d["ms_per_step"], d["ms_per_step_median"], d["roofline"]["avg_launch_ms"])
PYEOF
             ;;
    epipair) for v in 140 100 141 116 140 100; do timeout 300 python tools/linear_bench.py --dtype bf16 --variant $v $( [ $v != 100 ] && echo --no-check ) --rounds 5 --json $out/bf16_w16_paired_rows.jsonl > $out/linear_bf16_pair_$v.log 2>&1; tail -1 $out/linear_bf16_pair_$v.log | cut -c1-400; done ;;
    stpol)   for v in 100 142 143 144 145 100; do timeout 300 python tools/linear_bench.py --dtype bf16 --variant $v --rounds 5 --json $out/bf16_w16_store_policy.jsonl > $out/linear_bf16_stpol_$v.log 2>&1; tail -1 $out/linear_bf16_stpol_$v.log | cut -c1-400; done ;;
    chain256) timeout 600 python tools/mlp_chain_bench.py --dtype bf16 --width 256 --layers 3 --blocks 0,262144,131072,65536,32768 --rounds 7 --json $out/mlp_chain_row_blocks_w256.jsonl > $out/mlp_chain_w256.log 2>&1; cut -c1-300 $out/mlp_chain_w256.log | grep -v amdgpu.ids ;;
    storeonly) for v in 146 140 141 146; do timeout 300 python tools/linear_bench.py --dtype bf16 --variant $v --no-check --rounds 5 --json $out/bf16_w16_store_only_epilogue.jsonl > $out/linear_bf16_so_$v.log 2>&1; tail -1 $out/linear_bf16_so_$v.log | cut -c1-400; done ;;
    ldpad16) for pad in 0 64 128 32 0; do for v in 146 140; do timeout 300 python tools/linear_bench.py --dtype bf16 --variant $v --ld-pad $pad --no-check --rounds 5 --json $out/bf16_w16_ld_pad.jsonl > $out/linear_bf16_ldpad_${v}_$pad.log 2>&1; tail -1 $out/linear_bf16_ldpad_${v}_$pad.log | cut -c1-330; done; done ;;
    soak)    timeout 900 python tools/diag/w16_soak.py --shapes 60 > $out/w16_soak.log 2>&1; tail -2 $out/w16_soak.log
             timeout 900 python tools/diag/w16_soak.py --shapes 40 --special --seed 7 > $out/w16_soak_special.log 2>&1; tail -2 $out/w16_soak_special.log
             timeout 900 python tools/diag/w16_soak.py --shapes 40 --x3 --seed 3 > $out/w16_soak_x3.log 2>&1; tail -2 $out/w16_soak_x3.log ;;
    stagger) for st in 0 3 6 0 6 12; do M360_DIAG_STAGGER=$st timeout 300 python tools/linear_bench.py --dtype bf16 --variant 140 --no-check --rounds 7 > $out/linear_bf16_stagger_$st.log 2>&1; echo "stagger $st: $(tail -1 $out/linear_bf16_stagger_$st.log | cut -c1-330)" | tee -a $out/bf16_w16_start_stagger.txt; done ;;
    storeonly2) for v in 146 148 146 148; do timeout 300 python tools/linear_bench.py --dtype bf16 --variant $v --no-check --rounds 5 --json $out/bf16_w16_store_only_plain_vs_nt.jsonl > $out/linear_bf16_so2_$v.log 2>&1; tail -1 $out/linear_bf16_so2_$v.log | cut -c1-400; done ;;
    chainplain) for lib in libm360.so libm360_plainstores.so libm360.so libm360_plainstores.so; do echo "== $lib" | tee -a $out/mlp_chain_plain_vs_nt.log; M360_LIB=$PWD/mipnerf360_amd/$lib timeout 600 python tools/mlp_chain_bench.py --dtype bf16 --width 1024 --layers 6 --blocks 0,131072,65536,32768 --rounds 5 2>&1 | grep rows_per_block | cut -c1-300 | tee -a $out/mlp_chain_plain_vs_nt.log; done ;;
    chainplain2) for mode in "" "--reuse"; do for lib in libm360_plainstores.so libm360.so; do echo "== $lib $mode" | tee -a $out/mlp_chain_plain_vs_nt_blocks.log; M360_LIB=$PWD/mipnerf360_amd/$lib timeout 600 python tools/mlp_chain_bench.py --dtype bf16 --width 1024 --layers 6 --blocks 0,131072,98304,65536,49152,32768 --rounds 5 $mode 2>&1 | grep rows_per_block | cut -c1-330 | tee -a $out/mlp_chain_plain_vs_nt_blocks.log; done; done ;;
    chainplain3) for lib in libm360_plainstores.so libm360.so; do echo "== $lib --reuse width 256" | tee -a $out/mlp_chain_plain_vs_nt_blocks.log; M360_LIB=$PWD/mipnerf360_amd/$lib timeout 600 python tools/mlp_chain_bench.py --dtype bf16 --width 256 --layers 3 --blocks 0,262144,131072,65536 --rounds 7 --reuse 2>&1 | grep rows_per_block | cut -c1-330 | tee -a $out/mlp_chain_plain_vs_nt_blocks.log; done ;;
    rbtests) timeout 2400 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -k "row_blocks or temporal or paired or g19 or bf16 or c5 or fused or soak" > $out/pytest_rowblocks.log 2>&1; echo "pytest rc=$?" >> $out/pytest_rowblocks.log; tail -12 $out/pytest_rowblocks.log | cut -c1-300 ;;
    rbab)    for cfg in "--row-blocks 0" "--row-blocks 49152 --row-block-streams 1" "--row-blocks 24576 --row-block-streams 2" "--row-blocks 49152 --row-block-streams 2" "--row-blocks 32768 --row-block-streams 2" "--row-blocks 65536 --row-block-streams 2" "--row-blocks 0" "--row-blocks 24576 --row-block-streams 2"; do timeout 600 python bench.py --mlp-dtype bf16 --cpu-rays 0 --frame-steps 0 $cfg >> $out/row_blocks_ab.jsonl 2>> $out/row_blocks_ab.err; done
             python3 - <<PYEOF
import json
for l in open("$out/row_blocks_ab.jsonl"):
    d = json.loads(l); c = d["config"]; print(c.get("row_blocks"), c.get("row_block_streams"), d["ms_per_step"], d["ms_per_step_median"], d["roofline"]["avg_launch_ms"], d["roofline"]["frac"], d["parity"] if "parity" in d else "")
PYEOF
             ;;
    rbab2)   timeout 900 python tools/diag/row_blocks_ab.py > $out/row_blocks_ab_no_recorder.jsonl 2> $out/row_blocks_ab2.err; cat $out/row_blocks_ab_no_recorder.jsonl
             timeout 900 python tools/diag/row_blocks_ab.py --c5 --iters 20 --configs 0:1,24576:2,49152:1,32768:2,0:1 > $out/row_blocks_ab_no_recorder_c5.jsonl 2>> $out/row_blocks_ab2.err; cat $out/row_blocks_ab_no_recorder_c5.jsonl ;;
    rbtrace) R=$PWD; O=$R/$out; for cfg in 0:1 49152:1 24576:2; do ( cd /tmp && export TMPDIR=/tmp; timeout 600 rocprofv3 --kernel-trace --output-format csv -d $O/rbtrace_${cfg/:/_} -- python3 $R/tools/diag/row_blocks_ab.py --iters 3 --configs $cfg > $O/rbtrace_${cfg/:/_}.log 2>&1 ); done
             python3 - <<PYEOF
import csv, glob, os
for cfg in ("0_1", "49152_1", "24576_2"):
    fs = glob.glob("$out/rbtrace_" + cfg + "/*/*_kernel_trace.csv")
    if not fs: print(cfg, "no trace"); continue
    rows = sorted(csv.DictReader(open(max(fs, key=os.path.getmtime))), key=lambda r: int(r["Start_Timestamp"]))
    # the last forward: from the last stage_prologue with a preceding gap back to ... take the final N kernels between the last two 'sample/prologue' of the prop stage
    idx = [i for i, r in enumerate(rows) if "stage_prologue" in r["Kernel_Name"]]
    a, b = idx[-2], len(rows)     # the last forward starts at the second-to-last prologue (prop stage), ends at the end
    fw = rows[a:b]
    t0, t1 = int(fw[0]["Start_Timestamp"]), max(int(r["End_Timestamp"]) for r in fw)
    busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in fw)
    # union of intervals (two streams overlap)
    iv = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in fw)
    u, cs, ce = 0, iv[0][0], iv[0][1]
    for s_, e_ in iv[1:]:
        if s_ > ce: u += ce - cs; cs, ce = s_, e_
        else: ce = max(ce, e_)
    u += ce - cs
    print(cfg, "launches", len(fw), "span_us", (t1 - t0) / 1e3, "sum_of_kernels_us", busy / 1e3, "union_busy_us", u / 1e3, "idle_us", (t1 - t0 - u) / 1e3)
PYEOF
             ;;
    xcc)     /opt/rocm/bin/hipcc -O2 --offload-arch=gfx950 tools/xcc_probe.hip -o /tmp/xcc_probe.bin 2> $out/xcc_probe.err && timeout 120 /tmp/xcc_probe.bin > $out/xcc_probe.txt 2>> $out/xcc_probe.err; cat $out/xcc_probe.txt ;;
    chaintest) timeout 900 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -x -k "hidden_layer_chain or hidden_chain" > $out/pytest_chain.log 2>&1; echo "pytest rc=$?" >> $out/pytest_chain.log; tail -25 $out/pytest_chain.log | cut -c1-300 ;;
    chainbench) timeout 600 python tools/diag/chain_bench.py > $out/chain_bench.jsonl 2> $out/chain_bench.err; cat $out/chain_bench.jsonl; tail -3 $out/chain_bench.err
             timeout 600 python tools/diag/chain_bench.py --c5 --rounds 5 >> $out/chain_bench.jsonl 2>> $out/chain_bench.err; tail -2 $out/chain_bench.jsonl ;;
    chainab) for r in 1 2; do for f in "" "--no-chain"; do timeout 600 python bench.py --mlp-dtype bf16 --cpu-rays 0 --frame-steps 0 $f >> $out/hidden_chain_ab.jsonl 2>> $out/hidden_chain_ab.err; done; done
             timeout 600 python bench.py --config c5 --cpu-rays 0 --frame-steps 0 >> $out/hidden_chain_ab.jsonl 2>> $out/hidden_chain_ab.err; timeout 600 python bench.py --config c5 --cpu-rays 0 --frame-steps 0 --no-chain >> $out/hidden_chain_ab.jsonl 2>> $out/hidden_chain_ab.err
             python3 - <<PYEOF
import json
for l in open("$out/hidden_chain_ab.jsonl"):
    d = json.loads(l); c = d["config"]; r = d["roofline"] or {}; print(c.get("name"), "six launches" if c.get("no_chain") else "chain", d["ms_per_step"], d["ms_per_step_median"], r.get("avg_launch_ms"), r.get("frac"), d.get("parity", {}).get("max_abs_rgb") if isinstance(d.get("parity"), dict) else "")
PYEOF
             ;;
    chainbench256) timeout 600 python tools/diag/chain_bench.py --width 256 --layers 2 > $out/chain_bench_w256.jsonl 2> $out/chain_bench_w256.err; cat $out/chain_bench_w256.jsonl; tail -2 $out/chain_bench_w256.err ;;
    twotenants) timeout 1200 python tools/diag/chain_two_processes.py > $out/chain_two_processes.txt 2>&1; cat $out/chain_two_processes.txt | cut -c1-300 ;;
    chainsoak) timeout 900 python tools/diag/chain_soak.py --reps 300 > $out/chain_soak.log 2>&1; tail -3 $out/chain_soak.log
             timeout 900 python tools/diag/chain_soak.py --reps 60 --rows 2097152 >> $out/chain_soak.log 2>&1; tail -2 $out/chain_soak.log ;;
    smoke)   timeout 600 python __graft_entry__.py smoke > $out/smoke.log 2>&1; tail -2 $out/smoke.log ;;
    *) echo "unknown step $s" ;;
  esac
done
