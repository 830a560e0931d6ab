#!/bin/bash
# One GPU-box session: tests, bench lines, rocprofv3 stats and counter passes.  Usage: tools/gpu_session.sh <tag> [steps...]
# Everything lands under gpurun_out/<tag>/ (merged back by gpurun); `python tools/summarize_profiles.py <tag>` then copies what is to be
# judged into profiles/<tag>/ and regenerates profiles/traffic.json.  Counter passes are their own rocprofv3 runs with --kernel-trace only
# (never with --stats / --sys-trace), the program itself behind `--` (no env / bash -c hops), from /tmp.
# (Rounds 1-5 kept ~70 one-off steps here - probes, ablations, A/B runs; their outputs are profiles/r0N/, the recipes are in git history.)
tag=${1:-r06}; shift
out=gpurun_out/$tag; mkdir -p $out
steps=${@:-tests smoke prof pmc16 pmctrain summarize bench finpmc bf16 c5 x3 x3c5 b16prof train16 c4 c4b16 gloo2}
R=$PWD; O=$R/$out
PMC="--kernel-trace --output-format csv"
stamp() { python3 -c "import bench, json; print(json.dumps({'fp32': bench.kernel_source_sha(), 'bf16': bench.kernel_source_sha(bench.TRAFFIC_SOURCES_BF16)}))" > $O/$1; }
for s in $steps; do
  case $s in
    tests)   timeout 2400 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider > $out/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $out/pytest_gpu.log; tail -5 $out/pytest_gpu.log ;;
    smoke)   timeout 600 python __graft_entry__.py smoke > $out/smoke.log 2>&1; tail -2 $out/smoke.log ;;
    summarize) # profiles/traffic.json from the counter passes of THIS call, so that the bench lines behind it carry this round's counter traffic
             cp $out/bench.json $out/bench_before_counters.json 2>/dev/null; python3 tools/summarize_profiles.py $tag > $out/summarize.log 2>&1; tail -3 $out/summarize.log ;;
    bench)   timeout 900 python bench.py > $out/bench.json 2> $out/bench.err; tail -c 3000 $out/bench.json ;;
    bf16)    timeout 900 python bench.py --mlp-dtype bf16 --cpu-rays 0 > $out/bench_bf16.json 2> $out/bench_bf16.err; tail -c 2500 $out/bench_bf16.json ;;
    c5)      timeout 900 python bench.py --config c5 > $out/bench_c5.json 2> $out/bench_c5.err; tail -c 2500 $out/bench_c5.json ;;
    x3)      timeout 900 python bench.py --mlp-dtype bf16x3 > $out/bench_bf16x3.json 2> $out/bench_bf16x3.err; tail -c 1500 $out/bench_bf16x3.json ;;
    x3c5)    timeout 900 python bench.py --config c5 --mlp-dtype bf16x3 > $out/bench_c5_bf16x3.json 2> $out/bench_c5_bf16x3.err; tail -c 900 $out/bench_c5_bf16x3.json ;;
    c4)      timeout 900 python bench.py --config c4 --steps 2 --warmup 1 > $out/bench_c4.json 2> $out/bench_c4.err; tail -c 2500 $out/bench_c4.json ;;
    c4b16)   timeout 900 python bench.py --config c4 --mlp-dtype bf16 --steps 3 --warmup 1 > $out/bench_c4_bf16.json 2> $out/bench_c4_bf16.err; tail -c 700 $out/bench_c4_bf16.json
             timeout 900 python bench.py --config c4 --mlp-dtype bf16x3 --steps 3 --warmup 1 > $out/bench_c4_bf16x3.json 2> $out/bench_c4_bf16x3.err; tail -c 700 $out/bench_c4_bf16x3.json ;;
    gloo2)   timeout 900 python bench.py --gpus 2 --backend gloo --steps 5 --warmup 2 --frame-size 400x300 > $out/bench_gloo2.json 2> $out/bench_gloo2.err; tail -c 1200 $out/bench_gloo2.json ;;
    prof)    # the headline command under rocprofv3 --stats, then FETCH / WRITE / SQ counters of its dominant kernel in three separate passes
             B="python3 $R/bench.py --steps 3 --warmup 1 --cpu-rays 0 --frame-steps 0 --no-named"
             ( cd /tmp && export TMPDIR=/tmp
               timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $R/bench.py --steps 10 --warmup 3 --cpu-rays 0 --frame-steps 0 --no-named > $O/bench_under_rocprof.log 2>&1
               timeout 600 rocprofv3 $PMC --pmc FETCH_SIZE -d $O/pmc_fetch -- $B > $O/pmc_fetch.log 2>&1
               timeout 600 rocprofv3 $PMC --pmc WRITE_SIZE -d $O/pmc_write -- $B > $O/pmc_write.log 2>&1
               timeout 600 rocprofv3 $PMC --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_INSTS_MFMA SQ_LDS_BANK_CONFLICT -d $O/pmc_sq -- $B > $O/pmc_sq.log 2>&1 )
             stamp kernel_source_sha256_at_measurement.json   # the sources these counters were measured on, stamped HERE (the summariser copies it, never recomputes it)
             find $O -name "*.db" -delete 2>/dev/null; du -sh $O; tail -2 $O/bench_under_rocprof.log | cut -c1-300 ;;
    pmc16)   # the reduced-precision workloads of the driver line, each with its own three passes: c2_bf16 (the layer chain), c5_bf16 (configs[4]'s
             # per-GPU shape), c2_bf16x3 -> profiles/traffic.json: by_workload
             for w in "c2_bf16:--mlp-dtype bf16" "c5_bf16:--config c5" "c2_bf16x3:--mlp-dtype bf16x3"; do
               k=${w%%:*}; B="python3 $R/bench.py ${w#*:} --steps 3 --warmup 1 --cpu-rays 0 --frame-steps 0"
               ( cd /tmp && export TMPDIR=/tmp
                 timeout 600 rocprofv3 $PMC --pmc FETCH_SIZE -d $O/pmc_${k}_fetch -- $B > $O/pmc_${k}_fetch.log 2>&1
                 timeout 600 rocprofv3 $PMC --pmc WRITE_SIZE -d $O/pmc_${k}_write -- $B > $O/pmc_${k}_write.log 2>&1
                 timeout 600 rocprofv3 $PMC --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_MFMA SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS -d $O/pmc_${k}_sq -- $B > $O/pmc_${k}_sq.log 2>&1 )
             done
             stamp kernel_source_sha256_at_measurement_b16.json
             find $O -name "*.db" -delete 2>/dev/null; ls -d $O/pmc_c*_sq/*/ | head -3 ;;
    pmctrain) # counter traffic of the training GEMMs (weight gradient, input gradient, ReLU mask) of one 1024 x 1024 layer on 524 288 rows, bf16 and fp32
             for w in "bf16:--mlp-dtype bf16" "fp32:"; do
               k=${w%%:*}; B="python3 $R/tools/train_step_bench.py ${w#*:} --gemm-only --iters 3"
               ( cd /tmp && export TMPDIR=/tmp
                 timeout 600 rocprofv3 $PMC --pmc FETCH_SIZE -d $O/pmc_train_${k}_fetch -- $B > $O/pmc_train_${k}_fetch.log 2>&1
                 timeout 600 rocprofv3 $PMC --pmc WRITE_SIZE -d $O/pmc_train_${k}_write -- $B > $O/pmc_train_${k}_write.log 2>&1 )
             done
             python3 -c "import bench, json; print(json.dumps({'train': bench.kernel_source_sha(bench.TRAFFIC_SOURCES_TRAIN)}))" > $O/kernel_source_sha256_at_measurement_train.json
             find $O -name "*.db" -delete 2>/dev/null; ls -d $O/pmc_train_*/*/ | head -4 ;;
    b16prof) ( cd /tmp && export TMPDIR=/tmp; timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_b16 -- python3 $R/bench.py --mlp-dtype bf16 --steps 5 --warmup 2 --cpu-rays 0 > $O/bench_b16_under_rocprof.log 2>&1
               timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_x3 -- python3 $R/bench.py --mlp-dtype bf16x3 --steps 5 --warmup 2 --cpu-rays 0 > $O/bench_x3_under_rocprof.log 2>&1 )
             find $O/stats_b16 $O/stats_x3 -name "*.db" -delete 2>/dev/null
             cp $O/stats_b16/*/*_kernel_stats.csv $O/rocprofv3_kernel_stats_bench_bf16.csv 2>/dev/null; cp $O/stats_x3/*/*_kernel_stats.csv $O/rocprofv3_kernel_stats_bench_bf16x3.csv 2>/dev/null
             head -8 $O/rocprofv3_kernel_stats_bench_bf16.csv | cut -c1-160 ;;
    finpmc)  # what the HBM-bound kernels' microseconds are made of: SQ counters of the finishers / the encoder / the prologue, fp32 and bf16 steps
             ( cd /tmp && export TMPDIR=/tmp
               timeout 600 rocprofv3 $PMC --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU GRBM_GUI_ACTIVE -d $O/pmc_fin_a -- python3 $R/bench.py --steps 3 --warmup 1 --cpu-rays 0 --frame-steps 0 --no-named > $O/pmc_fin_a.log 2>&1
               timeout 600 rocprofv3 $PMC --pmc SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM SQ_LDS_BANK_CONFLICT -d $O/pmc_fin_b -- python3 $R/bench.py --steps 3 --warmup 1 --cpu-rays 0 --frame-steps 0 --no-named > $O/pmc_fin_b.log 2>&1
               timeout 600 rocprofv3 $PMC --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU GRBM_GUI_ACTIVE -d $O/pmc_fin_a16 -- python3 $R/bench.py --mlp-dtype bf16 --steps 3 --warmup 1 --cpu-rays 0 --frame-steps 0 > $O/pmc_fin_a16.log 2>&1 )
             find $O -name "*.db" -delete 2>/dev/null
             python3 tools/summarize_profiles.py --hbm-kernels $tag | cut -c1-400 ;;
    train16) # one iteration of train.py:53-82 in bf16 and fp32 - timings, then the bf16 iteration's kernels under rocprofv3
             timeout 600 python tools/train_step_bench.py --mlp-dtype bf16 --iters 5 > $out/train_step_bf16.json 2> $out/train_step_bf16.err; cat $out/train_step_bf16.json
             timeout 600 python tools/train_step_bench.py --iters 2 > $out/train_step_fp32.json 2> $out/train_step_fp32.err; cat $out/train_step_fp32.json
             ( cd /tmp && export TMPDIR=/tmp; timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_train16 -- python3 $R/tools/train_step_bench.py --mlp-dtype bf16 --iters 5 --no-gemms > $O/train_step_bf16_under_rocprof.log 2>&1 )
             find $O/stats_train16 -name "*.db" -delete 2>/dev/null; cp $O/stats_train16/*/*_kernel_stats.csv $O/rocprofv3_kernel_stats_train_bf16.csv 2>/dev/null; head -8 $O/rocprofv3_kernel_stats_train_bf16.csv | cut -c1-160 ;;
    *) echo "unknown step $s" ;;
  esac
done
