# Round-6 GPU runs (outputs under gpurun_out/r06/, summaries copied into profiles/ by hand afterwards).
#   bash tools/profile_round6.sh [part ...]    parts: quick attn suite bench bench_others prof pmc rig tp two_chain tpprobe t1
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r06
mkdir -p $O
cd $R
export GPU_MAX_HW_QUEUES=24
PARTS=${@:-suite bench prof pmc}
has() { case " $PARTS " in *" $1 "*) return 0;; esac; return 1; }
if has quick; then
  (timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_tp.py -q -x -k "attention or sdpa or gemm_configurations or single_rank" ) > $O/quick1.log 2>&1; echo "quick1 rc=$?"; tail -n 4 $O/quick1.log
  (timeout 900 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_cache.py -q -x -k "batch_invariance or window or cache" ) > $O/quick2.log 2>&1; echo "quick2 rc=$?"; tail -n 4 $O/quick2.log
fi
if has tp; then
  (timeout 900 python -m pytest tests/test_gpu_tp.py -q -x) > $O/tp_tests.log 2>&1; echo "tp tests rc=$?"; tail -n 4 $O/tp_tests.log
  timeout 600 python tools/tp_overlap_probe.py --cus 16,32 > $O/tp_overlap_probe.txt 2> $O/tp_overlap_probe.err; echo "overlap probe rc=$?"; cat $O/tp_overlap_probe.txt | cut -c1-900; tail -3 $O/tp_overlap_probe.err
fi
if has cumask; then
  timeout 300 python tools/cu_mask_probe.py > $O/cu_mask_probe.txt 2> $O/cu_mask_probe.err; echo "cumask rc=$?"; tail -n 5 $O/cu_mask_probe.txt; tail -2 $O/cu_mask_probe.err
fi
if has ab2; then
  timeout 600 python tools/block_ab.py --layers 2 --rounds 5 "gemm_tile_order=0" "gemm_tile_order=9904" > $O/block_ab2.txt 2> $O/block_ab2.err; echo "ab2 rc=$?"; cat $O/block_ab2.txt; tail -3 $O/block_ab2.err
fi
if has ab; then
  timeout 600 python tools/block_ab.py --layers 2 --rounds 5 "gemm_tile_order=9904" "gemm_tile_order=408" "gemm_tile_order=216" "gemm_tile_order=804" > $O/block_ab.txt 2> $O/block_ab.err; echo "ab rc=$?"; cat $O/block_ab.txt
fi
if has fetch; then
  cd /tmp && export TMPDIR=/tmp
  timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_order -o t -- python $R/tools/tile_order_fetch.py > /dev/null 2> $O/pmc_order.err; echo "fetch rc=$?"
  cd $R
  python tools/tile_order_fetch.py --summarise $(ls $O/pmc_order/*/*counter_collection.csv $O/pmc_order/*counter_collection.csv 2>/dev/null | head -1) > $O/tile_order_fetch.txt 2>&1; cat $O/tile_order_fetch.txt
  rm -rf $O/pmc_order
fi
if has peaked; then
  (timeout 1200 python -m pytest tests/test_gpu_peaked.py -q -x -s -k "not flat_weights") > $O/peaked.log 2>&1; echo "peaked rc=$?"; grep -E "passed|failed|error|peaked checkpoint|M layout|M variant" $O/peaked.log | cut -c1-600
fi
if has attn; then
  # the attention kernel alone: SQ counters of both forms at B = 1 / 2 (separate --pmc passes) and kernel time against L
  bash tools/attn16_pmc.sh "1,0" > $O/attn_pmc.txt 2>&1; echo "attn pmc rc=$?"; grep -c attn16 $O/attn_pmc.txt
  cd /tmp && export TMPDIR=/tmp
  for L in 640 1280 2438; do
    timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/attn_L$L -o a -- python $R/tools/attn16_dev.py --time --batch 1 --L $L --forms 1 > /dev/null 2>&1
    echo "L=$L $(grep attn16 $O/attn_L$L/*kernel_stats.csv | cut -d, -f2-7)"; rm -rf $O/attn_L$L
  done > $O/attn_time_vs_L.txt 2>&1
  for B in 1 2; do
    timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/attn_B$B -o a -- python $R/tools/attn16_dev.py --time --batch $B --forms 1 > /dev/null 2>&1
    echo "B=$B L=2438 form 1: $(grep attn16 $O/attn_B$B/*kernel_stats.csv | cut -d, -f2-7)  (calls, total ns, avg ns, %, min ns, max ns)"; rm -rf $O/attn_B$B
  done >> $O/attn_time_vs_L.txt 2>&1
  cd $R; cat $O/attn_time_vs_L.txt
  timeout 600 python tools/block_ab.py "attention_form=1" "attention_form=0" 2>&1 | grep "^B=" > $O/block_ab_attention.txt; cat $O/block_ab_attention.txt
fi
if has two_chain; then
  { timeout 300 python tools/two_chain_probe.py --prio 1 2>&1 | grep "^L="; timeout 300 python tools/two_chain_probe.py --prio 0 2>&1 | grep "^L="; } > $O/two_chain_probe.txt; cat $O/two_chain_probe.txt
fi
if has tpprobe; then
  timeout 900 python tools/tp_compute_probe.py 2>&1 | grep "^TP=\|lane 0" > $O/tp_compute_probe.txt; cat $O/tp_compute_probe.txt
fi
if has t1; then
  timeout 600 python bench.py --steps 4 --warmup 1 --no-cpu-baseline > $O/bench_config1_same_box_T0.json 2> $O/bench_t0.err; echo "t0 rc=$?"
  timeout 600 python bench.py --steps 4 --warmup 1 --no-cpu-baseline --temperature 1.0 > $O/bench_config1_readme_sampling_T1.json 2> $O/bench_t1.err; echo "t1 rc=$?"
fi
if has suite; then (timeout 1800 python -m pytest tests -q -m gpu) > $O/pytest_gpu.log 2>&1; echo "suite rc=$?"; tail -n 3 $O/pytest_gpu.log; fi
if has bench; then
  timeout 600 python bench.py --steps 4 --warmup 1 > $O/bench_config1.json 2> $O/bench_config1.err; echo "bench1 rc=$?"; tail -c 1500 $O/bench_config1.json
fi
if has bench_others; then
  timeout 300 python bench.py --config 0 --steps 4 --warmup 1 --no-cpu-baseline > $O/bench_config0.json 2> $O/bench_config0.err; echo "bench0 rc=$?"
  timeout 600 python bench.py --config 3 --steps 1 --warmup 1 --no-probe --no-cpu-baseline > $O/bench_config3.json 2> $O/bench_config3.err; echo "bench3 rc=$?"
  timeout 900 python bench.py --config 4 --graph on --steps 1 --warmup 0 --no-probe --no-cpu-baseline > $O/bench_config4.json 2> $O/bench_config4.err; echo "bench4 rc=$?"
fi
if has sweep; then
  # the 32-point table the planner's constants are held against (tests/test_host_logic.py), taken with the FINAL library's
  # -DMMADA_TUNE build (tools/build_tune.py: the product sources, no fork)
  V=100,300,301,302,303,1320,1256
  { echo "# tools/gemm_sweep.py on the round-6 library (-DMMADA_TUNE build of mmada_parallel_amd/csrc: tail K-tiles in a one-trip loop, 4x8 tile order for 320x256), cold operands (6 rotating copies), random bf16, STORE epilogue.  v100 = the planner's pick; v300-303 = 8-phase 320x256 / 256x256 / 160x256 / 320x128; v1320 / v1256 = the 16-wave kernel with BM 320 / 256.  TFLOP/s, median of 3 rounds.";
    MMADA_TUNE_PREBUILT=1 timeout 400 python tools/gemm_sweep.py --variants $V --rounds 3 --m 2440,4880;
    MMADA_TUNE_PREBUILT=1 timeout 400 python tools/gemm_sweep.py --variants $V --rounds 3 --m 4880,9760 --shapes qkv2:6144:4096,o2:4096:2048,gu2:12288:4096,dn2:4096:6144,qkv4:3072:4096,o4:4096:1024,gu4:6144:4096,dn4:4096:3072;
    MMADA_TUNE_PREBUILT=1 timeout 400 python tools/gemm_sweep.py --variants $V --rounds 3 --m 19520 --shapes qkv8:1536:4096,o8:4096:512,gu8:3072:4096,dn8:4096:1536,qkv4:3072:4096,o4:4096:1024,gu4:6144:4096,dn4:4096:3072; } > $O/gemm8_sweep_final.txt 2> $O/gemm8_sweep.err; echo "sweep rc=$?"; tail -n 12 $O/gemm8_sweep_final.txt
fi
if has bal; then
  # the balanced read schedule on the 320x256 tile (tuning build: configuration 6) against the shipped first schedule (0), every
  # GEMM of the block pinned to the tile: spill-free for every epilogue since the tail K-tiles sit in a one-trip loop
  MMADA_MI355X_LIB=$R/tools/libmmada_mi355x_tune.so timeout 600 python tools/block_ab.py --layers 2 --rounds 7 "gemm_config=0" "gemm_config=6" "gemm_config=4" > $O/block_ab_balanced.txt 2> $O/block_ab_balanced.err; echo "bal rc=$?"; cat $O/block_ab_balanced.txt; tail -2 $O/block_ab_balanced.err
fi
if has rig; then
  # the driver's multi-GPU command on the one-GPU rig (N tensor-parallel processes sharing this GPU over hipIpc, gloo control plane):
  # functional check of every data path the first multi-GPU session may select — not a throughput measurement
  for t in pull copy; do
    MMADA_TP_TRANSPORT=$t MMADA_BENCH_ONE_GPU=1 timeout 600 python bench.py --gpus 2 --steps 1 --warmup 1 --layers 4 --text-steps 16 --timesteps 8 --no-cpu-baseline --no-probe > $O/rig_tp2_$t.json 2> $O/rig_tp2_$t.err; echo "rig tp2 $t rc=$?"; tail -c 1800 $O/rig_tp2_$t.json | head -c 1800; echo
  done
  MMADA_TP_TRANSPORT=copy MMADA_BENCH_ONE_GPU=1 timeout 600 python bench.py --gpus 4 --steps 1 --warmup 1 --layers 4 --text-steps 16 --timesteps 8 --no-cpu-baseline --no-probe > $O/rig_tp4_copy.json 2> $O/rig_tp4_copy.err; echo "rig tp4 copy rc=$?"
fi
if has cfgab; then
  # every GEMM of the block pinned to one tile configuration in turn (product library): which tile each projection wants at B = 1 / 2
  timeout 600 python tools/block_ab.py --layers 2 --rounds 5 "gemm_config=-1" "gemm_config=0" "gemm_config=1" "gemm_config=2" "gemm_config=3" > $O/block_ab_configs.txt 2> $O/block_ab_configs.err; echo "cfgab rc=$?"; cat $O/block_ab_configs.txt; tail -2 $O/block_ab_configs.err
fi
if has diag; then
  # where the main loop's time goes, with the round-5 kernels: the DIAGNOSTIC builds of the 320x256 tile in the -DMMADA_TUNE library
  # (wrong results, timing only; numbers = the TFLOP/s the real GEMM would have at that duration).  300 production; 309 no MFMA;
  # 310 no LDS-DMA; 311 no ds_read; 312 MFMAs + barriers only; 313 LDS-DMA + barriers only; 314 the same, every request an L2 hit;
  # 315 whole kernel, every request an L2 hit; 304 static s_setprio for the late wave group; 306 the balanced read schedule
  { echo "# tools/gemm_sweep.py --variants 300,309,...: diagnostic builds of gemm8 320x256 (csrc/gemm8.hip launch_epi8, -DMMADA_TUNE), gate/up and down shapes, STORE epilogue, cold operands, random bf16";
    MMADA_TUNE_PREBUILT=1 timeout 400 python tools/gemm_sweep.py --variants 300,304,306,309,310,311,312,313,314,315 --rounds 25 --m 2440,4880 --shapes gateup:24576:4096,down:4096:12288; } > $O/gemm8_diagnostics.txt 2> $O/gemm8_diagnostics.err; echo "diag rc=$?"; cat $O/gemm8_diagnostics.txt; tail -2 $O/gemm8_diagnostics.err
fi
if has prof; then
  cd /tmp && export TMPDIR=/tmp
  rocprofv3 --kernel-trace --stats -d $O/kt -o r -- python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-probe > $O/bench_under_rocprof.json 2> $O/kt.err; echo "kt rc=$?"
  cd $R
  python tools/rocprof_summary.py $(ls $O/kt/*results.db 2>/dev/null | head -1) > $O/kernel_stats.csv 2>&1
  rm -rf $O/kt
  head -12 $O/kernel_stats.csv
fi
if has pmc; then
  cd /tmp && export TMPDIR=/tmp
  SHORT="--no-cpu-baseline --no-probe --text-steps 8 --timesteps 4 --warmup 0"
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_f -o f -- python $R/bench.py $SHORT > /dev/null 2> $O/pmc_f.err; echo "pmc_f rc=$?"
  rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_w -o w -- python $R/bench.py $SHORT > /dev/null 2> $O/pmc_w.err; echo "pmc_w rc=$?"
  rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d $O/pmc_t -o t -- python $R/bench.py $SHORT > /dev/null 2> $O/pmc_t.err; echo "pmc_t rc=$?"
  rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS --kernel-trace --output-format csv -d $O/pmc_q -o q -- python $R/bench.py $SHORT > /dev/null 2> $O/pmc_q.err; echo "pmc_q rc=$?"
  cd $R
  for x in f w t q; do f=$(ls $O/pmc_$x/*counter_collection.csv $O/pmc_$x/*/*counter_collection.csv 2>/dev/null | head -1); python tools/pmc_summary.py "$f" "gemm|attn|rmsnorm" > $O/pmc_$x.txt 2>&1; done
  python tools/traffic_from_pmc.py $O/pmc_f.txt $O/pmc_w.txt $O/pmc_t.txt > $O/traffic.json 2> $O/traffic.err; tail -3 $O/traffic.err
  rm -rf $O/pmc_f $O/pmc_w $O/pmc_t $O/pmc_q
fi
