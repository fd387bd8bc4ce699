# Round-4 evidence run on the GPU box (outputs under gpurun_out/r04/, copied into profiles/ by hand afterwards).
#   bash tools/profile_round4.sh [part ...]      parts: suite bench rig prof pmc (default: all)
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04
mkdir -p $O
cd $R
export GPU_MAX_HW_QUEUES=24
PARTS=${@:-suite bench rig prof pmc}
has() { case " $PARTS " in *" $1 "*) return 0;; esac; return 1; }
# a two-block, six-step bench first: if the benchmark script itself is broken, stop before spending the GPU time
timeout 300 python bench.py --layers 2 --text-steps 4 --timesteps 2 --no-cpu-baseline > $O/smoke_bench.json 2> $O/smoke_bench.err || { echo "smoke bench failed"; tail -5 $O/smoke_bench.err; exit 1; }
if has suite; then (timeout 1500 python -m pytest tests -q -m gpu) > $O/pytest_gpu.log 2>&1; echo "suite rc=$?"; tail -n 3 $O/pytest_gpu.log; fi
if has bench; then
  timeout 500 python bench.py --steps 4 --warmup 1 > $O/bench_config1.json 2> $O/bench_config1.err; echo "bench1 rc=$?"
  timeout 300 python bench.py --config 0 --steps 4 --warmup 1 > $O/bench_config0.json 2> $O/bench_config0.err; echo "bench0 rc=$?"
  timeout 600 python bench.py --config 3 --steps 1 --warmup 1 --no-probe > $O/bench_config3.json 2> $O/bench_config3.err; echo "bench3 rc=$?"
  timeout 900 python bench.py --config 4 --graph on --steps 1 --warmup 0 --no-probe > $O/bench_config4.json 2> $O/bench_config4.err; echo "bench4 rc=$?"
fi
if has rig; then
  # the driver's multi-GPU command on the one-GPU rig: N tensor-parallel processes sharing this GPU (hipIpc pull transport,
  # gloo control plane), hipGraph on; the line carries the eager-vs-graph launch probe.  Not a throughput measurement: the
  # processes time-slice one device (an exchange costs 0.13 ms with 2 processes and ~100 ms with 8), hence the reduced depth.
  for n in 2 4 8; do
    MMADA_BENCH_ONE_GPU=1 timeout 600 python bench.py --gpus $n --steps 1 --warmup 1 --layers 4 --text-steps 16 --timesteps 8 --no-cpu-baseline --no-probe > $O/rig_plain_tp$n.json 2> $O/rig_plain_tp$n.err; echo "rig plain tp$n rc=$? (python bench.py --gpus $n as a plain command: weak scaling, batch = $n jobs, one GPU shared)"
  done
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
  for x in f w t q; do f=$(ls $O/pmc_$x/*counter_collection.csv 2>/dev/null | head -1); python tools/pmc_summary.py "$f" "gemm|attn|rmsnorm" > $O/pmc_$x.txt 2>&1; done
  python tools/traffic_from_pmc.py $O/pmc_f.txt $O/pmc_w.txt $O/pmc_t.txt > $O/traffic.json 2> $O/traffic.err; tail -3 $O/traffic.err
  rm -rf $O/pmc_f $O/pmc_w $O/pmc_t $O/pmc_q
fi
