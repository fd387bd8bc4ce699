# Round-2 evidence run on the GPU box: test suite, the three bench lines, rocprofv3 kernel trace and PMC passes.
# Outputs under gpurun_out/r02/ (copied into profiles/ by hand afterwards).  Usage: bash tools/profile_round2.sh
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r02
mkdir -p $O
cd $R
export GPU_MAX_HW_QUEUES=12
# a two-block, six-step bench first: if the benchmark script itself is broken, stop before spending the GPU time
timeout 300 python bench.py --layers 2 --text-steps 4 --timesteps 2 --no-cpu-baseline > $O/smoke_bench.json 2> $O/smoke_bench.err || { echo "smoke bench failed"; tail -5 $O/smoke_bench.err; exit 1; }
(timeout 1100 python -m pytest tests -q -m gpu) > $O/pytest_gpu.log 2>&1; echo "suite rc=$?"
timeout 500 python bench.py > $O/bench_config1.json 2> $O/bench_config1.err; echo "bench1 rc=$?"
timeout 600 python bench.py --config 3 --steps 1 --warmup 1 > $O/bench_config3.json 2> $O/bench_config3.err; echo "bench3 rc=$?"
timeout 900 python bench.py --config 4 --graph on --steps 1 --warmup 0 > $O/bench_config4.json 2> $O/bench_config4.err; echo "bench4 rc=$?"
# hipGraph step on the multi-process tensor-parallel rig (2 ranks on this one GPU, library exchange inside the graph)
for g in off on; do
  MMADA_BENCH_ONE_GPU=1 timeout 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node=2 --master-addr 127.0.0.1 --master-port 29655 bench.py --gpus 2 --steps 1 --warmup 1 --layers 8 --text-steps 16 --timesteps 8 --no-cpu-baseline --graph $g > $O/rig_tp2_graph_$g.json 2> $O/rig_tp2_graph_$g.err; echo "rig graph $g rc=$?"
done
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $O/kt -o r -- python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline > $O/bench_under_rocprof.json 2> $O/kt.err; echo "kt rc=$?"
SHORT="--no-cpu-baseline --text-steps 8 --timesteps 4 --warmup 0"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_f -o f -- python $R/bench.py $SHORT > /dev/null 2> $O/pmc_f.err; echo "pmc_f rc=$?"
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_w -o w -- python $R/bench.py $SHORT > /dev/null 2> $O/pmc_w.err; echo "pmc_w rc=$?"
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d $O/pmc_t -o t -- python $R/bench.py $SHORT > /dev/null 2> $O/pmc_t.err; echo "pmc_t rc=$?"
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS --kernel-trace --output-format csv -d $O/pmc_q -o q -- python $R/bench.py $SHORT > /dev/null 2> $O/pmc_q.err; echo "pmc_q rc=$?"
cd $R
python tools/rocprof_summary.py $(ls $O/kt/*results.db 2>/dev/null | head -1) > $O/kernel_stats.csv 2>&1
for x in f w t q; do f=$(ls $O/pmc_$x/*counter_collection.csv 2>/dev/null | head -1); python tools/pmc_summary.py "$f" "gemm_bt|attn_fwd|rmsnorm" > $O/pmc_$x.txt 2>&1; done
rm -rf $O/kt $O/pmc_f $O/pmc_w $O/pmc_t $O/pmc_q
tail -n 3 $O/pytest_gpu.log; head -c 400 $O/bench_config1.json; echo; head -12 $O/kernel_stats.csv
