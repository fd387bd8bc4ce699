mkdir -p gpurun_out
export MMADA_TP_TIMEOUT_S=8
(timeout 300 python -m pytest tests/test_gpu_tp.py -q -s -m gpu) > gpurun_out/r2_tp.log 2>&1; echo tp rc=$?
(timeout 600 python -m pytest tests/test_gpu_model.py -k "graph or bench_multi_rank" -q -s -m gpu) > gpurun_out/r2_graph_rig.log 2>&1; echo graph_rig rc=$?
(timeout 400 python -m pytest tests/test_gpu_fullsize.py -k "graph or config0 or free_running or one_8b" -q -s -m gpu) > gpurun_out/r2_full.log 2>&1; echo full rc=$?
(timeout 300 python -m pytest tests/test_gpu_parity_depth.py -k each_op -q -s -m gpu) > gpurun_out/r2_perop.log 2>&1; echo perop rc=$?
timeout 300 python bench.py --config 3 --layers 2 --text-steps 8 --timesteps 4 --no-cpu-baseline --warmup 0 > gpurun_out/r2_b3_dbg.json 2> gpurun_out/r2_b3_dbg.err; echo b3 rc=$?
timeout 300 python bench.py --config 4 --batch 2 --layers 2 --text-steps 8 --timesteps 4 --no-cpu-baseline --warmup 0 > gpurun_out/r2_b4_dbg.json 2> gpurun_out/r2_b4_dbg.err; echo b4 rc=$?
timeout 300 python bench.py --graph on --steps 1 --warmup 1 --no-cpu-baseline > gpurun_out/r2_bench_graph.json 2> gpurun_out/r2_bench_graph.err; echo bgraph rc=$?
timeout 400 python bench.py --steps 1 --warmup 1 > gpurun_out/r2_bench2.json 2> gpurun_out/r2_bench2.err; echo bench rc=$?
for f in gpurun_out/r2_tp.log gpurun_out/r2_graph_rig.log gpurun_out/r2_full.log gpurun_out/r2_perop.log; do echo "== $f"; tail -n 4 $f; done
