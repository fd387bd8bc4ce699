mkdir -p gpurun_out
export MMADA_TP_TIMEOUT_S=8
(timeout 300 python -m pytest tests/test_gpu_tp.py -q -s -m gpu) > gpurun_out/r3_tp.log 2>&1; echo tp rc=$?
(timeout 600 python -m pytest tests/test_gpu_model.py -k "bench_multi_rank" -q -s -m gpu) > gpurun_out/r3_rig.log 2>&1; echo rig rc=$?
(MMADA_ATTN_WGS=3 timeout 400 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_model.py tests/test_gpu_fullsize.py -k "attn or sdpa or attention or block_stages or forward_hidden or one_8b or batch_invariance or consumed_row" -q -s -m gpu) > gpurun_out/r3_attn3.log 2>&1; echo attn3 rc=$?
MMADA_ATTN_WGS=2 timeout 300 python bench.py --steps 1 --warmup 1 --no-cpu-baseline > gpurun_out/r3_bench_wgs2.json 2> gpurun_out/r3_bench_wgs2.err; echo wgs2 rc=$?
MMADA_ATTN_WGS=3 timeout 300 python bench.py --steps 1 --warmup 1 --no-cpu-baseline > gpurun_out/r3_bench_wgs3.json 2> gpurun_out/r3_bench_wgs3.err; echo wgs3 rc=$?
timeout 600 python bench.py --config 3 --steps 1 --warmup 1 > gpurun_out/r3_bench_cfg3.json 2> gpurun_out/r3_bench_cfg3.err; echo cfg3 rc=$?
timeout 900 python bench.py --config 4 --graph on --steps 1 --warmup 0 > gpurun_out/r3_bench_cfg4.json 2> gpurun_out/r3_bench_cfg4.err; echo cfg4 rc=$?
for f in gpurun_out/r3_tp.log gpurun_out/r3_rig.log gpurun_out/r3_attn3.log; do echo "== $f"; tail -n 5 $f; done
