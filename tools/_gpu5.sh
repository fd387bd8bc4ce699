mkdir -p gpurun_out
export MMADA_TP_TIMEOUT_S=8
(timeout 300 python -m pytest tests/test_gpu_tp.py -q -s -m gpu) > gpurun_out/r5_tp.log 2>&1; echo tp rc=$?
(timeout 600 python -m pytest tests/test_gpu_model.py -k "bench_multi_rank" -q -s -m gpu) > gpurun_out/r5_rig.log 2>&1; echo rig rc=$?
(timeout 300 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_fullsize.py -k "attn or sdpa or attention or batch_invariance or consumed_row" -q -m gpu) > gpurun_out/r5_attn.log 2>&1; echo attn rc=$?
MMADA_ATTN_XCD=0 timeout 300 python bench.py --steps 1 --warmup 1 --no-cpu-baseline > gpurun_out/r5_bench_xcd0.json 2> gpurun_out/r5_bench_xcd0.err; echo xcd0 rc=$?
timeout 300 python bench.py --steps 1 --warmup 1 --no-cpu-baseline > gpurun_out/r5_bench_xcd1.json 2> gpurun_out/r5_bench_xcd1.err; echo xcd1 rc=$?
tail -n 4 gpurun_out/r5_tp.log gpurun_out/r5_rig.log gpurun_out/r5_attn.log
