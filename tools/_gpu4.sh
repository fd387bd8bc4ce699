mkdir -p gpurun_out
export MMADA_TP_TIMEOUT_S=8
(timeout 300 python -m pytest tests/test_gpu_tp.py -q -s -m gpu) > gpurun_out/r4_tp.log 2>&1; echo tp rc=$?
(MMADA_GEMM_BM=320 timeout 300 python -m pytest tests/test_gpu_kernels.py -k "gemm" -q -s -m gpu) > gpurun_out/r4_gemm320.log 2>&1; echo gemm320 rc=$?
for bm in 256 320 160; do MMADA_GEMM_BM=$bm timeout 200 python tools/gemm_sweep.py --variants 100 --m 2438,4876 --rounds 5 > gpurun_out/r4_sweep_bm$bm.log 2>&1; echo sweep$bm rc=$?; done
timeout 200 python tools/gemm_sweep.py --variants 100 --m 2438,4876 --rounds 5 > gpurun_out/r4_sweep_auto.log 2>&1; echo sweep_auto rc=$?
timeout 300 python bench.py --steps 1 --warmup 1 --no-cpu-baseline > gpurun_out/r4_bench.json 2> gpurun_out/r4_bench.err; echo bench rc=$?
MMADA_GEMM_NO320=1 timeout 300 python bench.py --steps 1 --warmup 1 --no-cpu-baseline > gpurun_out/r4_bench_no320.json 2> gpurun_out/r4_bench_no320.err; echo bench_no320 rc=$?
(timeout 1100 python -m pytest tests -q -m gpu --ignore=tests/test_gpu_tp.py) > gpurun_out/r4_suite.log 2>&1; echo suite rc=$?
tail -n 4 gpurun_out/r4_tp.log gpurun_out/r4_gemm320.log gpurun_out/r4_suite.log
cat gpurun_out/r4_sweep_bm256.log gpurun_out/r4_sweep_bm320.log gpurun_out/r4_sweep_bm160.log gpurun_out/r4_sweep_auto.log | grep -v "^$" | tail -40
