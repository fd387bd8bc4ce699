R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/s1
mkdir -p $O
cd $R
export GPU_MAX_HW_QUEUES=12
(timeout 400 python -m pytest tests/test_gpu_cache.py -q -x -s) > $O/cache.log 2>&1; echo "cache rc=$?"
tail -n 25 $O/cache.log
cd /tmp && export TMPDIR=/tmp
for b in 1 2; do
  timeout 200 rocprofv3 --kernel-trace --stats -d $O/kt$b -o a -- python $R/tools/attn_sweep.py --batch $b > $O/attn_b$b.log 2> $O/attn_b$b.err; echo "attn b$b rc=$?"
  cat $O/attn_b$b.log
  python $R/tools/rocprof_summary.py $(ls $O/kt$b/*results.db 2>/dev/null | head -1) 2>&1 | grep -i "attn_fwd\|kernel," | cut -c1-160
  rm -rf $O/kt$b
done
cd $R
(timeout 1100 python -m pytest tests -q -m gpu --deselect tests/test_gpu_cache.py) > $O/pytest_gpu.log 2>&1; echo "suite rc=$?"
tail -n 6 $O/pytest_gpu.log
