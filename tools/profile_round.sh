R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/r01b
python $R/bench.py > $R/gpurun_out/r01b/bench.json 2> $R/gpurun_out/r01b/bench.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/r01b/kt -o r -- python $R/bench.py --no-cpu-baseline > $R/gpurun_out/r01b/bench_prof.json 2> $R/gpurun_out/r01b/kt.err
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/r01b/pmc_f -o f -- python $R/bench.py --no-cpu-baseline --text-steps 8 --timesteps 4 > /dev/null 2> $R/gpurun_out/r01b/pmc_f.err
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/r01b/pmc_w -o w -- python $R/bench.py --no-cpu-baseline --text-steps 8 --timesteps 4 > /dev/null 2> $R/gpurun_out/r01b/pmc_w.err
cd $R
python tools/rocprof_summary.py gpurun_out/r01b/kt/r_results.db > gpurun_out/r01b/kernel_stats.csv 2>&1
for x in f w; do f=$(ls gpurun_out/r01b/pmc_$x/*counter_collection.csv 2>/dev/null | head -1); python tools/pmc_summary.py $f "gemm|attn" > gpurun_out/r01b/pmc_$x.txt 2>&1; done
rm -rf gpurun_out/r01b/kt gpurun_out/r01b/pmc_f gpurun_out/r01b/pmc_w
cat gpurun_out/r01b/bench.json | head -c 600
