R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS --kernel-trace --output-format csv -d $R/gpurun_out/pmc_sq -o q -- python $R/bench.py --no-cpu-baseline --text-steps 8 --timesteps 4 > /dev/null 2> $R/gpurun_out/pmc_sq.err
cd $R
f=$(ls gpurun_out/pmc_sq/*counter_collection.csv 2>/dev/null | head -1)
python tools/pmc_summary.py $f "gemm_bt|attn_fwd" > gpurun_out/pmc_sq_model.txt 2>&1
rm -rf gpurun_out/pmc_sq
head -50 gpurun_out/pmc_sq_model.txt
