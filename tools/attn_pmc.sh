# SQ counters of the attention forms (separate rocprofv3 --pmc pass, kernel trace only): bash tools/attn_pmc.sh "0,1"
R=${GRAFT_REPO_ROOT:-$(pwd)}
FORMS=${1:-0,1}
O=$R/gpurun_out/r04/attn_pmc
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for B in 1 2; do
  rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS --kernel-trace --output-format csv -d $O/q$B -o q -- python $R/tools/attn_sweep.py --batch $B --forms $FORMS --rounds 2 --iters 6 --warm 100 > $O/sweep_b$B.txt 2>&1
  f=$(ls $O/q$B/*counter_collection.csv 2>/dev/null | head -1)
  python $R/tools/pmc_summary.py "$f" "attn" > $O/pmc_b$B.txt 2>&1
  rm -rf $O/q$B
  echo "B=$B"; cat $O/pmc_b$B.txt
done
