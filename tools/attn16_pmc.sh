# SQ counters of the attention16 kernel forms (separate rocprofv3 --pmc passes, kernel trace only): bash tools/attn16_pmc.sh "1,3,3:2"
R=${GRAFT_REPO_ROOT:-$(pwd)}
FORMS=${1:-1,3}
O=$R/gpurun_out/attn16_pmc
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for B in 1 2; do
  rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS --kernel-trace --output-format csv -d $O/q$B -o q -- python $R/tools/attn16_dev.py --time --batch $B --forms $FORMS --rounds 2 --iters 6 --warm 100 > $O/sweep_b$B.txt 2>&1
  f=$(ls $O/q$B/*counter_collection.csv $O/q$B/*/*counter_collection.csv 2>/dev/null | head -1)
  python $R/tools/pmc_summary.py "$f" "attn" > $O/pmc_b$B.txt 2>&1
  rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU --kernel-trace --output-format csv -d $O/r$B -o r -- python $R/tools/attn16_dev.py --time --batch $B --forms $FORMS --rounds 2 --iters 6 --warm 100 > /dev/null 2>&1
  f=$(ls $O/r$B/*counter_collection.csv $O/r$B/*/*counter_collection.csv 2>/dev/null | head -1)
  python $R/tools/pmc_summary.py "$f" "attn" >> $O/pmc_b$B.txt 2>&1
  rm -rf $O/q$B $O/r$B
  echo "B=$B"; cat $O/pmc_b$B.txt
done
