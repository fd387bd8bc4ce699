# kernel-only times of the attention forms under rocprofv3 (kernel trace): bash tools/attn_prof.sh "0,1" OUTDIR
R=${GRAFT_REPO_ROOT:-$(pwd)}
FORMS=${1:-0,1}
O=${2:-$R/gpurun_out/r04/attn}
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for B in 1 2; do
  MMADA_TUNE_PREBUILT=1 rocprofv3 --kernel-trace -d $O/kt$B -o a -- python $R/tools/attn_sweep.py --batch $B --forms $FORMS --rounds 6 --iters 20 --warm 1500 > $O/sweep_b$B.txt 2>&1
  python $R/tools/rocprof_summary.py $(ls $O/kt$B/*results.db | head -1) 2>/dev/null | grep -E "attn|kernel,calls" | head -12 > $O/kernels_b$B.csv
  rm -rf $O/kt$B
  grep -E "form|err" $O/sweep_b$B.txt | tail -9; cat $O/kernels_b$B.csv
done
