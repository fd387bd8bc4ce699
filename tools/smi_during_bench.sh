# Sample rocm-smi (sclk, power, temperature) twice a second while bench.py runs: evidence for the sustained-clock discussion.
R=${GRAFT_REPO_ROOT:-.}
out=$R/gpurun_out/smi_during_bench.txt
( while true; do rocm-smi --showclocks --showpower --showtemp 2>/dev/null | grep -E "sclk|mclk|Power|Temperature \(Sensor (edge|junction)" | tr '\n' ' '; echo; sleep 0.5; done ) > $out &
SMI=$!
python $R/bench.py --no-cpu-baseline > $R/gpurun_out/smi_bench.json 2>/dev/null
kill $SMI
wc -l $out; sed -n '1p;20p;40p;60p;80p' $out | cut -c1-400
