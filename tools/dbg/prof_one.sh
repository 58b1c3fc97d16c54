# usage (GPU box): bash tools/dbg/prof_one.sh <tag> -- rocprofv3 kernel trace of bench.py, per-family sums of one replayed step
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; T=$1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $O/prof_$T -o x --output-format csv -- python3 $R/bench.py --no-cpu-baseline --steps 20 --warmup 8 --probe-launches 0 > $O/prof_$T.json 2> $O/prof_$T.err || { tail -3 $O/prof_$T.err; exit 1; }
python3 $R/tools/step_seq.py $O/prof_$T/x_kernel_trace.csv --step -3 --families > $O/prof_${T}_families.txt
cp $O/prof_$T/x_kernel_stats.csv $O/prof_${T}_stats.csv
head -16 $O/prof_${T}_families.txt
