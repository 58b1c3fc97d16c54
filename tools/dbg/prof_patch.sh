# usage (GPU box): bash tools/dbg/prof_patch.sh <tag> <module.attr=value> -- rocprofv3 kernel trace of bench.py run through ab_patch.py
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; T=$1; shift
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $O/prof_$T -o x --output-format csv -- python3 $R/tools/dbg/ab_patch.py "$@" -- --no-cpu-baseline --steps 20 --warmup 8 --probe-launches 0 > $O/prof_$T.json 2> $O/prof_$T.err || { tail -3 $O/prof_$T.err; exit 1; }
tail -1 $O/prof_$T.json
python3 $R/tools/step_seq.py $O/prof_$T/x_kernel_trace.csv --step -3 --window 350 3000 > $O/prof_${T}_head.txt
head -3 $O/prof_${T}_head.txt
