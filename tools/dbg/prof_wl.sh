# usage (GPU box): bash tools/dbg/prof_wl.sh TAG [bench flags...]   -> kernel trace of bench.py + per-family step breakdown
set -o pipefail
T=$1; shift; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $O/${T}_prof -o x --output-format csv -- python3 $R/bench.py --no-cpu-baseline --steps 20 --warmup 8 --probe-launches 0 --instep-steps 0 "$@" > $O/${T}_prof_bench.json 2> $O/${T}_prof.err
cd $R
tail -1 $O/${T}_prof_bench.json | cut -c1-200
python tools/step_seq.py $O/${T}_prof/x_kernel_trace.csv --step -3 --families --all-queues > $O/${T}_step_families.txt; head -24 $O/${T}_step_families.txt
python tools/step_seq.py $O/${T}_prof/x_kernel_trace.csv --step -3 > $O/${T}_step_seq.txt
rm -rf $O/${T}_prof
