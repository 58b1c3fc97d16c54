# usage (GPU box): bash tools/dbg/prof_ab.sh <alt .so> -- rocprofv3 kernel stats of bench.py with the in-tree library and with an alternative
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; alt=$1
cd /tmp && export TMPDIR=/tmp
for v in default alt; do
  if [ $v = alt ]; then export MTMP_LIB=$alt; else unset MTMP_LIB; fi
  rocprofv3 --kernel-trace --stats -d $O/ab_prof_$v -o x --output-format csv -- python3 $R/bench.py --no-cpu-baseline --steps 20 --warmup 8 --probe-launches 0 > $O/ab_prof_$v.json 2> $O/ab_prof_$v.err || { tail -3 $O/ab_prof_$v.err; exit 1; }
  python3 $R/tools/step_seq.py $O/ab_prof_$v/x_kernel_trace.csv --step -3 --families > $O/ab_prof_${v}_families.txt
  cp $O/ab_prof_$v/x_kernel_stats.csv $O/ab_prof_${v}_stats.csv
  echo "== $v"; head -24 $O/ab_prof_${v}_families.txt
done
