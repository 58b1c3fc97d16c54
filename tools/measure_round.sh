# usage (on the GPU box, through gpurun): bash tools/measure_round.sh r02x [quick]
# the round's measurement sequence: GPU tests, bench.py (+ forced-DDP staged run), smoke(), rocprofv3 kernel trace of the
# same command + summaries, and the FETCH_SIZE / WRITE_SIZE passes of the roofline kernel.  Everything lands in gpurun_out/.
set -o pipefail
T=$1; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
if [ "$2" != "quick" ]; then
  python -m pytest tests -m gpu -x -q > $O/${T}_tests.log 2>&1 || { tail -5 $O/${T}_tests.log; exit 1; }
  tail -1 $O/${T}_tests.log
  cp $O/parity_report.json $O/${T}_parity_report.json
  python bench.py > $O/${T}_bench.json 2> $O/${T}_bench.err || { tail -5 $O/${T}_bench.err; exit 1; }
  tail -1 $O/${T}_bench.json | cut -c1-300
  python bench.py --no-cpu-baseline --force-ddp > $O/${T}_bench_ddp1.json 2> $O/${T}_bench_ddp1.err || { tail -5 $O/${T}_bench_ddp1.err; exit 1; }
  tail -1 $O/${T}_bench_ddp1.json | cut -c1-200
  for w in ragged cfg5; do
    python bench.py --no-cpu-baseline --workload $w --steps 20 --warmup 6 > $O/${T}_bench_$w.json 2> $O/${T}_bench_$w.err || { tail -5 $O/${T}_bench_$w.err; exit 1; }
    tail -1 $O/${T}_bench_$w.json | cut -c1-200
  done
  python bench.py --no-cpu-baseline --workload ragged --steps 20 --warmup 6 --probe-launches 0 --instep-steps 0 --pack-rows 0 --skip-missing-images 0 > $O/${T}_bench_ragged_padded.json 2> /dev/null
  tail -1 $O/${T}_bench_ragged_padded.json | cut -c1-200
  python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
fi
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $O/${T}_prof -o x --output-format csv -- python3 $R/bench.py --no-cpu-baseline --steps 20 --warmup 8 --probe-launches 0 --instep-steps 0 > $O/${T}_prof_bench.json 2> $O/${T}_prof.err
cd $R
tail -1 $O/${T}_prof_bench.json | cut -c1-200
python tools/trace_summary.py $O/${T}_prof/x_kernel_trace.csv --json $O/${T}_trace_summary.json > /dev/null 2>&1
cp $O/${T}_prof/x_kernel_stats.csv $O/${T}_bench_kernel_stats.csv 2>/dev/null
python tools/step_seq.py $O/${T}_prof/x_kernel_trace.csv --step -3 --families > $O/${T}_step_families.txt; head -30 $O/${T}_step_families.txt
if [ "$2" != "quick" ]; then
  cd /tmp
  for c in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $c -d $O/pmc_rf_$c -o x --output-format csv -- python3 $R/bench.py --no-cpu-baseline --hip-graph 0 --steps 3 --warmup 2 --probe-launches 0 > $O/pmc_rf_$c.log 2>&1
    python3 $R/tools/pmc_summary.py $O/pmc_rf_$c/x_counter_collection.csv --match attn_fwd --json $O/${T}_pmc_rf_$c.json > /dev/null 2>&1
    echo done $c
  done
fi
