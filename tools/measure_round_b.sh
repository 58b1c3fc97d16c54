# second half of tools/measure_round.sh: the other workloads, smoke(), rocprofv3 kernel trace + summaries, PMC passes of the roofline kernel
set -o pipefail
T=$1; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
for w in ragged cfg5; do
  python bench.py --no-cpu-baseline --workload $w --steps 20 --warmup 6 > $O/${T}_bench_$w.json 2> $O/${T}_bench_$w.err || { tail -5 $O/${T}_bench_$w.err; exit 1; }
  tail -1 $O/${T}_bench_$w.json | cut -c1-200
done
python bench.py --no-cpu-baseline --workload ragged --steps 20 --warmup 6 --probe-launches 0 --instep-steps 0 --pack-rows 0 --skip-missing-images 0 > $O/${T}_bench_ragged_padded.json 2> /dev/null
tail -1 $O/${T}_bench_ragged_padded.json | cut -c1-200
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
bash tools/measure_round.sh $T quick
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c -d $O/pmc_rf_$c -o x --output-format csv -- python3 $R/bench.py --no-cpu-baseline --hip-graph 0 --steps 3 --warmup 2 --probe-launches 0 > $O/pmc_rf_$c.log 2>&1
  python3 $R/tools/pmc_summary.py $O/pmc_rf_$c/x_counter_collection.csv --match attn_fwd --json $O/${T}_pmc_rf_$c.json > /dev/null 2>&1
  echo done $c
done
