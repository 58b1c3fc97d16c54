# HBM traffic of the roofline kernel (vslt-stream attention forward): FETCH_SIZE and WRITE_SIZE in two separate --pmc passes
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c -d $R/gpurun_out/pmc_rf_$c -o x --output-format csv -- python3 $R/bench.py --no-cpu-baseline --hip-graph 0 --steps 3 --warmup 2 --probe-steps 0 > $R/gpurun_out/pmc_rf_$c.log 2>&1
  python3 $R/tools/pmc_summary.py $R/gpurun_out/pmc_rf_$c/x_counter_collection.csv --match attn_fwd --json $R/gpurun_out/pmc_rf_$c.json > /dev/null 2>&1
  echo done $c
done
