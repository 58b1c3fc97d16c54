cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
python bench.py > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/p_stats -o st -- python3 bench.py --no-cpu-baseline --steps 10 --warmup 3 > gpurun_out/p_stats.log 2>&1
python tools/trace_summary.py gpurun_out/p_stats/st_kernel_trace.csv --top 60 --json gpurun_out/trace_summary.json > gpurun_out/trace_summary.txt 2>&1
cp gpurun_out/p_stats/st_kernel_stats.csv gpurun_out/kernel_stats.csv; rm -rf gpurun_out/p_stats
rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/p_fetch -o f -- python3 bench.py --no-cpu-baseline --hip-graph 0 --steps 3 --warmup 2 --probe-steps 0 > gpurun_out/p_fetch.log 2>&1
ls gpurun_out/p_fetch > gpurun_out/p_fetch.ls
python tools/pmc_summary.py gpurun_out/p_fetch/f_counter_collection.csv --json gpurun_out/pmc_fetch.json > gpurun_out/pmc_fetch.txt 2>&1; rm -rf gpurun_out/p_fetch
rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/p_write -o w -- python3 bench.py --no-cpu-baseline --hip-graph 0 --steps 3 --warmup 2 --probe-steps 0 > gpurun_out/p_write.log 2>&1
python tools/pmc_summary.py gpurun_out/p_write/w_counter_collection.csv --json gpurun_out/pmc_write.json > gpurun_out/pmc_write.txt 2>&1; rm -rf gpurun_out/p_write
