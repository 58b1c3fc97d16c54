cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -x -q 2>&1 | tail -2
rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/p_write -o w -- python3 bench.py --no-cpu-baseline --hip-graph 0 --steps 3 --warmup 2 --probe-steps 0 > gpurun_out/p_write.log 2>&1
python tools/pmc_summary.py gpurun_out/p_write/w_counter_collection.csv --match attn_fwd --json gpurun_out/pmc_write_attn.json 2>&1 | cut -c1-300; rm -rf gpurun_out/p_write
rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/p_fetch -o f -- python3 bench.py --no-cpu-baseline --hip-graph 0 --steps 3 --warmup 2 --probe-steps 0 > gpurun_out/p_fetch.log 2>&1
python tools/pmc_summary.py gpurun_out/p_fetch/f_counter_collection.csv --match attn_fwd --json gpurun_out/pmc_fetch_attn.json 2>&1 | cut -c1-300; rm -rf gpurun_out/p_fetch
python bench.py --no-cpu-baseline 2>&1 | grep '"metric"' | cut -c1-700
