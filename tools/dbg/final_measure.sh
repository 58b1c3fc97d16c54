set -o pipefail
python -m pytest tests -m gpu -x -q > gpurun_out/r01t_tests.log 2>&1 || { tail -5 gpurun_out/r01t_tests.log; exit 1; }
tail -1 gpurun_out/r01t_tests.log
cp gpurun_out/parity_report.json gpurun_out/r01t_parity_report.json
python bench.py > gpurun_out/r01t_bench.json 2> gpurun_out/r01t_bench.err || { tail -5 gpurun_out/r01t_bench.err; exit 1; }
tail -1 gpurun_out/r01t_bench.json | cut -c1-400
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/r01t_prof -o x --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --steps 20 --warmup 8 > $GRAFT_REPO_ROOT/gpurun_out/r01t_prof_bench.json 2> $GRAFT_REPO_ROOT/gpurun_out/r01t_prof.err
cd $GRAFT_REPO_ROOT
tail -1 gpurun_out/r01t_prof_bench.json | cut -c1-200
python tools/trace_summary.py gpurun_out/r01t_prof/x_kernel_trace.csv --skip-last 4 --json gpurun_out/r01t_trace_summary.json > /dev/null 2>&1; ls -la gpurun_out/r01t_prof | head
python tools/step_seq.py gpurun_out/r01t_prof/x_kernel_trace.csv --step -8 --families > gpurun_out/r01t_step_families.txt; head -12 gpurun_out/r01t_step_families.txt
