cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
run() { tag=$1; shift; rocprofv3 --kernel-trace --output-format csv -d gpurun_out/p_$tag -o $tag -- python3 bench.py --no-cpu-baseline --steps 8 --warmup 3 "$@" > gpurun_out/p_$tag.log 2>&1; python tools/trace_summary.py gpurun_out/p_$tag/${tag}_kernel_trace.csv --top 70 > gpurun_out/sum_$tag.txt 2>&1; rm -rf gpurun_out/p_$tag; }
run eager --hip-graph 0
