python -m pytest tests -m gpu -x -q 2>&1 | tail -15 | cut -c1-250
python tools/bench_kernels.py --only attn 2>&1 | grep attn
python bench.py --no-cpu-baseline 2>&1 | tail -1 | cut -c1-330
