python -m pytest tests -m gpu -x -q -k "ln_bwd or training_step or encoder_layer" 2>&1 | tail -3
python tools/bench_kernels.py --only stream 2>&1 | grep ln_bwd
python bench.py --no-cpu-baseline --hip-graph 0 2>&1 | tail -1 | cut -c1-200
python bench.py --no-cpu-baseline --hip-graph 1 2>&1 | tail -1 | cut -c1-200
