for i in 1 2 3; do python -m pytest tests -m gpu -x -q 2>&1 | tail -1; done
python bench.py --no-cpu-baseline 2>&1 | tail -1 | cut -c1-330
python bench.py --no-cpu-baseline --hip-graph 0 2>&1 | tail -1 | cut -c1-330
MTMP_FORCE_DDP=1 python bench.py --no-cpu-baseline 2>&1 | tail -1 | cut -c1-330
MTMP_FORCE_DDP=1 python bench.py --no-cpu-baseline --hip-graph 0 2>&1 | tail -1 | cut -c1-330
