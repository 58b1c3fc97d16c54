python -m pytest tests -m gpu -x -q 2>&1 | tail -3
python bench.py --no-cpu-baseline 2>&1 | tail -1 | cut -c1-330
