python -m pytest tests -m gpu -x -q -k "evaluator or resume" 2>&1 | tail -5
