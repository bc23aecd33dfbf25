timeout 1200 python -m pytest tests -m gpu -q -x --timeout 600 2>&1 | tail -8
