timeout 900 python -m pytest tests -m gpu -q --timeout 300 2>&1 | tail -4
