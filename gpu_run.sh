timeout 900 python bench.py --no-cpu-baseline 2>&1 | tail -1
