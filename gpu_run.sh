python __graft_entry__.py --smoke 2>&1 | tail -2
timeout 900 python -m pytest tests -m gpu -q --timeout 300 2>&1 | tail -3
timeout 900 python bench.py > gpurun_out/bench_full.log 2>&1; tail -1 gpurun_out/bench_full.log
