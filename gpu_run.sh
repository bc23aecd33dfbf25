timeout 900 python -m pytest tests/test_gpu_coder.py tests/test_gpu_net.py -m gpu -q --timeout 300 2>&1 | tail -4
timeout 600 python tools/bench_dataset.py --n 64 2>&1 | tail -3
timeout 600 python tools/bench_dataset.py --n 64 --per-batch-coder 2>&1 | tail -3
GPU_MAX_HW_QUEUES=8 timeout 600 python tools/bench_dataset.py --n 64 2>&1 | tail -3
timeout 600 python bench.py --no-cpu-baseline 2>&1 | tail -1
