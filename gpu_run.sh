timeout 900 python -m pytest tests/test_gpu_net.py -m gpu -q --timeout 300 -x 2>&1 | tail -15
