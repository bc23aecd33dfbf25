timeout 600 python -m pytest tests/test_gpu_net.py -m gpu -q -x --timeout 300 -k "per_channel" 2>&1 | tail -12
