timeout 900 python -m pytest tests/test_gpu_coder.py tests/test_gpu_net.py -m gpu -q -x --timeout 300 2>&1 | tail -5
timeout 300 python tools/codec_probe.py --B 8 2>&1 | tail -1
