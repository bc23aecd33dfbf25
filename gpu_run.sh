timeout 900 python -m pytest tests -m gpu -q -x --timeout 300 2>&1 | tail -5
timeout 300 python tools/codec_probe.py --B 8 2>&1 | tail -1
