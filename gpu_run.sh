python tools/codec_probe.py --B 8
python tools/codec_probe.py --B 1
