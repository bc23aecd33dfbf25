for N in 8 16 32; do
echo chunks=$N
L3C_RGB_CHUNKS=$N timeout 300 python tools/codec_probe.py --B 8 2>&1 | tail -1
L3C_RGB_CHUNKS=$N timeout 300 python tools/codec_probe.py --B 64 2>&1 | tail -1
done
