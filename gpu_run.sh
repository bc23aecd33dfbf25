python tools/conv_probe.py
python tools/conv_probe.py --B 32
for b in 16 32; do timeout 600 python bench.py --steps 6 --warmup 2 --batch $b --no-cpu-baseline 2>&1 | tail -1 | python -c "
import sys, json
r = json.loads(sys.stdin.read())
print('B', r['config']['batch_per_gpu'], 'MPix/s', r['value'], 'ms/step', r['ms_per_step'], 'roofline', r['roofline']['achieved'], r['roofline']['all_mfma_convs'])"; done
