timeout 900 python -m pytest tests -m gpu -q --timeout 300 2>&1 | tail -3
timeout 900 python bench.py --no-cpu-baseline 2>&1 | tail -1 | python -c "
import sys, json
r = json.loads(sys.stdin.read())
print('B', r['config']['batch_per_gpu'], 'MPix/s', r['value'], 'ms/step', r['ms_per_step'], 'roofline', r['roofline']['achieved'], r['roofline']['all_mfma_convs'])"
