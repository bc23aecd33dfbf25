timeout 900 python -m pytest tests/test_gpu_net.py -m gpu -q -x --timeout 300 -k "auto_crop or cli or driver" 2>&1 | tail -5
timeout 900 python gpurun_dbg.py 2>&1 | grep -v "INFO\|Need to\|Stitching" | tail -4
