timeout 300 python gpurun_dbg.py 2>&1 | grep -v "^$" | tail -32
