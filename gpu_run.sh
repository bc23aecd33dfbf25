timeout 600 python gpurun_dbg.py 2>&1 | tail -8
timeout 900 python -m pytest tests/test_gpu_coder.py tests/test_gpu_net.py -m gpu -q -x --timeout 300 2>&1 | tail -8
timeout 300 python tools/codec_probe.py --B 8 2>&1 | tail -2
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_dec -o dec -- python $GRAFT_REPO_ROOT/tools/codec_probe.py --B 8 > /dev/null 2>&1
cd $GRAFT_REPO_ROOT && python tools/rocpd_summary.py gpurun_out/prof_dec/dec_results.db 2>&1 | head -6
