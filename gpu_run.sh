cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_dec3 -o dec -- python $GRAFT_REPO_ROOT/tools/codec_probe.py --B 8 > /dev/null 2>&1
cd $GRAFT_REPO_ROOT && python tools/rocpd_summary.py gpurun_out/prof_dec3/dec_results.db gpurun_out/r01_d_decode_kernel_stats.csv 2>&1 | head -12
