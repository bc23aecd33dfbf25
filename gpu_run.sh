cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 900 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_c -o bench -- python $R/bench.py > $R/gpurun_out/bench_prof.json 2> $R/gpurun_out/bench_prof.err
cd $R
python tools/rocpd_summary.py gpurun_out/prof_c/bench_results.db gpurun_out/r01_c_kernel_stats.csv 2>&1 | head -6
tail -1 gpurun_out/bench_prof.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['roofline'])"
