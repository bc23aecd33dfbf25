mkdir -p gpurun_out/pmc3
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc3 -o fetch -- python $GRAFT_REPO_ROOT/tools/conv_probe.py --iters 3 --B 32 > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc3 -o write -- python $GRAFT_REPO_ROOT/tools/conv_probe.py --iters 3 --B 32 > /dev/null 2>&1
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc3 -o l2 -- python $GRAFT_REPO_ROOT/tools/conv_probe.py --iters 3 --B 32 > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
nproc; free -g | head -2
timeout 900 python bench.py > gpurun_out/bench_full.log 2>&1; tail -1 gpurun_out/bench_full.log
