#!/bin/bash
# Regenerate every profiles/r06_* evidence file (E) with ONE gpurun call (round-2 verdict: "make the evidence reproducible by command"):
#
#     /usr/local/graft/bin/gpurun --timeout 2400 -- tools/make_evidence.sh      # on the GPU box: writes gpurun_out/evidence/
#     tools/make_evidence.sh --collect                                          # here: copies the summaries to profiles/r06_*
#     /usr/local/graft/bin/gpurun --timeout 900 -- tools/make_evidence.sh --bench-only   # the bench lines only (after host-side changes)
#
# Steps on the GPU box: (1) the default bench line (after the PMC passes, so that it carries their table); (2) the 768x512 parity suite (tests/test_gpu_headline.py writes the measured
# errors); (3) rocprofv3 --kernel-trace --stats over the same bench command; (4) three separate rocprofv3 --pmc passes (FETCH_SIZE /
# WRITE_SIZE / SQ + GRBM counters; never combined with a trace domain) over bench.py WITHOUT its decode / parity / extra legs, so that the
# profiled process launches exactly (warm-up + steps) identical steps; tools/pmc_bench.py keeps the launches of the timed steps only and
# stamps the table with the hash of the kernel sources; two more passes over a decode-only command for the decode-side kernels;
# (5) the other two configs at full size.
R=r06
if [ "$1" == "--collect" ]; then
    cd "$(dirname "$0")/.." || exit 1
    E=gpurun_out/evidence
    cp $E/bench_default.json profiles/${R}_bench_default.json
    cp $E/bench_profiled_run.json profiles/${R}_bench_default_profiled_run.json
    cp $E/kernel_stats.csv profiles/${R}_bench_default_kernel_stats.csv
    cp $E/pmc_bench.json profiles/${R}_pmc_bench.json
    cp $E/parity_768x512_default.json profiles/${R}_parity_768x512_default.json
    cp $E/parity_768x512_calibrated.json profiles/${R}_parity_768x512_calibrated.json
    cp $E/bench_dataset.json profiles/${R}_bench_dataset.json
    cp $E/bench_large.json profiles/${R}_bench_large.json
    cp $E/pytest_headline.log profiles/${R}_pytest_headline.log
    [ -f $E/decode_overlap.log ] && cp $E/decode_overlap.log profiles/${R}_decode_overlap.log
    [ -f $E/decode_timeline.log ] && cp $E/decode_timeline.log profiles/${R}_decode_timeline.log
    [ -f $E/decode_kernel_stats.csv ] && cp $E/decode_kernel_stats.csv profiles/${R}_decode_batch128_kernel_stats.csv
    [ -f $E/parity_truth.json ] && cp $E/parity_truth.json profiles/${R}_parity_truth.json
    [ -f $E/decode_modes.log ] && cp $E/decode_modes.log profiles/${R}_decode_window_modes.log
    [ -f $E/pytest_gpu.log ] && cp $E/pytest_gpu.log profiles/${R}_pytest_gpu.log
    [ -f $E/bench_files.json ] && cp $E/bench_files.json profiles/${R}_bench_files.json
    [ -f $E/tcc_per_variant.json ] && cp $E/tcc_per_variant.json profiles/${R}_tcc_per_variant.json
    [ -f $E/decode_concurrency_probe.log ] && cp $E/decode_concurrency_probe.log profiles/${R}_decode_concurrency_probe.log
    ls -la profiles/${R}_*
    exit 0
fi
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
E=$PWD/gpurun_out/evidence
if [ "$1" == "--bench-only" ]; then
    # the three bench lines only (host-side changes: the kernel sources, hence the committed PMC table and kernel statistics, are unchanged --
    # bench.py checks that by the source stamp); merged into the existing gpurun_out/evidence/ by --collect
    mkdir -p "$E"
    timeout 900 python bench.py > $E/bench_default.json 2> $E/bench_default.err
    timeout 900 python bench.py --config dataset > $E/bench_dataset.json 2> $E/bench_dataset.err
    timeout 600 python bench.py --config large > $E/bench_large.json 2> $E/bench_large.err
    timeout 900 python bench.py --config files > $E/bench_files.json 2> $E/bench_files.err
    tail -c 1200 $E/bench_default.json; cat $E/bench_dataset.json | cut -c1-300
    exit 0
fi
rm -rf "$E"; mkdir -p "$E"
LIGHT="--no-cpu-baseline --no-parity --no-extra-legs"
timeout 900 python -m pytest tests/test_gpu_headline.py -q > $E/pytest_headline.log 2>&1
timeout 600 python tools/parity_truth.py --out $E/parity_truth.json > $E/parity_truth.log 2>&1      # fp64 ground truth: the three-way table
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $E/smoke.log 2>&1; tail -2 $E/smoke.log
cp gpurun_out/parity_768x512_*.json $E/
# (--no-decode also skips the batch-1 latency leg: the profiled process then launches every encode-side kernel warm-up + steps times and nothing else, so the
# per-kernel averages of the summary are comparable with the HIP-event averages of the bench line; the decode-side kernels are profiled by the decode trace below)
timeout 600 rocprofv3 --kernel-trace --stats -d $E/stats -o run -- python bench.py $LIGHT --no-decode > $E/bench_profiled_run.json 2> $E/rocprof_stats.err
DB=$(find $E/stats -name '*.db' | head -1)
[ -n "$DB" ] && python tools/rocpd_summary.py "$DB" $E/kernel_stats.csv > $E/kernel_stats.txt
PMCARGS="--steps 2 --warmup 1 $LIGHT --no-kernel-events --no-decode"
timeout 600 rocprofv3 --pmc FETCH_SIZE -d $E/pmc/fetch -o run --output-format csv -- python bench.py $PMCARGS > /dev/null 2> $E/pmc_fetch.err
timeout 600 rocprofv3 --pmc WRITE_SIZE -d $E/pmc/write -o run --output-format csv -- python bench.py $PMCARGS > /dev/null 2> $E/pmc_write.err
timeout 600 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_MFMA \
    -d $E/pmc/sq -o run --output-format csv -- python bench.py $PMCARGS > /dev/null 2> $E/pmc_sq.err
# decode side: FETCH / WRITE passes over one batch-128 decode (tools/decode_profile.py), and its kernel trace for the overlap analysis
timeout 300 rocprofv3 --pmc FETCH_SIZE -d $E/pmcd/fetch -o run --output-format csv -- python tools/decode_profile.py 128 > /dev/null 2> $E/pmcd_fetch.err
timeout 300 rocprofv3 --pmc WRITE_SIZE GRBM_GUI_ACTIVE -d $E/pmcd/write -o run --output-format csv -- python tools/decode_profile.py 128 > /dev/null 2> $E/pmcd_write.err
timeout 300 rocprofv3 --kernel-trace --stats -d $E/dtrace -o run --output-format csv -- python tools/decode_profile.py 128 > /dev/null 2> $E/dtrace.err
python tools/decode_overlap.py $E/dtrace > $E/decode_overlap.log
( cd tools && python decode_timeline.py $E/dtrace > $E/decode_timeline.log )
find $E/dtrace -name '*kernel_stats.csv' -exec cp {} $E/decode_kernel_stats.csv \;
# the decode with window rows where the streams allow it (auto: the product), with full rows only (never: the round-4 decoder), both checkpoints
for ck in calibrated default; do for mode in auto never; do
  timeout 300 python tools/decode_profile.py 128 $ck $mode 2>&1 | grep "decode_batch B" | head -2 | tr '\n' ' ' | sed "s/^/$ck $mode: /"; echo
done; done > $E/decode_modes.log
python tools/pmc_bench.py $E/pmc 128 --steps 2 --warmup 1 --decode-root $E/pmcd > $E/pmc_bench.json
cp $E/pmc_bench.json profiles/${R}_pmc_bench.json      # (the box's copy: the bench line below then reports it as current)
timeout 900 python bench.py > $E/bench_default.json 2> $E/bench_default.err
timeout 900 python bench.py --config dataset > $E/bench_dataset.json 2> $E/bench_dataset.err
timeout 600 python bench.py --config large > $E/bench_large.json 2> $E/bench_large.err
timeout 900 python bench.py --config files > $E/bench_files.json 2> $E/bench_files.err
# L2 hit rates and memory-side bytes per kernel variant (round-5 verdict, next 1): TCC_HIT / TCC_MISS pass, joined with the FETCH / WRITE passes above
timeout 600 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum -d $E/pmc_tcc/hm -o run --output-format csv -- python bench.py $PMCARGS > /dev/null 2> $E/pmc_tcc.err
timeout 600 rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_REQ_sum -d $E/pmc_tcc/rd -o run --output-format csv -- python bench.py $PMCARGS > /dev/null 2>> $E/pmc_tcc.err
[ -d $E/pmc ] && python tools/pmc_tcc.py $E/pmc_tcc $E/pmc > $E/tcc_per_variant.json 2>> $E/pmc_tcc.err
rm -rf $E/pmc_tcc
timeout 900 python tools/decode_concurrency_probe.py 2>&1 | grep -v amdgpu.ids > $E/decode_concurrency_probe.log
# the whole GPU suite on the final code
timeout 3000 python -m pytest tests -m gpu -q > $E/pytest_gpu.log 2>&1; tail -3 $E/pytest_gpu.log
# keep the merge-back small: the raw traces stay on the box
rm -rf $E/stats $E/pmc $E/pmcd $E/dtrace $E/pmc_tcc
ls -la $E
tail -c 1500 $E/bench_default.json
