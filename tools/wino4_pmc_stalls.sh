#!/bin/bash
# Stall attribution of conv_wino4_kernel by counters (two rocprofv3 --pmc passes over tools/wino4_probe_variants.py; no trace domains):
#     /usr/local/graft/bin/gpurun --timeout 90 -- tools/wino4_pmc_stalls.sh
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=$PWD/gpurun_out/w4stalls; rm -rf $O; mkdir -p $O
timeout 40 rocprofv3 --pmc SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_WAVE_CYCLES \
    -d $O/a -o run --output-format csv -- python tools/wino4_probe_variants.py > $O/a.log 2>&1
timeout 40 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INST_LEVEL_LDS SQ_INSTS_LDS \
    -d $O/b -o run --output-format csv -- python tools/wino4_probe_variants.py > $O/b.log 2>&1
python - <<'PY'
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob('gpurun_out/w4stalls/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        n = r['Kernel_Name']
        if 'conv_wino4_kernel' not in n:
            continue
        key = 'res' if 'ILb0ELb1ELb0' in n else 'relu' if 'ILb1ELb0ELb0' in n else 'other'
        acc[key][r['Counter_Name']].append(float(r['Counter_Value']))
for k, c in acc.items():
    print(k, {n: round(sum(v) / len(v), 1) for n, v in sorted(c.items())}, 'launches', max(len(v) for v in c.values()))
PY
tail -2 $O/a.log $O/b.log
rm -rf $O/a $O/b
