#!/usr/bin/env python
"""Summarise a rocprofv3 (rocpd SQLite) kernel trace: per-kernel count / total / avg / min / max, like `--stats`.

    python tools/rocpd_summary.py gpurun_out/prof/xyz_results.db [out.csv]
"""
import csv
import sqlite3
import sys


def summarise(db_path):
    db = sqlite3.connect(db_path)
    cur = db.cursor()
    rows = list(cur.execute(
        "select s.kernel_name, count(*), sum(d.end - d.start), avg(d.end - d.start), min(d.end - d.start), "
        "max(d.end - d.start), max(s.arch_vgpr_count), max(s.accum_vgpr_count), max(s.sgpr_count), max(d.group_segment_size) "
        "from rocpd_kernel_dispatch d join rocpd_info_kernel_symbol s on d.kernel_id = s.id "
        "group by s.kernel_name order by 3 desc"))
    total = float(sum(r[2] for r in rows)) or 1.0
    return [{'kernel': r[0], 'calls': r[1], 'total_ms': r[2] / 1e6, 'pct': 100.0 * r[2] / total, 'avg_us': r[3] / 1e3,
             'min_us': r[4] / 1e3, 'max_us': r[5] / 1e3, 'vgpr': r[6], 'agpr': r[7], 'sgpr': r[8], 'lds_bytes': r[9]}
            for r in rows]


def main():
    rows = summarise(sys.argv[1])
    out = sys.argv[2] if len(sys.argv) > 2 else None
    if out:
        with open(out, 'w', newline='') as f:
            w = csv.DictWriter(f, fieldnames=list(rows[0].keys()))
            w.writeheader()
            for r in rows:
                w.writerow({k: (round(v, 3) if isinstance(v, float) else v) for k, v in r.items()})
    for r in rows[:20]:
        print('{total_ms:10.2f} ms {pct:5.1f}%  n={calls:5d} avg={avg_us:10.1f}us  v{vgpr}/a{agpr}/s{sgpr} lds={lds_bytes}  {k}'.format(
            k=r['kernel'][:90], **{k: v for k, v in r.items() if k != 'kernel'}))


if __name__ == '__main__':
    main()
