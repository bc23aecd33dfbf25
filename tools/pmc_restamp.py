#!/usr/bin/env python
"""Add the per-file source stamps (`csrc_files`) to a PMC table that only carries the combined `csrc_stamp`.

    python tools/pmc_restamp.py profiles/r05_pmc_bench.json <commit>

The files are read from `<commit>` (git show), NOT from the working tree, and the table is only touched when the combined stamp computed
from that commit's files reproduces the table's `csrc_stamp` -- i.e. when `<commit>` provably is the tree the counter passes ran on.
bench.py then accepts the table for the kernels whose sources are unchanged (bench.ROOFLINE_KERNEL_SOURCES) after OTHER sources moved on."""
import hashlib
import json
import subprocess
import sys


def main():
    path, commit = sys.argv[1], sys.argv[2]
    with open(path) as f:
        table = json.load(f)
    listed = subprocess.check_output(['git', 'ls-tree', '--name-only', commit, 'l3c-pytorch_amd/csrc/']).decode().split()
    names = sorted(n.split('/')[-1] for n in listed if n.endswith(('.hip', '.h'))) + ['../../include/l3c_hip.h']
    combined, per = hashlib.sha256(), {}
    for n in names:
        repo_path = 'include/l3c_hip.h' if n.startswith('..') else 'l3c-pytorch_amd/csrc/' + n
        data = subprocess.check_output(['git', 'show', '{}:{}'.format(commit, repo_path)])
        combined.update(n.encode() + b'\0' + data)
        per[n] = hashlib.sha256(data).hexdigest()[:16]
    if combined.hexdigest()[:16] != table.get('csrc_stamp'):
        print('the sources of {} give {}, the table was taken on {}: not touched'.format(commit, combined.hexdigest()[:16], table.get('csrc_stamp')))
        return 1
    table['csrc_files'] = per
    table['csrc_files_note'] = ('per-file stamps of the tree the passes ran on, added afterwards from commit {} (tools/pmc_restamp.py: its '
                                'sources reproduce csrc_stamp)'.format(commit))
    with open(path, 'w') as f:
        json.dump(table, f, indent=1, sort_keys=True)
        f.write('\n')
    print('ok: {} files stamped'.format(len(per)))
    return 0


if __name__ == '__main__':
    sys.exit(main())
