#!/usr/bin/env python
"""Development probe (not product): the RGB phase of a batch decode with the decoder chains and the table kernels on DISJOINT compute
units (hipExtStreamCreateWithCUMask streams, l3c_stream_create_cu_range), against the product's schedule (both on all CUs).
Round 4 measured no gain when a table launch took as long as a decode launch; with window rows the tables are a quarter of that.

usage: python tools/decode_cu_split_probe.py [B]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import bench  # noqa: E402
from l3c_pytorch_amd import _lib  # noqa: E402
from l3c_pytorch_amd.bitcoding.bitcoding import Bitcoding  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
cfg, sd, bp, bc, synthetic = bench.build_path('cr', 0, True)
imgs = torch.stack([synthetic.make_image(512, 768, i, 'natural') for i in range(B)]).cuda()
files = bc.encode_batch(imgs.float()).to_bytes()
torch.cuda.synchronize()
_, N_CU, _ = _lib.device_info()

_orig = Bitcoding._decode_rgb_pipelined
_table_stream = [None]


def _patched(self, *a, **k):
    tbl = _table_stream[0]
    if tbl is None:
        return _orig(self, *a, **k)
    outer = torch.cuda.current_stream()
    tbl.wait_stream(outer)
    with torch.cuda.stream(tbl):
        sym = _orig(self, *a, **k)
    outer.wait_stream(tbl)
    sym.record_stream(outer)
    return sym


Bitcoding._decode_rgb_pipelined = _patched


def run(label, chain_range, table_range):
    bc._coder_streams = None
    if chain_range:
        bc.coder_cus = chain_range[1]
        bc._coder_range = chain_range
    else:
        bc.coder_cus = 0
    _table_stream[0] = _lib.cu_range_stream(*table_range) if table_range else None
    ts = []
    ok = True
    for rep in range(4):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        dec, _ = bc.decode_batch(files)
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
        ok = ok and bool(torch.equal(dec.to(torch.uint8), imgs.to(torch.uint8)))
        del dec
    print('{:58s} {}  lossless={}'.format(label, ' '.join('{:.3f}'.format(t) for t in ts), ok), flush=True)


run('product (chains and tables on all {} CUs)'.format(N_CU), None, None)
for n in (32, 64, 96, 128):
    run('chains on the last {} CUs, tables on the first {}'.format(n, N_CU - n), (N_CU - n, n), (0, N_CU - n))
run('chains on the last 96 CUs, tables everywhere', (N_CU - 96, 96), None)
run('chains everywhere, tables on the first 192 CUs', None, (0, 192))
run('product again', None, None)
