#!/usr/bin/env python
"""Development: time of the fused interval kernel (l3c_dmll_encode_intervals) on the headline shapes, and a checksum of its output
(variants must agree bit for bit).   L3C_LIB=.../libl3c_hip_<variant>.so python tools/interval_probe.py [B]"""
import os
import sys
import zlib

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import l3c_pytorch_amd  # noqa: E402,F401
from l3c_pytorch_amd import ops  # noqa: E402
from oracle import cdf as ocdf  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
ONLY = sys.argv[2] if len(sys.argv) > 2 else ''
g = torch.Generator(device='cuda').manual_seed(0)
for name, (H, W, C, L, lo, hi, rgb) in (('RGB scale 512x768', (512, 768, 3, 256, 0, 255, True)), ('z scale 256x384', (256, 384, 5, 25, -1, 1, False))):
    if ONLY and ONLY not in name:
        continue
    Kp = (4 if rgb else 3) * C * 10
    P = torch.randn(B, H, W, Kp, device='cuda', generator=g)
    if rgb:
        P[..., 30:60] = torch.rand(B, H, W, 30, device='cuda', generator=g) * 280 - 10
        P[..., 60:90] = torch.rand(B, H, W, 30, device='cuda', generator=g) * 4 - 1
    sym = torch.randint(0, L, (B, C, H, W), device='cuda', generator=g).to(torch.int16)
    t = ocdf.coding_targets(lo, hi, L).cuda()
    for _ in range(2):
        iv = ops.dmll_encode_intervals(P, sym, t, C, 10, rgb)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        iv = ops.dmll_encode_intervals(P, sym, t, C, 10, rgb)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 5
    print('{:20s} B={:3d}: {:7.3f} ms per launch = {:6.1f} M symbols/s, checksum {:08x}'.format(
        name, B, ms, B * C * H * W / ms / 1e3, zlib.crc32(iv.cpu().numpy().tobytes())))
    del P, sym, iv
