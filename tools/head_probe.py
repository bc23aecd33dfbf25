#!/usr/bin/env python
"""Development probe: time the fused encoder head (l3c_dmll_encode_intervals) on random P at the RGB scale and a bottleneck scale."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import l3c_pytorch_amd  # noqa: E402,F401
from l3c_pytorch_amd import ops  # noqa: E402
from l3c_pytorch_amd.criterion.logistic_mixture import DiscretizedMixLogisticLoss  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
for name, (H, W, C, rgb, dm) in {'rgb 512x768': (512, 768, 3, True, DiscretizedMixLogisticLoss(True, 0, 255, 256)),
                                 'z 256x384': (256, 384, 5, False, DiscretizedMixLogisticLoss(False, -1, 1, 25))}.items():
    Kp = (4 if rgb else 3) * C * 10
    g = torch.Generator().manual_seed(0)
    P = torch.randn(B, H, W, Kp, generator=g).cuda()
    sym = torch.randint(0, dm.L, (B, C, H, W), generator=g).to(torch.int16).cuda()
    t = dm.coding_targets('cuda')
    for _ in range(2):
        ops.dmll_encode_intervals(P, sym, t, C, 10, rgb)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        iv = ops.dmll_encode_intervals(P, sym, t, C, 10, rgb)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 5
    print('{} B{}: {:.3f} ms  ({:.2f} TB/s of P read, {:.1f} Msym/s)  checksum {}'.format(
        name, B, ms, P.numel() * 4 / ms / 1e9, B * C * H * W / ms / 1e3, int(iv.to(torch.int64).sum())))
