"""Development: l3c_to_q_quantize alone (1x1 Cf -> C + hard quantiser) at the three bottleneck scales of a batch of 128 and two small shapes;
prints the time per call and a hash of (sym, bn_q, bn) -- run on the product library and on a variant (L3C_LIB=...) to compare both."""
import os, sys, time
sys.path.insert(0, os.getcwd())
import torch
from l3c_pytorch_amd import ops
torch.manual_seed(0)
lev = (torch.arange(25).float() * (2 / 24) - 1).cuda()
for (B, H, W) in ((128, 256, 384), (128, 128, 192), (128, 64, 96), (3, 40, 56), (1, 8, 24)):
    feat = torch.randn(B, H, W, 64, device='cuda')
    w = (torch.randn(5, 64, device='cuda') * 0.1).contiguous(); b = torch.randn(5, device='cuda') * 0.1
    out = ops.to_q_quantize(feat, w, b, lev, want_bn=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        out = ops.to_q_quantize(feat, w, b, lev, want_bn=True)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 10
    import hashlib
    h = hashlib.sha1(b''.join(t.cpu().numpy().tobytes() for t in out)).hexdigest()[:12]
    print(os.environ.get('L3C_LIB', 'product')[-22:], (B, H, W), '%.3f ms' % (dt * 1e3), h, flush=True)
