#!/usr/bin/env python
"""Ground truth under the parity gate (round-4 verdict, item 3): a THREE-WAY table per parameter group of P

    |HIP - fp64|      |oracle (torch-CPU fp32) - fp64|      |HIP - oracle|

where fp64 is the oracle's own forward (oracle/net.py: the reference's operations in the reference's order) evaluated in
double precision on the same weights.  The decoder side of all three is driven with the SAME bottlenecks (the fp32 oracle's
bn_q, exact quantiser levels in any precision), coarse to fine, each side carrying its own decoder features along -- as
tests/test_gpu_headline.py does -- so that a quantiser near-tie cannot void the comparison.  Cases: bench image 0 at 768x512 on
both synthetic checkpoints, and the reference-generated 64x96 fixture (tests/golden/net_cal_64x96.npz; there the "oracle" column
is the REFERENCE's own P).

    python tools/parity_truth.py [--out gpurun_out/parity_truth.json] [--no-hip] [--size H W]

Runs on the GPU box (HIP side through the product's get_P); with --no-hip only the CPU columns (works in the build container).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

import l3c_pytorch_amd  # noqa: E402,F401
from l3c_pytorch_amd.helpers import config_parser, synthetic  # noqa: E402
from oracle import net as onet  # noqa: E402
from tests.parity_gate import GROUPS  # noqa: E402


def group_table(a, b, num_params):
    """max |a - b| per parameter group (both (N, Kp, H, W), any float dtype), and the group's largest |b|"""
    a, b = a.double(), b.double()
    n = a.shape[1] // num_params
    out = {}
    for p, name in enumerate(GROUPS[:num_params]):
        d = (a[:, p * n:(p + 1) * n] - b[:, p * n:(p + 1) * n]).abs().max().item()
        out[name] = {'max_abs': d, 'max_value': b[:, p * n:(p + 1) * n].abs().max().item()}
    return out


def decoder_chain(bn_q, sd, dtype):
    """get_P coarse to fine on the given bottlenecks in `dtype` -> P[s], F_dec[s] (oracle/net.py:get_P)"""
    sd_t = {k: v.to(dtype) for k, v in sd.items()}
    P, Fd, f = [None] * 3, [None] * 3, None
    with torch.no_grad():
        for s in (2, 1, 0):
            P[s], f = onet.get_P(s, bn_q[s + 1].to(dtype), f, sd_t)
            Fd[s] = f
    return P, Fd


def hip_chain(bn_q, cfg, sd):
    from l3c_pytorch_amd.blueprints.multiscale_blueprint import MultiscaleBlueprint
    bp = MultiscaleBlueprint(cfg)
    bp.net.load_state_dict(sd, strict=True)
    bp.set_eval()
    P, Fd, f = [None] * 3, [None] * 3, None
    for s in (2, 1, 0):
        p, f = bp.net.get_P(s, bn_q[s + 1].cuda(), f)
        P[s], Fd[s] = p.cpu(), f.cpu()
    return P, Fd, bp


def three_way(P_h, P_o, P_64, Fd_h, Fd_o, Fd_64):
    rec = {}
    for s in (2, 1, 0):
        npar = 4 if s == 0 else 3
        row = {'oracle_vs_fp64': group_table(P_o[s], P_64[s], npar)}
        if P_h is not None:
            row['hip_vs_fp64'] = group_table(P_h[s], P_64[s], npar)
            row['hip_vs_oracle'] = group_table(P_h[s], P_o[s], npar)
        f = {'oracle_vs_fp64': (Fd_o[s].double() - Fd_64[s]).abs().max().item(), 'max_value': Fd_64[s].abs().max().item()}
        if Fd_h is not None:
            f['hip_vs_fp64'] = (Fd_h[s].double() - Fd_64[s]).abs().max().item()
            f['hip_vs_oracle'] = (Fd_h[s].double() - Fd_o[s].double()).abs().max().item()
        row['F_dec'] = f
        rec['scale%d' % s] = row
    return rec


def flat_gate(rec):
    """the gate the table supports: |x - fp64| < 1e-5 flat for every group whose largest value is <= 16, relative 1e-5 above"""
    worst = {}
    for side in ('hip_vs_fp64', 'oracle_vs_fp64'):
        w = 0.0
        for s, row in rec.items():
            if side not in row:
                continue
            for name, g in row[side].items():
                tol = 1e-5 * max(1.0, g['max_value'] / 16.0)
                w = max(w, g['max_abs'] / tol)
        worst[side] = w
    return worst


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--out', default=os.path.join(ROOT, 'gpurun_out', 'parity_truth.json'))
    ap.add_argument('--no-hip', action='store_true')
    ap.add_argument('--size', type=int, nargs=2, default=[512, 768])
    ap.add_argument('--threads', type=int, default=0)
    a = ap.parse_args()
    if a.threads:
        torch.set_num_threads(a.threads)
    cfg = config_parser.parse_builtin('ms', 'cr')
    H, W = a.size
    result = {'size': [H, W], 'threads': torch.get_num_threads(), 'cases': {}}
    img = synthetic.make_image(H, W, 0, 'natural').unsqueeze(0).float()
    for ckpt in ('calibrated', 'default'):
        sd = synthetic.make_state_dict(cfg, 0, calibrated=ckpt == 'calibrated')
        t0 = time.time()
        with torch.no_grad():
            o = onet.forward(img, sd)
        t1 = time.time()
        P_64, Fd_64 = decoder_chain(o.bn, sd, torch.float64)
        t2 = time.time()
        P_h = Fd_h = None
        if not a.no_hip:
            P_h, Fd_h, _ = hip_chain(o.bn, cfg, sd)
        rec = three_way(P_h, o.P, P_64, Fd_h, o.F_dec, Fd_64)
        case = {'table': rec, 'worst_over_flat_gate': flat_gate(rec), 'seconds': {'oracle_fp32_forward': t1 - t0, 'fp64_decoder_chain': t2 - t1}}
        result['cases']['%dx%d_%s' % (W, H, ckpt)] = case
        print(ckpt, json.dumps(case['worst_over_flat_gate']), flush=True)
    # the reference-generated 64x96 fixture (calibrated checkpoint): its P is the REFERENCE's
    fx = os.path.join(ROOT, 'tests', 'golden', 'net_cal_64x96.npz')
    if os.path.isfile(fx):
        sd = synthetic.make_state_dict(cfg, 0, calibrated=True)
        with np.load(fx) as z:
            names = set(z.files)
            bn = [None] + [torch.from_numpy(z['bn%d' % s]) for s in (1, 2, 3)] if 'bn1' in names else None
            P_ref = [torch.from_numpy(z['P%d' % s]) for s in range(3)] if 'P0' in names else None
            st0 = int(z['p0_stride']) if 'p0_stride' in names else 1      # the fixture keeps every st0-th pixel of P0
            keys = sorted(names)
        if bn is None or P_ref is None:
            result['fixture_64x96'] = {'skipped': 'unexpected fixture keys', 'keys': keys}
        else:
            P_64, Fd_64 = decoder_chain(bn, sd, torch.float64)
            P_o, Fd_o = decoder_chain(bn, sd, torch.float32)
            P_h = Fd_h = None
            if not a.no_hip:
                P_h, Fd_h, _ = hip_chain(bn, cfg, sd)
            sub = lambda Ps: None if Ps is None else [Ps[0][..., ::st0, ::st0], Ps[1], Ps[2]]   # noqa: E731
            P_h, P_64, P_o = sub(P_h), sub(P_64), sub(P_o)
            rec = three_way(P_h, P_ref, P_64, Fd_h, Fd_o, Fd_64)
            result['cases']['fixture_64x96_calibrated'] = {
                'table': rec, 'worst_over_flat_gate': flat_gate(rec),
                'oracle_equals_reference': all(torch.equal(P_o[s], P_ref[s]) for s in range(3))}
            print('fixture', json.dumps(flat_gate(rec)), flush=True)
    os.makedirs(os.path.dirname(a.out), exist_ok=True)
    with open(a.out, 'w') as f:
        json.dump(result, f, indent=1, sort_keys=True)
    print('written', a.out)


if __name__ == '__main__':
    main()
