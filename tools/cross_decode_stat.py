#!/usr/bin/env python
"""Cross-implementation decode statistic (round-3 verdict, weak 9 / next 3d): how often does a `.l3c` written by one implementation
decode with the other?  INFORMATIONAL -- SURVEY.md section 8c: two implementations of P never guarantee identical uint16 tables (one
entry off by one at a coded symbol derails the rest of that stream), and the reference itself changes its bitstream with the CPU
thread count.  A HIP-written file is only GUARANTEED to decode with the HIP decoder (tests: lossless for every size).

    python tools/cross_decode_stat.py [--n 24] [--h 256] [--w 384] [--full 2] [--out gpurun_out/cross_decode.json]

For each of `n` bench images (synthetic `natural`, seed = index; cropped to h x w so that the CPU oracle finishes in seconds) and
`full` images at the bench size 512 x 768, on the calibrated checkpoint:
  hip -> oracle:  oracle.bitcoding.decode(HIP file)  == image ?    (wrong sub-pixels counted)
  oracle -> hip:  Bitcoding.decode_batch(oracle file) == image ?
and the sizes of the two files.  The oracle is the checker here (tools/ is not the product)."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--n', type=int, default=24)
    ap.add_argument('--h', type=int, default=256)
    ap.add_argument('--w', type=int, default=384)
    ap.add_argument('--full', type=int, default=2)
    ap.add_argument('--threads', type=int, default=16)
    ap.add_argument('--out', default=os.path.join(ROOT, 'gpurun_out', 'cross_decode.json'))
    a = ap.parse_args()
    import l3c_pytorch_amd  # noqa: F401
    from l3c_pytorch_amd.bitcoding.bitcoding import Bitcoding
    from l3c_pytorch_amd.blueprints.multiscale_blueprint import MultiscaleBlueprint
    from l3c_pytorch_amd.helpers import config_parser, synthetic
    from oracle import bitcoding as obc
    torch.set_num_threads(a.threads)
    cfg = config_parser.parse_builtin('ms', 'cr')
    sd = synthetic.make_state_dict(cfg, 0, calibrated=True)
    bp = MultiscaleBlueprint(cfg)
    bp.net.load_state_dict(sd, strict=True)
    bp.set_eval()
    bc = Bitcoding(bp)
    rows = []
    t0 = time.time()
    cases = [(i, a.h, a.w) for i in range(a.n)] + [(i, 512, 768) for i in range(a.full)]
    for seed, h, w in cases:
        img = synthetic.make_image(512, 768, seed, 'natural')[:, :h, :w].contiguous().unsqueeze(0).long()
        hip_file = bc.encode_batch(img).to_bytes()[0]
        with torch.no_grad():
            ora_file = obc.encode(img, sd)
            dec_o, _ = obc.decode(hip_file, sd)
        dec_h, _ = bc.decode_batch([ora_file])
        own, _ = bc.decode_batch([hip_file])
        assert torch.equal(own.cpu(), img), 'HIP round trip must be lossless'
        rows.append({'seed': seed, 'size': [h, w], 'hip_bytes': len(hip_file), 'oracle_bytes': len(ora_file),
                     'oracle_decodes_hip_wrong_subpixels': int((dec_o != img).sum()),
                     'hip_decodes_oracle_wrong_subpixels': int((dec_h.cpu() != img).sum()), 'subpixels': int(img.numel())})
        print(rows[-1], flush=True)
    by = {}
    for r in rows:
        k = '{}x{}'.format(*r['size'])
        d = by.setdefault(k, {'images': 0, 'oracle_decodes_hip_file': 0, 'hip_decodes_oracle_file': 0, 'sizes_equal': 0})
        d['images'] += 1
        d['oracle_decodes_hip_file'] += r['oracle_decodes_hip_wrong_subpixels'] == 0
        d['hip_decodes_oracle_file'] += r['hip_decodes_oracle_wrong_subpixels'] == 0
        d['sizes_equal'] += r['hip_bytes'] == r['oracle_bytes']
    res = {'what': 'cross-implementation decode of .l3c files, calibrated checkpoint; counts of images decoded without a single wrong sub-pixel',
           'summary': by, 'seconds': round(time.time() - t0, 1), 'images': rows}
    os.makedirs(os.path.dirname(a.out), exist_ok=True)
    with open(a.out, 'w') as f:
        json.dump(res, f, indent=1)
    print(json.dumps(by))


if __name__ == '__main__':
    main()
