"""Generate tests/golden/*.npz from the UNMODIFIED reference (fab-jul/L3C-PyTorch @ /root/reference).

Run in the build container only:   python tests/golden/make_golden.py
The reference holds no golden vectors / KATs of its own for this path, so these fixtures ARE the
pin: every array below is an output of reference code (its C++ coder compiled by oracle/build_ref.py, its Python
modules imported through oracle/ref_import.py) on seeded inputs that the tests can regenerate.

  ac_kat.npz     range coder:   torchac_backend_cpu.encode_cdf / decode_cdf (torchac.cpp:263-269, :424-430)
  cdf_kat.npz    CDF tables:    torchac._get_uint16_cdf (torchac.py:174-178), Bitcoding._get_uniform_cdf (bitcoding.py:206-210)
  net_rgb_32x48.npz  RGB baselines (cr_rgb_shared with auto_recurse=3, cr_rgb): S pyramids (PIL bicubic), P, bpsp
  sample_32.npz  sampling:      MultiscaleBlueprint.sample_forward (multiscale_network.py:328-406) with the CPU generator seeded,
                                for the sample_scales the reference's tester uses ([], [0], [0, 1]; multiscale_tester.py:443-445)
                                and [0, 1, 2] (uniform prior on the coarsest scale)
  net_cal_32.npz, net_cal_64x96.npz   the net_32 fixture on the calibrated checkpoint + natural-like images (round 3)
  net_32.npz     config[0]:     MultiscaleBlueprint.forward / get_loss, DiscretizedMixLogisticLoss.cdf_step_non_shared,
                                MultiscaleNetwork.get_P, Bitcoding.encode -> file bytes, Bitcoding.decode
                                on one 32x32 image with the synthetic checkpoint (helpers/synthetic.py, seed 0)
"""
import os
import sys
import tempfile

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'oracle')]

import ref_import  # noqa: E402
import l3c_pytorch_amd  # noqa: E402,F401
from l3c_pytorch_amd.helpers import config_parser, synthetic  # noqa: E402

torch.set_num_threads(1)


# ---------------------------------------------------------------------------------------------------------------------
# coder KATs


def _random_table(rng, N, Lp, peaked=False):
    """strictly increasing rows in [0, 65535], like the mixture tables (first entry may be > 0)."""
    if peaked:
        w = np.full((N, Lp - 1), 1e-4)
        w[np.arange(N), rng.randint(0, Lp - 1, size=N)] = 1.0
    else:
        w = rng.gamma(0.3, size=(N, Lp - 1)) + 1e-3
    c = np.cumsum(w, axis=1)
    c = c / c[:, -1:]
    first = rng.randint(0, 300, size=(N, 1))
    body = np.round(c * (65535 - (Lp - 1) - first)).astype(np.int64) + np.arange(1, Lp) + first
    tab = np.concatenate([first, body], axis=1)
    assert (np.diff(tab, axis=1) > 0).all() and tab.max() <= 65535
    return tab.astype(np.uint16)


def _sample_symbols(rng, tab):
    """symbols roughly distributed like the table says (so the bitstream looks like a real one)."""
    N, Lp = tab.shape
    hi = np.concatenate([tab[:, 1:Lp - 1].astype(np.int64), np.full((N, 1), 65536)], axis=1)
    u = rng.randint(0, 65536, size=(N, 1))
    sym = (u >= hi).sum(axis=1)
    return np.clip(sym, 0, Lp - 2).astype(np.int16)


def ac_cases():
    rng = np.random.RandomState(1234)
    cases = {}

    def uniform(L):
        c = np.round(np.cumsum(np.full(L, np.float32(1) / np.float32(L), dtype=np.float32)) * 65536)
        return np.concatenate([[0], c]).astype(np.int64).astype(np.uint16)   # last entry wraps to 0

    for L, N in [(25, 96), (256, 77)]:
        row = uniform(L)
        tab = np.tile(row, (N, 1))
        cases['uniform_L%d' % L] = (tab, rng.randint(0, L, size=N).astype(np.int16))
    tab = _random_table(rng, 640, 257)
    cases['random_Lp257'] = (tab, _sample_symbols(rng, tab))
    tab = _random_table(rng, 1500, 26)
    cases['random_Lp26'] = (tab, _sample_symbols(rng, tab))
    tab = _random_table(rng, 900, 257, peaked=True)
    cases['peaked_Lp257'] = (tab, _sample_symbols(rng, tab))           # long runs of cheap symbols
    tab = _random_table(rng, 300, 257, peaked=True)
    cases['peaked_unlikely_Lp257'] = (tab, rng.randint(0, 256, size=300).astype(np.int16))  # mostly 1/65536-wide
    tab = _random_table(rng, 400, 26)
    cases['all_top_symbol_Lp26'] = (tab, np.full(400, 24, dtype=np.int16))
    cases['all_zero_symbol_Lp26'] = (tab.copy(), np.zeros(400, dtype=np.int16))
    tab = _random_table(rng, 1, 257)
    cases['single_symbol'] = (tab, np.array([200], dtype=np.int16))
    # interval straddling the midpoint for a long time -> many pending (underflow) bits
    tab = np.zeros((600, 3), dtype=np.uint16)
    tab[:, 1] = 32768
    tab[::2, 1] = 32767
    tab[1::2, 1] = 32769
    cases['underflow_Lp3'] = (tab, (np.arange(600) % 2 == 0).astype(np.int16))
    tab = _random_table(rng, 200, 257)
    tab[:, -1] = 0                                                      # wrapped final entry is never read
    cases['wrapped_last_entry'] = (tab, _sample_symbols(rng, tab))
    return cases


def make_ac_kat(backend):
    out = {}
    for name, (tab, sym) in ac_cases().items():
        N, Lp = tab.shape
        cdf_t = torch.from_numpy(tab.view(np.int16).copy()).reshape(1, 1, N, Lp)
        data = backend.encode_cdf(cdf_t, torch.from_numpy(sym))
        dec = backend.decode_cdf(cdf_t, data).numpy()
        assert (dec == sym).all(), name
        out[name + '/cdf'] = tab
        out[name + '/sym'] = sym
        out[name + '/bytes'] = np.frombuffer(data, dtype=np.uint8)
        print('  ac', name, N, Lp, len(data), 'bytes')
    # decoder behaviour on a truncated / foreign stream (zeros past EOF, torchac.cpp:104-108)
    tab, sym = ac_cases()['random_Lp257']
    cdf_t = torch.from_numpy(tab.view(np.int16).copy()).reshape(1, 1, *tab.shape)
    data = bytes(backend.encode_cdf(cdf_t, torch.from_numpy(sym)))[:200]
    out['truncated/cdf'] = tab
    out['truncated/bytes'] = np.frombuffer(data, dtype=np.uint8)
    out['truncated/decoded'] = backend.decode_cdf(cdf_t, data).numpy()
    np.savez_compressed(os.path.join(HERE, 'ac_kat.npz'), **out)


# ---------------------------------------------------------------------------------------------------------------------
# CDF KATs


def mixture_params(seed, K, H, W, rgb):
    g = torch.Generator().manual_seed(seed)
    logits = torch.randn(1, K, H, W, generator=g) * 2
    if rgb:
        mu = torch.rand(1, K, H, W, generator=g) * 300 - 20
        log_sigma = torch.rand(1, K, H, W, generator=g) * 9 - 7.5
    else:
        mu = torch.rand(1, K, H, W, generator=g) * 2.4 - 1.2
        log_sigma = torch.rand(1, K, H, W, generator=g) * 6 - 7.5
    return torch.softmax(logits, 1), mu, log_sigma.clamp(min=-7.)


def make_cdf_kat(ref_torchac, ref_bitcoding, ref_coders_helpers, ref_dmll_cls):
    out = {}
    for name, rgb, seed in [('rgb', True, 11), ('z', False, 12)]:
        dm = ref_dmll_cls(rgb_scale=rgb, x_min=0 if rgb else -1, x_max=255 if rgb else 1, L=256 if rgb else 25)
        targets = torch.linspace(dm.x_min - dm.bin_width / 2, dm.x_max + dm.bin_width / 2, dm.L + 1,
                                 dtype=torch.float32)   # coders_helpers.py:42-44
        pi, mu, ls = mixture_params(seed, 10, 8, 12, rgb)
        cdf = ref_torchac._get_uint16_cdf(pi, targets, mu, ls)
        out[name + '/targets'] = targets.numpy()
        out[name + '/pi'], out[name + '/mu'], out[name + '/log_sigma'] = pi.numpy(), mu.numpy(), ls.numpy()
        out[name + '/cdf'] = cdf.numpy().view(np.uint16)
        print('  cdf', name, cdf.shape)
    for L in (25, 256):
        u = ref_bitcoding._get_cdf_from_pr(ref_bitcoding._get_uniform_pr((1, 5, 3, 4), L))
        out['uniform_L%d' % L] = u.numpy().view(np.uint16)
    np.savez_compressed(os.path.join(HERE, 'cdf_kat.npz'), **out)


# ---------------------------------------------------------------------------------------------------------------------
# network / container fixture (config[0]: one 32x32 image, CPU)


def make_net_fixture(H=32, W=32, seed=0, img_seed=0, kind='uniform', calibrated=False, fname='net_32.npz', p0_stride=1):
    cfg_mine = config_parser.parse_builtin('ms', 'cr')
    sd = synthetic.make_state_dict(cfg_mine, seed, calibrated=calibrated)
    img = synthetic.make_image(H, W, img_seed, kind).unsqueeze(0).long()
    out = {'p0_stride': np.array(p0_stride)}
    with ref_import.reference_modules():
        from fjcommon import config_parser as rcp, no_op
        from blueprints.multiscale_blueprint import MultiscaleBlueprint
        from bitcoding.bitcoding import Bitcoding
        from bitcoding import coders_helpers
        cfg, _ = rcp.parse('configs/ms/cr.cf')
        bp = MultiscaleBlueprint(cfg)
        bp.net.load_state_dict(sd, strict=True)
        bp.set_eval()
        with torch.no_grad():
            o = bp.forward(img.float())
            loss = bp.get_loss(o)
            out['img'] = img.numpy().astype(np.uint8)
            for s in range(4):
                out['S%d' % s] = o.S[s].numpy().astype(np.int16)
            for s in range(3):
                out['P%d' % s] = (o.P[s][:, :, ::p0_stride, ::p0_stride] if s == 0 else o.P[s]).numpy().copy()
                out['bn%d' % (s + 1)] = o.bn[s + 1].numpy()
            out['bpsp'] = np.array([float(b) for b in loss.nonrecursive_bpsps], dtype=np.float64)
            # stage-wise intermediates (P2 parity ladder: feed each stage the reference's own inputs)
            x = bp.net.sub_rgb_mean(img.float())
            inp = x
            for s in range(3):
                inp = bp.net.heads[s](inp)
                if p0_stride == 1:
                    out['enc_in%d' % s] = inp.numpy().copy()
                e = bp.net.nets[s].enc(inp)
                if p0_stride == 1 or s > 0:
                    out['enc_F%d' % s] = e.F.numpy().copy()
                out['enc_bn%d' % s] = e.bn.numpy().copy() if False else bp.net.nets[s].enc.to_q(e.F).numpy().copy()
                inp = e.F
            f_prev = None
            for s in (2, 1, 0):
                P, f_prev = bp.net.get_P(s, o.bn[s + 1], f_prev)
                assert (P == o.P[s]).all()
                if p0_stride == 1 or s > 0:
                    out['dec_F%d' % s] = f_prev.numpy().copy()
            # per-channel coding parameters (logistic_mixture.py:134-141); clone P: the reference mutates it in place
            for s, dm, C, x_c in [(0, bp.losses.loss_dmol_rgb, 3, img.float()),
                                  (1, bp.losses.loss_dmol_n, 5, o.bn[1])]:
                helper = coders_helpers.CodingCDFNonshared(o.P[s].clone(), total_C=C, dmll=dm)
                for c in range(C):
                    cd = helper.get_next_C(x_c)
                    if p0_stride != 1 and s == 0:
                        if c == 2:      # the doubly lambda-coupled means only (fixture size)
                            out['cdfout0_c2/mu'] = cd.means_c[:, :, ::p0_stride, ::p0_stride].numpy().copy()
                        continue
                    if c == 0 or (s == 1 and c == 4):
                        out['cdfout%d_c%d/pi' % (s, c)] = cd.logit_probs_c_sm.numpy().copy()
                        out['cdfout%d_c%d/log_sigma' % (s, c)] = cd.log_scales_c.numpy().copy()
                    if s == 0 or c in (0, 4):
                        out['cdfout%d_c%d/mu' % (s, c)] = cd.means_c.numpy().copy()
                out['targets%d' % s] = helper.targets.numpy()
            # the container
            bc = Bitcoding(bp, times=no_op.NoOp)
            with tempfile.TemporaryDirectory() as d:
                p = os.path.join(d, 'x.l3c')
                bpsp_file = bc.encode(img.clone(), p)
                data = open(p, 'rb').read()
                dec = bc.decode(p)
            assert (dec == img).all()
            out['l3c'] = np.frombuffer(data, dtype=np.uint8)
            out['l3c_bpsp'] = np.array(bpsp_file)
            print('  net: bpsp', out['bpsp'], 'file', len(data), 'bytes ->', bpsp_file)
    np.savez_compressed(os.path.join(HERE, fname), **out)


def make_rgb_fixtures():
    """RGB baselines (config 5 family): cr_rgb_shared with auto_recurse=3 and cr_rgb, forward + get_loss on a 32x48 image."""
    out = {}
    img = synthetic.make_image(32, 48, 2, 'natural').unsqueeze(0).long()
    out['img'] = img.numpy().astype(np.uint8)
    for name, recurse in [('cr_rgb_shared', 3), ('cr_rgb', 0)]:
        cfg_mine = config_parser.parse_builtin('ms', name)
        sd = synthetic.make_state_dict(cfg_mine, 0)
        with ref_import.reference_modules():
            from fjcommon import config_parser as rcp
            from blueprints.multiscale_blueprint import MultiscaleBlueprint
            cfg, _ = rcp.parse('configs/ms/{}.cf'.format(name))
            bp = MultiscaleBlueprint(cfg)
            bp.net.load_state_dict(sd, strict=True)
            bp.set_eval()
            with torch.no_grad():
                o = bp.forward(img.float(), recurse)
                loss = bp.get_loss(o)
            for i, S in enumerate(o.S):
                out['{}/S{}'.format(name, i)] = S.numpy().astype(np.int16)
            for i, P in enumerate(o.P):   # the finest scale every 4th pixel only (fixture size)
                out['{}/P{}'.format(name, i)] = (P[:, :, ::4, ::4] if i == 0 else P).numpy().copy()
            out[name + '/nonrecursive_bpsps'] = np.array([float(b) for b in loss.nonrecursive_bpsps])
            if loss.recursive_bpsps is not None:
                out[name + '/recursive_bpsps'] = np.array([float(b) for b in loss.recursive_bpsps])
            print('  rgb', name, [tuple(S.shape) for S in o.S], [float(b) for b in loss.nonrecursive_bpsps])
    np.savez_compressed(os.path.join(HERE, 'net_rgb_32x48.npz'), **out)


def make_sample_fixture(H=32, W=32, seed=0, img_seed=3):
    """The draws come from torch's CPU generator: torch.manual_seed(SEED) immediately before the call, so a restatement that
    draws tensors of the same shapes in the same order (oracle.net.default_draw) sees the same uniforms."""
    cfg_mine = config_parser.parse_builtin('ms', 'cr')
    sd = synthetic.make_state_dict(cfg_mine, seed)
    img = synthetic.make_image(H, W, img_seed, 'natural').unsqueeze(0).long()
    out = {'img': img.numpy().astype(np.uint8)}
    with ref_import.reference_modules():
        from fjcommon import config_parser as rcp
        from blueprints.multiscale_blueprint import MultiscaleBlueprint
        cfg, _ = rcp.parse('configs/ms/cr.cf')
        bp = MultiscaleBlueprint(cfg)
        bp.net.load_state_dict(sd, strict=True)
        bp.set_eval()
        with torch.no_grad():
            for i, scales in enumerate(([], [0], [0, 1], [0, 1, 2])):
                torch.manual_seed(100 + i)
                x = bp.sample_forward(img.float(), scales)
                out['scales%d' % i] = np.array(scales, dtype=np.int64)
                out['seed%d' % i] = np.array(100 + i)
                out['x%d' % i] = x.numpy().copy()
                print('  sample', scales, tuple(x.shape), float(x.min()), float(x.max()))
    np.savez_compressed(os.path.join(HERE, 'sample_32.npz'), **out)


def make_calibrated_fixtures():
    """The same fixture on the CALIBRATED checkpoint (helpers/synthetic.make_state_dict(calibrated=True), fitted by
    make_calibrated.py) and natural-like images: mixtures that cover the data, every bottleneck level in use -- the regime in which
    a P error can move a table entry and a quantiser boundary.  The real reference encodes and decodes them (asserted above)."""
    make_net_fixture(32, 32, 0, 5, 'natural', True, 'net_cal_32.npz')
    make_net_fixture(64, 96, 0, 6, 'natural', True, 'net_cal_64x96.npz', p0_stride=2)


def main():
    if '--calibrated-only' in sys.argv:
        make_calibrated_fixtures()
        return
    with ref_import.reference_modules():
        import torchac_backend_cpu
        from torchac import torchac as ref_torchac
        from bitcoding import bitcoding as ref_bitcoding, coders_helpers
        from criterion.logistic_mixture import DiscretizedMixLogisticLoss
        print('coder KATs')
        make_ac_kat(torchac_backend_cpu)
        print('CDF KATs')
        make_cdf_kat(ref_torchac, ref_bitcoding, coders_helpers, DiscretizedMixLogisticLoss)
    print('network fixture')
    make_net_fixture()
    print('RGB baseline fixtures')
    make_rgb_fixtures()
    print('sampling fixture')
    make_sample_fixture()
    print('calibrated-checkpoint fixtures')
    make_calibrated_fixtures()
    for f in sorted(os.listdir(HERE)):
        if f.endswith('.npz'):
            print(f, os.path.getsize(os.path.join(HERE, f)), 'bytes')


if __name__ == '__main__':
    main()
