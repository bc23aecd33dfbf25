"""-m gpu: the logistic-mixture head kernels (csrc/dmll_kernels.hip) against the reference fixtures and the oracle."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import ac as oracle_ac, cdf as ocdf, dmll as odmll  # noqa: E402


def test_cdf_table_vs_reference_kat(golden):
    """fp32 path: equal to the reference's table except +-1 where frac(cdf*65280) is within float error of .5
    (SURVEY.md section 8c, P1).  Tolerance: |diff| <= 1 on < 0.2 % of the entries; strictly increasing rows."""
    from l3c_pytorch_amd import ops
    g = golden('cdf_kat.npz')
    for name in ('rgb', 'z'):
        args = [torch.from_numpy(g[name + '/' + k]).cuda() for k in ('targets', 'pi', 'mu', 'log_sigma')]
        table, flag = ops.cdf_table_mixture(*args)
        assert int(flag.item()) == 0
        got = table.cpu().numpy().view(np.uint16).astype(np.int64)
        ref = g[name + '/cdf'].astype(np.int64)
        d = np.abs(got - ref)[..., :-1]
        assert d.max() <= 1, (name, d.max())
        assert (d != 0).mean() < 2e-3, (name, (d != 0).mean())
        assert (np.diff(got[..., :-1], axis=-1) > 0).all()


def _rand_P(rng, B, H, W, C, K, rgb):
    Kp = (4 if rgb else 3) * C * K
    P = rng.randn(B, H, W, Kp).astype(np.float32)
    CK = C * K
    if rgb:
        P[..., CK:2 * CK] = rng.uniform(-20, 280, size=(B, H, W, CK))
        P[..., 2 * CK:3 * CK] = rng.uniform(-9, 2, size=(B, H, W, CK))
    else:
        P[..., CK:2 * CK] = rng.uniform(-1.2, 1.2, size=(B, H, W, CK))
        P[..., 2 * CK:3 * CK] = rng.uniform(-9, -1, size=(B, H, W, CK))
    return P


@pytest.mark.parametrize('rgb,C,H,W', [(True, 3, 9, 13), (False, 5, 8, 8), (True, 3, 16, 20), (False, 5, 3, 5)])
def test_fused_intervals_equal_table_intervals(rgb, C, H, W):
    """The encoder's fused head and the decoder's params->table path must give IDENTICAL intervals (bit-exact), and the
    stream coded from them must decode losslessly through the table."""
    from l3c_pytorch_amd import ops
    rng = np.random.RandomState(H * 100 + W)
    B, K = 2, 10
    L = 256 if rgb else 25
    spec = odmll.RGB if rgb else odmll.z_spec()
    P = torch.from_numpy(_rand_P(rng, B, H, W, C, K, rgb)).cuda()
    sym = torch.from_numpy(rng.randint(0, L, size=(B, C, H, W)).astype(np.int16)).cuda()
    sym[0, :, 0, 0] = L - 1                                     # top symbol -> c_high = 0x10000
    sym[0, :, 0, 1] = 0
    targets = ocdf.coding_targets(spec.x_min, spec.x_max, spec.L).cuda()
    iv_fused = ops.dmll_encode_intervals(P, sym, targets, C, K, rgb).cpu().numpy()
    tabs = []
    for c in range(C):
        pi, mu, ls = ops.dmll_channel_params(P, sym, C, K, rgb, c)
        table, flag = ops.cdf_table_mixture(targets, pi, mu, ls)
        assert int(flag.item()) == 0
        tabs.append(table)
    table_all = torch.stack(tabs, dim=1).reshape(B * C * H * W, L + 1)          # stream order b*C + c
    iv_table = ops.intervals_from_table(table_all, sym.reshape(B * C, H * W), B * C, H * W).cpu().numpy()
    assert (iv_fused == iv_table).all(), int((iv_fused != iv_table).sum())
    out, n = ops.ac_encode(torch.from_numpy(iv_fused).cuda(), B * C, H * W)
    n, out = n.cpu().numpy(), out.cpu().numpy()
    payloads = [out[i, :n[i]].tobytes() for i in range(B * C)]
    buf, offs, lens = ops.pack_streams(payloads)
    dec = ops.ac_decode(table_all, buf, offs, lens, B * C, H * W, True).reshape(B, C, H, W)
    assert torch.equal(dec, sym)
    # and the bytes are what the oracle coder produces from the same tables
    t_np = table_all.cpu().numpy().view(np.uint16).reshape(B * C, H * W, L + 1)
    s_np = sym.cpu().numpy().reshape(B * C, H * W)
    for s in range(B * C):
        assert payloads[s] == oracle_ac.encode(t_np[s], s_np[s]), s


def test_channel_params_vs_reference_fixture(golden):
    """CDFOut of the reference (logistic_mixture.py:134-141) on the reference's own P: log_sigma and the uncoupled means bit-equal,
    pi within 1e-6; the lambda-coupled RGB means mu + sigmoid(lambda) x (x up to 255, values up to ~500: 1 ulp = 3e-5) within
    1e-4 absolute = 3 ulp (the sigmoid's expf differs in the last bit between the CPU and the GPU)."""
    from l3c_pytorch_amd import ops
    g = golden('net_32.npz')
    P0 = torch.from_numpy(g['P0']).cuda().permute(0, 2, 3, 1).contiguous()
    sym0 = torch.from_numpy(g['img'].astype(np.int16)).cuda()
    for c in range(3):
        pi, mu, ls = ops.dmll_channel_params(P0, sym0, 3, 10, True, c)
        err = np.abs(mu.cpu().numpy() - g['cdfout0_c%d/mu' % c]).max()
        print('channel {}: max |mu - reference| = {:.3g} (max |mu| {:.1f})'.format(c, err, np.abs(g['cdfout0_c%d/mu' % c]).max()))
        assert err <= (0 if c == 0 else 1e-4), (c, err)
        if c == 0:
            assert np.allclose(pi.cpu().numpy(), g['cdfout0_c0/pi'], atol=1e-6)
            assert (ls.cpu().numpy() == g['cdfout0_c0/log_sigma']).all()
    P1 = torch.from_numpy(g['P1']).cuda().permute(0, 2, 3, 1).contiguous()
    for c in (0, 4):
        pi, mu, ls = ops.dmll_channel_params(P1, None, 5, 10, False, c)
        assert np.allclose(pi.cpu().numpy(), g['cdfout1_c%d/pi' % c], atol=1e-6)
        assert (mu.cpu().numpy() == g['cdfout1_c%d/mu' % c]).all()


def test_nll_vs_oracle_and_fixture_bpsp(golden):
    """bpsp per scale within 1e-5 relative of the reference's get_loss on the reference's P (P3 of the parity ladder)."""
    from l3c_pytorch_amd.criterion.logistic_mixture import DiscretizedMixLogisticLoss
    g = golden('net_32.npz')
    rgb = DiscretizedMixLogisticLoss(True, 0, 255, 256)
    z = DiscretizedMixLogisticLoss(False, -1, 1, 25)
    img = torch.from_numpy(g['img'].astype(np.float32)).cuda()
    n_sub = img.numel()
    for s, (loss, spec) in enumerate([(rgb, odmll.RGB), (z, odmll.z_spec()), (z, odmll.z_spec())]):
        P = torch.from_numpy(g['P%d' % s]).cuda()
        x = img if s == 0 else torch.from_numpy(g['bn%d' % s]).cuda()
        nll = loss(x, P).cpu()
        ref = odmll.nll(spec, x.cpu(), P.cpu())
        assert torch.allclose(nll, ref, rtol=2e-5, atol=2e-5), (s, (nll - ref).abs().max())
        bpsp = float(nll.sum()) / (np.log(2.) * n_sub)
        assert abs(bpsp - g['bpsp'][s]) <= 1e-5 * g['bpsp'][s], (s, bpsp, g['bpsp'][s])


def test_nll_edge_symbols():
    """x at x_min / x_max take the log_cdf_plus / log_one_minus_cdf_min branches (logistic_mixture.py:192-200)."""
    from l3c_pytorch_amd import ops
    rng = np.random.RandomState(9)
    P = torch.from_numpy(_rand_P(rng, 1, 4, 6, 3, 10, True))
    x = torch.from_numpy(rng.randint(0, 256, size=(1, 3, 4, 6)).astype(np.float32))
    x[0, :, 0, :3] = 0.
    x[0, :, 1, :3] = 255.
    got = ops.dmll_nll(P.cuda(), x.cuda(), 3, 10, True, 0, 255, 256).cpu()
    ref = odmll.nll(odmll.RGB, x, P.permute(0, 3, 1, 2).contiguous())
    assert torch.allclose(got, ref, rtol=2e-5, atol=2e-5), (got - ref).abs().max()


@pytest.mark.parametrize('rgb,C,H,W', [(True, 3, 9, 13), (False, 5, 16, 24), (True, 3, 64, 70)])
def test_sample_kernel_vs_oracle(rgb, C, H, W):
    """l3c_dmll_sample vs the oracle's restatement of _non_shared_sample on identical P and identical uniforms.  fp32 with
    device log/exp: values within 2e-3 absolute (values reach 255, the logistic tail multiplies log errors by sigma <= e^2)
    wherever both picked the same mixture component; the Gumbel-max may flip where the two best candidates are within
    rounding of each other -- tolerated on < 0.1 % of the pixel-channels."""
    from l3c_pytorch_amd import ops
    rng = np.random.RandomState(H * W)
    B, K = 2, 10
    P = _rand_P(rng, B, H, W, C, K, rgb)
    u_mix = rng.uniform(1e-5, 1 - 1e-5, size=(B, C, K, H, W)).astype(np.float32)
    u_log = rng.uniform(1e-5, 1 - 1e-5, size=(B, C, H, W)).astype(np.float32)
    got = ops.dmll_sample(torch.from_numpy(P).cuda(), torch.from_numpy(u_mix).cuda(), torch.from_numpy(u_log).cuda(),
                          C, K, rgb).cpu()
    spec = odmll.RGB if rgb else odmll.z_spec()
    want = odmll.sample(spec, torch.from_numpy(P).permute(0, 3, 1, 2).contiguous(), C, torch.from_numpy(u_mix),
                        torch.from_numpy(u_log))
    assert got.shape == want.shape == (B, C, H, W)
    bad = ((got - want).abs() > 2e-3).float().mean().item()
    assert bad < 1e-3, bad
    if rgb:
        assert got.min() >= 0 and got.max() <= 255


@pytest.mark.parametrize('rgb,C,c', [(True, 3, 0), (True, 3, 2), (False, 5, 3)])
def test_fused_table_rows_equal_the_two_kernel_table(rgb, C, c):
    """l3c_dmll_cdf_table (P -> rows of a pixel range, fused) == l3c_dmll_channel_params + l3c_cdf_table_mixture on the same
    pixels, bit for bit -- the chunk-pipelined RGB decoder and the encoder's fused intervals read the same numbers."""
    from l3c_pytorch_amd import ops
    rng = np.random.RandomState(c)
    B, K, H, W = 2, 10, 12, 20
    spec = odmll.RGB if rgb else odmll.z_spec()
    P = torch.from_numpy(_rand_P(rng, B, H, W, C, K, rgb)).cuda()
    sym = torch.from_numpy(rng.randint(0, spec.L, size=(B, C, H, W)).astype(np.int16)).cuda()
    targets = ocdf.coding_targets(spec.x_min, spec.x_max, spec.L).cuda()
    pi, mu, ls = ops.dmll_channel_params(P, sym if rgb else None, C, K, rgb, c)
    full, flag = ops.cdf_table_mixture(targets, pi, mu, ls)
    full = full.reshape(B, H * W, -1)
    for p0, n in [(0, H * W), (32, 64), (7, 50), (H * W - 33, 33)]:
        f = torch.zeros(1, dtype=torch.int32, device='cuda')
        part = ops.dmll_cdf_table(P, sym if rgb else None, targets, C, K, rgb, c, p0, n, f)
        assert torch.equal(part[..., :-1], full[:, p0:p0 + n, :-1]), (p0, n)
        assert int(f.item()) == int(flag.item())


@pytest.mark.parametrize('x_min,x_max,L', [(0, 255, 256), (-1, 1, 25)])
def test_coding_targets_on_the_device_are_the_oracles_bits(x_min, x_max, L):
    """SURVEY section 8 row a10 (bitcoding/coders_helpers.py:42-44): the bin edges the HIP head evaluates the CDF at are bit for
    bit the oracle's -- for the RGB scale (L = 256) and the bottleneck scales (L = 25) -- on the device, and through the
    reference-API iterator CodingCDFNonshared as well."""
    from oracle import cdf as ocdf
    from l3c_pytorch_amd.bitcoding.coders_helpers import CodingCDFNonshared
    from l3c_pytorch_amd.criterion.logistic_mixture import DiscretizedMixLogisticLoss
    dmll = DiscretizedMixLogisticLoss(rgb_scale=(L == 256), x_min=x_min, x_max=x_max, L=L)
    want = ocdf.coding_targets(x_min, x_max, L)
    got = dmll.coding_targets('cuda')
    assert got.is_cuda and got.dtype == torch.float32 and got.shape == (L + 1,)
    assert torch.equal(got.cpu(), want)
    assert got.cpu().numpy().tobytes() == want.numpy().tobytes()
    l = torch.zeros(1, 4, 2, 2, device='cuda')
    assert torch.equal(CodingCDFNonshared(l, 3 if L == 256 else 5, dmll).targets.cpu(), want)


@pytest.mark.parametrize('x_min,x_max,L', [(0, 255, 256), (-1, 1, 25)])
def test_host_and_device_linspace_give_the_same_bin_edges(x_min, x_max, L):
    """Round-4 advisor: the bin edges are computed on the HOST since round 4 (the oracle's form); rounds 1-3 and a GPU run of the
    reference (coders_helpers.py:42 builds them on l.device) computed them with the DEVICE's linspace kernel.  The `.l3c` container
    has no version field, so files of either kind must decode with either: on this stack the two kernels give the same bits for both
    alphabets -- asserted, so that a future torch / ROCm whose linspace kernels disagree shows up as a failing test and not as files
    that decode to garbage."""
    bin_width = (x_max - x_min) / (L - 1)
    host = torch.linspace(x_min - bin_width / 2, x_max + bin_width / 2, L + 1, dtype=torch.float32)
    dev = torch.linspace(x_min - bin_width / 2, x_max + bin_width / 2, L + 1, dtype=torch.float32, device='cuda')
    assert dev.cpu().numpy().tobytes() == host.numpy().tobytes()


def test_the_kernels_sigmoid_is_the_plain_one_on_every_float():
    """csrc/dmll_core.h: sigmoid_sat (saturated ends taken as constants, the middle without the library's range clamps, division
    scaling and fix-up) must return the bits of 1 / (1 + expf(-a)) for EVERY float -- the tables of rounds 1-3, the fixtures
    and the files written with them stay what they are.  All 2^32 bit patterns, on the device."""
    from l3c_pytorch_amd import _lib, ops
    bad = torch.zeros(1, dtype=torch.int64, device='cuda')
    first = torch.full((1,), -1, dtype=torch.int32, device='cuda')      # 0xFFFFFFFF
    _lib.call_xcheck('l3c_xcheck_sigmoid_exhaustive', ops.ptr(bad), ops.ptr(first), ops.stream())
    torch.cuda.synchronize()
    assert int(bad.item()) == 0, 'first differing input bits: 0x{:08x}'.format(int(first.item()) & 0xFFFFFFFF)
