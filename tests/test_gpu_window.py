"""-m gpu: WINDOW ROWS of the RGB decoder (round 5; include/l3c_hip.h l3c_ac_decode_part, csrc/dmll_core.h).

The decoder's table rows of a chunk are 65-entry windows around the mixture's mean where the stream's earlier chunks say that pays,
full 257-entry rows otherwise; a symbol outside its window is decoded from the pixel's full row, evaluated by the decoder wavefront
itself.  Whatever the row form, the decoded symbols are those of the classic full-row decoder (reference: torchac.cpp:276-381 on the
table of torchac_kernel.cu:26-76; bitcoding.py:248-266 is the per-channel loop this pipeline replaces):
  * the window rows ARE slices of the full rows (same device functions, same bits);
  * windowed decode == full-row decode == input on the bench images of BOTH checkpoints (the default-init one misses on ~95 % of the
    R and G symbols when forced into window rows), in every mode ('auto', 'always', 'never');
  * a stream engineered to miss on EVERY symbol; tables flagged as not validated (every symbol through the generic pass and a full
    row); garbage streams (decoded alike, whatever the row form); one-chunk and ragged-chunk images.
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import cdf as ocdf  # noqa: E402

K = 10


def _rand_P(rng, B, H, W, mean_lo=-20., mean_hi=280., ls_lo=-2., ls_hi=3.):
    """RGB-scale P (B,H,W,120): logits ~ N(0,1), means uniform, log sigma uniform, lambda ~ N(0,1)"""
    CK = 3 * K
    P = rng.randn(B, H, W, 4 * CK).astype(np.float32)
    P[..., CK:2 * CK] = rng.uniform(mean_lo, mean_hi, size=(B, H, W, CK))
    P[..., 2 * CK:3 * CK] = rng.uniform(ls_lo, ls_hi, size=(B, H, W, CK))
    return P


def _targets():
    return ocdf.coding_targets(0, 255, 256).cuda()


def _encode(P, sym):
    """-> payload bytes per stream (b * 3 + c), through the product's fused interval kernel and range coder"""
    from l3c_pytorch_amd import ops
    B, C, H, W = sym.shape
    iv = ops.dmll_encode_intervals(P, sym, _targets(), C, K, True)
    out, n = ops.ac_encode(iv, B * C, H * W)
    n, out = n.cpu().numpy(), out.cpu().numpy()
    return [out[i, :n[i]].tobytes() for i in range(B * C)]


def _bounds(HW, chunks):
    step = -(-HW // chunks)
    step = -(-step // 64) * 64                  # chunk boundaries on the 64-symbol store blocks
    return [(p0, min(step, HW - p0)) for p0 in range(0, HW, step)]


def _decode_channel(P, sym_true, payloads, c, stats_in, flag_value=0, chunks=1):
    """One channel of every image through l3c_dmll_cdf_table + l3c_ac_decode_chunks, `chunks` chunks, the previous channels taken
    from `sym_true` (so that the channels can be tested one by one).  stats_in: None (classic rows) or an int32 (B,) tensor that is
    used for EVERY chunk.  -> (decoded (B, HW) int16, stats_out of the last chunk or None)"""
    from l3c_pytorch_amd import ops
    B, C, H, W = sym_true.shape
    HW = H * W
    targets = _targets()
    buf, offs, lens = ops.pack_streams(payloads[c::C])
    out = torch.full((B, C, H, W), -7, dtype=torch.int16, device='cuda')
    flag = torch.full((1,), flag_value, dtype=torch.int32, device='cuda')
    states = [ops.ac_decode_state(B), ops.ac_decode_state(B)]
    bounds = _bounds(HW, chunks)
    stats_out = None
    for j, (p0, n) in enumerate(bounds):
        win = None
        if stats_in is not None:
            stats_out = torch.full((B,), -5, dtype=torch.int32, device='cuda')
            win = (stats_in, stats_out, P, sym_true, targets, p0, C, K, c)
        table = ops.dmll_cdf_table(P, sym_true, targets, C, K, True, c, p0, n, flag, window_stats=stats_in)
        part = ops.ac_decode_part(table.reshape(B * n, -1), buf, offs, lens, B, n, flag, states[(j + 1) & 1] if j else None,
                                  states[j & 1], j == len(bounds) - 1, out, C * HW, c * HW + p0, window=win)
        ops.ac_decode_chunks([part])
    torch.cuda.synchronize()
    return out[:, c].reshape(B, HW), stats_out


@pytest.mark.parametrize('c', [0, 1, 2])
def test_window_rows_are_slices_of_the_full_rows(c):
    from l3c_pytorch_amd import ops
    rng = np.random.RandomState(10 + c)
    B, H, W = 3, 12, 20
    HW = H * W
    P = torch.from_numpy(_rand_P(rng, B, H, W)).cuda()
    sym = torch.from_numpy(rng.randint(0, 256, size=(B, 3, H, W)).astype(np.int16)).cuda()
    t = _targets()
    full = ops.dmll_cdf_table(P, sym, t, 3, K, True, c, 0, HW).cpu().numpy().view(np.uint16).astype(np.int64)      # (B, HW, 257)
    # image 0: window rows; image 1: unknown statistics -> full rows; image 2: too many misses -> full rows
    stats = torch.tensor([0, -1, HW | 0x40000000], dtype=torch.int32, device='cuda')     # (bit 30: more than 1/64 of the chunk missed)
    flag = torch.zeros(1, dtype=torch.int32, device='cuda')
    mixed = ops.dmll_cdf_table(P, sym, t, 3, K, True, c, 0, HW, flag, window_stats=stats).cpu().numpy().view(np.uint16).astype(np.int64)
    assert int(flag.item()) == 0
    win = mixed[0].reshape(-1)[:HW * 65].reshape(HW, 65)
    w0 = win[:, 64]                            # entry 64: the window's offset; entries 0 .. 63 = cdf[w0 .. w0 + 63]
    assert w0.min() >= 0 and w0.max() <= 192 and len(np.unique(w0)) > 5
    for j in range(64):
        assert np.array_equal(win[:, j], full[0][np.arange(HW), w0 + j]), j
    assert (full[0][:, 0] > 0).any()           # cdf[0] is the mass below the first bin edge, not 0: a window at offset 0 must carry it
    for b in (1, 2):       # full rows whose dead last entry carries the offset a window would have had
        assert np.array_equal(mixed[b][:, :256], full[b][:, :256])
        assert mixed[b][:, 256].min() >= 0 and mixed[b][:, 256].max() <= 192
    # the offsets follow the mixture's mean: window centre within one symbol of the mean wherever the clamp is not active
    pi, mu, _ = ops.dmll_channel_params(P, sym, 3, K, True, c)
    mean = (pi * mu).sum(1).reshape(B, HW).cpu().numpy()
    want = np.clip(np.floor(np.clip(mean[0], 0, 255)) - 31, 0, 192)
    assert np.abs(w0 - want).max() <= 1          # (the kernel sums pi_k mu_k sequentially in fp32)
    assert np.abs(mixed[1][:, 256] - np.clip(np.floor(np.clip(mean[1], 0, 255)) - 31, 0, 192)).max() <= 1


@pytest.mark.parametrize('H,W,chunks', [(8, 24, 1), (40, 56, 3), (64, 96, 2)])
def test_every_symbol_a_miss_and_no_symbol_a_miss(H, W, chunks):
    """Means near 40 with small sigmas: symbols drawn near 40 never miss, symbols forced to 200..255 miss EVERY time (the decoder
    evaluates each pixel's full row itself) -- both decode to the input, and the reported miss counts say so."""
    rng = np.random.RandomState(H + W)
    B = 2
    HW = H * W
    P = torch.from_numpy(_rand_P(rng, B, H, W, mean_lo=38., mean_hi=42., ls_lo=0., ls_hi=1.5)).cuda()
    P[..., 3 * 3 * K:] = -30.0          # lambda = sigmoid(-30) ~ 0: no coupling, the means stay near 40 for G and B too
    near = torch.from_numpy(rng.randint(30, 52, size=(B, 3, H, W)).astype(np.int16)).cuda()
    far = torch.from_numpy(rng.randint(200, 256, size=(B, 3, H, W)).astype(np.int16)).cuda()
    far[0, :, 0, 0] = 255
    zeros = torch.zeros(B, dtype=torch.int32, device='cuda')
    for sym, expect_all_miss in ((near, False), (far, True)):
        payloads = _encode(P, sym)
        for c in range(3):
            classic, _ = _decode_channel(P, sym, payloads, c, None, chunks=chunks)
            assert torch.equal(classic, sym[:, c].reshape(B, HW))
            got, stats = _decode_channel(P, sym, payloads, c, zeros, chunks=chunks)
            assert torch.equal(got, sym[:, c].reshape(B, HW)), (c, expect_all_miss)
            last = _bounds(HW, chunks)[-1][1]
            assert stats.tolist() == [(last | 0x40000000) if expect_all_miss else 0] * B, (stats.tolist(), last)   # bit 30: too many for a window


def test_full_rows_count_what_a_window_would_have_missed():
    """statistics -1 (unknown): full rows, and the decoder reports how many symbols a window would have missed -- an estimate (every
    fourth ring block is looked at, x 4) of the number the window decoder counts for the same stream."""
    rng = np.random.RandomState(3)
    B, H, W = 2, 32, 48
    HW = H * W
    P = torch.from_numpy(_rand_P(rng, B, H, W, mean_lo=60., mean_hi=200., ls_lo=1.5, ls_hi=3.5)).cuda()
    sym = torch.from_numpy(np.clip(rng.normal(130, 45, size=(B, 3, H, W)), 0, 255).astype(np.int16)).cuda()
    payloads = _encode(P, sym)
    for c in range(3):
        got_w, stats_w = _decode_channel(P, sym, payloads, c, torch.zeros(B, dtype=torch.int32, device='cuda'))
        got_f, stats_f = _decode_channel(P, sym, payloads, c, torch.full((B,), -1, dtype=torch.int32, device='cuda'))
        assert torch.equal(got_w, sym[:, c].reshape(B, HW)) and torch.equal(got_f, got_w)
        w, f = [v & 0x3FFFFFFF for v in stats_w.tolist()], [v & 0x3FFFFFFF for v in stats_f.tolist()]
        assert 0 < min(w) and max(w) < HW
        for a, e in zip(w, f):
            assert a / 2.5 <= e <= a * 2.5, (w, f)


def test_unvalidated_table_and_garbage_streams_decode_alike_in_both_row_forms():
    """flag != 0 (table not validated): every symbol goes through the generic pass -- with window rows that means a full row per symbol and
    the reference's literal binary search -- and still decodes the stream; random bytes decode to the SAME symbols whatever the row
    form (torchac.cpp's wrapping arithmetic on a foreign stream)."""
    rng = np.random.RandomState(4)
    B, H, W = 2, 16, 40
    HW = H * W
    P = torch.from_numpy(_rand_P(rng, B, H, W, mean_lo=90., mean_hi=160., ls_lo=0.5, ls_hi=3.)).cuda()
    sym = torch.from_numpy(np.clip(rng.normal(125, 30, size=(B, 3, H, W)), 0, 255).astype(np.int16)).cuda()
    payloads = _encode(P, sym)
    zeros = torch.zeros(B, dtype=torch.int32, device='cuda')
    for c in range(3):
        got, stats = _decode_channel(P, sym, payloads, c, zeros, flag_value=1)
        assert torch.equal(got, sym[:, c].reshape(B, HW)), c
        assert stats.tolist() == [0x7FFFFFFF] * B         # the generic pass reports no usable statistics
    garbage = [rng.randint(0, 256, size=600).astype(np.uint8).tobytes() for _ in range(B * 3)]
    for c in range(3):
        a, _ = _decode_channel(P, sym, garbage, c, None)
        b, _ = _decode_channel(P, sym, garbage, c, zeros)
        assert torch.equal(a, b), c


@pytest.fixture(scope='module', params=['default', 'calibrated'])
def blueprint(request, l3c_checkpoint):
    from l3c_pytorch_amd.blueprints.multiscale_blueprint import MultiscaleBlueprint
    cfg, sd = l3c_checkpoint(request.param == 'calibrated')
    bp = MultiscaleBlueprint(cfg)
    bp.net.load_state_dict(sd, strict=True)
    bp.set_eval()
    bp.ckpt_name = request.param
    return bp


def test_windowed_decode_of_the_bench_images_in_every_mode(blueprint):
    """8 bench images (768x512) of both checkpoints: files decoded with window rows where the streams' statistics allow ('auto'), with
    window rows everywhere ('always': the default-init checkpoint then misses on most R and G symbols) and with full rows only ('never')
    all give back the input; on the calibrated checkpoint 'auto' really runs on window rows from the third chunk on."""
    from l3c_pytorch_amd.bitcoding.bitcoding import Bitcoding
    from l3c_pytorch_amd.helpers import synthetic
    H, W = 512, 768
    imgs = torch.stack([synthetic.make_image(H, W, i, 'natural') for i in range(8)])
    files = Bitcoding(blueprint).encode_batch(imgs).to_bytes()
    for mode in ('auto', 'never', 'always'):
        bc = Bitcoding(blueprint, rgb_window=mode)
        dec, _ = bc.decode_batch(files)
        assert torch.equal(dec.cpu(), imgs.long()), (blueprint.ckpt_name, mode)
        if mode == 'auto':
            stats = bc.last_rgb_window_stats.cpu().numpy()             # (3, chunks + 2, 8): slot j + 2 = chunk j (two probes first)
            assert (stats[:, :2] == -1).all() and (stats[:, 2:] >= 0).all()
            good = (stats[:, 2:] & 0x40000000) == 0                       # at most 1/64 of the chunk's symbols missed: window rows two chunks on
            print(blueprint.ckpt_name, 'share of (chunk, image) pairs that qualify for window rows, R G B:', good.mean(axis=(1, 2)),
                  'misses in the regular chunks, mean R G B:', (stats[:, 4:] & 0x3FFFFFFF).mean(axis=(1, 2)))
            if blueprint.ckpt_name == 'calibrated':
                assert good.mean() > 0.7
            else:
                assert good[:2].mean() < 0.5                              # mixtures in the wrong place: R and G stay on full rows
    # one image alone (32 chunks, no overlapped schedule) and its batch-invariance
    one, _ = Bitcoding(blueprint).decode_batch(files[3:4])
    assert torch.equal(one.cpu(), imgs[3:4].long())


def test_windowed_decode_under_the_overlapped_schedule(blueprint):
    """16 images (256x384): the schedule with tables and decoders two chunks apart on two streams -- the statistics a chunk's table kernel
    reads are those of the chunk before the previous one, complete by then."""
    from l3c_pytorch_amd.bitcoding.bitcoding import Bitcoding
    from l3c_pytorch_amd.helpers import synthetic
    imgs = torch.stack([synthetic.make_image(256, 384, 40 + i, 'natural') for i in range(16)])
    files = Bitcoding(blueprint).encode_batch(imgs).to_bytes()
    for mode in ('auto', 'always', 'never'):
        dec, _ = Bitcoding(blueprint, rgb_window=mode).decode_batch(files)
        assert torch.equal(dec.cpu(), imgs.long()), (blueprint.ckpt_name, mode)
