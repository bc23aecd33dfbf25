"""-m gpu: the HIP range coder (csrc/ac_kernels.hip) against the reference KATs and the oracle, through the C ABI.
Integer work: everything here is bit-exact."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import ac as oracle_ac  # noqa: E402


def _names(g):
    return sorted({k.split('/')[0] for k in g.files if k.endswith('/sym')})


def test_encode_decode_reference_kats(golden):
    from tests import gpu_util as gu
    g = golden('ac_kat.npz')
    for name in _names(g):
        tab, sym, ref = g[name + '/cdf'], g[name + '/sym'], g[name + '/bytes'].tobytes()
        got = gu.hip_encode_streams(tab[None], sym[None])[0]
        assert got == ref, (name, len(got), len(ref))
        for monotone in (True, False):
            dec = gu.hip_decode_streams(tab[None], [ref], monotone)[0]
            assert (dec == sym).all(), (name, monotone, int((dec != sym).sum()))


def test_decode_truncated_stream_like_reference(golden):
    from tests import gpu_util as gu
    g = golden('ac_kat.npz')
    dec = gu.hip_decode_streams(g['truncated/cdf'][None], [g['truncated/bytes'].tobytes()], False)[0]
    assert (dec == g['truncated/decoded']).all()


@pytest.mark.parametrize('S,N,Lp', [(1, 1, 257), (3, 65, 26), (70, 1000, 257), (130, 517, 26), (64, 64, 3), (5, 4097, 257),
                                    (9, 700, 130), (4, 900, 65), (3, 1300, 200), (2, 50, 2), (50, 600, 256)])
def test_many_streams_vs_oracle(S, N, Lp):
    """streams per lane pair (encode) / per wavefront (decode); S crosses the wavefront boundaries, N the 64-symbol blocks; alphabets
    of one row register (Lp <= 65: every lane a table entry at 65), of four with spare lanes (130, 200, 256: the compiled symbol of
    the lean decoder) and the full 257 (its hand-written loop); 50 streams: the small ring."""
    from tests import gpu_util as gu
    rng = np.random.RandomState(S * 1000 + N)
    tabs = gu.random_tables(rng, S, N, Lp, shape=rng.choice([0.05, 0.3, 2.0]))
    syms = gu.sample_symbols(rng, tabs)
    syms[::2] = rng.randint(0, Lp - 1, size=syms[::2].shape)       # unlikely symbols too
    got = gu.hip_encode_streams(tabs, syms)
    for s in range(S):
        assert got[s] == oracle_ac.encode(tabs[s], syms[s]), s
    for monotone in (True, False):
        dec = gu.hip_decode_streams(tabs, got, monotone)
        assert (dec == syms).all(), monotone


@pytest.mark.parametrize('n_groups', [1, 4, 70])
def test_grouped_launch_vs_oracle(n_groups):
    """l3c_ac_encode_groups: groups of different stream counts / lengths / alphabets in ONE launch pair; n_groups=70
    crosses the 64-descriptor upload packs.  Every stream must equal the oracle's bytes."""
    from l3c_pytorch_amd import ops
    from tests import gpu_util as gu
    rng = np.random.RandomState(n_groups)
    shapes = [(int(rng.choice([1, 3, 64, 65, 130])), int(rng.choice([1, 63, 64, 200, 1025])), int(rng.choice([3, 26, 257])))
              for _ in range(n_groups)]
    groups, want = [], []
    for S, N, Lp in shapes:
        tabs = gu.random_tables(rng, S, N, Lp, shape=rng.choice([0.05, 0.3, 2.0]))
        syms = gu.sample_symbols(rng, tabs)
        t = torch.from_numpy(np.ascontiguousarray(tabs).view(np.int16)).cuda().reshape(S * N, Lp)
        groups.append((ops.intervals_from_table(t, torch.from_numpy(syms).cuda(), S, N), S, N))
        want.append([oracle_ac.encode(tabs[s], syms[s]) for s in range(S)])
    res, _ = ops.ac_encode_groups(groups)
    for g, ((out, n), w) in enumerate(zip(res, want)):
        n, out = n.cpu().numpy(), out.cpu().numpy()
        for s in range(len(w)):
            assert out[s, :n[s]].tobytes() == w[s], (g, s, shapes[g])


@pytest.mark.parametrize('Lp', [26, 257])
def test_full_range_recurs_on_width_one_steps(Lp):
    """cdf steps of exactly 1 (or an aligned power of two) coded from the full 32-bit range renormalise back to the full
    range: low = 0, high = 2^32 - 1 recurs symbol after symbol, the case in which a masked 0x10000 entry scales to 2^32."""
    from tests import gpu_util as gu
    rng = np.random.RandomState(Lp)
    S, N = 4, 300
    tab = np.empty((S, N, Lp), dtype=np.uint16)
    tab[0] = np.arange(Lp)                                           # width-1 steps, all mass on the top symbol
    tab[1] = np.minimum(np.arange(Lp) * 2, 65535)                    # width 2
    tab[2] = np.concatenate([[0], 40000 + np.arange(Lp - 1)])        # a wide symbol 0, then width-1 steps
    tab[3] = (np.arange(Lp) * (65536 // Lp)).astype(np.uint16)       # near-uniform, for contrast
    syms = rng.randint(0, Lp - 2, size=(S, N)).astype(np.int16)      # never the heavy top symbol
    syms[2, ::3] = 0
    got = gu.hip_encode_streams(tab, syms)
    for s in range(S):
        assert got[s] == oracle_ac.encode(tab[s], syms[s]), s
    for monotone in (True, False):
        dec = gu.hip_decode_streams(tab, got, monotone)
        assert (dec == syms).all(), (monotone, [int((dec[s] != syms[s]).sum()) for s in range(S)])


@pytest.mark.parametrize('Lp,N', [(257, 700), (26, 3000)])
def test_foreign_streams_decode_like_the_reference(Lp, N):
    """Byte strings no encoder produced (random bytes, a stream coded with OTHER tables, an empty one): `value` leaves
    [low, high], the fast pass marks the stream and the generic pass must reproduce the reference's wrapping arithmetic
    (torchac.cpp:329) symbol for symbol -- alongside valid streams in the same launch, across several ring blocks."""
    from tests import gpu_util as gu
    rng = np.random.RandomState(Lp)
    S = 6
    tabs = gu.random_tables(rng, S, N, Lp, shape=0.3)
    syms = gu.sample_symbols(rng, tabs)
    other = gu.random_tables(rng, S, N, Lp, shape=2.0)
    payloads = [oracle_ac.encode(tabs[s], syms[s]) for s in range(S)]
    payloads[1] = rng.randint(0, 256, size=len(payloads[1]), dtype=np.uint8).tobytes()
    payloads[2] = oracle_ac.encode(other[2], gu.sample_symbols(rng, other)[2])
    payloads[3] = b''
    payloads[4] = payloads[4][:len(payloads[4]) // 3]
    want = np.stack([oracle_ac.decode(tabs[s], payloads[s], N) for s in range(S)])
    assert (want[0] == syms[0]).all() and (want[5] == syms[5]).all()
    for monotone in (True, False):
        dec = gu.hip_decode_streams(tabs, payloads, monotone)
        for s in range(S):
            assert (dec[s] == want[s]).all(), (monotone, s, int((dec[s] != want[s]).sum()))


@pytest.mark.parametrize('Lp,N,cuts', [(257, 1000, (0, 64, 576, 1000)), (26, 5000, (0, 2048, 2112, 4992, 5000)), (257, 130, (0, 130))])
def test_chunked_decode_resumes_exactly(Lp, N, cuts):
    """l3c_ac_decode_chunks: a stream decoded in pieces (state carried through the opaque per-stream record, table rows of
    each piece in their own buffer, validity as a DEVICE flag) gives the symbols of the one-shot decode; a foreign stream in
    the batch takes the generic pass chunk by chunk; a set flag sends everything through the generic pass."""
    from l3c_pytorch_amd import ops
    from tests import gpu_util as gu
    rng = np.random.RandomState(N)
    S = 5
    tabs = gu.random_tables(rng, S, N, Lp, shape=0.3)
    syms = gu.sample_symbols(rng, tabs)
    payloads = [oracle_ac.encode(tabs[s], syms[s]) for s in range(S)]
    payloads[2] = rng.randint(0, 256, size=len(payloads[2]), dtype=np.uint8).tobytes()
    want = np.stack([oracle_ac.decode(tabs[s], payloads[s], N) for s in range(S)])
    buf, offs, lens = ops.pack_streams(payloads)
    for flag_value in (0, 1):
        out = torch.full((S, N + 7), -7, dtype=torch.int16, device='cuda')      # row stride != chunk length
        flag = torch.full((1,), flag_value, dtype=torch.int32, device='cuda')
        states = [ops.ac_decode_state(S), ops.ac_decode_state(S)]
        for j in range(len(cuts) - 1):
            p0, n = cuts[j], cuts[j + 1] - cuts[j]
            chunk = torch.from_numpy(np.ascontiguousarray(tabs[:, p0:p0 + n]).view(np.int16)).cuda().reshape(S * n, Lp)
            ops.ac_decode_chunks([ops.ac_decode_part(chunk, buf, offs, lens, S, n, flag, states[(j + 1) & 1] if j else None,
                                                     states[j & 1], j == len(cuts) - 2, out, N + 7, p0)])
        got = out.cpu().numpy()
        assert (got[:, :N] == want).all(), (flag_value, [int((got[s, :N] != want[s]).sum()) for s in range(S)])
        assert (got[:, N:] == -7).all()


def test_uniform_row_broadcast():
    from l3c_pytorch_amd import ops
    from l3c_pytorch_amd.bitcoding.bitcoding import uniform_cdf_row
    rng = np.random.RandomState(3)
    row = uniform_cdf_row(25)
    S, N = 10, 6144
    sym = rng.randint(0, 25, size=(S, N)).astype(np.int16)
    iv = ops.intervals_from_table(row.cuda(), torch.from_numpy(sym).cuda(), S, N, broadcast_row=True)
    out, n = ops.ac_encode(iv, S, N)
    n, out = n.cpu().numpy(), out.cpu().numpy()
    payloads = [out[i, :n[i]].tobytes() for i in range(S)]
    for s in range(S):
        assert payloads[s] == oracle_ac.encode(row.numpy(), sym[s]), s
    buf, offs, lens = ops.pack_streams(payloads)
    dec = ops.ac_decode(row.cuda(), buf, offs, lens, S, N, True, broadcast_row=True).cpu().numpy()
    assert (dec == sym).all()


def test_monotone_check_kernel():
    from l3c_pytorch_amd import ops
    from tests import gpu_util as gu
    rng = np.random.RandomState(5)
    tab = gu.random_tables(rng, 1, 200, 257)[0]
    t = torch.from_numpy(tab.view(np.int16)).cuda()
    assert ops.table_is_monotone(t)
    tab[100, -1] = 0                       # the wrapped last entry is outside [0, Lp-2]: still fine
    assert ops.table_is_monotone(torch.from_numpy(tab.view(np.int16)).cuda())
    tab[150, 7] = tab[150, 6]
    assert not ops.table_is_monotone(torch.from_numpy(tab.view(np.int16)).cuda())


def test_torchac_facade_roundtrip_and_errors(golden):
    """reference API (torchac.py:87-166) with host tensors, as coders.py calls it."""
    from l3c_pytorch_amd import torchac
    g = golden('ac_kat.npz')
    tab, sym, ref = g['random_Lp26/cdf'], g['random_Lp26/sym'], g['random_Lp26/bytes'].tobytes()
    cdf = torch.from_numpy(tab.view(np.int16).copy()).reshape(1, 1, -1, 26)
    assert torchac.encode_cdf(cdf, torch.from_numpy(sym)) == ref
    dec = torchac.decode_cdf(cdf, ref)
    assert dec.dtype == torch.int16 and not dec.is_cuda and (dec.numpy() == sym).all()
    with pytest.raises(RuntimeError):
        torchac.encode_cdf(cdf.reshape(-1, 26), torch.from_numpy(sym))
    with pytest.raises(ValueError):
        torchac.encode_logistic_mixture(torch.zeros(26).cuda(), torch.zeros(1, 10, 4, 4), torch.zeros(1, 10, 4, 4),
                                        torch.zeros(1, 10, 4, 4), torch.zeros(16, dtype=torch.int16))
    assert torchac.CUDA_SUPPORTED and not torchac.CPU_SUPPORTED


def test_torchac_facade_refuses_tables_and_symbols_that_would_overrun():
    """encode_cdf on a user table: symbols outside [0, Lp-2] and empty intervals at a coded symbol are refused (the reference
    reads past the row / emits an undecodable stream; the HIP coder's output rows are sized for <= 16 bits per symbol)."""
    from l3c_pytorch_amd import torchac
    tab = torch.arange(0, 26 * 100, 100, dtype=torch.int32).to(torch.int16).reshape(1, 1, 1, 26).repeat(1, 2, 3, 1)
    sym = torch.tensor([0, 3, 24, 7, 1, 2], dtype=torch.int16)
    data = torchac.encode_cdf(tab, sym)
    assert torch.equal(torchac.decode_cdf(tab, data), sym)
    with pytest.raises(ValueError, match='symbols must lie'):
        torchac.encode_cdf(tab, torch.tensor([0, 3, 25, 7, 1, 2], dtype=torch.int16))
    with pytest.raises(ValueError, match='symbols must lie'):
        torchac.encode_cdf(tab, torch.tensor([0, -1, 2, 7, 1, 2], dtype=torch.int16))
    bad = tab.clone()
    bad[0, 0, 1, 4] = bad[0, 0, 1, 3]                  # row 1: entries 3 and 4 equal -> symbol 3 has an empty interval
    assert torchac.encode_cdf(bad, torch.tensor([0, 2, 24, 7, 1, 2], dtype=torch.int16))       # not coded there: fine
    with pytest.raises(ValueError, match='not increasing'):
        torchac.encode_cdf(bad, sym)                   # symbol 3 at row 1
