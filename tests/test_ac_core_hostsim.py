"""The product's coder state machine (csrc/ac_core.h -- the exact code the HIP kernels instantiate) executed on the host
CPU and checked bit-for-bit against the reference KATs and the oracle.  Covers the integer logic without a GPU."""
import numpy as np

from oracle import ac
from tests.hostsim import ac_hostsim as hs


def _names(g):
    return sorted({k.split('/')[0] for k in g.files if k.endswith('/sym')})


def test_hostsim_kats(golden):
    g = golden('ac_kat.npz')
    for n in _names(g):
        tab, sym, ref = g[n + '/cdf'], g[n + '/sym'], g[n + '/bytes'].tobytes()
        assert hs.encode(tab, sym, fast=True) == ref, n
        assert hs.encode(tab, sym, fast=2) == ref, n
        assert hs.encode(tab, sym, fast=False) == ref, n
        assert (hs.decode(tab, ref, len(sym), True) == sym).all(), n
        assert (hs.decode(tab, ref, len(sym), 2) == sym).all(), n
        assert (hs.decode(tab, ref, len(sym), 3) == sym).all(), (n, 'lean')
        assert (hs.decode(tab, ref, len(sym), False) == sym).all(), n


def test_hostsim_truncated_stream_matches_reference(golden):
    g = golden('ac_kat.npz')
    d = hs.decode(g['truncated/cdf'], g['truncated/bytes'].tobytes(), len(g['truncated/decoded']), False)
    assert (d == g['truncated/decoded']).all()


def _random_case(rng, it):
    Lp = int(rng.choice([2, 3, 5, 26, 257]))
    N = int(rng.randint(1, 2500))
    w = rng.gamma(rng.choice([0.05, 0.3, 2.0]), size=(N, Lp - 1)) + 1e-5
    c = np.cumsum(w, 1)
    c /= c[:, -1:]
    first = rng.randint(0, 500, size=(N, 1))
    tab = np.concatenate([first, np.round(c * (65535 - (Lp - 1) - first)) + np.arange(1, Lp) + first], 1)
    tab = tab.astype(np.int64).astype(np.uint16)
    if it % 2:
        sym = rng.randint(0, Lp - 1, size=N).astype(np.int16)
    else:
        hi = np.concatenate([tab[:, 1:Lp - 1].astype(np.int64), np.full((N, 1), 65536)], 1)
        sym = np.clip((rng.randint(0, 65536, size=(N, 1)) >= hi).sum(1), 0, Lp - 2).astype(np.int16)
    return tab, sym


_last_junk = [b'']


def junk_probe(rng):
    _last_junk[0] = rng.randint(0, 256, size=rng.randint(0, 200)).astype(np.uint8).tobytes()
    return _last_junk[0]


def test_hostsim_random_vs_oracle():
    rng = np.random.RandomState(0)
    for it in range(120):
        tab, sym = _random_case(rng, it)
        ref = ac.encode(tab, sym)
        assert hs.encode(tab, sym, fast=True) == ref, it
        assert hs.encode(tab, sym, fast=2) == ref, (it, 'lane pair')
        assert hs.encode(tab, sym, fast=False) == ref, it
        assert (hs.decode(tab, ref, len(sym), True) == sym).all(), it
        assert (hs.decode(tab, ref, len(sym), 2) == sym).all(), it
        assert (hs.decode(tab, ref, len(sym), 3) == sym).all(), (it, 'lean')
        assert (hs.decode(tab, junk_probe(rng), len(sym), 2) == ac.decode(tab, _last_junk[0])).all(), ('junk2', it)
        assert (hs.decode(tab, ref, len(sym), False) == sym).all(), it
        junk = rng.randint(0, 256, size=rng.randint(0, 200)).astype(np.uint8).tobytes()
        assert (hs.decode(tab, junk, len(sym), False) == ac.decode(tab, junk)).all(), ('junk', it)


def test_hostsim_degenerate_intervals():
    """Width-1 intervals force n == 32 (low == high) and long pending runs."""
    N = 300
    tab = np.zeros((N, 4), dtype=np.uint16)
    tab[:, 1] = 1
    tab[:, 2] = 2
    tab[:, 3] = 0
    for s in (0, 1, 2):
        sym = np.full(N, s, dtype=np.int16)
        ref = ac.encode(tab, sym)
        assert hs.encode(tab, sym) == ref and hs.encode(tab, sym, fast=2) == ref
        assert (hs.decode(tab, ref, N, True) == sym).all() and (hs.decode(tab, ref, N, 3) == sym).all()
    tab[:, 1] = 32768
    tab[:, 2] = 32769
    sym = np.tile(np.array([0, 2, 1, 1, 2, 0], dtype=np.int16), N // 6)
    ref = ac.encode(tab, sym)
    assert hs.encode(tab, sym) == ref and hs.encode(tab, sym, fast=False) == ref and hs.encode(tab, sym, fast=2) == ref
    assert (hs.decode(tab, ref, N, False) == sym).all()


def test_hostsim_dyadic_table_meets_the_full_range_again():
    """A uniform 256-symbol table (entries = multiples of 256): every interval is an aligned power of two, so the coder is back at
    the full 32-bit range (span = 2^32) after every symbol -- the case the lane-pair encoder carries as range = 2^32 - 1 and the
    lean decoder as its own instantiation."""
    rng = np.random.RandomState(3)
    N = 4000
    tab = np.tile((np.arange(257) * 256).astype(np.int64).astype(np.uint16), (N, 1))
    sym = rng.randint(0, 256, size=N).astype(np.int16)
    ref = ac.encode(tab, sym)
    assert hs.encode(tab, sym, fast=2) == ref and hs.encode(tab, sym, fast=True) == ref
    assert (hs.decode(tab, ref, N, 3) == sym).all() and (hs.decode(tab, ref, N, 2) == sym).all()


def test_hostsim_single_steps_of_the_three_forms_agree():
    """interval_update + renorm_counts (the reference's state machine in closed form), the encoder's lane-pair form and the decoder's
    (low, ~high, range) form on five million random steps from every kind of resting state: full range, just outside the underflow
    position, width-1 intervals, the top symbol -- the same bounds, the same n + m, the same record, the same underflow flip."""
    for seed in range(5):
        assert hs.step_forms_disagreements(1000000, seed) == 0, seed
