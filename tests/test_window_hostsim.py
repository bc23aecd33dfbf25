"""CPU tests of the window-row rules (round 5) -- the __host__ __device__ functions of l3c-pytorch_amd/csrc/dmll_core.h that the RGB
decoder (csrc/ac_kernels.hip) and its table kernel (csrc/dmll_kernels.hip) share, run on the host by tests/hostsim/window_hostsim.hip.

What a decoder does with a full row is the reference's binsearch (torchac.cpp:276-296): the largest m in [0, Lp - 2] with
cdf[m] <= count.  A 65-entry window row must give the SAME symbol or report a miss -- never a different symbol; the GPU tests
(tests/test_gpu_window.py) check that on whole streams, these check the rule itself exhaustively."""
import numpy as np
import pytest

from tests.hostsim import window_hostsim

LP = 257


@pytest.fixture(scope='module')
def lib():
    return window_hostsim.get()


def _rows(rng, n):
    """Strictly increasing uint16 rows as the table kernel builds them (torchac.py:174-213: quantised mass + l, so every symbol keeps a
    count of at least one); entry 0 is NOT zero in general, entry 256 is never read (here: garbage)."""
    rows = []
    for i in range(n):
        kind = i % 4
        if kind == 0:       # mass spread evenly
            inc = rng.integers(1, 400, size=LP - 1)
        elif kind == 1:     # a peaked mixture: nearly all the mass in a few bins
            inc = np.ones(LP - 1, np.int64)
            inc[rng.integers(0, LP - 1, size=3)] += rng.integers(5000, 20000, size=3)
        elif kind == 2:     # floor everywhere (every symbol exactly one count) with the mass below the first edge
            inc = np.ones(LP - 1, np.int64)
        else:               # mass at both ends
            inc = np.ones(LP - 1, np.int64)
            inc[0] += 20000
            inc[-1] += 20000
        first = int(rng.integers(0, 3000)) if kind != 2 else 40000
        cdf = first + np.concatenate([[0], np.cumsum(inc)])
        cdf = np.minimum(cdf, 65535 - (LP - 1) + np.arange(LP))            # keep it strictly increasing below 2^16
        row = cdf.astype(np.uint16)
        row[LP - 1] = rng.integers(0, 65536)
        assert (np.diff(row[:LP - 1].astype(np.int64)) > 0).all()
        rows.append(row)
    return rows


def _full_decode(row, count):
    return max(int((row[:LP - 1] <= count).sum()), 1) - 1


def test_constants_match_the_interface(lib):
    assert lib.hostsim_win_lp() == 65 and lib.hostsim_win_top() == 63           # include/l3c_hip.h: "65-entry WINDOW rows"
    assert lib.hostsim_win_max_offset() == 256 - 64
    assert lib.hostsim_win_bad() == 1 << 30


def test_window_decode_is_the_full_row_decode_or_a_miss(lib):
    rng = np.random.default_rng(5)
    checked = misses = 0
    for row in _rows(rng, 24):
        # counts at and around every entry (the decision boundaries), the extremes, and random ones
        edges = row[:LP - 1].astype(np.int64)
        counts = np.unique(np.clip(np.concatenate([edges - 1, edges, edges + 1, [0, 65535], rng.integers(0, 65536, size=64)]), 0, 65535))
        for w0 in (0, 1, 31, 97, 191, 192):
            for count in counts:
                x = _full_decode(row, int(count))
                got = lib.hostsim_window_decode(row.ctypes.data, w0, int(count))
                would_miss = bool(lib.hostsim_window_would_miss(x, w0))
                assert (got == -1) == would_miss, (w0, int(count), x, got)
                if got != -1:
                    assert got == x, (w0, int(count), x, got)
                checked += 1
                misses += got == -1
    assert checked > 50000 and 0 < misses < checked


def test_would_miss_is_exactly_the_symbols_outside_the_window(lib):
    for w0 in range(0, 193):
        for x in range(256):
            inside = (w0 + 1 <= x <= w0 + 62) or (w0 == 0 and x == 0) or (w0 == 192 and x == 255)
            assert bool(lib.hostsim_window_would_miss(x, w0)) == (not inside), (x, w0)
    # rank rule: ranks 1 .. 62 never miss, 0 and 63 only at the row's ends
    for w0 in range(0, 193):
        for xw in range(64):
            expect = (xw == 0 and w0 != 0) or (xw == 63 and w0 != 192)
            assert bool(lib.hostsim_window_miss(xw, w0)) == expect


def test_statistics_word(lib):
    bad = lib.hostsim_win_bad()
    assert not lib.hostsim_use_window(-1)                       # unknown: full rows
    assert lib.hostsim_use_window(0) and lib.hostsim_use_window(bad - 1)
    assert not lib.hostsim_use_window(bad) and not lib.hostsim_use_window(bad | 5)
    # a stream ON window rows stays up to 1/64 misses, one on full rows enters up to 1/128 (hysteresis)
    n = 12288
    assert lib.hostsim_window_stat(n // 64, n, 64) == n // 64
    assert lib.hostsim_window_stat(n // 64 + 1, n, 64) == (n // 64 + 1) | bad
    assert lib.hostsim_window_stat(n // 128, n, 128) == n // 128
    assert lib.hostsim_window_stat(n // 128 + 1, n, 128) == (n // 128 + 1) | bad
    assert lib.hostsim_window_stat(0, 0, 64) == 0               # an empty chunk is not a bad one
    # the count saturates below the flag, the verdict is taken on the true count
    assert lib.hostsim_window_stat(0xFFFFFFFF, 0xFFFFFFFF, 1) == bad - 1
    assert lib.hostsim_window_stat(0xFFFFFFFF, 1024, 64) == (bad - 1) | bad
    # the short probe chunks judge against their own length
    assert lib.hostsim_use_window(lib.hostsim_window_stat(8, 1024, 128))
    assert not lib.hostsim_use_window(lib.hostsim_window_stat(9, 1024, 128))
