// tests/hostsim/window_hostsim.hip -- TEST INFRASTRUCTURE: the host side of the window-row rules the RGB decoder and its table kernel
// share (l3c-pytorch_amd/csrc/dmll_core.h: use_window, window_stat, window_miss, window_would_miss are __host__ __device__), exported so
// that the CPU suite can check them exhaustively without a GPU.  Not part of the product path; no kernel is launched.
//
// Build: hipcc --offload-arch=gfx950 -O2 -shared -fPIC -I l3c-pytorch_amd/csrc -o tests/hostsim/_build/libwindow_hostsim.so tests/hostsim/window_hostsim.hip
#include "dmll_core.h"

extern "C" {
int hostsim_win_lp() { return l3c::kWinLp; }
int hostsim_win_top() { return l3c::kWinTop; }
int hostsim_win_max_offset() { return l3c::kWinMaxOffset; }
int hostsim_win_bad() { return l3c::kWinBad; }
int hostsim_use_window(int stat) { return l3c::use_window(stat) ? 1 : 0; }
int hostsim_window_stat(unsigned misses, unsigned n_sym, unsigned one_in) { return l3c::window_stat(misses, n_sym, one_in); }
int hostsim_window_miss(unsigned xw, unsigned w0) { return l3c::window_miss(xw, w0) ? 1 : 0; }
int hostsim_window_would_miss(unsigned x, unsigned w0) { return l3c::window_would_miss(x, w0) ? 1 : 0; }

// Decode ONE count against the window row of `full` (Lp = 257 entries, entry 256 never read: torchac.cpp:181) at offset w0, the way
// every decoder here ranks a row: x' = max(#{j < 64: e[j] <= count}, 1) - 1.  -> the symbol, or -1 for a miss by the product's rule.
int hostsim_window_decode(const unsigned short *full, unsigned w0, unsigned count) {
    unsigned below = 0;
    for (int j = 0; j < 64; ++j) below += full[w0 + j] <= count ? 1u : 0u;
    const unsigned xw = (below ? below : 1u) - 1u;
    if (l3c::window_miss(xw, w0)) return -1;
    return xw == (unsigned)l3c::kWinTop ? 255 : (int)(w0 + xw);      // rank 63 at the top window is the top symbol (w0 + 63 = 255)
}
}
