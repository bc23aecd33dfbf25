// tests/hostsim/ac_hostsim.cpp -- TEST INFRASTRUCTURE: runs the product's coder state machine
// (l3c-pytorch_amd/csrc/ac_core.h, the code the HIP kernels instantiate per lane / per wavefront) on the host CPU so
// that its integer logic can be checked against the reference KATs without a GPU.  Not part of the product path.
//
// Build: g++ -O2 -shared -fPIC -I l3c-pytorch_amd/csrc -o tests/hostsim/_build/libac_hostsim.so tests/hostsim/ac_hostsim.cpp
#include <cstring>
#include <vector>

#include "ac_core.h"

namespace {
struct VecStore {
    uint32_t *words;
    void operator()(uint32_t i, uint32_t w) const { words[i] = w; }
};
struct MemFetch {
    const uint8_t *p;
    uint32_t nbytes;
    uint32_t operator()(uint32_t i) const {
        uint32_t w = 0;
        for (int k = 0; k < 4; ++k) {
            const uint64_t off = (uint64_t)i * 4 + k;
            w = (w << 8) | (off < nbytes ? p[off] : 0u);
        }
        return w;
    }
};
}  // namespace

extern "C" {

// words of the interval stream in the layout of include/l3c_hip.h for ONE stream (n_streams == 1)
long long hostsim_encode(const uint16_t *cdf, long long row_stride, int Lp, const int16_t *sym, long long N,
                         uint8_t *out, long long cap, int fast) {
    std::vector<uint32_t> words((size_t)(N / 2 + 16));
    l3c::WordSink<VecStore> sink(VecStore{words.data()});
    uint32_t low = 0, high = 0xFFFFFFFFu, pending = 0;
    uint32_t u[2] = {0u, 0u}, range = 0xFFFFFFFFu;   // fast == 2: the lane pair of ac_state_kernel (role 0: low, role 1: ~high)
    for (long long i = 0; i < N; ++i) {
        const uint16_t *row = cdf + i * row_stride;
        const int x = sym[i];
        const uint32_t c_lo = row[x];
        const uint32_t c_hi = x == Lp - 2 ? 0x10000u : row[x + 1];
        const uint32_t w = l3c::pack_interval(c_lo, c_hi);
        if (fast == 2) {   // phase 1 as the two roles of a lane pair run it, phase 2's record from the pair of words it leaves
            uint32_t u1[2];
            for (int role = 0; role < 2; ++role)
                u1[role] = u[role] + l3c::role_term(range, l3c::role_word(c_lo, c_hi, role), l3c::role_round(role));
            const int t0 = l3c::role_shift(u1[0], u1[1]), t1 = l3c::role_shift(u1[1], u1[0]);
            u[0] = l3c::role_renorm(u1[0], t0);
            u[1] = l3c::role_renorm(u1[1], t1);
            range = l3c::role_range(u[0], u[1]);
            low = u[0];
            const uint32_t r = l3c::record_from_pair(u1[0], u1[1]);
            const uint32_t n = l3c::record_n(r);
            l3c::emit_record(n ? l3c::record_top(r) << ((32u - n) & 31u) : 0u, n | (l3c::record_m(r) << 8), pending, sink);
        } else if (fast) {   // the two-phase encoder's record path (what ac_state_kernel + ac_pack_kernel implement)
            uint32_t rl, rn;
            l3c::encode_state_step(low, high, w, rl, rn);
            const uint32_t r = l3c::pack_record(rl, rn);
            const uint32_t n = l3c::record_n(r);
            l3c::emit_record(n ? l3c::record_top(r) << ((32u - n) & 31u) : 0u, n | (l3c::record_m(r) << 8), pending, sink);
        } else {
            l3c::encode_symbol(low, high, pending, l3c::interval_lo(w), l3c::interval_hi(w), sink);
        }
    }
    l3c::encode_finish(low, pending, sink);
    const uint32_t n = sink.finish();
    if ((long long)n <= cap) std::memcpy(out, words.data(), n);
    return n;
}

void hostsim_decode(const uint16_t *cdf, long long row_stride, int Lp, const uint8_t *in, long long in_len,
                    int monotone, int16_t *sym_out, long long N) {
    l3c::WordSource<MemFetch> src(MemFetch{in, (uint32_t)in_len});
    uint32_t low = 0, high = 0xFFFFFFFFu;
    uint32_t value = src.take(32);
    const uint32_t top = (uint32_t)(Lp - 2);
    if (monotone == 3) {   // the lean fast decoder's state machine (ac_decode_lean_kernel): (low, ~high, range), one count n + m
        uint32_t nh = 0u, range = 0xFFFFFFFFu;
        for (long long i = 0; i < N; ++i) {
            const uint16_t *row = cdf + i * row_stride;
            const uint32_t d = value - low;
            if (d > range) {   // outside [low, high]: the kernel hands such a stream to the generic decoder
                for (long long j = i; j < N; ++j) sym_out[j] = -1;
                return;
            }
            auto scaled = [&](uint32_t c) { return (uint32_t)(((uint64_t)range * c + c) >> 16); };
            uint32_t rank = 0;
            for (uint32_t m = 0; m <= top; ++m) rank += scaled(row[m]) <= d;
            const uint32_t x = rank ? rank - 1 : 0;
            sym_out[i] = (int16_t)x;
            if (i == N - 1) break;
            uint32_t msb;
            const int c = l3c::lean_advance(low, nh, range, scaled(row[x]), x == top ? 0u : scaled(row[x + 1]), x == top, msb);
            value = (c ? ((value << c) | src.take(c)) : value) ^ msb;
        }
        return;
    }
    for (long long i = 0; i < N; ++i) {
        const uint16_t *row = cdf + i * row_stride;
        const uint32_t count = l3c::decode_count(low, high, value);
        uint32_t x;
        if (monotone == 2 && value >= low && value <= high) {
            // division-free rank (ac_decode_kernel fast path): cdf[m] <= count  <=>  (span * cdf[m]) >> 16 <= value - low
            const uint32_t range = high - low, d = value - low;
            uint32_t rank = 0;
            for (uint32_t m = 0; m <= top; ++m) rank += (uint32_t)(((uint64_t)range * row[m] + row[m]) >> 16) <= d;
            x = rank ? rank - 1 : 0;
            sym_out[i] = (int16_t)x;
            if (i == N - 1) break;
            // the scaled entries ARE the interval offsets: low' = low + t[x], high' = low - 1 + t[x+1] (top symbol: high)
            const uint32_t t_lo = (uint32_t)(((uint64_t)range * row[x] + row[x]) >> 16);
            const uint32_t new_high = x == top ? high : low - 1u + (uint32_t)(((uint64_t)range * row[x + 1] + row[x + 1]) >> 16);
            const uint32_t new_low = low + t_lo;
            int n, m2;
            uint32_t nl, nh;
            l3c::renorm_counts(new_low, new_high, n, m2, nl, nh);
            if (n >= 32) value = src.take(32);
            else if (n) value = (value << n) | src.take(n);
            if (m2) value = ((value << m2) ^ 0x80000000u) | src.take(m2);
            low = nl;
            high = nh;
            continue;
        } else if (monotone) {  // what the wavefront does: rank of `count` among the valid entries (ballot + popcount)
            uint32_t rank = 0;
            for (uint32_t m = 0; m <= top; ++m) rank += row[m] <= count;
            x = rank ? rank - 1 : 0;
        } else {
            x = l3c::ref_binsearch([&](uint32_t m) { return (uint32_t)row[m]; }, count, top);
        }
        sym_out[i] = (int16_t)x;
        if (i == N - 1) break;
        const uint32_t c_lo = row[x];
        const uint32_t c_hi = x == top ? 0x10000u : row[x + 1];
        l3c::decode_advance(low, high, value, c_lo, c_hi, src);
    }
}
// Random single steps from every kind of renormalised state: the lane-pair form (role_term / role_shift / role_renorm / role_range) and
// the decoder's lean_advance against interval_update + renorm_counts.  Returns the number of differing results (0 expected).
long long hostsim_step_forms_agree(long long n, unsigned long long seed) {
    unsigned long long x = seed * 0x9E3779B97F4A7C15ull + 1;
    auto rnd = [&]() { x ^= x << 13; x ^= x >> 7; x ^= x << 17; return (uint32_t)(x >> 16); };
    long long bad = 0;
    for (long long i = 0; i < n; ++i) {
        // a state the coder can be in: the full range, or low = 0..., high = 1... and not both in the underflow position
        uint32_t low, high;
        const uint32_t kind = rnd() % 8;
        if (kind == 0) { low = 0; high = 0xFFFFFFFFu; }
        else {
            low = rnd() & 0x7FFFFFFFu;
            high = rnd() | 0x80000000u;
            if (kind == 1) { low = 0x3FFFFFFFu; high = 0xC0000000u; }                    // just outside the underflow position
            if (kind == 2) { low &= 0x3FFFFFFFu; }                                        // low in the lowest quarter
            if ((low & 0x40000000u) && !(high & 0x40000000u)) high |= 0x40000000u;         // (the E3 position is never a resting state)
        }
        uint32_t c_lo = rnd() % 65536u, c_hi = rnd() % 65537u;
        if (kind == 3) { c_lo = 0; c_hi = 0x10000u; }
        if (kind == 4) { c_hi = c_lo + 1u; }                                              // width-1 interval
        if (kind == 5) { c_hi = 0x10000u; }
        if (c_hi <= c_lo) { const uint32_t t = c_lo; c_lo = c_hi; c_hi = t + 1u; }
        if (c_hi > 0x10000u) c_hi = 0x10000u;
        if (c_lo >= c_hi) c_lo = c_hi - 1u;
        // reference form
        uint32_t l1 = low, h1 = high;
        l3c::interval_update(l1, h1, c_lo, c_hi);
        int n1, m1;
        uint32_t lf, hf;
        l3c::renorm_counts(l1, h1, n1, m1, lf, hf);
        // lane pair
        const uint32_t range = high - low;
        uint32_t u1[2] = {low + l3c::role_term(range, l3c::role_word(c_lo, c_hi, 0), l3c::role_round(0)),
                          ~high + l3c::role_term(range, l3c::role_word(c_lo, c_hi, 1), l3c::role_round(1))};
        const int t = l3c::role_shift(u1[0], u1[1]);
        const uint32_t u0f = l3c::role_renorm(u1[0], t), u1f = l3c::role_renorm(u1[1], t);
        bool ok = u1[0] == l1 && u1[1] == ~h1 && t == n1 + m1 && u0f == lf && u1f == ~hf && l3c::role_range(u0f, u1f) == hf - lf;
        ok = ok && l3c::role_shift(u1[1], u1[0]) == t;
        const uint32_t rec = l3c::record_from_pair(u1[0], u1[1]);
        ok = ok && l3c::record_n(rec) == (uint32_t)n1 && l3c::record_m(rec) == (uint32_t)m1;
        // decoder form
        uint32_t dl = low, dnh = ~high, dr = range, msb;
        const uint32_t t_lo = (uint32_t)(((uint64_t)range * c_lo + c_lo) >> 16), t_hi = (uint32_t)(((uint64_t)range * c_hi + c_hi) >> 16);
        const int c = l3c::lean_advance(dl, dnh, dr, t_lo, t_hi, c_hi == 0x10000u, msb);
        ok = ok && c == n1 + m1 && dl == lf && dnh == ~hf && dr == hf - lf && msb == (m1 ? 0x80000000u : 0u);
        bad += ok ? 0 : 1;
    }
    return bad;
}
}

