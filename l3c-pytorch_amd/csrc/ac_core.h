// ac_core.h -- loop-free formulation of the reference range coder's per-symbol state machine.
//
// The reference (torchac.cpp:152-227 encode, :299-381 decode) renormalises with a bit-at-a-time `while(true)` loop.
// Here every symbol costs a fixed, branch-light sequence so that 64 lanes coding 64 different streams stay converged:
//
//   interval update   low' = low + ((span*c_low) >> 16),  high' = low - 1 + ((span*c_high) >> 16),  span = high-low+1
//   E1/E2 run         the loop's "high < 2^31 -> emit 0" / "low >= 2^31 -> emit 1" iterations are exactly the common
//                     binary prefix of low' and high': n = clz(low' ^ high') bits, equal to the top n bits of low'
//                     (the first of them is followed by the `pending` complemented bits, torchac.cpp:87-92)
//   E3 run            afterwards low = 0..., high = 1...; the "low >= 2^30 && high < 3*2^30 -> pending++" iterations
//                     continue while bit 30 of low is 1 and bit 30 of high is 0, i.e. m = number of leading ones of
//                     ((low & ~high) << 1); after an E3 step neither E1 nor E2 can fire again for this symbol.
//   decoder           the same shifts applied to `value`: value = (value << n) | next n bits, and the m E3 steps
//                     (value -= 2^30; value = value<<1 | bit) collapse to ((value << m) ^ 2^31) | next m bits (mod 2^32).
//
// Everything is integer and bit-exact; tests/hostsim compiles this header with g++ and runs the reference KATs through
// it on the CPU (test infrastructure -- the product only instantiates it inside HIP kernels).
#ifndef L3C_AC_CORE_H_
#define L3C_AC_CORE_H_

#include <stdint.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define L3C_HD __host__ __device__ __forceinline__
#define L3C_HDM __host__ __device__ __forceinline__   /* member functions */
#else
#define L3C_HD static inline
#define L3C_HDM inline
#endif

namespace l3c {

L3C_HD int clz32(uint32_t x) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __clz((int)x);  // 32 for x == 0
#else
    return x ? __builtin_clz(x) : 32;
#endif
}

L3C_HD uint32_t ones(int n) {  // n in [0, 32]
    return n >= 32 ? 0xFFFFFFFFu : ((1u << n) - 1u);
}

// Packed interval word c_low | (c_high - 1) << 16 of the one-lane form (encode_state_step; the kernels use role_word below).
L3C_HD uint32_t pack_interval(uint32_t c_lo, uint32_t c_hi) { return (c_lo & 0xFFFFu) | ((c_hi - 1u) << 16); }
L3C_HD uint32_t interval_lo(uint32_t w) { return w & 0xFFFFu; }
L3C_HD uint32_t interval_hi(uint32_t w) { return (w >> 16) + 1u; }

// torchac.cpp:177-184.  (span * c) >> 16 with span = range + 1 computed as (range*c + c) >> 16 in 64 bits; the
// truncation to 32 bits and the wrapping adds are the reference's uint32 assignments.
L3C_HD void interval_update(uint32_t &low, uint32_t &high, uint32_t c_lo, uint32_t c_hi) {
    const uint32_t range = high - low;
    const uint64_t a = (uint64_t)range * c_hi + c_hi;
    const uint64_t b = (uint64_t)range * c_lo + c_lo;
    high = low - 1u + (uint32_t)(a >> 16);
    low = low + (uint32_t)(b >> 16);
}

// Shift out the common prefix (n bits) and the underflow run (m bits) of low/high.  Returns n, m.
L3C_HD void renorm_counts(uint32_t low, uint32_t high, int &n, int &m, uint32_t &low_out, uint32_t &high_out) {
    n = clz32(low ^ high);
    if (n >= 32) {
        low = 0u;
        high = 0xFFFFFFFFu;
    } else {
        low <<= n;
        high = (high << n) | ones(n);
    }
    m = clz32(~((low & ~high) << 1));
    if (m) {
        low = (low << m) & 0x7FFFFFFFu;
        high = (high << m) | 0x80000000u | ones(m);
    }
    low_out = low;
    high_out = high;
}

// ---- encoder -------------------------------------------------------------------------------------------------------
// Sink concept: void put(uint32_t bits, int count)  -- append the low `count` (0..32) bits of `bits`, MSB first.

template <class Sink>
L3C_HD void put_with_pending(Sink &sink, uint32_t bit, uint32_t &pending) {
    sink.put(bit, 1);
    const uint32_t fill = bit ? 0u : 0xFFFFFFFFu;
    while (pending) {
        const int k = pending > 32u ? 32 : (int)pending;
        sink.put(fill & ones(k), k);
        pending -= (uint32_t)k;
    }
}

template <class Sink>
L3C_HD void encode_symbol(uint32_t &low, uint32_t &high, uint32_t &pending, uint32_t c_lo, uint32_t c_hi, Sink &sink) {
    interval_update(low, high, c_lo, c_hi);
    int n, m;
    uint32_t nl, nh;
    renorm_counts(low, high, n, m, nl, nh);
    if (n) {
        put_with_pending(sink, low >> 31, pending);
        if (n > 1) sink.put((low << 1) >> (33 - n), n - 1);
    }
    pending += (uint32_t)m;
    low = nl;
    high = nh;
}

// torchac.cpp:209-219: one more pending bit, the quadrant bit, then the caller pads to a byte boundary with zeros.
template <class Sink>
L3C_HD void encode_finish(uint32_t low, uint32_t &pending, Sink &sink) {
    pending += 1u;
    put_with_pending(sink, low < 0x40000000u ? 0u : 1u, pending);
}

// ---- decoder -------------------------------------------------------------------------------------------------------
// Source concept: uint32_t take(int count) -- next `count` (0..32) bits MSB first, zeros past the end of the stream.

// torchac.cpp:329: count = uint16(((value - low + 1) * 2^16 - 1) / span) in uint64 arithmetic.
L3C_HD uint32_t decode_count(uint32_t low, uint32_t high, uint32_t value) {
    const uint32_t range = high - low;
    const uint64_t span = (uint64_t)range + 1u;
    if (value >= low && value <= high) {
        // quotient < 2^16: float estimate (error < 1) + exact integer fix-up
        const uint32_t d = value - low;
        const uint64_t num = (((uint64_t)d + 1u) << 16) - 1u;
        float qf = ((float)d + 1.0f) * 65536.0f;
#if defined(__HIP_DEVICE_COMPILE__)
        qf = qf * __frcp_rn((float)range + 1.0f);
#else
        qf = qf / ((float)range + 1.0f);
#endif
        uint32_t q = (uint32_t)qf;
        int64_t r = (int64_t)num - (int64_t)((uint64_t)q * span);
        if (r < 0) {
            q -= 1u;
            r += (int64_t)span;
        }
        if (r < 0) {
            q -= 1u;
            r += (int64_t)span;
        }
        if (r >= (int64_t)span) {
            q += 1u;
            r -= (int64_t)span;
        }
        if (r >= (int64_t)span) q += 1u;
        return q & 0xFFFFu;
    }
    // corrupt / foreign stream: reproduce the reference's wrapping 64-bit arithmetic literally
    const uint64_t num = ((uint64_t)value - (uint64_t)low + 1u) * 0x10000u - 1u;
    return (uint32_t)(num / span) & 0xFFFFu;
}

// binsearch (torchac.cpp:276-296) over a row accessed through `fetch(m)`; exact hit returns m.
template <class Fetch>
L3C_HD uint32_t ref_binsearch(Fetch fetch, uint32_t target, uint32_t top_symbol) {
    uint32_t left = 0, right = top_symbol + 1u;
    while (left + 1u < right) {
        const uint32_t mid = (left + right) >> 1;
        const uint32_t v = fetch(mid);
        if (v < target) left = mid;
        else if (v > target) right = mid;
        else return mid;
    }
    return left;
}

template <class Source>
L3C_HD void decode_advance(uint32_t &low, uint32_t &high, uint32_t &value, uint32_t c_lo, uint32_t c_hi, Source &src) {
    interval_update(low, high, c_lo, c_hi);
    int n, m;
    uint32_t nl, nh;
    renorm_counts(low, high, n, m, nl, nh);
    if (n >= 32) value = src.take(32);
    else if (n) value = (value << n) | src.take(n);
    if (m) value = ((value << m) ^ 0x80000000u) | src.take(m);
    low = nl;
    high = nh;
}


// ---- bit sink / source on 32-bit words ------------------------------------------------------------------------------
// The bitstream is MSB-first per byte (torchac.cpp:69-78), i.e. a sequence of big-endian 32-bit words.

L3C_HD uint32_t bswap32(uint32_t x) { return __builtin_bswap32(x); }

// Store concept: void operator()(uint32_t word_index, uint32_t little_endian_word)
template <class Store>
struct WordSink {
    Store store;
    uint64_t acc;     // low `nb` bits are valid, older bits above them are stale
    int nb;           // 0..31 between calls
    uint32_t nwords;
    L3C_HDM explicit WordSink(Store s) : store(s), acc(0), nb(0), nwords(0) {}
    L3C_HDM void put(uint32_t bits, int count) {
        acc = (acc << count) | bits;
        nb += count;
        if (nb >= 32) {
            nb -= 32;
            store(nwords++, bswap32((uint32_t)(acc >> nb)));
        }
    }
    // zero-pad to a byte boundary (torchac.cpp:79-86); returns the stream length in bytes
    L3C_HDM uint32_t finish() {
        const uint32_t nbytes = nwords * 4u + (uint32_t)((nb + 7) >> 3);
        if (nb) store(nwords, bswap32((uint32_t)(acc << (32 - nb))));
        return nbytes;
    }
};

// Fetch concept: uint32_t operator()(uint32_t word_index) -- big-endian value of word `word_index` of the stream,
// bytes at or past the end of the stream read as zero (torchac.cpp:104-108).
template <class Fetch>
struct WordSource {
    Fetch fetch;
    uint64_t acc;
    int nb;
    uint32_t next;
    L3C_HDM explicit WordSource(Fetch f) : fetch(f), acc(0), nb(0), next(0) {}
    L3C_HDM uint32_t take(int count) {
        if (nb < count) {
            acc = (acc << 32) | fetch(next++);
            nb += 32;
        }
        nb -= count;
        return (uint32_t)(acc >> nb) & ones(count);
    }
};


// ---- two-phase encoder ---------------------------------------------------------------------------------------------------
// Phase 1 (serial per stream, branch-free): only the interval recurrence.  Per symbol it produces a record
//     rec_lo = low' (the lower bound right after the interval update, before renormalisation)
//     rec_nm = n | m << 8      n = length of the common prefix of low'/high' (0..32), m = underflow run (0..31)
// which is everything the bitstream depends on: the symbol emits -- iff n > 0 -- the top n bits of low', the first of them
// followed by the complements owed for the `pending` underflow bits accumulated since the previous emitting symbol, and then
// sets pending = m (or adds m when n == 0).  Phase 2 turns records into bits and is parallel over symbols: pending is a
// segmented sum of m, bit offsets are a prefix sum of the emitted lengths (csrc/ac_kernels.hip: ac_pack_kernel).
L3C_HD void encode_state_step(uint32_t &low, uint32_t &high, uint32_t w, uint32_t &rec_lo, uint32_t &rec_nm) {
    const uint32_t c_lo = interval_lo(w), c_hi = interval_hi(w);
    const uint32_t range = high - low;
    const uint32_t hi1 = low - 1u + (uint32_t)(((uint64_t)range * c_hi + c_hi) >> 16);
    const uint32_t lo1 = low + (uint32_t)(((uint64_t)range * c_lo + c_lo) >> 16);
    int n, m;
    renorm_counts(lo1, hi1, n, m, low, high);
    rec_lo = lo1;
    rec_nm = (uint32_t)n | ((uint32_t)m << 8);
}

// Packed record: top n bits of low' in bits 10.., n in bits 5..9, m in bits 0..4.  After renormalisation the interval is
// wider than 2^30 and c_high > c_low, so low' and high' differ by >= 2^14 - 2 and share at most 18 leading bits: for every
// valid (strictly increasing) table n <= 18 and the record fits 28 bits.  Tables with c_high <= c_low are outside the
// contract of the encoder (the reference produces an undecodable stream for them too).
L3C_HD uint32_t pack_record(uint32_t rec_lo, uint32_t rec_nm) {
    const uint32_t n = rec_nm & 0xFFu, m = rec_nm >> 8;
    const uint32_t top = n ? rec_lo >> ((32u - n) & 31u) : 0u;
    return (top << 10) | ((n & 31u) << 5) | (m & 31u);
}
L3C_HD uint32_t record_n(uint32_t r) { return (r >> 5) & 31u; }
L3C_HD uint32_t record_m(uint32_t r) { return r & 31u; }
L3C_HD uint32_t record_top(uint32_t r) { return r >> 10; }

// ---- phase 1 on a lane PAIR: the role-symmetric form of the recurrence (round 4) ------------------------------------------
// A lone wavefront issues one instruction every ~4 cycles whatever the dependencies, so the chain's cost is its instruction
// COUNT.  encode_state_step spends ~45 on a symbol: both bounds go through the same multiply / shift / mask sequence, and the
// interval word is unpacked and the record packed on the chain.  Here a stream is coded by TWO adjacent lanes, role 0 holding
// u = low and role 1 holding u = ~high, so that ONE instruction sequence updates both bounds:
//     role 0   low'   = low   + floor(span * c_low / 2^16)
//     role 1   ~high' = ~high + ceil(span * (2^16 - c_high) / 2^16)        (high - high' = span - floor(span * c_high / 2^16))
// i.e. u' = u + role_term(range, c, round) with the role's own word c (role_word) and round = 0 / 65535.  With
// span = range + 1 = rh * 2^16 + rl + 1 the term is rh * c + (((rl + 1) * c + round) >> 16): c <= 65535, so every product
// fits 32 bits and the multiplies are the full-rate 24-bit ones.  The renormalisation needs nothing but the pair's u':
//     a = u'_0 ^ u'_1 = ~(low' ^ high'),  z = u'_0 & u'_1 = low' & ~high'
//     n + m = clz(~(a | z << 1))     (n leading agreeing bits, the differing bit, then the m bits of the underflow run)
//     u = (u' << (n + m)) & 0x7FFFFFFF for BOTH roles,  range = ~(u_0 + u_1)
// and the record of the symbol is the pair (low', ~high') itself: phase 2, parallel over symbols, derives n, m and the
// emitted bits from it (record_from_pair), so nothing is unpacked or packed on the chain.  ~15 instructions per symbol.
// Precondition as for the whole encoder: c_high > c_low (then low' < high' and n + m <= 31).
L3C_HD uint32_t role_word(uint32_t c_lo, uint32_t c_hi, int role) { return role ? 0x10000u - c_hi : c_lo; }
L3C_HD uint32_t role_round(int role) { return role ? 0xFFFFu : 0u; }

L3C_HD uint32_t mul24(uint32_t a, uint32_t b) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __umul24(a, b);
#else
    return (a & 0xFFFFFFu) * (b & 0xFFFFFFu);
#endif
}

L3C_HD uint32_t role_term(uint32_t range, uint32_t c, uint32_t round) {
    const uint32_t rl = range & 0xFFFFu, rh = range >> 16;
    return mul24(rh, c) + ((mul24(rl, c) + c + round) >> 16);
}

L3C_HD int role_shift(uint32_t u_self, uint32_t u_other) {
    const uint32_t h = ~((u_self ^ u_other) | ((u_self & u_other) << 1));   // != 0 for low' < high'
#if defined(__HIP_DEVICE_COMPILE__)
    int t;
    asm("v_ffbh_u32 %0, %1" : "=v"(t) : "v"(h));   // clz32 costs a v_min on top; -1 for h == 0, the shifts use 5 bits
    return t;
#else
    return clz32(h) & 31;
#endif
}
L3C_HD uint32_t role_renorm(uint32_t u, int t) { return (u << (t & 31)) & 0x7FFFFFFFu; }
L3C_HD uint32_t role_range(uint32_t u_self, uint32_t u_other) { return ~(u_self + u_other); }

// The packed record (pack_record) of a symbol from the pair phase 1 left in its two interval words.
L3C_HD uint32_t record_from_pair(uint32_t low1, uint32_t not_high1) {
    int n, m;
    uint32_t nl, nh;
    renorm_counts(low1, ~not_high1, n, m, nl, nh);
    return pack_record(low1, (uint32_t)n | ((uint32_t)m << 8));
}

// ---- the decoder's state in the same form (csrc/ac_kernels.hip: ac_decode_lean_kernel) ---------------------------------------
// (low, nh = ~high, range = high - low).  t_lo / t_hi: the scaled table entries (span * cdf[x]) >> 16 and (span * cdf[x + 1]) >> 16
// the symbol search has produced (low' = low + t_lo, high' = low - 1 + t_hi; the top symbol keeps high).  Returns c = n + m, the
// number of stream bits to shift into `value`, and msb: 0x80000000 iff m > 0 -- the bit the underflow steps take out of low and
// out of value:  ((value << n | bits_n) << m ^ 2^31) | bits_m  ==  (value << c | bits_c) ^ msb.
L3C_HD int lean_advance(uint32_t &low, uint32_t &nh, uint32_t &range, uint32_t t_lo, uint32_t t_hi, bool top_symbol, uint32_t &msb) {
    const uint32_t lo = low + t_lo;
    const uint32_t nh1 = top_symbol ? nh : 0u - (low + t_hi);   // ~(low - 1 + t_hi)
    const uint32_t h = ~((lo ^ nh1) | ((lo & nh1) << 1));        // role_shift: != 0 as lo < hi
#if defined(__HIP_DEVICE_COMPILE__)
    const int c = h ? __builtin_clz(h) : 31;   // (h == 0 only for a table that is not strictly increasing: such a stream runs on garbage, defined garbage)
#else
    const int c = clz32(h) & 31;
#endif
    const uint32_t lo_s = lo << c;
    msb = lo_s & 0x80000000u;
    low = lo_s ^ msb;
    nh = (nh1 << c) & 0x7FFFFFFFu;
    range = ~(low + nh);
    return c;
}

// Literal (serial) emission of one record -- the definition phase 2 must reproduce; also used for its rare long runs.
template <class Sink>
L3C_HD void emit_record(uint32_t rec_lo, uint32_t rec_nm, uint32_t &pending, Sink &sink) {
    const int n = (int)(rec_nm & 0xFFu), m = (int)(rec_nm >> 8);
    if (n) {
        put_with_pending(sink, rec_lo >> 31, pending);
        if (n > 1) sink.put((rec_lo << 1) >> (33 - n), n - 1);
        pending = (uint32_t)m;
    } else {
        pending += (uint32_t)m;
    }
}

}  // namespace l3c
#endif
