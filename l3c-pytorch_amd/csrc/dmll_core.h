// dmll_core.h -- the logistic sigmoid of the discretised-mixture head, shared by the kernels of dmll_kernels.hip (encoder
// intervals and decoder tables go through the SAME functions: identical entries on both sides) and by the exhaustive check of
// the test-only library (csrc/xcheck_dmll.hip).
#ifndef L3C_DMLL_CORE_H_
#define L3C_DMLL_CORE_H_

#include <hip/hip_runtime.h>

namespace l3c {

// torch.sigmoid as the oracle states it: 1 / (1 + exp(-a)), every operation an individually rounded fp32 one.
__device__ __forceinline__ float sigmoid_f(float a) { return 1.0f / (1.0f + expf(-a)); }

// The SAME function, bit for bit (tests/test_gpu_head.py compares the two on every one of the 2^32 float bit patterns), with
// what it does not need left out -- the table kernel spends its time here (8116 evaluations per pixel):
//   a >= 16.7:  expf(-a) <= 5.6e-8 < 2^-24, so 1.0f + expf(-a) rounds to 1.0f and the quotient is 1.0f;
//   a <= -89:   expf(-a) overflows to +inf (e^89 > FLT_MAX), 1.0f + inf = inf, 1.0f / inf = 0.0f;
//   -80 < a < 16.7:  expf(-a) is the library's argument reduction, v_exp_f32 and v_ldexp_f32 WITHOUT its two range clamps
//               (underflow for -a < -103.97, overflow for -a > 88.72: neither can happen here), and 1 / d for 1 <= d < 2^116
//               is the IEEE expansion (v_rcp_f32, one Newton step, two residual corrections) WITHOUT v_div_scale (a
//               divisor in this range is not scaled) and v_div_fixup (no special case can occur): 16 instructions
//               instead of 29, and none of the wait states the clamps' compare-and-select pairs need;
//   -89 < a <= -80:  the library path as it is (one evaluation in a few thousand).
// A CDF row evaluates every mixture component at all Lp targets; a component only has unsaturated terms within (-89 .. 16.7)
// sigma of its mean -- for the entries outside that band the sigmoid costs two compares (a divergent branch: a wavefront
// takes a longer path only if one of its lanes needs it).
__device__ __forceinline__ float sigmoid_mid(float a) {   // -80 < a < 16.7
    const float kNegLog2e = __uint_as_float(0xbfb8aa3bu);       // -log2(e) rounded to fp32 ...
    const float kNegLog2eLo = __uint_as_float(0xb2a5705fu);     // ... and what the rounding left (-1.92596286e-8)
    const float t = kNegLog2e * a;                              // exp(-a) = 2^(-a log2 e)
    const float n = __builtin_rintf(t);
    float f = t - n;
    float c = __builtin_fmaf(a, kNegLog2e, -t);                 // the rounding error of t ...
    c = __builtin_fmaf(kNegLog2eLo, a, c);                      // ... and the low part of the constant
    f = f + c;
    const float e = __builtin_amdgcn_ldexpf(__builtin_amdgcn_exp2f(f), (int)n);
    const float d = 1.0f + e;
    float r = __builtin_amdgcn_rcpf(d);
    r = __builtin_fmaf(__builtin_fmaf(-d, r, 1.0f), r, r);
    float q = r;                                                // 1.0f * r
    q = __builtin_fmaf(__builtin_fmaf(-d, q, 1.0f), r, q);
    return __builtin_fmaf(__builtin_fmaf(-d, q, 1.0f), r, q);
}

__device__ __forceinline__ float sigmoid_sat(float a) {
    if (a >= 16.7f) return 1.0f;
    if (a <= -89.0f) return 0.0f;
    if (a > -80.0f) return sigmoid_mid(a);
    return sigmoid_f(a);   // also NaN
}

// ---- the mixture of one (pixel, channel): shared by the table / interval kernels (csrc/dmll_kernels.hip) and by the range decoder,
// which evaluates a whole row itself when a symbol falls outside the 64-entry window row it was handed (csrc/ac_kernels.hip).  ONE
// definition, every operation an individually rounded fp32 one in a fixed order (-ffp-contract=off, K walked sequentially): whoever
// evaluates an entry gets the same bits.
//   pi  = exp(l - max) / sum_k exp(l - max)            F.softmax(dim=K)
//   mu' = mu (+ sigmoid(lam) * x_prev ...)              logistic_mixture.py:262-272
//   ls  = max(log_sigma, -7)                            :260
//   cdf(l) = sum_k pi_k * sigmoid((t_l - mu'_k) * exp(-ls_k))   sequential k, mul then add   (torchac.py:181-200)
//   entry  = uint16(rint(cdf * (65536 - (Lp-1))) + l)            round-half-even, wraps mod 2^16 (torchac.py:203-213)
// Accessor concept: float operator()(int ch) -- channel `ch` (0..Kp-1) of the current pixel.
constexpr float kLogScalesMin = -7.0f;

struct MixStats {
    float max_logit, denom;
};

template <class Get>
__device__ __forceinline__ MixStats mix_stats(Get get, int C, int K, int c) {
    MixStats s;
    s.max_logit = get(c * K);
    for (int k = 1; k < K; ++k) s.max_logit = fmaxf(s.max_logit, get(c * K + k));
    s.denom = 0.0f;
    for (int k = 0; k < K; ++k) s.denom = s.denom + expf(get(c * K + k) - s.max_logit);
    return s;
}

struct MixComponent {
    float pi, mu, log_sigma;
};

// x0, x1: actual values of the previously coded channels (RGB scale only, c > 0).
// e_k = expf(logit_k - max): the softmax numerator.  Callers that have just computed it for the denominator pass it in (the
// same operation on the same operands: the same bits) instead of paying a second expf.
template <class Get>
__device__ __forceinline__ MixComponent mix_component_e(Get get, const MixStats &st, float e_k, int C, int K, int rgb, int c, int k,
                                                        float x0, float x1) {
    const int CK = C * K;
    MixComponent m;
    m.pi = e_k / st.denom;
    m.mu = get(CK + c * K + k);
    m.log_sigma = fmaxf(get(2 * CK + c * K + k), kLogScalesMin);
    if (rgb && c == 1) {
        m.mu = m.mu + sigmoid_f(get(3 * CK + k)) * x0;
    } else if (rgb && c == 2) {
        const float a = sigmoid_f(get(3 * CK + K + k)) * x0;
        const float b = sigmoid_f(get(3 * CK + 2 * K + k)) * x1;
        m.mu = m.mu + (a + b);
    }
    return m;
}

template <class Get>
__device__ __forceinline__ MixComponent mix_component(Get get, const MixStats &st, int C, int K, int rgb, int c, int k,
                                                      float x0, float x1) {
    return mix_component_e(get, st, expf(get(c * K + k) - st.max_logit), C, K, rgb, c, k, x0, x1);
}

__device__ __forceinline__ float cdf_term(float pi, float mu, float inv_sigma, float target) {
    return pi * sigmoid_sat((target - mu) * inv_sigma);
}

__device__ __forceinline__ uint32_t cdf_quantise(float cdf, float scale, int l) {
    return (uint32_t)((int)rintf(cdf * scale) + l) & 0xFFFFu;
}

// ---- window rows (round 5): the RGB decoder's 64-entry table rows ------------------------------------------------------------
// A full row has Lp = 257 entries; the symbol almost always lies near the mixture's mean.  A WINDOW row has kWinLp = 65 uint16:
//     e[j] = cdf[w0 + j], j = 0 .. 63            e[64] = w0 (the window's offset, 0 .. kWinMaxOffset)
// (cdf[0] is NOT zero in general: it is the mixture's mass below the first bin edge, torchac.py:181-213 -- so e[0] must be the true entry.)
// Ranked like a 64-symbol alphabet -- x' = max(#{j: e[j] <= count}, 1) - 1, as every decoder here ranks -- it yields
//     1 <= x' <= 62           symbol w0 + x', interval [e[x'], e[x' + 1]) -- exactly the full row's
//     x' == 0                 symbol <= w0: taken as exact (symbol 0, interval [e[0], e[1])) iff w0 == 0, otherwise a MISS
//     x' == 63                symbol >= w0 + 63: exact (the top symbol 255, interval [e[63], 2^16)) iff w0 == kWinMaxOffset, else a MISS
// On a miss the decoder evaluates further entries of that pixel's row itself (the functions above: the same bits): the next 64 in the
// direction of the miss, as another window, until the symbol is inside (csrc/ac_kernels.hip: window_row_at).  Which form the rows of an
// image's chunk have is a pure function of the miss count the decoder reported two chunks earlier (see use_window).
constexpr int kWinLp = 65;
constexpr int kWinTop = 63;
constexpr int kWinMaxOffset = 192;      // 256 - 64

// Statistics word of a (stream, chunk): the number of misses (of would-be misses when its rows were full), with kWinBad set when that is
// more than 1/64 of the chunk's symbols; negative = unknown.  The WRITER (the chunk's decoder) judges the rate against its own chunk
// length, so that chunks of different lengths (the short probe chunks a stream starts with) need no bookkeeping on the reader's side.
constexpr int kWinBad = 0x40000000;
__host__ __device__ __forceinline__ bool use_window(int stat) { return stat >= 0 && stat < kWinBad; }
// one_in: the rate above which the chunk counts as bad -- 1/64 for a stream that is on window rows (it stays), 1/128 for one on full
// rows (it enters): the hysteresis keeps streams near the limit from flipping, and an estimate from full rows is a sample.
__host__ __device__ __forceinline__ int window_stat(unsigned misses, unsigned n_sym, unsigned one_in) {
    const unsigned m = misses < (unsigned)kWinBad ? misses : (unsigned)kWinBad - 1u;
    return (int)((unsigned long long)misses * one_in <= n_sym ? m : (m | (unsigned)kWinBad));
}

// offset of the window around the mixture's mean (sum_k pi_k mu_k, sequential): symbols w0 + 1 .. w0 + 62 decode without a miss
__device__ __forceinline__ int window_offset(float mean) {
    float m = mean == mean ? mean : 0.0f;
    m = fminf(fmaxf(m, 0.0f), 255.0f);
    const int w = (int)floorf(m) - 31;
    return w < 0 ? 0 : (w > kWinMaxOffset ? kWinMaxOffset : w);
}

__host__ __device__ __forceinline__ bool window_miss(unsigned xw, unsigned w0) {   // xw: rank inside the window, 0 .. 63
    return (xw == 0u && w0 != 0u) || (xw == (unsigned)kWinTop && w0 != (unsigned)kWinMaxOffset);
}
// would the symbol x of a FULL row have been a miss of the window at w0 (statistics of chunks decoded from full rows)
__host__ __device__ __forceinline__ bool window_would_miss(unsigned x, unsigned w0) {
    return !(x - w0 - 1u <= 61u) && !(x == 0u && w0 == 0u) && !(x == 255u && w0 == (unsigned)kWinMaxOffset);
}

}  // namespace l3c
#endif
