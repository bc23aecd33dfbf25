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

}  // namespace l3c
#endif
