// conv_small.hip -- the thin (3-, 5-channel) ends of the conv stack; all HBM-bound, no MFMA.
//
//   rgb_head_kernel      sub_rgb_mean -> MeanShift(1/128) -> conv3x3 3->Cf      (multiscale_network.py:241, head.py:26-41)
//   to_q_quantize_kernel conv1x1 Cf->C + hard quantiser                          (net.py:144-148, quantizer.py:72-87)
//   dec_head_kernel      conv1x1 C->Cf (+ fused coarser features)                (net.py:178-180)
//   sym_to_bn_kernel     to_bn                                                   (quantizer.py:44-47)
//   meanshift_planar / to_u8 / resample_u8 / u8_to_sym_bn   the RGB baselines' bicubic pyramid encoder, Pillow-exact
//                        (net.py:65-80, images_loader.py:277-288; coefficient tables from helpers/pil_resample.py)
// Wide tensors are pixel-major: 16 consecutive lanes own the 64 channels of one pixel as float4s, so a wavefront
// reads/writes 4 pixels x 256 B = 1 KB contiguous per instruction.
#include "l3c_common.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int RH_TH = 8, RH_TW = 32, RH_NT = 4;   // tile of the 3x3 head; tiles a block walks down its column

__global__ __launch_bounds__(256) void rgb_head_kernel(const float *__restrict__ img, const float *__restrict__ w1,
                                                       const float *__restrict__ b1, const float *__restrict__ w2,
                                                       const float *__restrict__ b2, const float *__restrict__ w3,
                                                       const float *__restrict__ b3, int H, int W, int Cf,
                                                       float *__restrict__ out, float *__restrict__ shifted_out) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *s_w = smem;                                   // [27][Cf]  (tap-major: (ky*3+kx)*3 + ci)
    float *s_in = smem + 27 * Cf;                        // [3][RH_TH+2][RH_TW+2]
    constexpr int IH = RH_TH + 2, IW = RH_TW + 2;
    const int tid = threadIdx.x;
    const int b = blockIdx.z;
    const int ox0 = blockIdx.x * RH_TW;
    for (int i = tid; i < 27 * Cf; i += 256) {
        const int co = i % Cf, r = i / Cf;               // r = (ky*3+kx)*3 + ci
        const int ci = r % 3, tap = r / 3;
        s_w[i] = w3[((size_t)co * 3 + ci) * 9 + tap];
    }
    __syncthreads();
    const int quads = Cf / 4;                            // 16 lanes per pixel when Cf == 64
    const int q = tid % quads;
    const int pix_per_pass = 256 / quads;
    // A thread keeps its four output channels for every pixel it visits: their 27 weight quads live in registers (read from
    // LDS per pixel they were 4/5 of the kernel's LDS traffic), and a block walks RH_NT tiles down its column so that the
    // weight gather and the register fill are paid once per RH_NT * 256 pixels.  Same products, same order.
    f32x4 wreg[27];
#pragma unroll
    for (int t = 0; t < 27; ++t) wreg[t] = *reinterpret_cast<const f32x4 *>(&s_w[t * Cf + q * 4]);
    const f32x4 bias = *reinterpret_cast<const f32x4 *>(&b3[q * 4]);
    const size_t plane = (size_t)H * W;
    const float *im = img + (size_t)b * 3 * plane;
    const int Hb = H, Wb = W;
    if (ox0 >= Wb) return;
    for (int nt = 0; nt < RH_NT; ++nt) {
        const int oy0 = (blockIdx.y * RH_NT + nt) * RH_TH;
        if (oy0 >= Hb) break;
        if (nt) __syncthreads();                         // the previous tile's s_in is no longer read
        for (int i = tid; i < IH * IW; i += 256) {
            const int r = i / IW, c = i % IW;
            const int y = oy0 + r - 1, x = ox0 + c - 1;
            float z[3] = {0.f, 0.f, 0.f};
            if (y >= 0 && y < Hb && x >= 0 && x < Wb) {
                float v[3], u[3];
#pragma unroll
                for (int k = 0; k < 3; ++k) v[k] = im[k * plane + (size_t)y * W + x];
#pragma unroll
                for (int o = 0; o < 3; ++o) u[o] = ((v[0] * w1[o * 3 + 0] + v[1] * w1[o * 3 + 1]) + v[2] * w1[o * 3 + 2]) + b1[o];
#pragma unroll
                for (int o = 0; o < 3; ++o) z[o] = ((u[0] * w2[o * 3 + 0] + u[1] * w2[o * 3 + 1]) + u[2] * w2[o * 3 + 2]) + b2[o];
                if (shifted_out && r >= 1 && r <= RH_TH && c >= 1 && c <= RH_TW) {
#pragma unroll
                    for (int o = 0; o < 3; ++o) shifted_out[((size_t)b * 3 + o) * plane + (size_t)y * W + x] = z[o];
                }
            }
#pragma unroll
            for (int o = 0; o < 3; ++o) s_in[(o * IH + r) * IW + c] = z[o];
        }
        __syncthreads();
        for (int pp = tid / quads; pp < RH_TH * RH_TW; pp += pix_per_pass) {
            const int r = pp / RH_TW, c = pp % RH_TW;
            const int y = oy0 + r, x = ox0 + c;
            if (y >= Hb || x >= Wb) continue;
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                for (int kx = 0; kx < 3; ++kx)
#pragma unroll
                    for (int ci = 0; ci < 3; ++ci) {
                        // fused multiply-adds (round 5: the file is built with -ffp-contract=off, and as separate multiplications and
                        // additions these 27 x 4 terms made the kernel VALU-bound at 2.8 TB/s: ~250 instructions per 16-byte store)
                        const float v = s_in[(ci * IH + r + ky) * IW + c + kx];
                        const f32x4 wq = wreg[(ky * 3 + kx) * 3 + ci];
#pragma unroll
                        for (int e = 0; e < 4; ++e) acc[e] = __builtin_fmaf(v, wq[e], acc[e]);
                    }
            *reinterpret_cast<f32x4 *>(&out[(((size_t)b * H + y) * W + x) * Cf + q * 4]) = acc + bias;
        }
    }
}

__global__ __launch_bounds__(256) void to_q_quantize_kernel(const float *__restrict__ feat, const float *__restrict__ w,
                                                            const float *__restrict__ bias, const float *__restrict__ levels,
                                                            int64_t B, int64_t HW, int Cf, int C, int L,
                                                            int16_t *__restrict__ sym, float *__restrict__ bn_q,
                                                            float *__restrict__ bn) {
    const int64_t total = B * HW;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t b = i / HW, n = i % HW;
        const f32x4 *px = reinterpret_cast<const f32x4 *>(feat + i * Cf);
        float acc[8];
#pragma unroll
        for (int c = 0; c < 8; ++c) acc[c] = 0.0f;
        for (int k4 = 0; k4 < Cf / 4; ++k4) {
            const f32x4 v = px[k4];
#pragma unroll
            for (int c = 0; c < 8; ++c)
                if (c < C) {
                    const float *wc = w + c * Cf + k4 * 4;
                    acc[c] = fmaf(v[3], wc[3], fmaf(v[2], wc[2], fmaf(v[1], wc[1], fmaf(v[0], wc[0], acc[c]))));
                }
        }
#pragma unroll
        for (int c = 0; c < 8; ++c)
            if (c < C) {
                const float x = acc[c] + bias[c];
                int best = 0;
                float dbest = (x - levels[0]) * (x - levels[0]);
                for (int l = 1; l < L; ++l) {
                    const float d = (x - levels[l]) * (x - levels[l]);
                    if (d < dbest) {   // first minimum wins (torch.min)
                        dbest = d;
                        best = l;
                    }
                }
                const int64_t o = (b * C + c) * HW + n;
                sym[o] = (int16_t)best;
                bn_q[o] = levels[best];
                if (bn) bn[o] = x;
            }
    }
}

// The same arithmetic (a thread per pixel, the products of a pixel summed in the same order), the features staged through LDS: read
// straight from memory a wavefront's load touches 64 pixels' 256-byte records for 16 bytes each (measured: 1.5 TB/s, 2.7 ms a step
// for 4.2 GB); here a block's tile of 256 pixels is ONE contiguous 64 KB run, fetched with 16 coalesced 16-byte loads per thread that are
// all in flight together, and a thread then reads its pixel's record from LDS (row stride Cf + 4 floats: 16 lanes of a ds_read_b128
// cover the 64 banks once).  Cf % 4 == 0, Cf <= 64.
#ifndef L3C_TQ_PIX
#define L3C_TQ_PIX 256
#endif
#ifndef L3C_TQ_GRID
#define L3C_TQ_GRID 512       // blocks of a launch (two per CU): each walks its tiles with the next one's loads in flight [1.276 ms against 1.301 with 2048, 1.394 with a block per tile]
#endif
constexpr int TQ_PIX = L3C_TQ_PIX, TQ_MAX_CF = 64, TQ_LD = TQ_MAX_CF + 4;
__global__ __launch_bounds__(TQ_PIX) void to_q_quantize_tile_kernel(const float *__restrict__ feat, const float *__restrict__ w,
                                                                    const float *__restrict__ bias, const float *__restrict__ levels,
                                                                    int64_t B, int64_t HW, int Cf, int C, int L,
                                                                    int16_t *__restrict__ sym, float *__restrict__ bn_q,
                                                                    float *__restrict__ bn) {
    __shared__ __attribute__((aligned(16))) float tile[TQ_PIX * TQ_LD];
    const int64_t total = B * HW;
    const int tid = threadIdx.x;
    const int quads = Cf / 4;
    constexpr int U = TQ_MAX_CF / 4;                        // 16-byte loads per thread and tile when Cf == 64
    const int64_t stride = (int64_t)gridDim.x * TQ_PIX;
    f32x4 v[U];
    auto fetch = [&](int64_t t0) {                          // all loads of a tile in flight together; past the end: the tile's first piece
        const int npix = (int)((total - t0) < TQ_PIX ? (total - t0) : TQ_PIX);
        const int n4 = npix * quads;
        const f32x4 *src = reinterpret_cast<const f32x4 *>(feat + t0 * Cf);
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int i = tid + u * TQ_PIX;
            v[u] = src[i < n4 ? i : 0];
        }
    };
    int64_t t0 = (int64_t)blockIdx.x * TQ_PIX;
    if (t0 < total) fetch(t0);
    for (; t0 < total; t0 += stride) {
        const int npix = (int)((total - t0) < TQ_PIX ? (total - t0) : TQ_PIX);
        const int n4 = npix * quads;
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int i = tid + u * TQ_PIX;
            if (i < n4) {
                const int p = i / quads, q = i - p * quads;
                *reinterpret_cast<f32x4 *>(&tile[p * TQ_LD + q * 4]) = v[u];
            }
        }
        __syncthreads();
        if (t0 + stride < total) fetch(t0 + stride);        // the next tile's loads fly while this one is computed
        if (tid < npix) {
            const int64_t i = t0 + tid;
            const int64_t b = i / HW, n = i % HW;
            const f32x4 *px = reinterpret_cast<const f32x4 *>(&tile[tid * TQ_LD]);
            float acc[8];
#pragma unroll
            for (int c = 0; c < 8; ++c) acc[c] = 0.0f;
            for (int k4 = 0; k4 < quads; ++k4) {
                const f32x4 x4 = px[k4];
#pragma unroll
                for (int c = 0; c < 8; ++c)
                    if (c < C) {
                        const float *wc = w + c * Cf + k4 * 4;
                        acc[c] = fmaf(x4[3], wc[3], fmaf(x4[2], wc[2], fmaf(x4[1], wc[1], fmaf(x4[0], wc[0], acc[c]))));
                    }
            }
#pragma unroll
            for (int c = 0; c < 8; ++c)
                if (c < C) {
                    const float x = acc[c] + bias[c];
                    int best = 0;
                    float dbest = (x - levels[0]) * (x - levels[0]);
                    for (int l = 1; l < L; ++l) {
                        const float d = (x - levels[l]) * (x - levels[l]);
                        if (d < dbest) {   // first minimum wins (torch.min)
                            dbest = d;
                            best = l;
                        }
                    }
                    const int64_t o = (b * C + c) * HW + n;
                    sym[o] = (int16_t)best;
                    bn_q[o] = levels[best];
                    if (bn) bn[o] = x;
                }
        }
        __syncthreads();                                    // the tile is overwritten by the next turn
    }
}

// A thread owns one channel quad (q = tid % quads: its 4 x C weights and its bias stay in registers) and walks pixels of ONE
// image (blockIdx.y) with 32-bit indices -- the first version divided 64-bit indices per element and re-read the weights per
// pixel: 1.7 TB/s; the kernel only moves 5 + 64 (+ 64) floats per pixel.
constexpr int DH_MAX_C = 8;
__global__ __launch_bounds__(256) void dec_head_kernel(const float *__restrict__ bn_q, const float *__restrict__ w,
                                                       const float *__restrict__ bias, const float *__restrict__ fuse,
                                                       int64_t B, int64_t HW, int C, int Cf, float *__restrict__ out) {
    const int quads = Cf / 4;
    const int q = threadIdx.x % quads, pl = threadIdx.x / quads, ppb = 256 / quads;
    const int64_t b = blockIdx.y;
    float wr[4][DH_MAX_C];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int c = 0; c < DH_MAX_C; ++c) wr[j][c] = c < C ? w[(q * 4 + j) * C + c] : 0.0f;
    const f32x4 bq = *reinterpret_cast<const f32x4 *>(&bias[q * 4]);
    const float *bn_b = bn_q + b * C * HW;
    const float *fuse_b = fuse ? fuse + b * HW * Cf + q * 4 : nullptr;
    float *out_b = out + b * HW * Cf + q * 4;
    const int hw = (int)HW;
    for (int64_t n = (int64_t)blockIdx.x * ppb + pl; n < hw; n += (int64_t)gridDim.x * ppb) {   // (hw may approach 2^31)
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int c = 0; c < DH_MAX_C; ++c) {
            if (c < C) {
                const float v = bn_b[(int64_t)c * HW + n];
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[j] = fmaf(v, wr[j][c], acc[j]);
            }
        }
        f32x4 r = acc + bq;
        if (fuse_b) r = r + *reinterpret_cast<const f32x4 *>(&fuse_b[(int64_t)n * Cf]);
        *reinterpret_cast<f32x4 *>(&out_b[(int64_t)n * Cf]) = r;
    }
}

__global__ __launch_bounds__(256) void sym_to_bn_kernel(const int16_t *__restrict__ sym, int64_t n, float bin_width,
                                                        float x_min, float *__restrict__ bn) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        bn[i] = (float)sym[i] * bin_width + x_min;   // two roundings (file is built with -ffp-contract=off)
}

// ---- RGB baselines: bicubic pyramid encoder (modules/net.py:72-80, dataloaders/images_loader.py:277-288) ------------------------

__global__ __launch_bounds__(256) void meanshift_planar_kernel(const float *__restrict__ img, const float *__restrict__ w,
                                                               const float *__restrict__ b, int64_t B, int64_t HW,
                                                               float *__restrict__ out) {
    const int64_t total = B * HW;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t bi = i / HW, n = i % HW;
        const float *px = img + bi * 3 * HW + n;
        const float v0 = px[0], v1 = px[HW], v2 = px[2 * HW];
#pragma unroll
        for (int o = 0; o < 3; ++o)
            out[(bi * 3 + o) * HW + n] = ((v0 * w[o * 3 + 0] + v1 * w[o * 3 + 1]) + v2 * w[o * 3 + 2]) + b[o];
    }
}

// x + mean -> clamp(0, 255) -> round half to even -> uint8   (net.py:73-74)
__global__ __launch_bounds__(256) void to_u8_kernel(const float *__restrict__ x, float m0, float m1, float m2, int64_t B,
                                                    int64_t HW, uint8_t *__restrict__ out) {
    const int64_t total = B * 3 * HW;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)((i / HW) % 3);
        const float v = x[i] + (c == 0 ? m0 : c == 1 ? m1 : m2);
        out[i] = (uint8_t)rintf(fminf(fmaxf(v, 0.0f), 255.0f));
    }
}

// One pass of Pillow's ImagingResample for 8-bit data (Resample.c: ImagingResampleHorizontal_8bpc / Vertical_8bpc):
// out = clip8((2^21 + sum_k in[xmin + k] * kk[k]) >> 22).  `planes` images of H x W, resampled along W (axis 0) or H (axis 1).
__global__ __launch_bounds__(256) void resample_u8_kernel(const uint8_t *__restrict__ in, int64_t planes, int H, int W, int axis,
                                                          int out_size, const int *__restrict__ bounds,
                                                          const int *__restrict__ kk, int ksize, uint8_t *__restrict__ out) {
    const int oh = axis ? out_size : H, ow = axis ? W : out_size;
    const int64_t total = planes * oh * ow;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int ox = (int)(i % ow), oy = (int)((i / ow) % oh);
        const int64_t pl = i / ow / oh;
        const int o = axis ? oy : ox;
        const int first = bounds[o * 2], count = bounds[o * 2 + 1];
        const int *k = kk + (int64_t)o * ksize;
        const uint8_t *src = in + pl * H * W;
        int acc = 1 << 21;
        if (axis) {
            for (int t = 0; t < count; ++t) acc += (int)src[(int64_t)(first + t) * W + ox] * k[t];
        } else {
            for (int t = 0; t < count; ++t) acc += (int)src[(int64_t)oy * W + first + t] * k[t];
        }
        acc >>= 22;   // arithmetic shift, then clip8
        out[i] = (uint8_t)(acc < 0 ? 0 : (acc > 255 ? 255 : acc));
    }
}

// symbols = pixel values (L = 256), bn = value - mean   (net.py:76-80)
__global__ __launch_bounds__(256) void u8_to_sym_bn_kernel(const uint8_t *__restrict__ in, float m0, float m1, float m2,
                                                           int64_t B, int64_t HW, int16_t *__restrict__ sym,
                                                           float *__restrict__ bn) {
    const int64_t total = B * 3 * HW;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)((i / HW) % 3);
        const int v = in[i];
        sym[i] = (int16_t)v;
        bn[i] = (float)v - (c == 0 ? m0 : c == 1 ? m1 : m2);
    }
}

int grid_1d(int64_t total, int block) {
    int64_t g = (total + block - 1) / block;
    return (int)(g < 1 ? 1 : (g > 8192 ? 8192 : g));
}

}  // namespace

extern "C" {

int l3c_rgb_head(const float *img, const float *w1, const float *b1, const float *w2, const float *b2, const float *w3,
                 const float *b3, int B, int H, int W, int Cf, float *out, float *shifted_out, l3c_stream_t stream) {
    L3C_REQUIRE(img && w1 && b1 && w2 && b2 && w3 && b3 && out, "null pointer");
    L3C_REQUIRE(B > 0 && B < 65536 && H > 0 && W > 0, "bad shape");
    L3C_REQUIRE(Cf % 4 == 0 && Cf >= 4 && Cf <= 256 && 256 % (Cf / 4) == 0, "Cf must be 4 * a divisor of 256");
    const size_t lds = (size_t)(27 * Cf + 3 * (RH_TH + 2) * (RH_TW + 2)) * sizeof(float);
    const dim3 grid((unsigned)((W + RH_TW - 1) / RH_TW), (unsigned)((H + RH_TH * RH_NT - 1) / (RH_TH * RH_NT)), (unsigned)B);
    hipLaunchKernelGGL(rgb_head_kernel, grid, dim3(256), lds, l3c::as_stream(stream), img, w1, b1, w2, b2, w3, b3, H, W,
                       Cf, out, shifted_out);
    return l3c::check_launch("rgb_head_kernel");
}

int l3c_to_q_quantize(const float *feat, const float *w, const float *b, const float *levels, int64_t B, int64_t HW,
                      int Cf, int C, int L, int16_t *sym, float *bn_q, float *bn, l3c_stream_t stream) {
    L3C_REQUIRE(feat && w && b && levels && sym && bn_q, "null pointer");
    L3C_REQUIRE(B > 0 && HW > 0 && Cf % 4 == 0 && C > 0 && C <= 8 && L > 0 && L <= 32767, "bad shape (C <= 8)");
#ifndef L3C_TO_Q_DIRECT
    if (Cf <= TQ_MAX_CF && (reinterpret_cast<uintptr_t>(feat) & 15) == 0) {
        int64_t tiles = (B * HW + TQ_PIX - 1) / TQ_PIX;
        if (tiles > L3C_TQ_GRID) tiles = L3C_TQ_GRID;
        hipLaunchKernelGGL(to_q_quantize_tile_kernel, dim3((unsigned)tiles), dim3(TQ_PIX), 0, l3c::as_stream(stream), feat, w, b,
                           levels, B, HW, Cf, C, L, sym, bn_q, bn);
        return l3c::check_launch("to_q_quantize_tile_kernel");
    }
#endif
    hipLaunchKernelGGL(to_q_quantize_kernel, dim3(grid_1d(B * HW, 256)), dim3(256), 0, l3c::as_stream(stream), feat, w, b,
                       levels, B, HW, Cf, C, L, sym, bn_q, bn);
    return l3c::check_launch("to_q_quantize_kernel");
}

int l3c_dec_head(const float *bn_q, const float *w, const float *b, const float *fuse, int64_t B, int64_t HW, int C,
                 int Cf, float *out, l3c_stream_t stream) {
    L3C_REQUIRE(bn_q && w && b && out, "null pointer");
    L3C_REQUIRE(B > 0 && B < 65536 && HW > 0 && HW < (1ll << 31) && C > 0 && C <= DH_MAX_C, "bad shape (C <= 8, one image < 2^31 pixels)");
    L3C_REQUIRE(Cf % 4 == 0 && Cf >= 4 && Cf <= 1024 && 256 % (Cf / 4) == 0, "Cf must be 4 * a divisor of 256");
    const int64_t ppb = 256 / (Cf / 4);
    int64_t gx = (HW + ppb - 1) / ppb;
    if (gx > 2048) gx = 2048;
    hipLaunchKernelGGL(dec_head_kernel, dim3((unsigned)gx, (unsigned)B), dim3(256), 0, l3c::as_stream(stream), bn_q,
                       w, b, fuse, B, HW, C, Cf, out);
    return l3c::check_launch("dec_head_kernel");
}

int l3c_meanshift_planar(const float *img, const float *w, const float *b, int64_t B, int64_t HW, float *out,
                         l3c_stream_t stream) {
    L3C_REQUIRE(img && w && b && out && B > 0 && HW > 0, "bad arguments");
    hipLaunchKernelGGL(meanshift_planar_kernel, dim3(grid_1d(B * HW, 256)), dim3(256), 0, l3c::as_stream(stream), img, w, b, B,
                       HW, out);
    return l3c::check_launch("meanshift_planar_kernel");
}

int l3c_rgb_to_u8(const float *x, const float *mean3_host, int64_t B, int64_t HW, uint8_t *out, l3c_stream_t stream) {
    L3C_REQUIRE(x && mean3_host && out && B > 0 && HW > 0, "bad arguments");
    hipLaunchKernelGGL(to_u8_kernel, dim3(grid_1d(B * 3 * HW, 256)), dim3(256), 0, l3c::as_stream(stream), x, mean3_host[0],
                       mean3_host[1], mean3_host[2], B, HW, out);
    return l3c::check_launch("to_u8_kernel");
}

int l3c_resample_u8(const uint8_t *in, int64_t planes, int H, int W, int axis, int out_size, const int32_t *bounds,
                    const int32_t *kk, int ksize, uint8_t *out, l3c_stream_t stream) {
    L3C_REQUIRE(in && bounds && kk && out, "null pointer");
    L3C_REQUIRE(planes > 0 && H > 0 && W > 0 && out_size > 0 && ksize > 0 && (axis == 0 || axis == 1), "bad shape");
    const int64_t total = planes * (axis ? (int64_t)out_size * W : (int64_t)H * out_size);
    hipLaunchKernelGGL(resample_u8_kernel, dim3(grid_1d(total, 256)), dim3(256), 0, l3c::as_stream(stream), in, planes, H, W,
                       axis, out_size, bounds, kk, ksize, out);
    return l3c::check_launch("resample_u8_kernel");
}

int l3c_u8_to_sym_bn(const uint8_t *in, const float *mean3_host, int64_t B, int64_t HW, int16_t *sym, float *bn,
                     l3c_stream_t stream) {
    L3C_REQUIRE(in && mean3_host && sym && bn && B > 0 && HW > 0, "bad arguments");
    hipLaunchKernelGGL(u8_to_sym_bn_kernel, dim3(grid_1d(B * 3 * HW, 256)), dim3(256), 0, l3c::as_stream(stream), in,
                       mean3_host[0], mean3_host[1], mean3_host[2], B, HW, sym, bn);
    return l3c::check_launch("u8_to_sym_bn_kernel");
}

int l3c_sym_to_bn(const int16_t *sym, int64_t n, float bin_width, float x_min, float *bn, l3c_stream_t stream) {
    L3C_REQUIRE(sym && bn && n > 0, "bad arguments");
    hipLaunchKernelGGL(sym_to_bn_kernel, dim3(grid_1d(n, 256)), dim3(256), 0, l3c::as_stream(stream), sym, n, bin_width,
                       x_min, bn);
    return l3c::check_launch("sym_to_bn_kernel");
}
}
