// conv_mfma.hip -- fp32 implicit-GEMM convolution on v_mfma_f32_32x32x2_f32 (gfx950), plus a plain-VALU cross-check.
//
// Replaces the cuDNN convolutions behind the reference's conv factory (pytorch_ext.py:57-61) for every 64-/192-input-
// channel layer of the L3C stack: 3x3 (dilation 1, 2, 4), 5x5 stride 2, 1x1; epilogues: bias,
// ReLU, residual add (edsr.py:83-86, net.py:142, :181), PixelShuffle(2) (edsr.py:98-99).
//
// GEMM view: M = output pixels, N = output channels, K = taps x input channels.  Activations are pixel-major (NHWC) so
// that an input pixel's channels are contiguous: that makes the global->LDS staging coalesced (128 B per pixel per
// channel chunk), lets one ds_read_b128 deliver the A operands of four consecutive MFMA k-steps, and makes the D
// fragment (col = lane&31 = output channel) store 128 contiguous bytes per pixel.
//
// Block = 256 threads = 4 wavefronts, output tile TH x 32 pixels (TH = 4*MT rows) x 64 output channels:
//   wave w owns rows w*MT .. w*MT+MT-1 (each row of 32 pixels = one MFMA M-tile) x 2 N-tiles of 32 channels
//   -> MT*2 accumulators of 16 VGPRs each (v_mfma_f32_32x32x2_f32: A[i=lane&31][k=lane>>5], B[k=lane>>5][j=lane&31])
// Input channels are processed in chunks of CK: the (TH*S + halo) x (32*S + halo) x CK input patch is staged in LDS with
// a pixel stride of CK+4 floats (conflict-free ds_read_b128 across 32 consecutive pixels), then every tap walks it.
// K ordering trick: lane-half h = lane>>5 reads channels 8g+4h .. 8g+4h+3 with ONE b128 read and feeds element t to
// k-step t, so a k-step pairs channels (8g+t, 8g+4+t); the weights are pre-packed in exactly that fragment order
// (l3c_conv_pack_weights) and stream from L2 as one coalesced dwordx4 per lane per 4 k-steps.
// Stride-2 convs store even and odd input columns in separate halves of each LDS row so lanes still read consecutive
// pixels.  Everything is deterministic: fixed k order, no atomics, the tile schedule depends only on (H, W).
#include "l3c_common.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

struct ConvParams {
    const float *in;
    const float *w;
    const float *bias;
    const float *res;
    float *out;
    int in_cstride, in_coff, res_cstride, res_coff, out_cstride, out_coff;
    int B, Hin, Win, Cin, Cout, Hout, Wout;
    int pad, epilogue;
    int tiles_x, tiles_y, n_chunks_o;
    int total_blocks;
};

constexpr int TW = 32;

template <int KS, int STRIDE, int DIL, int CK, int MT>
struct Geo {
    static constexpr int TH = 4 * MT;
    static constexpr int IH = (TH - 1) * STRIDE + (KS - 1) * DIL + 1;
    static constexpr int IW = (TW - 1) * STRIDE + (KS - 1) * DIL + 1;
    static constexpr int IWH = (IW + 1) / 2;                      // half row (stride 2: even | odd columns)
    static constexpr int IWL = STRIDE == 2 ? 2 * IWH : IW;        // LDS row length in pixels
    static constexpr int PS = CK + 4;                             // LDS pixel stride in floats
    static constexpr int LDS_BYTES = IH * IWL * PS * 4;
    static constexpr int TAPS = KS * KS;
};

__device__ __forceinline__ int xcd_remap(int bid, int total) {
    // blocks are dispatched round-robin over the 8 XCDs (observed, speed only): give each XCD a contiguous range of
    // tiles so neighbouring tiles (shared halo) and the layer's weights stay in one L2.  Bijective for any `total`.
    const int q = total >> 3, r = total & 7;
    const int xcd = bid & 7, slot = bid >> 3;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + slot;
}

template <int KS, int STRIDE, int DIL, int CK, int MT>
__global__ __launch_bounds__(256, 3) void conv_mfma_kernel(const ConvParams p) {
    using G = Geo<KS, STRIDE, DIL, CK, MT>;
    extern __shared__ __attribute__((aligned(16))) float lds[];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int half = lane >> 5, lx = lane & 31;

    int w = xcd_remap(blockIdx.x, p.total_blocks);
    const int tile = w % (p.tiles_x * p.tiles_y);
    w /= p.tiles_x * p.tiles_y;
    const int chunk_o = w % p.n_chunks_o;
    const int b = w / p.n_chunks_o;
    const int ty = tile / p.tiles_x, tx = tile % p.tiles_x;
    const int oy0 = ty * G::TH, ox0 = tx * TW;
    const int iy0 = oy0 * STRIDE - p.pad, ix0 = ox0 * STRIDE - p.pad;

    f32x16 acc[MT][2];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mt][nt][r] = 0.0f;

    const int n_cc = p.Cin / CK;
    const int groups_total = p.Cin / 8;
    const f32x4 *wp = reinterpret_cast<const f32x4 *>(p.w) + (size_t)chunk_o * groups_total * G::TAPS * 2 * 64 + lane;
    const float *in_b = p.in + (size_t)b * p.Hin * p.Win * p.in_cstride + p.in_coff;

    for (int cc = 0; cc < n_cc; ++cc) {
        if (cc) __syncthreads();  // everyone is done reading the previous chunk
        // ---- stage the input patch: [IH][IW] pixels x CK channels, zero outside the image ----
        // All global loads of the chunk are issued before the first LDS write (one memory round trip per chunk, not one
        // per iteration); the ~NIT*4 staging VGPRs are dead again before the MFMA loop starts.
        constexpr int V = CK / 4;  // float4 per pixel
        constexpr int TOTAL = G::IH * G::IW * V;
        constexpr int NIT = (TOTAL + 255) / 256;
        f32x4 stage[NIT];
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int i = tid + it * 256;
            const int c4 = i % V, pix = i / V;
            const int r = pix / G::IW, ci = pix % G::IW;
            const int iy = iy0 + r, ix = ix0 + ci;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (i < TOTAL && !(p.epilogue & 256) && iy >= 0 && iy < p.Hin && ix >= 0 && ix < p.Win)
                v = *reinterpret_cast<const f32x4 *>(in_b + ((size_t)iy * p.Win + ix) * p.in_cstride + cc * CK + c4 * 4);
            stage[it] = v;
        }
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int i = tid + it * 256;
            const int c4 = i % V, pix = i / V;
            const int r = pix / G::IW, ci = pix % G::IW;
            const int lc = STRIDE == 2 ? (ci & 1) * G::IWH + (ci >> 1) : ci;
            if (i < TOTAL) *reinterpret_cast<f32x4 *>(&lds[(r * G::IWL + lc) * G::PS + c4 * 4]) = stage[it];
        }
        __syncthreads();

        // ---- taps x channel groups ----
        for (int tap = 0; tap < G::TAPS; ++tap) {
            const int ky = tap / KS, kx = tap % KS;
#pragma unroll
            for (int g = 0; g < CK / 8; ++g) {
                const int gabs = cc * (CK / 8) + g;
                f32x4 bfrag[2];
#pragma unroll
                for (int nt = 0; nt < 2; ++nt) bfrag[nt] = wp[(p.epilogue & 512) ? 0 : ((size_t)(gabs * G::TAPS + tap) * 2 + nt) * 64];
                f32x4 afrag[MT];
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
                    const int r = (wave * MT + mt) * STRIDE + ky * DIL;
                    const int lc = STRIDE == 2 ? ((kx * DIL) & 1) * G::IWH + lx + ((kx * DIL) >> 1) : lx + kx * DIL;
                    afrag[mt] = *reinterpret_cast<const f32x4 *>(&lds[(r * G::IWL + lc) * G::PS + g * 8 + half * 4]);
                }
#pragma unroll
                for (int t = 0; t < 4; ++t)
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                        for (int nt = 0; nt < 2; ++nt)
                            acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(afrag[mt][t], bfrag[nt][t], acc[mt][nt], 0, 0, 0);
            }
        }
    }

    // ---- epilogue: D[i][j]: j = lane&31 (channel), i = (r&3) + 8*(r>>2) + 4*(lane>>5) (pixel within the row) ----
    const bool relu = p.epilogue & L3C_EPI_RELU;
    const bool shuffle = p.epilogue & L3C_EPI_PIXEL_SHUFFLE;
    const bool interior = oy0 + G::TH <= p.Hout && ox0 + TW <= p.Wout && chunk_o * 64 + 64 <= p.Cout;
    if (interior && !shuffle) {
        // whole tile inside the image: no per-element bounds checks, 32-bit offsets from block-uniform bases
        float *obase = p.out + (((size_t)b * p.Hout + oy0) * p.Wout + ox0) * p.out_cstride + p.out_coff + chunk_o * 64;
        const float *rbase = p.res ? p.res + (((size_t)b * p.Hout + oy0) * p.Wout + ox0) * p.res_cstride + p.res_coff + chunk_o * 64 : nullptr;
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
            const float bias = p.bias[chunk_o * 64 + nt * 32 + lx];
            float resv[MT][16];
            if (rbase) {   // all residual loads of this half in flight before the first use
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int pix = (wave * MT + mt) * p.Wout + (r & 3) + 8 * (r >> 2) + 4 * half;
                        resv[mt][r] = rbase[pix * p.res_cstride + nt * 32 + lx];
                    }
            }
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int pix = (wave * MT + mt) * p.Wout + (r & 3) + 8 * (r >> 2) + 4 * half;
                    if ((p.epilogue & 1024) && acc[mt][nt][r] != 12345.f) continue;
                    float v = acc[mt][nt][r] + bias;
                    if (relu) v = fmaxf(v, 0.0f);
                    if (rbase) v = v + resv[mt][r];
                    obase[pix * p.out_cstride + nt * 32 + lx] = v;
                }
        }
        return;
    }
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
        const int co = chunk_o * 64 + nt * 32 + lx;
        if (co >= p.Cout) continue;
        const float bias = p.bias[co];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const int oy = oy0 + wave * MT + mt;
            if (oy >= p.Hout) continue;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int ox = ox0 + (r & 3) + 8 * (r >> 2) + 4 * half;
                if (ox >= p.Wout) continue;
                float v = acc[mt][nt][r] + bias;
                if (relu) v = fmaxf(v, 0.0f);
                if (p.res) v = v + p.res[(((size_t)b * p.Hout + oy) * p.Wout + ox) * p.res_cstride + p.res_coff + co];
                if (shuffle) {
                    const size_t oyy = 2 * oy + ((co >> 1) & 1), oxx = 2 * ox + (co & 1);
                    p.out[(((size_t)b * 2 * p.Hout + oyy) * 2 * p.Wout + oxx) * p.out_cstride + p.out_coff + (co >> 2)] = v;
                } else {
                    p.out[(((size_t)b * p.Hout + oy) * p.Wout + ox) * p.out_cstride + p.out_coff + co] = v;
                }
            }
        }
    }
}

// Weight packing: OIHW -> [chunk_o][cin/8][tap][nt][lane][4].
__global__ __launch_bounds__(256) void pack_weights_kernel(const float *__restrict__ w, int Cout, int Cin, int KS,
                                                           float *__restrict__ packed, int64_t total) {
    const int taps = KS * KS;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        int64_t r = i;
        const int t = r % 4;  r /= 4;
        const int lane = r % 64;  r /= 64;
        const int nt = r % 2;  r /= 2;
        const int tap = r % taps;  r /= taps;
        const int g = r % (Cin / 8);  r /= (Cin / 8);
        const int chunk_o = (int)r;
        const int co = chunk_o * 64 + nt * 32 + (lane & 31);
        const int ci = g * 8 + 4 * (lane >> 5) + t;
        packed[i] = co < Cout ? w[((size_t)co * Cin + ci) * taps + tap] : 0.0f;
    }
}

// Plain-VALU statement of the same contract (unpacked OIHW weights): one thread per (pixel, output channel).
__global__ __launch_bounds__(256) void conv_direct_kernel(const ConvParams p, int KS, int stride, int dil) {
    const int64_t total = (int64_t)p.B * p.Hout * p.Wout * p.Cout;
    const bool relu = p.epilogue & L3C_EPI_RELU;
    const bool shuffle = p.epilogue & L3C_EPI_PIXEL_SHUFFLE;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int co = i % p.Cout;
        const int ox = (i / p.Cout) % p.Wout;
        const int oy = (i / p.Cout / p.Wout) % p.Hout;
        const int b = (int)(i / p.Cout / p.Wout / p.Hout);
        float acc = 0.0f;
        for (int ky = 0; ky < KS; ++ky)
            for (int kx = 0; kx < KS; ++kx) {
                const int iy = oy * stride - p.pad + ky * dil, ix = ox * stride - p.pad + kx * dil;
                if (iy < 0 || iy >= p.Hin || ix < 0 || ix >= p.Win) continue;
                const float *px = p.in + (((size_t)b * p.Hin + iy) * p.Win + ix) * p.in_cstride + p.in_coff;
                const float *wk = p.w + (size_t)co * p.Cin * KS * KS + ky * KS + kx;
                for (int ci = 0; ci < p.Cin; ++ci) acc = fmaf(px[ci], wk[(size_t)ci * KS * KS], acc);
            }
        float v = acc + p.bias[co];
        if (relu) v = fmaxf(v, 0.0f);
        if (p.res) v = v + p.res[(((size_t)b * p.Hout + oy) * p.Wout + ox) * p.res_cstride + p.res_coff + co];
        if (shuffle) {
            const size_t oyy = 2 * oy + ((co >> 1) & 1), oxx = 2 * ox + (co & 1);
            p.out[(((size_t)b * 2 * p.Hout + oyy) * 2 * p.Wout + oxx) * p.out_cstride + p.out_coff + (co >> 2)] = v;
        } else {
            p.out[(((size_t)b * p.Hout + oy) * p.Wout + ox) * p.out_cstride + p.out_coff + co] = v;
        }
    }
}

int fill_params(const l3c_conv_desc *d, ConvParams &p) {
    L3C_REQUIRE(d, "null descriptor");
    L3C_REQUIRE(d->in && d->packed_w && d->bias && d->out, "null pointer in descriptor");
    L3C_REQUIRE(d->B > 0 && d->Hin > 0 && d->Win > 0 && d->Cin > 0 && d->Cout > 0, "bad shape");
    L3C_REQUIRE(d->KS == 1 || d->KS == 3 || d->KS == 5, "KS must be 1, 3 or 5");
    L3C_REQUIRE(d->stride == 1 || (d->stride == 2 && d->KS == 5), "stride 2 is only provided for the 5x5 down conv");
    L3C_REQUIRE(d->dilation == 1 || (d->KS == 3 && (d->dilation == 2 || d->dilation == 4)), "dilation must be 1, 2 or 4 (3x3)");
    L3C_REQUIRE(d->in_cstride % 4 == 0 && d->in_coff % 4 == 0, "input channel stride/offset must be multiples of 4");
    L3C_REQUIRE(d->in_coff + d->Cin <= d->in_cstride, "input channel slice out of range");
    L3C_REQUIRE(!(d->epilogue & L3C_EPI_RESIDUAL) || d->residual, "residual epilogue without residual pointer");
    L3C_REQUIRE(!((d->epilogue & L3C_EPI_PIXEL_SHUFFLE) && (d->epilogue & L3C_EPI_RESIDUAL)), "pixel shuffle + residual not provided");
    L3C_REQUIRE(!(d->epilogue & L3C_EPI_PIXEL_SHUFFLE) || d->Cout % 4 == 0, "pixel shuffle needs Cout % 4 == 0");
    p.in = d->in;  p.w = d->packed_w;  p.bias = d->bias;
    p.res = (d->epilogue & L3C_EPI_RESIDUAL) ? d->residual : nullptr;
    p.out = d->out;
    p.in_cstride = d->in_cstride;  p.in_coff = d->in_coff;
    p.res_cstride = d->res_cstride;  p.res_coff = d->res_coff;
    p.out_cstride = d->out_cstride;  p.out_coff = d->out_coff;
    p.B = d->B;  p.Hin = d->Hin;  p.Win = d->Win;  p.Cin = d->Cin;  p.Cout = d->Cout;
    p.pad = d->dilation == 1 ? d->KS / 2 : d->dilation;
    const int ext = (d->KS - 1) * d->dilation + 1;
    p.Hout = (d->Hin + 2 * p.pad - ext) / d->stride + 1;
    p.Wout = (d->Win + 2 * p.pad - ext) / d->stride + 1;
    L3C_REQUIRE(p.Hout > 0 && p.Wout > 0, "empty output");
    p.epilogue = d->epilogue;
    return L3C_OK;
}

template <int KS, int STRIDE, int DIL, int CK, int MT>
int launch(ConvParams &p, hipStream_t stream) {
    using G = Geo<KS, STRIDE, DIL, CK, MT>;
    static_assert(G::LDS_BYTES <= 64 * 1024, "LDS tile too large");
    L3C_REQUIRE(p.Cin % CK == 0, "Cin must be a multiple of the channel chunk");
    p.tiles_x = (p.Wout + TW - 1) / TW;
    p.tiles_y = (p.Hout + G::TH - 1) / G::TH;
    p.n_chunks_o = (p.Cout + 63) / 64;
    const int64_t total = (int64_t)p.tiles_x * p.tiles_y * p.n_chunks_o * p.B;
    L3C_REQUIRE(total < (1ll << 31), "grid too large");
    p.total_blocks = (int)total;
    hipLaunchKernelGGL((conv_mfma_kernel<KS, STRIDE, DIL, CK, MT>), dim3((unsigned)total), dim3(256), G::LDS_BYTES, stream, p);
    return l3c::check_launch("conv_mfma_kernel");
}

}  // namespace

extern "C" {

int64_t l3c_conv_packed_words(int Cout, int Cin, int KS) {
    return (int64_t)((Cout + 63) / 64) * 64 * Cin * KS * KS;
}

int l3c_conv_pack_weights(const float *w_oihw, int Cout, int Cin, int KS, float *packed, l3c_stream_t stream) {
    L3C_REQUIRE(w_oihw && packed, "null pointer");
    L3C_REQUIRE(Cout > 0 && Cin > 0 && Cin % 8 == 0, "Cin must be a multiple of 8");
    L3C_REQUIRE(KS == 1 || KS == 3 || KS == 5, "KS must be 1, 3 or 5");
    const int64_t total = l3c_conv_packed_words(Cout, Cin, KS);
    int64_t g = (total + 255) / 256;
    hipLaunchKernelGGL(pack_weights_kernel, dim3((unsigned)(g > 4096 ? 4096 : g)), dim3(256), 0, l3c::as_stream(stream),
                       w_oihw, Cout, Cin, KS, packed, total);
    return l3c::check_launch("pack_weights_kernel");
}

int l3c_conv_mfma(const l3c_conv_desc *d, l3c_stream_t stream) {
    ConvParams p;
    const int rc = fill_params(d, p);
    if (rc != L3C_OK) return rc;
    L3C_REQUIRE(d->Cin % 16 == 0, "Cin must be a multiple of 16");
    hipStream_t s = l3c::as_stream(stream);
    if (d->KS == 3 && d->dilation == 1) return (d->epilogue & 2048) ? launch<3, 1, 1, 32, 2>(p, s) : launch<3, 1, 1, 16, 2>(p, s);
    if (d->KS == 3 && d->dilation == 2) return launch<3, 1, 2, 16, 2>(p, s);
    if (d->KS == 3 && d->dilation == 4) return launch<3, 1, 4, 16, 2>(p, s);
    if (d->KS == 5) return launch<5, 2, 1, 16, 1>(p, s);
    return launch<1, 1, 1, 32, 2>(p, s);
}

int l3c_conv_direct(const l3c_conv_desc *d, l3c_stream_t stream) {
    ConvParams p;
    const int rc = fill_params(d, p);
    if (rc != L3C_OK) return rc;
    const int64_t total = (int64_t)p.B * p.Hout * p.Wout * p.Cout;
    int64_t g = (total + 255) / 256;
    hipLaunchKernelGGL(conv_direct_kernel, dim3((unsigned)(g > 65536 ? 65536 : g)), dim3(256), 0, l3c::as_stream(stream), p,
                       d->KS, d->stride, d->dilation);
    return l3c::check_launch("conv_direct_kernel");
}
}
