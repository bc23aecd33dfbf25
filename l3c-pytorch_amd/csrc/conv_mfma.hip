// conv_mfma.hip -- fp32 implicit-GEMM convolution on v_mfma_f32_32x32x2_f32 (gfx950), plus a plain-VALU cross-check.
//
// Two kernels: conv_lds_kernel ("v3", all 3x3 layers: both MFMA operands from LDS, weights by LDS-DMA) further down, and
// conv_mfma_kernel ("v1", 5x5 stride 2 and 1x1: weights per wave from L2) described first.  Only a -DL3C_DEV_PROBES build
// (csrc/build.py --dev-probes, tools/conv_probe.py) honours the flags 0x100..0x800 in `epilogue`, which select development-probe
// instantiations (parts of the kernel removed: WRONG results) / the v1 kernel for A/B measurements; the product library
// rejects every epilogue bit it does not document.
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
#include <stdlib.h>

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
    int total_blocks;      // number of work items (tile x output-channel chunk x image)
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

// DBG (development probes only, never dispatched in production): 1 = no weight loads, 2 = also no LDS reads,
// 3 = also no staging and no stores (MFMA + barriers only)
template <int KS, int STRIDE, int DIL, int CK, int MT, int DBG = 0>
__global__ __launch_bounds__(256, (MT >= 4 ? 2 : 3)) void conv_mfma_kernel(const ConvParams p) {
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

    // Software pipeline (explicit: hipcc schedules loads only a few MFMAs ahead of their use otherwise).  A "step" is one
    // (tap, 8-channel group) = 16 MFMAs per wave.  Weight fragments are fetched PFB steps ahead into a register ring that
    // keeps running across chunk boundaries (weights do not depend on LDS, so they also cover the staging round trip);
    // A fragments are read from LDS one step ahead.  sched_barrier keeps the compiler from sinking the loads back.
    constexpr int GPC = CK / 8;                       // channel groups per chunk
    constexpr int NSTEPS = G::TAPS * GPC;
    constexpr int R = (NSTEPS % 3 == 0) ? 3 : 2;      // ring size, divides NSTEPS so ring slots line up across chunks
    constexpr int PFB = R - 1;
    const int total_steps = n_cc * NSTEPS;
    f32x4 bq[R][2];
    f32x4 aq[2][MT];
    auto load_b = [&](int cc_, int st, f32x4 (&dst)[2]) {   // st is a compile-time step index at every call site
        const int tap = st / GPC, g = st % GPC;
        const f32x4 *wc = wp + (size_t)cc_ * GPC * G::TAPS * 2 * 64;
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
            if (DBG >= 1) dst[nt] = f32x4{1.f, 2.f, 3.f, (float)st};
            else dst[nt] = wc[((g * G::TAPS + tap) * 2 + nt) * 64];
        }
    };
    auto load_a = [&](int st, f32x4 (&dst)[MT]) {
        const int tap = st / GPC, g = st % GPC;
        const int ky = tap / KS, kx = tap % KS;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const int r = (wave * MT + mt) * STRIDE + ky * DIL;
            const int lc = STRIDE == 2 ? ((kx * DIL) & 1) * G::IWH + lx + ((kx * DIL) >> 1) : lx + kx * DIL;
            if (DBG >= 2) dst[mt] = f32x4{(float)lx, 1.f, (float)st, 2.f};
            else dst[mt] = *reinterpret_cast<const f32x4 *>(&lds[(r * G::IWL + lc) * G::PS + g * 8 + half * 4]);
        }
    };
#pragma unroll
    for (int st = 0; st < PFB; ++st) load_b(0, st, bq[st]);

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
            if (DBG < 3 && i < TOTAL && iy >= 0 && iy < p.Hin && ix >= 0 && ix < p.Win)
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
        load_a(0, aq[0]);
#pragma unroll
        for (int st = 0; st < NSTEPS; ++st) {
            if (st + PFB < NSTEPS) {
                load_b(cc, st + PFB, bq[(st + PFB) % R]);
            } else if (cc + 1 < n_cc) {
                load_b(cc + 1, st + PFB - NSTEPS, bq[(st + PFB) % R]);
            }
            if (st + 1 < NSTEPS) load_a(st + 1, aq[(st + 1) & 1]);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                    for (int nt = 0; nt < 2; ++nt)
                        acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(aq[st & 1][mt][t], bq[st % R][nt][t], acc[mt][nt], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    (void)total_steps;

    // ---- epilogue: D[i][j]: j = lane&31 (channel), i = (r&3) + 8*(r>>2) + 4*(lane>>5) (pixel within the row) ----
    const bool relu = p.epilogue & L3C_EPI_RELU;
    const bool shuffle = p.epilogue & L3C_EPI_PIXEL_SHUFFLE;
    const bool interior = oy0 + G::TH <= p.Hout && ox0 + TW <= p.Wout;
    if (interior && !shuffle) {
        // whole tile inside the image: no per-element bounds checks, 32-bit offsets from block-uniform bases; a partial last
        // output-channel chunk (Kp = 120, 150) only needs a per-lane channel mask
        float *obase = p.out + (((size_t)b * p.Hout + oy0) * p.Wout + ox0) * p.out_cstride + p.out_coff + chunk_o * 64;
        const float *rbase = p.res ? p.res + (((size_t)b * p.Hout + oy0) * p.Wout + ox0) * p.res_cstride + p.res_coff + chunk_o * 64 : nullptr;
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
            if (chunk_o * 64 + nt * 32 + lx >= p.Cout) continue;
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
                    float v = acc[mt][nt][r] + bias;
                    if (relu) v = fmaxf(v, 0.0f);
                    if (rbase) v = v + resv[mt][r];
                    if (DBG < 3 || v == 12345.678f) obase[pix * p.out_cstride + nt * 32 + lx] = v;
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

// ---- v3: weights through LDS ---------------------------------------------------------------------------------------------
// Measured on the v1 structure (development probes): MFMA + barriers alone run at 146.7 TFLOP/s, adding the LDS A-fragment
// reads costs 5 %, adding the per-wave GLOBAL weight-fragment loads costs another 20 % -- every wave of every block streams
// the same 1 KB weight fragments through the vector memory path.  v3 therefore feeds BOTH operands from LDS:
//   * block = 512 threads = 8 waves on a 16 x 32 output tile (wave w: rows 2w, 2w+1) x 64 output channels, one block per CU
//     (2 waves per SIMD); the 8 waves share one copy of the weights;
//   * per 16-channel chunk the weight slab (taps x 2 groups x 2 N-tiles x 1 KB, contiguous in the packed layout) is copied
//     by LDS-DMA (`global_load_lds_dwordx4`, lane-linear = exactly the fragment layout) into one of two LDS slabs, one chunk
//     ahead; the input patch of the next chunk is prefetched into registers during the MFMA loop and written to the
//     (single, padded) LDS patch between two barriers;
//   * the MFMA loop contains no vector-memory instruction at all: ds_read_b128 with immediate offsets, one step ahead.
// CK_ = input channels per chunk; WN = waves along N: the 8 waves are WM x WN, each wave MT=2 rows x 64 output channels, so a
// block covers a (WM*2) x 32 tile and WN consecutive 64-channel output chunks.  3x3: CK 16, WN 1 (16 x 32 tile).
// 1x1 (192 -> 120/150): CK 64, WN 2 (8 x 32 tile, both output chunks of Kp = 120 in one block: the patch is staged once).
// NW_ = waves per block (8, or 4: two blocks then share a CU and cover each other's barrier / DMA stalls), SLABS_ = weight slab
// buffers (2 = the next chunk's slab is DMA'd during the MFMA loop; 1 = fetched between the two barriers, for the NW_ = 4
// layout whose LDS budget is 80 KB per block).
template <int KS, int DIL, int CK_ = 16, int WN_ = 1, int NW_ = 8, int SLABS_ = 2>
struct Geo3 {
    static constexpr int NW = NW_, SLABS = SLABS_, THREADS = NW_ * 64;
    static constexpr int MT = 2, CK = CK_, GPC = CK_ / 8, WN = WN_, WM = NW_ / WN_;
    static constexpr int TH = WM * MT;
    static constexpr int IH = TH + (KS - 1) * DIL, IW = TW + (KS - 1) * DIL;
    static constexpr int PS = CK + 4;
    static constexpr int TAPS = KS * KS;
    static constexpr int A_FLOATS = IH * IW * PS;
    static constexpr int BW_FLOATS = TAPS * GPC * 2 * 256;          // one chunk's weight slab for ONE 64-channel output chunk
    static constexpr int B_FLOATS = WN * BW_FLOATS;
    static constexpr int LDS_BYTES = (A_FLOATS + SLABS * B_FLOATS) * 4;
};

template <int KS, int DIL, int CK_, int WN_, int NW_, int SLABS_>
__global__ __launch_bounds__(NW_ * 64, 2) void conv_lds_kernel(const ConvParams p) {
    using G = Geo3<KS, DIL, CK_, WN_, NW_, SLABS_>;
    constexpr int NT_ = G::THREADS;
    constexpr int MT = G::MT, CK = G::CK, GPC = G::GPC, NSTEPS = G::TAPS * GPC;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float *lds_a = lds;
    float *lds_b = lds + G::A_FLOATS;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int half = lane >> 5, lx = lane & 31;
    const int wm = wave / G::WN, wn = wave % G::WN;   // wave -> (row pair, 64-channel output chunk)
    const int n_cc = p.Cin / CK;
    const int groups_total = p.Cin / 8;
    const int tiles = p.tiles_x * p.tiles_y;

    // persistent: a contiguous range of work items (tile x 64 output channels x image) per block, neighbouring ranges on
    // one XCD; the block walks a flat sequence of stages (item, chunk) and always fetches one stage ahead, also across items
    const int nb = gridDim.x;
    const int kb = xcd_remap(blockIdx.x, nb);
    const int qi = p.total_blocks / nb, ri = p.total_blocks % nb;
    const int first = kb * qi + (kb < ri ? kb : ri);
    const int count = qi + (kb < ri ? 1 : 0);
    const int n_stages = count * n_cc;

    constexpr int NB_DMA = G::B_FLOATS / 256, NBW = G::BW_FLOATS / 256;
    const int n_groups_o = (p.n_chunks_o + G::WN - 1) / G::WN;
    constexpr int V = CK / 4;
    constexpr int TOTAL = G::IH * G::IW * V;
    constexpr int NIT = (TOTAL + NT_ - 1) / NT_;
    f32x4 stage_regs[NIT];
    // stage s: weight slab -> LDS slab (s % SLABS) by DMA ...
    auto fetch_slab = [&](int s_) {
        const int item = first + s_ / n_cc, cc = s_ % n_cc;
        const int group_o = (item / tiles) % n_groups_o;
        float *dst = lds_b + (s_ % G::SLABS) * G::B_FLOATS;
        for (int j = wave; j < NB_DMA; j += G::NW) {
            // slab part of output chunk group_o*WN + j / NBW (clamped: a partial last group re-reads the last chunk)
            int co = group_o * G::WN + j / NBW;
            co = co < p.n_chunks_o ? co : p.n_chunks_o - 1;
            const float *src = p.w + ((size_t)co * groups_total * G::TAPS * 2 + (size_t)cc * GPC * G::TAPS * 2) * 256 + (j % NBW) * 256 + lane * 4;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                             (__attribute__((address_space(3))) void *)(dst + j * 256), 16, 0, 0);
        }
    };
    // ... and its input patch -> registers
    auto fetch_patch = [&](int s_) {
        const int item = first + s_ / n_cc, cc = s_ % n_cc;
        const int tile = item % tiles, b = item / tiles / n_groups_o;
        const int iy0 = (tile / p.tiles_x) * G::TH - p.pad, ix0 = (tile % p.tiles_x) * TW - p.pad;
        const float *in_b = p.in + (size_t)b * p.Hin * p.Win * p.in_cstride + p.in_coff + cc * CK;
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int i = tid + it * NT_;
            const int c4 = i % V, pix = i / V;
            const int r = pix / G::IW, ci = pix % G::IW;
            const int iy = iy0 + r, ix = ix0 + ci;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (i < TOTAL && iy >= 0 && iy < p.Hin && ix >= 0 && ix < p.Win)
                v = *reinterpret_cast<const f32x4 *>(in_b + ((size_t)iy * p.Win + ix) * p.in_cstride + c4 * 4);
            stage_regs[it] = v;
        }
    };
    auto store_patch = [&]() {
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int i = tid + it * NT_;
            const int c4 = i % V, pix = i / V;
            if (i < TOTAL) *reinterpret_cast<f32x4 *>(&lds_a[pix * G::PS + c4 * 4]) = stage_regs[it];
        }
    };

    if (n_stages > 0) {
        if (G::SLABS == 2) fetch_slab(0);
        fetch_patch(0);
    }
    const float *a_lane = lds_a + ((wm * MT) * G::IW + lx) * G::PS + half * 4;
    int stage = 0;
    for (int it_ = 0; it_ < count; ++it_) {
        const int item = first + it_;
        f32x16 acc[MT][2];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[mt][nt][r] = 0.0f;

        for (int cc = 0; cc < n_cc; ++cc, ++stage) {
            if (stage) __syncthreads();                        // everyone is done reading the patch and slab (stage+1)&1
            if (G::SLABS == 1) fetch_slab(stage);              // single slab: only now free (the co-resident block computes meanwhile)
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this thread's patch registers and slab DMA have landed
            store_patch();
            __syncthreads();                                   // patch + slab of this stage visible to the whole block
            if (stage + 1 < n_stages) {
                if (G::SLABS == 2) fetch_slab(stage + 1);
                fetch_patch(stage + 1);
            }
            const float *b_lane = lds_b + (stage % G::SLABS) * G::B_FLOATS + wn * G::BW_FLOATS + lane * 4;
            f32x4 aq[2][MT], bq[2][2];
            auto load_ab = [&](int st, f32x4 (&a)[MT], f32x4 (&bb)[2]) {
                const int tap = st / GPC, g = st % GPC;
                const int ky = tap / KS, kx = tap % KS;
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
                    a[mt] = *reinterpret_cast<const f32x4 *>(a_lane + ((mt + ky * DIL) * G::IW + kx * DIL) * G::PS + g * 8);
#pragma unroll
                for (int nt = 0; nt < 2; ++nt)
                    bb[nt] = *reinterpret_cast<const f32x4 *>(b_lane + ((g * G::TAPS + tap) * 2 + nt) * 256);
            };
            load_ab(0, aq[0], bq[0]);
#pragma unroll
            for (int st = 0; st < NSTEPS; ++st) {
                if (st + 1 < NSTEPS) load_ab(st + 1, aq[(st + 1) & 1], bq[(st + 1) & 1]);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int t = 0; t < 4; ++t)
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                        for (int nt = 0; nt < 2; ++nt)
                            acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(aq[st & 1][mt][t], bq[st & 1][nt][t], acc[mt][nt], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        }

        // ---- epilogue of this item; its stores overlap the next item's first fetch (already in flight) ----
        const int tile = item % tiles, chunk_o = ((item / tiles) % n_groups_o) * G::WN + wn, b = item / tiles / n_groups_o;
        const int oy0 = (tile / p.tiles_x) * G::TH, ox0 = (tile % p.tiles_x) * TW;
        if (chunk_o >= p.n_chunks_o) continue;   // wave-uniform: this wave's output chunk does not exist (partial last group)
        // launder the lane coordinates once per item: otherwise the ~128 loop-invariant per-lane store addresses are
        // hoisted out of the item loop, live across the MFMA loop and spill
        int e_lx = lx, e_half = half, e_wave = wm;
        asm volatile("" : "+v"(e_lx), "+v"(e_half), "+v"(e_wave));
        const bool relu = p.epilogue & L3C_EPI_RELU;
        const bool shuffle = p.epilogue & L3C_EPI_PIXEL_SHUFFLE;
        const bool interior = oy0 + G::TH <= p.Hout && ox0 + TW <= p.Wout;
        if (interior && !shuffle) {
            float *obase = p.out + (((size_t)b * p.Hout + oy0) * p.Wout + ox0) * p.out_cstride + p.out_coff + chunk_o * 64;
            const float *rbase = p.res ? p.res + (((size_t)b * p.Hout + oy0) * p.Wout + ox0) * p.res_cstride + p.res_coff + chunk_o * 64 : nullptr;
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) {
                if (chunk_o * 64 + nt * 32 + e_lx >= p.Cout) continue;
                const float bias = p.bias[chunk_o * 64 + nt * 32 + e_lx];
                float resv[MT][16];
                if (rbase) {
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            const int pix = (e_wave * MT + mt) * p.Wout + (r & 3) + 8 * (r >> 2) + 4 * e_half;
                            resv[mt][r] = rbase[pix * p.res_cstride + nt * 32 + e_lx];
                        }
                }
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int pix = (e_wave * MT + mt) * p.Wout + (r & 3) + 8 * (r >> 2) + 4 * e_half;
                        float v = acc[mt][nt][r] + bias;
                        if (relu) v = fmaxf(v, 0.0f);
                        if (rbase) v = v + resv[mt][r];
                        obase[pix * p.out_cstride + nt * 32 + e_lx] = v;
                    }
            }
        } else {
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) {
                const int co = chunk_o * 64 + nt * 32 + e_lx;
                if (co >= p.Cout) continue;
                const float bias = p.bias[co];
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
                    const int oy = oy0 + e_wave * MT + mt;
                    if (oy >= p.Hout) continue;
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int ox = ox0 + (r & 3) + 8 * (r >> 2) + 4 * e_half;
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

template <int KS, int STRIDE, int DIL, int CK, int MT, int DBG = 0>
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
    hipLaunchKernelGGL((conv_mfma_kernel<KS, STRIDE, DIL, CK, MT, DBG>), dim3((unsigned)total), dim3(256), G::LDS_BYTES, stream, p);
    return l3c::check_launch("conv_mfma_kernel");
}

int num_cus() {
    static int n = 0;
    if (!n) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) n = prop.multiProcessorCount;
        if (n <= 0) n = 256;
    }
    return n;
}

template <int KS, int DIL, int CK_ = 16, int WN_ = 1, int NW_ = 8, int SLABS_ = 2>
int launch_lds(ConvParams &p, hipStream_t stream) {
    using G = Geo3<KS, DIL, CK_, WN_, NW_, SLABS_>;
    static_assert(G::LDS_BYTES <= 160 * 1024, "LDS budget exceeded");
    L3C_REQUIRE(p.Cin % G::CK == 0, "Cin must be a multiple of the channel chunk");
    p.tiles_x = (p.Wout + TW - 1) / TW;
    p.tiles_y = (p.Hout + G::TH - 1) / G::TH;
    p.n_chunks_o = (p.Cout + 63) / 64;
    const int64_t total = (int64_t)p.tiles_x * p.tiles_y * ((p.n_chunks_o + G::WN - 1) / G::WN) * p.B;
    L3C_REQUIRE(total < (1ll << 31), "grid too large");
    p.total_blocks = (int)total;
    static bool attr_set[64] = {};   // > 64 KB of dynamic LDS needs the opt-in, once per device (and per instantiation: static)
    int dev = 0;
    {
        const int rc = l3c::check_hip(hipGetDevice(&dev), "hipGetDevice");
        if (rc != L3C_OK) return rc;
    }
    L3C_REQUIRE(dev >= 0 && dev < 64, "device index out of range");
    if (!attr_set[dev]) {
        const int rc = l3c::check_hip(hipFuncSetAttribute(reinterpret_cast<const void *>(&conv_lds_kernel<KS, DIL, CK_, WN_, NW_, SLABS_>),
                                                          hipFuncAttributeMaxDynamicSharedMemorySize, G::LDS_BYTES),
                                      "hipFuncSetAttribute");
        if (rc != L3C_OK) return rc;
        attr_set[dev] = true;
    }
    // Items per block: the kernel walks a contiguous range of items with cross-item prefetch (the next tile's first stage is
    // fetched during the current tile's last chunk, its epilogue stores overlap that fetch).  A fully persistent grid
    // (one block per CU, static partition) loses more to interference from the concurrently running coder waves than it
    // gains; a few items per block keeps the hardware dispatcher's dynamic balancing.  L3C_CONV_ITEMS_PER_BLOCK overrides.
    static const int items_per_block = getenv("L3C_CONV_ITEMS_PER_BLOCK") ? atoi(getenv("L3C_CONV_ITEMS_PER_BLOCK")) : 1;
    const int ipb = items_per_block > 0 ? items_per_block : 1;
    const unsigned grid = (unsigned)((total + ipb - 1) / ipb);
    hipLaunchKernelGGL((conv_lds_kernel<KS, DIL, CK_, WN_, NW_, SLABS_>), dim3(grid), dim3(G::THREADS), G::LDS_BYTES, stream, p);
    return l3c::check_launch("conv_lds_kernel");
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
#ifdef L3C_DEV_PROBES
    const bool v1 = d->epilogue & 2048;
    if (d->KS == 3 && d->dilation == 1 && (d->epilogue & 0x300)) {
        const int dbg = (d->epilogue >> 8) & 3;
        return dbg == 1 ? launch<3, 1, 1, 16, 2, 1>(p, s) : dbg == 2 ? launch<3, 1, 1, 16, 2, 2>(p, s) : launch<3, 1, 1, 16, 2, 3>(p, s);
    }
    if (d->KS == 3 && v1) return d->dilation == 1 ? launch<3, 1, 1, 16, 2>(p, s) : d->dilation == 2 ? launch<3, 1, 2, 16, 2>(p, s) : launch<3, 1, 4, 16, 2>(p, s);
#else
    const bool v1 = false;
    L3C_REQUIRE((d->epilogue & ~(L3C_EPI_RELU | L3C_EPI_RESIDUAL | L3C_EPI_PIXEL_SHUFFLE)) == 0, "unknown epilogue bits");
#endif
    if (d->KS == 3 && d->dilation == 1) return launch_lds<3, 1>(p, s);
    if (d->KS == 3 && d->dilation == 2) return launch_lds<3, 2>(p, s);
    if (d->KS == 3 && d->dilation == 4) return launch_lds<3, 4>(p, s);
    if (d->KS == 5) return launch<5, 2, 1, 16, 1>(p, s);
    // the 1x1 192->Kp layer: v3 with both output chunks of Kp = 120 in one block (patch staged once) measures 83 TFLOP/s vs
    // 78 for v1; with an odd number of chunks (Kp = 150) half a block would idle and v1 wins (64 vs 58)
    if (!v1 && d->Cin % 64 == 0 && ((d->Cout + 63) / 64) % 2 == 0) return launch_lds<1, 1, 64, 2>(p, s);
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
