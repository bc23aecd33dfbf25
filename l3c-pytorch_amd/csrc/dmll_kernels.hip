// dmll_kernels.hip -- discretised logistic-mixture head on the GPU.
//
// Replaces, on the coding path: criterion/logistic_mixture.py (_extract_non_shared_c :248-275, cdf_step_non_shared
// :134-141, forward :146-207) and torchac_kernel.cu (calculate_cdf_kernel :26-76 == torchac.py:_get_uint16_cdf :174-213).
//
// Bit-reproducibility contract: the encoder never materialises a table (l3c_dmll_encode_intervals evaluates only the two
// entries per symbol that the coder reads) while the decoder does (l3c_dmll_channel_params -> l3c_cdf_table_mixture ->
// l3c_ac_decode).  Both go through the SAME device functions below (mix_stats / mix_component / cdf_entry), every
// floating-point operation is an individually rounded fp32 op in a fixed order (this file is compiled with
// -ffp-contract=off, K is walked sequentially, no atomics), so the two paths produce identical table entries.
//
// Arithmetic follows the torch-CPU statement of the reference kernel (which is what the oracle pins):
//   pi  = exp(l - max) / sum_k exp(l - max)            F.softmax(dim=K)
//   mu' = mu (+ sigmoid(lam) * x_prev ...)              logistic_mixture.py:262-272
//   ls  = max(log_sigma, -7)                            :260
//   cdf(l) = sum_k pi_k * sigmoid((t_l - mu'_k) * exp(-ls_k))   sequential k, mul then add   (torchac.py:181-200)
//   entry  = uint16(rint(cdf * (65536 - (Lp-1))) + l)            round-half-even, wraps mod 2^16 (torchac.py:203-213)
#include "ac_core.h"
#include "dmll_core.h"
#include "l3c_common.h"

namespace {

using l3c::sigmoid_f;
using l3c::sigmoid_sat;   // csrc/dmll_core.h, with the mixture functions below
using l3c::kLogScalesMin;
using l3c::MixStats;
using l3c::MixComponent;
using l3c::mix_stats;
using l3c::mix_component;
using l3c::mix_component_e;
using l3c::cdf_term;
using l3c::cdf_quantise;

// ---- P tile -> LDS ------------------------------------------------------------------------------------------------------
// npix pixel rows of Kp floats (contiguous in memory) into tile[pixel][ld = Kp + 1] (the odd stride keeps the per-pixel column reads of
// the compute phases conflict-free).  Kp % 4 == 0 (the RGB scale: 120) and a 16-byte aligned source: one 16-byte load per four values and
// one division per load -- by multiplication with a host-made reciprocal (exact for the < 2^16 indices of a tile) -- instead of a
// 4-byte load and an integer division per value (round 5: the divisions were a quarter of the interval kernel's VALU work).
struct TileDiv {
    unsigned q4, magic;   // q4 = Kp / 4 (0: scalar fill); i / q4 == (i * magic) >> 20 for i < 65536 / ... (checked by the host)
};
static TileDiv tile_div(int Kp, int max_pix) {
    TileDiv d{0u, 0u};
#ifdef L3C_IV_SCALAR_FILL
    return d;       // (A/B builds: the scalar fill of rounds 1-4)
#endif
    if (Kp % 4 != 0) return d;
    const unsigned q = (unsigned)(Kp / 4);
    const unsigned magic = ((1u << 20) + q - 1u) / q;
    for (unsigned i = 0; i < (unsigned)max_pix * q; ++i)
        if (i / q != (unsigned)(((unsigned long long)i * magic) >> 20)) return d;   // (never for the shapes of this path; falls back to scalar)
    d.q4 = q;
    d.magic = magic;
    return d;
}
// NT threads (a compile-time constant) fill; the loads of a batch of U turns are all issued before the first value is stored, so that the
// batch pays one memory round trip (with a run-time stride the compiler keeps one load in flight per turn: the 150-channel tiles of
// the bottleneck scales then took 30 round trips, twice the time of the whole kernel before).
template <int NT>
__device__ __forceinline__ void fill_tile(float *__restrict__ tile, const float *__restrict__ src, int npix, int Kp, int ld, TileDiv dv, int tid) {
    if (dv.q4 && (reinterpret_cast<uintptr_t>(src) & 15) == 0) {
        constexpr int U = 4;
        const float4 *src4 = reinterpret_cast<const float4 *>(src);
        const int n4 = npix * (int)dv.q4;
        for (int i0 = tid; i0 < n4; i0 += NT * U) {
            float4 v[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int i = i0 + u * NT;
                v[u] = src4[i < n4 ? i : i0];
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int i = i0 + u * NT;
                if (i < n4) {
                    const int p = (int)(((unsigned)i * dv.magic) >> 20);
                    float *dst = tile + p * ld + (i - p * (int)dv.q4) * 4;
                    dst[0] = v[u].x;
                    dst[1] = v[u].y;
                    dst[2] = v[u].z;
                    dst[3] = v[u].w;
                }
            }
        }
    } else {
        constexpr int U = 8;
        const int n = npix * Kp;
        for (int i0 = tid; i0 < n; i0 += NT * U) {
            float v[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int i = i0 + u * NT;
                v[u] = src[i < n ? i : i0];
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int i = i0 + u * NT;
                if (i < n) tile[(i / Kp) * ld + (i % Kp)] = v[u];
            }
        }
    }
}

// ---- per-channel parameters (CDFOut) ---------------------------------------------------------------------------------

__global__ __launch_bounds__(256) void channel_params_kernel(const float *__restrict__ P, const int16_t *__restrict__ sym,
                                                             int64_t B, int64_t HW, int C, int K, int rgb, int c,
                                                             float *__restrict__ pi, float *__restrict__ mu,
                                                             float *__restrict__ log_sigma) {
    const int Kp = (rgb ? 4 : 3) * C * K;
    const int64_t total = B * HW;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t b = i / HW, n = i % HW;
        const float *px = P + i * Kp;
        auto get = [&](int ch) { return px[ch]; };
        float x0 = 0.f, x1 = 0.f;
        if (rgb && c > 0) {
            x0 = (float)sym[(b * C + 0) * HW + n];
            if (c > 1) x1 = (float)sym[(b * C + 1) * HW + n];
        }
        const MixStats st = mix_stats(get, C, K, c);
        for (int k = 0; k < K; ++k) {
            const MixComponent m = mix_component(get, st, C, K, rgb, c, k, x0, x1);
            const int64_t o = (b * K + k) * HW + n;
            pi[o] = m.pi;
            mu[o] = m.mu;
            log_sigma[o] = m.log_sigma;
        }
    }
}

// ---- full uint16 table (decoder side; the reference's calculate_cdf_kernel) -------------------------------------------

constexpr int kTablePix = 32;  // pixels per block
constexpr int kMaxK = 16;

__global__ __launch_bounds__(256) void cdf_table_kernel(const float *__restrict__ targets, const float *__restrict__ pi,
                                                        const float *__restrict__ mu, const float *__restrict__ ls,
                                                        int64_t HW, int K, int Lp, uint16_t *__restrict__ cdf) {
    __shared__ float s_pi[kTablePix][kMaxK], s_mu[kTablePix][kMaxK], s_inv[kTablePix][kMaxK];
    __shared__ float s_t[260];
    const int64_t img = blockIdx.y;
    const int64_t pix0 = (int64_t)blockIdx.x * kTablePix;
    const int npix = (int)((HW - pix0) < kTablePix ? (HW - pix0) : kTablePix);
    const int tid = threadIdx.x;
    for (int i = tid; i < kTablePix * K; i += 256) {
        const int p = i % kTablePix, k = i / kTablePix;
        if (p < npix) {
            const int64_t o = (img * K + k) * HW + pix0 + p;
            s_pi[p][k] = pi[o];
            s_mu[p][k] = mu[o];
            s_inv[p][k] = expf(-ls[o]);
        }
    }
    for (int i = tid; i < Lp; i += 256) s_t[i] = targets[i];
    __syncthreads();
    const float scale = (float)(65536 - (Lp - 1));
    const int count = npix * Lp;
    uint16_t *out = cdf + (img * HW + pix0) * Lp;
    const bool aligned4 = ((reinterpret_cast<uintptr_t>(out) & 3) == 0);
    for (int e = tid * 2; e < count; e += 512) {
        uint32_t v[2];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int ee = e + h < count ? e + h : e;
            const int p = ee / Lp, l = ee - p * Lp;
            const float t = s_t[l];
            float acc = 0.0f;
            for (int k = 0; k < K; ++k) acc = acc + cdf_term(s_pi[p][k], s_mu[p][k], s_inv[p][k], t);
            v[h] = cdf_quantise(acc, scale, l);
        }
        if (aligned4 && e + 1 < count) {
            *reinterpret_cast<uint32_t *>(out + e) = v[0] | (v[1] << 16);
        } else {
            out[e] = (uint16_t)v[0];
            if (e + 1 < count) out[e + 1] = (uint16_t)v[1];
        }
    }
}

// ---- decoder, fused: P (+ the channels decoded so far) -> uint16 table rows of ONE channel over a pixel range -----------
// channel_params_kernel + cdf_table_kernel in one pass, without the (B,K,H,W) x 3 parameter arrays in between, for the
// chunk-pipelined RGB decode (R, G and B of an image are decoded a chunk apart, each chunk's table built just in time from
// the symbols the previous channel has produced).  Same device functions, same operation order as the two-kernel path and
// as encode_intervals_kernel: identical entries.
// win_stats (round 5; nullptr: the classic form above): per image the miss count its decoder reported two chunks earlier.  The rows of image
// b then are WINDOW rows (csrc/dmll_core.h: 65 entries around the mixture's mean, a quarter of the sigmoids and of the bytes) when
// use_window(win_stats[b]) says so -- the decoder evaluates the same function of the same number -- and full rows otherwise,
// whose entry Lp - 1 (never read by any coder: the top symbol's upper bound is 2^16) then carries the window offset, so that a decoder
// working on full rows can still count what a window would have missed.  Either way image b's rows start at its full-size slot
// cdf + b * range_len * Lp.
// GROUPED (round 6): blockIdx.z picks one of up to kMaxTableParts parts -- the channels of one pipeline step of the RGB decode (R chunk
// j, G chunk j - D, B chunk j - 2 D) or the C channels of a bottleneck scale -- so that a step's tables are ONE launch (the decode was
// within 30 % of being bound by the host's launch rate).  Parts share P, sym, targets and the shape; ranges may differ in length (grid.x
// covers the longest; the surplus blocks of a shorter part leave at once).
constexpr int kMaxTableParts = 8;
struct TablePart {
    int c;
    int64_t range0, range_len;
    uint16_t *cdf;
    int32_t *not_monotone;
    const int32_t *win_stats;
    // RAGGED batches (round 6, l3c_decode_rgb_ragged: images of DIFFERENT sizes in one launch; null = every image [HW] pixels, rows
    // [range0, range0 + range_len)): per image b its pixel range of this part and the byte offset of its rows inside `cdf`
    const int64_t *r_pix0, *r_npix, *r_table_off;
};
struct TableParts {
    TablePart part[kMaxTableParts];
    const int64_t *r_pixbase, *r_hw;   // ragged: image b's pixels start at pixel r_pixbase[b] of P (and its C planes at element C * r_pixbase[b] of sym), it has r_hw[b] of them
};

__global__ __launch_bounds__(256) void cdf_table_from_P_kernel(const float *__restrict__ P, const int16_t *__restrict__ sym,
                                                               const float *__restrict__ targets, int64_t HW, int C, int K,
                                                               int rgb, int Lp, const TableParts parts, TileDiv dv) {
    extern __shared__ __attribute__((aligned(16))) float tile[];   // [kTablePix][Kp + 1]
    __shared__ float s_pi[kTablePix][kMaxK], s_mu[kTablePix][kMaxK], s_inv[kTablePix][kMaxK];
    __shared__ float s_e[kTablePix][kMaxK], s_max[kTablePix];
    __shared__ float s_t[260];
    const TablePart &part = parts.part[blockIdx.z];
    const int c = part.c;
    int32_t *__restrict__ not_monotone = part.not_monotone;
    const int32_t *__restrict__ win_stats = part.win_stats;
    const int Kp = (rgb ? 4 : 3) * C * K;
    const int ld = Kp + 1;
    const int64_t b = blockIdx.y;
    // image b: its pixels in P / sym, its range of this part, its rows in the table -- a rectangular batch, or a RAGGED one (per-image arrays)
    const bool ragged = parts.r_hw != nullptr;
    if (ragged) HW = parts.r_hw[b];
    const int64_t img_pix = ragged ? parts.r_pixbase[b] : b * HW;
    const int64_t range0 = ragged ? part.r_pix0[b] : part.range0, range_len = ragged ? part.r_npix[b] : part.range_len;
    uint16_t *__restrict__ cdf = ragged ? part.cdf + (part.r_table_off[b] >> 1) : part.cdf + b * range_len * Lp;   // this image's rows (its full-size slot)
    P += img_pix * Kp;
    if (sym) sym += img_pix * C;
    const int64_t off = (int64_t)blockIdx.x * kTablePix;          // offset inside the range
    if (off >= range_len) return;                                 // (uniform: a shorter part of a grouped launch, a smaller image of a ragged one)
    const int64_t pix0 = range0 + off;
    const int npix = (int)((range_len - off) < kTablePix ? (range_len - off) : kTablePix);
    const int tid = threadIdx.x;
    const bool window = win_stats && l3c::use_window(win_stats[b]);   // uniform over the image's blocks
    const float *src = P + pix0 * Kp;
    fill_tile<256>(tile, src, npix, Kp, ld, dv, tid);
    for (int i = tid; i < Lp; i += 256) s_t[i] = targets[i];
    __syncthreads();
    // The mixture parameters of the 32 pixels, one (pixel, component) item per thread in both phases (round 5; one thread per pixel
    // walked mix_stats before: 32 of 256 lanes busy for what is, with 65-entry rows, a third of the block's work).  Same operations in the
    // same order as mix_stats / mix_component: the maximum by the same fmaxf chain, the softmax numerators e_k = expf(l_k - max) computed
    // once, the denominator their SEQUENTIAL sum (every thread of a pixel adds the same ten values in the same order).
    for (int i = tid; i < kTablePix * K; i += 256) {
        const int p = i % kTablePix, k = i / kTablePix;
        if (p < npix) {
            const float *px = tile + p * ld;
            float mx = px[c * K];
            for (int j = 1; j < K; ++j) mx = fmaxf(mx, px[c * K + j]);
            s_e[p][k] = expf(px[c * K + k] - mx);
            if (k == 0) s_max[p] = mx;
        }
    }
    __syncthreads();
    for (int i = tid; i < kTablePix * K; i += 256) {
        const int p = i % kTablePix, k = i / kTablePix;
        if (p < npix) {
            const float *px = tile + p * ld;
            const int64_t n = pix0 + p;
            float x0 = 0.f, x1 = 0.f;
            if (rgb && c > 0) {
                x0 = (float)sym[0 * HW + n];
                if (c > 1) x1 = (float)sym[1 * HW + n];
            }
            MixStats st;
            st.max_logit = s_max[p];
            st.denom = 0.0f;
            for (int j = 0; j < K; ++j) st.denom = st.denom + s_e[p][j];
            const MixComponent m = mix_component_e([&](int ch) { return px[ch]; }, st, s_e[p][k], C, K, rgb, c, k, x0, x1);
            s_pi[p][k] = m.pi;
            s_mu[p][k] = m.mu;
            s_inv[p][k] = expf(-m.log_sigma);
        }
    }
    __syncthreads();
    const float scale = (float)(65536 - (Lp - 1));
    auto pixel_w0 = [&](int p) {   // the window's offset: around the mixture's mean sum_k pi_k mu_k (sequential)
        float mean = 0.0f;
        for (int k = 0; k < K; ++k) mean = mean + s_pi[p][k] * s_mu[p][k];
        return l3c::window_offset(mean);
    };
    const int lane = tid & 63;
    if (window) {
        // WINDOW rows: 32 pixels x 64 evaluated entries = 8 per thread -- thread t: pixel t / 8, entries 8 (t % 8) .. + 7 of its window.  The
        // pixel's parameters are read once per component for all eight entries; a pixel's row lives in eight adjacent lanes, so the
        // monotonicity check needs the next lane's first entry and nothing else.
        const int p = tid >> 3, q = tid & 7;
        bool bad = false;
        if (p < npix) {
            const int w0 = pixel_w0(p);
            const int l0 = w0 + q * 8;
            float tt[8], acc[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                tt[i] = s_t[l0 + i];
                acc[i] = 0.0f;
            }
            for (int k = 0; k < K; ++k) {
                const float pi = s_pi[p][k], mu = s_mu[p][k], inv = s_inv[p][k];
#pragma unroll
                for (int i = 0; i < 8; ++i) acc[i] = acc[i] + cdf_term(pi, mu, inv, tt[i]);
            }
            uint32_t v[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) v[i] = cdf_quantise(acc[i], scale, l0 + i);
            uint16_t *row = cdf + (off + p) * l3c::kWinLp + q * 8;
#pragma unroll
            for (int i = 0; i < 8; ++i) row[i] = (uint16_t)v[i];
            if (q == 7) row[8] = (uint16_t)w0;          // entry 64: the window's offset
#pragma unroll
            for (int i = 0; i < 7; ++i) bad = bad || !(v[i] < v[i + 1]);
            const uint32_t next = (uint32_t)__shfl_down((int)v[0], 1, 64);   // entry 8 (q + 1) of the same pixel (q < 7: the same wavefront)
            bad = bad || (q < 7 && !(v[7] < next));
        } else {
            (void)__shfl_down(0, 1, 64);
        }
        if (not_monotone && __any(bad) && lane == 0) atomicOr(not_monotone, 1);
        return;
    }
    const int count = npix * Lp;
    uint16_t *out = cdf + off * Lp;
    const bool aligned4 = ((reinterpret_cast<uintptr_t>(out) & 3) == 0);
    // The strict-monotonicity check the decoder needs (l3c_cdf_check_monotone: entries 0 .. Lp-2 of every row) is done on the
    // entries while they are in registers instead of re-reading the table (a 24 GB pass per batch of 128 otherwise): a thread
    // holds entries e, e + 1; entry e + 2 is the next lane's first; what a wavefront's last lane needs comes from another
    // wavefront or the next turn of this loop and goes through s_edge, checked after the loop.
    __shared__ uint32_t s_edge[kTablePix * 260 / 128 + 2][2];   // per run of 128 entries: its first and its last entry
    bool bad = false;
    for (int e = tid * 2; e < count; e += 512) {
        uint32_t v[2];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int ee = e + h < count ? e + h : e;
            const int p = ee / Lp, l = ee - p * Lp;
            if (win_stats && l == Lp - 1) {      // rows of a windowed part: the (never decoded) last entry carries the offset a window would have
                v[h] = (uint32_t)pixel_w0(p);
            } else {
                const float t = s_t[l];
                float acc = 0.0f;
                for (int k = 0; k < K; ++k) acc = acc + cdf_term(s_pi[p][k], s_mu[p][k], s_inv[p][k], t);
                v[h] = cdf_quantise(acc, scale, l);
            }
        }
        if (aligned4 && e + 1 < count) {
            *reinterpret_cast<uint32_t *>(out + e) = v[0] | (v[1] << 16);
        } else {
            out[e] = (uint16_t)v[0];
            if (e + 1 < count) out[e + 1] = (uint16_t)v[1];
        }
        if (not_monotone) {   // pairs (m, m + 1) with m <= Lp - 3 inside a row: entry index mod Lp
            const int l0 = e % Lp;
            bad = bad || (e + 1 < count && l0 <= Lp - 3 && !(v[0] < v[1]));
            const uint32_t next = (uint32_t)__shfl_down((int)v[0], 1, 64);   // entry e + 2 (lanes 0 .. 62)
            const int l1 = l0 + 1 < Lp ? l0 + 1 : 0;
            bad = bad || (lane < 63 && e + 2 < count && l1 <= Lp - 3 && !(v[1] < next));
            if (lane == 0) s_edge[e >> 7][0] = v[0];
            if (lane == 63) s_edge[e >> 7][1] = v[1];
        }
    }
    if (not_monotone) {
        __syncthreads();
        const int runs = (count + 127) >> 7;   // run j: entries 128 j .. 128 j + 127; its last entry against the next run's first
        for (int j = tid; j + 1 < runs; j += 256) {
            const int m = (128 * j + 127) % Lp;
            bad = bad || (m <= Lp - 3 && !(s_edge[j][1] < s_edge[j + 1][0]));
        }
        if (__any(bad) && lane == 0) atomicOr(not_monotone, 1);
    }
}

// ---- fused encoder head: P + symbols -> packed coding intervals -------------------------------------------------------

constexpr int kHeadPix = 64;  // == interval block length, so one block writes whole 256-byte interval runs

// (Tried: kHeadPix * C threads per block -- one (pixel, channel) item per thread, no idle lanes in the compute phase: 4.4 ->
// 6.8 ms per launch at batch 128 [PMC]: fewer wavefronts per CU for the LDS-bound tile fill outweigh the busier lanes.)
template <int NT>
__global__ __launch_bounds__(NT) void encode_intervals_kernel(const float *__restrict__ P, const int16_t *__restrict__ sym,
                                                               const float *__restrict__ targets, int64_t HW, int C, int K,
                                                               int rgb, int Lp, uint32_t *__restrict__ iv, TileDiv dv) {
    extern __shared__ __attribute__((aligned(16))) float tile[];   // [kHeadPix][Kp + 1]
    const int Kp = (rgb ? 4 : 3) * C * K;
    const int ld = Kp + 1;
    const int64_t b = blockIdx.y;
    const int64_t chunk = blockIdx.x;
    const int64_t pix0 = chunk * kHeadPix;
    const int npix = (int)((HW - pix0) < kHeadPix ? (HW - pix0) : kHeadPix);
    const int tid = threadIdx.x;
    const float *src = P + (b * HW + pix0) * Kp;
    constexpr int nthreads = NT;
    fill_tile<NT>(tile, src, npix, Kp, ld, dv, tid);   // (a wavefront per pixel row instead: 4.4 -> 6.8 ms)
    __syncthreads();
    const float scale = (float)(65536 - (Lp - 1));
    const int64_t n_streams = (int64_t)gridDim.y * C;
    for (int w = tid; w < kHeadPix * C; w += nthreads) {
        const int p = w % kHeadPix, c = w / kHeadPix;
        uint32_t word0 = 0, word1 = 0;   // the stream's lane pair reads one word per role (csrc/ac_core.h: role_word)
        if (p < npix) {
            const float *px = tile + p * ld;
            auto get = [&](int ch) { return px[ch]; };
            const int64_t n = pix0 + p;
            const int x = sym[(b * C + c) * HW + n];
            float x0 = 0.f, x1 = 0.f;
            if (rgb && c > 0) {
                x0 = (float)sym[(b * C + 0) * HW + n];
                if (c > 1) x1 = (float)sym[(b * C + 1) * HW + n];
            }
            const float t_lo = targets[x];
            const float t_hi = targets[x + 1];
            // softmax statistics with the numerators kept (mix_stats' two loops, same operations in the same order)
            MixStats st;
            st.max_logit = get(c * K);
            for (int k = 1; k < K; ++k) st.max_logit = fmaxf(st.max_logit, get(c * K + k));
            float e[kMaxK];
            st.denom = 0.0f;
#pragma unroll
            for (int k = 0; k < kMaxK; ++k) {
                e[k] = k < K ? expf(get(c * K + k) - st.max_logit) : 0.0f;
                if (k < K) st.denom = st.denom + e[k];
            }
            float acc_lo = 0.0f, acc_hi = 0.0f;
#pragma unroll
            for (int k = 0; k < kMaxK; ++k) {
                if (k >= K) break;
                const MixComponent m = mix_component_e(get, st, e[k], C, K, rgb, c, k, x0, x1);
                const float inv = expf(-m.log_sigma);
                acc_lo = acc_lo + cdf_term(m.pi, m.mu, inv, t_lo);
                acc_hi = acc_hi + cdf_term(m.pi, m.mu, inv, t_hi);
            }
            const uint32_t c_lo = cdf_quantise(acc_lo, scale, x);
            const uint32_t c_hi = (x == Lp - 2) ? 0x10000u : cdf_quantise(acc_hi, scale, x + 1);
            word0 = l3c::role_word(c_lo, c_hi, 0);
            word1 = l3c::role_word(c_lo, c_hi, 1);
        }
        uint32_t *dst = iv + (chunk * n_streams + b * C + c) * (2 * kHeadPix) + p;
        dst[0] = word0;
        dst[kHeadPix] = word1;
    }
}

// ---- negative log-likelihood map ---------------------------------------------------------------------------------------

__device__ __forceinline__ float softplus_f(float x) { return x > 20.0f ? x : log1pf(expf(x)); }  // F.softplus defaults

__global__ __launch_bounds__(256) void nll_kernel(const float *__restrict__ P, const float *__restrict__ xs, int64_t HW,
                                                  int C, int K, int rgb, float x_lower, float x_upper, float half_bin,
                                                  float *__restrict__ nll) {
    extern __shared__ __attribute__((aligned(16))) float tile[];
    const int Kp = (rgb ? 4 : 3) * C * K;
    const int CK = C * K;
    const int ld = Kp + 1;
    const int64_t b = blockIdx.y;
    const int64_t pix0 = (int64_t)blockIdx.x * kHeadPix;
    const int npix = (int)((HW - pix0) < kHeadPix ? (HW - pix0) : kHeadPix);
    const int tid = threadIdx.x;
    const float *src = P + (b * HW + pix0) * Kp;
    for (int i = tid; i < npix * Kp; i += 256) tile[(i / Kp) * ld + (i % Kp)] = src[i];
    __syncthreads();
    for (int w = tid; w < kHeadPix * C; w += 256) {
        const int p = w % kHeadPix, c = w / kHeadPix;
        if (p >= npix) continue;
        const float *px = tile + p * ld;
        const int64_t n = pix0 + p;
        const float x = xs[(b * C + c) * HW + n];
        float x0 = 0.f, x1 = 0.f;
        if (rgb && c > 0) {
            x0 = xs[(b * C + 0) * HW + n];
            if (c > 1) x1 = xs[(b * C + 1) * HW + n];
        }
        // log_softmax over K of the mixture logits (logistic_mixture.py:334-337)
        float m = px[c * K];
        for (int k = 1; k < K; ++k) m = fmaxf(m, px[c * K + k]);
        float se = 0.0f;
        for (int k = 0; k < K; ++k) se = se + expf(px[c * K + k] - m);
        const float lse = logf(se);
        // first pass: weighted log-probs, keep their max; second pass: log-sum-exp (:340-345)
        float lp[kMaxK];
        float mx = -INFINITY;
#pragma unroll
        for (int k = 0; k < kMaxK; ++k) {
            if (k < K) {
                float mu = px[CK + c * K + k];
                if (rgb && c == 1) mu = mu + sigmoid_f(px[3 * CK + k]) * x0;
                if (rgb && c == 2) mu = (mu + sigmoid_f(px[3 * CK + K + k]) * x0) + sigmoid_f(px[3 * CK + 2 * K + k]) * x1;
                const float inv = expf(-fmaxf(px[2 * CK + c * K + k], kLogScalesMin));
                const float centered = x - mu;
                const float plus_in = inv * (centered + half_bin);
                const float min_in = inv * (centered - half_bin);
                const float delta = sigmoid_f(plus_in) - sigmoid_f(min_in);
                float lprob = logf(fmaxf(delta, 1e-12f));
                if (x > x_upper) lprob = -softplus_f(min_in);
                if (x < x_lower) lprob = plus_in - softplus_f(plus_in);
                lp[k] = lprob + ((px[c * K + k] - m) - lse);
                mx = fmaxf(mx, lp[k]);
            }
        }
        float s = 0.0f;
#pragma unroll
        for (int k = 0; k < kMaxK; ++k)
            if (k < K) s = s + expf(lp[k] - mx);
        nll[(b * C + c) * HW + n] = -(logf(s) + mx);
    }
}

// ---- sampling (DiscretizedMixLogisticLoss._non_shared_sample, logistic_mixture.py:277-323) ------------------------------------
// One thread per pixel walks the C channels: component by Gumbel-max over the mixture logits (first maximum wins, as
// torch.argmax), then the inverse logistic CDF of the chosen component; the RGB scale couples the channels through the
// lambda coefficients OF THE COMPONENTS CHOSEN FOR G AND B and clamps to [0, 255] (:305-322).  The uniforms are inputs, so
// the kernel is a pure function (and comparable with the oracle).
__global__ __launch_bounds__(256) void sample_kernel(const float *__restrict__ P, const float *__restrict__ u_mix,
                                                     const float *__restrict__ u_log, int64_t HW, int C, int K, int rgb,
                                                     float *__restrict__ x_out) {
    extern __shared__ __attribute__((aligned(16))) float tile[];
    const int Kp = (rgb ? 4 : 3) * C * K;
    const int CK = C * K;
    const int ld = Kp + 1;
    const int64_t b = blockIdx.y;
    const int64_t pix0 = (int64_t)blockIdx.x * kHeadPix;
    const int npix = (int)((HW - pix0) < kHeadPix ? (HW - pix0) : kHeadPix);
    const int tid = threadIdx.x;
    const float *src = P + (b * HW + pix0) * Kp;
    for (int i = tid; i < npix * Kp; i += 256) tile[(i / Kp) * ld + (i % Kp)] = src[i];
    __syncthreads();
    for (int p = tid; p < npix; p += 256) {
        const float *px = tile + p * ld;
        const int64_t n = pix0 + p;
        float xs[3] = {0.f, 0.f, 0.f};
        for (int c = 0; c < C; ++c) {
            int sel = 0;
            float best = -INFINITY;
            for (int k = 0; k < K; ++k) {
                const float u = u_mix[((b * C + c) * K + k) * HW + n];
                const float g = px[c * K + k] - logf(-logf(u));
                if (g > best) {
                    best = g;
                    sel = k;
                }
            }
            const float mu = px[CK + c * K + sel];
            const float ls = fmaxf(px[2 * CK + c * K + sel], kLogScalesMin);
            const float u = u_log[(b * C + c) * HW + n];
            float x = mu + expf(ls) * (logf(u) - logf(1.0f - u));
            if (rgb) {
                if (c == 1) x = x + sigmoid_f(px[3 * CK + sel]) * xs[0];
                if (c == 2) x = (x + sigmoid_f(px[3 * CK + K + sel]) * xs[0]) + sigmoid_f(px[3 * CK + 2 * K + sel]) * xs[1];
                x = fminf(fmaxf(x, 0.0f), 255.0f);
                xs[c] = x;
            }
            x_out[(b * C + c) * HW + n] = x;
        }
    }
}

int grid_1d(int64_t total, int block) {
    int64_t g = (total + block - 1) / block;
    return (int)(g < 1 ? 1 : (g > 2048 ? 2048 : g));
}

}  // namespace

extern "C" {

int l3c_dmll_channel_params(const float *P, const int16_t *sym, int64_t B, int64_t HW, int C, int K, int rgb, int c,
                            float *pi, float *mu, float *log_sigma, l3c_stream_t stream) {
    L3C_REQUIRE(P && pi && mu && log_sigma, "null pointer");
    L3C_REQUIRE(B > 0 && HW > 0 && C > 0 && K > 0 && c >= 0 && c < C, "bad shape");
    L3C_REQUIRE(!rgb || C == 3, "lambda coupling is only defined for C == 3 (logistic_mixture.py:44-51)");
    L3C_REQUIRE(!(rgb && c > 0) || sym, "sym required for the RGB scale, c > 0");
    hipLaunchKernelGGL(channel_params_kernel, dim3(grid_1d(B * HW, 256)), dim3(256), 0, l3c::as_stream(stream), P, sym,
                       B, HW, C, K, rgb, c, pi, mu, log_sigma);
    return l3c::check_launch("channel_params_kernel");
}

int l3c_cdf_table_mixture(const float *targets, const float *pi, const float *mu, const float *log_sigma, int64_t n_img,
                          int64_t HW, int K, int Lp, uint16_t *cdf, int32_t *not_monotone, l3c_stream_t stream) {
    L3C_REQUIRE(targets && pi && mu && log_sigma && cdf, "null pointer");
    L3C_REQUIRE(n_img > 0 && n_img < 65536 && HW > 0, "bad shape");
    L3C_REQUIRE(K > 0 && K <= kMaxK, "K out of range (1..16)");
    L3C_REQUIRE(Lp >= 2 && Lp <= 260, "Lp out of range (2..260)");
    const dim3 grid((unsigned)((HW + kTablePix - 1) / kTablePix), (unsigned)n_img);
    hipLaunchKernelGGL(cdf_table_kernel, grid, dim3(256), 0, l3c::as_stream(stream), targets, pi, mu, log_sigma, HW, K,
                       Lp, cdf);
    int rc = l3c::check_launch("cdf_table_kernel");
    if (rc == L3C_OK && not_monotone) rc = l3c_cdf_check_monotone(cdf, n_img * HW, Lp, not_monotone, stream);
    return rc;
}

static int cdf_table_parts_launch(const float *P, const int16_t *sym, const float *targets, int64_t B, int64_t HW, int C, int K, int rgb,
                                  int Lp, const l3c_table_part *parts, int n_parts, l3c_stream_t stream, const l3c_ragged_batch *rag = nullptr,
                                  const l3c_ragged_part *rag_parts = nullptr) {
    L3C_REQUIRE(P && targets && parts, "null pointer");
    L3C_REQUIRE(n_parts > 0 && n_parts <= kMaxTableParts, "1..8 parts per call");
    L3C_REQUIRE(B > 0 && B < 65536 && HW > 0 && C > 0, "bad shape");
    L3C_REQUIRE(K > 0 && K <= kMaxK, "K out of range (1..16)");
    L3C_REQUIRE(Lp >= 2 && Lp <= 260, "Lp out of range (2..260)");
    L3C_REQUIRE(!rgb || C == 3, "lambda coupling is only defined for C == 3");
    const int Kp = (rgb ? 4 : 3) * C * K;
    const size_t lds = (size_t)kTablePix * (Kp + 1) * sizeof(float);
    L3C_REQUIRE(lds <= 48 * 1024, "Kp too large for the LDS tile");
    TableParts tp{};
    int64_t longest = 0;
    for (int i = 0; i < n_parts; ++i) {
        const l3c_table_part &q = parts[i];
        L3C_REQUIRE(q.cdf, "null table pointer in part");
        L3C_REQUIRE(q.c >= 0 && q.c < C, "channel out of range");
        L3C_REQUIRE(rag || (q.pix0 >= 0 && q.npix > 0 && q.pix0 + q.npix <= HW), "pixel range outside the image");
        L3C_REQUIRE(!(rgb && q.c > 0) || sym, "the RGB scale needs the symbols of the channels decoded so far");
        L3C_REQUIRE(!q.window_stats || Lp == 257, "window rows are defined for the 256-symbol alphabet (Lp == 257)");
        tp.part[i] = TablePart{q.c, q.pix0, q.npix, q.cdf, q.not_monotone, q.window_stats, nullptr, nullptr, nullptr};
        if (rag) {      // q.npix = the LONGEST range of the part (it sizes the grid); the per-image ranges come from the device arrays
            L3C_REQUIRE(rag_parts && rag_parts[i].pix0 && rag_parts[i].npix && rag_parts[i].table_off, "ragged part without its arrays");
            tp.part[i].r_pix0 = rag_parts[i].pix0;
            tp.part[i].r_npix = rag_parts[i].npix;
            tp.part[i].r_table_off = rag_parts[i].table_off;
        }
        longest = q.npix > longest ? q.npix : longest;
    }
    if (rag) {
        L3C_REQUIRE(rag->pixbase && rag->hw, "ragged batch without its arrays");
        tp.r_pixbase = rag->pixbase;
        tp.r_hw = rag->hw;
    }
    const dim3 grid((unsigned)((longest + kTablePix - 1) / kTablePix), (unsigned)B, (unsigned)n_parts);
    hipLaunchKernelGGL(cdf_table_from_P_kernel, grid, dim3(256), lds, l3c::as_stream(stream), P, sym, targets, HW, C, K, rgb, Lp, tp,
                       tile_div(Kp, kTablePix));
    return l3c::check_launch("cdf_table_from_P_kernel");   // (the kernel has checked the rows while it held them: not_monotone)
}

int l3c_dmll_cdf_table(const float *P, const int16_t *sym, const float *targets, int64_t B, int64_t HW, int C, int K, int rgb,
                       int c, int64_t pix0, int64_t npix, int Lp, uint16_t *cdf, int32_t *not_monotone, const int32_t *window_stats,
                       l3c_stream_t stream) {
    L3C_REQUIRE(cdf, "null pointer");
    const l3c_table_part part{c, pix0, npix, cdf, not_monotone, window_stats};
    return cdf_table_parts_launch(P, sym, targets, B, HW, C, K, rgb, Lp, &part, 1, stream);
}

int l3c_dmll_cdf_table_parts(const float *P, const int16_t *sym, const float *targets, int64_t B, int64_t HW, int C, int K, int rgb,
                             int Lp, const l3c_table_part *parts_host, int n_parts, l3c_stream_t stream) {
    return cdf_table_parts_launch(P, sym, targets, B, HW, C, K, rgb, Lp, parts_host, n_parts, stream);
}

int l3c_dmll_cdf_table_ragged(const float *P, const int16_t *sym, const float *targets, const l3c_ragged_batch *batch, int C, int K, int rgb,
                              int Lp, const l3c_table_part *parts_host, const l3c_ragged_part *ragged_parts_host, int n_parts,
                              l3c_stream_t stream) {
    L3C_REQUIRE(batch && ragged_parts_host, "null pointer");
    L3C_REQUIRE(batch->B > 0 && batch->max_hw > 0, "bad ragged batch");
    return cdf_table_parts_launch(P, sym, targets, batch->B, batch->max_hw, C, K, rgb, Lp, parts_host, n_parts, stream, batch, ragged_parts_host);
}

int l3c_dmll_encode_intervals(const float *P, const int16_t *sym, const float *targets, int64_t B, int64_t HW, int C,
                              int K, int rgb, int Lp, uint32_t *intervals, l3c_stream_t stream) {
    L3C_REQUIRE(P && sym && targets && intervals, "null pointer");
    L3C_REQUIRE(B > 0 && B < 65536 && HW > 0 && C > 0 && K > 0, "bad shape");
    L3C_REQUIRE(K <= kMaxK, "K out of range");
    L3C_REQUIRE(!rgb || C == 3, "lambda coupling is only defined for C == 3");
    L3C_REQUIRE(Lp >= 2 && Lp <= 65536, "Lp out of range");
    const int Kp = (rgb ? 4 : 3) * C * K;
    const size_t lds = (size_t)kHeadPix * (Kp + 1) * sizeof(float);
    L3C_REQUIRE(lds <= 64 * 1024, "Kp too large for the LDS tile");
    const dim3 grid((unsigned)((HW + kHeadPix - 1) / kHeadPix), (unsigned)B);
    // Threads per block [measured, round 5, profiles/r05_interval_kernel_threads.log, batch 32]: the bottleneck scales' 320 (pixel, channel)
    // items on 320 threads -- one each -- 1.28 ms against 1.40 ms on 256 (where a second, quarter-full turn of the compute loop follows the
    // first); the RGB scale's 192 items stay on 256 threads: 2.16 ms against 2.30 ms on 192 (the fourth wavefront idles through the compute
    // phase but helps fill the tile).
#ifdef L3C_IV_THREADS_256
    const int nthreads = 256;
#else
    const int nthreads = kHeadPix * C == 320 ? 320 : 256;
#endif
    if (nthreads == 320)
        hipLaunchKernelGGL(encode_intervals_kernel<320>, grid, dim3(320), lds, l3c::as_stream(stream), P, sym, targets, HW, C, K, rgb, Lp,
                           intervals, tile_div(Kp, kHeadPix));
    else
        hipLaunchKernelGGL(encode_intervals_kernel<256>, grid, dim3(256), lds, l3c::as_stream(stream), P, sym, targets, HW, C, K, rgb, Lp,
                           intervals, tile_div(Kp, kHeadPix));
    return l3c::check_launch("encode_intervals_kernel");
}

int l3c_dmll_nll(const float *P, const float *x, int64_t B, int64_t HW, int C, int K, int rgb, float x_min, float x_max,
                 int L, float *nll, l3c_stream_t stream) {
    L3C_REQUIRE(P && x && nll, "null pointer");
    L3C_REQUIRE(B > 0 && B < 65536 && HW > 0 && C > 0, "bad shape");
    L3C_REQUIRE(K > 0 && K <= kMaxK, "K out of range (1..16)");
    L3C_REQUIRE(!rgb || C == 3, "lambda coupling is only defined for C == 3");
    L3C_REQUIRE(L >= 2, "L out of range");
    const int Kp = (rgb ? 4 : 3) * C * K;
    const size_t lds = (size_t)kHeadPix * (Kp + 1) * sizeof(float);
    L3C_REQUIRE(lds <= 64 * 1024, "Kp too large for the LDS tile");
    // bounds and half bin exactly as logistic_mixture.py:113-116 computes them (in double), then to fp32
    const double bin = ((double)x_max - (double)x_min) / (double)(L - 1);
    const float x_lower = (float)((double)x_min + 0.001), x_upper = (float)((double)x_max - 0.001);
    const dim3 grid((unsigned)((HW + kHeadPix - 1) / kHeadPix), (unsigned)B);
    hipLaunchKernelGGL(nll_kernel, grid, dim3(256), lds, l3c::as_stream(stream), P, x, HW, C, K, rgb, x_lower, x_upper,
                       (float)(bin / 2.0), nll);
    return l3c::check_launch("nll_kernel");
}

int l3c_dmll_sample(const float *P, const float *u_mix, const float *u_logistic, int64_t B, int64_t HW, int C, int K,
                    int rgb, float *x, l3c_stream_t stream) {
    L3C_REQUIRE(P && u_mix && u_logistic && x, "null pointer");
    L3C_REQUIRE(B > 0 && B < 65536 && HW > 0 && C > 0, "bad shape");
    L3C_REQUIRE(K > 0 && K <= kMaxK, "K out of range (1..16)");
    L3C_REQUIRE(!rgb || C == 3, "lambda coupling is only defined for C == 3");
    const int Kp = (rgb ? 4 : 3) * C * K;
    const size_t lds = (size_t)kHeadPix * (Kp + 1) * sizeof(float);
    L3C_REQUIRE(lds <= 64 * 1024, "Kp too large for the LDS tile");
    const dim3 grid((unsigned)((HW + kHeadPix - 1) / kHeadPix), (unsigned)B);
    hipLaunchKernelGGL(sample_kernel, grid, dim3(256), lds, l3c::as_stream(stream), P, u_mix, u_logistic, HW, C, K, rgb, x);
    return l3c::check_launch("sample_kernel");
}
}
