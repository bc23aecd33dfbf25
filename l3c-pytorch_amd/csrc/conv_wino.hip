// conv_wino.hip -- 3x3 / stride 1 convolution (dilation 1, 2, 4) by Winograd F(2x2, 3x3) on the fp32 MFMA.
//
// 2.25x fewer multiplications than the implicit GEMM of conv_mfma.hip: every 2x2 output tile is
//     Y = A^T [ sum_ci (G g G^T) (.) (B^T d B) ] A
// with d the 4x4 input tile.  The sum over input channels of the element-wise products is, for each of the 16 positions (xi, nu)
// of the transformed tile, a plain GEMM  M[xi,nu] (tiles x Cout) = V[xi,nu] (tiles x Cin) * U[xi,nu] (Cin x Cout)  -- that part
// runs on v_mfma_f32_32x32x2_f32; the transforms are a few dozen additions per tile and channel.
//
//   * block = 256 threads = 4 wavefronts on a 4 x 32 output tile = 2 x 16 = 32 Winograd tiles, 64 output channels.
//     Wavefront (nj, ph) owns 32 tiles x 32 output channels (nj) for the EIGHT positions of two columns nu = 2 ph, 2 ph + 1 of
//     the transformed tile: 8 accumulator fragments = 128 registers.  A wavefront therefore needs at most 256 registers and a
//     block 64 KB of LDS: TWO blocks share a CU (two wavefronts per SIMD), and while one block is in its prologue, at its
//     chunk barrier or in its output transform, the matrix pipe runs the other one's MFMAs -- the hardware interleaves
//     what a single 512-register wavefront per SIMD could only approximate by hand;
//   * the output transform is linear, so each wavefront transforms ITS two columns (A^T M_ph A_ph) and writes the partial
//     2 x 2 outputs to LDS as [pixel][channel]; wavefront (nj, ph) then finishes tile row ph: it reads both halves in the
//     STORE layout (four channels adjacent in memory per lane) and leaves 8 x 16-byte stores per lane;
//   * input channels in chunks of 8: the raw 6 x 34 patch of chunk k is fetched into registers during chunk k - 3, written to
//     LDS during k - 2, transformed during k - 1 by all 256 threads (thread = tile x channel pair x output-row pair: 12
//     ds_read_b64, 16 packed additions, 8 ds_write_b64, bank-conflict-free) into the other V[16][32 tiles][8] buffer WHILE the
//     MFMAs of the current chunk run -- the slices of the transform are dealt out between the MFMAs;
//   * the pre-transformed weights U (l3c_conv_wino_pack_weights: 16/9 of the 3x3 weights, MFMA fragment order) never touch
//     LDS: position q of a chunk is ONE coalesced 16-byte buffer load per lane straight into the B operand registers, issued
//     a whole chunk ahead (they come from L2: 256 KB per 64 output channels, shared by every block);
//   * per position one ds_read_b128 of V and one operand quad of U feed four MFMAs (the k-ordering trick of conv_mfma.hip).
// A dilated conv is the dense conv on each of the dil x dil interleaved sub-grids of the image: same kernel, strided indexing.
// fp32 throughout; the result differs from the direct convolution by rounding only (transform coefficients are 1, 1/2).
#include "l3c_common.h"
#include "../../include/l3c_xcheck.h"

#include <stdlib.h>

#include <atomic>
#include <type_traits>

namespace {

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// n / d for 0 <= n < 2^31 as a multiplication (Granlund-Montgomery: m = floor(2^(31+L) / d) + 1, L = ceil(log2 d))
struct WinoDiv {
    unsigned m, sh;   // d == 1: m = 0
    __device__ __forceinline__ unsigned div(unsigned n) const { return m ? __umulhi(n, m) >> sh : n; }
};
static WinoDiv wino_div(unsigned d) {
    WinoDiv r{0, 0};
    if (d <= 1) return r;
    unsigned L = 0;
    while ((1ull << L) < d) ++L;
    r.m = (unsigned)(((1ull << (31 + L)) / d) + 1);
    r.sh = L - 1;
    return r;
}

struct WinoParams {
    const float *in;
    const float *u;
    const float *bias;
    const float *res;
    float *out;
    int in_cstride, in_coff, res_cstride, res_coff, out_cstride, out_coff;
    int B, H, W, Cin, Cout;
    int dil;          // 1, 2 or 4: the output grid splits into dil x dil interleaved sub-grids, each an ordinary 3x3 conv
    int dil_log2;
    WinoDiv div_groups, div_groups_x, div_chunks;
#ifdef L3C_WINO_TIMELINE
    unsigned long long *dbg;   // development build: per-wavefront s_memtime stamps (tools/wino_timeline.py)
#endif
    int epilogue;
    int tiles_x, tiles_y, n_chunks_o, total_blocks;
    int tpb, groups_x;   // a block walks up to tpb horizontally adjacent tiles; groups_x = ceil(tiles_x / tpb)
};

constexpr int WT_H = 4, WT_W = 32;               // output tile of a block: 2 x 16 Winograd tiles
constexpr int WP_H = WT_H + 2, WP_W = WT_W + 2;  // input patch
constexpr int N_TILES = (WT_H / 2) * (WT_W / 2);
constexpr int WCK = 8;                           // input channels per chunk
constexpr int WINO_TPB_MAX = 6;                  // tiles a block walks (24, 12, 6, 3 tiles per row at the L3C resolutions); 3 -> 6: +1.5 %
constexpr int PSR = 12;                          // LDS stride of a raw patch pixel (floats)
constexpr int PSV = 8;                           // LDS stride of a transformed tile (floats), its two 4-float groups swizzled
constexpr int RAW_FLOATS = WP_H * WP_W * PSR;
constexpr int V_FLOATS = 16 * N_TILES * PSV;     // one buffer (two: the next chunk is transformed during the MFMA loop)
constexpr int U_FLOATS = 16 * 2 * 64 * 4;        // packed weights of one input chunk x one 64-channel output chunk
constexpr int Y_FLOATS = WT_H * WT_W * 32;       // output exchange: the 2x2 outputs of one channel half, [pixel][32 channels]
constexpr int LDS_FLOATS = 16384;                // 65 536 bytes: two blocks per CU
constexpr int WINO_LDS_BYTES = LDS_FLOATS * 4;
// LDS map (floats): [V0][raw1][spare][raw0][V1].  After the last chunk g of a tile the pipeline (which runs on into the next
// tile) still needs V[(g+1) & 1] and raw[g & 1]; V[g & 1] and raw[(g+1) & 1] are free -- with this map they and the spare
// floats between them are ONE contiguous run of 9840 floats at either end, enough for the 2 x Y_FLOATS output exchange.
constexpr int V_OFF0 = 0, RAW_OFF1 = V_FLOATS, RAW_OFF0 = LDS_FLOATS - V_FLOATS - RAW_FLOATS, V_OFF1 = LDS_FLOATS - V_FLOATS;
static_assert(RAW_OFF1 + RAW_FLOATS <= RAW_OFF0, "LDS layout");
static_assert(2 * Y_FLOATS <= RAW_OFF0 && LDS_FLOATS - 2 * Y_FLOATS >= RAW_OFF1 + RAW_FLOATS, "the exchange must fit the free run");

__device__ __forceinline__ int xcd_remap_w(int bid, int total) {
    const int q = total >> 3, r = total & 7;
    const int xcd = bid & 7, slot = bid >> 3;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + slot;
}

// Development probes (csrc/build.py --wino-probe N; results are WRONG, only the time means something): bit 0 no patch loads,
// 1 no patch stores to LDS, 2 no input transform, 3 no weight loads inside the loop, 4 no chunk barrier, 5 no output stores,
// (tried and dropped: the patch fetch behind the chunk's weight loads instead of ahead of them -- no difference).  Never defined
// in the product build.
#ifndef L3C_WINO_PROBE
#define L3C_WINO_PROBE 0
#endif
#ifdef L3C_WINO_TIMELINE
#define L3C_WINO_STAMP(i) __builtin_amdgcn_sched_barrier(0); dbg_t[i] = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_sched_barrier(0);
#else
#define L3C_WINO_STAMP(i)
#endif

// a - b as ONE plain v_sub_f32 (the optimiser folds extract / subtract / insert on the two halves of a float2 back into a packed
// v_pk_add_f32 whatever the source says; an asm statement it cannot)
__device__ __forceinline__ float sub_scalar(float a, float b) {
    float r;
    asm("v_sub_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

// RELU / RES / SHUFFLE: the epilogue variant, compile-time (no per-element selects).
//
// A block walks n_t <= p.tpb horizontally adjacent tiles (same image, sub-grid, tile row and output-channel chunk, hence the
// same weights).  Its chunk pipeline -- patch of chunk g+3 being fetched, g+2 stored to LDS, g+1 transformed, g multiplied --
// runs over ONE global chunk sequence g = tile * n_cc + cc: the fetches that the last chunks of a tile issue are the first
// patches of the NEXT tile, so only the first tile of a block pays the prologue's memory round trip.  Between two tiles
// sits the output transform of the finished one; it works in the part of LDS the pipeline does not need at that point.
template <bool RELU, bool RES, bool SHUFFLE>
__global__ __launch_bounds__(256, 2) void conv_wino_kernel(const WinoParams p) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    auto raw_buf = [&](int par) { return lds + (par ? RAW_OFF1 : RAW_OFF0); };   // patch of chunk g in raw[g & 1]
    auto v_buf = [&](int par) { return lds + (par ? V_OFF1 : V_OFF0); };         // transformed tiles of chunk g in V[g & 1]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // uniform, and the compiler knows it
    const int half = lane >> 5, lx = lane & 31;
    const int nj = wave & 1, ph = wave >> 1;     // output-channel half; column pair nu = 2 ph, 2 ph + 1 of the transformed tile
    const int n_cc = p.Cin / WCK;
#ifdef L3C_WINO_TIMELINE
    unsigned long long dbg_t[12] = {};
#endif
    L3C_WINO_STAMP(0)

    // block -> (image, output-channel chunk, sub-grid, tile row, group of tiles): divisions by multiplication with
    // host-computed reciprocals
    unsigned w = (unsigned)xcd_remap_w(blockIdx.x, p.total_blocks);
    const unsigned groups = (unsigned)(p.groups_x * p.tiles_y);
    const unsigned w_t = p.div_groups.div(w);
    const unsigned grp = w - w_t * groups;
    w = w_t;
    const int dl = p.dil_log2, dil = 1 << dl;
    const int phase = (int)(w & ((1u << (2 * dl)) - 1));   // which of the dil x dil sub-grids (dilated conv = dense conv on each)
    w >>= 2 * dl;
    const int b = (int)p.div_chunks.div(w);
    const int chunk_o = (int)(w - (unsigned)b * (unsigned)p.n_chunks_o);
    const int py = phase >> dl, px = phase & (dil - 1);
    const unsigned t_y = p.div_groups_x.div(grp);
    const int sy0 = (int)t_y * WT_H;                                       // tile row origin in sub-grid coordinates
    const int tx_first = (int)(grp - t_y * (unsigned)p.groups_x) * p.tpb;  // first tile of this block
    const int n_t = min(p.tpb, p.tiles_x - tx_first);

    constexpr int N_PIECES = WP_H * WP_W * 2;    // 16-byte pieces of a patch (a pixel's 8 channels = 2 pieces)
    constexpr int NIT = (N_PIECES + 255) / 256;
    static_assert(NIT == 2, "the patch prefetch pieces are dealt out by hand below");
    f32x4 stage_regs[2][NIT];   // two patches in flight: set g & 1 holds the patch of chunk g + 2, fetched during chunk g - 2
    // this thread's patch elements: byte offsets relative to the image base, for the tile the prefetch pointer is in.  The
    // image is one buffer descriptor: a piece outside it (zero padding) gets an offset beyond the buffer and reads as zero.
    // The lane index, recomputed where it is used rarely (v_mbcnt; the opaque mask keeps the compiler from hoisting the
    // computation -- and everything derived from it -- out of the tile loop into registers held across the MFMA loop)
    auto fresh_lane = [&]() {
        unsigned m = ~0u;
        asm volatile("" : "+s"(m));
        return (int)__builtin_amdgcn_mbcnt_hi(m, __builtin_amdgcn_mbcnt_lo(m, 0u));
    };
    int patch_off[NIT];
    auto set_patch_tile = [&](int t) {   // t: tile of the block (recomputed from scratch: once per tile, no registers held)
        const int x0 = px + dil * ((tx_first + t) * WT_W - 1);
        const int tid_f = wave * 64 + fresh_lane();
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int i = tid_f + it * 256;
            const int c4 = i & 1, pix = i >> 1;
            const int r = pix / WP_W, ci = pix % WP_W;
            const int iy = py + dil * (sy0 - 1 + r), ix = x0 + dil * ci;
            const bool ok = i < N_PIECES && iy >= 0 && iy < p.H && ix >= 0 && ix < p.W;
            patch_off[it] = ok ? ((iy * p.W + ix) * p.in_cstride + c4 * 4) * 4 : 0x7ffffff0;
        }
    };
    // the prefetch pointer: (tile, chunk) of the next patch to fetch; past the block's last chunk it stays there (the loads
    // remain unconditional, their data is never used)
    int pf_tile = 0, pf_cc = 0;
    auto pf_advance = [&]() {
        if (++pf_cc == n_cc) {
            if (pf_tile + 1 < n_t) {
                pf_cc = 0;
                set_patch_tile(++pf_tile);
            } else {
                pf_cc = n_cc - 1;
            }
        }
    };
    const auto in_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float *>(p.in + (size_t)b * p.H * p.W * p.in_cstride + p.in_coff), 0, p.H * p.W * p.in_cstride * 4, 0x00020000);
    // B operands: uniform descriptor + fixed per-lane byte offset + scalar offset, no vector address arithmetic.  Every load
    // inside the chunk loop is UNCONDITIONAL (the last chunk re-requests slices it already has): a load behind a branch
    // would make the compiler's count of outstanding loads imprecise and turn its waits into vmcnt(0).
    const auto u_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(p.u + (size_t)chunk_o * n_cc * U_FLOATS), 0,
                                                          n_cc * U_FLOATS * 4, 0x00020000);
    const int u_lane = (ph * 2 * 512 + nj * 256 + lane * 4) * 4;   // the wavefront's first position is 2 ph
    f32x4 bq[8];   // operand a = 0..7 <-> position (a >> 1) * 4 + 2 ph + (a & 1)
    auto fetch_b = [&](int cc, int a) {
        bq[a] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(
                                              u_rsrc, u_lane, (cc * U_FLOATS + ((a >> 1) * 4 + (a & 1)) * 512) * 4, 0));
    };
    auto fetch_patch_piece = [&](int set, int it) {
        if constexpr (L3C_WINO_PROBE & 1) return;
        stage_regs[set][it] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(in_rsrc, patch_off[it], pf_cc * WCK * 4, 0));
    };
    auto store_pieces = [&](int par, const f32x4 (&regs)[NIT]) {
        float *dst = raw_buf(par);
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int i = tid + it * 256;
            const int c4 = i & 1, pix = i >> 1;
            if (i < N_PIECES) *reinterpret_cast<f32x4 *>(&dst[pix * PSR + c4 * 4]) = regs[it];
        }
    };
    // Input transform B^T d B (B^T = [1 0 -1 0; 0 1 1 0; 0 -1 1 0; 0 1 0 -1]).  Wavefront w = (tile row t_ty, output-row pair
    // t_h): lane = (tile column, channel pair) computes rows xi = 2 t_h, 2 t_h + 1 of its tile's transform -- these need the
    // three input rows t_h .. t_h + 2 -- for both channels: 8 tiles x 4 pairs per 32 lanes cover all 64 LDS banks on the
    // reads (pixel stride 12 floats) and on the writes (tile stride 8 floats).  The two 4-float groups of a tile are swapped
    // for tiles 8..15 of a row (XOR swizzle), which makes the MFMA A-fragment reads (16 tiles x 16 B per pass) conflict-free.
    const int t_tx = lane >> 2, t_cq = lane & 3, t_ty = wave & 1, t_h = wave >> 1;
    // rows xi = 2 t_h, 2 t_h + 1 of B^T d as  e0 = x - z,  e1 = s y + z  with (x, y, z; s) = (d0, d1, d2; +1) for t_h = 0 and
    // (d2, d3, d1; -1) for t_h = 1: ONE instruction stream for all four wavefronts, the difference is in the LDS addresses
    const int t_src = (2 * t_tx) * PSR + 2 * t_cq;                  // the lane's part of the source address (ONE register) ...
    const int t_row[3] = {2 * t_ty + 2 * t_h, 2 * t_ty + 1 + 2 * t_h, 2 * t_ty + 2 - t_h};   // ... and the wavefront's: patch rows
    const float t_s = t_h ? -1.0f : 1.0f;
    const int t_dst = (t_ty * 16 + t_tx) * PSV + (((t_cq >> 1) ^ (t_tx >> 3)) * 4) + (t_cq & 1) * 2 + (2 * t_h * 4) * N_TILES * PSV;
    f32x2 d[3][4], e[2][4];
    auto transform_load = [&](int par, int i) {   // i = 0, 1, 2: x, y, z
        const float *src = raw_buf(par) + t_src + t_row[i] * WP_W * PSR;   // (three per-lane addresses, hoisted out of the loop)
#pragma unroll
        for (int j = 0; j < 4; ++j) d[i][j] = *reinterpret_cast<const f32x2 *>(src + j * PSR);
    };
    auto transform_rows = [&](int k) {   // e[k] = row 2 t_h + k of B^T d
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (k == 0) {   // (component-wise on purpose, and the file is built with -fno-slp-vectorize: packed fp32 VALU beside MFMAs
                            // costs more issue time than two plain ones -- MI355X_MICROARCH.md, "price of one filler")
                e[0][j][0] = sub_scalar(d[0][j][0], d[2][j][0]);
                e[0][j][1] = sub_scalar(d[0][j][1], d[2][j][1]);
            } else {   // s = +-1: the product is exact
                e[1][j][0] = __builtin_fmaf(t_s, d[1][j][0], d[2][j][0]);
                e[1][j][1] = __builtin_fmaf(t_s, d[1][j][1], d[2][j][1]);
            }
        }
    };
    auto transform_cols = [&](int k) {   // e[k] <- e[k] B
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const float c0 = e[k][0][h] - e[k][2][h], c1 = e[k][1][h] + e[k][2][h], c2 = e[k][2][h] - e[k][1][h], c3 = e[k][1][h] - e[k][3][h];
            e[k][0][h] = c0;
            e[k][1][h] = c1;
            e[k][2][h] = c2;
            e[k][3][h] = c3;
        }
    };
    auto transform_write = [&](float *Vdst, int k) {   // finished row 2 t_h + k -> positions 4 (2 t_h + k) .. + 3
        float *dst = Vdst + t_dst + (k * 4) * N_TILES * PSV;
#pragma unroll
        for (int j = 0; j < 4; ++j) *reinterpret_cast<f32x2 *>(dst + j * N_TILES * PSV) = e[k][j];
    };

    f32x16 acc[8];   // never cleared: the first MFMA of every accumulator of a tile takes C = 0 ...
    // ... except the one of position (xi, nu) = (1, 1), which starts from the BIAS: A^T has a 1 in column 1 of both rows, so
    // M[1][1] enters all four outputs of a tile with weight 1 -- the bias rides through the output transform for free.  That
    // position is operand 3 of the ph = 0 wavefronts; D-layout: lane = output channel.
    const int co_lane = chunk_o * 64 + nj * 32 + lx;
    const float bias_init = (ph == 0 && co_lane < p.Cout) ? p.bias[co_lane] : 0.0f;
    auto init_bias = [&]() {   // (volatile: sixteen copies made here, per tile -- not hoisted into sixteen registers held forever)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            float v;
            asm volatile("v_mov_b32 %0, %1" : "=v"(v) : "v"(bias_init));
            acc[3][r] = v;
        }
    };

    // Prologue.  Invariants at the start of the MFMA loop of chunk g:  V[g & 1] complete and visible; bq = its B operands (in
    // flight); a0[0], a1[0] = the A fragments of its first position pair; raw[(g+1) & 1] = patch of chunk g + 1, visible; the
    // staging sets g & 1 and (g+1) & 1 = the patches of chunks g + 2 and g + 3 (in flight: a patch has TWO chunks' time to
    // arrive -- with one the loop stalled on HBM latency whenever a chunk touched new cache lines [timeline: the loop of a
    // tile whose prefetches were dummies ran in 24 k cycles, with real ones in 32 k]); the prefetch pointer is at chunk g + 4.
    // The number of chunks is even (Cin % 16 == 0), so g & 1 = cc & 1 is a compile-time constant of each chunk body.
    // ALL of the prologue's loads go out at once -- ONE memory round trip (the co-resident block can cover only so much).
    // The B operands AFTER the last patch fetch: the loads then are outstanding in the same order as at the top of every later
    // chunk (patch, then B), and the loop's waits -- which the compiler derives for the worst path -- fit the steady state.
    set_patch_tile(0);
    f32x4 first_regs[2][NIT];
#pragma unroll
    for (int k = 0; k < 2; ++k) {
#pragma unroll
        for (int it = 0; it < NIT; ++it)
            first_regs[k][it] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(in_rsrc, patch_off[it], pf_cc * WCK * 4, 0));
        pf_advance();
    }
#pragma unroll
    for (int k = 0; k < 2; ++k) {
#pragma unroll
        for (int it = 0; it < NIT; ++it) fetch_patch_piece(k, it);
        pf_advance();
    }
#pragma unroll
    for (int a = 0; a < 8; ++a) fetch_b(0, a);
    L3C_WINO_STAMP(1)
    store_pieces(0, first_regs[0]);
    store_pieces(1, first_regs[1]);
    L3C_WINO_STAMP(2)
    __syncthreads();
    L3C_WINO_STAMP(3)
#pragma unroll
    for (int i = 0; i < 3; ++i) transform_load(0, i);
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        transform_rows(k);
        transform_cols(k);
        transform_write(v_buf(0), k);
    }
    __syncthreads();

    // A fragment of (tile, half): the 4-float group `half`, swizzled like the writes; position stride N_TILES * PSV
    const int a_lane = lx * PSV + ((half ^ ((lx >> 3) & 1)) * 4) + (2 * ph) * N_TILES * PSV;
    auto a_pos = [&](int a) { return ((a >> 1) * 4 + (a & 1)) * N_TILES * PSV; };   // operand a of this wavefront
    // Issue is in order: the non-MFMA instructions are dealt out BETWEEN the MFMAs (each keeps the matrix pipe busy for 64
    // cycles; whatever does not fit is covered by the co-resident block's wavefront on the same SIMD).  Positions are taken
    // in pairs so that consecutive MFMAs alternate between two accumulators; the 8 gaps of a pair carry
    //   1, 2: the A fragment reads of the next pair
    //   3 .. 6: pair 0: store patch g + 2 (fetched during chunk g - 1), fetch patch g + 3, first LDS reads of the transform of
    //           chunk g + 1; pair 1: its remaining reads and the row stage; pair 2: the column stage and the writes to V
    //   7, 8: the B operands of the SAME positions for chunk g + 1 -- each register quad is reloaded right after its last
    //         MFMA of this chunk was issued, a whole chunk ahead of its next use
    // The chunk's ONE barrier sits between its pairs 2 and 3: by then this wave has written its share of V[(g+1) & 1] and
    // stored the patch of chunk g + 2, so the last pair's 8 MFMAs run while the barrier releases and the first fragments of
    // chunk g + 1 arrive.
    f32x4 a0[2], a1[2];
    L3C_WINO_STAMP(4)
    a0[0] = *reinterpret_cast<const f32x4 *>(v_buf(0) + a_lane + a_pos(0));
    a1[0] = *reinterpret_cast<const f32x4 *>(v_buf(0) + a_lane + a_pos(1));
    // Nothing of the prologue stays in flight: the tile loop is entered from here and from its own back edge, and the compiler
    // sizes the waits at a join for the path with the FEWEST operations behind the awaited load -- with the prologue's loads
    // still counted, the first chunk of every later tile would wait for the previous tile's output stores to be acknowledged.
    __builtin_amdgcn_s_waitcnt(0x0f70);   // vmcnt(0)
    auto chunk = [&](const int cc, auto first_c, auto par_c) __attribute__((always_inline)) {
        constexpr bool FIRST = decltype(first_c)::value;
        constexpr int par = decltype(par_c)::value;   // = cc & 1 = g & 1
        // No branches around loads: past the block's last chunk the pipeline fetches, stores and transforms data that nobody
        // reads -- cheaper than seventeen uniform branches per chunk, and it keeps the compiler's load counts exact.
        const int cc_b = cc + 1 == n_cc ? 0 : cc + 1;   // the next chunk's weights: the next tile starts over with chunk 0
        const float *a_cur = v_buf(par) + a_lane;
        const float *a_nxt = v_buf(par ^ 1) + a_lane;
        float *v_next = v_buf(par ^ 1);
        const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#define L3C_WINO_MFMA(Q, T, A, B)                                                                  \
    __builtin_amdgcn_sched_barrier(0);                                                             \
    acc[Q] = __builtin_amdgcn_mfma_f32_32x32x2f32((A)[T], (B)[T], (FIRST && (T) == 0 && (Q) != 3) ? zero16 : acc[Q], 0, 0, 0); \
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int pp = 0; pp < 4; ++pp) {
            const int q = 2 * pp, cb = pp & 1, nb = cb ^ 1;
            const f32x4 A0 = a0[cb], A1 = a1[cb], B0 = bq[q], B1 = bq[q + 1];
            if (pp == 3) {
                // everything chunk g + 1 needs from this wave is issued: patch g + 2 stored, V[(g+1) & 1] written -- the
                // barrier (LDS operations only: the loads in flight stay in flight), then the first fragments of chunk g + 1
                if constexpr (!(L3C_WINO_PROBE & 16)) __syncthreads();
                a0[nb] = *reinterpret_cast<const f32x4 *>(a_nxt + a_pos(0));
                a1[nb] = *reinterpret_cast<const f32x4 *>(a_nxt + a_pos(1));
            }
            L3C_WINO_MFMA(q, 0, A0, B0)
            if (pp < 3) a0[nb] = *reinterpret_cast<const f32x4 *>(a_cur + a_pos(q + 2));
            L3C_WINO_MFMA(q + 1, 0, A1, B1)
            if (pp < 3) a1[nb] = *reinterpret_cast<const f32x4 *>(a_cur + a_pos(q + 3));
            L3C_WINO_MFMA(q, 1, A0, B0)
            if (pp == 3) pf_advance();       // the prefetch pointer moves on (into the next tile: new column offsets); in a gap of its
                                             // own: its (rarely taken) branch ends a basic block, and work placed behind it bunches up there
            if (pp == 0 && !(L3C_WINO_PROBE & 2)) store_pieces(par, stage_regs[par]);   // patch g + 2 -> raw[g & 1] (its previous patch was transformed during g - 1)
            if (pp == 1 && !(L3C_WINO_PROBE & 4)) transform_load(par ^ 1, 1);
            if (pp == 2 && !(L3C_WINO_PROBE & 4)) transform_cols(0);
            L3C_WINO_MFMA(q + 1, 1, A1, B1)
            if (pp == 0) fetch_patch_piece(par, 0);   // patch g + 4 into the set just stored
            if (pp == 1 && !(L3C_WINO_PROBE & 4)) transform_rows(0);
            if (pp == 2 && !(L3C_WINO_PROBE & 4)) transform_write(v_next, 0);
            L3C_WINO_MFMA(q, 2, A0, B0)
            if (pp == 0 && !(L3C_WINO_PROBE & 4)) transform_load(par ^ 1, 0);
            if (pp == 1 && !(L3C_WINO_PROBE & 4)) transform_rows(1);
            if (pp == 2 && !(L3C_WINO_PROBE & 4)) transform_cols(1);
            L3C_WINO_MFMA(q + 1, 2, A1, B1)
            if (pp == 0) fetch_patch_piece(par, 1);
            if (pp == 0 && !(L3C_WINO_PROBE & 4)) transform_load(par ^ 1, 2);
            if (pp == 2 && !(L3C_WINO_PROBE & 4)) transform_write(v_next, 1);
            L3C_WINO_MFMA(q, 3, A0, B0)
            if constexpr (!(L3C_WINO_PROBE & 8)) fetch_b(cc_b, q);
            L3C_WINO_MFMA(q + 1, 3, A1, B1)
            if constexpr (!(L3C_WINO_PROBE & 8)) fetch_b(cc_b, q + 1);
        }
#undef L3C_WINO_MFMA
    };

    constexpr int S = SHUFFLE ? 2 : 1;
    constexpr int OOB = 0x7ffffff0;   // a byte offset beyond every buffer: the access is dropped by the range check
    // Output / residual addressing: ONE uniform descriptor at the block's first output row (a descriptor with a per-lane field
    // would cost a waterfall loop per access; based at the row, not at the image: the 32-bit offsets then span a few rows,
    // whatever the image size -- the 192-channel concat of a 2000 x 1500 image is 2.3 GB), one per-lane byte offset, and a
    // scalar offset per group of 8 pixels that carries the tile origin.  A lane whose channels do not exist, or -- in a tile that sticks out of the image -- whose pixel does not exist,
    // gets the out-of-range offset instead.  Pixel shuffle (dil = 1): conv pixel (oy, ox), channel co -> pixel
    // (2 oy + (co >> 1 & 1), 2 ox + (co & 1)), channel co >> 2 of a 2H x 2W image: the same walk with doubled strides, the
    // sub-pixel in the lane offset.
    const int col_b = S * dil * p.out_cstride * 4, row_b = S * dil * (S * p.W) * p.out_cstride * 4;   // one conv pixel / row on
    const int rcol_b = dil * p.res_cstride * 4, rrow_b = dil * p.W * p.res_cstride * 4;
    const auto o_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        p.out + (((size_t)b * (S * p.H) + S * (py + dil * sy0)) * (S * p.W) + S * px) * p.out_cstride + p.out_coff +
            (SHUFFLE ? chunk_o * 16 : chunk_o * 64), 0, OOB, 0x00020000);
    const auto r_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float *>(RES ? p.res + (((size_t)b * p.H + py + dil * sy0) * p.W + px) * p.res_cstride + p.res_coff + chunk_o * 64 : p.bias),
        0, RES ? OOB : 0, 0x00020000);
    const bool rows_in = py + dil * (sy0 + WT_H - 1) < p.H;

    for (int t = 0; t < n_t; ++t) {
        L3C_WINO_STAMP(5)
        init_bias();
        chunk(0, std::true_type{}, std::integral_constant<int, 0>{});
        chunk(1, std::false_type{}, std::integral_constant<int, 1>{});
        for (int cc = 2; cc < n_cc; cc += 2) {
            chunk(cc, std::false_type{}, std::integral_constant<int, 0>{});
            chunk(cc + 1, std::false_type{}, std::integral_constant<int, 1>{});
        }

        // ---- output transform Y = A^T M A (A^T = [1 1 1 0; 0 1 -1 -1]).  This wavefront holds columns nu = 2 ph, 2 ph + 1 of
        // M: with t_i[n] = sum_xi A^T[i][xi] M[xi][2 ph + n] its share of Y[i][j] is  sum_n t_i[n] A^T[j][2 ph + n]:
        //   ph = 0:  P[i][0] = t_i[0] + t_i[1],  P[i][1] = t_i[1];        ph = 1:  P[i][0] = t_i[0],  P[i][1] = -t_i[0] - t_i[1].
        // D register r of a fragment belongs to tile (ty, tx) = (r >> 3, (r & 3) + 8 ((r >> 2) & 1) + 4 half), channel lx.
        // Wavefront (nj, ph) FINISHES tile row ph (output rows 2 ph, 2 ph + 1) of its 32 channels: it hands its share of the
        // OTHER tile row to its partner (nj, 1 - ph) through LDS as [pixel][channel], adds the partner's share of its own row
        // in place, and reads the sums back in the STORE layout -- four values adjacent in memory per lane -- so that the results
        // leave as 8 x 16-byte stores per lane (full 128-byte lines): the store instructions, not the bytes, are what the
        // epilogue waits for.  The exchange (2 x 16 KB) lives in the LDS run the pipeline has just released (see the LDS map).
        L3C_WINO_STAMP(6)
        // Everything the epilogue derives from the lane index is recomputed for every tile (the empty asm hides the index from
        // the loop-invariant code motion): held across the chunk loop these values would push the accumulators out of the
        // register file.
        const int ln = fresh_lane();
        const int half = ln >> 5, lx = ln & 31;
        // Store layout: a lane ends up with FOUR values that are adjacent in memory -- plain: pixel e_pl of a group of 8,
        // channels 4 e_q .. 4 e_q + 3 of the wavefront's 32; pixel shuffle: pixel e_pl, sub-pixel e_s, and the four conv
        // channels 16 e_g + 4 j + e_s (j = 0..3), which are adjacent OUTPUT channels of that sub-pixel.
        const int e_pl = ln >> 3, e_q = ln & 7, e_s = (ln >> 1) & 3, e_g = ln & 1;
        auto e_cw = [&](int j) { return SHUFFLE ? 16 * e_g + 4 * j + e_s : 4 * e_q + j; };   // channel inside the wavefront's 32
        const bool lane_ok = chunk_o * 64 + nj * 32 + e_cw(0) < p.Cout;   // Cout % 4 (pixel shuffle: % 16) == 0: all four or none
        const int o_lane = !lane_ok ? OOB : 2 * ph * row_b + e_pl * col_b +
                           (SHUFFLE ? (((e_s >> 1) * (2 * p.W) + (e_s & 1)) * p.out_cstride + nj * 8 + 4 * e_g) * 4 : (nj * 32 + 4 * e_q) * 4);
        const int r_lane = lane_ok ? 2 * ph * rrow_b + e_pl * rcol_b + (nj * 32 + 4 * e_q) * 4 : OOB;
        const int oy_l = py + dil * (sy0 + 2 * ph), ox_l = px + dil * e_pl;   // this lane's first pixel (ox: + dil sx0)
        const int sx0 = (tx_first + t) * WT_W;
        const bool interior = rows_in && px + dil * (sx0 + WT_W - 1) < p.W;
        // per-lane offsets of the k-th group of 8 pixels (row k >> 2, columns 8 (k & 3) .. + 7 of the tile row)
        auto lane_off = [&](int base, int k) {
            const bool ok = interior | ((oy_l + dil * (k >> 2) < p.H) & (ox_l + dil * (sx0 + (k & 3) * 8) < p.W));
            return ok ? base : OOB;
        };
        float *Yn = lds + (LDS_FLOATS - 2 * Y_FLOATS) + nj * Y_FLOATS;   // [128 pixels][32 channels] of this nj (the last chunk of a tile is odd)
        float *Yw = Yn + half * 8 * 32 + lx;
        float own[8][2][2];   // this wavefront's share of its own tile row (D registers r = 8 ph .. 8 ph + 7)
        auto hand_over = [&](auto ph_c) __attribute__((always_inline)) {
            constexpr int PH = decltype(ph_c)::value;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float tt[2][2], P[2][2];
#pragma unroll
                for (int n = 0; n < 2; ++n) {
                    tt[0][n] = (acc[0 + n][r] + acc[2 + n][r]) + acc[4 + n][r];
                    tt[1][n] = (acc[2 + n][r] - acc[4 + n][r]) - acc[6 + n][r];
                }
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    P[i][0] = PH == 0 ? tt[i][0] + tt[i][1] : tt[i][0];
                    P[i][1] = PH == 0 ? tt[i][1] : (-tt[i][0]) - tt[i][1];
                }
                const int ty = r >> 3, tx = (r & 3) + 8 * ((r >> 2) & 1);   // + 4 half: in Yw
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        if (ty == PH) own[r & 7][i][j] = P[i][j];
                        else Yw[((2 * ty + i) * WT_W + 2 * tx + j) * 32] = P[i][j];
                    }
                __builtin_amdgcn_sched_barrier(0);   // in this order: every step frees eight accumulator registers
            }
        };
        if (ph == 0) hand_over(std::integral_constant<int, 0>{});
        else hand_over(std::integral_constant<int, 1>{});
        f32x4 resv[8];
        if constexpr (RES) {   // the residual values in the store layout (the accumulators are dead: they take their registers)
            const int r_tile = sx0 * rcol_b;
#pragma unroll
            for (int k = 0; k < 8; ++k)
                resv[k] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(
                                                        r_rsrc, lane_off(r_lane, k), r_tile + (k >> 2) * rrow_b + (k & 3) * 8 * rcol_b, 0));
        }
        L3C_WINO_STAMP(7)
        __syncthreads();
        L3C_WINO_STAMP(8)
        {
            float *Yo = Yw + 2 * ph * WT_W * 32;   // this wavefront's own tile row
            float theirs[8][2][2];
#pragma unroll
            for (int r8 = 0; r8 < 8; ++r8)
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) theirs[r8][i][j] = Yo[(i * WT_W + 2 * ((r8 & 3) + 8 * ((r8 >> 2) & 1)) + j) * 32];
#pragma unroll
            for (int r8 = 0; r8 < 8; ++r8)
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        Yo[(i * WT_W + 2 * ((r8 & 3) + 8 * ((r8 >> 2) & 1)) + j) * 32] = theirs[r8][i][j] + own[r8][i][j];
        }
        // the sums of this wavefront's tile row were written by this wavefront: no block barrier, the wavefront's LDS operations
        // execute in order (the fence keeps the compiler from moving the reads of other lanes' values above the writes)
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        // pixel k * 8 + e_pl of tile row ph (64 pixels: two image rows of 32)
        const float *Yr = Yn + (2 * ph * WT_W + e_pl) * 32;
        f32x4 v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const float *y0 = Yr + k * 8 * 32;
            if constexpr (SHUFFLE) v[k] = f32x4{y0[e_cw(0)], y0[e_cw(1)], y0[e_cw(2)], y0[e_cw(3)]};
            else v[k] = *reinterpret_cast<const f32x4 *>(y0 + 4 * e_q);
        }
        L3C_WINO_STAMP(9)
        // the next tile's pipeline writes into the exchange's LDS: every wavefront must have read its sums first
        if (t + 1 < n_t) __syncthreads();
        int o_off[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if constexpr (RELU) v[k][j] = fmaxf(v[k][j], 0.0f);
            if constexpr (RES) v[k] = v[k] + resv[k];
            o_off[k] = lane_off(o_lane, k);
        }
        // ALL eight results first, then the eight stores back to back, each from its own registers: on gfx950 a VALU write to
        // the data registers of a 16-byte buffer store that was issued just before (register soffset: the compiler sees no
        // hazard and inserts no wait states) can overtake the store's read of its last dwords -- measured: the 4th dword
        // of the last lanes of each 16-lane group came out as the NEXT store's value.
        const int o_tile = sx0 * col_b;
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int k = 0; k < ((L3C_WINO_PROBE & 32) ? 1 : 8); ++k)
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(__attribute__((__vector_size__(4 * sizeof(unsigned)))) unsigned,
                                                                      (L3C_WINO_PROBE & 32) ? ((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7])) : v[k]),
                                                   o_rsrc, o_off[k], o_tile + (k >> 2) * row_b + (k & 3) * 8 * col_b, 0);
        __builtin_amdgcn_sched_barrier(0);
        // the first A fragments of the next tile once more (V[0] has been complete since the last chunk's barrier): read
        // again here, they need no registers during the output transform
        a0[0] = *reinterpret_cast<const f32x4 *>(v_buf(0) + a_lane + a_pos(0));
        a1[0] = *reinterpret_cast<const f32x4 *>(v_buf(0) + a_lane + a_pos(1));
    }
#ifdef L3C_WINO_TIMELINE
    if (p.dbg && lane == 0) {   // stamps of the block's first prologue and LAST tile (tools/wino_timeline.py)
        unsigned long long *o = p.dbg + ((size_t)blockIdx.x * 4 + wave) * 16;
        dbg_t[10] = __builtin_amdgcn_s_memtime();
        o[12] = __builtin_amdgcn_s_getreg((16 - 1) << 11 | 0 << 6 | 4) | (__builtin_amdgcn_s_getreg((4 - 1) << 11 | 0 << 6 | 20) << 16);
        o[13] = blockIdx.x;
        o[14] = (unsigned long long)n_t;
#pragma unroll
        for (int i = 0; i < 11; ++i) o[i] = dbg_t[i];
    }
#endif
}

// OIHW 3x3 weights -> U = G g G^T (G = [1 0 0; .5 .5 .5; .5 -.5 .5; 0 0 1]) in the kernel's slab order
// [Cout/64][Cin/8][16 positions][2 n-tiles][64 lanes][4]: lane (n = lane % 32, half = lane / 32), element t holds
// U[position][co = chunk * 64 + n_tile * 32 + n][ci = cc * 8 + half * 4 + t].
__global__ __launch_bounds__(256) void pack_wino_kernel(const float *__restrict__ w, int Cout, int Cin, float *__restrict__ packed,
                                                        int64_t total) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        int64_t r = i;
        const int t = r % 4;  r /= 4;
        const int lane = r % 64;  r /= 64;
        const int ntile = r % 2;  r /= 2;
        const int pos = r % 16;  r /= 16;
        const int cc = r % (Cin / 8);  r /= (Cin / 8);
        const int chunk = (int)r;
        const int co = chunk * 64 + ntile * 32 + (lane & 31);
        const int ci = cc * 8 + (lane >> 5) * 4 + t;
        float u = 0.0f;
        if (co < Cout) {
            const float *g = w + ((size_t)co * Cin + ci) * 9;
            const int xi = pos >> 2, nu = pos & 3;
            float gg[3];   // row xi of G g
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                const float g0 = g[0 * 3 + j], g1 = g[1 * 3 + j], g2 = g[2 * 3 + j];
                gg[j] = xi == 0 ? g0 : xi == 3 ? g2 : xi == 1 ? ((g0 + g1) + g2) * 0.5f : ((g0 - g1) + g2) * 0.5f;
            }
            u = nu == 0 ? gg[0] : nu == 3 ? gg[2] : nu == 1 ? ((gg[0] + gg[1]) + gg[2]) * 0.5f : ((gg[0] - gg[1]) + gg[2]) * 0.5f;
        }
        packed[i] = u;
    }
}

}  // namespace

#ifdef L3C_WINO_TIMELINE
static unsigned long long *g_wino_dbg = nullptr;
extern "C" void l3c_conv_wino_set_debug(void *ptr) { g_wino_dbg = (unsigned long long *)ptr; }
#endif

static std::atomic<int> g_wino_tpb{getenv("L3C_WINO_TPB") ? atoi(getenv("L3C_WINO_TPB")) : 0};
// fewest blocks a launch may be cut down to by walking several tiles per block: 8 per block slot of a 256-CU part (development:
// L3C_WINO_MIN_BLOCKS)
static const long long g_wino_min_blocks = getenv("L3C_WINO_MIN_BLOCKS") ? atoll(getenv("L3C_WINO_MIN_BLOCKS")) : 8 * 512;

extern "C" {

int l3c_conv_wino_set_tiles_per_block(int n) { return g_wino_tpb.exchange(n < 0 ? 0 : n > 64 ? 64 : n); }

int64_t l3c_conv_wino_packed_words(int Cout, int Cin) { return (int64_t)((Cout + 63) / 64) * (Cin / 8) * U_FLOATS; }

int l3c_conv_wino_pack_weights(const float *w_oihw, int Cout, int Cin, float *packed, l3c_stream_t stream) {
    L3C_REQUIRE(w_oihw && packed, "null pointer");
    L3C_REQUIRE(Cout > 0 && Cin > 0 && Cin % 16 == 0, "Cin must be a multiple of 16");
    const int64_t total = l3c_conv_wino_packed_words(Cout, Cin);
    int64_t g = (total + 255) / 256;
    hipLaunchKernelGGL(pack_wino_kernel, dim3((unsigned)(g > 4096 ? 4096 : g)), dim3(256), 0, l3c::as_stream(stream), w_oihw,
                       Cout, Cin, packed, total);
    return l3c::check_launch("pack_wino_kernel");
}

int l3c_conv_wino(const l3c_conv_desc *d, l3c_stream_t stream) {
    L3C_REQUIRE(d, "null descriptor");
    L3C_REQUIRE(d->in && d->packed_w && d->bias && d->out, "null pointer in descriptor");
    L3C_REQUIRE(d->KS == 3 && d->stride == 1, "Winograd F(2x2,3x3): 3x3, stride 1 only");
    L3C_REQUIRE(d->dilation == 1 || d->dilation == 2 || d->dilation == 4, "dilation must be 1, 2 or 4");
    L3C_REQUIRE(d->dilation == 1 || !(d->epilogue & L3C_EPI_PIXEL_SHUFFLE), "pixel shuffle with dilation not provided");
    L3C_REQUIRE(d->B > 0 && d->Hin > 0 && d->Win > 0 && d->Cout > 0, "bad shape");
    L3C_REQUIRE(d->Cin > 0 && d->Cin % (2 * WCK) == 0, "Cin must be a multiple of 16 (an even number of 8-channel chunks)");
    L3C_REQUIRE(d->in_cstride % 4 == 0 && d->in_coff % 4 == 0, "input channel stride/offset must be multiples of 4");
    L3C_REQUIRE(d->out_cstride % 4 == 0 && d->out_coff % 4 == 0, "output channel stride/offset must be multiples of 4 (16-byte stores)");
    L3C_REQUIRE(!(d->epilogue & L3C_EPI_RESIDUAL) || (d->res_cstride % 4 == 0 && d->res_coff % 4 == 0),
                "residual channel stride/offset must be multiples of 4 (16-byte loads)");
    L3C_REQUIRE(((uintptr_t)d->in | (uintptr_t)d->out | (uintptr_t)d->packed_w | ((d->epilogue & L3C_EPI_RESIDUAL) ? (uintptr_t)d->residual : 0)) % 16 == 0,
                "input, output, residual and packed weights must be 16-byte aligned");
    L3C_REQUIRE(d->Cout % ((d->epilogue & L3C_EPI_PIXEL_SHUFFLE) ? 16 : 4) == 0,
                "Cout must be a multiple of 4 (pixel shuffle: 16): a lane stores four adjacent channels");
    L3C_REQUIRE((d->epilogue & ~(L3C_EPI_RELU | L3C_EPI_RESIDUAL | L3C_EPI_PIXEL_SHUFFLE)) == 0, "unknown epilogue bits");
    L3C_REQUIRE(d->in_coff + d->Cin <= d->in_cstride, "input channel slice out of range");
    // (the epilogue stores / loads through descriptors whose range is the constant OOB: the hardware check does not bound a slice)
    L3C_REQUIRE(d->out_coff + ((d->epilogue & L3C_EPI_PIXEL_SHUFFLE) ? d->Cout / 4 : d->Cout) <= d->out_cstride, "output channel slice out of range");
    L3C_REQUIRE(!(d->epilogue & L3C_EPI_RESIDUAL) || d->res_coff + d->Cout <= d->res_cstride, "residual channel slice out of range");
    L3C_REQUIRE(!(d->epilogue & L3C_EPI_RESIDUAL) || d->residual, "residual epilogue without residual pointer");
    L3C_REQUIRE(!((d->epilogue & L3C_EPI_PIXEL_SHUFFLE) && (d->epilogue & L3C_EPI_RESIDUAL)), "pixel shuffle + residual not provided");
    WinoParams p{};
    p.in = d->in;  p.u = d->packed_w;  p.bias = d->bias;
    p.res = (d->epilogue & L3C_EPI_RESIDUAL) ? d->residual : nullptr;
    p.out = d->out;
    p.in_cstride = d->in_cstride;  p.in_coff = d->in_coff;
    p.res_cstride = d->res_cstride;  p.res_coff = d->res_coff;
    p.out_cstride = d->out_cstride;  p.out_coff = d->out_coff;
    p.B = d->B;  p.H = d->Hin;  p.W = d->Win;  p.Cin = d->Cin;  p.Cout = d->Cout;
    p.epilogue = d->epilogue;
    p.dil = d->dilation;
    p.dil_log2 = d->dilation == 4 ? 2 : d->dilation == 2 ? 1 : 0;
    p.tiles_x = ((p.W + p.dil - 1) / p.dil + WT_W - 1) / WT_W;   // tiles of the (largest) sub-grid
    p.tiles_y = ((p.H + p.dil - 1) / p.dil + WT_H - 1) / WT_H;
    p.n_chunks_o = (p.Cout + 63) / 64;
    // tiles per block: as many as leave every one of the 2 x CUs block slots several blocks to run (the first tile of a block
    // pays the prologue's memory round trip, the others do not); L3C_WINO_TPB overrides (development)
    const int64_t rows = (int64_t)p.tiles_y * p.dil * p.dil * p.n_chunks_o * p.B;
    const int tpb_set = g_wino_tpb.load(std::memory_order_relaxed);
    int tpb = tpb_set > 0 ? tpb_set : WINO_TPB_MAX;
    if (tpb > p.tiles_x) tpb = p.tiles_x;
    if (tpb_set <= 0) {
        // the fewest groups per tile row that still give the launch enough blocks, then EVEN groups: a row of 6 tiles cut 5 + 1
        // (what counting tpb down by one arrives at) leaves half the blocks with five times the work of the others
        // [measured: batch 64 ran 4 % below its neighbours for that reason]
        int g = (p.tiles_x + WINO_TPB_MAX - 1) / WINO_TPB_MAX;
        while (g < p.tiles_x && rows * g < g_wino_min_blocks) ++g;
        tpb = (p.tiles_x + g - 1) / g;
    }
    p.tpb = tpb;
    p.groups_x = (p.tiles_x + tpb - 1) / tpb;
    const int64_t total = rows * p.groups_x;
    L3C_REQUIRE(total < (1ll << 31), "grid too large");
    // 32-bit byte offsets (buffer addressing): inside one image of the input; inside the 8 x dilation rows of a block's output /
    // residual tile row
    const int64_t S2 = (d->epilogue & L3C_EPI_PIXEL_SHUFFLE) ? 4 : 1;
    L3C_REQUIRE((int64_t)p.H * p.W * p.in_cstride * 4 < 0x7ffffff0ll, "one image of the input must stay below 2 GB");
    L3C_REQUIRE(S2 * 8 * p.dil * (p.W + 64 * p.dil) * p.out_cstride * 4 < 0x7ffffff0ll && 8ll * p.dil * (p.W + 64 * p.dil) * p.res_cstride * 4 < 0x7ffffff0ll,
                "image too wide for 32-bit offsets inside a tile row");
    p.total_blocks = (int)total;
    p.div_groups = wino_div((unsigned)(p.groups_x * p.tiles_y));
    p.div_groups_x = wino_div((unsigned)p.groups_x);
    p.div_chunks = wino_div((unsigned)p.n_chunks_o);
#ifdef L3C_WINO_TIMELINE
    p.dbg = g_wino_dbg;
#endif
    typedef void (*kernel_t)(const WinoParams);
    static const kernel_t variants[5] = {conv_wino_kernel<false, false, false>, conv_wino_kernel<true, false, false>,
                                         conv_wino_kernel<false, true, false>, conv_wino_kernel<true, true, false>,
                                         conv_wino_kernel<false, false, true>};
    static bool attr_set[64][5] = {};   // 64 KB of dynamic LDS needs the opt-in, per device
    int dev = 0;
    {
        const int rc = l3c::check_hip(hipGetDevice(&dev), "hipGetDevice");
        if (rc != L3C_OK) return rc;
    }
    L3C_REQUIRE(dev >= 0 && dev < 64, "device index out of range");
    const bool relu = d->epilogue & L3C_EPI_RELU, res = d->epilogue & L3C_EPI_RESIDUAL, shuffle = d->epilogue & L3C_EPI_PIXEL_SHUFFLE;
    L3C_REQUIRE(!(shuffle && relu), "pixel shuffle + ReLU not provided");
    const int v = shuffle ? 4 : (relu ? 1 : 0) + (res ? 2 : 0);
    if (!attr_set[dev][v]) {
        const int rc = l3c::check_hip(hipFuncSetAttribute(reinterpret_cast<const void *>(variants[v]),
                                                          hipFuncAttributeMaxDynamicSharedMemorySize, WINO_LDS_BYTES),
                                      "hipFuncSetAttribute");
        if (rc != L3C_OK) return rc;
        attr_set[dev][v] = true;
    }
    hipLaunchKernelGGL(variants[v], dim3((unsigned)total), dim3(256), WINO_LDS_BYTES, l3c::as_stream(stream), p);
    return l3c::check_launch("conv_wino_kernel");
}
}
