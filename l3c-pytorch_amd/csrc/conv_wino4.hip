// conv_wino4.hip -- 3x3 / stride 1 convolution (dilation 1, 2, 4) by Winograd F(4x4, 3x3) on the fp32 MFMA.
//
// 4x fewer multiplications than the direct form and 1.78x fewer than F(2x2,3x3) (conv_wino.hip): every 4x4 output tile is
//     Y = A^T [ sum_ci (G g G^T) (.) (B^T d B) ] A,        d the 6x6 input tile, interpolation points {0, 1, -1, 1/2, -2, inf}
//     (the set with the smallest fp32 error for F(4,3): half that of the textbook {0, +-1, +-2, inf}, measured on the whole forward:
//     profiles/r03_wino_f43_numerics.log; the transforms cost ~1.7x the additions, which this kernel has room for):
//     B^T = [2 -3 -4 3 2 0; 0 -2 1 5 2 0; 0 -2 5 -1 -2 0; 0 2 1 -2 -1 0; 0 1 -2 -1 2 0; 0 2 -3 -4 3 2]
//     G   = [1/2 0 0; 1/6 1/6 1/6; 1/6 -1/6 1/6; 16/15 8/15 4/15; 1/30 -1/15 2/15; 0 0 1/2]
//     A^T = [1 1 1 1 1 0; 0 1 -1 1/2 -2 0; 0 1 1 1/4 4 0; 0 1 -1 1/8 -8 1]
// The sum over input channels is, for each of the 36 positions (xi, nu) of the transformed tile, a GEMM
//     M[xi,nu] (tiles x Cout) = V[xi,nu] (tiles x Cin) * U[xi,nu] (Cin x Cout)
// on v_mfma_f32_16x16x4_f32.  fp32 throughout; the error against the direct convolution was measured BEFORE the kernel was
// written (tests/numerics/wino_f43_numerics.py -> profiles/r03_wino_f43_numerics.log: the whole L3C forward with this algorithm
// in fp32 stays within 5e-6 of the fp32 oracle relative to each tensor's largest magnitude; north_star allows 1e-5).
//
//   * block = 256 threads = 4 wavefronts on a 16 x 16 output tile = 4 x 4 Winograd tiles, 64 output channels.  Wavefront w
//     owns ALL 36 positions of the 16 tiles for 16 output channels: 36 accumulator fragments of 4 registers = 144 registers
//     (D layout of the 16x16 MFMA: lane = output channel, lane / 16 = tile row, register = tile column).  Every lane therefore
//     holds complete transformed tiles and the output transform A^T M A runs in registers: no exchange, no barrier.
//     <= 256 registers and 68 KB of LDS per block: TWO blocks per CU, the matrix pipe runs one block's MFMAs while the other
//     is in its prologue, at its chunk barrier or in its output transform (the lesson of conv_wino.hip);
//   * input channels in chunks of 8 = two k-steps of the MFMA.  The raw 18 x 18 patch of chunk g + 3 is being fetched into
//     registers (buffer descriptor over the image: zero padding = the range check), the patch of g + 2 stored to LDS, the
//     one of g + 1 transformed by all 256 threads (thread = tile x channel x half of the rows xi; bank-conflict-free) into
//     the other V buffer WHILE the 72 MFMAs of chunk g run; ONE barrier per chunk;
//   * V in LDS as [position pair][tile][k group][position of the pair][2 channels]: one ds_read_b128 per lane feeds the four
//     MFMAs of a position pair; the pre-transformed weights U (l3c_conv_wino4_pack_weights, the same fragment order) never
//     touch LDS: one 16-byte buffer load per lane and position pair straight into the B operands, four pairs ahead;
//   * the bias rides in the accumulator of position (1, 1): column 1 of A^T is all ones, so M[1][1] enters all 16 outputs of
//     a tile with weight 1;
//   * a block walks up to 6 horizontally adjacent tiles over ONE chunk pipeline (the fetches the last chunks of a tile issue
//     are the first patches of the next tile).
// A dilated conv is the dense conv on each of the dil x dil interleaved sub-grids of the image (strided indexing).
#include "l3c_common.h"

#include <stdlib.h>

#include <atomic>
#include <type_traits>

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

// n / d for 0 <= n < 2^31 as a multiplication (Granlund-Montgomery)
struct W4Div {
    unsigned m, sh;   // d == 1: m = 0
    __device__ __forceinline__ unsigned div(unsigned n) const { return m ? __umulhi(n, m) >> sh : n; }
};
static W4Div w4_div(unsigned d) {
    W4Div r{0, 0};
    if (d <= 1) return r;
    unsigned L = 0;
    while ((1ull << L) < d) ++L;
    r.m = (unsigned)(((1ull << (31 + L)) / d) + 1);
    r.sh = L - 1;
    return r;
}

struct Wino4Params {
    const float *in;
    const float *u;
    const float *bias;
    const float *res;
    float *out;
    int in_cstride, in_coff, res_cstride, res_coff, out_cstride, out_coff;
    int B, H, W, Cin, Cout;
    int dil, dil_log2;
    int Ho, Wo;                // output image (== H, W except in the polyphase form)
    int poly, in_py, in_px;    // polyphase form of a stride-2 convolution: the input is the sub-grid (in_py, in_px) of step 2, the output dense
    int poly_ncp;              // > 0: ALL FOUR phases in this launch -- the chunk sequence runs through the phases, poly_ncp chunks each
    W4Div div_groups, div_groups_x, div_chunks;
    int tiles_x, tiles_y, n_chunks_o, total_blocks;
    int tpb, groups_x;
    unsigned long long *dbg;   // development (-DL3C_W4_TIMELINE): per-wavefront s_memtime stamps, nullptr otherwise
};

constexpr int OT = 16;                          // output tile of a block: 4 x 4 Winograd tiles of 4 x 4 pixels
constexpr int PW = OT + 2;                      // input patch side
constexpr int CK = 8;                           // input channels per chunk
constexpr int NPP = 18;                         // position pairs: pp = xi * 3 + nu / 2
constexpr int PSR = 12;                         // LDS stride of a raw patch pixel (floats): 4 * PSR * tx covers distinct banks
constexpr int RAW_FLOATS = PW * PW * PSR;       // 3888
constexpr int VPP = 16 * 16;                    // floats of one position pair in V: [16 tiles][4 k groups][2 positions][2 channels]
constexpr int V_FLOATS = NPP * VPP;             // 4608
constexpr int U_CHUNK_FLOATS = NPP * 4 * 64 * 4;   // packed weights of one input chunk x one 64-channel output chunk
constexpr int V_OFF0 = 0, V_OFF1 = V_FLOATS, RAW_OFF0 = 2 * V_FLOATS, RAW_OFF1 = 2 * V_FLOATS + RAW_FLOATS;
constexpr int LDS_MAIN_FLOATS = 2 * V_FLOATS + 2 * RAW_FLOATS;
constexpr int WIN_ROW_FLOATS = 4 * 4 * 16;      // one output row of the four tiles of a tile column: [4 tile rows q][4 pixels j][16 channels]
// The output window of a wavefront (up to 4 rows = 4 KB) lives in V[1]: the last chunk of a tile reads V[1] and every wavefront has issued
// its last read of it before the chunk barrier, the next tile's first chunk transforms into it only after the barrier that ends the
// output transform -- no LDS of its own (round 3: 10 KB per block for the windows).
static_assert(4 * 4 * WIN_ROW_FLOATS <= V_FLOATS, "the four wavefronts' windows must fit V[1]");
constexpr int LDS_FLOATS = LDS_MAIN_FLOATS;
constexpr int W4_LDS_BYTES = LDS_FLOATS * 4;    // 67 968: two blocks per CU
constexpr int W4_TPB_MAX = 6;
constexpr int OOB = 0x7ffffff0;                 // a byte offset beyond every buffer: the access is dropped by the range check
constexpr int N_PIECES = PW * PW * 2;           // 16-byte pieces of a patch (a pixel's 8 channels = 2 pieces)
constexpr int NIT = (N_PIECES + 255) / 256;     // 3

__device__ __forceinline__ int xcd_remap4(int bid, int total) {
    const int q = total >> 3, r = total & 7;
    const int xcd = bid & 7, slot = bid >> 3;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + slot;
}

// one 6-vector through B^T (rows of the input transform), all six outputs
__device__ __forceinline__ void bt6(const float (&w)[6], float (&o)[6]) {
    o[0] = __builtin_fmaf(2.0f, w[0] + w[4], __builtin_fmaf(3.0f, w[3] - w[1], -4.0f * w[2]));
    o[1] = __builtin_fmaf(2.0f, w[4], __builtin_fmaf(5.0f, w[3], __builtin_fmaf(-2.0f, w[1], w[2])));
    o[2] = __builtin_fmaf(-2.0f, w[4], __builtin_fmaf(-2.0f, w[1], __builtin_fmaf(5.0f, w[2], -w[3])));
    const float p = w[1] - w[3], q = w[2] - w[4];
    o[3] = __builtin_fmaf(2.0f, p, q);
    o[4] = __builtin_fmaf(-2.0f, q, p);
    o[5] = __builtin_fmaf(2.0f, w[1] + w[5], __builtin_fmaf(3.0f, w[4] - w[2], -4.0f * w[3]));
}

// one 6-vector through A^T (output transform), four outputs
__device__ __forceinline__ void at6(float m0, float m1, float m2, float m3, float m4, float m5, float (&y)[4]) {
    const float s1 = m1 + m2, d1 = m1 - m2;
    y[0] = (m0 + s1) + (m3 + m4);
    y[1] = d1 + __builtin_fmaf(-2.0f, m4, 0.5f * m3);
    y[2] = s1 + __builtin_fmaf(4.0f, m4, 0.25f * m3);
    y[3] = (d1 + __builtin_fmaf(-8.0f, m4, 0.125f * m3)) + m5;
}

// Development probes (csrc/build.py --variant NAME "-DL3C_W4_PROBE=N"; results are WRONG, only the time means something): bit 0 no
// patch fetch, 1 no input transform, 2 no weight loads inside the loop, 3 no chunk barrier, 4 no output transform / stores (one
// store per tile keeps the accumulators alive), 5 no patch stores to LDS, 6 no MFMA (one v_add in its place: the data-movement
// skeleton alone), 7 all weight loads from the same 4 KB (L1 hits: the instruction stream without the L2 traffic), 8 the input addressed
// as [C/8][H][W][8], 9 every block fetches the same patch pixels (the fetch instructions without their HBM / L2 latency), 10 half the weight
// loads (what a kernel with 32 tiles per weight fragment would issue).  Never defined in the product build.
#ifndef L3C_W4_PROBE
#define L3C_W4_PROBE 0
#endif
// Schedule knob (A/B builds): TR_LEAD = pairs between the LDS reads of a raw patch column and the arithmetic that consumes them (0: the
// s_waitcnt lgkmcnt(0) in front of the column arithmetic sits one MFMA behind its ds_reads; 1: a whole pair, the waits become
// lgkmcnt(4..5)).  [measured, 64->64 at 256x384x32: 0.768 / 0.766 ms -- the LDS latency was not what idles the matrix pipe; likewise
// s_setprio 1 inside the chunk loop and 0 in the output transform, or the other way round: 0.758 / 0.766 ms, within the noise, not kept]
#ifndef L3C_W4_TR_LEAD
#define L3C_W4_TR_LEAD 1
#endif
// Depth of the weight ring = its lookahead in position pairs (4 or 6; a slot is a register quad).
#ifndef L3C_W4_RING
#define L3C_W4_RING 6
#endif
// Pair at which a chunk stores the staged patch to LDS and fetches the next one (16: right behind the weight burst, see the chunk lambda).
// Residual quads in flight in the output transform (round k's residual is loaded RES_RING - 1 rounds before its store).
#ifndef L3C_W4_RES_RING
#define L3C_W4_RES_RING 4
#endif
// Cache-policy bits of the output stores / the residual loads (A/B builds; [measured, 64->64 at 256x384x32, relu / residual variant:
// 0 / 0: 0.743 / 0.858 ms; stores nt: 0.762 / 0.873; residual nt: 0.746 / 0.871; both: 0.761 / 0.889; stores sc0: 0.751 / 0.847] -- left at 0)
#ifndef L3C_W4_OUT_AUX
#define L3C_W4_OUT_AUX 0
#endif
#ifndef L3C_W4_RES_AUX
#define L3C_W4_RES_AUX 0
#endif
#ifndef L3C_W4_FETCH_PP
#define L3C_W4_FETCH_PP 16
#endif
// Output rows of a tile column that leave through the LDS window in ONE round (1, 2 or 4; round 4).  Round 3 handed one row at a time through
// a double-buffered 1.25 KB window: 16 rounds per tile of (4 ds_write_b32, wait, ds_read_b128, wait, store) whose LDS round trips a
// wavefront could only cover with the next row's dozen VALU instructions -- the time line (profiles/r04_wino4_timeline_before.log) shows
// 10 k cycles per tile in the output transform (16 k with a residual) for ~650 VALU instructions.  With 4 rows per round the reads of a
// round are issued right behind its 16 writes (a wavefront's LDS operations execute in order: the window needs no double buffering) and
// are consumed after the NEXT tile column's whole first transform stage (~120 VALU instructions): no exposed LDS latency, 4 rounds.
// Depth of the A-fragment ring (register quads of V read ahead from LDS): 2 = the pair after the current one, 3 = two pairs ahead.  A
// wavefront that has the matrix pipe to itself (its SIMD partner in its prologue / output transform, or a launch too small for two
// blocks per CU) runs a pair in 128 cycles -- about one ds_read_b128 round trip.  NPP % depth must be 0.  [measured, 64->64 at 256x384,
// batch 1 / 4 / 32: depth 3 = depth 2 to the microsecond (0.036 / 0.105 / 0.746 ms) and spills 1-3 registers: the A reads are not what a
// lone wavefront waits for; left at 2]
#ifndef L3C_W4_ARING
#define L3C_W4_ARING 2
#endif
static_assert(L3C_W4_ARING == 2 || L3C_W4_ARING == 3, "A ring depth");
#ifndef L3C_W4_EPI_ROWS
#define L3C_W4_EPI_ROWS 4
#endif
static_assert(L3C_W4_EPI_ROWS == 1 || L3C_W4_EPI_ROWS == 2 || L3C_W4_EPI_ROWS == 4, "rows per round");
static_assert(L3C_W4_RING == 4 || L3C_W4_RING == 6, "ring depth");
// slot of pair q (numbered on into the next chunk) in a chunk of buffer parity par: 18 pairs per chunk = 2 mod 4, 0 mod 6
__device__ __forceinline__ constexpr int ring_slot(int q, int par) { return L3C_W4_RING == 4 ? ((q + 2 * par) & 3) : q % 6; }

// POLY: the polyphase forms of a stride-2 convolution (l3c_conv_wino4_phase / _stride2) get an instantiation of their own, so that the
// profiler's kernel names tell the 5x5 stride-2 launches from the 3x3 ones (tools/pmc_bench.py).  A name tag only: the body keeps reading
// p.poly at run time (folding it changed the register allocation of the ReLU variant: 255 -> 256 VGPRs + 2 spills)
template <bool RELU, bool RES, bool SHUFFLE, bool POLY>
__global__ __launch_bounds__(256, 2) void conv_wino4_kernel(const Wino4Params p) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n_cc = p.Cin / CK;

    // block -> (image, output-channel chunk, sub-grid, tile row, group of tiles)
    // (the output-channel chunk is the FASTEST index: the blocks that compute different output channels of one tile -- four for the
    // 64 -> 256 PixelShuffle tail -- are neighbours in the launch order of one XCD and share the tile's input patches through its L2)
    unsigned w = (unsigned)xcd_remap4(blockIdx.x, p.total_blocks);
    const unsigned w_c = p.div_chunks.div(w);
    const int chunk_o = (int)(w - w_c * (unsigned)p.n_chunks_o);
    w = w_c;
    const unsigned groups = (unsigned)(p.groups_x * p.tiles_y);
    const unsigned w_t = p.div_groups.div(w);
    const unsigned grp = w - w_t * groups;
    w = w_t;
    const int dl = p.dil_log2, dil = 1 << dl;
    const int phase = (int)(w & ((1u << (2 * dl)) - 1));
    w >>= 2 * dl;
    const int b = (int)w;
    const int py = phase >> dl, px = phase & (dil - 1);
    const unsigned t_y = p.div_groups_x.div(grp);
    const int sy0 = (int)t_y * OT;                                         // tile row origin in sub-grid coordinates
    const int tx_first = (int)(grp - t_y * (unsigned)p.groups_x) * p.tpb;  // first tile of this block
    const int Ho_b = p.Ho, Wo_b = p.Wo;
    const int n_t = min(p.tpb, p.tiles_x - tx_first);

    // ---- patch fetch: thread (r0 = tid / 36 < 6, column c = tid % 36 / 2, half = tid & 1) fetches the 16-byte pieces (channels
    // 4 half .. + 3 of the chunk) of the patch pixels (r0 + 6 k, c), k = 0..2: ONE per-lane byte offset, the rows 6 apart by a
    // uniform offset.  Zero padding = the descriptor's range check: a row below the image lies beyond the buffer, a row above it at
    // a negative (= huge unsigned) offset, and a thread whose column is outside the image starts from an offset that keeps all
    // three of its pieces out of range.  216 threads fetch, the other 40 idle (offsets out of range, nothing stored).
    const int pf_r0 = tid / (2 * PW), pf_j = tid - pf_r0 * (2 * PW);
    const bool pf_thread = pf_r0 < 6;
    // input sampling: the sub-grid (py, px) of step dil -- or, for one phase of a stride-2 convolution, (in_py, in_px) of step 2
    // -- or, with poly_ncp > 0, all four phases one after the other: the chunk sequence of a tile is (phase 0: poly_ncp chunks of the
    // input's channels, phase 1: ...), the weights are those of a convolution with 4 Cin input channels (phase-major)
    const int istep = p.poly ? 2 : dil;
    const int ncp = p.poly_ncp > 0 ? p.poly_ncp : n_cc, n_phase = p.poly_ncp > 0 ? 4 : 1;
    constexpr bool PLANAR_PROBE = (L3C_W4_PROBE & 256) != 0;   // timing only: input addressed as [C/8][H][W][8] (a chunk's pieces contiguous)
    const int in_pix_b = PLANAR_PROBE ? CK * 4 : p.in_cstride * 4;
    const int pf_row_b = 6 * istep * p.W * in_pix_b;                           // six patch rows on, in bytes
    const int pf_lds = (pf_r0 * PW + (pf_j >> 1)) * PSR + (pf_j & 1) * 4;       // LDS position of piece 0 (floats); + k * 6 * PW * PSR
    int patch_off;
    auto fresh_tid = [&]() {   // the thread index, recomputed where it is used rarely (kept out of the registers the MFMA loop holds)
        unsigned m = ~0u;
        asm volatile("" : "+s"(m));
        return wave * 64 + (int)__builtin_amdgcn_mbcnt_hi(m, __builtin_amdgcn_mbcnt_lo(m, 0u));
    };
    auto set_patch_tile = [&](int t, int ph = 0) {
        const int tf = fresh_tid();
        const int r0 = tf / (2 * PW), j = tf - r0 * (2 * PW);
        const int ipy = p.poly ? (p.poly_ncp > 0 ? ph >> 1 : p.in_py) : py, ipx = p.poly ? (p.poly_ncp > 0 ? ph & 1 : p.in_px) : px;
        const int iy = ipy + istep * (sy0 - 1 + r0), ix = ipx + istep * ((tx_first + t) * OT - 1 + (j >> 1));
        const bool ok = r0 < 6 && ix >= 0 && ix < p.W;
        // (iy may be negative: the offset wraps to a huge unsigned value and the load returns zero; iy + 6 k then comes back in range)
        patch_off = ok ? (iy * p.W + ix) * in_pix_b + (j & 1) * 16 : (int)0xC0000000;   // (+ 2 rows: still beyond 2 GB)
        if constexpr (L3C_W4_PROBE & 512) patch_off = (r0 * PW + (j >> 1)) * in_pix_b + (j & 1) * 16;   // timing only: every block fetches the same pixels (cache hits)
    };
    int pf_tile = 0, pf_ph = 0, pf_cc = 0;   // the prefetch pointer: (tile, phase, chunk of the phase) of the next patch to fetch
    auto pf_advance = [&]() {
        if (++pf_cc == ncp) {
            if (pf_ph + 1 < n_phase) {
                pf_cc = 0;
                set_patch_tile(pf_tile, ++pf_ph);
            } else if (pf_tile + 1 < n_t) {
                pf_cc = 0;
                pf_ph = 0;
                set_patch_tile(++pf_tile, 0);
            } else {
                pf_cc = ncp - 1;   // past the block's last chunk: the loads stay unconditional, their data is never used
            }
        }
    };
    const auto in_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float *>(p.in + (size_t)b * p.H * p.W * p.in_cstride + p.in_coff), 0, p.H * p.W * p.in_cstride * 4, 0x00020000);
    f32x4 stage[NIT];
    auto fetch_piece = [&](int it) {
        if constexpr (L3C_W4_PROBE & 1) return;
        if constexpr (PLANAR_PROBE) {
            const auto rs = __builtin_amdgcn_make_buffer_rsrc(
                const_cast<float *>(p.in + (size_t)b * p.H * p.W * p.in_cstride + (size_t)pf_cc * p.H * p.W * CK), 0, p.H * p.W * CK * 4, 0x00020000);
            stage[it] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, patch_off + it * pf_row_b, 0, 0));
            return;
        }
        stage[it] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(in_rsrc, patch_off + it * pf_row_b, pf_cc * CK * 4, 0));
    };
    auto store_piece = [&](float *dst, int it, const f32x4 &v) {
        if (pf_thread) *reinterpret_cast<f32x4 *>(&dst[pf_lds + it * (6 * PW * PSR)]) = v;
    };

    // ---- B operands: uniform descriptor + fixed per-lane byte offset + scalar offset; all loads unconditional
    const auto u_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(p.u + (size_t)chunk_o * n_cc * U_CHUNK_FLOATS), 0,
                                                          n_cc * U_CHUNK_FLOATS * 4, 0x00020000);
    const int u_lane = (wave * 64 + lane) * 16;
    f32x4 b_ring[L3C_W4_RING];
    auto fetch_b = [&](int cc, int pp) {
        if constexpr (L3C_W4_PROBE & 1024) {   // timing only: HALF the weight stream (every other pair's fragment is not loaded)
            if (pp & 1) return f32x4{1.f, 2.f, 3.f, 4.f};
        }
        if constexpr (L3C_W4_PROBE & 128) return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(u_rsrc, u_lane, (pp & 3) * (4 * 64 * 16), 0));
        return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(u_rsrc, u_lane, (cc * NPP + pp) * (4 * 64 * 16), 0));
    };

    // ---- input transform roles: thread = (tile (ty, tx), channel c, half TH of the rows xi)
    const int t_c = lane & 7, t_tx = (lane >> 3) & 3, t_ty = (wave & 1) * 2 + (lane >> 5), t_h = wave >> 1;
    const int t_src = ((4 * t_ty + t_h) * PW + 4 * t_tx) * PSR + t_c;
    const int t_dst = (t_ty * 4 + t_tx) * 16 + (t_c >> 1) * 4 + (t_c & 1) + (3 * t_h) * 3 * VPP;

    // ---- A fragments: lane (tile m = lane & 15, k group lane >> 4)
    const int a_lane = (lane & 15) * 16 + (lane >> 4) * 4;
    constexpr int AR = L3C_W4_ARING;
    static_assert(NPP % AR == 0, "pairs per chunk must be a multiple of the A ring depth");
    f32x4 a_ring[AR];

    f32x4 acc[36];
    // PixelShuffle: block chunk_o computes SUB-PIXEL chunk_o of the 2 x 2 output pixels -- its 64 conv channels are 4 oc + chunk_o, oc = 0..63
    // (the caller packs the weights in that order, see l3c_conv_wino4) -- so that a wavefront's 16 channels are 16 ADJACENT output channels
    // of one output pixel: 64-byte runs per pixel like every other variant (round 5; before, a wavefront held the 4 sub-pixels of 4 output
    // channels and every output pixel's 256-byte line left in sixteen 16-byte pieces from sixteen wavefronts: 2.8x the algorithmic bytes)
    const int co_lane = SHUFFLE ? 4 * (wave * 16 + (lane & 15)) + chunk_o : chunk_o * 64 + wave * 16 + (lane & 15);
    const float bias_init = co_lane < p.Cout ? p.bias[co_lane] : 0.0f;

    // ---- output addressing: ONE uniform descriptor at the block's first output row, a per-lane byte offset, uniform offsets
    constexpr int S = SHUFFLE ? 2 : 1;
    const int col_b = S * dil * p.out_cstride * 4, row_b = S * dil * (S * p.Wo) * p.out_cstride * 4;   // one conv pixel / row on
    const int rcol_b = dil * p.res_cstride * 4, rrow_b = dil * p.Wo * p.res_cstride * 4;
    const auto o_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        p.out + (((size_t)b * (S * p.Ho) + S * (py + dil * sy0) + (SHUFFLE ? (chunk_o >> 1) : 0)) * (S * p.Wo) + S * px + (SHUFFLE ? (chunk_o & 1) : 0)) *
                    p.out_cstride + p.out_coff + (SHUFFLE ? 0 : chunk_o * 64), 0, OOB, 0x00020000);
    const auto r_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float *>(RES ? p.res + (((size_t)b * p.Ho + py + dil * sy0) * p.Wo + px) * p.res_cstride + p.res_coff + chunk_o * 64 : p.bias),
        0, RES ? OOB : 0, 0x00020000);

#ifdef L3C_W4_TIMELINE
    // stamps of wavefront (block, wave) at dbg[((block * 4 + wave) * 32) + k]: 0 start, 1 prologue done, then per tile (2 + 2 t) loop done,
    // (3 + 2 t) output transform done; 15 = number of tiles; 16 + 2 r / 17 + 2 r: inside the output transform of tile 0, tile column r: window
    // written (both transform stages done) / previous round stored and this round's reads issued; 24..27 / 28..31: inside the chunk loops of tiles 0 / 1
    // (after chunks 0, 2, 4, 6).  s_memtime ticks = shader cycles.  Blocks beyond the buffer are not recorded.
    unsigned long long *dbg_w = (p.dbg && blockIdx.x < 8192) ? p.dbg + ((size_t)blockIdx.x * 4 + wave) * 32 : nullptr;
    auto stamp = [&](int k) {
        if (dbg_w && k != 15 && k < 32) {
            const unsigned long long t = __builtin_amdgcn_s_memtime();
            if (lane == 0) dbg_w[k] = t;
        }
    };
    stamp(0);
    if (dbg_w && lane == 0) dbg_w[15] = (unsigned long long)n_t;
#else
    auto stamp = [](int) {};
#endif
    auto body = [&](auto th_c) __attribute__((always_inline)) {
        constexpr int TH = decltype(th_c)::value;
        float T[3][6];    // rows xi = 3 TH .. 3 TH + 2 of B^T d for this thread's (tile, channel)
        float L[2][5];    // raw rows TH .. TH + 4 of one patch column (two columns in flight)
        auto tr_load = [&](const float *src, int col) {
#pragma unroll
            for (int i = 0; i < 5; ++i) L[col & 1][i] = src[(i * PW + col) * PSR];
        };
        // (the transform is dealt out between the MFMAs in slices of 4..8 instructions: a gap of 32 matrix-pipe cycles has room for ~7)
        auto tr_col_a = [&](int col) {
            const float(&l)[5] = L[col & 1];
            if constexpr (TH == 0) {   // xi = 0, 1, 2 from raw rows 0 .. 4 (l[i] = row i)
                T[0][col] = __builtin_fmaf(2.0f, l[0] + l[4], __builtin_fmaf(3.0f, l[3] - l[1], -4.0f * l[2]));
            } else {                   // xi = 3, 4, 5 from raw rows 1 .. 5 (l[i] = row 1 + i)
                const float pp_ = l[0] - l[2], qq_ = l[1] - l[3];
                T[0][col] = __builtin_fmaf(2.0f, pp_, qq_);
                T[1][col] = __builtin_fmaf(-2.0f, qq_, pp_);
            }
        };
        auto tr_col_b = [&](int col) {
            const float(&l)[5] = L[col & 1];
            if constexpr (TH == 0) {
                T[1][col] = __builtin_fmaf(2.0f, l[4], __builtin_fmaf(5.0f, l[3], __builtin_fmaf(-2.0f, l[1], l[2])));
                T[2][col] = __builtin_fmaf(-2.0f, l[4], __builtin_fmaf(-2.0f, l[1], __builtin_fmaf(5.0f, l[2], -l[3])));
            } else {
                T[2][col] = __builtin_fmaf(2.0f, l[0] + l[4], __builtin_fmaf(3.0f, l[3] - l[1], -4.0f * l[2]));
            }
        };
        auto tr_col = [&](int col) {
            tr_col_a(col);
            tr_col_b(col);
        };
        float R[6];
        auto tr_row_part = [&](int k, int part) {   // row k through B^T in three slices (== bt6)
            const float(&w)[6] = T[k];
            if (part == 0) {
                R[0] = __builtin_fmaf(2.0f, w[0] + w[4], __builtin_fmaf(3.0f, w[3] - w[1], -4.0f * w[2]));
                R[1] = __builtin_fmaf(2.0f, w[4], __builtin_fmaf(5.0f, w[3], __builtin_fmaf(-2.0f, w[1], w[2])));
            } else if (part == 1) {
                R[2] = __builtin_fmaf(-2.0f, w[4], __builtin_fmaf(-2.0f, w[1], __builtin_fmaf(5.0f, w[2], -w[3])));
                const float p_ = w[1] - w[3], q_ = w[2] - w[4];
                R[3] = __builtin_fmaf(2.0f, p_, q_);
                R[4] = __builtin_fmaf(-2.0f, q_, p_);
            } else {
                R[5] = __builtin_fmaf(2.0f, w[1] + w[5], __builtin_fmaf(3.0f, w[4] - w[2], -4.0f * w[3]));
            }
        };
        auto tr_row = [&](int k) {
            tr_row_part(k, 0);
            tr_row_part(k, 1);
            tr_row_part(k, 2);
        };
        auto tr_write = [&](float *dst, int k) {   // row xi = 3 TH + k: positions (xi, 0..5) = pairs 3 xi .. 3 xi + 2
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                dst[(3 * k + j) * VPP] = R[2 * j];
                dst[(3 * k + j) * VPP + 2] = R[2 * j + 1];
            }
        };

        // ---- prologue: patches 0, 1 -> raw[0], raw[1]; patch 2 in flight; V[0] = transform of patch 0; B of the first pairs
        set_patch_tile(0);
        {
            f32x4 first[2][NIT];
#pragma unroll
            for (int k = 0; k < 2; ++k) {
#pragma unroll
                for (int it = 0; it < NIT; ++it)
                    first[k][it] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(in_rsrc, patch_off + it * pf_row_b, pf_cc * CK * 4, 0));
                pf_advance();
            }
#pragma unroll
            for (int it = 0; it < NIT; ++it) fetch_piece(it);
            pf_advance();
#pragma unroll
            for (int q = 0; q < L3C_W4_RING; ++q) b_ring[q] = fetch_b(0, q);
#pragma unroll
            for (int it = 0; it < NIT; ++it) store_piece(lds + RAW_OFF0, it, first[0][it]);
#pragma unroll
            for (int it = 0; it < NIT; ++it) store_piece(lds + RAW_OFF1, it, first[1][it]);
        }
        __syncthreads();
#pragma unroll
        for (int col = 0; col < 6; ++col) {
            tr_load(lds + RAW_OFF0 + t_src, col);
            tr_col(col);
        }
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            tr_row(k);
            tr_write(lds + V_OFF0 + t_dst, k);
        }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < AR; ++j) a_ring[j] = *reinterpret_cast<const f32x4 *>(lds + V_OFF0 + a_lane + j * VPP);
        __builtin_amdgcn_s_waitcnt(0x0f70);   // vmcnt(0): nothing of the prologue stays in flight (see conv_wino.hip)
        stamp(1);

        // Chunk g (buffer parity par = g & 1) -- invariants at its start: V[par] complete and visible; raw[par ^ 1] = patch of
        // chunk g + 1, visible; `stage` = patch of chunk g + 2 (in flight); a_ring = A of pairs 0, 1; the prefetch pointer is at chunk
        // g + 3; B operands: b_ring (slot (pair + 2 par) & 3) holds pairs 0..2 (FIRST chunk of a tile: 0..3), b_burst pairs 3..9.
        //
        // vmcnt is IN ORDER: a B load issued behind a patch fetch cannot be consumed before the patch has arrived (HBM latency for
        // one chunk in four -- a pixel's 32 channels share a 128-byte line -- and L2 latency otherwise), and with four pairs of
        // lookahead that stalled the wave in every chunk [probe: no patch fetch -15 %].  So the B operands of pairs 3..9 of the NEXT
        // chunk are fetched in one burst (seven register quads: the registers the input transform has just released -- it runs in
        // pairs 7..15, the burst is consumed by pair 9; [measured: columns in pairs 10..12 two at a time 171, 6..11 168, 7..12 174 MPix/s]) IMMEDIATELY BEFORE the patch fetch at pair 16: the first B load issued behind
        // the patch is the one of pair 10 at pair 6 of the next chunk, needed at pair 10 -- the patch has 12 pairs' time to arrive.
        // KIND 0: first chunk of a tile (all its B operands come through the ring: the chunk before it was the last of the previous
        // tile or the prologue), 1: middle, 2: last chunk of a tile (no burst: the output transform needs the registers).
        f32x4 b_burst[7];
        auto chunk = [&](const int cc, auto kind_c, auto par_c) __attribute__((always_inline)) {
            constexpr int KIND = decltype(kind_c)::value;
            constexpr bool FIRST = KIND == 0, LAST = KIND == 2;
            constexpr int par = decltype(par_c)::value;
            const int cc_b = cc + 1 == n_cc ? 0 : cc + 1;   // the next chunk's weights (the next tile starts over with chunk 0)
            const float *a_cur = lds + (par ? V_OFF1 : V_OFF0) + a_lane;
            const float *a_nxt = lds + (par ? V_OFF0 : V_OFF1) + a_lane;
            float *v_next = lds + (par ? V_OFF0 : V_OFF1) + t_dst;
            const float *r_src = lds + (par ? RAW_OFF0 : RAW_OFF1) + t_src;   // raw[par ^ 1]
            float *r_dst = lds + (par ? RAW_OFF1 : RAW_OFF0);                 // raw[par]
            const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
            constexpr bool TR = !(L3C_W4_PROBE & 2), ST = !(L3C_W4_PROBE & 32), LB = !(L3C_W4_PROBE & 4);
#define L3C_W4_MFMA(Q, A, B, ZERO)                                                                          \
    __builtin_amdgcn_sched_barrier(0);                                                                      \
    if constexpr (L3C_W4_PROBE & 64) { asm volatile("v_add_f32 %0, %1, %2" : "+v"(acc[Q][0]) : "v"(A), "v"(B)); }          \
    else acc[Q] = __builtin_amdgcn_mfma_f32_16x16x4f32((A), (B), (FIRST && (ZERO) && (Q) != 7) ? zero4 : acc[Q], 0, 0, 0); \
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int pp = 0; pp < NPP; ++pp) {
                const f32x4 &A = a_ring[pp % AR];
                const bool from_burst = !FIRST && pp >= 3 && pp <= 9;
                const f32x4 &Bv = from_burst ? b_burst[pp >= 3 && pp <= 9 ? pp - 3 : 0] : b_ring[ring_slot(pp, par)];
                if (pp == NPP - 1) {
                    // everything chunk g + 1 needs from this wave is issued: V[par ^ 1] written, patch g + 2 stored
                    if constexpr (!(L3C_W4_PROBE & 8)) __syncthreads();
#pragma unroll
                    for (int j = 0; j + 1 < AR; ++j) a_ring[j] = *reinterpret_cast<const f32x4 *>(a_nxt + j * VPP);
                }
                L3C_W4_MFMA(2 * pp, A[0], Bv[0], true)
                // one patch column per pair, arithmetic in pairs 7..12 (the burst's registers free up as its pairs 3..9 are consumed), its
                // five LDS reads TR_LEAD pairs earlier (two columns in flight: L[col & 1])
                if (TR && pp >= 7 - L3C_W4_TR_LEAD && pp <= 12 - L3C_W4_TR_LEAD) tr_load(r_src, pp - 7 + L3C_W4_TR_LEAD);
                if (TR && pp >= 13 && pp <= 15) tr_row_part(pp - 13, 0);
                if (ST && pp == L3C_W4_FETCH_PP) store_piece(r_dst, 0, stage[0]);
                if (ST && pp == L3C_W4_FETCH_PP) store_piece(r_dst, 1, stage[1]);
                L3C_W4_MFMA(2 * pp + 1, A[2], Bv[2], true)
                if (TR && pp >= 7 && pp <= 12) tr_col_a(pp - 7);
                if (TR && pp >= 13 && pp <= 15) tr_row_part(pp - 13, 1);
                if (ST && pp == L3C_W4_FETCH_PP) store_piece(r_dst, 2, stage[2]);
                if (LB && !LAST && pp == 15) {
#pragma unroll
                    for (int k = 0; k < 3; ++k) b_burst[k] = fetch_b(cc_b, 3 + k);
                }
                L3C_W4_MFMA(2 * pp, A[1], Bv[1], false)
                if (TR && pp >= 7 && pp <= 12) tr_col_b(pp - 7);
                if (TR && pp >= 13 && pp <= 15) {
                    tr_row_part(pp - 13, 2);
                    tr_write(v_next, pp - 13);
                }
                L3C_W4_MFMA(2 * pp + 1, A[3], Bv[3], false)
                // (the rings are reloaded behind the pair's last MFMA: A and Bv are references into them)
                if (pp + AR < NPP) a_ring[pp % AR] = *reinterpret_cast<const f32x4 *>(a_cur + (pp + AR) * VPP);
                if (pp == NPP - 1) a_ring[AR - 1] = *reinterpret_cast<const f32x4 *>(a_nxt + (AR - 1) * VPP);
                // the ring: pair pp + 4 -- of this chunk, or (numbered on) of the next one, whose pairs 3..9 come from the burst
                if constexpr (LB) {
                    const int nxt = pp + L3C_W4_RING;
                    const bool ring_load = nxt < NPP ? (FIRST || nxt >= 10) : (LAST || nxt - NPP <= 2);
                    if (ring_load) b_ring[ring_slot(nxt, par)] = nxt < NPP ? fetch_b(cc, nxt) : fetch_b(cc_b, nxt - NPP);
                }
                if (pp == 16) {
                    if (LB && !LAST) {
#pragma unroll
                        for (int k = 3; k < 7; ++k) b_burst[k] = fetch_b(cc_b, 3 + k);
                    }
                }
                if (pp == L3C_W4_FETCH_PP) {
#pragma unroll
                    for (int it = 0; it < NIT; ++it) fetch_piece(it);   // patch of chunk g + 3, behind the burst
                    pf_advance();
                }
            }
#undef L3C_W4_MFMA
        };

        for (int t = 0; t < n_t; ++t) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {   // (volatile: four copies made here, per tile -- not four registers held forever)
                float v;
                asm volatile("v_mov_b32 %0, %1" : "=v"(v) : "v"(bias_init));
                acc[7][r] = v;
            }
            chunk(0, std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{});
#ifdef L3C_W4_TIMELINE
            if (t < 2) stamp(24 + 4 * t);                        // after chunk 0 of tiles 0 and 1
#endif
            for (int cc = 1; cc + 1 < n_cc; cc += 2) {
                chunk(cc, std::integral_constant<int, 1>{}, std::integral_constant<int, 1>{});
                chunk(cc + 1, std::integral_constant<int, 1>{}, std::integral_constant<int, 0>{});
#ifdef L3C_W4_TIMELINE
                if (t < 2 && cc <= 5) stamp(24 + 4 * t + (cc + 1) / 2);   // after chunks 2, 4, 6
#endif
            }
            chunk(n_cc - 1, std::integral_constant<int, 2>{}, std::integral_constant<int, 1>{});
            stamp(2 + 2 * t);

            // ---- output transform Y = A^T M A, in registers: lane (q = lane >> 4, n = lane & 15) holds, in register r of the 36
            // fragments, the transformed tile (ty, tx) = (q, r) of output channel n.  The results leave through a
            // wavefront-private LDS window that turns "one channel, 16 pixels per lane" into "four adjacent channels of one pixel
            // per lane": row (r, i) = output row i of the four tiles (., r) -> 16 pixels x 16 channels: four ds_write_b32, one
            // ds_read_b128, one 16-byte store (and residual load) per lane -- 16 of each per tile instead of 64 four-byte ones;
            // L3C_W4_EPI_ROWS rows per round.
            // Everything the epilogue derives from the lane index is recomputed per tile (the opaque mask hides the index from the
            // loop-invariant code motion): held across the chunk loop these values would push the accumulators out of the registers.
            const int ln = fresh_tid() & 63;
            const int q = ln >> 4, n = ln & 15;
            const int q2 = ln >> 4, j2 = (ln >> 2) & 3, c4 = ln & 3;   // store layout: pixel (4 q2 + i, 4 r + j2), channels 4 c4 ..
            const int sx0 = (tx_first + t) * OT;
            // Cout % 4 == 0 (shuffle: the block's channels are output channels 0 .. Cout / 4 - 1 of its sub-pixel)
            const bool lane_ok = SHUFFLE ? 4 * (wave * 16 + c4 * 4) < p.Cout : chunk_o * 64 + wave * 16 + c4 * 4 < p.Cout;
            int r_lane = 0;
            // (shuffle: conv pixel (y, x) -> output pixel (2 y + chunk_o / 2, 2 x + chunk_o % 2): row_b / col_b step two output pixels, the
            // sub-pixel sits in the descriptor's base)
            const int o_lane = 4 * q2 * row_b + j2 * col_b + (wave * 16 + c4 * 4) * 4;
            if constexpr (RES) r_lane = 4 * q2 * rrow_b + j2 * rcol_b + (wave * 16 + c4 * 4) * 4;
            const int oy_l = py + dil * (sy0 + 4 * q2), ox_l = px + dil * (sx0 + j2);
            // window of this wavefront, in V[1]: [row of the round][q][j][16 channels] (a 2-way bank conflict on the four-byte writes -- lanes
            // n and n + 32 q -- costs a ds_write_b32 nothing; the 16-byte reads cover 1 KB contiguously)
            constexpr int ER = L3C_W4_EPI_ROWS;
            float *win = lds + V_OFF1 + wave * (ER * WIN_ROW_FLOATS);
            float *w_dst = win + q * 64 + n;
            const float *w_src = win + q2 * 64 + j2 * 16 + c4 * 4;
            auto lane_off = [&](int base, int r, int i) {
                const bool ok = lane_ok && oy_l + dil * i < Ho_b && ox_l + dil * 4 * r < Wo_b;
                return ok ? base : OOB;
            };
            f32x4 resv[L3C_W4_RES_RING];
            auto res_load = [&](int k) {   // row k = r * 4 + i
                if constexpr (RES)
                    resv[k % L3C_W4_RES_RING] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(
                                                                r_rsrc, lane_off(r_lane, k >> 2, k & 3), (k & 3) * rrow_b + (sx0 + 4 * (k >> 2)) * rcol_b, L3C_W4_RES_AUX));
            };
            f32x4 rd[ER];                  // the rows of the last round, read back in the store layout (in flight until `finish`)
            auto finish_row = [&](int k, f32x4 v) {      // row k: registers -> memory
                if constexpr (RELU) {   // (fmaxf would first canonicalise the value read from LDS: a second v_max per element)
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        float m;
                        asm("v_max_f32 %0, 0, %1" : "=v"(m) : "v"(v[e]));
                        v[e] = m;
                    }
                }
                if constexpr (RES) v = v + resv[k % L3C_W4_RES_RING];
                const int vo = lane_off(o_lane, k >> 2, k & 3), so = (k & 3) * row_b + (sx0 + 4 * (k >> 2)) * col_b;
                __builtin_amdgcn_sched_barrier(0);
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(__attribute__((__vector_size__(4 * sizeof(unsigned)))) unsigned, v),
                                                       o_rsrc, vo, so, L3C_W4_OUT_AUX);
                // gfx950: a VALU write to the data registers of a 16-byte buffer store with a REGISTER soffset, issued right behind
                // it, overtakes the store's read of its last dwords, and the compiler inserts no wait states for this form
                // (found with conv_wino.hip; tools/check_store_hazard.py scans the ISA at build time)
                asm volatile("s_nop 1");
                __builtin_amdgcn_sched_barrier(0);
            };
            auto finish = [&](int round) {               // the ER rows of a round; behind each store the residual of a later row is requested
#pragma unroll
                for (int e = 0; e < ER; ++e) {
                    const int k = round * ER + e;
                    finish_row(k, rd[e]);
                    if (k + L3C_W4_RES_RING < 16) res_load(k + L3C_W4_RES_RING);   // (into the register quad this row has just released)
                }
            };
            if constexpr (L3C_W4_PROBE & 16) {
                f32x4 sum = acc[0];
#pragma unroll
                for (int i = 1; i < 36; ++i) sum = sum + acc[i];
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(__attribute__((__vector_size__(4 * sizeof(unsigned)))) unsigned, sum),
                                                       o_rsrc, lane_off(o_lane, 0, 0), sx0 * col_b, 0);
                continue;
            }
#pragma unroll
            for (int k = 0; k < L3C_W4_RES_RING; ++k) res_load(k);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float tcol[4][6];   // t[i][nu] = sum_xi A^T[i][xi] M[xi][nu]
#pragma unroll
                for (int nu = 0; nu < 6; ++nu) {
                    float y[4];
                    at6(acc[0 * 6 + nu][r], acc[1 * 6 + nu][r], acc[2 * 6 + nu][r], acc[3 * 6 + nu][r], acc[4 * 6 + nu][r], acc[5 * 6 + nu][r], y);
#pragma unroll
                    for (int i = 0; i < 4; ++i) tcol[i][nu] = y[i];
                }
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int k = r * 4 + i, e = i % ER;
                    float y[4];
                    at6(tcol[i][0], tcol[i][1], tcol[i][2], tcol[i][3], tcol[i][4], tcol[i][5], y);
#pragma unroll
                    for (int j = 0; j < 4; ++j) w_dst[e * WIN_ROW_FLOATS + j * 16] = y[j];
                    if (e == ER - 1) {
                        // Order of a round, enforced through the memory operations (the optimiser otherwise sinks the transform arithmetic
                        // below the stores and waits for the LDS reads right behind their issue): this round's window writes -- which
                        // need the whole transform of this tile column -- | the PREVIOUS round's rows leave `rd` for memory (their LDS
                        // reads have had that arithmetic, ~120 VALU instructions, to arrive) | this round's reads are issued into `rd`.
                        asm volatile("" ::: "memory");
                        if (t == 0) stamp(16 + 2 * r);
                        if (k >= ER) finish(k / ER - 1);
                        // the window is this wavefront's own and its LDS operations execute in order: no block barrier, no double
                        // buffering (the next round's writes are issued behind these reads); the fences keep the compiler from moving
                        // the reads of other lanes' values above the writes, and the next writes above the reads
                        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                        __builtin_amdgcn_wave_barrier();
#pragma unroll
                        for (int e2 = 0; e2 < ER; ++e2) rd[e2] = *reinterpret_cast<const f32x4 *>(w_src + e2 * WIN_ROW_FLOATS);
                        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                        __builtin_amdgcn_wave_barrier();
                        if (t == 0) stamp(17 + 2 * r);
                    }
                }
            }
            finish(16 / ER - 1);
            // V[1] goes back to the pipeline: the next tile's first chunk transforms into it (at its pair 13, with no barrier in between)
            if (t + 1 < n_t) __syncthreads();
            stamp(3 + 2 * t);
        }
    };
    if (t_h == 0) body(std::integral_constant<int, 0>{});
    else body(std::integral_constant<int, 1>{});
}

// OIHW 3x3 weights -> U = G g G^T (computed in double, rounded once) in the kernel's fragment order
// [Cout/64][Cin/8][18 position pairs][4 waves][64 lanes][4]: lane (n = lane % 16, kq = lane / 16), element e * 2 + s holds
// U[position (xi = pp / 3, nu = 2 (pp % 3) + e)][co = chunk * 64 + wave * 16 + n][ci = cc * 8 + 2 kq + s].
__global__ __launch_bounds__(256) void pack_wino4_kernel(const float *__restrict__ w, int Cout, int Cin, float *__restrict__ packed,
                                                         int64_t total) {
    const double G[6][3] = {{0.5, 0.0, 0.0},           {1.0 / 6, 1.0 / 6, 1.0 / 6},     {1.0 / 6, -1.0 / 6, 1.0 / 6},
                            {16.0 / 15, 8.0 / 15, 4.0 / 15}, {1.0 / 30, -1.0 / 15, 2.0 / 15}, {0.0, 0.0, 0.5}};
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        int64_t r = i;
        const int idx = r % 4;  r /= 4;
        const int lane = r % 64;  r /= 64;
        const int wv = r % 4;  r /= 4;
        const int pp = r % NPP;  r /= NPP;
        const int cc = r % (Cin / 8);  r /= (Cin / 8);
        const int chunk = (int)r;
        const int co = chunk * 64 + wv * 16 + (lane & 15);
        const int ci = cc * 8 + 2 * (lane >> 4) + (idx & 1);
        const int xi = pp / 3, nu = 2 * (pp % 3) + (idx >> 1);
        double u = 0.0;
        if (co < Cout) {
            const float *g = w + ((size_t)co * Cin + ci) * 9;
            for (int a = 0; a < 3; ++a)
                for (int c = 0; c < 3; ++c) u += G[xi][a] * (double)g[a * 3 + c] * G[nu][c];
        }
        packed[i] = (float)u;
    }
}

}  // namespace

static std::atomic<int> g_w4_tpb{getenv("L3C_WINO4_TPB") ? atoi(getenv("L3C_WINO4_TPB")) : 0};
static const long long g_w4_min_blocks = getenv("L3C_WINO4_MIN_BLOCKS") ? atoll(getenv("L3C_WINO4_MIN_BLOCKS")) : 8 * 512;

#ifdef L3C_W4_TIMELINE
static unsigned long long *g_w4_dbg = nullptr;
extern "C" void l3c_conv_wino4_set_debug(void *ptr) { g_w4_dbg = (unsigned long long *)ptr; }
#endif

extern "C" {

int l3c_conv_wino4_set_tiles_per_block(int n) { return g_w4_tpb.exchange(n < 0 ? 0 : n > 64 ? 64 : n); }

int64_t l3c_conv_wino4_packed_words(int Cout, int Cin) { return (int64_t)((Cout + 63) / 64) * (Cin / 8) * U_CHUNK_FLOATS; }

int l3c_conv_wino4_pack_weights(const float *w_oihw, int Cout, int Cin, float *packed, l3c_stream_t stream) {
    L3C_REQUIRE(w_oihw && packed, "null pointer");
    L3C_REQUIRE(Cout > 0 && Cin > 0 && Cin % 16 == 0, "Cin must be a multiple of 16");
    const int64_t total = l3c_conv_wino4_packed_words(Cout, Cin);
    int64_t g = (total + 255) / 256;
    hipLaunchKernelGGL(pack_wino4_kernel, dim3((unsigned)(g > 4096 ? 4096 : g)), dim3(256), 0, l3c::as_stream(stream), w_oihw,
                       Cout, Cin, packed, total);
    return l3c::check_launch("pack_wino4_kernel");
}

static int conv_wino4_launch(const l3c_conv_desc *d, int poly, int phase_y, int phase_x, l3c_stream_t stream);   // poly: 0 no, 1 one phase, 2 all four

int l3c_conv_wino4(const l3c_conv_desc *d, l3c_stream_t stream) {
    L3C_REQUIRE(d, "null descriptor");
    L3C_REQUIRE(d->KS == 3 && d->stride == 1, "Winograd F(4x4,3x3): 3x3, stride 1 only");
    return conv_wino4_launch(d, 0, 0, 0, stream);
}

int l3c_conv_wino4_phase(const l3c_conv_desc *d, int phase_y, int phase_x, l3c_stream_t stream) {
    L3C_REQUIRE(d, "null descriptor");
    L3C_REQUIRE(d->KS == 3 && d->stride == 2 && d->dilation == 1, "polyphase form: 3x3 phase kernel, stride 2, no dilation");
    L3C_REQUIRE((phase_y == 0 || phase_y == 1) && (phase_x == 0 || phase_x == 1), "phase must be 0 or 1");
    L3C_REQUIRE(d->Hin % 2 == 0 && d->Win % 2 == 0, "polyphase form: even input size");
    L3C_REQUIRE(!(d->epilogue & (L3C_EPI_PIXEL_SHUFFLE | L3C_EPI_RELU)), "polyphase form: bias (+ residual) only");
    return conv_wino4_launch(d, 1, phase_y, phase_x, stream);
}

int l3c_conv_wino4_stride2(const l3c_conv_desc *d, l3c_stream_t stream) {
    L3C_REQUIRE(d, "null descriptor");
    L3C_REQUIRE(d->KS == 5 && d->stride == 2 && d->dilation == 1, "fused polyphase form: 5x5, stride 2, padding 2");
    L3C_REQUIRE(d->Hin % 2 == 0 && d->Win % 2 == 0, "polyphase form: even input size");
    L3C_REQUIRE(d->epilogue == 0, "fused polyphase form: bias only");
    return conv_wino4_launch(d, 2, 0, 0, stream);
}

static int conv_wino4_launch(const l3c_conv_desc *d, int poly, int phase_y, int phase_x, l3c_stream_t stream) {
    L3C_REQUIRE(d->in && d->packed_w && d->bias && d->out, "null pointer in descriptor");
    L3C_REQUIRE(d->dilation == 1 || d->dilation == 2 || d->dilation == 4, "dilation must be 1, 2 or 4");
    L3C_REQUIRE(d->dilation == 1 || !(d->epilogue & L3C_EPI_PIXEL_SHUFFLE), "pixel shuffle with dilation not provided");
    L3C_REQUIRE(d->B > 0 && d->Hin > 0 && d->Win > 0 && d->Cout > 0, "bad shape");
    L3C_REQUIRE(d->Cin > 0 && d->Cin % (2 * CK) == 0, "Cin must be a multiple of 16 (an even number of 8-channel chunks)");
    L3C_REQUIRE(d->in_cstride % 4 == 0 && d->in_coff % 4 == 0, "input channel stride/offset must be multiples of 4");
    L3C_REQUIRE(d->out_cstride % 4 == 0 && d->out_coff % 4 == 0, "output channel stride/offset must be multiples of 4 (16-byte stores)");
    L3C_REQUIRE(!(d->epilogue & L3C_EPI_RESIDUAL) || (d->res_cstride % 4 == 0 && d->res_coff % 4 == 0),
                "residual channel stride/offset must be multiples of 4 (16-byte loads)");
    L3C_REQUIRE(((uintptr_t)d->in | (uintptr_t)d->out | (uintptr_t)d->packed_w | ((d->epilogue & L3C_EPI_RESIDUAL) ? (uintptr_t)d->residual : 0)) % 16 == 0,
                "input, output, residual and packed weights must be 16-byte aligned");
    L3C_REQUIRE(d->Cout % 4 == 0, "Cout must be a multiple of 4: a lane stores four adjacent channels");
    L3C_REQUIRE(!(d->epilogue & L3C_EPI_PIXEL_SHUFFLE) || d->Cout == 256,
                "pixel shuffle: Cout must be 256 (four sub-pixel blocks of 64 output channels; weights packed sub-pixel-major)");
    L3C_REQUIRE((d->epilogue & ~(L3C_EPI_RELU | L3C_EPI_RESIDUAL | L3C_EPI_PIXEL_SHUFFLE)) == 0, "unknown epilogue bits");
    L3C_REQUIRE(d->in_coff + d->Cin <= d->in_cstride, "input channel slice out of range");
    L3C_REQUIRE(d->out_coff + ((d->epilogue & L3C_EPI_PIXEL_SHUFFLE) ? d->Cout / 4 : d->Cout) <= d->out_cstride, "output channel slice out of range");
    L3C_REQUIRE(!(d->epilogue & L3C_EPI_RESIDUAL) || d->residual, "residual epilogue without residual pointer");
    L3C_REQUIRE(!(d->epilogue & L3C_EPI_RESIDUAL) || d->res_coff + d->Cout <= d->res_cstride, "residual channel slice out of range");
    L3C_REQUIRE(!((d->epilogue & L3C_EPI_PIXEL_SHUFFLE) && (d->epilogue & (L3C_EPI_RESIDUAL | L3C_EPI_RELU))),
                "pixel shuffle + residual / ReLU not provided");
    Wino4Params p{};
    p.in = d->in;  p.u = d->packed_w;  p.bias = d->bias;
    p.res = (d->epilogue & L3C_EPI_RESIDUAL) ? d->residual : nullptr;
    p.out = d->out;
    p.in_cstride = d->in_cstride;  p.in_coff = d->in_coff;
    p.res_cstride = d->res_cstride;  p.res_coff = d->res_coff;
    p.out_cstride = d->out_cstride;  p.out_coff = d->out_coff;
    p.B = d->B;  p.H = d->Hin;  p.W = d->Win;  p.Cin = poly == 2 ? 4 * d->Cin : d->Cin;  p.Cout = d->Cout;
    p.poly = poly != 0;  p.in_py = phase_y;  p.in_px = phase_x;  p.poly_ncp = poly == 2 ? d->Cin / CK : 0;
    p.Ho = poly ? d->Hin / 2 : d->Hin;  p.Wo = poly ? d->Win / 2 : d->Win;
    p.dil = d->dilation;
    p.dil_log2 = d->dilation == 4 ? 2 : d->dilation == 2 ? 1 : 0;
    p.tiles_x = ((p.Wo + p.dil - 1) / p.dil + OT - 1) / OT;   // tiles of the (largest) sub-grid
    p.tiles_y = ((p.Ho + p.dil - 1) / p.dil + OT - 1) / OT;
    p.n_chunks_o = (p.Cout + 63) / 64;
    const int64_t rows = (int64_t)p.tiles_y * p.dil * p.dil * p.n_chunks_o * p.B;
    const int tpb_set = g_w4_tpb.load(std::memory_order_relaxed);
    int tpb = tpb_set > 0 ? tpb_set : W4_TPB_MAX;
    if (tpb > p.tiles_x) tpb = p.tiles_x;
    if (tpb_set <= 0) {   // the fewest EVEN groups per tile row that still give the launch enough blocks (conv_wino.hip)
        int g = (p.tiles_x + W4_TPB_MAX - 1) / W4_TPB_MAX;
        while (g < p.tiles_x && rows * g < g_w4_min_blocks) ++g;
        tpb = (p.tiles_x + g - 1) / g;
    }
    p.tpb = tpb;
    p.groups_x = (p.tiles_x + tpb - 1) / tpb;
    const int64_t total = rows * p.groups_x;
    L3C_REQUIRE(total < (1ll << 31), "grid too large");
    const int64_t S2 = (d->epilogue & L3C_EPI_PIXEL_SHUFFLE) ? 4 : 1;
    L3C_REQUIRE((int64_t)p.H * p.W * p.in_cstride * 4 < 0x7ffffff0ll, "one image of the input must stay below 2 GB");
    L3C_REQUIRE(S2 * 20 * p.dil * (p.W + 64 * p.dil) * p.out_cstride * 4 < 0x7ffffff0ll && 20ll * p.dil * (p.W + 64 * p.dil) * p.res_cstride * 4 < 0x7ffffff0ll,
                "image too wide for 32-bit offsets inside a tile row");
    p.total_blocks = (int)total;
#ifdef L3C_W4_TIMELINE
    p.dbg = g_w4_dbg;
#endif
    p.div_groups = w4_div((unsigned)(p.groups_x * p.tiles_y));
    p.div_groups_x = w4_div((unsigned)p.groups_x);
    p.div_chunks = w4_div((unsigned)p.n_chunks_o);
    typedef void (*kernel_t)(const Wino4Params);
    static const kernel_t variants[7] = {conv_wino4_kernel<false, false, false, false>, conv_wino4_kernel<true, false, false, false>,
                                         conv_wino4_kernel<false, true, false, false>, conv_wino4_kernel<true, true, false, false>,
                                         conv_wino4_kernel<false, false, true, false>,
                                         conv_wino4_kernel<false, false, false, true>, conv_wino4_kernel<false, true, false, true>};
    static bool attr_set[64][7] = {};   // > 64 KB of dynamic LDS needs the opt-in, per device
    int dev = 0;
    {
        const int rc = l3c::check_hip(hipGetDevice(&dev), "hipGetDevice");
        if (rc != L3C_OK) return rc;
    }
    L3C_REQUIRE(dev >= 0 && dev < 64, "device index out of range");
    const bool relu = d->epilogue & L3C_EPI_RELU, res = d->epilogue & L3C_EPI_RESIDUAL, shuffle = d->epilogue & L3C_EPI_PIXEL_SHUFFLE;
    const int v = poly ? (res ? 6 : 5) : shuffle ? 4 : (relu ? 1 : 0) + (res ? 2 : 0);   // (the polyphase forms: bias (+ residual) only)
    if (!attr_set[dev][v]) {
        const int rc = l3c::check_hip(hipFuncSetAttribute(reinterpret_cast<const void *>(variants[v]),
                                                          hipFuncAttributeMaxDynamicSharedMemorySize, W4_LDS_BYTES),
                                      "hipFuncSetAttribute");
        if (rc != L3C_OK) return rc;
        attr_set[dev][v] = true;
    }
    hipLaunchKernelGGL(variants[v], dim3((unsigned)total), dim3(256), W4_LDS_BYTES, l3c::as_stream(stream), p);
    return l3c::check_launch("conv_wino4_kernel");
}
}
