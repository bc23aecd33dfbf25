// conv_wino.hip -- 3x3 / stride 1 convolution (dilation 1, 2, 4) by Winograd F(2x2, 3x3) on the fp32 MFMA.
//
// 2.25x fewer multiplications than the implicit GEMM of conv_mfma.hip: every 2x2 output tile is
//     Y = A^T [ sum_ci (G g G^T) (.) (B^T d B) ] A
// with d the 4x4 input tile.  The sum over input channels of the element-wise products is, for each of the 16 positions (xi, nu)
// of the transformed tile, a plain GEMM  M[xi,nu] (tiles x Cout) = V[xi,nu] (tiles x Cin) * U[xi,nu] (Cin x Cout)  -- that part
// runs on v_mfma_f32_32x32x2_f32; the transforms are a few dozen additions per tile and channel.
//
//   * block = 256 threads = 4 wavefronts (one per SIMD) on an 8 x 32 output tile = 4 x 16 = 64 Winograd tiles, 64 output
//     channels; wavefront w owns 32 tiles (two tile rows) x 32 output channels for ALL 16 positions: 16 accumulator
//     fragments = 256 registers (the AccVGPR half of the unified file), so the output transform is register-local --
//     D register r of every fragment belongs to the same (tile, channel);
//   * input channels in chunks of 8: the raw 10 x 34 patch is register-prefetched two chunks ahead, written to LDS one
//     chunk ahead, and transformed by all 256 threads (thread = tile x channel pair: 16 ds_read_b64, 32 packed additions,
//     16 ds_write_b64, bank-conflict-free) into the other V[16][64 tiles][8] buffer WHILE the MFMAs of the current chunk run
//     -- the slices of the transform are interleaved into the MFMA loop;
//   * the pre-transformed weights U (l3c_conv_wino_pack_weights: 16/9 of the 3x3 weights, MFMA fragment order) stream through
//     a double-buffered LDS slab by LDS-DMA, one chunk ahead;
//   * per position one ds_read_b128 of V and one of U feed four MFMAs (the k-ordering trick of conv_mfma.hip).
// A dilated conv is the dense conv on each of the dil x dil interleaved sub-grids of the image: same kernel, strided indexing.
// fp32 throughout; the result differs from the direct convolution by rounding only (transform coefficients are 1, 1/2).
#include "l3c_common.h"

namespace {

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

struct WinoParams {
    const float *in;
    const float *u;
    const float *bias;
    const float *res;
    float *out;
    int in_cstride, in_coff, res_cstride, res_coff, out_cstride, out_coff;
    int B, H, W, Cin, Cout;
    int dil;          // 1, 2 or 4: the output grid splits into dil x dil interleaved sub-grids, each an ordinary 3x3 conv
    int epilogue;
    int tiles_x, tiles_y, n_chunks_o, total_blocks;
};

constexpr int WT_H = 8, WT_W = 32;               // output tile of a block
constexpr int WP_H = WT_H + 2, WP_W = WT_W + 2;  // input patch
constexpr int WCK = 8;                           // input channels per chunk
constexpr int PSR = 12;                          // LDS stride of a raw patch pixel (floats)
constexpr int PSV = 8;                           // LDS stride of a transformed tile (floats), its two 4-float groups swizzled
constexpr int RAW_FLOATS = WP_H * WP_W * PSR;
constexpr int V_FLOATS = 16 * 64 * PSV;          // one buffer (two: the next chunk is transformed during the MFMA loop)
constexpr int U_FLOATS = 16 * 2 * 64 * 4;        // packed weights of one input chunk x one 64-channel output chunk
constexpr int WINO_LDS_BYTES = (2 * RAW_FLOATS + 2 * V_FLOATS) * 4;   // 98 176 bytes

__device__ __forceinline__ int xcd_remap_w(int bid, int total) {
    const int q = total >> 3, r = total & 7;
    const int xcd = bid & 7, slot = bid >> 3;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + slot;
}

// RELU / RES / SHUFFLE: the epilogue variant, compile-time (no per-element selects).
template <bool RELU, bool RES, bool SHUFFLE>
__global__ __launch_bounds__(256, 1) void conv_wino_kernel(const WinoParams p) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float *raw = lds;                            // two buffers: patch of chunk k in raw[k & 1]
    float *V = lds + 2 * RAW_FLOATS;             // two buffers

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int half = lane >> 5, lx = lane & 31;
    const int mi = wave & 1, nj = wave >> 1;
    const int n_cc = p.Cin / WCK;

    int w = xcd_remap_w(blockIdx.x, p.total_blocks);
    const int tiles = p.tiles_x * p.tiles_y;
    const int tile = w % tiles;
    w /= tiles;
    const int phase = w % (p.dil * p.dil);   // which of the dil x dil sub-grids (dilated conv = dense conv on each of them)
    w /= p.dil * p.dil;
    const int chunk_o = w % p.n_chunks_o;
    const int b = w / p.n_chunks_o;
    const int dil = p.dil, py = phase / dil, px = phase % dil;
    const int sy0 = (tile / p.tiles_x) * WT_H, sx0 = (tile % p.tiles_x) * WT_W;   // tile origin in sub-grid coordinates

    constexpr int NIT = (WP_H * WP_W * 2 + 255) / 256;
    f32x4 stage_regs[NIT];
    // this thread's patch elements: offsets relative to the image / channel-chunk base, fixed for the whole block
    int64_t patch_off[NIT];
    bool patch_ok[NIT];
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int i = tid + it * 256;
        const int c4 = i & 1, pix = i >> 1;
        const int r = pix / WP_W, ci = pix % WP_W;
        const int iy = py + dil * (sy0 - 1 + r), ix = px + dil * (sx0 - 1 + ci);
        patch_ok[it] = pix < WP_H * WP_W && iy >= 0 && iy < p.H && ix >= 0 && ix < p.W;
        patch_off[it] = patch_ok[it] ? ((int64_t)iy * p.W + ix) * p.in_cstride + c4 * 4 : 0;   // padding: any valid address
    }
    const float *in_b = p.in + (size_t)b * p.H * p.W * p.in_cstride + p.in_coff;
    // B operands (transformed weights): the packed layout IS the MFMA fragment order, so position q of chunk cc is ONE
    // coalesced 16-byte load per lane straight into the operand registers (uniform base + per-lane offset) -- no LDS staging.
    // Buffer addressing: uniform descriptor + per-lane byte offset (fixed) + scalar offset -- no vector address arithmetic.
    // Every load inside the chunk loop is UNCONDITIONAL (the last chunks re-request a slice they already have): a load
    // behind a branch would make the compiler's count of outstanding loads imprecise and turn its waits into vmcnt(0).
    const auto u_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(p.u + (size_t)chunk_o * n_cc * U_FLOATS), 0,
                                                          n_cc * U_FLOATS * 4, 0x00020000);
    const int u_lane = (nj * 256 + lane * 4) * 4;
    f32x4 bq[16];
    auto fetch_b = [&](int cc, int q) {
        bq[q] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(u_rsrc, u_lane, (cc * U_FLOATS + q * 512) * 4, 0));
    };
    auto fetch_patch_piece = [&](int cc, int it) {   // zero padding is applied by store_patch
        stage_regs[it] = *reinterpret_cast<const f32x4 *>(in_b + cc * WCK + patch_off[it]);
    };
    auto fetch_patch = [&](int cc) {
#pragma unroll
        for (int it = 0; it < NIT; ++it) fetch_patch_piece(cc, it);
    };
    static_assert(NIT == 3, "the patch prefetch pieces are dealt out by hand below");
    auto store_patch = [&](int cc) {
        float *dst = raw + (cc & 1) * RAW_FLOATS;
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int i = tid + it * 256;
            const int c4 = i & 1, pix = i >> 1;
            const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
            if (pix < WP_H * WP_W) *reinterpret_cast<f32x4 *>(&dst[pix * PSR + c4 * 4]) = patch_ok[it] ? stage_regs[it] : zero;
        }
    };
    // Input transform B^T d B (B^T = [1 0 -1 0; 0 1 1 0; 0 -1 1 0; 0 1 0 -1]) of this thread's (tile, channel pair): wave w
    // takes tile row w, lane = (tile column, channel pair) -- 8 tiles x 4 pairs per 32 lanes cover all 64 LDS banks on the
    // reads (pixel stride 12 floats) and on the writes (tile stride 8 floats).  The two 4-float groups of a tile are swapped
    // for tiles 8..15 of a row (XOR swizzle), which makes the MFMA A-fragment reads (16 tiles x 16 B per pass) conflict-free.
    const int t_tx = lane >> 2, t_cq = lane & 3, t_ty = wave;
    const float *t_src = raw + ((2 * t_ty) * WP_W + 2 * t_tx) * PSR + 2 * t_cq;
    const int t_dst = (t_ty * 16 + t_tx) * PSV + (((t_cq >> 1) ^ (t_tx >> 3)) * 4) + (t_cq & 1) * 2;
    f32x2 d[4][4];
    auto transform_load = [&](int cc, int i) {
        const float *src = t_src + (cc & 1) * RAW_FLOATS;
#pragma unroll
        for (int j = 0; j < 4; ++j) d[i][j] = *reinterpret_cast<const f32x2 *>(src + (i * WP_W + j) * PSR);
    };
    auto transform_rows_col = [&](int j) {   // column j of d <- B^T d
        const f32x2 b0 = d[0][j] - d[2][j], b1 = d[1][j] + d[2][j], b2 = d[2][j] - d[1][j], b3 = d[1][j] - d[3][j];
        d[0][j] = b0;
        d[1][j] = b1;
        d[2][j] = b2;
        d[3][j] = b3;
    };
    auto transform_rows = [&]() {   // d <- B^T d  (in place, column by column)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const f32x2 b0 = d[0][j] - d[2][j], b1 = d[1][j] + d[2][j], b2 = d[2][j] - d[1][j], b3 = d[1][j] - d[3][j];
            d[0][j] = b0;
            d[1][j] = b1;
            d[2][j] = b2;
            d[3][j] = b3;
        }
    };
    auto transform_cols_row = [&](int i) {   // row i of d <- d B
        const f32x2 c0 = d[i][0] - d[i][2], c1 = d[i][1] + d[i][2], c2 = d[i][2] - d[i][1], c3 = d[i][1] - d[i][3];
        d[i][0] = c0;
        d[i][1] = c1;
        d[i][2] = c2;
        d[i][3] = c3;
    };
    auto transform_write = [&](float *Vdst, int i) {   // finished row i -> positions 4 i .. 4 i + 3
        float *dst = Vdst + t_dst + (i * 4) * 64 * PSV;
#pragma unroll
        for (int j = 0; j < 4; ++j) *reinterpret_cast<f32x2 *>(dst + j * 64 * PSV) = d[i][j];
    };
    auto transform_store = [&](float *Vdst, int i) {   // row i of (B^T d) B -> positions 4 i .. 4 i + 3
        float *dst = Vdst + t_dst + (i * 4) * 64 * PSV;
        *reinterpret_cast<f32x2 *>(dst + 0 * 64 * PSV) = d[i][0] - d[i][2];
        *reinterpret_cast<f32x2 *>(dst + 1 * 64 * PSV) = d[i][1] + d[i][2];
        *reinterpret_cast<f32x2 *>(dst + 2 * 64 * PSV) = d[i][2] - d[i][1];
        *reinterpret_cast<f32x2 *>(dst + 3 * 64 * PSV) = d[i][1] - d[i][3];
    };

    f32x16 acc[16];
#pragma unroll
    for (int q = 0; q < 16; ++q)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[q][r] = 0.0f;

    // Prologue.  Invariants at the start of the MFMA loop of chunk c:  V[c & 1] and slab c & 1 complete and visible; the operand
    // fragments of its first position pair loaded; raw[(c+1) & 1] = patch of chunk c + 1, visible; the staging registers free.
    fetch_patch(0);
#pragma unroll
    for (int q = 0; q < 16; ++q) fetch_b(0, q);
    store_patch(0);
    if (n_cc > 1) fetch_patch(1);
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i) transform_load(0, i);
    transform_rows();
#pragma unroll
    for (int i = 0; i < 4; ++i) transform_store(V, i);
    if (n_cc > 1) store_patch(1);
    __syncthreads();

    // A fragment of (tile, half): the 4-float group `half`, swizzled like the writes
    const int a_tile = 32 * mi + lx;
    const float *a_lane = V + a_tile * PSV + ((half ^ ((a_tile >> 3) & 1)) * 4);
    // One wavefront per SIMD: nothing else hides this wave's non-MFMA instructions, and issue is in order -- so they are
    // dealt out BETWEEN the MFMAs (each keeps the matrix pipe busy for 64 cycles).  Positions are taken in pairs so that
    // consecutive MFMAs alternate between two accumulators; the 8 gaps of a pair carry
    //   1, 2: the A fragment reads of the next pair            3, 5: LDS traffic of the next chunk's input transform
    //   4, 6: its arithmetic / the patch prefetch of chunk c + 2 (pairs 0, 1) / its store to LDS (pair 6)
    //   7, 8: the B operands of the SAME positions for chunk c + 1 -- each register quad is reloaded right after its last
    //         MFMA of this chunk was issued, a whole chunk (> 4000 cycles) before its next use
    // The chunk's ONE barrier sits between its pairs 6 and 7: by then this wave has written its share of V[(c+1) & 1] and
    // stored the patch of chunk c + 2, so the last pair's 8 MFMAs run while the barrier releases and the first fragments of
    // chunk c + 1 arrive -- the matrix pipe does not drain at the chunk boundary.
    f32x4 a0[2], a1[2];
    a0[0] = *reinterpret_cast<const f32x4 *>(a_lane);
    a1[0] = *reinterpret_cast<const f32x4 *>(a_lane + 64 * PSV);
    for (int cc = 0; cc < n_cc; ++cc) {
        const bool more = cc + 1 < n_cc, more2 = cc + 2 < n_cc;
        const int cc_b = more ? cc + 1 : cc, cc_p = more2 ? cc + 2 : cc;   // what the (unconditional) prefetches ask for
        const float *a_cur = a_lane + (cc & 1) * V_FLOATS;
        const float *a_nxt = a_lane + ((cc + 1) & 1) * V_FLOATS;
        float *v_next = V + ((cc + 1) & 1) * V_FLOATS;
#define L3C_WINO_MFMA(Q, T, A, B)                                                                  \
    __builtin_amdgcn_sched_barrier(0);                                                             \
    acc[Q] = __builtin_amdgcn_mfma_f32_32x32x2f32((A)[T], (B)[T], acc[Q], 0, 0, 0);                 \
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int pp = 0; pp < 8; ++pp) {
            const int q = 2 * pp, cur = pp & 1, nxt = cur ^ 1;
            const f32x4 A0 = a0[cur], A1 = a1[cur], B0 = bq[q], B1 = bq[q + 1];
            if (pp == 7) {
                // everything chunk c + 1 needs from this wave is issued: patch c + 2 stored, V[(c+1) & 1] written -- the
                // barrier (LDS operations only: the B loads in flight stay in flight), then the first fragments of chunk c + 1
                __syncthreads();
                if (more) {
                    a0[nxt] = *reinterpret_cast<const f32x4 *>(a_nxt);
                    a1[nxt] = *reinterpret_cast<const f32x4 *>(a_nxt + 64 * PSV);
                }
            }
            L3C_WINO_MFMA(q, 0, A0, B0)
            if (pp < 7) a0[nxt] = *reinterpret_cast<const f32x4 *>(a_cur + (q + 2) * 64 * PSV);
            L3C_WINO_MFMA(q + 1, 0, A1, B1)
            if (pp < 7) a1[nxt] = *reinterpret_cast<const f32x4 *>(a_cur + (q + 3) * 64 * PSV);
            L3C_WINO_MFMA(q, 1, A0, B0)
            if (more) {
                if (pp < 2) transform_load(cc + 1, 2 * pp);
                if (pp == 5) transform_write(v_next, 0);
                if (pp == 6) transform_write(v_next, 2);
            }
            L3C_WINO_MFMA(q + 1, 1, A1, B1)
            if (pp == 0) fetch_patch_piece(cc_p, 0);
            if (pp == 1) fetch_patch_piece(cc_p, 2);
            if (pp == 6 && more2) store_patch(cc + 2);   // raw[c & 1]: read by the transform of chunk c during loop c - 1
            if (more) {
                if (pp == 2 || pp == 3) transform_rows_col(2 * (pp - 2));
                if (pp == 4 || pp == 5) transform_cols_row(2 * (pp - 4));
            }
            L3C_WINO_MFMA(q, 2, A0, B0)
            if (more) {
                if (pp < 2) transform_load(cc + 1, 2 * pp + 1);
                if (pp == 5) transform_write(v_next, 1);
                if (pp == 6) transform_write(v_next, 3);
            }
            L3C_WINO_MFMA(q + 1, 2, A1, B1)
            if (pp == 0) fetch_patch_piece(cc_p, 1);
            if (more) {
                if (pp == 2 || pp == 3) transform_rows_col(2 * (pp - 2) + 1);
                if (pp == 4 || pp == 5) transform_cols_row(2 * (pp - 4) + 1);
            }
            L3C_WINO_MFMA(q, 3, A0, B0)
            fetch_b(cc_b, q);
            L3C_WINO_MFMA(q + 1, 3, A1, B1)
            fetch_b(cc_b, q + 1);
        }
#undef L3C_WINO_MFMA
    }

    // ---- output transform Y = A^T M A (A^T = [1 1 1 0; 0 1 -1 -1]), register-local, + bias / ReLU / residual / store ----
    const int co = chunk_o * 64 + nj * 32 + lx;
    if (co >= p.Cout) return;
    const float bias = p.bias[co];
    const bool interior = py + dil * (sy0 + WT_H - 1) < p.H && px + dil * (sx0 + WT_W - 1) < p.W;
    auto y_of = [&](int r, float (&y)[2][2]) {
        float t0[4], t1[4];
#pragma unroll
        for (int nu = 0; nu < 4; ++nu) {
            t0[nu] = (acc[0 + nu][r] + acc[4 + nu][r]) + acc[8 + nu][r];
            t1[nu] = (acc[4 + nu][r] - acc[8 + nu][r]) - acc[12 + nu][r];
        }
        y[0][0] = (t0[0] + t0[1]) + t0[2];
        y[0][1] = (t0[1] - t0[2]) - t0[3];
        y[1][0] = (t1[0] + t1[1]) + t1[2];
        y[1][1] = (t1[1] - t1[2]) - t1[3];
    };
    if (interior) {
        // tile (ty, tx) of this wave: ty = 2 mi + (r >> 3), tx = (r & 3) + 8 ((r >> 2) & 1) + 4 half.  Buffer addressing:
        // descriptor at the block's first output pixel (uniform), ONE per-lane byte offset, and a scalar offset per
        // (row, pixel) -- no vector address arithmetic.  Pixel shuffle (dil = 1): conv pixel (oy, ox), channel co -> pixel
        // (2 oy + (co >> 1 & 1), 2 ox + (co & 1)), channel co >> 2 of a 2H x 2W image, i.e. the same walk with doubled strides
        // and the sub-pixel folded into the lane offset.
        constexpr int S = SHUFFLE ? 2 : 1;
        const int64_t col_b = (int64_t)S * dil * p.out_cstride * 4, row_b = (int64_t)S * dil * (S * p.W) * p.out_cstride * 4;
        float *o_blk = p.out + (((size_t)b * (S * p.H) + S * (py + dil * sy0)) * (S * p.W) + S * (px + dil * sx0)) * p.out_cstride +
                       p.out_coff + (SHUFFLE ? chunk_o * 16 : chunk_o * 64);
        const auto o_rsrc = __builtin_amdgcn_make_buffer_rsrc(o_blk, 0, 0x7fffffff, 0x00020000);
        const int cw = nj * 32 + lx;   // channel inside the 64-channel chunk
        const int o_lane = (int)(4 * mi * row_b + 8 * half * col_b) +
                           (SHUFFLE ? ((((cw >> 1) & 1) * (2 * p.W) + (cw & 1)) * p.out_cstride + (cw >> 2)) * 4 : cw * 4);
        // all 64 residual values of the lane first (one round trip instead of 64), then transform + store
        float resv[16][2][2];
        if constexpr (RES) {
            const int64_t rrow_b = (int64_t)dil * p.W * p.res_cstride * 4, rcol_b = (int64_t)dil * p.res_cstride * 4;
            const float *r_blk = p.res + (((size_t)b * p.H + py + dil * sy0) * p.W + px + dil * sx0) * p.res_cstride + p.res_coff + chunk_o * 64;
            const auto r_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(r_blk), 0, 0x7fffffff, 0x00020000);
            const int r_lane = (int)(4 * mi * rrow_b + 8 * half * rcol_b) + cw * 4;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int ty2 = 2 * (r >> 3), tx2 = 2 * ((r & 3) + 8 * ((r >> 2) & 1));
#pragma unroll
                for (int dy = 0; dy < 2; ++dy)
#pragma unroll
                    for (int dx = 0; dx < 2; ++dx)
                        resv[r][dy][dx] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(
                            r_rsrc, r_lane, (int)((ty2 + dy) * rrow_b + (tx2 + dx) * rcol_b), 0));
            }
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            float y[2][2];
            y_of(r, y);
            const int ty2 = 2 * (r >> 3), tx2 = 2 * ((r & 3) + 8 * ((r >> 2) & 1));
#pragma unroll
            for (int dy = 0; dy < 2; ++dy)
#pragma unroll
                for (int dx = 0; dx < 2; ++dx) {
                    float v = y[dy][dx] + bias;
                    if constexpr (RELU) v = fmaxf(v, 0.0f);
                    if constexpr (RES) v = v + resv[r][dy][dx];
                    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), o_rsrc, o_lane,
                                                          (int)((ty2 + dy) * row_b + (tx2 + dx) * col_b), 0);
                }
            if (r & 1) __builtin_amdgcn_sched_barrier(0);   // two tiles at a time (keeps the register demand flat)
        }
        return;
    }
    // tiles that stick out of the image: per-element bounds checks and addresses
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int tl = (r & 3) + 8 * (r >> 2) + 4 * half;       // tile inside this wave's 32
        const int ty = 2 * mi + (tl >> 4), tx = tl & 15;
        float y[2][2];
        y_of(r, y);
#pragma unroll
        for (int dy = 0; dy < 2; ++dy)
#pragma unroll
            for (int dx = 0; dx < 2; ++dx) {
                const int oy = py + dil * (sy0 + 2 * ty + dy), ox = px + dil * (sx0 + 2 * tx + dx);
                if (oy >= p.H || ox >= p.W) continue;
                float v = y[dy][dx] + bias;
                if constexpr (RELU) v = fmaxf(v, 0.0f);
                if constexpr (RES) v = v + p.res[(((size_t)b * p.H + oy) * p.W + ox) * p.res_cstride + p.res_coff + co];
                if constexpr (SHUFFLE) {
                    const size_t oyy = 2 * oy + ((co >> 1) & 1), oxx = 2 * ox + (co & 1);
                    p.out[(((size_t)b * 2 * p.H + oyy) * 2 * p.W + oxx) * p.out_cstride + p.out_coff + (co >> 2)] = v;
                } else {
                    p.out[(((size_t)b * p.H + oy) * p.W + ox) * p.out_cstride + p.out_coff + co] = v;
                }
            }
    }
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

extern "C" {

int64_t l3c_conv_wino_packed_words(int Cout, int Cin) { return (int64_t)((Cout + 63) / 64) * (Cin / 8) * U_FLOATS; }

int l3c_conv_wino_pack_weights(const float *w_oihw, int Cout, int Cin, float *packed, l3c_stream_t stream) {
    L3C_REQUIRE(w_oihw && packed, "null pointer");
    L3C_REQUIRE(Cout > 0 && Cin > 0 && Cin % 8 == 0, "Cin must be a multiple of 8");
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
    L3C_REQUIRE(d->Cin > 0 && d->Cin % WCK == 0, "Cin must be a multiple of 8");
    L3C_REQUIRE(d->in_cstride % 4 == 0 && d->in_coff % 4 == 0, "input channel stride/offset must be multiples of 4");
    L3C_REQUIRE(d->in_coff + d->Cin <= d->in_cstride, "input channel slice out of range");
    L3C_REQUIRE(!(d->epilogue & L3C_EPI_RESIDUAL) || d->residual, "residual epilogue without residual pointer");
    L3C_REQUIRE(!((d->epilogue & L3C_EPI_PIXEL_SHUFFLE) && (d->epilogue & L3C_EPI_RESIDUAL)), "pixel shuffle + residual not provided");
    L3C_REQUIRE(!(d->epilogue & L3C_EPI_PIXEL_SHUFFLE) || d->Cout % 4 == 0, "pixel shuffle needs Cout % 4 == 0");
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
    p.tiles_x = ((p.W + p.dil - 1) / p.dil + WT_W - 1) / WT_W;   // tiles of the (largest) sub-grid
    p.tiles_y = ((p.H + p.dil - 1) / p.dil + WT_H - 1) / WT_H;
    p.n_chunks_o = (p.Cout + 63) / 64;
    const int64_t total = (int64_t)p.tiles_x * p.tiles_y * p.dil * p.dil * p.n_chunks_o * p.B;
    L3C_REQUIRE(total < (1ll << 31), "grid too large");
    p.total_blocks = (int)total;
    typedef void (*kernel_t)(const WinoParams);
    static const kernel_t variants[5] = {conv_wino_kernel<false, false, false>, conv_wino_kernel<true, false, false>,
                                         conv_wino_kernel<false, true, false>, conv_wino_kernel<true, true, false>,
                                         conv_wino_kernel<false, false, true>};
    static bool attr_set[5] = {false, false, false, false, false};   // > 64 KB of dynamic LDS needs the opt-in
    const bool relu = d->epilogue & L3C_EPI_RELU, res = d->epilogue & L3C_EPI_RESIDUAL, shuffle = d->epilogue & L3C_EPI_PIXEL_SHUFFLE;
    L3C_REQUIRE(!(shuffle && relu), "pixel shuffle + ReLU not provided");
    const int v = shuffle ? 4 : (relu ? 1 : 0) + (res ? 2 : 0);
    if (!attr_set[v]) {
        const int rc = l3c::check_hip(hipFuncSetAttribute(reinterpret_cast<const void *>(variants[v]),
                                                          hipFuncAttributeMaxDynamicSharedMemorySize, WINO_LDS_BYTES),
                                      "hipFuncSetAttribute");
        if (rc != L3C_OK) return rc;
        attr_set[v] = true;
    }
    hipLaunchKernelGGL(variants[v], dim3((unsigned)total), dim3(256), WINO_LDS_BYTES, l3c::as_stream(stream), p);
    return l3c::check_launch("conv_wino_kernel");
}
}
