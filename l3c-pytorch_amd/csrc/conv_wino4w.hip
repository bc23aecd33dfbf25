// conv_wino4w.hip -- PROBE (round 6, test-only library): Winograd F(4x4,3x3) on v_mfma_f32_32x32x2_f32.
//
// The round-5 verdict asked for the alternative decomposition of csrc/conv_wino4.hip to be BUILT on one layer shape and measured, instead
// of priced on paper a third time: "a wavefront owns 9 of 36 positions for 32 tiles x 32 channels = 144 accumulators, four wavefronts
// cover the 36 positions, M exchanged through LDS for the output transform: half the MFMA, weight-load and A-read instructions per FLOP".
// This is that kernel, for 3x3 / stride 1 / dilation 1 layers with Cin % 16 == 0, bias (+ ReLU):
//
//   * block = 512 threads = 8 wavefronts on a 16 x 32 output tile = 4 x 8 Winograd tiles (32 = the M dimension of the MFMA), 64 output
//     channels.  Wavefront w = (pg = w & 3, h = w >> 2) owns positions 9 pg .. 9 pg + 8 of the 36 for output channels 32 h .. 32 h + 31:
//     9 accumulator fragments of 16 registers = 144 registers, the same budget as conv_wino4_kernel -- but 132 KB of LDS and 8
//     wavefronts: ONE block per CU, both wavefronts of a SIMD in the same block;
//   * per 8-channel chunk a wavefront issues 36 MFMAs of 64 cycles (conv_wino4_kernel: 72 of 32), 9 ds_read_b128 of A (V in LDS as
//     [position][k half][tile][k step]: one read = the four k-steps of a position) and 9 16-byte weight loads (18 + 18 there); the input
//     transform is the same arithmetic on 512 threads (thread = tile x channel x half of the rows xi);
//   * the price: the output transform needs all 36 positions of a (tile, channel) and they live in four wavefronts.  Four rounds (one
//     per tile row): every wavefront writes its accumulators of the round's 8 tiles to X[position][tile][channel] (73 728 bytes -- exactly
//     the two V buffers, free between the last chunk and the next tile's first transform), barrier, thread (tile = wavefront, channel =
//     lane) reads its 36 values, transforms, stores 16 pixels (a wavefront's store = one pixel's 64 channels = 256 contiguous bytes),
//     barrier.  The matrix pipe idles through all of it: there is no second block on the CU to cover it.
// Chunk pipeline as in conv_wino4_kernel: patch g + 3 in flight, g + 2 stored to LDS, g + 1 transformed while chunk g multiplies; a block
// walks several horizontally adjacent tiles with the pipeline running through the tile boundaries (the next tile's first transform
// happens after the epilogue, because the epilogue's exchange uses the V buffers).
// Reference call sites this layer shape covers: modules/edsr.py:63-89 (ResBlock convs), modules/net.py:136-148, :173-184.
#include "l3c_common.h"
#include "../../include/l3c_xcheck.h"

#include <type_traits>

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

struct W4wParams {
    const float *in;
    const float *u;
    const float *bias;
    float *out;
    int in_cstride, in_coff, out_cstride, out_coff;
    int B, H, W, Cin, Cout;
    int tiles_x, tiles_y, tpb, groups_x;
    int relu;
};

constexpr int OH = 16, OW = 32;                 // output tile of a block: 4 x 8 Winograd tiles
constexpr int PH = OH + 2, PWD = OW + 2;        // input patch
constexpr int CK = 8;                           // input channels per chunk = four k-steps of the 32x32x2 MFMA
constexpr int NP = 36;
constexpr int PSR = 12;                         // LDS stride of a raw patch pixel (floats)
constexpr int RAW_FLOATS = PH * PWD * PSR;      // 7344
constexpr int VP = 2 * 32 * 4;                  // floats of one position in V: [k half][tile][k step]
constexpr int V_FLOATS = NP * VP;               // 9216
constexpr int V_OFF0 = 0, V_OFF1 = V_FLOATS, RAW_OFF0 = 2 * V_FLOATS, RAW_OFF1 = 2 * V_FLOATS + RAW_FLOATS;
constexpr int LDS_FLOATS = 2 * V_FLOATS + 2 * RAW_FLOATS;
constexpr int W4W_LDS_BYTES = LDS_FLOATS * 4;   // 132 480
static_assert(NP * 8 * 64 == 2 * V_FLOATS, "the exchange of one round (36 positions x 8 tiles x 64 channels) is exactly the two V buffers");
constexpr int U_CHUNK_FLOATS = NP * 2 * 64 * 4; // packed weights of one input chunk x one 64-channel output chunk: [position][h][lane][4]
constexpr int OOB = 0x7ffffff0;
constexpr int ROW_SLOTS = 2 * PWD;              // 16-byte pieces of a patch row: 68
constexpr int NIT = 3;                          // pieces per fetching thread: rows r0, r0 + 6, r0 + 12
constexpr int BR = 6;                           // depth of the weight ring (positions ahead); a chunk has 9: slot (q + 3 par) % 6
constexpr int AR = 3;                           // depth of the A ring
// Development probes (csrc/build.py --variant NAME "-DL3C_W4W_PROBE=N"; WRONG results, only the time means something): bit 0 no output
// transform / exchange (one store per tile keeps the accumulators alive), 1 the exchange and the transform without the global stores, 2 no input
// transform inside the chunk loop, 3 no MFMA (one v_add in its place), 4 no patch fetch, 5 no weight loads inside the loop
#ifndef L3C_W4W_PROBE
#define L3C_W4W_PROBE 0
#endif
// Output transform: 0 = first version (four rounds of 8 tiles through one 73 728-byte buffer, two barriers a round), 1 = eight half rounds through
// two buffers, the next half round's writes issued before this one's reads (one barrier a half round).  [measured, 64 -> 64 at 256x384x32, tiles
// per block 6 / 12: version 0 0.747 / 0.734 ms, version 1 0.766 / 0.755 ms (profiles/r06_wino4w_probe.log): the pipelined form reads more (30 values
// per thread and half round instead of 36 per round) and wins nothing back -- left at 0]
#ifndef L3C_W4W_EPI
#define L3C_W4W_EPI 0
#endif

// one 6-vector through A^T (output transform), four outputs -- as conv_wino4.hip
__device__ __forceinline__ void at6(float m0, float m1, float m2, float m3, float m4, float m5, float (&y)[4]) {
    const float s1 = m1 + m2, d1 = m1 - m2;
    y[0] = (m0 + s1) + (m3 + m4);
    y[1] = d1 + __builtin_fmaf(-2.0f, m4, 0.5f * m3);
    y[2] = s1 + __builtin_fmaf(4.0f, m4, 0.25f * m3);
    y[3] = (d1 + __builtin_fmaf(-8.0f, m4, 0.125f * m3)) + m5;
}

__global__ __launch_bounds__(512, 2) void conv_wino4w_kernel(const W4wParams p) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int pg = wave & 3, h = wave >> 2;
    const int n_cc = p.Cin / CK;

    // block -> (image, tile row, group of tiles)
    unsigned w = blockIdx.x;
    const unsigned gx = w % (unsigned)p.groups_x;
    w /= (unsigned)p.groups_x;
    const unsigned t_y = w % (unsigned)p.tiles_y;
    const int b = (int)(w / (unsigned)p.tiles_y);
    const int sy0 = (int)t_y * OH;
    const int tx_first = (int)gx * p.tpb;
    const int n_t = min(p.tpb, p.tiles_x - tx_first);

    // ---- patch fetch: thread (r0 = tid / 68 < 6, slot j = tid % 68: column j / 2, half j & 1) fetches the 16-byte pieces of the patch
    // pixels (r0 + 6 k, column), k = 0..2; zero padding = the descriptor's range check (see conv_wino4.hip)
    const int pf_r0 = tid / ROW_SLOTS, pf_j = tid - pf_r0 * ROW_SLOTS;
    const bool pf_thread = pf_r0 < 6;
    const int in_pix_b = p.in_cstride * 4;
    const int pf_row_b = 6 * p.W * in_pix_b;
    const int pf_lds = (pf_r0 * PWD + (pf_j >> 1)) * PSR + (pf_j & 1) * 4;
    int patch_off;
    auto set_patch_tile = [&](int t) {
        const int iy = sy0 - 1 + pf_r0, ix = (tx_first + t) * OW - 1 + (pf_j >> 1);
        const bool ok = pf_thread && ix >= 0 && ix < p.W;
        patch_off = ok ? (iy * p.W + ix) * in_pix_b + (pf_j & 1) * 16 : (int)0xC0000000;
    };
    int pf_tile = 0, pf_cc = 0;
    auto pf_advance = [&]() {
        if (++pf_cc == n_cc) {
            if (pf_tile + 1 < n_t) {
                pf_cc = 0;
                set_patch_tile(++pf_tile);
            } else {
                pf_cc = n_cc - 1;   // past the block's last chunk: the loads stay unconditional, their data is never used
            }
        }
    };
    const auto in_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float *>(p.in + (size_t)b * p.H * p.W * p.in_cstride + p.in_coff), 0, p.H * p.W * p.in_cstride * 4, 0x00020000);
    f32x4 stage[NIT];
    auto fetch_patch = [&]() {
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            if constexpr (L3C_W4W_PROBE & 16) stage[it] = f32x4{1.f, 2.f, 3.f, 4.f};
            else stage[it] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(in_rsrc, patch_off + it * pf_row_b, pf_cc * CK * 4, 0));
        }
        pf_advance();
    };
    auto store_patch = [&](float *dst) {
        if (pf_thread) {
#pragma unroll
            for (int it = 0; it < NIT; ++it) *reinterpret_cast<f32x4 *>(&dst[pf_lds + it * (6 * PWD * PSR)]) = stage[it];
        }
    };

    // ---- B operands: [chunk][position][h][lane][4 k-steps]; uniform descriptor + per-lane byte offset + scalar offset
    const auto u_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(p.u), 0, n_cc * U_CHUNK_FLOATS * 4, 0x00020000);
    const int u_lane = (h * 64 + lane) * 16;
    f32x4 b_ring[BR];
    auto fetch_b = [&](int cc, int q) {   // position q of the wavefront's nine, chunk cc
        return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(u_rsrc, u_lane, (cc * NP + 9 * pg + q) * (2 * 64 * 16), 0));
    };

    // ---- input transform roles: thread = (tile (ty = wave & 3, tx = lane >> 3), channel c = lane & 7, half th = wave >> 2 of the rows xi)
    const int t_c = lane & 7, t_tx = lane >> 3, t_ty = wave & 3;
    const int t_src = ((4 * t_ty + h) * PWD + 4 * t_tx) * PSR + t_c;
    const int t_dst = (3 * h) * 6 * VP + (t_c & 1) * 128 + (t_ty * 8 + t_tx) * 4 + (t_c >> 1);

    // ---- A fragments: lane (tile = lane & 31, k half = lane >> 5), the wavefront's positions 9 pg ..
    const int a_lane = 9 * pg * VP + (lane >> 5) * 128 + (lane & 31) * 4;
    f32x4 a_ring[AR];

    f32x16 acc[9];
    const int co_lane = h * 32 + (lane & 31);
    const float bias_init = (pg == 0 && co_lane < p.Cout) ? p.bias[co_lane] : 0.0f;   // rides in position (1, 1) = 7: column 1 of A^T is all ones

    const auto o_rsrc = __builtin_amdgcn_make_buffer_rsrc(p.out + ((size_t)b * p.H + sy0) * p.W * p.out_cstride + p.out_coff, 0, OOB, 0x00020000);

    auto body = [&](auto th_c) __attribute__((always_inline)) {
        constexpr int TH = decltype(th_c)::value;
        float T[3][6];
        float L[2][5];
        auto tr_load = [&](const float *src, int col) {
#pragma unroll
            for (int i = 0; i < 5; ++i) L[col & 1][i] = src[(i * PWD + col) * PSR];
        };
        auto tr_col = [&](int col) {
            const float(&l)[5] = L[col & 1];
            if constexpr (TH == 0) {   // xi = 0, 1, 2 from raw rows 0 .. 4
                T[0][col] = __builtin_fmaf(2.0f, l[0] + l[4], __builtin_fmaf(3.0f, l[3] - l[1], -4.0f * l[2]));
                T[1][col] = __builtin_fmaf(2.0f, l[4], __builtin_fmaf(5.0f, l[3], __builtin_fmaf(-2.0f, l[1], l[2])));
                T[2][col] = __builtin_fmaf(-2.0f, l[4], __builtin_fmaf(-2.0f, l[1], __builtin_fmaf(5.0f, l[2], -l[3])));
            } else {                   // xi = 3, 4, 5 from raw rows 1 .. 5
                const float pp_ = l[0] - l[2], qq_ = l[1] - l[3];
                T[0][col] = __builtin_fmaf(2.0f, pp_, qq_);
                T[1][col] = __builtin_fmaf(-2.0f, qq_, pp_);
                T[2][col] = __builtin_fmaf(2.0f, l[0] + l[4], __builtin_fmaf(3.0f, l[3] - l[1], -4.0f * l[2]));
            }
        };
        auto tr_row = [&](float *dst, int k) {   // row xi = 3 TH + k through B^T, positions (xi, 0..5)
            const float(&w_)[6] = T[k];
            float R[6];
            R[0] = __builtin_fmaf(2.0f, w_[0] + w_[4], __builtin_fmaf(3.0f, w_[3] - w_[1], -4.0f * w_[2]));
            R[1] = __builtin_fmaf(2.0f, w_[4], __builtin_fmaf(5.0f, w_[3], __builtin_fmaf(-2.0f, w_[1], w_[2])));
            R[2] = __builtin_fmaf(-2.0f, w_[4], __builtin_fmaf(-2.0f, w_[1], __builtin_fmaf(5.0f, w_[2], -w_[3])));
            const float p_ = w_[1] - w_[3], q_ = w_[2] - w_[4];
            R[3] = __builtin_fmaf(2.0f, p_, q_);
            R[4] = __builtin_fmaf(-2.0f, q_, p_);
            R[5] = __builtin_fmaf(2.0f, w_[1] + w_[5], __builtin_fmaf(3.0f, w_[4] - w_[2], -4.0f * w_[3]));
#pragma unroll
            for (int nu = 0; nu < 6; ++nu) dst[(k * 6 + nu) * VP] = R[nu];
        };
        auto transform_all = [&](const float *raw, float *v) {   // a whole patch at once (prologue, and between two tiles)
#pragma unroll
            for (int col = 0; col < 6; ++col) {
                tr_load(raw + t_src, col);
                tr_col(col);
            }
#pragma unroll
            for (int k = 0; k < 3; ++k) tr_row(v + t_dst, k);
        };

        // ---- prologue: patches 0, 1 -> raw[0], raw[1]; patch 2 in flight; V[0] = transform of patch 0; B of the first positions
        set_patch_tile(0);
        {
            f32x4 first[2][NIT];
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                fetch_patch();
#pragma unroll
                for (int it = 0; it < NIT; ++it) first[k][it] = stage[it];
            }
            fetch_patch();
            if (pf_thread) {
#pragma unroll
                for (int k = 0; k < 2; ++k)
#pragma unroll
                    for (int it = 0; it < NIT; ++it)
                        *reinterpret_cast<f32x4 *>(&lds[(k ? RAW_OFF1 : RAW_OFF0) + pf_lds + it * (6 * PWD * PSR)]) = first[k][it];
            }
        }
#pragma unroll
        for (int q = 0; q < BR; ++q) b_ring[q] = fetch_b(0, q);
        __syncthreads();
        transform_all(lds + RAW_OFF0, lds + V_OFF0);
        __syncthreads();

        // Chunk g (buffer parity par) -- invariants at its start: V[par] complete and visible; raw[par ^ 1] = patch g + 1, visible; `stage` =
        // patch g + 2 (in flight); the prefetch pointer is at chunk g + 3; b_ring holds positions 0 .. BR - 1 of this chunk; a_ring is loaded
        // at the start.  LAST (last chunk of a tile): no transform (the epilogue's exchange uses both V buffers; the next tile's first
        // transform follows the epilogue).
        auto chunk = [&](const int cc, auto last_c, auto par_c, auto first_c) __attribute__((always_inline)) {
            constexpr bool LASTC = decltype(last_c)::value, FIRST = decltype(first_c)::value;
            constexpr bool LAST = LASTC || (L3C_W4W_PROBE & 4);   // (probe bit 2: no transform in any chunk)
            constexpr int par = decltype(par_c)::value;
            const int cc_b = cc + 1 == n_cc ? 0 : cc + 1;
            const float *a_cur = lds + (par ? V_OFF1 : V_OFF0) + a_lane;
            const float *a_nxt = lds + (par ? V_OFF0 : V_OFF1) + a_lane;
            float *v_next = lds + (par ? V_OFF0 : V_OFF1) + t_dst;
            const float *r_src = lds + (par ? RAW_OFF0 : RAW_OFF1) + t_src;   // raw[par ^ 1]
            float *r_dst = lds + (par ? RAW_OFF1 : RAW_OFF0);                 // raw[par]
            if constexpr (FIRST) {
#pragma unroll
                for (int j = 0; j < AR; ++j) a_ring[j] = *reinterpret_cast<const f32x4 *>(a_cur + j * VP);
            }
#define L3C_W4W_MFMA(Q, S)                                                                               \
    __builtin_amdgcn_sched_barrier(0);                                                                   \
    if constexpr (L3C_W4W_PROBE & 8) { asm volatile("v_add_f32 %0, %1, %2" : "+v"(acc[Q][0]) : "v"(A[S]), "v"(Bv[S])); } \
    else acc[Q] = __builtin_amdgcn_mfma_f32_32x32x2f32(A[S], Bv[S], acc[Q], 0, 0, 0);                   \
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int q = 0; q < 9; ++q) {
                const f32x4 A = a_ring[q % AR];
                const f32x4 Bv = b_ring[(q + 3 * par) % BR];
                if (q == 8) {
                    // everything chunk g + 1 needs from this wavefront is issued: V[par ^ 1] written, patch g + 2 stored
                    __syncthreads();
                    if constexpr (!LASTC) {
#pragma unroll
                        for (int j = 0; j + 1 < AR; ++j) a_ring[j] = *reinterpret_cast<const f32x4 *>(a_nxt + j * VP);
                    }
                }
                L3C_W4W_MFMA(q, 0)
                if (!LAST && q <= 5) tr_load(r_src, q);
                if (q == 1) store_patch(r_dst);
                L3C_W4W_MFMA(q, 1)
                if (!LAST && q >= 1 && q <= 6) tr_col(q - 1);
                L3C_W4W_MFMA(q, 2)
                if (!LAST && q == 6) tr_row(v_next, 0);
                if (!LAST && q == 7) tr_row(v_next, 1);
                L3C_W4W_MFMA(q, 3)
                if (!LAST && q == 7) tr_row(v_next, 2);
                // the rings are reloaded behind the position's last MFMA
                if (q + AR < 9) a_ring[q % AR] = *reinterpret_cast<const f32x4 *>(a_cur + (q + AR) * VP);
                if (!LASTC && q == 8) a_ring[AR - 1] = *reinterpret_cast<const f32x4 *>(a_nxt + (AR - 1) * VP);
                if constexpr (!(L3C_W4W_PROBE & 32)) {
                    const int nxt = q + BR;
                    b_ring[(q + 3 * par) % BR] = nxt < 9 ? fetch_b(cc, nxt) : fetch_b(cc_b, nxt - 9);
                }
                if (q == 2) fetch_patch();   // patch of chunk g + 3, behind the ring loads of the next positions
            }
#undef L3C_W4W_MFMA
        };
        static_assert(9 % AR == 0 && 18 % BR == 0, "ring slots must repeat every two chunks");

        using F = std::false_type;
        using Tt = std::true_type;
        for (int t = 0; t < n_t; ++t) {
#pragma unroll
            for (int q = 0; q < 9; ++q)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[q][r] = 0.0f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float v;
                asm volatile("v_mov_b32 %0, %1" : "=v"(v) : "v"(bias_init));
                acc[7][r] = v;
            }
            chunk(0, F{}, std::integral_constant<int, 0>{}, Tt{});
            for (int cc = 1; cc + 1 < n_cc; cc += 2) {
                chunk(cc, F{}, std::integral_constant<int, 1>{}, F{});
                chunk(cc + 1, F{}, std::integral_constant<int, 0>{}, F{});
            }
            chunk(n_cc - 1, Tt{}, std::integral_constant<int, 1>{}, F{});
            // (the barrier inside the last chunk precedes its last position's MFMAs, whose A operands are in registers: every wavefront has
            // issued its last read of V before any other one can pass that barrier and reach the writes below)

            // ---- output transform: four rounds, one per tile row.  X[position][tile of the row][channel] over V[0] | V[1].
            [[maybe_unused]] float *X = lds + V_OFF0;
            const int sx0 = (tx_first + t) * OW;
            if constexpr (L3C_W4W_PROBE & 1) {
                f32x16 sum = acc[0];
#pragma unroll
                for (int q = 1; q < 9; ++q) sum = sum + acc[q];
                float v = 0.f;
#pragma unroll
                for (int r = 0; r < 16; ++r) v += sum[r];
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), o_rsrc, (sx0 * p.out_cstride + lane + 64 * wave) * 4, 0, 0);
                __syncthreads();
                if (t + 1 < n_t) {
                    transform_all(lds + RAW_OFF0, lds + V_OFF0);
                    __syncthreads();
                }
                continue;
            }
#if L3C_W4W_EPI == 1
            // PIPELINED (second version): eight HALF rounds k = (tile row i = k >> 1, column pair jh = k & 1) of four tiles -- registers 4 i + 2 jh + jj,
            // jj = 0, 1, i.e. tile columns 4 (lane >> 5) + 2 jh + jj -- through TWO 36 864-byte buffers (V[0], V[1]): the accumulators of half round
            // k + 1 are written while half round k is read and transformed, one barrier per half round.  Readers: thread (tile slot ts = pg,
            // row half = TH, channel = lane) computes output rows 2 TH, 2 TH + 1 of its tile from rows xi = TH .. TH + 4 of the transformed tile
            // (A^T has a zero in column 5 of rows 0..2 and in column 0 of rows 1..3): 30 reads, the same arithmetic as at6.
            {
                auto write_half = [&](auto k_c) __attribute__((always_inline)) {
                    constexpr int k = decltype(k_c)::value, i = k >> 1, jh = k & 1;
                    float *x_dst = lds + ((k & 1) ? V_OFF1 : V_OFF0) + (9 * pg * 4 + 2 * (lane >> 5)) * 64 + h * 32 + (lane & 31);
#pragma unroll
                    for (int q = 0; q < 9; ++q)
#pragma unroll
                        for (int jj = 0; jj < 2; ++jj) x_dst[(q * 4 + jj) * 64] = acc[q][4 * i + 2 * jh + jj];
                };
                auto read_half = [&](auto k_c) __attribute__((always_inline)) {
                    constexpr int k = decltype(k_c)::value, i = k >> 1, jh = k & 1;
                    const float *x_src = lds + ((k & 1) ? V_OFF1 : V_OFF0) + pg * 64 + lane;     // tile slot pg = 2 (writer's lane half) + jj
                    float tr[2][6];
#pragma unroll
                    for (int nu = 0; nu < 6; ++nu) {
                        const float m1 = x_src[(1 * 6 + nu) * 256], m2 = x_src[(2 * 6 + nu) * 256], m3 = x_src[(3 * 6 + nu) * 256], m4 = x_src[(4 * 6 + nu) * 256];
                        const float s1 = m1 + m2, d1 = m1 - m2;
                        if constexpr (TH == 0) {
                            const float m0 = x_src[(0 * 6 + nu) * 256];
                            tr[0][nu] = (m0 + s1) + (m3 + m4);
                            tr[1][nu] = d1 + __builtin_fmaf(-2.0f, m4, 0.5f * m3);
                        } else {
                            const float m5 = x_src[(5 * 6 + nu) * 256];
                            tr[0][nu] = s1 + __builtin_fmaf(4.0f, m4, 0.25f * m3);
                            tr[1][nu] = (d1 + __builtin_fmaf(-8.0f, m4, 0.125f * m3)) + m5;
                        }
                    }
                    const int oy0 = sy0 + 4 * i + 2 * TH, ox0 = sx0 + 4 * (4 * (pg >> 1) + 2 * jh + (pg & 1));
#pragma unroll
                    for (int r = 0; r < 2; ++r) {
                        float y[4];
                        at6(tr[r][0], tr[r][1], tr[r][2], tr[r][3], tr[r][4], tr[r][5], y);
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            float v = y[j];
                            if (p.relu) v = fmaxf(v, 0.0f);
                            const bool ok = lane < p.Cout && oy0 + r < p.H && ox0 + j < p.W;
                            const int off = ok ? (((4 * i + 2 * TH + r) * p.W + ox0 + j) * p.out_cstride + lane) * 4 : OOB;
                            if constexpr (L3C_W4W_PROBE & 2) {
                                if (v == 123.456f) __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), o_rsrc, off, 0, 0);
                            } else
                            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), o_rsrc, off, 0, 0);
                        }
                    }
                };
                write_half(std::integral_constant<int, 0>{});
                __syncthreads();
#define L3C_W4W_HALF(K)                                                               \
    if constexpr (K < 7) write_half(std::integral_constant<int, (K < 7 ? K + 1 : 7)>{}); \
    asm volatile("" ::: "memory");                                                    \
    read_half(std::integral_constant<int, K>{});                                      \
    __syncthreads();
                L3C_W4W_HALF(0) L3C_W4W_HALF(1) L3C_W4W_HALF(2) L3C_W4W_HALF(3) L3C_W4W_HALF(4) L3C_W4W_HALF(5) L3C_W4W_HALF(6) L3C_W4W_HALF(7)
#undef L3C_W4W_HALF
            }
#else
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                // this wavefront's accumulators of tile row i: register 4 i + j = tile (row i, column 4 (lane >> 5) + j), channel 32 h + lane % 32
                float *x_dst = X + (9 * pg * 8 + 4 * (lane >> 5)) * 64 + h * 32 + (lane & 31);
#pragma unroll
                for (int q = 0; q < 9; ++q)
#pragma unroll
                    for (int j = 0; j < 4; ++j) x_dst[(q * 8 + j) * 64] = acc[q][4 * i + j];
                __syncthreads();
                // thread (tile column = wavefront, channel = lane): its 36 positions, the transform, 16 pixels
                const float *x_src = X + wave * 64 + lane;
                float tcol[4][6];
#pragma unroll
                for (int nu = 0; nu < 6; ++nu) {
                    float y[4];
                    at6(x_src[(0 * 6 + nu) * 512], x_src[(1 * 6 + nu) * 512], x_src[(2 * 6 + nu) * 512], x_src[(3 * 6 + nu) * 512],
                        x_src[(4 * 6 + nu) * 512], x_src[(5 * 6 + nu) * 512], y);
#pragma unroll
                    for (int r = 0; r < 4; ++r) tcol[r][nu] = y[r];
                }
                const int oy0 = sy0 + 4 * i, ox0 = sx0 + 4 * wave;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float y[4];
                    at6(tcol[r][0], tcol[r][1], tcol[r][2], tcol[r][3], tcol[r][4], tcol[r][5], y);
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        float v = y[j];
                        if (p.relu) v = fmaxf(v, 0.0f);
                        const bool ok = lane < p.Cout && oy0 + r < p.H && ox0 + j < p.W;
                        const int off = ok ? (((4 * i + r) * p.W + ox0 + j) * p.out_cstride + lane) * 4 : OOB;
                        if constexpr (L3C_W4W_PROBE & 2) {
                            if (v == 123.456f) __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), o_rsrc, off, 0, 0);
                        } else
                        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), o_rsrc, off, 0, 0);
                    }
                }
                __syncthreads();
            }
#endif
            // ---- the next tile's first transform (its patch 0 sits in raw[0], patch 1 in raw[1], patch 2 is in flight)
            if (t + 1 < n_t) {
                transform_all(lds + RAW_OFF0, lds + V_OFF0);
                __syncthreads();
            }
        }
    };
    if (h == 0) body(std::integral_constant<int, 0>{});
    else body(std::integral_constant<int, 1>{});
}

// OIHW 3x3 weights -> U = G g G^T (double, rounded once) as [Cin/8][36 positions][2 halves][64 lanes][4]: lane (n = lane % 32, k half =
// lane / 32), element s holds U[position][co = 32 h + n][ci = 8 cc + 2 s + k half]
__global__ __launch_bounds__(256) void pack_wino4w_kernel(const float *__restrict__ w, int Cout, int Cin, float *__restrict__ packed, int64_t total) {
    const double G[6][3] = {{0.5, 0.0, 0.0},           {1.0 / 6, 1.0 / 6, 1.0 / 6},     {1.0 / 6, -1.0 / 6, 1.0 / 6},
                            {16.0 / 15, 8.0 / 15, 4.0 / 15}, {1.0 / 30, -1.0 / 15, 2.0 / 15}, {0.0, 0.0, 0.5}};
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        int64_t r = i;
        const int s = r % 4;  r /= 4;
        const int lane = r % 64;  r /= 64;
        const int hh = r % 2;  r /= 2;
        const int pos = r % NP;  r /= NP;
        const int cc = (int)r;
        const int co = hh * 32 + (lane & 31);
        const int ci = cc * 8 + 2 * s + (lane >> 5);
        const int xi = pos / 6, nu = pos % 6;
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

extern "C" {

int64_t l3c_conv_wino4w_packed_words(int Cout, int Cin) { return Cout > 64 ? -1 : (int64_t)(Cin / 8) * U_CHUNK_FLOATS; }

int l3c_conv_wino4w_pack_weights(const float *w_oihw, int Cout, int Cin, float *packed, l3c_stream_t stream) {
    L3C_REQUIRE(w_oihw && packed, "null pointer");
    L3C_REQUIRE(Cout > 0 && Cout <= 64 && Cin > 0 && Cin % 16 == 0, "probe: Cout <= 64, Cin a multiple of 16");
    const int64_t total = l3c_conv_wino4w_packed_words(Cout, Cin);
    int64_t g = (total + 255) / 256;
    hipLaunchKernelGGL(pack_wino4w_kernel, dim3((unsigned)(g > 4096 ? 4096 : g)), dim3(256), 0, l3c::as_stream(stream), w_oihw, Cout, Cin, packed, total);
    return l3c::check_launch("pack_wino4w_kernel");
}

int l3c_conv_wino4w(const l3c_conv_desc *d, int tiles_per_block, l3c_stream_t stream) {
    L3C_REQUIRE(d && d->in && d->packed_w && d->bias && d->out, "null pointer in descriptor");
    L3C_REQUIRE(d->KS == 3 && d->stride == 1 && d->dilation == 1, "probe: 3x3, stride 1, dilation 1");
    L3C_REQUIRE(d->B > 0 && d->Hin > 0 && d->Win > 0 && d->Cout > 0 && d->Cout <= 64, "probe: Cout <= 64");
    L3C_REQUIRE(d->Cin > 0 && d->Cin % 16 == 0, "Cin must be a multiple of 16");
    L3C_REQUIRE((d->epilogue & ~L3C_EPI_RELU) == 0, "probe: bias (+ ReLU) only");
    L3C_REQUIRE(d->in_cstride % 4 == 0 && d->in_coff % 4 == 0, "input channel stride/offset must be multiples of 4");
    L3C_REQUIRE(((uintptr_t)d->in | (uintptr_t)d->packed_w) % 16 == 0, "input and packed weights must be 16-byte aligned");
    L3C_REQUIRE((int64_t)d->Hin * d->Win * d->in_cstride * 4 < 0x7ffffff0ll, "one image of the input must stay below 2 GB");
    L3C_REQUIRE(20ll * (d->Win + 64) * d->out_cstride * 4 < 0x7ffffff0ll, "image too wide for 32-bit offsets inside a tile row");
    W4wParams p{};
    p.in = d->in;  p.u = d->packed_w;  p.bias = d->bias;  p.out = d->out;
    p.in_cstride = d->in_cstride;  p.in_coff = d->in_coff;  p.out_cstride = d->out_cstride;  p.out_coff = d->out_coff;
    p.B = d->B;  p.H = d->Hin;  p.W = d->Win;  p.Cin = d->Cin;  p.Cout = d->Cout;
    p.tiles_x = (p.W + OW - 1) / OW;
    p.tiles_y = (p.H + OH - 1) / OH;
    int tpb = tiles_per_block > 0 ? tiles_per_block : 3;
    if (tpb > p.tiles_x) tpb = p.tiles_x;
    p.tpb = tpb;
    p.groups_x = (p.tiles_x + tpb - 1) / tpb;
    p.relu = (d->epilogue & L3C_EPI_RELU) != 0;
    const int64_t total = (int64_t)p.groups_x * p.tiles_y * p.B;
    L3C_REQUIRE(total < (1ll << 31), "grid too large");
    static bool attr_set[64] = {};
    int dev = 0;
    int rc = l3c::check_hip(hipGetDevice(&dev), "hipGetDevice");
    if (rc != L3C_OK) return rc;
    L3C_REQUIRE(dev >= 0 && dev < 64, "device index out of range");
    if (!attr_set[dev]) {
        rc = l3c::check_hip(hipFuncSetAttribute(reinterpret_cast<const void *>(conv_wino4w_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, W4W_LDS_BYTES),
                            "hipFuncSetAttribute");
        if (rc != L3C_OK) return rc;
        attr_set[dev] = true;
    }
    hipLaunchKernelGGL(conv_wino4w_kernel, dim3((unsigned)total), dim3(512), W4W_LDS_BYTES, l3c::as_stream(stream), p);
    return l3c::check_launch("conv_wino4w_kernel");
}
}
