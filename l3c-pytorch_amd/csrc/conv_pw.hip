// conv_pw.hip -- pointwise (1x1) convolution Cin -> Cout <= 160 on the fp32 MFMA: the 192 -> Kp layer that ends every
// probability classifier (reference prob_clf.py:71-74, `self.lin`; Kp = 120 for the RGB scale, 150 for the bottleneck scales).
//
// A 1x1 conv has no spatial structure: it is the GEMM  out[pixel][co] = sum_ci in[pixel][ci] * w[co][ci]  over the flattened
// B*H*W pixels.  The kernel is the Winograd kernel's skeleton (conv_wino.hip) with its transforms taken out:
//   * block = NW wavefronts (NW = 4 for Cout <= 128, 5 for Cout <= 160) on a tile of 128 consecutive pixels; wavefront nj owns
//     output channels 32 nj .. 32 nj + 31 for all 128 pixels: 4 accumulator fragments = 64 registers -- two blocks per CU;
//   * input channels in chunks of 32 (one full 128-byte line per pixel): the chunk of the tile is fetched into registers
//     (buffer descriptor over the tile: pixels beyond the tensor read as zero), stored to LDS one chunk later ([pixel][32 + 4
//     pad] floats: conflict-free for the 16-byte A-fragment reads) and multiplied one chunk after that -- a chunk is 64 MFMAs
//     per wavefront, so a load has > 4000 cycles to arrive;
//   * the weights never touch LDS: per chunk four 16-byte buffer loads per lane straight into the B-operand registers (one
//     quad feeds 16 MFMAs), each reloaded for the next chunk right after its last use; ALL loop loads unconditional;
//   * ONE barrier per chunk, placed before its last group of MFMAs (whose A fragments were read before it), so that the LDS
//     buffer of the chunk before can be overwritten right after;
//   * a block walks several tiles over one chunk pipeline (the loads the last chunks issue are the next tile's first); the
//     epilogue needs no LDS: D fragments (lane = output channel) + bias go out as 4-byte stores whose 32 lanes cover a 128-byte
//     run of one pixel's channels -- no barrier between tiles.
// fp32 throughout; differs from the other 1x1 kernels (conv_mfma.hip) by the order of the 192-term sums only.
#include "l3c_common.h"

#include <stdlib.h>

#include <type_traits>

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

struct PwParams {
    const float *in;
    const float *w;      // packed by l3c_conv_pw_pack_weights
    const float *bias;
    float *out;
    int in_cstride, in_coff, out_cstride, out_coff;
    int Cin, Cout;
    long long n_pix;     // B * H * W
    int n_tiles, tpb, total_blocks;
};

constexpr int PW_TM = 128;            // pixels per tile
constexpr int PW_CK = 32;             // input channels per chunk
constexpr int PW_PS = PW_CK + 4;      // LDS stride of a pixel (floats): 16-byte reads of 8 consecutive pixels hit distinct banks
constexpr int PW_BUF = PW_TM * PW_PS; // one chunk of a tile (floats)
constexpr int PW_LDS_BYTES = 2 * PW_BUF * 4;   // 36 864

__device__ __forceinline__ int xcd_remap_pw(int bid, int total) {
    const int q = total >> 3, r = total & 7;
    const int xcd = bid & 7, slot = bid >> 3;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + slot;
}

// X5 (round 5; NW == 4 only): 128 < Cout <= 160 on FOUR wavefronts -- wavefront nj owns channel group nj for all 128 pixels AND the fifth
// group (channels 128 .. 159) for pixel fragment nj: 80 instead of 64 MFMAs per chunk, the same for every wavefront, two blocks per CU
// (176 registers) = two wavefronts on every SIMD.  The five-wavefront form (NW == 5) puts ten wavefronts of two blocks on four SIMDs
// (3, 3, 2, 2): 83 % at best.  Same products in the same order per output: bit-identical.
template <int NW, bool X5 = false>
__global__ __launch_bounds__(NW * 64, X5 ? 2 : 3) void conv_pw_kernel(const PwParams p) {
    static_assert(!X5 || NW == 4, "the split fifth group: four wavefronts");
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int T = NW * 64;
    constexpr int N_PIECES = PW_TM * (PW_CK / 4);            // 16-byte pieces of a chunk of a tile: 1024
    constexpr int NIT = (N_PIECES + T - 1) / T;              // 4 per thread
    const int tid = threadIdx.x, lane = tid & 63;
    const int nj = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int half = lane >> 5, lx = lane & 31;
    const int n_cc = p.Cin / PW_CK;                          // even (host)

    const int blk = xcd_remap_pw(blockIdx.x, p.total_blocks);
    const int tile0 = blk * p.tpb;
    const int n_t = min(p.tpb, p.n_tiles - tile0);

    // this thread's pieces of a chunk: pixel i / 8, channels 4 (i % 8) .. + 3; byte offset inside the tile
    int piece_off[NIT];
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int i = tid + it * T;
        piece_off[it] = i < N_PIECES ? ((i >> 3) * p.in_cstride + (i & 7) * 4) * 4 : 0x7ffffff0;
    }
    // the prefetch pointer: (tile, chunk) of the next fetch, and the descriptor of its tile (pixels beyond the tensor read as zero)
    int pf_tile = 0, pf_cc = 0;
    auto tile_rsrc = [&](int t) {
        const long long px0 = (long long)(tile0 + t) * PW_TM;
        const long long rem = p.n_pix - px0;
        const int rows = rem < PW_TM ? (int)rem : PW_TM;
        return __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(p.in + px0 * p.in_cstride + p.in_coff), 0,
                                                 rows * p.in_cstride * 4, 0x00020000);
    };
    auto in_rsrc = tile_rsrc(0);
    auto pf_advance = [&]() {
        if (++pf_cc == n_cc) {
            if (pf_tile + 1 < n_t) {
                pf_cc = 0;
                in_rsrc = tile_rsrc(++pf_tile);
            } else {
                pf_cc = n_cc - 1;      // past the block's last chunk: the loads stay unconditional, their data is never used
            }
        }
    };
    f32x4 stage[NIT];
    auto fetch_chunk = [&]() {
#pragma unroll
        for (int it = 0; it < NIT; ++it)
            stage[it] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(in_rsrc, piece_off[it], pf_cc * PW_CK * 4, 0));
    };
    auto store_chunk = [&](int par) {
        float *dst = lds + par * PW_BUF;
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int i = tid + it * T;
            if (i < N_PIECES) *reinterpret_cast<f32x4 *>(&dst[(i >> 3) * PW_PS + (i & 7) * 4]) = stage[it];
        }
    };
    // B operands: [chunk][group of 8 k][32-channel group][lane][4]; lane (n = lane % 32, half) holds w[32 nj + n][8 g + 4 half + t]
    const int n_groups_o = (p.Cout + 31) / 32;
    const auto w_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(p.w), 0, n_cc * 4 * n_groups_o * 256 * 4, 0x00020000);
    const int w_lane = (nj * 256 + lane * 4) * 4;
    f32x4 bq[4], bx[4];   // bx: the fifth channel group's weights (X5)
    auto fetch_b = [&](int cc, int g) {
        bq[g] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(w_rsrc, w_lane, ((cc * 4 + g) * n_groups_o) * 1024, 0));
        if constexpr (X5)
            bx[g] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(w_rsrc, (4 * 256 + lane * 4) * 4, ((cc * 4 + g) * n_groups_o) * 1024, 0));
    };

    // prologue: chunks 0 and 1 of the first tile
    fetch_chunk();
    pf_advance();
    store_chunk(0);                       // (waits for the loads)
    fetch_chunk();
    pf_advance();
#pragma unroll
    for (int g = 0; g < 4; ++g) fetch_b(0, g);
    __syncthreads();
    const int a_lane = lx * PW_PS + half * 4;                  // A fragment of pixel (32 m + lx): channels 8 g + 4 half .. + 3
    f32x4 af[2][4], afx[2];   // afx (X5): pixel fragment nj once more, for the fifth channel group (a second LDS read instead of 48 selects per chunk)
#pragma unroll
    for (int m = 0; m < 4; ++m) af[0][m] = *reinterpret_cast<const f32x4 *>(lds + a_lane + m * 32 * PW_PS);
    if constexpr (X5) afx[0] = *reinterpret_cast<const f32x4 *>(lds + a_lane + nj * 32 * PW_PS);
    __builtin_amdgcn_s_waitcnt(0x0f70);   // vmcnt(0): the loop head is also reached from its own back edge (see conv_wino.hip)

    f32x16 acc[4], accx;
    // One chunk: 4 groups of 8 input channels x 4 pixel fragments x 4 k-steps = 64 MFMAs.  Invariants at its start: buffer
    // par holds chunk c; af[0] = the A fragments of its group 0; bq = its weights (in flight); `stage` = chunk c + 1 (in flight);
    // the prefetch pointer is at chunk c + 2.
    auto chunk = [&](const int cc, auto first_c, auto par_c) __attribute__((always_inline)) {
        constexpr bool FIRST = decltype(first_c)::value;
        constexpr int par = decltype(par_c)::value;
        const int cc_b = cc + 1 == n_cc ? 0 : cc + 1;
        const float *a_cur = lds + par * PW_BUF + a_lane;
        const float *a_nxt = lds + (par ^ 1) * PW_BUF + a_lane;
        const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int cb = g & 1, nb = cb ^ 1;
            if (g == 3) {
                // this wave has stored chunk c + 1 and read every fragment of chunk c it needs: the barrier, then the first
                // fragments of chunk c + 1
                __syncthreads();
#pragma unroll
                for (int m = 0; m < 4; ++m) af[nb][m] = *reinterpret_cast<const f32x4 *>(a_nxt + m * 32 * PW_PS);
                if constexpr (X5) afx[nb] = *reinterpret_cast<const f32x4 *>(a_nxt + nj * 32 * PW_PS);
            }
            const f32x4 B = bq[g];
            const f32x4 BX = bx[g];
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                if constexpr (X5) {   // pixel fragment nj x the fifth channel group
                    __builtin_amdgcn_sched_barrier(0);
                    accx = __builtin_amdgcn_mfma_f32_32x32x2f32(afx[cb][t], BX[t], (FIRST && g == 0 && t == 0) ? zero16 : accx, 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                    if (t == 0 && g < 3) afx[nb] = *reinterpret_cast<const f32x4 *>(a_cur + nj * 32 * PW_PS + (g + 1) * 8);
                }
#pragma unroll
                for (int m = 0; m < 4; ++m) {
                    __builtin_amdgcn_sched_barrier(0);
                    acc[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[cb][m][t], B[t], (FIRST && g == 0 && t == 0) ? zero16 : acc[m], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                    // the non-MFMA work of the group, dealt out between the MFMAs
                    if (t == 0 && g < 3) af[nb][m] = *reinterpret_cast<const f32x4 *>(a_cur + m * 32 * PW_PS + (g + 1) * 8);
                    if (g == 0 && t == 1 && m == 0) store_chunk(par ^ 1);     // chunk c + 1 -> the other buffer (free since the last barrier)
                    if (g == 0 && t == 2 && m == 0) fetch_chunk();            // chunk c + 2 into the staging registers just stored
                    if (g == 0 && t == 3 && m == 0) pf_advance();
                    if (t == 3 && m == 3) fetch_b(cc_b, g);                   // this group's weights for chunk c + 1
                }
            }
        }
    };

    const int n_lane = nj * 32 + lx;
    const bool ch_ok = n_lane < p.Cout;
    const float bias_v = ch_ok ? p.bias[n_lane] : 0.0f;
    const int o_lane = ch_ok ? (4 * half * p.out_cstride + n_lane) * 4 : 0x7ffffff0;
    for (int t = 0; t < n_t; ++t) {
        chunk(0, std::true_type{}, std::integral_constant<int, 0>{});
        chunk(1, std::false_type{}, std::integral_constant<int, 1>{});
        for (int cc = 2; cc < n_cc; cc += 2) {
            chunk(cc, std::false_type{}, std::integral_constant<int, 0>{});
            chunk(cc + 1, std::false_type{}, std::integral_constant<int, 1>{});
        }
        // epilogue: D register r of fragment m = pixel 32 m + (r & 3) + 4 half + 8 (r >> 2), channel 32 nj + lx.  The pixel goes
        // into the PER-LANE offset (the range check ignores the scalar offset): pixels beyond the tensor are dropped by it.
        const long long px0 = (long long)(tile0 + t) * PW_TM;
        const long long rem = p.n_pix - px0;
        const int rows = rem < PW_TM ? (int)rem : PW_TM;
        const auto o_rsrc = __builtin_amdgcn_make_buffer_rsrc(p.out + px0 * p.out_cstride + p.out_coff, 0, rows * p.out_cstride * 4, 0x00020000);
        int ocs4 = p.out_cstride * 4;
        asm volatile("" : "+s"(ocs4));   // per tile: 64 precomputed per-lane offsets held across the MFMA loop would cost 64 registers
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int pix = 32 * m + (r & 3) + 8 * (r >> 2);
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, acc[m][r] + bias_v), o_rsrc,
                                                      o_lane + pix * ocs4, 0, 0);
            }
        if constexpr (X5) {   // the fifth group's fragment of this wavefront: pixels 32 nj + ..., channels 128 + lx
            const int nx = 128 + lx;
            const bool okx = nx < p.Cout;
            const float bias_x = okx ? p.bias[nx] : 0.0f;
            const int ox_lane = okx ? (4 * half * p.out_cstride + nx) * 4 : 0x7ffffff0;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int pix = 32 * nj + (r & 3) + 8 * (r >> 2);
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, accx[r] + bias_x), o_rsrc, ox_lane + pix * ocs4, 0, 0);
            }
        }
    }
}

// OIHW 1x1 weights [Cout][Cin] -> [Cin/32][4 groups of 8 k][ceil(Cout/32)][64 lanes][4]: lane (n = lane % 32, half = lane / 32),
// element t holds w[co = 32 j + n][ci = 32 cc + 8 g + 4 half + t] (0 beyond Cout).
__global__ __launch_bounds__(256) void pack_pw_kernel(const float *__restrict__ w, int Cout, int Cin, float *__restrict__ packed,
                                                      int64_t total) {
    const int ngo = (Cout + 31) / 32;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        int64_t r = i;
        const int t = r % 4;  r /= 4;
        const int lane = r % 64;  r /= 64;
        const int j = r % ngo;  r /= ngo;
        const int g = r % 4;  r /= 4;
        const int cc = (int)r;
        const int co = 32 * j + (lane & 31);
        const int ci = 32 * cc + 8 * g + 4 * (lane >> 5) + t;
        packed[i] = co < Cout ? w[(size_t)co * Cin + ci] : 0.0f;
    }
}

}  // namespace

extern "C" {

int64_t l3c_conv_pw_packed_words(int Cout, int Cin) { return (int64_t)(Cin / 32) * 4 * ((Cout + 31) / 32) * 256; }

int l3c_conv_pw_pack_weights(const float *w_oi, int Cout, int Cin, float *packed, l3c_stream_t stream) {
    L3C_REQUIRE(w_oi && packed, "null pointer");
    L3C_REQUIRE(Cout > 0 && Cout <= 160 && Cin > 0 && Cin % 64 == 0, "Cin must be a multiple of 64, Cout <= 160");
    const int64_t total = l3c_conv_pw_packed_words(Cout, Cin);
    int64_t g = (total + 255) / 256;
    hipLaunchKernelGGL(pack_pw_kernel, dim3((unsigned)(g > 4096 ? 4096 : g)), dim3(256), 0, l3c::as_stream(stream), w_oi, Cout, Cin,
                       packed, total);
    return l3c::check_launch("pack_pw_kernel");
}

int l3c_conv_pw(const l3c_conv_desc *d, l3c_stream_t stream) {
    L3C_REQUIRE(d, "null descriptor");
    L3C_REQUIRE(d->in && d->packed_w && d->bias && d->out, "null pointer in descriptor");
    L3C_REQUIRE(d->KS == 1 && d->stride == 1 && d->dilation == 1, "pointwise kernel: 1x1, stride 1");
    L3C_REQUIRE(d->epilogue == 0, "pointwise kernel: bias only (no epilogue flags)");
    L3C_REQUIRE(d->B > 0 && d->Hin > 0 && d->Win > 0, "bad shape");
    L3C_REQUIRE(d->Cin > 0 && d->Cin % 64 == 0, "Cin must be a multiple of 64");
    L3C_REQUIRE(d->Cout > 0 && d->Cout <= 160, "Cout must be <= 160");
    L3C_REQUIRE(d->in_cstride % 4 == 0 && d->in_coff % 4 == 0 && ((uintptr_t)d->in | (uintptr_t)d->packed_w) % 16 == 0,
                "input channel stride/offset must be multiples of 4, input and weights 16-byte aligned");
    L3C_REQUIRE(d->in_coff + d->Cin <= d->in_cstride && d->out_coff + d->Cout <= d->out_cstride, "channel slice out of range");
    L3C_REQUIRE((int64_t)d->in_cstride * 4 * 128 < (1ll << 30) && (int64_t)d->out_cstride * 4 * 128 < (1ll << 30), "channel stride too large");
    PwParams p{};
    p.in = d->in;  p.w = d->packed_w;  p.bias = d->bias;  p.out = d->out;
    p.in_cstride = d->in_cstride;  p.in_coff = d->in_coff;  p.out_cstride = d->out_cstride;  p.out_coff = d->out_coff;
    p.Cin = d->Cin;  p.Cout = d->Cout;
    p.n_pix = (long long)d->B * d->Hin * d->Win;
    const int64_t tiles = (p.n_pix + PW_TM - 1) / PW_TM;
    L3C_REQUIRE(tiles < (1ll << 31), "grid too large");
    p.n_tiles = (int)tiles;
    int tpb = 4;                                   // tiles per block, as many as leave every block slot several blocks
    while (tpb > 1 && (tiles + tpb - 1) / tpb < 8 * 512) --tpb;
    p.tpb = tpb;
    p.total_blocks = (int)((tiles + tpb - 1) / tpb);
    if (d->Cout <= 128) {
        hipLaunchKernelGGL(conv_pw_kernel<4>, dim3((unsigned)p.total_blocks), dim3(256), PW_LDS_BYTES, l3c::as_stream(stream), p);
    } else {
#ifdef L3C_PW_FIVE_WAVES
        hipLaunchKernelGGL(conv_pw_kernel<5>, dim3((unsigned)p.total_blocks), dim3(320), PW_LDS_BYTES, l3c::as_stream(stream), p);
#else
        hipLaunchKernelGGL((conv_pw_kernel<4, true>), dim3((unsigned)p.total_blocks), dim3(256), PW_LDS_BYTES, l3c::as_stream(stream), p);
#endif
    }
    return l3c::check_launch("conv_pw_kernel");
}
}
