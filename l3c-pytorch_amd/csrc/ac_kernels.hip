// ac_kernels.hip -- range coder on the GPU (replaces the CPU coder of torchac.cpp).
//
//   intervals_from_table_kernel   table + symbols -> packed (c_low, c_high) words          (fully parallel)
//   ac_encode_kernel              one stream per LANE, 64 streams per wavefront: the serial integer state machine of
//                                 csrc/ac_core.h on pre-computed intervals; interval words are read 16 at a time
//                                 (4 x dwordx4, prefetched one group ahead), output words are written per lane
//   ac_decode_kernel              one stream per WAVEFRONT: lanes hold the CDF row of the current symbol (prefetched 4
//                                 symbols ahead); `count` is ranked against the row with v_cmp + s_bcnt1 (ballot/popcount)
//                                 or, for tables not known to be monotone, with the reference's literal binary search
//                                 over v_readlane; the coder state lives in SGPRs (wave-uniform)
//   check_monotone_kernel         flags tables that are not strictly increasing (selects the decode path)
#include "ac_core.h"
#include "l3c_common.h"

namespace {

constexpr int kChunk = 64;  // symbols per interval block (see l3c_interval_words)

__global__ __launch_bounds__(256) void intervals_from_table_kernel(const uint16_t *__restrict__ cdf, int64_t row_stride,
                                                                   int Lp, const int16_t *__restrict__ sym,
                                                                   int64_t n_streams, int64_t n_sym,
                                                                   uint32_t *__restrict__ iv) {
    const int64_t n_chunks = (n_sym + kChunk - 1) / kChunk;
    const int64_t total = n_chunks * n_streams * kChunk;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t j = i % kChunk;
        const int64_t s = (i / kChunk) % n_streams;
        const int64_t t = (i / kChunk / n_streams) * kChunk + j;
        uint32_t w = 0;
        if (t < n_sym) {
            const int x = sym[s * n_sym + t];
            const uint16_t *row = cdf + (row_stride ? (s * n_sym + t) * row_stride : 0);
            const uint32_t c_lo = row[x];
            const uint32_t c_hi = (x == Lp - 2) ? 0x10000u : (uint32_t)row[x + 1];
            w = l3c::pack_interval(c_lo, c_hi);
        }
        iv[i] = w;
    }
}

struct LaneStore {
    uint32_t *words;
    bool active;
    __device__ __forceinline__ void operator()(uint32_t i, uint32_t w) const {
        if (active) words[i] = w;
    }
};

__global__ __launch_bounds__(64) void ac_encode_kernel(const uint32_t *__restrict__ iv, int64_t n_streams, int64_t n_sym,
                                                       uint8_t *__restrict__ out, int64_t out_stride,
                                                       uint32_t *__restrict__ out_nbytes) {
    int64_t s = (int64_t)blockIdx.x * 64 + threadIdx.x;
    const bool active = s < n_streams;
    if (!active) s = n_streams - 1;  // keep the wavefront converged; stores are predicated
    l3c::WordSink<LaneStore> sink(LaneStore{reinterpret_cast<uint32_t *>(out + s * out_stride), active});
    uint32_t low = 0, high = 0xFFFFFFFFu, pending = 0;

    const int64_t n_groups = (n_sym + 15) / 16;  // groups of 16 symbols; 4 groups per 64-symbol block
    auto group_ptr = [&](int64_t g) {
        return reinterpret_cast<const uint4 *>(iv + ((g >> 2) * n_streams + s) * kChunk + (g & 3) * 16);
    };
    uint4 cur[4], nxt[4];
    {
        const uint4 *p = group_ptr(0);
#pragma unroll
        for (int k = 0; k < 4; ++k) cur[k] = p[k];
    }
    const int64_t full_groups = n_sym / 16;
    for (int64_t g = 0; g < full_groups; ++g) {
        if (g + 1 < n_groups) {
            const uint4 *p = group_ptr(g + 1);
#pragma unroll
            for (int k = 0; k < 4; ++k) nxt[k] = p[k];
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const uint32_t w[4] = {cur[k].x, cur[k].y, cur[k].z, cur[k].w};
#pragma unroll
            for (int j = 0; j < 4; ++j)
                l3c::encode_symbol_fast(low, high, pending, l3c::interval_lo(w[j]), l3c::interval_hi(w[j]), sink);
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) cur[k] = nxt[k];
    }
    if (full_groups < n_groups) {   // ragged tail (< 16 symbols), already in `cur`
        const int valid = (int)(n_sym - full_groups * 16);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const uint32_t w[4] = {cur[k].x, cur[k].y, cur[k].z, cur[k].w};
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (k * 4 + j < valid)
                    l3c::encode_symbol_fast(low, high, pending, l3c::interval_lo(w[j]), l3c::interval_hi(w[j]), sink);
        }
    }
    l3c::encode_finish(low, pending, sink);
    const uint32_t nbytes = sink.finish();
    if (active) out_nbytes[s] = nbytes;
}

// ---- decoder -------------------------------------------------------------------------------------------------------

// Wave-uniform word source: a 64-word window of the stream per VGPR (lane l holds word base + l), the next window
// prefetched; fetch(i) is a v_readlane.
struct WaveFetch {
    const uint32_t *words;   // 4-byte aligned start of the stream
    uint32_t nbytes;
    uint32_t base;           // first word index of `cur`
    uint32_t cur, nxt;       // per-lane window registers
    int lane;
    __device__ __forceinline__ uint32_t load_window(uint32_t b) const {
        const uint32_t idx = b + (uint32_t)lane;
        const uint32_t byte0 = idx * 4u;
        uint32_t w = 0;
        if (byte0 < nbytes) {
            w = l3c::bswap32(words[idx]);
            const uint32_t rem = nbytes - byte0;          // valid bytes in this word
            if (rem < 4u) w &= 0xFFFFFFFFu << (8u * (4u - rem));
        }
        return w;
    }
    __device__ __forceinline__ void init() {
        base = 0;
        cur = load_window(0);
        nxt = load_window(64);
    }
    __device__ __forceinline__ uint32_t operator()(uint32_t i) {
        if (i >= base + 64u) {   // wave-uniform
            base += 64u;
            cur = nxt;
            nxt = load_window(base + 64u);
        }
        return (uint32_t)__builtin_amdgcn_readlane((int)cur, (int)(i - base));
    }
};

template <int NJ>
struct Row {
    uint32_t e[NJ];  // lane l holds entries l + 64*j; entries past the top symbol hold 0x10000 (never <= count)
};

template <int NJ>
__device__ __forceinline__ Row<NJ> load_row(const uint16_t *tab, int64_t row_stride, int64_t i, int64_t n_sym, int top,
                                            int lane) {
    Row<NJ> r;
    const uint16_t *row = tab + (row_stride ? i * row_stride : 0);
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const int idx = lane + 64 * j;
        r.e[j] = (i < n_sym && idx <= top) ? (uint32_t)row[idx] : 0x10000u;
    }
    return r;
}

template <int NJ>
__device__ __forceinline__ uint32_t row_fetch(const Row<NJ> &r, uint32_t m) {
    uint32_t v = r.e[0];
#pragma unroll
    for (int j = 1; j < NJ; ++j) v = ((m >> 6) == (uint32_t)j) ? r.e[j] : v;   // m is wave-uniform
    return (uint32_t)__builtin_amdgcn_readlane((int)v, (int)(m & 63u));
}

template <int NJ>
__global__ __launch_bounds__(64) void ac_decode_kernel(const uint16_t *__restrict__ cdf, int64_t row_stride, int Lp,
                                                       const uint8_t *__restrict__ in,
                                                       const int64_t *__restrict__ in_offsets,
                                                       const uint32_t *__restrict__ in_nbytes, int64_t n_sym,
                                                       int monotone, int16_t *__restrict__ sym_out) {
    const int64_t s = blockIdx.x;
    const int lane = threadIdx.x;
    const int top = Lp - 2;
    const uint16_t *tab = cdf + (row_stride ? s * n_sym * row_stride : 0);
    int16_t *dst = sym_out + s * n_sym;

    WaveFetch wf;
    wf.words = reinterpret_cast<const uint32_t *>(in + in_offsets[s]);
    wf.nbytes = in_nbytes[s];
    wf.lane = lane;
    wf.init();
    l3c::WordSource<WaveFetch &> src(wf);
    uint32_t low = 0, high = 0xFFFFFFFFu;
    uint32_t value = src.take(32);

    constexpr int D = 4;  // rows in flight
    Row<NJ> ring[D];
#pragma unroll
    for (int d = 0; d < D; ++d) ring[d] = load_row<NJ>(tab, row_stride, d, n_sym, top, lane);

    int packed_out = 0;  // lane (i & 63) keeps symbol i until the 64-symbol block is stored
    for (int64_t i0 = 0; i0 < n_sym; i0 += D) {
#pragma unroll
        for (int d = 0; d < D; ++d) {
            const int64_t i = i0 + d;
            if (i < n_sym) {   // wave-uniform
                const Row<NJ> row = ring[d];
                ring[d] = load_row<NJ>(tab, row_stride, i + D, n_sym, top, lane);
                const uint32_t count = l3c::decode_count(low, high, value);
                uint32_t x;
                if (monotone) {
                    uint32_t rank = 0;
#pragma unroll
                    for (int j = 0; j < NJ; ++j) rank += (uint32_t)__popcll(__ballot(row.e[j] <= count));
                    x = rank ? rank - 1u : 0u;
                } else {
                    x = l3c::ref_binsearch([&](uint32_t m) { return row_fetch<NJ>(row, m); }, count, (uint32_t)top);
                }
                if ((int)(i & 63) == lane) packed_out = (int)x;
                if ((i & 63) == 63 || i == n_sym - 1) {
                    const int64_t t = (i & ~(int64_t)63) + lane;
                    if (t <= i) dst[t] = (int16_t)packed_out;
                }
                if (i != n_sym - 1) {
                    const uint32_t c_lo = row_fetch<NJ>(row, x);
                    const uint32_t c_hi = (x == (uint32_t)top) ? 0x10000u : row_fetch<NJ>(row, x + 1u);
                    l3c::decode_advance(low, high, value, c_lo, c_hi, src);
                }
            }
        }
    }
}

__global__ __launch_bounds__(256) void check_monotone_kernel(const uint16_t *__restrict__ cdf, int64_t n_rows, int Lp,
                                                             int32_t *__restrict__ flag) {
    // one thread per (row, entry m in [0, Lp-3]): requires cdf[m] < cdf[m+1]
    const int64_t per_row = Lp - 2;
    const int64_t total = n_rows * per_row;
    bool bad = false;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = i / per_row, m = i % per_row;
        const uint16_t *row = cdf + r * Lp;
        bad |= !(row[m] < row[m + 1]);
    }
    if (__any(bad) && (threadIdx.x & 63) == 0) atomicOr(flag, 1);
}

int grid_for(int64_t total, int block, int max_blocks = 256 * 8) {
    int64_t g = (total + block - 1) / block;
    if (g < 1) g = 1;
    return (int)(g > max_blocks ? max_blocks : g);
}

}  // namespace

extern "C" {

int64_t l3c_interval_words(int64_t n_streams, int64_t n_sym) {
    return ((n_sym + kChunk - 1) / kChunk) * n_streams * kChunk;
}

int64_t l3c_ac_max_bytes(int64_t n_sym) {
    // <= 16 bits per symbol plus up to 31 underflow bits carried at any time plus the 2-bit flush, rounded up to a
    // whole number of 32-bit words, plus one spare word for the final partial-word store
    return ((2 * n_sym + 16 + 3) / 4) * 4 + 8;
}

int l3c_ac_intervals_from_table(const uint16_t *cdf, int64_t row_stride, int Lp, const int16_t *sym, int64_t n_streams,
                                int64_t n_sym, uint32_t *intervals, l3c_stream_t stream) {
    L3C_REQUIRE(cdf && sym && intervals, "null pointer");
    L3C_REQUIRE(Lp >= 2 && Lp <= 65536, "Lp out of range");
    L3C_REQUIRE(row_stride == 0 || row_stride == Lp, "row_stride must be 0 or Lp");
    L3C_REQUIRE(n_streams > 0 && n_sym > 0, "empty input");
    const int64_t total = l3c_interval_words(n_streams, n_sym);
    hipLaunchKernelGGL(intervals_from_table_kernel, dim3(grid_for(total, 256)), dim3(256), 0, l3c::as_stream(stream),
                       cdf, row_stride, Lp, sym, n_streams, n_sym, intervals);
    return l3c::check_launch("intervals_from_table_kernel");
}

int l3c_ac_encode(const uint32_t *intervals, int64_t n_streams, int64_t n_sym, uint8_t *out, int64_t out_stride_bytes,
                  uint32_t *out_nbytes, l3c_stream_t stream) {
    L3C_REQUIRE(intervals && out && out_nbytes, "null pointer");
    L3C_REQUIRE(n_streams > 0 && n_sym > 0, "empty input");
    L3C_REQUIRE(out_stride_bytes % 4 == 0 && out_stride_bytes >= l3c_ac_max_bytes(n_sym), "output stride too small");
    L3C_REQUIRE((reinterpret_cast<uintptr_t>(out) & 3) == 0 && (reinterpret_cast<uintptr_t>(intervals) & 15) == 0,
                "misaligned buffer");
    const int blocks = (int)((n_streams + 63) / 64);
    hipLaunchKernelGGL(ac_encode_kernel, dim3(blocks), dim3(64), 0, l3c::as_stream(stream), intervals, n_streams, n_sym,
                       out, out_stride_bytes, out_nbytes);
    return l3c::check_launch("ac_encode_kernel");
}

int l3c_ac_decode(const uint16_t *cdf, int64_t row_stride, int Lp, const uint8_t *in, const int64_t *in_offsets,
                  const uint32_t *in_nbytes, int64_t n_streams, int64_t n_sym, int monotone, int16_t *sym_out,
                  l3c_stream_t stream) {
    L3C_REQUIRE(cdf && in && in_offsets && in_nbytes && sym_out, "null pointer");
    L3C_REQUIRE(Lp >= 2 && Lp <= 257, "Lp out of range (2..257)");
    L3C_REQUIRE(row_stride == 0 || row_stride == Lp, "row_stride must be 0 or Lp");
    L3C_REQUIRE(n_streams > 0 && n_sym > 0, "empty input");
    L3C_REQUIRE((reinterpret_cast<uintptr_t>(in) & 3) == 0, "input must be 4-byte aligned (and every offset a multiple of 4)");
    const dim3 grid((unsigned)n_streams), block(64);
    if (Lp - 1 <= 64)
        hipLaunchKernelGGL(ac_decode_kernel<1>, grid, block, 0, l3c::as_stream(stream), cdf, row_stride, Lp, in,
                           in_offsets, in_nbytes, n_sym, monotone, sym_out);
    else
        hipLaunchKernelGGL(ac_decode_kernel<4>, grid, block, 0, l3c::as_stream(stream), cdf, row_stride, Lp, in,
                           in_offsets, in_nbytes, n_sym, monotone, sym_out);
    return l3c::check_launch("ac_decode_kernel");
}

int l3c_cdf_check_monotone(const uint16_t *cdf, int64_t n_rows, int Lp, int32_t *flag_out, l3c_stream_t stream) {
    L3C_REQUIRE(cdf && flag_out, "null pointer");
    L3C_REQUIRE(Lp >= 2 && n_rows > 0, "bad shape");
    if (Lp == 2) return L3C_OK;  // a single valid entry per row is trivially monotone
    hipLaunchKernelGGL(check_monotone_kernel, dim3(grid_for(n_rows * (Lp - 2), 256)), dim3(256), 0,
                       l3c::as_stream(stream), cdf, n_rows, Lp, flag_out);
    return l3c::check_launch("check_monotone_kernel");
}
}
