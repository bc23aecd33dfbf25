// ac_kernels.hip -- range coder on the GPU (replaces the CPU coder of torchac.cpp).
//
//   intervals_from_table_kernel   table + symbols -> the two role words of every symbol     (fully parallel)
//   ac_state_kernel               encoder phase 1, one stream per LANE PAIR (32 streams per wavefront): only the serial
//                                 interval recurrence in the role-symmetric form of csrc/ac_core.h, branch-free; each
//                                 interval word is replaced in place by its role's bound after the update (low' / ~high')
//   ac_pack_kernel                encoder phase 2, one stream per WAVEFRONT, 64 symbols per step: pending runs by a segmented
//                                 wave scan, bit offsets by a prefix sum, bits merged with LDS atomics, words written coalesced
//   ac_decode_lean_kernel         the fast pass, one stream per WAVEFRONT: the table streams through an LDS ring by LDS-DMA,
//                                 lanes hold the CDF row of the current symbol; the symbol is ranked against the scaled row
//                                 with v_cmp + s_bcnt1 (division-free); the coder state lives in SGPRs (wave-uniform); full
//                                 ring blocks of the 256-symbol alphabet run a hand-written, branch-free loop
//   ac_decode_ring_kernel         the generic decoder on the same ring (streams the fast pass gave up on, tables not known
//                                 to be monotone): the reference's literal binary search over v_readlane
//   ac_decode_const_row_kernel    the same for one row shared by all symbols (the uniform prior of the coarsest scale)
//   check_monotone_kernel         flags tables that are not strictly increasing (selects the decode path)
#include <vector>

#include "ac_core.h"
#include "dmll_core.h"
#include "l3c_common.h"

namespace {

constexpr int kChunk = 64;  // symbols per interval block (see l3c_interval_words)

// Interval words (include/l3c_hip.h): per 64-symbol block and stream TWO runs of 64 words, one per role of the lane pair
// that codes the stream (csrc/ac_core.h: role_word).
__device__ __forceinline__ int64_t iv_index(int64_t chunk, int64_t n_streams, int64_t s, int role, int j) {
    return ((chunk * n_streams + s) * 2 + role) * kChunk + j;
}

__global__ __launch_bounds__(256) void intervals_from_table_kernel(const uint16_t *__restrict__ cdf, int64_t row_stride,
                                                                   int Lp, const int16_t *__restrict__ sym,
                                                                   int64_t n_streams, int64_t n_sym,
                                                                   uint32_t *__restrict__ iv) {
    const int64_t n_chunks = (n_sym + kChunk - 1) / kChunk;
    const int64_t total = n_chunks * n_streams * kChunk;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int j = (int)(i % kChunk);
        const int64_t s = (i / kChunk) % n_streams;
        const int64_t chunk = i / kChunk / n_streams;
        const int64_t t = chunk * kChunk + j;
        uint32_t w0 = 0, w1 = 0;
        if (t < n_sym) {
            const int x = sym[s * n_sym + t];
            const uint16_t *row = cdf + (row_stride ? (s * n_sym + t) * row_stride : 0);
            const uint32_t c_lo = row[x];
            const uint32_t c_hi = (x == Lp - 2) ? 0x10000u : (uint32_t)row[x + 1];
            w0 = l3c::role_word(c_lo, c_hi, 0);
            w1 = l3c::role_word(c_lo, c_hi, 1);
        }
        iv[iv_index(chunk, n_streams, s, 0, j)] = w0;
        iv[iv_index(chunk, n_streams, s, 1, j)] = w1;
    }
}

// ---- encoder, phase 1: interval recurrence, one stream per lane PAIR ----------------------------------------------------
// Lanes 2i / 2i+1 are the two roles of stream i (csrc/ac_core.h: role 0 holds low, role 1 ~high; the partner's value comes
// through a DPP quad permute, which the compiler folds into the xor / and / add that consumes it).  A lane reads its role's
// 64 words of a block as 16 x dwordx4 (the next block already in flight) and overwrites each with u' -- same address, so no
// extra memory and no read/write hazard (a lane only ever touches its own 64-word runs, in order).
// The grouped launches read their buffer pointers from a descriptor in memory, so the compiler cannot prove them global and
// emits FLAT accesses -- whose completion it can only await with vmcnt(0): the block-ahead prefetch below (and phase 2's ring
// of records) then waited for the loads it had just issued, a full HBM round trip per 64 symbols (18.5 -> 11 ms per
// 393 216-symbol stream in phase 1).  Typed global pointers give global_load / global_store and exact vmcnt counts.
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(1))) u32x4 gu32x4;
typedef __attribute__((address_space(1))) uint32_t gu32;

__device__ __forceinline__ uint32_t pair_partner(uint32_t v) {
    return (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0xB1 /* quad_perm [1,0,3,2] */, 0xF, 0xF, true);
}

// One symbol of a lane pair (csrc/ac_core.h: role_term / role_shift / role_renorm, restated for the DEPTH of the dependent
// chain: a lone wavefront waits ~7 cycles for the result of the previous instruction, so what counts is the longest path from
// `range` to the next `range`, 9 instructions here):
//     range -> rl * c -> + c + round -> u' = (u + rh * c) + (q >> 16)          [3; u + rh * c beside them]
//     u' -> u' << 1 -> & partner's -> ~((u' ^ o) | .) -> ffbh = n + m          [4; o = partner's u' and S = u' + o beside them]
//     n + m -> S << (n + m) -> ~ = range                                        [2: range = ((high' - low' + 1) << (n + m)) - 1
//                                                                                   and high' - low' + 1 = -S]
// the renormalised bound u = (u' << (n + m)) & 0x7FFFFFFF is only needed by the NEXT symbol's last addition.
__device__ __forceinline__ uint32_t pair_step(uint32_t &u, uint32_t &range, uint32_t c, uint32_t round) {
    const uint32_t q = l3c::mul24(range & 0xFFFFu, c) + c + round;
    const uint32_t base = u + l3c::mul24(range >> 16, c);
    uint32_t u1;                                          // base + (q >> 16) == u + role_term(range, c, round), the shift as an operand select
    asm("v_add_u32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1" : "=v"(u1) : "v"(base), "v"(q));
    const uint32_t o1 = pair_partner(u1);
    const uint32_t sum = u1 + o1;                         // low' + ~high' = -(high' - low' + 1)
    const uint32_t us = u1 << 1;
    const uint32_t zz = us & pair_partner(us);            // (low' & ~high') << 1
    const uint32_t h = ~((u1 ^ o1) | zz);
    // == role_shift(u1, o1).  h != 0 for low' < high'; for an interval that violates the encoder's precondition (c_high <= c_low) h may be
    // 0, for which __builtin_clz is undefined to the COMPILER: v_ffbh_u32 returns -1 there and the shifts use its low 5 bits (the stream
    // is then garbage, as the reference's is, but nothing the optimiser may assume is violated; phase 2 bounds its stores)
    int t;
    asm("v_ffbh_u32 %0, %1" : "=v"(t) : "v"(h));
    t &= 31;
    range = ~(sum << t);                                  // == role_range of the renormalised pair
    u = (u1 << t) & 0x7FFFFFFFu;                          // == role_renorm(u1, t)
    asm volatile("" : "+v"(range));   // keep `range` a value of its own: its two halves then feed the 24-bit multiplies as SDWA selects
    return u1;
}

__device__ __forceinline__ void ac_state_body(uint32_t *__restrict__ iv, int64_t n_streams, int64_t n_sym,
                                              uint32_t *__restrict__ final_low, int64_t block) {
    int64_t s = block * 32 + (threadIdx.x >> 1);
    const int role = threadIdx.x & 1;
    const bool active = s < n_streams;
    if (!active) s = n_streams - 1;   // keep the wavefront converged; duplicates rewrite identical values
    uint32_t u = 0, range = 0xFFFFFFFFu;   // low = 0, high = 0xFFFFFFFF
    const uint32_t round = l3c::role_round(role);
    __builtin_amdgcn_s_setprio(3);    // a few long-lived latency-bound waves next to MFMA-heavy kernels: issue first

    // One 64-symbol block (16 x dwordx4 per lane) is processed while the next one is already in flight: ~2 us of serial
    // work per block hides the HBM latency even when the conv kernels of the next batch saturate the memory system.
    const int64_t n_chunks = (n_sym + kChunk - 1) / kChunk;
    auto chunk_ptr = [&](int64_t c) { return (gu32x4 *)(iv + iv_index(c, n_streams, s, role, 0)); };
    u32x4 cur[16], nxt[16];
    {
        const gu32x4 *p = chunk_ptr(0);
#pragma unroll
        for (int k = 0; k < 16; ++k) cur[k] = p[k];
    }
    for (int64_t c = 0; c < n_chunks; ++c) {
        {   // UNCONDITIONAL (the last block is fetched again and dropped): with a skipped prefetch as a second path into the
            // code below the compiler has to await `cur` with vmcnt(0) -- i.e. the loads just issued
            const gu32x4 *p = chunk_ptr(c + 1 < n_chunks ? c + 1 : c);
#pragma unroll
            for (int k = 0; k < 16; ++k) nxt[k] = p[k];
        }
        gu32x4 *dst = chunk_ptr(c);
        const int64_t left = n_sym - c * kChunk;
        if (left >= kChunk) {
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                uint32_t w[4] = {cur[k].x, cur[k].y, cur[k].z, cur[k].w};
#pragma unroll
                for (int j = 0; j < 4; ++j) w[j] = pair_step(u, range, w[j], round);
                if (active) dst[k] = u32x4{w[0], w[1], w[2], w[3]};
            }
        } else {   // ragged last block
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                uint32_t w[4] = {cur[k].x, cur[k].y, cur[k].z, cur[k].w};
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (k * 4 + j < (int)left) w[j] = pair_step(u, range, w[j], round);   // wave-uniform test
                if (active) dst[k] = u32x4{w[0], w[1], w[2], w[3]};
            }
        }
#pragma unroll
        for (int k = 0; k < 16; ++k) cur[k] = nxt[k];
    }
    if (active && role == 0) final_low[s] = u;
}

// One group of equally long streams for the grouped launches (mirrors l3c_ac_group of include/l3c_hip.h).
struct AcGroup {
    uint32_t *intervals;
    uint8_t *out;
    uint32_t *out_nbytes;
    uint32_t *final_low;
    int64_t n_streams, n_sym, out_stride;
};

struct AcGroupPack {
    static constexpr int N = 64;
    AcGroup g[N];
};

__global__ __launch_bounds__(64) void ac_groups_upload_kernel(AcGroupPack pack, AcGroup *__restrict__ dst, int n) {
    if ((int)threadIdx.x < n) dst[threadIdx.x] = pack.g[threadIdx.x];
}

__global__ __launch_bounds__(64) void ac_state_kernel(uint32_t *__restrict__ iv, int64_t n_streams, int64_t n_sym,
                                                      uint32_t *__restrict__ final_low) {
    ac_state_body(iv, n_streams, n_sym, final_low, blockIdx.x);
}

// Grouped launch: block -> (group, 32-stream block inside the group) by walking the (short) descriptor array.
__global__ __launch_bounds__(64) void ac_state_groups_kernel(const AcGroup *__restrict__ groups, int n_groups) {
    int64_t blk = blockIdx.x;
    int g = 0;
    for (; g < n_groups; ++g) {
        const int64_t nb = (groups[g].n_streams + 31) / 32;
        if (blk < nb) break;
        blk -= nb;
    }
    if (g >= n_groups) return;
    const AcGroup gr = groups[g];
    ac_state_body(gr.intervals, gr.n_streams, gr.n_sym, gr.final_low, blk);
}

// ---- encoder, phase 2: records -> bits, one stream per wavefront, 64 symbols per step ------------------------------------
struct GlobalWordStore {
    uint32_t *words;
    bool on;
    uint32_t cap;   // words of the stream's output slot: nothing is ever stored beyond it (see ac_pack_body)
    __device__ __forceinline__ void operator()(uint32_t i, uint32_t w) const {
        if (on && i < cap) words[i] = w;
    }
};

// Wave scans by DPP (round 4; before: __shfl_up = ds_bpermute, ~100 cycles of LDS-crossbar latency per level, 19 of them on
// the serial path of every 64-symbol step).  dpp0<CTRL, ROWS>(v): the value of the lane CTRL names, 0 where there is none
// (row_shr beyond the start of a 16-lane row) or for rows outside ROWS.  Hillis-Steele inside the rows (row_shr 1, 2, 4, 8),
// then lane 15 of rows 0 / 2 into rows 1 / 3 (row_bcast:15) and lane 31 into rows 2, 3 (row_bcast:31).
template <int CTRL, int ROWS = 0xF>
__device__ __forceinline__ uint32_t dpp0(uint32_t v) {
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, ROWS, 0xF, false);
}
#define L3C_WAVE_SCAN_LEVELS(STEP) \
    STEP(0x111, 0xF) STEP(0x112, 0xF) STEP(0x114, 0xF) STEP(0x118, 0xF) STEP(0x142, 0xA) STEP(0x143, 0xC)

__device__ __forceinline__ uint32_t wave_inclusive_sum(uint32_t v) {
#define L3C_STEP(CTRL, ROWS) v += dpp0<CTRL, ROWS>(v);
    L3C_WAVE_SCAN_LEVELS(L3C_STEP)
#undef L3C_STEP
    return v;
}

// segmented inclusive sum: S_j = flag_j ? val_j : S_{j-1} + val_j; on return `flag` says whether any lane <= j had its flag set
__device__ __forceinline__ uint32_t wave_segmented_sum(uint32_t val, uint32_t &flag) {
#define L3C_STEP(CTRL, ROWS)                          \
    {                                                 \
        const uint32_t v_up = dpp0<CTRL, ROWS>(val);  \
        const uint32_t f_up = dpp0<CTRL, ROWS>(flag); \
        val += flag ? 0u : v_up;                      \
        flag |= f_up;                                 \
    }
    L3C_WAVE_SCAN_LEVELS(L3C_STEP)
#undef L3C_STEP
    return val;
}

__device__ __forceinline__ uint32_t wave_prev_lane(uint32_t v) { return dpp0<0x138>(v); }   // wave_shr:1; lane 0 reads 0

// Shared memory of a pack block: the bit window of one step (<= 31 carried bits + kPackWaves * 64 symbols of <= 32 bits) and what the
// wavefronts tell each other.
constexpr int kPackWaves = 4;                          // wavefronts per stream: a step takes 4 x 64 symbols (round 4; one before)
constexpr int kPackThreads = kPackWaves * 64;
struct PackShared {
    uint32_t buf[kPackWaves * 64 + 16];                // the window, in words
    uint32_t rec[kPackThreads];                        // the step's records (for the serial path of long pending runs)
    uint32_t flag[kPackWaves], val[kPackWaves];        // per wavefront: does it hold an emitter; its pending run with no carry-in
    uint32_t bits[kPackWaves];                         // per wavefront: bits it emits
    uint32_t rare;                                     // some symbol of the step emits more than 32 bits
    uint32_t state[4];                                 // after the serial path: pending, bit_off (2 words), carry_word
};

__device__ __forceinline__ uint32_t uni(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }

// One stream per BLOCK of kPackWaves wavefronts; a step codes kPackWaves consecutive 64-symbol blocks, wavefront w the w-th of
// them.  Inside a wavefront as before (segmented scan of the pending runs, prefix sum of the bits); between the wavefronts the
// pending run and the bit offset are carried by a serial composition of four summaries: a wavefront with an emitter ends with
// its own run (val), one without passes carry-in + its sum on.
__device__ __forceinline__ void ac_pack_body(PackShared &sh, const uint32_t *__restrict__ rec, int64_t n_streams,
                                             int64_t n_sym, const uint32_t *__restrict__ final_low,
                                             uint8_t *__restrict__ out, int64_t out_stride,
                                             uint32_t *__restrict__ out_nbytes, int64_t s) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    uint32_t *words = reinterpret_cast<uint32_t *>(out + s * out_stride);
    // The slot holds l3c_ac_max_bytes(n_sym) bytes: enough for every stream whose intervals satisfy c_high > c_low (<= 16 bits a symbol).
    // Intervals that do not (the public entry points cannot validate them on the host) may ask for up to 31 bits a symbol: the stores stop
    // at the slot's end and the stream's length is reported as L3C_AC_OVERRUN instead of writing into the next stream's slot.
    const uint32_t cap_words = (uint32_t)(out_stride >> 2);
    gu32 *gwords = (gu32 *)words;              // see the note at gu32: exact vmcnt counts for the ring of records below
    const gu32 *grec = (const gu32 *)rec;
    uint32_t pending = 0;          // block-uniform
    uint64_t bit_off = 0;          // bits emitted so far (block-uniform)
    uint32_t carry_word = 0;       // the incomplete output word, MSB aligned (block-uniform)

    const int64_t n_chunks = (n_sym + kChunk - 1) / kChunk;
    const int64_t n_steps = (n_chunks + kPackWaves - 1) / kPackWaves;
    constexpr int PF = 4;          // records in flight: 4 steps
    uint32_t ring[PF], ring_hi[PF];   // what phase 1 left of a symbol: low' and ~high' (csrc/ac_core.h: record_from_pair)
    auto chunk_of = [&](int64_t step) {   // this wavefront's block of the step, clamped to the stream (a clamped block is not used)
        const int64_t c = step * kPackWaves + wave;
        return c < n_chunks ? c : n_chunks - 1;
    };
#pragma unroll
    for (int d = 0; d < PF; ++d) {
        ring[d] = grec[iv_index(chunk_of(d), n_streams, s, 0, lane)];
        ring_hi[d] = grec[iv_index(chunk_of(d), n_streams, s, 1, lane)];
    }
    for (int64_t t0 = 0; t0 < n_steps; t0 += PF)
#pragma unroll
    for (int d = 0; d < PF; ++d) {
        const int64_t step = t0 + d;
        if (step >= n_steps) break;   // block-uniform
        const int64_t c = step * kPackWaves + wave;
        const uint32_t r = (c < n_chunks && c * kChunk + lane < n_sym) ? l3c::record_from_pair(ring[d], ring_hi[d]) : 0u;   // a zero record emits nothing
        {   // unconditional, clamped: see ac_state_body
            ring[d] = grec[iv_index(chunk_of(step + PF), n_streams, s, 0, lane)];
            ring_hi[d] = grec[iv_index(chunk_of(step + PF), n_streams, s, 1, lane)];
        }
        const uint32_t n = l3c::record_n(r), m = l3c::record_m(r), top = l3c::record_top(r);
        const bool emits = n != 0;
        // segmented inclusive scan: S_j = emits_j ? m_j : S_{j-1} + m_j  (a run of pending bits restarts at every emitter)
        uint32_t flag = emits ? 1u : 0u;
        uint32_t val = wave_segmented_sum(m, flag);
        sh.rec[tid] = r;
        if (lane == 63) {
            sh.flag[wave] = flag;
            sh.val[wave] = val;
        }
        if (tid == 0) sh.rare = 0u;
        sh.buf[tid] = 0u;                              // (the window; word 0 gets the carried bits below)
        if (tid < 16) sh.buf[kPackThreads + tid] = 0u;
        __syncthreads();
        // the pending run this wavefront starts with, and the one the step ends with
        uint32_t pend_in = pending, pend_out = pending;
#pragma unroll
        for (int u = 0; u < kPackWaves; ++u) {
            const uint32_t f = uni(sh.flag[u]), v_u = uni(sh.val[u]);
            pend_out = f ? v_u : pend_out + v_u;
            if (u + 1 == wave) pend_in = pend_out;
        }
        if (!flag) val += pend_in;                     // no emitter at or before this lane: the carried run continues
        // pending BEFORE symbol j = S_{j-1} (exclusive), S_{-1} = carried pending
        uint32_t p_before = wave_prev_lane(val);
        if (lane == 0) p_before = pend_in;
        const uint32_t e = emits ? n + p_before : 0u;  // bits this symbol emits
        const uint32_t incl = wave_inclusive_sum(e);
        if (lane == 63) sh.bits[wave] = incl;
        if (__any(e > 32u) && lane == 0) sh.rare = 1u;
        if (tid == 0) sh.buf[0] = carry_word;
        __syncthreads();
        if (__builtin_expect(uni(sh.rare) != 0u, 0)) {
            // a pending run of >= 15 bits: emit this step serially through the literal path (the first wavefront, from the records)
            if (wave == 0) {
                const uint32_t nb = (uint32_t)(bit_off & 31u);
                l3c::WordSink<GlobalWordStore> sink(GlobalWordStore{words, lane == 0, cap_words});
                sink.nwords = (uint32_t)(bit_off >> 5);
                sink.nb = (int)nb;
                sink.acc = nb ? (uint64_t)(carry_word >> (32u - nb)) : 0u;
                uint32_t pend = pending;
                for (int j = 0; j < kPackThreads; ++j) {
                    const uint32_t rj = uni(sh.rec[j]);
                    const uint32_t nj = l3c::record_n(rj);
                    const uint32_t lo_j = nj ? l3c::record_top(rj) << ((32u - nj) & 31u) : 0u;
                    l3c::emit_record(lo_j, nj | (l3c::record_m(rj) << 8), pend, sink);
                }
                if (lane == 0) {
                    const uint64_t off = (uint64_t)sink.nwords * 32u + (uint32_t)sink.nb;
                    sh.state[0] = pend;
                    sh.state[1] = (uint32_t)off;
                    sh.state[2] = (uint32_t)(off >> 32);
                    sh.state[3] = sink.nb ? (uint32_t)(sink.acc << (32 - sink.nb)) : 0u;
                }
            }
            __syncthreads();
            pending = uni(sh.state[0]);
            bit_off = (uint64_t)uni(sh.state[1]) | ((uint64_t)uni(sh.state[2]) << 32);
            carry_word = uni(sh.state[3]);
            __syncthreads();
            continue;
        }

        // bit position of every symbol inside this step's window: the carried bits, the wavefronts before, the lanes before
        uint32_t base = (uint32_t)(bit_off & 31u), total = 0;
#pragma unroll
        for (int u = 0; u < kPackWaves; ++u) {
            const uint32_t b_u = uni(sh.bits[u]);
            if (u < wave) base += b_u;
            total += b_u;
        }
        const uint32_t pos = base + incl - e;
        // value: top n bits with the pending complements inserted after the first = top + (ones(p) << (n-1))
        const uint32_t v = emits ? top + (l3c::ones((int)p_before) << ((n - 1u) & 31u)) : 0u;
        if (e) {
            const uint32_t wi = pos >> 5;
            const int shf = 32 - (int)(pos & 31u) - (int)e;
            if (shf >= 0) {
                atomicOr(&sh.buf[wi], v << shf);
            } else {
                atomicOr(&sh.buf[wi], v >> (-shf));
                atomicOr(&sh.buf[wi + 1], v << (32 + shf));
            }
        }
        __syncthreads();
        const uint32_t window_bits = (uint32_t)(bit_off & 31u) + total;
        const uint32_t full = window_bits >> 5;        // complete words in the window (<= kPackThreads + 1)
        const uint32_t first_word = (uint32_t)(bit_off >> 5);
        if ((uint32_t)tid < full && first_word + tid < cap_words) gwords[first_word + tid] = l3c::bswap32(sh.buf[tid]);
        if ((uint32_t)tid + kPackThreads < full && first_word + kPackThreads + tid < cap_words)
            gwords[first_word + kPackThreads + tid] = l3c::bswap32(sh.buf[kPackThreads + tid]);
        carry_word = (window_bits & 31u) ? uni(sh.buf[full]) : 0u;
        __syncthreads();
        bit_off += total;
        pending = pend_out;
    }
    // flush (torchac.cpp:209-219): pending + 1 complements after the quadrant bit, zero padding to a byte
    if (wave == 0) {
        const uint32_t nb = (uint32_t)(bit_off & 31u);
        l3c::WordSink<GlobalWordStore> sink(GlobalWordStore{words, lane == 0, cap_words});
        sink.nwords = (uint32_t)(bit_off >> 5);
        sink.nb = (int)nb;
        sink.acc = nb ? (uint64_t)(carry_word >> (32u - nb)) : 0u;
        uint32_t pend = pending;
        l3c::encode_finish(final_low[s], pend, sink);
        const uint32_t nbytes = sink.finish();
        if (lane == 0) out_nbytes[s] = (uint64_t)nbytes <= (uint64_t)out_stride ? nbytes : 0xFFFFFFFFu /* L3C_AC_OVERRUN */;
    }
}

__global__ __launch_bounds__(kPackThreads) void ac_pack_kernel(const uint32_t *__restrict__ rec, int64_t n_streams, int64_t n_sym,
                                                                const uint32_t *__restrict__ final_low, uint8_t *__restrict__ out,
                                                                int64_t out_stride, uint32_t *__restrict__ out_nbytes) {
    __shared__ PackShared sh;
    ac_pack_body(sh, rec, n_streams, n_sym, final_low, out, out_stride, out_nbytes, blockIdx.x);
}

__global__ __launch_bounds__(kPackThreads) void ac_pack_groups_kernel(const AcGroup *__restrict__ groups, int n_groups) {
    __shared__ PackShared sh;
    int64_t blk = blockIdx.x;
    int g = 0;
    for (; g < n_groups; ++g) {
        if (blk < groups[g].n_streams) break;
        blk -= groups[g].n_streams;
    }
    if (g >= n_groups) return;
    const AcGroup gr = groups[g];
    ac_pack_body(sh, gr.intervals, gr.n_streams, gr.n_sym, gr.final_low, gr.out, gr.out_stride, gr.out_nbytes, blk);
}

// ---- decoder -------------------------------------------------------------------------------------------------------
// One stream per WAVEFRONT.  The coder state (low, high, value, bit reader) is wave-uniform and lives in SGPRs; the 64 lanes
// hold the CDF row of the current symbol (lane l: entries l + 64 j), so the symbol search is one compare + ballot + popcount
// per 64 entries instead of the reference's binary search (torchac.cpp:286-307), and the two interval bounds are v_readlanes.
//
// Row registers are NAMED scalars (Regs<1> / Regs<4>), not arrays: with arrays the compiler turned the wave-uniform select
// over j into an indexed load from an LDS-promoted alloca -- two LDS round trips on the serial chain of every symbol.

template <int NJ>
struct Regs;
template <>
struct Regs<1> {
    uint32_t a;
};
template <>
struct Regs<4> {
    uint32_t a, b, c, d;
};

__device__ __forceinline__ uint32_t lane_read(uint32_t v, uint32_t l) {
    return (uint32_t)__builtin_amdgcn_readlane((int)v, (int)l);
}
__device__ __forceinline__ uint32_t count_le(uint32_t v, uint32_t bound) {
    return (uint32_t)__popcll(__builtin_amdgcn_ballot_w64(v <= bound));
}
__device__ __forceinline__ uint32_t scale16(uint32_t range, uint32_t c) {   // (span * c) >> 16, span = range + 1
    return (uint32_t)(((uint64_t)range * c + c) >> 16);
}

// entries past the top symbol become 0x10000: never <= a 16-bit count, and scaled they exceed every value - low
__device__ __forceinline__ void regs_mask(Regs<1> &r, int lane, int top) { r.a = lane <= top ? r.a : 0x10000u; }
__device__ __forceinline__ void regs_mask(Regs<4> &r, int lane, int top) {
    r.a = lane <= top ? r.a : 0x10000u;
    r.b = lane + 64 <= top ? r.b : 0x10000u;
    r.c = lane + 128 <= top ? r.c : 0x10000u;
    r.d = lane + 192 <= top ? r.d : 0x10000u;
}
__device__ __forceinline__ Regs<1> regs_scale(const Regs<1> &r, uint32_t range) { return Regs<1>{scale16(range, r.a)}; }
__device__ __forceinline__ Regs<4> regs_scale(const Regs<4> &r, uint32_t range) {
    return Regs<4>{scale16(range, r.a), scale16(range, r.b), scale16(range, r.c), scale16(range, r.d)};
}
__device__ __forceinline__ uint32_t regs_rank(const Regs<1> &r, uint32_t bound) { return count_le(r.a, bound); }
__device__ __forceinline__ uint32_t regs_rank(const Regs<4> &r, uint32_t bound) {
    return (count_le(r.a, bound) + count_le(r.b, bound)) + (count_le(r.c, bound) + count_le(r.d, bound));
}
// Lanes holding real table entries (index <= top), as wave-uniform 64-bit masks: the fast decoder ranks UNMASKED rows and
// ands the ballots with these (a masked entry 0x10000 would scale to 2^32 = 0 when the interval is the full 32-bit range,
// which recurs whenever a symbol's interval is an aligned power of two, e.g. a width-1 cdf step coded from the full range).
template <int NJ>
struct ValidLanes;
template <>
struct ValidLanes<1> {
    uint64_t a;
};
template <>
struct ValidLanes<4> {
    uint64_t a, b, c, d;
};
__device__ __forceinline__ ValidLanes<1> valid_lanes1(int lane, int top) {
    return ValidLanes<1>{__builtin_amdgcn_ballot_w64(lane <= top)};
}
__device__ __forceinline__ ValidLanes<4> valid_lanes4(int lane, int top) {
    return ValidLanes<4>{__builtin_amdgcn_ballot_w64(lane <= top), __builtin_amdgcn_ballot_w64(lane + 64 <= top),
                         __builtin_amdgcn_ballot_w64(lane + 128 <= top), __builtin_amdgcn_ballot_w64(lane + 192 <= top)};
}
__device__ __forceinline__ uint32_t count_le_valid(uint32_t v, uint32_t bound, uint64_t valid) {
    return (uint32_t)__popcll(__builtin_amdgcn_ballot_w64(v <= bound) & valid);
}
__device__ __forceinline__ uint32_t regs_fetch(const Regs<1> &r, uint32_t m) { return lane_read(r.a, m & 63u); }
__device__ __forceinline__ uint32_t regs_fetch(const Regs<4> &r, uint32_t m) {   // m is wave-uniform
    const uint32_t l = m & 63u, j = m >> 6;
    const uint32_t a = lane_read(r.a, l), b = lane_read(r.b, l), c = lane_read(r.c, l), d = lane_read(r.d, l);
    const uint32_t ab = j & 1u ? b : a, cd = j & 1u ? d : c;
    return j & 2u ? cd : ab;
}
__device__ __forceinline__ Regs<1> regs_load(const uint16_t *row, int lane, int top) {
    return Regs<1>{lane <= top ? (uint32_t)row[lane] : 0x10000u};
}
__device__ __forceinline__ void regs_load_into(Regs<1> &r, const uint16_t *row, int lane, int top) {
    r.a = lane <= top ? (uint32_t)row[lane] : 0x10000u;
}
__device__ __forceinline__ void regs_load_into(Regs<4> &r, const uint16_t *row, int lane, int top) {
    r.a = lane <= top ? (uint32_t)row[lane] : 0x10000u;
    r.b = lane + 64 <= top ? (uint32_t)row[lane + 64] : 0x10000u;
    r.c = lane + 128 <= top ? (uint32_t)row[lane + 128] : 0x10000u;
    r.d = lane + 192 <= top ? (uint32_t)row[lane + 192] : 0x10000u;
}

template <int N>
__device__ __forceinline__ void vm_wait() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// Wave-uniform bit reader.  The current 64-word window of the stream sits in one VGPR (lane l: word base + l, byte-swapped
// and tail-masked); the NEXT window is requested by LDS-DMA into a 2 x 256-byte LDS buffer when a window becomes current and
// picked up from there 64 words later.  No VGPR is the target of a load in the decode loop, so the compiler has nothing to
// put an s_waitcnt vmcnt on (every such wait would also drain the table DMAs of the ring decoder).  `age` counts table blocks
// started since the request: after 3 blocks the in-order vmcnt waits of the ring decoder have covered it, otherwise (short
// streams, the constant-row kernel) the switch waits for everything.
// The reader is a bit POSITION plus the two stream words around it in an SGPR pair: the next 32 bits are one 64-bit shift
// away (peek32), consuming c <= 32 bits into `value` is one more (shift_in), and a word is pulled out of the window VGPR
// only when the position crosses a word boundary (every ~6 symbols) -- no per-symbol refill logic.
struct WaveBits {
    const uint32_t *words;   // 4-byte aligned start of the stream
    uint32_t nbytes;
    int lane;
    uint32_t lds;            // LDS byte address of the 2 window buffers
    uint32_t base;           // first word index of `cur`
    uint32_t cur;            // per-lane window register
    uint32_t age;
    uint32_t pos;            // bit position of the next unread bit
    uint64_t w01;            // stream words (pos >> 5) : (pos >> 5) + 1, zeros past the end

    __device__ __forceinline__ void request(uint32_t b) {   // window starting at word b -> buffer (b / 64) & 1
        const uint32_t idx = b + (uint32_t)lane;
        if (idx * 4u < nbytes)    // lanes past the end leave stale LDS behind; pick_up() zeroes them
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(words + idx),
                                             (__attribute__((address_space(3))) void *)(uintptr_t)(lds + ((b >> 6) & 1u) * 256u),
                                             4, 0, 0);
        age = 0;
    }
    __device__ __forceinline__ uint32_t pick_up(uint32_t b) const {
        uint32_t raw;
        const uint32_t addr = lds + ((b >> 6) & 1u) * 256u + (uint32_t)lane * 4u;
        asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(raw) : "v"(addr) : "memory");
        const uint32_t byte0 = (b + (uint32_t)lane) * 4u;
        uint32_t w = l3c::bswap32(raw);
        if (byte0 < nbytes) {
            const uint32_t rem = nbytes - byte0;   // valid bytes in this word
            if (rem < 4u) w &= 0xFFFFFFFFu << (8u * (4u - rem));
        } else {
            w = 0u;
        }
        return w;
    }
    __device__ __forceinline__ void next_window() {   // the requested window becomes current, the one after it is requested
        base += 64u;
        if (age < 3u) vm_wait<0>();
        cur = pick_up(base);
        request(base + 64u);
    }
    // word `idx` of the stream; idx never decreases and grows by at most one window between calls
    __device__ __forceinline__ uint32_t word(uint32_t idx) {
        if (__builtin_expect(idx - base >= 64u, 0)) next_window();   // window exhausted (wave-uniform, once per 64 words)
        return lane_read(cur, idx - base);
    }
    // lane l: word base + 32 + l of the stream -- the second half of the current window and the first half of the requested one
    __device__ __forceinline__ uint32_t late_window() {
        if (age < 3u) {
            vm_wait<0>();
            age = 3u;
        }
        const uint32_t nxt = pick_up(base + 64u);
        const uint32_t a = (uint32_t)__builtin_amdgcn_ds_bpermute((lane + 32) * 4, (int)cur);    // lanes 0..31: cur[32 + l]
        const uint32_t b = (uint32_t)__builtin_amdgcn_ds_bpermute((lane - 32) * 4, (int)nxt);    // lanes 32..63: nxt[l - 32]
        return lane < 32 ? a : b;
    }
    // pos0: where a previous chunk of the same stream stopped (0 = start of the stream)
    __device__ __forceinline__ void init(const uint32_t *w, uint32_t n, int lane_, uint32_t lds_addr, uint32_t pos0 = 0) {
        words = w;
        nbytes = n;
        lane = lane_;
        lds = lds_addr;
        pos = pos0;
        const uint32_t widx = pos0 >> 5;
        base = widx & ~63u;
        request(base);
        request(base + 64u);
        vm_wait<0>();
        cur = pick_up(base);
        age = 3;
        const uint32_t w0 = word(widx);
        w01 = ((uint64_t)w0 << 32) | word(widx + 1u);
    }
    __device__ __forceinline__ uint32_t peek32() const {   // the next 32 bits, MSB first
        return (uint32_t)((w01 << (pos & 31u)) >> 32);
    }
    __device__ __forceinline__ void skip(int count) {       // count in 0..32
        const uint32_t before = pos >> 5;
        pos += (uint32_t)count;
        const uint32_t after = pos >> 5;
        if (after != before) w01 = (w01 << 32) | word(after + 1u);
    }
    // (value << count) | next `count` bits, count in 0..32
    __device__ __forceinline__ uint32_t shift_in(uint32_t value, int count) {
        const uint32_t r = (uint32_t)(((((uint64_t)value) << 32) | peek32()) << count >> 32);
        skip(count);
        return r;
    }
    __device__ __forceinline__ uint32_t take(int count) {   // next `count` (0..32) bits, MSB first; zeros past the end
        return shift_in(0u, count);
    }
};

// One symbol: search it in `row` (masked by regs_mask), and -- unless it is the stream's last symbol -- advance the state.
// Fast path (validated table, value inside [low, high] as for every stream this coder or the reference produced):
// division-free.  cdf[m] <= count  <=>  t[m] = (span * cdf[m]) >> 16 <= value - low, and the scaled entries are exactly the
// offsets of the interval update (low' = low + t[x], high' = low - 1 + t[x+1]; torchac.cpp:329-345).  low <= high is an
// invariant of validated tables, so value - low <= high - low is the in-range test.  Masked lanes scale to range + 1 > d.
// Otherwise: the reference's arithmetic literally (wrapping 64-bit count, its binary search on non-monotone rows).
template <int NJ>
__device__ __forceinline__ uint32_t decode_symbol(const Regs<NJ> &row, uint32_t &low, uint32_t &high, uint32_t &value,
                                                  WaveBits &src, int top, bool monotone, bool advance) {
    const uint32_t range = high - low, d = value - low;
    uint32_t x;
    if (monotone && d <= range && range != 0xFFFFFFFFu) {   // full range (first symbol): a masked entry would scale to 2^32
        const Regs<NJ> t = regs_scale(row, range);
        const uint32_t rank = regs_rank(t, d);
        x = (rank > 1u ? rank : 1u) - 1u;
        if (advance) {
            const uint32_t t_lo = regs_fetch(t, x);
            const uint32_t t_hi = regs_fetch(t, x + 1u);   // x == top: lanes past the top hold range + 1
            const uint32_t new_low = low + t_lo;
            const uint32_t new_high = x == (uint32_t)top ? high : low - 1u + t_hi;
            int n, m;
            l3c::renorm_counts(new_low, new_high, n, m, low, high);
            const int c = n + m;
            if (c <= 32) {   // one read: ((value << n | bits_n) << m ^ msb) | bits_m  ==  (value << c | bits_c) ^ msb
                const uint32_t bits = src.take(c);
                value = ((uint32_t)(((uint64_t)value << c)) | bits) ^ (m ? 0x80000000u : 0u);
            } else {
                value = n >= 32 ? src.take(32) : ((value << n) | src.take(n));
                value = ((value << m) ^ 0x80000000u) | src.take(m);
            }
        }
        return x;
    }
    const uint32_t count = l3c::decode_count(low, high, value);
    if (monotone) {
        const uint32_t rank = regs_rank(row, count);
        x = (rank > 1u ? rank : 1u) - 1u;
    } else {
        x = l3c::ref_binsearch([&](uint32_t m) { return regs_fetch(row, m); }, count, (uint32_t)top);
    }
    if (advance) {
        const uint32_t c_lo = regs_fetch(row, x);
        const uint32_t c_hi = (x == (uint32_t)top) ? 0x10000u : regs_fetch(row, x + 1u);
        l3c::decode_advance(low, high, value, c_lo, c_hi, src);
    }
    return x;
}

__device__ __forceinline__ uint32_t mul_hi(uint32_t a, uint32_t b) { return (uint32_t)(((uint64_t)a * b) >> 32); }
// lane (i & 63) keeps symbol i until the 64-symbol block is stored with one coalesced write
__device__ __forceinline__ void keep_symbol(int16_t *dst, uint32_t i, uint32_t n_sym, uint32_t x, int lane, int &kept) {
    if ((int)(i & 63u) == lane) kept = (int)x;   // (v_writelane_b32 would need two SGPR operands: over gfx9's constant-bus limit)
    if ((i & 63u) == 63u || i == n_sym - 1u) {
        const uint32_t t = (i & ~63u) + (uint32_t)lane;
        if (t <= i) dst[t] = (int16_t)kept;
    }
}

// Every symbol uses the SAME row (row_stride == 0: the uniform prior of the coarsest scale, bitcoding.py:297-323).
template <int NJ>
__global__ __launch_bounds__(64) void ac_decode_const_row_kernel(const uint16_t *__restrict__ cdf, int Lp,
                                                                 const uint8_t *__restrict__ in,
                                                                 const int64_t *__restrict__ in_offsets,
                                                                 const uint32_t *__restrict__ in_nbytes, uint32_t n_sym,
                                                                 int monotone, int16_t *__restrict__ sym_out) {
    const int64_t s = blockIdx.x;
    const int lane = threadIdx.x;
    const int top = Lp - 2;
    int16_t *dst = sym_out + s * (int64_t)n_sym;
    __shared__ __attribute__((aligned(16))) uint8_t window[512];
    Regs<NJ> row;
    regs_load_into(row, cdf, lane, top);
    WaveBits src;
    src.init(reinterpret_cast<const uint32_t *>(in + in_offsets[s]), in_nbytes[s], lane,
             (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint8_t *)window);
    uint32_t low = 0, high = 0xFFFFFFFFu;
    uint32_t value = src.take(32);
    int kept = 0;
    for (uint32_t i = 0; i < n_sym; ++i) {
        const uint32_t x = decode_symbol<NJ>(row, low, high, value, src, top, monotone != 0, i != n_sym - 1u);
        keep_symbol(dst, i, n_sym, x, lane, kept);
    }
}

// Per-symbol rows (row_stride == Lp).  The first version of this kernel kept 4 rows in flight in registers and measured
// ~630 ns per symbol: the serial chain spent most of its time waiting for HBM.  Here the table streams through an LDS ring
// with LDS-DMA (`global_load_lds_dwordx4`, no VGPR staging): NB = 4 blocks of R rows; block k+3 is requested when block k
// starts and block k+1 must have landed by then (s_waitcnt vmcnt(2*IPB): vector-memory operations retire in order), so 2-3
// blocks (>= 32 symbols, several microseconds) of lookahead are always in flight.  The row of symbol i+1 is read from LDS
// into registers while symbol i is being decoded.
//   * The LDS reads are inline asm: the compiler cannot prove that a ds_read does not alias an outstanding LDS-DMA write and
//     would put vmcnt(0) before every one of them.  They are issued at the top of an iteration (lds_row_issue) and waited
//     for at its end (lds_row_take).
//   * DMA windows are 16-byte granules at absolute addresses.  A granule is only requested if it holds at least one byte
//     of the table (others are redirected to the table's last granule), so the reads never leave the pages of the table.
// IPB_: KB per ring block.  The wide-alphabet (RGB, Lp = 257) decoder comes in two sizes, picked per launch: 9 KB blocks (17 rows per
// block, 36.5 KB of LDS per stream: the longest DMA lookahead, for a few streams on an empty machine) and 3 KB blocks (5 rows, 12.5 KB:
// for large batches, where the table kernel of the next chunk step runs beside the decoders and every KB of LDS a decoder holds
// keeps table blocks off its CU).  [measured, profiles/r04_decode_isolation_experiments.log: 3 KB blocks 0.622 -> 0.570 s per batch of
// 128, 0.188 -> 0.199 s for one image]
template <int NJ, int IPB_ = (NJ == 1 ? 3 : 9)>
struct RingCfg {
    static constexpr int IPB = IPB_;   // 1 KB DMA instructions per block
    static constexpr int NB = 4;
    static constexpr int BLOCK_BYTES = IPB * 1024;
};

__device__ __forceinline__ int ring_rows_per_block(int Lp, int block_bytes) {
    const int r = (block_bytes - 16) / (Lp * 2);
    return r < 32 ? r : 32;
}

__device__ __forceinline__ void lds_row_issue(uint32_t addr, Regs<1> &r) {
    asm volatile("ds_read_u16 %0, %1" : "=&v"(r.a) : "v"(addr));
}
__device__ __forceinline__ void lds_row_issue(uint32_t addr, Regs<4> &r) {
    asm volatile(
        "ds_read_u16 %0, %4\n\tds_read_u16 %1, %4 offset:128\n\tds_read_u16 %2, %4 offset:256\n\tds_read_u16 %3, %4 offset:384"
        : "=&v"(r.a), "=&v"(r.b), "=&v"(r.c), "=&v"(r.d)
        : "v"(addr));
}
// Wait for the reads of lds_row_issue and move the row out of the registers they target, in ONE asm statement: between the
// two statements `pending` has no uses, so the compiler has no reason to copy registers whose LDS data is still in flight
// (it cannot know about that), and what leaves this statement are ordinary, complete values.
__device__ __forceinline__ void lds_row_take(Regs<1> &row, const Regs<1> &pending) {
    asm volatile("s_waitcnt lgkmcnt(0)\n\tv_mov_b32 %0, %1" : "=&v"(row.a) : "v"(pending.a));
}
__device__ __forceinline__ void lds_row_take(Regs<4> &row, const Regs<4> &pending) {
    asm volatile("s_waitcnt lgkmcnt(0)\n\tv_mov_b32 %0, %4\n\tv_mov_b32 %1, %5\n\tv_mov_b32 %2, %6\n\tv_mov_b32 %3, %7"
                 : "=&v"(row.a), "=&v"(row.b), "=&v"(row.c), "=&v"(row.d)
                 : "v"(pending.a), "v"(pending.b), "v"(pending.c), "v"(pending.d));
}

// Coder state of one stream between two chunks of it (l3c_ac_decode_chunk); 32 bytes, opaque to the caller.
struct DecodeState {
    uint32_t low, high, value, pos;
    uint32_t pad[4];
};
static_assert(sizeof(DecodeState) == 32, "l3c_ac_decode_state_bytes");

// ---- window rows (round 5; csrc/dmll_core.h) --------------------------------------------------------------------------------
// The RGB decoder's table rows come in two forms per (image, chunk): 65-entry WINDOW rows around the mixture's mean or full 257-entry
// rows; which one is a pure function of the miss count the stream's decoder wrote two chunks earlier (l3c::use_window), evaluated
// alike by the table kernel and here.  A symbol outside its window -- x' = 0 or 63 where that is not the alphabet's end -- makes the
// wavefront evaluate the pixel's full row itself (window_full_row: the table kernel's device functions, the same bits) and rank
// against that.  Window streams are decoded by the <1, ..., WINDOW> instantiations (one row register, like the bottleneck scales),
// full-row streams by the classic ones; both are launched over the same grid and a block leaves at once when the stream is not its kind.
struct WindowCtx {
    const int32_t *stats_in;       // [n_streams] misses two chunks ago (negative: unknown); nullptr: no window rows at all
    int32_t *stats_out;            // [n_streams] misses of this chunk
    const float *P;                // [B][HW][4 C K]
    const int16_t *sym;            // [B][C][HW]: the channels decoded so far
    const float *targets;          // [257]
    int64_t HW, pix0;
    int C, K, c;
};

// (no "memory" clobber: behind one the compiler re-reads everything it holds from memory -- the kernel arguments among it -- once per
// ring block; the ring's contents are ordered by the volatile DMA waits, which volatile asm statements are never moved across)
__device__ __forceinline__ uint32_t lds_read_u16_now(uint32_t addr) {
    uint32_t v;
    asm volatile("ds_read_u16 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr));
    return v;
}

// All 256 entries of the full row of pixel n of image b (lane l: entries l, l + 64, l + 128, l + 192), by the wavefront itself: every
// lane walks the K components (redundantly: the parameters are wave-uniform) and sums its four entries in the table kernel's order.
__device__ __forceinline__ Regs<4> window_full_row(const WindowCtx &w, int64_t b, int64_t n, int lane) {
    const int C = w.C, K = w.K, c = w.c;
    const float *px = w.P + (b * w.HW + n) * (int64_t)(4 * C * K);
    auto get = [&](int ch) { return px[ch]; };
    float x0 = 0.f, x1 = 0.f;
    if (c > 0) {
        x0 = (float)w.sym[(b * C + 0) * w.HW + n];
        if (c > 1) x1 = (float)w.sym[(b * C + 1) * w.HW + n];
    }
    const l3c::MixStats st = l3c::mix_stats(get, C, K, c);
    const float t0 = w.targets[lane], t1 = w.targets[lane + 64], t2 = w.targets[lane + 128], t3 = w.targets[lane + 192];
    float a0 = 0.0f, a1 = 0.0f, a2 = 0.0f, a3 = 0.0f;
    for (int k = 0; k < K; ++k) {
        const l3c::MixComponent m = l3c::mix_component(get, st, C, K, 1, c, k, x0, x1);
        const float inv = expf(-m.log_sigma);
        a0 = a0 + l3c::cdf_term(m.pi, m.mu, inv, t0);
        a1 = a1 + l3c::cdf_term(m.pi, m.mu, inv, t1);
        a2 = a2 + l3c::cdf_term(m.pi, m.mu, inv, t2);
        a3 = a3 + l3c::cdf_term(m.pi, m.mu, inv, t3);
    }
    const float scale = (float)(65536 - 256);
    return Regs<4>{l3c::cdf_quantise(a0, scale, lane), l3c::cdf_quantise(a1, scale, lane + 64), l3c::cdf_quantise(a2, scale, lane + 128),
                   l3c::cdf_quantise(a3, scale, lane + 192)};
}

// entries 0 .. 255 strictly increasing?  (the table kernel has only checked the window's part of this row)
__device__ __forceinline__ bool full_row_monotone(const Regs<4> &r, int lane) {
    const uint32_t na = (uint32_t)__shfl_down((int)r.a, 1, 64), nb = (uint32_t)__shfl_down((int)r.b, 1, 64);
    const uint32_t nc = (uint32_t)__shfl_down((int)r.c, 1, 64), nd = (uint32_t)__shfl_down((int)r.d, 1, 64);
    bool bad = lane < 63 && (!(r.a < na) || !(r.b < nb) || !(r.c < nc) || !(r.d < nd));
    const uint32_t b0 = lane_read(r.b, 0), c0 = lane_read(r.c, 0), d0 = lane_read(r.d, 0);
    bad = bad || (lane == 63 && (!(r.a < b0) || !(r.b < c0) || !(r.c < d0)));
    return !__any(bad);
}

struct DecodeArgs {
    const uint16_t *cdf;           // rows of THIS chunk: [n_streams][n_sym][Lp]
    int Lp;
    int64_t table_bytes;
    const uint8_t *in;
    const int64_t *in_offsets;
    const uint32_t *in_nbytes;
    int64_t n_streams;
    uint32_t n_sym;                // symbols of this chunk
    int monotone;                  // host-side knowledge about the table (used when `flag` is null)
    const int32_t *flag;           // device-side: != 0 -> table not validated (overrides `monotone`)
    int force;                     // generic pass: decode every stream, not only the marked ones
    int final_chunk;               // the stream ends with this chunk: its last symbol does not advance the state
    const DecodeState *state_in;   // null: start of the stream
    DecodeState *state_out;        // null: not needed
    int16_t *sym_out;
    int64_t sym_stride, sym_offset;   // stream s writes sym_out[s * sym_stride + sym_offset + i]
    WindowCtx win;                 // window rows (round 5): win.stats_in == nullptr -> classic rows
    // RAGGED batches (round 6, l3c_decode_rgb_ragged: streams of DIFFERENT lengths -- images of different sizes -- in one launch; null = every
    // stream n_sym symbols, rows at s * n_sym * Lp, output at s * sym_stride + sym_offset): per stream s its symbol count of this chunk, the
    // byte offset of its rows inside `cdf`, its image's first pixel / pixel count / this chunk's first pixel (the output goes to
    // sym_out + r_C * r_pixbase[s] + r_c * r_hw[s] + r_pix0[s]; the window context's P / sym are offset likewise)
    const int64_t *r_npix, *r_table_off, *r_pixbase, *r_hw, *r_pix0;
    int r_C, r_c;
};

// the stream's view of a part: rectangular or ragged
struct StreamView {
    uint32_t n_sym;
    uint64_t rows;                 // address of the stream's first row
    int16_t *dst;
    WindowCtx win;                 // (P, sym, HW, pix0 of THIS stream's image when ragged; index the image as 0 then)
    int64_t img;                   // image index to hand to the window functions
};
__device__ __forceinline__ StreamView stream_view(const DecodeArgs &a, int64_t s) {
    StreamView v;
    v.win = a.win;
    if (a.r_npix) {
        const int64_t pb = a.r_pixbase[s], hw = a.r_hw[s], p0 = a.r_pix0[s];
        v.n_sym = (uint32_t)a.r_npix[s];
        v.rows = reinterpret_cast<uint64_t>(a.cdf) + (uint64_t)a.r_table_off[s];
        v.dst = a.sym_out + a.r_C * pb + a.r_c * hw + p0;
        if (a.win.stats_in) {
            v.win.P = a.win.P + pb * (int64_t)(4 * a.win.C * a.win.K);
            v.win.sym = a.win.sym ? a.win.sym + pb * a.win.C : nullptr;
            v.win.HW = hw;
            v.win.pix0 = p0;
        }
        v.img = 0;
    } else {
        v.n_sym = a.n_sym;
        v.rows = reinterpret_cast<uint64_t>(a.cdf) + (uint64_t)s * a.n_sym * ((uint64_t)a.Lp * 2u);
        v.dst = a.sym_out + s * a.sym_stride + a.sym_offset;
        v.img = s;
    }
    return v;
}

// The GENERIC decoder: the reference's arithmetic literally (decode_symbol), for every stream when the table is not validated,
// otherwise only for the streams the fast pass (ac_decode_lean_kernel, below) has marked by the sentinel -1 in the chunk's
// first output symbol (state_out untouched) -- streams whose value left [low, high], which this coder never writes.
// Up to 8 independent decode calls in one launch (blockIdx.y selects the part): the chunk-pipelined RGB decode runs the R, G
// and B chunks of one pipeline step side by side without relying on several HIP streams reaching distinct hardware queues.
struct DecodeArgsPack {
    static constexpr int N = 8;
    DecodeArgs part[N];
};

// WINDOW: the instantiation for the streams whose rows of this chunk are window rows (NJ == 1): every symbol is ranked inside its window;
// a miss -- and, when the table is not validated, EVERY symbol -- goes through the pixel's full row, evaluated here, with the reference's
// literal arithmetic (decode_symbol<4>), so that this pass decodes what the classic generic pass decodes from full rows.
template <int NJ, int IPB_, bool WINDOW>
__device__ __forceinline__ void ring_decode_body(const DecodeArgs &a, uint8_t *ring) {
    static_assert(!WINDOW || NJ == 1, "window rows fill one row register");
    using C = RingCfg<NJ, IPB_>;
    const uint16_t *cdf = a.cdf;
    const int64_t table_bytes = a.table_bytes;
    const bool has_win = a.win.stats_in != nullptr;
    if (WINDOW != (has_win && l3c::use_window(a.win.stats_in[blockIdx.x]))) return;   // not this kernel's kind of stream
    const int Lp = WINDOW ? l3c::kWinLp : a.Lp;
    const bool validated = a.flag ? (*a.flag == 0) : (a.monotone != 0);
    const int64_t s = blockIdx.x;
    const StreamView sv = stream_view(a, s);
    const uint32_t n_sym = sv.n_sym;
    const int lane = threadIdx.x;
    const int top = Lp - 2;
    const uint32_t row_bytes = (uint32_t)Lp * 2u;
    const uint32_t R = (uint32_t)ring_rows_per_block(Lp, C::BLOCK_BYTES);
    const uint32_t n_blocks = (n_sym + R - 1u) / R;
    const uint64_t tab0 = reinterpret_cast<uint64_t>(cdf);
    const uint64_t stream0 = sv.rows;   // this stream's first row (its full-size slot)
    const uint64_t last_granule = (tab0 + (uint64_t)table_bytes - 1u) & ~(uint64_t)15;
    int16_t *dst = sv.dst;
    if (!a.force && validated && dst[0] != (int16_t)-1) return;   // decoded by the fast pass (ac_decode_lean_kernel)
    const uint32_t ring_base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint8_t *)ring;

    auto request_block = [&](uint32_t k) {   // DMA the 16-byte granules holding rows [kR, (k+1)R) into slot k % NB
        const uint64_t a = (stream0 + (uint64_t)k * R * row_bytes) & ~(uint64_t)15;
        uint8_t *slot = ring + (k % C::NB) * C::BLOCK_BYTES;
#pragma unroll
        for (int q = 0; q < C::IPB; ++q) {
            uint64_t g = a + (uint64_t)(q * 1024 + lane * 16);
            g = g <= last_granule ? g : last_granule;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)g,
                                             (__attribute__((address_space(3))) void *)(slot + q * 1024), 16, 0, 0);
        }
    };
    auto block_addr = [&](uint32_t k) -> uint32_t {   // LDS byte address of this lane's first entry of row k * R
        const uint64_t b = stream0 + (uint64_t)k * R * row_bytes;
        return ring_base + (k % C::NB) * C::BLOCK_BYTES + (uint32_t)(b & 15u) + (uint32_t)lane * 2u;
    };

    WaveBits src;
    const uint32_t *words = reinterpret_cast<const uint32_t *>(a.in + a.in_offsets[s]);
    uint32_t low = 0, high = 0xFFFFFFFFu, value;
    if (a.state_in) {
        const DecodeState st = a.state_in[s];
        src.init(words, a.in_nbytes[s], lane, ring_base + C::NB * C::BLOCK_BYTES, st.pos);
        low = st.low;
        high = st.high;
        value = st.value;
    } else {
        src.init(words, a.in_nbytes[s], lane, ring_base + C::NB * C::BLOCK_BYTES);
        value = src.take(32);
    }
    const uint32_t no_advance = a.final_chunk ? n_sym - 1u : 0xFFFFFFFFu;   // torchac.cpp:335-337

#pragma unroll
    for (uint32_t k = 0; k < (uint32_t)C::NB - 1u; ++k)
        if (k < n_blocks) request_block(k);
    vm_wait<0>();

    Regs<NJ> row, pending;
    lds_row_issue(block_addr(0), pending);
    lds_row_take(row, pending);

    int kept = 0;
    uint32_t i = 0;
    for (uint32_t k = 0; k < n_blocks; ++k) {
        // Request block k + NB - 1 (its slot held block k - 1, fully consumed), then make sure block k + 1 -- whose first row
        // is prefetched at the end of this block -- has landed: only the two newest requests may stay in flight.
        if (k + C::NB - 1u < n_blocks) {
            request_block(k + C::NB - 1u);
            if (k > 0) vm_wait<2 * C::IPB>();   // k == 0: blocks 0 .. NB-2 were waited for above
        } else {
            vm_wait<0>();
        }
        src.age += 1u;
        const uint32_t i_end = (k + 1u) * R < n_sym ? (k + 1u) * R : n_sym;
        const uint32_t i_cross = (k + 1u) * R - 1u;   // the symbol whose successor row lives in block k + 1
        const uint32_t addr_cross = block_addr(k + 1u);
        uint32_t addr_next = block_addr(k) + (i - k * R + 1u) * row_bytes;
        for (; i < i_end; ++i) {
            regs_mask(row, lane, top);
            lds_row_issue(i == i_cross ? addr_cross : addr_next, pending);   // row i + 1 (past the end: never used)
            addr_next += row_bytes;
            uint32_t x = 0;
            if constexpr (WINDOW) {
                bool need_full = !validated;
                if (validated) {   // rank inside the window first
                    const uint32_t xw = decode_symbol<1>(row, low, high, value, src, top, true, false);   // no advance: the state is untouched
                    // the row's entry 64: its window offset (the read also lands the prefetched row; lds_row_take below still moves it)
                    const uint32_t w0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)lds_read_u16_now(
                        block_addr(k) - (uint32_t)lane * 2u + (i - k * R) * row_bytes + 2u * (uint32_t)l3c::kWinTop + 2u));
                    if (l3c::window_miss(xw, w0)) {
                        need_full = true;
                    } else {
                        decode_symbol<1>(row, low, high, value, src, top, true, i != no_advance);
                        x = w0 + xw;
                    }
                }
                lds_row_take(row, pending);   // (before the long path: no LDS read in flight while the full row is evaluated)
                if (need_full) {
                    const Regs<4> full = window_full_row(sv.win, sv.img, sv.win.pix0 + (int64_t)i, lane);
                    x = decode_symbol<4>(full, low, high, value, src, 255, validated && full_row_monotone(full, lane), i != no_advance);
                }
                keep_symbol(dst, i, n_sym, x, lane, kept);
            } else {
                x = decode_symbol<NJ>(row, low, high, value, src, top, validated, i != no_advance);
                keep_symbol(dst, i, n_sym, x, lane, kept);
                lds_row_take(row, pending);   // the only take of the loop, on every path (tools/check_asm_prefetch.py)
            }
        }
    }
    if (a.state_out && lane == 0) a.state_out[s] = DecodeState{low, high, value, src.pos, {0u, 0u, 0u, 0u}};
    // a stream that needed this pass reports no usable statistics: the chunk after the next one gets full rows
    if (has_win && a.win.stats_out && lane == 0) a.win.stats_out[s] = 0x7FFFFFFF;
}

// WITH_WINDOW: the launch may hold streams on window rows (RGB parts of a windowed decode): such a block runs the <1, 3, WINDOW> body, the
// others the classic one -- in ONE kernel, so that the two kinds of streams of a pipeline step run side by side (as two launches on one HIP
// stream they would run one after the other: the chains are latency-bound, the step would take the SUM of its slowest window stream and
// its slowest full-row stream).
template <int NJ, int IPB_ = (NJ == 1 ? 3 : 9), bool WITH_WINDOW = false>
__global__ __launch_bounds__(64) void ac_decode_ring_kernel(const DecodeArgsPack pack) {
    using C = RingCfg<NJ, IPB_>;
    static_assert(!WITH_WINDOW || (NJ == 4 && IPB_ >= 3), "the window body's ring (4 x 3 KB) must fit the classic one");
    __shared__ __attribute__((aligned(16))) uint8_t ring[C::NB * C::BLOCK_BYTES + 512];   // + the bit reader's windows
    const DecodeArgs &a = pack.part[blockIdx.y];
    if ((int64_t)blockIdx.x >= a.n_streams) return;
    if constexpr (WITH_WINDOW) {
        if (a.win.stats_in != nullptr && l3c::use_window(a.win.stats_in[blockIdx.x])) {
            ring_decode_body<1, 3, true>(a, ring);
            return;
        }
    }
    ring_decode_body<NJ, IPB_, false>(a, ring);
}

// ---- the lean fast decoder (round 4) --------------------------------------------------------------------------------------
// The fast pass over validated tables.  Round 3's version of it (the generic loop above with a division-free symbol) spent
// ~130 instructions on a symbol, and a lone wavefront issues one every ~4 cycles whatever they are: the chain costs its
// instruction COUNT.  Same ring, same arithmetic, same DecodeState between chunks, ~40 % fewer instructions per symbol:
//   * rows land in the HIGH half-words of two alternating register sets (ds_read_u16_d16_hi; the low halves are zero and
//     stay zero): no shift, no copy out of a staging set; the loop is unrolled by two rows
//   * the state is (low, ~high, range) as in the encoder's lane pairs (csrc/ac_core.h): ONE count n + m renormalises both
//     bounds, range = ~(low + ~high), the underflow flip of `value` is the bit the shift pushes out of low
//   * the bit reader is a 64-bit buffer of upcoming bits: consuming n + m <= 31 bits is one 64-bit shift of value:buffer,
//     a word is pulled out of the window register only when fewer than 32 bits are left (every ~5 symbols)
//   * full ring blocks hold neither the stream's last symbol nor a ragged end: no per-symbol tests for either; decoded
//     symbols are kept in lane (row within the block) and stored once per block
//   * Lp == 257 (the RGB alphabet): all 256 entries of the four registers are table entries, no validity masks
// A stream whose value leaves [low, high] (never for a stream of this coder or the reference) marks itself and is redone by
// the generic instantiation, exactly as before.
template <int NJ>
struct RowHi;
template <>
struct RowHi<1> {
    uint32_t a;
};
template <>
struct RowHi<4> {
    uint32_t a, b, c, d;
};
__device__ __forceinline__ void row_hi_issue(uint32_t addr, RowHi<1> &r) {
    asm volatile("ds_read_u16_d16_hi %0, %1" : "+v"(r.a) : "v"(addr));
}
__device__ __forceinline__ void row_hi_issue(uint32_t addr, RowHi<4> &r) {
    asm volatile(
        "ds_read_u16_d16_hi %0, %4\n\tds_read_u16_d16_hi %1, %4 offset:128\n\tds_read_u16_d16_hi %2, %4 offset:256\n\t"
        "ds_read_u16_d16_hi %3, %4 offset:384"
        : "+v"(r.a), "+v"(r.b), "+v"(r.c), "+v"(r.d)
        : "v"(addr));
}
// the reads of row_hi_issue have landed; ties the registers to the wait so that no use moves above it
__device__ __forceinline__ void row_hi_wait(RowHi<1> &r) { asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(r.a)); }
__device__ __forceinline__ void row_hi_wait(RowHi<4> &r) {
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(r.a), "+v"(r.b), "+v"(r.c), "+v"(r.d));
}
__device__ __forceinline__ RowHi<1> row_mul_hi(const RowHi<1> &r, uint32_t s) { return RowHi<1>{mul_hi(r.a, s)}; }
__device__ __forceinline__ RowHi<4> row_mul_hi(const RowHi<4> &r, uint32_t s) {
    return RowHi<4>{mul_hi(r.a, s), mul_hi(r.b, s), mul_hi(r.c, s), mul_hi(r.d, s)};
}
template <bool ALLVALID>
__device__ __forceinline__ uint32_t row_rank(const RowHi<1> &t, uint32_t d, const ValidLanes<1> &v) {
    return ALLVALID ? count_le(t.a, d) : count_le_valid(t.a, d, v.a);
}
template <bool ALLVALID>
__device__ __forceinline__ uint32_t row_rank(const RowHi<4> &t, uint32_t d, const ValidLanes<4> &v) {
    if (ALLVALID) return (count_le(t.a, d) + count_le(t.b, d)) + (count_le(t.c, d) + count_le(t.d, d));
    return (count_le_valid(t.a, d, v.a) + count_le_valid(t.b, d, v.b)) + (count_le_valid(t.c, d, v.c) + count_le_valid(t.d, d, v.d));
}
// entries x and x1 = x + 1 of a row: pick the register holding x once (wave-uniform selects), two v_readlanes; only when x is
// the last lane of a register does x1 live in the next one
__device__ __forceinline__ void row_fetch2(const RowHi<1> &r, uint32_t x, uint32_t x1, uint32_t &lo, uint32_t &hi) {
    lo = lane_read(r.a, x & 63u);
    hi = lane_read(r.a, x1 & 63u);
}
__device__ __forceinline__ void row_fetch2(const RowHi<4> &r, uint32_t x, uint32_t x1, uint32_t &lo, uint32_t &hi) {
    const uint32_t j = x >> 6;
    const uint32_t ab = j & 1u ? r.b : r.a, cd = j & 1u ? r.d : r.c;
    const uint32_t sel = j & 2u ? cd : ab;
    lo = lane_read(sel, x & 63u);
    if (__builtin_expect((x1 & 63u) != 0u, 1)) {
        hi = lane_read(sel, x1 & 63u);
    } else {
        const uint32_t j1 = x1 >> 6;   // 1..4; 4 (x == top == 255) is replaced by the caller
        const uint32_t bc = j1 & 1u ? r.b : r.c;
        hi = lane_read(j1 == 3u ? r.d : bc, 0u);
    }
}

struct LeanState {
    uint32_t low, nh, range;          // nh = ~high, range = high - low
    uint64_t vb;                      // high word: value; low word: scratch (the next 32 bits are copied in before a shift)
    uint64_t buf;                     // upcoming bits of the stream, MSB first
    uint32_t nbits;                   // valid bits in buf: >= 32 before every symbol
    uint32_t widx;                    // next stream word to enter buf
    uint32_t bad;                     // != 0: value left [low, high] at some symbol (checked once per ring block)
};

// One symbol on a validated table.  A value outside [low, high] only raises st.bad (the state then runs on garbage until the
// block ends: every index it forms is masked by the hardware, stream words past the end read as zero).
template <int NJ, bool ALLVALID, bool FULL>
__device__ __forceinline__ void lean_symbol_body(const RowHi<NJ> &row, const ValidLanes<NJ> &valid, uint32_t top, bool advance,
                                                 LeanState &st, WaveBits &src, uint32_t &x) {
    const uint32_t d = (uint32_t)(st.vb >> 32) - st.low;
    st.bad |= d > st.range ? 1u : 0u;
    uint32_t x1, t_lo, t_hi;
    RowHi<NJ> t;
    if (FULL) t = row;
    else t = row_mul_hi(row, st.range + 1u);
    const uint32_t rank = row_rank<ALLVALID>(t, d, valid);
    x1 = rank > 1u ? rank : 1u;            // x + 1
    asm("" : "+s"(x1));                    // keep it scalar: max - 1 would be canonicalised to a VALU-only saturating subtract
    x = x1 - 1u;
    row_fetch2(t, x, x1, t_lo, t_hi);
    if (advance) {   // x == top: t_hi is not a table entry, replaced
        uint32_t msb;
        const int c = l3c::lean_advance(st.low, st.nh, st.range, t_lo, t_hi, x == top, msb);
        // ((value << n | bits_n) << m ^ msb) | bits_m  ==  (value << c | bits_c) ^ msb
        st.vb = (((st.vb & 0xFFFFFFFF00000000ull) | (st.buf >> 32)) << c) ^ ((uint64_t)msb << 32);
        st.buf <<= c;
        st.nbits -= (uint32_t)c;
        if (__builtin_expect(st.nbits < 32u, 0)) {
            st.buf |= (uint64_t)src.word(st.widx) << (32u - st.nbits);
            st.widx += 1u;
            st.nbits += 32u;
        }
    }
}
// The mixture of pixel n of image b, one component per lane (lane k < K: pi_k, mu_k, 1 / sigma_k; the softmax's denominator is the
// SEQUENTIAL sum of the numerators, gathered lane by lane), and from it the 64 entries cdf[wbase + lane] as a window row of the lean
// decoder (entries in the high half-words): what the table kernel would have written for a window at offset wbase.
struct LaneMixture {
    float pi, mu, inv;
};
__device__ __forceinline__ LaneMixture window_lane_mixture(const WindowCtx &w, int64_t b, int64_t n, int lane) {
    const int C = w.C, K = w.K, c = w.c;
    const float *px = w.P + (b * w.HW + n) * (int64_t)(4 * C * K);
    auto get = [&](int ch) { return px[ch]; };
    float x0 = 0.f, x1 = 0.f;
    if (c > 0) {
        x0 = (float)w.sym[(b * C + 0) * w.HW + n];
        if (c > 1) x1 = (float)w.sym[(b * C + 1) * w.HW + n];
    }
    const int k = lane < K ? lane : K - 1;
    l3c::MixStats st;
    st.max_logit = get(c * K);
    for (int j = 1; j < K; ++j) st.max_logit = fmaxf(st.max_logit, get(c * K + j));
    const float e_k = expf(get(c * K + k) - st.max_logit);
    st.denom = 0.0f;
    for (int j = 0; j < K; ++j) st.denom = st.denom + __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, e_k), j));
    const l3c::MixComponent m = l3c::mix_component_e(get, st, e_k, C, K, 1, c, k, x0, x1);
    return LaneMixture{m.pi, m.mu, expf(-m.log_sigma)};
}
__device__ __forceinline__ RowHi<1> window_row_at(const WindowCtx &w, const LaneMixture &mx, int wbase, int lane, bool &monotone) {
    const float t = w.targets[wbase + lane];
    float acc = 0.0f;
    for (int j = 0; j < w.K; ++j) {
        const float pi = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, mx.pi), j));
        const float mu = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, mx.mu), j));
        const float inv = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, mx.inv), j));
        acc = acc + l3c::cdf_term(pi, mu, inv, t);
    }
    const uint32_t v = l3c::cdf_quantise(acc, (float)(65536 - 256), wbase + lane);
    const uint32_t nxt = (uint32_t)__shfl_down((int)v, 1, 64);
    monotone = !__any(lane < 63 && !(v < nxt));
    return RowHi<1>{v << 16};
}

// A symbol of a WINDOW row (csrc/dmll_core.h; entries 0 .. 63 in the high half-words like every lean row; w0: the window's offset, the
// row's entry 64, read from the ring by the caller).
// Ranked inside the window; unless that is a miss, the state advances as for a 64-symbol alphabet whose top symbol is entry 63, and
// x = w0 + x'.  On a MISS nothing is touched and its direction is returned (-1: the symbol lies below the window, +1: above; 0: decoded):
// the caller then evaluates the 64 entries next to the window on that side (window_row_at, once the prefetched LDS reads have landed) and
// tries again, until the symbol is inside.
template <bool FULLRANGE>
__device__ __forceinline__ int lean_symbol_window_body(const RowHi<1> &row, uint32_t w0, bool advance, LeanState &st, WaveBits &src, uint32_t &x) {
    const uint32_t d = (uint32_t)(st.vb >> 32) - st.low;
    RowHi<1> t;
    if (FULLRANGE) t = row;
    else t = row_mul_hi(row, st.range + 1u);
    const uint32_t rank = count_le(t.a, d);
    uint32_t x1 = rank > 1u ? rank : 1u;
    asm("" : "+s"(x1));
    const uint32_t xw = x1 - 1u;
    if (l3c::window_miss(xw, w0) && d <= st.range) return xw == 0u ? -1 : 1;   // (a value outside [low, high] is not a miss: it marks the stream below)
    st.bad |= d > st.range ? 1u : 0u;
    uint32_t t_lo, t_hi;
    row_fetch2(t, xw, x1, t_lo, t_hi);
    x = w0 + xw;
    if (advance) {
        uint32_t msb;
        const int c = l3c::lean_advance(st.low, st.nh, st.range, t_lo, t_hi, xw == (uint32_t)l3c::kWinTop, msb);
        st.vb = (((st.vb & 0xFFFFFFFF00000000ull) | (st.buf >> 32)) << c) ^ ((uint64_t)msb << 32);
        st.buf <<= c;
        st.nbits -= (uint32_t)c;
        if (__builtin_expect(st.nbits < 32u, 0)) {
            st.buf |= (uint64_t)src.word(st.widx) << (32u - st.nbits);
            st.widx += 1u;
            st.nbits += 32u;
        }
    }
    return 0;
}
__device__ __forceinline__ int lean_symbol_window(const RowHi<1> &row, uint32_t w0, bool advance, LeanState &st, WaveBits &src, uint32_t &x) {
    if (__builtin_expect(st.range == 0xFFFFFFFFu, 0)) return lean_symbol_window_body<true>(row, w0, advance, st, src, x);
    return lean_symbol_window_body<false>(row, w0, advance, st, src, x);
}

template <int NJ, bool ALLVALID>
__device__ __forceinline__ void lean_symbol(const RowHi<NJ> &row, const ValidLanes<NJ> &valid, uint32_t top, bool advance,
                                            LeanState &st, WaveBits &src, uint32_t &x) {
    // the interval is the whole 32-bit range (span = 2^32: the first symbol, and again whenever a symbol's interval is an aligned
    // power of two): (span * cdf) >> 16 is cdf << 16 itself -- a separate instantiation of the whole symbol
    if (__builtin_expect(st.range == 0xFFFFFFFFu, 0)) lean_symbol_body<NJ, ALLVALID, true>(row, valid, top, advance, st, src, x);
    else lean_symbol_body<NJ, ALLVALID, false>(row, valid, top, advance, st, src, x);
}

// ---- a FULL ring block of the 256-symbol alphabet as one hand-written loop ---------------------------------------------------
// What a lone wavefront pays for (tools/issue_microbench.hip, ns per instruction at ~2.4 GHz): 1.75 for almost anything, vector
// or scalar, dependent or not -- but 10 for a v_cmp -> s_bcnt1 pair on the same vcc (four pairs back to back, as the compiler
// emits them: 40; four v_cmp into four SGPR pairs and then four s_bcnt1: 20), +8 when a v_readlane result is needed by the
// scalar unit at once, 8.4 for a taken branch, 4.2 for a conditional branch that is not taken, 1.75 for every s_nop.  The
// compiler's symbol (lean_symbol above) has ~85 instructions with six conditional branches, two or three of them taken, and
// the compares on one vcc: 290 ns.  Here a symbol is 66 instructions WITHOUT A BRANCH:
//   * the conditions that need other code are only RECORDED -- value outside [low, high] (bad), the interval being the whole
//     32-bit range at a symbol (minspan == 0: span = range + 1 wraps) -- and the caller decodes the block again from the saved
//     state with lean_symbol if one was; the caller hands over a bit window that cannot run out inside the block
//   * the register holding entry x (and the one holding x + 1) is picked by VGPR indexing (s_set_gpr_idx_on over v96..v100,
//     fixed registers so that they are consecutive) instead of scalar tests, v_cndmask and the hand-over back
//   * the refill of the bit buffer is arithmetic: the word is read every symbol (v_readlane, off the chain) and or-ed in as
//     zero when it is not needed
//   * value : next 32 bits, the bit buffer and the refill word live in fixed SGPR pairs (s[94:101]), whose halves the 32-bit
//     instructions can name; the decoded symbol goes to lane (row in the block) of `kept` by v_writelane with M0 as the lane
//     select (gfx9's constant bus takes one SGPR)
// Rows: set A holds the current row on entry and the first row of the next block on exit; the loop takes two rows a turn.
// The loop names PHYSICAL registers -- s94..s101 (allocatable on gfx950 up to s101; VCC / FLAT_SCRATCH / XNACK_MASK sit above), v96..v100,
// M0 through s_set_gpr_idx -- so it is tied to this target: any other one must not build it silently.  tools/check_asm_prefetch.py and
// tools/check_registers.py inspect the compiled kernel at build time; -DL3C_DECODE_ASM_LOOP=0 builds the decoder with the
// compiler-generated symbol (lean_symbol) everywhere, bit-identical and ~1.7x slower per symbol, as the fallback.
#ifndef L3C_DECODE_ASM_LOOP
#define L3C_DECODE_ASM_LOOP 1
#endif
#if L3C_DECODE_ASM_LOOP && defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx950__)
#error "ac_decode_lean_kernel's hand-written loop is written for gfx950 (physical SGPR / VGPR numbers); build with -DL3C_DECODE_ASM_LOOP=0 elsewhere"
#endif
#define L3C_ROW_READS(R0, R1, R2, R3, ADDR)                                                                        \
    "ds_read_u16_d16_hi " R0 ", " ADDR "\n\tds_read_u16_d16_hi " R1 ", " ADDR " offset:128\n\t"                     \
    "ds_read_u16_d16_hi " R2 ", " ADDR " offset:256\n\tds_read_u16_d16_hi " R3 ", " ADDR " offset:384\n\t"
#define L3C_ROW_READ1(R0, ADDR) "ds_read_u16_d16_hi " R0 ", " ADDR "\n\t"
// d = value - low; the scaled row; one compare per register into an SGPR pair of its own
#define L3C_SYMBOL_SCALE4(R0, R1, R2, R3)                                                                          \
    "s_add_u32 %[span], %[range], 1\n\t"                                                                            \
    "s_sub_u32 %[d], s97, %[low]\n\t"                                                                               \
    "v_mul_hi_u32 v96, " R0 ", %[span]\n\tv_mul_hi_u32 v97, " R1 ", %[span]\n\t"                                    \
    "v_mul_hi_u32 v98, " R2 ", %[span]\n\tv_mul_hi_u32 v99, " R3 ", %[span]\n\t"                                    \
    "v_cmp_ge_u32_e64 %[m0], %[d], v96\n\tv_cmp_ge_u32_e64 %[m1], %[d], v97\n\t"                                    \
    "v_cmp_ge_u32_e64 %[m2], %[d], v98\n\tv_cmp_ge_u32_e64 %[m3], %[d], v99\n\t"
#define L3C_SYMBOL_SCALE1(R0)                                                                                      \
    "s_add_u32 %[span], %[range], 1\n\t"                                                                            \
    "s_sub_u32 %[d], s97, %[low]\n\t"                                                                               \
    "v_mul_hi_u32 v96, " R0 ", %[span]\n\t"                                                                         \
    "v_cmp_ge_u32_e64 %[m0], %[d], v96\n\t"
// x1 = max(rank, 1), x = x1 - 1, lo = t[x], hi = t[x1].  Four registers: the one holding entry x (and the one holding x + 1; v100
// for x == top, replaced below) is picked by VGPR indexing
#define L3C_SYMBOL_RANK4(SET_M0_TO_ROW)                                                                            \
    "s_bcnt1_i32_b64 %[r0], %[m0]\n\ts_bcnt1_i32_b64 %[r1], %[m1]\n\t"                                              \
    "s_bcnt1_i32_b64 %[r2], %[m2]\n\ts_bcnt1_i32_b64 %[r3], %[m3]\n\t"                                              \
    "s_add_i32 %[r0], %[r0], %[r1]\n\ts_add_i32 %[r2], %[r2], %[r3]\n\ts_add_i32 %[r0], %[r0], %[r2]\n\t"           \
    "s_max_u32 %[x1], %[r0], 1\n\ts_add_i32 %[x], %[x1], -1\n\t"                                                    \
    "s_lshr_b32 %[r1], %[x], 6\n\ts_lshr_b32 %[r2], %[x1], 6\n\t"                                                   \
    "s_set_gpr_idx_on %[r1], 1\n\tv_mov_b32 %[sel], v96\n\t"                                                        \
    "s_set_gpr_idx_idx %[r2]\n\tv_mov_b32 %[sel1], v96\n\t"                                                         \
    "s_set_gpr_idx_off\n\t"                                                                                         \
    SET_M0_TO_ROW "\n\t"                                                                                            \
    "v_readlane_b32 %[lo], %[sel], %[x]\n\tv_readlane_b32 %[hi], %[sel1], %[x1]\n\t"
#define L3C_SYMBOL_RANK1(SET_M0_TO_ROW)                                                                            \
    "s_and_b64 %[m0], %[m0], %[valid]\n\t"                 /* lanes past the top symbol hold no table entries */      \
    "s_bcnt1_i32_b64 %[r0], %[m0]\n\t"                                                                              \
    "s_max_u32 %[x1], %[r0], 1\n\ts_add_i32 %[x], %[x1], -1\n\t"                                                    \
    SET_M0_TO_ROW "\n\t"                                                                                            \
    "v_readlane_b32 %[lo], v96, %[x]\n\tv_readlane_b32 %[hi], v96, %[x1]\n\t"
// the interval update, the renormalisation, the bit buffer (see the list above)
#define L3C_SYMBOL_ADVANCE(BETWEEN)                                                                                \
    "v_readlane_b32 %[w], %[cur], %[wrel]\n\t"             /* the next stream word, whether needed or not */         \
    "v_writelane_b32 %[kept], %[x], m0\n\t"                                                                         \
    "s_min_u32 %[minspan], %[minspan], %[span]\n\t"                                                                 \
    "s_sub_u32 %[t0], %[range], %[d]\n\t"                  /* SCC = d > range */                                    \
    "s_addc_u32 %[bad], %[bad], 0\n\t"                                                                              \
    BETWEEN                                                                                                         \
    "s_add_u32 %[lo], %[lo], %[low]\n\t"                   /* low' */                                               \
    "s_add_u32 %[hi], %[hi], %[low]\n\ts_sub_u32 %[hi], 0, %[hi]\n\t"      /* ~high' = -(low + t_hi) */             \
    "s_cmp_eq_u32 %[x], %[top]\n\ts_cselect_b32 %[hi], %[nh], %[hi]\n\t"   /* the top symbol keeps high */           \
    "s_and_b32 %[r0], %[lo], %[hi]\n\ts_xor_b32 %[r1], %[lo], %[hi]\n\ts_lshl_b32 %[r0], %[r0], 1\n\t"              \
    "s_nor_b32 %[r0], %[r0], %[r1]\n\ts_flbit_i32_b32 %[c], %[r0]\n\t"     /* c = n + m */                          \
    "s_lshl_b32 %[lo], %[lo], %[c]\n\ts_and_b32 %[r1], %[lo], 0x80000000\n\ts_xor_b32 %[low], %[lo], %[r1]\n\t"     \
    "s_lshl_b32 %[hi], %[hi], %[c]\n\ts_and_b32 %[nh], %[hi], 0x7fffffff\n\t"                                       \
    "s_add_u32 %[r2], %[low], %[nh]\n\ts_not_b32 %[range], %[r2]\n\t"                                               \
    "s_mov_b32 s96, s99\n\ts_lshl_b64 s[96:97], s[96:97], %[c]\n\ts_xor_b32 s97, s97, %[r1]\n\t"  /* value */        \
    "s_lshl_b64 s[98:99], s[98:99], %[c]\n\ts_sub_u32 %[nbits], %[nbits], %[c]\n\t"                                 \
    "s_cmp_lt_u32 %[nbits], 32\n\ts_cselect_b32 s100, %[w], 0\n\ts_cselect_b32 %[r3], 1, 0\n\t"                     \
    "s_sub_u32 %[r2], 32, %[nbits]\n\ts_lshl_b64 s[94:95], s[100:101], %[r2]\n\ts_or_b64 s[98:99], s[98:99], s[94:95]\n\t" \
    "s_lshl_b32 %[r2], %[r3], 5\n\ts_add_u32 %[nbits], %[nbits], %[r2]\n\ts_add_u32 %[wrel], %[wrel], %[r3]\n\t"
// PREFETCH: the reads of the NEXT row (other register set) and whatever else does not depend on this symbol, placed where
// the scalar unit waits for the compares.  BETWEEN: independent scalar work placed where it waits for the v_readlanes.
#define L3C_SYMBOL(R0, R1, R2, R3, PREFETCH, SET_M0_TO_ROW, BETWEEN) \
    L3C_SYMBOL_SCALE4(R0, R1, R2, R3) PREFETCH L3C_SYMBOL_RANK4(SET_M0_TO_ROW) L3C_SYMBOL_ADVANCE(BETWEEN)
#define L3C_SYMBOL1(R0, PREFETCH, SET_M0_TO_ROW, BETWEEN) \
    L3C_SYMBOL_SCALE1(R0) PREFETCH L3C_SYMBOL_RANK1(SET_M0_TO_ROW) L3C_SYMBOL_ADVANCE(BETWEEN)
// the row after the block's last lives in the next block
#define L3C_CROSS_SELECT \
    "s_add_u32 %[t0], %[j], 2\n\ts_cmp_eq_u32 %[t0], %[R]\n\ts_cselect_b64 vcc, -1, 0\n\tv_cndmask_b32 %[a2], %[addr], %[across], vcc\n\t"
#define L3C_BLOCK_STATE_OPERANDS                                                                                   \
    [kept] "+v"(kept), [addr] "+v"(addr_next), [low] "+s"(st.low), [nh] "+s"(st.nh), [range] "+s"(st.range),         \
    [nbits] "+s"(st.nbits), [wrel] "+s"(wrel), [bad] "+s"(st.bad), [minspan] "+s"(minspan), [j] "+s"(j), [value] "+s"(value), \
    [buf] "+s"(st.buf), [span] "=&s"(span), [d] "=&s"(d), [t0] "=&s"(t0), [r0] "=&s"(r0), [r1] "=&s"(r1), [r2] "=&s"(r2), \
    [r3] "=&s"(r3), [x] "=&s"(x), [x1] "=&s"(x1), [lo] "=&s"(lo), [hi] "=&s"(hi), [w] "=&s"(w), [c] "=&s"(c), [a2] "=&v"(a2)
#define L3C_BLOCK_CLOBBERS "s94", "s95", "s96", "s97", "s98", "s99", "s100", "s101", "m0", "scc", "vcc", "memory"

// WINDOW rows: the same symbol, plus the test for a MISS right behind the rank (csrc/dmll_core.h).  The rank sits at an edge of the window
// iff x1 = x' + 1 is 1 or 64, i.e. (x1 & 62) == 0: ONE s_and and a branch that is not taken (4.2 ns) on the common path.  At an edge the
// out-of-line handler decides with the row's window offset -- entry 64 of the row, read one symbol ahead by lane 0 (ds_read_u16 at the
// lane address + 128) and handed to the scalar unit at the start of the symbol --: x' = 0 is exact iff w0 == 0, x' = 63 iff w0 == 192
// (then back into the symbol); otherwise the loop is LEFT before the symbol has touched the state, with `hit` = 1 / 2 (first / second row
// of the pair), the prefetched row landed, `j` at the pair.  The caller decodes the missed symbol from the pixel's full row, finishes the
// pair with the compiled symbol and comes back for the rest of the block.
#define L3C_SYMBOL_RANK1W(SET_M0_TO_ROW, EDGE, BACK)                                                               \
    "s_and_b64 %[m0], %[m0], %[valid]\n\t"                                                                          \
    "s_bcnt1_i32_b64 %[r0], %[m0]\n\t"                                                                              \
    "s_max_u32 %[x1], %[r0], 1\n\ts_add_i32 %[x], %[x1], -1\n\t"                                                    \
    "s_and_b32 %[t0], %[x1], 62\n\t"                                                                                \
    "s_cbranch_scc0 " EDGE "\n"                                                                                     \
    BACK ":\n\t"                                                                                                    \
    SET_M0_TO_ROW "\n\t"                                                                                            \
    "v_readlane_b32 %[lo], v96, %[x]\n\tv_readlane_b32 %[hi], v96, %[x1]\n\t"
#define L3C_SYMBOL1W(R0, PREFETCH, SET_M0_TO_ROW, BETWEEN, EDGE, BACK) \
    L3C_SYMBOL_SCALE1(R0) PREFETCH L3C_SYMBOL_RANK1W(SET_M0_TO_ROW, EDGE, BACK) L3C_SYMBOL_ADVANCE(BETWEEN)
// x' at an edge: t0 = w0 (low edge) or 192 - w0 (high edge); 0 -> the symbol is exact, back; else leave with hit = N
#define L3C_EDGE_HANDLER(LABEL, BACK, N)                                                                           \
    LABEL ":\n\t"                                                                                                   \
    "s_sub_u32 %[t0], 192, %[wv]\n\t"                                                                               \
    "s_cmp_eq_u32 %[x1], 1\n\ts_cselect_b32 %[t0], %[wv], %[t0]\n\t"                                                \
    "s_cmp_eq_u32 %[t0], 0\n\ts_cbranch_scc1 " BACK "\n\t"                                                          \
    "s_mov_b32 %[hit], " N "\n\ts_waitcnt lgkmcnt(0)\n\ts_branch 9f\n"

#if L3C_DECODE_ASM_LOOP   // (the functions below name physical registers: not even parsed for the compiled-symbol fallback build)
__device__ __forceinline__ void lean_block_uniform(LeanState &st, uint32_t &wrel, uint32_t &minspan, uint32_t value);

// rows j0 (even; set A holds it, wA lane 0 its entry 64) .. R - 1 of a block of window rows; returns with hit != 0 and j at the pair of
// the missed row, or hit == 0 and j == R
__device__ __forceinline__ void lean_block_asm_window(RowHi<1> &A, RowHi<1> &B, uint32_t &wA, const ValidLanes<1> &valid, LeanState &st,
                                                      uint32_t &wrel, uint32_t &minspan, uint32_t window, int &kept, uint32_t &addr_next,
                                                      uint32_t addr_cross, uint32_t row_bytes, uint32_t R, uint32_t top, uint32_t &j,
                                                      uint32_t &hit) {
    uint64_t m0;
    uint32_t span, d, t0, r0, r1, r2, r3, x, x1, lo, hi, w, c, a2, wv, wB = 0;
    uint32_t value = (uint32_t)(st.vb >> 32);
    asm volatile(
        "s_mov_b32 s97, %[value]\n\ts_mov_b64 s[98:99], %[buf]\n\ts_mov_b32 s101, 0\n"
        "1:\n\t"
        "v_readfirstlane_b32 %[wv], %[wA]\n\t"
        L3C_SYMBOL1W("%[a0]", L3C_ROW_READ1("%[b0]", "%[addr]") "ds_read_u16 %[wB], %[addr] offset:128\n\tv_add_u32 %[addr], %[rowb], %[addr]\n\t",
                     "s_mov_b32 m0, %[j]", L3C_CROSS_SELECT, "2f", "3")
        "s_waitcnt lgkmcnt(0)\n\t"
        "v_readfirstlane_b32 %[wv], %[wB]\n\t"
        L3C_SYMBOL1W("%[b0]", L3C_ROW_READ1("%[a0]", "%[a2]") "ds_read_u16 %[wA], %[a2] offset:128\n\tv_add_u32 %[addr], %[rowb], %[addr]\n\t",
                     "s_add_u32 m0, %[j], 1", "s_add_u32 %[j], %[j], 2\n\t", "4f", "5")
        "s_waitcnt lgkmcnt(0)\n\t"
        "s_cmp_lt_u32 %[j], %[R]\n\ts_cbranch_scc1 1b\n\t"
        "s_branch 9f\n"
        L3C_EDGE_HANDLER("2", "3b", "1")
        L3C_EDGE_HANDLER("4", "5b", "2")
        "9:\n\t"
        "s_mov_b32 %[value], s97\n\ts_mov_b64 %[buf], s[98:99]"
        : [a0] "+v"(A.a), [b0] "+v"(B.a), [wA] "+v"(wA), [wB] "+v"(wB), [m0] "=&s"(m0), [wv] "=&s"(wv), [hit] "+s"(hit),
          [kept] "+v"(kept), [addr] "+v"(addr_next), [low] "+s"(st.low), [nh] "+s"(st.nh), [range] "+s"(st.range),
          [nbits] "+s"(st.nbits), [wrel] "+s"(wrel), [bad] "+s"(st.bad), [minspan] "+s"(minspan), [j] "+s"(j), [value] "+s"(value),
          [buf] "+s"(st.buf), [span] "=&s"(span), [d] "=&s"(d), [t0] "=&s"(t0), [r0] "=&s"(r0), [r1] "=&s"(r1), [r2] "=&s"(r2),
          [r3] "=&s"(r3), [x] "=&s"(x), [x1] "=&s"(x1), [lo] "=&s"(lo), [hi] "=&s"(hi), [w] "=&s"(w), [c] "=&s"(c), [a2] "=&v"(a2)
        : [cur] "v"(window), [across] "v"(addr_cross), [rowb] "s"(row_bytes), [R] "s"(R), [top] "s"(top), [valid] "s"(valid.a)
        : "v96", L3C_BLOCK_CLOBBERS);
    lean_block_uniform(st, wrel, minspan, value);
    j = (uint32_t)__builtin_amdgcn_readfirstlane((int)j);
    hit = (uint32_t)__builtin_amdgcn_readfirstlane((int)hit);
}

// what an asm statement returns counts as divergent: say that the state is wave-uniform (it already sits in SGPRs)
__device__ __forceinline__ void lean_block_uniform(LeanState &st, uint32_t &wrel, uint32_t &minspan, uint32_t value) {
    st.low = (uint32_t)__builtin_amdgcn_readfirstlane((int)st.low);
    st.nh = (uint32_t)__builtin_amdgcn_readfirstlane((int)st.nh);
    st.range = (uint32_t)__builtin_amdgcn_readfirstlane((int)st.range);
    st.nbits = (uint32_t)__builtin_amdgcn_readfirstlane((int)st.nbits);
    st.bad = (uint32_t)__builtin_amdgcn_readfirstlane((int)st.bad);
    wrel = (uint32_t)__builtin_amdgcn_readfirstlane((int)wrel);
    minspan = (uint32_t)__builtin_amdgcn_readfirstlane((int)minspan);
    value = (uint32_t)__builtin_amdgcn_readfirstlane((int)value);
    const uint32_t bl = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)st.buf);
    const uint32_t bh = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(st.buf >> 32));
    st.buf = ((uint64_t)bh << 32) | bl;
    st.vb = (uint64_t)value << 32;
}

// ONE row register (alphabets of up to 64 symbols: the bottleneck scales): the same loop, one compare, the valid-lane mask
__device__ __forceinline__ void lean_block_asm(RowHi<1> &A, RowHi<1> &B, const ValidLanes<1> &valid, LeanState &st, uint32_t &wrel,
                                               uint32_t &minspan, uint32_t window, int &kept, uint32_t addr_next,
                                               uint32_t addr_cross, uint32_t row_bytes, uint32_t R, uint32_t top) {
    uint64_t m0;
    uint32_t span, d, t0, r0, r1, r2, r3, x, x1, lo, hi, w, c, a2;
    uint32_t value = (uint32_t)(st.vb >> 32), j = 0;
    asm volatile(
        "s_mov_b32 s97, %[value]\n\ts_mov_b64 s[98:99], %[buf]\n\ts_mov_b32 s101, 0\n"
        "1:\n\t"
        L3C_SYMBOL1("%[a0]", L3C_ROW_READ1("%[b0]", "%[addr]") "v_add_u32 %[addr], %[rowb], %[addr]\n\t", "s_mov_b32 m0, %[j]",
                    L3C_CROSS_SELECT)
        "s_waitcnt lgkmcnt(0)\n\t"
        L3C_SYMBOL1("%[b0]", L3C_ROW_READ1("%[a0]", "%[a2]") "v_add_u32 %[addr], %[rowb], %[addr]\n\t", "s_add_u32 m0, %[j], 1",
                    "s_add_u32 %[j], %[j], 2\n\t")
        "s_waitcnt lgkmcnt(0)\n\t"
        "s_cmp_lt_u32 %[j], %[R]\n\ts_cbranch_scc1 1b\n\t"
        "s_mov_b32 %[value], s97\n\ts_mov_b64 %[buf], s[98:99]"
        : [a0] "+v"(A.a), [b0] "+v"(B.a), [m0] "=&s"(m0), L3C_BLOCK_STATE_OPERANDS
        : [cur] "v"(window), [across] "v"(addr_cross), [rowb] "s"(row_bytes), [R] "s"(R), [top] "s"(top), [valid] "s"(valid.a)
        : "v96", L3C_BLOCK_CLOBBERS);
    lean_block_uniform(st, wrel, minspan, value);
}

__device__ __forceinline__ void lean_block_asm(RowHi<4> &A, RowHi<4> &B, const ValidLanes<4> &, LeanState &st, uint32_t &wrel,
                                               uint32_t &minspan, uint32_t window, int &kept, uint32_t addr_next,
                                               uint32_t addr_cross, uint32_t row_bytes, uint32_t R, uint32_t top) {
    uint64_t m0, m1, m2, m3;
    uint32_t span, d, t0, r0, r1, r2, r3, x, x1, lo, hi, w, c, sel, sel1, a2;
    uint32_t value = (uint32_t)(st.vb >> 32), j = 0;
    asm volatile(
        "s_mov_b32 s97, %[value]\n\ts_mov_b64 s[98:99], %[buf]\n\ts_mov_b32 s101, 0\n"
        "1:\n\t"
        L3C_SYMBOL("%[a0]", "%[a1]", "%[a2_]", "%[a3]",
                   L3C_ROW_READS("%[b0]", "%[b1]", "%[b2]", "%[b3]", "%[addr]") "v_add_u32 %[addr], %[rowb], %[addr]\n\t",
                   "s_mov_b32 m0, %[j]", L3C_CROSS_SELECT)
        "s_waitcnt lgkmcnt(0)\n\t"
        L3C_SYMBOL("%[b0]", "%[b1]", "%[b2]", "%[b3]",
                   L3C_ROW_READS("%[a0]", "%[a1]", "%[a2_]", "%[a3]", "%[a2]") "v_add_u32 %[addr], %[rowb], %[addr]\n\t",
                   "s_add_u32 m0, %[j], 1", "s_add_u32 %[j], %[j], 2\n\t")
        "s_waitcnt lgkmcnt(0)\n\t"
        "s_cmp_lt_u32 %[j], %[R]\n\ts_cbranch_scc1 1b\n\t"
        "s_mov_b32 %[value], s97\n\ts_mov_b64 %[buf], s[98:99]"
        : [a0] "+v"(A.a), [a1] "+v"(A.b), [a2_] "+v"(A.c), [a3] "+v"(A.d), [b0] "+v"(B.a), [b1] "+v"(B.b), [b2] "+v"(B.c),
          [b3] "+v"(B.d), [m0] "=&s"(m0), [m1] "=&s"(m1), [m2] "=&s"(m2), [m3] "=&s"(m3), [sel] "=&v"(sel), [sel1] "=&v"(sel1),
          L3C_BLOCK_STATE_OPERANDS
        : [cur] "v"(window), [across] "v"(addr_cross), [rowb] "s"(row_bytes), [R] "s"(R), [top] "s"(top)
        : "v96", "v97", "v98", "v99", "v100", L3C_BLOCK_CLOBBERS);
    lean_block_uniform(st, wrel, minspan, value);
}
#endif   // L3C_DECODE_ASM_LOOP

// WINDOW (round 5): the instantiation for streams whose rows of this chunk are 65-entry window rows -- one row register, the bottleneck
// scales' loop.  The hand-written loop runs a block as if no symbol could miss; afterwards every lane compares ITS row's rank with its
// row's window offset (one LDS read per block): a real miss in the block -- the first one is always seen, the state before it is
// valid -- sends the block through the careful symbols below, which evaluate a missed pixel's full row in the wavefront.
template <int NJ, bool ALLVALID, int IPB_, bool WINDOW>
__device__ __forceinline__ void lean_decode_body(const DecodeArgs &a, uint8_t *ring) {
    static_assert(!WINDOW || (NJ == 1 && ALLVALID), "window rows fill one row register, all 64 lanes hold entries");
    using C = RingCfg<NJ, IPB_>;
    // ([measured, round 5] s_setprio 3 here -- a few hundred latency-bound wavefronts beside the table kernel's thousands -- changes nothing:
    // 0.415 vs 0.404 s per batch of 128)
    const uint16_t *cdf = a.cdf;
    const int64_t table_bytes = a.table_bytes;
    const bool has_win = a.win.stats_in != nullptr;
    if (WINDOW != (has_win && l3c::use_window(a.win.stats_in[blockIdx.x]))) return;   // not this kernel's kind of stream
    const int Lp = WINDOW ? l3c::kWinLp : a.Lp;
    const bool validated = a.flag ? (*a.flag == 0) : (a.monotone != 0);
    const int64_t s = blockIdx.x;
    const StreamView sv = stream_view(a, s);
    const uint32_t n_sym = sv.n_sym;
    const int lane = threadIdx.x;
    const uint32_t top = (uint32_t)(Lp - 2);
    const uint32_t row_bytes = (uint32_t)Lp * 2u;
    const uint32_t R = (uint32_t)ring_rows_per_block(Lp, C::BLOCK_BYTES) & ~1u;   // even: the loop below takes rows in pairs
    const uint32_t n_blocks = (n_sym + R - 1u) / R;
    const uint64_t tab0 = reinterpret_cast<uint64_t>(cdf);
    const uint64_t stream0 = sv.rows;   // this stream's first row (its full-size slot)
    const uint64_t last_granule = (tab0 + (uint64_t)table_bytes - 1u) & ~(uint64_t)15;
    int16_t *dst = sv.dst;
    if (!validated) {   // not a table for the fast path: leave the whole chunk to the generic pass
        if (lane == 0) dst[0] = (int16_t)-1;
        return;
    }
    const uint32_t ring_base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint8_t *)ring;

    auto request_block = [&](uint32_t k) {   // DMA the 16-byte granules holding rows [kR, (k+1)R) into slot k % NB
        const uint64_t a = (stream0 + (uint64_t)k * R * row_bytes) & ~(uint64_t)15;
        uint8_t *slot = ring + (k % C::NB) * C::BLOCK_BYTES;
        if (a + (uint64_t)(C::IPB * 1024 - 16) <= last_granule) {
            // the whole block lies inside the table (all but the last blocks of the last stream): one address per 4 KB, the
            // 1 KB steps as the instruction's immediate offset, which moves the global and the LDS address alike
#pragma unroll
            for (int q0 = 0; q0 < C::IPB; q0 += 4) {
                const uint64_t g = a + (uint64_t)(q0 * 1024 + lane * 16);
                const auto gp = (const __attribute__((address_space(1))) void *)g;
                const auto lp = (__attribute__((address_space(3))) void *)(slot + q0 * 1024);
                __builtin_amdgcn_global_load_lds(gp, lp, 16, 0, 0);
                if (q0 + 1 < C::IPB) __builtin_amdgcn_global_load_lds(gp, lp, 16, 1024, 0);
                if (q0 + 2 < C::IPB) __builtin_amdgcn_global_load_lds(gp, lp, 16, 2048, 0);
                if (q0 + 3 < C::IPB) __builtin_amdgcn_global_load_lds(gp, lp, 16, 3072, 0);
            }
            return;
        }
#pragma unroll
        for (int q = 0; q < C::IPB; ++q) {
            uint64_t g = a + (uint64_t)(q * 1024 + lane * 16);
            g = g <= last_granule ? g : last_granule;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)g,
                                             (__attribute__((address_space(3))) void *)(slot + q * 1024), 16, 0, 0);
        }
    };
    auto block_base = [&](uint32_t k) -> uint32_t {   // LDS byte address of entry 0 of row k * R
        const uint64_t b = stream0 + (uint64_t)k * R * row_bytes;
        return ring_base + (k % C::NB) * C::BLOCK_BYTES + (uint32_t)(b & 15u);
    };
    auto block_addr = [&](uint32_t k) -> uint32_t { return block_base(k) + (uint32_t)lane * 2u; };   // ... of this lane's first entry

    WaveBits src;
    const uint32_t *words = reinterpret_cast<const uint32_t *>(a.in + a.in_offsets[s]);
    LeanState st;
    {
        uint32_t pos0 = 0;
        uint32_t value = 0;
        st.low = 0;
        st.nh = 0;
        st.bad = 0;
        if (a.state_in) {
            const DecodeState in = a.state_in[s];
            pos0 = in.pos;
            st.low = in.low;
            st.nh = ~in.high;
            value = in.value;
        }
        src.init(words, a.in_nbytes[s], lane, ring_base + C::NB * C::BLOCK_BYTES, pos0);   // w01 = words (pos0 >> 5), + 1
        st.buf = src.w01 << (pos0 & 31u);
        st.nbits = 64u - (pos0 & 31u);
        st.widx = (pos0 >> 5) + 2u;
        if (!a.state_in) {   // value = the first 32 bits
            value = (uint32_t)(st.buf >> 32);
            st.buf <<= 32;
            st.nbits -= 32u;
        }
        st.vb = (uint64_t)value << 32;
        if (st.nbits < 32u) {
            st.buf |= (uint64_t)src.word(st.widx) << (32u - st.nbits);
            st.widx += 1u;
            st.nbits += 32u;
        }
        st.range = ~(st.low + st.nh);
    }
    const uint32_t no_advance = a.final_chunk ? n_sym - 1u : 0xFFFFFFFFu;   // torchac.cpp:335-337

#pragma unroll
    for (uint32_t k = 0; k < (uint32_t)C::NB - 1u; ++k)
        if (k < n_blocks) request_block(k);
    vm_wait<0>();

    RowHi<NJ> rowA, rowB;
    if constexpr (NJ == 1) {
        rowA = RowHi<1>{0u};
        rowB = RowHi<1>{0u};
    } else {
        rowA = RowHi<4>{0u, 0u, 0u, 0u};
        rowB = RowHi<4>{0u, 0u, 0u, 0u};
    }
    row_hi_issue(block_addr(0), rowA);
    row_hi_wait(rowA);

    ValidLanes<NJ> valid;
    if constexpr (NJ == 1) valid = valid_lanes1(lane, (int)top);
    else valid = valid_lanes4(lane, (int)top);
    int kept = 0;
    uint32_t misses = 0;   // WINDOW: symbols that fell outside their window; classic rows of a windowed part: symbols a window would have missed
    int miss = 0;          // WINDOW: direction of the careful symbol's miss (0: decoded)
    uint32_t w0_j = 0;     // WINDOW: the window offset of the row the careful symbol works on
    auto row_w0_addr = [&](uint32_t k, uint32_t j) -> uint32_t { return block_base(k) + j * row_bytes + 2u * (uint32_t)l3c::kWinTop + 2u; };   // entry 64
    // One careful symbol.  WINDOW: a miss leaves the state untouched (L3C_LEAN_FIXUP then decodes the symbol from the pixel's full row,
    // AFTER the wait for the prefetched row: no LDS read is in flight while that long path runs)
#define L3C_LEAN_SYMBOL(ROW, ADVANCE, J)                                                                                   \
    if constexpr (WINDOW) {  /* (the row's entry 64: its window offset; the read also lands the prefetched row early) */   \
        w0_j = (uint32_t)__builtin_amdgcn_readfirstlane((int)lds_read_u16_now(row_w0_addr(k, (uint32_t)(J))));              \
        miss = lean_symbol_window(ROW, w0_j, ADVANCE, st, src, x);                                                         \
    } else {                                                                                                               \
        lean_symbol<NJ, ALLVALID>(ROW, valid, top, ADVANCE, st, src, x);                                                   \
    }
#define L3C_LEAN_FIXUP(J, ADVANCE)                                                                                         \
    if constexpr (WINDOW) {                                                                                                \
        if (__builtin_expect(miss != 0, 0)) {                                                                              \
            /* further windows of the same pixel's row, 62 symbols on in the direction of the miss (so that the symbol next to the old */ \
            /* window is inside the new one), until the symbol is inside: the direction never turns, the alphabet's ends are exact */ \
            const LaneMixture lm = window_lane_mixture(sv.win, sv.img, sv.win.pix0 + (int64_t)(i0 + (J)), lane);           \
            int wb = (int)w0_j;                                                                                            \
            do {                                                                                                           \
                wb = miss > 0 ? (wb + 62 < l3c::kWinMaxOffset ? wb + 62 : l3c::kWinMaxOffset) : (wb > 62 ? wb - 62 : 0);    \
                bool mono;                                                                                                 \
                const RowHi<1> rw = window_row_at(sv.win, lm, wb, lane, mono);                                             \
                if (!mono) {                                                                                               \
                    st.bad = 1u;   /* a row the fast pass must not rank: the generic pass decodes this chunk */            \
                    break;                                                                                                 \
                }                                                                                                          \
                miss = lean_symbol_window(rw, (uint32_t)wb, ADVANCE, st, src, x);                                          \
            } while (miss != 0);                                                                                           \
            misses += 1u;                                                                                                  \
        }                                                                                                                  \
        x -= w0_j;   /* WINDOW: `kept` holds window-relative values (as the hand-written loop leaves them) until the block ends */ \
    }                                                                                                                      \
    kept = lane == (int)(J) ? (int)x : kept;
    for (uint32_t k = 0; k < n_blocks; ++k) {
        // Request block k + NB - 1 (its slot held block k - 1, fully consumed), then make sure block k + 1 -- whose first row
        // is prefetched at the end of this block -- has landed: only the two newest requests may stay in flight.
        if (k + C::NB - 1u < n_blocks) {
            request_block(k + C::NB - 1u);
            if (k > 0) vm_wait<2 * C::IPB>();   // k == 0: blocks 0 .. NB-2 were waited for above
        } else {
            vm_wait<0>();
        }
        src.age += 1u;
        const uint32_t i0 = k * R;
        const uint32_t rows = n_sym - i0 < R ? n_sym - i0 : R;
        const uint32_t addr_cross = block_addr(k + 1u);
        uint32_t addr_next = block_addr(k) + row_bytes;
        uint32_t x = 0;
        const bool full_block = k + 1u < n_blocks;   // a full block that does not hold the stream's last symbol
        bool done = false;
#if L3C_DECODE_ASM_LOOP
        if constexpr (L3C_DECODE_ASM_LOOP && (NJ == 1 || ALLVALID)) {
            // the hand-written loop, unless the bit window could run out inside the block (a symbol takes at most one word)
            if (full_block) {
                // the bit window: a block takes at most R <= 32 words (one a symbol).  Window indices stay below 64 if the block
                // starts below 32; otherwise it reads a window that starts 32 words later (late_window)
                if (st.widx - src.base >= 64u) src.next_window();
                uint32_t wrel = st.widx - src.base, wofs = 0u, window = src.cur;
                if (wrel >= 32u) {
                    window = src.late_window();
                    wofs = 32u;
                    wrel -= 32u;
                }
                const LeanState saved = st;
                const uint32_t saved_misses = misses;   // (a block that is decoded again must not count its misses twice)
                uint32_t minspan = 0xFFFFFFFFu;
                if constexpr (WINDOW) {
                    // rows in pairs through the hand-written loop; it leaves at a MISS (hit = 1 / 2: first / second row of pair j), with the
                    // state as before that symbol: the pair is finished here -- the missed symbol from the pixel's full row -- and the loop
                    // re-entered for the rest of the block
                    uint32_t j = 0u, hit = 0u;
                    uint32_t addr = addr_next;
                    done = true;
                    while (true) {
                        uint32_t wA = lds_read_u16_now(row_w0_addr(k, j));
                        hit = 0u;
                        lean_block_asm_window(rowA, rowB, wA, valid, st, wrel, minspan, window, kept, addr, addr_cross, row_bytes, R, top, j, hit);
                        st.widx = src.base + wofs + wrel;
                        if (__builtin_expect(minspan == 0u || st.bad != 0u, 0)) {   // the whole 32-bit range / a bad value: the block again, the careful way
                            done = false;
                            break;
                        }
                        if (hit == 0u) break;
                        if (hit == 1u) {   // A: row j (the miss), B: row j + 1 (landed), addr: row j + 2
                            L3C_LEAN_SYMBOL(rowA, true, j)
                            L3C_LEAN_FIXUP(j, true)
                            row_hi_issue(j + 2u == R ? addr_cross : addr, rowA);
                            addr += row_bytes;
                            L3C_LEAN_SYMBOL(rowB, true, j + 1u)
                            row_hi_wait(rowA);
                            L3C_LEAN_FIXUP(j + 1u, true)
                        } else {           // B: row j + 1 (the miss), A: row j + 2 (landed), addr: row j + 3
                            L3C_LEAN_SYMBOL(rowB, true, j + 1u)
                            L3C_LEAN_FIXUP(j + 1u, true)
                        }
                        j += 2u;
                        if (__builtin_expect(st.bad != 0u, 0)) {
                            done = false;
                            break;
                        }
                        if (j >= R) break;
                        // the bit window for the rest of the block (<= R - j words from here), as at the block's start
                        if (st.widx - src.base >= 64u) src.next_window();
                        wrel = st.widx - src.base;
                        wofs = 0u;
                        window = src.cur;
                        if (wrel >= 32u) {
                            window = src.late_window();
                            wofs = 32u;
                            wrel -= 32u;
                        }
                    }
                } else {
                    lean_block_asm(rowA, rowB, valid, st, wrel, minspan, window, kept, addr_next, addr_cross, row_bytes, R, top);
                    st.widx = src.base + wofs + wrel;
                    done = minspan != 0u && st.bad == 0u;
                }
                if (__builtin_expect(!done, 0)) {   // a symbol met the whole 32-bit range or a bad value: again, the careful way
                    st = saved;
                    misses = saved_misses;
                    row_hi_issue(block_addr(k), rowA);
                    row_hi_wait(rowA);
                }
            }
        }
#endif
        if (done) {
        } else if (full_block) {   // rows in pairs, A then B
            for (uint32_t j = 0; j < R; j += 2u) {
                row_hi_issue(addr_next, rowB);
                addr_next += row_bytes;
                L3C_LEAN_SYMBOL(rowA, true, j)
                row_hi_wait(rowB);
                L3C_LEAN_FIXUP(j, true)
                row_hi_issue(j + 2u == R ? addr_cross : addr_next, rowA);   // the row after the block's last lives in block k + 1
                addr_next += row_bytes;
                L3C_LEAN_SYMBOL(rowB, true, j + 1u)
                row_hi_wait(rowA);
                L3C_LEAN_FIXUP(j + 1u, true)
            }
        } else {                   // the last block: ragged, and its last symbol may not advance the state
            for (uint32_t j = 0; j < rows; ++j) {
                const bool advance = i0 + j != no_advance;
                if (j & 1u) {
                    row_hi_issue(addr_next, rowA);   // past the stream's last row: never used
                    L3C_LEAN_SYMBOL(rowB, advance, j)
                    row_hi_wait(rowA);
                } else {
                    row_hi_issue(addr_next, rowB);
                    L3C_LEAN_SYMBOL(rowA, advance, j)
                    row_hi_wait(rowB);
                }
                addr_next += row_bytes;
                L3C_LEAN_FIXUP(j, advance)
            }
        }
        if (__builtin_expect(st.bad != 0u, 0)) {
            // a stream that leaves the fast path: mark it for the generic pass and stop (same lane, after any block store: the
            // last write to dst[0])
            vm_wait<0>();
            if (lane == 0) dst[0] = (int16_t)-1;
            return;
        }
        if constexpr (WINDOW) {   // `kept` is window-relative on every path: + the row's offset (entry 64 of row `lane`)
            const uint32_t r_l = (uint32_t)lane < rows ? (uint32_t)lane : rows - 1u;
            kept += (int)lds_read_u16_now(row_w0_addr(k, r_l));
        }
        if constexpr (!WINDOW && NJ == 4) {
            // full rows of a windowed part: what would a window have missed?  The (never decoded) entry Lp - 1 of row j carries its offset
            // (l3c_dmll_cdf_table); block k is still in its ring slot
            // -- sampled on every fourth block (x 4): an estimate is all the decision needs, and a block here is only 4 .. 16 symbols
            if (has_win && (k & 3u) == 0u) {
                const uint32_t r_l = (uint32_t)lane < rows ? (uint32_t)lane : rows - 1u;
                const uint32_t w0_l = lds_read_u16_now(block_base(k) + r_l * row_bytes + (uint32_t)(Lp - 1) * 2u);
                const bool m_l = (uint32_t)lane < rows && l3c::window_would_miss((uint32_t)kept & 0xFFFFu, w0_l);
                misses += 4u * (uint32_t)__popcll(__builtin_amdgcn_ballot_w64(m_l));
            }
        }
        if ((uint32_t)lane < rows) dst[i0 + (uint32_t)lane] = (int16_t)kept;
    }
#undef L3C_LEAN_SYMBOL
#undef L3C_LEAN_FIXUP
    if (a.state_out && lane == 0)
        a.state_out[s] = DecodeState{st.low, ~st.nh, (uint32_t)(st.vb >> 32), st.widx * 32u - st.nbits, {0u, 0u, 0u, 0u}};
    if (has_win && a.win.stats_out && lane == 0) a.win.stats_out[s] = l3c::window_stat(misses, n_sym, WINDOW ? 64u : 128u);
}

// WITH_WINDOW: see ac_decode_ring_kernel -- window-row streams and full-row streams of a launch side by side in one kernel
template <int NJ, bool ALLVALID, int IPB_ = (NJ == 1 ? 3 : 9), bool WITH_WINDOW = false>
__global__ __launch_bounds__(64) void ac_decode_lean_kernel(const DecodeArgsPack pack) {
    using C = RingCfg<NJ, IPB_>;
    static_assert(!WITH_WINDOW || (NJ == 4 && ALLVALID && IPB_ >= 3), "window rows: parts of the 256-symbol alphabet; the window body's ring must fit");
    __shared__ __attribute__((aligned(16))) uint8_t ring[C::NB * C::BLOCK_BYTES + 512];   // + the bit reader's windows
    const DecodeArgs &a = pack.part[blockIdx.y];
    if ((int64_t)blockIdx.x >= a.n_streams) return;
    if constexpr (WITH_WINDOW) {
        if (a.win.stats_in != nullptr && l3c::use_window(a.win.stats_in[blockIdx.x])) {
            lean_decode_body<1, true, 3, true>(a, ring);
            return;
        }
    }
    lean_decode_body<NJ, ALLVALID, IPB_, false>(a, ring);
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

// All parts must be of one alphabet class (Lp <= 65 or larger: the kernel instantiation) and agree on `fast_pass`.
int launch_ring_decode(DecodeArgsPack pack, int n_parts, bool fast_pass, hipStream_t st) {
    int64_t max_streams = 0;
    for (int i = 0; i < n_parts; ++i) max_streams = pack.part[i].n_streams > max_streams ? pack.part[i].n_streams : max_streams;
    const dim3 grid((unsigned)max_streams, (unsigned)n_parts), block(64);
    const bool small = pack.part[0].Lp - 1 <= 64;
    // wide alphabet: the small ring from 48 streams per launch on (a batch of 16 images: the decoders then share the machine with the
    // table kernel of the next chunk step); the result does not depend on the ring size
    const bool crowd = max_streams * n_parts >= 48;
    bool any_win = false;   // some part's table may hold window rows: their streams are decoded by the <1, ..., WINDOW> instantiations
    for (int i = 0; i < n_parts; ++i) any_win = any_win || pack.part[i].win.stats_in != nullptr;
    if (fast_pass) {   // streams that leave the fast path (or all, if the table is not validated) mark themselves
        bool full = true;   // every part codes the 256-symbol alphabet: all entries of the four row registers are table entries
        for (int i = 0; i < n_parts; ++i) full = full && pack.part[i].Lp == 257;
        if (any_win) {   // (l3c_ac_decode_chunks has checked: every windowed part codes the 256-symbol alphabet)
            if (crowd) hipLaunchKernelGGL((ac_decode_lean_kernel<4, true, 3, true>), grid, block, 0, st, pack);
            else hipLaunchKernelGGL((ac_decode_lean_kernel<4, true, 9, true>), grid, block, 0, st, pack);
        } else
        if (small) hipLaunchKernelGGL((ac_decode_lean_kernel<1, false>), grid, block, 0, st, pack);
        else if (crowd && full) hipLaunchKernelGGL((ac_decode_lean_kernel<4, true, 3>), grid, block, 0, st, pack);
        else if (crowd) hipLaunchKernelGGL((ac_decode_lean_kernel<4, false, 3>), grid, block, 0, st, pack);
        else if (full) hipLaunchKernelGGL((ac_decode_lean_kernel<4, true>), grid, block, 0, st, pack);
        else hipLaunchKernelGGL((ac_decode_lean_kernel<4, false>), grid, block, 0, st, pack);
        const int rc = l3c::check_launch("ac_decode_lean_kernel");
        if (rc != L3C_OK) return rc;
    }
    for (int i = 0; i < n_parts; ++i) pack.part[i].force = fast_pass ? 0 : 1;
    if (any_win) {
        if (crowd) hipLaunchKernelGGL((ac_decode_ring_kernel<4, 3, true>), grid, block, 0, st, pack);
        else hipLaunchKernelGGL((ac_decode_ring_kernel<4, 9, true>), grid, block, 0, st, pack);
    } else if (small) hipLaunchKernelGGL((ac_decode_ring_kernel<1>), grid, block, 0, st, pack);
    else if (crowd) hipLaunchKernelGGL((ac_decode_ring_kernel<4, 3>), grid, block, 0, st, pack);
    else hipLaunchKernelGGL((ac_decode_ring_kernel<4>), grid, block, 0, st, pack);
    return l3c::check_launch("ac_decode_ring_kernel<generic>");
}

}  // namespace

extern "C" {

int64_t l3c_interval_words(int64_t n_streams, int64_t n_sym) {
    return ((n_sym + kChunk - 1) / kChunk) * n_streams * kChunk * 2;
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

int64_t l3c_ac_encode_workspace_bytes(int64_t n_streams) { return ((n_streams * 4 + 255) / 256) * 256; }

int l3c_ac_encode(uint32_t *intervals, int64_t n_streams, int64_t n_sym, uint8_t *out, int64_t out_stride_bytes,
                  uint32_t *out_nbytes, void *workspace, l3c_stream_t stream) {
    L3C_REQUIRE(intervals && out && out_nbytes && workspace, "null pointer");
    L3C_REQUIRE(n_streams > 0 && n_sym > 0 && n_streams < (1ll << 31), "empty input");
    L3C_REQUIRE(out_stride_bytes % 4 == 0 && out_stride_bytes >= l3c_ac_max_bytes(n_sym), "output stride too small");
    L3C_REQUIRE((reinterpret_cast<uintptr_t>(out) & 3) == 0 && (reinterpret_cast<uintptr_t>(intervals) & 15) == 0 &&
                    (reinterpret_cast<uintptr_t>(workspace) & 3) == 0, "misaligned buffer");
    uint32_t *final_low = static_cast<uint32_t *>(workspace);
    const int blocks = (int)((n_streams + 31) / 32);   // a lane pair per stream
    hipLaunchKernelGGL(ac_state_kernel, dim3(blocks), dim3(64), 0, l3c::as_stream(stream), intervals, n_streams, n_sym,
                       final_low);
    int rc = l3c::check_launch("ac_state_kernel");
    if (rc != L3C_OK) return rc;
    hipLaunchKernelGGL(ac_pack_kernel, dim3((unsigned)n_streams), dim3(kPackThreads), 0, l3c::as_stream(stream), intervals,
                       n_streams, n_sym, final_low, out, out_stride_bytes, out_nbytes);
    return l3c::check_launch("ac_pack_kernel");
}

int64_t l3c_ac_encode_groups_workspace_bytes(int n_groups, int64_t total_streams) {
    return (((int64_t)n_groups * (int64_t)sizeof(AcGroup) + 255) / 256) * 256 + ((total_streams * 4 + 255) / 256) * 256;
}

int l3c_ac_encode_groups(const l3c_ac_group *groups_host, int n_groups, void *workspace, l3c_stream_t stream) {
    L3C_REQUIRE(groups_host && workspace && n_groups > 0 && n_groups <= 65536, "bad arguments");
    L3C_REQUIRE((reinterpret_cast<uintptr_t>(workspace) & 15) == 0, "misaligned workspace");
    std::vector<AcGroup> host((size_t)n_groups);
    int64_t state_blocks = 0, pack_blocks = 0, streams = 0;
    const size_t desc_bytes = (((size_t)n_groups * sizeof(AcGroup) + 255) / 256) * 256;
    uint32_t *final_low = reinterpret_cast<uint32_t *>(static_cast<char *>(workspace) + desc_bytes);
    for (int g = 0; g < n_groups; ++g) {
        const l3c_ac_group &in = groups_host[g];
        L3C_REQUIRE(in.intervals && in.out && in.out_nbytes, "null pointer in group");
        L3C_REQUIRE(in.n_streams > 0 && in.n_sym > 0, "empty group");
        L3C_REQUIRE(in.out_stride_bytes % 4 == 0 && in.out_stride_bytes >= l3c_ac_max_bytes(in.n_sym), "output stride too small");
        L3C_REQUIRE((reinterpret_cast<uintptr_t>(in.out) & 3) == 0 && (reinterpret_cast<uintptr_t>(in.intervals) & 15) == 0,
                    "misaligned buffer in group");
        host[g] = AcGroup{in.intervals, in.out, in.out_nbytes, final_low + streams, in.n_streams, in.n_sym, in.out_stride_bytes};
        state_blocks += (in.n_streams + 31) / 32;
        pack_blocks += in.n_streams;
        streams += in.n_streams;
    }
    L3C_REQUIRE(pack_blocks < (1ll << 31), "too many streams");
    // The descriptors travel as by-value kernel arguments (copied at launch time, so nothing here has to outlive the
    // call and no host-side synchronisation is needed), 64 per launch.
    AcGroup *dev = static_cast<AcGroup *>(workspace);
    int rc;
    for (int g0 = 0; g0 < n_groups; g0 += AcGroupPack::N) {
        AcGroupPack pack;
        const int n = n_groups - g0 < AcGroupPack::N ? n_groups - g0 : AcGroupPack::N;
        for (int i = 0; i < n; ++i) pack.g[i] = host[(size_t)(g0 + i)];
        hipLaunchKernelGGL(ac_groups_upload_kernel, dim3(1), dim3(64), 0, l3c::as_stream(stream), pack, dev + g0, n);
        rc = l3c::check_launch("ac_groups_upload_kernel");
        if (rc != L3C_OK) return rc;
    }
    hipLaunchKernelGGL(ac_state_groups_kernel, dim3((unsigned)state_blocks), dim3(64), 0, l3c::as_stream(stream), dev, n_groups);
    rc = l3c::check_launch("ac_state_groups_kernel");
    if (rc != L3C_OK) return rc;
    hipLaunchKernelGGL(ac_pack_groups_kernel, dim3((unsigned)pack_blocks), dim3(kPackThreads), 0, l3c::as_stream(stream), dev, n_groups);
    return l3c::check_launch("ac_pack_groups_kernel");
}

int l3c_ac_decode(const uint16_t *cdf, int64_t row_stride, int Lp, const uint8_t *in, const int64_t *in_offsets,
                  const uint32_t *in_nbytes, int64_t n_streams, int64_t n_sym, int monotone, int16_t *sym_out,
                  l3c_stream_t stream) {
    L3C_REQUIRE(cdf && in && in_offsets && in_nbytes && sym_out, "null pointer");
    L3C_REQUIRE(Lp >= 2 && Lp <= 257, "Lp out of range (2..257)");
    L3C_REQUIRE(row_stride == 0 || row_stride == Lp, "row_stride must be 0 or Lp");
    L3C_REQUIRE(n_streams > 0 && n_sym > 0, "empty input");
    L3C_REQUIRE((reinterpret_cast<uintptr_t>(in) & 3) == 0, "input must be 4-byte aligned (and every offset a multiple of 4)");
    L3C_REQUIRE(n_sym < (1ll << 31), "n_sym out of range");
    L3C_REQUIRE((reinterpret_cast<uintptr_t>(cdf) & 1) == 0, "table must be 2-byte aligned");
    const dim3 grid((unsigned)n_streams), block(64);
    const hipStream_t st = l3c::as_stream(stream);
    const uint32_t n = (uint32_t)n_sym;
    if (row_stride == 0) {
        if (Lp - 1 <= 64)
            hipLaunchKernelGGL(ac_decode_const_row_kernel<1>, grid, block, 0, st, cdf, Lp, in, in_offsets, in_nbytes, n, monotone, sym_out);
        else
            hipLaunchKernelGGL(ac_decode_const_row_kernel<4>, grid, block, 0, st, cdf, Lp, in, in_offsets, in_nbytes, n, monotone, sym_out);
        return l3c::check_launch("ac_decode_const_row_kernel");
    }
    DecodeArgs a{};
    a.cdf = cdf;
    a.Lp = Lp;
    a.table_bytes = n_streams * n_sym * (int64_t)Lp * 2;
    a.in = in;
    a.in_offsets = in_offsets;
    a.in_nbytes = in_nbytes;
    a.n_sym = n;
    a.monotone = monotone;
    a.final_chunk = 1;
    a.sym_out = sym_out;
    a.sym_stride = n_sym;
    a.n_streams = n_streams;
    DecodeArgsPack pack{};
    pack.part[0] = a;
    return launch_ring_decode(pack, 1, /*fast_pass=*/monotone != 0, st);
}

int64_t l3c_ac_decode_state_bytes(void) { return (int64_t)sizeof(DecodeState); }

int l3c_ac_decode_chunks(const l3c_ac_decode_part *parts, int n_parts, l3c_stream_t stream) {
    L3C_REQUIRE(parts && n_parts > 0 && n_parts <= DecodeArgsPack::N, "1..8 parts per call");
    DecodeArgsPack pack{};
    for (int i = 0; i < n_parts; ++i) {
        const l3c_ac_decode_part &q = parts[i];
        L3C_REQUIRE(q.cdf && q.in && q.in_offsets && q.in_nbytes && q.sym_out, "null pointer in part");
        L3C_REQUIRE(q.Lp >= 2 && q.Lp <= 257, "Lp out of range (2..257)");
        L3C_REQUIRE((q.Lp - 1 <= 64) == (parts[0].Lp - 1 <= 64), "parts of one call must share the alphabet class (Lp <= 65 or not)");
        L3C_REQUIRE((q.not_monotone_flag != nullptr) == (parts[0].not_monotone_flag != nullptr), "parts must agree on having a validity flag");
        const bool ragged = q.r_npix != nullptr;
        L3C_REQUIRE(q.n_streams > 0 && q.n_sym > 0 && q.n_sym < (1ll << 31), "bad shape");
        L3C_REQUIRE(ragged || (q.sym_offset >= 0 && q.sym_stride >= q.n_sym), "bad output layout");
        L3C_REQUIRE(!ragged || (q.r_table_off && q.r_pixbase && q.r_hw && q.r_pix0 && q.r_C > 0 && q.r_c >= 0 && q.r_c < q.r_C && q.r_table_bytes > 0),
                    "ragged part: every per-stream array, the plane count and the table size are needed");
        L3C_REQUIRE((reinterpret_cast<uintptr_t>(q.in) & 3) == 0 && (reinterpret_cast<uintptr_t>(q.cdf) & 1) == 0, "misaligned input");
        L3C_REQUIRE(((reinterpret_cast<uintptr_t>(q.state_in) | reinterpret_cast<uintptr_t>(q.state_out)) & 7) == 0, "misaligned state");
        L3C_REQUIRE(q.state_in != q.state_out || !q.state_in, "state_in and state_out must differ (a marked stream is decoded twice)");
        DecodeArgs &a = pack.part[i];
        a.cdf = q.cdf;
        a.Lp = q.Lp;
        a.table_bytes = ragged ? q.r_table_bytes : q.n_streams * q.n_sym * (int64_t)q.Lp * 2;
        a.r_npix = q.r_npix;  a.r_table_off = q.r_table_off;  a.r_pixbase = q.r_pixbase;  a.r_hw = q.r_hw;  a.r_pix0 = q.r_pix0;
        a.r_C = q.r_C;  a.r_c = q.r_c;
        a.in = q.in;
        a.in_offsets = q.in_offsets;
        a.in_nbytes = q.in_nbytes;
        a.n_streams = q.n_streams;
        a.n_sym = (uint32_t)q.n_sym;
        a.monotone = 0;
        a.flag = q.not_monotone_flag;
        a.final_chunk = q.final_chunk;
        a.state_in = static_cast<const DecodeState *>(q.state_in);
        a.state_out = static_cast<DecodeState *>(q.state_out);
        a.sym_out = q.sym_out;
        a.sym_stride = q.sym_stride;
        a.sym_offset = q.sym_offset;
        if (q.window_stats_in) {
            L3C_REQUIRE(q.Lp == 257 && q.C == 3 && q.K > 0 && q.K <= 16 && q.c >= 0 && q.c < 3, "window rows: RGB scale only (Lp 257, C 3, K <= 16)");
            L3C_REQUIRE(q.P && q.targets && (q.c == 0 || q.sym_all), "window rows: P, targets and the decoded channels are needed to evaluate a missed row");
            L3C_REQUIRE(q.not_monotone_flag, "window rows: the table's validity flag is required");
            L3C_REQUIRE(ragged || (q.HW > 0 && q.pix0 >= 0 && q.pix0 + q.n_sym <= q.HW), "window rows: chunk outside the image");
            a.win = WindowCtx{q.window_stats_in, q.window_stats_out, q.P, q.sym_all, q.targets, q.HW, q.pix0, q.C, q.K, q.c};
        }
    }
    for (int i = 0; i < n_parts; ++i)   // one kernel decodes window-row and full-row streams side by side: its full-row body is the 256-symbol one
        for (int j = 0; j < n_parts; ++j)
            L3C_REQUIRE(!parts[i].window_stats_in || parts[j].Lp == 257, "a call with window rows: every part must code the 256-symbol alphabet (Lp 257)");
    return launch_ring_decode(pack, n_parts, /*fast_pass=*/parts[0].not_monotone_flag != nullptr, l3c::as_stream(stream));
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
