// decode_pipeline.hip -- the chunk-pipelined decode of the RGB scale, issued by ONE host call (round 6).
//
// Reference: the decoder's per-channel loop (src/bitcoding/bitcoding.py:212-266: for every channel, build the CDF from the channels
// decoded so far, then torchac.decode_logistic_mixture -> torchac.cpp:299-381).  Channel c's means depend on the decoded values of the
// channels < c AT THE SAME PIXEL only (src/criterion/logistic_mixture.py:262-272), so R, G and B are three serial chains that only have
// to stay a chunk of pixels apart: pipeline step t handles chunk t - D c of channel c -- its table rows are built straight from P and the
// symbols decoded so far (ONE grouped table launch, l3c_dmll_cdf_table_parts), then ONE grouped decoder launch (l3c_ac_decode_chunks)
// resumes the range decoders of all active channels side by side.
//
// Until round 5 the host layer (bitcoding.py: _decode_rgb_pipelined) issued these ~150 launches one by one through ctypes, allocating
// a table tensor per launch: 0.24 s of host time for a 0.35 s decode of 128 images -- within 30 % of being bound by the host.  Here the
// whole schedule is a C loop over a caller-owned workspace: a few microseconds per launch, no allocation, no host synchronisation.
//
//   D = 1: everything on `main`.
//   D = 2: the channels stay TWO chunks apart, so the tables of step t + 1 need only the symbols of step t - 1 and are built on `main`
//          WHILE `side` decodes step t: max(table, decode) per step instead of their sum.  Two table slot sets (step parity): the tables
//          of step t are written after `main` has waited for the decode of step t - 2, the last reader of that slot set.
#include "l3c_common.h"

#include <string.h>

namespace {

constexpr int C3 = 3;
constexpr int LP = 257;
constexpr int64_t ALIGN = 256;

inline int64_t up(int64_t n) { return (n + ALIGN - 1) / ALIGN * ALIGN; }

struct Layout {
    int64_t flags, states, stats, scratch, tables, table_bytes, state_bytes, total;
};

// rectangular: B images, max_npix pixels in the longest chunk; ragged (streams > 0): B = 1, max_npix = the largest chunk's TOTAL over the images
Layout layout(int64_t B, int64_t max_npix, int n_chunks, int lag, int64_t streams = 0) {
    Layout l{};
    const int64_t S = streams > 0 ? streams : B;
    l.state_bytes = up(S * l3c_ac_decode_state_bytes());
    l.table_bytes = up(B * max_npix * LP * 2);
    int64_t p = 0;
    l.flags = p;    p += up(C3 * 4);
    l.states = p;   p += C3 * 2 * l.state_bytes;
    l.stats = p;    p += up((int64_t)C3 * (n_chunks + 2) * S * 4);
    l.scratch = p;  p += up(S * 4);
    l.tables = p;   p += (lag == 2 ? 2 : 1) * C3 * l.table_bytes;
    l.total = p;
    return l;
}

// A handful of events per device and host thread, created on first use and reused: an event may be recorded again as soon as every wait
// on its previous record has been ISSUED (hipStreamWaitEvent captures the record it finds), which the schedule below guarantees.
struct EventPool {
    static constexpr int N = 4;
    hipEvent_t ev[N] = {};
    bool made = false;
};
thread_local EventPool g_pool[64];

int events_for_current_device(EventPool **out) {
    int dev = 0;
    int rc = l3c::check_hip(hipGetDevice(&dev), "hipGetDevice");
    if (rc != L3C_OK) return rc;
    if (dev < 0 || dev >= 64) return l3c::fail(L3C_ERR_INVALID_ARG, "%s", "device index out of range");
    EventPool &p = g_pool[dev];
    if (!p.made) {
        for (int i = 0; i < EventPool::N; ++i) {
            rc = l3c::check_hip(hipEventCreateWithFlags(&p.ev[i], hipEventDisableTiming), "hipEventCreateWithFlags");
            if (rc != L3C_OK) return rc;
        }
        p.made = true;
    }
    *out = &p;
    return L3C_OK;
}

}  // namespace

extern "C" {

int64_t l3c_decode_rgb_workspace_bytes(int64_t B, int64_t max_chunk_npix, int n_chunks, int lag) {
    if (B <= 0 || max_chunk_npix <= 0 || n_chunks <= 0 || (lag != 1 && lag != 2)) return -1;
    return layout(B, max_chunk_npix, n_chunks, lag).total;
}

int64_t l3c_decode_rgb_stats_offset(int64_t B, int64_t max_chunk_npix, int n_chunks, int lag) {
    if (B <= 0 || max_chunk_npix <= 0 || n_chunks <= 0 || (lag != 1 && lag != 2)) return -1;
    return layout(B, max_chunk_npix, n_chunks, lag).stats;
}
}

namespace {

// one schedule for both forms: `rag` == nullptr -> the rectangular batch of `d`
struct Ragged {
    const int64_t *hw_host;        // [B]
    const int64_t *pix0_host;      // [n_chunks][B]
    const int64_t *npix_host;      // [n_chunks][B]
    const int64_t *tables_dev;     // pixbase [B] | hw [B] | pix0 [n_chunks][B] | npix [n_chunks][B] | table_off [n_chunks][B]
};

int decode_rgb_impl(const l3c_rgb_decode_desc *d, const Ragged *rag, int64_t max_npix, l3c_stream_t main_stream, l3c_stream_t side_stream) {
    const Layout l = layout(rag ? 1 : d->B, max_npix, d->n_chunks, d->lag, rag ? d->B : 0);
    L3C_REQUIRE(d->workspace_bytes >= l.total, "workspace too small (l3c_decode_rgb(_ragged)_workspace_bytes)");
    const hipStream_t main = l3c::as_stream(main_stream), side = d->lag == 2 ? l3c::as_stream(side_stream) : main;
    EventPool *pool = nullptr;
    int rc = events_for_current_device(&pool);
    if (rc != L3C_OK) return rc;
    hipEvent_t ev_tables = pool->ev[0];                 // main -> side: the tables of a step (waited for right after its record)
    hipEvent_t *ev_decoded = &pool->ev[1];              // side -> main: step t's symbols, ring of three (waited for at step t + 2)

    uint8_t *ws = static_cast<uint8_t *>(d->workspace);
    int32_t *flags = reinterpret_cast<int32_t *>(ws + l.flags);
    int32_t *stats = reinterpret_cast<int32_t *>(ws + l.stats);
    int32_t *scratch = reinterpret_cast<int32_t *>(ws + l.scratch);
    const int64_t B = d->B, HW = d->HW;
    const int n_ch = d->n_chunks, D = d->lag;
    // flags[c] is only ever SET by the table kernels (a decode launch that sees it mid-way re-decodes its chunk from state_in with the
    // generic path); stats: -1 = unknown = full rows (auto), 0 = window rows from the start (always)
    rc = l3c::check_hip(hipMemsetAsync(flags, 0, C3 * 4, main), "hipMemsetAsync");
    if (rc != L3C_OK) return rc;
    if (d->window_mode) {
        rc = l3c::check_hip(hipMemsetAsync(stats, d->window_mode == 1 ? 0xFF : 0, (size_t)C3 * (n_ch + 2) * B * 4, main), "hipMemsetAsync");
        if (rc != L3C_OK) return rc;
    }
    if (D == 2) {   // the workspace initialisation and whatever the caller enqueued on `main` (sym, the packed streams) before the side stream reads it
        rc = l3c::check_hip(hipEventRecord(ev_tables, main), "hipEventRecord");
        if (rc == L3C_OK) rc = l3c::check_hip(hipStreamWaitEvent(side, ev_tables, 0), "hipStreamWaitEvent");
        if (rc != L3C_OK) return rc;
    }
    l3c_ragged_batch rb{};
    const int64_t *dev_pix0 = nullptr, *dev_npix = nullptr, *dev_off = nullptr;
    if (rag) {
        int64_t max_hw = 0;
        for (int64_t b = 0; b < B; ++b) max_hw = rag->hw_host[b] > max_hw ? rag->hw_host[b] : max_hw;
        rb = l3c_ragged_batch{B, max_hw, rag->tables_dev, rag->tables_dev + B};
        dev_pix0 = rag->tables_dev + 2 * B;
        dev_npix = dev_pix0 + (int64_t)n_ch * B;
        dev_off = dev_npix + (int64_t)n_ch * B;
    }
    const int n_steps = n_ch + D * (C3 - 1);
    int last_decoded = -1;
    for (int t = 0; t < n_steps; ++t) {
        l3c_table_part tp[C3];
        l3c_ragged_part rp[C3];
        l3c_ac_decode_part dp[C3];
        int n = 0;
        uint8_t *slot = ws + l.tables + (D == 2 ? (t & 1) : 0) * C3 * l.table_bytes;
        for (int c = 0; c < C3; ++c) {
            const int j = t - D * c;
            if (j < 0 || j >= n_ch) continue;
            int64_t p0 = 0, np = 0, total = 0;
            if (rag) {
                for (int64_t b = 0; b < B; ++b) {
                    const int64_t nb = rag->npix_host[(int64_t)j * B + b];
                    np = nb > np ? nb : np;
                    total += nb;
                }
            } else {
                p0 = d->chunk_pix0_host[j];
                np = d->chunk_npix_host[j];
            }
            uint16_t *table = reinterpret_cast<uint16_t *>(slot + c * l.table_bytes);
            int32_t *st_in = d->window_mode ? stats + ((int64_t)c * (n_ch + 2) + j) * B : nullptr;
            int32_t *st_out = d->window_mode == 2 ? scratch : d->window_mode == 1 ? stats + ((int64_t)c * (n_ch + 2) + j + 2) * B : nullptr;
            tp[n] = l3c_table_part{c, p0, np, table, flags + c, st_in};
            l3c_ac_decode_part q;
            memset(&q, 0, sizeof(q));
            q.cdf = table;
            q.Lp = LP;
            q.in = d->in;
            q.in_offsets = d->in_offsets + (int64_t)c * B;
            q.in_nbytes = d->in_nbytes + (int64_t)c * B;
            q.n_streams = B;
            q.n_sym = np;
            q.not_monotone_flag = flags + c;
            q.state_in = j ? ws + l.states + (c * 2 + ((j + 1) & 1)) * l.state_bytes : nullptr;
            q.state_out = ws + l.states + (c * 2 + (j & 1)) * l.state_bytes;
            q.final_chunk = j == n_ch - 1;
            q.sym_out = d->sym;
            q.sym_stride = C3 * HW;
            q.sym_offset = c * HW + p0;
            if (rag) {
                rp[n] = l3c_ragged_part{dev_pix0 + (int64_t)j * B, dev_npix + (int64_t)j * B, dev_off + (int64_t)j * B};
                q.r_npix = rp[n].npix;
                q.r_table_off = rp[n].table_off;
                q.r_pixbase = rb.pixbase;
                q.r_hw = rb.hw;
                q.r_pix0 = rp[n].pix0;
                q.r_C = C3;
                q.r_c = c;
                q.r_table_bytes = total * LP * 2;
            }
            if (d->window_mode) {
                q.window_stats_in = st_in;
                q.window_stats_out = st_out;
                q.P = d->P;
                q.sym_all = d->sym;
                q.targets = d->targets;
                q.HW = HW;
                q.pix0 = p0;
                q.C = C3;
                q.K = d->K;
                q.c = c;
            }
            dp[n++] = q;
        }
        if (!n) continue;
        if (D == 2 && t >= 2 && last_decoded >= t - 2) {   // the symbols of step t - 2 (older steps are ordered before it on the side stream)
            rc = l3c::check_hip(hipStreamWaitEvent(main, ev_decoded[(t - 2) % 3], 0), "hipStreamWaitEvent");
            if (rc != L3C_OK) return rc;
        }
        rc = rag ? l3c_dmll_cdf_table_ragged(d->P, d->sym, d->targets, &rb, C3, d->K, 1, LP, tp, rp, n, main)
                 : l3c_dmll_cdf_table_parts(d->P, d->sym, d->targets, B, HW, C3, d->K, 1, LP, tp, n, main);
        if (rc != L3C_OK) return rc;
        if (D == 2) {
            rc = l3c::check_hip(hipEventRecord(ev_tables, main), "hipEventRecord");
            if (rc == L3C_OK) rc = l3c::check_hip(hipStreamWaitEvent(side, ev_tables, 0), "hipStreamWaitEvent");
            if (rc != L3C_OK) return rc;
        }
        rc = l3c_ac_decode_chunks(dp, n, side);
        if (rc != L3C_OK) return rc;
        if (D == 2) {
            rc = l3c::check_hip(hipEventRecord(ev_decoded[t % 3], side), "hipEventRecord");
            if (rc != L3C_OK) return rc;
            last_decoded = t;
        }
    }
    if (D == 2 && last_decoded >= 0) {   // the caller's stream continues after the last symbols
        rc = l3c::check_hip(hipStreamWaitEvent(main, ev_decoded[last_decoded % 3], 0), "hipStreamWaitEvent");
        if (rc != L3C_OK) return rc;
    }
    return L3C_OK;
}

}  // namespace

extern "C" {

int l3c_decode_rgb(const l3c_rgb_decode_desc *d, l3c_stream_t main_stream, l3c_stream_t side_stream) {
    L3C_REQUIRE(d, "null descriptor");
    L3C_REQUIRE(d->P && d->targets && d->sym && d->in && d->in_offsets && d->in_nbytes && d->workspace, "null pointer in descriptor");
    L3C_REQUIRE(d->B > 0 && d->B < 65536 && d->HW > 0 && d->K > 0 && d->K <= 16, "bad shape");
    L3C_REQUIRE(d->n_chunks > 0 && d->n_chunks <= 4096 && d->chunk_pix0_host && d->chunk_npix_host, "bad chunk list");
    L3C_REQUIRE(d->lag == 1 || d->lag == 2, "lag must be 1 (one stream) or 2 (tables and decoders overlapped on two streams)");
    L3C_REQUIRE(d->lag == 1 || side_stream != main_stream, "lag 2 needs a side stream that is not the main stream");
    L3C_REQUIRE(d->window_mode >= 0 && d->window_mode <= 2, "window_mode: 0 never, 1 auto, 2 always");
    L3C_REQUIRE((reinterpret_cast<uintptr_t>(d->workspace) & (ALIGN - 1)) == 0, "workspace must be 256-byte aligned");
    int64_t max_npix = 0, next = 0;
    for (int j = 0; j < d->n_chunks; ++j) {
        L3C_REQUIRE(d->chunk_pix0_host[j] == next && d->chunk_npix_host[j] > 0, "chunks must tile [0, HW) in order");
        L3C_REQUIRE(j + 1 == d->n_chunks || d->chunk_npix_host[j] % 64 == 0, "chunk boundaries must lie on the 64-symbol store blocks");
        next += d->chunk_npix_host[j];
        max_npix = d->chunk_npix_host[j] > max_npix ? d->chunk_npix_host[j] : max_npix;
    }
    L3C_REQUIRE(next == d->HW, "chunks must tile [0, HW) in order");
    return decode_rgb_impl(d, nullptr, max_npix, main_stream, side_stream);
}

int64_t l3c_decode_rgb_ragged_workspace_bytes(int64_t B, int64_t max_chunk_total_npix, int n_chunks, int lag) {
    if (B <= 0 || max_chunk_total_npix <= 0 || n_chunks <= 0 || (lag != 1 && lag != 2)) return -1;
    return layout(1, max_chunk_total_npix, n_chunks, lag, B).total;
}

int l3c_decode_rgb_ragged(const l3c_rgb_ragged_desc *r, l3c_stream_t main_stream, l3c_stream_t side_stream) {
    L3C_REQUIRE(r, "null descriptor");
    L3C_REQUIRE(r->P && r->targets && r->sym && r->in && r->in_offsets && r->in_nbytes && r->workspace && r->hw_host && r->tables_dev,
                "null pointer in descriptor");
    L3C_REQUIRE(r->B > 0 && r->B < 65536 && r->K > 0 && r->K <= 16, "bad shape");
    L3C_REQUIRE(r->n_chunks > 0 && r->n_chunks <= 4096 && r->chunk_pix0_host && r->chunk_npix_host, "bad chunk list");
    L3C_REQUIRE(r->lag == 1 || r->lag == 2, "lag must be 1 (one stream) or 2 (tables and decoders overlapped on two streams)");
    L3C_REQUIRE(r->lag == 1 || side_stream != main_stream, "lag 2 needs a side stream that is not the main stream");
    L3C_REQUIRE(r->window_mode >= 0 && r->window_mode <= 2, "window_mode: 0 never, 1 auto, 2 always");
    L3C_REQUIRE((reinterpret_cast<uintptr_t>(r->workspace) & (ALIGN - 1)) == 0, "workspace must be 256-byte aligned");
    int64_t max_total = 0;
    for (int64_t b = 0; b < r->B; ++b) {
        int64_t next = 0;
        for (int j = 0; j < r->n_chunks; ++j) {
            const int64_t p0 = r->chunk_pix0_host[(int64_t)j * r->B + b], np = r->chunk_npix_host[(int64_t)j * r->B + b];
            L3C_REQUIRE(p0 == next && np > 0, "every image's chunks must tile [0, HW_b) in order, none empty");
            L3C_REQUIRE(j + 1 == r->n_chunks || np % 64 == 0, "chunk boundaries must lie on the 64-symbol store blocks");
            next += np;
        }
        L3C_REQUIRE(next == r->hw_host[b], "every image's chunks must tile [0, HW_b) in order");
    }
    for (int j = 0; j < r->n_chunks; ++j) {
        int64_t total = 0;
        for (int64_t b = 0; b < r->B; ++b) total += r->chunk_npix_host[(int64_t)j * r->B + b];
        max_total = total > max_total ? total : max_total;
    }
    l3c_rgb_decode_desc d{};
    d.P = r->P;  d.targets = r->targets;  d.sym = r->sym;  d.B = r->B;  d.HW = 0;  d.K = r->K;
    d.in = r->in;  d.in_offsets = r->in_offsets;  d.in_nbytes = r->in_nbytes;
    d.n_chunks = r->n_chunks;  d.lag = r->lag;  d.window_mode = r->window_mode;
    d.workspace = r->workspace;  d.workspace_bytes = r->workspace_bytes;
    const Ragged rag{r->hw_host, r->chunk_pix0_host, r->chunk_npix_host, r->tables_dev};
    return decode_rgb_impl(&d, &rag, max_total, main_stream, side_stream);
}
}
