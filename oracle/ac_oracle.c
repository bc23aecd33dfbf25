/*
 * oracle/ac_oracle.c -- CPU restatement of the reference's adaptive range coder.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the product package (l3c-pytorch_amd/) may link, import or call
 * this file; it exists so tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg can check the HIP
 * coder bit-for-bit.  It is itself pinned against the real reference (oracle/_ref/torchac_backend_cpu.so,
 * built from /root/reference/src/torchac/torchac_backend/torchac.cpp) and against tests/golden/ac_kat.npz.
 *
 * Algorithm restated (file:line in /root/reference/src/torchac/torchac_backend/torchac.cpp):
 *   - 32-bit low/high interval, 16-bit CDF precision, bit-wise output with pending ("underflow") bits:
 *     encode()  :152-227      bit sink OutCacheString :63-93
 *   - decoder with the reference's own binary search (exact-hit early return) and the quirk that the
 *     LAST symbol skips the state update: decode() :299-381, binsearch() :276-296, bit source :96-128
 *   - c_high of the top symbol (sym == Lp-2) is the constant 0x10000, entry Lp-1 of a row is never read
 *     (:181, :340).
 * Table layout: N rows of Lp uint16, row-major (get_cdf_ptr :131-145).  `row_stride` == 0 broadcasts one
 * row to every symbol (the uniform-prior scale, bitcoding.py:297-323, uses N identical rows).
 *
 * Build: gcc -O2 -shared -fPIC -o oracle/_build/libac_oracle.so oracle/ac_oracle.c
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef struct {
    uint8_t *buf;
    size_t cap;
    size_t n;      /* whole bytes written */
    uint8_t acc;   /* partial byte, MSB first */
    int fill;      /* bits in acc */
    int overflow;
} bit_sink;

static void sink_put(bit_sink *s, int bit) {
    s->acc = (uint8_t)((s->acc << 1) | (bit & 1));
    if (++s->fill == 8) {
        if (s->n < s->cap) s->buf[s->n] = s->acc; else s->overflow = 1;
        s->n++;
        s->fill = 0;
        s->acc = 0;
    }
}

static void sink_put_with_pending(bit_sink *s, int bit, uint64_t *pending) {
    sink_put(s, bit);
    while (*pending) { sink_put(s, !bit); --*pending; }
}

/*
 * Encode N symbols.  Returns the number of bytes produced (may exceed `cap`: then nothing past cap was
 * stored and the caller must retry with a larger buffer).
 */
long long l3c_oracle_ac_encode(const uint16_t *cdf, long long row_stride, int Lp,
                               const int16_t *sym, long long N,
                               uint8_t *out, long long cap) {
    bit_sink s = { out, (size_t)cap, 0, 0, 0, 0 };
    uint32_t low = 0, high = 0xFFFFFFFFu;
    uint64_t pending = 0;
    const int top_symbol = Lp - 2;

    for (long long i = 0; i < N; ++i) {
        const uint16_t *row = cdf + i * row_stride;
        const int x = sym[i];
        const uint64_t span = (uint64_t)high - (uint64_t)low + 1;
        const uint32_t c_lo = row[x];
        const uint32_t c_hi = (x == top_symbol) ? 0x10000u : row[x + 1];
        high = (low - 1) + (uint32_t)((span * c_hi) >> 16);
        low = low + (uint32_t)((span * c_lo) >> 16);
        for (;;) {
            if (high < 0x80000000u) {
                sink_put_with_pending(&s, 0, &pending);
            } else if (low >= 0x80000000u) {
                sink_put_with_pending(&s, 1, &pending);
            } else if (low >= 0x40000000u && high < 0xC0000000u) {
                ++pending;
                low = (low << 1) & 0x7FFFFFFFu;
                high = (high << 1) | 0x80000001u;
                continue;
            } else {
                break;
            }
            low <<= 1;
            high = (high << 1) | 1u;
        }
    }
    ++pending;
    sink_put_with_pending(&s, low < 0x40000000u ? 0 : 1, &pending);
    if (s.fill) { while (s.fill) sink_put(&s, 0); }
    return (long long)s.n;
}

typedef struct {
    const uint8_t *buf;
    size_t len;
    size_t pos;
    uint8_t cur;
    int left;
} bit_source;

static void source_shift_in(bit_source *b, uint32_t *value) {
    if (b->left == 0) {
        if (b->pos == b->len) { *value <<= 1; return; }   /* zeros past the end */
        b->cur = b->buf[b->pos++];
        b->left = 8;
    }
    *value = (*value << 1) | ((b->cur >> (b->left - 1)) & 1u);
    --b->left;
}

static uint16_t ref_binsearch(const uint16_t *row, uint16_t target, uint16_t top_symbol) {
    uint16_t left = 0, right = (uint16_t)(top_symbol + 1);
    while (left + 1 < right) {
        const uint16_t m = (uint16_t)((left + right) / 2);
        const uint16_t v = row[m];
        if (v < target) left = m;
        else if (v > target) right = m;
        else return m;
    }
    return left;
}

void l3c_oracle_ac_decode(const uint16_t *cdf, long long row_stride, int Lp,
                          const uint8_t *in, long long in_len,
                          int16_t *sym_out, long long N) {
    bit_source b = { in, (size_t)in_len, 0, 0, 0 };
    uint32_t low = 0, high = 0xFFFFFFFFu, value = 0;
    const int top_symbol = Lp - 2;
    for (int i = 0; i < 32; ++i) source_shift_in(&b, &value);

    for (long long i = 0; i < N; ++i) {
        const uint16_t *row = cdf + i * row_stride;
        const uint64_t span = (uint64_t)high - (uint64_t)low + 1;
        const uint16_t count =
            (uint16_t)((((uint64_t)value - (uint64_t)low + 1) * 0x10000u - 1) / span);
        const uint16_t x = ref_binsearch(row, count, (uint16_t)top_symbol);
        sym_out[i] = (int16_t)x;
        if (i == N - 1) break;
        const uint32_t c_lo = row[x];
        const uint32_t c_hi = (x == top_symbol) ? 0x10000u : row[x + 1];
        high = (low - 1) + (uint32_t)((span * c_hi) >> 16);
        low = low + (uint32_t)((span * c_lo) >> 16);
        for (;;) {
            if (low >= 0x80000000u || high < 0x80000000u) {
                low <<= 1;
                high = (high << 1) | 1u;
            } else if (low >= 0x40000000u && high < 0xC0000000u) {
                low = (low << 1) & 0x7FFFFFFFu;
                high = (high << 1) | 0x80000001u;
                value -= 0x40000000u;
            } else {
                break;
            }
            source_shift_in(&b, &value);
        }
    }
}
