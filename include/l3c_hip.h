/*
 * l3c_hip.h -- C ABI of libl3c_hip.so: the MI355X-native (gfx950) L3C inference path.
 *
 * This is the drop-in boundary.  The reference's native FFI for this path is the pybind11 torch
 * extension `torchac_backend_gpu` / `torchac_backend_cpu` (src/torchac/torchac_backend/torchac.cpp:433-443) plus the cuDNN
 * convolutions it reaches through torch.nn; everything below replaces one of those call sites with a HIP kernel behind
 * plain pointers and sizes -- no torch types, no hidden allocation, no hidden synchronisation.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer unless its name ends in `_host`;
 *   - `stream` is a hipStream_t passed as void* (NULL = the null stream); all work is enqueued asynchronously;
 *   - every function returns L3C_OK (0) or a negative status; l3c_last_error() returns a thread-local message;
 *   - buffers are caller-owned; sizes are element counts unless the name says `_bytes`;
 *   - activations are fp32 "pixel-major" (NHWC): element (b, y, x, c) lives at ((b*H + y)*W + x)*cstride + coff + c,
 *     where `cstride` is the number of channels per pixel in memory (lets a conv write a channel slice of a wider tensor);
 *   - symbols are int16, planar (b, c, y*W + x) like the reference hands them to its coder (bitcoding/coders.py:46-51).
 */
#ifndef L3C_HIP_H_
#define L3C_HIP_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define L3C_OK 0
#define L3C_ERR_INVALID_ARG (-1)
#define L3C_ERR_HIP (-2)
#define L3C_ERR_UNSUPPORTED (-3)

#define L3C_ABI_VERSION 4

typedef void *l3c_stream_t;

/* ---- library ------------------------------------------------------------------------------------------------------ */

int l3c_abi_version(void);
/*
 * Generation of the BITSTREAM this build reads and writes.  The `.l3c` container has no version field (reference:
 * src/bitcoding/bitcoding.py:326-375) and the format is only decodable by a decoder whose kernels reproduce the encoder's P -- and
 * from it the 16-bit table entries -- BIT FOR BIT (the reference's own check is the round trip of
 * src/test/multiscale_tester.py:373).  So every change that alters a bit of P on purpose (a different summation order in a decoder-side
 * convolution, a different transcendental in the mixture head, another compiler contraction setting) bumps this number, and files of
 * an older generation are NOT readable by this build.  tests/golden/hip_*.l3c + hip_bitstream.json are files and P hashes written by the
 * generation they name; tests/test_gpu_bitstream.py fails when today's build no longer decodes them or no longer reproduces the hashes
 * (i.e. when a kernel changed P bits WITHOUT the bump).  `l3c.py enc` prints it.
 *   1  rounds 1-2 (F(2x2,3x3) convolutions)        2  round 3 (F(4x4,3x3), polyphase 5x5)        3  round 5 (fused multiply-adds in the 3->64 head)
 */
#define L3C_BITSTREAM_GENERATION 3
int l3c_bitstream_generation(void);
/* Thread-local description of the last failing call ("" if none). */
const char *l3c_last_error(void);
/* Name / CU count / gcnArchName of the current HIP device; fails when no GPU is visible (there is no CPU fallback). */
int l3c_device_info(char *name_host, int name_cap, int *num_cu_host, char *arch_host, int arch_cap);

/*
 * A HIP stream whose kernels only run on compute units [first_cu, first_cu + n_cu) (hipExtStreamCreateWithCUMask).
 * Used to give the latency-bound range-coder wavefronts CUs of their own while the MFMA-bound conv kernels of the next
 * batch run on the complementary range: co-resident MFMA waves slow the coder's serial chain down 2.3x otherwise.
 */
int l3c_stream_create_cu_range(int first_cu, int n_cu, l3c_stream_t *stream_out_host);
/* The general form: bit i of the mask (32 CUs per word, `mask_host` is a HOST array of n_words words) = compute unit i may run this stream's
 * kernels.  The dispatcher deals workgroups to the 8 XCDs round robin whatever the mask says, so a mask should hold the same number of
 * CUs of every XCD (helpers/runtime.balanced_cu_mask). */
int l3c_stream_create_cu_mask(const uint32_t *mask_host, int n_words, l3c_stream_t *stream_out_host);
int l3c_stream_destroy(l3c_stream_t stream);

/* ---- arithmetic coder (replaces torchac.cpp) ---------------------------------------------------------------------- */

/*
 * Coding interval of one symbol: TWO words, one per role of the lane pair that codes the stream (csrc/ac_core.h):
 * role 0 = c_low, role 1 = 65536 - c_high (c_high in 1..65536).  Interval streams are stored in blocks of 64 symbols,
 * per block and stream one 64-word run per role: word(stream s, symbol t, role r) = iv[(((t/64)*n_streams + s)*2 + r)*64 + t%64],
 * so the kernel that writes them and the lane that codes role r of stream s both touch contiguous 256-byte runs.
 * The buffer is opaque between the l3c_*intervals* call that fills it and the l3c_ac_encode* call that consumes it;
 * l3c_interval_words(n_streams, n_sym) gives its size in uint32 words.
 */
int64_t l3c_interval_words(int64_t n_streams, int64_t n_sym);

/*
 * Intervals from an explicit CDF table (the reference's encode_cdf path, torchac.cpp:263-269 -> encode() :174-182:
 * c_low = cdf[i*Lp + s], c_high = s == Lp-2 ? 0x10000 : cdf[i*Lp + s + 1]).
 *   cdf        uint16 [n_streams][n_sym][Lp] when row_stride == Lp; row_stride == 0 broadcasts ONE row of Lp entries
 *              to every symbol of every stream (uniform prior of the coarsest scale, bitcoding.py:297-323)
 *   sym        int16 [n_streams][n_sym]
 */
int l3c_ac_intervals_from_table(const uint16_t *cdf, int64_t row_stride, int Lp, const int16_t *sym,
                                int64_t n_streams, int64_t n_sym, uint32_t *intervals, l3c_stream_t stream);

/*
 * Range-encode n_streams independent symbol streams of equal length.  Bit-exact restatement of encode()
 * (torchac.cpp:152-227): 32-bit low/high, 16-bit precision, pending-bit carry handling, final flush of `pending+1`
 * bits, zero padding to a byte boundary, MSB-first.  Two launches:
 *   phase 1  one stream per lane PAIR (32 per wavefront; one lane updates low, its neighbour ~high, with the same
 *            ~15 instructions per symbol): the serial interval recurrence only; every interval word is REPLACED IN PLACE
 *            by the bound right after the interval update (low' / ~high'), from which phase 2 derives what is emitted
 *   phase 2  one stream per wavefront, 64 symbols per step: records -> bits (wave scans + LDS merge), coalesced stores
 * Precondition: c_high > c_low for every symbol (strictly increasing table rows), as for the reference.  Intervals that violate it
 * (the entry points cannot validate device-side intervals) produce garbage, as the reference does, but never a store outside the
 * stream's own out_stride_bytes slot: such a stream reports out_nbytes = L3C_AC_OVERRUN when it asked for more.
 *   intervals  in/out, CLOBBERED
 *   out        uint8 [n_streams][out_stride_bytes]; out_stride_bytes % 4 == 0 and >= l3c_ac_max_bytes(n_sym)
 *   out_nbytes uint32 [n_streams]   number of bytes produced per stream, or L3C_AC_OVERRUN
 *   workspace  l3c_ac_encode_workspace_bytes(n_streams) bytes, 4-byte aligned
 */
#define L3C_AC_OVERRUN 0xFFFFFFFFu
int64_t l3c_ac_max_bytes(int64_t n_sym);
int64_t l3c_ac_encode_workspace_bytes(int64_t n_streams);
int l3c_ac_encode(uint32_t *intervals, int64_t n_streams, int64_t n_sym, uint8_t *out, int64_t out_stride_bytes,
                  uint32_t *out_nbytes, void *workspace, l3c_stream_t stream);

/*
 * Grouped form of l3c_ac_encode: ONE phase-1 launch and ONE phase-2 launch cover any number of groups of equally long
 * streams (the four scales of a batch; or every batch of a heterogeneous image set), so all of them are coded concurrently
 * without relying on stream-level concurrency.  Descriptors are given in HOST memory; pointers inside are device pointers.
 *   workspace  l3c_ac_encode_groups_workspace_bytes(n_groups, total number of streams) bytes, 16-byte aligned
 */
typedef struct {
    uint32_t *intervals;       /* in/out, clobbered; l3c_interval_words(n_streams, n_sym) words */
    uint8_t *out;              /* [n_streams][out_stride_bytes] */
    uint32_t *out_nbytes;      /* [n_streams] */
    int64_t n_streams, n_sym, out_stride_bytes;
} l3c_ac_group;
int64_t l3c_ac_encode_groups_workspace_bytes(int n_groups, int64_t total_streams);
int l3c_ac_encode_groups(const l3c_ac_group *groups_host, int n_groups, void *workspace, l3c_stream_t stream);

/*
 * Range-decode n_streams streams, one wavefront per stream (the 64 lanes hold the CDF row of the current symbol and
 * rank the decoder's `count` against it).  Bit-exact restatement of decode() (torchac.cpp:299-381) including the
 * reference binary search (binsearch :276-296) and the skipped state update of the last symbol.
 *   cdf         uint16 rows as above (row_stride == Lp or 0), Lp <= 257
 *   in          uint8, stream s occupies in[in_offsets[s] .. in_offsets[s] + in_nbytes[s]); bits past the end read 0
 *   monotone    1: every row is known to be strictly increasing over [0, Lp-2] (rank == reference binsearch; fast path)
 *               0: use the literal binary search of the reference for every symbol
 *   sym_out     int16 [n_streams][n_sym]
 */
int l3c_ac_decode(const uint16_t *cdf, int64_t row_stride, int Lp, const uint8_t *in, const int64_t *in_offsets,
                  const uint32_t *in_nbytes, int64_t n_streams, int64_t n_sym, int monotone, int16_t *sym_out,
                  l3c_stream_t stream);

/* flag_out[0] |= 1 when some row of the table is not strictly increasing over entries [0, Lp-2]. */
/*
 * Chunked, grouped form of l3c_ac_decode for streams whose tables become available piecewise (the RGB scale: the table of G
 * at a pixel needs R decoded at that pixel, logistic_mixture.py:262-272).  Each part decodes symbols [sym_offset, sym_offset +
 * n_sym) of its streams from the table rows of exactly that range, resuming from / saving the coder state; the 1..8 parts
 * of a call are independent and run side by side in ONE launch pair (R chunk j+2, G chunk j+1, B chunk j of a pipeline step).
 *   cdf                [n_streams][n_sym][Lp] rows of THIS chunk
 *   not_monotone_flag  device int32 (as written by l3c_cdf_check_monotone / l3c_dmll_cdf_table), read by the kernel -- no host
 *                      synchronisation; null: the table is treated as not validated (reference's literal search)
 *   state_in / _out    [n_streams] x l3c_ac_decode_state_bytes(); state_in null = start of the streams; must differ
 *   final_chunk        non-zero when the streams end with this chunk (their last symbol skips the update, torchac.cpp:335-337)
 *   sym_out            stream s writes sym_out[s * sym_stride + sym_offset + i], i < n_sym
 * WINDOW ROWS (ABI version 3; all fields NULL / 0: classic rows).  With window_stats_in set, `cdf` was built by l3c_dmll_cdf_table
 * with the same window_stats array: stream s's rows -- at its full-size slot cdf + s * n_sym * Lp -- are 65-entry WINDOW rows around
 * the mixture's mean when window_stats_in[s] says the stream missed at most 1/64 of the symbols two chunks earlier (0 <= value < 2^30; a
 * quarter of the table arithmetic and bytes of the 257-entry rows), and full rows otherwise.  A symbol outside its window makes the decoder wavefront
 * evaluate the full row of that pixel itself from P / sym_all / targets (the table kernel's device functions: the same bits), so
 * the decoded symbols never depend on the row form.  window_stats_out[s] receives the stream's miss count of THIS chunk (what a window
 * would have missed when the rows were full), bit 30 set when that is more than 1/64 of the chunk's symbols (INT32_MAX when the stream went
 * through the generic pass): feed it to the table call and the
 * decode part of the chunk after the next one.  RGB scale only (Lp == 257, C == 3), one stream per image: stream s = image s of P.
 *   P, sym_all, targets, HW, C, K, c   as for l3c_dmll_cdf_table;   pix0   first pixel of this chunk (sym_all index of symbol 0)
 */
typedef struct {
    const uint16_t *cdf;
    int Lp;
    const uint8_t *in;
    const int64_t *in_offsets;
    const uint32_t *in_nbytes;
    int64_t n_streams, n_sym;
    const int32_t *not_monotone_flag;
    const void *state_in;
    void *state_out;
    int final_chunk;
    int16_t *sym_out;
    int64_t sym_stride, sym_offset;
    const int32_t *window_stats_in;
    int32_t *window_stats_out;
    const float *P;
    const int16_t *sym_all;
    const float *targets;
    int64_t HW, pix0;
    int C, K, c;
    /* RAGGED batches (round 6; all NULL / 0: the rectangular form above).  Streams of DIFFERENT lengths -- the images of a set decode have
     * different sizes -- in one launch: stream s decodes r_npix[s] symbols from the rows at BYTE offset r_table_off[s] of `cdf` (its full-size
     * slot of r_npix[s] * Lp entries) and writes them to sym_out + r_C * r_pixbase[s] + r_c * r_hw[s] + r_pix0[s], i.e. into channel r_c of an
     * image whose r_C planes of r_hw[s] symbols start at element r_C * r_pixbase[s]; P / sym_all of the window fields are indexed the same way
     * (image s: pixels r_pixbase[s] .. + r_hw[s]).  n_sym / sym_stride / sym_offset / HW / pix0 are ignored; r_table_bytes = size of the table. */
    const int64_t *r_npix, *r_table_off, *r_pixbase, *r_hw, *r_pix0;   /* device, [n_streams] each */
    int r_C, r_c;
    int64_t r_table_bytes;
} l3c_ac_decode_part;
int64_t l3c_ac_decode_state_bytes(void);
int l3c_ac_decode_chunks(const l3c_ac_decode_part *parts, int n_parts, l3c_stream_t stream);

int l3c_cdf_check_monotone(const uint16_t *cdf, int64_t n_rows, int Lp, int32_t *flag_out, l3c_stream_t stream);

/*
 * The whole RGB scale of a batch in ONE host call (round 6): the chunk-pipelined decode of R, G and B -- channel c's table at a pixel
 * needs the channels < c decoded at that pixel (reference: criterion/logistic_mixture.py:262-272; the per-channel decoder loop it
 * replaces: bitcoding/bitcoding.py:212-266 -> torchac.cpp:299-381).  Pipeline step t handles chunk t - lag * c of channel c: ONE
 * grouped table launch (l3c_dmll_cdf_table_parts) on `main_stream`, then ONE grouped decoder launch (l3c_ac_decode_chunks) on
 * `side_stream` (lag 2: the tables of step t + 1 are built while step t decodes; lag 1: everything on main_stream, side_stream unused).
 * Returns with everything enqueued and main_stream ordered after the last symbols; no host synchronisation, no allocation: table
 * slots, coder states, validity flags and window statistics live in the caller's workspace (l3c_decode_rgb_workspace_bytes).
 *   P            fp32 pixel-major [B][HW][120] (K = 10: 4 * 3 * K)      targets  fp32 [257]
 *   sym          int16 planar [B][3][HW], ZEROED by the caller; receives the symbols, read back for the lambda coupling
 *   in / in_offsets / in_nbytes   the streams, CHANNEL-major: stream (c, b) at index c * B + b (4-byte aligned, zero padded: l3c_container_read)
 *   chunks       n_chunks (pix0, npix) ranges tiling [0, HW) in order, boundaries on multiples of 64; HOST arrays
 *   window_mode  0: full 257-entry rows; 1: 65-entry window rows wherever the stream's statistics two chunks earlier say that pays
 *                (see l3c_ac_decode_part; the first two chunks of a channel are then decoded on full rows: make them short probes);
 *                2: window rows from the first chunk on (tests)
 */
typedef struct {
    const float *P;
    const float *targets;
    int16_t *sym;
    int64_t B, HW;
    int K;
    const uint8_t *in;
    const int64_t *in_offsets;
    const uint32_t *in_nbytes;
    int n_chunks;
    const int64_t *chunk_pix0_host;
    const int64_t *chunk_npix_host;
    int lag;
    int window_mode;
    void *workspace;
    int64_t workspace_bytes;
} l3c_rgb_decode_desc;
/*
 * The same for a RAGGED batch (round 6): B images of DIFFERENT sizes decoded in lock step -- one grouped table launch and one decoder
 * launch per pipeline step for all of them -- so that a set of differently sized images (the reference's folder evaluation,
 * test/multiscale_tester.py:353-381) is not decoded one latency-bound image after the other.  Every image has the same NUMBER of chunks;
 * image b's chunk j is pixels [chunk_pix0_host[j * B + b], + chunk_npix_host[j * B + b]).  P / sym ragged as for
 * l3c_dmll_cdf_table_ragged with pixbase[b] = sum of hw_host[0 .. b).  `tables_dev`: device int64 array the CALLER has uploaded:
 * pixbase [B] | hw [B] | pix0 [n_chunks][B] | npix [n_chunks][B] | table_off [n_chunks][B], table_off[j][b] = 514 * sum of npix[j][0 .. b).
 */
typedef struct {
    const float *P;
    const float *targets;
    int16_t *sym;
    int64_t B;
    const int64_t *hw_host;
    int K;
    const uint8_t *in;
    const int64_t *in_offsets;
    const uint32_t *in_nbytes;
    int n_chunks;
    const int64_t *chunk_pix0_host;
    const int64_t *chunk_npix_host;
    const int64_t *tables_dev;
    int lag;
    int window_mode;
    void *workspace;
    int64_t workspace_bytes;
} l3c_rgb_ragged_desc;
int64_t l3c_decode_rgb_ragged_workspace_bytes(int64_t B, int64_t max_chunk_total_npix, int n_chunks, int lag);
int l3c_decode_rgb_ragged(const l3c_rgb_ragged_desc *desc_host, l3c_stream_t main_stream, l3c_stream_t side_stream);
int64_t l3c_decode_rgb_workspace_bytes(int64_t B, int64_t max_chunk_npix, int n_chunks, int lag);
/* byte offset, inside the workspace, of the window statistics int32 [3][n_chunks + 2][B] (slot j + 2 = what chunk j's decoders reported; tests) */
int64_t l3c_decode_rgb_stats_offset(int64_t B, int64_t max_chunk_npix, int n_chunks, int lag);
int l3c_decode_rgb(const l3c_rgb_decode_desc *desc_host, l3c_stream_t main_stream, l3c_stream_t side_stream);

/* ---- logistic-mixture head (replaces torchac_kernel.cu + criterion/logistic_mixture.py on the coding path) --------- */

/*
 * Per-channel mixture parameters, the reference's CDFOut (logistic_mixture.py:134-141, :248-275):
 *   P      fp32 pixel-major [B][HW][Kp], Kp = (rgb ? 4 : 3) * C * K, channel index p*(C*K) + c*K + k
 *   sym    int16 planar [B][C][HW]; only read for rgb && c > 0 (lambda coupling with the ACTUAL values of the previous
 *          channels: mu_G += sigmoid(lam_0) x_R, mu_B += sigmoid(lam_1) x_R + sigmoid(lam_2) x_G)
 *   out    pi (softmax over K), mu, log_sigma (clamped at -7): fp32 planar [B][K][HW] each -- the 1KHW layout of CDFOut
 */
int l3c_dmll_channel_params(const float *P, const int16_t *sym, int64_t B, int64_t HW, int C, int K, int rgb, int c,
                            float *pi, float *mu, float *log_sigma, l3c_stream_t stream);

/*
 * uint16 CDF table of a logistic mixture -- calculate_cdf_kernel (torchac_kernel.cu:26-76) / _get_uint16_cdf
 * (torchac.py:174-213):  cdf[n][l] = uint16(lrintf(sum_k pi[k][n] * sigmoid((t[l] - mu[k][n]) * exp(-ls[k][n]))
 *                                                  * (65536 - (Lp - 1))) + l)
 *   targets fp32 [Lp]; pi, mu, log_sigma fp32 planar [n_img][K][HW]; cdf uint16 [n_img][HW][Lp]
 *   not_monotone (may be NULL) int32, |= 1 if a row is not strictly increasing over [0, Lp-2]
 */
int l3c_cdf_table_mixture(const float *targets, const float *pi, const float *mu, const float *log_sigma,
                          int64_t n_img, int64_t HW, int K, int Lp, uint16_t *cdf, int32_t *not_monotone,
                          l3c_stream_t stream);

/*
 * Decoder, fused: the uint16 table rows of channel c for pixels [pix0, pix0 + npix) of every image, straight from P and the
 * symbols of the channels decoded so far (l3c_dmll_channel_params + l3c_cdf_table_mixture in one pass; identical entries).
 *   cdf [B][npix][Lp];  not_monotone: optional flag, set (never cleared) if a row is not strictly increasing
 *   window_stats (ABI version 3; NULL: every row a full row): int32 [B], see l3c_ac_decode_part -- image b's rows are 65-entry
 *   window rows (entries 0..63 = cdf[w0 .. w0 + 63], entry 64 = the window's offset w0) packed from the start of its full-size slot
 *   cdf + b * npix * Lp when 0 <= window_stats[b] < 2^30; otherwise full rows whose (never read) entry Lp - 1
 *   carries w0.  Lp must be 257.
 */
int l3c_dmll_cdf_table(const float *P, const int16_t *sym, const float *targets, int64_t B, int64_t HW, int C, int K, int rgb,
                       int c, int64_t pix0, int64_t npix, int Lp, uint16_t *cdf, int32_t *not_monotone, const int32_t *window_stats,
                       l3c_stream_t stream);

/*
 * Grouped form (round 6): the tables of 1..8 parts -- the channels of one pipeline step of the RGB decode, each over its own pixel
 * range, or the C channels of a bottleneck scale -- in ONE launch.  Parts share P, sym, targets and the shape; every part's arguments as
 * for l3c_dmll_cdf_table.  `parts_host` is a HOST array.
 */
typedef struct {
    int c;
    int64_t pix0, npix;
    uint16_t *cdf;               /* [B][npix][Lp] */
    int32_t *not_monotone;       /* may be NULL */
    const int32_t *window_stats; /* may be NULL */
} l3c_table_part;
int l3c_dmll_cdf_table_parts(const float *P, const int16_t *sym, const float *targets, int64_t B, int64_t HW, int C, int K, int rgb,
                             int Lp, const l3c_table_part *parts_host, int n_parts, l3c_stream_t stream);
/*
 * RAGGED form (round 6): the images of the batch have DIFFERENT sizes -- P is [sum_b HW_b][Kp] with image b's pixels from pixel pixbase[b]
 * on, sym holds image b's C planes of HW_b symbols from element C * pixbase[b] on -- and every part gives, per image, its own pixel
 * range and the byte offset of its rows in the part's table.  parts_host[i].pix0 is ignored, .npix = the LONGEST range of the part.
 */
typedef struct {
    int64_t B;                 /* images */
    int64_t max_hw;            /* the largest HW_b */
    const int64_t *pixbase;    /* device [B] */
    const int64_t *hw;         /* device [B] */
} l3c_ragged_batch;
typedef struct {
    const int64_t *pix0;       /* device [B]: first pixel of image b's range of this part */
    const int64_t *npix;       /* device [B]: its length (0: nothing for this image) */
    const int64_t *table_off;  /* device [B]: BYTE offset of image b's rows inside the part's table */
} l3c_ragged_part;
int l3c_dmll_cdf_table_ragged(const float *P, const int16_t *sym, const float *targets, const l3c_ragged_batch *batch_host, int C, int K,
                              int rgb, int Lp, const l3c_table_part *parts_host, const l3c_ragged_part *ragged_parts_host, int n_parts,
                              l3c_stream_t stream);

/*
 * Fused encoder head: straight from the network output P and the symbols to the packed coding intervals of every
 * channel of one scale -- only the two table entries the encoder reads (torchac.cpp:180-181) are evaluated, with the
 * same per-entry arithmetic as l3c_cdf_table_mixture, so the result is bit-identical to building the table first.
 *   stream index = b*C + c, n_streams = B*C, n_sym = HW; intervals sized by l3c_interval_words(B*C, HW)
 */
int l3c_dmll_encode_intervals(const float *P, const int16_t *sym, const float *targets, int64_t B, int64_t HW, int C,
                              int K, int rgb, int Lp, uint32_t *intervals, l3c_stream_t stream);

/*
 * Negative log-likelihood map of DiscretizedMixLogisticLoss.forward (logistic_mixture.py:146-207):
 *   x      fp32 planar [B][C][HW] targets (pixel values for rgb, bottleneck values for z scales)
 *   nll    fp32 planar [B][C][HW], nats
 */
int l3c_dmll_nll(const float *P, const float *x, int64_t B, int64_t HW, int C, int K, int rgb, float x_min,
                 float x_max, int L, float *nll, l3c_stream_t stream);

/*
 * Sample from the mixture (DiscretizedMixLogisticLoss._non_shared_sample, logistic_mixture.py:277-323): component by
 * Gumbel-max over the logits, value by the inverse logistic CDF; rgb: lambda coupling with the chosen components + clamp to
 * [0, 255].  The uniforms are inputs (the reference draws them with uniform_(1e-5, 1 - 1e-5), :286, :300):
 *   u_mix       fp32 planar [B][C][K][HW]        u_logistic  fp32 planar [B][C][HW]        x  fp32 planar [B][C][HW] (not rounded)
 */
int l3c_dmll_sample(const float *P, const float *u_mix, const float *u_logistic, int64_t B, int64_t HW, int C, int K,
                    int rgb, float *x, l3c_stream_t stream);

/* ---- convolution stack (replaces the cuDNN convs behind modules/{net,edsr,head,prob_clf}.py) ---------------------- */

/*
 * Weight pre-packing for l3c_conv_mfma: w_oihw fp32 [Cout][Cin][KS][KS] (the checkpoint layout) -> MFMA fragment
 * order.  l3c_conv_packed_words gives the size of `packed` in floats.  Cin % 16 == 0.
 */
int64_t l3c_conv_packed_words(int Cout, int Cin, int KS);
int l3c_conv_pack_weights(const float *w_oihw, int Cout, int Cin, int KS, float *packed, l3c_stream_t stream);

#define L3C_EPI_RELU 1         /* max(0, .) after the bias */
#define L3C_EPI_RESIDUAL 2     /* += residual[pixel][cout] after bias (and relu) */
#define L3C_EPI_PIXEL_SHUFFLE 4 /* out[(2y+i, 2x+j)][n>>2] = conv[(y,x)][n], i = (n>>1)&1, j = n&1 (nn.PixelShuffle(2)) */

typedef struct {
    const float *in;       /* [B][Hin][Win][in_cstride], channels in_coff .. in_coff+Cin */
    int in_cstride, in_coff;
    const float *packed_w; /* from l3c_conv_pack_weights */
    const float *bias;     /* [Cout] */
    const float *residual; /* [B][Hout][Wout][res_cstride] (+res_coff), or NULL */
    int res_cstride, res_coff;
    float *out;            /* [B][Hout][Wout][out_cstride] (+out_coff); with PIXEL_SHUFFLE: [B][2Hout][2Wout][..] */
    int out_cstride, out_coff;
    int B, Hin, Win, Cin, Cout;
    int KS;                /* 1, 3 or 5 */
    int stride;            /* 1, or 2 (KS == 5 only) */
    int dilation;          /* 1, 2 or 4 (KS == 3 only); padding is KS/2 when dilation == 1 else dilation */
    int epilogue;          /* L3C_EPI_* flags */
} l3c_conv_desc;

/* fp32 implicit-GEMM convolution on v_mfma_f32_32x32x2_f32; Cin % 16 == 0 (64 or 192 on this path). */
int l3c_conv_mfma(const l3c_conv_desc *desc_host, l3c_stream_t stream);
/*
 * The same convolution by Winograd F(4x4, 3x3) (interpolation points 0, 1, -1, 1/2, -2, inf): 4x fewer multiplications than the
 * direct form; the 36 per-position GEMMs on v_mfma_f32_16x16x4_f32, output transform in registers.
 * `packed_w` must come from l3c_conv_wino4_pack_weights (G g G^T computed in double precision, l3c_conv_wino4_packed_words(Cout,
 * Cin) floats).  Replaces the cuDNN call sites of the 3x3 layers (modules/edsr.py:63-89, net.py:136-148, :173-184,
 * prob_clf.py:71-74); differs from the direct convolution by fp32 rounding (measured: the L3C forward stays within 5e-6 of the
 * fp32 reference relative to each tensor's largest magnitude, profiles/r03_wino_f43_numerics.log).  Cin % 16 == 0; input and
 * packed weights 16-byte aligned, input channel stride / offset multiples of 4; output / residual: any channel slice.
 * L3C_EPI_PIXEL_SHUFFLE (the 64 -> 256 tail of edsr.Upsampler + nn.PixelShuffle(2), ABI version 3): Cout == 256, and `packed_w` must be
 * packed from the weights in SUB-PIXEL-MAJOR order -- row 64 s + oc of the tensor handed to l3c_conv_wino4_pack_weights = row 4 oc + s of
 * the layer's OIHW weights (s = 2 i + j: the sub-pixel, oc: the output channel) -- so that block s of a tile computes 64 adjacent output
 * channels of one sub-pixel and stores 64-byte runs; `bias` stays in the layer's own order.
 */
int64_t l3c_conv_wino4_packed_words(int Cout, int Cin);
int l3c_conv_wino4_pack_weights(const float *w_oihw, int Cout, int Cin, float *packed, l3c_stream_t stream);
int l3c_conv_wino4(const l3c_conv_desc *desc_host, l3c_stream_t stream);
/*
 * One phase of a STRIDE-2 convolution in its polyphase form (the 5x5 stride-2 padding-2 `down` layers, reference net.py:102,141):
 *     out[y][x] = sum_{i,j} w[i][j] in[2y-2+i][2x-2+j]  =  sum over the four input phases (a, b) of a 3x3, padding-1, stride-1
 * convolution of the sub-grid in[2r+a][2c+b] with the phase kernel w_ab[u][v] = w[2u+a][2v+b] (zero where 2u+a or 2v+b > 4) -- each
 * of which runs on the F(4x4,3x3) kernel: 9 instead of 25 multiplications per output (4 x 36/16).  desc: KS = 3, stride = 2,
 * dilation 1, Hin / Win = the FULL-resolution input (even), `packed_w` = l3c_conv_wino4_pack_weights of the phase kernel, output
 * [B][Hin/2][Win/2][..]; epilogue 0 or L3C_EPI_RESIDUAL (the caller accumulates the phases: the first launch carries the bias, the
 * other three a zero bias and residual == out, in place).
 */
int l3c_conv_wino4_phase(const l3c_conv_desc *desc_host, int phase_y, int phase_x, l3c_stream_t stream);
/*
 * The same 5x5 stride-2 padding-2 convolution with ALL FOUR phases in one launch: the kernel's chunk sequence runs through the phases
 * (input channels of phase (0,0), then (0,1), (1,0), (1,1)), so there is one output transform and no read-modify-write of the
 * output.  desc: KS = 5, stride = 2, Cin = the input's channels, bias-only epilogue; `packed_w` = l3c_conv_wino4_pack_weights(w_cat,
 * Cout, 4 * Cin) of the four phase kernels concatenated along the INPUT-channel axis, phase-major: w_cat[:, (2a+b) Cin + ci] = w_ab[:, ci].
 */
int l3c_conv_wino4_stride2(const l3c_conv_desc *desc_host, l3c_stream_t stream);
int l3c_conv_wino4_set_tiles_per_block(int n);
/*
 * Pointwise (KS == 1, stride 1) convolution Cin -> Cout <= 160 as a pixel x channel GEMM on the fp32 MFMA: the 192 -> Kp layer that
 * ends every probability classifier (reference prob_clf.py:71-74).  `packed_w` must come from l3c_conv_pw_pack_weights
 * (l3c_conv_pw_packed_words(Cout, Cin) floats).  Cin % 64 == 0; bias only (epilogue == 0); input / weights 16-byte aligned and
 * input channel stride / offset multiples of 4.  Differs from l3c_conv_mfma by the order of the Cin-term sums only.
 */
int64_t l3c_conv_pw_packed_words(int Cout, int Cin);
int l3c_conv_pw_pack_weights(const float *w_oi, int Cout, int Cin, float *packed, l3c_stream_t stream);
int l3c_conv_pw(const l3c_conv_desc *desc_host, l3c_stream_t stream);
/* Same contract on plain VALU FMAs with unpacked OIHW weights (packed_w = w_oihw): a device-side cross-check. */
int l3c_conv_direct(const l3c_conv_desc *desc_host, l3c_stream_t stream);

/*
 * RGB head: img planar fp32 [B][3][H][W] (0..255) -> sub_rgb_mean (1x1, 3->3) -> MeanShift(1/128) (1x1, 3->3) ->
 * conv 3x3 3->Cf, zero padding applied to the mean-shifted image.  (multiscale_network.py:241, head.py:26-41)
 *   w1,b1  sub_rgb_mean [3][3],[3];  w2,b2 heads.0.head.0 [3][3],[3];  w3 [Cf][3][3][3], b3 [Cf];  out pixel-major
 * shifted_out (may be NULL): planar [B][3][H][W], the input of the 3x3 conv, for stage-wise parity tests.
 */
int l3c_rgb_head(const float *img, const float *w1, const float *b1, const float *w2, const float *b2,
                 const float *w3, const float *b3, int B, int H, int W, int Cf, float *out, float *shifted_out,
                 l3c_stream_t stream);

/*
 * Encoder output: 1x1 conv Cf -> C (`to_q`, net.py:113-121) fused with the hard quantiser (quantizer.py:72-87:
 * symbols = argmin_l (x - levels_l)^2, first minimum wins; x_hard = levels[symbols]).
 *   feat pixel-major [B][HW][Cf]; w [C][Cf], b [C], levels [L]
 *   sym int16 planar [B][C][HW]; bn_q fp32 planar [B][C][HW]; bn (may be NULL) pre-quantisation values, planar
 */
int l3c_to_q_quantize(const float *feat, const float *w, const float *b, const float *levels, int64_t B, int64_t HW,
                      int Cf, int C, int L, int16_t *sym, float *bn_q, float *bn, l3c_stream_t stream);

/*
 * Decoder input: 1x1 conv C -> Cf on the quantised bottleneck (+ the coarser decoder's features), net.py:178-180.
 *   bn_q planar [B][C][HW]; w [Cf][C], b [Cf]; fuse pixel-major [B][HW][Cf] or NULL; out pixel-major [B][HW][Cf]
 *   Limits: C <= 8 (the bottleneck's weights live in registers; q.C is 5 or 3 in every shipped config), Cf % 4 == 0 and
 *   256 % (Cf / 4) == 0, HW < 2^31.  Anything else returns L3C_ERR_INVALID_ARG.
 */
int l3c_dec_head(const float *bn_q, const float *w, const float *b, const float *fuse, int64_t B, int64_t HW, int C,
                 int Cf, float *out, l3c_stream_t stream);

/*
 * RGB baselines (BicubicSubsampling encoder, modules/net.py:65-80): the reference leaves the GPU for PIL's BICUBIC resize
 * (dataloaders/images_loader.py:277-288).  Here the pyramid stays on the device and is bit-exact with Pillow:
 *   l3c_meanshift_planar  1x1 conv 3->3 on a planar image (sub_rgb_mean, multiscale_network.py:241); w [3][3], b [3] on device
 *   l3c_rgb_to_u8         uint8(round(clamp(x + mean, 0, 255)))                      planar fp32 [B][3][HW] -> uint8
 *   l3c_resample_u8       ONE pass of Pillow's ImagingResample (8 bpc, 22-bit fixed point) along W (axis 0) or H (axis 1);
 *                         bounds int32 [out_size][2] = (first, count), kk int32 [out_size][ksize], both on the device,
 *                         computed on the host exactly as Pillow does (helpers/pil_resample.py)
 *   l3c_u8_to_sym_bn      symbols (int16) = pixel values, bn = value - mean
 * `mean3_host` is a HOST pointer to the three channel means (0.4488, 0.4371, 0.4040) * 255 as fp32.
 */
int l3c_meanshift_planar(const float *img, const float *w, const float *b, int64_t B, int64_t HW, float *out,
                         l3c_stream_t stream);
int l3c_rgb_to_u8(const float *x, const float *mean3_host, int64_t B, int64_t HW, uint8_t *out, l3c_stream_t stream);
int l3c_resample_u8(const uint8_t *in, int64_t planes, int H, int W, int axis, int out_size, const int32_t *bounds,
                    const int32_t *kk, int ksize, uint8_t *out, l3c_stream_t stream);
int l3c_u8_to_sym_bn(const uint8_t *in, const float *mean3_host, int64_t B, int64_t HW, int16_t *sym, float *bn,
                     l3c_stream_t stream);

/*
 * `.l3c` file assembly on the device (the byte format of bitcoding/bitcoding.py:326-375): writes the B files of a batch back
 * to back into `dst` -- file b at dst + file_offset[b] -- from the coder's per-scale output rows:
 *     u16 x4 padding | for scale = coarsest .. 0:  u8 C, u16 H, u16 W | per channel: u32 nbytes, payload | 46 E2 84 92
 *   scales       coarsest first (file order), HOST array; pointers inside are device pointers
 *   padding      device uint16 [B][4]: left, right, top, bottom          file_offset  device int64 [B]
 * File b is 8 + sum_scales (5 + 4*C + 4) + its payload bytes long; the caller sizes `dst` and scans the offsets.
 */
typedef struct {
    const uint8_t *out;       /* [B*C][stride], as written by l3c_ac_encode(_groups) */
    const uint32_t *nbytes;   /* [B*C] */
    int64_t stride;
    int C, H, W;
} l3c_container_scale;
int l3c_container_write(const l3c_container_scale *scales, int n_scales, int64_t B, const uint16_t *padding,
                        const int64_t *file_offset, uint8_t *dst, l3c_stream_t stream);

/*
 * The inverse for the decoder (round 6): the entropy-coded streams of MANY files, which lie at arbitrary byte offsets inside the files
 * (bitcoding.py:326-375: u32 nbytes + payload per channel), copied out of ONE device buffer holding the raw files into the form the
 * range decoder reads -- every stream 4-byte aligned and followed by at least 4 zero bytes (bits past the end read 0, torchac.cpp:96-128).
 * The host parses only the framing (a few length fields per file) and uploads the files as they are, in one copy.
 *   files        the raw bytes of the files, back to back (device), 4-byte aligned and readable up to the next multiple of 4 of its size
 *   src_offset   int64 [n_streams]: byte position of stream s's payload inside `files`
 *   dst_offset   int64 [n_streams]: byte position of stream s inside `dst`, a multiple of 4; the caller leaves ((nbytes + 3) / 4) * 4 + 4
 *                bytes per stream, all of which are written (payload, then zeros)
 *   nbytes       uint32 [n_streams];   max_nbytes   the largest of them (the host has read every length field: it sizes the launch)
 */
int l3c_container_read(const uint8_t *files, const int64_t *src_offset, const int64_t *dst_offset, const uint32_t *nbytes,
                       int64_t n_streams, uint32_t max_nbytes, uint8_t *dst, l3c_stream_t stream);

/* symbols -> bottleneck values, to_bn (quantizer.py:44-47): float(S) * bin + x_min, two separately rounded fp32 ops. */
int l3c_sym_to_bn(const int16_t *sym, int64_t n, float bin_width, float x_min, float *bn, l3c_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* L3C_HIP_H_ */
