/*
 * l3c_xcheck.h -- C ABI of libl3c_hip_xcheck.so, a TEST-ONLY library (csrc/build.py builds it next to the product library; the
 * product never loads it).  It holds the round-1/2 Winograd F(2x2,3x3) convolution kernel (csrc/conv_wino.hip), kept as an
 * independent second implementation the tests compare the product's F(4x4,3x3) kernel with (tests/test_gpu_conv.py), the exhaustive
 * check of the head's sigmoid (csrc/xcheck_dmll.hip), plus the library-level entry points of l3c_api.hip (error text).  Same conventions as include/l3c_hip.h.
 */
#ifndef L3C_XCHECK_H_
#define L3C_XCHECK_H_

#include "l3c_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/*
 * The l3c_conv_mfma convolution for KS == 3, stride 1, dilation 1, 2 or 4 by Winograd F(2x2, 3x3): 2.25x fewer multiplications, the
 * 16 per-position GEMMs on the fp32 MFMA.  `packed_w` of the descriptor must come from l3c_conv_wino_pack_weights (the
 * transformed weights G g G^T, l3c_conv_wino_packed_words(Cout, Cin) floats).  Differs from l3c_conv_mfma by fp32 rounding.
 * Requirements (L3C_ERR_INVALID_ARG otherwise): Cin % 16 == 0; Cout % 4 == 0 (with PIXEL_SHUFFLE: % 16, dilation 1, no
 * RELU / RESIDUAL); every channel stride / offset a multiple of 4 and every pointer 16-byte aligned (16-byte accesses);
 * one image of each tensor below 2 GB; no epilogue bits other than L3C_EPI_*.
 * l3c_conv_wino_set_tiles_per_block: a block walks up to n horizontally adjacent 4 x 32 output tiles with its load pipeline
 * running through the tile boundaries; 0 (default, or the environment variable L3C_WINO_TPB at load time) picks n per launch
 * from the grid size.  The result does not depend on n bit for bit.  Process-wide; returns the previous value.
 */
int64_t l3c_conv_wino_packed_words(int Cout, int Cin);
int l3c_conv_wino_pack_weights(const float *w_oihw, int Cout, int Cin, float *packed, l3c_stream_t stream);
int l3c_conv_wino(const l3c_conv_desc *desc_host, l3c_stream_t stream);
int l3c_conv_wino_set_tiles_per_block(int n);

/*
 * PROBE (round 6; csrc/conv_wino4w.hip): Winograd F(4x4,3x3) on v_mfma_f32_32x32x2_f32 -- a wavefront owns 9 of the 36 positions for 32
 * tiles x 32 channels, four wavefronts cover the positions, M is exchanged through LDS for the output transform -- built to MEASURE the
 * decomposition the round-5 verdict asked about against the product's conv_wino4_kernel (16x16x4 MFMA, all 36 positions per wavefront,
 * output transform in registers).  3x3, stride 1, dilation 1, Cout <= 64, Cin % 16 == 0, bias (+ L3C_EPI_RELU).
 */
int64_t l3c_conv_wino4w_packed_words(int Cout, int Cin);
int l3c_conv_wino4w_pack_weights(const float *w_oihw, int Cout, int Cin, float *packed, l3c_stream_t stream);
int l3c_conv_wino4w(const l3c_conv_desc *desc_host, int tiles_per_block, l3c_stream_t stream);

/*
 * csrc/dmll_core.h: sigmoid_sat (what the kernels evaluate) against 1 / (1 + expf(-a)) (what it has to equal) on every one of
 * the 2^32 float bit patterns.  *mismatches_dev (uint64, device, zeroed by the caller) += number of differing results,
 * *first_bad_dev (uint32, device, 0xFFFFFFFF from the caller) = smallest differing bit pattern.
 */
int l3c_xcheck_sigmoid_exhaustive(unsigned long long *mismatches_dev, uint32_t *first_bad_dev, l3c_stream_t stream);

#ifdef __cplusplus
}
#endif

#endif
