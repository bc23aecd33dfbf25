// container.hip -- `.l3c` file assembly on the GPU (replaces the host-side byte shuffling of bitcoding.py:326-375 for batches).
//
// The range coder leaves every stream's bytes in its own row of a per-scale buffer.  A file is
//     u16 x4 padding | for scale = coarsest .. 0:  u8 C, u16 H, u16 W | per channel: u32 nbytes, payload | magic 46 E2 84 92
// (all little-endian).  Given the per-image file offsets (an exclusive scan of the file sizes, done by the caller), ONE launch
// writes all files of a batch back to back into one buffer: block (stream j, image b) finds its place from the byte counts of
// the streams before it, writes its length field (+ the scale header / magic / padding header it is next to) and copies its
// payload -- dword-wise, re-aligned with v_alignbyte_b32, since file offsets are byte-granular.  One D2H copy of exactly the
// files' bytes then replaces the per-scale padded copies and the host-side joins.
#include "l3c_common.h"

namespace {

struct ScaleDesc {
    const uint8_t *out;        // [B * C][stride] coder output rows (4-byte aligned rows)
    const uint32_t *nbytes;    // [B * C]
    int64_t stride;
    int C, H, W;
};

struct ContainerArgs {
    static constexpr int MAX_SCALES = 8;
    ScaleDesc scale[MAX_SCALES];   // coarsest first: file order
    int n_scales;
    int streams_per_image;
    const uint16_t *padding;       // [B][4] left, right, top, bottom
    const int64_t *file_offset;    // [B]
    uint8_t *dst;
};

__device__ __forceinline__ void put_bytes(uint8_t *p, uint32_t v, int n) {
    for (int i = 0; i < n; ++i) p[i] = (uint8_t)(v >> (8 * i));
}

__global__ __launch_bounds__(256) void container_write_kernel(const ContainerArgs a) {
    const int64_t b = blockIdx.y;
    int j = blockIdx.x;            // stream of this image, file order
    // locate (scale k, channel c) and the byte position of this stream's length field inside the file
    int64_t pos = 8;
    int k = 0;
    for (; k < a.n_scales; ++k) {
        const ScaleDesc &s = a.scale[k];
        if (j < s.C) break;
        pos += 5 + 4;
        for (int c = 0; c < s.C; ++c) pos += 4 + (int64_t)s.nbytes[b * s.C + c];
        j -= s.C;
    }
    const ScaleDesc &s = a.scale[k];
    const int c = j;
    pos += 5;
    for (int cc = 0; cc < c; ++cc) pos += 4 + (int64_t)s.nbytes[b * s.C + cc];
    const uint32_t n = s.nbytes[b * s.C + c];
    uint8_t *file = a.dst + a.file_offset[b];
    if (threadIdx.x == 0) {
        if (k == 0 && c == 0)
            for (int i = 0; i < 4; ++i) put_bytes(file + 2 * i, a.padding[b * 4 + i], 2);
        if (c == 0) {
            file[pos - 5] = (uint8_t)s.C;
            put_bytes(file + pos - 4, (uint32_t)s.H, 2);
            put_bytes(file + pos - 2, (uint32_t)s.W, 2);
        }
        put_bytes(file + pos, n, 4);
        if (c == s.C - 1) put_bytes(file + pos + 4 + n, 0x9284E246u, 4);   // 46 E2 84 92
    }
    // payload: src rows are 4-byte aligned, the destination is wherever the bytes before it ended
    const uint8_t *src = s.out + (b * s.C + c) * s.stride;
    uint8_t *dst = file + pos + 4;
    const uint32_t head = (uint32_t)((4 - (reinterpret_cast<uintptr_t>(dst) & 3)) & 3);   // bytes up to the first aligned dword
    const uint32_t h = head < n ? head : n;
    if (threadIdx.x < h) dst[threadIdx.x] = src[threadIdx.x];
    const uint32_t body = (n - h) / 4;                        // aligned destination dwords
    const uint32_t *src_w = reinterpret_cast<const uint32_t *>(src);
    uint32_t *dst_w = reinterpret_cast<uint32_t *>(dst + h);
    const uint32_t last_word = (n + 3) / 4;                   // source words holding valid bytes: [0, last_word)
    for (uint32_t i = threadIdx.x; i < body; i += blockDim.x) {
        // destination dword i holds source bytes h + 4 i .. h + 4 i + 3
        const uint32_t q = (h + 4 * i) >> 2;
        const uint32_t lo = src_w[q];
        const uint32_t hi = (q + 1 < last_word) ? src_w[q + 1] : 0u;
        dst_w[i] = h ? __builtin_amdgcn_alignbyte(hi, lo, h) : lo;
    }
    const uint32_t tail0 = h + 4 * body;
    if (threadIdx.x < n - tail0) dst[tail0 + threadIdx.x] = src[tail0 + threadIdx.x];
}

// one block per (stream, 16 KB slice of its padded length): output dword i = source bytes 4 i .. 4 i + 3 (zero beyond the payload)
__global__ __launch_bounds__(256) void container_read_kernel(const uint8_t *__restrict__ files, const int64_t *__restrict__ src_offset,
                                                             const int64_t *__restrict__ dst_offset, const uint32_t *__restrict__ nbytes,
                                                             uint8_t *__restrict__ dst) {
    const int64_t s = blockIdx.y;
    const uint32_t n = nbytes[s];
    const uint32_t words = (n + 3) / 4 + 1;                  // the padded stream: payload, zeros up to a dword, one zero dword
    const uint8_t *src = files + src_offset[s];
    uint32_t *out = reinterpret_cast<uint32_t *>(dst + dst_offset[s]);
    const uint32_t mis = (uint32_t)(reinterpret_cast<uintptr_t>(src) & 3);
    const uint32_t *src_w = reinterpret_cast<const uint32_t *>(src - mis);   // aligned words around the payload (inside the file buffer)
    const uint32_t last = (mis + n + 3) / 4;                 // aligned source words holding payload bytes: [0, last)
    for (uint32_t i = blockIdx.x * 4096u + threadIdx.x; i < words && i < (blockIdx.x + 1) * 4096u; i += 256u) {
        uint32_t v = 0;
        if (4 * i < n) {
            const uint32_t lo = src_w[i];
            const uint32_t hi = (i + 1 < last) ? src_w[i + 1] : 0u;
            v = mis ? __builtin_amdgcn_alignbyte(hi, lo, mis) : lo;
            const uint32_t left = n - 4 * i;                 // payload bytes in this dword (1..4 of them when left < 4)
            if (left < 4) v &= (1u << (8 * left)) - 1u;
        }
        out[i] = v;
    }
}

}  // namespace

extern "C" {

int l3c_container_read(const uint8_t *files, const int64_t *src_offset, const int64_t *dst_offset, const uint32_t *nbytes,
                       int64_t n_streams, uint32_t max_nbytes, uint8_t *dst, l3c_stream_t stream) {
    L3C_REQUIRE(files && src_offset && dst_offset && nbytes && dst, "null pointer");
    L3C_REQUIRE(n_streams > 0 && n_streams < 65536, "1..65535 streams per call");
    L3C_REQUIRE((reinterpret_cast<uintptr_t>(files) & 3) == 0 && (reinterpret_cast<uintptr_t>(dst) & 3) == 0, "buffers must be 4-byte aligned");
    // grid.x covers the longest stream (the host has read every length field); a block whose slice lies beyond its own stream does nothing
    const unsigned slices = (unsigned)(((uint64_t)max_nbytes + 3) / 4 + 1 + 4095) / 4096;
    hipLaunchKernelGGL(container_read_kernel, dim3(slices, (unsigned)n_streams), dim3(256), 0, l3c::as_stream(stream), files, src_offset,
                       dst_offset, nbytes, dst);
    return l3c::check_launch("container_read_kernel");
}

int l3c_container_write(const l3c_container_scale *scales, int n_scales, int64_t B, const uint16_t *padding,
                        const int64_t *file_offset, uint8_t *dst, l3c_stream_t stream) {
    L3C_REQUIRE(scales && padding && file_offset && dst, "null pointer");
    L3C_REQUIRE(n_scales > 0 && n_scales <= ContainerArgs::MAX_SCALES, "1..8 scales");
    L3C_REQUIRE(B > 0 && B < 65536, "bad batch size");
    ContainerArgs a{};
    a.n_scales = n_scales;
    a.padding = padding;
    a.file_offset = file_offset;
    a.dst = dst;
    for (int k = 0; k < n_scales; ++k) {
        L3C_REQUIRE(scales[k].out && scales[k].nbytes && scales[k].C > 0 && scales[k].C < 256, "bad scale descriptor");
        L3C_REQUIRE(scales[k].H > 0 && scales[k].H < 65536 && scales[k].W > 0 && scales[k].W < 65536, "scale shape does not fit u16");
        L3C_REQUIRE(scales[k].stride % 4 == 0 && (reinterpret_cast<uintptr_t>(scales[k].out) & 3) == 0, "stream rows must be 4-byte aligned");
        a.scale[k] = ScaleDesc{scales[k].out, scales[k].nbytes, scales[k].stride, scales[k].C, scales[k].H, scales[k].W};
        a.streams_per_image += scales[k].C;
    }
    hipLaunchKernelGGL(container_write_kernel, dim3((unsigned)a.streams_per_image, (unsigned)B), dim3(256), 0,
                       l3c::as_stream(stream), a);
    return l3c::check_launch("container_write_kernel");
}
}
