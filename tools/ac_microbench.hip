// tools/ac_microbench.hip -- development probe: where does the lane-per-stream encoder spend its time?
// hipcc --offload-arch=gfx950 -O3 -std=c++17 -I l3c-pytorch_amd/csrc tools/ac_microbench.hip -o /tmp/ac_microbench
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#include "ac_core.h"

struct LaneStore {
    uint32_t *words;
    bool active;
    __device__ __forceinline__ void operator()(uint32_t i, uint32_t w) const { if (active) words[i] = w; }
};
struct NullStore {
    uint32_t *sinkhole;
    __device__ __forceinline__ void operator()(uint32_t i, uint32_t w) const { if (w == 0x12345u && i == 77777777u) *sinkhole = w; }
};

template <int MODE, int LEAN>   // 0: full, 1: no stores, 2: no loads (intervals synthesised from a register LCG), 3: no loads no stores
__global__ __launch_bounds__(64) void enc(const uint32_t *__restrict__ iv, int64_t S, int64_t N, uint8_t *out, int64_t stride, uint32_t *nb) {
    int64_t s = (int64_t)blockIdx.x * 64 + threadIdx.x;
    const bool active = s < S;
    if (!active) s = S - 1;
    uint32_t low = 0, high = 0xFFFFFFFFu, pending = 0;
    uint32_t lcg = 12345u + (uint32_t)s;
    auto run = [&](auto &sink) {
        const int64_t groups = N / 16;
        uint4 cur[4], nxt[4];
        auto gp = [&](int64_t g) { return reinterpret_cast<const uint4 *>(iv + ((g >> 2) * S + s) * 64 + (g & 3) * 16); };
        if (MODE < 2) { for (int k = 0; k < 4; ++k) cur[k] = gp(0)[k]; }
        for (int64_t g = 0; g < groups; ++g) {
            if (MODE < 2 && g + 1 < groups) { for (int k = 0; k < 4; ++k) nxt[k] = gp(g + 1)[k]; }
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                uint32_t w[4];
                if (MODE < 2) { w[0] = cur[k].x; w[1] = cur[k].y; w[2] = cur[k].z; w[3] = cur[k].w; }
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    if (MODE >= 2) { lcg = lcg * 1664525u + 1013904223u; uint32_t lo = (lcg >> 8) & 0xFF00u; w[j] = lo | ((lo + 255u) << 16); }
                    if (LEAN) { uint32_t rl, rn; l3c::encode_state_step(low, high, w[j], rl, rn); sink.put(l3c::pack_record(rl, rn) & 1u, 1); }
                    else l3c::encode_symbol(low, high, pending, l3c::interval_lo(w[j]), l3c::interval_hi(w[j]), sink);
                }
            }
            if (MODE < 2) { for (int k = 0; k < 4; ++k) cur[k] = nxt[k]; }
        }
        l3c::encode_finish(low, pending, sink);
        uint32_t n = sink.finish();
        if (active) nb[s] = n;
    };
    if (MODE == 0 || MODE == 2) { l3c::WordSink<LaneStore> sink(LaneStore{(uint32_t *)(out + s * stride), active}); run(sink); }
    else { l3c::WordSink<NullStore> sink(NullStore{(uint32_t *)out}); run(sink); }
}

// pure dependent-chain probes: K dependent VALU ops per iteration
__global__ __launch_bounds__(64) void chain(uint32_t *out, int iters) {
    uint32_t a = threadIdx.x, b = 0x9E3779B9u;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int k = 0; k < 16; ++k) { a = (a ^ b) + (a << 3); }
    }
    out[threadIdx.x] = a;
}

int main() {
    const int64_t S = 48, N = 393216;
    const int64_t words = (N / 64) * S * 64;
    std::vector<uint32_t> h(words);
    uint32_t x = 1;
    for (auto &w : h) { x = x * 1664525u + 1013904223u; uint32_t bits = 2 + ((x >> 28) % 13); uint32_t width = 65536u >> bits; uint32_t lo = ((x >> 4) % (65536u - width)); w = lo | ((lo + width - 1u) << 16); }   // 2..14 bits/symbol
    uint32_t *iv; uint8_t *out; uint32_t *nb;
    const int64_t stride = ((2 * N + 19) / 4) * 4 + 8;
    hipMalloc(&iv, words * 4); hipMalloc(&out, S * stride); hipMalloc(&nb, S * 4);
    hipMemcpy(iv, h.data(), words * 4, hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto time = [&](const char *name, auto launch) {
        launch(); hipDeviceSynchronize();
        hipEventRecord(e0); launch(); hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("%-28s %8.2f ms  %7.1f ns/symbol\n", name, ms, ms * 1e6 / N);
    };
    time("literal encode_symbol", [&] { hipLaunchKernelGGL((enc<0, 0>), dim3(1), dim3(64), 0, 0, iv, S, N, out, stride, nb); });
    time("state step only", [&] { hipLaunchKernelGGL((enc<0, 1>), dim3(1), dim3(64), 0, 0, iv, S, N, out, stride, nb); });
    time("state step, no stores", [&] { hipLaunchKernelGGL((enc<1, 1>), dim3(1), dim3(64), 0, 0, iv, S, N, out, stride, nb); });
    time("state step, no loads/stores", [&] { hipLaunchKernelGGL((enc<3, 1>), dim3(1), dim3(64), 0, 0, iv, S, N, out, stride, nb); });
    uint32_t *o; hipMalloc(&o, 256);
    const int iters = 1 << 20;
    hipLaunchKernelGGL(chain, dim3(1), dim3(64), 0, 0, o, iters); hipDeviceSynchronize();
    hipEventRecord(e0); hipLaunchKernelGGL(chain, dim3(1), dim3(64), 0, 0, o, iters); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("dependent VALU chain: %.2f ns per op (2 dependent ops per step: xor, lshl_add)\n", ms * 1e6 / ((double)iters * 16 * 2));
    return 0;
}
