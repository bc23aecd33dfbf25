// xcheck_dmll.hip -- TEST-ONLY (libl3c_hip_xcheck.so): sigmoid_sat of csrc/dmll_core.h against the plain statement
// 1 / (1 + expf(-a)) on EVERY float bit pattern.  The product never loads this.
#include "../../include/l3c_xcheck.h"
#include "dmll_core.h"
#include "l3c_common.h"

namespace {

__global__ __launch_bounds__(256) void sigmoid_exhaustive_kernel(unsigned long long *mismatches, uint32_t *first_bad) {
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    unsigned long long bad = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < (1ull << 32); i += stride) {
        const uint32_t bits = (uint32_t)i;
        const float a = __uint_as_float(bits);
        const float want = l3c::sigmoid_f(a), got = l3c::sigmoid_sat(a);
        const bool same = __float_as_uint(want) == __float_as_uint(got) || (want != want && got != got);
        if (!same) {
            ++bad;
            atomicMin(first_bad, bits);
        }
    }
    if (bad) atomicAdd(mismatches, bad);
}

}  // namespace

extern "C" int l3c_xcheck_sigmoid_exhaustive(unsigned long long *mismatches_dev, uint32_t *first_bad_dev, l3c_stream_t stream) {
    L3C_REQUIRE(mismatches_dev && first_bad_dev, "null pointer");
    hipLaunchKernelGGL(sigmoid_exhaustive_kernel, dim3(4096), dim3(256), 0, l3c::as_stream(stream), mismatches_dev, first_bad_dev);
    return l3c::check_launch("sigmoid_exhaustive_kernel");
}
