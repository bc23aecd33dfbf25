// l3c_api.hip -- library-level entry points of libl3c_hip.so (version, error text, device query).
#include <string.h>

#include "l3c_common.h"

namespace l3c {
char *error_buffer() {
    static thread_local char buf[512] = {0};
    return buf;
}
}  // namespace l3c

extern "C" {

int l3c_abi_version(void) { return L3C_ABI_VERSION; }
int l3c_bitstream_generation(void) { return L3C_BITSTREAM_GENERATION; }

const char *l3c_last_error(void) { return l3c::error_buffer(); }

int l3c_device_info(char *name_host, int name_cap, int *num_cu_host, char *arch_host, int arch_cap) {
    int dev = 0;
    int rc = l3c::check_hip(hipGetDevice(&dev), "hipGetDevice");
    if (rc != L3C_OK) return rc;
    hipDeviceProp_t prop;
    rc = l3c::check_hip(hipGetDeviceProperties(&prop, dev), "hipGetDeviceProperties");
    if (rc != L3C_OK) return rc;
    if (name_host && name_cap > 0) {
        strncpy(name_host, prop.name, (size_t)name_cap - 1);
        name_host[name_cap - 1] = 0;
    }
    if (arch_host && arch_cap > 0) {
        strncpy(arch_host, prop.gcnArchName, (size_t)arch_cap - 1);
        arch_host[arch_cap - 1] = 0;
    }
    if (num_cu_host) *num_cu_host = prop.multiProcessorCount;
    return L3C_OK;
}

int l3c_stream_create_cu_range(int first_cu, int n_cu, l3c_stream_t *stream_out_host) {
    L3C_REQUIRE(stream_out_host, "null pointer");
    int dev = 0;
    int rc = l3c::check_hip(hipGetDevice(&dev), "hipGetDevice");
    if (rc != L3C_OK) return rc;
    hipDeviceProp_t prop;
    rc = l3c::check_hip(hipGetDeviceProperties(&prop, dev), "hipGetDeviceProperties");
    if (rc != L3C_OK) return rc;
    const int total = prop.multiProcessorCount;
    L3C_REQUIRE(first_cu >= 0 && n_cu > 0 && first_cu + n_cu <= total, "CU range outside the device");
    uint32_t mask[32] = {0};
    L3C_REQUIRE(total <= 32 * 32, "device has more CUs than the mask holds");
    for (int cu = first_cu; cu < first_cu + n_cu; ++cu) mask[cu >> 5] |= 1u << (cu & 31);
    hipStream_t st = nullptr;
    rc = l3c::check_hip(hipExtStreamCreateWithCUMask(&st, (uint32_t)((total + 31) / 32), mask), "hipExtStreamCreateWithCUMask");
    if (rc != L3C_OK) return rc;
    *stream_out_host = reinterpret_cast<l3c_stream_t>(st);
    return L3C_OK;
}

int l3c_stream_create_cu_mask(const uint32_t *mask_host, int n_words, l3c_stream_t *stream_out_host) {
    L3C_REQUIRE(mask_host && stream_out_host && n_words > 0 && n_words <= 32, "bad mask");
    hipStream_t st = nullptr;
    const int rc = l3c::check_hip(hipExtStreamCreateWithCUMask(&st, (uint32_t)n_words, mask_host), "hipExtStreamCreateWithCUMask");
    if (rc != L3C_OK) return rc;
    *stream_out_host = reinterpret_cast<l3c_stream_t>(st);
    return L3C_OK;
}

int l3c_stream_destroy(l3c_stream_t stream) {
    return l3c::check_hip(hipStreamDestroy(l3c::as_stream(stream)), "hipStreamDestroy");
}
}
