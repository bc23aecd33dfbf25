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
}
