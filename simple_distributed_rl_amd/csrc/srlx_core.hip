// srlx_core.hip -- error reporting and device queries of libsrlx.so
#include "srlx_common.h"

namespace srlx {
static thread_local char g_err[512] = "";
void set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
}  // namespace srlx

extern "C" {

const char *srlx_last_error(void) { return srlx::g_err; }
int srlx_version(void) { return SRLX_VERSION; }

int srlx_device_count(int *out_count) {
    SRLX_REQUIRE(out_count, "device_count: NULL argument");
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) {
        // no device / no driver is a valid answer, not a failure of the query
        (void)hipGetLastError();
        n = 0;
    }
    *out_count = n;
    return SRLX_OK;
}

int srlx_device_info(int device, char *arch_name, int arch_name_len, int *cu_count, int64_t *hbm_bytes) {
    hipDeviceProp_t prop;
    SRLX_HIP(hipGetDeviceProperties(&prop, device));
    if (arch_name && arch_name_len > 0) {
        strncpy(arch_name, prop.gcnArchName, (size_t)arch_name_len - 1);
        arch_name[arch_name_len - 1] = 0;
    }
    if (cu_count) *cu_count = prop.multiProcessorCount;
    if (hbm_bytes) *hbm_bytes = (int64_t)prop.totalGlobalMem;
    return SRLX_OK;
}

}  // extern "C"
