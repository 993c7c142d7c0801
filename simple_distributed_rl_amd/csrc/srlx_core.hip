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

namespace {
__global__ void k_stamp(unsigned long long *buf, int i) { buf[i] = wall_clock64(); }
}  // namespace

extern "C" {

// Measurement aid (tools/lockstep_phases.py): a one-thread launch that writes the device's constant-rate wall clock (100 MHz on MI355X) into d_buf[index] --
// a timestamp that survives capture into a HIP graph, where external event records are refused by this HIP runtime.
int srlx_debug_stamp(uint64_t *d_buf, int index, void *stream) {
    SRLX_REQUIRE(d_buf && index >= 0, "debug_stamp: bad argument");
    hipLaunchKernelGGL(k_stamp, dim3(1), dim3(1), 0, (hipStream_t)stream, (unsigned long long *)d_buf, index);
    SRLX_HIP(hipGetLastError());
    return SRLX_OK;
}

// A HIP stream of a given priority LEVEL (-1 high, 0 normal, 1 low; hipStreamNonBlocking).  HIP keeps one pool of hardware queues per level: a stream of its
// own level never shares a hardware queue with the normal-priority internal streams a HIP graph replays its branches on (tools/README.md, finding 5).
int srlx_stream_create(int priority_level, void **out_stream) {
    SRLX_REQUIRE(out_stream && priority_level >= -1 && priority_level <= 1, "stream_create: level -1 / 0 / 1");
    hipStream_t s = nullptr;
    SRLX_HIP(hipStreamCreateWithPriority(&s, hipStreamNonBlocking, priority_level));
    *out_stream = (void *)s;
    return SRLX_OK;
}

int srlx_stream_destroy(void *stream) {
    if (stream) SRLX_HIP(hipStreamDestroy((hipStream_t)stream));
    return SRLX_OK;
}

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
