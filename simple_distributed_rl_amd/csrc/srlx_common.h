// srlx_common.h -- shared host-side plumbing of libsrlx.so (error reporting, scratch arenas).
// gfx950 / ROCm only; there is deliberately no other backend.
#pragma once

#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "srlx.h"

namespace srlx {

void set_error(const char *fmt, ...);

#define SRLX_HIP(expr)                                                                         \
    do {                                                                                       \
        hipError_t _e = (expr);                                                                \
        if (_e != hipSuccess) {                                                                \
            ::srlx::set_error("%s:%d: %s failed: %s", __FILE__, __LINE__, #expr, hipGetErrorString(_e)); \
            return SRLX_ERR_HIP;                                                               \
        }                                                                                      \
    } while (0)

#define SRLX_REQUIRE(cond, ...)              \
    do {                                     \
        if (!(cond)) {                       \
            ::srlx::set_error(__VA_ARGS__);  \
            return SRLX_ERR_INVALID;         \
        }                                    \
    } while (0)

#define SRLX_TRY(expr)                 \
    do {                               \
        int _s = (expr);               \
        if (_s != SRLX_OK) return _s;  \
    } while (0)

// Grow-only device / pinned-host arena.  Growth is a hipMalloc, which is illegal while a
// stream is being captured into a graph: capture users warm up (uncaptured) first.
struct Arena {
    void *ptr = nullptr;
    size_t bytes = 0;
    bool pinned_host = false;

    int reserve(size_t need) {
        if (need <= bytes) return SRLX_OK;
        size_t want = bytes ? bytes : 4096;
        while (want < need) want *= 2;
        void *p = nullptr;
        if (pinned_host) {
            SRLX_HIP(hipHostMalloc(&p, want, hipHostMallocDefault));
            if (ptr) SRLX_HIP(hipHostFree(ptr));
        } else {
            SRLX_HIP(hipMalloc(&p, want));
            if (ptr) SRLX_HIP(hipFree(ptr));  // hipFree synchronises: earlier users are done
        }
        ptr = p;
        bytes = want;
        return SRLX_OK;
    }
    void release() {
        if (!ptr) return;
        if (pinned_host)
            (void)hipHostFree(ptr);
        else
            (void)hipFree(ptr);
        ptr = nullptr;
        bytes = 0;
    }
};

// carve 256-byte aligned pieces out of an arena
struct Carver {
    char *base;
    size_t off = 0;
    explicit Carver(void *p) : base((char *)p) {}
    template <typename T>
    T *take(size_t n) {
        T *r = (T *)(base + off);
        off += (n * sizeof(T) + 255) & ~(size_t)255;
        return r;
    }
    static size_t padded(size_t bytes) { return (bytes + 255) & ~(size_t)255; }
};

// keyed counter RNG of the vectorised path (definition; restated in oracle/hot_path_oracle.py: rng_u64 / u53)
using u64 = unsigned long long;
__host__ __device__ __forceinline__ u64 mix64(u64 z) {
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
__host__ __device__ __forceinline__ u64 rng_u64(u64 seed, u64 a, u64 b) {
    return mix64(mix64(seed + a * 0xD1342543DE82EF95ull) + b * 0xAEF17502108EF2D9ull);
}
__host__ __device__ __forceinline__ double u53(u64 x) { return (double)(x >> 11) * (1.0 / 9007199254740992.0); }

struct DeviceGuard {
    int prev = -1;
    bool ok = false;
    explicit DeviceGuard(int dev) {
        if (hipGetDevice(&prev) == hipSuccess && (prev == dev || hipSetDevice(dev) == hipSuccess)) ok = true;
        if (prev == dev) prev = -1;
    }
    ~DeviceGuard() {
        if (prev >= 0) (void)hipSetDevice(prev);
    }
};

}  // namespace srlx
