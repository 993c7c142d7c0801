// srlx_rank.hip -- rank-based prioritised replay: the device side of
// srl/rl/memories/priority_memories/rankbased_memory.py:42-58.  The reference argsorts ALL N priorities on every
// sample() (np.argsort(-priorities[:N]), ~10 ms at N = 1e5 on the host); which RANKS get drawn depends only on N, alpha
// and numpy's generator (np.random.choice over rank probabilities) and stays on the host, bit-identical; the data-dependent
// part -- rank -> buffer index -- is one descending radix sort of (priority, index) pairs in HBM (rocPRIM through
// hipCUB: a library sort, the op is a plain sort) + a gather of the drawn ranks.
#include <hipcub/hipcub.hpp>

#include "srlx_common.h"

struct srlx_rank {
    int device;
    int64_t capacity;
    float *prio;       // [capacity]
    float *keys_out;   // [capacity]
    int *idx_in, *idx_out;
    void *tmp;
    size_t tmp_bytes;
};

namespace {
__global__ void __launch_bounds__(256) k_iota(int *p, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) p[i] = (int)i;
}
__global__ void __launch_bounds__(256) k_scatter_prio(float *prio, int64_t n, const int64_t *idx, const float *val, int64_t start, int64_t cap) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    prio[idx ? idx[i] : (start + i) % cap] = val[i];
}
__global__ void __launch_bounds__(256) k_gather_rank(const int *sorted_idx, int64_t n, const int64_t *ranks, int64_t *out) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) out[i] = sorted_idx[ranks[i]];
}
}  // namespace

extern "C" {

int srlx_rank_create(srlx_rank_t **out, int64_t capacity, int device) {
    SRLX_REQUIRE(out, "rank_create: out is NULL");
    *out = nullptr;
    SRLX_REQUIRE(capacity > 0 && capacity < ((int64_t)1 << 31), "rank_create: capacity out of range");
    int ndev = 0;
    SRLX_HIP(hipGetDeviceCount(&ndev));
    SRLX_REQUIRE(device >= 0 && device < ndev, "rank_create: device %d not present", device);
    srlx::DeviceGuard guard(device);
    srlx_rank *h = new srlx_rank();
    memset(h, 0, sizeof(*h));
    h->device = device;
    h->capacity = capacity;
    hipError_t e = hipcub::DeviceRadixSort::SortPairsDescending(nullptr, h->tmp_bytes, (const float *)nullptr, (float *)nullptr, (const int *)nullptr,
                                                                 (int *)nullptr, (int)capacity);  // size query
    if (e == hipSuccess) e = hipMalloc(&h->prio, sizeof(float) * capacity);
    if (e == hipSuccess) e = hipMalloc(&h->keys_out, sizeof(float) * capacity);
    if (e == hipSuccess) e = hipMalloc(&h->idx_in, sizeof(int) * capacity);
    if (e == hipSuccess) e = hipMalloc(&h->idx_out, sizeof(int) * capacity);
    if (e == hipSuccess) e = hipMalloc(&h->tmp, h->tmp_bytes ? h->tmp_bytes : 256);
    if (e != hipSuccess) {
        srlx::set_error("rank_create: %s", hipGetErrorString(e));
        srlx_rank_destroy(h);
        return e == hipErrorOutOfMemory ? SRLX_ERR_NOMEM : SRLX_ERR_HIP;
    }
    SRLX_HIP(hipMemset(h->prio, 0, sizeof(float) * capacity));
    hipLaunchKernelGGL(k_iota, dim3((unsigned)((capacity + 255) / 256)), dim3(256), 0, nullptr, h->idx_in, capacity);
    SRLX_HIP(hipDeviceSynchronize());
    *out = h;
    return SRLX_OK;
}

int srlx_rank_destroy(srlx_rank_t *h) {
    if (!h) return SRLX_OK;
    srlx::DeviceGuard guard(h->device);
    for (void *p : {(void *)h->prio, (void *)h->keys_out, (void *)h->idx_in, (void *)h->idx_out, h->tmp})
        if (p) (void)hipFree(p);
    delete h;
    return SRLX_OK;
}

/* priorities[idx[i]] = val[i] (update, :60-62) or, idx == NULL, priorities[(start + i) % capacity] = val[i] (add, :33-40) */
int srlx_rank_set(srlx_rank_t *h, int64_t n, const int64_t *d_idx, const float *d_val, int64_t start, void *stream) {
    SRLX_REQUIRE(h && d_val && n > 0 && start >= 0, "rank_set: bad argument");
    srlx::DeviceGuard guard(h->device);
    hipLaunchKernelGGL(k_scatter_prio, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, h->prio, n, d_idx, d_val, start, h->capacity);
    SRLX_HIP(hipGetLastError());
    return SRLX_OK;
}

/* out[i] = index of the ranks[i]-th largest of priorities[0 .. n_live)  (np.argsort(-priorities[:N])[ranks], :47,54) */
int srlx_rank_select(srlx_rank_t *h, int64_t n_live, int64_t n, const int64_t *d_ranks, int64_t *d_out, void *stream) {
    SRLX_REQUIRE(h && d_ranks && d_out && n > 0 && n_live > 0 && n_live <= h->capacity, "rank_select: bad argument");
    srlx::DeviceGuard guard(h->device);
    size_t bytes = h->tmp_bytes;
    SRLX_HIP(hipcub::DeviceRadixSort::SortPairsDescending(h->tmp, bytes, (const float *)h->prio, h->keys_out, (const int *)h->idx_in, h->idx_out, (int)n_live, 0, 32,
                                                          (hipStream_t)stream));
    hipLaunchKernelGGL(k_gather_rank, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (const int *)h->idx_out, n, d_ranks, d_out);
    SRLX_HIP(hipGetLastError());
    return SRLX_OK;
}

int srlx_rank_priorities(srlx_rank_t *h, float **d_prio) {
    SRLX_REQUIRE(h && d_prio, "rank_priorities: NULL argument");
    *d_prio = h->prio;
    return SRLX_OK;
}

}  // extern "C"
