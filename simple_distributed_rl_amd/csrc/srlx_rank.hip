// srlx_rank.hip -- rank-based prioritised replay: the device side of
// srl/rl/memories/priority_memories/rankbased_memory.py:42-58.  The reference argsorts ALL N priorities on every
// sample() (np.argsort(-priorities[:N]), ~10 ms at N = 1e5 on the host); which RANKS get drawn depends only on N, alpha
// and numpy's generator (np.random.choice over rank probabilities) and stays on the host, bit-identical; the data-dependent
// part -- rank -> buffer index -- is one descending radix sort of (priority, index) pairs in HBM (hand-written below) + a gather of the
// drawn ranks.
#include "srlx_common.h"

// ---- a stable LSD radix sort of (float32 key, int32 index) pairs, descending, written for gfx950 ---------------------------------
// Four passes over 8-bit digits of the order-preserving integer image of the key.  Per pass:
//   k_rs_hist     every workgroup counts the digits of its tile (kTile keys, four waves x contiguous chunks) -> hist[digit][workgroup]
//   k_rs_scan     one workgroup turns hist into exclusive offsets in (digit, workgroup) order: where each workgroup's keys of each digit start
//   k_rs_scatter  every wave re-walks its chunk 64 keys at a time IN ORDER; a key's slot = the scanned base of (digit, workgroup) + keys of that digit
//                 in earlier waves of the tile + in earlier 64-key groups of this wave (an LDS counter per wave and digit) + among lower lanes of the
//                 group (eight ballots give the lanes holding the same digit) -- so equal keys keep their input order (ties end up in index order,
//                 which is what srlx_rank_select documents)
// ~16 bytes moved per key and pass; at the memory's sizes (1e5 .. 1e6 keys) a whole sort is a few tens of microseconds and not on the
// Rainbow hot path (rank-based replay is SURVEY 8 f3, "next").
namespace {
using u32 = unsigned int;
constexpr int kTile = 4096, kWaves = 4, kChunk = kTile / kWaves;  // keys per workgroup / waves per workgroup / keys per wave

__device__ __forceinline__ u32 desc_image(float f) {  // larger float -> smaller integer
    const u32 u = __float_as_uint(f);
    const u32 asc = (u & 0x80000000u) ? ~u : (u | 0x80000000u);
    return ~asc;
}

__global__ void __launch_bounds__(256) k_rs_prepare(const float *__restrict__ prio, int64_t n, u32 *__restrict__ keys, int *__restrict__ idx) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) keys[i] = desc_image(prio[i]), idx[i] = (int)i;
}

__global__ void __launch_bounds__(256) k_rs_hist(const u32 *__restrict__ keys, int64_t n, int shift, u32 *__restrict__ hist, int nblocks) {
    __shared__ u32 cnt[256];
    cnt[threadIdx.x] = 0;
    __syncthreads();
    const int64_t base = (int64_t)blockIdx.x * kTile;
    for (int j = threadIdx.x; j < kTile; j += 256)
        if (base + j < n) atomicAdd(&cnt[(keys[base + j] >> shift) & 255u], 1u);
    __syncthreads();
    hist[(int64_t)threadIdx.x * nblocks + blockIdx.x] = cnt[threadIdx.x];
}

__global__ void __launch_bounds__(1024) k_rs_scan(u32 *__restrict__ hist, int64_t total) {  // exclusive scan in place, one workgroup
    __shared__ u32 wsum[16];
    __shared__ u32 carry;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    if (t == 0) carry = 0;
    __syncthreads();
    for (int64_t c0 = 0; c0 < total; c0 += 1024) {
        const int64_t i = c0 + t;
        const u32 v = i < total ? hist[i] : 0u;
        u32 x = v;
        for (int off = 1; off < 64; off <<= 1) {
            const u32 y = __shfl_up(x, off);
            if (lane >= off) x += y;
        }
        if (lane == 63) wsum[wave] = x;
        __syncthreads();
        u32 wbase = 0;
        for (int w = 0; w < wave; w++) wbase += wsum[w];
        const u32 excl = carry + wbase + x - v;
        if (i < total) hist[i] = excl;
        __syncthreads();
        if (t == 1023) carry = excl + v;
        __syncthreads();
    }
}

__global__ void __launch_bounds__(256) k_rs_scatter(const u32 *__restrict__ keys, const int *__restrict__ idx, int64_t n, int shift, const u32 *__restrict__ offs,
                                                    int nblocks, u32 *__restrict__ keys_out, int *__restrict__ idx_out) {
    __shared__ u32 wcnt[kWaves][256];  // phase A: digits per wave chunk; phase B: running slot of (wave, digit)
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    for (int j = t; j < kWaves * 256; j += 256) (&wcnt[0][0])[j] = 0;
    __syncthreads();
    const int64_t base = (int64_t)blockIdx.x * kTile + (int64_t)wave * kChunk;
    for (int j = lane; j < kChunk; j += 64)
        if (base + j < n) atomicAdd(&wcnt[wave][(keys[base + j] >> shift) & 255u], 1u);
    __syncthreads();
    {   // digit d = t: global start of this workgroup's keys of digit d, then the waves' chunks one after the other
        u32 run = offs[(int64_t)t * nblocks + blockIdx.x];
        for (int w = 0; w < kWaves; w++) {
            const u32 c = wcnt[w][t];
            wcnt[w][t] = run;
            run += c;
        }
    }
    __syncthreads();
    for (int j0 = 0; j0 < kChunk; j0 += 64) {
        const int64_t i = base + j0 + lane;
        const bool live = i < n;
        const u32 k = live ? keys[i] : 0u;
        const u32 d = (k >> shift) & 255u;
        unsigned long long peers = __ballot(live);
#pragma unroll
        for (int bit = 0; bit < 8; bit++) {
            const unsigned long long m = __ballot((d >> bit) & 1u);
            peers &= ((d >> bit) & 1u) ? m : ~m;
        }
        if (live) {
            const u32 before = (u32)__popcll(peers & ((1ull << lane) - 1ull));
            const u32 slot = wcnt[wave][d] + before;
            keys_out[slot] = k;
            idx_out[slot] = idx[i];
        }
        // the highest lane of every digit group advances that digit's running slot (one writer per digit: no race; the wave runs in lock step)
        if (live && (peers >> lane) == 1ull) wcnt[wave][d] += (u32)__popcll(peers);
    }
}
}  // namespace

struct srlx_rank {
    int device;
    int64_t capacity;
    float *prio;            // [capacity]
    u32 *keys[2];           // ping-pong key images
    int *idx[2];            // ping-pong indices; after the four passes the sorted order is in idx[0]
    u32 *hist;              // [256][workgroups]
    int nblocks_max;
};

namespace {
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
    h->nblocks_max = (int)((capacity + kTile - 1) / kTile);
    hipError_t e = hipMalloc(&h->prio, sizeof(float) * capacity);
    for (int k = 0; k < 2 && e == hipSuccess; k++) {
        e = hipMalloc(&h->keys[k], sizeof(u32) * capacity);
        if (e == hipSuccess) e = hipMalloc(&h->idx[k], sizeof(int) * capacity);
    }
    if (e == hipSuccess) e = hipMalloc(&h->hist, sizeof(u32) * 256 * (size_t)h->nblocks_max);
    if (e != hipSuccess) {
        srlx::set_error("rank_create: %s", hipGetErrorString(e));
        srlx_rank_destroy(h);
        return e == hipErrorOutOfMemory ? SRLX_ERR_NOMEM : SRLX_ERR_HIP;
    }
    SRLX_HIP(hipMemset(h->prio, 0, sizeof(float) * capacity));
    SRLX_HIP(hipDeviceSynchronize());
    *out = h;
    return SRLX_OK;
}

int srlx_rank_destroy(srlx_rank_t *h) {
    if (!h) return SRLX_OK;
    srlx::DeviceGuard guard(h->device);
    for (void *p : {(void *)h->prio, (void *)h->keys[0], (void *)h->keys[1], (void *)h->idx[0], (void *)h->idx[1], (void *)h->hist})
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
    hipStream_t st = (hipStream_t)stream;
    const int nblocks = (int)((n_live + kTile - 1) / kTile);
    hipLaunchKernelGGL(k_rs_prepare, dim3((unsigned)((n_live + 255) / 256)), dim3(256), 0, st, h->prio, n_live, h->keys[0], h->idx[0]);
    for (int pass = 0; pass < 4; pass++) {  // four 8-bit digits, least significant first; an even number of passes ends in buffer 0
        const int a = pass & 1, b = a ^ 1, shift = 8 * pass;
        hipLaunchKernelGGL(k_rs_hist, dim3(nblocks), dim3(256), 0, st, h->keys[a], n_live, shift, h->hist, nblocks);
        hipLaunchKernelGGL(k_rs_scan, dim3(1), dim3(1024), 0, st, h->hist, (int64_t)256 * nblocks);
        hipLaunchKernelGGL(k_rs_scatter, dim3(nblocks), dim3(256), 0, st, h->keys[a], h->idx[a], n_live, shift, h->hist, nblocks, h->keys[b], h->idx[b]);
    }
    hipLaunchKernelGGL(k_gather_rank, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, (const int *)h->idx[0], n, d_ranks, d_out);
    SRLX_HIP(hipGetLastError());
    return SRLX_OK;
}

int srlx_rank_priorities(srlx_rank_t *h, float **d_prio) {
    SRLX_REQUIRE(h && d_prio, "rank_priorities: NULL argument");
    *d_prio = h->prio;
    return SRLX_OK;
}

}  // extern "C"
