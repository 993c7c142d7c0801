// srlx_ngu.hip -- Never-Give-Up intrinsic reward on the device:
//   episodic novelty  : srl/algorithms/agent57_light/agent57_light.py:473-513 (the reference walks a Python
//                       deque with one np.linalg.norm per stored embedding, per environment step)
//   lifelong novelty  : agent57_light.py:515-529 (1 + RND mean squared error, clipped to [1, L])
//   mixed priorities  : srl/algorithms/agent57_light/model_torch.py:367-373, 442
//
// Episodic memory layout: one bounded memory per environment, structure-of-arrays [E][D][capacity] float32
// so that thread i reads dimension d of entry i at mem[d*capacity + i] -- every load of a wavefront is one
// contiguous 256-byte run, and each thread accumulates its squared distance over d in index order
// (a fixed, documented order: the reference's order is whatever the BLAS sdot of the host does).
// Bound: HBM, D*4 bytes per live entry per step (128 B at D = 32).
//
// k-nearest selection: every thread keeps its own ascending top-K (K <= 16, in registers, branch-free
// insertion), the workgroup then extracts the k smallest heads with k rounds of a (value, lane) min
// reduction.  Long memories are split over several workgroups per environment; a second kernel merges
// the per-workgroup candidates, evaluates the pseudo-count formula in numpy's float32 evaluation order
// (np.mean / np.sum of a short vector: 8 partial sums, then the tail) and appends the new embedding.
#include "srlx_common.h"

namespace {

using i64 = int64_t;
using u8 = unsigned char;

constexpr int kMaxK = 16;
constexpr int kThreads = 256;
constexpr int kMaxSplit = 16;

__device__ __forceinline__ float f_sqrt(float x) { return (float)__dsqrt_rn((double)x); }  // correctly rounded fp32 sqrt

// numpy's float32 pairwise summation (numpy/_core/src/umath/loops_utils.h.src) over f(0..n): plain loop below 8
// elements, 8 partial sums up to 128, recursive halving above.
template <typename F>
__device__ float np_pairwise_sum(const F &f, i64 lo, i64 n) {
    if (n < 8) {
        float acc = 0.f;  // numpy starts from -0.0; identical for the non-negative inputs used here
        for (i64 i = 0; i < n; i++) acc = acc + f(lo + i);
        return acc;
    }
    if (n <= 128) {
        float r[8];
        for (int j = 0; j < 8; j++) r[j] = f(lo + j);
        i64 i = 8;
        for (; i < n - (n % 8); i += 8)
            for (int j = 0; j < 8; j++) r[j] = r[j] + f(lo + i + j);
        float acc = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
        for (; i < n; i++) acc = acc + f(lo + i);
        return acc;
    }
    i64 n2 = n / 2;
    n2 -= n2 % 8;
    return np_pairwise_sum(f, lo, n2) + np_pairwise_sum(f, lo + n2, n - n2);
}

struct NguDev {
    i64 E, cap;
    int D, k, split;
    float eps, cluster, c, first;
    float *mem;    // [E][D][cap]
    i64 *count;    // [E] embeddings appended since the env's last reset
    float *cand;   // [E][split][kMaxK] per-workgroup candidates (ascending, +inf padded)
};

__device__ __forceinline__ void topk_insert(float (&best)[kMaxK], float d) {
#pragma unroll
    for (int i = 0; i < kMaxK; i++) {
        const float lo = fminf(best[i], d);
        d = fmaxf(best[i], d);
        best[i] = lo;
    }
}

// Workgroup-wide extraction of the k smallest values held in per-thread ascending lists.
// out[0..k) ascending (LDS or global), +inf when fewer than k values exist.
__device__ void wg_select_k(float (&best)[kMaxK], int k, float *out, float *s_val, int *s_lane) {
    const int t = threadIdx.x;
    for (int r = 0; r < k; r++) {
        float v = best[0];
        int who = t;
        for (int off = 32; off > 0; off >>= 1) {
            const float ov = __shfl_xor(v, off);
            const int ow = __shfl_xor(who, off);
            if (ov < v || (ov == v && ow < who)) {
                v = ov;
                who = ow;
            }
        }
        if ((t & 63) == 0) {
            s_val[t >> 6] = v;
            s_lane[t >> 6] = who;
        }
        __syncthreads();
        float bv = s_val[0];
        int bw = s_lane[0];
        for (int w = 1; w < kThreads / 64; w++)
            if (s_val[w] < bv || (s_val[w] == bv && s_lane[w] < bw)) {
                bv = s_val[w];
                bw = s_lane[w];
            }
        if (t == bw) {  // pop the winner's head
#pragma unroll
            for (int i = 0; i + 1 < kMaxK; i++) best[i] = best[i + 1];
            best[kMaxK - 1] = INFINITY;
        }
        if (t == 0) out[r] = bv;
        __syncthreads();
    }
}

// agent57_light.py:493-513 from the ascending k nearest distances (n_near of them are real).
__device__ float pseudo_count_reward(const float *near, int n_near, float eps, float cluster, float c, float *tmp) {
    const float ave = np_pairwise_sum([&](i64 i) { return near[i]; }, 0, n_near) / (float)n_near;  // np.mean
    for (int i = 0; i < n_near; i++) {
        float dn = (ave == 0.0f) ? near[i] : near[i] / ave;
        dn = fmaxf(dn - cluster, 0.f);
        tmp[i] = eps / (dn + eps);
    }
    const float visits = np_pairwise_sum([&](i64 i) { return tmp[i]; }, 0, n_near);  // np.sum
    return 1.0f / (f_sqrt(visits) + c);
}

__device__ void finish_env(const NguDev &a, i64 e, i64 live, const float *near, float *reward, float *tmp) {
    // thread 0 of the finishing workgroup: reward, then append (the oldest entry is overwritten once full,
    // like collections.deque(maxlen), :311,491)
    const int n_near = (int)(live < a.k ? live : a.k);
    reward[e] = live == 0 ? a.first : pseudo_count_reward(near, n_near, a.eps, a.cluster, a.c, tmp);
}

__global__ void __launch_bounds__(kThreads) k_ngu_knn(NguDev a, const float *emb, const u8 *reset, const u8 *active, float *reward) {
    __shared__ float s_q[256];
    __shared__ float s_val[kThreads / 64];
    __shared__ int s_lane[kThreads / 64];
    __shared__ float s_near[kMaxK];
    __shared__ float s_tmp[kMaxK];
    const i64 e = blockIdx.x;
    const int part = blockIdx.y, t = threadIdx.x;
    if (active && !active[e]) return;
    const i64 cnt = (reset && reset[e]) ? 0 : a.count[e];
    const i64 live = cnt < a.cap ? cnt : a.cap;
    for (int d = t; d < a.D; d += kThreads) s_q[d] = emb[e * a.D + d];
    __syncthreads();
    float best[kMaxK];
#pragma unroll
    for (int i = 0; i < kMaxK; i++) best[i] = INFINITY;
    const float *m = a.mem + e * (i64)a.D * a.cap;
    for (i64 i = (i64)part * kThreads + t; i < live; i += (i64)a.split * kThreads) {
        float acc = 0.f;
        for (int d = 0; d < a.D; d++) {
            const float diff = m[(i64)d * a.cap + i] - s_q[d];
            acc = acc + diff * diff;
        }
        const float dist = f_sqrt(acc);
        if (dist < best[kMaxK - 1]) topk_insert(best, dist);
    }
    if (a.split == 1) {
        wg_select_k(best, a.k, s_near, s_val, s_lane);
        if (t == 0) finish_env(a, e, live, s_near, reward, s_tmp);
        __syncthreads();  // every distance of this env has been taken: the new entry may overwrite the oldest
        float *mw = a.mem + e * (i64)a.D * a.cap + (cnt % a.cap);
        for (int d = t; d < a.D; d += kThreads) mw[(i64)d * a.cap] = s_q[d];
        if (t == 0) a.count[e] = cnt + 1;
    } else {
        wg_select_k(best, a.k, a.cand + (e * a.split + part) * kMaxK, s_val, s_lane);
    }
}

// split > 1: merge split*k candidates of one environment (one wavefront per env)
__global__ void __launch_bounds__(64) k_ngu_finish(NguDev a, const float *emb, const u8 *reset, const u8 *active, float *reward) {
    __shared__ float s_c[kMaxSplit * kMaxK];
    __shared__ float s_near[kMaxK];
    __shared__ float s_tmp[kMaxK];
    const i64 e = blockIdx.x;
    const int t = threadIdx.x;
    if (active && !active[e]) return;
    const i64 cnt = (reset && reset[e]) ? 0 : a.count[e];
    const i64 live = cnt < a.cap ? cnt : a.cap;
    const int n = a.split * kMaxK;
    for (int i = t; i < n; i += 64) {
        const int p = i / kMaxK, j = i % kMaxK;
        s_c[i] = j < a.k ? a.cand[(e * a.split + p) * kMaxK + j] : INFINITY;
    }
    __syncthreads();
    for (int r = 0; r < a.k; r++) {
        float v = INFINITY;
        int who = 0x7fffffff;
        for (int i = t; i < n; i += 64)
            if (s_c[i] < v) {
                v = s_c[i];
                who = i;
            }
        for (int off = 32; off > 0; off >>= 1) {
            const float ov = __shfl_xor(v, off);
            const int ow = __shfl_xor(who, off);
            if (ov < v || (ov == v && ow < who)) {
                v = ov;
                who = ow;
            }
        }
        if (t == 0) {
            s_near[r] = v;
            if (who != 0x7fffffff) s_c[who] = INFINITY;
        }
        __syncthreads();
    }
    if (t == 0) finish_env(a, e, live, s_near, reward, s_tmp);
    float *mw = a.mem + e * (i64)a.D * a.cap + (cnt % a.cap);
    for (int d = t; d < a.D; d += 64) mw[(i64)d * a.cap] = emb[e * a.D + d];
    if (t == 0) a.count[e] = cnt + 1;
}

__global__ void __launch_bounds__(256) k_ngu_lifelong(i64 n, int D, const float *target, const float *train, float lmax, float *reward) {
    const i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    // np.square(t - p).mean(): float32 squares added in numpy's pairwise order, then divided by D
    const float *tp = target + i * D, *pp = train + i * D;
    const float sum = np_pairwise_sum(
        [&](i64 j) {
            const float d = tp[j] - pp[j];
            return d * d;
        },
        0, D);
    float r = 1.0f + sum / (float)D;
    if (r < 1.f) r = 1.f;
    if (r > lmax) r = lmax;
    reward[i] = r;
}

__global__ void __launch_bounds__(256) k_agent57_priority(i64 B, int A, const float *target_ext, const float *q_ext, const float *target_int,
                                                          const float *q_int, const int32_t *actions, const int32_t *actor_idx, const float *beta_list,
                                                          float *td_ext_out, float *td_int_out, float *pri) {
    const i64 b = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    const int a = actions[b];
    const float te = target_ext[b] - q_ext[b * A + a];  // model_torch.py:442
    float p = te;
    if (target_int) {
        const float ti = target_int[b] - q_int[b * A + a];
        p = te + beta_list[actor_idx[b]] * ti;  // :371-372
        if (td_int_out) td_int_out[b] = ti;
    }
    if (td_ext_out) td_ext_out[b] = te;
    pri[b] = fabsf(p);
}

}  // namespace

struct srlx_ngu {
    int device;
    NguDev d;
};

extern "C" {

int srlx_ngu_create(srlx_ngu_t **out, int64_t n_envs, int emb_dim, int64_t capacity, int k, double epsilon, double cluster_distance, double pseudo_counts,
                    int device) {
    SRLX_REQUIRE(out, "ngu_create: NULL out");
    *out = nullptr;
    SRLX_REQUIRE(n_envs > 0 && capacity > 0 && emb_dim > 0 && emb_dim <= 256, "ngu_create: need n_envs>0, capacity>0, 0<emb_dim<=256");
    SRLX_REQUIRE(k >= 1 && k <= kMaxK, "ngu_create: episodic_count_max must be in 1..%d (got %d)", kMaxK, k);
    int ndev = 0;
    SRLX_HIP(hipGetDeviceCount(&ndev));
    SRLX_REQUIRE(device >= 0 && device < ndev, "ngu_create: no such device %d", device);
    srlx::DeviceGuard g(device);
    srlx_ngu *h = new srlx_ngu();
    h->device = device;
    NguDev &d = h->d;
    d.E = n_envs;
    d.cap = capacity;
    d.D = emb_dim;
    d.k = k;
    // enough workgroups to fill 256 CUs when there are few environments with long memories
    i64 split = (capacity + 4 * kThreads - 1) / (4 * kThreads);
    const i64 fill = (1024 + n_envs - 1) / n_envs;
    if (split > fill) split = fill;
    if (split > kMaxSplit) split = kMaxSplit;
    if (split < 1) split = 1;
    d.split = (int)split;
    d.eps = (float)epsilon;
    d.cluster = (float)cluster_distance;
    d.c = (float)pseudo_counts;
    d.first = (float)(1.0 / pseudo_counts);  // :485
    d.mem = nullptr;
    d.count = nullptr;
    d.cand = nullptr;
    hipError_t e1 = hipMalloc(&d.mem, sizeof(float) * (size_t)n_envs * emb_dim * capacity);
    hipError_t e2 = hipMalloc(&d.count, sizeof(i64) * (size_t)n_envs);
    hipError_t e3 = hipMalloc(&d.cand, sizeof(float) * (size_t)n_envs * d.split * kMaxK);
    if (e1 != hipSuccess || e2 != hipSuccess || e3 != hipSuccess) {
        if (d.mem) (void)hipFree(d.mem);
        if (d.count) (void)hipFree(d.count);
        if (d.cand) (void)hipFree(d.cand);
        delete h;
        srlx::set_error("ngu_create: hipMalloc of %lld x %d x %lld floats failed", (long long)n_envs, emb_dim, (long long)capacity);
        return SRLX_ERR_HIP;
    }
    SRLX_HIP(hipMemset(d.count, 0, sizeof(i64) * (size_t)n_envs));
    *out = h;
    return SRLX_OK;
}

int srlx_ngu_destroy(srlx_ngu_t *h) {
    if (!h) return SRLX_OK;
    srlx::DeviceGuard g(h->device);
    (void)hipFree(h->d.mem);
    (void)hipFree(h->d.count);
    (void)hipFree(h->d.cand);
    delete h;
    return SRLX_OK;
}

int srlx_ngu_reset(srlx_ngu_t *h, void *stream) {
    SRLX_REQUIRE(h, "ngu_reset: NULL handle");
    SRLX_HIP(hipMemsetAsync(h->d.count, 0, sizeof(i64) * (size_t)h->d.E, (hipStream_t)stream));
    return SRLX_OK;
}

int srlx_ngu_counts(srlx_ngu_t *h, int64_t **d_counts) {
    SRLX_REQUIRE(h && d_counts, "ngu_counts: NULL argument");
    *d_counts = h->d.count;
    return SRLX_OK;
}

int srlx_ngu_episodic_reward(srlx_ngu_t *h, const float *d_emb, const uint8_t *d_reset, const uint8_t *d_active, float *d_reward, void *stream) {
    SRLX_REQUIRE(h && d_emb && d_reward, "ngu_episodic_reward: NULL argument");
    const NguDev &d = h->d;
    hipLaunchKernelGGL(k_ngu_knn, dim3((unsigned)d.E, (unsigned)d.split), dim3(kThreads), 0, (hipStream_t)stream, d, d_emb, d_reset, d_active, d_reward);
    if (d.split > 1)
        hipLaunchKernelGGL(k_ngu_finish, dim3((unsigned)d.E), dim3(64), 0, (hipStream_t)stream, d, d_emb, d_reset, d_active, d_reward);
    SRLX_HIP(hipGetLastError());
    return SRLX_OK;
}

int srlx_ngu_lifelong_reward(int64_t n, int dim, const float *d_target, const float *d_train, double lifelong_max, float *d_reward, void *stream) {
    SRLX_REQUIRE(n > 0 && dim > 0 && d_target && d_train && d_reward, "ngu_lifelong_reward: bad argument");
    hipLaunchKernelGGL(k_ngu_lifelong, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (i64)n, dim, d_target, d_train, (float)lifelong_max,
                       d_reward);
    SRLX_HIP(hipGetLastError());
    return SRLX_OK;
}

int srlx_agent57_priority(int64_t batch, int n_actions, const float *d_target_ext, const float *d_q_ext, const float *d_target_int, const float *d_q_int,
                          const int32_t *d_actions, const int32_t *d_actor_idx, const float *d_beta_list, float *d_td_ext, float *d_td_int, float *d_priorities,
                          void *stream) {
    SRLX_REQUIRE(batch > 0 && n_actions > 0 && d_target_ext && d_q_ext && d_actions && d_priorities, "agent57_priority: bad argument");
    SRLX_REQUIRE(!d_target_int || (d_q_int && d_actor_idx && d_beta_list), "agent57_priority: intrinsic inputs incomplete");
    hipLaunchKernelGGL(k_agent57_priority, dim3((unsigned)((batch + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (i64)batch, n_actions, d_target_ext, d_q_ext,
                       d_target_int, d_q_int, d_actions, d_actor_idx, d_beta_list, d_td_ext, d_td_int, d_priorities);
    SRLX_HIP(hipGetLastError());
    return SRLX_OK;
}

}  // extern "C"
