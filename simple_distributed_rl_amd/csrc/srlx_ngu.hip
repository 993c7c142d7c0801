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
    // q_ext == NULL: target_ext / target_int already hold the TD errors (Agent57's sequence means, agent57/model_torch.py:388-391)
    const int a = actions ? actions[b] : 0;
    const float te = q_ext ? target_ext[b] - q_ext[b * A + a] : target_ext[b];  // agent57_light/model_torch.py:442
    float p = te;
    if (target_int) {
        const float ti = q_int ? target_int[b] - q_int[b * A + a] : target_int[b];
        p = te + beta_list[actor_idx[b]] * ti;  // :371-372
        if (td_int_out) td_int_out[b] = ti;
    }
    if (td_ext_out) td_ext_out[b] = te;
    pri[b] = fabsf(p);
}

// ------------------------------------------------------------------------------------------
// Agent57 (R2D2-style) sequence retrace target + Huber loss + gradient seed + TD means in one launch:
// srl/algorithms/agent57/agent57.py:301-379 (calc_target_q) and model_torch.py:469-492 (_train_q after the
// forwards).  One thread per batch row walks its sequence (S <= 1024); float32 arithmetic in numpy's order.
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ float signf(float x) { return x > 0.f ? 1.f : (x < 0.f ? -1.f : 0.f); }
__device__ __forceinline__ float rescaling(float x) { return signf(x) * (f_sqrt(fabsf(x) + 1.0f) - 1.0f) + 0.001f * x; }
__device__ __forceinline__ float inverse_rescaling(float x) {
    float n = f_sqrt(1.0f + (float)(4.0 * 0.001) * ((fabsf(x) + 1.0f) + 0.001f)) - 1.0f;
    n = n / (float)(2.0 * 0.001);
    return signf(x) * ((n * n) - 1.0f);
}

struct SeqArgs {
    i64 B;
    int S, A;
    const float *q, *q_target;  // [B][S+1][A]
    const int32_t *actions;     // [B][S]
    const float *rewards, *dones;  // [B][S]; dones = 0 after a terminal step, else 1
    const u8 *invalid;          // [B][S][A] for the S "next" positions, or NULL
    const float *discounts, *weights;  // [B]
    double retrace_h;
    int double_dqn, rescale;
    float *target;   // [S][B]
    float *loss;     // [1]
    float *grad_q;   // [B][S+1][A]
    float *td_mean;  // [B]  mean_t(action_q - target)
    float *scratch;  // [B][S] gains, then reused
    u8 *pi;          // [B][S]
};

__global__ void __launch_bounds__(256) k_agent57_seq_td(SeqArgs a) {
    __shared__ float red[256];
    const int S = a.S, A = a.A;
    float loss_part = 0.f;
    const float inv_n = 1.0f / (float)(a.B * S);
    for (i64 b = (i64)blockIdx.x * blockDim.x + threadIdx.x; b < a.B; b += (i64)gridDim.x * blockDim.x) {
        const float *q = a.q + b * (S + 1) * A, *qt = a.q_target + b * (S + 1) * A;
        float *gq = a.grad_q + b * (S + 1) * A;
        for (int i = 0; i < (S + 1) * A; i++) gq[i] = 0.f;
        const float disc = a.discounts[b], w = a.weights[b];
        float *gains = a.scratch + b * S;
        u8 *pi = a.pi + b * S;
        for (int t = 0; t < S; t++) {  // agent57.py:317-341
            const float *nq = q + (t + 1) * A, *nqt = qt + (t + 1) * A;
            const u8 *inv = a.invalid ? a.invalid + (b * S + t) * A : nullptr;
            const float *sel = a.double_dqn ? nq : nqt;
            int best = 0;
            float bv = 0.f;
            for (int k = 0; k < A; k++) {
                const float v = (inv && inv[k]) ? -INFINITY : sel[k];
                if (k == 0 || v > bv) {
                    best = k;
                    bv = v;
                }
            }
            float maxq = (!a.double_dqn && inv && inv[best]) ? -INFINITY : nqt[best];
            if (a.rescale) maxq = inverse_rescaling(maxq);
            float g = a.rewards[b * S + t] + (a.dones[b * S + t] * disc) * maxq;
            if (a.rescale) g = rescaling(g);
            gains[t] = g;
            pi[t] = a.actions[b * S + t] == best;  // :351
        }
        // retrace_seq[t] and discounts_seq[t] are needed back to front: precompute the products forward (:355-366)
        // coef[t] = retrace_seq[t] * discounts_seq[t]; stored over the `target` column of this row temporarily
        float retrace = 1.0f, dseq = disc;
        for (int t = 0; t < S; t++) {
            a.target[(i64)t * a.B + b] = retrace * dseq;
            if (t + 1 < S) {
                retrace = (float)((double)retrace * (a.retrace_h * (pi[t] ? 1.0 : 0.0)));  // float32 *= float64 array
                dseq = dseq * disc;
            }
        }
        float next_td = 0.f;
        for (int t = S - 1; t >= 0; t--) {  // :369-376
            const float coef = a.target[(i64)t * a.B + b];
            const float aq = q[t * A + a.actions[b * S + t]];
            const float tq = gains[t] + coef * next_td;
            a.target[(i64)t * a.B + b] = tq;
            next_td = tq - aq;
            // HuberLoss(target*w, action_q*w), delta = 1, mean over S*B (model_torch.py:483)
            const float d = aq * w - tq * w;
            const float ad = fabsf(d);
            loss_part += ad < 1.0f ? 0.5f * d * d : ad - 0.5f;
            const float dl = ad < 1.0f ? d : signf(d);
            gq[t * A + a.actions[b * S + t]] += inv_n * dl * w;
            gains[t] = aq - tq;
        }
        // np.mean(action_q - target, axis=0) of a C-contiguous [S][B] array: rows are added in order t = 0..S-1 (model_torch.py:491)
        float td_sum = 0.f;
        for (int t = 0; t < S; t++) td_sum = td_sum + gains[t];
        a.td_mean[b] = td_sum / (float)S;
    }
    red[threadIdx.x] = loss_part;
    __syncthreads();
    for (int s2 = blockDim.x >> 1; s2 > 0; s2 >>= 1) {
        if (threadIdx.x < s2) red[threadIdx.x] += red[threadIdx.x + s2];
        __syncthreads();
    }
    if (threadIdx.x == 0) atomicAdd(a.loss, red[0] * inv_n);
}


// ------------------------------------------------------------------------------------------------------------------
// Sliding-window UCB meta-controller, one per environment (agent57_light.py:317-353: `_calc_actor_index`).
// Every actor process of the reference keeps ONE controller; E device-resident environments keep E of them: the window of the
// last `window - 1` (arm, episode reward) pairs as a ring, per-arm counts (every arm starts at 1) and reward sums in float64 like
// the Python floats.  One thread per environment; only environments whose episode just ended (`done`) advance.
//   u[e][0] < epsilon  -> a uniformly random arm floor(u[e][1] * N); otherwise the arm of maximal
//   reward/count + beta * sqrt(log(n_recent) / count), ties broken uniformly with u[e][2] (get_random_max_index).
// ------------------------------------------------------------------------------------------------------------------
struct UcbDev {
    i64 E;
    int N, window;
    int32_t *ring_arm;   // [E][window]
    float *ring_reward;  // [E][window]
    int32_t *head, *n_recent, *count;  // [E], [E], [E][N]
    double *sum;         // [E][N]
    int32_t *arm;        // [E] current arm (-1 before the first episode)
};

__global__ void __launch_bounds__(256) k_ucb_step(UcbDev s, const unsigned char *done, const float *episode_reward, const double *u, double epsilon, double beta) {
    const i64 e = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= s.E || (done && !done[e])) return;
    int32_t *cnt = s.count + e * s.N;
    double *sum = s.sum + e * s.N;
    int n = s.n_recent[e];
    const int a_prev = s.arm[e];
    if (a_prev >= 0) {  // the episode that just ended was played by a_prev
        const float r = episode_reward[e];
        const int slot = (s.head[e] + n) % s.window;
        s.ring_arm[e * s.window + slot] = a_prev;
        s.ring_reward[e * s.window + slot] = r;
        n += 1;
        cnt[a_prev] += 1;
        sum[a_prev] += (double)r;
        if (n >= s.window) {  // `if len(recent) >= window: pop(0)`
            const int h0 = s.head[e];
            const int a_old = s.ring_arm[e * s.window + h0];
            cnt[a_old] -= 1;
            sum[a_old] -= (double)s.ring_reward[e * s.window + h0];
            s.head[e] = (h0 + 1) % s.window;
            n -= 1;
        }
        s.n_recent[e] = n;
    }
    int next;
    if (n < s.N) {
        next = n;  // every arm once, in order
    } else if (u[e * 3] < epsilon) {
        next = (int)(u[e * 3 + 1] * s.N);
        next = next >= s.N ? s.N - 1 : next;
    } else {
        const double ln = log((double)n);
        double best = -INFINITY;
        int ties = 0;
        for (int i = 0; i < s.N; i++) {
            const double v = sum[i] / (double)cnt[i] + beta * sqrt(ln / (double)cnt[i]);
            if (v > best) best = v, ties = 1;
            else if (v == best) ties++;
        }
        int pick = (int)(u[e * 3 + 2] * ties);
        pick = pick >= ties ? ties - 1 : pick;
        next = 0;
        for (int i = 0; i < s.N; i++) {
            const double v = sum[i] / (double)cnt[i] + beta * sqrt(ln / (double)cnt[i]);
            if (v == best && pick-- == 0) {
                next = i;
                break;
            }
        }
    }
    s.arm[e] = next;
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
    SRLX_REQUIRE(batch > 0 && n_actions > 0 && d_target_ext && d_priorities && (!d_q_ext || d_actions), "agent57_priority: bad argument");
    SRLX_REQUIRE(!d_target_int || (d_actor_idx && d_beta_list && (!d_q_ext) == (!d_q_int)), "agent57_priority: intrinsic inputs incomplete");
    hipLaunchKernelGGL(k_agent57_priority, dim3((unsigned)((batch + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (i64)batch, n_actions, d_target_ext, d_q_ext,
                       d_target_int, d_q_int, d_actions, d_actor_idx, d_beta_list, d_td_ext, d_td_int, d_priorities);
    SRLX_HIP(hipGetLastError());
    return SRLX_OK;
}

int srlx_agent57_seq_td(int64_t batch, int seq_len, int n_actions, const float *d_q, const float *d_q_target, const int32_t *d_actions,
                        const float *d_rewards, const float *d_dones, const uint8_t *d_invalid_next, const float *d_discounts, const float *d_weights,
                        double retrace_h, int enable_double_dqn, int enable_rescale, float *d_target, float *d_loss, float *d_grad_q, float *d_td_mean,
                        float *d_scratch, void *stream) {
    SRLX_REQUIRE(batch > 0 && seq_len >= 1 && n_actions >= 1, "agent57_seq_td: bad sizes");
    SRLX_REQUIRE(d_q && d_q_target && d_actions && d_rewards && d_dones && d_discounts && d_weights && d_target && d_loss && d_grad_q && d_td_mean && d_scratch,
                 "agent57_seq_td: NULL argument");
    SeqArgs a{};
    a.B = batch;
    a.S = seq_len;
    a.A = n_actions;
    a.q = d_q;
    a.q_target = d_q_target;
    a.actions = d_actions;
    a.rewards = d_rewards;
    a.dones = d_dones;
    a.invalid = d_invalid_next;
    a.discounts = d_discounts;
    a.weights = d_weights;
    a.retrace_h = retrace_h;
    a.double_dqn = enable_double_dqn;
    a.rescale = enable_rescale;
    a.target = d_target;
    a.loss = d_loss;
    a.grad_q = d_grad_q;
    a.td_mean = d_td_mean;
    a.scratch = d_scratch;
    a.pi = (u8 *)(d_scratch + batch * seq_len);
    SRLX_HIP(hipMemsetAsync(d_loss, 0, sizeof(float), (hipStream_t)stream));
    hipLaunchKernelGGL(k_agent57_seq_td, dim3((unsigned)((batch + 255) / 256)), dim3(256), 0, (hipStream_t)stream, a);
    SRLX_HIP(hipGetLastError());
    return SRLX_OK;
}


int srlx_agent57_ucb_step(int64_t n_envs, int n_arms, int window, int32_t *d_ring_arm, float *d_ring_reward, int32_t *d_head, int32_t *d_n_recent,
                          int32_t *d_count, double *d_sum, int32_t *d_arm, const uint8_t *d_done, const float *d_episode_reward, const double *d_u,
                          double epsilon, double beta, void *stream) {
    SRLX_REQUIRE(n_envs > 0 && n_arms > 0 && window > 1 && d_ring_arm && d_ring_reward && d_head && d_n_recent && d_count && d_sum && d_arm && d_episode_reward && d_u,
                 "agent57_ucb_step: bad argument");
    UcbDev s{n_envs, n_arms, window, d_ring_arm, d_ring_reward, d_head, d_n_recent, d_count, d_sum, d_arm};
    hipLaunchKernelGGL(k_ucb_step, dim3((unsigned)((n_envs + 255) / 256)), dim3(256), 0, (hipStream_t)stream, s, d_done, d_episode_reward, d_u, epsilon, beta);
    SRLX_HIP(hipGetLastError());
    return SRLX_OK;
}

}  // extern "C"
