// srlx_ppo_net.hip -- PPO's actor-critic (srl/algorithms/ppo/ppo.py:55-99 with the default blocks: in -> 64 -> 64 -> {64 -> V, 64 -> (loc, log_scale)}) as four kernels:
//   k_ppo_rollout   : the WHOLE rollout of an iteration in one launch -- T steps of E Pendulum-shaped environments: network forward, Normal policy sample +
//                     log-probability (ppo.py:316-339), environment step with auto-reset, the [T][E] buffers, episode returns, V(s_T) and the GAE scan (:389-404).
//                     One workgroup owns 16 environments for all T steps (an environment's steps depend on each other, the environments do not): weights and
//                     activations live in LDS (the 64 x 64 layers as v_mfma_f32_16x16x4_f32 tiles), nothing but the buffers goes to HBM.
//   k_ppo_minibatch : one minibatch of the update in one launch -- gather of the permuted samples, forward, compute_train_loss + gradient seeds (:102-169), the whole
//                     backward pass; every workgroup walks tiles of 64 samples (the three 64 x 64 layers on v_mfma_f32_32x32x2_f32) and keeps ITS sum of the
//                     12 931 parameter gradients in registers; per-workgroup partial gradients go to HBM once.
//   k_ppo_reduce    : partial gradients -> the flat gradient (fixed order: deterministic) + the three reported losses.
//   k_ppo_adam      : global-norm clip (:240-241, torch.nn.utils.clip_grad_norm_) + Adam (torch.optim.Adam) over the flat parameter vector, one workgroup; between
//                     k_ppo_reduce and k_ppo_adam sits the data-parallel job's ONE all-reduce of the flat gradient (device/ppo.py).
// float32 throughout, fmaf accumulation in ascending input order; the per-sample policy / loss / environment arithmetic is srlx_ppo_math.h, shared with the
// one-purpose kernels of srlx_ppo.hip.  Parameters are ONE flat float32 vector in torch's `ActorCritic.parameters()` order (weights [out][in]).
// Bounds: VALU (f32 FMA) -- about 76 kFLOP per sample and update (forward + backward), 25 kFLOP per environment step; HBM traffic is the buffers only.
#include "srlx_adam_math.h"
#include "srlx_common.h"
#include "srlx_ppo_math.h"

namespace {

using i64 = int64_t;
using u8 = unsigned char;
using u64 = unsigned long long;
using srlxp::LossCfg;

constexpr int H = 64;       // width of every hidden layer (the reference's default blocks)
constexpr int OBS_MAX = 8;  // observation dimensions
constexpr int A_MAX = 4;    // action dimensions
constexpr int RE = 16;      // environments per workgroup in the rollout / forward kernels
constexpr int S = 64;       // samples per tile in the minibatch kernel

struct NetOff {
    int w1, b1, w2, b2, wv, bv, wvo, bvo, wp, bp, wloc, bloc, wls, bls, total;
};
__host__ __device__ inline NetOff net_off(int obs, int A) {
    NetOff o;
    int p = 0;
    o.w1 = p, p += H * obs;
    o.b1 = p, p += H;
    o.w2 = p, p += H * H;
    o.b2 = p, p += H;
    o.wv = p, p += H * H;
    o.bv = p, p += H;
    o.wvo = p, p += H;
    o.bvo = p, p += 1;
    o.wp = p, p += H * H;
    o.bp = p, p += H;
    o.wloc = p, p += A * H;
    o.bloc = p, p += A;
    o.wls = p, p += A * H;
    o.bls = p, p += A;
    o.total = p;
    return o;
}

// LDS image of the small tensors (everything but the three 64 x 64 matrices)
struct Small {
    float w1[H * OBS_MAX], b1[H], b2[H], bv[H], bp[H], wvo[H], wloc[A_MAX * H], wls[A_MAX * H], bvo[4], bloc[A_MAX], bls[A_MAX];  // (a multiple of 16 bytes)
};

__device__ __forceinline__ void load_small(Small &sm, const float *__restrict__ p, const NetOff &o, int obs, int A) {
    const int t = threadIdx.x, n = blockDim.x;
    for (int i = t; i < H * obs; i += n) sm.w1[i] = p[o.w1 + i];
    for (int i = t; i < H; i += n) {
        sm.b1[i] = p[o.b1 + i];
        sm.b2[i] = p[o.b2 + i];
        sm.bv[i] = p[o.bv + i];
        sm.bp[i] = p[o.bp + i];
        sm.wvo[i] = p[o.wvo + i];
    }
    for (int i = t; i < A * H; i += n) {
        sm.wloc[i] = p[o.wloc + i];
        sm.wls[i] = p[o.wls + i];
    }
    if (t < A) {
        sm.bloc[t] = p[o.bloc + t];
        sm.bls[t] = p[o.bls + t];
    }
    if (t == 0) sm.bvo[0] = p[o.bvo];
}

// W [j][k] in HBM -> Wt [k][j] in LDS (the forward's operand: a thread reads four consecutive units of one input)
__device__ __forceinline__ void load_transposed(float *__restrict__ wt, const float *__restrict__ w) {
    for (int i = threadIdx.x; i < H * H; i += blockDim.x) {
        const int j = i >> 6, k = i & 63;
        wt[k * H + j] = w[i];
    }
}

__device__ __forceinline__ float4 ld4(const float *p) { return *reinterpret_cast<const float4 *>(p); }
__device__ __forceinline__ void st4(float *p, float4 v) { *reinterpret_cast<float4 *>(p) = v; }
__device__ __forceinline__ float4 relu4(float4 v) { return make_float4(fmaxf(v.x, 0.f), fmaxf(v.y, 0.f), fmaxf(v.z, 0.f), fmaxf(v.w, 0.f)); }
__device__ __forceinline__ float4 fma4(float s, float4 w, float4 a) { return make_float4(fmaf(s, w.x, a.x), fmaf(s, w.y, a.y), fmaf(s, w.z, a.z), fmaf(s, w.w, a.w)); }

// first layer: out[j0 .. j0 + 3] = b1 + sum_o x[o] * w1[j][o]
__device__ __forceinline__ float4 first_row(const float *__restrict__ x, const Small &sm, int obs, int j0) {
    float acc[4];
#pragma unroll
    for (int i = 0; i < 4; i++) {
        float a = sm.b1[j0 + i];
        for (int o = 0; o < obs; o++) a = fmaf(x[o], sm.w1[(j0 + i) * obs + o], a);
        acc[i] = a;
    }
    return make_float4(acc[0], acc[1], acc[2], acc[3]);
}

__device__ __forceinline__ float dot64(const float *__restrict__ a, const float *__restrict__ b, float bias) {
    float acc = bias;
#pragma unroll 8
    for (int k = 0; k < H; k++) acc = fmaf(a[k], b[k], acc);
    return acc;
}

// The forward of RE = 16 rows held in LDS (x [RE][OBS_MAX]) by 256 threads.  The first layer and the heads on the vector pipe; the three 64 x 64 layers as
// v_mfma_f32_16x16x4_f32 tiles: wave w owns units 16 w .. 16 w + 15 of all 16 rows, 16 chained MFMAs per layer (lane l supplies in[row l & 15][k] and
// Wt[k][unit l & 15] for k = (l >> 4) + 4 q; D[4 (l >> 4) + i][l & 15] = acc[i]) -- a layer is 0.5 k clocks of matrix pipe instead of ~3 k clocks of LDS-latency-
// bound FMAs (one workgroup per CU, one wave per SIMD).  Activation rows are 65 floats long ("lane i reads row i" without bank conflicts).
// heads [RE][1 + 2 A]: v, loc, log_scale.  value_only: the policy branch is skipped (V(s_T)).  Ends behind a barrier.
constexpr int LDF = 65;
typedef float f32x4 __attribute__((ext_vector_type(4)));
struct FwdLds {
    float wt2[H * H], wtv[H * H], wtp[H * H];
    Small sm;
    float x[RE * OBS_MAX], h1[RE * LDF], h2[RE * LDF], hv[RE * LDF], hp[RE * LDF], heads[RE * (1 + 2 * A_MAX)];
};

__device__ __forceinline__ void mfma16_dense(const float *__restrict__ in, const float *__restrict__ wt, const float *__restrict__ bias, float *__restrict__ out) {
    const int lane = threadIdx.x & 63, n0 = (threadIdx.x >> 6) * 16, c = lane & 15, g = lane >> 4;
    const float b = bias[n0 + c];
    f32x4 acc = {b, b, b, b};
#pragma unroll
    for (int q = 0; q < 16; q++) {
        const int k = g + 4 * q;
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(in[c * LDF + k], wt[k * H + n0 + c], acc, 0, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < 4; i++) out[(4 * g + i) * LDF + n0 + c] = fmaxf(acc[i], 0.f);
}

__device__ __forceinline__ void forward_rows(FwdLds &L, int obs, int A, bool value_only) {
    const int tid = threadIdx.x, e = tid >> 4, j0 = (tid & 15) * 4;
    {
        const float4 h = relu4(first_row(L.x + e * OBS_MAX, L.sm, obs, j0));
        float *o = L.h1 + e * LDF + j0;
        o[0] = h.x, o[1] = h.y, o[2] = h.z, o[3] = h.w;
    }
    __syncthreads();
    mfma16_dense(L.h1, L.wt2, L.sm.b2, L.h2);
    __syncthreads();
    mfma16_dense(L.h2, L.wtv, L.sm.bv, L.hv);
    if (!value_only) mfma16_dense(L.h2, L.wtp, L.sm.bp, L.hp);
    __syncthreads();
    const int n_out = 1 + 2 * A;
    if (tid < RE * n_out) {
        const int r = tid % RE, o = tid / RE;  // (consecutive lanes: consecutive rows)
        float v;
        if (o == 0)
            v = dot64(L.hv + r * LDF, L.sm.wvo, L.sm.bvo[0]);
        else if (value_only)
            v = 0.f;
        else if (o <= A)
            v = dot64(L.hp + r * LDF, L.sm.wloc + (o - 1) * H, L.sm.bloc[o - 1]);
        else
            v = dot64(L.hp + r * LDF, L.sm.wls + (o - 1 - A) * H, L.sm.bls[o - 1 - A]);
        L.heads[r * (1 + 2 * A_MAX) + o] = v;
    }
    __syncthreads();
}

__device__ __forceinline__ void load_forward_weights(FwdLds &L, const float *__restrict__ params, const NetOff &o, int obs, int A) {
    load_transposed(L.wt2, params + o.w2);
    load_transposed(L.wtv, params + o.wv);
    load_transposed(L.wtp, params + o.wp);
    load_small(L.sm, params, o, obs, A);
}

// ---- plain forward (evaluation, tests, rollouts of environments other than the built-in one) -----------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_ppo_forward(i64 n, int obs, int A, const float *__restrict__ params, const float *__restrict__ x, float *__restrict__ v, float *__restrict__ loc,
                                                     float *__restrict__ ls) {
    extern __shared__ __align__(16) unsigned char lds_raw[];
    FwdLds &L = *reinterpret_cast<FwdLds *>(lds_raw);
    const NetOff o = net_off(obs, A);
    load_forward_weights(L, params, o, obs, A);
    const int tid = threadIdx.x;
    for (i64 r0 = (i64)blockIdx.x * RE; r0 < n; r0 += (i64)gridDim.x * RE) {
        __syncthreads();
        if (tid < RE * obs) {
            const int r = tid / obs, c = tid % obs;
            L.x[r * OBS_MAX + c] = r0 + r < n ? x[(r0 + r) * obs + c] : 0.f;
        }
        __syncthreads();
        forward_rows(L, obs, A, false);
        if (tid < RE && r0 + tid < n) {
            const float *hd = L.heads + tid * (1 + 2 * A_MAX);
            v[r0 + tid] = hd[0];
            for (int a = 0; a < A; a++) {
                loc[(r0 + tid) * A + a] = hd[1 + a];
                ls[(r0 + tid) * A + a] = hd[1 + A + a];
            }
        }
    }
}

// ---- the rollout -------------------------------------------------------------------------------------------------------------------------------------------------------
struct RolloutArgs {
    i64 E, T;
    int A;
    const float *params;
    float *env_state;  // [E][2] th, thdot
    int32_t *t_in_ep;  // [E]
    float *env_obs;    // [E][3]: the observation the rollout starts from / ends at
    i64 episode_len;
    u64 env_seed, act_seed;
    const i64 *env_counter, *act_counter;  // both advance by T per rollout (k_advance2, behind this kernel)
    float ls_lo, ls_hi;
    double discount, lam;
    float *b_obs /*[T+1][E][3]*/, *b_act /*[T][E][A]*/, *b_logp, *b_val /*[T][E]*/, *b_rew;
    u8 *b_done;
    float *b_adv, *last_v /*[E]*/, *episode_return /*[E]*/, *finished /*[2] sum, count*/;
};

__global__ void __launch_bounds__(256) k_ppo_rollout(RolloutArgs a) {
    extern __shared__ __align__(16) unsigned char lds_raw[];
    FwdLds &L = *reinterpret_cast<FwdLds *>(lds_raw);
    float *t_rew = reinterpret_cast<float *>(lds_raw + sizeof(FwdLds));  // [T][RE]
    float *t_val = t_rew + a.T * RE;
    float *t_done = t_val + a.T * RE;
    float *zbuf = t_done + a.T * RE;  // [T][RE][A]: the policy's standard-normal draws of the whole rollout, made up front by all threads (double-precision log / cos off the step loop)
    const int obs = 3, A = a.A, tid = threadIdx.x;
    const NetOff o = net_off(obs, A);
    load_forward_weights(L, a.params, o, obs, A);
    const i64 e0 = (i64)blockIdx.x * RE, eg = e0 + tid;  // (tid < RE: this thread's environment)
    float th = 0.f, thd = 0.f, er = 0.f, fin_sum = 0.f, fin_cnt = 0.f;
    int tstep = 0;
    if (tid < RE) {
        th = a.env_state[2 * eg], thd = a.env_state[2 * eg + 1], tstep = a.t_in_ep[eg], er = a.episode_return[eg];
        for (int c = 0; c < 3; c++) {
            const float v = a.env_obs[3 * eg + c];
            L.x[tid * OBS_MAX + c] = v;
            a.b_obs[3 * eg + c] = v;
        }
    }
    const u64 c_act = (u64)a.act_counter[0], c_env = (u64)a.env_counter[0];
    for (i64 w = tid; w < a.T * RE * A; w += 256) {
        const i64 t = w / (RE * A);
        const int rem = (int)(w % (RE * A)), e = rem / A, d = rem % A;
        zbuf[w] = srlxp::normal_z(a.act_seed, c_act + (u64)t, (e0 + e) * A + d);
    }
    __syncthreads();
    for (i64 t = 0; t < a.T; t++) {
        forward_rows(L, obs, A, false);
        if (tid < RE) {
            const float *hd = L.heads + tid * (1 + 2 * A_MAX);
            float act0 = 0.f;
            for (int d = 0; d < A; d++) {
                float ac, lp;
                srlxp::normal_act_from_z(hd[1 + d], hd[1 + A + d], a.ls_lo, a.ls_hi, zbuf[(t * RE + tid) * A + d], 0, ac, lp);
                a.b_act[(t * a.E + eg) * A + d] = ac;
                a.b_logp[(t * a.E + eg) * A + d] = lp;
                if (d == 0) act0 = ac;
            }
            float o0, o1, o2, rw;
            u8 dn;
            srlxp::pendulum_one(th, thd, tstep, act0, a.episode_len, a.env_seed, c_env + (u64)t, eg, o0, o1, o2, rw, dn);
            const i64 k = t * a.E + eg;
            a.b_val[k] = hd[0];
            a.b_rew[k] = rw;
            a.b_done[k] = dn;
            float *ob = a.b_obs + ((t + 1) * a.E + eg) * 3;
            ob[0] = o0, ob[1] = o1, ob[2] = o2;
            L.x[tid * OBS_MAX + 0] = o0, L.x[tid * OBS_MAX + 1] = o1, L.x[tid * OBS_MAX + 2] = o2;
            t_rew[t * RE + tid] = rw, t_val[t * RE + tid] = hd[0], t_done[t * RE + tid] = dn ? 1.f : 0.f;
            er += rw;
            if (dn) fin_sum += er, fin_cnt += 1.f, er = 0.f;
        }
        __syncthreads();
    }
    forward_rows(L, obs, A, true);  // V(s_T): a horizon cut inside an episode bootstraps from it, an episode end never does (ppo.py:396-397)
    if (tid < RE) {
        const float lv = L.heads[tid * (1 + 2 * A_MAX)];
        a.last_v[eg] = lv;
        const float g = (float)a.discount, gl = (float)(a.discount * a.lam);
        float gae = 0.f;
        for (i64 i = a.T - 1; i >= 0; i--) {  // (the arithmetic of k_gae_scan, srlx_train.hip)
            const float r = t_rew[i * RE + tid], v = t_val[i * RE + tid];
            float delta;
            if (t_done[i * RE + tid] != 0.f) {
                delta = r - v;
                gae = 0.f;
            } else if (i == a.T - 1) {
                delta = (r + g * lv) - v;
            } else {
                delta = (r + g * t_val[(i + 1) * RE + tid]) - v;
            }
            gae = delta + gl * gae;
            a.b_adv[i * a.E + eg] = gae;
        }
        a.env_state[2 * eg] = th, a.env_state[2 * eg + 1] = thd, a.t_in_ep[eg] = tstep, a.episode_return[eg] = er;
        for (int c = 0; c < 3; c++) a.env_obs[3 * eg + c] = L.x[tid * OBS_MAX + c];
        if (fin_cnt > 0.f) {
            atomicAdd(&a.finished[0], fin_sum);
            atomicAdd(&a.finished[1], fin_cnt);
        }
    }
}

__global__ void k_advance2(i64 *c0, i64 *c1, i64 n) {
    c0[0] += n;
    c1[0] += n;
}

// ---- one minibatch: forward + loss + backward ----------------------------------------------------------------------------------------------------------------------
struct MbArgs {
    i64 mb;           // samples in this minibatch
    const i64 *perm;  // [mb] rows of the [T * E] buffers
    int obs, A;
    const float *params;
    const float *b_obs, *b_act, *b_logp, *b_adv, *b_vt, *b_val;
    LossCfg cfg;
    float *partials;  // [gridDim.x][stride]: per-workgroup gradient sums (parameter order) + 3 loss sums
    int stride;
};

// The three 64 x 64 layers run on the matrix cores: v_mfma_f32_32x32x2_f32 (f32 in, f32 accumulate: an fmaf chain per output, the f32 MFMA peak equals the vector
// peak -- the gain is that an operand element is read from LDS once per 32 outputs instead of once per 4, and that 32 chained MFMAs keep a SIMD busy where the
// vector loop waited for LDS).  A tile is 64 samples; each of the four waves owns one 32 x 32 block of every 64 x 64 product (forward: [sample][unit], data
// gradient: [sample][input], weight gradient: [unit][input] with the SAMPLES as the K dimension -- accumulated in registers across the workgroup's tiles).
// LDS rows are 65 floats long: "lane i reads row i" and "lane i reads column i" are both conflict-free, so no matrix is kept twice.
constexpr int LD = 65;
static_assert(S == 64 && H == 64 && RE == 16, "mfma_block: K = 64 for every product; mfma16_dense: 16 rows");
constexpr int HS = 12;  // floats per row of the heads / seeds tables (16-byte rows)
constexpr int kVecs = 6 + 2 * A_MAX + OBS_MAX;  // vectors of 64 partial sums a thread row keeps (b1, b2, bv, bp, wvo, head biases, wloc[], wls[], w1[][c])
typedef float f32x16 __attribute__((ext_vector_type(16)));

struct MbLds {
    float w2[H * LD], wv[H * LD], wp[H * LD];  // [unit][input]
    Small sm;
    float x[S * OBS_MAX], h1[S * LD], h2[S * LD], hv[S * LD], hp[S * LD], d2[S * LD];
    float heads[S * HS], seeds[S * HS];
    float red[3 * S];
};

// acc[r] <-> D[(r & 3) + 8 (r >> 2) + 4 h][i] of this lane (i = lane & 31, h = lane >> 5); lane supplies A[i][k], B[k][i] for k = h, h + 2, ...
template <class FA, class FB>
__device__ __forceinline__ f32x16 mfma_block(f32x16 acc, FA a_at, FB b_at) {  // K = 64: 32 chained MFMAs (measured: two interleaved chains, a 16-deep unroll or
    // straight-line code with all operands fetched first are slower or spill -- about 100 clocks per MFMA against the pipe's 64)
    const int lane = threadIdx.x & 63, i = lane & 31, h = lane >> 5;
#pragma unroll 8
    for (int q = 0; q < 32; q++) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a_at(i, h + 2 * q), b_at(h + 2 * q, i), acc, 0, 0, 0);
    return acc;
}
__device__ __forceinline__ int acc_row(int r) { return (r & 3) + 8 * (r >> 2) + 4 * ((threadIdx.x & 63) >> 5); }

// out[s][j] = relu(bias[j] + sum_k in[s][k] W[j][k]) for this wave's block (samples m0.., units n0..)
__device__ __forceinline__ void mfma_dense(const float *__restrict__ in, const float *__restrict__ w, const float *__restrict__ bias, float *__restrict__ out, int m0, int n0) {
    const int i = threadIdx.x & 31;
    f32x16 acc;
    const float b = bias[n0 + i];
#pragma unroll
    for (int r = 0; r < 16; r++) acc[r] = b;
    acc = mfma_block(acc, [&](int row, int k) { return in[(m0 + row) * LD + k]; }, [&](int k, int col) { return w[(n0 + col) * LD + k]; });
#pragma unroll
    for (int r = 0; r < 16; r++) out[(m0 + acc_row(r)) * LD + n0 + i] = fmaxf(acc[r], 0.f);
}

// acc += dz[s][j] W[j][k] for this wave's block (samples m0.., inputs n0..)
__device__ __forceinline__ f32x16 mfma_dgrad(f32x16 acc, const float *__restrict__ dz, const float *__restrict__ w, int m0, int n0) {
    return mfma_block(acc, [&](int row, int j) { return dz[(m0 + row) * LD + j]; }, [&](int j, int col) { return w[j * LD + n0 + col]; });
}

// acc += sum_s dz[s][j] h[s][k] for this wave's block (units m0.., inputs n0..)
__device__ __forceinline__ f32x16 mfma_wgrad(f32x16 acc, const float *__restrict__ dz, const float *__restrict__ h, int m0, int n0) {
    return mfma_block(acc, [&](int row, int s) { return dz[s * LD + m0 + row]; }, [&](int s, int col) { return h[s * LD + n0 + col]; });
}

__global__ void __launch_bounds__(256) k_ppo_minibatch(MbArgs a) {
    extern __shared__ __align__(16) unsigned char lds_raw[];
    MbLds &L = *reinterpret_cast<MbLds *>(lds_raw);
    const int obs = a.obs, A = a.A, tid = threadIdx.x, n_out = 1 + 2 * A;
    const NetOff o = net_off(obs, A);
#pragma unroll 4
    for (int i = tid; i < H * H; i += 256) {
        const int j = i >> 6, k = i & 63;
        L.w2[j * LD + k] = a.params[o.w2 + i];
        L.wv[j * LD + k] = a.params[o.wv + i];
        L.wp[j * LD + k] = a.params[o.wp + i];
    }
    load_small(L.sm, a.params, o, obs, A);
    const int wave = tid >> 6, m0 = (wave >> 1) * 32, n0 = (wave & 1) * 32, li = tid & 31;
    // the matrices: this wave's block of W [m0 + acc_row(r)][n0 + li], summed over the workgroup's tiles
    f32x16 g_w2, g_wv, g_wp;
#pragma unroll
    for (int r = 0; r < 16; r++) g_w2[r] = 0.f, g_wv[r] = 0.f, g_wp[r] = 0.f;
    // the vectors: thread (unit u = tid & 63, part = tid >> 6) works on samples 16 part .. 16 part + 15 of every tile; the four parts meet once, behind the tile loop.
    // Fixed extents (8 observation dimensions, 4 action dimensions, zero-padded): no run-time trip counts in the per-sample loops
    const int u = tid & 63, part = tid >> 6, s_lo = 16 * part;
    float g_w1[OBS_MAX] = {}, g_b1 = 0.f, g_b2 = 0.f, g_bv = 0.f, g_bp = 0.f, g_wvo = 0.f, g_wloc[A_MAX] = {}, g_wls[A_MAX] = {};
    float g_head_b = 0.f;                         // u < 12: the seeds' column u (0: bvo, 1 + d: bloc[d], 5 + d: bls[d])
    float s_pol = 0.f, s_val = 0.f, s_ent = 0.f;  // tid < S: loss sums
    __syncthreads();
    float w1u[OBS_MAX], wloc_u[A_MAX], wls_u[A_MAX];
#pragma unroll
    for (int c = 0; c < OBS_MAX; c++) w1u[c] = c < obs ? L.sm.w1[u * obs + c] : 0.f;
#pragma unroll
    for (int d = 0; d < A_MAX; d++) wloc_u[d] = d < A ? L.sm.wloc[d * H + u] : 0.f, wls_u[d] = d < A ? L.sm.wls[d * H + u] : 0.f;
    const float b1u = L.sm.b1[u], wvo_u = L.sm.wvo[u];
    const i64 tiles = (a.mb + S - 1) / S;
    for (i64 tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
        __syncthreads();
        const i64 row = tile * S + tid;  // (tid < S)
        i64 idx = -1;
        float in_vt = 0.f, in_adv = 0.f, in_val = 0.f, in_act[A_MAX], in_logp[A_MAX];  // this sample's loss inputs: gathered now, used behind the forward
        if (tid < S) {
            idx = row < a.mb ? a.perm[row] : -1;
            for (int c = 0; c < OBS_MAX; c++) L.x[tid * OBS_MAX + c] = (idx >= 0 && c < obs) ? a.b_obs[idx * obs + c] : 0.f;
            if (idx >= 0) {
                in_vt = a.b_vt[idx], in_adv = a.b_adv[idx], in_val = a.b_val[idx];
                for (int d = 0; d < A; d++) in_act[d] = a.b_act[idx * A + d], in_logp[d] = a.b_logp[idx * A + d];
            }
        }
        __syncthreads();
        // ---- forward ----
#pragma unroll 4
        for (int s = s_lo; s < s_lo + 16; s++) {
            const float4 x0 = ld4(L.x + s * OBS_MAX), x1 = ld4(L.x + s * OBS_MAX + 4);
            float acc = b1u;
            acc = fmaf(x0.x, w1u[0], acc), acc = fmaf(x0.y, w1u[1], acc), acc = fmaf(x0.z, w1u[2], acc), acc = fmaf(x0.w, w1u[3], acc);
            acc = fmaf(x1.x, w1u[4], acc), acc = fmaf(x1.y, w1u[5], acc), acc = fmaf(x1.z, w1u[6], acc), acc = fmaf(x1.w, w1u[7], acc);
            L.h1[s * LD + u] = fmaxf(acc, 0.f);
        }
        __syncthreads();
        mfma_dense(L.h1, L.w2, L.sm.b2, L.h2, m0, n0);
        __syncthreads();
        mfma_dense(L.h2, L.wv, L.sm.bv, L.hv, m0, n0);
        mfma_dense(L.h2, L.wp, L.sm.bp, L.hp, m0, n0);
        __syncthreads();
        for (int w = tid; w < S * n_out; w += 256) {
            const int r = w % S, oo = w / S;  // (consecutive lanes: consecutive rows of an LD = 65 matrix: conflict-free)
            const float *hrow = (oo == 0 ? L.hv : L.hp) + r * LD;
            const float *wrow = oo == 0 ? L.sm.wvo : (oo <= A ? L.sm.wloc + (oo - 1) * H : L.sm.wls + (oo - 1 - A) * H);
            L.heads[r * HS + oo] = dot64(hrow, wrow, oo == 0 ? L.sm.bvo[0] : (oo <= A ? L.sm.bloc[oo - 1] : L.sm.bls[oo - 1 - A]));
        }
        __syncthreads();
        // ---- loss + gradient seeds (compute_train_loss, ppo.py:102-169): one thread per sample; seeds row: [0] d/dv, [1 + d] d/dloc, [5 + d] d/dlog_scale ----
        if (tid < S) {
            float sd[HS] = {};
            const float *hd = L.heads + tid * HS;
            if (idx >= 0) {
                const float v = hd[0];
                const float adv = a.cfg.baseline_advantage ? in_adv - v : in_adv;
                float ent = 0.f;
                for (int d = 0; d < A; d++) {
                    float term, e1;
                    srlxp::policy_normal(a.cfg, hd[1 + d], hd[1 + A + d], in_act[d], in_logp[d], adv, term, e1, sd[1 + d], sd[5 + d]);
                    s_pol += term;
                    ent += e1;
                }
                s_ent += ent;
                s_val += srlxp::value_term(a.cfg, v, in_vt, a.cfg.value_clip ? in_val : 0.f, sd[0]);
            }
#pragma unroll
            for (int q = 0; q < HS; q += 4) st4(L.seeds + tid * HS + q, make_float4(sd[q], sd[q + 1], sd[q + 2], sd[q + 3]));
        }
        __syncthreads();
        // ---- the heads' gradients, and the gradients at the value / policy blocks' pre-activations in place of their activations ----
#pragma unroll 4
        for (int s = s_lo; s < s_lo + 16; s++) {
            const float4 q0 = ld4(L.seeds + s * HS), q1 = ld4(L.seeds + s * HS + 4), q2 = ld4(L.seeds + s * HS + 8);
            const float sloc[4] = {q0.y, q0.z, q0.w, q1.x}, sls[4] = {q1.y, q1.z, q1.w, q2.x};
            const float hv = L.hv[s * LD + u], hp = L.hp[s * LD + u];
            g_wvo = fmaf(q0.x, hv, g_wvo);
            float acc = 0.f;
#pragma unroll
            for (int d = 0; d < A_MAX; d++) {
                g_wloc[d] = fmaf(sloc[d], hp, g_wloc[d]);
                g_wls[d] = fmaf(sls[d], hp, g_wls[d]);
                acc = fmaf(sloc[d], wloc_u[d], acc);
                acc = fmaf(sls[d], wls_u[d], acc);
            }
            if (u < HS) g_head_b += L.seeds[s * HS + u];
            const float zv = hv > 0.f ? q0.x * wvo_u : 0.f, zp = hp > 0.f ? acc : 0.f;
            g_bv += zv, g_bp += zp;
            L.hv[s * LD + u] = zv, L.hp[s * LD + u] = zp;  // (this thread read them, this thread replaces them)
        }
        __syncthreads();
        // ---- value / policy blocks: weight gradients, and the gradient at the trunk's second pre-activation ----
        g_wv = mfma_wgrad(g_wv, L.hv, L.h2, m0, n0);
        g_wp = mfma_wgrad(g_wp, L.hp, L.h2, m0, n0);
        {
            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; r++) acc[r] = 0.f;
            acc = mfma_dgrad(acc, L.hv, L.wv, m0, n0);
            acc = mfma_dgrad(acc, L.hp, L.wp, m0, n0);
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const int at = (m0 + acc_row(r)) * LD + n0 + li;
                L.d2[at] = L.h2[at] > 0.f ? acc[r] : 0.f;
            }
        }
        __syncthreads();
        // ---- trunk, second layer: weight gradient; the gradient at the first pre-activation goes where hv was ----
        g_w2 = mfma_wgrad(g_w2, L.d2, L.h1, m0, n0);
        {
            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; r++) acc[r] = 0.f;
            acc = mfma_dgrad(acc, L.d2, L.w2, m0, n0);
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const int at = (m0 + acc_row(r)) * LD + n0 + li;
                L.hv[at] = L.h1[at] > 0.f ? acc[r] : 0.f;
            }
        }
        __syncthreads();
        // ---- trunk, first layer ----
#pragma unroll 4
        for (int s = s_lo; s < s_lo + 16; s++) {
            const float d = L.hv[s * LD + u];
            const float4 x0 = ld4(L.x + s * OBS_MAX), x1 = ld4(L.x + s * OBS_MAX + 4);
            g_b1 += d;
            g_b2 += L.d2[s * LD + u];
            g_w1[0] = fmaf(d, x0.x, g_w1[0]), g_w1[1] = fmaf(d, x0.y, g_w1[1]), g_w1[2] = fmaf(d, x0.z, g_w1[2]), g_w1[3] = fmaf(d, x0.w, g_w1[3]);
            g_w1[4] = fmaf(d, x1.x, g_w1[4]), g_w1[5] = fmaf(d, x1.y, g_w1[5]), g_w1[6] = fmaf(d, x1.z, g_w1[6]), g_w1[7] = fmaf(d, x1.w, g_w1[7]);
        }
    }
    // ---- the four parts of every vector meet (part 0 + 1 + 2 + 3, in that order), through the activations' LDS ----
    __syncthreads();
    {
        static_assert(4 * kVecs * H <= 2 * S * LD, "the partial sums' scratch spans h1 and h2");
        float *sc = L.h1 + part * kVecs * H;  // [part][vector][unit] (h1 and h2 are adjacent and dead by now)
        int q = 0;
        sc[(q++) * H + u] = g_b1, sc[(q++) * H + u] = g_b2, sc[(q++) * H + u] = g_bv, sc[(q++) * H + u] = g_bp, sc[(q++) * H + u] = g_wvo, sc[(q++) * H + u] = g_head_b;
        for (int d = 0; d < A_MAX; d++) sc[(q++) * H + u] = g_wloc[d], sc[(q++) * H + u] = g_wls[d];
        for (int c = 0; c < OBS_MAX; c++) sc[(q++) * H + u] = g_w1[c];
    }
    __syncthreads();
    if (part == 0) {
        const int per = kVecs * H;
        auto total = [&](int q) { return ((L.h1[q * H + u] + L.h1[per + q * H + u]) + L.h1[2 * per + q * H + u]) + L.h1[3 * per + q * H + u]; };
        int q = 0;
        g_b1 = total(q++), g_b2 = total(q++), g_bv = total(q++), g_bp = total(q++), g_wvo = total(q++), g_head_b = total(q++);
        for (int d = 0; d < A_MAX; d++) g_wloc[d] = total(q++), g_wls[d] = total(q++);
        for (int c = 0; c < OBS_MAX; c++) g_w1[c] = total(q++);
    }
    // ---- this workgroup's sums -> HBM ----
    float *out = a.partials + (i64)blockIdx.x * a.stride;
#pragma unroll
    for (int r = 0; r < 16; r++) {
        const int at = (m0 + acc_row(r)) * H + n0 + li;
        out[o.w2 + at] = g_w2[r], out[o.wv + at] = g_wv[r], out[o.wp + at] = g_wp[r];
    }
    if (tid < H) {
        for (int c = 0; c < obs; c++) out[o.w1 + tid * obs + c] = g_w1[c];
        out[o.b1 + tid] = g_b1, out[o.b2 + tid] = g_b2, out[o.bv + tid] = g_bv, out[o.bp + tid] = g_bp, out[o.wvo + tid] = g_wvo;
        for (int d = 0; d < A; d++) out[o.wloc + d * H + tid] = g_wloc[d], out[o.wls + d * H + tid] = g_wls[d];
        if (tid == 0) out[o.bvo] = g_head_b;
        if (tid >= 1 && tid <= A) out[o.bloc + tid - 1] = g_head_b;
        if (tid >= 5 && tid < 5 + A) out[o.bls + tid - 5] = g_head_b;
    }
    __syncthreads();
    if (tid < S) L.red[tid] = s_pol, L.red[S + tid] = s_val, L.red[2 * S + tid] = s_ent;
    __syncthreads();
    if (tid < 3) {
        float acc = 0.f;
        for (int s = 0; s < S; s++) acc += L.red[tid * S + s];
        out[o.total + tid] = acc;
    }
}

// partial[w][p] -> grad[p]: four lanes per parameter, each sums every fourth partial in ascending order, then (s0 + s1) + (s2 + s3) -- a fixed order: deterministic.
// The three loss sums -> the values the reference reports (weighted means).
__global__ void __launch_bounds__(256) k_ppo_reduce(int n_wg, int P, int stride, const float *__restrict__ partials, float *__restrict__ grad, float *__restrict__ losses, LossCfg cfg) {
    const int t = blockIdx.x * 256 + threadIdx.x, p = t >> 2, q = t & 3;
    float acc = 0.f;
    if (p < P + 3) {
#pragma unroll 8
        for (int w = q; w < n_wg; w += 4) acc += partials[(i64)w * stride + p];
    }
    const float a1 = __shfl_xor(acc, 1);
    acc = (q & 1) ? a1 + acc : acc + a1;  // (both lanes of a pair hold s_even + s_odd, added in that order)
    const float a2 = __shfl_xor(acc, 2);
    acc = (q & 2) ? a2 + acc : acc + a2;
    if (q != 0 || p >= P + 3) return;
    if (p < P)
        grad[p] = acc;
    else if (losses)
        losses[p - P] = p - P == 0 ? -cfg.inv_bk * acc : (p - P == 1 ? cfg.value_w * cfg.inv_b * acc : cfg.entropy_w * -cfg.inv_b * acc);
}

// The gradient scaled by grad_scale (1 / world size behind the data-parallel all-reduce); global-norm clip; Adam.
__global__ void __launch_bounds__(1024) k_ppo_adam(int P, float *__restrict__ params, const float *__restrict__ grad, float *__restrict__ m, float *__restrict__ v, i64 *__restrict__ step,
                                                    double lr, double b1, double b2, double eps, float max_norm, float grad_scale) {
    // ceil(P / 1024) workgroups: every one computes the WHOLE vector's norm (52 KB out of L2, the same sums in the same order everywhere), then steps its own 1 024
    // parameters -- no grid-wide exchange for the clip factor; the last workgroup out (step[1]: an arrival counter) advances the step count.
    __shared__ float red[1024];
    constexpr int kPer = 16;  // elements per thread of the norm pass (>= the largest geometry's 14.3 K / 1024), fully unrolled: every load in flight at once
    const int tid = threadIdx.x, mine = blockIdx.x * 1024 + tid;
    const i64 steps_taken = step[0];
    const bool in_mine = mine < P;
    float pp = 0.f, mm = 0.f, vv = 0.f;
    if (in_mine) pp = params[mine], mm = m[mine], vv = v[mine];
    float g[kPer];
    float ss = 0.f;
#pragma unroll
    for (int k = 0; k < kPer; k++) {
        const int p = tid + 1024 * k;
        g[k] = p < P ? grad[p] * grad_scale : 0.f;
    }
#pragma unroll
    for (int k = 0; k < kPer; k++) ss = fmaf(g[k], g[k], ss);
    red[tid] = ss;
    __syncthreads();
    for (int s = 512; s > 0; s >>= 1) {
        if (tid < s) red[tid] += red[tid + s];
        __syncthreads();
    }
    const float clip = max_norm > 0.f ? fminf(max_norm / (sqrtf(red[0]) + 1e-6f), 1.0f) : 1.0f;  // torch.nn.utils.clip_grad_norm_
    if (in_mine) {
        const srlx::AdamCoef c = srlx::adam_coef(lr, b1, b2, eps, steps_taken);
        float gc = 0.f;
#pragma unroll
        for (int k = 0; k < kPer; k++)
            if (k == (int)blockIdx.x) gc = g[k] * clip;  // (this thread's own element is g[blockIdx.x]: selected without a run-time register index)
        srlx::adam_one(pp, gc, mm, vv, c);  // (`grad` itself stays as it is: every workgroup reads all of it for the norm)
        params[mine] = pp, m[mine] = mm, v[mine] = vv;
    }
    __syncthreads();
    if (tid == 0) {
        __threadfence();
        const unsigned long long old = atomicAdd(reinterpret_cast<unsigned long long *>(step + 1), 1ull);
        if (old == gridDim.x - 1) {  // every workgroup has read step[0] (it arrives behind that read)
            step[1] = 0;
            step[0] = steps_taken + 1;
        }
    }
}

static_assert(H * OBS_MAX + 3 * H * H + 2 * A_MAX * H + 5 * H + 1 + 2 * A_MAX <= 16 * 1024, "k_ppo_adam: sixteen elements per thread");
bool geometry_ok(int obs, int A) { return obs >= 1 && obs <= OBS_MAX && A >= 1 && A <= A_MAX; }

}  // namespace

extern "C" {

int srlx_ppo_net_param_count(int obs_dim, int action_dim) { return geometry_ok(obs_dim, action_dim) ? net_off(obs_dim, action_dim).total : -1; }

int srlx_ppo_net_forward(int64_t n, int obs_dim, int action_dim, const float *d_params, const float *d_obs, float *d_v, float *d_loc, float *d_log_scale, void *stream) {
    SRLX_REQUIRE(n > 0 && geometry_ok(obs_dim, action_dim) && d_params && d_obs && d_v && d_loc && d_log_scale, "ppo_net_forward: bad argument");
    static bool attr = false;
    if (!attr) {
        SRLX_HIP(hipFuncSetAttribute((const void *)k_ppo_forward, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(FwdLds)));
        attr = true;
    }
    i64 wgs = (n + RE - 1) / RE;
    if (wgs > 1024) wgs = 1024;
    hipLaunchKernelGGL(k_ppo_forward, dim3((unsigned)wgs), dim3(256), sizeof(FwdLds), (hipStream_t)stream, (i64)n, obs_dim, action_dim, d_params, d_obs, d_v, d_loc, d_log_scale);
    SRLX_HIP(hipGetLastError());
    return SRLX_OK;
}

int srlx_ppo_net_rollout_max_horizon(int action_dim) {  // what fits the workgroup's LDS beside weights and activations
    if (!geometry_ok(3, action_dim)) return -1;
    const long long t = ((long long)160 * 1024 - (long long)sizeof(FwdLds)) / ((long long)RE * (3 + action_dim) * (long long)sizeof(float));
    return (int)(t < 1024 ? t : 1024);
}

int srlx_ppo_net_rollout(int64_t n_envs, int64_t horizon, int action_dim, const float *d_params, float *d_env_state, int32_t *d_step_in_episode, float *d_env_obs,
                         int64_t episode_len, uint64_t env_seed, int64_t *d_env_counter, uint64_t act_seed, int64_t *d_act_counter, double log_scale_min,
                         double log_scale_max, double discount, double gae_lambda, float *d_b_obs, float *d_b_act, float *d_b_logp, float *d_b_val, float *d_b_rew,
                         uint8_t *d_b_done, float *d_b_adv, float *d_last_v, float *d_episode_return, float *d_finished, void *stream) {
    SRLX_REQUIRE(n_envs > 0 && n_envs % RE == 0, "ppo_net_rollout: the environment count must be a multiple of 16");
    SRLX_REQUIRE(horizon > 0 && geometry_ok(3, action_dim) && horizon <= srlx_ppo_net_rollout_max_horizon(action_dim) && episode_len > 0,
                 "ppo_net_rollout: bad geometry (horizon <= srlx_ppo_net_rollout_max_horizon)");
    SRLX_REQUIRE(d_params && d_env_state && d_step_in_episode && d_env_obs && d_env_counter && d_act_counter && d_b_obs && d_b_act && d_b_logp && d_b_val && d_b_rew && d_b_done &&
                     d_b_adv && d_last_v && d_episode_return && d_finished,
                 "ppo_net_rollout: NULL argument");
    const size_t lds = sizeof(FwdLds) + (size_t)horizon * RE * (3 + action_dim) * sizeof(float);
    SRLX_REQUIRE(lds <= 160 * 1024, "ppo_net_rollout: horizon too long for the workgroup's LDS");
    static size_t lds_set = 0;
    if (lds > lds_set) {
        SRLX_HIP(hipFuncSetAttribute((const void *)k_ppo_rollout, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        lds_set = lds;
    }
    RolloutArgs a{n_envs, horizon, action_dim, d_params, d_env_state, d_step_in_episode, d_env_obs, episode_len, (u64)env_seed, (u64)act_seed, d_env_counter, d_act_counter,
                  (float)log_scale_min, (float)log_scale_max, discount, gae_lambda, d_b_obs, d_b_act, d_b_logp, d_b_val, d_b_rew, d_b_done, d_b_adv, d_last_v, d_episode_return,
                  d_finished};
    hipLaunchKernelGGL(k_ppo_rollout, dim3((unsigned)(n_envs / RE)), dim3(256), lds, (hipStream_t)stream, a);
    hipLaunchKernelGGL(k_advance2, dim3(1), dim3(1), 0, (hipStream_t)stream, d_env_counter, d_act_counter, (i64)horizon);
    SRLX_HIP(hipGetLastError());
    return SRLX_OK;
}

int srlx_ppo_net_minibatch(int64_t minibatch, const int64_t *d_rows, int obs_dim, int action_dim, const float *d_params, const float *d_b_obs, const float *d_b_act,
                           const float *d_b_logp, const float *d_b_adv, const float *d_b_v_target, const float *d_b_val, double log_scale_min, double log_scale_max,
                           int baseline_advantage, int surrogate_clip, double policy_clip_range, int enable_value_clip, double value_clip_range, double value_loss_weight,
                           double entropy_weight, float *d_partials, float *d_grad, float *d_losses, void *stream) {
    SRLX_REQUIRE(minibatch > 0 && geometry_ok(obs_dim, action_dim), "ppo_net_minibatch: bad geometry");
    SRLX_REQUIRE(d_rows && d_params && d_b_obs && d_b_act && d_b_logp && d_b_adv && d_b_v_target && d_b_val && d_partials && d_grad, "ppo_net_minibatch: NULL argument");
    static bool attr = false;
    if (!attr) {
        SRLX_HIP(hipFuncSetAttribute((const void *)k_ppo_minibatch, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(MbLds)));
        attr = true;
    }
    const NetOff o = net_off(obs_dim, action_dim);
    const int stride = (o.total + 3 + 3) & ~3;
    i64 tiles = (minibatch + S - 1) / S;
    const int wgs = (int)(tiles < 256 ? tiles : 256);
    MbArgs a{};
    a.mb = minibatch;
    a.perm = d_rows;
    a.obs = obs_dim, a.A = action_dim;
    a.params = d_params;
    a.b_obs = d_b_obs, a.b_act = d_b_act, a.b_logp = d_b_logp, a.b_adv = d_b_adv, a.b_vt = d_b_v_target, a.b_val = d_b_val;
    a.cfg = LossCfg{(float)log_scale_min, (float)log_scale_max, baseline_advantage, surrogate_clip, enable_value_clip, (float)policy_clip_range, (float)value_clip_range,
                    (float)value_loss_weight, (float)entropy_weight, 1.0f / (float)minibatch, 1.0f / (float)(minibatch * action_dim)};
    a.partials = d_partials;
    a.stride = stride;
    hipLaunchKernelGGL(k_ppo_minibatch, dim3((unsigned)wgs), dim3(256), sizeof(MbLds), (hipStream_t)stream, a);
    hipLaunchKernelGGL(k_ppo_reduce, dim3((unsigned)((4 * (o.total + 3) + 255) / 256)), dim3(256), 0, (hipStream_t)stream, wgs, o.total, stride, d_partials, d_grad, d_losses, a.cfg);
    SRLX_HIP(hipGetLastError());
    return SRLX_OK;
}

int srlx_ppo_net_partials_floats(int obs_dim, int action_dim) {
    if (!geometry_ok(obs_dim, action_dim)) return -1;
    return 256 * ((net_off(obs_dim, action_dim).total + 3 + 3) & ~3);
}

int srlx_ppo_net_adam(int obs_dim, int action_dim, float *d_params, float *d_grad, float *d_exp_avg, float *d_exp_avg_sq, int64_t *d_step, double lr, double beta1, double beta2,
                      double eps, double max_grad_norm, double grad_scale, void *stream) {
    SRLX_REQUIRE(geometry_ok(obs_dim, action_dim) && d_params && d_grad && d_exp_avg && d_exp_avg_sq && d_step, "ppo_net_adam: bad argument");
    hipLaunchKernelGGL(k_ppo_adam, dim3((unsigned)((net_off(obs_dim, action_dim).total + 1023) / 1024)), dim3(1024), 0, (hipStream_t)stream, net_off(obs_dim, action_dim).total, d_params, d_grad, d_exp_avg, d_exp_avg_sq, (i64 *)d_step, lr, beta1,
                       beta2, eps, (float)max_grad_norm, (float)grad_scale);
    SRLX_HIP(hipGetLastError());
    return SRLX_OK;
}

}  // extern "C"
