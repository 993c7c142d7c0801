// srlx_ppo.hip -- PPO on the vectorised path (BASELINE config 5):
//   srlx_ppo_normal_act   : Normal policy head -> action sample + log-probability for E environments in one
//                           launch (srl/algorithms/ppo/ppo.py:316-339, srl/rl/tf/distributions/normal_dist_block.py:13-20,64-74)
//   srlx_ppo_loss_normal  : the whole of compute_train_loss (ppo.py:102-169) for a Normal policy, forward AND the
//                           gradient seeds d loss / d (loc, log_scale, v) in one launch; the host backpropagates the
//                           seeds through the network (same scheme as the Huber seed of the DQN family)
//   srlx_ppo_loss_logpi   : the same for an arbitrary policy head given its log-probabilities (Categorical):
//                           seeds d loss / d (new_logpi, v)
//   srlx_pendulum_step    : the Pendulum-shaped synthetic workload of BASELINE config 5 (obs (cos, sin, thdot),
//                           one torque) for E lock-stepped environments
// The reference module imports TensorFlow and cannot be imported in the build container: these follow the source
// lines only (parity UNPINNED, like srlx_gae_scan); tests check them against oracle/hot_path_oracle.py and
// against torch autograd of the same formula.
// All of them are tiny elementwise/reduction kernels (B x action_dim floats): their point is launch count --
// one launch instead of ~40 framework kernels per minibatch.
#include "srlx_common.h"
#include "srlx_ppo_math.h"

namespace {

using i64 = int64_t;
using u8 = unsigned char;
using srlxp::LossCfg;

__device__ __forceinline__ float block_sum(float v, float *red) {
    const int t = threadIdx.x;
    red[t] = v;
    __syncthreads();
    for (int s = blockDim.x >> 1; s > 0; s >>= 1) {
        if (t < s) red[t] += red[t + s];
        __syncthreads();
    }
    const float r = red[0];
    __syncthreads();
    return r;
}

__global__ void __launch_bounds__(256) k_normal_act(i64 n, const float *loc, const float *log_scale, float ls_lo, float ls_hi, unsigned long long seed,
                                                    const i64 *counter, int deterministic, float *action, float *logprob) {
    const i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    srlxp::normal_act_one(loc[i], log_scale[i], ls_lo, ls_hi, seed, deterministic ? 0ull : (unsigned long long)counter[0], i, deterministic, action[i], logprob[i]);
}

struct PpoArgs {
    i64 B;
    int K;  // action dimensions (log-probabilities per sample)
    const float *loc, *log_scale, *action;  // normal variant
    const float *new_logpi;                 // logpi variant
    const float *old_logpi, *advantage, *v, *v_target, *old_v;
    float ls_lo, ls_hi;
    int baseline_advantage, surrogate_clip, value_clip;
    float policy_clip, value_clip_range, value_w, entropy_w;
    float *losses;  // [3] policy, value, entropy (weighted, as the reference reports them)
    float *d_loc, *d_log_scale, *d_logpi, *d_v;
};

// One workgroup per launch slice; partial sums are accumulated with float atomics into losses[] (zeroed by the host
// side memset) -- the three scalars are reporting values, the gradient seeds do not depend on them.
template <bool NORMAL>
__global__ void __launch_bounds__(256) k_ppo_loss(PpoArgs a) {
    __shared__ float red[256];
    LossCfg c{a.ls_lo, a.ls_hi, a.baseline_advantage, a.surrogate_clip, a.value_clip, a.policy_clip, a.value_clip_range, a.value_w, a.entropy_w,
              1.0f / (float)a.B, 1.0f / (float)(a.B * a.K)};
    float s_pol = 0.f, s_val = 0.f, s_ent = 0.f;
    for (i64 b = (i64)blockIdx.x * blockDim.x + threadIdx.x; b < a.B; b += (i64)gridDim.x * blockDim.x) {
        const float v = a.v[b], vt = a.v_target[b];
        const float adv = a.baseline_advantage ? a.advantage[b] - v : a.advantage[b];  // ppo.py:121-122 (v is a stop_gradient there)
        float ent = 0.f;
        for (int k = 0; k < a.K; k++) {
            const i64 i = b * a.K + k;
            float term, e1;
            if (NORMAL)
                srlxp::policy_normal(c, a.loc[i], a.log_scale[i], a.action[i], a.old_logpi[i], adv, term, e1, a.d_loc[i], a.d_log_scale[i]);
            else
                a.d_logpi[i] = srlxp::policy_lp_terms(c, a.new_logpi[i], a.old_logpi[i], adv, term, e1);
            s_pol += term;
            ent += e1;
        }
        s_ent += ent;
        float g_v;
        s_val += srlxp::value_term(c, v, vt, a.value_clip ? a.old_v[b] : 0.f, g_v);
        a.d_v[b] = g_v;
    }
    const float p = block_sum(s_pol, red), vl = block_sum(s_val, red), en = block_sum(s_ent, red);
    if (threadIdx.x == 0) {
        atomicAdd(&a.losses[0], -c.inv_bk * p);
        atomicAdd(&a.losses[1], a.value_w * c.inv_b * vl);
        atomicAdd(&a.losses[2], a.entropy_w * -c.inv_b * en);
    }
}

__global__ void __launch_bounds__(256) k_pendulum(i64 E, float *state /*[E][2] th, thdot*/, int32_t *t_in_ep, const float *action, i64 episode_len,
                                                  unsigned long long seed, const i64 *counter, float *obs /*[E][3]*/, float *reward, u8 *done) {
    const i64 e = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= E) return;
    float th = state[2 * e], thd = state[2 * e + 1];
    int t = t_in_ep[e];
    srlxp::pendulum_one(th, thd, t, action[e], episode_len, seed, (unsigned long long)counter[0], e, obs[3 * e], obs[3 * e + 1], obs[3 * e + 2], reward[e], done[e]);
    state[2 * e] = th;
    state[2 * e + 1] = thd;
    t_in_ep[e] = t;
}

__global__ void k_advance1(i64 *c) { c[0] += 1; }

}  // namespace

extern "C" {

int srlx_ppo_normal_act(int64_t n, const float *d_loc, const float *d_log_scale, double log_scale_min, double log_scale_max, uint64_t seed,
                        int64_t *d_counter, int deterministic, float *d_action, float *d_logprob, void *stream) {
    SRLX_REQUIRE(n > 0 && d_loc && d_log_scale && d_action && d_logprob && (deterministic || d_counter), "ppo_normal_act: bad argument");
    hipLaunchKernelGGL(k_normal_act, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (i64)n, d_loc, d_log_scale, (float)log_scale_min,
                       (float)log_scale_max, (unsigned long long)seed, (const i64 *)d_counter, deterministic, d_action, d_logprob);
    if (!deterministic) hipLaunchKernelGGL(k_advance1, dim3(1), dim3(1), 0, (hipStream_t)stream, d_counter);
    SRLX_HIP(hipGetLastError());
    return SRLX_OK;
}

static int ppo_loss_common(PpoArgs &a, bool normal, void *stream) {
    SRLX_REQUIRE(a.B > 0 && a.K > 0 && a.old_logpi && a.advantage && a.v && a.v_target && a.losses && a.d_v, "ppo_loss: bad argument");
    SRLX_REQUIRE(!a.value_clip || a.old_v, "ppo_loss: enable_value_clip needs old_v");
    SRLX_HIP(hipMemsetAsync(a.losses, 0, 3 * sizeof(float), (hipStream_t)stream));
    i64 blocks = (a.B + 255) / 256;
    if (blocks > 1024) blocks = 1024;
    if (normal)
        hipLaunchKernelGGL(k_ppo_loss<true>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, a);
    else
        hipLaunchKernelGGL(k_ppo_loss<false>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, a);
    SRLX_HIP(hipGetLastError());
    return SRLX_OK;
}

int srlx_ppo_loss_normal(int64_t batch, int action_dim, const float *d_loc, const float *d_log_scale, double log_scale_min, double log_scale_max,
                         const float *d_action, const float *d_old_logpi, const float *d_advantage, const float *d_v, const float *d_v_target,
                         const float *d_old_v, int baseline_advantage, int surrogate_clip, double policy_clip_range, int enable_value_clip,
                         double value_clip_range, double value_loss_weight, double entropy_weight, float *d_losses, float *d_grad_loc,
                         float *d_grad_log_scale, float *d_grad_v, void *stream) {
    SRLX_REQUIRE(d_loc && d_log_scale && d_action && d_grad_loc && d_grad_log_scale, "ppo_loss_normal: NULL argument");
    PpoArgs a{};
    a.B = batch;
    a.K = action_dim;
    a.loc = d_loc;
    a.log_scale = d_log_scale;
    a.action = d_action;
    a.old_logpi = d_old_logpi;
    a.advantage = d_advantage;
    a.v = d_v;
    a.v_target = d_v_target;
    a.old_v = d_old_v;
    a.ls_lo = (float)log_scale_min;
    a.ls_hi = (float)log_scale_max;
    a.baseline_advantage = baseline_advantage;
    a.surrogate_clip = surrogate_clip;
    a.value_clip = enable_value_clip;
    a.policy_clip = (float)policy_clip_range;
    a.value_clip_range = (float)value_clip_range;
    a.value_w = (float)value_loss_weight;
    a.entropy_w = (float)entropy_weight;
    a.losses = d_losses;
    a.d_loc = d_grad_loc;
    a.d_log_scale = d_grad_log_scale;
    a.d_v = d_grad_v;
    return ppo_loss_common(a, true, stream);
}

int srlx_ppo_loss_logpi(int64_t batch, int n_logpi, const float *d_new_logpi, const float *d_old_logpi, const float *d_advantage, const float *d_v,
                        const float *d_v_target, const float *d_old_v, int baseline_advantage, int surrogate_clip, double policy_clip_range,
                        int enable_value_clip, double value_clip_range, double value_loss_weight, double entropy_weight, float *d_losses,
                        float *d_grad_logpi, float *d_grad_v, void *stream) {
    SRLX_REQUIRE(d_new_logpi && d_grad_logpi, "ppo_loss_logpi: NULL argument");
    PpoArgs a{};
    a.B = batch;
    a.K = n_logpi;
    a.new_logpi = d_new_logpi;
    a.old_logpi = d_old_logpi;
    a.advantage = d_advantage;
    a.v = d_v;
    a.v_target = d_v_target;
    a.old_v = d_old_v;
    a.baseline_advantage = baseline_advantage;
    a.surrogate_clip = surrogate_clip;
    a.value_clip = enable_value_clip;
    a.policy_clip = (float)policy_clip_range;
    a.value_clip_range = (float)value_clip_range;
    a.value_w = (float)value_loss_weight;
    a.entropy_w = (float)entropy_weight;
    a.losses = d_losses;
    a.d_logpi = d_grad_logpi;
    a.d_v = d_grad_v;
    return ppo_loss_common(a, false, stream);
}

int srlx_pendulum_step(int64_t n_envs, float *d_state, int32_t *d_step_in_episode, const float *d_action, int64_t episode_len, uint64_t seed,
                       int64_t *d_counter, float *d_obs, float *d_reward, uint8_t *d_done, void *stream) {
    SRLX_REQUIRE(n_envs > 0 && d_state && d_step_in_episode && d_action && d_counter && d_obs && d_reward && d_done && episode_len > 0,
                 "pendulum_step: bad argument");
    hipLaunchKernelGGL(k_pendulum, dim3((unsigned)((n_envs + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (i64)n_envs, d_state, d_step_in_episode, d_action,
                       (i64)episode_len, (unsigned long long)seed, (const i64 *)d_counter, d_obs, d_reward, d_done);
    hipLaunchKernelGGL(k_advance1, dim3(1), dim3(1), 0, (hipStream_t)stream, d_counter);
    SRLX_HIP(hipGetLastError());
    return SRLX_OK;
}

}  // extern "C"
