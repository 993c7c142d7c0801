// srlx_train.hip -- fused learner arithmetic: n-step/retrace TD target + Huber loss + gradient seed +
// priority recompute in one launch; 1-step (double-)DQN target; GAE reverse scan; multi-tensor Adam step.
//
// These replace host-side numpy between network calls in the reference:
//   srl/algorithms/rainbow/rainbow.py:226-287        (calc_target_q after the two forwards)
//   srl/algorithms/rainbow/model_torch.py:103-105,113 (selected Q, HuberLoss(target*w, q*w), |target-q|)
//   srl/algorithms/dqn/dqn.py:144-176, srl/algorithms/rainbow/rainbow_nomultisteps.py:10-43
//   srl/algorithms/ppo/ppo.py:389-404
//   srl/algorithms/rainbow/model_torch.py:71,109      (torch.optim.Adam step)
// All are tiny (B x n x A floats): latency-bound single-workgroup kernels whose point is to keep the
// learner step free of device<->host hops (the reference does four per train()).
// float32 arithmetic follows numpy's evaluation order of the cited lines.
#include "srlx_adam_math.h"
#include "srlx_common.h"
#include "srlx_td_math.h"

namespace {

using i64 = int64_t;
using u8 = unsigned char;

using srlx::TdArgs;
using srlx::argmax_masked;
using srlx::inverse_rescaling;
using srlx::rescaling;
constexpr int kMaxStep = srlx::kTdMaxStep;

__global__ void __launch_bounds__(256) k_nstep_td_huber_priority(TdArgs a) {
    __shared__ double red[256];
    const int t = threadIdx.x, T = blockDim.x;
    const double loss_acc = srlx::td_rows(a, t, T, a.grad_q0, true);
    red[t] = loss_acc;
    __syncthreads();
    for (int s = T >> 1; s > 0; s >>= 1) {
        if (t < s) red[t] += red[t + s];
        __syncthreads();
    }
    if (t == 0) a.loss[0] = (float)(red[0] / (double)a.B);
}

struct DqnArgs {
    i64 B;
    int A;
    const float *q_on_next, *q_tg_next, *rewards, *undone;
    const u8 *invalid_next;
    double discount;
    const float *discount_per_sample;
    int double_dqn, rescale, f64_accum;
    float *target;
};

__global__ void __launch_bounds__(256) k_dqn_target(DqnArgs a) {
    __shared__ float red[256];
    const int t = threadIdx.x, T = blockDim.x;
    // np.min over the WHOLE (B, A) array of the net that is masked (dqn.py:160,164)
    const float *masked = a.double_dqn ? a.q_on_next : a.q_tg_next;
    float mn = INFINITY;
    for (i64 i = t; i < a.B * a.A; i += T) mn = fminf(mn, masked[i]);
    red[t] = mn;
    __syncthreads();
    for (int s = T >> 1; s > 0; s >>= 1) {
        if (t < s) red[t] = fminf(red[t], red[t + s]);
        __syncthreads();
    }
    const float qmin = red[0];
    for (i64 b = t; b < a.B; b += T) {
        const float *row = masked + b * a.A;
        const u8 *inv = a.invalid_next ? a.invalid_next + b * a.A : nullptr;
        int best = 0;
        float bv = 0.f;
        for (int k = 0; k < a.A; k++) {
            const float v = (inv && inv[k]) ? qmin : row[k];
            if (k == 0 || v > bv) {
                best = k;
                bv = v;
            }
        }
        // double: value of the target net at the online argmax (:161-162); else max of the masked target row (:165)
        float maxq = a.double_dqn ? a.q_tg_next[b * a.A + best] : bv;
        if (a.rescale) maxq = inverse_rescaling(maxq);
        float tq;
        if (a.f64_accum) {  // dqn.py:171 with an int `undone` array: numpy promotes to float64, cast at :176
            double v = (double)a.rewards[b] + ((double)a.undone[b] * a.discount) * (double)maxq;
            tq = (float)v;
            if (a.rescale) tq = (float)((v > 0 ? 1.0 : (v < 0 ? -1.0 : 0.0)) * (sqrt(fabs(v) + 1.0) - 1.0) + 0.001 * v);
        } else {  // rainbow_nomultisteps.py:38 all float32
            // agent57_light.py:263: the same expression with the sampled actor's gamma (float32 [B])
            const float g = a.discount_per_sample ? a.discount_per_sample[b] : (float)a.discount;
            tq = a.rewards[b] + (a.undone[b] * g) * maxq;
            if (a.rescale) tq = rescaling(tq);
        }
        a.target[b] = tq;
    }
}

// one thread per environment, reverse scan over the horizon (ppo.py:389-404)
__global__ void __launch_bounds__(256) k_gae_scan(i64 E, i64 T, const float *rewards, const float *values, const u8 *done,
                                                   const float *last_values, double discount, double lam, float *adv) {
    const i64 e = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= E) return;
    const float g = (float)discount;
    const float gl = (float)(discount * lam);  // python: discount * gae_discount (float64) * np.float32 -> float32
    float gae = 0.f;
    for (i64 i = T - 1; i >= 0; i--) {
        const i64 k = i * E + e;
        float delta;
        if (done[k]) {
            delta = rewards[k] - values[k];  // :396-397 last step of an episode: no bootstrap
            gae = 0.f;
        } else if (i == T - 1) {
            delta = last_values ? (rewards[k] + g * last_values[e]) - values[k] : rewards[k] - values[k];
        } else {
            delta = (rewards[k] + g * values[k + E]) - values[k];  // :399
        }
        gae = delta + gl * gae;  // :400
        adv[k] = gae;
    }
}

// ------------------------------------------------------------------------------------------
// Adam (model_torch.py:71: torch.optim.Adam(lr); :109 optimizer.step()) for every parameter tensor of the network in
// ONE launch: 28 B of traffic per element (p, g, m, v read; p, m, v written), HBM bound -- 8.0 M parameters = 224 MB.
// The tensor table travels as a kernel argument; a workgroup owns one 4096-element chunk of one tensor.
// The per-element arithmetic (torch's order of operations) is srlx_adam_math.h.
// ------------------------------------------------------------------------------------------
constexpr int kAdamMaxTensors = 24;
constexpr int kAdamChunk = 4096;
struct AdamTable {
    float *p[kAdamMaxTensors];
    const float *g[kAdamMaxTensors];
    float *m[kAdamMaxTensors];
    float *v[kAdamMaxTensors];
    i64 n[kAdamMaxTensors];
    int chunk_start[kAdamMaxTensors + 1];
    int n_tensors;
};

__global__ void __launch_bounds__(256) k_adam(AdamTable tb, double lr, double beta1, double beta2, double eps, const i64 *d_step) {
    int ti = 0;
    while (ti + 1 < tb.n_tensors && (int)blockIdx.x >= tb.chunk_start[ti + 1]) ti++;
    const i64 off = (i64)((int)blockIdx.x - tb.chunk_start[ti]) * kAdamChunk;
    const i64 n = tb.n[ti];
    float *p = tb.p[ti] + off, *m = tb.m[ti] + off, *v = tb.v[ti] + off;
    const float *g = tb.g[ti] + off;
    const i64 left = n - off;
    const int cnt = left < kAdamChunk ? (int)left : kAdamChunk;
    const int t = threadIdx.x;
    // Every load of the kernel -- the step count included -- is issued before anything is computed: one memory round trip per launch
    // instead of two (inside the lock-step loop a round trip costs this kernel more than its arithmetic).
    if (cnt == kAdamChunk) {  // chunks start at multiples of 4096 floats of a 16-byte aligned tensor
        constexpr int R = kAdamChunk / (256 * 4);
        float4 pp[R], mm[R], vv[R], gg[R];
#pragma unroll
        for (int r = 0; r < R; r++) {
            const int k = (r * 256 + t) * 4;
            pp[r] = *reinterpret_cast<float4 *>(p + k), mm[r] = *reinterpret_cast<float4 *>(m + k), vv[r] = *reinterpret_cast<float4 *>(v + k);
            gg[r] = *reinterpret_cast<const float4 *>(g + k);
        }
        const srlx::AdamCoef c = srlx::adam_coef(lr, beta1, beta2, eps, *d_step);
#pragma unroll
        for (int r = 0; r < R; r++) {
            const int k = (r * 256 + t) * 4;
            srlx::adam_one(pp[r].x, gg[r].x, mm[r].x, vv[r].x, c);
            srlx::adam_one(pp[r].y, gg[r].y, mm[r].y, vv[r].y, c);
            srlx::adam_one(pp[r].z, gg[r].z, mm[r].z, vv[r].z, c);
            srlx::adam_one(pp[r].w, gg[r].w, mm[r].w, vv[r].w, c);
            *reinterpret_cast<float4 *>(p + k) = pp[r];
            *reinterpret_cast<float4 *>(m + k) = mm[r];
            *reinterpret_cast<float4 *>(v + k) = vv[r];
        }
    } else {
        constexpr int R = kAdamChunk / 256;
        float pp[R], mm[R], vv[R], gg[R];
#pragma unroll
        for (int r = 0; r < R; r++) {
            const int k = r * 256 + t;
            if (k < cnt) pp[r] = p[k], mm[r] = m[k], vv[r] = v[k], gg[r] = g[k];
        }
        const srlx::AdamCoef c = srlx::adam_coef(lr, beta1, beta2, eps, *d_step);
#pragma unroll
        for (int r = 0; r < R; r++) {
            const int k = r * 256 + t;
            if (k < cnt) {
                srlx::adam_one(pp[r], gg[r], mm[r], vv[r], c);
                p[k] = pp[r], m[k] = mm[r], v[k] = vv[r];
            }
        }
    }
}

}  // namespace

extern "C" {

int srlx_nstep_td_huber_priority(int64_t batch, int n_step, int n_actions, const float *d_q_on_next,
                                 const float *d_q_tg_next, const float *d_q_on_0, const int32_t *d_actions,
                                 const float *d_rewards, const float *d_terminated, const uint8_t *d_invalid_next,
                                 const float *d_weights, double discount, double retrace_h, int enable_double_dqn,
                                 int enable_rescale, float *d_target, float *d_loss, float *d_grad_q0,
                                 float *d_priorities, void *stream) {
    SRLX_REQUIRE(batch > 0 && n_step >= 1 && n_step <= kMaxStep && n_actions >= 1, "nstep_td: bad sizes (n_step <= %d)", kMaxStep);
    SRLX_REQUIRE(d_q_on_next && d_q_tg_next && d_q_on_0 && d_actions && d_rewards && d_terminated && d_weights,
                 "nstep_td: NULL input");
    SRLX_REQUIRE(d_target && d_loss && d_grad_q0 && d_priorities, "nstep_td: NULL output");
    TdArgs a{batch, n_step, n_actions, d_q_on_next, d_q_tg_next, d_q_on_0, d_actions, d_rewards, d_terminated,
             d_invalid_next, d_weights, discount, retrace_h, enable_double_dqn, enable_rescale, d_target, d_loss,
             d_grad_q0, d_priorities, (i64)n_step * n_actions, (i64)n_actions};
    srlx::td_fill_discounts(a);
    hipLaunchKernelGGL(k_nstep_td_huber_priority, dim3(1), dim3(256), 0, (hipStream_t)stream, a);
    SRLX_HIP(hipGetLastError());
    return SRLX_OK;
}

int srlx_nstep_td_huber_priority_packed(int64_t batch, int n_step, int n_actions, const float *d_q_on_all, const float *d_q_tg_next,
                                        const int32_t *d_actions, const float *d_rewards, const float *d_terminated,
                                        const uint8_t *d_invalid_next, const float *d_weights, double discount, double retrace_h,
                                        int enable_double_dqn, int enable_rescale, float *d_target, float *d_loss, float *d_grad_q0,
                                        float *d_priorities, void *stream) {
    SRLX_REQUIRE(batch > 0 && n_step >= 1 && n_step <= kMaxStep && n_actions >= 1, "nstep_td_packed: bad sizes (n_step <= %d)", kMaxStep);
    SRLX_REQUIRE(d_q_on_all && d_q_tg_next && d_actions && d_rewards && d_terminated && d_weights, "nstep_td_packed: NULL input");
    SRLX_REQUIRE(d_target && d_loss && d_grad_q0 && d_priorities, "nstep_td_packed: NULL output");
    const i64 row = (i64)(n_step + 1) * n_actions;
    TdArgs a{batch, n_step, n_actions, d_q_on_all + n_actions, d_q_tg_next, d_q_on_all, d_actions, d_rewards, d_terminated,
             d_invalid_next, d_weights, discount, retrace_h, enable_double_dqn, enable_rescale, d_target, d_loss,
             d_grad_q0, d_priorities, row, row};
    srlx::td_fill_discounts(a);
    hipLaunchKernelGGL(k_nstep_td_huber_priority, dim3(1), dim3(256), 0, (hipStream_t)stream, a);
    SRLX_HIP(hipGetLastError());
    return SRLX_OK;
}

int srlx_dqn_target(int64_t batch, int n_actions, const float *d_q_on_next, const float *d_q_tg_next,
                    const float *d_rewards, const float *d_undone, const uint8_t *d_invalid_next, double discount,
                    const float *d_discount_per_sample, int enable_double_dqn, int enable_rescale, int f64_accum, float *d_target,
                    void *stream) {
    SRLX_REQUIRE(batch > 0 && n_actions >= 1, "dqn_target: bad sizes");
    SRLX_REQUIRE(!(d_discount_per_sample && f64_accum), "dqn_target: a per-sample discount is float32 arithmetic (f64_accum must be 0)");
    SRLX_REQUIRE(d_q_tg_next && d_rewards && d_undone && d_target && (d_q_on_next || !enable_double_dqn), "dqn_target: NULL argument");
    DqnArgs a{batch, n_actions, d_q_on_next, d_q_tg_next, d_rewards, d_undone, d_invalid_next, discount, d_discount_per_sample,
              enable_double_dqn, enable_rescale, f64_accum, d_target};
    hipLaunchKernelGGL(k_dqn_target, dim3(1), dim3(256), 0, (hipStream_t)stream, a);
    SRLX_HIP(hipGetLastError());
    return SRLX_OK;
}

int srlx_gae_scan(int64_t n_envs, int64_t horizon, const float *d_rewards, const float *d_values, const uint8_t *d_done,
                  const float *d_last_values, double discount, double gae_lambda, float *d_adv, void *stream) {
    SRLX_REQUIRE(n_envs > 0 && horizon > 0 && d_rewards && d_values && d_done && d_adv, "gae_scan: bad argument");
    hipLaunchKernelGGL(k_gae_scan, dim3((unsigned)((n_envs + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (i64)n_envs,
                       (i64)horizon, d_rewards, d_values, d_done, d_last_values, discount, gae_lambda, d_adv);
    SRLX_HIP(hipGetLastError());
    return SRLX_OK;
}

int srlx_adam_step(int n_tensors, float *const *d_params, const float *const *d_grads, float *const *d_exp_avg, float *const *d_exp_avg_sq,
                   const int64_t *numels, double lr, double beta1, double beta2, double eps, const int64_t *d_step, void *stream) {
    SRLX_REQUIRE(n_tensors > 0 && n_tensors <= kAdamMaxTensors, "adam_step: 1..%d tensors per call (got %d)", kAdamMaxTensors, n_tensors);
    SRLX_REQUIRE(d_params && d_grads && d_exp_avg && d_exp_avg_sq && numels && d_step, "adam_step: NULL argument");
    AdamTable tb{};
    tb.n_tensors = n_tensors;
    int chunks = 0;
    for (int i = 0; i < n_tensors; i++) {
        SRLX_REQUIRE(d_params[i] && d_grads[i] && d_exp_avg[i] && d_exp_avg_sq[i] && numels[i] > 0, "adam_step: tensor %d is NULL or empty", i);
        SRLX_REQUIRE(((uintptr_t)d_params[i] | (uintptr_t)d_grads[i] | (uintptr_t)d_exp_avg[i] | (uintptr_t)d_exp_avg_sq[i]) % 16 == 0,
                     "adam_step: tensor %d is not 16-byte aligned", i);
        tb.p[i] = d_params[i];
        tb.g[i] = d_grads[i];
        tb.m[i] = d_exp_avg[i];
        tb.v[i] = d_exp_avg_sq[i];
        tb.n[i] = numels[i];
        tb.chunk_start[i] = chunks;
        chunks += (int)((numels[i] + kAdamChunk - 1) / kAdamChunk);
    }
    tb.chunk_start[n_tensors] = chunks;
    hipLaunchKernelGGL(k_adam, dim3((unsigned)chunks), dim3(256), 0, (hipStream_t)stream, tb, lr, beta1, beta2, eps, (const i64 *)d_step);
    SRLX_HIP(hipGetLastError());
    return SRLX_OK;
}

}  // extern "C"
