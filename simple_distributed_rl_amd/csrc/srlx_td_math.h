// srlx_td_math.h -- the n-step / retrace TD target, Huber loss, gradient seed and priority of one sampled item, shared by the
// stand-alone kernel k_nstep_td_huber_priority (srlx_train.hip) and the backward pass's head kernel, which can evaluate it in
// its prologue (srlx_qnet_backward_td_u8, srlx_qnet_bwd.hip).  Replaces the numpy between the network calls:
//   srl/algorithms/rainbow/rainbow.py:226-287         (calc_target_q after the two forwards)
//   srl/algorithms/rainbow/model_torch.py:103-105,113 (selected Q, HuberLoss(target*w, q*w), |target-q|)
// float32 arithmetic follows numpy's evaluation order of the cited lines.
#pragma once
#include <math.h>

#include "srlx_common.h"

namespace srlx {
using i64 = int64_t;
using u8 = unsigned char;

// srl/rl/functions.py:10-17 evaluated in float32 like numpy does on a float32 array
__device__ __forceinline__ float signf(float x) { return x > 0.f ? 1.f : (x < 0.f ? -1.f : 0.f); }
__device__ __forceinline__ float f_sqrt(float x) { return (float)__dsqrt_rn((double)x); }  // correctly rounded fp32 sqrt
__device__ __forceinline__ float rescaling(float x) {
    const float eps = 0.001f;
    return signf(x) * (f_sqrt(fabsf(x) + 1.0f) - 1.0f) + eps * x;
}
__device__ __forceinline__ float inverse_rescaling(float x) {
    const float eps = 0.001f;
    float n = f_sqrt(1.0f + (float)(4.0 * 0.001) * ((fabsf(x) + 1.0f) + eps)) - 1.0f;
    n = n / (float)(2.0 * 0.001);
    return signf(x) * ((n * n) - 1.0f);
}

__device__ __forceinline__ int argmax_masked(const float *q, const u8 *inv, int A) {
    int best = 0;
    float bv = 0.f;
    for (int a = 0; a < A; a++) {
        const float v = (inv && inv[a]) ? -INFINITY : q[a];
        if (a == 0 || v > bv) {  // np.argmax: first maximum
            best = a;
            bv = v;
        }
    }
    return best;
}

constexpr int kTdMaxStep = 32;

struct TdArgs {
    i64 B;
    int n, A;
    const float *q_on_next, *q_tg_next, *q_on_0;
    const int32_t *actions;
    const float *rewards, *terminated;
    const u8 *invalid_next;
    const float *weights;
    double discount, retrace_h;
    int double_dqn, rescale;
    float *target, *loss, *grad_q0, *priorities;
    i64 on_next_stride, on_0_stride;  // floats between consecutive items (dense: n*A and A; packed [B][n+1][A]: (n+1)*A for both)
    // multi_discounts (rainbow.py:182): float32(discount ** m), evaluated ON THE HOST by td_fill_discounts -- the C library's pow(), which is what Python's `**`
    // calls -- instead of three device pow() calls per item in the head kernel's prologue (round 5: ~1000 instructions each on the update's critical chain)
    float dm[kTdMaxStep];
    // round 6 (Agent57_light's 1-step targets, agent57_light.py:218-268 + model_torch.py:384-443; srlx_qnet_set_td_extras): a per-sample discount (the sampled
    // actor's gamma, float32 [B]; NULL: `discount`) and the SIGNED TD error target - q (NULL: not stored; the priorities are mixed from both networks' errors)
    const float *disc_ps;
    float *td_signed;
};

inline void td_fill_discounts(TdArgs &a) {
    for (int m = 0; m < kTdMaxStep; m++) a.dm[m] = m < a.n ? (float)pow(a.discount, (double)m) : 0.f;
}

// rows t, t + T, ... of the batch: target / gradient seed / priority of each; returns this thread's sum of Huber terms (float64).
// `grad_out` [B][A] may be LDS; `write_global`: also store target, priorities (and grad_q0 when grad_out is elsewhere).
__device__ __forceinline__ double td_rows(const TdArgs &a, int t, int T, float *grad_out, bool write_global) {
    const int n = a.n, A = a.A;
    double loss_acc = 0.0;
    for (i64 b = t; b < a.B; b += T) {
        const float disc_f = a.disc_ps ? a.disc_ps[b] : (float)a.discount;
        const float *qon = a.q_on_next + b * a.on_next_stride;
        const float *qtg = a.q_tg_next + b * n * A;
        const int32_t *act = a.actions + b * n;
        const u8 *inv = a.invalid_next ? a.invalid_next + b * n * A : nullptr;
        int nact[kTdMaxStep];
        float td[kTdMaxStep];
        for (int m = 0; m < n; m++) {
            // rainbow.py:245-253: greedy next action from the online net (double DQN) or the target net
            const float *sel = a.double_dqn ? qon + m * A : qtg + m * A;
            nact[m] = argmax_masked(sel, inv ? inv + m * A : nullptr, A);
            float maxq = qtg[m * A + nact[m]];
            if (a.rescale) maxq = inverse_rescaling(maxq);  // :255-256
            float gain = a.rewards[b * n + m] + ((1.0f - a.terminated[b * n + m]) * disc_f) * maxq;  // :258
            if (a.rescale) gain = rescaling(gain);  // :260-261
            // :231-233 action value of the online net for steps 1..n-1, 0 for the first step
            const float qsel = (m == 0) ? 0.f : qon[(m - 1) * A + act[m]];
            td[m] = gain - qsel;  // :263
        }
        // :268-286 retrace coefficients (float64 in the reference) and the discounted sum in float32
        double c = 1.0;
        float target = 0.f;
        for (int m = 0; m < n; m++) {
            if (m > 0) {
                const bool pi = (act[m] == nact[m]);  // argmax(n_action)[m-1] == n_act_idx[:,1:][m-1]
                c *= a.retrace_h * (pi ? 1.0 : 0.0);
            }
            const float dm = a.dm[m];  // float32(discount ** m) (:182), from the host
            const float term = (float)((double)(td[m] * dm) * c);
            target = target + term;
        }
        if (write_global) a.target[b] = target;

        // model_torch.py:103-105,113
        const int a0 = act[0];
        const float q0 = a.q_on_0[b * a.on_0_stride + a0];
        const float w = a.weights[b];
        const float tw = target * w, qw = q0 * w;
        const float diff = tw - qw;
        const float z = fabsf(diff);
        loss_acc += (z < 1.0f) ? 0.5 * (double)z * (double)z : (double)z - 0.5;  // HuberLoss, delta = 1
        const float dclamp = diff > 1.0f ? 1.0f : (diff < -1.0f ? -1.0f : diff);
        const float gsel = -(w * dclamp) / (float)a.B;  // d mean(huber(tw - q*w)) / d q
        for (int k = 0; k < A; k++) grad_out[b * A + k] = k == a0 ? gsel : 0.f;
        if (write_global) {
            if (grad_out != a.grad_q0)
                for (int k = 0; k < A; k++) a.grad_q0[b * A + k] = k == a0 ? gsel : 0.f;
            a.priorities[b] = fabsf(target - q0);
            if (a.td_signed) a.td_signed[b] = target - q0;  // model_torch.py:442
        }
    }
    return loss_acc;
}

}  // namespace srlx
