// srlx_ppo_math.h -- the per-sample arithmetic of the PPO path, shared by the one-purpose kernels of srlx_ppo.hip (policy sampling, loss + gradient seeds,
// the Pendulum-shaped environment) and the fused network kernels of srlx_ppo_net.hip (whole rollout / whole minibatch in one launch): one definition, same bits.
// Reference lines: srl/algorithms/ppo/ppo.py:102-169 (compute_train_loss), :316-339 (policy), srl/rl/tf/distributions/normal_dist_block.py:13-20,64-74,144-149.
#pragma once
#include "srlx_common.h"

namespace srlxp {

using i64 = int64_t;
using u8 = unsigned char;
using u64 = unsigned long long;

constexpr float kHalfLog2Pi = 0.91893853320467274178f;  // 0.5 * log(2 pi)
constexpr float kLogFloor = -13.815510557964274f;      // math.log(1e-6), ppo.py:322

__device__ __forceinline__ float clampf(float x, float lo, float hi) { return fminf(fmaxf(x, lo), hi); }

// action = loc + exp(log_scale) * N(0,1) (Box-Muller on two keyed uniforms); log_prob per dimension, floored at log(1e-6).  i = flat index of (environment, dimension).
// The standard-normal draw depends on (seed, counter, index) only -- a rollout kernel draws all T steps' values up front (normal_z) and applies them per step
// (normal_act_from_z): the same expression, the same bits as normal_act_one.
__device__ __forceinline__ float normal_z(u64 seed, u64 c, i64 i) {
    const double u1 = 1.0 - srlx::u53(srlx::rng_u64(seed, c, (u64)(2 * i)));  // (0, 1]
    const double u2 = srlx::u53(srlx::rng_u64(seed, c, (u64)(2 * i + 1)));
    return (float)(sqrt(-2.0 * log(u1)) * cos(6.283185307179586 * u2));
}

__device__ __forceinline__ void normal_act_from_z(float loc, float log_scale, float ls_lo, float ls_hi, float z, int deterministic, float &action, float &logprob) {
    const float ls = clampf(log_scale, ls_lo, ls_hi);  // enable_stable_gradients clip, normal_dist_block.py:144-149
    const float sd = expf(ls);
    float a = loc;
    if (!deterministic) a = loc + sd * z;
    const float q = (a - loc) / sd;
    action = a;
    logprob = fmaxf(-kHalfLog2Pi - ls - 0.5f * (q * q), kLogFloor);
}

__device__ __forceinline__ void normal_act_one(float loc, float log_scale, float ls_lo, float ls_hi, u64 seed, u64 c, i64 i, int deterministic, float &action, float &logprob) {
    normal_act_from_z(loc, log_scale, ls_lo, ls_hi, deterministic ? 0.f : normal_z(seed, c, i), deterministic, action, logprob);
}

struct LossCfg {
    float ls_lo, ls_hi;
    int baseline_advantage, surrogate_clip, value_clip;
    float policy_clip, value_clip_range, value_w, entropy_w;
    float inv_b, inv_bk;  // 1 / B, 1 / (B * K)
};

// One (sample, dimension) of the policy loss given the log-probability lp of the taken action: surrogate term, entropy term, d loss / d lp
__device__ __forceinline__ float policy_lp_terms(const LossCfg &a, float lp, float old_lp, float adv, float &term, float &ent) {
    const float ratio = expf(lp - old_lp);  // :126
    float g_ratio;                          // d policy_term / d ratio
    if (a.surrogate_clip) {                 // :127-137
        const float rc = clampf(ratio, 1.0f - a.policy_clip, 1.0f + a.policy_clip);
        const float lu = ratio * adv, lc = rc * adv;
        term = fminf(lu, lc);
        g_ratio = lu <= lc ? adv : 0.f;  // tf.minimum routes the gradient to its first argument on ties
    } else {  // surrogate_type == "" (:148-149)
        term = ratio * adv;
        g_ratio = adv;
    }
    const float elp = expf(lp);
    ent = -elp * lp;  // :166
    // d loss / d lp: policy (-mean over B*K), entropy (weight * -mean over B of the per-sample sum)
    return -a.inv_bk * g_ratio * ratio + a.entropy_w * a.inv_b * (elp * lp + elp);
}

// Normal head: log-probability of `action` and the seeds d loss / d loc, d loss / d log_scale
__device__ __forceinline__ void policy_normal(const LossCfg &a, float loc, float ls_raw, float action, float old_lp, float adv, float &term, float &ent, float &d_loc, float &d_ls) {
    const float ls = clampf(ls_raw, a.ls_lo, a.ls_hi);
    const float q = (action - loc) / expf(ls);
    const float lp = -kHalfLog2Pi - ls - 0.5f * (q * q);  // normal_dist_block.py:13-20
    const float g_lp = policy_lp_terms(a, lp, old_lp, adv, term, ent);
    const bool pass = ls_raw >= a.ls_lo && ls_raw <= a.ls_hi;  // clip_by_value passes the gradient inside the range
    d_loc = g_lp * (q / expf(clampf(ls_raw, a.ls_lo, a.ls_hi)));
    d_ls = pass ? g_lp * (q * q - 1.0f) : 0.f;
}

// value loss :152-158: returns the summand, g_v = d loss / d v
__device__ __forceinline__ float value_term(const LossCfg &a, float v, float vt, float ov, float &g_v) {
    const float e1 = v - vt;
    float s, g;
    if (a.value_clip) {
        const float vc = clampf(v, ov - a.value_clip_range, ov + a.value_clip_range);
        const float e2 = vc - vt;
        const float l1 = e1 * e1, l2 = e2 * e2;
        s = fmaxf(l1, l2);
        // tf.maximum routes the gradient to its first argument on ties; the clipped branch only inside the range
        g = l1 >= l2 ? 2.0f * e1 : ((v >= ov - a.value_clip_range && v <= ov + a.value_clip_range) ? 2.0f * e2 : 0.f);
    } else {
        s = e1 * e1;
        g = 2.0f * e1;
    }
    g_v = a.value_w * a.inv_b * g;
    return s;
}

// Pendulum dynamics (the classic-control task config 5 is shaped on): th'' = 3g/(2l) sin th + 3/(m l^2) u; time limit = truncation with auto-reset
__device__ __forceinline__ void pendulum_one(float &th, float &thd, int &t, float action, i64 episode_len, u64 seed, u64 c, i64 e, float &o0, float &o1, float &o2, float &reward,
                                             u8 &done) {
    const float g = 10.0f, m = 1.0f, l = 1.0f, dt = 0.05f, max_speed = 8.0f, max_torque = 2.0f;
    const float u = clampf(action, -max_torque, max_torque);
    const float pi = 3.14159265358979323846f;
    float an = fmodf(th + pi, 2.0f * pi);
    if (an < 0.f) an += 2.0f * pi;
    an -= pi;  // angle_normalize
    reward = -(an * an + 0.1f * thd * thd + 0.001f * u * u);
    thd = clampf(thd + (3.0f * g / (2.0f * l) * sinf(th) + 3.0f / (m * l * l) * u) * dt, -max_speed, max_speed);
    th = th + thd * dt;
    t = t + 1;
    const bool end = t >= episode_len;  // a time limit: truncation, not termination
    done = end ? 1 : 0;
    if (end) {  // auto-reset: th ~ U(-pi, pi), thdot ~ U(-1, 1)
        th = (float)((2.0 * srlx::u53(srlx::rng_u64(seed ^ 0x70656e64ull, c, (u64)(2 * e))) - 1.0) * 3.14159265358979323846);
        thd = (float)(2.0 * srlx::u53(srlx::rng_u64(seed ^ 0x70656e64ull, c, (u64)(2 * e + 1))) - 1.0);
        t = 0;
    }
    o0 = cosf(th);
    o1 = sinf(th);
    o2 = thd;
}

}  // namespace srlxp
