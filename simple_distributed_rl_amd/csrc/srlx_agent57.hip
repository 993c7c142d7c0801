// srlx_agent57.hip -- what Agent57_light does AROUND its five networks, per lock-step and per update, as a handful of launches.
//
// Replaces, for E lock-stepped environments / one sampled batch (reference paths relative to the repository root):
//   srl/algorithms/agent57_light/agent57_light.py:355-375   Worker.policy: q = q_ext + beta[arm] q_int, epsilon[arm]-greedy          k_a57_policy
//   srl/algorithms/agent57_light/agent57_light.py:377-432   Worker.on_step: the item's fields, previous action / rewards, episode
//                                                           reward (the worker object's attributes, here per-environment arrays)      k_a57_post
//   srl/algorithms/agent57_light/agent57_light.py:288-311   Worker.on_reset: random previous action, zero previous rewards           k_a57_begin
//   srl/algorithms/agent57_light/agent57_light.py:165-216   change_batches_format: the UVFA inputs of a sampled batch                k_a57_gather
//   srl/algorithms/agent57_light/model_torch.py:70-99,341-348   the embedding network's tail (concat -> dense -> LayerNorm -> dense -> softmax), MSE
//                                                           against the one-hot action, backward, Adam                               k_a57_emb_tail
//   srl/algorithms/agent57_light/model_torch.py:105-117,353-362 the lifelong (RND) network's LayerNorm, MSE against the target network, backward, Adam
//                                                                                                                                     k_a57_rnd_tail
// Rounds 2-5 ran all of this as torch elementwise / indexing / hipBLASLt launches (~150 per lock-step, 62 % of the kernel time of the configs[3] workload);
// the image trunks and the first dense layers are srlx_qnet handles (head_mode 1 for the embedding / RND networks, UVFA columns for the Q-networks).
// The tails are tiny (B = 32 rows, 64 -> 128 -> A): one workgroup each, float32, fixed summation orders (deterministic run to run).
#include "srlx_adam_math.h"
#include "srlx_common.h"

namespace {

using i64 = int64_t;
using u8 = unsigned char;
using srlx::rng_u64;
using srlx::u53;
using srlx::u64;

// ---- actors ------------------------------------------------------------------------------------------------------------------------------------------------
struct PolicyArgs {
    i64 E;
    int A;
    const float *q_ext, *q_int;
    const int32_t *arm;                 // NULL: evaluation (arm 0, test_beta / test_epsilon: agent57_light.py:294-297)
    const float *beta_list, *eps_list;  // [actor_num]
    float test_beta, test_eps;
    u64 seed;
    const i64 *counter;                 // read only (the ring commit advances it)
    int32_t *actions;
    float *q_out;                       // [E][A] or NULL
};

__global__ void __launch_bounds__(256) k_a57_policy(PolicyArgs a) {
    const i64 e = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= a.E) return;
    const int arm = a.arm ? a.arm[e] : 0;
    const float beta = a.arm ? a.beta_list[arm] : a.test_beta, eps = a.arm ? a.eps_list[arm] : a.test_eps;
    const u64 c = (u64)a.counter[0];
    const int A = a.A;
    int act = 0;
    float bv = -INFINITY;
    for (int j = 0; j < A; j++) {
        const float q = a.q_ext[e * A + j] + beta * a.q_int[e * A + j];  // :363
        if (a.q_out) a.q_out[e * A + j] = q;
        if (j == 0 || q > bv) act = j, bv = q;  // np.argmax: first maximum
    }
    if (u53(rng_u64(a.seed, c, (u64)(2 * e))) < (double)eps) {  // random.random() < epsilon (:365-367)
        int pick = (int)(u53(rng_u64(a.seed, c, (u64)(2 * e + 1))) * (double)A);
        act = pick >= A ? A - 1 : pick;
    }
    a.actions[e] = act;
}

struct PostArgs {
    i64 E;
    const int32_t *actions, *arm;
    const float *rewards;
    const u8 *reset_lane;            // lanes whose lock-step only delivered a new episode's first frame (they took no action)
    const float *episodic, *lifelong;  // NULL: no intrinsic reward
    int32_t *prev_action;
    float *prev_r_ext, *prev_r_int, *episode_reward;
    // the item fields the frame store does not keep, row `slot` of [ring slot][env] arrays
    float *x_r_int, *x_prev_r_ext, *x_prev_r_int;
    int32_t *x_actor, *x_prev_action;
};

__global__ void __launch_bounds__(256) k_a57_post(PostArgs a) {
    const i64 e = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= a.E) return;
    const bool live = a.reset_lane[e] == 0;
    // what the lane held when it acted (:420-432)
    a.x_actor[e] = a.arm[e];
    a.x_prev_action[e] = a.prev_action[e];
    a.x_prev_r_ext[e] = a.prev_r_ext[e];
    a.x_prev_r_int[e] = a.prev_r_int[e];
    const float r_int = (live && a.episodic) ? a.episodic[e] * a.lifelong[e] : 0.f;  // :383-391
    a.x_r_int[e] = r_int;
    if (live) {  // :393-417
        a.prev_action[e] = a.actions[e];
        a.prev_r_ext[e] = a.rewards[e];
        a.prev_r_int[e] = r_int;
        a.episode_reward[e] = a.episode_reward[e] + a.rewards[e];
    }
}

// on_reset for the lanes whose episode has just ended (done; NULL: all): random previous action (keyed generator: the counter is only read -- the ring commit of
// the lock-step has advanced it), zero previous rewards and episode reward; reset_lane := done
__global__ void __launch_bounds__(256) k_a57_begin(i64 E, int A, const u8 *done, u64 seed, const i64 *counter, int32_t *prev_action, float *prev_r_ext, float *prev_r_int,
                                                   float *episode_reward, u8 *reset_lane, u8 *live_lane) {
    const i64 e = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= E) return;
    if (live_lane) live_lane[e] = done ? !done[e] : 1;
    if (!done || done[e]) {
        int a = (int)(u53(rng_u64(seed, (u64)counter[0], (u64)e)) * (double)A);  // random.randint(0, action_num - 1) (:300)
        prev_action[e] = a >= A ? A - 1 : a;
        prev_r_ext[e] = 0.f, prev_r_int[e] = 0.f, episode_reward[e] = 0.f;
    }
    if (reset_lane) reset_lane[e] = done ? done[e] : 0;
}

// ---- learner: the UVFA inputs of a sampled batch (rows interleaved like the online network's one pass: 2 b = s_0 with the inputs the actor saw, 2 b + 1 = s_1
// with what it saw one step later), the target network's rows (s_1 only), the sampled actor's discount, the intrinsic rewards ----------------------------------
struct GatherArgs {
    int B;
    i64 E;
    const i64 *loc_env, *loc_slot;
    const int32_t *actions;          // [B] (n_step = 1)
    const float *rewards;            // [B] extrinsic
    const float *x_r_int, *x_prev_r_ext, *x_prev_r_int;
    const int32_t *x_actor, *x_prev_action;
    const float *discount_list;
    float *on_r_ext, *on_r_int;      // [2 B]
    int32_t *on_action, *on_actor;   // [2 B]
    float *tg_r_ext, *tg_r_int;      // [B]
    int32_t *tg_action, *tg_actor;   // [B]
    float *discount, *r_int;         // [B]
};

__global__ void __launch_bounds__(64) k_a57_gather(GatherArgs a) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= a.B) return;
    const i64 at = a.loc_slot[b] * a.E + a.loc_env[b];
    const int actor = a.x_actor[at];
    const float ri = a.x_r_int[at], re = a.rewards[b];
    const int act = a.actions[b];
    a.on_r_ext[2 * b] = a.x_prev_r_ext[at], a.on_r_int[2 * b] = a.x_prev_r_int[at], a.on_action[2 * b] = a.x_prev_action[at], a.on_actor[2 * b] = actor;  // model_torch.py:427-433
    a.on_r_ext[2 * b + 1] = re, a.on_r_int[2 * b + 1] = ri, a.on_action[2 * b + 1] = act, a.on_actor[2 * b + 1] = actor;                                  // :294-299
    a.tg_r_ext[b] = re, a.tg_r_int[b] = ri, a.tg_action[b] = act, a.tg_actor[b] = actor;
    a.discount[b] = a.discount_list[actor];  // :287
    a.r_int[b] = ri;
}

// ---- multi-GPU: one rank's lock-step as a packed record (the layout srlx_store_commit_step_packed takes, with Agent57_light's five further item fields as extra
// floats: intrinsic reward, arm, previous action, previous extrinsic / intrinsic reward), and its inverse on the learner rank -----------------------------------
constexpr int kA57Fields = 5;
__global__ void __launch_bounds__(256) k_a57_pack(i64 E, const int32_t *actions, const float *rewards, const u8 *terminated, const u8 *done, const float *x_r_int,
                                                  const int32_t *x_actor, const int32_t *x_prev_action, const float *x_prev_r_ext, const float *x_prev_r_int, u8 *rec) {
    const i64 e = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= E) return;
    reinterpret_cast<int32_t *>(rec)[e] = actions[e];
    reinterpret_cast<float *>(rec + 4 * E)[e] = rewards[e];
    rec[8 * E + e] = terminated[e];
    rec[9 * E + e] = done[e];
    float *x = reinterpret_cast<float *>(rec + 10 * E) + e * kA57Fields;
    x[0] = x_r_int[e], x[1] = (float)x_actor[e], x[2] = (float)x_prev_action[e], x[3] = x_prev_r_ext[e], x[4] = x_prev_r_int[e];
}

// environment g of the learner's replay = lane g % per of record g / per; the fields land in row (*pos % L) of the learner's [ring slot][environment] arrays -- the
// slot the ring commit of the same slab writes (the position is read on the device: the launch replays inside a captured update)
__global__ void __launch_bounds__(256) k_a57_unpack(i64 total, i64 per, const u8 *rec, i64 stride, const i64 *pos, i64 L, float *x_r_int, int32_t *x_actor,
                                                    int32_t *x_prev_action, float *x_prev_r_ext, float *x_prev_r_int) {
    const i64 g = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= total) return;
    const i64 slot = ((pos[0] % L) + L) % L;
    const float *x = reinterpret_cast<const float *>(rec + (g / per) * stride + 10 * per) + (g % per) * kA57Fields;
    const i64 at = slot * total + g;
    x_r_int[at] = x[0], x_actor[at] = (int32_t)x[1], x_prev_action[at] = (int32_t)x[2], x_prev_r_ext[at] = x[3], x_prev_r_int[at] = x[4];
}

// ---- learner: the two small tails -----------------------------------------------------------------------------------------------------------------------
struct AdamHyper {
    double lr, beta1, beta2, eps;
    const i64 *d_step;  // optimiser steps already taken (device scalar; the caller advances it after the update)
};
struct Tensor3 {  // parameter, gradient out, optimiser state; m == NULL: no step (the gradient is still written)
    float *p, *g, *m, *v;
};
__device__ __forceinline__ void finish(const Tensor3 &t, i64 i, float g, const srlx::AdamCoef &c, bool adam, float *mirror = nullptr) {
    if (t.g) t.g[i] = g;
    if (adam && t.m) {
        float p = t.p[i], m = t.m[i], v = t.v[i];
        srlx::adam_one(p, g, m, v, c);
        t.p[i] = p, t.m[i] = m, t.v[i] = v;
        if (mirror) mirror[i] = p;
    } else if (mirror) {
        mirror[i] = t.p[i];
    }
}

// mean / rstd of one row of `n` values held in LDS, by one wave (biased variance, eps inside the root: nn.LayerNorm)
__device__ __forceinline__ void wave_ln_stats(const float *row, int n, float eps, int lane, float &mean, float &rstd) {
    float s = 0.f;
    for (int j = lane; j < n; j += 64) s += row[j];
    for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off);
    mean = s / (float)n;
    float s2 = 0.f;
    for (int j = lane; j < n; j += 64) s2 += (row[j] - mean) * (row[j] - mean);
    for (int off = 32; off > 0; off >>= 1) s2 += __shfl_xor(s2, off);
    rstd = 1.0f / sqrtf(s2 / (float)n + eps);
}

struct EmbTailArgs {
    int B, D, Hd, A;         // batch, embedding width (the input is 2 D wide), hidden width, actions
    const float *emb;        // [2 B][D]: rows 2 b = f(s), 2 b + 1 = f(s')
    const int32_t *actions;  // [B]
    Tensor3 w1, b1, lnw, lnb, w2, b2;  // out_block Linear [Hd][2 D], LayerNorm [Hd], out_block_out1 Linear [A][Hd]
    float ln_eps;
    float *loss;             // [1]
    float *d_emb;            // [2 B][D]
    AdamHyper ad;
};

__global__ void __launch_bounds__(256) k_a57_emb_tail(EmbTailArgs a) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int B = a.B, D2 = 2 * a.D, Hd = a.Hd, A = a.A;
    float *x = sm;                 // [B][D2]
    float *a1 = x + B * D2;        // [B][Hd]  post-ReLU
    float *xh = a1 + B * Hd;       // [B][Hd]  normalised
    float *dz = xh + B * Hd;       // [B][Hd]  dy, then dz1
    float *pr = dz + B * Hd;       // [B][A]   probabilities, then d logits
    float *rs = pr + B * A;        // [B]      1 / std
    float *red = rs + B;           // [256]
    // the tail's own parameters in LDS (a row of w1 padded by one float: "lane j reads row j" and "lane k reads column k" are both conflict-free) -- read from
    // global memory inside the loops below, the [Hd][2 D] weight cost a strided load per multiply-add and the launch 140 us; staged once it costs ~25
    const int W1 = D2 + 1;
    float *w1s = red + 256;        // [Hd][D2 + 1]
    float *w2s = w1s + Hd * W1;    // [A][Hd]
    float *gms = w2s + A * Hd;     // [Hd] LayerNorm weight
    float *bts = gms + Hd;         // [Hd] LayerNorm bias
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const bool adam = a.ad.d_step != nullptr;
    const srlx::AdamCoef cf = adam ? srlx::adam_coef(a.ad.lr, a.ad.beta1, a.ad.beta2, a.ad.eps, *a.ad.d_step) : srlx::AdamCoef{};
    for (int i = t; i < Hd * D2; i += 256) w1s[(i / D2) * W1 + i % D2] = a.w1.p[i];
    for (int i = t; i < A * Hd; i += 256) w2s[i] = a.w2.p[i];
    for (int i = t; i < Hd; i += 256) gms[i] = a.lnw.p[i], bts[i] = a.lnb.p[i];
    for (int i = t; i < B * D2; i += 256) x[i] = a.emb[i];  // (row b of x = rows 2 b, 2 b + 1 of emb back to back: torch.cat([f(s), f(s')], dim=1), model_torch.py:90)
    __syncthreads();
    for (int i = t; i < B * Hd; i += 256) {  // out_block: Linear + ReLU
        const int b = i / Hd, j = i % Hd;
        float s = a.b1.p[j];
        const float *w = w1s + j * W1, *xb = x + b * D2;
#pragma unroll 8
        for (int k = 0; k < D2; k++) s += w[k] * xb[k];
        a1[i] = s > 0.f ? s : 0.f;
    }
    __syncthreads();
    for (int b = wave; b < B; b += 4) {  // LayerNorm
        float mean, rstd;
        wave_ln_stats(a1 + b * Hd, Hd, a.ln_eps, lane, mean, rstd);
        for (int j = lane; j < Hd; j += 64) xh[b * Hd + j] = (a1[b * Hd + j] - mean) * rstd;
        if (lane == 0) rs[b] = rstd;
    }
    __syncthreads();
    for (int i = t; i < B * A; i += 256) {  // out_block_out1
        const int b = i / A, k = i % A;
        float s = a.b2.p[k];
        const float *w = w2s + k * Hd;
#pragma unroll 8
        for (int j = 0; j < Hd; j++) s += w[j] * (xh[b * Hd + j] * gms[j] + bts[j]);
        pr[i] = s;
    }
    __syncthreads();
    float lsum = 0.f;
    for (int b = t; b < B; b += 256) {  // softmax, MSE against the one-hot action (model_torch.py:343), d loss / d logits
        float *p = pr + b * A;
        float mx = p[0];
        for (int k = 1; k < A; k++) mx = p[k] > mx ? p[k] : mx;
        float z = 0.f;
        for (int k = 0; k < A; k++) p[k] = expf(p[k] - mx), z += p[k];
        float dot = 0.f;
        const float inv_n = 1.0f / (float)(B * A);
        for (int k = 0; k < A; k++) {
            p[k] = p[k] / z;
            const float d = p[k] - (a.actions[b] == k ? 1.f : 0.f);
            lsum += d * d;
            dot += (2.0f * d * inv_n) * p[k];
        }
        for (int k = 0; k < A; k++) {
            const float d = p[k] - (a.actions[b] == k ? 1.f : 0.f);
            p[k] = p[k] * (2.0f * d * inv_n - dot);  // softmax backward
        }
    }
    red[t] = lsum;
    __syncthreads();
    if (t == 0) {
        float s = 0.f;
        for (int i = 0; i < (B < 256 ? B : 256); i++) s += red[i];
        a.loss[0] = s / (float)(B * A);
    }
    // dy[b][j] = sum_k dl[b][k] w2[k][j]  (before w2 takes its step)
    for (int i = t; i < B * Hd; i += 256) {
        const int b = i / Hd, j = i % Hd;
        float s = 0.f;
        for (int k = 0; k < A; k++) s += pr[b * A + k] * w2s[k * Hd + j];
        dz[i] = s;
    }
    __syncthreads();
    // gradients of out_block_out1 (y = xh * gamma + beta with the gamma / beta of this forward)
    for (int i = t; i < A * Hd; i += 256) {
        const int k = i / Hd, j = i % Hd;
        const float gm = gms[j], bt = bts[j];
        float s = 0.f;
#pragma unroll 8
        for (int b = 0; b < B; b++) s += pr[b * A + k] * (xh[b * Hd + j] * gm + bt);
        a.w2.g[i] = s;  // (the step is applied below, once every reader of w2 / gamma / beta is through)
    }
    if (t < A) {
        float s = 0.f;
        for (int b = 0; b < B; b++) s += pr[b * A + t];
        a.b2.g[t] = s;
    }
    // LayerNorm backward: d gamma / d beta over the batch, dx per row
    for (int j = t; j < Hd; j += 256) {
        float sg = 0.f, sb = 0.f;
#pragma unroll 8
        for (int b = 0; b < B; b++) sg += dz[b * Hd + j] * xh[b * Hd + j], sb += dz[b * Hd + j];
        a.lnw.g[j] = sg, a.lnb.g[j] = sb;
    }
    __syncthreads();
    for (int b = wave; b < B; b += 4) {
        float m1 = 0.f, m2 = 0.f;
        for (int j = lane; j < Hd; j += 64) {
            const float dxh = dz[b * Hd + j] * gms[j];
            m1 += dxh, m2 += dxh * xh[b * Hd + j];
        }
        for (int off = 32; off > 0; off >>= 1) m1 += __shfl_xor(m1, off), m2 += __shfl_xor(m2, off);
        m1 /= (float)Hd, m2 /= (float)Hd;
        const float rstd = rs[b];
        for (int j = lane; j < Hd; j += 64) {
            const float dxh = dz[b * Hd + j] * gms[j];
            const float dx = rstd * ((dxh - m1) - xh[b * Hd + j] * m2);
            dz[b * Hd + j] = a1[b * Hd + j] > 0.f ? dx : 0.f;  // ReLU of out_block
        }
    }
    __syncthreads();
    // the tail's own parameters are read no more: out_block_out1 / LayerNorm take their steps
    if (adam) {
        for (int i = t; i < A * Hd; i += 256) finish(a.w2, i, a.w2.g[i], cf, true);
        if (t < A) finish(a.b2, t, a.b2.g[t], cf, true);
        for (int j = t; j < Hd; j += 256) finish(a.lnw, j, a.lnw.g[j], cf, true), finish(a.lnb, j, a.lnb.g[j], cf, true);
    }
    // d input (before w1 takes its step): d_emb row 2 b = columns 0..D-1 of row b, row 2 b + 1 = columns D..2D-1
    for (int i = t; i < B * D2; i += 256) {
        const int b = i / D2, k = i % D2;
        float s = 0.f;
#pragma unroll 8
        for (int j = 0; j < Hd; j++) s += dz[b * Hd + j] * w1s[j * W1 + k];
        a.d_emb[i] = s;
    }
    __syncthreads();
    // w1: eight elements per thread and pass -- their parameter / moment loads are issued BEFORE the sums (one element at a time, every optimiser step waited for
    // its own three loads behind the previous element's stores: 44 of the launch's 122 us)
    for (int i0 = t; i0 < Hd * D2; i0 += 8 * 256) {
        float pp[8], mm[8], vv[8], gs[8];
        const bool step = adam && a.w1.m;
#pragma unroll
        for (int q = 0; q < 8; q++) {
            const int i = i0 + 256 * q;
            const bool in = i < Hd * D2 && step;
            pp[q] = in ? a.w1.p[i] : 0.f, mm[q] = in ? a.w1.m[i] : 0.f, vv[q] = in ? a.w1.v[i] : 0.f;
        }
#pragma unroll
        for (int q = 0; q < 8; q++) {
            const int i = i0 + 256 * q, j = i / D2, k = i % D2;
            float s = 0.f;
            if (i < Hd * D2) {
#pragma unroll 8
                for (int b = 0; b < B; b++) s += dz[b * Hd + j] * x[b * D2 + k];
            }
            gs[q] = s;
        }
#pragma unroll
        for (int q = 0; q < 8; q++) {
            const int i = i0 + 256 * q;
            if (i >= Hd * D2) continue;
            if (a.w1.g) a.w1.g[i] = gs[q];
            if (step) {
                srlx::adam_one(pp[q], gs[q], mm[q], vv[q], cf);
                a.w1.p[i] = pp[q], a.w1.m[i] = mm[q], a.w1.v[i] = vv[q];
            }
        }
    }
    for (int j = t; j < Hd; j += 256) {
        float s = 0.f;
#pragma unroll 8
        for (int b = 0; b < B; b++) s += dz[b * Hd + j];
        finish(a.b1, j, s, cf, adam);
    }
}

struct RndTailArgs {
    int B, D;
    i64 ld;                // floats between consecutive rows of h / target (>= D: every second row of an interleaved [s_0, s_1] pass)
    const float *h;        // [B] rows of D: post-ReLU hidden layer of the predictor network
    const float *target;   // [B] rows of D: the target network's output (LayerNorm applied)
    Tensor3 lnw, lnb;
    float ln_eps;
    float *loss;           // [1]
    float *d_h;            // [B][D]
    float *mirror_w, *mirror_b;  // the updated LayerNorm parameters once more (the actors' copy of the set the update publishes into) or NULL
    AdamHyper ad;
};

__global__ void __launch_bounds__(256) k_a57_rnd_tail(RndTailArgs a) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int B = a.B, D = a.D;
    float *xh = sm;            // [B][D]
    float *dy = xh + B * D;    // [B][D]
    float *rs = dy + B * D;    // [B]
    float *red = rs + B;       // [4]
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const bool adam = a.ad.d_step != nullptr;
    const srlx::AdamCoef cf = adam ? srlx::adam_coef(a.ad.lr, a.ad.beta1, a.ad.beta2, a.ad.eps, *a.ad.d_step) : srlx::AdamCoef{};
    for (int i = t; i < B * D; i += 256) xh[i] = a.h[(i64)(i / D) * a.ld + i % D];
    __syncthreads();
    float lsum = 0.f;
    const float inv_n = 1.0f / (float)(B * D);
    for (int b = wave; b < B; b += 4) {
        float mean, rstd;
        wave_ln_stats(xh + b * D, D, a.ln_eps, lane, mean, rstd);
        for (int j = lane; j < D; j += 64) {
            const float n = (xh[b * D + j] - mean) * rstd;
            const float y = n * a.lnw.p[j] + a.lnb.p[j];
            const float d = y - a.target[(i64)b * a.ld + j];  // mse_loss(target, train) (model_torch.py:357)
            lsum += d * d;
            xh[b * D + j] = n;
            dy[b * D + j] = 2.0f * d * inv_n;
        }
        if (lane == 0) rs[b] = rstd;
    }
    for (int off = 32; off > 0; off >>= 1) lsum += __shfl_xor(lsum, off);
    if (lane == 0) red[wave] = lsum;
    __syncthreads();
    if (t == 0) a.loss[0] = (((red[0] + red[1]) + red[2]) + red[3]) * inv_n;
    for (int b = wave; b < B; b += 4) {  // LayerNorm backward + the ReLU of the hidden layer is the caller's (k_hidden_bwd masks by h1 > 0)
        float m1 = 0.f, m2 = 0.f;
        for (int j = lane; j < D; j += 64) {
            const float dxh = dy[b * D + j] * a.lnw.p[j];
            m1 += dxh, m2 += dxh * xh[b * D + j];
        }
        for (int off = 32; off > 0; off >>= 1) m1 += __shfl_xor(m1, off), m2 += __shfl_xor(m2, off);
        m1 /= (float)D, m2 /= (float)D;
        for (int j = lane; j < D; j += 64) {
            const float dxh = dy[b * D + j] * a.lnw.p[j];
            a.d_h[b * D + j] = rs[b] * ((dxh - m1) - xh[b * D + j] * m2);
        }
    }
    __syncthreads();
    for (int j = t; j < D; j += 256) {
        float sg = 0.f, sb = 0.f;
        for (int b = 0; b < B; b++) sg += dy[b * D + j] * xh[b * D + j], sb += dy[b * D + j];
        finish(a.lnw, j, sg, cf, adam, a.mirror_w);
        finish(a.lnb, j, sb, cf, adam, a.mirror_b);
    }
}

}  // namespace

extern "C" {

int srlx_agent57_policy(int64_t n_envs, int n_actions, const float *d_q_ext, const float *d_q_int, const int32_t *d_arm, const float *d_beta_list, const float *d_eps_list,
                        double test_beta, double test_epsilon, uint64_t seed, const int64_t *d_counter, int32_t *d_actions, float *d_q_out, void *stream) {
    SRLX_REQUIRE(n_envs > 0 && n_actions >= 1 && d_q_ext && d_q_int && d_counter && d_actions && (!d_arm || (d_beta_list && d_eps_list)), "agent57_policy: bad argument");
    PolicyArgs a{n_envs, n_actions, d_q_ext, d_q_int, d_arm, d_beta_list, d_eps_list, (float)test_beta, (float)test_epsilon, (u64)seed, d_counter, d_actions, d_q_out};
    hipLaunchKernelGGL(k_a57_policy, dim3((unsigned)((n_envs + 255) / 256)), dim3(256), 0, (hipStream_t)stream, a);
    SRLX_HIP(hipGetLastError());
    return SRLX_OK;
}

int srlx_agent57_post_step(int64_t n_envs, const int32_t *d_actions, const int32_t *d_arm, const float *d_rewards, const uint8_t *d_reset_lane, const float *d_episodic,
                           const float *d_lifelong, int32_t *d_prev_action, float *d_prev_r_ext, float *d_prev_r_int, float *d_episode_reward, float *d_x_r_int,
                           float *d_x_prev_r_ext, float *d_x_prev_r_int, int32_t *d_x_actor, int32_t *d_x_prev_action, void *stream) {
    SRLX_REQUIRE(n_envs > 0 && d_actions && d_arm && d_rewards && d_reset_lane && d_prev_action && d_prev_r_ext && d_prev_r_int && d_episode_reward && d_x_r_int &&
                     d_x_prev_r_ext && d_x_prev_r_int && d_x_actor && d_x_prev_action && (!d_episodic == !d_lifelong),
                 "agent57_post_step: bad argument");
    PostArgs a{n_envs, d_actions, d_arm, d_rewards, d_reset_lane, d_episodic, d_lifelong, d_prev_action, d_prev_r_ext, d_prev_r_int, d_episode_reward, d_x_r_int,
               d_x_prev_r_ext, d_x_prev_r_int, d_x_actor, d_x_prev_action};
    hipLaunchKernelGGL(k_a57_post, dim3((unsigned)((n_envs + 255) / 256)), dim3(256), 0, (hipStream_t)stream, a);
    SRLX_HIP(hipGetLastError());
    return SRLX_OK;
}

int srlx_agent57_begin_episodes(int64_t n_envs, int n_actions, const uint8_t *d_done, uint64_t seed, const int64_t *d_counter, int32_t *d_prev_action, float *d_prev_r_ext,
                                float *d_prev_r_int, float *d_episode_reward, uint8_t *d_reset_lane, uint8_t *d_live_lane, void *stream) {
    SRLX_REQUIRE(n_envs > 0 && n_actions >= 1 && d_counter && d_prev_action && d_prev_r_ext && d_prev_r_int && d_episode_reward, "agent57_begin_episodes: bad argument");
    hipLaunchKernelGGL(k_a57_begin, dim3((unsigned)((n_envs + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (i64)n_envs, n_actions, d_done, (u64)seed, d_counter,
                       d_prev_action, d_prev_r_ext, d_prev_r_int, d_episode_reward, d_reset_lane, d_live_lane);
    SRLX_HIP(hipGetLastError());
    return SRLX_OK;
}

int srlx_agent57_gather_inputs(int64_t batch, int64_t n_envs, const int64_t *d_loc_env, const int64_t *d_loc_slot, const int32_t *d_actions, const float *d_rewards,
                               const float *d_x_r_int, const float *d_x_prev_r_ext, const float *d_x_prev_r_int, const int32_t *d_x_actor, const int32_t *d_x_prev_action,
                               const float *d_discount_list, float *d_on_r_ext, float *d_on_r_int, int32_t *d_on_action, int32_t *d_on_actor, float *d_tg_r_ext,
                               float *d_tg_r_int, int32_t *d_tg_action, int32_t *d_tg_actor, float *d_discount, float *d_r_int, void *stream) {
    SRLX_REQUIRE(batch > 0 && n_envs > 0 && d_loc_env && d_loc_slot && d_actions && d_rewards && d_x_r_int && d_x_prev_r_ext && d_x_prev_r_int && d_x_actor && d_x_prev_action &&
                     d_discount_list && d_on_r_ext && d_on_r_int && d_on_action && d_on_actor && d_tg_r_ext && d_tg_r_int && d_tg_action && d_tg_actor && d_discount && d_r_int,
                 "agent57_gather_inputs: NULL argument");
    GatherArgs a{(int)batch, n_envs, d_loc_env, d_loc_slot, d_actions, d_rewards, d_x_r_int, d_x_prev_r_ext, d_x_prev_r_int, d_x_actor, d_x_prev_action, d_discount_list,
                 d_on_r_ext, d_on_r_int, d_on_action, d_on_actor, d_tg_r_ext, d_tg_r_int, d_tg_action, d_tg_actor, d_discount, d_r_int};
    hipLaunchKernelGGL(k_a57_gather, dim3((unsigned)((batch + 63) / 64)), dim3(64), 0, (hipStream_t)stream, a);
    SRLX_HIP(hipGetLastError());
    return SRLX_OK;
}

int srlx_agent57_pack_record(int64_t n_envs, const int32_t *d_actions, const float *d_rewards, const uint8_t *d_terminated, const uint8_t *d_done, const float *d_x_r_int,
                             const int32_t *d_x_actor, const int32_t *d_x_prev_action, const float *d_x_prev_r_ext, const float *d_x_prev_r_int, uint8_t *d_record, void *stream) {
    SRLX_REQUIRE(n_envs > 0 && n_envs % 4 == 0 && d_actions && d_rewards && d_terminated && d_done && d_x_r_int && d_x_actor && d_x_prev_action && d_x_prev_r_ext && d_x_prev_r_int &&
                     d_record, "agent57_pack_record: bad argument (n_envs in multiples of 4: the float fields sit behind 10 * n_envs bytes)");
    hipLaunchKernelGGL(k_a57_pack, dim3((unsigned)((n_envs + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (i64)n_envs, d_actions, d_rewards, d_terminated, d_done, d_x_r_int,
                       d_x_actor, d_x_prev_action, d_x_prev_r_ext, d_x_prev_r_int, d_record);
    SRLX_HIP(hipGetLastError());
    return SRLX_OK;
}

int srlx_agent57_unpack_fields(int64_t n_records, int64_t envs_per_record, const uint8_t *d_records, int64_t record_stride, const int64_t *d_position, int64_t ring_len,
                               float *d_x_r_int, int32_t *d_x_actor, int32_t *d_x_prev_action, float *d_x_prev_r_ext, float *d_x_prev_r_int, void *stream) {
    SRLX_REQUIRE(n_records > 0 && envs_per_record > 0 && envs_per_record % 4 == 0 && d_records && record_stride >= (10 + 4 * kA57Fields) * envs_per_record && d_position &&
                     ring_len > 0 && d_x_r_int && d_x_actor && d_x_prev_action && d_x_prev_r_ext && d_x_prev_r_int, "agent57_unpack_fields: bad argument");
    const i64 total = n_records * envs_per_record;
    hipLaunchKernelGGL(k_a57_unpack, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, total, (i64)envs_per_record, d_records, (i64)record_stride, d_position,
                       (i64)ring_len, d_x_r_int, d_x_actor, d_x_prev_action, d_x_prev_r_ext, d_x_prev_r_int);
    SRLX_HIP(hipGetLastError());
    return SRLX_OK;
}

// params / grads / exp_avg / exp_avg_sq: HOST arrays of 6 device pointers in the order out_block weight [hidden][2 emb_dim], out_block bias, LayerNorm weight, LayerNorm
// bias, out_block_out1 weight [n_actions][hidden], out_block_out1 bias.  exp_avg == NULL (or d_steps_taken == NULL): gradients only.
int srlx_agent57_emb_tail(int64_t batch, int emb_dim, int hidden, int n_actions, const float *d_emb, const int32_t *d_actions, float *const *d_params, float *const *d_grads,
                          float *const *d_exp_avg, float *const *d_exp_avg_sq, double ln_eps, double lr, double beta1, double beta2, double eps, const int64_t *d_steps_taken,
                          float *d_loss, float *d_grad_emb, void *stream) {
    SRLX_REQUIRE(batch > 0 && batch <= 64 && emb_dim > 0 && hidden > 0 && n_actions >= 1 && n_actions <= 32 && d_emb && d_actions && d_params && d_grads && d_loss && d_grad_emb,
                 "agent57_emb_tail: bad argument (batch <= 64, n_actions <= 32)");
    for (int k = 0; k < 6; k++) SRLX_REQUIRE(d_params[k] && d_grads[k], "agent57_emb_tail: parameter / gradient %d is NULL", k);
    const bool adam = d_exp_avg && d_exp_avg_sq && d_steps_taken;
    Tensor3 t[6];
    for (int k = 0; k < 6; k++) t[k] = Tensor3{d_params[k], d_grads[k], adam ? d_exp_avg[k] : nullptr, adam ? d_exp_avg_sq[k] : nullptr};
    EmbTailArgs a{(int)batch, emb_dim, hidden, n_actions, d_emb, d_actions, t[0], t[1], t[2], t[3], t[4], t[5], (float)ln_eps, d_loss, d_grad_emb,
                  AdamHyper{lr, beta1, beta2, eps, adam ? d_steps_taken : nullptr}};
    const size_t lds = ((size_t)batch * 2 * emb_dim + 3 * (size_t)batch * hidden + (size_t)batch * n_actions + batch + 256 + (size_t)hidden * (2 * emb_dim + 1) +
                        (size_t)n_actions * hidden + 2 * (size_t)hidden) * sizeof(float);
    SRLX_REQUIRE(lds <= 160 * 1024, "agent57_emb_tail: %zu bytes of LDS (batch x hidden too large)", lds);
    static size_t lds_set = 0;
    if (lds > lds_set) {
        SRLX_HIP(hipFuncSetAttribute((const void *)k_a57_emb_tail, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        lds_set = lds;
    }
    hipLaunchKernelGGL(k_a57_emb_tail, dim3(1), dim3(256), lds, (hipStream_t)stream, a);
    SRLX_HIP(hipGetLastError());
    return SRLX_OK;
}

int srlx_agent57_rnd_tail(int64_t batch, int dim, int64_t row_stride, const float *d_hidden, const float *d_target, float *d_ln_w, float *d_ln_b, float *d_grad_ln_w, float *d_grad_ln_b,
                          float *d_exp_avg_w, float *d_exp_avg_sq_w, float *d_exp_avg_b, float *d_exp_avg_sq_b, float *d_mirror_w, float *d_mirror_b, double ln_eps, double lr,
                          double beta1, double beta2, double eps, const int64_t *d_steps_taken, float *d_loss, float *d_grad_hidden, void *stream) {
    SRLX_REQUIRE(batch > 0 && batch <= 64 && dim > 0 && row_stride >= dim && d_hidden && d_target && d_ln_w && d_ln_b && d_loss && d_grad_hidden,
                 "agent57_rnd_tail: bad argument (batch <= 64, row_stride >= dim)");
    const bool adam = d_exp_avg_w && d_exp_avg_sq_w && d_exp_avg_b && d_exp_avg_sq_b && d_steps_taken;
    RndTailArgs a{(int)batch, dim, (i64)row_stride, d_hidden, d_target, Tensor3{d_ln_w, d_grad_ln_w, adam ? d_exp_avg_w : nullptr, adam ? d_exp_avg_sq_w : nullptr},
                  Tensor3{d_ln_b, d_grad_ln_b, adam ? d_exp_avg_b : nullptr, adam ? d_exp_avg_sq_b : nullptr}, (float)ln_eps, d_loss, d_grad_hidden, d_mirror_w, d_mirror_b,
                  AdamHyper{lr, beta1, beta2, eps, adam ? d_steps_taken : nullptr}};
    const size_t lds = (2 * (size_t)batch * dim + batch + 4) * sizeof(float);
    SRLX_REQUIRE(lds <= 64 * 1024, "agent57_rnd_tail: %zu bytes of LDS", lds);
    hipLaunchKernelGGL(k_a57_rnd_tail, dim3(1), dim3(256), lds, (hipStream_t)stream, a);
    SRLX_HIP(hipGetLastError());
    return SRLX_OK;
}

}  // extern "C"
