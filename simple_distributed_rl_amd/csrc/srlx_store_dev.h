// srlx_store_dev.h -- the device-side view of the transition store (srlx_rollout.hip) and the item look-ups on it, shared with srlx_per.hip so that the
// learner's PER draw and its gather (item location, n-step scalars, frame-offset tables) can run as ONE launch (srlx_per_sample_gather_train).
#pragma once
#include "srlx_common.h"

namespace srlxs {

using i64 = int64_t;
using u64 = unsigned long long;
using u8 = unsigned char;

constexpr u8 kTerm = 1, kDone = 2, kInvalid = 4;

using srlx::mix64;
using srlx::rng_u64;
using srlx::u53;

__device__ __forceinline__ i64 posmod(i64 a, i64 m) {
    i64 r = a % m;
    return r < 0 ? r + m : r;
}

struct StoreDev {
    i64 E, L, F;
    int obs_dtype, W, n, A, reward_clip;
    u64 seed;
    i64 item_len;
    void *obs;
    int32_t *action;
    float *reward;
    u8 *flags;
    int32_t *step_in_ep;
    i64 *pos;  // [0] ring position p, [1] rng counter
    u8 *needs_reset;
};

struct ItemMeta {
    i64 e, q;
    int jd;  // first transition index that ended the episode (n if none)
    int pad;
};

// per sampled item: locate (env, position), find the episode end inside the window, emit the scalars.
// (Round 5: the ring position is a 64-bit count, and every posmod() of it was a 64-bit software division -- ten per item, ~2.5 us of the learner's draw + gather launch.
// Two are left, of the position itself; everything relative to them is 32-bit wrap-around arithmetic.  The episode-end scan loads its n flags without a break: the loads
// no longer wait for one another.)
__device__ __forceinline__ int wrap32(int r, int m) { return r >= m ? r - m : (r < 0 ? r + m : r); }  // r in (-m, 2m)
__device__ __forceinline__ ItemMeta item_meta(const StoreDev &s, i64 b, const i64 *tree_idx, int32_t *actions, float *rewards, float *terminated) {
    ItemMeta out;
    const i64 N = s.E * s.item_len;
    i64 j = tree_idx[b] - (N - 1);
    if (j < 0) j = 0;
    if (j >= N) j = N - 1;
    i64 e, tau;
    if (N < ((i64)1 << 31)) {  // (32-bit division: every store of this build)
        const unsigned uj = (unsigned)j, uE = (unsigned)s.E;
        e = uj % uE, tau = uj / uE;
    } else {
        e = j % s.E, tau = j / s.E;
    }
    const i64 p_last = s.pos[0] - 1;
    const int il = (int)s.item_len, L = (int)s.L;
    const int d = wrap32((int)posmod(p_last, s.item_len) - (int)tau, il);  // posmod(p_last - tau, item_len): tau < item_len
    const i64 p_add = p_last - d;
    i64 q = p_add - (s.n - 1);
    if (q < 0) q = 0;
    const int qm = (int)posmod(q, s.L);
    const i64 base = e * s.L;
    int jd = s.n;
    for (int k = s.n - 1; k >= 0; k--)
        if (s.flags[base + wrap32(qm + k, L)] & kDone) jd = k;  // (the first transition that ended the episode)
    out.e = e;
    out.q = q;
    out.jd = jd;
    out.pad = qm;  // q mod L, for frame_offset_q
    for (int k = 0; k < s.n; k++) {
        const i64 r = base + wrap32(qm + k, L);
        if (k <= jd) {
            actions[b * s.n + k] = s.action[r];
            rewards[b * s.n + k] = s.reward[r];
            terminated[b * s.n + k] = (s.flags[r] & kTerm) ? 1.f : 0.f;
        } else {
            // terminal padding (rainbow.py:354-372): random action, reward 0, terminated 1.  The action is
            // a fixed function of the item so that re-sampling the item reproduces it.
            actions[b * s.n + k] = (int32_t)(rng_u64(s.seed ^ 0x70616464ull, (u64)(e * 0x100000000ll + (q & 0xffffffffll)), (u64)k) % (u64)s.A);
            rewards[b * s.n + k] = 0.f;
            terminated[b * s.n + k] = 1.f;
        }
    }
    return out;
}
// frame c of the stack at position q + kk of environment e, given qm = q mod L (ItemMeta.pad): frame_offset(s, e, q + kk, c) without the 64-bit divisions
__device__ __forceinline__ i64 frame_offset_q(const StoreDev &s, i64 e, int qm, int kk, int c) {
    const int back = s.W - 1 - c, L = (int)s.L;
    if (back > s.step_in_ep[e * s.L + wrap32(qm + kk, L)]) return -1;
    return (e * s.L + wrap32(qm + kk - back, L)) * s.F;  // (kk <= n, back < W, n + W <= L)
}
__device__ __forceinline__ i64 frame_offset(const StoreDev &s, i64 e, i64 x, int c) {
    const int back = s.W - 1 - c;
    if (back > s.step_in_ep[e * s.L + posmod(x, s.L)]) return -1;
    return (e * s.L + posmod(x - back, s.L)) * s.F;
}

// the learner's whole "gather" for up to kTrainItems items by one workgroup of 256 threads: item location + n-step scalars (item_meta) and both offset tables
// (s_0..s_n for the online network, s_1..s_n for the target network); `sm` = kTrainItems ItemMeta in LDS; b0 = first item, cnt = items of this workgroup
constexpr int kTrainItems = 64;
__device__ __forceinline__ void gather_train_items(const StoreDev &s, i64 b0, int cnt, const i64 *tree_idx, ItemMeta *meta, int32_t *actions, float *rewards,
                                                   float *terminated, i64 *off_all, i64 *off_next, ItemMeta *sm) {
    const int t = threadIdx.x;
    if (t < cnt) {
        sm[t] = item_meta(s, b0 + t, tree_idx, actions, rewards, terminated);
        if (meta) meta[b0 + t] = sm[t];
    }
    __syncthreads();
    const int S = s.n + 1, W = s.W;
    for (int x = t; x < cnt * S * W; x += blockDim.x) {
        const int c = x % W, k = (x / W) % S, bl = x / (W * S);
        const ItemMeta m = sm[bl];
        const int kk = k < m.jd + 1 ? k : m.jd + 1;  // states after the terminal one repeat it (rainbow.py:358)
        const i64 off = frame_offset_q(s, m.e, m.pad, kk, c);
        off_all[((b0 + bl) * S + k) * W + c] = off;
        if (off_next && k >= 1) off_next[((b0 + bl) * s.n + (k - 1)) * W + c] = off;
    }
}

}  // namespace srlxs

// srlx_rollout.hip: the device view of a store handle + a scratch array of `batch` ItemMeta (grown on demand; not during graph capture)
int srlx_store_dev_view(srlx_store_t *h, int64_t batch, srlxs::StoreDev *out, srlxs::ItemMeta **meta);
