// srlx_episode.hip -- episode bookkeeping for E device-resident environments.
//
// Replaces, for the vectorised sequence loop, the host-side accounting the reference does per single step:
//   srl/base/env/env_run.py:334-352       (step counter, per-episode reward sums)
//   srl/base/run/core_play.py:200-214     (episode_rewards_list / last_episode_* at episode end)
// The environments' rewards and done flags never leave HBM; this kernel keeps a running return and length per
// environment and appends every episode that ends in the lock-step to a ring of (return, length) records -- in
// environment order, so the record sequence is deterministic -- plus running totals the host reads through a pinned
// mailbox whenever it wants (no per-step synchronisation).
#include "srlx_common.h"

namespace {
using i64 = int64_t;
using u8 = unsigned char;

struct EpTotals {
    i64 episodes;       // episodes finished since the ledger was cleared
    i64 steps;          // environment steps accounted
    double return_sum;  // sum of the finished episodes' returns
    i64 length_sum;     // sum of the finished episodes' lengths
};

__global__ void __launch_bounds__(1024) k_episode_account(i64 E, const float *__restrict__ rewards, const u8 *__restrict__ done,
                                                          const u8 *__restrict__ skip, float *__restrict__ ep_return, int32_t *__restrict__ ep_len, float *__restrict__ ring, i64 cap,
                                                          EpTotals *__restrict__ tot) {
    __shared__ int wcount[16], wlive[16];
    __shared__ double wret[16];
    __shared__ i64 wlen[16];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const i64 ep0 = tot->episodes;
    i64 appended = 0, stepped = 0;
    double ret_acc = 0.0;
    i64 len_acc = 0;
    for (i64 c0 = 0; c0 < E; c0 += 1024) {
        const i64 e = c0 + t;
        bool fin = false;
        float R = 0.f;
        int L = 0;
        const bool live = e < E && !(skip && skip[e]);
        if (live) {
            R = ep_return[e] + rewards[e];
            L = ep_len[e] + 1;
            fin = done[e] != 0;
            ep_return[e] = fin ? 0.f : R;
            ep_len[e] = fin ? 0 : L;
        }
        const unsigned long long m = __ballot(fin);
        const unsigned long long ml = __ballot(live);
        const int before = __popcll(m & ((1ull << lane) - 1ull));
        // fixed-order sums: lanes in order inside a wave (shuffle tree), then waves in order
        double r = fin ? (double)R : 0.0;
        i64 l = fin ? (i64)L : 0;
        if (lane == 0) wlive[wave] = __popcll(ml);
        for (int off = 32; off > 0; off >>= 1) {
            r += __shfl_down(r, off);
            l += __shfl_down(l, off);
        }
        if (lane == 0) wcount[wave] = __popcll(m), wret[wave] = r, wlen[wave] = l;
        __syncthreads();
        int woff = 0, total = 0;
        for (int w = 0; w < 16; w++) {
            if (w < wave) woff += wcount[w];
            total += wcount[w];
        }
        if (fin) {
            const i64 slot = (ep0 + appended + woff + before) % cap;
            ring[2 * slot] = R;
            ring[2 * slot + 1] = (float)L;
        }
        if (t == 0)
            for (int w = 0; w < 16; w++) ret_acc += wret[w], len_acc += wlen[w], stepped += wlive[w];
        appended += total;
        __syncthreads();
    }
    if (t == 0) {
        tot->episodes = ep0 + appended;
        tot->steps += stepped;
        tot->return_sum += ret_acc;
        tot->length_sum += len_acc;
    }
}
}  // namespace

extern "C" {

int srlx_episode_account(int64_t n_envs, const float *d_rewards, const uint8_t *d_done, const uint8_t *d_skip, float *d_ep_return, int32_t *d_ep_len,
                         float *d_ring, int64_t ring_cap, void *d_totals, void *stream) {
    SRLX_REQUIRE(n_envs > 0 && d_rewards && d_done && d_ep_return && d_ep_len && d_ring && ring_cap > 0 && d_totals, "episode_account: bad argument");
    hipLaunchKernelGGL(k_episode_account, dim3(1), dim3(1024), 0, (hipStream_t)stream, n_envs, d_rewards, d_done, d_skip, d_ep_return, d_ep_len, d_ring,
                       ring_cap, (EpTotals *)d_totals);
    SRLX_HIP(hipGetLastError());
    return SRLX_OK;
}

}  // extern "C"
