// srlx_rollout.hip -- device-resident transition store + batched policy step for E lock-stepped envs.
//
// Replaces (vectorised engine): WorkerRun frame stacking / tracking ring
// (srl/base/rl/worker_run.py:310-358,548-610; srl/base/spaces/box.py:303-312), the Rainbow worker's
// n-step item assembly with terminal padding (srl/algorithms/rainbow/rainbow.py:331-400), the
// zlib+pickle item storage (srl/rl/memories/priority_replay_buffer.py:205-217,242-243), the nested-list
// -> ndarray batch assembly (rainbow.py:190-194) and epsilon-greedy selection (rainbow.py:301-329).
//
// HBM layout:  frames [E][L][F] uint8 (or float32) -- an item's n-step window (W+n frames) is ONE
// contiguous span of the ring; scalars action/reward/flags/step_in_episode [E][L].  A stacked fp32
// observation is never stored: it is rebuilt from W uint8 frames on every read (7 056 B read instead of
// 112 896 B per state; the reference stores 451 584 B of float pixels per item).
//
// All frame movers handle 16 bytes per lane (global_load_dwordx4 -> 4 x global_store_dwordx4 for the
// u8 -> f32 expansion); they are HBM-bound: algorithmic bytes per frame = F read + 4F written.
#include <new>

#include "srlx_common.h"
#include "srlx_store_dev.h"
#include "srlx_td_math.h"

namespace {

using namespace srlxs;

// u8/255 in float32, correctly rounded (image_processor.py:140-142 `state.astype(float32); state /= 255`)
__device__ __forceinline__ float norm_u8(unsigned b) { return __fdiv_rn((float)b, 255.0f); }

// one 16-byte chunk of a frame: u8 -> 16 normalised floats (or zeros)
__device__ __forceinline__ void emit_chunk_u8(const u8 *src_frame, bool zero, i64 chunk, float *dst_frame) {
    float4 *d = reinterpret_cast<float4 *>(dst_frame) + chunk * 4;
    if (zero) {
        const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
        d[0] = z;
        d[1] = z;
        d[2] = z;
        d[3] = z;
        return;
    }
    const uint4 v = reinterpret_cast<const uint4 *>(src_frame)[chunk];
    const unsigned w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int k = 0; k < 4; k++)
        d[k] = make_float4(norm_u8(w[k] & 255u), norm_u8((w[k] >> 8) & 255u), norm_u8((w[k] >> 16) & 255u),
                           norm_u8(w[k] >> 24));
}

// stacked observation at ring position x of env e: channel c holds the frame W-1-c steps back, zeros
// before the episode start (worker_run.py:277,316-322)
template <bool VEC>
__device__ __forceinline__ void emit_stack_elem(const StoreDev &s, i64 e, i64 x, int c, i64 unit, float *dst_frame) {
    const int back = s.W - 1 - c;
    const i64 rx = posmod(x, s.L);
    const bool zero = back > s.step_in_ep[e * s.L + rx];
    const i64 rf = posmod(x - back, s.L);
    if (s.obs_dtype == SRLX_OBS_U8) {
        const u8 *src = (const u8 *)s.obs + (e * s.L + rf) * s.F;
        if (VEC) {
            emit_chunk_u8(src, zero, unit, dst_frame);
        } else {
            dst_frame[unit] = zero ? 0.f : norm_u8(src[unit]);
        }
    } else {
        const float *src = (const float *)s.obs + (e * s.L + rf) * s.F;
        if (VEC) {
            reinterpret_cast<float4 *>(dst_frame)[unit] = zero ? make_float4(0.f, 0.f, 0.f, 0.f) : reinterpret_cast<const float4 *>(src)[unit];
        } else {
            dst_frame[unit] = zero ? 0.f : src[unit];
        }
    }
}

// u8 fast path: ONE workgroup expands ONE frame.  A lane loads kFrameUnroll independent dwords (a wave
// reads 256 contiguous bytes per load instruction) before it converts and stores them as float4 (1 KiB
// contiguous per store instruction): with a single dword per lane a CU can only keep 8 KiB of reads in
// flight and the kernel is latency-bound at ~3 TB/s (measured); the frame lookup (episode-start test,
// ring offset) is workgroup-uniform, no per-lane integer division.
constexpr int kFrameUnroll = 8;  // 8 * 256 lanes * 4 B = 8 KiB per workgroup pass (a 84x84 frame is 7 056 B)

__device__ __forceinline__ void emit_frame_dwords(const StoreDev &s, i64 e, i64 x, int c, float *dst_frame) {
    const int back = s.W - 1 - c;
    const i64 rx = posmod(x, s.L);
    const bool zero = back > s.step_in_ep[e * s.L + rx];
    const i64 rf = posmod(x - back, s.L);
    const i64 nd = s.F / 4;
    const unsigned *src = reinterpret_cast<const unsigned *>((const u8 *)s.obs + (e * s.L + rf) * s.F);
    float4 *dst = reinterpret_cast<float4 *>(dst_frame);
    for (i64 i0 = threadIdx.x; i0 < nd; i0 += 256 * kFrameUnroll) {
        unsigned w[kFrameUnroll];
#pragma unroll
        for (int u = 0; u < kFrameUnroll; u++) {
            const i64 i = i0 + u * 256;
            w[u] = (!zero && i < nd) ? src[i] : 0u;
        }
#pragma unroll
        for (int u = 0; u < kFrameUnroll; u++) {
            const i64 i = i0 + u * 256;
            if (i < nd)
                dst[i] = make_float4(norm_u8(w[u] & 255u), norm_u8((w[u] >> 8) & 255u), norm_u8((w[u] >> 16) & 255u), norm_u8(w[u] >> 24));
        }
    }
}

__global__ void __launch_bounds__(256) k_stack_current_u8(StoreDev s, float *out) {
    const i64 fc = blockIdx.x;  // (e, c)
    emit_frame_dwords(s, fc / s.W, s.pos[0], (int)(fc % s.W), out + fc * s.F);
}

// units per frame for the vector / scalar paths
__host__ __device__ __forceinline__ i64 units_per_frame(i64 F, int obs_dtype, bool vec) {
    if (!vec) return F;
    return obs_dtype == SRLX_OBS_U8 ? F / 16 : F / 4;
}

template <bool VEC>
__global__ void __launch_bounds__(256) k_stack_current(StoreDev s, float *out) {
    const i64 upf = units_per_frame(s.F, s.obs_dtype, VEC);
    const i64 total = s.E * s.W * upf;
    const i64 p = s.pos[0];
    for (i64 t = (i64)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (i64)gridDim.x * blockDim.x) {
        const i64 unit = t % upf;
        const i64 fc = t / upf;
        const int c = (int)(fc % s.W);
        const i64 e = fc / s.W;
        emit_stack_elem<VEC>(s, e, p, c, unit, out + fc * s.F);
    }
}

__global__ void k_reset_all(StoreDev s, const void *first_obs) {
    const i64 upf = s.obs_dtype == SRLX_OBS_U8 ? s.F : s.F * 4;  // bytes
    const i64 total = s.E * upf;
    for (i64 t = (i64)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (i64)gridDim.x * blockDim.x) {
        const i64 e = t / upf, b = t % upf;
        ((u8 *)s.obs)[(e * s.L) * upf + b] = ((const u8 *)first_obs)[t];
    }
    for (i64 e = (i64)blockIdx.x * blockDim.x + threadIdx.x; e < s.E; e += (i64)gridDim.x * blockDim.x) {
        s.needs_reset[e] = 0;
        s.step_in_ep[e * s.L] = 0;
        s.flags[e * s.L] = 0;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) s.pos[0] = 0;
}

__global__ void k_advance(i64 *counter) { counter[0] += 1; }

// ---- one lock-step commit as ONE launch (k_commit_frames + k_commit_scalars + k_advance + the next lock-step's k_frame_table_current) ----------
// Every block copies its share of the frames (grid-stride, 16 B per lane); the first ceil(E / 256) blocks also do the per-environment scalars, the
// item mask and -- a pure function of (environment, p + 1, the step_in_ep value this thread has just written) -- the frame-offset table of the NEXT
// policy pass.  advance != 0: the ring position moves inside the launch.  Every block reads p when it starts and takes a ticket when it is done; the
// block that draws the last ticket is the only one left running, moves p and rewinds the ticket counter (pos[4]) for the next launch.
// advance == 0: p stays (an engine that overlaps a learner moves it after joining it: the learner's item lookup reads p; ring slot p + 1 itself is
// referenced by no item, so frames and scalars may land while the learner still runs).  `bump`: one more int64 counter the launch advances (block 0).
// Packed source (srlx_store_commit_step_packed: the slabs actor ranks ship to a learner rank, device/dist.py): environment e = (rank block e / per, lane e % per);
// a block's record = [action int32 x per | reward float32 x per | terminated u8 x per | done u8 x per | extra float32 x per x k].  est / est_out: the first
// extra field of ANOTHER packed buffer of the same shape (the actor-side initial-priority estimates, which travel one slab behind the items they belong to)
// written as one float per environment, -2 where this commit completed no item for the lane (what srlx_per_add(SRLX_PRIO_EST_F32) expects).
struct PackedSrc {
    const u8 *base;  // NULL: separate arrays
    i64 stride;      // bytes between rank blocks
    i64 per;         // environments per rank block
    int k;           // extra float32 fields per environment
    const u8 *est;   // packed buffer carrying the estimates, or NULL
    float *est_out;  // [E]
};
__global__ void __launch_bounds__(256) k_commit_step(StoreDev s, const int32_t *__restrict__ actions, const float *__restrict__ rewards, const u8 *__restrict__ terminated,
                                                     const u8 *__restrict__ done, const void *__restrict__ next_obs, u8 *__restrict__ item_mask,
                                                     i64 *__restrict__ next_table, int advance, i64 *__restrict__ bump, PackedSrc pk, i64 host_pos) {
    const i64 p = host_pos >= 0 ? host_pos : s.pos[0];  // (host_pos: srlx_store_commit_step_at -- the device position then belongs to the learner's view alone)
    const i64 r = posmod(p, s.L), r1 = posmod(p + 1, s.L);
    const i64 fb = s.obs_dtype == SRLX_OBS_U8 ? s.F : s.F * 4;  // frame bytes
    if ((fb & 15) == 0) {
        const i64 cpf = fb / 16, total = s.E * cpf;
        for (i64 t = (i64)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (i64)gridDim.x * blockDim.x) {
            const i64 e = t / cpf, c = t % cpf;
            reinterpret_cast<uint4 *>((u8 *)s.obs + (e * s.L + r1) * fb)[c] = reinterpret_cast<const uint4 *>(next_obs)[t];
        }
    } else {
        const i64 total = s.E * fb;
        for (i64 t = (i64)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (i64)gridDim.x * blockDim.x) {
            const i64 e = t / fb, c = t % fb;
            ((u8 *)s.obs)[(e * s.L + r1) * fb + c] = ((const u8 *)next_obs)[t];
        }
    }
    const i64 e = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    if (e < s.E) {  // the scalars of environment e (k_commit_scalars)
        const i64 base = e * s.L;
        int sie1;
        if (s.needs_reset[e]) {
            // position p holds the previous episode's terminal frame: no transition starts here
            s.flags[base + r] = kInvalid;
            s.action[base + r] = 0;
            s.reward[base + r] = 0.f;
            sie1 = 0;
            s.needs_reset[e] = 0;
        } else {
            float rew;
            int32_t act;
            u8 d, tm;
            if (pk.base) {
                const u8 *rec = pk.base + (e / pk.per) * pk.stride;
                const i64 i = e % pk.per;
                act = reinterpret_cast<const int32_t *>(rec)[i];
                rew = reinterpret_cast<const float *>(rec + 4 * pk.per)[i];
                tm = rec[8 * pk.per + i] ? 1 : 0;
                d = rec[9 * pk.per + i] ? 1 : 0;
            } else {
                act = actions[e];
                rew = rewards[e];
                d = done[e] ? 1 : 0, tm = terminated[e] ? 1 : 0;
            }
            if (s.reward_clip) rew = rew < 0.f ? -1.f : (rew > 0.f ? 1.f : 0.f);  // rainbow.py:337-343
            s.flags[base + r] = (tm ? kTerm : 0) | (d ? kDone : 0);
            s.action[base + r] = act;
            s.reward[base + r] = rew;
            sie1 = s.step_in_ep[base + r] + 1;
            s.needs_reset[e] = d;
        }
        s.step_in_ep[base + r1] = sie1;
        if (item_mask) {
            const i64 q = p - (s.n - 1);
            const u8 has = (q >= 0 && !(s.flags[base + posmod(q, s.L)] & kInvalid)) ? 1 : 0;
            item_mask[e] = has;
            if (pk.est_out)
                pk.est_out[e] = !has ? -2.f : (pk.est ? reinterpret_cast<const float *>(pk.est + (e / pk.per) * pk.stride + 10 * pk.per)[(e % pk.per) * pk.k] : -1.f);
        }
        if (next_table)  // frame_offset(s, e, p + 1, c) with step_in_ep[p + 1] = sie1
            for (int c = 0; c < s.W; c++) {
                const int back = s.W - 1 - c;
                next_table[e * s.W + c] = back > sie1 ? -1 : (base + posmod(p + 1 - back, s.L)) * s.F;
            }
    }
    if (bump && blockIdx.x == 0 && threadIdx.x == 0) *bump += 1;  // nobody in THIS launch reads it (the policy pass that did is an earlier launch): no ticket needed
    if (advance) {  // p is read by every block: the block that draws the last ticket moves it (the launcher keeps the grid small: one atomic per block on one address)
        __shared__ int last;
        __syncthreads();  // this block's reads of p are done
        if (threadIdx.x == 0) last = atomicAdd(reinterpret_cast<unsigned *>(s.pos + 4), 1u) == gridDim.x - 1;
        __syncthreads();
        if (last && threadIdx.x == 0) {
            s.pos[0] = p + 1;
            *reinterpret_cast<unsigned *>(s.pos + 4) = 0u;
        }
    }
}

// ------------------------------------------------------------------------------------------
// gather_nstep
// ------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_gather_meta(StoreDev s, i64 B, const i64 *tree_idx, ItemMeta *meta,
                                                      int32_t *actions, float *rewards, float *terminated) {
    const i64 b = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    meta[b] = item_meta(s, b, tree_idx, actions, rewards, terminated);
}

// (environment, ring slot of the first transition, ring slot of the transition before it or -1 at an episode start) of sampled items:
// what an engine needs to gather per-step fields it keeps in its own [ring slot][env] arrays next to the store's
__global__ void __launch_bounds__(256) k_locate(StoreDev s, i64 B, const i64 *tree_idx, i64 *env, i64 *slot, i64 *prev_slot) {
    const i64 b = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    const i64 N = s.E * s.item_len;
    i64 j = tree_idx[b] - (N - 1);
    j = j < 0 ? 0 : (j >= N ? N - 1 : j);
    const i64 e = j % s.E, tau = j / s.E;
    const i64 p_last = s.pos[0] - 1;
    i64 q = p_last - posmod(p_last - tau, s.item_len) - (s.n - 1);
    if (q < 0) q = 0;
    const i64 r = posmod(q, s.L);
    env[b] = e;
    slot[b] = r;
    if (prev_slot) prev_slot[b] = s.step_in_ep[e * s.L + r] == 0 ? -1 : posmod(q - 1, s.L);
}

__global__ void __launch_bounds__(256) k_gather_obs_u8(StoreDev s, const ItemMeta *meta, float *out) {
    const i64 fc = blockIdx.x;  // (b, k, c)
    const int S = s.n + 1;
    const int c = (int)(fc % s.W);
    const i64 bk = fc / s.W;
    const int k = (int)(bk % S);
    const ItemMeta m = meta[bk / S];
    const int kk = k < m.jd + 1 ? k : m.jd + 1;  // states after the terminal one repeat it (rainbow.py:358)
    emit_frame_dwords(s, m.e, m.q + kk, c, out + fc * s.F);
}

// states k_begin .. k_begin+k_count-1 only (the learner needs float32 pixels for s_0 alone: s_1..s_n go
// through srlx_qnet_forward_u8)
__global__ void __launch_bounds__(256) k_gather_obs_sel_u8(StoreDev s, const ItemMeta *meta, int k_begin, int k_count, float *out) {
    const i64 fc = blockIdx.x;  // (b, k', c)
    const int c = (int)(fc % s.W);
    const i64 bk = fc / s.W;
    const int k = k_begin + (int)(bk % k_count);
    const ItemMeta m = meta[bk / k_count];
    const int kk = k < m.jd + 1 ? k : m.jd + 1;
    emit_frame_dwords(s, m.e, m.q + kk, c, out + fc * s.F);
}

template <bool VEC>
__global__ void __launch_bounds__(256) k_gather_obs(StoreDev s, i64 B, const ItemMeta *meta, float *out) {
    const i64 upf = units_per_frame(s.F, s.obs_dtype, VEC);
    const int S = s.n + 1;
    const i64 total = B * S * s.W * upf;
    for (i64 t = (i64)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (i64)gridDim.x * blockDim.x) {
        const i64 unit = t % upf;
        const i64 fc = t / upf;  // (b, k, c)
        const int c = (int)(fc % s.W);
        const i64 bk = fc / s.W;
        const int k = (int)(bk % S);
        const i64 b = bk / S;
        const ItemMeta m = meta[b];
        const int kk = k < m.jd + 1 ? k : m.jd + 1;  // states after the terminal one repeat it (rainbow.py:358)
        emit_stack_elem<VEC>(s, m.e, m.q + kk, c, unit, out + fc * s.F);
    }
}

// ------------------------------------------------------------------------------------------
// frame-offset tables for srlx_qnet_forward_u8: per sample and stacked channel the BYTE offset of the
// uint8 frame inside the ring (or -1 for the all-zero history before an episode start), so that the
// first convolution can read the ring directly and the float32 stacked observation never exists.
// ------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_frame_table_current(StoreDev s, i64 *out) {
    const i64 t = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= s.E * s.W) return;
    out[t] = frame_offset(s, t / s.W, s.pos[0], (int)(t % s.W));
}
__global__ void __launch_bounds__(256) k_frame_table_items(StoreDev s, i64 B, const ItemMeta *meta, int k_begin, int k_count, i64 *out) {
    const i64 t = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= B * k_count * s.W) return;
    const int c = (int)(t % s.W);
    const i64 bk = t / s.W;
    const int k = k_begin + (int)(bk % k_count);
    const ItemMeta m = meta[bk / k_count];
    const int kk = k < m.jd + 1 ? k : m.jd + 1;
    out[t] = frame_offset_q(s, m.e, m.pad, kk, c);
}

// the learner's whole "gather" in one launch: item location + n-step scalars (k_gather_meta) and both offset tables
// (s_0..s_n for the online network, s_1..s_n for the target network); 64 items per workgroup, their metadata in LDS
__global__ void __launch_bounds__(256) k_gather_train(StoreDev s, i64 B, const i64 *tree_idx, ItemMeta *meta, int32_t *actions, float *rewards,
                                                       float *terminated, i64 *off_all, i64 *off_next) {
    __shared__ ItemMeta sm[kTrainItems];
    const i64 b0 = (i64)blockIdx.x * kTrainItems;
    const int cnt = (int)(B - b0 < kTrainItems ? B - b0 : kTrainItems);
    gather_train_items(s, b0, cnt, tree_idx, meta, actions, rewards, terminated, off_all, off_next, sm);
}

// ------------------------------------------------------------------------------------------
// epsilon-greedy (rainbow.py:301-329)
// ------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_eps_greedy(i64 E, int A, const float *q, const float *eps, const double *u,
                                                     const u8 *invalid, int32_t *actions) {
    const i64 e = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= E) return;
    const float *qe = q + e * A;
    const u8 *inv = invalid ? invalid + e * A : nullptr;
    if (u[2 * e] < (double)eps[e]) {  // random.random() < epsilon (:317)
        int nv = 0;
        for (int a = 0; a < A; a++) nv += !(inv && inv[a]);
        int pick = (int)(u[2 * e + 1] * (double)nv);
        if (pick >= nv) pick = nv - 1;
        int act = 0;
        for (int a = 0; a < A; a++) {
            if (inv && inv[a]) continue;
            if (pick == 0) {
                act = a;
                break;
            }
            pick--;
        }
        actions[e] = act;
    } else {  // q[invalid] = -inf; argmax (first maximum, like np.argmax) (:321-325)
        int best = 0;
        float bv = -INFINITY;
        bool have = false;
        for (int a = 0; a < A; a++) {
            const float v = (inv && inv[a]) ? -INFINITY : qe[a];
            if (!have || v > bv) {
                best = a;
                bv = v;
                have = true;
            }
        }
        actions[e] = best;
    }
}

__global__ void __launch_bounds__(256) k_rng_uniform(u64 seed, const i64 *counter, i64 n, double *out) {
    const u64 c = (u64)counter[0];
    for (i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (i64)gridDim.x * blockDim.x)
        out[i] = u53(rng_u64(seed, c, (u64)i));
}
// same values from a single workgroup that also advances the counter (every thread has read it before the barrier)
__global__ void __launch_bounds__(1024) k_rng_uniform_wg(u64 seed, i64 *counter, i64 n, double *out) {
    const u64 c = (u64)counter[0];
    for (i64 i = threadIdx.x; i < n; i += blockDim.x) out[i] = u53(rng_u64(seed, c, (u64)i));
    __syncthreads();
    if (threadIdx.x == 0) counter[0] = (i64)c + 1;
}

// ------------------------------------------------------------------------------------------
// synthetic environment batch
// ------------------------------------------------------------------------------------------
// `pos_arg` >= 0: the ring position as a launch argument (srlx_synth_env_step_at: a store whose device-resident position trails its commits); < 0: the device's
__global__ void __launch_bounds__(256) k_synth_frames(StoreDev s, void *next_obs, i64 pos_arg) {
    const i64 fb = s.obs_dtype == SRLX_OBS_U8 ? s.F : s.F * 4;
    const i64 p1 = (pos_arg >= 0 ? pos_arg : s.pos[0]) + 1;
    if (s.obs_dtype == SRLX_OBS_U8 && (fb & 15) == 0) {
        const i64 cpf = fb / 16, total = s.E * cpf;
        for (i64 t = (i64)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (i64)gridDim.x * blockDim.x) {
            const i64 e = t / cpf, c = t % cpf;
            const u64 key = (u64)(e * 0x100000000ll + (p1 & 0xffffffffll));
            const u64 a = rng_u64(s.seed, key, (u64)(2 * c)), b = rng_u64(s.seed, key, (u64)(2 * c + 1));
            reinterpret_cast<uint4 *>(next_obs)[t] = make_uint4((unsigned)a, (unsigned)(a >> 32), (unsigned)b, (unsigned)(b >> 32));
        }
    } else if (s.obs_dtype == SRLX_OBS_U8) {
        const i64 total = s.E * fb;
        for (i64 t = (i64)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (i64)gridDim.x * blockDim.x) {
            const i64 e = t / fb, c = t % fb;
            const u64 key = (u64)(e * 0x100000000ll + (p1 & 0xffffffffll));
            ((u8 *)next_obs)[t] = (u8)(rng_u64(s.seed, key, (u64)(c / 8)) >> (8 * (c % 8)));
        }
    } else {
        const i64 total = s.E * s.F;
        for (i64 t = (i64)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (i64)gridDim.x * blockDim.x) {
            const i64 e = t / s.F, c = t % s.F;
            const u64 key = (u64)(e * 0x100000000ll + (p1 & 0xffffffffll));
            ((float *)next_obs)[t] = (float)(2.0 * u53(rng_u64(s.seed, key, (u64)c)) - 1.0);
        }
    }
}
// frames AND scalars of one synthetic lock-step in one launch (the first ceil(E / 256) blocks also do the scalars)
__device__ __forceinline__ void synth_scalars_one(const StoreDev &s, i64 e, i64 episode_len, float *rewards, u8 *terminated, u8 *done, i64 pos_arg);
__global__ void __launch_bounds__(256) k_synth_env(StoreDev s, i64 episode_len, void *next_obs, float *rewards, u8 *terminated, u8 *done, i64 pos_arg) {
    const i64 fb = s.F;  // uint8 frames of 16-byte multiples only (the launcher checks)
    const i64 p1 = (pos_arg >= 0 ? pos_arg : s.pos[0]) + 1;
    const i64 cpf = fb / 16, total = s.E * cpf;
    for (i64 t = (i64)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (i64)gridDim.x * blockDim.x) {
        const i64 e = t / cpf, c = t % cpf;
        const u64 key = (u64)(e * 0x100000000ll + (p1 & 0xffffffffll));
        const u64 a = rng_u64(s.seed, key, (u64)(2 * c)), b = rng_u64(s.seed, key, (u64)(2 * c + 1));
        reinterpret_cast<uint4 *>(next_obs)[t] = make_uint4((unsigned)a, (unsigned)(a >> 32), (unsigned)b, (unsigned)(b >> 32));
    }
    const i64 e = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    if (e < s.E) synth_scalars_one(s, e, episode_len, rewards, terminated, done, pos_arg);
}
__device__ __forceinline__ void synth_scalars_one(const StoreDev &s, i64 e, i64 episode_len, float *rewards, u8 *terminated, u8 *done, i64 pos_arg) {
    const i64 p = pos_arg >= 0 ? pos_arg : s.pos[0];
    if (s.needs_reset[e]) {
        rewards[e] = 0.f;
        terminated[e] = 0;
        done[e] = 0;
        return;
    }
    const u64 key = (u64)(e * 0x100000000ll + (p & 0xffffffffll));
    rewards[e] = (float)((int)(rng_u64(s.seed ^ 0x726577ull, key, 0) % 3ull) - 1);
    const int sie = s.step_in_ep[e * s.L + posmod(p, s.L)];
    const u8 d = (sie + 1 >= episode_len) ? 1 : 0;
    terminated[e] = d;
    done[e] = d;
}
__global__ void __launch_bounds__(256) k_synth_scalars(StoreDev s, i64 episode_len, float *rewards, u8 *terminated, u8 *done, i64 pos_arg) {
    const i64 e = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    if (e < s.E) synth_scalars_one(s, e, episode_len, rewards, terminated, done, pos_arg);
}

}  // namespace

struct srlx_store {
    StoreDev d;
    int device;
    srlx::Arena scratch;
    bool vec;
};

namespace {
hipStream_t pick(srlx_store *, void *st) { return (hipStream_t)st; }  // NULL = HIP's default stream
int grid_for(i64 total, int cap = 256 * 16) {
    i64 b = (total + 255) / 256;
    if (b < 1) b = 1;
    return (int)(b < cap ? b : cap);
}
}  // namespace

extern "C" {

int srlx_store_create(srlx_store_t **out, int64_t n_envs, int64_t ring_len, int64_t obs_elems, int obs_dtype, int window,
                      int n_step, int n_actions, int reward_clip, uint64_t seed, int device) {
    SRLX_REQUIRE(out, "store_create: out is NULL");
    SRLX_REQUIRE(n_envs > 0 && obs_elems > 0 && window >= 1 && n_step >= 1 && n_actions >= 1, "store_create: bad sizes");
    SRLX_REQUIRE(obs_dtype == SRLX_OBS_U8 || obs_dtype == SRLX_OBS_F32, "store_create: bad obs_dtype %d", obs_dtype);
    SRLX_REQUIRE(ring_len > (i64)n_step + window, "store_create: ring_len must exceed n_step + window");
    int ndev = 0;
    SRLX_HIP(hipGetDeviceCount(&ndev));
    SRLX_REQUIRE(device >= 0 && device < ndev, "store_create: device %d not present", device);
    srlx::DeviceGuard guard(device);
    srlx_store *h = new (std::nothrow) srlx_store();
    if (!h) return SRLX_ERR_NOMEM;
    memset(&h->d, 0, sizeof(h->d));
    h->device = device;
    StoreDev &d = h->d;
    d.E = n_envs;
    d.L = ring_len;
    d.F = obs_elems;
    d.obs_dtype = obs_dtype;
    d.W = window;
    d.n = n_step;
    d.A = n_actions;
    d.reward_clip = reward_clip;
    d.seed = seed;
    d.item_len = ring_len - (n_step + window);
    const size_t eb = obs_dtype == SRLX_OBS_U8 ? 1 : 4;
    const size_t cells = (size_t)n_envs * (size_t)ring_len;
    h->vec = obs_dtype == SRLX_OBS_U8 ? (obs_elems % 16 == 0) : (obs_elems % 4 == 0);
    hipError_t e = hipMalloc(&d.obs, cells * (size_t)obs_elems * eb);
    if (e == hipSuccess) e = hipMalloc((void **)&d.action, cells * 4);
    if (e == hipSuccess) e = hipMalloc((void **)&d.reward, cells * 4);
    if (e == hipSuccess) e = hipMalloc((void **)&d.flags, cells);
    if (e == hipSuccess) e = hipMalloc((void **)&d.step_in_ep, cells * 4);
    if (e == hipSuccess) e = hipMalloc((void **)&d.pos, 64);
    if (e == hipSuccess) e = hipMalloc((void **)&d.needs_reset, (size_t)n_envs);
    if (e == hipSuccess) e = hipMemset(d.flags, 0, cells);
    if (e == hipSuccess) e = hipMemset(d.step_in_ep, 0, cells * 4);
    if (e == hipSuccess) e = hipMemset(d.action, 0, cells * 4);
    if (e == hipSuccess) e = hipMemset(d.reward, 0, cells * 4);
    if (e == hipSuccess) e = hipMemset(d.pos, 0, 64);
    if (e == hipSuccess) e = hipMemset(d.needs_reset, 0, (size_t)n_envs);
    if (e != hipSuccess) {
        srlx::set_error("store_create: %s", hipGetErrorString(e));
        srlx_store_destroy(h);
        return e == hipErrorOutOfMemory ? SRLX_ERR_NOMEM : SRLX_ERR_HIP;
    }
    *out = h;
    return SRLX_OK;
}

int srlx_store_destroy(srlx_store_t *h) {
    if (!h) return SRLX_OK;
    srlx::DeviceGuard guard(h->device);
    (void)hipDeviceSynchronize();
    StoreDev &d = h->d;
    if (d.obs) (void)hipFree(d.obs);
    if (d.action) (void)hipFree(d.action);
    if (d.reward) (void)hipFree(d.reward);
    if (d.flags) (void)hipFree(d.flags);
    if (d.step_in_ep) (void)hipFree(d.step_in_ep);
    if (d.pos) (void)hipFree(d.pos);
    if (d.needs_reset) (void)hipFree(d.needs_reset);
    h->scratch.release();
    delete h;
    return SRLX_OK;
}

int64_t srlx_store_item_len(const srlx_store_t *h) { return h ? h->d.item_len : -1; }
int64_t srlx_store_per_capacity(const srlx_store_t *h) { return h ? h->d.E * h->d.item_len : -1; }

int srlx_store_reset_all(srlx_store_t *h, const void *d_first_obs, void *stream) {
    SRLX_REQUIRE(h && d_first_obs, "store_reset_all: NULL argument");
    srlx::DeviceGuard guard(h->device);
    hipStream_t st = pick(h, stream);
    const i64 bytes = h->d.E * h->d.F * (h->d.obs_dtype == SRLX_OBS_U8 ? 1 : 4);
    hipLaunchKernelGGL(k_reset_all, dim3(grid_for(bytes)), dim3(256), 0, st, h->d, d_first_obs);
    SRLX_HIP(hipGetLastError());
    return SRLX_OK;
}

int srlx_store_stack_current(srlx_store_t *h, float *d_out, void *stream) {
    SRLX_REQUIRE(h && d_out, "store_stack_current: NULL argument");
    srlx::DeviceGuard guard(h->device);
    hipStream_t st = pick(h, stream);
    const StoreDev &d = h->d;
    const i64 total = d.E * d.W * units_per_frame(d.F, d.obs_dtype, h->vec);
    if (d.obs_dtype == SRLX_OBS_U8 && d.F % 16 == 0 && d.E * d.W < ((i64)1 << 31))
        hipLaunchKernelGGL(k_stack_current_u8, dim3((unsigned)(d.E * d.W)), dim3(256), 0, st, d, d_out);
    else if (h->vec)
        hipLaunchKernelGGL(k_stack_current<true>, dim3(grid_for(total, 256 * 32)), dim3(256), 0, st, d, d_out);
    else
        hipLaunchKernelGGL(k_stack_current<false>, dim3(grid_for(total, 256 * 32)), dim3(256), 0, st, d, d_out);
    SRLX_HIP(hipGetLastError());
    return SRLX_OK;
}

int srlx_store_commit_step_ex(srlx_store_t *h, const int32_t *d_actions, const float *d_rewards, const uint8_t *d_terminated, const uint8_t *d_done,
                              const void *d_next_obs, uint8_t *d_item_mask, int64_t *d_next_frame_table, int advance, int64_t *d_bump, void *stream) {
    SRLX_REQUIRE(h && d_actions && d_rewards && d_terminated && d_done && d_next_obs, "store_commit_step: NULL argument");
    SRLX_REQUIRE(!d_next_frame_table || h->d.obs_dtype == SRLX_OBS_U8, "store_commit_step: frame tables exist for uint8 stores only");
    srlx::DeviceGuard guard(h->device);
    hipStream_t st = pick(h, stream);
    const StoreDev &d = h->d;
    const i64 fb = d.F * (d.obs_dtype == SRLX_OBS_U8 ? 1 : 4);
    int grid = grid_for(d.E * (fb / 16 + 1), advance ? 256 : 2048);  // (advance: 256 tickets on one address cost ~3 us, 2048 cost 25)
    const int need = (int)((d.E + 255) / 256);  // the scalar part needs one thread per environment
    if (grid < need) grid = need;
    hipLaunchKernelGGL(k_commit_step, dim3((unsigned)grid), dim3(256), 0, st, d, d_actions, d_rewards, d_terminated, d_done, d_next_obs, d_item_mask,
                       (i64 *)d_next_frame_table, advance, (i64 *)d_bump, PackedSrc{}, (i64)-1);
    SRLX_HIP(hipGetLastError());
    return SRLX_OK;
}

int srlx_store_commit_step_at(srlx_store_t *h, int64_t position, const int32_t *d_actions, const float *d_rewards, const uint8_t *d_terminated, const uint8_t *d_done,
                              const void *d_next_obs, uint8_t *d_item_mask, int64_t *d_next_frame_table, int64_t *d_bump, void *stream) {
    SRLX_REQUIRE(h && d_actions && d_rewards && d_terminated && d_done && d_next_obs && position >= 0, "store_commit_step_at: bad argument");
    SRLX_REQUIRE(!d_next_frame_table || h->d.obs_dtype == SRLX_OBS_U8, "store_commit_step_at: frame tables exist for uint8 stores only");
    srlx::DeviceGuard guard(h->device);
    hipStream_t st = pick(h, stream);
    const StoreDev &d = h->d;
    const i64 fb = d.F * (d.obs_dtype == SRLX_OBS_U8 ? 1 : 4);
    int grid = grid_for(d.E * (fb / 16 + 1), 2048);
    const int need = (int)((d.E + 255) / 256);
    if (grid < need) grid = need;
    hipLaunchKernelGGL(k_commit_step, dim3((unsigned)grid), dim3(256), 0, st, d, d_actions, d_rewards, d_terminated, d_done, d_next_obs, d_item_mask,
                       (i64 *)d_next_frame_table, 0, (i64 *)d_bump, PackedSrc{}, (i64)position);
    SRLX_HIP(hipGetLastError());
    return SRLX_OK;
}

int srlx_store_set_item_slack(srlx_store_t *h, int slack) {
    SRLX_REQUIRE(h && slack >= 0 && h->d.L - (h->d.n + h->d.W) - slack >= 1, "store_set_item_slack: bad slack");
    h->d.item_len = h->d.L - (h->d.n + h->d.W) - slack;
    return SRLX_OK;
}

int srlx_store_commit_step_packed(srlx_store_t *h, const uint8_t *d_records, int64_t record_stride, int64_t envs_per_record, int extra_floats, const void *d_next_obs,
                                  uint8_t *d_item_mask, const uint8_t *d_est_records, float *d_est_out, int advance, void *stream) {
    SRLX_REQUIRE(h && d_records && d_next_obs && d_item_mask, "store_commit_step_packed: NULL argument");
    SRLX_REQUIRE(envs_per_record > 0 && h->d.E % envs_per_record == 0 && extra_floats >= 0 && record_stride >= (10 + 4 * (int64_t)extra_floats) * envs_per_record &&
                     record_stride % 4 == 0 && envs_per_record % 4 == 0,
                 "store_commit_step_packed: %lld environments do not split into records of %lld (stride %lld bytes, %d extra fields; multiples of 4)", (long long)h->d.E,
                 (long long)envs_per_record, (long long)record_stride, extra_floats);
    SRLX_REQUIRE(!d_est_records || (extra_floats >= 1 && d_est_out), "store_commit_step_packed: estimates ride in the first extra field and need d_est_out");
    srlx::DeviceGuard guard(h->device);
    hipStream_t st = pick(h, stream);
    const StoreDev &d = h->d;
    const i64 fb = d.F * (d.obs_dtype == SRLX_OBS_U8 ? 1 : 4);
    int grid = grid_for(d.E * (fb / 16 + 1), advance ? 256 : 2048);
    const int need = (int)((d.E + 255) / 256);
    if (grid < need) grid = need;
    hipLaunchKernelGGL(k_commit_step, dim3((unsigned)grid), dim3(256), 0, st, d, nullptr, nullptr, nullptr, nullptr, d_next_obs, d_item_mask, nullptr, advance, nullptr,
                       PackedSrc{d_records, (i64)record_stride, (i64)envs_per_record, extra_floats, d_est_records, d_est_out}, (i64)-1);
    SRLX_HIP(hipGetLastError());
    return SRLX_OK;
}

int srlx_store_commit_step(srlx_store_t *h, const int32_t *d_actions, const float *d_rewards, const uint8_t *d_terminated,
                           const uint8_t *d_done, const void *d_next_obs, uint8_t *d_item_mask, void *stream) {
    return srlx_store_commit_step_ex(h, d_actions, d_rewards, d_terminated, d_done, d_next_obs, d_item_mask, nullptr, 1, nullptr, stream);
}

int srlx_store_advance(srlx_store_t *h, void *stream) {
    SRLX_REQUIRE(h, "store_advance: NULL handle");
    srlx::DeviceGuard guard(h->device);
    hipLaunchKernelGGL(k_advance, dim3(1), dim3(1), 0, pick(h, stream), h->d.pos);
    SRLX_HIP(hipGetLastError());
    return SRLX_OK;
}

int srlx_store_views(srlx_store_t *h, void **d_pos, void **d_needs_reset, void **d_step_in_ep) {
    SRLX_REQUIRE(h, "store_views: NULL handle");
    if (d_pos) *d_pos = h->d.pos;
    if (d_needs_reset) *d_needs_reset = h->d.needs_reset;
    if (d_step_in_ep) *d_step_in_ep = h->d.step_in_ep;
    return SRLX_OK;
}

int srlx_store_gather_nstep(srlx_store_t *h, int64_t batch, const int64_t *d_tree_idx, float *d_obs, int32_t *d_actions,
                            float *d_rewards, float *d_terminated, void *stream) {
    SRLX_REQUIRE(h && d_tree_idx && d_obs && d_actions && d_rewards && d_terminated, "store_gather_nstep: NULL argument");
    SRLX_REQUIRE(batch > 0, "store_gather_nstep: batch must be positive");
    srlx::DeviceGuard guard(h->device);
    hipStream_t st = pick(h, stream);
    const StoreDev &d = h->d;
    SRLX_TRY(h->scratch.reserve((size_t)batch * sizeof(ItemMeta)));
    ItemMeta *meta = (ItemMeta *)h->scratch.ptr;
    hipLaunchKernelGGL(k_gather_meta, dim3((unsigned)((batch + 255) / 256)), dim3(256), 0, st, d, (i64)batch, d_tree_idx, meta,
                       d_actions, d_rewards, d_terminated);
    const i64 total = batch * (d.n + 1) * d.W * units_per_frame(d.F, d.obs_dtype, h->vec);
    const i64 frames = batch * (d.n + 1) * d.W;
    if (d.obs_dtype == SRLX_OBS_U8 && d.F % 16 == 0 && frames < ((i64)1 << 31))
        hipLaunchKernelGGL(k_gather_obs_u8, dim3((unsigned)frames), dim3(256), 0, st, d, meta, d_obs);
    else if (h->vec)
        hipLaunchKernelGGL(k_gather_obs<true>, dim3(grid_for(total, 256 * 32)), dim3(256), 0, st, d, (i64)batch, meta, d_obs);
    else
        hipLaunchKernelGGL(k_gather_obs<false>, dim3(grid_for(total, 256 * 32)), dim3(256), 0, st, d, (i64)batch, meta, d_obs);
    SRLX_HIP(hipGetLastError());
    return SRLX_OK;
}

int srlx_store_locate(srlx_store_t *h, int64_t batch, const int64_t *d_tree_idx, int64_t *d_env, int64_t *d_slot, int64_t *d_prev_slot, void *stream) {
    SRLX_REQUIRE(h && d_tree_idx && d_env && d_slot && batch > 0, "store_locate: bad argument");
    srlx::DeviceGuard guard(h->device);
    hipLaunchKernelGGL(k_locate, dim3((unsigned)((batch + 255) / 256)), dim3(256), 0, (hipStream_t)stream, h->d, batch, d_tree_idx, d_env, d_slot, d_prev_slot);
    SRLX_HIP(hipGetLastError());
    return SRLX_OK;
}

int srlx_policy_epsilon_greedy(int64_t n_envs, int n_actions, const float *d_q, const float *d_eps, const double *d_u,
                               const uint8_t *d_invalid, int32_t *d_actions, void *stream) {
    SRLX_REQUIRE(n_envs > 0 && n_actions > 0 && d_q && d_eps && d_u && d_actions, "policy_epsilon_greedy: bad argument");
    hipLaunchKernelGGL(k_eps_greedy, dim3((unsigned)((n_envs + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (i64)n_envs,
                       n_actions, d_q, d_eps, d_u, d_invalid, d_actions);
    SRLX_HIP(hipGetLastError());
    return SRLX_OK;
}

int srlx_rng_uniform(uint64_t seed, int64_t *d_counter, int64_t n, double *d_out, void *stream) {
    SRLX_REQUIRE(d_counter && d_out && n > 0, "rng_uniform: bad argument");
    if (n <= 8192) {  // the per-step calls (B + slack, 2 E uniforms): one workgroup draws and advances the counter itself
        hipLaunchKernelGGL(k_rng_uniform_wg, dim3(1), dim3(1024), 0, (hipStream_t)stream, (u64)seed, d_counter, (i64)n, d_out);
        SRLX_HIP(hipGetLastError());
        return SRLX_OK;
    }
    hipLaunchKernelGGL(k_rng_uniform, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, (u64)seed, d_counter, (i64)n, d_out);
    hipLaunchKernelGGL(k_advance, dim3(1), dim3(1), 0, (hipStream_t)stream, d_counter);
    SRLX_HIP(hipGetLastError());
    return SRLX_OK;
}

// Actor-side initial priorities (rainbow.py:389-398): one thread per lane -- the item the last commit completed for it (item_meta: scalars, terminal padding, where the
// episode ends inside the window), then the n-step / retrace TD of srlx_td_math.h:td_rows on the lane's CACHED Q rows (a ring over the acting passes; the online rows in
// both roles).  Replaces ~20 small torch launches on the serial tail of a lock-step.
__global__ void __launch_bounds__(256) k_actor_td(StoreDev s, i64 first_slot, i64 cap, const float *__restrict__ q_hist, int base_slot, const u8 *__restrict__ mask,
                                                  double discount, double retrace_h, int double_dqn, int rescale, float *__restrict__ est) {
    const i64 e = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= s.E) return;
    if (!mask[e]) {
        est[e] = -2.f;
        return;
    }
    const int n = s.n, A = s.A, n1 = n + 1;
    i64 tidx = (first_slot + e) % cap + (cap - 1);
    int32_t act[srlx::kTdMaxStep];
    float rew[srlx::kTdMaxStep], ter[srlx::kTdMaxStep];
    const ItemMeta m = item_meta(s, 0, &tidx, act, rew, ter);
    if (m.jd < n || m.e != e) {  // an episode ends inside the window (or the slot is not this lane's: a caller error) -> max_priority
        est[e] = -1.f;
        return;
    }
    auto row = [&](int k) { return q_hist + (((i64)((base_slot + k) % n1)) * s.E + e) * A; };
    const float disc_f = (float)discount;
    float td[srlx::kTdMaxStep];
    int nact[srlx::kTdMaxStep];
    for (int k = 0; k < n; k++) {  // srlx_td_math.h:td_rows with q_on_next = q_tg_next = rows 1..n, q_on_0 = row 0
        const float *nx = row(k + 1);
        nact[k] = srlx::argmax_masked(nx, nullptr, A);
        float maxq = nx[nact[k]];
        if (rescale) maxq = srlx::inverse_rescaling(maxq);
        float gain = rew[k] + ((1.0f - ter[k]) * disc_f) * maxq;
        if (rescale) gain = srlx::rescaling(gain);
        const float qsel = (k == 0) ? 0.f : row(k)[act[k]];
        td[k] = gain - qsel;
    }
    (void)double_dqn;  // one set of rows: the greedy action and its value come from the same network
    double c = 1.0;
    float target = 0.f;
    for (int k = 0; k < n; k++) {
        if (k > 0) c *= retrace_h * ((act[k] == nact[k]) ? 1.0 : 0.0);  // td_rows: pi = (act[m] == nact[m])
        const float dm = (float)pow(discount, (double)k);
        target = target + (float)((double)(td[k] * dm) * c);
    }
    est[e] = fabsf(target - row(0)[act[0]]);
}

int srlx_store_actor_td(srlx_store_t *h, int64_t first_slot, int64_t per_capacity, const float *d_q_hist, int base_slot, const uint8_t *d_item_mask, double discount,
                        double retrace_h, int enable_double_dqn, int enable_rescale, float *d_est, void *stream) {
    SRLX_REQUIRE(h && d_q_hist && d_item_mask && d_est && per_capacity > 0 && first_slot >= 0 && base_slot >= 0, "store_actor_td: bad argument");
    SRLX_REQUIRE(h->d.n <= srlx::kTdMaxStep, "store_actor_td: n_step <= %d", srlx::kTdMaxStep);
    srlx::DeviceGuard guard(h->device);
    hipLaunchKernelGGL(k_actor_td, dim3((unsigned)((h->d.E + 255) / 256)), dim3(256), 0, pick(h, stream), h->d, (i64)first_slot, (i64)per_capacity, d_q_hist, base_slot,
                       d_item_mask, discount, retrace_h, enable_double_dqn, enable_rescale, d_est);
    SRLX_HIP(hipGetLastError());
    return SRLX_OK;
}

// The frames a frame-offset table points at, packed row by row into one message buffer, and the table re-based onto that buffer (-1 = "no frame" stays):
// what a replay rank ships to a learner rank instead of its ring (device/replay_role.py; the network kernels take a base pointer + offset table, so the
// learner evaluates the packed frames exactly as it would the ring).
__global__ void __launch_bounds__(256) k_pack_frames(const u8 *__restrict__ base, const i64 *__restrict__ off, i64 frame_bytes, u8 *__restrict__ out, i64 *__restrict__ rel) {
    const i64 row = blockIdx.x;
    const i64 o = off[row];
    if (threadIdx.x == 0) rel[row] = o < 0 ? -1 : row * frame_bytes;
    if (o < 0) return;
    const uint4 *src = reinterpret_cast<const uint4 *>(base + o);
    uint4 *dst = reinterpret_cast<uint4 *>(out + row * frame_bytes);
    for (i64 k = threadIdx.x; k < frame_bytes / 16; k += blockDim.x) dst[k] = src[k];
}

int srlx_pack_frames(const uint8_t *d_frame_base, const int64_t *d_frame_off, int64_t rows, int64_t frame_bytes, uint8_t *d_out, int64_t *d_rel_off, void *stream) {
    SRLX_REQUIRE(d_frame_base && d_frame_off && d_out && d_rel_off && rows > 0 && frame_bytes > 0 && frame_bytes % 16 == 0, "pack_frames: bad argument (16-byte frames)");
    hipLaunchKernelGGL(k_pack_frames, dim3((unsigned)rows), dim3(256), 0, (hipStream_t)stream, d_frame_base, (const i64 *)d_frame_off, (i64)frame_bytes, d_out, (i64 *)d_rel_off);
    SRLX_HIP(hipGetLastError());
    return SRLX_OK;
}

// A keyed pseudo-random PERMUTATION of 0..n-1 without a sort: a four-round Feistel network over the smallest even number of bits that holds n - 1,
// walked until it lands below n (a bijection restricted to a subset by cycle walking stays a bijection; fewer than four rounds of walking on average).
// Key = (seed, *counter): a captured graph draws a fresh permutation at every replay with nothing but device state -- the PPO engine's minibatch
// shuffles (device/ppo.py; torch.randperm as a graph node, or eager between replays of its large graphs, did not replay reliably).
__global__ void __launch_bounds__(256) k_rng_permutation(u64 seed, const i64 *counter, i64 n, int half_bits, i64 *out) {
    const u64 c = (u64)counter[0] + blockIdx.y;  // (blockIdx.y: the y-th of several permutations drawn in one launch, under the counter values successive calls would see)
    out += (i64)blockIdx.y * n;
    const u64 mask = (1ull << half_bits) - 1;
    for (i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (i64)gridDim.x * blockDim.x) {
        u64 x = (u64)i;
        do {
            u64 l = x >> half_bits, r = x & mask;
#pragma unroll
            for (int round = 0; round < 4; round++) {
                const u64 f = rng_u64(seed + 0x9E3779B97F4A7C15ull * (u64)(round + 1), c, r) & mask;
                const u64 nl = r;
                r = l ^ f;
                l = nl;
            }
            x = (l << half_bits) | r;
        } while (x >= (u64)n);
        out[i] = (i64)x;
    }
}

int srlx_rng_permutation(uint64_t seed, int64_t *d_counter, int64_t n, int64_t *d_out, void *stream) {
    SRLX_REQUIRE(d_counter && d_out && n > 0 && n < ((int64_t)1 << 40), "rng_permutation: bad argument");
    int bits = 2;
    while (((int64_t)1 << bits) < n) bits += 2;
    hipLaunchKernelGGL(k_rng_permutation, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, (u64)seed, d_counter, (i64)n, bits / 2, (i64 *)d_out);
    hipLaunchKernelGGL(k_advance, dim3(1), dim3(1), 0, (hipStream_t)stream, d_counter);
    SRLX_HIP(hipGetLastError());
    return SRLX_OK;
}

__global__ void k_advance_by(i64 *c, i64 k) { c[0] += k; }

int srlx_rng_permutations(uint64_t seed, int64_t *d_counter, int64_t n, int count, int64_t *d_out, void *stream) {
    SRLX_REQUIRE(d_counter && d_out && n > 0 && n < ((int64_t)1 << 40) && count >= 1 && count <= 65535, "rng_permutations: bad argument");
    int bits = 2;
    while (((int64_t)1 << bits) < n) bits += 2;
    hipLaunchKernelGGL(k_rng_permutation, dim3(grid_for(n), (unsigned)count), dim3(256), 0, (hipStream_t)stream, (u64)seed, d_counter, (i64)n, bits / 2, (i64 *)d_out);
    hipLaunchKernelGGL(k_advance_by, dim3(1), dim3(1), 0, (hipStream_t)stream, d_counter, (i64)count);
    SRLX_HIP(hipGetLastError());
    return SRLX_OK;
}

int srlx_synth_env_step(srlx_store_t *h, int64_t episode_len, void *d_next_obs, float *d_rewards, uint8_t *d_terminated,
                        uint8_t *d_done, void *stream) {
    return srlx_synth_env_step_at(h, -1, episode_len, d_next_obs, d_rewards, d_terminated, d_done, stream);
}

int srlx_synth_env_step_at(srlx_store_t *h, int64_t position, int64_t episode_len, void *d_next_obs, float *d_rewards, uint8_t *d_terminated, uint8_t *d_done,
                           void *stream) {
    SRLX_REQUIRE(h && d_next_obs && d_rewards && d_terminated && d_done && episode_len > 0, "synth_env_step: bad argument");
    const i64 pos_arg = position;
    srlx::DeviceGuard guard(h->device);
    hipStream_t st = pick(h, stream);
    const StoreDev &d = h->d;
    const i64 fb = d.F * (d.obs_dtype == SRLX_OBS_U8 ? 1 : 4);
    if (d.obs_dtype == SRLX_OBS_U8 && (fb & 15) == 0) {  // frames and scalars in one launch
        int grid = grid_for(d.E * (fb / 16 + 1), 2048);
        const int need = (int)((d.E + 255) / 256);
        if (grid < need) grid = need;
        hipLaunchKernelGGL(k_synth_env, dim3((unsigned)grid), dim3(256), 0, st, d, (i64)episode_len, d_next_obs, d_rewards, d_terminated, d_done, pos_arg);
    } else {
        hipLaunchKernelGGL(k_synth_frames, dim3(grid_for(d.E * (fb / 16 + 1))), dim3(256), 0, st, d, d_next_obs, pos_arg);
        hipLaunchKernelGGL(k_synth_scalars, dim3((unsigned)((d.E + 255) / 256)), dim3(256), 0, st, d, (i64)episode_len, d_rewards,
                           d_terminated, d_done, pos_arg);
    }
    SRLX_HIP(hipGetLastError());
    return SRLX_OK;
}


int srlx_store_obs_base(srlx_store_t *h, void **d_base, int64_t *frame_bytes) {
    SRLX_REQUIRE(h && d_base, "store_obs_base: NULL argument");
    *d_base = h->d.obs;
    if (frame_bytes) *frame_bytes = h->d.F * (h->d.obs_dtype == SRLX_OBS_U8 ? 1 : 4);
    return SRLX_OK;
}

int srlx_store_frame_table_current(srlx_store_t *h, int64_t *d_out, void *stream) {
    SRLX_REQUIRE(h && d_out, "store_frame_table_current: NULL argument");
    SRLX_REQUIRE(h->d.obs_dtype == SRLX_OBS_U8, "store_frame_table_current: uint8 stores only");
    srlx::DeviceGuard guard(h->device);
    const i64 n = h->d.E * h->d.W;
    hipLaunchKernelGGL(k_frame_table_current, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, pick(h, stream), h->d, (i64 *)d_out);
    SRLX_HIP(hipGetLastError());
    return SRLX_OK;
}

int srlx_store_gather_items(srlx_store_t *h, int64_t batch, const int64_t *d_tree_idx, int k_begin, int k_count, int64_t *d_frame_off,
                            int32_t *d_actions, float *d_rewards, float *d_terminated, void *stream) {
    SRLX_REQUIRE(h && d_tree_idx && d_frame_off && d_actions && d_rewards && d_terminated, "store_gather_items: NULL argument");
    SRLX_REQUIRE(batch > 0 && k_begin >= 0 && k_count > 0 && k_begin + k_count <= h->d.n + 1, "store_gather_items: bad range");
    SRLX_REQUIRE(h->d.obs_dtype == SRLX_OBS_U8, "store_gather_items: uint8 stores only");
    srlx::DeviceGuard guard(h->device);
    hipStream_t st = pick(h, stream);
    SRLX_TRY(h->scratch.reserve((size_t)batch * sizeof(ItemMeta)));
    ItemMeta *meta = (ItemMeta *)h->scratch.ptr;
    hipLaunchKernelGGL(k_gather_meta, dim3((unsigned)((batch + 255) / 256)), dim3(256), 0, st, h->d, (i64)batch, d_tree_idx, meta, d_actions,
                       d_rewards, d_terminated);
    const i64 n = batch * k_count * h->d.W;
    hipLaunchKernelGGL(k_frame_table_items, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, h->d, (i64)batch, meta, k_begin, k_count,
                       (i64 *)d_frame_off);
    SRLX_HIP(hipGetLastError());
    return SRLX_OK;
}

int srlx_store_gather_train(srlx_store_t *h, int64_t batch, const int64_t *d_tree_idx, int64_t *d_frame_off_all, int64_t *d_frame_off_next,
                            int32_t *d_actions, float *d_rewards, float *d_terminated, void *stream) {
    SRLX_REQUIRE(h && d_tree_idx && d_frame_off_all && d_actions && d_rewards && d_terminated, "store_gather_train: NULL argument");
    SRLX_REQUIRE(batch > 0, "store_gather_train: empty batch");
    SRLX_REQUIRE(h->d.obs_dtype == SRLX_OBS_U8, "store_gather_train: uint8 stores only");
    srlx::DeviceGuard guard(h->device);
    hipStream_t st = pick(h, stream);
    SRLX_TRY(h->scratch.reserve((size_t)batch * sizeof(ItemMeta)));
    ItemMeta *meta = (ItemMeta *)h->scratch.ptr;
    hipLaunchKernelGGL(k_gather_train, dim3((unsigned)((batch + kTrainItems - 1) / kTrainItems)), dim3(256), 0, st, h->d, (i64)batch, d_tree_idx, meta,
                       d_actions, d_rewards, d_terminated, (i64 *)d_frame_off_all, (i64 *)d_frame_off_next);
    SRLX_HIP(hipGetLastError());
    return SRLX_OK;
}

int srlx_store_gather_obs(srlx_store_t *h, int64_t batch, int k_begin, int k_count, float *d_obs, void *stream) {
    // float32 stacked observations of states k_begin..k_begin+k_count-1 of the items located by the LAST
    // srlx_store_gather_items / srlx_store_gather_nstep call on this handle (same stream)
    SRLX_REQUIRE(h && d_obs && batch > 0 && k_begin >= 0 && k_count > 0 && k_begin + k_count <= h->d.n + 1, "store_gather_obs: bad argument");
    SRLX_REQUIRE(h->d.obs_dtype == SRLX_OBS_U8 && h->d.F % 16 == 0, "store_gather_obs: uint8 stores with 16-byte frames only");
    SRLX_REQUIRE(h->scratch.bytes >= (size_t)batch * sizeof(ItemMeta), "store_gather_obs: no gather_items call preceded");
    srlx::DeviceGuard guard(h->device);
    hipLaunchKernelGGL(k_gather_obs_sel_u8, dim3((unsigned)(batch * k_count * h->d.W)), dim3(256), 0, pick(h, stream), h->d,
                       (const ItemMeta *)h->scratch.ptr, k_begin, k_count, d_obs);
    SRLX_HIP(hipGetLastError());
    return SRLX_OK;
}

}  // extern "C"

// (C++ linkage: internal to the library, srlx_store_dev.h)
int srlx_store_dev_view(srlx_store_t *h, int64_t batch, srlxs::StoreDev *out, srlxs::ItemMeta **meta) {
    SRLX_REQUIRE(h && out && batch > 0, "store_dev_view: bad argument");
    *out = h->d;
    if (meta) {
        SRLX_TRY(h->scratch.reserve((size_t)batch * sizeof(ItemMeta)));
        *meta = (ItemMeta *)h->scratch.ptr;
    }
    return SRLX_OK;
}

