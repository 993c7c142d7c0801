// srlx_per.hip -- GPU-resident proportional prioritized replay (sum-tree) for gfx950.
//
// Replaces srl/rl/memories/priority_memories/proportional_memory.py:13-205 (and its pybind11
// twin cpp_module/src/proportional_memory.cpp) behind the C ABI of include/srlx.h.
//
// Data layout in HBM (one handle):
//   T     : the reference's implicit heap (node i has children 2i+1 / 2i+2, leaf slot j is node
//           j+N-1) stored BLOCKED: every 128-byte line holds the 2+4+8 descendants (relative
//           depths 1..3) of one "owner" node, groups of three levels aligned to the leaf level.
//           Inside a line the seven LEFT children come first (64 bytes), the seven right children
//           second: a descent only ever compares against left children (proportional_memory.py:61),
//           so a root->leaf walk of a 1M-leaf tree touches 7 half-lines instead of 21 lines and the
//           right half is read once, for the priority of the leaf that was reached.  Logical
//           node indices (what sample() returns and update() takes) are unchanged; backup()/restore()
//           convert to/from the reference's flat heap order.
//   state : { float64 max_priority; int64 size; int64 write; } -- lives on the device so that
//           add/sample/update are HIP-graph capturable (no host-side scalars frozen at capture).
//
// Bit-exactness contract (SURVEY.md section 7 "hard parts"): the reference propagates fp64
// deltas leaf->root per update, in call order (proportional_memory.py:49-54,81-86).  fp64 adds
// do not re-associate, so every kernel below applies, for every tree node, exactly the
// reference's sequence of `tree[node] += change_i` in list order i.  Parallelism comes from
// processing different NODES concurrently, never from re-ordering one node's additions.
// Compile with -ffp-contract=off (see Makefile): a fused multiply-add would change roundings.
//
// Kernels and their bounds (algorithmic bytes per unit; DESIGN.md has the derivation):
//   per_sample : (depth+1)*8 B tree reads + 8 B uniform + 16 B out per draw  -> HBM/L2 bound
//   per_update : 16 B leaf R/W + depth*16 B ancestor RMW per index           -> latency bound at B=32
//   per_add    : same as update + 8 B priority                               -> root chain (n fp64 adds)
#include <new>
#include <type_traits>

#include "srlx_common.h"
#include "srlx_store_dev.h"

namespace {

using i64 = int64_t;
using u64 = unsigned long long;

struct PerState {
    double max_priority;
    i64 size;
    i64 write;
    i64 pad;
};

constexpr int kWgSample = 256;       // threads of the single-workgroup sample kernel
constexpr i64 kSmallSampleMax = 8192;  // uniforms handled by the single-workgroup path
constexpr int kWgUpdate = 256;
constexpr int kUpdateChunk = 1024;   // indices per general-update launch (LDS resident)
constexpr int kWgAdd = 256;
constexpr i64 kSmallAddMax = 8192;   // adds handled by the single-workgroup path (changes in LDS: 64 KB at 8192 = 8 GPUs x 1024 envs)

__host__ __device__ __forceinline__ int node_depth(i64 x) {
#ifdef __HIP_DEVICE_COMPILE__
    return 63 - __clzll((u64)(x + 1));
#else
    return 63 - __builtin_clzll((unsigned long long)(x + 1));
#endif
}

// Blocked physical layout of the heap.  Levels below the root are cut into groups of three, aligned to
// the deepest level D: the top group has h0 = ((D-1) % 3) + 1 levels (1..h0), then G groups of 3.  The
// nodes of a group that descend from the same "owner" (the ancestor on the level just above the
// group) share one 16-double block.  A node at relative depth r (1..3) with in-block offset m (0..2^r-1)
// with even m (a left child) sits in slot {0,1,3}[r-1] + (m>>1) (slots 0..6: the line's first half); the right children fill the second half DEEPEST FIRST --
// relative depth 3 in slots 8..11, depth 2 in 12..13, depth 1 in 14 -- so that the bulk walk's last group gets every value it can need (7 left children to
// compare against, 4 right leaves) from the first 96 bytes of the line; block 0 slot 7 is the root.  base[g] = index of the first block of group g.
// A LEAF on depth D-1 (capacities that are not powers of two) is also MIRRORED into the slot its left child would have (`Tree::mirror`): the walk reads such a
// leaf's priority from the left half it fetched anyway, whichever side of its parent the leaf hangs on (round 5: the dependent read of the right sibling after
// the walk cost 4.5 of the 48 us of a 2^20-draw call).
constexpr int kRootSlot = 7;
__host__ __device__ __forceinline__ int slot_in_block(int r, int64_t m) {
    const int h = (int)(m >> 1);
    if (m & 1) return r == 1 ? 14 : (r == 2 ? 12 + h : 8 + h);
    return (r == 1 ? 0 : (r == 2 ? 1 : 3)) + h;
}
// slot of the right sibling of the left child in slot s
__host__ __device__ __forceinline__ int right_of_left_slot(int s) { return s == 0 ? 14 : (s < 3 ? s + 11 : s + 5); }
struct Tree {
    double *T;
    i64 len;  // 2 * capacity - 1 logical nodes
    int D, h0, G;
    i64 base[12];

    // base[g] in closed form, g >= 1: 1 + 2^h0 (8^(g-1) - 1) / 7.  A lane-dependent index into the by-value array makes hipcc copy the array to scratch
    // memory (every kernel that calls phys() then starts with scratch stores and pays scratch latency per look-up: the 1024-leaf add spent 11 of its 27 us
    // there).  (8^j - 1) / 7 is taken as a bit pattern: written with the 64-bit division, the expression came out wrong inside the add kernels (ROCm 7.2 hipcc,
    // gfx950: correct in a stand-alone probe, tools/README.md) -- tests/test_per_gpu.py::test_full_size_1M_bulk_paths caught it.
    __device__ __forceinline__ i64 base_of(int g) const { return 1 + (i64)((0x1249249249249249ull & ((1ull << (3 * (g - 1))) - 1ull)) << h0); }  // (8^(g-1) - 1) / 7 = 0b...001001001
    __device__ __forceinline__ i64 phys(i64 i) const {
        if (i == 0) return kRootSlot;
        const int d = node_depth(i);
        const i64 q = i + 1 - ((i64)1 << d);
        if (d <= h0) return slot_in_block(d, q);
        const int k = d - h0 - 1;
        const int g = 1 + k / 3, r = k % 3 + 1;
        const i64 blk = base_of(g) + (q >> r);
        return 16 * blk + slot_in_block(r, q & (((i64)1 << r) - 1));
    }
    __device__ __forceinline__ double get(i64 i) const { return T[phys(i)]; }
    __device__ __forceinline__ void set(i64 i, double v) const {
        T[phys(i)] = v;
        mirror(i, v);
    }
    // every store to a LEAF goes through here (or through set): a leaf on depth D-1 also lives where its left child would
    __device__ __forceinline__ void mirror(i64 i, double v) const {
        if (2 * i + 1 >= len && D >= 1 && node_depth(i) == D - 1) T[phys(2 * i + 1)] = v;
    }
    // block that holds the children of owner node `owner` on level `level` (level = 0, h0, h0+3, ...)
    __device__ __forceinline__ i64 block_of_owner(i64 owner, int level) const {
        if (level == 0) return 0;
        return base_of(1 + (level - h0) / 3) + (owner + 1 - ((i64)1 << level));
    }
};

// (|x|+eps)^alpha.  kind F64: numpy float64 semantics (sqrt fast path at 0.5 like np.power);
// kind F32: the expression evaluated in float32 (numpy keeps a float32 array in float32,
// proportional_memory.py:172), correctly rounded, then widened.
__device__ __forceinline__ double transform_f64(double v, double eps, double alpha) {
    double x = fabs(v) + eps;
    if (alpha == 0.5) return __dsqrt_rn(x);
    if (alpha == 1.0) return x;
    return pow(x, alpha);
}
__device__ __forceinline__ double transform_f32(float v, double eps, double alpha) {
    float x = fabsf(v) + (float)eps;
    float a = (float)alpha;
    // sqrt in fp64 then one rounding to fp32 is the correctly rounded fp32 sqrt (53 >= 2*24+2);
    // __fsqrt_rn lowers to the 1-ulp native instruction on gfx950 (measured: 15% of values differ)
    if (a == 0.5f) return (double)(float)__dsqrt_rn((double)x);
    if (a == 1.0f) return (double)x;
    return (double)(float)pow((double)x, (double)a);
}
__device__ __forceinline__ double load_prio(const void *prio, int kind, i64 i, double eps, double alpha,
                                            double max_priority) {
    switch (kind) {
        case SRLX_PRIO_NONE: return max_priority;
        case SRLX_PRIO_F64: return transform_f64(((const double *)prio)[i], eps, alpha);
        case SRLX_PRIO_F32: return transform_f32(((const float *)prio)[i], eps, alpha);
        case SRLX_PRIO_NONE_MASKED: return ((const unsigned char *)prio)[i] ? max_priority : 0.0;
        case SRLX_PRIO_EST_F32: {
            const float x = ((const float *)prio)[i];
            return x < -1.5f ? 0.0 : (x < 0.f ? max_priority : transform_f64((double)x, eps, alpha));
        }
        default: return ((const double *)prio)[i];
    }
}

// ------------------------------------------------------------------------------------------
// descent: proportional_memory.py:56-66 (_retrieve) + :88-92 (get).  Defined next to the bulk walkers below: the same
// block-wise walk, seven dependent half-line fetches per draw instead of twenty dependent pair loads.
// ------------------------------------------------------------------------------------------
__device__ void descend(const Tree &tr, double val, i64 &out_idx, double &out_p);

__device__ __forceinline__ double beta_of(double beta_initial, double beta_steps, i64 step) {
    // proportional_memory.py:138-140
    double beta = beta_initial + ((1.0 - beta_initial) * (double)step) / beta_steps;
    return beta > 1.0 ? 1.0 : beta;
}

// importance weight (size * p / total)^(-beta)  (proportional_memory.py:163-164).  exp2(-beta*log2(x)) is
// within a few fp64 ulp of pow(x, -beta) (|log2 x| < 64, beta <= 1) at a third of the instruction count.
__device__ __forceinline__ double is_weight(double size, double p, double total, double beta) {
    return exp2(-beta * log2(size * (p / total)));
}

// out-of-line copy for the bulk walk: inlined, its polynomial constants are hoisted into VGPRs for the whole kernel and
// push the walk (128-VGPR budget at 16 waves per CU) into scratch spills
__device__ __attribute__((noinline)) double is_weight_call(double size, double p, double total, double beta) {
    return is_weight(size, p, total, beta);
}

// block-wide max of one double per thread (blockDim.x <= 1024, power of two)
__device__ __forceinline__ double block_max(double v, double *red) {
    const int t = threadIdx.x;
    red[t] = v;
    __syncthreads();
    for (int s = blockDim.x >> 1; s > 0; s >>= 1) {
        if (t < s) red[t] = fmax(red[t], red[t + s]);
        __syncthreads();
    }
    double r = red[0];
    __syncthreads();
    return r;
}

// block-wide exclusive scan of one int per thread; returns exclusive prefix, *total = sum
__device__ __forceinline__ int block_exscan(int v, int *buf, int *total) {
    const int t = threadIdx.x;
    const int n = blockDim.x;
    buf[t] = v;
    __syncthreads();
    for (int off = 1; off < n; off <<= 1) {
        int add = (t >= off) ? buf[t - off] : 0;
        __syncthreads();
        buf[t] += add;
        __syncthreads();
    }
    int incl = buf[t];
    *total = buf[n - 1];
    __syncthreads();
    return incl - v;
}

struct SampleArgs {
    Tree tr;
    const PerState *state;
    double beta_initial, beta_steps;
    i64 step;
    const i64 *d_step;
    int has_duplicate;
    const double *uniforms;  // NULL: uniform j = u53(rng_u64(key_seed, *key_counter, j)), and the call advances *key_counter (srlx_per_sample_keyed)
    u64 key_seed;
    i64 *key_counter;
    i64 n_uniforms;
    i64 batch;
    // scratch
    i64 *cand_idx;
    double *cand_p;
    i64 *map;
    double *wtmp;
    // outputs
    i64 *out_idx;
    double *out_w;
    float *out_w32;
    i64 *out_used;
};

// ------------------------------------------------------------------------------------------
// per_sample, single workgroup (the B=32/64 learner call): descent of every supplied uniform,
// in-order acceptance (zero-priority and duplicate rejects consume a uniform each, :146-157),
// prefix-sum compaction, IS weights and max-normalisation (:163-167) in ONE launch.
// ------------------------------------------------------------------------------------------
static inline size_t sample_wg_lds(i64 M) { return (size_t)kWgSample * (8 + 4) + (size_t)((M + 15) & ~(i64)15) + 16 + (M <= kWgSample ? (size_t)kWgSample * 32 : 0); }

__device__ __forceinline__ bool sample_wg_body(const SampleArgs &a, unsigned char *smem) {
    double *red = reinterpret_cast<double *>(smem);                 // blockDim doubles
    int *ibuf = reinterpret_cast<int *>(red + blockDim.x);          // blockDim ints
    unsigned char *flags = reinterpret_cast<unsigned char *>(ibuf + blockDim.x);  // n_uniforms bytes

    const int t = threadIdx.x, T = blockDim.x;
    const i64 M = a.n_uniforms, B = a.batch;
    // the candidates, the compaction map and the raw weights: in LDS when one uniform per thread is all there is (the learner's B = 32 / 64 draw, the host shim's
    // first attempt) -- through the global scratch every phase boundary was a write -> barrier -> read round trip to L2 (sample_wg_lds() sizes the launch)
    i64 *cand_idx = a.cand_idx, *map = a.map;
    double *cand_p = a.cand_p, *wtmp = a.wtmp;
    if (M <= T) {
        cand_idx = reinterpret_cast<i64 *>(flags + ((M + 15) & ~(i64)15));
        cand_p = reinterpret_cast<double *>(cand_idx + T);
        map = reinterpret_cast<i64 *>(cand_p + T);
        wtmp = reinterpret_cast<double *>(map + T);
    }
    const double total = a.tr.T[kRootSlot];  // :135 (root)

    // phase 1: one descent per uniform (coalesced uniform reads, strided assignment)
    const u64 kc = a.uniforms ? 0 : (u64)a.key_counter[0];
    for (i64 j = t; j < M; j += T) {
        i64 idx;
        double p;
        const double u = a.uniforms ? a.uniforms[j] : srlx::u53(srlx::rng_u64(a.key_seed, kc, (u64)j));
        descend(a.tr, u * total, idx, p);  // :147-148
        cand_idx[j] = idx;
        cand_p[j] = p;
    }
    __syncthreads();
    if (!a.uniforms && t == 0) a.key_counter[0] = (i64)kc + 1;  // like srlx_rng_uniform: one counter value per call (every thread has read it)

    // phase 2: acceptance.  A draw is rejected if its leaf priority is 0 (:150-152) or, without
    // duplicates, if an earlier non-zero draw already produced the same leaf (:155-156).
    for (i64 j = t; j < M; j += T) {
        bool ok = cand_p[j] != 0.0;
        if (ok && !a.has_duplicate) {
            const i64 me = cand_idx[j];
            for (i64 k = 0; k < j; k++)
                if (cand_idx[k] == me && cand_p[k] != 0.0) {
                    ok = false;
                    break;
                }
        }
        flags[j] = ok ? 1 : 0;
    }
    __syncthreads();

    // phase 3: ordered compaction.  Thread t owns the contiguous chunk [t*c, (t+1)*c).
    const i64 c = (M + T - 1) / T;
    const i64 lo = (i64)t * c, hi = (lo + c < M) ? lo + c : M;
    int cnt = 0;
    for (i64 j = lo; j < hi; j++) cnt += flags[j];
    int total_ok;
    int pos = block_exscan(cnt, ibuf, &total_ok);
    for (i64 j = lo; j < hi; j++) {
        if (flags[j]) {
            if (pos < B) map[pos] = j;
            if (pos == B - 1) *a.out_used = j + 1;  // uniforms consumed = index of the B-th accept + 1
            pos++;
        }
    }
    if (t == 0 && total_ok < B) *a.out_used = -1;
    __syncthreads();
    if (total_ok < B) return false;

    // phase 4: importance weights (:163-167)
    // (d_step == key_counter: the call's own draw number is the step -- an engine whose every draw feeds exactly one update; the counter itself has moved on by now)
    const i64 step = a.d_step ? (a.d_step == a.key_counter ? (i64)kc : *a.d_step) : a.step;
    const double beta = beta_of(a.beta_initial, a.beta_steps, step);
    const double size = (double)a.state->size;
    double wmax_local = 0.0;
    for (i64 i = t; i < B; i += T) {
        const i64 j = map[i];
        const double w = is_weight(size, cand_p[j], total, beta);
        wtmp[i] = w;
        a.out_idx[i] = cand_idx[j];
        wmax_local = fmax(wmax_local, w);
    }
    const double wmax = block_max(wmax_local, red);
    for (i64 i = t; i < B; i += T) {
        const double w = wtmp[i] / wmax;
        if (a.out_w) a.out_w[i] = w;
        if (a.out_w32) a.out_w32[i] = (float)w;
    }
    return true;
}

__global__ void __launch_bounds__(kWgSample) k_sample_wg(SampleArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    (void)sample_wg_body(a, smem);
}

// The learner's draw AND its gather as one launch (round 4): the single-workgroup sampler, then -- same workgroup, the indices it has just written -- item location,
// n-step scalars and both frame-offset tables of the store (srlx_store_dev.h: what k_gather_train of srlx_rollout.hip does).  B <= 64 items.
struct GatherArgs {
    srlxs::StoreDev store;
    srlxs::ItemMeta *meta;
    int32_t *actions;
    float *rewards, *terminated;
    i64 *off_all, *off_next;
};
__global__ void __launch_bounds__(kWgSample) k_sample_gather_wg(SampleArgs a, GatherArgs g) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    __shared__ srlxs::ItemMeta sm[srlxs::kTrainItems];
    if (!sample_wg_body(a, smem)) return;  // (not enough accepted draws: out_used = -1, the batch is not touched)
    __syncthreads();  // out_idx is complete (written by this workgroup: visible through the CU's own cache)
    srlxs::gather_train_items(g.store, 0, (int)a.batch, a.out_idx, g.meta, g.actions, g.rewards, g.terminated, g.off_all, g.off_next, sm);
}

// ------------------------------------------------------------------------------------------
// per_sample, bulk path (thousands..millions of draws per launch: prefetching learners, the
// PER micro-benchmark).  The walk is done one 128-byte block (three levels) at a time: the blocks of
// the top groups (levels <= 11 of a 1M-leaf tree, 37.5 KiB) are staged in LDS, each lower group costs
// ONE cache line per draw (its 7 x 16-byte loads are issued together, the three decisions are then
// register selects), and every lane walks kIlp independent draws so that their line fetches overlap.
// The IS weight of the no-rejection fast path is fused in.
// Measured alternatives on MI355X (1M draws from 1M leaves, whole call): one
// dependent 16-byte pair load per level 104 us (blocked layout) / 76 us (flat heap, 64 KiB LDS top);
// 8-lane cooperative line fetch through LDS 206 us; this kernel 84 us.  All of them are bound by the
// L2-miss traffic of the bottom three levels (~190 B per draw out of the Infinity Cache), not by HBM.
// ------------------------------------------------------------------------------------------
constexpr int kBulkTopSmall = 296;        // left halves staged by a 256-thread workgroup of the flat walk (18.5 KiB)
constexpr int kBulkTopBig = 2400;         // ... by a 1024-thread workgroup (150 KiB + 8 KiB reduction buffer of the 160 KiB)
constexpr i64 kBulkBigMin = (i64)1 << 19; // draws from which the one-workgroup-per-CU configuration is used

// One draw in flight.  The reference reads the priority of the leaf it reached after the walk (:88-92); here the
// walk remembers the last decision instead: the left value it compared against and where that node's right
// sibling lives, so the right half of a line is only ever read for that one value.
struct Draw {
    i64 idx;
    double val;
    double pl;     // value of the left child at the last decision (the root total before any decision)
    i64 rpos;      // physical position of its right sibling
    bool wl;       // last decision went left
    bool live;     // not at a leaf yet
};

// walk up to `levels` levels inside the block whose 8 left slots are in registers (four 16-byte loads)
__device__ __forceinline__ void walk_block_regs(const double (&L)[8], i64 blockpos, int levels, i64 len, Draw &w) {
    i64 left = 2 * w.idx + 1;
    if (left >= len) { w.live = false; return; }
    double l = L[0];
    bool go = w.val <= l;  // :61
    const int s1 = go ? 0 : 1;
    w.val = go ? w.val : w.val - l;
    w.idx = left + s1;
    w.pl = l; w.wl = go; w.rpos = blockpos + 14;
    if (levels < 2) return;
    left = 2 * w.idx + 1;
    if (left >= len) { w.live = false; return; }
    l = s1 ? L[2] : L[1];
    go = w.val <= l;
    const int s2 = 2 * s1 + (go ? 0 : 1);
    w.val = go ? w.val : w.val - l;
    w.idx = left + (go ? 0 : 1);
    w.pl = l; w.wl = go; w.rpos = blockpos + 12 + s1;
    if (levels < 3) return;
    left = 2 * w.idx + 1;
    if (left >= len) { w.live = false; return; }
    const double la = (s2 & 1) ? L[4] : L[3], lb = (s2 & 1) ? L[6] : L[5];
    l = (s2 >> 1) ? lb : la;
    go = w.val <= l;
    w.val = go ? w.val : w.val - l;
    w.idx = left + (go ? 0 : 1);
    w.pl = l; w.wl = go; w.rpos = blockpos + 8 + s2;
}

// The LAST group's block out of registers (bulk walk; the group spans the depths D-2 .. D, so every live draw ends in it): left half L[0..7] and the four deepest
// right children R4[0..3] (slots 8..11).  Ends the walk: pl = the priority of the leaf reached, wl = true (nothing is left to read).
__device__ __forceinline__ void walk_last_regs(const double (&L)[8], const double (&R4)[4], i64 len, Draw &w) {
    i64 left = 2 * w.idx + 1;  // (a node on depth D-2 has children)
    double l = L[0];
    bool go = w.val <= l;  // :61
    const int s1 = go ? 0 : 1;
    w.val = go ? w.val : w.val - l;
    w.idx = left + s1;
    left = 2 * w.idx + 1;
    l = s1 ? L[2] : L[1];
    go = w.val <= l;
    const int s2 = 2 * s1 + (go ? 0 : 1);
    w.val = go ? w.val : w.val - l;
    w.idx = left + (go ? 0 : 1);
    left = 2 * w.idx + 1;
    const double la = (s2 & 1) ? L[4] : L[3], lb = (s2 & 1) ? L[6] : L[5];
    l = (s2 >> 1) ? lb : la;  // the left child -- or, for a leaf on depth D-1, the leaf's own mirror
    const double ra = (s2 & 1) ? R4[1] : R4[0], rb = (s2 & 1) ? R4[3] : R4[2];
    const double r = (s2 >> 1) ? rb : ra;
    const bool leaf = left >= len;
    go = leaf || w.val <= l;
    w.idx = leaf ? w.idx : left + (go ? 0 : 1);
    w.pl = go ? l : r;
    w.wl = true;
    w.live = false;
}

// one draw of the small (single-workgroup) sampler: block by block out of memory
__device__ void descend(const Tree &tr, double val, i64 &out_idx, double &out_p) {
    Draw w;
    w.idx = 0;
    w.val = val;
    w.pl = tr.T[kRootSlot];
    w.wl = true;
    w.rpos = 0;
    w.live = tr.len > 1;
    int level = 0;
    for (int g = 0; g <= tr.G && w.live; g++) {
        const int levels = g == 0 ? tr.h0 : 3;
        const i64 bp = tr.block_of_owner(w.idx, level) * 16;
        const double2 *src = reinterpret_cast<const double2 *>(tr.T + bp);
        double L[8];
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const double2 v = src[k];
            L[2 * k] = v.x;
            L[2 * k + 1] = v.y;
        }
        walk_block_regs(L, bp, levels, tr.len, w);
        level += levels;
    }
    out_idx = w.idx;
    out_p = w.wl ? w.pl : tr.T[w.rpos];
}

// the same walk out of a block staged in LDS: one dependent 8-byte read per level
__device__ __forceinline__ void walk_block_lds(const double *blk, i64 blockpos, int levels, i64 len, Draw &w) {
    int pos = 0;
#pragma unroll
    for (int r = 1; r <= 3; r++) {
        if (r > levels) break;
        const i64 left = 2 * w.idx + 1;
        if (left >= len) { w.live = false; return; }
        const int slot = (r == 1 ? 0 : (r == 2 ? 1 : 3)) + pos;
        const double l = blk[slot];
        const bool go = w.val <= l;  // :61
        w.val = go ? w.val : w.val - l;
        w.idx = left + (go ? 0 : 1);
        w.pl = l; w.wl = go; w.rpos = blockpos + right_of_left_slot(slot);
        pos = 2 * pos + (go ? 0 : 1);
    }
}

// Dynamic LDS: the LEFT halves (64 bytes) of the first `lds_blocks` blocks, then blockDim doubles for the reduction.
// Large calls run one 1024-thread workgroup per CU with 2341 staged blocks (150 KB: the 14 top levels of a 1M-leaf
// tree, so only two lines per draw are fetched from memory); small calls run 256-thread workgroups with 293 blocks.
// The walk has two phases.  LDS-staged groups whose levels all lie above the shallowest leaf (5 of the 7 groups of a
// 1M-leaf tree) need no end-of-tree test: they run branch-free on 32-bit in-level offsets with the kIlp draws of a
// lane interleaved level by level, so their dependent LDS reads overlap.  The remaining groups use the checked
// walkers above.  (Measured: the kernel is bound by the two scattered line fetches per draw, not by instruction
// issue -- the branch-free phase alone changes its time by < 1 %.)

// one branch-free decision (:61-66): left if val <= l, else subtract and go right
__device__ __forceinline__ void step_free(double l, double &val, unsigned &pos) {
    const bool go = val <= l;
    val = go ? val : val - l;
    pos = 2 * pos + (go ? 0u : 1u);
}

constexpr int kIlp = 2;
// Control words of the bulk sampler (handle-owned, 32 x u64).  [0] epoch, written by k_compact_bulk only; [1] the epoch the running call uses, written by
// k_descend_bulk only (kernels of one call and of successive calls are stream-ordered, so neither word is ever written by the kernel that reads it);
// [2 + 2 s], [3 + 2 s] for s = epoch & 1: zero-priority draws seen by the walk / max weight of the draws < B (bits) -- a call accumulates into set s and clears
// set 1 - s, so nothing has to be re-armed behind it (round 5: the third launch, k_finish_slow, is gone); [6] max weight of the compacted draws (bits),
// [7] tile tickets, [8] workgroups that left the compaction: re-armed by the compaction's last workgroup.
constexpr int kCtlEpoch = 0, kCtlUse = 1, kCtlSets = 2, kCtlSlowMax = 6, kCtlTickets = 7, kCtlDone = 8, kCtlWords = 32;
// kAllStaged: every block of the tree is staged in LDS (small trees); else the last group -- and for deep trees the groups just above it -- is fetched.
// Every group but the last lies above the shallowest leaf (the last group spans the depths D-2 .. D), so all of them are walked branch-free on 32-bit in-level
// offsets, out of LDS or out of the four 16-byte pieces of a fetched line's left half; only the last group can end a walk.
template <int kThreads, bool kAllStaged>
__global__ void __launch_bounds__(kThreads, 4) k_descend_bulk(SampleArgs a, int lds_blocks, u64 *ctl) {
    extern __shared__ __attribute__((aligned(16))) double bulk_smem[];
    double *top = bulk_smem;                      // block b, left slot s (0..7) at top[8 * b + s]
    double *red = bulk_smem + (size_t)lds_blocks * 8;
    const Tree tr = a.tr;
    {   // staging: eight loads of a thread in flight before its first LDS store (a rolled loop is one dependent memory round trip per 16 KB: 3 us of the call)
        constexpr int kStage = 8;
        const int pieces = lds_blocks * 4;
        int k0 = threadIdx.x;
        for (; k0 + (kStage - 1) * kThreads < pieces; k0 += kStage * kThreads) {
            double2 v[kStage];
#pragma unroll
            for (int u = 0; u < kStage; u++) {
                const int k = k0 + u * kThreads;
                v[u] = reinterpret_cast<const double2 *>(tr.T)[(k >> 2) * 8 + (k & 3)];
            }
#pragma unroll
            for (int u = 0; u < kStage; u++) reinterpret_cast<double2 *>(top)[k0 + u * kThreads] = v[u];
        }
        for (; k0 < pieces; k0 += kThreads) reinterpret_cast<double2 *>(top)[k0] = reinterpret_cast<const double2 *>(tr.T)[(k0 >> 2) * 8 + (k0 & 3)];
    }
    const u64 epoch = ctl[kCtlEpoch];
    u64 *zero_count = ctl + kCtlSets + 2 * (epoch & 1), *wmax_bits = zero_count + 1;
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        ctl[kCtlUse] = epoch;
        ctl[kCtlSets + 2 * (1 - (epoch & 1))] = 0;  // the other set: the next call's, untouched by this one
        ctl[kCtlSets + 2 * (1 - (epoch & 1)) + 1] = 0;
    }
    __syncthreads();
    const double total = top[kRootSlot];
    const i64 M = a.n_uniforms, B = a.batch, len = tr.len;
    const i64 step = a.d_step ? *a.d_step : a.step;
    const double beta = beta_of(a.beta_initial, a.beta_steps, step);
    const double size = (double)a.state->size;
    const i64 stride = (i64)gridDim.x * blockDim.x;
    unsigned zeros = 0;
    double wmax = 0.0;
    for (i64 base = (i64)blockIdx.x * blockDim.x + threadIdx.x; base < M; base += stride * kIlp) {
        double val[kIlp];
        unsigned q[kIlp];  // offset of the current owner node inside its level
#pragma unroll
        for (int d = 0; d < kIlp; d++) {
            const i64 j = base + d * stride;
            val[d] = j < M ? a.uniforms[j] * total : 0.0;  // :147 (a draw past M walks the left edge, harmlessly)
            q[d] = 0;
        }
        int level = 0;  // level of the current owner nodes (same for every draw)
        for (int g = 0; g < tr.G; g++) {  // the groups above the last
            const int levels = g == 0 ? tr.h0 : 3;
            const unsigned gbase = (unsigned)tr.base[g];
            unsigned pos[kIlp];
            if (kAllStaged || g == 0 || tr.base[g + 1] <= lds_blocks) {  // staged: one dependent 8-byte LDS read per level, the draws of a lane interleaved
                const double *blk[kIlp];
                double l[kIlp];
#pragma unroll
                for (int d = 0; d < kIlp; d++) {
                    blk[d] = top + (size_t)(gbase + q[d]) * 8;
                    pos[d] = 0;
                }
#pragma unroll
                for (int d = 0; d < kIlp; d++) l[d] = blk[d][0];
#pragma unroll
                for (int d = 0; d < kIlp; d++) step_free(l[d], val[d], pos[d]);
                if (levels >= 2) {
#pragma unroll
                    for (int d = 0; d < kIlp; d++) l[d] = blk[d][1 + pos[d]];
#pragma unroll
                    for (int d = 0; d < kIlp; d++) step_free(l[d], val[d], pos[d]);
                }
                if (levels >= 3) {
#pragma unroll
                    for (int d = 0; d < kIlp; d++) l[d] = blk[d][3 + pos[d]];
#pragma unroll
                    for (int d = 0; d < kIlp; d++) step_free(l[d], val[d], pos[d]);
                }
            } else {  // fetched (three levels): the left half of one line per draw, the decisions are register selects
                double2 P[kIlp][4];
#pragma unroll
                for (int d = 0; d < kIlp; d++) {
                    const double2 *src = reinterpret_cast<const double2 *>(tr.T + (size_t)(gbase + q[d]) * 16);
#pragma unroll
                    for (int k = 0; k < 4; k++) P[d][k] = src[k];
                }
#pragma unroll
                for (int d = 0; d < kIlp; d++) {
                    pos[d] = 0;
                    step_free(P[d][0].x, val[d], pos[d]);
                    const unsigned s1 = pos[d];
                    step_free(s1 ? P[d][1].x : P[d][0].y, val[d], pos[d]);
                    const unsigned s2 = pos[d];
                    const double la = (s2 & 1) ? P[d][2].x : P[d][1].y, lb = (s2 & 1) ? P[d][3].x : P[d][2].y;
                    step_free((s2 >> 1) ? lb : la, val[d], pos[d]);
                }
            }
#pragma unroll
            for (int d = 0; d < kIlp; d++) q[d] = (q[d] << levels) + pos[d];
            level += levels;
        }
        // the last group
        Draw w[kIlp];
#pragma unroll
        for (int d = 0; d < kIlp; d++) {
            const i64 j = base + d * stride;
            w[d].live = j < M && len > 1;
            w[d].val = val[d];
            w[d].idx = (i64)q[d] + (((i64)1 << level) - 1);
            w[d].pl = total;
            w[d].wl = true;
            w[d].rpos = 0;
        }
        double pr[kIlp];
        if (kAllStaged || tr.G == 0) {  // out of LDS, with end-of-tree tests; then the one read of a right child: the leaf's own priority
#pragma unroll
            for (int d = 0; d < kIlp; d++)
                if (w[d].live) {
                    const i64 blk = tr.block_of_owner(w[d].idx, level);
                    walk_block_lds(top + blk * 8, blk * 16, tr.G == 0 ? tr.h0 : 3, len, w[d]);
                }
#pragma unroll
            for (int d = 0; d < kIlp; d++) pr[d] = w[d].wl ? 0.0 : tr.T[w[d].rpos];
        } else {  // fetched: 96 bytes per draw, and the leaf's priority comes out of them (no dependent read behind the walk)
            double L[kIlp][8], R4[kIlp][4];
            const unsigned gbase = (unsigned)tr.base[tr.G];
#pragma unroll
            for (int d = 0; d < kIlp; d++) {
                const double2 *src = reinterpret_cast<const double2 *>(tr.T + (size_t)(gbase + q[d]) * 16);
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    const double2 v = src[k];
                    L[d][2 * k] = v.x;
                    L[d][2 * k + 1] = v.y;
                }
#pragma unroll
                for (int k = 0; k < 2; k++) {
                    const double2 v = src[4 + k];
                    R4[d][2 * k] = v.x;
                    R4[d][2 * k + 1] = v.y;
                }
            }
#pragma unroll
            for (int d = 0; d < kIlp; d++) {
                if (w[d].live) walk_last_regs(L[d], R4[d], len, w[d]);
                pr[d] = 0.0;
            }
        }
#pragma unroll
        for (int d = 0; d < kIlp; d++) {
            const i64 j = base + d * stride;
            if (j >= M) continue;
            const double p = w[d].wl ? w[d].pl : pr[d];
            const double wi = p == 0.0 ? 0.0 : is_weight_call(size, p, total, beta);  // :163-164, computed once per draw
            // scratch for the rejecting path (struct of arrays) + the speculative output: with nothing rejected draw j IS output j
            (j < B ? a.out_idx : a.cand_idx)[j] = w[d].idx;  // (the compaction reads draw j's leaf from where it was written: tile sources are in registers before a tile publishes)
            a.cand_p[j] = wi;
            if (p == 0.0) {
                zeros++;
            } else if (j < B) {  // fast path: with no rejection output i is draw i
                wmax = fmax(wmax, wi);
            }
        }
    }
    if (zeros) atomicAdd(zero_count, (u64)zeros);
    const double m = block_max(wmax, red);
    if (threadIdx.x == 0 && m > 0.0) atomicMax(wmax_bits, (u64)__double_as_longlong(m));  // positive doubles order like their bits
}

// ------------------------------------------------------------------------------------------
// After the walk: ONE launch (round 5; the control words are described at k_descend_bulk).
//   nothing was rejected (the common case): output i is draw i, the walk already wrote the indices, one pass writes the
//                   normalised weights (:163-167);
//   else            the ORDERED compaction of the accepted draws (in-order rejection, :146-157) over the whole device: one
//                   2048-draw tile per workgroup, tile offsets by a decoupled look-back scan (a tile publishes its count, then
//                   its first wave sums 64 predecessors per step until it meets a resolved prefix); the LAST workgroup to leave
//                   normalises the compacted weights and re-arms tickets and look-back state (rare path: a draw is rejected only
//                   when it lands exactly on a zero-priority leaf's boundary).  No per-call memset anywhere: safe under HIP-graph replay.
// (A counting-sorted "binned" walk -- sort the draws by the subtree they reach so that every walk finishes in
// LDS -- was measured at 100 us against 81 us per 2^20 draws and removed; profiles/NOTES.md keeps the numbers.)
// ------------------------------------------------------------------------------------------
constexpr int kTileThreads = 256, kTilePer = 8, kTile = kTileThreads * kTilePer;
constexpr u64 kStAggregate = 1ull << 62, kStPrefix = 2ull << 62, kStMask = 3ull << 62;

__global__ void __launch_bounds__(kTileThreads) k_compact_bulk(SampleArgs a, u64 *ctl, u64 *tile_state, i64 ntiles, int slow_workgroups) {
    __shared__ int ibuf[kTileThreads];
    __shared__ double dred[kTileThreads];
    __shared__ i64 s_prefix;
    __shared__ unsigned s_tile;
    const i64 M = a.n_uniforms, B = a.batch;
    const int t = threadIdx.x;
    const u64 epoch = ctl[kCtlUse];
    const u64 *set = ctl + kCtlSets + 2 * (epoch & 1);
    if (blockIdx.x == 0 && t == 0) ctl[kCtlEpoch] = epoch + 1;  // (read by the NEXT call's walk only)
    if (set[0] == 0) {  // nothing was rejected: output i is draw i, the walk already wrote the indices
        if (blockIdx.x == 0 && t == 0) *a.out_used = (M >= B) ? B : -1;
        if (M < B) return;
        const double wmax = __longlong_as_double((long long)set[1]);
        for (i64 i = (i64)blockIdx.x * blockDim.x + t; i < B; i += (i64)gridDim.x * blockDim.x) {
            const double w = a.cand_p[i] / wmax;  // :167
            if (a.out_w) a.out_w[i] = w;
            if (a.out_w32) a.out_w32[i] = (float)w;
        }
        return;
    }
    if ((int)blockIdx.x >= slow_workgroups) return;  // the grid is sized for the normalising pass; the compaction wants fewer, persistent workgroups
    const int participants = (int)gridDim.x < slow_workgroups ? (int)gridDim.x : slow_workgroups;
    // tiles are handed out in scheduling order, so every predecessor a tile waits for has been taken by a running workgroup
    for (;;) {
        __syncthreads();
        if (t == 0) s_tile = (unsigned)atomicAdd(&ctl[kCtlTickets], 1ull);
        __syncthreads();
        const i64 tile = s_tile;
        if (tile >= ntiles) break;
        const i64 j0 = tile * kTile + (i64)t * kTilePer;
        double w[kTilePer];
        i64 src[kTilePer];  // the draws' leaves, in registers BEFORE this tile publishes anything: a later tile may then overwrite out_idx[j < B] in place
        int cnt = 0;
#pragma unroll
        for (int k = 0; k < kTilePer; k++) {
            const i64 j = j0 + k;
            w[k] = j < M ? a.cand_p[j] : 0.0;
            src[k] = j < M ? (j < B ? a.out_idx : a.cand_idx)[j] : 0;
            cnt += w[k] != 0.0 ? 1 : 0;
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the loads have RETURNED (a published count must not overtake them)
        int tot;
        const int local = block_exscan(cnt, ibuf, &tot);
        if (t < 64) {  // first wave: decoupled look-back
            if (t == 0)
                __hip_atomic_store(&tile_state[tile], (tile == 0 ? kStPrefix : kStAggregate) | (u64)tot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            i64 excl = 0, hi = tile - 1;
            bool done = tile == 0;
            while (!done) {
                const i64 p = hi - t;
                u64 v = kStPrefix;  // "tiles" before the first: resolved, nothing accepted
                if (p >= 0) {
                    do {
                        v = __hip_atomic_load(&tile_state[p], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    } while ((v & kStMask) == 0);
                }
                const u64 resolved = __ballot((v & kStMask) == kStPrefix);
                const int first = resolved ? __ffsll((unsigned long long)resolved) - 1 : 64;
                i64 part = t <= first ? (i64)(v & ~kStMask) : 0;
#pragma unroll
                for (int off = 32; off > 0; off >>= 1) part += __shfl_xor(part, off);
                excl += part;
                done = resolved != 0;
                hi -= 64;
            }
            if (t == 0) {
                if (tile != 0) __hip_atomic_store(&tile_state[tile], kStPrefix | (u64)(excl + tot), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                s_prefix = excl;
            }
        }
        __syncthreads();
        const i64 prefix = s_prefix;
        i64 pos = prefix + local;
        double wm = 0.0;
#pragma unroll
        for (int k = 0; k < kTilePer; k++) {
            if (w[k] == 0.0) continue;
            if (pos < B) {
                a.out_idx[pos] = src[k];
                a.wtmp[pos] = w[k];
                wm = fmax(wm, w[k]);
                if (pos == B - 1) *a.out_used = j0 + k + 1;  // uniforms consumed = index of the B-th accept + 1
            }
            pos++;
        }
        wm = block_max(wm, dred);
        if (t == 0) {
            if (wm > 0.0) atomicMax(&ctl[kCtlSlowMax], (u64)__double_as_longlong(wm));  // positive doubles order like their bits
            if (tile == ntiles - 1 && prefix + tot < B) *a.out_used = -1;
        }
    }
    // leave: the last workgroup out normalises and re-arms (release of this workgroup's writes, ticket, acquire by the last one)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (t == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        s_tile = (unsigned)__hip_atomic_fetch_add(&ctl[kCtlDone], 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
    if ((int)s_tile != participants - 1) return;
    if (t == 0) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    __syncthreads();
    for (i64 i = t; i < ntiles; i += kTileThreads) tile_state[i] = 0;
    const i64 used = __hip_atomic_load(a.out_used, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const double wmax = __longlong_as_double((long long)__hip_atomic_load(&ctl[kCtlSlowMax], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
    if (used >= 0)
        for (i64 i = t; i < B; i += kTileThreads) {
            const double w = a.wtmp[i] / wmax;  // :167
            if (a.out_w) a.out_w[i] = w;
            if (a.out_w32) a.out_w32[i] = (float)w;
        }
    __syncthreads();
    if (t == 0) ctl[kCtlSlowMax] = ctl[kCtlTickets] = ctl[kCtlDone] = 0;
}

// ------------------------------------------------------------------------------------------
// per_update, general indices (proportional_memory.py:171-177), one workgroup, n <= kUpdateChunk.
//   1. p_i = transform(priority_i); change_i = p_i - (value the leaf holds when step i runs)
//      -- a repeated index sees the earlier write of the same call (:173-175)
//   2. the last occurrence of each leaf stores p_i
//   3. every touched ancestor is owned by ONE thread which replays `node += change_i` for the
//      contributing i in list order (the reference's _propagate order for that node)
//   4. max_priority = max(max_priority, max_i p_i)  (:176-177)
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ bool is_ancestor(i64 a, int da, i64 x, int dx) {
    return dx > da && (((x + 1) >> (dx - da)) == a + 1);
}

__global__ void __launch_bounds__(1024) k_update_wg(Tree tr, PerState *state, i64 n,
                                                          const i64 *indices, const void *prio, int kind, double eps,
                                                          double alpha, int *err_flag, i64 *bump) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    if (bump && threadIdx.x == 0) *bump += 1;  // srlx_per_set_update_counter: the caller's count of priority write-backs (read by later launches only)
    double *s_p = reinterpret_cast<double *>(smem);
    double *s_chg = s_p + n;
    double *red = s_chg + n;                                    // blockDim doubles (only [0] is used: the maximum as a bit pattern)
    i64 *s_idx = reinterpret_cast<i64 *>(red + blockDim.x);     // n
    int *s_dep = reinterpret_cast<int *>(s_idx + n);            // n
    int *s_shared = s_dep + n;                                  // n: depth of the deepest node index i shares with ANY other index of the call (-1: none)
    const int t = threadIdx.x, T = blockDim.x;
    const double maxp0 = state->max_priority;
    if (t == 0) reinterpret_cast<u64 *>(red)[0] = 0ull;

    double pmax = 0.0;
    for (i64 i = t; i < n; i += T) {
        i64 x = indices[i];
        if (x < 0 || x >= tr.len) {  // the reference would raise IndexError; flag and neutralise
            *err_flag = 1;
            x = 0;
        }
        const double p = load_prio(prio, kind, i, eps, alpha, maxp0);
        s_p[i] = p;
        s_idx[i] = x;
        s_dep[i] = node_depth(x);
        s_shared[i] = -1;
        pmax = fmax(pmax, p);
    }
    __syncthreads();
    // The learner's call (n * depth <= 2 x the workgroup: 32 / 64 indices of a 1M-leaf tree): every value the call will read from the tree -- the old leaf of index
    // t, the old value of the <= 2 ancestors a thread owns a task for -- is REQUESTED here, before the pair pass and the scans below, which then run while the loads
    // travel (round 5: the leaf read and the ancestor read were two dependent memory round trips behind them).  A load of a node that turns out to be somebody
    // else's is dropped.
    const int maxd = tr.D;
    const i64 tasks = n * (i64)maxd;
    // n <= 64 (the learner's calls): the WAVE FORM of the ancestor pass below -- lane i of every wave holds index i, wave w owns the tree depths w and w + 16
    const bool wave_form = n <= 64 && T == 1024 && maxd <= 32;
    const bool pre = !wave_form && tasks <= 2 * (i64)T && n <= T;
    double pre_v[2] = {0.0, 0.0}, pre_leaf = 0.0;
    i64 pre_pa[2] = {0, 0};
    unsigned wf_key[2] = {~0u, ~0u};  // position of lane i's ancestor within depth w + 16 r (~0: index i has no ancestor there)
    if (wave_form) {
        const int lane = t & 63, w = t >> 6;
        const i64 x = lane < n ? s_idx[lane] : 0;
        const int dep = lane < n ? s_dep[lane] : 0;
#pragma unroll
        for (int r = 0; r < 2; r++) {
            const int d = w + 16 * r;
            if (d >= dep) continue;
            wf_key[r] = (unsigned)(((x + 1) >> (dep - d)) - ((i64)1 << d));
            pre_pa[r] = tr.phys((((i64)1 << d) - 1) + wf_key[r]);
            pre_v[r] = tr.T[pre_pa[r]];
        }
        if (t < n) pre_leaf = tr.get(x);
    }
    if (pre) {
#pragma unroll
        for (int r = 0; r < 2; r++) {
            const i64 task = t + (i64)r * T;
            if (task >= tasks) continue;
            const i64 i = task % n;
            const int k = (int)(task / n) + 1;
            if (k > s_dep[i]) continue;
            pre_pa[r] = tr.phys(((s_idx[i] + 1) >> k) - 1);
            pre_v[r] = tr.T[pre_pa[r]];
        }
        if (t < n) pre_leaf = tr.get(s_idx[t]);
    }
    // Where do the indices' root paths meet?  Below the deepest node it shares with any other index, an index is the ONLY contributor of its ancestors
    // (14 of the 20 levels of a 64-index call on a 1M-leaf tree), and an index that shares nothing at its own depth has no duplicate: those cases need
    // neither the ownership scan nor the ordered replay below -- O(n) LDS look-ups per task, which is where a call's time went (ablation at 32 / 64 /
    // 128 indices: 11.7 / 24.2 / 53.1 us with the scans, 7.9 / 11.6 / 16.8 us without).  n (n - 1) / 2 pair tests, once: depth of the lowest common ancestor
    // of (i, j) from the heap indices aligned to the shallower one.  (Even n: slot (i, j <= i) of the half square stands for the pair (n-1-i, n-1-j).)
    const bool even = (n & 1) == 0;
    for (i64 pair = t; pair < (even ? n * n / 2 : n * n); pair += T) {
        i64 i = pair / n, j = pair % n;
        if (even && j <= i) i = n - 1 - i, j = n - 1 - j;
        if (i >= j) continue;
        const int di = s_dep[i], dj = s_dep[j], dm = di < dj ? di : dj;
        const u64 u = (u64)(s_idx[i] + 1) >> (di - dm), v = (u64)(s_idx[j] + 1) >> (dj - dm);
        const int lca = u == v ? dm : dm - (64 - __clzll(u ^ v));
        atomicMax(&s_shared[i], lca);
        atomicMax(&s_shared[j], lca);
    }
    __syncthreads();

    unsigned last_me = 0u;  // bit q: this thread's q-th index is the last occurrence of its leaf (kUpdateChunk / kWgUpdate <= 32 rounds)
    int q = 0;
    for (i64 i = t; i < n; i += T, q++) {
        const i64 x = s_idx[i];
        i64 prev = -1;
        bool last = true;
        if (s_shared[i] >= s_dep[i]) {  // somebody shares this very node: look for duplicates of the index
            for (i64 j = 0; j < i; j++)
                if (s_idx[j] == x) prev = j;
            for (i64 j = i + 1; j < n; j++)
                if (s_idx[j] == x) {
                    last = false;
                    break;
                }
        }
        const double before = prev >= 0 ? s_p[prev] : (pre || wave_form ? pre_leaf : tr.get(x));
        s_chg[i] = s_p[i] - before;  // :83
        last_me |= (last ? 1u : 0u) << q;
    }
    __syncthreads();  // every old leaf value has been read
    q = 0;
    for (i64 i = t; i < n; i += T, q++)
        if ((last_me >> q) & 1u) tr.set(s_idx[i], s_p[i]);  // :85

    // ancestors (:49-54).  task (i, k): the k-th ancestor of index i; the first i that reaches a
    // node owns it.
    if (wave_form) {
        // Every lane with an ancestor on this depth replays `node += change_j` over the call's indices in list order (the reference's order for that node): the other
        // indices' node positions and changes come out of the wave's own registers (v_readlane: no LDS round trips, no ownership scan -- lanes under the same node
        // compute the same sum and store the same value).  A depth on which no index of the wave shares a node (the pair pass above) skips the replay.  Round 5:
        // the LDS scans this replaces were 6.7 of a 64-index call's 14.0 us (tools/_upd_abl.sh).
        const int lane = t & 63, w = t >> 6;
        const double chg = lane < n ? s_chg[lane] : 0.0;
        const int sh = lane < n ? s_shared[lane] : -1;
        const int clo = __double2loint(chg), chi = __double2hiint(chg);
#pragma unroll
        for (int r = 0; r < 2; r++) {
            const int d = w + 16 * r;
            const unsigned key = wf_key[r];
            const bool valid = key != ~0u;
            double v = pre_v[r];
            if (__any(valid && d <= sh)) {
                // (fully unrolled over constant lane numbers; lanes past n hold no index: their position never equals a valid one)
                auto replay = [&](auto cnt) {
#pragma unroll
                    for (int j = 0; j < decltype(cnt)::value; j++) {
                        const unsigned kj = (unsigned)__builtin_amdgcn_readlane((int)key, j);
                        const double sum = v + __hiloint2double(__builtin_amdgcn_readlane(chi, j), __builtin_amdgcn_readlane(clo, j));
                        v = kj == key ? sum : v;
                    }
                };
                if (n <= 32)
                    replay(std::integral_constant<int, 32>{});
                else
                    replay(std::integral_constant<int, 64>{});
            } else {
                v += chg;
            }
            if (valid) tr.T[pre_pa[r]] = v;
        }
    }
    for (i64 task = t; !wave_form && task < tasks; task += T) {
        const i64 i = task % n;  // level-major: a wave's lanes sit on the same few levels, so whole waves (the deep levels) skip the scans
        const int k = (int)(task / n) + 1;
        const int dx = s_dep[i];
        if (k > dx) continue;
        const int da = dx - k;
        const i64 a = ((s_idx[i] + 1) >> k) - 1;
        const bool alone = da > s_shared[i];  // nobody else's path passes through this node
        bool owner = true;
        if (!alone) {  // no early exit: with a `break` every iteration waits for its own LDS loads (~150 clocks each, serial); unrolled, eight travel together
#pragma unroll 8
            for (i64 j = 0; j < i; j++) owner &= !is_ancestor(a, da, s_idx[j], s_dep[j]);
        }
        if (!owner) continue;
        const int r = (int)(task / T);
        const i64 pa = pre ? ((r & 1) ? pre_pa[1] : pre_pa[0]) : tr.phys(a);  // (selects, not a run-time index: that would put the arrays into scratch memory)
        double v = pre ? ((r & 1) ? pre_v[1] : pre_v[0]) : tr.T[pa];
        v += s_chg[i];
        if (!alone) {
#pragma unroll 8
            for (i64 j = i + 1; j < n; j++) {
                const double c = s_chg[j];
                if (is_ancestor(a, da, s_idx[j], s_dep[j])) v += c;
            }
        }
        tr.T[pa] = v;
    }

    // max_priority (:176-177): wave maxima, then one LDS atomic per wave (priorities are >= 0: their bit patterns order like the values)
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) pmax = fmax(pmax, __shfl_xor(pmax, off));
    if ((t & 63) == 0 && pmax > 0.0) atomicMax(reinterpret_cast<u64 *>(red), (u64)__double_as_longlong(pmax));
    __syncthreads();
    const double m = __longlong_as_double((long long)reinterpret_cast<u64 *>(red)[0]);
    if (t == 0 && kind != SRLX_PRIO_NONE && maxp0 < m) state->max_priority = m;
}

// ------------------------------------------------------------------------------------------
// per_add of n consecutive ring slots (proportional_memory.py:120-129, :71-79).
// The batch covers at most 3 runs of tree nodes that are contiguous AND on one tree depth
// (ring wrap-around x the two leaf depths of a non-power-of-two capacity).  Within a run the
// leaves under an ancestor are a contiguous slice of the batch, so each ancestor is owned by
// one thread that adds that slice in order; runs are processed in batch order.
// ------------------------------------------------------------------------------------------
struct Run {
    i64 i_lo, i_hi;  // batch positions [i_lo, i_hi)
    i64 x_lo;        // tree node of batch position i_lo
    int dl;          // depth of the run's leaves
};

// run number `want` of the batch (false: there is none) -- make_runs without the array (a lane-uniform but run-time index would put it into scratch memory)
__device__ __forceinline__ bool get_run(i64 cap, i64 write, i64 n, int want, Run &out) {
    const i64 tree_len = 2 * cap - 1;
    const int D = node_depth(tree_len - 1);
    const i64 first_deep = ((i64)1 << D) - 1;
    const i64 sb = first_deep - (cap - 1);
    int nr = 0;
    i64 done = 0;
#pragma unroll
    for (int piece = 0; piece < 2; piece++) {
        const i64 s_lo = piece == 0 ? write : 0;
        const i64 s_hi = piece == 0 ? (write + n < cap ? write + n : cap) : (write + n - cap);
        if (piece == 1 && write + n <= cap) break;
        const i64 mid = (sb > s_lo && sb < s_hi) ? sb : s_lo;
#pragma unroll
        for (int c = 0; c < 2; c++) {
            const i64 a = c == 0 ? s_lo : mid, b = c == 0 ? mid : s_hi;
            if (b <= a) continue;
            if (nr == want) {
                out.i_lo = done;
                out.i_hi = done + (b - a);
                out.x_lo = a + cap - 1;
                out.dl = (a < sb) ? D - 1 : D;
                return true;
            }
            done += b - a;
            nr++;
        }
    }
    return false;
}

// number of ancestor nodes of a run
__device__ __forceinline__ i64 run_tasks(const Run &r) {
    const i64 x_hi = r.x_lo + (r.i_hi - r.i_lo) - 1;
    i64 tot = 0;
    for (int d = r.dl - 1; d >= 0; d--) tot += (((x_hi + 1) >> (r.dl - d)) - ((r.x_lo + 1) >> (r.dl - d))) + 1;
    return tot;
}

// thread `tid` of `nthreads` processes its share of a run's ancestor nodes
__device__ __forceinline__ void run_ancestors(const Tree &tr, const double *chg, const Run &r, i64 tid, i64 nthreads) {
    const i64 cnt = r.i_hi - r.i_lo;
    const i64 x_hi = r.x_lo + cnt - 1;
    const i64 tasks = run_tasks(r);
    for (i64 task = tid; task < tasks; task += nthreads) {
        i64 rem = task;
        int d = r.dl - 1;
        i64 a = 0;
        for (; d >= 0; d--) {
            const i64 a_lo = ((r.x_lo + 1) >> (r.dl - d)) - 1;
            const i64 a_hi = ((x_hi + 1) >> (r.dl - d)) - 1;
            const i64 c = a_hi - a_lo + 1;
            if (rem < c) {
                a = a_lo + rem;
                break;
            }
            rem -= c;
        }
        const int s = r.dl - d;
        i64 first = ((a + 1) << s) - 1, last = ((a + 2) << s) - 2;
        if (first < r.x_lo) first = r.x_lo;
        if (last > x_hi) last = x_hi;
        const double *c0 = chg + r.i_lo + (first - r.x_lo);
        const i64 m = last - first + 1;
        const i64 pa = tr.phys(a);
        double v = tr.T[pa];
        // additions strictly in list order (the reference's propagate order).  The upper levels' owners add ALL n changes: one dependent fp64 add per change
        // is the floor of the whole call, so the loads must not sit on that chain -- blocks of 16 changes, the next block requested before the current one is added
        // (the scheduling barriers keep hipcc from sinking the loads to their uses: 8 loads then 8 dependent adds per trip cost 2.6x the adds alone)
        i64 k = 0;
        if (m >= 32) {
            double ca[16], cb[16];
#pragma unroll
            for (int u = 0; u < 16; u++) ca[u] = c0[u];
            for (; k + 32 <= m; k += 32) {
#pragma unroll
                for (int u = 0; u < 16; u++) cb[u] = c0[k + 16 + u];
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int u = 0; u < 16; u++) v += ca[u];
                __builtin_amdgcn_sched_barrier(0);
                if (k + 48 <= m) {
#pragma unroll
                    for (int u = 0; u < 16; u++) ca[u] = c0[k + 32 + u];
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int u = 0; u < 16; u++) v += cb[u];
                __builtin_amdgcn_sched_barrier(0);
            }
            if (k + 16 <= m) {  // (ca holds block k)
#pragma unroll
                for (int u = 0; u < 16; u++) v += ca[u];
                k += 16;
            }
        }
        for (; k < m; k++) v += c0[k];
        tr.T[pa] = v;
    }
}

struct AddArgs {
    Tree tree;
    i64 cap;
    PerState *state;
    i64 n;
    const void *prio;
    int kind;
    double eps, alpha;
    double *chg;     // scratch, n doubles
    i64 start_slot;  // -1: append at state->write (add); >= 0: rewrite these ring slots in place (set_range)
    int commit;      // advance write/size (add) or not (set_range)
    int track_max;   // raise max_priority like update() does (:176-177)
    double maxp_snapshot;  // bulk path only: filled from *maxp_dev by the leaf kernel's caller
    const double *maxp_dev;
    i64 *bump0, *bump1;  // srlx_per_set_add_counters: int64 device counters an appending add advances by one (NULL: none)
};

__device__ __forceinline__ i64 add_start(const AddArgs &a) { return a.start_slot >= 0 ? a.start_slot : a.state->write; }

__device__ __forceinline__ void add_leaf(const AddArgs &a, i64 i, i64 write, double maxp) {
    i64 slot = write + i;
    if (slot >= a.cap) slot -= a.cap;
    const i64 x = slot + a.cap - 1;
    const double p = load_prio(a.prio, a.kind, i, a.eps, a.alpha, maxp);
    const i64 px = a.tree.phys(x);
    a.chg[i] = p - a.tree.T[px];
    a.tree.T[px] = p;
    a.tree.mirror(x, p);
    // priorities are >= 0, so their bit patterns order like the values
    if (a.track_max && p > maxp) atomicMax((u64 *)&a.state->max_priority, (u64)__double_as_longlong(p));
}

__device__ __forceinline__ void add_commit(const AddArgs &a) {
    if (!a.commit) return;
    PerState *s = a.state;
    i64 w = s->write + a.n;
    if (w >= a.cap) w -= a.cap;
    s->write = w;
    i64 z = s->size + a.n;
    s->size = z > a.cap ? a.cap : z;
    if (a.bump0) *a.bump0 += 1;  // readers of these counters ran in earlier launches of the stream
    if (a.bump1) *a.bump1 += 1;
}

// n <= kSmallAddMax: everything in one launch; the per-leaf changes live in LDS so that the root
// owner's n dependent fp64 additions are fed from LDS, not from memory
__global__ void __launch_bounds__(512) k_add_wg(AddArgs a) {
    extern __shared__ __attribute__((aligned(16))) double s_chg[];  // n doubles
    a.chg = s_chg;
    const int t = threadIdx.x, T = blockDim.x;
    const i64 write = add_start(a);
    const double maxp = a.state->max_priority;
    __syncthreads();  // every thread has read max_priority before anyone raises it
    for (i64 i = t; i < a.n; i += T) add_leaf(a, i, write, maxp);
    __syncthreads();
    for (int r = 0; r < 4; r++) {
        Run run;
        if (!get_run(a.cap, write, a.n, r, run)) break;
        run_ancestors(a.tree, a.chg, run, t, T);
        __syncthreads();
    }
    if (t == 0) add_commit(a);
}

// ---- n <= kTinyAddMax appending adds by ONE wave (round 6: the b1 shim's host loop adds one item between two samples, srl/rl/memories/priority_replay_buffer.py
// :205-217; the 256-thread machinery above costs ~10 us for that).  The adds are applied one after the other like the reference's loop (proportional_memory.py:120-129):
// leaf write, then lane d adds the change to the ancestor on depth d -- a node is always touched by the same lane, in add order: the fp64 sums of k_add_wg, bit for bit.
constexpr int kTinyAddMax = 16;
__device__ __forceinline__ void add_tiny_body(const AddArgs &a, int lane) {
    const i64 write = add_start(a);
    const double maxp = a.state->max_priority;
    double seen_max = maxp;
    for (i64 i = 0; i < a.n; i++) {
        i64 slot = write + i;
        if (slot >= a.cap) slot -= a.cap;
        const i64 x = slot + a.cap - 1;
        const double p = load_prio(a.prio, a.kind, i, a.eps, a.alpha, maxp);
        const i64 px = a.tree.phys(x);
        const double change = p - a.tree.T[px];  // (every lane reads the leaf before lane 0 overwrites it: the wave runs in lock-step and the load is waited for below)
        const int dx = node_depth(x);
        __builtin_amdgcn_s_waitcnt(0);
        if (lane == 0) {
            a.tree.T[px] = p;
            a.tree.mirror(x, p);
        }
        if (lane < dx) {
            const i64 node = ((x + 1) >> (dx - lane)) - 1;  // the ancestor on depth `lane`
            const i64 pn = a.tree.phys(node);
            a.tree.T[pn] = a.tree.T[pn] + change;
        }
        if (p > seen_max) seen_max = p;
    }
    if (lane == 0) {
        if (a.track_max && seen_max > maxp) a.state->max_priority = seen_max;
        add_commit(a);
    }
}
__global__ void __launch_bounds__(64) k_add_tiny(AddArgs a) { add_tiny_body(a, threadIdx.x); }

// The shim's `sample` with the adds queued since the last observation of the tree INSIDE the launch (values by value in the kernel arguments: no staging at all),
// and a completion flag in host-visible memory for the caller to spin on (a stream synchronisation costs the host more than the kernel runs).
struct TinyVals {
    double v[kTinyAddMax];
};
__global__ void __launch_bounds__(kWgSample) k_add_sample_wg(AddArgs add, TinyVals vals, SampleArgs a, unsigned long long *done_flag, unsigned long long ticket) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    __shared__ double s_vals[kTinyAddMax];
    if (add.n > 0) {
        if (threadIdx.x < kTinyAddMax) s_vals[threadIdx.x] = vals.v[threadIdx.x];
        __syncthreads();
        add.prio = s_vals;
        if (threadIdx.x < 64) add_tiny_body(add, threadIdx.x);
        __threadfence();
        __syncthreads();  // the tree the draw walks includes the adds
    }
    (void)sample_wg_body(a, smem);
    if (done_flag) {  // results first (they may live in host memory), then the flag
        __threadfence_system();
        __syncthreads();
        if (threadIdx.x == 0) __hip_atomic_store(done_flag, ticket, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

// bulk: leaf pass, then one ancestor launch per run (empty runs exit), then commit
__global__ void __launch_bounds__(256) k_add_leaf_bulk(AddArgs a) {
    const i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < a.n) add_leaf(a, i, add_start(a), *a.maxp_dev);
}
__global__ void __launch_bounds__(256) k_add_anc_bulk(AddArgs a, int r) {
    Run run;
    if (!get_run(a.cap, add_start(a), a.n, r, run)) return;
    run_ancestors(a.tree, a.chg, run, (i64)blockIdx.x * blockDim.x + threadIdx.x, (i64)gridDim.x * blockDim.x);
}
__global__ void k_add_commit(AddArgs a) { add_commit(a); }
// max_priority as it was before this call (every add with priority=None uses that value, and
// track_max compares against it); copied aside so the leaf kernel's atomics cannot feed back
__global__ void k_snapshot_max(const PerState *s, double *out) { *out = s->max_priority; }

__global__ void k_state_init(PerState *s) {
    s->max_priority = 1.0;
    s->size = 0;
    s->write = 0;
    s->pad = 0;
}
__global__ void k_state_set(PerState *s, double mp, i64 size, i64 write) {
    s->max_priority = mp;
    s->size = size;
    s->write = write;
}
// blocked layout <-> the reference's flat heap order (backup()/restore(), proportional_memory.py:179-205)
__global__ void __launch_bounds__(256) k_to_heap(Tree tr, double *heap) {
    for (i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x; i < tr.len; i += (i64)gridDim.x * blockDim.x) heap[i] = tr.get(i);
}
__global__ void __launch_bounds__(256) k_from_heap(Tree tr, const double *heap) {
    for (i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x; i < tr.len; i += (i64)gridDim.x * blockDim.x) tr.set(i, heap[i]);
}

}  // namespace

// ==========================================================================================
// host side
// ==========================================================================================
struct srlx_per {
    int device;
    i64 capacity, tree_len;
    double alpha, beta_initial, beta_steps, epsilon;
    int has_duplicate;
    Tree tree;        // blocked device layout (tree.T is the allocation)
    i64 n_blocks;     // 128-byte blocks allocated
    int lds_blocks;   // leading blocks the bulk sampler stages in LDS
    int lds_blocks_big;  // ... by its one-workgroup-per-CU configuration
    int n_cu;
    u64 *d_ctl;       // control words of the bulk sampler (kCtl*: epochs, two counter sets, compaction tickets)
    u64 *d_tiles;     // look-back state of the bulk compaction (zero between calls)
    i64 tiles_cap;
    PerState *d_state;
    int *d_err;
    i64 *d_update_counter;  // BORROWED (srlx_per_set_update_counter) or NULL
    i64 *d_add_counter[2];  // BORROWED (srlx_per_set_add_counters) or NULL
    i64 size, write;  // host mirror
    srlx::Arena scratch;  // device
    srlx::Arena staging;  // device copies of host-mode arguments / results
    srlx::Arena pinned;   // pinned host
    // on_device = 2 (round 6: the b1 shim's asynchronous host mode): a ring of device-visible pinned slots; add / update copy their host arguments into a slot and
    // launch kernels that read it over the link -- no staging copy, no stream synchronisation -- and sample reads its uniforms from and writes its results to one
    // (ONE synchronisation per call).  ring_ev[k] is recorded behind the launch that reads slot k; a slot is reused only once its event has completed.
    static constexpr int kRingSlots = 16;
    static constexpr size_t kRingSlotBytes = 16 * 1024;
    char *ring;
    hipEvent_t ring_ev[kRingSlots];
    bool ring_busy[kRingSlots];
    int ring_next;
};

namespace {

hipStream_t pick_stream(srlx_per *, void *stream) { return (hipStream_t)stream; }  // NULL = HIP's default stream

size_t prio_elem_bytes(int kind) { return (kind == SRLX_PRIO_F32 || kind == SRLX_PRIO_EST_F32) ? 4 : (kind == SRLX_PRIO_NONE_MASKED ? 1 : 8); }

// a free slot of the asynchronous host mode's pinned ring (NULL: the arguments do not fit one slot -- the caller takes the synchronous path)
int ring_acquire(srlx_per *h, size_t bytes, char **out, int *slot) {
    *out = nullptr;
    if (bytes > srlx_per::kRingSlotBytes) return SRLX_OK;
    if (!h->ring) {
        SRLX_HIP(hipHostMalloc((void **)&h->ring, srlx_per::kRingSlots * srlx_per::kRingSlotBytes, hipHostMallocDefault));
        for (auto &e : h->ring_ev) SRLX_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    }
    const int k = h->ring_next;
    h->ring_next = (k + 1) % srlx_per::kRingSlots;
    if (h->ring_busy[k]) SRLX_HIP(hipEventSynchronize(h->ring_ev[k]));
    h->ring_busy[k] = false;
    *out = h->ring + (size_t)k * srlx_per::kRingSlotBytes;
    *slot = k;
    return SRLX_OK;
}
int ring_release(srlx_per *h, int slot, hipStream_t st) {
    SRLX_HIP(hipEventRecord(h->ring_ev[slot], st));
    h->ring_busy[slot] = true;
    return SRLX_OK;
}

int launch_add(srlx_per *h, i64 n, const void *d_prio, int kind, hipStream_t st, i64 start_slot = -1) {
    SRLX_TRY(h->scratch.reserve(srlx::Carver::padded((size_t)n * 8) + 256));
    const bool append = start_slot < 0;
    double *snap = (double *)((char *)h->scratch.ptr + srlx::Carver::padded((size_t)n * 8));
    AddArgs a{h->tree, h->capacity, h->d_state, n, d_prio, kind, h->epsilon, h->alpha, (double *)h->scratch.ptr,
              start_slot, append ? 1 : 0, append ? 0 : 1, 0.0, snap, append ? h->d_add_counter[0] : nullptr, append ? h->d_add_counter[1] : nullptr};
    if (n <= kTinyAddMax && append) {
        hipLaunchKernelGGL(k_add_tiny, dim3(1), dim3(64), 0, st, a);
    } else if (n <= kSmallAddMax) {
        hipLaunchKernelGGL(k_add_wg, dim3(1), dim3(n < 512 ? kWgAdd : 512), (size_t)n * sizeof(double), st, a);  // (512 threads: a 256-register budget keeps the pipelined chain blocks out of scratch)
    } else {
        const int blocks = (int)((n + 255) / 256);
        hipLaunchKernelGGL(k_snapshot_max, dim3(1), dim3(1), 0, st, h->d_state, snap);
        hipLaunchKernelGGL(k_add_leaf_bulk, dim3(blocks), dim3(256), 0, st, a);
        // ~2n ancestor nodes; the root owner walks all n changes, so more threads do not help it
        i64 anc_threads = 2 * n + 64;
        int anc_blocks = (int)((anc_threads + 255) / 256);
        if (anc_blocks > 2048) anc_blocks = 2048;
        for (int r = 0; r < 4; r++) hipLaunchKernelGGL(k_add_anc_bulk, dim3(anc_blocks), dim3(256), 0, st, a, r);
        hipLaunchKernelGGL(k_add_commit, dim3(1), dim3(1), 0, st, a);
    }
    SRLX_HIP(hipGetLastError());
    if (append) {  // host mirror
        h->write = (h->write + n) % h->capacity;
        h->size = (h->size + n > h->capacity) ? h->capacity : h->size + n;
    }
    return SRLX_OK;
}

int launch_update(srlx_per *h, i64 n, const i64 *d_idx, const void *d_prio, int kind, hipStream_t st) {
    const size_t eb = prio_elem_bytes(kind);
    for (i64 off = 0; off < n; off += kUpdateChunk) {
        const i64 m = (n - off < kUpdateChunk) ? n - off : kUpdateChunk;
        // one thread per (index, ancestor) task where that fits: the B = 32 learner call is 640 tasks, one round at 1024 threads
        const int threads = (m <= 64 || m * (i64)h->tree.D > kWgUpdate) ? 1024 : kWgUpdate;  // (m <= 64: the kernel's wave form wants its 16 waves)
        const size_t lds = (size_t)m * (8 + 8 + 8 + 4 + 4) + (size_t)threads * 8 + 16;
        hipLaunchKernelGGL(k_update_wg, dim3(1), dim3(threads), lds, st, h->tree, h->d_state, m,
                           d_idx + off, (const void *)((const char *)d_prio + (size_t)off * eb), kind, h->epsilon,
                           h->alpha, h->d_err, off + m >= n ? h->d_update_counter : nullptr);
    }
    SRLX_HIP(hipGetLastError());
    return SRLX_OK;
}

// scratch layout for sample
struct SampleScratch {
    static size_t bytes(i64 M, i64 B, bool bulk) {
        using C = srlx::Carver;
        if (!bulk) return C::padded((size_t)M * 8) * 2 + C::padded((size_t)B * 8) * 2;
        return C::padded((size_t)B * 8) + C::padded((size_t)M * 8) * 2;  // compacted weights | leaf per draw | weight per draw
    }
};

int launch_sample(srlx_per *h, i64 B, i64 step, const i64 *d_step, const double *d_u, i64 M, i64 *d_idx, double *d_w,
                  float *d_w32, i64 *d_used, hipStream_t st, u64 key_seed = 0, i64 *key_counter = nullptr) {
    const bool bulk = M > kSmallSampleMax;
    // "d_step == key_counter" (the call's own draw number is the beta step; sample_wg_body) exists on the one-workgroup path only: the bulk sampler reads *d_step
    // in later launches, when the counter has already moved -- refuse instead of annealing beta a step off, depending on launch order
    SRLX_REQUIRE(!(bulk && d_step && (const void *)d_step == (const void *)key_counter), "per_sample: d_step == the draw counter needs the one-workgroup sampler (at most 8192 uniforms)");
    SRLX_TRY(h->scratch.reserve(SampleScratch::bytes(M, B, bulk)));
    srlx::Carver cv(h->scratch.ptr);
    SampleArgs a{};
    a.tr = h->tree;
    a.state = h->d_state;
    a.beta_initial = h->beta_initial;
    a.beta_steps = h->beta_steps;
    a.step = step;
    a.d_step = d_step;
    a.has_duplicate = h->has_duplicate;
    a.uniforms = d_u;
    a.key_seed = key_seed;
    a.key_counter = key_counter;
    a.n_uniforms = M;
    a.batch = B;
    a.out_idx = d_idx;
    a.out_w = d_w;
    a.out_w32 = d_w32;
    a.out_used = d_used;

    if (!bulk) {
        a.cand_idx = cv.take<i64>(M);
        a.cand_p = cv.take<double>(M);
        a.map = cv.take<i64>(B);
        a.wtmp = cv.take<double>(B);
        const size_t lds = sample_wg_lds(M);
        hipLaunchKernelGGL(k_sample_wg, dim3(1), dim3(kWgSample), lds, st, a);
    } else {
        if (!h->has_duplicate) {
            srlx::set_error("per_sample: has_duplicate=False is limited to %lld uniforms per call", (long long)kSmallSampleMax);
            return SRLX_ERR_UNSUPPORTED;
        }
        a.wtmp = cv.take<double>(B);
        a.cand_idx = cv.take<i64>(M);
        a.cand_p = cv.take<double>(M);  // un-normalised IS weight of the draw (0 = zero-priority leaf, rejected)
        u64 *ctl = h->d_ctl;
        const i64 ntiles = (M + kTile - 1) / kTile;
        if (ntiles > h->tiles_cap) {  // look-back state of the compaction, zero between calls
            if (h->d_tiles) SRLX_HIP(hipFree(h->d_tiles));
            h->d_tiles = nullptr;
            h->tiles_cap = 0;
            const i64 cap = ntiles < 1024 ? 1024 : 2 * ntiles;
            SRLX_HIP(hipMalloc((void **)&h->d_tiles, (size_t)cap * 8));
            SRLX_HIP(hipMemset(h->d_tiles, 0, (size_t)cap * 8));
            h->tiles_cap = cap;
        }
        const bool big = M >= kBulkBigMin;
        const int threads = big ? 1024 : 256;
        const int top_blocks = big ? h->lds_blocks_big : h->lds_blocks;
        const int resident = big ? h->n_cu : 4 * h->n_cu;
        const i64 want = (M + threads * kIlp - 1) / (threads * kIlp);
        const int blocks = (int)(want < resident ? want : resident);
        const size_t lds = (size_t)top_blocks * 64 + (size_t)threads * 8;
        const bool all_staged = h->n_blocks <= top_blocks;
        if (big && all_staged)
            hipLaunchKernelGGL((k_descend_bulk<1024, true>), dim3(blocks), dim3(threads), lds, st, a, top_blocks, ctl);
        else if (big)
            hipLaunchKernelGGL((k_descend_bulk<1024, false>), dim3(blocks), dim3(threads), lds, st, a, top_blocks, ctl);
        else if (all_staged)
            hipLaunchKernelGGL((k_descend_bulk<256, true>), dim3(blocks), dim3(threads), lds, st, a, top_blocks, ctl);
        else
            hipLaunchKernelGGL((k_descend_bulk<256, false>), dim3(blocks), dim3(threads), lds, st, a, top_blocks, ctl);
        // one launch finishes either path: the normalising pass when nothing was rejected, else the ordered compaction
        const i64 fw = (B + 256 * 4 - 1) / (256 * 4);
        const i64 cw = 8 * (i64)h->n_cu, need = ntiles > fw ? ntiles : fw;
        hipLaunchKernelGGL(k_compact_bulk, dim3((unsigned)(need < cw ? need : cw)), dim3(kTileThreads), 0, st, a, ctl, h->d_tiles, ntiles, 2 * h->n_cu);
    }
    SRLX_HIP(hipGetLastError());
    return SRLX_OK;
}

}  // namespace

extern "C" {

int srlx_per_create(srlx_per_t **out, int64_t capacity, double alpha, double beta_initial, double beta_steps,
                    int has_duplicate, double epsilon, int device) {
    SRLX_REQUIRE(out != nullptr, "per_create: out is NULL");
    SRLX_REQUIRE(capacity > 0 && capacity <= ((i64)1 << 30), "per_create: capacity %lld out of range", (long long)capacity);
    SRLX_REQUIRE(beta_steps != 0.0, "per_create: beta_steps must be non-zero");
    int ndev = 0;
    SRLX_HIP(hipGetDeviceCount(&ndev));
    SRLX_REQUIRE(device >= 0 && device < ndev, "per_create: device %d not present (%d devices)", device, ndev);
    srlx::DeviceGuard guard(device);
    srlx_per *h = new (std::nothrow) srlx_per();
    if (!h) return SRLX_ERR_NOMEM;
    h->device = device;
    h->capacity = capacity;
    h->tree_len = 2 * capacity - 1;
    h->alpha = alpha;
    h->beta_initial = beta_initial;
    h->beta_steps = beta_steps;
    h->epsilon = epsilon;
    h->has_duplicate = has_duplicate ? 1 : 0;
    h->pinned.pinned_host = true;
    h->size = h->write = 0;
    h->d_state = nullptr;
    h->d_err = nullptr;
    h->d_ctl = nullptr;
    h->d_tiles = nullptr;
    h->tiles_cap = 0;
    {   // geometry of the blocked layout (see struct Tree)
        Tree &t = h->tree;
        t.T = nullptr;
        t.len = h->tree_len;
        t.D = node_depth(h->tree_len - 1);
        for (int g = 0; g < 12; g++) t.base[g] = 0;
        if (t.D == 0) {
            t.h0 = 0;
            t.G = 0;
            h->n_blocks = 1;
        } else {
            t.h0 = (t.D - 1) % 3 + 1;
            t.G = (t.D - t.h0) / 3;
            t.base[1] = 1;
            for (int g = 1; g <= t.G; g++) t.base[g + 1] = t.base[g] + ((i64)1 << (t.h0 + 3 * (g - 1)));
            h->n_blocks = t.G >= 1 ? t.base[t.G + 1] : 1;
        }
        h->lds_blocks = 1;
        h->lds_blocks_big = 1;
        for (int g = 1; g <= t.G; g++) {
            if (t.base[g + 1] <= kBulkTopSmall) h->lds_blocks = (int)t.base[g + 1];
            if (t.base[g + 1] <= kBulkTopBig) h->lds_blocks_big = (int)t.base[g + 1];
        }
        if (h->n_blocks < h->lds_blocks) h->lds_blocks = (int)h->n_blocks;
        if (h->n_blocks < h->lds_blocks_big) h->lds_blocks_big = (int)h->n_blocks;
        hipDeviceProp_t prop;
        SRLX_HIP(hipGetDeviceProperties(&prop, device));
        h->n_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
        SRLX_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k_descend_bulk<1024, false>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                     kBulkTopBig * 64 + 1024 * 8));
        SRLX_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k_descend_bulk<1024, true>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                     kBulkTopBig * 64 + 1024 * 8));
    }
    hipError_t e = hipMalloc((void **)&h->tree.T, 128 * (size_t)h->n_blocks);
    if (e == hipSuccess) e = hipMalloc((void **)&h->d_state, sizeof(PerState));
    if (e == hipSuccess) e = hipMalloc((void **)&h->d_err, sizeof(int));
    if (e == hipSuccess) e = hipMalloc((void **)&h->d_ctl, kCtlWords * 8);
    if (e == hipSuccess) e = hipMemset(h->d_ctl, 0, kCtlWords * 8);
    if (e != hipSuccess) {
        srlx::set_error("per_create: %s", hipGetErrorString(e));
        srlx_per_destroy(h);
        return e == hipErrorOutOfMemory ? SRLX_ERR_NOMEM : SRLX_ERR_HIP;
    }
    *out = h;
    int s = srlx_per_clear(h, nullptr);
    if (s == SRLX_OK && hipStreamSynchronize(nullptr) != hipSuccess) s = SRLX_ERR_HIP;
    if (s != SRLX_OK) {
        srlx_per_destroy(h);
        *out = nullptr;
    }
    return s;
}

int srlx_per_set_has_duplicate(srlx_per_t *h, int has_duplicate) {
    SRLX_REQUIRE(h, "per_set_has_duplicate: NULL handle");
    h->has_duplicate = has_duplicate ? 1 : 0;
    return SRLX_OK;
}

int srlx_per_set_update_counter(srlx_per_t *h, int64_t *d_counter) {
    SRLX_REQUIRE(h, "per_set_update_counter: NULL handle");
    h->d_update_counter = d_counter;
    return SRLX_OK;
}

int srlx_per_set_add_counters(srlx_per_t *h, int64_t *d_counter0, int64_t *d_counter1) {
    SRLX_REQUIRE(h, "per_set_add_counters: NULL handle");
    h->d_add_counter[0] = d_counter0;
    h->d_add_counter[1] = d_counter1;
    return SRLX_OK;
}

int srlx_per_destroy(srlx_per_t *h) {
    if (!h) return SRLX_OK;
    srlx::DeviceGuard guard(h->device);
    (void)hipDeviceSynchronize();
    if (h->tree.T) (void)hipFree(h->tree.T);
    if (h->d_state) (void)hipFree(h->d_state);
    if (h->d_err) (void)hipFree(h->d_err);
    if (h->d_ctl) (void)hipFree(h->d_ctl);
    if (h->d_tiles) (void)hipFree(h->d_tiles);
    h->scratch.release();
    h->staging.release();
    h->pinned.release();
    if (h->ring) {
        (void)hipHostFree(h->ring);
        for (auto &e : h->ring_ev) (void)hipEventDestroy(e);
    }
    delete h;
    return SRLX_OK;
}

int srlx_per_clear(srlx_per_t *h, void *stream) {
    SRLX_REQUIRE(h, "per_clear: NULL handle");
    srlx::DeviceGuard guard(h->device);
    hipStream_t st = pick_stream(h, stream);
    SRLX_HIP(hipMemsetAsync(h->tree.T, 0, 128 * (size_t)h->n_blocks, st));
    SRLX_HIP(hipMemsetAsync(h->d_err, 0, sizeof(int), st));
    hipLaunchKernelGGL(k_state_init, dim3(1), dim3(1), 0, st, h->d_state);
    SRLX_HIP(hipGetLastError());
    h->size = h->write = 0;
    return SRLX_OK;
}

int64_t srlx_per_length(const srlx_per_t *h) { return h ? h->size : -1; }
int64_t srlx_per_capacity(const srlx_per_t *h) { return h ? h->capacity : -1; }

int srlx_per_add(srlx_per_t *h, int64_t n, const void *prio, int prio_kind, int on_device, void *stream) {
    SRLX_REQUIRE(h, "per_add: NULL handle");
    SRLX_REQUIRE(n >= 0 && n <= h->capacity, "per_add: n=%lld must be in [0, capacity=%lld]", (long long)n,
                 (long long)h->capacity);
    SRLX_REQUIRE(prio_kind >= SRLX_PRIO_NONE && prio_kind <= SRLX_PRIO_EST_F32, "per_add: bad prio_kind %d", prio_kind);
    SRLX_REQUIRE(prio_kind == SRLX_PRIO_NONE || prio != nullptr, "per_add: prio is NULL");
    if (n == 0) return SRLX_OK;
    srlx::DeviceGuard guard(h->device);
    hipStream_t st = pick_stream(h, stream);
    if (on_device == 1 || prio_kind == SRLX_PRIO_NONE) return launch_add(h, n, prio, prio_kind, st);
    const size_t bytes = (size_t)n * prio_elem_bytes(prio_kind);
    if (on_device == 2) {  // host arrays, asynchronous: the kernel reads a device-visible pinned slot
        char *slot_ptr;
        int slot = 0;
        SRLX_TRY(ring_acquire(h, bytes, &slot_ptr, &slot));
        if (slot_ptr) {
            memcpy(slot_ptr, prio, bytes);
            SRLX_TRY(launch_add(h, n, slot_ptr, prio_kind, st));
            return ring_release(h, slot, st);
        }
    }
    SRLX_TRY(h->pinned.reserve(bytes));
    SRLX_TRY(h->staging.reserve(bytes));
    memcpy(h->pinned.ptr, prio, bytes);
    SRLX_HIP(hipMemcpyAsync(h->staging.ptr, h->pinned.ptr, bytes, hipMemcpyHostToDevice, st));
    SRLX_TRY(launch_add(h, n, h->staging.ptr, prio_kind, st));
    SRLX_HIP(hipStreamSynchronize(st));
    return SRLX_OK;
}

int srlx_per_set_range(srlx_per_t *h, int64_t first_slot, int64_t n, const void *prio, int prio_kind, int on_device,
                       void *stream) {
    SRLX_REQUIRE(h, "per_set_range: NULL handle");
    SRLX_REQUIRE(first_slot >= 0 && first_slot < h->capacity && n >= 0 && n <= h->capacity, "per_set_range: bad range");
    SRLX_REQUIRE(prio_kind >= SRLX_PRIO_F64 && prio_kind <= SRLX_PRIO_RAW && prio, "per_set_range: bad prio");
    if (n == 0) return SRLX_OK;
    srlx::DeviceGuard guard(h->device);
    hipStream_t st = pick_stream(h, stream);
    if (on_device) return launch_add(h, n, prio, prio_kind, st, first_slot);
    const size_t bytes = (size_t)n * prio_elem_bytes(prio_kind);
    SRLX_TRY(h->pinned.reserve(bytes));
    SRLX_TRY(h->staging.reserve(bytes));
    memcpy(h->pinned.ptr, prio, bytes);
    SRLX_HIP(hipMemcpyAsync(h->staging.ptr, h->pinned.ptr, bytes, hipMemcpyHostToDevice, st));
    SRLX_TRY(launch_add(h, n, h->staging.ptr, prio_kind, st, first_slot));
    SRLX_HIP(hipStreamSynchronize(st));
    return SRLX_OK;
}

int srlx_per_sample(srlx_per_t *h, int64_t batch_size, int64_t step, const int64_t *d_step, const double *uniforms,
                    int64_t n_uniforms, int64_t *out_idx, double *out_w, float *out_w32, int64_t *out_used,
                    int on_device, void *stream) {
    SRLX_REQUIRE(h, "per_sample: NULL handle");
    SRLX_REQUIRE(batch_size > 0, "per_sample: batch_size must be positive");
    SRLX_REQUIRE(uniforms && n_uniforms >= batch_size, "per_sample: need at least batch_size uniforms (%lld < %lld)",
                 (long long)n_uniforms, (long long)batch_size);
    SRLX_REQUIRE(out_idx && out_used, "per_sample: out_idx/out_used are NULL");
    srlx::DeviceGuard guard(h->device);
    hipStream_t st = pick_stream(h, stream);
    if (on_device == 1)
        return launch_sample(h, batch_size, step, d_step, uniforms, n_uniforms, out_idx, out_w, out_w32, out_used, st);

    SRLX_REQUIRE(d_step == nullptr, "per_sample: d_step needs on_device=1");
    using C = srlx::Carver;
    if (on_device == 2) {  // host arrays through ONE device-visible pinned slot: uniforms in, results out, one synchronisation, no copy commands
        const size_t need = C::padded((size_t)n_uniforms * 8) + C::padded((size_t)batch_size * 8) * 2 + C::padded((size_t)batch_size * 4) + C::padded(8);
        char *slot_ptr;
        int slot = 0;
        SRLX_TRY(ring_acquire(h, need, &slot_ptr, &slot));
        if (slot_ptr) {
            C sp(slot_ptr);
            double *m_u = sp.take<double>(n_uniforms);
            i64 *m_idx = sp.take<i64>(batch_size);
            double *m_w = sp.take<double>(batch_size);
            float *m_w32 = sp.take<float>(batch_size);
            i64 *m_used = sp.take<i64>(1);
            memcpy(m_u, uniforms, (size_t)n_uniforms * 8);
            SRLX_TRY(launch_sample(h, batch_size, step, nullptr, m_u, n_uniforms, m_idx, m_w, m_w32, m_used, st));
            SRLX_HIP(hipStreamSynchronize(st));
            *out_used = *m_used;
            if (*m_used < 0) {
                srlx::set_error("per_sample: %lld uniforms were not enough for %lld accepted draws", (long long)n_uniforms, (long long)batch_size);
                return SRLX_ERR_UNIFORMS_EXHAUSTED;
            }
            memcpy(out_idx, m_idx, (size_t)batch_size * 8);
            if (out_w) memcpy(out_w, m_w, (size_t)batch_size * 8);
            if (out_w32) memcpy(out_w32, m_w32, (size_t)batch_size * 4);
            return SRLX_OK;
        }
    }
    const size_t in_bytes = C::padded((size_t)n_uniforms * 8);
    const size_t out_bytes = C::padded((size_t)batch_size * 8) * 2 + C::padded((size_t)batch_size * 4) + C::padded(8);
    SRLX_TRY(h->pinned.reserve(in_bytes + out_bytes));
    SRLX_TRY(h->staging.reserve(in_bytes + out_bytes));
    C hp(h->pinned.ptr), dp(h->staging.ptr);
    double *h_u = hp.take<double>(n_uniforms);
    double *d_u = dp.take<double>(n_uniforms);
    i64 *h_idx = hp.take<i64>(batch_size), *d_idx = dp.take<i64>(batch_size);
    double *h_w = hp.take<double>(batch_size), *d_w = dp.take<double>(batch_size);
    float *h_w32 = hp.take<float>(batch_size), *d_w32 = dp.take<float>(batch_size);
    i64 *h_used = hp.take<i64>(1), *d_used = dp.take<i64>(1);
    memcpy(h_u, uniforms, (size_t)n_uniforms * 8);
    SRLX_HIP(hipMemcpyAsync(d_u, h_u, (size_t)n_uniforms * 8, hipMemcpyHostToDevice, st));
    SRLX_TRY(launch_sample(h, batch_size, step, nullptr, d_u, n_uniforms, d_idx, d_w, d_w32, d_used, st));
    // results sit contiguously after the uniforms: one copy back
    SRLX_HIP(hipMemcpyAsync(h_idx, d_idx, out_bytes, hipMemcpyDeviceToHost, st));
    SRLX_HIP(hipStreamSynchronize(st));
    *out_used = *h_used;
    if (*h_used < 0) {
        srlx::set_error("per_sample: %lld uniforms were not enough for %lld accepted draws", (long long)n_uniforms,
                        (long long)batch_size);
        return SRLX_ERR_UNIFORMS_EXHAUSTED;
    }
    memcpy(out_idx, h_idx, (size_t)batch_size * 8);
    if (out_w) memcpy(out_w, h_w, (size_t)batch_size * 8);
    if (out_w32) memcpy(out_w32, h_w32, (size_t)batch_size * 4);
    return SRLX_OK;
}

// The b1 shim's `sample` (round 6): `n_add` <= 16 adds queued since the tree was last observed (HOST values: final leaf priorities, SRLX_PRIO_RAW, or NULL with
// SRLX_PRIO_NONE) and the draw as ONE launch -- the add values travel in the kernel arguments, the uniforms and the results through a device-visible pinned slot --
// and the host spins on a completion flag in that slot instead of synchronising the stream.  Results as srlx_per_sample(on_device = 0).
static int sample_after_adds_impl(srlx_per_t *h, int64_t n_add, const double *add_values, int add_kind, int64_t batch_size, int64_t step, const double *uniforms,
                                  const uint32_t *mt_words, int64_t n_uniforms, int64_t *out_idx, double *out_w, float *out_w32, int64_t *out_used, int64_t *out_slots,
                                  void *stream) {
    SRLX_REQUIRE(h, "per_sample_after_adds: NULL handle");
    SRLX_REQUIRE(n_add >= 0 && n_add <= kTinyAddMax && (n_add == 0 || add_kind == SRLX_PRIO_NONE || (add_kind == SRLX_PRIO_RAW && add_values)),
                 "per_sample_after_adds: at most %d adds, SRLX_PRIO_RAW values or SRLX_PRIO_NONE", kTinyAddMax);
    SRLX_REQUIRE(batch_size > 0 && (uniforms || mt_words) && n_uniforms >= batch_size && n_uniforms <= kSmallSampleMax && out_idx && out_used,
                 "per_sample_after_adds: batch_size <= n_uniforms <= %lld", (long long)kSmallSampleMax);
    srlx::DeviceGuard guard(h->device);
    hipStream_t st = pick_stream(h, stream);
    using C = srlx::Carver;
    const size_t need = C::padded((size_t)n_uniforms * 8) + C::padded((size_t)batch_size * 8) * 2 + C::padded((size_t)batch_size * 4) + C::padded(8) + C::padded(8);
    char *slot_ptr;
    int slot = 0;
    SRLX_TRY(ring_acquire(h, need, &slot_ptr, &slot));
    if (!slot_ptr) {  // does not fit a slot: the plain calls
        SRLX_REQUIRE(uniforms && !out_slots, "per_sample_after_adds_mt: the draw does not fit a pinned slot");
        if (n_add > 0) SRLX_TRY(srlx_per_add(h, n_add, add_values, add_kind, 0, stream));
        return srlx_per_sample(h, batch_size, step, nullptr, uniforms, n_uniforms, out_idx, out_w, out_w32, out_used, 0, stream);
    }
    C sp(slot_ptr);
    double *m_u = sp.take<double>(n_uniforms);
    i64 *m_idx = sp.take<i64>(batch_size);
    double *m_w = sp.take<double>(batch_size);
    float *m_w32 = sp.take<float>(batch_size);
    i64 *m_used = sp.take<i64>(1);
    volatile unsigned long long *m_flag = (volatile unsigned long long *)sp.take<unsigned long long>(1);
    if (uniforms)
        memcpy(m_u, uniforms, (size_t)n_uniforms * 8);
    else  // CPython's random.random() (Modules/_randommodule.c: random_random) on consecutive MT19937 outputs
        for (i64 j = 0; j < n_uniforms; j++) m_u[j] = ((double)(mt_words[2 * j] >> 5) * 67108864.0 + (double)(mt_words[2 * j + 1] >> 6)) * (1.0 / 9007199254740992.0);
    static unsigned long long ticket = 0;
    const unsigned long long want = ++ticket;
    *m_flag = 0;
    const i64 M = n_uniforms, B = batch_size;
    SRLX_TRY(h->scratch.reserve(SampleScratch::bytes(M, B, false)));
    srlx::Carver cv(h->scratch.ptr);
    SampleArgs a{};
    a.tr = h->tree, a.state = h->d_state, a.beta_initial = h->beta_initial, a.beta_steps = h->beta_steps, a.step = step, a.d_step = nullptr;
    a.has_duplicate = h->has_duplicate, a.uniforms = m_u, a.key_seed = 0, a.key_counter = nullptr, a.n_uniforms = M, a.batch = B;
    a.out_idx = m_idx, a.out_w = m_w, a.out_w32 = m_w32, a.out_used = m_used;
    a.cand_idx = cv.take<i64>(M), a.cand_p = cv.take<double>(M), a.map = cv.take<i64>(B), a.wtmp = cv.take<double>(B);
    AddArgs add{h->tree, h->capacity, h->d_state, n_add, nullptr, add_kind, h->epsilon, h->alpha, nullptr, -1, 1, 0, 0.0, nullptr, h->d_add_counter[0], h->d_add_counter[1]};
    TinyVals vals{};
    for (int k = 0; k < n_add && add_values; k++) vals.v[k] = add_values[k];
    const size_t lds = sample_wg_lds(M);
    hipLaunchKernelGGL(k_add_sample_wg, dim3(1), dim3(kWgSample), lds, st, add, vals, a, (unsigned long long *)m_flag, want);
    SRLX_HIP(hipGetLastError());
    if (n_add > 0) {  // host mirror
        h->write = (h->write + n_add) % h->capacity;
        h->size = (h->size + n_add > h->capacity) ? h->capacity : h->size + n_add;
    }
    // spin on the flag (the kernel's last store, system scope, behind its results); a stream synchronisation as the fallback after ~2 ms
    bool done = false;
    for (long spins = 0; spins < 2000000; spins++) {
        if (*m_flag == want) {
            done = true;
            break;
        }
        __builtin_ia32_pause();
    }
    if (!done) SRLX_HIP(hipStreamSynchronize(st));
    __atomic_thread_fence(__ATOMIC_ACQUIRE);
    *out_used = *m_used;
    if (*m_used < 0) {
        srlx::set_error("per_sample: %lld uniforms were not enough for %lld accepted draws", (long long)n_uniforms, (long long)batch_size);
        return SRLX_ERR_UNIFORMS_EXHAUSTED;
    }
    memcpy(out_idx, m_idx, (size_t)batch_size * 8);
    if (out_w) memcpy(out_w, m_w, (size_t)batch_size * 8);
    if (out_w32) memcpy(out_w32, m_w32, (size_t)batch_size * 4);
    if (out_slots)
        for (i64 j = 0; j < batch_size; j++) out_slots[j] = m_idx[j] - (h->capacity - 1);  // tree index -> data slot (leaf j <-> node j + N - 1)
    return SRLX_OK;
}

int srlx_per_sample_after_adds(srlx_per_t *h, int64_t n_add, const double *add_values, int add_kind, int64_t batch_size, int64_t step, const double *uniforms,
                               int64_t n_uniforms, int64_t *out_idx, double *out_w, float *out_w32, int64_t *out_used, void *stream) {
    SRLX_REQUIRE(uniforms, "per_sample_after_adds: NULL uniforms");
    return sample_after_adds_impl(h, n_add, add_values, add_kind, batch_size, step, uniforms, nullptr, n_uniforms, out_idx, out_w, out_w32, out_used, nullptr, stream);
}

int srlx_per_sample_after_adds_mt(srlx_per_t *h, int64_t n_add, const double *add_values, int add_kind, int64_t batch_size, int64_t step, const uint32_t *mt_words,
                                  int64_t n_uniforms, int64_t *out_idx, double *out_w, float *out_w32, int64_t *out_used, int64_t *out_slots, void *stream) {
    SRLX_REQUIRE(mt_words, "per_sample_after_adds_mt: NULL words");
    return sample_after_adds_impl(h, n_add, add_values, add_kind, batch_size, step, nullptr, mt_words, n_uniforms, out_idx, out_w, out_w32, out_used, out_slots, stream);
}

int srlx_per_sample_keyed(srlx_per_t *h, int64_t batch_size, const int64_t *d_step, uint64_t seed, int64_t *d_counter, int64_t n_uniforms,
                          int64_t *d_out_idx, double *d_out_w, float *d_out_w32, int64_t *d_out_used, void *stream) {
    SRLX_REQUIRE(h, "per_sample_keyed: NULL handle");
    SRLX_REQUIRE(batch_size > 0 && n_uniforms >= batch_size, "per_sample_keyed: need at least batch_size uniforms (%lld < %lld)", (long long)n_uniforms,
                 (long long)batch_size);
    SRLX_REQUIRE(n_uniforms <= kSmallSampleMax, "per_sample_keyed: at most %lld uniforms per call (the single-workgroup sampler)", (long long)kSmallSampleMax);
    SRLX_REQUIRE(d_step && d_counter && d_out_idx && d_out_used, "per_sample_keyed: NULL argument");
    srlx::DeviceGuard guard(h->device);
    return launch_sample(h, batch_size, 0, d_step, nullptr, n_uniforms, d_out_idx, d_out_w, d_out_w32, d_out_used, pick_stream(h, stream), seed, d_counter);
}

int srlx_per_sample_gather_train(srlx_per_t *h, srlx_store_t *store, int64_t batch_size, const int64_t *d_step, uint64_t seed, int64_t *d_counter, int64_t n_uniforms,
                                 int64_t *d_out_idx, float *d_out_w32, int64_t *d_out_used, int64_t *d_frame_off_all, int64_t *d_frame_off_next, int32_t *d_actions,
                                 float *d_rewards, float *d_terminated, void *stream) {
    SRLX_REQUIRE(h && store, "per_sample_gather_train: NULL handle");
    SRLX_REQUIRE(batch_size > 0 && batch_size <= srlxs::kTrainItems && n_uniforms >= batch_size && n_uniforms <= kSmallSampleMax,
                 "per_sample_gather_train: 1..%d items, batch_size..%lld uniforms", srlxs::kTrainItems, (long long)kSmallSampleMax);
    SRLX_REQUIRE(d_step && d_counter && d_out_idx && d_out_w32 && d_out_used && d_frame_off_all && d_actions && d_rewards && d_terminated, "per_sample_gather_train: NULL argument");
    srlx::DeviceGuard guard(h->device);
    hipStream_t st = pick_stream(h, stream);
    GatherArgs g{};
    SRLX_TRY(srlx_store_dev_view(store, batch_size, &g.store, &g.meta));
    SRLX_REQUIRE(g.store.obs_dtype == SRLX_OBS_U8, "per_sample_gather_train: uint8 stores only");
    SRLX_REQUIRE(h->capacity == g.store.E * g.store.item_len, "per_sample_gather_train: the tree has %lld leaves, the store %lld items (leaf j <-> environment j %% E, ring time j / E)",
                 (long long)h->capacity, (long long)(g.store.E * g.store.item_len));
    g.actions = d_actions, g.rewards = d_rewards, g.terminated = d_terminated, g.off_all = (i64 *)d_frame_off_all, g.off_next = (i64 *)d_frame_off_next;
    const i64 M = n_uniforms, B = batch_size;
    SRLX_TRY(h->scratch.reserve(SampleScratch::bytes(M, B, false)));
    srlx::Carver cv(h->scratch.ptr);
    SampleArgs a{};
    a.tr = h->tree, a.state = h->d_state, a.beta_initial = h->beta_initial, a.beta_steps = h->beta_steps, a.step = 0, a.d_step = d_step;
    a.has_duplicate = h->has_duplicate, a.uniforms = nullptr, a.key_seed = seed, a.key_counter = d_counter, a.n_uniforms = M, a.batch = B;
    a.out_idx = d_out_idx, a.out_w = nullptr, a.out_w32 = d_out_w32, a.out_used = d_out_used;
    a.cand_idx = cv.take<i64>(M), a.cand_p = cv.take<double>(M), a.map = cv.take<i64>(B), a.wtmp = cv.take<double>(B);
    const size_t lds = sample_wg_lds(M);
    hipLaunchKernelGGL(k_sample_gather_wg, dim3(1), dim3(kWgSample), lds, st, a, g);
    SRLX_HIP(hipGetLastError());
    return SRLX_OK;
}

int srlx_per_update(srlx_per_t *h, int64_t n, const int64_t *indices, const void *prio, int prio_kind, int on_device,
                    void *stream) {
    SRLX_REQUIRE(h, "per_update: NULL handle");
    SRLX_REQUIRE(n >= 0, "per_update: negative n");
    SRLX_REQUIRE(prio_kind >= SRLX_PRIO_F64 && prio_kind <= SRLX_PRIO_RAW, "per_update: bad prio_kind %d", prio_kind);
    if (n == 0) return SRLX_OK;
    SRLX_REQUIRE(indices && prio, "per_update: NULL array");
    srlx::DeviceGuard guard(h->device);
    hipStream_t st = pick_stream(h, stream);
    if (on_device == 1) return launch_update(h, n, indices, prio, prio_kind, st);

    for (i64 i = 0; i < n; i++)
        SRLX_REQUIRE(indices[i] >= 0 && indices[i] < h->tree_len, "per_update: index %lld out of range [0,%lld)",
                     (long long)indices[i], (long long)h->tree_len);
    using C = srlx::Carver;
    const size_t ib = C::padded((size_t)n * 8), pb = C::padded((size_t)n * prio_elem_bytes(prio_kind));
    if (on_device == 2) {  // host arrays, asynchronous (see srlx_per_add)
        char *slot_ptr;
        int slot = 0;
        SRLX_TRY(ring_acquire(h, ib + pb, &slot_ptr, &slot));
        if (slot_ptr) {
            memcpy(slot_ptr, indices, (size_t)n * 8);
            memcpy(slot_ptr + ib, prio, (size_t)n * prio_elem_bytes(prio_kind));
            SRLX_TRY(launch_update(h, n, (const i64 *)slot_ptr, slot_ptr + ib, prio_kind, st));
            return ring_release(h, slot, st);
        }
    }
    SRLX_TRY(h->pinned.reserve(ib + pb));
    SRLX_TRY(h->staging.reserve(ib + pb));
    C hp(h->pinned.ptr), dp(h->staging.ptr);
    i64 *h_i = hp.take<i64>(n), *d_i = dp.take<i64>(n);
    char *h_p = (char *)h->pinned.ptr + ib, *d_p = (char *)h->staging.ptr + ib;
    memcpy(h_i, indices, (size_t)n * 8);
    memcpy(h_p, prio, (size_t)n * prio_elem_bytes(prio_kind));
    SRLX_HIP(hipMemcpyAsync(d_i, h_i, ib + (size_t)n * prio_elem_bytes(prio_kind), hipMemcpyHostToDevice, st));
    SRLX_TRY(launch_update(h, n, d_i, d_p, prio_kind, st));
    SRLX_HIP(hipStreamSynchronize(st));
    return SRLX_OK;
}

int srlx_per_backup(srlx_per_t *h, double *max_priority, int64_t *size, int64_t *write, double *tree_host) {
    SRLX_REQUIRE(h && max_priority && size && write, "per_backup: NULL argument");
    srlx::DeviceGuard guard(h->device);
    SRLX_HIP(hipDeviceSynchronize());
    PerState s;
    SRLX_HIP(hipMemcpy(&s, h->d_state, sizeof(s), hipMemcpyDeviceToHost));
    *max_priority = s.max_priority;
    *size = s.size;
    *write = s.write;
    h->size = s.size;
    h->write = s.write;
    if (tree_host) {
        SRLX_TRY(h->staging.reserve(sizeof(double) * (size_t)h->tree_len));
        hipLaunchKernelGGL(k_to_heap, dim3(1024), dim3(256), 0, nullptr, h->tree, (double *)h->staging.ptr);
        SRLX_HIP(hipGetLastError());
        SRLX_HIP(hipMemcpy(tree_host, h->staging.ptr, sizeof(double) * (size_t)h->tree_len, hipMemcpyDeviceToHost));
    }
    int err = 0;
    SRLX_HIP(hipMemcpy(&err, h->d_err, sizeof(int), hipMemcpyDeviceToHost));
    if (err) {
        srlx::set_error("per: an earlier on-device update() received an out-of-range tree index");
        return SRLX_ERR_INVALID;
    }
    return SRLX_OK;
}

int srlx_per_restore(srlx_per_t *h, double max_priority, int64_t size, int64_t write, const double *tree_host) {
    SRLX_REQUIRE(h && tree_host, "per_restore: NULL argument");
    SRLX_REQUIRE(size >= 0 && size <= h->capacity && write >= 0 && write < h->capacity, "per_restore: bad size/write");
    srlx::DeviceGuard guard(h->device);
    SRLX_HIP(hipDeviceSynchronize());
    SRLX_TRY(h->staging.reserve(sizeof(double) * (size_t)h->tree_len));
    SRLX_HIP(hipMemcpy(h->staging.ptr, tree_host, sizeof(double) * (size_t)h->tree_len, hipMemcpyHostToDevice));
    SRLX_HIP(hipMemsetAsync(h->tree.T, 0, 128 * (size_t)h->n_blocks, nullptr));
    hipLaunchKernelGGL(k_from_heap, dim3(1024), dim3(256), 0, nullptr, h->tree, (const double *)h->staging.ptr);
    SRLX_HIP(hipGetLastError());
    hipLaunchKernelGGL(k_state_set, dim3(1), dim3(1), 0, nullptr, h->d_state, max_priority, (i64)size, (i64)write);
    SRLX_HIP(hipGetLastError());
    SRLX_HIP(hipStreamSynchronize(nullptr));
    h->size = size;
    h->write = write;
    return SRLX_OK;
}

int srlx_per_restore_resized(srlx_per_t *h, int64_t old_capacity, int64_t old_size, const double *old_tree_host) {
    SRLX_REQUIRE(h && old_tree_host, "per_restore_resized: NULL argument");
    SRLX_REQUIRE(old_capacity > 0 && old_size >= 0 && old_size <= old_capacity, "per_restore_resized: bad sizes");
    SRLX_TRY(srlx_per_clear(h, nullptr));
    const double *leaves = old_tree_host + (old_capacity - 1);
    for (i64 off = 0; off < old_size; off += h->capacity) {
        const i64 m = (old_size - off < h->capacity) ? old_size - off : h->capacity;
        SRLX_TRY(srlx_per_add(h, m, leaves + off, SRLX_PRIO_RAW, 0, nullptr));
    }
    SRLX_HIP(hipStreamSynchronize(nullptr));
    return SRLX_OK;
}

int srlx_per_tree_ptr(srlx_per_t *h, void **d_tree, int64_t *tree_len) {
    SRLX_REQUIRE(h && d_tree, "per_tree_ptr: NULL argument");
    *d_tree = h->tree.T;
    if (tree_len) *tree_len = h->n_blocks * 16;
    return SRLX_OK;
}

int srlx_per_max_priority(srlx_per_t *h, double *d_out, void *stream) {
    SRLX_REQUIRE(h && d_out, "per_max_priority: NULL argument");
    srlx::DeviceGuard guard(h->device);
    hipLaunchKernelGGL(k_snapshot_max, dim3(1), dim3(1), 0, pick_stream(h, stream), h->d_state, d_out);
    SRLX_HIP(hipGetLastError());
    return SRLX_OK;
}

int srlx_per_state_ptr(srlx_per_t *h, void **d_state) {
    SRLX_REQUIRE(h && d_state, "per_state_ptr: NULL argument");
    *d_state = h->d_state;
    return SRLX_OK;
}

int srlx_per_refresh(srlx_per_t *h, void *stream) {
    SRLX_REQUIRE(h, "per_refresh: NULL handle");
    srlx::DeviceGuard guard(h->device);
    hipStream_t st = pick_stream(h, stream);
    SRLX_TRY(h->pinned.reserve(sizeof(PerState)));
    SRLX_HIP(hipMemcpyAsync(h->pinned.ptr, h->d_state, sizeof(PerState), hipMemcpyDeviceToHost, st));
    SRLX_HIP(hipStreamSynchronize(st));
    const PerState *s = (const PerState *)h->pinned.ptr;
    h->size = s->size;
    h->write = s->write;
    return SRLX_OK;
}

}  // extern "C"
