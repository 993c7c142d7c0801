// srlx_qnet_int.h -- the srlx_qnet handle shared by the forward (srlx_qnet.hip) and backward (srlx_qnet_bwd.hip) kernels.
#pragma once
#include "srlx_common.h"

struct srlx_qnet {
    int device;
    int H, W, Wn, F1, hidden, A, dueling;
    int OH1, OW1, OH2, OW2, OH3, OW3;
    int64_t max_batch;
    int flat;  // OH3*OW3*2*F1
    const float *w1, *b1, *w2, *b2, *w3, *b3, *wf, *bf, *v2w, *v2b, *a2w, *a2b;  // BORROWED: the torch parameters themselves
    float *act1, *act2, *act3, *partial;
    int max_splits;
    // training (srlx_qnet_enable_training): post-ReLU hidden layer of every forward row + gradient scratch
    int64_t max_train;
    float *h1;                            // [max_batch][2*hidden]
    float *dh1, *dact3, *dact2, *dact1;   // [max_train][...]
    float *fc_part;                       // [kFcSplits][max_train][flat] (first dense layer's data gradient, batches above 32)
    float *dh1t;                          // [2*hidden][32]: dh1 transposed, rows of samples >= batch are zero (matrix-core data gradient, batch <= 32)
    float *w_part;                        // weight-gradient partial sums (largest layer)
    float *dxpad, *w_t, *w_t2;            // padded data gradient (per parity class), transposed filters of conv3 / conv2
    size_t w_part_floats;
    hipStream_t side;                     // weight-gradient branch of the backward pass (forks from / joins the caller's stream)
    hipEvent_t ev_fork, ev_d3, ev_d2, ev_d1, ev_join, ev_wt;
    // Adam applied to the first dense layer's weight inside its weight-gradient kernel (srlx_qnet_fuse_adam_fc1), which then
    // runs once the data gradient has read the weights
    float *adam_m, *adam_v;               // BORROWED optimiser state of wf (NULL: the gradient is written out instead)
    double adam_lr, adam_b1, adam_b2, adam_eps;
    const int64_t *adam_step;             // BORROWED device scalar: optimiser steps already taken
    // Adam for every OTHER tensor inside the launch that finishes its gradient (srlx_qnet_fuse_adam_rest; round 5: k_adam was a launch + a dependent-launch gap on the
    // update's tail): convolution weights and biases in k_reduce_parts' epilogue, the first dense layer's bias and the head's second layers in the small-vector range
    // of the packing launch (srlx_qnet_publish).  Indexed like the gradient list of srlx_qnet_backward_u8 (entry 6 = the first dense layer's weight: unused here).
    const float *rest_g[12];              // BORROWED gradient buffers (the ones every backward call is handed)
    float *rest_m[12], *rest_v[12];       // BORROWED optimiser state
    bool rest_on, rest_armed;             // armed: a backward pass has run and the packing launch that completes its optimiser step has not
    int64_t *step_snap;                   // owned device scalar: the step count as this update's Adam launches see it (the packing launch itself advances the count)
    hipEvent_t probe0, probe1;            // optional, caller-owned: recorded right around the convolution kernel launch(es) of the next forward (srlx_qnet_set_probe)
    hipEvent_t probe_fc0, probe_fc1;      // the same around the first dense layer's GEMM launch (srlx_qnet_set_probe_fc1)
    uint64_t *fc1_span;                   // caller-owned or NULL: the NEXT operand-planes first-dense-layer launch leaves min(first workgroup in), max(last workgroup out) of the device wall clock there (srlx_qnet_set_fc1_span)
    // NoisyLinear dense layers (srlx_qnet_bind_noisy, srlx_noisy.hip): wf..a2b above then point at `eff`, the effective tensors
    // mu + sigma * eps of the current noise draw; order of the six: wf, bf, v2w, v2b, a2w, a2b
    const float *mu[6], *sig[6];          // BORROWED torch parameters
    float *eff[6];                        // owned
    int64_t eff_n[6];
    unsigned long long noisy_seed;
    int64_t *d_draw;                      // device: [0] id of the next draw, [1] id of the draw `eff` holds
    float *g_sig[6];                      // BORROWED gradient tensors of the sigmas (srlx_qnet_bind_noisy_grads)
    int *range_flag;                      // device word: bit l set when an activation of layer l + 1 left float16's range in the two-part split (srlx_qnet_range_flags)
    void *fused_dbg;                      // optional device buffer [8 waves][8] of phase timestamps (srlx_qnet_set_debug; NULL in production)
    bool side_external;                   // h->side was handed in (srlx_qnet_set_side_stream): not ours to destroy
    bool wt_from_forward;                 // the last forward already built w_t / w_t2 (fused path of a training handle)
    float *wpack;                         // conv filters in MFMA-fragment order (srlx_qnet_fused.hip), rebuilt per forward
    // pre-split bf16 operand planes of the first dense layer (srlx_fc1_planes.hip; srlx_qnet_enable_fc1_planes): [row][K/8][3 parts][8 bf16]
    void *wf_planes;                      // [2*hidden][flat] as planes: valid after srlx_qnet_refresh_fc1_planes
    void *a3_planes;                      // [max_batch][flat] as planes: written by the convolution kernel's epilogue (or by a split pass)
    bool planes_valid;                    // wf_planes holds the weight the next forward should use
    bool a3_planes_fresh;                 // the convolution kernel of THIS forward wrote a3_planes itself
    bool want_planes_out;                 // this forward continues into the dense layers (set by forward_u8_impl; a convs-only call needs float32 act3)
    // ---- round 4: weights handed from the learner to the actors without a copy on the lock-step's serial tail ----
    // packed filters that stay valid between forwards: srlx_qnet_publish packs them right after the optimiser step (the NEXT forward of this handle skips
    // k_pack_filters); `pack_sticky`: a forward's own pack stays valid too (a target network: its weights change at a sync only).  srlx_qnet_weights_changed clears.
    bool pack_valid, pack_sticky;
    // an ACTOR handle's two published parameter sets (srlx_qnet_actor_sets_enable): everything a policy pass reads -- packed convolution filters, the first dense
    // layer as bf16 operand planes, biases and the head's vectors -- written by the learner's update (srlx_qnet_publish + the fused Adam's plane epilogue) into the
    // set the actors are NOT reading; srlx_qnet_actor_set_select flips which one the next forwards read.
    struct ActorSet {
        float *wpack;     // kPackFloats
        void *wf_planes;  // [flat/32][2 hidden][3 parts][4 k-groups][8 bf16]
        float *small;     // b1 | b2 | b3 | bf | v2w | v2b | a2w | a2b (each padded to 4 floats)
    } aset[2];
    int aset_cur;                         // -1: the bound parameters; 0 / 1: forwards read this set
    float *wpack_own;                     // this handle's own packed-filter buffer while a set is selected
    void *wf_planes_own;
    const float *bound[12];               // what srlx_qnet_bind bound (restored by srlx_qnet_actor_set_select(h, -1))
    bool planes_small;                    // operand planes also for launches below 512 rows (a learner's 96 / 128-row passes: srlx_qnet_set_planes_small)
    const void *wf_planes_ext;            // BORROWED weight planes for the next forwards (an actor set's: the learner's online network reads what it published)
    int fc1_neighbour;                    // > 0: chip-filling first-dense-layer launches use k_fc1_planes_h (half-CU workgroups) with this many K splits
    size_t partial_floats;                // allocation of `partial`
    float *c1_gpart;                      // conv1 weight gradient: group partial sums [Wn][4][32 x 64 + 32] of the in-launch reduction
    unsigned *c1_cnt;                     // ... and its arrival tickets [Wn][5] (zero between launches)
    // round 4 (second half): the replay's priority write-back rides on the weight-gradient branch (srlx_qnet_set_priority_sink), and the first dense layer's
    // weight gradient can take a branch of its own (side2; srlx_qnet_set_fc1_branch)
    struct srlx_per *sink_per;            // NULL: no sink
    const int64_t *sink_idx;
    const void *sink_prio;
    int64_t sink_n;
    int sink_kind;
    bool partial_used;                    // a forward has used `partial` (srlx_qnet_set_fc1_neighbour may no longer move it)
    hipEvent_t sink_done;                 // caller-owned or NULL: recorded on the sink's branch right behind the write-back (srlx_qnet_set_sink_done)
    hipEvent_t sink_wait;                 // caller-owned or NULL: the sink's branch waits for it first (srlx_qnet_set_sink_wait)
    bool main_first;                      // srlx_qnet_set_main_first
    int dgrad_split;                      // srlx_qnet_set_dgrad_split: K splits of conv3's data-gradient GEMM (0 / 1: none, 2)
    hipStream_t sink_stream;              // caller-owned or NULL: the write-back runs there (behind the fork point) instead of first on the weight-gradient branch (srlx_qnet_set_sink_stream)
    int fc1_order;                        // 0 (default) / 1 / 2: srlx_qnet_set_fc1_branch
    hipStream_t side2;
    hipEvent_t ev_join2;
    uint64_t *stamp_buf;                  // measurement aid (srlx_qnet_set_stamp_buffer): srlx_debug_stamp launches at fixed points of the backward pass
    hipEvent_t ev_td;                     // caller-owned or NULL: recorded right behind the head kernel of every backward pass (srlx_qnet_set_td_event)
    void *adam_planes_out;                // the fused Adam of the first dense layer ALSO writes the updated weight as operand planes here (NULL: off)
    // epsilon-greedy fused into the head kernel of the NEXT forward (srlx_qnet_forward_u8_policy)
    struct Policy {
        const float *eps;
        unsigned long long seed;
        const int64_t *counter;
        const unsigned char *invalid;
        int32_t *actions;
        float *q_copy;
    } pol;
    // ---- round 6: Agent57(_light)'s networks on this handle (srl/algorithms/agent57_light/model_torch.py:18-117) ----
    // UVFA inputs of a Q-network (:35-64: previous extrinsic / intrinsic reward, one-hot previous action, one-hot actor, concatenated behind the image features): their
    // columns of the first dense layer are kept apart as wx [X][2 hidden] (column-major: unit u of column c at c * 2 hidden + u) and enter the layer as rank-1 terms
    // in the head kernel -- scalar * column for the rewards, one column gather for each one-hot -- instead of K columns of the GEMM; no concatenated input exists.
    struct Uvfa {
        const float *wx;                  // what the next forwards read: the bound tensor, or a selected actor set's copy
        const float *wx_bound;            // BORROWED master (srlx_qnet_bind_uvfa)
        int X, c_ext, c_int, c_act, n_act_in, c_actor, n_actor;  // columns; c_* = first column of the input or -1 (absent)
        const float *r_ext, *r_int;       // per-row inputs of the next forwards / backward passes (srlx_qnet_set_uvfa_inputs; device, BORROWED)
        const int32_t *action, *actor;
        float *g_wx, *m_wx, *v_wx;        // gradient buffer and optimiser state (srlx_qnet_fuse_adam_uvfa): the step rides on the packing launch
    } uvfa;
    // per-sample discount and signed TD error for the fused TD prologue of srlx_qnet_backward_td_u8 (srlx_qnet_set_td_extras)
    const float *td_disc_ps;
    float *td_signed;
    // head_mode 1 (srlx_qnet_set_head_mode): the handle ends behind the first dense layer -- q / grad_q of the forward / backward entry points are the post-ReLU
    // hidden layer's first out_cols units [rows][out_cols] (the embedding and RND networks, model_torch.py:70-117); ln_w != NULL: a LayerNorm over all 2 hidden
    // units follows in the forward (inference handles only)
    int head_mode, out_cols;
    const float *ln_w, *ln_b;
    float ln_eps;
};

// what the head kernels need of srlx_qnet::Uvfa, by value
struct srlx_uvfa_dev {
    const float *wx, *r_ext, *r_int;
    const int32_t *action, *actor;
    int c_ext, c_int, c_act, c_actor, n_act_in, n_actor;
};
inline srlx_uvfa_dev srlx_uvfa_args(const srlx_qnet *h) {
    const srlx_qnet::Uvfa &u = h->uvfa;
    return srlx_uvfa_dev{u.wx, u.r_ext, u.r_int, u.action, u.actor, u.c_ext, u.c_int, u.c_act, u.c_actor, u.n_act_in, u.n_actor};
}

// offsets (floats) of the vectors inside ActorSet::small
struct srlx_small_layout {
    int b1, b2, b3, bf, v2w, v2b, a2w, a2b, wx, total;  // wx: the UVFA columns (round 6; empty without them)
};
inline srlx_small_layout srlx_small_offsets(const srlx_qnet *h) {
    auto pad = [](int n) { return (n + 3) & ~3; };
    srlx_small_layout L;
    L.b1 = 0;
    L.b2 = L.b1 + pad(h->F1);
    L.b3 = L.b2 + pad(2 * h->F1);
    L.bf = L.b3 + pad(2 * h->F1);
    L.v2w = L.bf + pad(2 * h->hidden);
    L.v2b = L.v2w + pad(h->hidden);
    L.a2w = L.v2b + 4;
    L.a2b = L.a2w + pad(h->A * h->hidden);
    L.wx = L.a2b + pad(h->A);
    L.total = L.wx + pad(h->uvfa.X * 2 * h->hidden);
    return L;
}

// srlx_noisy.hip: (re)materialise the effective dense-layer tensors with a fresh draw (no-op for a plain network)
int srlx_qnet_noisy_refresh(srlx_qnet *h, hipStream_t st);
// gradients of the sigmas from the gradients of the effective tensors (g[6..11] of srlx_qnet_backward_u8) and the draw `eff` holds
int srlx_qnet_noisy_sigma_grads(srlx_qnet *h, float *const *g, hipStream_t st);
// the dense layers (FC1 split-K + head) of srlx_qnet.hip over `rows` activation rows starting at act3 + first*flat, row stride `stride` rows
int srlx_qnet_dense_rows(srlx_qnet *h, int64_t rows, int64_t stride, float *d_q, hipStream_t st);

// srlx_qnet_fused.hip: conv1 -> conv2 -> conv3 in one kernel (activations in LDS); false when the geometry is not the Atari one
bool srlx_conv_h16();  // the convolutions' split: two float16 parts (default) or three bf16 parts (SRLX_CONV_BF16X3=1)
bool srlx_qnet_fused_convs(srlx_qnet *h, int64_t batch, const uint8_t *d_frame_base, const int64_t *d_frame_off, hipStream_t st);

int srlx_qnet_fused_convs_multi(srlx_qnet *const *hs, int n, int64_t batch, const uint8_t *d_frame_base, const int64_t *d_frame_off, hipStream_t st);

// srlx_fc1_planes.hip: the first dense layer of chip-filling launches as a conversion-free GEMM on pre-split bf16 operand planes
int srlx_fc1_planes_alloc(srlx_qnet *h);
int srlx_fc1_planes_split_weight(srlx_qnet *h, const float *src, float *copy_dst, hipStream_t st, void *planes_dst = nullptr);
size_t srlx_fc1_planes_weight_bytes(const srlx_qnet *h);
// srlx_qnet_fused.hip: pack `src`'s convolution filters (its own wpack + transposed filters when it trains) and, with `dst_set`, also into an actor set together
// with the small vectors
int srlx_qnet_pack_publish(srlx_qnet *src, srlx_qnet::ActorSet *dst_set, const srlx_small_layout *L, hipStream_t st, int64_t *bump = nullptr, bool adam_small = false);
size_t srlx_qnet_pack_bytes();
int srlx_fc1_planes_split_act(srlx_qnet *h, int64_t rows, hipStream_t st);
bool srlx_fc1_planes_applicable(const srlx_qnet *h, int64_t rows);
int srlx_fc1_planes_gemm(srlx_qnet *h, int64_t rows, int splits, int kps, hipStream_t st);

// implicit-GEMM data gradient on the matrix cores (defined next to k_gemm in srlx_qnet.hip)
int srlx_qnet_dgrad_gemm(const float *dY, int B, int QH, int QW, int OH, int OW, int CO, int KH, int KW, int S, const float *wT, int CI, float *dXq,
                         hipStream_t st, int ksplits = 1);
