// srlx_qnet_bwd.hip -- backward pass of the DQN-image + dueling Q-network for the learner's B = 32 gradient step.
//
// Replaces `loss.backward()` of srl/algorithms/rainbow/model_torch.py:107-108 (and the framework forward it needs)
// on the vectorised path: the forward of srlx_qnet.hip keeps its activations (NHWC act1..act3, the post-ReLU
// hidden layer), the fused TD kernel provides d loss / d Q, and the kernels below produce every parameter
// gradient directly in the memory layout of the torch parameters (conv2/conv3 channels_last, conv1 NCHW taps,
// the fused [2*hidden][flat] first dense layer), where the fused Adam reads them.
//
// At B = 32 the whole backward pass is 2.6 GFLOP against 64 MB of weight-sized traffic: every kernel here is
// HBM/latency bound, not MFMA bound (the two dense-layer kernels stream the 32 MB matrix once each).  The dense layer and
// conv1 use the matrix cores all the same -- with the batch (or the pixel pair) as the K dimension an MFMA needs one operand
// load per 2 x 32 x 32 FMAs where a plain FMA kernel needs one LDS broadcast per FMA.  Summation orders are fixed (no
// atomics): results are deterministic and agree with autograd to float32 round-off (tests/test_qnet_gpu.py, 1e-4 relative).
//
// Replicate padding: a border input pixel is read by several (output, tap) pairs; the data-gradient kernels gather
// over host-built pair tables instead of scattering.
#include <new>

#include "srlx_adam_math.h"
#include "srlx_qnet_int.h"
#include "srlx_td_math.h"

namespace {

using i64 = int64_t;
using u8 = unsigned char;
using f32x16 = __attribute__((ext_vector_type(16))) float;

constexpr int kFcSplits = 16;   // splits of the hidden dimension in the dense data-gradient
constexpr int kWgSplits = 64;   // partial tensors of a conv weight gradient: one per sample (max_train <= 64)
constexpr int kC1Pad = 88, kC1Frame = kC1Pad * kC1Pad, kC1Chunks = 4;  // conv1 weight gradient: pixel chunks per (sample, frame) the scratch is sized for; the launch uses c1_chunks() of them

__device__ __forceinline__ int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }
__device__ __forceinline__ float byte_to_unit(unsigned b) {  // u8 / 255, correctly rounded (see srlx_qnet.hip)
    const float x = (float)b, rcp = 1.0f / 255.0f;
    const float q = x * rcp;
    return fmaf(fmaf(-q, 255.0f, x), rcp, q);
}

// ---- dueling head + second dense layers (dueling_network.py:43-58) ------------------------------------------
// 64 hidden units per workgroup x 4 sample slices (thread = (unit, slice); slice s takes samples s, s+4, ...): the
// dependent chain over the batch is a quarter as long, the four partial sums are combined in slice order.
// dq [B][A] is dense, h1 rows at i*sample_stride.
template <int AMAX>
__global__ void __launch_bounds__(256) k_head_bwd(int B, i64 sstride, int hidden, int A, int dueling, const float *dq,
                                                   const float *__restrict__ h1, const float *__restrict__ v2w, const float *__restrict__ a2w,
                                                   float *__restrict__ dh1, float *__restrict__ dh1t /*[N1][32] or NULL*/, float *__restrict__ g_bf,
                                                   float *__restrict__ g_v2w, float *__restrict__ g_v2b, float *__restrict__ g_a2w, float *__restrict__ g_a2b,
                                                   srlx::TdArgs td, int with_td) {
    extern __shared__ __attribute__((aligned(16))) float sm[];  // dv[B], da[B][A], part[3 + AMAX][256] (+ with_td: dq[B][A], 256 doubles)
    float *dv = sm, *da = sm + B, *part = sm + B + B * A;
    if (with_td) {
        // srlx_qnet_backward_td_u8: d loss / d Q is not an input -- every workgroup evaluates the TD target / Huber gradient of the
        // B items itself (a few hundred flops per item, the same rows in the same order as k_nstep_td_huber_priority), workgroup 0 also
        // stores target, priorities, gradient seed and the loss: one launch less on the learner's chain
        float *sdq = part + (3 + AMAX) * 256;
        int off = B + 2 * B * A + (3 + AMAX) * 256;
        off += off & 1;
        double *red = reinterpret_cast<double *>(sm + off);
        const int t = threadIdx.x;
        const double la = srlx::td_rows(td, t, 256, sdq, blockIdx.x == 0);
        if (blockIdx.x == 0) {
            red[t] = la;
            __syncthreads();
            for (int s = 128; s > 0; s >>= 1) {
                if (t < s) red[t] += red[t + s];
                __syncthreads();
            }
            if (t == 0) td.loss[0] = (float)(red[0] / (double)td.B);
        }
        __syncthreads();
        dq = sdq;
    }
    for (int b = threadIdx.x; b < B; b += blockDim.x) {
        float s = 0.f;
        for (int j = 0; j < A; j++) s += dq[b * A + j];
        dv[b] = s;  // q_j = v + a_j - f(a): every q_j passes its gradient to v
        for (int j = 0; j < A; j++) da[b * A + j] = dueling == 0 ? dq[b * A + j] - s / (float)A : dq[b * A + j];
    }
    __syncthreads();
    const int ul = threadIdx.x & 63, slice = threadIdx.x >> 6;
    const int u = blockIdx.x * 64 + ul;
    const int N1 = 2 * hidden;
    float gv = 0.f, gbv = 0.f, gba = 0.f, ga[AMAX];
#pragma unroll
    for (int j = 0; j < AMAX; j++) ga[j] = 0.f;
    if (u < hidden) {
        const float wv = v2w[u];
        float wa[AMAX];
#pragma unroll
        for (int j = 0; j < AMAX; j++) wa[j] = j < A ? a2w[j * hidden + u] : 0.f;
#pragma unroll 8
        for (int b = slice; b < B; b += 4) {
            const float hv = h1[(i64)b * sstride * N1 + u], ha = h1[(i64)b * sstride * N1 + hidden + u];
            gv += dv[b] * hv;
            float s = 0.f;
#pragma unroll
            for (int j = 0; j < AMAX; j++)
                if (j < A) {
                    ga[j] += da[b * A + j] * ha;
                    s += da[b * A + j] * wa[j];
                }
            const float dhv = hv > 0.f ? dv[b] * wv : 0.f, dha = ha > 0.f ? s : 0.f;  // ReLU of the first dense layer
            dh1[(i64)b * N1 + u] = dhv;
            dh1[(i64)b * N1 + hidden + u] = dha;
            if (dh1t) dh1t[u * 32 + b] = dhv, dh1t[(hidden + u) * 32 + b] = dha;
            gbv += dhv;
            gba += dha;
        }
        if (dh1t)  // the matrix-core data gradient multiplies whole 32-sample tiles: absent samples contribute zeros
            for (int b = slice; b < 32; b += 4)
                if (b >= B) dh1t[u * 32 + b] = 0.f, dh1t[(hidden + u) * 32 + b] = 0.f;
    }
    part[0 * 256 + threadIdx.x] = gv;
    part[1 * 256 + threadIdx.x] = gbv;
    part[2 * 256 + threadIdx.x] = gba;
#pragma unroll
    for (int j = 0; j < AMAX; j++)
        if (j < A) part[(3 + j) * 256 + threadIdx.x] = ga[j];
    __syncthreads();
    if (slice == 0 && u < hidden) {
        auto sum4 = [&](int row) { return ((part[row * 256 + ul] + part[row * 256 + 64 + ul]) + part[row * 256 + 128 + ul]) + part[row * 256 + 192 + ul]; };
        g_v2w[u] = sum4(0);
        g_bf[u] = sum4(1);
        g_bf[hidden + u] = sum4(2);
#pragma unroll
        for (int j = 0; j < AMAX; j++)
            if (j < A) g_a2w[j * hidden + u] = sum4(3 + j);
    }
    if (blockIdx.x == 0 && threadIdx.x <= A) {
        float s = 0.f;
        if (threadIdx.x == A) {
            for (int b = 0; b < B; b++) s += dv[b];
            g_v2b[0] = s;
        } else {
            for (int b = 0; b < B; b++) s += da[b * A + threadIdx.x];
            g_a2b[threadIdx.x] = s;
        }
    }
}

// ---- UVFA columns of an Agent57(_light) Q-network (srlx_qnet_int.h): g_wx[c][u] = sum_b dh1[b][u] * x[b][c], x = (previous rewards, one-hot previous action,
// one-hot actor) of the rows that carry gradient (b * sstride of the forward); one thread per (column, unit), the batch in order ------------------------------
__global__ void __launch_bounds__(256) k_uvfa_wgrad(int B, i64 sstride, int N1, int X, srlx_uvfa_dev uv, const float *__restrict__ dh1, float *__restrict__ g_wx) {
    const int id = blockIdx.x * 256 + threadIdx.x;
    if (id >= X * N1) return;
    const int c = id / N1, u = id % N1;
    float s = 0.f;
    for (int b = 0; b < B; b++) {
        const i64 row = (i64)b * sstride;
        float x;
        if (c == uv.c_ext) x = uv.r_ext[row];
        else if (c == uv.c_int) x = uv.r_int[row];
        else if (uv.c_act >= 0 && c >= uv.c_act && c < uv.c_act + uv.n_act_in) x = uv.action[row] == c - uv.c_act ? 1.f : 0.f;
        else x = (uv.c_actor >= 0 && uv.actor[row] == c - uv.c_actor) ? 1.f : 0.f;
        s += dh1[(i64)b * N1 + u] * x;
    }
    g_wx[id] = s;
}

// ---- head_mode 1: the gradient arrives at the post-ReLU hidden layer's first out_cols units, g [B][out_cols] (the other units of a padded layer get none):
// dh1 = g masked by the ReLU, its transpose for the matrix-core data gradient, and the bias gradient (batch in slice order, as k_head_bwd) -------------------
__global__ void __launch_bounds__(256) k_hidden_bwd(int B, i64 sstride, int N1, int out_cols, const float *__restrict__ g, const float *__restrict__ h1,
                                                    float *__restrict__ dh1, float *__restrict__ dh1t, float *__restrict__ g_bf) {
    __shared__ float part[256];
    const int ul = threadIdx.x & 63, slice = threadIdx.x >> 6;
    const int u = blockIdx.x * 64 + ul;
    float gb = 0.f;
    if (u < N1) {
        for (int b = slice; b < B; b += 4) {
            const float hv = h1[(i64)b * sstride * N1 + u];
            const float d = (u < out_cols && hv > 0.f) ? g[(i64)b * out_cols + u] : 0.f;
            dh1[(i64)b * N1 + u] = d;
            if (dh1t) dh1t[u * 32 + b] = d;
            gb += d;
        }
        if (dh1t)
            for (int b = slice; b < 32; b += 4)
                if (b >= B) dh1t[u * 32 + b] = 0.f;
    }
    part[threadIdx.x] = gb;
    __syncthreads();
    if (slice == 0 && u < N1) g_bf[u] = ((part[ul] + part[64 + ul]) + part[128 + ul]) + part[192 + ul];
}

// ---- first dense layer: weight gradient  g_wf[n][k] = sum_b dh1[b][n] * act3[b][k]  on the matrix cores -------------------------
// One wave per 32 x 32 tile of the 32 MB matrix: the batch is the K dimension of v_mfma_f32_32x32x2_f32 (two samples per
// instruction), both operands come straight from global memory as coalesced 128-byte rows (dh1[b][n0 + i], act3[b][k0 + i]), no
// LDS.  ADAM = true (srlx_qnet_fuse_adam_fc1): the gradient is never written -- each lane applies the optimiser step to its
// 16 weights and their two moment estimates in place (their loads are issued before the MFMAs): 226 MB of traffic for
// "write g, then Adam" becomes 161 MB, and the update of 97 % of the parameters runs beside the convolution gradients instead
// of after them.  Both variants run the same instruction sequence, so the fused update is bit-equal to the separate one.
// PLANES (with ADAM): the updated weight is ALSO written as the two float16 part planes the actors' first dense layer multiplies (srlx_fc1_planes.hip, weight layout
// [K/32 slabs][N1 rows][2 parts][4 k-groups][8 f16]; three bf16 parts in rounds 4-5): a lane holds column k0 + i of sixteen rows, so part p of a row's 32 k is 64 contiguous bytes written by the
// 32 lanes of a half-wave, two bytes each -- the actors' private copy of 97 % of the parameters needs neither a copy nor a splitting pass on the lock-step's tail.
template <bool ADAM, bool PLANES = false>
__global__ void __launch_bounds__(256) k_fc1_wgrad(int B, i64 sstride, int N1, int K, const float *__restrict__ dh1, const float *__restrict__ act3,
                                                   float *__restrict__ g_wf, float *__restrict__ wf, float *__restrict__ m, float *__restrict__ v, double lr, double beta1,
                                                   double beta2, double eps, const i64 *__restrict__ d_step, _Float16 *__restrict__ planes = nullptr) {
    const int lane = threadIdx.x & 63, i = lane & 31, h = lane >> 5;
    const int k0 = (blockIdx.x * 4 + (threadIdx.x >> 6)) * 32, n0 = blockIdx.y * 32;
    if (k0 >= K) return;  // (wave-uniform; no barriers in this kernel)
    float pp[16], mm[16], vv[16];
    if (ADAM)
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const i64 at = (i64)(n0 + (r & 3) + 8 * (r >> 2) + 4 * h) * K + k0 + i;  // C/D layout: row = hidden unit, column = lane & 31
            pp[r] = wf[at], mm[r] = m[at], vv[r] = v[at];
        }
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; r++) acc[r] = 0.f;
    const float *pa = dh1 + (i64)h * N1 + n0 + i, *pb = act3 + (i64)h * sstride * K + k0 + i;
    const int steps = (B + 1) / 2;
    for (int s0 = 0; s0 < steps; s0 += 8) {
        float av[8], bv[8];
#pragma unroll
        for (int q = 0; q < 8; q++) {
            const bool live = 2 * (s0 + q) + h < B;
            av[q] = live ? pa[(i64)(s0 + q) * 2 * N1] : 0.f;
            bv[q] = live ? pb[(i64)(s0 + q) * 2 * sstride * K] : 0.f;
        }
#pragma unroll
        for (int q = 0; q < 8; q++) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[q], bv[q], acc, 0, 0, 0);
    }
    if (ADAM) {
        const srlx::AdamCoef c = srlx::adam_coef(lr, beta1, beta2, eps, *d_step);
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const i64 at = (i64)(n0 + (r & 3) + 8 * (r >> 2) + 4 * h) * K + k0 + i;
            srlx::adam_one(pp[r], acc[r], mm[r], vv[r], c);
            wf[at] = pp[r], m[at] = mm[r], v[at] = vv[r];
            if (PLANES) {
                _Float16 *row = planes + (((i64)(k0 >> 5) * N1 + n0 + (r & 3) + 8 * (r >> 2) + 4 * h) * 64) + i;  // 64 f16 per (slab, row): hi at + 0, lo at + 32
                const _Float16 hi = (_Float16)pp[r];
                row[0] = hi;
                row[32] = (_Float16)((pp[r] - (float)hi) * 2048.0f);
            }
        }
    } else {
#pragma unroll
        for (int r = 0; r < 16; r++) g_wf[(i64)(n0 + (r & 3) + 8 * (r >> 2) + 4 * h) * K + k0 + i] = acc[r];
    }
}

// ---- first dense layer: data gradient, split over the hidden dimension (reads the 32 MB matrix once) ---------
template <int BT>
__global__ void __launch_bounds__(256) k_fc1_dgrad(int B, int N1, int K, const float *__restrict__ dh1, const float *__restrict__ wf, float *__restrict__ part) {
    __shared__ float sd[BT * 64];
    const int per = N1 / kFcSplits;  // <= 64
    const int n0 = blockIdx.y * per;
    const int k = blockIdx.x * 256 + threadIdx.x;
    for (int idx = threadIdx.x; idx < BT * per; idx += 256) {
        const int b = idx / per, nn = idx % per;
        sd[b * 64 + nn] = b < B ? dh1[(i64)b * N1 + n0 + nn] : 0.f;
    }
    __syncthreads();
    if (k >= K) return;
    float acc[BT];
#pragma unroll
    for (int b = 0; b < BT; b++) acc[b] = 0.f;
#pragma unroll 8
    for (int nn = 0; nn < per; nn++) {
        const float w = wf[(i64)(n0 + nn) * K + k];
#pragma unroll
        for (int b = 0; b < BT; b++) acc[b] += sd[b * 64 + nn] * w;
    }
#pragma unroll
    for (int b = 0; b < BT; b++)
        if (b < B) part[((i64)blockIdx.y * B + b) * K + k] = acc[b];
}

// dact3[b][k] = act3[b][k] > 0 ? sum_s part[s][b][k] : 0   (ReLU of conv3)
__global__ void __launch_bounds__(256) k_fc1_dgrad_reduce(int B, i64 sstride, int K, const float *__restrict__ part, const float *__restrict__ act3,
                                                          float *__restrict__ dact3) {
    const unsigned i = blockIdx.x * 256u + threadIdx.x;  // B * K <= 64 * 7744: 32-bit index arithmetic
    if (i >= (unsigned)(B * K)) return;
    const int b = (int)(i / (unsigned)K), k = (int)(i % (unsigned)K);
    float s = 0.f;
#pragma unroll
    for (int sp = 0; sp < kFcSplits; sp++) s += part[((i64)sp * B + b) * K + k];
    dact3[i] = act3[(i64)b * sstride * K + k] > 0.f ? s : 0.f;
}

// ---- first dense layer, batch <= 32: data gradient on the matrix cores, no partial sums ------------------------
//   dact3[b][k] = [act3[b][k] > 0] * sum_n dh1[b][n] * wf[n][k]
// The batch IS the 32-row tile of v_mfma_f32_32x32x2_f32: every weight is used by exactly one MFMA, so the B operand is loaded
// straight from global memory (lane (i, h) reads wf[n + h][k0 + i]: two coalesced 128-byte rows per instruction) and the A
// operand from the transposed gradient dh1t[n + h][i] (256 contiguous bytes, L2 resident).  A workgroup owns one 32-column tile
// of k; its four waves split the hidden dimension and their 32 x 32 partial tiles are summed through LDS in wave order.
// (The VALU kernel below spends 32 LDS broadcasts per 32 FMAs and needs a second launch to add up its 16 splits.)
__global__ void __launch_bounds__(256) k_fc1_dgrad_mfma(int B, i64 sstride, int N1, int K, const float *__restrict__ dh1t, const float *__restrict__ wf,
                                                        const float *__restrict__ act3, float *__restrict__ dact3) {
    __shared__ float red[3][16][64];
    const int t = threadIdx.x, wave = t >> 6, lane = t & 63, i = lane & 31, h = lane >> 5;
    const int k0 = blockIdx.x * 32;
    const int per = N1 / 4;  // hidden units of this wave (a multiple of 16: hidden % 32 == 0)
    const float *pa = dh1t + ((i64)wave * per + h) * 32 + i;
    const float *pb = wf + ((i64)wave * per + h) * K + k0 + i;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; r++) acc[r] = 0.f;
    float a0[8], b0[8], a1[8], b1[8];
    auto fetch = [&](float *a, float *b, int s0) {  // MFMA steps s0 .. s0+7 (two hidden units per step)
#pragma unroll
        for (int q = 0; q < 8; q++) {
            a[q] = pa[(i64)(s0 + q) * 64];
            b[q] = pb[(i64)(s0 + q) * 2 * K];
        }
    };
    const int steps = per / 2;  // a multiple of 8
    fetch(a0, b0, 0);
    for (int s0 = 0; s0 < steps; s0 += 16) {
        if (s0 + 8 < steps) fetch(a1, b1, s0 + 8);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int q = 0; q < 8; q++) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[q], b0[q], acc, 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        if (s0 + 8 >= steps) break;
        if (s0 + 16 < steps) fetch(a0, b0, s0 + 16);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int q = 0; q < 8; q++) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[q], b1[q], acc, 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
    }
    if (wave > 0)
#pragma unroll
        for (int r = 0; r < 16; r++) red[wave - 1][r][lane] = acc[r];
    __syncthreads();
    if (wave == 0)
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const int b = (r & 3) + 8 * (r >> 2) + 4 * h;  // C/D layout: row = sample, column = lane & 31
            if (b < B) {
                const float sum = ((acc[r] + red[0][r][lane]) + red[1][r][lane]) + red[2][r][lane];
                dact3[(i64)b * K + k0 + i] = act3[(i64)b * sstride * K + k0 + i] > 0.f ? sum : 0.f;
            }
        }
}

// ---- convolution weight gradient (NHWC input X, NHWC output gradient dY already masked by its ReLU), on the matrix cores ----
//   part[b][co][tap][ci] = sum over the pixels m of sample b of dY[b, m][co] * X[b, clamp(oy*S + ky - P), clamp(ox*S + kx - P), ci]
// One wave per (tap, 32 output channels, 32 input channels, sample): the sample's output pixels are the K dimension of
// v_mfma_f32_32x32x2_f32 (two pixels per instruction); lane (i, h) loads dY[m + h][co0 + i] and X[pixel(m + h, tap)][ci0 + i] straight
// from global memory -- both 128-byte rows -- eight steps ahead of the MFMAs that consume them.  No LDS, no barriers: the
// workgroups fit next to the actors' 127 KB convolution workgroups, where the staged FMA kernel this replaces ran 5x slower
// inside the lock-step loop than alone.  CO = 64.
struct ConvGeo {
    int H, W, CI, OH, OW, CO, KH, KW, S, P;
};
template <int CI>
__global__ void __launch_bounds__(256) k_conv_wgrad_mfma(ConvGeo g, i64 sstride, const float *__restrict__ X, const float *__restrict__ dY, float *__restrict__ part,
                                                         float *__restrict__ bias_part /*[sample][CO]*/) {
    constexpr int NCI = CI / 32, CO = 64;
    const int lane = threadIdx.x & 63, i = lane & 31, h = lane >> 5;
    const int combo = blockIdx.x * 4 + (threadIdx.x >> 6), taps = g.KH * g.KW;
    if (combo >= taps * 2 * NCI) return;  // (wave-uniform)
    const int cit = combo % NCI, cot = (combo / NCI) & 1, tap = combo / (2 * NCI);
    const int ky = tap / g.KW, kx = tap % g.KW;
    const i64 b = blockIdx.y;
    const int per_img = g.OH * g.OW, steps = (per_img + 1) / 2;
    const float *pa = dY + b * per_img * CO + cot * 32 + i;
    const float *px = X + b * sstride * g.H * g.W * CI + cit * 32 + i;
    int m = h, oy = 0, ox = h;  // this lane's pixel of the next step to fetch (OW >= 2)
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; r++) acc[r] = 0.f;
    float bsum = 0.f;
    float a0[8], x0[8], a1[8], x1[8];
    auto fetch = [&](float *a, float *x) __attribute__((always_inline)) {
#pragma unroll
        for (int q = 0; q < 8; q++) {
            const bool live = m < per_img;
            const int iy = clampi(oy * g.S + ky - g.P, 0, g.H - 1), ix = clampi(ox * g.S + kx - g.P, 0, g.W - 1);
            a[q] = live ? pa[(i64)m * CO] : 0.f;
            x[q] = live ? px[((i64)iy * g.W + ix) * CI] : 0.f;
            m += 2, ox += 2;
            if (ox >= g.OW) ox -= g.OW, oy++;
        }
    };
    auto mfma8 = [&](const float *a, const float *x) __attribute__((always_inline)) {
#pragma unroll
        for (int q = 0; q < 8; q++) {
            bsum += a[q];
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[q], x[q], acc, 0, 0, 0);
        }
    };
    fetch(a0, x0);
    for (int s0 = 0; s0 < steps; s0 += 16) {
        if (s0 + 8 < steps) fetch(a1, x1);
        __builtin_amdgcn_sched_barrier(0);
        mfma8(a0, x0);
        __builtin_amdgcn_sched_barrier(0);
        if (s0 + 8 >= steps) break;
        if (s0 + 16 < steps) fetch(a0, x0);
        __builtin_amdgcn_sched_barrier(0);
        mfma8(a1, x1);
        __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int r = 0; r < 16; r++) {
        const int co = cot * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;  // C/D layout: row = output channel, column = input channel
        part[((b * CO + co) * taps + tap) * CI + cit * 32 + i] = acc[r];
    }
    if (tap == 0 && cit == 0) {  // bias gradient of this sample: the dY column sums the A operand already walked (even pixels + odd pixels)
        const float tot = bsum + __shfl_xor(bsum, 32);
        if (h == 0) bias_part[b * CO + cot * 32 + i] = tot;
    }
}

// out[i] = sum_p part[p][i] for a weight gradient (n entries) and, in the same launch, its bias gradient (nb entries).
// 256 outputs (64 lanes x float4) x 4 slices per workgroup: slice s adds its run of parts in order (the loads of a run are independent: up to 16 in
// flight), the slices are then added in order -- a fixed summation order.  (256 threads, not more: a 1024-thread workgroup waits
// for sixteen free wave slots on one CU, which beside the actors' convolution workgroups took 25 us per launch.)
struct AdamArgs {
    double lr, beta1, beta2, eps;
    const i64 *d_step;
};
// ADAM (srlx_qnet_fuse_adam_rest): the optimiser step of the weight / bias right behind the sum, in place -- the arithmetic and the operand order of k_adam
// (srlx_adam_math.h), so the fused update is bit-equal to the separate launch.  `snap` (one launch per update): the step count is copied for the packing launch.
template <bool ADAM>
__global__ void __launch_bounds__(256) k_reduce_parts(const float *__restrict__ part, int P, i64 n, float *__restrict__ out, const float *__restrict__ bpart, int nb,
                                                      float *__restrict__ bout, float *__restrict__ pw = nullptr, float *__restrict__ mw = nullptr,
                                                      float *__restrict__ vw = nullptr, float *__restrict__ pb = nullptr, float *__restrict__ mb = nullptr,
                                                      float *__restrict__ vb = nullptr, AdamArgs ad = AdamArgs{}, i64 *__restrict__ snap = nullptr) {
    __shared__ float4 sm[4][64];
    const int lane = threadIdx.x & 63, sl = threadIdx.x >> 6;
    i64 i = ((i64)blockIdx.x * 64 + lane) * 4;  // four consecutive outputs per lane (n and nb are multiples of 4)
    const bool live = i < n + nb;
    if (ADAM && snap && blockIdx.x == 0 && threadIdx.x == 0) *snap = *ad.d_step;
    if (i >= n) {
        i -= n;
        part = bpart;
        n = nb;
        out = bout;
        pw = pb, mw = mb, vw = vb;
    }
    float4 pp, mm, vv;
    i64 step = 0;
    if (ADAM && sl == 0 && live) {  // requested ahead of the partial sums: one memory round trip for the launch
        pp = *reinterpret_cast<const float4 *>(pw + i), mm = *reinterpret_cast<const float4 *>(mw + i), vv = *reinterpret_cast<const float4 *>(vw + i);
        step = *ad.d_step;
    }
    const int per = (P + 3) / 4, p_lo = sl * per, p_hi = p_lo + per < P ? p_lo + per : P;
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    if (live)
#pragma unroll 8
        for (int p = p_lo; p < p_hi; p++) {
            const float4 v = *reinterpret_cast<const float4 *>(part + (i64)p * n + i);
            s.x += v.x, s.y += v.y, s.z += v.z, s.w += v.w;
        }
    sm[sl][lane] = s;
    __syncthreads();
    if (sl == 0 && live) {
        const float4 a = sm[1][lane], b = sm[2][lane], c = sm[3][lane];
        const float4 g = make_float4(((s.x + a.x) + b.x) + c.x, ((s.y + a.y) + b.y) + c.y, ((s.z + a.z) + b.z) + c.z, ((s.w + a.w) + b.w) + c.w);
        *reinterpret_cast<float4 *>(out + i) = g;
        if (ADAM) {
            const srlx::AdamCoef cf = srlx::adam_coef(ad.lr, ad.beta1, ad.beta2, ad.eps, step);
            srlx::adam_one(pp.x, g.x, mm.x, vv.x, cf);
            srlx::adam_one(pp.y, g.y, mm.y, vv.y, cf);
            srlx::adam_one(pp.z, g.z, mm.z, vv.z, cf);
            srlx::adam_one(pp.w, g.w, mm.w, vv.w, cf);
            *reinterpret_cast<float4 *>(pw + i) = pp;
            *reinterpret_cast<float4 *>(mw + i) = mm;
            *reinterpret_cast<float4 *>(vw + i) = vv;
        }
    }
}

// ---- convolution data gradient through replicate padding ---------------------------------------------------
// Replicate padding is an explicit pad followed by a plain convolution, so the data gradient is the plain transposed
// convolution evaluated on the PADDED grid (an implicit GEMM on the matrix cores, srlx_qnet_dgrad_gemm: rows = padded
// pixels, K = taps x CO, N = CI, filters transposed to [ci][tap][co]) followed by folding the pad rows / columns back
// onto the border pixels they replicate, fused with the ReLU mask of the layer input.
// wT[cls][ci][(a * KWS + b') * CO + co] = W[co][(cy + S a) * KW + (cx + S b')][ci],  cls = cy * S + cx
__global__ void __launch_bounds__(256) k_transpose_filter(const float *__restrict__ W, int CO, int KH, int KW, int S, int CI, float *__restrict__ wT) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    const int taps = KH * KW;
    if (i >= CO * taps * CI) return;
    const int ci = i % CI, tap = (i / CI) % taps, co = i / (CI * taps);  // W[co][tap][ci]
    const int ky = tap / KW, kx = tap % KW, KWS = KW / S, Kc = (KH / S) * KWS * CO;
    const int cls = (ky % S) * S + (kx % S);
    wT[((i64)cls * CI + ci) * Kc + ((ky / S) * KWS + kx / S) * CO + co] = W[i];
}

// dX[b][iy][ix][ci] = [X > 0] * sum over the padded positions that replicate (iy, ix) of dXq[class][b][py / S][px / S][ci]
// one thread per four channels (float4 loads / stores; CI is a multiple of 32)
// nsplit > 1: dxq is nsplit split-K partial slabs (split_stride floats apart), added per position in split order
__global__ void __launch_bounds__(256) k_fold_pad(int B, i64 sstride, int H, int W, int CI, int P, int HP, int WP, int S, int QH, int QW,
                                                  const float *__restrict__ dxq, const float *__restrict__ X, float *__restrict__ dX, int nsplit = 1, i64 split_stride = 0) {
    // 32-bit index arithmetic: B <= 64 samples of at most 21 x 21 x 64 values (64-bit divisions cost more than the kernel's traffic)
    const unsigned i4 = blockIdx.x * 256u + threadIdx.x;
    const unsigned c4n = (unsigned)CI / 4u;
    if (i4 >= (unsigned)(B * H * W) * c4n) return;
    const unsigned uw = (unsigned)W, uh = (unsigned)H;
    const int ci = 4 * (int)(i4 % c4n), ix = (int)((i4 / c4n) % uw), iy = (int)((i4 / (c4n * uw)) % uh), b = (int)(i4 / (c4n * uw * uh));
    const int y0 = iy == 0 ? 0 : iy + P, y1 = iy == H - 1 ? HP - 1 : iy + P;
    const int x0 = ix == 0 ? 0 : ix + P, x1 = ix == W - 1 ? WP - 1 : ix + P;
    const i64 cls_stride = (i64)B * QH * QW * CI;
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int py = y0; py <= y1; py++)
        for (int px = x0; px <= x1; px++) {
            const float *src = dxq + ((py % S) * S + px % S) * cls_stride + (((i64)b * QH + py / S) * QW + px / S) * CI + ci;
            float4 v = *reinterpret_cast<const float4 *>(src);
            for (int sp = 1; sp < nsplit; sp++) {
                const float4 w = *reinterpret_cast<const float4 *>(src + sp * split_stride);
                v.x += w.x, v.y += w.y, v.z += w.z, v.w += w.w;
            }
            s.x += v.x, s.y += v.y, s.z += v.z, s.w += v.w;
        }
    const float4 x = *reinterpret_cast<const float4 *>(X + (((i64)b * sstride * H + iy) * W + ix) * CI + ci);
    *reinterpret_cast<float4 *>(dX + (i64)i4 * 4) = make_float4(x.x > 0.f ? s.x : 0.f, x.y > 0.f ? s.y : 0.f, x.z > 0.f ? s.z : 0.f, x.w > 0.f ? s.w : 0.f);
}

// ---- conv1 weight gradient straight from the uint8 ring, on the matrix cores ------------------------------------
//   part[b, chunk][co][c*64 + ky*8 + kx] = sum over the chunk's pixels p of dY1[b][p][co] * frame_c[4 oy + ky - 3][4 ox + kx - 3] / 255
// workgroup = (sample, frame c of the window, quarter of the output pixels): one padded 88 x 88 uint8 frame, the chunk's dY rows
// and a pixel -> window-origin table in LDS.  The product is a 32 (co) x 64 (taps of frame c) x pixels GEMM: wave (tt, ph)
// owns taps ky in [4 tt, 4 tt + 4) and the pixel pairs jj = ph mod 2; per MFMA a lane reads one dY value, one table entry and
// one byte.  The two pixel interleaves are added through LDS (ph order): a sample leaves 4 partial tensors.
__global__ void __launch_bounds__(256) k_conv1_wgrad_mfma(const u8 *__restrict__ base, const i64 *__restrict__ frame_off, i64 sstride, int Wn, int H, int W,
                                                          int OH, int OW, int per, const float *__restrict__ dY1, float *__restrict__ part,
                                                          float *__restrict__ bias_part, float *__restrict__ gpart, unsigned *__restrict__ tickets,
                                                          float *__restrict__ g_w1, float *__restrict__ g_b1, int nch, int in_launch) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    u8 *fr = smem;                                                           // [88][88]
    int *poff = reinterpret_cast<int *>(smem + kC1Frame);                    // [per]
    float *sy = reinterpret_cast<float *>(smem + kC1Frame + (size_t)per * 4);  // [per][32]
    float *red = sy + (size_t)per * 32;                                      // [2][16][64]
    const int t = threadIdx.x, wave = t >> 6, lane = t & 63, i = lane & 31, h = lane >> 5;
    const i64 b = blockIdx.x;
    const int c = blockIdx.y / nch, half = blockIdx.y % nch;  // (`half`: the pixel chunk)
    const int M = OH * OW, p_lo = half * per, cnt = (p_lo + per < M ? p_lo + per : M) - p_lo;
    {
        const i64 off = frame_off[b * sstride * Wn + c];
        unsigned *dst = reinterpret_cast<unsigned *>(fr);
        if (W == kC1Pad - 4) {  // the Atari geometry: one (unaligned) dword load per padded dword, the border bytes placed by v_perm_b32 (as k_convnet_fused stages its frames)
            constexpr int kCols = kC1Pad / 4;
            for (int idx = t; idx < kC1Pad * kCols; idx += 256) {
                unsigned o = 0u;
                if (off >= 0) {
                    const int r = idx / kCols, d = idx % kCols;
                    unsigned v;
                    __builtin_memcpy(&v, base + off + (i64)clampi(r - 3, 0, H - 1) * W + clampi(4 * d - 3, 0, W - 4), 4);
                    o = __builtin_amdgcn_perm(0u, v, d == 0 ? 0x00000000u : (d == kCols - 1 ? 0x03030201u : 0x03020100u));
                }
                dst[idx] = o;
            }
        } else
        for (int idx = t; idx < kC1Pad * (kC1Pad / 4); idx += 256) {
            unsigned o = 0u;
            if (off >= 0) {
                const int r = idx / (kC1Pad / 4), x = 4 * (idx % (kC1Pad / 4)) - 3;
                const u8 *row = base + off + (i64)clampi(r - 3, 0, H - 1) * W;
#pragma unroll
                for (int q = 0; q < 4; q++) o |= (unsigned)row[clampi(x + q, 0, W - 1)] << (8 * q);
            }
            dst[idx] = o;
        }
    }
    for (int idx = t; idx < per * 32; idx += 256) sy[idx] = idx < cnt * 32 ? dY1[((i64)b * M + p_lo) * 32 + idx] : 0.f;  // pad rows multiply as zeros
    for (int p = t; p < per; p += 256) {
        const int pp = p < cnt ? p_lo + p : p_lo;
        poff[p] = 4 * (pp / OW) * kC1Pad + 4 * (pp % OW);
    }
    __syncthreads();
    if (c == 0) {  // bias gradient of this (sample, half): column sums of its dY rows, eight slices added in order
        float bs = 0.f;
        for (int p = t >> 5; p < cnt; p += 8) bs += sy[p * 32 + i];
        red[t] = bs;
        __syncthreads();
        if (t < 32) {
            float tot = red[t];
            for (int q = 1; q < 8; q++) tot += red[q * 32 + t];
            bias_part[((i64)b * nch + half) * 32 + t] = tot;
        }
        __syncthreads();
    }
    const int tt = wave & 1, ph = wave >> 1;
    const int tap = (4 * tt + (i >> 3)) * kC1Pad + (i & 7);
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; r++) acc[r] = 0.f;
    const int pairs = per / 2;
    for (int jj = ph; jj < pairs; jj += 2) {
        const int p = 2 * jj + h;
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(sy[p * 32 + i], byte_to_unit(fr[poff[p] + tap]), acc, 0, 0, 0);
    }
    if (ph == 1)
#pragma unroll
        for (int r = 0; r < 16; r++) red[(tt * 16 + r) * 64 + lane] = acc[r];
    __syncthreads();
    if (ph == 0) {
        const int K = Wn * 64;
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const int co = (r & 3) + 8 * (r >> 2) + 4 * h;  // C/D layout: row = output channel, column = tap
            part[(((i64)b * nch + half) * 32 + co) * K + c * 64 + tt * 32 + i] = acc[r] + red[(tt * 16 + r) * 64 + lane];
        }
    }
    // ---- the reduction over (sample, pixel chunk) inside the launch (round 4: it was a launch of its own, k_reduce_parts -- 4 us of work that cost the update's
    // critical path 17 us beside the actors).  Fixed summation order, the one k_reduce_parts used: the samples in four groups, a group's parts added in (sample,
    // chunk) order by the LAST workgroup of the group to arrive (ticket per frame and group), the four group sums added in order by the last group to finish.
    // Publication (cdna_hip_programming.md, in-launch split-K reduction): plain stores -> every wave drains its stores (s_waitcnt vmcnt(0)) -> barrier -> ONE lane's
    // agent-scope RELEASE fence (+ the restated wait) -> relaxed ticket; the reducer: ticket -> one agent-scope ACQUIRE fence -> barrier -> plain loads.  (Round 4
    // tried write-through sc1 stores without a release: 16 us faster and stale partial sums in about one run of six -- the write-throughs are not what a later
    // buffer_wbl2 waits for; __threadfence(), release AND acquire in every workgroup, is what the first version used.)  The reducer: ticket -> ONE agent-scope acquire fence (this CU's L1
    // drops what it holds of other CUs' lines) -> barrier -> plain loads (cdna_hip_programming.md, in-launch split-K reduction).  Tickets rewind themselves.
    if (!in_launch) return;  // the partial sums are added up by a k_reduce_parts launch behind this one (conv_chain)
    const int B = gridDim.x, K = Wn * 64;
    const int gs = (B + 3) / 4, g = (int)b / gs, ng = (B + gs - 1) / gs;
    const int b_lo = g * gs, b_hi = b_lo + gs < B ? b_lo + gs : B;
    __shared__ int s_last;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (t == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        s_last = __hip_atomic_fetch_add(&tickets[c * 5 + g], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (unsigned)((b_hi - b_lo) * nch) - 1u;
    }
    __syncthreads();
    if (!s_last) return;
    if (t == 0) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    __syncthreads();
    float4 *gp = reinterpret_cast<float4 *>(gpart + ((i64)c * 4 + g) * (32 * 64 + 32));
    for (int q = t; q < 32 * 16; q += 256) {
        const int co = q >> 4, tap4 = (q & 15) * 4;
        float4 sum = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll 8
        for (int p = b_lo * nch; p < b_hi * nch; p++) {
            const float4 v = *reinterpret_cast<const float4 *>(part + ((i64)p * 32 + co) * K + c * 64 + tap4);
            sum.x += v.x, sum.y += v.y, sum.z += v.z, sum.w += v.w;
        }
        gp[q] = sum;
    }
    if (c == 0 && t < 32) {
        float bs = 0.f;
        for (int p = b_lo * nch; p < b_hi * nch; p++) bs += bias_part[(i64)p * 32 + t];
        reinterpret_cast<float *>(gp)[32 * 64 + t] = bs;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (t == 0) {
        __hip_atomic_store(&tickets[c * 5 + g], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        s_last = __hip_atomic_fetch_add(&tickets[c * 5 + 4], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (unsigned)ng - 1u;
    }
    __syncthreads();
    if (!s_last) return;
    if (t == 0) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    __syncthreads();
    const float *g0 = gpart + (i64)c * 4 * (32 * 64 + 32);
    for (int q = t; q < 32 * 16; q += 256) {
        const int co = q >> 4, tap4 = (q & 15) * 4;
        float4 sum = reinterpret_cast<const float4 *>(g0)[q];
        for (int k = 1; k < ng; k++) {
            const float4 v = reinterpret_cast<const float4 *>(g0 + (i64)k * (32 * 64 + 32))[q];
            sum.x += v.x, sum.y += v.y, sum.z += v.z, sum.w += v.w;
        }
        *reinterpret_cast<float4 *>(g_w1 + (i64)co * K + c * 64 + tap4) = sum;
    }
    if (c == 0 && t < 32) {
        float bs = g0[32 * 64 + t];
        for (int k = 1; k < ng; k++) bs += g0[(i64)k * (32 * 64 + 32) + 32 * 64 + t];
        g_b1[t] = bs;
    }
    if (t == 0) __hip_atomic_store(&tickets[c * 5 + 4], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

}  // namespace

extern "C" {

int srlx_qnet_enable_training(srlx_qnet_t *h, int64_t max_train_batch) {
    SRLX_REQUIRE(h, "qnet_enable_training: NULL handle");
    SRLX_REQUIRE(max_train_batch > 0 && max_train_batch <= 64 && max_train_batch <= h->max_batch, "qnet_enable_training: 1 <= max_train_batch <= 64");
    SRLX_REQUIRE(h->F1 == 32 && h->dueling != 1 && h->H == h->W && h->W % 4 == 0 && (2 * h->hidden) % kFcSplits == 0 &&
                     2 * h->hidden / kFcSplits <= 64 && 4 * (h->OH1 - 1) + 8 <= kC1Pad,
                 "qnet_enable_training: the backward kernels cover the DQN image block with 32 filters, square frames, hidden <= 512, dueling average / none");
    if (h->max_train >= max_train_batch) return SRLX_OK;
    SRLX_REQUIRE(h->max_train == 0, "qnet_enable_training: already enabled with a smaller batch");
    srlx::DeviceGuard guard(h->device);
    const int N1 = 2 * h->hidden;
    const size_t c3 = (size_t)2 * h->F1 * 9 * 2 * h->F1, c2 = (size_t)2 * h->F1 * 16 * h->F1, c1 = (size_t)max_train_batch * kC1Chunks * 32 * h->Wn * 64;
    const size_t wp = kWgSplits * (c3 > c2 ? c3 : c2) + c1;  // [conv2 / conv3 per-sample parts][conv1 (sample, pixel chunk) parts]
    h->w_part_floats = wp;
    struct {
        float **p;
        size_t n;
    } bufs[] = {{&h->h1, (size_t)h->max_batch * N1},
                {&h->dh1, (size_t)max_train_batch * N1},
                {&h->dh1t, (size_t)N1 * 32},
                {&h->dact3, (size_t)max_train_batch * h->flat},
                {&h->dact2, (size_t)max_train_batch * h->OH2 * h->OW2 * 2 * h->F1},
                {&h->dact1, (size_t)max_train_batch * h->OH1 * h->OW1 * h->F1},
                {&h->fc_part, (size_t)kFcSplits * max_train_batch * h->flat},
                {&h->dxpad, (size_t)max_train_batch * (4 * (size_t)((h->OH1 + 5) / 2) * ((h->OW1 + 5) / 2) * h->F1 > 2 * (size_t)(h->OH2 + 2) * (h->OW2 + 2) * 2 * h->F1
                                                             ? 4 * (size_t)((h->OH1 + 5) / 2) * ((h->OW1 + 5) / 2) * h->F1
                                                             : 2 * (size_t)(h->OH2 + 2) * (h->OW2 + 2) * 2 * h->F1) + 128 * 64},
                {&h->w_t, c3},
                {&h->w_t2, c2},
                {&h->w_part, wp + 64 + 2 * kWgSplits * 64 + 64 * kC1Chunks * 32}};  // + bias partials: [sample][64] conv2/conv3 (x2), [sample, chunk][32] conv1
    for (auto &b : bufs) {
        hipError_t e = hipMalloc((void **)b.p, b.n * sizeof(float));
        if (e != hipSuccess) {
            srlx::set_error("qnet_enable_training: %s", hipGetErrorString(e));
            return e == hipErrorOutOfMemory ? SRLX_ERR_NOMEM : SRLX_ERR_HIP;
        }
    }
    // the weight-gradient branch of the backward pass runs on its own stream, forked from / joined to the caller's
    // (one side stream, created WITHOUT a priority: a second side stream, or hipStreamCreateWithPriority at either end of the range,
    // changes which hardware queues the learner's branches land on -- measured: the update then no longer overlaps the actors' pass
    // at all (0.84 instead of 0.755 ms per lock-step); GPU_MAX_HW_QUEUES above the default 4 triples the update's own time)
    SRLX_HIP(hipMalloc((void **)&h->c1_gpart, (size_t)h->Wn * 4 * (32 * 64 + 32) * sizeof(float)));
    SRLX_HIP(hipMalloc((void **)&h->c1_cnt, (size_t)h->Wn * 5 * sizeof(unsigned)));
    SRLX_HIP(hipMemset(h->c1_cnt, 0, (size_t)h->Wn * 5 * sizeof(unsigned)));
    SRLX_HIP(hipStreamCreateWithFlags(&h->side, hipStreamNonBlocking));
    // (side2 -- srlx_qnet_set_fc1_branch(h, 2) -- is created on first use: one more stream in the process changes which hardware queues the others land on, and with two
    //  queues that decides whether the update overlaps the actors' pass at all: measured 0.61 against 0.50 ms per lock-step with an UNUSED extra stream)
    for (hipEvent_t *e : {&h->ev_fork, &h->ev_d3, &h->ev_d2, &h->ev_d1, &h->ev_join, &h->ev_wt, &h->ev_join2})
        SRLX_HIP(hipEventCreateWithFlags(e, hipEventDisableTiming));
    h->max_train = max_train_batch;
    return SRLX_OK;
}

int srlx_qnet_fuse_adam_fc1(srlx_qnet_t *h, float *d_exp_avg, float *d_exp_avg_sq, double lr, double beta1, double beta2, double eps, const int64_t *d_steps_taken) {
    SRLX_REQUIRE(h, "qnet_fuse_adam_fc1: NULL handle");
    SRLX_REQUIRE(h->max_train > 0, "qnet_fuse_adam_fc1: call srlx_qnet_enable_training first");
    if (!d_exp_avg) {  // back to writing the gradient out
        h->adam_m = h->adam_v = nullptr;
        h->adam_step = nullptr;
        return SRLX_OK;
    }
    SRLX_REQUIRE(!h->sig[0], "qnet_fuse_adam_fc1: NoisyLinear layers need the gradient of the effective weight (for mu AND sigma): not fusable");
    SRLX_REQUIRE(d_exp_avg_sq && d_steps_taken && lr > 0.0 && beta1 >= 0.0 && beta1 < 1.0 && beta2 >= 0.0 && beta2 < 1.0 && eps >= 0.0, "qnet_fuse_adam_fc1: bad argument");
    h->adam_m = d_exp_avg, h->adam_v = d_exp_avg_sq;
    h->adam_lr = lr, h->adam_b1 = beta1, h->adam_b2 = beta2, h->adam_eps = eps;
    h->adam_step = d_steps_taken;
    return SRLX_OK;
}

int srlx_qnet_fuse_adam_rest(srlx_qnet_t *h, const float *const *d_grads, float *const *d_exp_avg, float *const *d_exp_avg_sq) {
    SRLX_REQUIRE(h, "qnet_fuse_adam_rest: NULL handle");
    if (!d_grads) {
        h->rest_on = h->rest_armed = false;
        return SRLX_OK;
    }
    SRLX_REQUIRE(h->adam_m && h->adam_step, "qnet_fuse_adam_rest: call srlx_qnet_fuse_adam_fc1 first (its hyper-parameters and step count are used)");
    SRLX_REQUIRE(d_exp_avg && d_exp_avg_sq && !h->sig[0] && h->bound[0], "qnet_fuse_adam_rest: bad argument (NoisyLinear networks are not fusable; parameters must be bound)");
    for (int k = 0; k < 12; k++) {
        if (k == 6) continue;
        SRLX_REQUIRE(d_grads[k] && d_exp_avg[k] && d_exp_avg_sq[k], "qnet_fuse_adam_rest: tensor %d has a NULL buffer", k);
        h->rest_g[k] = d_grads[k], h->rest_m[k] = d_exp_avg[k], h->rest_v[k] = d_exp_avg_sq[k];
    }
    srlx::DeviceGuard guard(h->device);
    if (!h->step_snap) SRLX_HIP(hipMalloc((void **)&h->step_snap, sizeof(int64_t)));
    h->rest_on = true;
    h->rest_armed = false;
    return SRLX_OK;
}

// Two branches (fork/join with events; capturable into a HIP graph): the data-gradient chain, then conv1's weight gradient (which needs the end of
// it), stay on the caller's stream; the other weight gradients run on h->side as soon as the activation gradient each needs exists -- the first
// dense layer's with Adam in its epilogue when the optimiser state is bound.
#define SRLX_STAMP(idx, stream) \
    if (h->stamp_buf) SRLX_TRY(srlx_debug_stamp(h->stamp_buf, (idx), (stream)))

static int chain_prologue(srlx_qnet_t *h, hipStream_t st, int part = 3, bool with_uvfa = false, int uvfa_B = 0, i64 uvfa_ss = 1) {  // part 1: the fork point on `st`; part 2: the side branch's first launches; 3: both
    hipStream_t sd = h->side;
    if (part & 1) SRLX_HIP(hipEventRecord(h->ev_fork, st));
    if (!(part & 2)) return SRLX_OK;
    SRLX_HIP(hipStreamWaitEvent(sd, h->ev_fork, 0));
    // the replay's priority write-back (model_torch.py:113-114) needs the TD kernel's output only: first thing on the weight-gradient branch instead of the
    // last launch of the update (it was 9 us + a launch boundary at the very end of the learner's critical path) -- or, with a sink stream (a learner rank: the
    // write-back waits for the slab's 150 us ingest, and the weight gradients must not queue behind that wait), on the caller's stream, which joins it itself
    hipStream_t sk = h->sink_per && h->sink_stream ? h->sink_stream : sd;
    if (sk != sd) SRLX_HIP(hipStreamWaitEvent(sk, h->ev_fork, 0));
    if (h->sink_per && h->sink_wait) SRLX_HIP(hipStreamWaitEvent(sk, h->sink_wait, 0));
    if (h->sink_per) SRLX_TRY(srlx_per_update(h->sink_per, h->sink_n, h->sink_idx, h->sink_prio, h->sink_kind, 1, sk));
    if (h->sink_per && h->sink_done) SRLX_HIP(hipEventRecord(h->sink_done, sk));
    SRLX_STAMP(21, sd);
    if (h->uvfa.X > 0 && h->uvfa.g_wx && with_uvfa) {  // the UVFA columns' gradient needs dh1 only: beside the data-gradient chain
        const int N1 = 2 * h->hidden;
        hipLaunchKernelGGL(k_uvfa_wgrad, dim3((unsigned)((h->uvfa.X * N1 + 255) / 256)), dim3(256), 0, sd, uvfa_B, uvfa_ss, N1, h->uvfa.X, srlx_uvfa_args(h), h->dh1, h->uvfa.g_wx);
    }
    const int C2 = 2 * h->F1;
    // the transposed filters of the two data-gradient GEMMs depend on the weights only: the fused forward of a training handle has
    // built them already (k_pack_filters); otherwise they are built here, ahead of the chain that needs them
    if (!h->wt_from_forward) {
        hipLaunchKernelGGL(k_transpose_filter, dim3((unsigned)((C2 * 9 * C2 + 255) / 256)), dim3(256), 0, st, h->w3, C2, 3, 3, 1, C2, h->w_t);
        hipLaunchKernelGGL(k_transpose_filter, dim3((unsigned)((C2 * 16 * h->F1 + 255) / 256)), dim3(256), 0, st, h->w2, C2, 4, 4, 2, h->F1, h->w_t2);
    }
    return SRLX_OK;
}

// From h->dact3 (the gradient at conv3's output, ReLU mask applied, rows 0..B-1) to the six convolution gradients g[0..5]; `with_fc1`: the first dense
// layer's weight gradient rides on the side stream (the whole-network backward).
static int conv_chain(srlx_qnet_t *h, int B, i64 ss, const uint8_t *d_frame_base, const int64_t *d_frame_off, float *const *g, hipStream_t st, bool with_fc1) {
    const int N1 = 2 * h->hidden, K = h->flat, C2 = 2 * h->F1;
    float *g_w1 = g[0], *g_b1 = g[1], *g_w2 = g[2], *g_b2 = g[3], *g_w3 = g[4], *g_b3 = g[5], *g_wf = with_fc1 ? g[6] : nullptr;
    float *bias_part = h->w_part + h->w_part_floats;  // [splits][CO] partial bias sums
    const size_t c3 = (size_t)2 * h->F1 * 9 * 2 * h->F1, c2 = (size_t)2 * h->F1 * 16 * h->F1;
    hipStream_t sd = h->side;
    SRLX_HIP(hipEventRecord(h->ev_d3, st));  // dact3 exists; nothing reads the first dense layer's weights any more
    {   // conv3 (3x3 stride 1 pad 1, act2 -> act3) data gradient on the padded grid (OH2 + 2)^2
        const int HP = h->OH2 + 2, WP = h->OW2 + 2;
        // split over K in two (round 5): 254 workgroups of 18 K-slabs -- one per CU, each a serial chain of stage / barrier / multiply -- become 508 of 9 that
        // overlap two to a CU; the fold adds the two partial slabs per padded position
        // (srlx_qnet_set_dgrad_split: for a handle that has the GPU to itself -- learner rank 0.293 -> 0.287 ms per period, the GEMM 21.2 -> 12.2 us; beside a GPU's own
        // actors the chip is full either way: 0.4184 against 0.4173 ms per lock-step)
        const int nsplit = (C2 == 64 && h->dgrad_split == 2) ? 2 : 1;
        SRLX_TRY(srlx_qnet_dgrad_gemm(h->dact3, B, HP, WP, h->OH3, h->OW3, C2, 3, 3, 1, h->w_t, C2, h->dxpad, st, nsplit));
        const i64 tot = (i64)B * h->OH2 * h->OW2 * C2;
        hipLaunchKernelGGL(k_fold_pad, dim3((unsigned)((tot / 4 + 255) / 256)), dim3(256), 0, st, B, ss, h->OH2, h->OW2, C2, 1, HP, WP, 1, HP, WP, h->dxpad, h->act2, h->dact2, nsplit,
                           (i64)B * HP * WP * C2);
    }
    SRLX_STAMP(18, st);
    SRLX_HIP(hipEventRecord(h->ev_d2, st));
    {   // conv2 (4x4 stride 2 pad 2, act1 -> act2) data gradient on the padded grid (OH1 + 4)^2, four parity classes
        const int HP = h->OH1 + 4, WP = h->OW1 + 4, QH = (HP + 1) / 2, QW = (WP + 1) / 2;
        SRLX_TRY(srlx_qnet_dgrad_gemm(h->dact2, B, QH, QW, h->OH2, h->OW2, C2, 4, 4, 2, h->w_t2, h->F1, h->dxpad, st));
        const i64 tot = (i64)B * h->OH1 * h->OW1 * h->F1;
        hipLaunchKernelGGL(k_fold_pad, dim3((unsigned)((tot / 4 + 255) / 256)), dim3(256), 0, st, B, ss, h->OH1, h->OW1, h->F1, 2, HP, WP, 2, QH, QW, h->dxpad, h->act1, h->dact1);
    }
    SRLX_STAMP(19, st);
    SRLX_HIP(hipEventRecord(h->ev_d1, st));
    // ---- weight gradients of conv3, conv2 and the first dense layer (side stream).  The dense layer's comes last: with Adam in its
    // epilogue it streams 160 MB, and beside the data-gradient chain it tripled the duration of that chain's pad-fold kernels
    const dim3 fg((unsigned)((K / 32 + 3) / 4), (unsigned)(N1 / 32));  // one wave per 32 x 32 tile of the weight
    if (with_fc1 && !h->adam_m)
        hipLaunchKernelGGL(k_fc1_wgrad<false>, fg, dim3(256), 0, sd, B, ss, N1, K, h->dh1, h->act3, g_wf, nullptr, nullptr, nullptr, 0.0, 0.0, 0.0, 0.0, nullptr);
    SRLX_HIP(hipStreamWaitEvent(sd, h->ev_d3, 0));
    // where the Adam-fused first-dense-layer weight gradient goes (srlx_qnet_set_fc1_branch): 0 = last on the side stream (rounds 2-3),
    // 1 = first on the side stream (as soon as the data gradient has read the weights: ev_d3), 2 = a branch of its own (side2) from ev_d3 -- a THIRD concurrent
    // branch of a captured update: only where the actors' stream does not share a hardware-queue pool with the graph's internal streams (tools/README.md, 5)
    const int fc1_order = h->fc1_order;
    auto launch_fc1_adam = [&](hipStream_t s2) {
        if (h->adam_planes_out)
            hipLaunchKernelGGL((k_fc1_wgrad<true, true>), fg, dim3(256), 0, s2, B, ss, N1, K, h->dh1, h->act3, nullptr, const_cast<float *>(h->wf), h->adam_m, h->adam_v,
                               h->adam_lr, h->adam_b1, h->adam_b2, h->adam_eps, h->adam_step, (_Float16 *)h->adam_planes_out);
        else
            hipLaunchKernelGGL(k_fc1_wgrad<true>, fg, dim3(256), 0, s2, B, ss, N1, K, h->dh1, h->act3, nullptr, const_cast<float *>(h->wf), h->adam_m, h->adam_v, h->adam_lr,
                               h->adam_b1, h->adam_b2, h->adam_eps, h->adam_step);
    };
    const bool fc1_adam = with_fc1 && h->adam_m;
    if (fc1_adam && fc1_order == 1) launch_fc1_adam(sd);
    if (fc1_adam && fc1_order == 2) {
        if (!h->side2) SRLX_HIP(hipStreamCreateWithFlags(&h->side2, hipStreamNonBlocking));
        SRLX_HIP(hipStreamWaitEvent(h->side2, h->ev_d3, 0));
        launch_fc1_adam(h->side2);
        SRLX_HIP(hipEventRecord(h->ev_join2, h->side2));
    }
    ConvGeo g3{h->OH2, h->OW2, C2, h->OH3, h->OW3, C2, 3, 3, 1, 1};
    hipLaunchKernelGGL(k_conv_wgrad_mfma<64>, dim3((9 * 2 * 2 + 3) / 4, (unsigned)B), dim3(256), 0, sd, g3, ss, h->act2, h->dact3, h->w_part, bias_part);
    // the convolution tensors' optimiser steps ride on their reductions (srlx_qnet_fuse_adam_rest); a tensor's raw weights have no reader left in this update:
    // the data-gradient GEMMs read transposed copies, the forwards packed ones, and the packing launch that rebuilds both runs behind the join
    const bool rest = h->rest_on && with_fc1;
    const AdamArgs ad{h->adam_lr, h->adam_b1, h->adam_b2, h->adam_eps, h->adam_step};
    auto reduce = [&](hipStream_t s_, const float *part_, int P_, i64 n_, float *gw_, const float *bp_, int nb_, float *gb_, int wi, i64 *snap_) {
        const dim3 grid((unsigned)((n_ + nb_ + 255) / 256));
        if (rest)
            hipLaunchKernelGGL(k_reduce_parts<true>, grid, dim3(256), 0, s_, part_, P_, n_, gw_, bp_, nb_, gb_, const_cast<float *>(h->bound[wi]), h->rest_m[wi], h->rest_v[wi],
                               const_cast<float *>(h->bound[wi + 1]), h->rest_m[wi + 1], h->rest_v[wi + 1], ad, snap_);
        else
            hipLaunchKernelGGL(k_reduce_parts<false>, grid, dim3(256), 0, s_, part_, P_, n_, gw_, bp_, nb_, gb_);
    };
    reduce(sd, h->w_part, B, (i64)C2 * 9 * C2, g_w3, bias_part, C2, g_b3, 4, nullptr);
    SRLX_STAMP(22, sd);
    SRLX_HIP(hipStreamWaitEvent(sd, h->ev_d2, 0));
    ConvGeo g2{h->OH1, h->OW1, h->F1, h->OH2, h->OW2, C2, 4, 4, 2, 2};
    hipLaunchKernelGGL(k_conv_wgrad_mfma<32>, dim3((16 * 2 * 1 + 3) / 4, (unsigned)B), dim3(256), 0, sd, g2, ss, h->act1, h->dact2, h->w_part, bias_part);
    reduce(sd, h->w_part, B, (i64)C2 * 16 * h->F1, g_w2, bias_part, C2, g_b2, 2, nullptr);
    SRLX_STAMP(23, sd);
    if (fc1_adam && fc1_order != 1 && fc1_order != 2) launch_fc1_adam(sd);  // Adam in the epilogue updates the weights in place: after ev_d3, when the data gradient has read them
    SRLX_STAMP(24, sd);
    SRLX_HIP(hipEventRecord(h->ev_join, sd));
    // conv1's weight gradient needs the END of the data-gradient chain: it follows it on the caller's stream instead of queueing behind
    // the conv2 / conv3 weight gradients on the side stream
    // output pixels per chunk, even (the MFMA consumes pixel pairs).  Two chunks per (sample, frame) (45 KB of LDS); the partial sums are added up by a
    // k_reduce_parts launch behind the kernel (an in-launch ticket reduction saved the launch and cost 1.6 % per lock-step: `__threadfence()` in 512 workgroups).
    constexpr int c1_chunks = 2;
    constexpr bool c1_in_launch = false;
    const int per = ((h->OH1 * h->OW1 + c1_chunks - 1) / c1_chunks + 1) & ~1;
    const size_t lds = (size_t)kC1Frame + (size_t)per * 4 + (size_t)per * 32 * sizeof(float) + 2 * 16 * 64 * sizeof(float);
    SRLX_REQUIRE(lds <= 64 * 1024, "qnet_backward_u8: conv1 staging needs %zu bytes of LDS", lds);
    float *c1_part = h->w_part + kWgSplits * (c3 > c2 ? c3 : c2), *c1_bias = bias_part + 2 * kWgSplits * 64;  // its own scratch: runs beside conv2's reduction
    hipLaunchKernelGGL(k_conv1_wgrad_mfma, dim3((unsigned)B, (unsigned)(c1_chunks * h->Wn)), dim3(256), lds, st, d_frame_base, d_frame_off, ss, h->Wn, h->H, h->W, h->OH1, h->OW1, per,
                       h->dact1, c1_part, c1_bias, h->c1_gpart, h->c1_cnt, g_w1, g_b1, c1_chunks, c1_in_launch ? 1 : 0);
    if (!c1_in_launch)  // the same sums in the same order (four slices of consecutive partial tensors, then ((s0 + s1) + s2) + s3), behind a kernel boundary
        reduce(st, c1_part, B * c1_chunks, (i64)32 * h->Wn * 64, g_w1, c1_bias, 32, g_b1, 0, h->step_snap);
    SRLX_STAMP(20, st);
    SRLX_HIP(hipStreamWaitEvent(st, h->ev_join, 0));
    if (fc1_adam && fc1_order == 2) SRLX_HIP(hipStreamWaitEvent(st, h->ev_join2, 0));
    return SRLX_OK;
}

// the incoming gradient at conv3's OUTPUT (after its ReLU), rows 0..B-1 compact -> h->dact3 with the ReLU mask of the kept activations (row b * ss)
__global__ void __launch_bounds__(256) k_relu_mask_rows(const float *__restrict__ gy, const float *__restrict__ act, i64 ss, i64 row, i64 total, float *__restrict__ out) {
    const i64 q = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    if (q * 4 >= total) return;
    const i64 b = (q * 4) / row, k = (q * 4) % row;
    const float4 gv = *reinterpret_cast<const float4 *>(gy + q * 4), a = *reinterpret_cast<const float4 *>(act + b * ss * row + k);
    *reinterpret_cast<float4 *>(out + q * 4) = make_float4(a.x > 0.f ? gv.x : 0.f, a.y > 0.f ? gv.y : 0.f, a.z > 0.f ? gv.z : 0.f, a.w > 0.f ? gv.w : 0.f);
}

// `td` != NULL: d loss / d Q is computed by the head kernel itself (srlx_qnet_backward_td_u8), d_grad_q is not read
static int backward_impl(srlx_qnet_t *h, int64_t batch, int64_t sample_stride, const uint8_t *d_frame_base, const int64_t *d_frame_off, const float *d_grad_q,
                         const srlx::TdArgs *td, float *const *g, void *stream) {
    SRLX_REQUIRE(h && d_frame_base && d_frame_off && (d_grad_q || td) && g, "qnet_backward_u8: NULL argument");
    SRLX_REQUIRE(h->max_train > 0, "qnet_backward_u8: call srlx_qnet_enable_training first");
    SRLX_REQUIRE(batch > 0 && batch <= h->max_train && sample_stride >= 1 && batch * sample_stride <= h->max_batch, "qnet_backward_u8: batch %lld x stride %lld out of range",
                 (long long)batch, (long long)sample_stride);
    for (int i = 0; i < 12; i++) SRLX_REQUIRE(g[i], "qnet_backward_u8: gradient buffer %d is NULL", i);
    if (h->rest_on) {  // srlx_qnet_fuse_adam_rest: this pass applies the convolution tensors' optimiser steps itself and leaves the small vectors' to the packing launch
        SRLX_REQUIRE(!h->rest_armed, "qnet_backward_u8: the previous pass's optimiser step was never completed (srlx_qnet_publish must follow every backward pass "
                                     "of a handle with srlx_qnet_fuse_adam_rest)");
        for (int i = 7; i < 12; i++) SRLX_REQUIRE(g[i] == h->rest_g[i], "qnet_backward_u8: gradient buffer %d is not the one srlx_qnet_fuse_adam_rest was given", i);
    }
    srlx::DeviceGuard guard(h->device);
    hipStream_t st = (hipStream_t)stream;
    const int B = (int)batch, N1 = 2 * h->hidden, K = h->flat, A = h->A;
    const i64 ss = sample_stride;
    float *g_bf = g[7], *g_v2w = g[8], *g_v2b = g[9], *g_a2w = g[10], *g_a2b = g[11];

    // head + second layers -> dh1 (masked by the first layer's ReLU), bias gradient of the first layer
    const bool mfma_dgrad = B <= 32;  // the batch fits one 32-row MFMA tile
    if (h->head_mode == 1) {  // the handle ends behind the first dense layer: d_grad_q is the gradient at its first out_cols post-ReLU units
        SRLX_REQUIRE(!td && !h->ln_w, "qnet_backward_u8: a hidden-layer handle takes the gradient of its output (no TD prologue, no LayerNorm inside)");
        hipLaunchKernelGGL(k_hidden_bwd, dim3((unsigned)((N1 + 63) / 64)), dim3(256), 0, st, B, ss, N1, h->out_cols, d_grad_q, h->h1, h->dh1, mfma_dgrad ? h->dh1t : nullptr, g_bf);
    } else {
        const dim3 hg((unsigned)((h->hidden + 63) / 64));
        const size_t hl = (size_t)(B + B * A + (3 + (A <= 8 ? 8 : (A <= 16 ? 16 : 32))) * 256 + (td ? B * A + 2 + 512 : 0)) * sizeof(float);
        float *dh1t = mfma_dgrad ? h->dh1t : nullptr;
        const srlx::TdArgs tda = td ? *td : srlx::TdArgs{};
        const int with_td = td ? 1 : 0;
        if (A <= 8)
            hipLaunchKernelGGL(k_head_bwd<8>, hg, dim3(256), hl, st, B, ss, h->hidden, A, h->dueling, d_grad_q, h->h1, h->v2w, h->a2w, h->dh1, dh1t, g_bf, g_v2w, g_v2b, g_a2w,
                               g_a2b, tda, with_td);
        else if (A <= 16)
            hipLaunchKernelGGL(k_head_bwd<16>, hg, dim3(256), hl, st, B, ss, h->hidden, A, h->dueling, d_grad_q, h->h1, h->v2w, h->a2w, h->dh1, dh1t, g_bf, g_v2w, g_v2b, g_a2w,
                               g_a2b, tda, with_td);
        else
            hipLaunchKernelGGL(k_head_bwd<32>, hg, dim3(256), hl, st, B, ss, h->hidden, A, h->dueling, d_grad_q, h->h1, h->v2w, h->a2w, h->dh1, dh1t, g_bf, g_v2w, g_v2b, g_a2w,
                               g_a2b, tda, with_td);
    }
    SRLX_STAMP(16, st);
    if (h->ev_td) SRLX_HIP(hipEventRecord(h->ev_td, st));  // target / loss / priorities exist: the caller's priority write-back need not wait for the gradients
    // srlx_qnet_set_main_first: the data-gradient chain's first kernel is RECORDED ahead of the side branch's launches.  A captured graph's concurrent branches
    // land on streams in recording order (the chain a node's first-recorded successor starts stays on its stream): with the critical chain recorded first it keeps
    // one hardware queue from the head kernel to the packing launch instead of hopping queues at the fork (-1.2 % per single-GPU lock-step, -4 % per period of a
    // learner rank, update alone 0.313 -> 0.273 ms; profiles/r5_ab_ingest_order.txt)
    const bool main_first = h->main_first;
    const bool uvfa = h->uvfa.X > 0;
    SRLX_REQUIRE(!uvfa || h->uvfa.g_wx, "qnet_backward_u8: a UVFA network needs a gradient buffer for its columns (srlx_qnet_fuse_adam_uvfa)");
    SRLX_TRY(chain_prologue(h, st, main_first && mfma_dgrad ? 1 : 3, uvfa, B, ss));
    // ---- data-gradient chain (caller's stream)
    if (mfma_dgrad) {
        hipLaunchKernelGGL(k_fc1_dgrad_mfma, dim3((unsigned)(K / 32)), dim3(256), 0, st, B, ss, N1, K, h->dh1t, h->wf, h->act3, h->dact3);
        if (main_first) SRLX_TRY(chain_prologue(h, st, 2, uvfa, B, ss));
    } else {
        hipLaunchKernelGGL(k_fc1_dgrad<64>, dim3((unsigned)((K + 255) / 256), kFcSplits), dim3(256), 0, st, B, N1, K, h->dh1, h->wf, h->fc_part);
        hipLaunchKernelGGL(k_fc1_dgrad_reduce, dim3((unsigned)(((i64)B * K + 255) / 256)), dim3(256), 0, st, B, ss, K, h->fc_part, h->act3, h->dact3);
    }
    SRLX_STAMP(17, st);
    SRLX_TRY(conv_chain(h, B, ss, d_frame_base, d_frame_off, g, st, true));
    // NoisyLinear: d loss / d sigma = d loss / d W_effective * eps of the draw the forward used (regenerated, not stored)
    SRLX_TRY(srlx_qnet_noisy_sigma_grads(h, g, st));
    SRLX_HIP(hipGetLastError());
    h->rest_armed = h->rest_on;
    return SRLX_OK;
}

int srlx_qnet_backward_u8(srlx_qnet_t *h, int64_t batch, int64_t sample_stride, const uint8_t *d_frame_base, const int64_t *d_frame_off,
                          const float *d_grad_q, float *const *g, void *stream) {
    SRLX_REQUIRE(d_grad_q, "qnet_backward_u8: NULL argument");
    return backward_impl(h, batch, sample_stride, d_frame_base, d_frame_off, d_grad_q, nullptr, g, stream);
}

int srlx_qnet_backward_convs_u8(srlx_qnet_t *h, int64_t batch, int64_t sample_stride, const uint8_t *d_frame_base, const int64_t *d_frame_off, const float *d_grad_features,
                                float *const *g, void *stream) {
    SRLX_REQUIRE(h && d_frame_base && d_frame_off && d_grad_features && g, "qnet_backward_convs_u8: NULL argument");
    SRLX_REQUIRE(h->max_train > 0, "qnet_backward_convs_u8: call srlx_qnet_enable_training first");
    SRLX_REQUIRE(batch > 0 && batch <= h->max_train && sample_stride >= 1 && batch * sample_stride <= h->max_batch, "qnet_backward_convs_u8: batch %lld x stride %lld out of range",
                 (long long)batch, (long long)sample_stride);
    for (int i = 0; i < 6; i++) SRLX_REQUIRE(g[i], "qnet_backward_convs_u8: gradient buffer %d is NULL", i);
    srlx::DeviceGuard guard(h->device);
    hipStream_t st = (hipStream_t)stream;
    SRLX_TRY(chain_prologue(h, st));
    const i64 total = (i64)batch * h->flat;
    hipLaunchKernelGGL(k_relu_mask_rows, dim3((unsigned)((total / 4 + 255) / 256)), dim3(256), 0, st, d_grad_features, h->act3, (i64)sample_stride, (i64)h->flat, total, h->dact3);
    SRLX_TRY(conv_chain(h, (int)batch, (i64)sample_stride, d_frame_base, d_frame_off, g, st, false));
    SRLX_HIP(hipGetLastError());
    return SRLX_OK;
}

int srlx_qnet_backward_td_u8(srlx_qnet_t *h, int64_t batch, int n_step, const uint8_t *d_frame_base, const int64_t *d_frame_off, const float *d_q_on_all,
                             const float *d_q_tg_next, const int32_t *d_actions, const float *d_rewards, const float *d_terminated, const uint8_t *d_invalid_next,
                             const float *d_weights, double discount, double retrace_h, int enable_double_dqn, int enable_rescale, float *d_target, float *d_loss,
                             float *d_grad_q0, float *d_priorities, float *const *g, void *stream) {
    SRLX_REQUIRE(h, "qnet_backward_td_u8: NULL handle");
    SRLX_REQUIRE(batch > 0 && batch <= 256 && n_step >= 1 && n_step <= srlx::kTdMaxStep, "qnet_backward_td_u8: bad sizes (n_step <= %d)", srlx::kTdMaxStep);
    SRLX_REQUIRE(d_q_on_all && d_q_tg_next && d_actions && d_rewards && d_terminated && d_weights, "qnet_backward_td_u8: NULL input");
    SRLX_REQUIRE(d_target && d_loss && d_grad_q0 && d_priorities, "qnet_backward_td_u8: NULL output");
    const int A = h->A;
    const int64_t row = (int64_t)(n_step + 1) * A;
    srlx::TdArgs td{batch, n_step, A, d_q_on_all + A, d_q_tg_next, d_q_on_all, d_actions, d_rewards, d_terminated, d_invalid_next, d_weights, discount, retrace_h,
                          enable_double_dqn, enable_rescale, d_target, d_loss, d_grad_q0, d_priorities, row, row};
    srlx::td_fill_discounts(td);
    td.disc_ps = h->td_disc_ps, td.td_signed = h->td_signed;  // (srlx_qnet_set_td_extras; NULL: Rainbow's scalar discount, priorities only)
    return backward_impl(h, batch, n_step + 1, d_frame_base, d_frame_off, nullptr, &td, g, stream);
}

}  // extern "C"
