// srlx_noisy.hip -- NoisyLinear dense layers for the matrix-core Q-network.
//
// Replaces srl/rl/torch_/modules/noisy_linear.py:26-52 (the dense layers of the reference's own Atari configuration,
// rainbow.py:116-148 `enable_noisy_dense=True`): every forward call draws independent Gaussian noise for every weight and
// bias,  W = w_mu + w_sigma * eps_w,  b = b_mu + b_sigma * eps_b,  eps ~ N(0, 1),  and all rows of the call share that draw.
// The reference draws 2 x (7744 x 512) + ... normals with torch.randn per forward (47 ms per policy step on the CPU,
// SURVEY.md section 3.1).  Here one launch materialises the six effective tensors of a draw from a keyed counter generator --
// eps(draw, tensor, i) is a pure function, so the backward pass regenerates it for d loss / d sigma = d loss / d W * eps instead of
// storing 32 MB of noise -- and the GEMM / head kernels read them exactly like plain weights.  A draw costs 64 MB of reads and
// 32 MB of writes (HBM-bound, ~20 us) against the 150 us dense-layer GEMM it feeds; generating the noise inside the GEMM's
// B-operand loader instead would regenerate every normal once per 128-row tile (8 times at 1024 rows) on the VALU slots that
// the f32 MFMAs need.
//
// Parity with the reference is statistical (its noise comes from torch's Philox stream): tests/test_noisy_gpu.py checks
// mean / variance of the effective weights and of Q over many draws against torch's NoisyLinear, exact equality at sigma = 0,
// and the gradients against torch autograd of the same effective weights.
#include "srlx_qnet_int.h"

namespace {
using i64 = int64_t;
using u64 = unsigned long long;
using srlx::rng_u64;

struct NoisySet {
    const float *mu[6], *sig[6];
    float *out[6];
    i64 begin[7];  // prefix sums of the element PAIR counts (one Box-Muller evaluation yields the normals of elements 2j, 2j+1)
};

// two independent standard normals from one 64-bit draw (Box-Muller on two 24-bit uniforms; u1 in (0, 1])
__device__ __forceinline__ float2 normal_pair(u64 x) {
    const float u1 = (float)((unsigned)(x >> 40) + 1u) * (1.0f / 16777216.0f);
    const float u2 = (float)((unsigned)(x >> 8) & 0xFFFFFFu) * (1.0f / 16777216.0f);
    const float r = sqrtf(-2.0f * logf(u1));
    float s, c;
    sincosf(6.283185307179586f * u2, &s, &c);
    return make_float2(r * c, r * s);
}

__device__ __forceinline__ int find_tensor(const i64 *begin, i64 j) {
    int t = 0;
#pragma unroll
    for (int k = 1; k < 6; k++) t += j >= begin[k] ? 1 : 0;
    return t;
}

__global__ void __launch_bounds__(256) k_noisy_eff(NoisySet s, const i64 *n_elems, u64 seed, i64 *draw) {
    const i64 id = draw[0];
    if (blockIdx.x == 0 && threadIdx.x == 0) draw[1] = id;
    const i64 total = s.begin[6];
    for (i64 j = (i64)blockIdx.x * blockDim.x + threadIdx.x; j < total; j += (i64)gridDim.x * blockDim.x) {
        const int t = find_tensor(s.begin, j);
        const i64 p = j - s.begin[t], e = 2 * p, n = n_elems[t];
        const float2 z = normal_pair(rng_u64(seed + 0x9E37ull * (u64)(t + 1), (u64)id, (u64)p));
        s.out[t][e] = s.mu[t][e] + s.sig[t][e] * z.x;
        if (e + 1 < n) s.out[t][e + 1] = s.mu[t][e + 1] + s.sig[t][e + 1] * z.y;
    }
}

struct SigmaGradSet {
    const float *g_eff[6];
    float *g_sig[6];
    i64 begin[7];
};

__global__ void __launch_bounds__(256) k_noisy_sigma_grad(SigmaGradSet s, const i64 *n_elems, u64 seed, const i64 *draw) {
    const i64 id = draw[1];
    const i64 total = s.begin[6];
    for (i64 j = (i64)blockIdx.x * blockDim.x + threadIdx.x; j < total; j += (i64)gridDim.x * blockDim.x) {
        const int t = find_tensor(s.begin, j);
        const i64 p = j - s.begin[t], e = 2 * p, n = n_elems[t];
        const float2 z = normal_pair(rng_u64(seed + 0x9E37ull * (u64)(t + 1), (u64)id, (u64)p));
        s.g_sig[t][e] = s.g_eff[t][e] * z.x;
        if (e + 1 < n) s.g_sig[t][e + 1] = s.g_eff[t][e + 1] * z.y;
    }
}

i64 fill_begin(const srlx_qnet *h, i64 *begin) {
    begin[0] = 0;
    for (int t = 0; t < 6; t++) begin[t + 1] = begin[t] + (h->eff_n[t] + 1) / 2;
    return begin[6];
}
}  // namespace

int srlx_qnet_noisy_refresh(srlx_qnet *h, hipStream_t st) {
    if (!h->sig[0]) return SRLX_OK;
    NoisySet s;
    for (int t = 0; t < 6; t++) s.mu[t] = h->mu[t], s.sig[t] = h->sig[t], s.out[t] = h->eff[t];
    const i64 total = fill_begin(h, s.begin);
    const unsigned grid = (unsigned)((total + 1023) / 1024 < 4096 ? (total + 1023) / 1024 : 4096);
    hipLaunchKernelGGL(k_noisy_eff, dim3(grid), dim3(256), 0, st, s, h->d_draw + 2, h->noisy_seed, h->d_draw);
    SRLX_HIP(hipGetLastError());
    return SRLX_OK;
}

int srlx_qnet_noisy_sigma_grads(srlx_qnet *h, float *const *g, hipStream_t st) {
    if (!h->sig[0]) return SRLX_OK;
    SRLX_REQUIRE(h->g_sig[0], "qnet_backward_u8: the network is noisy but no sigma gradient tensors are bound (srlx_qnet_bind_noisy_grads)");
    SigmaGradSet s;
    for (int t = 0; t < 6; t++) s.g_eff[t] = g[6 + t], s.g_sig[t] = h->g_sig[t];
    const i64 total = fill_begin(h, s.begin);
    const unsigned grid = (unsigned)((total + 1023) / 1024 < 4096 ? (total + 1023) / 1024 : 4096);
    hipLaunchKernelGGL(k_noisy_sigma_grad, dim3(grid), dim3(256), 0, st, s, h->d_draw + 2, h->noisy_seed, h->d_draw);
    SRLX_HIP(hipGetLastError());
    return SRLX_OK;
}

extern "C" {

int srlx_qnet_bind_noisy(srlx_qnet_t *h, const float *const *d_sigma, uint64_t seed) {
    SRLX_REQUIRE(h && d_sigma, "qnet_bind_noisy: NULL argument");
    SRLX_REQUIRE(h->w1, "qnet_bind_noisy: bind the parameters first (srlx_qnet_bind: the dense-layer entries are the mu tensors)");
    for (int t = 0; t < 6; t++) SRLX_REQUIRE(d_sigma[t], "qnet_bind_noisy: sigma %d is NULL", t);
    srlx::DeviceGuard guard(h->device);
    const int N1 = 2 * h->hidden;
    const int64_t n[6] = {(int64_t)N1 * h->flat, N1, h->hidden, 1, (int64_t)h->A * h->hidden, h->A};
    const float *mu_now[6] = {h->wf, h->bf, h->v2w, h->v2b, h->a2w, h->a2b};
    const bool first = h->eff[0] == nullptr;
    for (int t = 0; t < 6; t++) {
        if (first) {
            h->eff_n[t] = n[t];
            SRLX_HIP(hipMalloc((void **)&h->eff[t], (size_t)n[t] * sizeof(float)));
            h->mu[t] = mu_now[t];
        }
        h->sig[t] = d_sigma[t];
    }
    if (first) {
        SRLX_HIP(hipMalloc((void **)&h->d_draw, 8 * sizeof(int64_t)));
        int64_t init[8] = {0, -1, n[0], n[1], n[2], n[3], n[4], n[5]};
        SRLX_HIP(hipMemcpy(h->d_draw, init, sizeof(init), hipMemcpyHostToDevice));
    }
    h->noisy_seed = seed;
    // from now on the dense kernels read the effective tensors
    h->wf = h->eff[0], h->bf = h->eff[1], h->v2w = h->eff[2], h->v2b = h->eff[3], h->a2w = h->eff[4], h->a2b = h->eff[5];
    return SRLX_OK;
}

int srlx_qnet_bind_noisy_grads(srlx_qnet_t *h, float *const *d_grad_sigma) {
    SRLX_REQUIRE(h && d_grad_sigma, "qnet_bind_noisy_grads: NULL argument");
    for (int t = 0; t < 6; t++) {
        SRLX_REQUIRE(d_grad_sigma[t], "qnet_bind_noisy_grads: gradient tensor %d is NULL", t);
        h->g_sig[t] = d_grad_sigma[t];
    }
    return SRLX_OK;
}

int srlx_qnet_noisy_effective(srlx_qnet_t *h, int which, float *d_out, int64_t *n_elems, int64_t *draw_id, void *stream) {
    SRLX_REQUIRE(h && which >= 0 && which < 6 && h->sig[0], "qnet_noisy_effective: not a noisy network or bad index");
    srlx::DeviceGuard guard(h->device);
    if (n_elems) *n_elems = h->eff_n[which];
    if (d_out) SRLX_HIP(hipMemcpyAsync(d_out, h->eff[which], (size_t)h->eff_n[which] * sizeof(float), hipMemcpyDeviceToDevice, (hipStream_t)stream));
    if (draw_id) {
        SRLX_HIP(hipStreamSynchronize((hipStream_t)stream));
        SRLX_HIP(hipMemcpy(draw_id, h->d_draw + 1, sizeof(int64_t), hipMemcpyDeviceToHost));
    }
    return SRLX_OK;
}

int srlx_qnet_redraw_rows(srlx_qnet_t *h, int64_t rows, int64_t row_stride, float *d_q, void *stream) {
    SRLX_REQUIRE(h && d_q, "qnet_redraw_rows: NULL argument");
    SRLX_REQUIRE(rows > 0 && row_stride >= 1 && rows * row_stride <= h->max_batch, "qnet_redraw_rows: rows %lld x stride %lld out of range", (long long)rows,
                 (long long)row_stride);
    srlx::DeviceGuard guard(h->device);
    hipStream_t st = (hipStream_t)stream;
    SRLX_TRY(srlx_qnet_noisy_refresh(h, st));
    return srlx_qnet_dense_rows(h, rows, row_stride, d_q, st);
}

}  // extern "C"
