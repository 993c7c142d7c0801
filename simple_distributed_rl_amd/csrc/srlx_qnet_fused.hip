// srlx_qnet_fused.hip -- conv1 -> conv2 -> conv3 of the DQN image block as ONE kernel, one workgroup per sample.
//
// Replaces the three launches k_conv1_u8 + k_gemm<AConv> x 2 of srlx_qnet.hip for the Atari geometry (84 x 84 x 4 frames,
// 32 / 64 / 64 filters; srl/rl/torch_/blocks/dqn_image_block.py:29-54).  The implicit-GEMM kernels spend a third of the f32
// matrix-core peak on their im2col: 93 VALU instructions per 32 MFMAs for clamped addresses, and f32 MFMAs share issue
// slots with VALU on gfx950.  Here a workgroup (8 waves) owns one sample end to end:
//   frames  4 x 88 x 88 uint8 (replicate padding materialised)     31.0 KB of LDS   <- the ring, through the frame-offset table
//   act1    441 pixels x 32 channels f32, pixel stride 36           63.5 KB          <- conv1 (filters in registers, as k_conv1_u8)
//   act2    121 pixels x 64 channels f32, pixel stride 68           32.9 KB          <- conv2
//   act3    121 x 64 f32 -> HBM (NHWC flatten, the FC1 GEMM's A operand)             <- conv3
// The activations never leave the CU, and because a lane's GEMM rows (output pixels) are fixed for a whole layer, every
// im2col address is computed ONCE per tile and tap before the K loop, which is then ds_read_b128 (A fragments out of LDS),
// global_load_dwordx4 (B fragments: the filters stream from L2, 272 KB shared by every workgroup) and MFMAs only.
// Tiles: 32 x 32 outputs per wave (v_mfma_f32_32x32x2_f32, exact f32); conv2 / conv3 are 4 pixel tiles x 2 channel tiles = one
// tile per wave, conv1 14 pixel tiles over the 8 waves.  Pixel strides 36 / 68 floats spread the 16-lane groups of a
// ds_read_b128 over the 64 banks.  With `act1_out` / `act2_out` the activations are ALSO written to HBM for the backward
// pass of a training forward.
#include "srlx_adam_math.h"
#include "srlx_qnet_int.h"

namespace {

using i64 = int64_t;
using u8 = unsigned char;
using f32x16 = __attribute__((ext_vector_type(16))) float;
using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
using f16x8 = __attribute__((ext_vector_type(8))) _Float16;
using f16x4 = __attribute__((ext_vector_type(4))) _Float16;

constexpr int kPad = 88, kFrame = kPad * kPad;     // staged frame side / bytes
constexpr int kP1 = 21, kM1 = kP1 * kP1;           // conv1 output side / pixels
constexpr int kP2 = 11, kM2 = kP2 * kP2;           // conv2 = conv3 output side / pixels
constexpr int kS1 = 36, kS2 = 68;                  // LDS pixel strides (floats) of act1 / act2
constexpr int kWaves = 8;
constexpr size_t kLdsF32 = 4 * kFrame + (size_t)kM1 * kS1 * 4 + (size_t)kM2 * kS2 * 4;  // every activation float32 (the f32-pipe A/B variants)
// The split-bf16 kernel keeps act1 / act2 in LDS as three bf16 part planes per pixel ([part][channel]; 16 bytes of padding make the pixel stride an odd
// multiple of 16 bytes, so the 16 lanes of a ds_read_b128 phase hit 16 different bank quads):
//   [frames 31.0 KB | conv1's shared filter parts 32.1 KB] -> later act2 planes (121 x 400 B = 47.3 KB) | act1 planes 441 x 208 B = 89.6 KB
// the K-quarter reductions park their partials in the act1 region once its readers are done.
constexpr int kPB1 = 3 * 32 * 2 + 16, kPB2 = 3 * 64 * 2 + 16;
// Round 6, the TWO-PART FLOAT16 SPLIT (H16; the default): x = hi + lo / 2048 with hi = f16(x), lo = f16((x - hi) * 2048) -- 22 significand bits (the residual is exact in
// float32, and scaled by 2^11 it is a NORMAL f16 wherever hi is: nothing is lost to f16's short exponent range below |x| = 2^-14, where the absolute error is 2^-36) --
// and x * w = hi * hi' + (hi * lo' + lo * hi') / 2048 with the lo * lo' term (<= 2^-24 |x w|) dropped: THREE exact products per 16 K instead of six, into two
// accumulators (the unscaled and the 2^11-scaled sums, joined once per K quarter).  Error per product <= 3 * 2^-24: float32 round-off.  Two parts of two bytes = the
// float32's own four bytes: pixel strides 144 / 272 bytes (odd multiples of 16).  Range: an activation above 65 504 would overflow hi; the epilogues
// then set a bit of the handle's range word (srlx_qnet_range_flags; the host side raises where it synchronises anyway) -- SRLX_CONV_BF16X3=1 is the path for such a network.
constexpr int kPB1h = 2 * 32 * 2 + 16, kPB2h = 2 * 64 * 2 + 16;
constexpr float kLoScale = 2048.0f, kLoInv = 1.0f / 2048.0f;
constexpr size_t kOffA1P = 4 * kFrame + (size_t)kM2 * kS2 * 4;
constexpr size_t kLdsPlanes = kOffA1P + (size_t)kM1 * kPB1;
constexpr size_t kLdsBytes = kLdsPlanes > kLdsF32 ? kLdsPlanes : kLdsF32;
static_assert((size_t)kM2 * kPB2 <= kOffA1P && kOffA1P % 16 == 0 && kLdsBytes <= 160 * 1024, "LDS plan of the split-bf16 kernel");
// what the float16 kernel asks for: act1 planes end at 127 392, the K-quarter scratch (96 KB behind the act2 planes) at 132 096 -- 28 KB of the CU's LDS stay free for a
// neighbour's workgroup (the learner's small kernels run beside the actors' pass)
constexpr size_t kLdsH16A = kOffA1P + (size_t)kM1 * kPB1h, kLdsH16B = ((size_t)kM2 * kPB2h + 1023) / 1024 * 1024 + 2 * 4 * 3 * 4 * 64 * 16;
constexpr size_t kLdsH16 = kLdsH16A > kLdsH16B ? kLdsH16A : kLdsH16B;

__device__ __forceinline__ int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

// Filters in MFMA-fragment order.  A wave's B fragment of one K-slab is, per lane (i = lane & 31, h = lane >> 5), the 16 floats
// W[n0 + i][slab*32 + 16 h .. + 15]: read from the torch layout ([n][K] rows of 512 / 576 / 256 floats) every lane of a load
// instruction touches its own 128-byte line -- 64 tag look-ups per instruction, and with eight waves streaming filters the
// texture-address unit is as busy as the matrix pipe.  Packed as [slab][n-tile][v][lane][4] a load instruction of a wave is one
// contiguous KiB (8 lines).  311 KB for the three layers, rebuilt by one small launch per forward (the filters are the torch
// parameters themselves and change with every optimiser step).
constexpr int kW1 = 32 * 256, kW2 = 64 * 512, kW3 = 64 * 576;
constexpr int kW1B = 16 * 3 * 64 * 4;  // conv1's filters once more as split-bf16 MFMA fragments: [step 16][part 3][lane 64] x 16 bytes (in floats)
constexpr int kW2B = 32 * 2 * 3 * 64 * 4, kW3B = 36 * 2 * 3 * 64 * 4;  // conv2 / conv3 filters as split-bf16 fragments: [step][n-tile][part][lane] x 16 bytes (in floats)
constexpr int kPackFloats = kW1 + kW2 + kW3 + kW1B + kW2B + kW3B;

// A training handle's launch also builds the transposed filters of the backward pass's two data-gradient GEMMs (wT3 / wT2: layout of
// k_transpose_filter in srlx_qnet_bwd.hip) -- they depend on the weights only, and the weights do not change between this forward and
// its backward, so two launches leave the learner's critical path.
// `out2` (optional): a second copy of the packed buffer -- an actor handle's published set (srlx_qnet_publish) -- and `sm`: the network's small vectors
// (biases, the head's second layers) copied element by element into that set's block by the threads behind the packing ranges.
constexpr int kSmallVecs = 9;  // b1 b2 b3 bf v2w v2b a2w a2b + the UVFA columns of an Agent57(_light) Q-network (round 6; length 0 elsewhere)
struct SmallCopy {
    const float *src[kSmallVecs];
    int off[kSmallVecs + 1];  // destination offsets (floats); off[kSmallVecs] = total
    int len[kSmallVecs];  // elements of each vector (the padding behind it is never read or written)
    float *dst;  // NULL: nothing to copy
    int first;   // first thread index of the copy range
    long long *bump;  // int64 device counter advanced by the launch (NULL: none): the update's step count, whose readers all ran in earlier launches
    // srlx_qnet_fuse_adam_rest: the optimiser step of vector k right here (g[k] != NULL), in place, before it is copied -- every element has exactly one thread.
    // The step count comes from `snap` (copied by an earlier launch of the update): thread 0 of this launch advances the count itself.
    const float *g[kSmallVecs];
    float *m[kSmallVecs], *v[kSmallVecs];
    double lr, beta1, beta2, eps;
    const long long *snap;
    int h16;  // the matrix-pipe fragments as two float16 parts (the default) instead of three bf16 parts
};
__global__ void __launch_bounds__(256) k_pack_filters(const float *__restrict__ w1, const float *__restrict__ w2, const float *__restrict__ w3,
                                                      float *__restrict__ out, float *__restrict__ wT3, float *__restrict__ wT2, float *__restrict__ out2, SmallCopy sm) {
    // one thread per float4 of the packed float32 buffer, then one per (step, lane) of conv1's split-bf16 fragments
    const int q = blockIdx.x * blockDim.x + threadIdx.x;
    const int n4 = (kW1 + kW2 + kW3) / 4;
    if (sm.bump && q == 0) *sm.bump += 1;
    if (sm.first > 0 && q >= sm.first) {
        const int x = q - sm.first;
        if (x >= sm.off[kSmallVecs]) return;
        int v = 0;
#pragma unroll
        for (int k = 1; k < kSmallVecs; k++) v += x >= sm.off[k] ? 1 : 0;
        const int j = x - sm.off[v];
        if (j >= sm.len[v]) return;
        // (selects over the by-value tables: a lane-dependent index would send them to scratch memory)
        const float *src = sm.src[0], *gp = sm.g[0];
        float *mp = sm.m[0], *vp = sm.v[0];
#pragma unroll
        for (int k = 1; k < kSmallVecs; k++)
            if (v == k) src = sm.src[k], gp = sm.g[k], mp = sm.m[k], vp = sm.v[k];
        float pv = src[j];
        if (gp) {
            float mv = mp[j], vv = vp[j];
            const srlx::AdamCoef cf = srlx::adam_coef(sm.lr, sm.beta1, sm.beta2, sm.eps, (int64_t)*sm.snap);
            srlx::adam_one(pv, gp[j], mv, vv, cf);
            const_cast<float *>(src)[j] = pv;
            mp[j] = mv;
            vp[j] = vv;
        }
        if (sm.dst) sm.dst[x] = pv;
        return;
    }
    if (q >= n4 && q < n4 + 16 * 64) {
        // conv1 on the bf16 matrix pipe, exactly: w / 255 (float32, as the float32 path folds it) = p0 + p1 + p2 with three bf16 parts (8 + 8 + 8
        // mantissa bits; each remainder is exact in float32), the pixel operand is a uint8 and exact in ONE bf16, every partial product is exact in
        // the float32 accumulator.  Fragment of v_mfma_f32_32x32x16_bf16: lane (i, h) holds k = 16 step + 8 h + 0..7 of filter row i.
        const int idx = q - n4, lane = idx & 63, step = idx >> 6, i = lane & 31, hh = lane >> 5;
        if (sm.h16) {  // w * 256 / 255 = hi + lo / 2048 (times 256: the parts of a small filter stay normal float16s; the kernel's epilogue divides)
            f16x8 hp[2];
#pragma unroll
            for (int j = 0; j < 8; j++) {
                const float r = w1[i * 256 + step * 16 + hh * 8 + j] * (1.0f / 255.0f) * 256.0f;
                const _Float16 hi = (_Float16)r;
                hp[0][j] = hi, hp[1][j] = (_Float16)((r - (float)hi) * kLoScale);
            }
            f16x8 *dst = reinterpret_cast<f16x8 *>(out + kW1 + kW2 + kW3), *d2 = out2 ? reinterpret_cast<f16x8 *>(out2 + kW1 + kW2 + kW3) : nullptr;
#pragma unroll
            for (int t = 0; t < 2; t++) {
                dst[(step * 2 + t) * 64 + lane] = hp[t];
                if (d2) d2[(step * 2 + t) * 64 + lane] = hp[t];
            }
            return;
        }
        bf16x8 part[3];
#pragma unroll
        for (int j = 0; j < 8; j++) {
            float r = w1[i * 256 + step * 16 + hh * 8 + j] * (1.0f / 255.0f);
#pragma unroll
            for (int t = 0; t < 3; t++) {
                const __bf16 b = (__bf16)r;
                part[t][j] = b;
                r -= (float)b;
            }
        }
        bf16x8 *dst = reinterpret_cast<bf16x8 *>(out + kW1 + kW2 + kW3);
#pragma unroll
        for (int t = 0; t < 3; t++) dst[(step * 3 + t) * 64 + lane] = part[t];
        if (out2) {
            bf16x8 *d2 = reinterpret_cast<bf16x8 *>(out2 + kW1 + kW2 + kW3);
#pragma unroll
            for (int t = 0; t < 3; t++) d2[(step * 3 + t) * 64 + lane] = part[t];
        }
        return;
    }
    if (q >= n4 + 16 * 64 && q < n4 + 16 * 64 + (32 + 36) * 2 * 64) {
        // conv2 / conv3 filters as three bf16 parts in MFMA-fragment order (k_convnet_fused<.., C23B16 = true>): lane (i, h) of (step, n-tile) holds
        // k = 16 step + 8 h + 0..7 of filter row n-tile * 32 + i; K = (tap, channel) as in the float32 path
        int idx = q - n4 - 16 * 64;
        const bool third = idx >= 32 * 2 * 64;
        if (third) idx -= 32 * 2 * 64;
        const int lane = idx & 63, nt = (idx >> 6) & 1, step = idx >> 7, i = lane & 31, hh = lane >> 5;
        const float *src = third ? w3 : w2;
        const int K = third ? 576 : 512;
        if (sm.h16) {
            f16x8 hp[2];
#pragma unroll
            for (int j = 0; j < 8; j++) {
                const float r = src[(nt * 32 + i) * K + step * 16 + hh * 8 + j];
                const _Float16 hi = (_Float16)r;
                hp[0][j] = hi, hp[1][j] = (_Float16)((r - (float)hi) * kLoScale);
            }
            const size_t o = kW1 + kW2 + kW3 + kW1B + (third ? kW2B : 0);
            f16x8 *dst = reinterpret_cast<f16x8 *>(out + o), *d2 = out2 ? reinterpret_cast<f16x8 *>(out2 + o) : nullptr;
#pragma unroll
            for (int t = 0; t < 2; t++) {
                dst[((step * 2 + nt) * 2 + t) * 64 + lane] = hp[t];
                if (d2) d2[((step * 2 + nt) * 2 + t) * 64 + lane] = hp[t];
            }
            return;
        }
        bf16x8 part[3];
#pragma unroll
        for (int j = 0; j < 8; j++) {
            float r = src[(nt * 32 + i) * K + step * 16 + hh * 8 + j];
#pragma unroll
            for (int t = 0; t < 3; t++) {
                const __bf16 b = (__bf16)r;
                part[t][j] = b;
                r -= (float)b;
            }
        }
        bf16x8 *dst = reinterpret_cast<bf16x8 *>(out + kW1 + kW2 + kW3 + kW1B + (third ? kW2B : 0));
#pragma unroll
        for (int t = 0; t < 3; t++) dst[((step * 2 + nt) * 3 + t) * 64 + lane] = part[t];
        if (out2) {
            bf16x8 *d2 = reinterpret_cast<bf16x8 *>(out2 + kW1 + kW2 + kW3 + kW1B + (third ? kW2B : 0));
#pragma unroll
            for (int t = 0; t < 3; t++) d2[((step * 2 + nt) * 3 + t) * 64 + lane] = part[t];
        }
        return;
    }
    if (q >= n4) {
        if (!wT3) return;
        int i = q - n4 - 16 * 64 - (32 + 36) * 2 * 64;
        if (i < kW3) {  // conv3: 64 x (3 x 3) x 64, stride 1: wT[ci][tap * 64 + co] = W[co][tap][ci]
            const int ci = i % 64, tap = (i / 64) % 9, co = i / (64 * 9);
            wT3[ci * 576 + tap * 64 + co] = w3[i];
        } else if ((i -= kW3) < kW2) {  // conv2: 64 x (4 x 4) x 32, stride 2: four parity classes, wT[cls][ci][((ky/2) * 2 + kx/2) * 64 + co]
            const int ci = i % 32, tap = (i / 32) % 16, co = i / (32 * 16);
            const int ky = tap / 4, kx = tap % 4, cls = (ky % 2) * 2 + (kx % 2);
            wT2[(cls * 32 + ci) * 256 + ((ky / 2) * 2 + kx / 2) * 64 + co] = w2[i];
        }
        return;
    }
    const float *src;
    int K, tiles, local;
    if (q < kW1 / 4) src = w1, K = 256, tiles = 1, local = q;
    else if (q < (kW1 + kW2) / 4) src = w2, K = 512, tiles = 2, local = q - kW1 / 4;
    else src = w3, K = 576, tiles = 2, local = q - (kW1 + kW2) / 4;
    const int lane = local & 63, v = (local >> 6) & 3, rest = local >> 8;  // rest = slab * tiles + nt
    const int nt = rest % tiles, slab = rest / tiles;
    const int i = lane & 31, hh = lane >> 5;
    const float4 val = *reinterpret_cast<const float4 *>(src + (i64)(nt * 32 + i) * K + slab * 32 + 16 * hh + 4 * v);
    reinterpret_cast<float4 *>(out)[q] = val;
    if (out2) reinterpret_cast<float4 *>(out2)[q] = val;
}

// one 32 (pixels) x 32 (channels) tile of a convolution whose input sits in LDS (pixel-major, `stride` floats per pixel):
//   acc += sum over slabs of A[pixel rows][32 k] * W[32 channel rows][32 k]^T,  slab = (tap, 32-channel group)
// `abase[t]` = LDS float index of this lane's row for tap t (clamped pixel * stride + 16 h); `wpk` = packed filters of the layer
// + (nt * 4 * 64 + lane) * 4: the lane's float4 of (slab 0, v = 0); slabs are 2 * 4 * 64 * 4 floats apart, v's 64 * 4.
template <int TAPS, int CG>  // CG = 32-channel groups per tap (1 for C = 32, 2 for C = 64)
__device__ __forceinline__ void tile_from_lds(const float *__restrict__ lds_in, const int (&abase)[TAPS], const float *__restrict__ wpk, f32x16 &acc) {
    constexpr int S = TAPS * CG;
    static_assert(S % 2 == 0, "the slab loop is unrolled by two (ping-pong fragment registers)");
    float4 a0[4], b0[4], a1[4], b1[4];
    auto load = [&](int sl, float4(&A)[4], float4(&B)[4]) {
        const float *pa = lds_in + abase[sl / CG] + (sl % CG) * 32;
        const float *pb = wpk + sl * (2 * 4 * 64 * 4);
#pragma unroll
        for (int v = 0; v < 4; v++) B[v] = *reinterpret_cast<const float4 *>(pb + v * 256);  // filters: L2 -> registers (the long latency first)
#pragma unroll
        for (int v = 0; v < 4; v++) A[v] = *reinterpret_cast<const float4 *>(pa + 4 * v);  // activations: LDS -> registers
    };
    auto mfma16 = [&](const float4(&A)[4], const float4(&B)[4]) {
#pragma unroll
        for (int v = 0; v < 4; v++) {
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(A[v].x, B[v].x, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(A[v].y, B[v].y, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(A[v].z, B[v].z, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(A[v].w, B[v].w, acc, 0, 0, 0);
        }
    };
    // software pipeline, one slab ahead: the fragments of slab s+1 are requested BEFORE the 16 MFMAs of slab s.  hipcc's scheduler
    // otherwise sinks every load next to its first use (one dwordx4 + s_waitcnt vmcnt(0) per 4 MFMAs: the matrix pipe then idles
    // for an L2 round trip per quarter slab); the scheduling barriers pin the order, the waitcnts stay the compiler's.
    load(0, a0, b0);
#pragma unroll
    for (int sl = 0; sl < S; sl += 2) {
        load(sl + 1, a1, b1);
        __builtin_amdgcn_sched_barrier(0);
        mfma16(a0, b0);
        __builtin_amdgcn_sched_barrier(0);
        if (sl + 2 < S) load(sl + 2, a0, b0);
        __builtin_amdgcn_sched_barrier(0);
        mfma16(a1, b1);
        __builtin_amdgcn_sched_barrier(0);
    }
}

// conv2 / conv3 on the bf16 matrix pipe (float32 x float32 from six exact partial products, like k_gemm_b16 of srlx_qnet.hip).  With one 32 x 32
// tile per wave an MFMA needs 2 KB of operands -- 256 B per clock per CU at the bf16 pipe's rate, twice what LDS delivers and four times what L2
// does -- so here a wave owns a 64 x 64 output BLOCK (2 pixel tiles x both channel tiles: four accumulators, every fragment used twice) over a
// QUARTER of K; the eight waves are 2 blocks x 4 K quarters and the quarters are summed through LDS afterwards.
// Both operands arrive PRE-SPLIT: the filters by k_pack_filters, the activations by the epilogue of the layer that produced them (act1 / act2 live in LDS
// as three bf16 part planes per pixel: [part][channel], kPB1 / kPB2 bytes per pixel).  Splitting the A fragments inside this loop cost 88 vector
// instructions per 24 MFMAs, and on gfx950 vector instructions take their issue slots from the matrix pipe (tools/mfma_peak.hip): the loop ran at 60 % of its
// MFMA time.  Now a K step is 6 global_load_dwordx4 + 6 ds_read_b128 + 24 MFMAs and the address arithmetic of one tap.
//   LAYER 2: 4 x 4 stride 2 pad 2 over act1 (21 x 21 x 32), K = 16 taps x 32 = 32 steps of 16;  LAYER 3: 3 x 3 stride 1 pad 1 over act2 (11 x 11 x 64), 36 steps
// The MFMA operands are SWAPPED (filters as the A operand), so each accumulator tile comes out transposed -- rows = channels, columns = pixels: a lane
// holds runs of four consecutive channels of ONE pixel, i.e. 8-byte pieces of the next layer's part planes (and float4s of the float32 tensors).
template <int LAYER, bool H16>
__device__ __forceinline__ void block_planes(const unsigned char *__restrict__ lds_in, const bf16x8 *__restrict__ wfrag, int blk, int kq, int lane, f32x16 (&acc)[2][2]) {
    constexpr int S = LAYER == 2 ? 32 : 36, SPT = LAYER == 2 ? 2 : 4, KW = LAYER == 2 ? 4 : 3, STR = LAYER == 2 ? 2 : 1, PAD = LAYER == 2 ? 2 : 1;
    constexpr int IN = LAYER == 2 ? kP1 : kP2, PB = LAYER == 2 ? (H16 ? kPB1h : kPB1) : (H16 ? kPB2h : kPB2), CB = LAYER == 2 ? 64 : 128, QS = S / 4;
    constexpr int NP = H16 ? 2 : 3;  // parts per value
    const int i = lane & 31, h = lane >> 5;
    int oy[2], ox[2];
#pragma unroll
    for (int a = 0; a < 2; a++) {
        const int m = (2 * blk + a) * 32 + i < kM2 ? (2 * blk + a) * 32 + i : kM2 - 1;
        oy[a] = m / kP2, ox[a] = m % kP2;
    }
    f32x16 lo[H16 ? 2 : 1][H16 ? 2 : 1];  // H16: the 2^11-scaled sums (hi * lo' + lo * hi')
#pragma unroll
    for (int a = 0; a < 2; a++)
#pragma unroll
        for (int n = 0; n < 2; n++)
#pragma unroll
            for (int r = 0; r < 16; r++) {
                acc[a][n][r] = 0.f;
                if constexpr (H16) lo[a][n][r] = 0.f;
            }
    bf16x8 fa0[2][NP], fa1[2][NP];  // [pixel tile][part]  (H16: sixteen bytes of f16 in the same carrier type)
    bf16x8 fb0[2][NP], fb1[2][NP];  // [channel tile][part]
    auto load = [&](int sl, bf16x8(&fa)[2][NP], bf16x8(&fb)[2][NP]) __attribute__((always_inline)) {
        // which K steps a quarter owns: conv2 -- kernel row kq (8 consecutive steps); conv3 -- the 16-channel group kq of every tap, so that the tap of
        // step sl is a compile-time constant and the clamped pixel address costs a handful of instructions (a quarter of consecutive steps: 355 VALU per pass)
        const int sp = LAYER == 2 ? kq * QS + sl : sl * SPT + kq, tap = LAYER == 2 ? sp / SPT : sl, cg = LAYER == 2 ? sp % SPT : kq, ky = tap / KW, kx = tap % KW;
#pragma unroll
        for (int n = 0; n < 2; n++)
#pragma unroll
            for (int q = 0; q < NP; q++) fb[n][q] = wfrag[((sp * 2 + n) * NP + q) * 64 + lane];  // filters: L2 -> registers (the long latency first)
#pragma unroll
        for (int a = 0; a < 2; a++) {
            const int iy = clampi(oy[a] * STR + ky - PAD, 0, IN - 1), ix = clampi(ox[a] * STR + kx - PAD, 0, IN - 1);
            const unsigned char *pa = lds_in + (iy * IN + ix) * PB + cg * 32 + 16 * h;
#pragma unroll
            for (int q = 0; q < NP; q++) fa[a][q] = *reinterpret_cast<const bf16x8 *>(pa + q * CB);
        }
    };
    auto mfma_step = [&](const bf16x8(&fa)[2][NP], const bf16x8(&fb)[2][NP]) __attribute__((always_inline)) {
        if constexpr (H16) {
            constexpr int pq[3][2] = {{1, 0}, {0, 1}, {0, 0}};  // (activation part, filter part): the two cross terms into `lo`, hi * hi' into `acc`
#pragma unroll
            for (int c = 0; c < 3; c++)
#pragma unroll
                for (int a = 0; a < 2; a++)
#pragma unroll
                    for (int n = 0; n < 2; n++) {
                        f32x16 &d = c < 2 ? lo[a][n] : acc[a][n];
                        d = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, fb[n][pq[c][1]]), __builtin_bit_cast(f16x8, fa[a][pq[c][0]]), d, 0, 0, 0);
                    }
        } else {
            constexpr int pq[6][2] = {{2, 0}, {0, 2}, {1, 1}, {1, 0}, {0, 1}, {0, 0}};  // smallest partial products first
#pragma unroll
            for (int c = 0; c < 6; c++)
#pragma unroll
                for (int a = 0; a < 2; a++)
#pragma unroll
                    for (int n = 0; n < 2; n++)
                        acc[a][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[n][pq[c][1]], fa[a][pq[c][0]], acc[a][n], 0, 0, 0);
        }
    };
    load(0, fa0, fb0);
#pragma unroll
    for (int sl = 0; sl < QS; sl += 2) {
        if (sl + 1 < QS) load(sl + 1, fa1, fb1);
        __builtin_amdgcn_sched_barrier(0);
        mfma_step(fa0, fb0);
        __builtin_amdgcn_sched_barrier(0);
        if (sl + 2 < QS) load(sl + 2, fa0, fb0);
        __builtin_amdgcn_sched_barrier(0);
        if (sl + 1 < QS) mfma_step(fa1, fb1);
        __builtin_amdgcn_sched_barrier(0);
    }
    if constexpr (H16) {  // this K quarter's sum = the unscaled part + the scaled part / 2048
#pragma unroll
        for (int a = 0; a < 2; a++)
#pragma unroll
            for (int n = 0; n < 2; n++)
#pragma unroll
                for (int r = 0; r < 16; r++) acc[a][n][r] = __builtin_fmaf(lo[a][n][r], kLoInv, acc[a][n][r]);
    }
}

// bias + ReLU on four consecutive channels of one pixel (one quarter g of a transposed accumulator tile), then the exact three-way bf16 split
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
__device__ __forceinline__ float4 bias_relu4(const f32x16 &acc, int g, const float4 &bias) {
    float4 v = make_float4(acc[4 * g] + bias.x, acc[4 * g + 1] + bias.y, acc[4 * g + 2] + bias.z, acc[4 * g + 3] + bias.w);
    v.x = v.x > 0.f ? v.x : 0.f, v.y = v.y > 0.f ? v.y : 0.f, v.z = v.z > 0.f ? v.z : 0.f, v.w = v.w > 0.f ? v.w : 0.f;
    return v;
}
__device__ __forceinline__ void split3(const float4 &x, bf16x4 (&part)[3]) {
    float v[4] = {x.x, x.y, x.z, x.w};
#pragma unroll
    for (int p = 0; p < 3; p++)
#pragma unroll
        for (int e = 0; e < 4; e++) {
            const __bf16 q = (__bf16)v[e];
            part[p][e] = q;
            v[e] -= (float)q;
        }
}

// the two-part float16 split of four values: hi = f16(x), lo = f16((x - hi) * 2048)
__device__ __forceinline__ void split2(const float4 &x, f16x4 (&part)[2]) {
    const float v[4] = {x.x, x.y, x.z, x.w};
#pragma unroll
    for (int e = 0; e < 4; e++) {
        const _Float16 hi = (_Float16)v[e];
        part[0][e] = hi;
        part[1][e] = (_Float16)((v[e] - (float)hi) * kLoScale);
    }
}

// Sums the four K-quarter partial blocks of every output tile through `scratch` (96 KB of LDS nobody else uses meanwhile).  Wave (blk, kq) ends up
// OWNING tile kq = (pixel tile kq >> 1, channel tile kq & 1) of its block: every wave parks its partials of the three tiles it does not own (twelve
// ds_write_b128), ONE barrier, the owner adds the four partials in K order (twelve ds_read_b128); the epilogues then run on all eight waves at once.
// (Round by round through 24 KB -- four rounds of two barriers -- this took 5.4 k clocks per layer, a tenth of the kernel.)
constexpr size_t kScratchBytes = 2 * 4 * 3 * 4 * 64 * 16;
__device__ __forceinline__ f32x16 reduce_quarters(float *__restrict__ scratch, int blk, int kq, int lane, f32x16 (&acc)[2][2]) {
    float4 *sc = reinterpret_cast<float4 *>(scratch);
#pragma unroll
    for (int t = 0; t < 4; t++) {
        const f32x16 &part = acc[t >> 1][t & 1];
        if (kq != t) {
            const int slot = (blk * 4 + t) * 3 + (kq < t ? kq : kq - 1);
#pragma unroll
            for (int v = 0; v < 4; v++) sc[(slot * 4 + v) * 64 + lane] = make_float4(part[4 * v], part[4 * v + 1], part[4 * v + 2], part[4 * v + 3]);
        }
    }
    __syncthreads();
    f32x16 sum;
#pragma unroll
    for (int t = 0; t < 4; t++) {
        const f32x16 &mine = acc[t >> 1][t & 1];
        if (kq == t) {
#pragma unroll
            for (int v = 0; v < 4; v++) {
                float4 part[4];
#pragma unroll
                for (int k = 0; k < 4; k++)
                    part[k] = k == t ? make_float4(mine[4 * v], mine[4 * v + 1], mine[4 * v + 2], mine[4 * v + 3]) : sc[(((blk * 4 + t) * 3 + (k < t ? k : k - 1)) * 4 + v) * 64 + lane];
                sum[4 * v] = ((part[0].x + part[1].x) + part[2].x) + part[3].x;  // K quarters in order
                sum[4 * v + 1] = ((part[0].y + part[1].y) + part[2].y) + part[3].y;
                sum[4 * v + 2] = ((part[0].z + part[1].z) + part[2].z) + part[3].z;
                sum[4 * v + 3] = ((part[0].w + part[1].w) + part[2].w) + part[3].w;
            }
        }
    }
    return sum;
}

// BIG = a launch of at least 512 samples (the actors' policy pass): a template parameter only so that profiles list the chip-filling
// launches and the learner's 96 / 128-sample launches as separate kernels (same code).
// PLANES (chip-filling launches of a handle with operand planes, srlx_fc1_planes.hip): act3 is written as the three bf16 part planes the first dense
// layer's GEMM reads -- [K/32 slabs][batch rows][4 k-groups][3 parts][8 bf16], K = pixel * 64 + channel -- instead of float32; `act3` then points at them.
// PLANES = 2 (round 4, the learner's passes): BOTH -- float32 act3 for the backward pass and the planes (at `planes_out`, `plane_rows` rows per K-slab: the launch's
// rows rounded up to the GEMM's 128-row tile) for the first dense layer.
template <bool BIG, bool C1B16, bool C23B16, int PLANES = 0, bool H16 = false>
__device__ __forceinline__ void convnet_fused_body(const u8 *__restrict__ base, const i64 *__restrict__ frame_off, const float *__restrict__ wpk,
                                                   const float *__restrict__ b1, const float *__restrict__ b2, const float *__restrict__ b3, float *__restrict__ act3,
                                                   float *__restrict__ act1_out, float *__restrict__ act2_out, unsigned long long *__restrict__ dbg,
                                                   unsigned char *__restrict__ planes_out, long long plane_rows, long long n_samples, i64 b, bool stamp_wg,
                                                   int *__restrict__ range_flag) {
    static_assert(!PLANES || C23B16, "operand planes come out of the split-bf16 conv3 only");
    static_assert(!C23B16 || C1B16, "the split-bf16 conv2 / conv3 read the part planes conv1's split-bf16 epilogue writes");
    static_assert(!H16 || C23B16, "the float16 split is a variant of the part-plane kernel");
    constexpr int PB1 = H16 ? kPB1h : kPB1, PB2 = H16 ? kPB2h : kPB2;
    constexpr bool PL = C23B16;  // act1 / act2 as bf16 part planes in LDS (kPB1 / kPB2 bytes per pixel) instead of float32
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    u8 *fr = smem;                                                 // [4][88][88]
    float *a1 = reinterpret_cast<float *>(smem + 4 * kFrame);      // [441][36]            (float32 layout)
    float *a2 = a1 + kM1 * kS1;                                    // [121][68]
    unsigned char *a1p = smem + kOffA1P, *a2p = smem;              // [441][kPB1], [121][kPB2] (plane layout; act2 overlays the frames and conv1's filters)
    int t = threadIdx.x;
    const int wave = t >> 6;
    constexpr int H = 84, W = 84, NT = 64 * kWaves;
    auto stamp = [&](int k) {  // phase timestamps of every wave of workgroup 0 (tools/fused_phases.py); dbg is NULL in production
        if (dbg && stamp_wg && (threadIdx.x & 63) == 0) dbg[wave * 8 + k] = __builtin_amdgcn_s_memtime();
    };
    const int lane = t & 63, h = lane >> 5, i = lane & 31;
    stamp(0);

    // ---- stage the four frames (uint8, replicate padding materialised): padded dword (row r, dword d) covers image columns
    //      clamp(4d - 3 .. 4d), image row clamp(r - 3); all loads of all frames are in flight before the first LDS store
    //      A lane owns the SAME four padded dwords (t + 512 j) of every frame, so the clamped source offset and the border byte pattern are computed
    //      once per lane and dword -- the address arithmetic used to be 1000 VALU instructions per wave (8 k of the kernel's 70 k clocks per sample:
    //      vector instructions take their issue slots from the matrix pipe on gfx950); (row, dword) advance by 512 = 23 x 22 + 6 per j.
    // conv1's filter fragments do not depend on the sample: requested first, they travel during the frame-offset round trip (see the conv1 section)
    bf16x8 bw2[C1B16 && !H16 ? 16 : 1], wtmp[C1B16 ? 4 : 1];
    if constexpr (C1B16) {
        const bf16x8 *wsrc = reinterpret_cast<const bf16x8 *>(wpk + kW1 + kW2 + kW3);
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const int e = t + NT * j, l = e & 63, sp2 = e >> 6;  // sp2 = step * 2 + part
            wtmp[j] = H16 ? wsrc[e] : wsrc[((sp2 >> 1) * 3 + (sp2 & 1)) * 64 + l];  // (H16: [step][2 parts][lane], both parts go to LDS)
        }
        if constexpr (!H16) {
#pragma unroll
            for (int sp = 0; sp < 16; sp++) bw2[sp] = wsrc[(sp * 3 + 2) * 64 + lane];
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    {
        constexpr int kDw = kPad * (kPad / 4);              // 1936 dwords per frame
        constexpr int kCols = kPad / 4;                      // 22 dwords per padded row
        constexpr int kPer = (kDw + NT - 1) / NT;            // 4 dwords per lane and frame (the last one for t < 400 only)
        static_assert(NT == 23 * kCols + 6, "the (row, dword) stepping below assumes 512 lanes over 22-dword rows");
        unsigned v[4][kPer], goff[kPer], sel[kPer];
        i64 o[4];
#pragma unroll
        for (int q = 0; q < 4; q++) o[q] = frame_off[b * 4 + q];  // one round trip, then every frame load is independent
        int r = t / kCols, c = t % kCols;
#pragma unroll
        for (int j = 0; j < kPer; j++) {
            goff[j] = (unsigned)(clampi(r - 3, 0, H - 1) * W + clampi(4 * c - 3, 0, W - 4));
            // border dwords: column clamp(x + p) sits at byte clamp(x + p) - xs of the loaded dword (x = 4 c - 3, xs = clamp(x)): left edge = byte 0 four
            // times, right edge (c = 21: columns 81 82 83 83 out of the dword at 80) = bytes 1 2 3 3; v_perm_b32 selectors, zero frames stay zero
            sel[j] = c == 0 ? 0x00000000u : (c == kCols - 1 ? 0x03030201u : 0x03020100u);
            c += 6, r += 23;
            if (c >= kCols) c -= kCols, r += 1;
        }
#pragma unroll
        for (int q = 0; q < 4; q++)
#pragma unroll
            for (int j = 0; j < kPer; j++) {
                v[q][j] = 0u;
                if ((j + 1 < kPer || t + NT * j < kDw) && o[q] >= 0) __builtin_memcpy(&v[q][j], base + o[q] + goff[j], 4);
            }
#pragma unroll
        for (int q = 0; q < 4; q++)
#pragma unroll
            for (int j = 0; j < kPer; j++)
                if (j + 1 < kPer || t + NT * j < kDw) reinterpret_cast<unsigned *>(fr)[q * kDw + t + NT * j] = __builtin_amdgcn_perm(0u, v[q][j], sel[j]);
    }
    // ---- conv1 on the bf16 matrix pipe, exactly (see k_pack_filters): the A operand is the uint8 pixel itself (one bf16), the filter / 255 is
    //      the sum of three bf16 parts, v_mfma_f32_32x32x16_bf16 accumulates the exact partial products in float32.  3 MFMAs of 32 cycles per
    //      16 K instead of 8 of 64: the layer's matrix time drops 5.3x; what remains is the LDS / conversion stream (two dwords and twelve
    //      VALU instructions per step).  Step s = (frame c = s >> 2, kernel rows 2 (s & 3) + h), a lane's 8 k = the 8 kernel columns.
    if constexpr (C1B16) {
        // Filter fragments: every wave needs all 48 KB of them.  Fetched per wave they cost the CU's load path 384 KB per sample (4.8 k clocks of the
        // staging phase).  The act2 region of LDS is idle until conv2: the workgroup fetches parts 0 and 1 of all 16 steps ONCE into it (32 KB,
        // [step][part][lane] x 16 B), only part 2 stays in registers (16 x 16 B per lane).
        bf16x8 *wl = reinterpret_cast<bf16x8 *>(smem + (PL ? (size_t)4 * kFrame : (size_t)4 * kFrame + (size_t)kM1 * kS1 * 4));
        static_assert(16 * 2 * 64 * 16 <= kM2 * kS2 * 4, "conv1's shared filter parts must fit into the act2 region");
#pragma unroll
        for (int j = 0; j < 4; j++) wl[t + NT * j] = wtmp[j];
        const float bias = b1[i];
        float4 bias4[4];  // plane layout: the tile comes out transposed (rows = channels), a lane's quarter g = channels 8 g + 4 h .. + 3
#pragma unroll
        for (int g = 0; g < 4; g++) bias4[g] = PL ? *reinterpret_cast<const float4 *>(b1 + 8 * g + 4 * h) : make_float4(0.f, 0.f, 0.f, 0.f);
        stamp(1);
        __syncthreads();
        stamp(2);
        constexpr int tiles = (kM1 + 31) / 32;  // 14
        const int wrot = (wave + (int)(b & 7)) & 7;  // rotate the 2/2/2/2/2/2/1/1 split with the sample index
        for (int tile = wrot; tile < tiles; tile += kWaves) {
            const int m = tile * 32 + i < kM1 ? tile * 32 + i : kM1 - 1;
            const int oy = m / kP1, ox = m % kP1;
            const u8 *win = fr + (4 * oy + h) * kPad + 4 * ox;
            f32x16 acc, acl;  // (H16: acl = the 2^11-scaled sum of pixel * filter lo part)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[r] = 0.f, acl[r] = 0.f;
            struct Step {
                unsigned x, y;   // the lane's 8 pixels
                bf16x8 f0, f1;   // filter parts 0 / 1 of the step (LDS)
            } wa, wb;
            auto fetch = [&](int sp, Step &w) __attribute__((always_inline)) {
                const u8 *p = win + (sp >> 2) * kFrame + 2 * (sp & 3) * kPad;
                w.x = *reinterpret_cast<const unsigned *>(p);
                w.y = *reinterpret_cast<const unsigned *>(p + 4);
                w.f0 = wl[(sp * 2) * 64 + lane];
                w.f1 = wl[(sp * 2 + 1) * 64 + lane];
            };
            auto mfma3 = [&](int sp, const Step &w) __attribute__((always_inline)) {
                if constexpr (H16) {  // the pixel is exact in ONE f16 as well; filter * 256 / 255 = hi + lo / 2048: two products per step
                    // a pixel byte n zero-extended to sixteen bits IS the float16 denormal n * 2^-24, and the matrix pipe keeps denormal inputs (tools/mfma_f16_denorm.hip):
                    // one v_perm_b32 per two pixels instead of two conversions per pixel; the epilogue's scale carries the 2^24
                    const unsigned pk[4] = {__builtin_amdgcn_perm(0u, w.x, 0x0c010c00u), __builtin_amdgcn_perm(0u, w.x, 0x0c030c02u),
                                            __builtin_amdgcn_perm(0u, w.y, 0x0c010c00u), __builtin_amdgcn_perm(0u, w.y, 0x0c030c02u)};
                    f16x8 a;
                    __builtin_memcpy(&a, pk, 16);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, w.f0), a, acc, 0, 0, 0);
                    acl = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, w.f1), a, acl, 0, 0, 0);
                    return;
                }
                bf16x8 a;
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    a[j] = (__bf16)(float)((w.x >> (8 * j)) & 255u);
                    a[4 + j] = (__bf16)(float)((w.y >> (8 * j)) & 255u);
                }
                if constexpr (PL) {  // operands swapped: the transposed tile (rows = channels, columns = pixels), same sums
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w.f0, a, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w.f1, a, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bw2[sp], a, acc, 0, 0, 0);
                } else {
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, w.f0, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, w.f1, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, bw2[sp], acc, 0, 0, 0);
                }
            };
            fetch(0, wa);
#pragma unroll
            for (int sp = 0; sp < 16; sp += 2) {
                fetch(sp + 1, wb);
                __builtin_amdgcn_sched_barrier(0);
                mfma3(sp, wa);
                __builtin_amdgcn_sched_barrier(0);
                if (sp + 2 < 16) fetch(sp + 2, wa);
                __builtin_amdgcn_sched_barrier(0);
                mfma3(sp + 1, wb);
                __builtin_amdgcn_sched_barrier(0);
            }
            if constexpr (PL) {
                // transposed C/D layout: col = lane & 31 = pixel, row = (r & 3) + 8 (r >> 2) + 4 h = channel: four consecutive channels of one pixel per quarter
                // g = r >> 2 -> bias, ReLU, the exact three-way bf16 split ONCE per activation, 8-byte pieces of the pixel's three part planes
                const int mm = tile * 32 + i;
                if constexpr (H16) {  // filters packed times 2^8 (their f16 parts stay normal), pixels times 2^-24: sum = (acc + acl / 2048) * 2^16, every factor a power of two
#pragma unroll
                    for (int r = 0; r < 16; r++) acc[r] = __builtin_fmaf(acl[r], 32.0f, acc[r] * 65536.0f);
                }
                if (mm < kM1) {
                    float top = 0.f;
#pragma unroll
                    for (int g = 0; g < 4; g++) {
                        const float4 v = bias_relu4(acc, g, bias4[g]);
                        if (act1_out) *reinterpret_cast<float4 *>(act1_out + (b * kM1 + mm) * 32 + 8 * g + 4 * h) = v;
                        if constexpr (H16) {
                            top = fmaxf(fmaxf(top, fmaxf(v.x, v.y)), fmaxf(v.z, v.w));
                            f16x4 part[2];
                            split2(v, part);
#pragma unroll
                            for (int q = 0; q < 2; q++) *reinterpret_cast<f16x4 *>(a1p + mm * PB1 + q * 64 + (8 * g + 4 * h) * 2) = part[q];
                        } else {
                            bf16x4 part[3];
                            split3(v, part);
#pragma unroll
                            for (int q = 0; q < 3; q++) *reinterpret_cast<bf16x4 *>(a1p + mm * kPB1 + q * 64 + (8 * g + 4 * h) * 2) = part[q];
                        }
                    }
                    if (H16 && !(top <= 65504.0f)) atomicOr(range_flag, 1);  // outside float16: loud on the host side (srlx_qnet_range_flags), never silent
                }
            } else {
                // C/D layout: col = lane & 31 (channel), row = (r & 3) + 8 (r >> 2) + 4 h (pixel)
#pragma unroll
                for (int r = 0; r < 16; r++) {
                    const int mm = tile * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                    if (mm < kM1) {
                        float v = acc[r] + bias;
                        v = v > 0.f ? v : 0.f;
                        a1[mm * kS1 + i] = v;
                        if (act1_out) act1_out[(b * kM1 + mm) * 32 + i] = v;
                    }
                }
            }
        }
    }
    // ---- conv1, float32 MFMAs (SRLX_CONV1_F32=1): this lane's B fragments of all eight K-slabs (filter row i, k = slab*32 + 16 h + 0..15), 1/255 folded in
    if constexpr (!C1B16) {
        float bfr[8][16];
        const float *wp = wpk + lane * 4;  // conv1's filters lead the packed buffer: [slab][v][lane][4]
#pragma unroll
        for (int sl = 0; sl < 8; sl++)
#pragma unroll
            for (int v = 0; v < 4; v++) {
                const float4 x = *reinterpret_cast<const float4 *>(wp + (sl * 4 + v) * 256);
                bfr[sl][4 * v] = x.x * (1.0f / 255.0f), bfr[sl][4 * v + 1] = x.y * (1.0f / 255.0f), bfr[sl][4 * v + 2] = x.z * (1.0f / 255.0f),
                bfr[sl][4 * v + 3] = x.w * (1.0f / 255.0f);
            }
        const float bias = b1[i];
        stamp(1);
        __syncthreads();
        stamp(2);
        constexpr int tiles = (kM1 + 31) / 32;  // 14
        const int wrot = (wave + (int)(b & 7)) & 7;  // rotate the 2/2/2/2/2/2/1/1 split with the sample index
        for (int tile = wrot; tile < tiles; tile += kWaves) {
            const int m = tile * 32 + i < kM1 ? tile * 32 + i : kM1 - 1;
            const int oy = m / kP1, ox = m % kP1;
            const u8 *win = fr + (4 * oy + 2 * h) * kPad + 4 * ox;
            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; r++) acc[r] = 0.f;
            // eight K-slabs (frame c, kernel rows kyb..kyb+3); the four dwords of slab g+1 are read from LDS before the 16 MFMAs of slab g
            unsigned wa[4], wb[4];
            auto fetch = [&](int g, unsigned(&w)[4]) {
                const u8 *p = win + (g >> 1) * kFrame + (g & 1) * 4 * kPad;
                w[0] = *reinterpret_cast<const unsigned *>(p);
                w[1] = *reinterpret_cast<const unsigned *>(p + 4);
                w[2] = *reinterpret_cast<const unsigned *>(p + kPad);
                w[3] = *reinterpret_cast<const unsigned *>(p + kPad + 4);
            };
            auto mfma16 = [&](int g, const unsigned(&w)[4]) {
#pragma unroll
                for (int s = 0; s < 16; s++) {
                    const float a = (float)((w[s >> 2] >> (8 * (s & 3))) & 255u);
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bfr[g][s], acc, 0, 0, 0);
                }
            };
            fetch(0, wa);
#pragma unroll
            for (int g = 0; g < 8; g += 2) {
                fetch(g + 1, wb);
                __builtin_amdgcn_sched_barrier(0);
                mfma16(g, wa);
                __builtin_amdgcn_sched_barrier(0);
                if (g + 2 < 8) fetch(g + 2, wa);
                __builtin_amdgcn_sched_barrier(0);
                mfma16(g + 1, wb);
                __builtin_amdgcn_sched_barrier(0);
            }
            // C/D layout: col = lane & 31 (channel), row = (r & 3) + 8 (r >> 2) + 4 h (pixel)
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const int mm = tile * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                if (mm < kM1) {
                    float v = acc[r] + bias;
                    v = v > 0.f ? v : 0.f;
                    a1[mm * kS1 + i] = v;
                    if (act1_out) act1_out[(b * kM1 + mm) * 32 + i] = v;
                }
            }
        }
    }
    stamp(3);
    __syncthreads();
    stamp(4);
    // conv3's output tile (pixel tile mt, channel tile nt): bias + ReLU, act3 to HBM
    auto store_act3 = [&](int mt, int nt, const f32x16 &acc) __attribute__((always_inline)) {
        const float bias = b3[nt * 32 + i];
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const int mm = mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
            if (mm < kM2) {
                const float v = acc[r] + bias;
                act3[(b * kM2 + mm) * 64 + nt * 32 + i] = v > 0.f ? v : 0.f;
            }
        }
    };
    if constexpr (C23B16) {
        const int blk = wave & 1, kq = wave >> 1;  // 64 x 64 output block (pixel tiles 2 blk, 2 blk + 1; both channel tiles) x K quarter
        f32x16 acc4[2][2];
        // the K-quarter partials park in the 96 KB behind the act2 planes: act1 (and the tail of conv1's filter region) is dead once conv2's block loops are done
        constexpr size_t kOffScratch = ((size_t)kM2 * PB2 + 1023) / 1024 * 1024;
        static_assert(kOffScratch + kScratchBytes <= (H16 ? kLdsH16 : kLdsBytes), "reduction scratch");
        float *scratch = reinterpret_cast<float *>(smem + kOffScratch);
        const bf16x8 *wf2 = reinterpret_cast<const bf16x8 *>(wpk + kW1 + kW2 + kW3 + kW1B), *wf3 = reinterpret_cast<const bf16x8 *>(wpk + kW1 + kW2 + kW3 + kW1B + kW2B);
        block_planes<2, H16>(a1p, wf2, blk, kq, lane, acc4);
        stamp(5);
        __syncthreads();  // every wave has read its last act1 fragment
        const int mt = 2 * blk + (kq >> 1), nt = kq & 1;  // the tile this wave owns after the reduction: channels nt * 32 + 8 g + 4 h + e of pixel mt * 32 + i
        const int pix = mt * 32 + i;
        {
            const f32x16 sum = reduce_quarters(scratch, blk, kq, lane, acc4);
            if (pix < kM2) {
                float top = 0.f;
#pragma unroll
                for (int g = 0; g < 4; g++) {
                    const float4 v = bias_relu4(sum, g, *reinterpret_cast<const float4 *>(b2 + nt * 32 + 8 * g + 4 * h));
                    if (act2_out) *reinterpret_cast<float4 *>(act2_out + (b * kM2 + pix) * 64 + nt * 32 + 8 * g + 4 * h) = v;
                    if constexpr (H16) {
                        top = fmaxf(fmaxf(top, fmaxf(v.x, v.y)), fmaxf(v.z, v.w));
                        f16x4 part[2];
                        split2(v, part);
#pragma unroll
                        for (int q = 0; q < 2; q++) *reinterpret_cast<f16x4 *>(a2p + pix * PB2 + q * 128 + (nt * 32 + 8 * g + 4 * h) * 2) = part[q];
                    } else {
                        bf16x4 part[3];
                        split3(v, part);
#pragma unroll
                        for (int q = 0; q < 3; q++) *reinterpret_cast<bf16x4 *>(a2p + pix * kPB2 + q * 128 + (nt * 32 + 8 * g + 4 * h) * 2) = part[q];
                    }
                }
                if (H16 && !(top <= 65504.0f)) atomicOr(range_flag, 2);
            }
        }
        __syncthreads();  // act2 is complete
        stamp(6);
        block_planes<3, H16>(a2p, wf3, blk, kq, lane, acc4);
        const f32x16 sum = reduce_quarters(scratch, blk, kq, lane, acc4);
        if constexpr (PLANES != 0) {
            // The operand planes of one sample are 242 chunks of 128 bytes (K-slab = (pixel, channel half): [4 k-groups][2 float16 parts][8]; srlx_fc1_planes.hip) at a
            // stride of rows * 128 bytes.  Written from the accumulator layout they would be 8-byte pieces, every lane of a store instruction in a line of its own (that
            // cost ~6 k clocks of the CU's store path per sample with the three-part planes of rounds 3-5).  The tile's parts are parked in LDS ([channel half][k-group]
            // [part][pixel] x 16 bytes: conflict-free for the writers; 129-pixel stride: at most three-way conflicts for the readers) and leave as 16-byte pieces, eight
            // consecutive lanes per chunk = one full line.  These planes are ALWAYS the two-part float16 split (the first dense layer has no bf16 variant any more):
            // an activation above 65 504 sets bit 2 of the handle's range word.
            constexpr int kStg = 129 * 16;
            static_assert(16 * kStg <= kOffScratch + kScratchBytes, "plane staging");
            unsigned char *stg = smem;
            __syncthreads();  // every owner has read its partial sums: the scratch (and the act2 planes in front of it) may be overwritten
            if (pix < kM2) {
                float top = 0.f;
#pragma unroll
                for (int g = 0; g < 4; g++) {
                    const float4 v = bias_relu4(sum, g, *reinterpret_cast<const float4 *>(b3 + nt * 32 + 8 * g + 4 * h));
                    if constexpr (PLANES == 2) *reinterpret_cast<float4 *>(act3 + (b * kM2 + pix) * 64 + nt * 32 + 8 * g + 4 * h) = v;
                    top = fmaxf(fmaxf(top, fmaxf(v.x, v.y)), fmaxf(v.z, v.w));
                    f16x4 part[2];
                    split2(v, part);
#pragma unroll
                    for (int q = 0; q < 2; q++) *reinterpret_cast<f16x4 *>(stg + ((nt * 4 + g) * 2 + q) * kStg + pix * 16 + h * 8) = part[q];
                }
                if (!(top <= 65504.0f)) atomicOr(range_flag, 4);
            }
            __syncthreads();
            // chunk piece c = slab * 8 + (k-group * 2 + part), slab = pixel * 2 + channel half: destination ((slab * rows + b) * 8 + k-group * 2 + part) * 16
            const i64 rows = PLANES == 2 ? (i64)plane_rows : (i64)n_samples;
            unsigned char *dst = (PLANES == 2 ? planes_out : reinterpret_cast<unsigned char *>(act3)) + b * 128;
            constexpr int kPieces = 2 * kM2 * 8;
#pragma unroll
            for (int k = 0; k < (kPieces + NT - 1) / NT; k++) {
                const int c = t + NT * k;
                if (c < kPieces) {
                    const int sl = c >> 3, w = c & 7;
                    const uint4 val = *reinterpret_cast<const uint4 *>(stg + ((sl & 1) * 8 + w) * kStg + (sl >> 1) * 16);
                    *reinterpret_cast<uint4 *>(dst + (i64)sl * rows * 128 + w * 16) = val;
                }
            }
        } else if (pix < kM2) {
#pragma unroll
            for (int g = 0; g < 4; g++)
                *reinterpret_cast<float4 *>(act3 + (b * kM2 + pix) * 64 + nt * 32 + 8 * g + 4 * h) =
                    bias_relu4(sum, g, *reinterpret_cast<const float4 *>(b3 + nt * 32 + 8 * g + 4 * h));
        }
    } else {
        const int mt = wave >> 1, nt = wave & 1;  // conv2 / conv3: one 32 x 32 output tile per wave
        const int m = mt * 32 + i < kM2 ? mt * 32 + i : kM2 - 1;
        const int oy = m / kP2, ox = m % kP2;
        // ---- conv2: 4 x 4 stride 2 pad 2 over act1 (21 x 21 x 32), K = 16 taps x 32
        {
            int ab[16];
    #pragma unroll
            for (int tp = 0; tp < 16; tp++) {
                const int iy = clampi(2 * oy - 2 + (tp >> 2), 0, kP1 - 1), ix = clampi(2 * ox - 2 + (tp & 3), 0, kP1 - 1);
                ab[tp] = (iy * kP1 + ix) * kS1 + 16 * h;
            }
            f32x16 acc;
    #pragma unroll
            for (int r = 0; r < 16; r++) acc[r] = 0.f;
            tile_from_lds<16, 1>(a1, ab, wpk + kW1 + (nt * 4 * 64 + lane) * 4, acc);
            const float bias = b2[nt * 32 + i];
    #pragma unroll
            for (int r = 0; r < 16; r++) {
                const int mm = mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                if (mm < kM2) {
                    float v = acc[r] + bias;
                    v = v > 0.f ? v : 0.f;
                    a2[mm * kS2 + nt * 32 + i] = v;
                    if (act2_out) act2_out[(b * kM2 + mm) * 64 + nt * 32 + i] = v;
                }
            }
        }
        stamp(5);
        __syncthreads();
        stamp(6);
        // ---- conv3: 3 x 3 stride 1 pad 1 over act2 (11 x 11 x 64), K = 9 taps x 64
        {
            int ab[9];
    #pragma unroll
            for (int tp = 0; tp < 9; tp++) {
                const int iy = clampi(oy - 1 + tp / 3, 0, kP2 - 1), ix = clampi(ox - 1 + tp % 3, 0, kP2 - 1);
                ab[tp] = (iy * kP2 + ix) * kS2 + 16 * h;
            }
            f32x16 acc;
    #pragma unroll
            for (int r = 0; r < 16; r++) acc[r] = 0.f;
            tile_from_lds<9, 2>(a2, ab, wpk + kW1 + kW2 + (nt * 4 * 64 + lane) * 4, acc);
            store_act3(mt, nt, acc);
        }
    }
    stamp(7);
}

template <bool BIG, bool C1B16, bool C23B16, int PLANES = 0, bool H16 = false>
__global__ void __launch_bounds__(64 * kWaves) k_convnet_fused(const u8 *__restrict__ base, const i64 *__restrict__ frame_off, const float *__restrict__ wpk,
                                                                const float *__restrict__ b1, const float *__restrict__ b2, const float *__restrict__ b3,
                                                                float *__restrict__ act3,
                                                                float *__restrict__ act1_out, float *__restrict__ act2_out, unsigned long long *__restrict__ dbg,
                                                                unsigned char *__restrict__ planes_out = nullptr, long long plane_rows = 0, long long n_samples = 0,
                                                                long long first_sample = 0, int *__restrict__ range_flag = nullptr) {
    // (a chip-filling pass may come as several launches of consecutive samples: first_sample)
    convnet_fused_body<BIG, C1B16, C23B16, PLANES, H16>(base, frame_off, wpk, b1, b2, b3, act3, act1_out, act2_out, dbg, planes_out, plane_rows, n_samples,
                                                   (i64)blockIdx.x + first_sample, blockIdx.x == 0, range_flag);
}

// Several networks' image blocks over the SAME frames as ONE launch (round 6: Agent57_light's five networks all evaluate the state the ring commit has just made
// current -- two UVFA Q-networks for the next policy step, the embedding network and the two RND networks for the intrinsic reward, agent57_light.py:355-391): workgroup
// (net, sample) = (blockIdx.x / n_samples, blockIdx.x % n_samples) runs the chip-filling PLANES body on net's packed filters and writes net's operand planes.  Five
// launches of 4 workgroup rounds each pay five ramps and five tails (a third of their time at E = 1024); one launch of 20 rounds pays one.
constexpr int kMultiMax = 8;
struct MultiNets {
    const float *wpk[kMultiMax], *b1[kMultiMax], *b2[kMultiMax], *b3[kMultiMax];
    float *act3[kMultiMax];
    int *flag[kMultiMax];
};
template <bool H16>
__global__ void __launch_bounds__(64 * kWaves) k_convnet_fused_multi(const u8 *__restrict__ base, const i64 *__restrict__ frame_off, MultiNets nets, long long n_samples) {
    const int net = (int)(blockIdx.x / (unsigned)n_samples);
    const i64 b = (i64)(blockIdx.x % (unsigned)n_samples);
    convnet_fused_body<true, true, true, 1, H16>(base, frame_off, nets.wpk[net], nets.b1[net], nets.b2[net], nets.b3[net], nets.act3[net], nullptr, nullptr, nullptr, nullptr, 0,
                                                 n_samples, b, false, nets.flag[net]);
}

}  // namespace

// SRLX_CONV_BF16X3=1: the convolutions' matrix-pipe products as six of the nine products of three bf16 parts (rounds 3-5) instead of three products of two float16
// parts (A/B switch, and the path for networks whose activations leave float16's range); read once per process
bool srlx_conv_h16() {
    static const bool bf16x3 = (getenv("SRLX_CONV_BF16X3") && getenv("SRLX_CONV_BF16X3")[0] == '1') ||
                               (getenv("SRLX_CONV23_F32") && getenv("SRLX_CONV23_F32")[0] == '1');  // (that A/B variant's conv1 reads the bf16 fragments)
    return !bf16x3;
}

size_t srlx_qnet_pack_bytes() { return (size_t)kPackFloats * sizeof(float); }

// k_pack_filters over `src`'s bound filters into its own packed buffer (+ the transposed filters of a training handle) and, with `dst_set`, a second copy
// into an actor set together with the small vectors (layout *L)
int srlx_qnet_pack_publish(srlx_qnet *src, srlx_qnet::ActorSet *dst_set, const srlx_small_layout *L, hipStream_t st, int64_t *bump, bool adam_small) {
    float *&own = src->aset_cur >= 0 ? src->wpack_own : src->wpack;
    if (!own) SRLX_HIP(hipMalloc((void **)&own, (size_t)kPackFloats * sizeof(float)));
    const bool keep = src->max_train > 0;
    const float *const *b = src->aset_cur >= 0 ? src->bound : nullptr;  // a handle reading a set still packs its BOUND weights
    const float *w1 = b ? b[0] : src->w1, *w2 = b ? b[2] : src->w2, *w3 = b ? b[4] : src->w3;
    int pack_threads = (kW1 + kW2 + kW3) / 4 + 16 * 64 + (32 + 36) * 2 * 64 + (keep ? kW3 + kW2 : 0);
    SmallCopy sm{};
    sm.h16 = srlx_conv_h16() ? 1 : 0;
    sm.bump = (long long *)bump;
    const bool adam = adam_small && src->rest_on && src->rest_armed;
    srlx_small_layout own_layout;
    if (adam && !L) {  // no set to publish into: the small vectors still get their optimiser step (any dense layout of the eight will do)
        own_layout = srlx_small_offsets(src);
        L = &own_layout;
    }
    if (dst_set || adam) {
        const float *v[kSmallVecs] = {b ? b[1] : src->b1, b ? b[3] : src->b2, b ? b[5] : src->b3, b ? b[7] : src->bf, b ? b[8] : src->v2w, b ? b[9] : src->v2b,
                                      b ? b[10] : src->a2w, b ? b[11] : src->a2b, src->uvfa.wx_bound};
        const int off[kSmallVecs + 1] = {L->b1, L->b2, L->b3, L->bf, L->v2w, L->v2b, L->a2w, L->a2b, L->wx, L->total};
        const int len[kSmallVecs] = {src->F1, 2 * src->F1, 2 * src->F1, 2 * src->hidden, src->hidden, 1, src->A * src->hidden, src->A, src->uvfa.X * 2 * src->hidden};
        for (int k = 0; k < kSmallVecs; k++) sm.src[k] = v[k], sm.len[k] = len[k];
        for (int k = 0; k <= kSmallVecs; k++) sm.off[k] = off[k];
        sm.dst = dst_set ? dst_set->small : nullptr;
        sm.first = ((pack_threads + 255) / 256) * 256;
        pack_threads = sm.first + L->total;
        if (adam) {  // the convolution biases (vectors 0..2) took their step in k_reduce_parts; 3..7 = gradient list entries 7..11
            for (int k = 3; k < 8; k++) sm.g[k] = src->rest_g[k + 4], sm.m[k] = src->rest_m[k + 4], sm.v[k] = src->rest_v[k + 4], sm.src[k] = src->bound[k + 4];
            if (src->uvfa.X > 0 && src->uvfa.m_wx) sm.g[8] = src->uvfa.g_wx, sm.m[8] = src->uvfa.m_wx, sm.v[8] = src->uvfa.v_wx;  // the UVFA columns' step (srlx_qnet_fuse_adam_uvfa)
            sm.lr = src->adam_lr, sm.beta1 = src->adam_b1, sm.beta2 = src->adam_b2, sm.eps = src->adam_eps;
            sm.snap = (const long long *)src->step_snap;
            src->rest_armed = false;
        }
    }
    hipLaunchKernelGGL(k_pack_filters, dim3((pack_threads + 255) / 256), dim3(256), 0, st, w1, w2, w3, own, keep ? src->w_t : nullptr, keep ? src->w_t2 : nullptr,
                       dst_set ? dst_set->wpack : nullptr, sm);
    SRLX_HIP(hipGetLastError());
    return SRLX_OK;
}

// Launcher: true when the fused kernel covers this handle's geometry (then act3 -- and act1 / act2 when training is enabled -- are valid).
bool srlx_qnet_fused_convs(srlx_qnet *h, int64_t batch, const uint8_t *d_frame_base, const int64_t *d_frame_off, hipStream_t st) {
    if (!(h->H == 84 && h->W == 84 && h->Wn == 4 && h->F1 == 32)) return false;
    static bool attr_set = false;
    if (!attr_set) {
        const void *kerns[] = {(const void *)k_convnet_fused<true, true, true, 1>, (const void *)k_convnet_fused<false, true, true, 2>,
                               (const void *)k_convnet_fused<true, true, true>,  (const void *)k_convnet_fused<false, true, true>,
                               (const void *)k_convnet_fused<true, true, true, 1, true>, (const void *)k_convnet_fused<false, true, true, 2, true>,
                               (const void *)k_convnet_fused<true, true, true, 0, true>,  (const void *)k_convnet_fused<false, true, true, 0, true>,
                               (const void *)k_convnet_fused<true, true, false>, (const void *)k_convnet_fused<false, true, false>,
                               (const void *)k_convnet_fused<true, false, false>, (const void *)k_convnet_fused<false, false, false>};
        for (const void *kp : kerns)
            if (hipFuncSetAttribute(kp, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLdsBytes) != hipSuccess) return false;
        attr_set = true;
    }
    const bool keep = h->max_train > 0;  // a training handle: the backward pass reads act1 / act2 and the transposed filters
    if (!h->pack_valid) {  // (valid: packed right after the last optimiser step by srlx_qnet_publish, a selected actor set, or a sticky pack of unchanged weights)
        if (srlx_qnet_pack_publish(h, nullptr, nullptr, st) != SRLX_OK) return false;
        h->pack_valid = h->pack_sticky;
    }
    h->wt_from_forward = keep;
    static const bool c1_f32 = getenv("SRLX_CONV1_F32") && getenv("SRLX_CONV1_F32")[0] == '1';  // A/B switch: conv1 on the float32 matrix pipe
    float *out3 = h->act3;
    const bool h16 = srlx_conv_h16();
    const size_t lds = h16 && !c1_f32 ? kLdsH16 : kLdsBytes;
    // one workgroup per sample.  (Measured and dropped in round 4: fewer, sample-walking workgroups -- the loop form cost 20 % in code generation -- and the launch
    // cut into chunks of consecutive samples so that the update's kernels get compute units earlier: -1.1 % / +0.4 % per lock-step on two boxes; profiles/NOTES.md.)
    auto launch = [&](auto kern) {
        hipLaunchKernelGGL(kern, dim3((unsigned)batch), dim3(64 * kWaves), lds, st, d_frame_base, d_frame_off, h->wpack, h->b1, h->b2, h->b3, out3,
                           keep ? h->act1 : nullptr, keep ? h->act2 : nullptr, (unsigned long long *)h->fused_dbg, (unsigned char *)nullptr, 0ll, (long long)batch, 0ll, h->range_flag);
    };
    static const bool c23_f32 = getenv("SRLX_CONV23_F32") && getenv("SRLX_CONV23_F32")[0] == '1';  // A/B switch: conv2 / conv3 on the float32 matrix pipe
    if (h->probe0 && hipEventRecord(h->probe0, st) != hipSuccess) return false;  // measurement hook: exactly this kernel, on its launch stream
    // the dense layers will run on operand planes (srlx_qnet_dense_rows's own condition): conv3 writes them itself, float32 act3 is not produced
    static const bool no_planes_out = (getenv("SRLX_NO_CONV_PLANES") && getenv("SRLX_NO_CONV_PLANES")[0] == '1') ||  // A/B switch: float32 act3 + a split pass
                                      (getenv("SRLX_NO_PLANES_GEMM") && getenv("SRLX_NO_PLANES_GEMM")[0] == '1');
    h->a3_planes_fresh = false;
    if (h->want_planes_out && !no_planes_out && !c1_f32 && !c23_f32 && !keep && h->planes_valid && !h->eff[0] && srlx_fc1_planes_applicable(h, batch) && batch >= 512) {
        out3 = reinterpret_cast<float *>(h->a3_planes);
        h16 ? launch(k_convnet_fused<true, true, true, 1, true>) : launch(k_convnet_fused<true, true, true, 1>);
        h->a3_planes_fresh = true;
    } else if (h->want_planes_out && !no_planes_out && !c1_f32 && !c23_f32 && h->planes_valid && !h->eff[0] && srlx_fc1_planes_applicable(h, batch)) {
        // a learner's pass (96 / 128 rows; `planes_small`): float32 act3 for its backward pass AND the planes for the first dense layer, rows padded to the GEMM's tile
        const long long prow = (batch + 127) / 128 * 128;
        auto launch2 = [&](auto kern) {
            hipLaunchKernelGGL(kern, dim3((unsigned)batch), dim3(64 * kWaves), lds, st, d_frame_base, d_frame_off, h->wpack, h->b1, h->b2, h->b3, out3,
                               keep ? h->act1 : nullptr, keep ? h->act2 : nullptr, (unsigned long long *)h->fused_dbg, (unsigned char *)h->a3_planes, prow, (long long)batch, 0ll, h->range_flag);
        };
        h16 ? launch2(k_convnet_fused<false, true, true, 2, true>) : launch2(k_convnet_fused<false, true, true, 2>);
        h->a3_planes_fresh = true;
    } else if (batch >= 512)
        c1_f32 ? launch(k_convnet_fused<true, false, false>) : c23_f32 ? launch(k_convnet_fused<true, true, false>)
               : h16 ? launch(k_convnet_fused<true, true, true, 0, true>) : launch(k_convnet_fused<true, true, true>);
    else
        c1_f32 ? launch(k_convnet_fused<false, false, false>) : c23_f32 ? launch(k_convnet_fused<false, true, false>)
               : h16 ? launch(k_convnet_fused<false, true, true, 0, true>) : launch(k_convnet_fused<false, true, true>);
    if (h->probe1 && hipEventRecord(h->probe1, st) != hipSuccess) return false;
    if (h->stamp_buf && srlx_debug_stamp(h->stamp_buf, 10, st) != SRLX_OK) return false;  // (measurement aid: the convolution launch of a forward pass is done)
    return hipGetLastError() == hipSuccess;
}

// conv1 -> conv2 -> conv3 of `n` handles over the same `batch` frame stacks as one launch; every handle is left as after its own chip-filling forward's convolution
// kernel (operand planes fresh): srlx_qnet_forward_dense_planes continues each.
int srlx_qnet_fused_convs_multi(srlx_qnet *const *hs, int n, int64_t batch, const uint8_t *d_frame_base, const int64_t *d_frame_off, hipStream_t st) {
    SRLX_REQUIRE(n >= 1 && n <= kMultiMax && batch >= 512, "qnet_forward_convs_multi: 1..%d handles, chip-filling batches", kMultiMax);
    static bool attr_set = false;
    if (!attr_set) {
        SRLX_HIP(hipFuncSetAttribute((const void *)k_convnet_fused_multi<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLdsBytes));
        SRLX_HIP(hipFuncSetAttribute((const void *)k_convnet_fused_multi<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLdsBytes));
        attr_set = true;
    }
    MultiNets nets{};
    for (int k = 0; k < n; k++) {
        srlx_qnet *h = hs[k];
        SRLX_REQUIRE(h && h->H == 84 && h->W == 84 && h->Wn == 4 && h->F1 == 32, "qnet_forward_convs_multi: handle %d is not the 84 x 84 x 4 / 32-filter geometry", k);
        SRLX_REQUIRE(h->max_train == 0 && h->planes_valid && !h->eff[0] && srlx_fc1_planes_applicable(h, batch) && batch <= h->max_batch,
                     "qnet_forward_convs_multi: handle %d must be an inference handle with valid operand planes for %lld rows", k, (long long)batch);
        if (!h->pack_valid) {
            SRLX_TRY(srlx_qnet_pack_publish(h, nullptr, nullptr, st));
            h->pack_valid = h->pack_sticky;
        }
        h->wt_from_forward = false;
        nets.wpk[k] = h->wpack, nets.b1[k] = h->b1, nets.b2[k] = h->b2, nets.b3[k] = h->b3, nets.act3[k] = reinterpret_cast<float *>(h->a3_planes), nets.flag[k] = h->range_flag;
    }
    if (hs[0]->probe0) SRLX_HIP(hipEventRecord(hs[0]->probe0, st));
    if (srlx_conv_h16())
        hipLaunchKernelGGL(k_convnet_fused_multi<true>, dim3((unsigned)(n * batch)), dim3(64 * kWaves), kLdsH16, st, d_frame_base, d_frame_off, nets, (long long)batch);
    else
        hipLaunchKernelGGL(k_convnet_fused_multi<false>, dim3((unsigned)(n * batch)), dim3(64 * kWaves), kLdsBytes, st, d_frame_base, d_frame_off, nets, (long long)batch);
    if (hs[0]->probe1) SRLX_HIP(hipEventRecord(hs[0]->probe1, st));
    hs[0]->probe0 = hs[0]->probe1 = nullptr;
    for (int k = 0; k < n; k++) hs[k]->a3_planes_fresh = true, hs[k]->want_planes_out = true;
    SRLX_HIP(hipGetLastError());
    return SRLX_OK;
}
