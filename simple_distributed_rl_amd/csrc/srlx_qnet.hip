// srlx_qnet.hip -- exact-fp32 inference of the DQN-image + dueling Q-network on the CDNA4 matrix cores.
//
// Replaces, for the no-grad forwards of the hot path (the actor's policy step over E environments and
// the learner's online/target evaluation of s_1..s_n), the stock torch modules of
//   srl/rl/torch_/blocks/dqn_image_block.py:10-67   (conv 8/4 -> 4/2 -> 3/1, replicate padding, ReLU)
//   srl/rl/torch_/blocks/dueling_network.py:8-59     (V/A heads, Q = V + A - mean A)
//   srl/algorithms/rainbow/model_torch.py:55-67      (pred_q / pred_target_q incl. their H2D/D2H hops)
// These dense layers are the only MFMA-shaped work on the path.  Every layer is an implicit GEMM
//   C[M, N] = A[M, K] * W[N, K]^T (+ bias, ReLU),  M = samples x output pixels,
// computed with v_mfma_f32_32x32x2_f32 (f32 in, f32 accumulate: bit-for-bit an fmaf chain, no TF32-like
// shortcut exists on gfx950), so results stay within f32 round-off of the reference's f32 network
// (the summation order differs from MIOpen's; tests/test_qnet_gpu.py bounds it at 1e-5 relative).
//
// Weights are NOT copied: the kernels read the torch parameters in place (srlx_qnet_bind).  The engine's
// torch module keeps conv2/conv3 weights in channels_last memory ([Cout][ky][kx][Cin] = the K order of an
// NHWC implicit GEMM), the fused V/A first layer as one [2*hidden][flat] matrix whose columns follow the
// NHWC flatten order, so autograd/Adam and these kernels share one copy of the parameters.
//
// Data movement:
//   conv1 reads the uint8 frame ring DIRECTLY through a per-sample frame-offset table (frame stacking,
//   zero history at episode start, u8/255 normalisation and replicate padding happen in the tile loader):
//   the float32 stacked observation (112 896 B per sample) is never written.
//   Activations are NHWC float32, so the K dimension of conv2/conv3 is contiguous in memory and a
//   layer's output tile is the next layer's input without a transpose.
//   FC1 (7744 -> 2 x hidden, V and A heads concatenated) is split along K so that even a 96-sample learner
//   batch fills the chip; the per-split partial sums are reduced, biased and ReLU-ed by the head kernel,
//   which also does the tiny second layers and the dueling combine.  No atomics: results are deterministic.
//
// Tile: 128 (M) x 64|32 (N) x 32 (K) per 256-thread workgroup, 4 waves; LDS rows padded to 36 floats so
// the 16-float fragment reads (4 x ds_read_b128 per lane) are bank-conflict free; global->register
// prefetch of the next K-slab overlaps the 16 MFMAs of the current one.
#include <new>

#include "srlx_qnet_int.h"

namespace {

using i64 = int64_t;
using u8 = unsigned char;
using f32x16 = __attribute__((ext_vector_type(16))) float;
using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;

constexpr int BM = 128, BK = 32, LDT = 36;

__device__ __forceinline__ int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

// ---- A-operand loaders: one float4 (4 consecutive k) of one GEMM row ---------------------------------
struct APlain {  // row-major [M][K]
    const float *A;
    i64 lda;
    struct Row {
        const float *p;
    };
    __device__ __forceinline__ Row row(i64 m, i64 M) const { return Row{m < M ? A + m * lda : nullptr}; }
    __device__ __forceinline__ float4 load4(const Row &r, int k0, int c4) const {
        return r.p ? *reinterpret_cast<const float4 *>(r.p + k0 + c4) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
};

struct AConv {  // NHWC float32 input [B][H][W][C]; k = (ky * KW + kx) * C + c; replicate padding.  C is a multiple of 32,
                // so the 32 k of one slab share their filter tap: (ky, kx) come from a per-slab table, no per-lane division
    const float *in;
    int H, W, C, KW, S, P, OH, OW;
    int tap[80];  // slab (k0 / 32) -> filter tap ky | kx << 8: dwords, so that the uniform lookup is a scalar load (byte tables are
                  // fetched with per-lane global_load_ubyte + vmcnt(0), which serialised the four row fetches of a slab)
    struct Row {
        const float *img;  // null: row beyond M
        int iy0, ix0;
    };
    __device__ __forceinline__ Row row(i64 m, i64 M) const {
        if (m >= M) return Row{nullptr, 0, 0};
        const int per = OH * OW;
        const i64 b = m / per;
        const int pix = (int)(m % per);
        const int oy = pix / OW, ox = pix % OW;
        return Row{in + b * (i64)H * W * C, oy * S - P, ox * S - P};
    }
    __device__ __forceinline__ float4 load4(const Row &r, int k0, int c4) const {
        if (!r.img) return make_float4(0.f, 0.f, 0.f, 0.f);
        const int slab = k0 >> 5;                 // uniform across the workgroup
        const int c = (k0 & (C - 1) & ~31) + c4;  // C is 32 or a multiple of 64 that is a power of two
        const int tp = tap[slab];
        const int iy = clampi(r.iy0 + (tp & 255), 0, H - 1), ix = clampi(r.ix0 + (tp >> 8), 0, W - 1);
        return *reinterpret_cast<const float4 *>(r.img + ((i64)iy * W + ix) * C + c);
    }
    void fill_taps(int K) {
        for (int sl = 0; sl < K / 32 && sl < 80; sl++) {
            const int kyx = (sl * 32) / C;
            tap[sl] = (kyx / KW) | ((kyx % KW) << 8);
        }
    }
};

// u8 / 255 correctly rounded in three operations (quotient estimate + one fma residual correction); equal to the IEEE
// division for every byte value (checked exhaustively with exact arithmetic), a fifth of the instructions of __fdiv_rn
__device__ __forceinline__ float byte_to_unit(unsigned b) {
    const float x = (float)b, rcp = 1.0f / 255.0f;
    const float q = x * rcp;
    return fmaf(fmaf(-q, 255.0f, x), rcp, q);
}

struct AU8 {  // uint8 frames addressed through frame_off[sample][Wn] (bytes from `base`, < 0: zero frame);
              // k = c * 64 + ky * 8 + kx (torch conv weight order, 8x8 kernel); replicate padding; value = u8 / 255
    const u8 *base;
    const i64 *frame_off;
    int Wn, H, W, S, P, OH, OW;
    struct Row {
        const i64 *offs;  // null: row beyond M
        int iy0, ix0;
    };
    __device__ __forceinline__ Row row(i64 m, i64 M) const {
        if (m >= M) return Row{nullptr, 0, 0};
        const int per = OH * OW;
        const i64 b = m / per;
        const int pix = (int)(m % per);
        return Row{frame_off + b * Wn, (pix / OW) * S - P, (pix % OW) * S - P};
    }
    __device__ __forceinline__ float4 load4(const Row &r, int k0, int c4) const {
        if (!r.offs) return make_float4(0.f, 0.f, 0.f, 0.f);
        const int c = k0 >> 6;  // uniform: a 32-wide slab never straddles a frame (64 taps)
        const int ky = ((k0 & 63) + c4) >> 3, kx = c4 & 7;
        const i64 off = r.offs[c];
        if (off < 0) return make_float4(0.f, 0.f, 0.f, 0.f);
        const u8 *row = base + off + (i64)clampi(r.iy0 + ky, 0, H - 1) * W;
        const int x = r.ix0 + kx;
        unsigned b0, b1, b2, b3;
        if (x >= 0 && x + 3 < W) {  // interior: one (unaligned) 4-byte load
            unsigned w;
            __builtin_memcpy(&w, row + x, 4);
            b0 = w & 255u, b1 = (w >> 8) & 255u, b2 = (w >> 16) & 255u, b3 = w >> 24;
        } else {  // replicate padding at the left / right border
            b0 = row[clampi(x, 0, W - 1)], b1 = row[clampi(x + 1, 0, W - 1)], b2 = row[clampi(x + 2, 0, W - 1)], b3 = row[clampi(x + 3, 0, W - 1)];
        }
        return make_float4(byte_to_unit(b0), byte_to_unit(b1), byte_to_unit(b2), byte_to_unit(b3));
    }
};

// ---- conv1 straight from the uint8 frame ring ---------------------------------------------------------
// The generic implicit GEMM re-gathers every byte of a frame four times from HBM-resident memory and pays the
// address arithmetic per element; with N = 32 there are only 16 MFMAs per K-slab to hide that behind (measured:
// 200 us at 1024 samples = 24 % of the matrix-core peak).  Here ONE workgroup owns ONE sample: its window of four
// frames is staged once into LDS (uint8, replicate padding materialised: 88 x 88 per frame = 31 KB, so several
// workgroups share a CU and one's staging hides behind the others' MFMAs), every lane keeps ITS slice of the 32 x 256
// filter matrix in registers for all the tiles of its wave (8 K-slabs x 16 floats), and every A fragment is two 8-byte
// LDS reads + 16 exact u8/255 conversions in registers.
// Fragment order: slab (frame c, kernel rows kyb..kyb+3); lane (i, h) holds rows kyb + 2h, kyb + 2h + 1, kx = 0..7,
// i.e. k = c*64 + kyb*8 + 16h + s -- the same 16 contiguous floats of the [32][256] weight row for B.
constexpr int kC1Pad = 88;                     // padded frame side: 4*20 + 8
constexpr int kC1Frame = kC1Pad * kC1Pad;      // bytes per staged frame

__global__ void __launch_bounds__(256, 3) k_conv1_u8(const u8 *__restrict__ base, const i64 *__restrict__ frame_off, int Wn, int H, int W, int OH, int OW,
                                                  const float *__restrict__ w1, const float *__restrict__ b1, float *__restrict__ act1) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    u8 *fr = smem;  // [4][88][88]
    const int t = threadIdx.x, wave = t >> 6, lane = t & 63;
    const i64 b = blockIdx.x;
    constexpr int K = 256;  // Wn == 4 (checked by the launcher)
    // frames: padded dword (row r, dword d) covers padded columns 4d..4d+3 = image columns clamp(4d - 3 .. 4d), image row
    // clamp(r - 3).  The ring lives in HBM: all loads of a frame are issued before the first LDS store (8 in flight per
    // lane), each is ONE in-bounds unaligned dword whose bytes are re-picked at the left / right border.
    constexpr int kPer = (kC1Pad * (kC1Pad / 4) + 255) / 256;  // 8 dwords per lane per frame
    for (int c0 = 0; c0 < Wn; c0 += 4) {  // up to four frames (32 loads per lane) in flight together
        i64 off[4];
#pragma unroll
        for (int q = 0; q < 4; q++) off[q] = c0 + q < Wn ? frame_off[b * Wn + c0 + q] : -1;
        unsigned v[4][kPer];
#pragma unroll
        for (int q = 0; q < 4; q++)
#pragma unroll
            for (int j = 0; j < kPer; j++) {
                const int idx = t + 256 * j;
                v[q][j] = 0u;
                if (off[q] >= 0 && idx < kC1Pad * (kC1Pad / 4)) {
                    const int r = idx / (kC1Pad / 4), d = idx % (kC1Pad / 4);
                    __builtin_memcpy(&v[q][j], base + off[q] + (i64)clampi(r - 3, 0, H - 1) * W + clampi(4 * d - 3, 0, W - 4), 4);
                }
            }
#pragma unroll
        for (int q = 0; q < 4; q++) {
            if (c0 + q >= Wn) break;
            unsigned *dst = reinterpret_cast<unsigned *>(fr + (c0 + q) * kC1Frame);
#pragma unroll
            for (int j = 0; j < kPer; j++) {
                const int idx = t + 256 * j;
                if (idx >= kC1Pad * (kC1Pad / 4)) continue;
                const int x = 4 * (idx % (kC1Pad / 4)) - 3, xs = clampi(x, 0, W - 4);
                unsigned o = v[q][j];
                if (off[q] >= 0 && x != xs) {  // border: column clamp(x + p) sits at byte clamp(x + p) - xs of the loaded dword
                    o = 0u;
#pragma unroll
                    for (int p = 0; p < 4; p++) o |= ((v[q][j] >> (8 * (clampi(x + p, 0, W - 1) - xs))) & 255u) << (8 * p);
                }
                dst[idx] = o;
            }
        }
    }
    // this lane's B fragments of all eight K-slabs: filter row n = lane & 31, k = slab*32 + 16 (lane >> 5) + 0..15
    float bfr[8][16];
    {
        const float *wp = w1 + (i64)(lane & 31) * K + 16 * (lane >> 5);
#pragma unroll
        for (int sl = 0; sl < 8; sl++)
#pragma unroll
            for (int v = 0; v < 4; v++) {
                const float4 x = *reinterpret_cast<const float4 *>(wp + sl * 32 + 4 * v);
                // 1/255 is folded into the filter slice ONCE per lane: the A operand is then the raw byte (one v_cvt_f32_ubyte per
                // MFMA instead of cvt + 3-op exact division; f32 MFMAs and VALU instructions share the SIMD's issue slots).
                // byte * RN(w/255) differs from RN(byte/255) * w by <= 1 ulp per product: Q-values stay inside the 1e-5 bar.
                bfr[sl][4 * v] = x.x * (1.0f / 255.0f), bfr[sl][4 * v + 1] = x.y * (1.0f / 255.0f), bfr[sl][4 * v + 2] = x.z * (1.0f / 255.0f),
                bfr[sl][4 * v + 3] = x.w * (1.0f / 255.0f);
            }
    }
    __syncthreads();
    const int h = lane >> 5, i = lane & 31;
    const int M = OH * OW, tiles = (M + 31) / 32;
    const float bias = b1[i];
    // 14 tiles over 4 waves is 4/4/3/3: rotate the assignment with the sample index so that every SIMD of a CU gets the same
    // number of tiles over the samples it hosts
    const int wrot = (wave + (int)(b & 3)) & 3;
    for (int tile = wrot * gridDim.y + blockIdx.y; tile < tiles; tile += 4 * gridDim.y) {  // gridDim.y workgroups share a sample at small batches
        const int m = tile * 32 + i < M ? tile * 32 + i : M - 1;
        const int oy = m / OW, ox = m % OW;
        const u8 *win = fr + (4 * oy + 2 * h) * kC1Pad + 4 * ox;  // this lane's first kernel row inside a slab
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; r++) acc[r] = 0.f;
#pragma unroll
        for (int c = 0; c < 4; c++) {
#pragma unroll
            for (int kyb = 0; kyb < 8; kyb += 4) {
                const u8 *p = win + c * kC1Frame + kyb * kC1Pad;
                unsigned w[4];
                w[0] = *reinterpret_cast<const unsigned *>(p);
                w[1] = *reinterpret_cast<const unsigned *>(p + 4);
                w[2] = *reinterpret_cast<const unsigned *>(p + kC1Pad);
                w[3] = *reinterpret_cast<const unsigned *>(p + kC1Pad + 4);
#pragma unroll
                for (int s = 0; s < 16; s++) {
                    const float a = (float)((w[s >> 2] >> (8 * (s & 3))) & 255u);
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bfr[2 * c + kyb / 4][s], acc, 0, 0, 0);
                }
            }
        }
        // C/D layout of the 32x32 MFMA: col = lane & 31 (output channel), row = (r & 3) + 8 (r >> 2) + 4 h (pixel)
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const int mm = tile * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
            if (mm < M) {
                const float v = acc[r] + bias;
                act1[(b * M + mm) * 32 + i] = v > 0.f ? v : 0.f;
            }
        }
    }
}

// Workgroup barrier that orders LDS traffic only.  `__syncthreads()` also drains vmcnt, i.e. it makes every wave wait for
// the global loads of the NEXT K-slab that were issued a few hundred cycles earlier -- the prefetch then hides at most one
// MFMA phase (0.85 us) of memory latency and the matrix pipe idles behind the barrier.  The prefetched registers are
// consumed after the next barrier; the compiler inserts the vmcnt wait there.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// ---- the GEMM ------------------------------------------------------------------------------------------
template <class AL, int BN, bool RELU, bool SPLITK, int TBM = 128>
__global__ void __launch_bounds__(256) k_gemm(AL al, const float *__restrict__ Bw, const float *__restrict__ bias, float *__restrict__ C, i64 M, int N,
                                              int K, int k_per_split, i64 zstride_w = 0, i64 zstride_c = 0) {
    if (!SPLITK) {  // batched GEMMs (blockIdx.z): same A loader, one weight matrix and one output per z
        Bw += blockIdx.z * zstride_w;
        C += blockIdx.z * zstride_c;
    }
    // two LDS buffers: slab s+1 is stored while slab s is still being read by slower waves -- ONE barrier per K-slab instead of two
    __shared__ __attribute__((aligned(16))) float As2[2][TBM * LDT];
    __shared__ __attribute__((aligned(16))) float Bs2[2][BN * LDT];
    const int t = threadIdx.x, wave = t >> 6, lane = t & 63;
    // XCD-aware tile order for the split-K launches (FC1).  Workgroups go round-robin over the 8 XCDs in linear-id order, each XCD has its
    // own 4 MB L2: with the plain mapping (x = M tile fastest, 8 M tiles) XCD i computes M tile i against EVERY weight tile, i.e. the 32 MB
    // weight matrix is fetched once per XCD -- rocprofv3 FETCH_SIZE: 302 MB per launch at 1024 rows against 80 MB algorithmic.  Handing XCD i
    // the i-th CONTIGUOUS eighth of the (split, N tile, M tile) space gives it one K range x half the N tiles x all M tiles: every weight tile
    // is fetched by one XCD, every activation K-slice by two.
    unsigned bx = blockIdx.x, by = blockIdx.y, bz = blockIdx.z;
    if (SPLITK) {
        const unsigned gx = gridDim.x, gy = gridDim.y, total = gx * gy * gridDim.z;
        if (total % 8 == 0) {
            const unsigned lin = bx + gx * (by + gy * bz), tile = (lin % 8) * (total / 8) + lin / 8;
            bx = tile % gx, by = (tile / gx) % gy, bz = tile / (gx * gy);
        }
    }
    const i64 m0 = (i64)bx * TBM;
    const int n0 = by * BN;
    const int kbeg = SPLITK ? bz * k_per_split : 0;
    const int kend = SPLITK ? (kbeg + k_per_split < K ? kbeg + k_per_split : K) : K;
    constexpr int WN = BN / 32, WM = 4 / WN, MT = TBM / (32 * WM), NB = BN / 32, MR = TBM / 32;
    const int wn = wave % WN, wm = wave / WN;
    f32x16 acc[MT];
#pragma unroll
    for (int i = 0; i < MT; i++)
#pragma unroll
        for (int r = 0; r < 16; r++) acc[i][r] = 0.f;

    const int lrow = t >> 3, c4 = (t & 7) * 4;  // this lane stages rows lrow + 32 j, columns c4..c4+3 of a tile
    typename AL::Row rows[MR];
#pragma unroll
    for (int j = 0; j < MR; j++) rows[j] = al.row(m0 + lrow + 32 * j, M);
    float4 ra[MR], rb[NB];
    auto fetch = [&](int k0) {
#pragma unroll
        for (int j = 0; j < MR; j++) ra[j] = al.load4(rows[j], k0, c4);
#pragma unroll
        for (int j = 0; j < NB; j++) {
            const int n = n0 + lrow + 32 * j;
            rb[j] = n < N ? *reinterpret_cast<const float4 *>(Bw + (i64)n * K + k0 + c4) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    auto stage = [&](int buf) {
#pragma unroll
        for (int j = 0; j < MR; j++) *reinterpret_cast<float4 *>(&As2[buf][(lrow + 32 * j) * LDT + c4]) = ra[j];
#pragma unroll
        for (int j = 0; j < NB; j++) *reinterpret_cast<float4 *>(&Bs2[buf][(lrow + 32 * j) * LDT + c4]) = rb[j];
    };
    const int h = lane >> 5, i = lane & 31;
    int cur = 0;
    if (kbeg < kend) {
        fetch(kbeg);
        stage(0);
    }
    lds_barrier();
    for (int k0 = kbeg; k0 < kend; k0 += BK) {
        const bool more = k0 + BK < kend;
        if (more) fetch(k0 + BK);  // overlaps with the MFMAs below
        const float *As = As2[cur], *Bs = Bs2[cur];
        // fragments: lane (i, h) holds k = 16 h + s, s = 0..15, of row/column i (A and B use the same k order)
        float bf[16];
#pragma unroll
        for (int v = 0; v < 4; v++) {
            const float4 x = *reinterpret_cast<const float4 *>(&Bs[(wn * 32 + i) * LDT + 16 * h + 4 * v]);
            bf[4 * v] = x.x, bf[4 * v + 1] = x.y, bf[4 * v + 2] = x.z, bf[4 * v + 3] = x.w;
        }
#pragma unroll
        for (int ms = 0; ms < MT; ms++) {
            float af[16];
#pragma unroll
            for (int v = 0; v < 4; v++) {
                const float4 x = *reinterpret_cast<const float4 *>(&As[(wm * 32 * MT + ms * 32 + i) * LDT + 16 * h + 4 * v]);
                af[4 * v] = x.x, af[4 * v + 1] = x.y, af[4 * v + 2] = x.z, af[4 * v + 3] = x.w;
            }
#pragma unroll
            for (int s = 0; s < 16; s++) acc[ms] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[s], bf[s], acc[ms], 0, 0, 0);
        }
        if (more) stage(cur ^ 1);  // the other buffer: every wave left it before the barrier that ended the previous slab
        lds_barrier();
        cur ^= 1;
    }
    // epilogue: C/D layout of the 32x32 MFMA: col = lane & 31, row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5)
    float *Cz = SPLITK ? C + (i64)bz * M * N : C;
    const int n = n0 + wn * 32 + i;
    const float bv = (!SPLITK && bias && n < N) ? bias[n] : 0.f;
#pragma unroll
    for (int ms = 0; ms < MT; ms++)
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const i64 m = m0 + wm * 32 * MT + ms * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
            if (m < M && n < N) {
                float v = acc[ms][r] + bv;
                if (RELU) v = v > 0.f ? v : 0.f;
                Cz[m * N + n] = v;
            }
        }
}

// ---- FC1 on the bf16 matrix pipe: float32 x float32 from six exact partial products ---------------------------------------------------
// A float32 is the sum of three bf16 parts exactly (8 + 8 + 8 mantissa bits; each remainder is exact in float32).  a * b is evaluated
// as the six largest of the nine partial products a_p * b_q (p + q <= 2), each exact in the float32 accumulator of
// v_mfma_f32_32x32x16_bf16; the three dropped ones are below 2^-24 |a b|, the size of float32's own rounding of the product.
// 6 MFMAs of 32 cycles per 16 K against 8 of 64 on the float32 pipe.  The splitting is done while STAGING: the operands are read as float32
// (activation rows with a row stride, the weight in place), each thread turns its 24 floats per K-slab into three bf16 parts (~130 VALU
// instructions, in the shadow of the slab's 24 MFMAs) and stores them to three plane tiles in LDS.  (Pre-split operand planes in HBM -- one
// splitting pass over the 32 MB weight per forward, act3 written as planes by the convolution kernel -- were built first and measured slower inside
// the lock-step loop: 0.573 against 0.545 ms.)  Tiles, split-K and XCD-aware order as k_gemm.
constexpr int kRowB = 80;  // LDS bytes per tile row: 32 bf16 (64 B) + 16 B pad: the 16 lanes of a ds_read_b128 pass start 20 banks apart
// H16 (round 6; the first dense layer's FORWARD only): two float16 parts per operand, hi = f16(x), lo = f16((x - hi) * 2048) -- three exact products per 16 K into an unscaled
// and a 2^11-scaled accumulator, joined once per split (srlx_qnet_fused.hip has the derivation; forward error = float32 round-off) instead of six of three bf16 parts.
// The data-gradient GEMMs stay on the three-part bf16 split: gradients live far below float16's normal range.  Per 16-k step, in this order: lo += a_lo b_hi,
// lo += a_hi b_lo, acc += a_hi b_hi -- the order k_fc1_planes / k_fc1_planes_h keep (bit-identical results).
template <class AL, int BN, bool SPLITK, int TBM = 128, bool H16 = false>
__global__ void __launch_bounds__(256) k_gemm_s16(AL al, const float *__restrict__ Bw, float *__restrict__ C, i64 M, int N, int K, int k_per_split, i64 zstride_w = 0,
                                                  i64 zstride_c = 0) {
    if (!SPLITK) {  // batched GEMMs (blockIdx.z): same A loader, one weight matrix and one output per z
        Bw += blockIdx.z * zstride_w;
        C += blockIdx.z * zstride_c;
    }
    constexpr int WN = BN / 32, WM = 4 / WN, MT = TBM / (32 * WM), NB = BN / 32, MR = TBM / 32;
    __shared__ __attribute__((aligned(16))) unsigned char As[3 * TBM * kRowB];
    __shared__ __attribute__((aligned(16))) unsigned char Bs[3 * BN * kRowB];
    const int t = threadIdx.x, wave = t >> 6, lane = t & 63, i = lane & 31, h = lane >> 5;
    unsigned bx = blockIdx.x, by = blockIdx.y, bz = blockIdx.z;
    if (SPLITK) {  // XCD-aware tile order (see k_gemm)
        const unsigned gx = gridDim.x, gy = gridDim.y, total = gx * gy * gridDim.z;
        if (total % 8 == 0) {
            const unsigned lin = bx + gx * (by + gy * bz), tile = (lin % 8) * (total / 8) + lin / 8;
            bx = tile % gx, by = (tile / gx) % gy, bz = tile / (gx * gy);
        }
    }
    const i64 m0 = (i64)bx * TBM;
    const int n0 = by * BN;
    const int kbeg = SPLITK ? bz * k_per_split : 0;
    const int kend = SPLITK ? (kbeg + k_per_split < K ? kbeg + k_per_split : K) : K;
    const int wn = wave % WN, wm = wave / WN;
    f32x16 acc[MT], lo[H16 ? MT : 1];
#pragma unroll
    for (int a = 0; a < MT; a++)
#pragma unroll
        for (int r = 0; r < 16; r++) {
            acc[a][r] = 0.f;
            if constexpr (H16) lo[a][r] = 0.f;
        }
    const int lrow = t >> 3, c4 = (t & 7) * 4;  // this lane stages rows lrow + 32 j, columns c4..c4+3 of a tile (as k_gemm)
    typename AL::Row rows[MR];
#pragma unroll
    for (int j = 0; j < MR; j++) rows[j] = al.row(m0 + lrow + 32 * j, M);
    float4 ra[MR], rb[NB];
    auto fetch = [&](int k0) {
#pragma unroll
        for (int j = 0; j < MR; j++) ra[j] = al.load4(rows[j], k0, c4);
#pragma unroll
        for (int j = 0; j < NB; j++) {
            const int n = n0 + lrow + 32 * j;
            rb[j] = n < N ? *reinterpret_cast<const float4 *>(Bw + (i64)n * K + k0 + c4) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    auto put = [&](unsigned char *tile, int rows_, int row, const float4 &x) __attribute__((always_inline)) {  // three bf16 parts of four floats -> the three plane tiles
        float r[4] = {x.x, x.y, x.z, x.w};
        if constexpr (H16) {
            typedef __attribute__((ext_vector_type(4))) _Float16 f16x4;
            f16x4 hp, lp;
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const _Float16 hi = (_Float16)r[j];
                hp[j] = hi, lp[j] = (_Float16)((r[j] - (float)hi) * 2048.0f);
            }
            *reinterpret_cast<f16x4 *>(&tile[row * kRowB + 2 * c4]) = hp;
            *reinterpret_cast<f16x4 *>(&tile[(rows_ + row) * kRowB + 2 * c4]) = lp;
            return;
        }
#pragma unroll
        for (int p = 0; p < 3; p++) {
            typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
            bf16x4 part;
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const __bf16 b = (__bf16)r[j];
                part[j] = b;
                r[j] -= (float)b;
            }
            *reinterpret_cast<bf16x4 *>(&tile[(p * rows_ + row) * kRowB + 2 * c4]) = part;
        }
    };
    if (kbeg < kend) fetch(kbeg);
    for (int k0 = kbeg; k0 < kend; k0 += BK) {
#pragma unroll
        for (int j = 0; j < MR; j++) put(As, TBM, lrow + 32 * j, ra[j]);
#pragma unroll
        for (int j = 0; j < NB; j++) put(Bs, BN, lrow + 32 * j, rb[j]);
        lds_barrier();
        if (k0 + BK < kend) fetch(k0 + BK);  // overlaps with the MFMAs below
#pragma unroll
        for (int ks = 0; ks < 2; ks++) {
            if constexpr (H16) {
                typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
                f16x8 bh[2], ah[MT][2];
#pragma unroll
                for (int p = 0; p < 2; p++) bh[p] = *reinterpret_cast<const f16x8 *>(&Bs[(p * BN + wn * 32 + i) * kRowB + (2 * ks + h) * 16]);
#pragma unroll
                for (int ms = 0; ms < MT; ms++)
#pragma unroll
                    for (int p = 0; p < 2; p++) ah[ms][p] = *reinterpret_cast<const f16x8 *>(&As[(p * TBM + wm * 32 * MT + ms * 32 + i) * kRowB + (2 * ks + h) * 16]);
#pragma unroll
                for (int ms = 0; ms < MT; ms++) lo[ms] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[ms][1], bh[0], lo[ms], 0, 0, 0);
#pragma unroll
                for (int ms = 0; ms < MT; ms++) lo[ms] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[ms][0], bh[1], lo[ms], 0, 0, 0);
#pragma unroll
                for (int ms = 0; ms < MT; ms++) acc[ms] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[ms][0], bh[0], acc[ms], 0, 0, 0);
                continue;
            }
            bf16x8 bf[3];
#pragma unroll
            for (int p = 0; p < 3; p++) bf[p] = *reinterpret_cast<const bf16x8 *>(&Bs[(p * BN + wn * 32 + i) * kRowB + (2 * ks + h) * 16]);
            bf16x8 af[MT][3];
#pragma unroll
            for (int ms = 0; ms < MT; ms++)
#pragma unroll
                for (int p = 0; p < 3; p++) af[ms][p] = *reinterpret_cast<const bf16x8 *>(&As[(p * TBM + wm * 32 * MT + ms * 32 + i) * kRowB + (2 * ks + h) * 16]);
            // smallest partial products first; the accumulators alternate so that no MFMA waits for the one issued just before it
            constexpr int pq[6][2] = {{2, 0}, {0, 2}, {1, 1}, {1, 0}, {0, 1}, {0, 0}};
#pragma unroll
            for (int c = 0; c < 6; c++)
#pragma unroll
                for (int ms = 0; ms < MT; ms++) acc[ms] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[ms][pq[c][0]], bf[pq[c][1]], acc[ms], 0, 0, 0);
        }
        lds_barrier();
    }
    float *Cz = SPLITK ? C + (i64)bz * M * N : C;
    const int n = n0 + wn * 32 + i;
#pragma unroll
    for (int ms = 0; ms < MT; ms++)
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const i64 m = m0 + wm * 32 * MT + ms * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
            float out = acc[ms][r];
            if constexpr (H16) out = __builtin_fmaf(lo[ms][r], 1.0f / 2048.0f, out);
            if (m < M && n < N) Cz[m * N + n] = out;
        }
}

// ---- head: reduce FC1 splits (+bias, ReLU), second layers, dueling combine; one workgroup per sample ----
constexpr int kMaxActions = 32;
// AMAX = 8 / 16 / 32 >= A (round 5: with the loops over kMaxActions = 32 and a run-time A, every one of the accumulate / shuffle / store loops carried 32 uniform
// branches for 6 live actions, and thread 0 summed the waves' partial rows with 56 dependent LDS reads: 16.5 us for the learner's 128 rows, of which the loads were 7)
template <int AMAX>
__global__ void __launch_bounds__(512) k_head(const float *__restrict__ partial, int splits, i64 M, int hidden, const float *__restrict__ b1,
                                              const float *__restrict__ v2w, const float *__restrict__ v2b, const float *__restrict__ a2w,
                                              const float *__restrict__ a2b, int A, int dueling, float *__restrict__ q, float *__restrict__ h1, i64 ostride,
                                              i64 *__restrict__ draw, srlx_qnet::Policy pol, srlx_uvfa_dev uv) {
    __shared__ float red[8][kMaxActions + 1];  // one row per wave (256 or 512 threads)
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const i64 m = blockIdx.x;
    // UVFA inputs of this row (agent57_light/model_torch.py:52-62): the rewards scale their columns of the first dense layer, each one-hot selects one
    float x_ext = 0.f, x_int = 0.f;
    const float *col_act = nullptr, *col_actor = nullptr, *col_ext = nullptr, *col_int = nullptr;
    if (uv.wx) {
        const int n1 = 2 * hidden;
        if (uv.c_ext >= 0) x_ext = uv.r_ext[m], col_ext = uv.wx + (i64)uv.c_ext * n1;
        if (uv.c_int >= 0) x_int = uv.r_int[m], col_int = uv.wx + (i64)uv.c_int * n1;
        if (uv.c_act >= 0) col_act = uv.wx + (i64)(uv.c_act + uv.action[m]) * n1;
        if (uv.c_actor >= 0) col_actor = uv.wx + (i64)(uv.c_actor + uv.actor[m]) * n1;
    }
    const i64 mo = m * ostride;  // row of q / h1 this sample's results go to (the partial sums are dense over the launch's rows)
    const int N1 = 2 * hidden;
    if (draw && blockIdx.x == 0 && t == 0) draw[0] += 1;  // NoisyLinear: the draw this pass used is spent (every reader of draw[0] ran in an earlier launch)
    float v = 0.f, adv[AMAX];
#pragma unroll
    for (int j = 0; j < AMAX; j++) adv[j] = 0.f;
    const int nwaves = blockDim.x >> 6;
    for (int u = t; u < hidden; u += blockDim.x) {
        // split sums in a fixed order with eight independent chains (sixteen loads in flight per iteration)
        float v8[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, a8[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        const float *p = partial + m * N1 + u;
        const i64 ss = M * N1;
        int s = 0;
        if (splits > 8 && splits <= 32) {
            // the learner's 96 / 128-row launches (31 splits): all 62 loads in flight at once instead of four dependent rounds of sixteen; the same sums in the same
            // order (chain q takes splits q, q + 8, q + 16, q + 24; an absent split adds +0.f, which changes no float that came out of an addition with +0.f)
            float pv[32], pa[32];
#pragma unroll
            for (int q = 0; q < 32; q++) {
                pv[q] = q < splits ? p[q * ss] : 0.f;
                pa[q] = q < splits ? p[q * ss + hidden] : 0.f;
            }
#pragma unroll
            for (int r = 0; r < 4; r++)
#pragma unroll
                for (int q = 0; q < 8; q++) {
                    if (8 * r + q < splits) {
                        v8[q] += pv[8 * r + q];
                        a8[q] += pa[8 * r + q];
                    }
                }
            s = splits;
        }
        for (; s + 8 <= splits; s += 8)
#pragma unroll
            for (int q = 0; q < 8; q++) {
                v8[q] += p[(s + q) * ss];
                a8[q] += p[(s + q) * ss + hidden];
            }
        for (; s < splits; s++) {
            v8[s & 7] += p[s * ss];
            a8[s & 7] += p[s * ss + hidden];
        }
        float hv = b1[u] + (((v8[0] + v8[1]) + (v8[2] + v8[3])) + ((v8[4] + v8[5]) + (v8[6] + v8[7])));
        float ha = b1[hidden + u] + (((a8[0] + a8[1]) + (a8[2] + a8[3])) + ((a8[4] + a8[5]) + (a8[6] + a8[7])));
        if (uv.wx) {  // the UVFA columns' share of the pre-activation, in the reference's column order
            float xv = 0.f, xa = 0.f;
            if (col_ext) xv += x_ext * col_ext[u], xa += x_ext * col_ext[hidden + u];
            if (col_int) xv += x_int * col_int[u], xa += x_int * col_int[hidden + u];
            if (col_act) xv += col_act[u], xa += col_act[hidden + u];
            if (col_actor) xv += col_actor[u], xa += col_actor[hidden + u];
            hv += xv, ha += xa;
        }
        hv = hv > 0.f ? hv : 0.f;
        ha = ha > 0.f ? ha : 0.f;
        if (h1) {  // training: the backward pass needs the hidden layer
            h1[mo * N1 + u] = hv;
            h1[mo * N1 + hidden + u] = ha;
        }
        v += hv * v2w[u];
#pragma unroll
        for (int j = 0; j < AMAX; j++)
            if (j < A) adv[j] += ha * a2w[j * hidden + u];
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        v += __shfl_xor(v, off);
#pragma unroll
        for (int j = 0; j < AMAX; j++)
            if (j < A) adv[j] += __shfl_xor(adv[j], off);
    }
    if (lane == 0) {
        red[wave][kMaxActions] = v;
#pragma unroll
        for (int j = 0; j < AMAX; j++)
            if (j < A) red[wave][j] = adv[j];
    }
    __syncthreads();
    if (wave != 0) return;
    // the waves' partial rows: lane j < A sums column j, lane A the value stream's, each over the waves in wave order (what thread 0 did alone: the same sums);
    // thread 0 then collects them through the wave's registers
    float colsum = 0.f;
    {
        const int col = lane < A ? lane : kMaxActions;
        if (lane <= A)
            for (int w = 0; w < nwaves; w++) colsum += red[w][col];
    }
    float out[AMAX];
#pragma unroll
    for (int j = 0; j < AMAX; j++) out[j] = j < A ? __shfl(colsum, j) : 0.f;
    v = __shfl(colsum, A);
    if (t == 0) {
        v += v2b[0];
        float mean = 0.f, mx = -INFINITY;
#pragma unroll
        for (int j = 0; j < AMAX; j++)
            if (j < A) {
                out[j] = out[j] + a2b[j];
                mean += out[j];
                mx = out[j] > mx ? out[j] : mx;
            }
        mean /= (float)A;
        const float sub = dueling == 0 ? mean : (dueling == 1 ? mx : 0.f);  // "average" / "max" / "" (dueling_network.py:49-56)
#pragma unroll
        for (int j = 0; j < AMAX; j++)
            if (j < A) {
                out[j] = v + out[j] - sub;
                q[mo * A + j] = out[j];
                if (pol.q_copy) pol.q_copy[mo * A + j] = out[j];
            }
        // ---- the batched Worker.policy step's selection (rainbow.py:301-329), fused: epsilon-greedy on the Q row this thread has just finished, with the two
        //      uniforms srlx_rng_uniform(seed, counter, 2 E) would have written for the row (u[2 e], u[2 e + 1]); arithmetic = k_eps_greedy (srlx_rollout.hip).
        //      The counter is only READ here (one workgroup per row): the caller advances it once per pass, in a later launch.
        if (pol.actions) {
            const unsigned long long c = (unsigned long long)pol.counter[0];
            const unsigned char *inv = pol.invalid ? pol.invalid + m * A : nullptr;
            int act = 0;
            if (srlx::u53(srlx::rng_u64(pol.seed, c, (unsigned long long)(2 * m))) < (double)pol.eps[m]) {  // random.random() < epsilon (:317)
                int nv = 0;
                for (int a = 0; a < A; a++) nv += !(inv && inv[a]);
                int pick = (int)(srlx::u53(srlx::rng_u64(pol.seed, c, (unsigned long long)(2 * m + 1))) * (double)nv);
                if (pick >= nv) pick = nv - 1;
                for (int a = 0; a < A; a++) {
                    if (inv && inv[a]) continue;
                    if (pick == 0) {
                        act = a;
                        break;
                    }
                    pick--;
                }
            } else {  // q[invalid] = -inf; first maximum, like np.argmax (:321-325)
                float bv = -INFINITY;
                bool have = false;
#pragma unroll
                for (int a = 0; a < AMAX; a++)
                    if (a < A) {
                        const float x = (inv && inv[a]) ? -INFINITY : out[a];
                        if (!have || x > bv) act = a, bv = x, have = true;
                    }
            }
            pol.actions[m] = act;
        }
    }
}

// ---- head_mode 1: the handle ends behind the first dense layer (the embedding / RND networks of Agent57_light, model_torch.py:70-117) ----
// one workgroup per row: split sums in split order + bias + ReLU -> h1 (training) and out[row][0..out_cols); ln_w != NULL: LayerNorm over all N1 units first
// (nn.LayerNorm: biased variance, eps inside the root, :112,116), the lifelong networks' last layer.
__global__ void __launch_bounds__(256) k_hidden_out(const float *__restrict__ partial, int splits, i64 M, int N1, const float *__restrict__ b1, float *__restrict__ out,
                                                    int out_cols, float *__restrict__ h1, i64 ostride, const float *__restrict__ ln_w, const float *__restrict__ ln_b,
                                                    float ln_eps) {
    __shared__ float red[2][4];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const i64 m = blockIdx.x, mo = m * ostride;
    constexpr int kPer = 4;  // N1 <= 1024
    float hv[kPer];
    float s1 = 0.f;
#pragma unroll
    for (int j = 0; j < kPer; j++) {
        const int u = t + 256 * j;
        hv[j] = 0.f;
        if (u < N1) {
            const float *p = partial + m * N1 + u;
            const i64 ss = M * N1;
            float c4[4] = {0.f, 0.f, 0.f, 0.f};
            int s = 0;
            for (; s + 4 <= splits; s += 4)
#pragma unroll
                for (int q = 0; q < 4; q++) c4[q] += p[(s + q) * ss];
            for (; s < splits; s++) c4[s & 3] += p[s * ss];
            float v = b1[u] + ((c4[0] + c4[1]) + (c4[2] + c4[3]));
            v = v > 0.f ? v : 0.f;
            hv[j] = v;
            if (h1) h1[mo * N1 + u] = v;
            s1 += v;
        }
    }
    if (ln_w) {
        for (int off = 32; off > 0; off >>= 1) s1 += __shfl_xor(s1, off);
        if (lane == 0) red[0][wave] = s1;
        __syncthreads();
        const float mean = (((red[0][0] + red[0][1]) + red[0][2]) + red[0][3]) / (float)N1;
        float s2 = 0.f;
#pragma unroll
        for (int j = 0; j < kPer; j++)
            if (t + 256 * j < N1) s2 += (hv[j] - mean) * (hv[j] - mean);
        for (int off = 32; off > 0; off >>= 1) s2 += __shfl_xor(s2, off);
        if (lane == 0) red[1][wave] = s2;
        __syncthreads();
        const float var = (((red[1][0] + red[1][1]) + red[1][2]) + red[1][3]) / (float)N1;
        const float rstd = 1.0f / sqrtf(var + ln_eps);
#pragma unroll
        for (int j = 0; j < kPer; j++) {
            const int u = t + 256 * j;
            if (u < N1) hv[j] = ((hv[j] - mean) * rstd) * ln_w[u] + ln_b[u];
        }
    }
#pragma unroll
    for (int j = 0; j < kPer; j++) {
        const int u = t + 256 * j;
        if (u < out_cols) out[mo * out_cols + u] = hv[j];
    }
}

// stacked float32 NCHW [B][Wn][H][W] -> frame table for AU8 is not possible; for float input conv1 uses
// this NCHW loader instead (k = c*KH*KW + ky*KW + kx, like AU8)
struct ANchw {
    const float *in;
    int Wn, H, W, S, P, OH, OW;
    struct Row {
        const float *img;
        int iy0, ix0;
    };
    __device__ __forceinline__ Row row(i64 m, i64 M) const {
        if (m >= M) return Row{nullptr, 0, 0};
        const int per = OH * OW;
        const i64 b = m / per;
        const int pix = (int)(m % per);
        return Row{in + b * (i64)Wn * H * W, (pix / OW) * S - P, (pix % OW) * S - P};
    }
    __device__ __forceinline__ float4 load4(const Row &r, int k0, int c4) const {
        if (!r.img) return make_float4(0.f, 0.f, 0.f, 0.f);
        const int c = k0 >> 6, ky = ((k0 & 63) + c4) >> 3, kx = c4 & 7;  // 8x8 kernel
        const float *row = r.img + ((i64)c * H + clampi(r.iy0 + ky, 0, H - 1)) * W;
        const int x = r.ix0 + kx;
        return make_float4(row[clampi(x, 0, W - 1)], row[clampi(x + 1, 0, W - 1)], row[clampi(x + 2, 0, W - 1)], row[clampi(x + 3, 0, W - 1)]);
    }
};


// data gradient of a convolution as implicit GEMMs over the PADDED input grid (replicate padding = an explicit pad
// followed by a plain convolution, so its gradient is the plain transposed convolution on the padded grid; the caller
// folds the border rows/columns back).  With stride S only the taps ky = py mod S (+ S a) reach a padded row py, so the
// grid is split into S*S parity classes, one GEMM each (blockIdx.z), with K = (KH/S)(KW/S) CO instead of KH KW CO:
// row m = (b, qy, qx) with py = S qy + cy; k = (a, b', co); value = dY[b][qy - a][qx - b'][co] where that output exists.
struct ADgrad {
    const float *dY;
    int QH, QW, OH, OW, CO;
    int tap[80];  // slab -> a | b' << 8 (dwords: scalar loads)
    struct Row {
        const float *img;  // null: row beyond M
        int qy, qx;
    };
    __device__ __forceinline__ Row row(i64 m, i64 M) const {
        if (m >= M) return Row{nullptr, 0, 0};
        const int per = QH * QW;
        const i64 b = m / per;
        const int pix = (int)(m % per);
        return Row{dY + b * (i64)OH * OW * CO, pix / QW, pix % QW};
    }
    __device__ __forceinline__ float4 load4(const Row &r, int k0, int c4) const {
        if (!r.img) return make_float4(0.f, 0.f, 0.f, 0.f);
        const int slab = k0 >> 5;
        const int tp = tap[slab];
        const int oy = r.qy - (tp & 255), ox = r.qx - (tp >> 8);
        if (oy < 0 || ox < 0 || oy >= OH || ox >= OW) return make_float4(0.f, 0.f, 0.f, 0.f);
        const int co = (k0 & (CO - 1) & ~31) + c4;
        return *reinterpret_cast<const float4 *>(r.img + ((i64)oy * OW + ox) * CO + co);
    }
    void fill_taps(int K, int KWS) {
        for (int sl = 0; sl < K / 32 && sl < 80; sl++) {
            const int t = (sl * 32) / CO;
            tap[sl] = (t / KWS) | ((t % KWS) << 8);
        }
    }
};

}  // namespace

namespace {
int conv_out(int in, int k, int s, int p) { return (in + 2 * p - k) / s + 1; }

template <class AL, int BN, bool RELU, bool SPLITK>
void launch_gemm(const AL &al, const float *Bw, const float *bias, float *C, i64 M, int N, int K, int splits, hipStream_t st) {
    int kps = K;
    if (SPLITK) {
        kps = ((K / BK + splits - 1) / splits) * BK;
    }
    const i64 wgs = ((M + BM - 1) / BM) * ((N + BN - 1) / BN) * (SPLITK ? splits : 1);
    if constexpr (BN == 64) if (wgs < 200) {  // small batches: 64-row tiles double the workgroup count (one 32x32 block per wave)
        dim3 grid((unsigned)((M + 63) / 64), (unsigned)((N + BN - 1) / BN), SPLITK ? (unsigned)splits : 1u);
        hipLaunchKernelGGL((k_gemm<AL, BN, RELU, SPLITK, 64>), grid, dim3(256), 0, st, al, Bw, bias, C, M, N, K, kps);
        return;
    }
    dim3 grid((unsigned)((M + BM - 1) / BM), (unsigned)((N + BN - 1) / BN), SPLITK ? (unsigned)splits : 1u);
    hipLaunchKernelGGL((k_gemm<AL, BN, RELU, SPLITK>), grid, dim3(256), 0, st, al, Bw, bias, C, M, N, K, kps);
}

int run_dense(srlx_qnet *h, i64 B, float *d_q, hipStream_t st);

// conv2 + conv3 as implicit GEMMs over act1 (every geometry; the Atari geometry takes the fused kernel instead), then the dense layers
int run_tail(srlx_qnet *h, i64 B, float *d_q, hipStream_t st) {
    // conv2: 4x4 stride 2 pad 2 on act1 [B][OH1][OW1][F1]
    AConv c2{h->act1, h->OH1, h->OW1, h->F1, 4, 2, 2, h->OH2, h->OW2, {}};
    c2.fill_taps(16 * h->F1);
    if (h->probe0) SRLX_HIP(hipEventRecord(h->probe0, st));
    launch_gemm<AConv, 64, true, false>(c2, h->w2, h->b2, h->act2, B * h->OH2 * h->OW2, 2 * h->F1, 16 * h->F1, 1, st);
    // conv3: 3x3 stride 1 pad 1
    AConv c3{h->act2, h->OH2, h->OW2, 2 * h->F1, 3, 1, 1, h->OH3, h->OW3, {}};
    c3.fill_taps(9 * 2 * h->F1);
    launch_gemm<AConv, 64, true, false>(c3, h->w3, h->b3, h->act3, B * h->OH3 * h->OW3, 2 * h->F1, 9 * 2 * h->F1, 1, st);
    if (h->probe1) SRLX_HIP(hipEventRecord(h->probe1, st));
    h->probe0 = h->probe1 = nullptr;  // one forward only
    return d_q ? run_dense(h, B, d_q, st) : SRLX_OK;  // (d_q == NULL: the convolutions only, srlx_qnet_forward_convs_u8)
}

int run_dense(srlx_qnet *h, i64 B, float *d_q, hipStream_t st) {
    SRLX_TRY(srlx_qnet_noisy_refresh(h, st));  // NoisyLinear: one noise draw per forward call (noisy_linear.py:35-52); no-op for plain layers
    return srlx_qnet_dense_rows(h, B, 1, d_q, st);
}
}  // namespace

// The dense layers over `rows` activation rows act3[i * stride] (i < rows): FC1 split along K so that ~512 workgroups exist whatever
// the batch, then the head (split reduction + bias + ReLU, second layers, dueling combine) writing q / h1 rows i * stride.
int srlx_qnet_dense_rows(srlx_qnet *h, int64_t B, int64_t stride, float *d_q, hipStream_t st) {
    const int N1 = 2 * h->hidden;
    h->partial_used = true;
    const i64 tiles = ((B + BM - 1) / BM) * ((N1 + 63) / 64);
    int splits = (int)((512 + tiles - 1) / tiles);  // ~512 workgroups (the learner's 96 / 128 rows at 1024: no faster; at 384 / 256 / 192: +2 / +7 / +9 % per period of a learner rank)
    const int ksteps = h->flat / BK;
    if (h->fc1_neighbour > 0 && h->planes_valid && srlx_fc1_planes_applicable(h, B) && B >= 512) splits = h->fc1_neighbour;
    if (splits > ksteps) splits = ksteps;
    if (splits > h->max_splits) splits = h->max_splits;
    {   // (a narrow layer has few column tiles and would ask for more K splits than the partial-sum buffer holds rows for: srlx_qnet_set_fc1_neighbour sizes it for more)
        const size_t fit = h->partial_floats / ((size_t)((B + 127) / 128 * 128) * N1);
        if ((size_t)splits > fit) splits = (int)fit;
    }
    if (splits < 1) splits = 1;
    const int kps = ((ksteps + splits - 1) / splits);
    const int used = (ksteps + kps - 1) / kps;  // splits that actually own a K range
    static const bool fc1_f32 = getenv("SRLX_FC1_F32") && getenv("SRLX_FC1_F32")[0] == '1';  // A/B switch: FC1 on the float32 matrix pipe
    // chip-filling launches of a handle with valid weight planes (the actors' pass): conversion-free GEMM on pre-split operands, bit-identical to k_gemm_s16
    static const bool no_planes_gemm = getenv("SRLX_NO_PLANES_GEMM") && getenv("SRLX_NO_PLANES_GEMM")[0] == '1';  // measurement only (a selected set's pass then reads the BOUND float32 weight)
    const bool planes = !fc1_f32 && !no_planes_gemm && stride == 1 && h->planes_valid && !h->eff[0] && srlx_fc1_planes_applicable(h, B);
    if (planes && !h->a3_planes_fresh) SRLX_TRY(srlx_fc1_planes_split_act(h, B, st));  // (a convolution path that wrote float32 act3 only)
    h->a3_planes_fresh = false;
    if (h->probe_fc0) SRLX_HIP(hipEventRecord(h->probe_fc0, st));
    const i64 Mp = planes ? (B + 127) / 128 * 128 : B;  // row stride of the split-K partial slabs (the planes GEMM pads small launches to its 128-row tile)
    if (planes) {
        SRLX_REQUIRE((size_t)used * Mp * N1 <= h->partial_floats, "qnet: split-K partial buffer too small for %d splits of %lld rows", used, (long long)Mp);
        void *own = h->wf_planes;
        if (h->wf_planes_ext) h->wf_planes = const_cast<void *>(h->wf_planes_ext);
        const int rc = srlx_fc1_planes_gemm(h, B, splits, kps, st);
        h->wf_planes = own;
        SRLX_TRY(rc);
    } else if (!fc1_f32 && h->flat % BK == 0) {
        const dim3 grid((unsigned)((B + 127) / 128), (unsigned)((N1 + 63) / 64), (unsigned)splits);
        APlain fa{h->act3, (i64)h->flat * stride};
        hipLaunchKernelGGL((k_gemm_s16<APlain, 64, true, 128, true>), grid, dim3(256), 0, st, fa, h->wf, h->partial, B, N1, h->flat, kps * BK);
    } else {
        APlain fa{h->act3, (i64)h->flat * stride};
        launch_gemm<APlain, 64, false, true>(fa, h->wf, nullptr, h->partial, B, N1, h->flat, splits, st);
    }
    if (h->probe_fc1) SRLX_HIP(hipEventRecord(h->probe_fc1, st));
    h->probe_fc0 = h->probe_fc1 = nullptr;  // one forward only
    if (h->stamp_buf) SRLX_TRY(srlx_debug_stamp(h->stamp_buf, 11, st));  // (measurement aid: the first dense layer's launch is done)
    // small launches (the learner's 128 / 96 rows) are one workgroup per row and far from filling the chip: twice the threads per row
    const dim3 hgrid((unsigned)B), hblock(B <= 256 && h->hidden > 256 ? 512 : 256);
    i64 *const hdraw = h->sig[0] ? h->d_draw : nullptr;
    const srlx_uvfa_dev uv = srlx_uvfa_args(h);
    SRLX_REQUIRE(!uv.wx || ((uv.c_ext < 0 || uv.r_ext) && (uv.c_int < 0 || uv.r_int) && (uv.c_act < 0 || uv.action) && (uv.c_actor < 0 || uv.actor)),
                 "qnet_forward: a UVFA network needs its per-row inputs (srlx_qnet_set_uvfa_inputs)");
    if (h->head_mode == 1)  // the handle ends behind the first dense layer: d_q is the hidden layer's first out_cols units (+ LayerNorm)
        hipLaunchKernelGGL(k_hidden_out, hgrid, dim3(256), 0, st, h->partial, used, Mp, N1, h->bf, d_q, h->out_cols, h->h1, (i64)stride, h->ln_w, h->ln_b, h->ln_eps);
    else if (h->A <= 8)
        hipLaunchKernelGGL(k_head<8>, hgrid, hblock, 0, st, h->partial, used, Mp, h->hidden, h->bf, h->v2w, h->v2b, h->a2w, h->a2b, h->A, h->dueling, d_q, h->h1, (i64)stride, hdraw, h->pol, uv);
    else if (h->A <= 16)
        hipLaunchKernelGGL(k_head<16>, hgrid, hblock, 0, st, h->partial, used, Mp, h->hidden, h->bf, h->v2w, h->v2b, h->a2w, h->a2b, h->A, h->dueling, d_q, h->h1, (i64)stride, hdraw, h->pol, uv);
    else
        hipLaunchKernelGGL(k_head<32>, hgrid, hblock, 0, st, h->partial, used, Mp, h->hidden, h->bf, h->v2w, h->v2b, h->a2w, h->a2b, h->A, h->dueling, d_q, h->h1, (i64)stride, hdraw, h->pol, uv);
    h->pol = srlx_qnet::Policy{};  // one forward only
    if (h->stamp_buf) SRLX_TRY(srlx_debug_stamp(h->stamp_buf, 12, st));  // (... and the head)
    SRLX_HIP(hipGetLastError());
    return SRLX_OK;
}

// per parity class z: dXq[z][M = B*QH*QW][CI] = ADgrad(dY) x WT[z][CI][K]^T, K = (KH/S)(KW/S) CO  (srlx_qnet_bwd.hip)
int srlx_qnet_dgrad_gemm(const float *dY, int B, int QH, int QW, int OH, int OW, int CO, int KH, int KW, int S, const float *wT, int CI, float *dXq,
                         hipStream_t st, int ksplits) {
    SRLX_REQUIRE(KH % S == 0 && KW % S == 0, "dgrad_gemm: the kernel size must be a multiple of the stride");
    ADgrad a{dY, QH, QW, OH, OW, CO, {}};
    const int K = (KH / S) * (KW / S) * CO;
    SRLX_REQUIRE((CO == 32 || CO == 64) && K % BK == 0 && K / 32 <= 80 && (CI == 32 || CI == 64), "dgrad_gemm: unsupported channel counts");
    a.fill_taps(K, KW / S);
    const i64 M = (i64)B * QH * QW;
    const unsigned Z = (unsigned)(S * S);
    static const bool dgrad_f32 = getenv("SRLX_DGRAD_F32") && getenv("SRLX_DGRAD_F32")[0] == '1';  // A/B switch: the data-gradient GEMMs on the float32 matrix pipe
    if (CI == 64 && ksplits > 1) {  // stride 1 only (blockIdx.z is the K split here, the parity class otherwise): ksplits partial slabs of M x CI floats
        SRLX_REQUIRE(S == 1 && (K / BK) % ksplits == 0 && !dgrad_f32, "dgrad_gemm: K splits need stride 1 and a K that divides");
        dim3 grid((unsigned)((M + 63) / 64), 1, (unsigned)ksplits);
        hipLaunchKernelGGL((k_gemm_s16<ADgrad, 64, true, 64>), grid, dim3(256), 0, st, a, wT, dXq, M, CI, K, K / ksplits);
    } else if (CI == 64) {
        dim3 grid((unsigned)((M + 63) / 64), 1, Z);
        if (dgrad_f32)
            hipLaunchKernelGGL((k_gemm<ADgrad, 64, false, false, 64>), grid, dim3(256), 0, st, a, wT, nullptr, dXq, M, CI, K, K, (i64)CI * K, M * CI);
        else  // six exact bf16 partial products, operands split while staging (k_gemm_s16)
            hipLaunchKernelGGL((k_gemm_s16<ADgrad, 64, false, 64>), grid, dim3(256), 0, st, a, wT, dXq, M, CI, K, K, (i64)CI * K, M * CI);
    } else {
        dim3 grid((unsigned)((M + BM - 1) / BM), 1, Z);
        if (dgrad_f32)
            hipLaunchKernelGGL((k_gemm<ADgrad, 32, false, false>), grid, dim3(256), 0, st, a, wT, nullptr, dXq, M, CI, K, K, (i64)CI * K, M * CI);
        else
            hipLaunchKernelGGL((k_gemm_s16<ADgrad, 32, false>), grid, dim3(256), 0, st, a, wT, dXq, M, CI, K, K, (i64)CI * K, M * CI);
    }
    SRLX_HIP(hipGetLastError());
    return SRLX_OK;
}

extern "C" {

int srlx_qnet_create(srlx_qnet_t **out, int in_h, int in_w, int window, int filters, int hidden, int n_actions, int dueling_type, int64_t max_batch,
                     int device) {
    SRLX_REQUIRE(out, "qnet_create: out is NULL");
    SRLX_REQUIRE(in_h >= 8 && in_w >= 8 && window >= 1 && filters % 32 == 0 && hidden % 32 == 0 && n_actions >= 1 && n_actions <= kMaxActions && max_batch > 0,
                 "qnet_create: unsupported shape (filters and hidden must be multiples of 32, n_actions <= %d)", kMaxActions);
    SRLX_REQUIRE((window * 64) % BK == 0, "qnet_create: window*64 must be a multiple of %d", BK);
    SRLX_REQUIRE(filters == 32 || filters == 64 || filters == 128, "qnet_create: filters must be 32, 64 or 128 (channel counts are powers of two, <= 80 K-slabs)");
    int ndev = 0;
    SRLX_HIP(hipGetDeviceCount(&ndev));
    SRLX_REQUIRE(device >= 0 && device < ndev, "qnet_create: device %d not present", device);
    srlx::DeviceGuard guard(device);
    srlx_qnet *h = new (std::nothrow) srlx_qnet();
    if (!h) return SRLX_ERR_NOMEM;
    memset(h, 0, sizeof(*h));
    h->device = device;
    h->H = in_h, h->W = in_w, h->Wn = window, h->F1 = filters, h->hidden = hidden, h->A = n_actions, h->dueling = dueling_type;
    h->OH1 = conv_out(in_h, 8, 4, 3), h->OW1 = conv_out(in_w, 8, 4, 3);
    h->OH2 = conv_out(h->OH1, 4, 2, 2), h->OW2 = conv_out(h->OW1, 4, 2, 2);
    h->OH3 = conv_out(h->OH2, 3, 1, 1), h->OW3 = conv_out(h->OW2, 3, 1, 1);
    h->flat = h->OH3 * h->OW3 * 2 * filters;
    h->max_batch = max_batch;
    h->max_splits = 64;
    h->aset_cur = -1;
    SRLX_REQUIRE(h->flat % BK == 0, "qnet_create: flattened size %d must be a multiple of %d", h->flat, BK);
    const size_t f = sizeof(float);
    struct {
        float **p;
        size_t n;
    } bufs[] = {{&h->act1, (size_t)max_batch * h->OH1 * h->OW1 * filters},
                {&h->act2, (size_t)max_batch * h->OH2 * h->OW2 * 2 * filters},
                {&h->act3, (size_t)max_batch * h->flat},
                // FC1 split-K partial sums: splits(B) * B <= 4096 + B for every batch B (see run_tail)
                {&h->partial, (size_t)(4096 + 128 + max_batch) * 2 * hidden}};
    h->partial_floats = bufs[3].n;
    for (auto &b : bufs) {
        hipError_t e = hipMalloc((void **)b.p, b.n * f);
        if (e != hipSuccess) {
            srlx::set_error("qnet_create: %s", hipGetErrorString(e));
            srlx_qnet_destroy(h);
            return e == hipErrorOutOfMemory ? SRLX_ERR_NOMEM : SRLX_ERR_HIP;
        }
    }
    if (hipMalloc((void **)&h->range_flag, sizeof(int)) != hipSuccess || hipMemset(h->range_flag, 0, sizeof(int)) != hipSuccess) {
        srlx::set_error("qnet_create: the range flag");
        srlx_qnet_destroy(h);
        return SRLX_ERR_HIP;
    }
    *out = h;
    return SRLX_OK;
}

// Bit l of *out: an activation of convolution l + 1 exceeded 65 504 in some forward pass since the handle was created -- the two-part float16 split of the fused
// convolution kernel cannot hold it (its result is then meaningless: run the process with SRLX_CONV_BF16X3=1).  Blocking copy: call where the host has synchronised.
int srlx_qnet_range_flags(srlx_qnet_t *h, int *out) {
    SRLX_REQUIRE(h && out, "qnet_range_flags: NULL argument");
    srlx::DeviceGuard guard(h->device);
    SRLX_HIP(hipMemcpy(out, h->range_flag, sizeof(int), hipMemcpyDeviceToHost));
    return SRLX_OK;
}

int srlx_qnet_destroy(srlx_qnet_t *h) {
    if (!h) return SRLX_OK;
    srlx::DeviceGuard guard(h->device);
    (void)hipDeviceSynchronize();
    float *all[] = {h->act1, h->act2, h->act3, h->partial, h->h1, h->dh1, h->dh1t, h->dact3, h->dact2, h->dact1, h->fc_part, h->w_part, h->dxpad, h->w_t, h->w_t2};
    for (float *p : all)
        if (p) (void)hipFree(p);
    for (float *p : h->eff)
        if (p) (void)hipFree(p);
    if (h->range_flag) (void)hipFree(h->range_flag);
    if (h->c1_gpart) (void)hipFree(h->c1_gpart);
    if (h->c1_cnt) (void)hipFree(h->c1_cnt);
    if (h->d_draw) (void)hipFree(h->d_draw);
    if (h->step_snap) (void)hipFree(h->step_snap);
    if (h->aset_cur >= 0) h->wpack = h->wpack_own, h->wf_planes = h->wf_planes_own;
    for (auto &st_ : h->aset) {
        if (st_.wpack) (void)hipFree(st_.wpack);
        if (st_.wf_planes) (void)hipFree(st_.wf_planes);
        if (st_.small) (void)hipFree(st_.small);
    }
    if (h->wpack) (void)hipFree(h->wpack);
    if (h->wf_planes) (void)hipFree(h->wf_planes);
    if (h->a3_planes) (void)hipFree(h->a3_planes);
    if (h->side && !h->side_external) (void)hipStreamDestroy(h->side);
    if (h->side2) (void)hipStreamDestroy(h->side2);
    for (hipEvent_t e : {h->ev_fork, h->ev_d3, h->ev_d2, h->ev_d1, h->ev_join, h->ev_wt, h->ev_join2})
        if (e) (void)hipEventDestroy(e);
    delete h;
    return SRLX_OK;
}

int srlx_qnet_bind(srlx_qnet_t *h, const float *const *p) {
    SRLX_REQUIRE(h && p, "qnet_bind: NULL argument");
    for (int i = 0; i < 12; i++) SRLX_REQUIRE(p[i], "qnet_bind: parameter %d is NULL", i);
    for (int i = 0; i < 12; i++) h->bound[i] = p[i];
    h->pack_valid = false;  // other weights: the packed filters are stale
    if (h->aset_cur >= 0) {  // forwards keep reading the selected set; the filters / first dense layer it was made from are the bound ones
        h->w1 = p[0], h->w2 = p[2], h->w3 = p[4], h->wf = p[6];
        h->pack_valid = true;
        return SRLX_OK;
    }
    h->w1 = p[0], h->b1 = p[1], h->w2 = p[2], h->b2 = p[3], h->w3 = p[4], h->b3 = p[5];
    if (h->eff[0]) {  // NoisyLinear: the dense-layer entries are the mu tensors; the kernels keep reading the effective tensors
        for (int t = 0; t < 6; t++) h->mu[t] = p[6 + t];
        return SRLX_OK;
    }
    h->wf = p[6], h->bf = p[7], h->v2w = p[8], h->v2b = p[9], h->a2w = p[10], h->a2b = p[11];
    return SRLX_OK;
}

int srlx_qnet_enable_fc1_planes(srlx_qnet_t *h) {
    SRLX_REQUIRE(h, "qnet_enable_fc1_planes: NULL handle");
    SRLX_REQUIRE(!h->eff[0], "qnet_enable_fc1_planes: NoisyLinear layers draw a new effective weight per forward: no persistent planes");
    srlx::DeviceGuard guard(h->device);
    return srlx_fc1_planes_alloc(h);
}

int srlx_qnet_refresh_fc1_planes(srlx_qnet_t *h, const float *d_src_wf, float *d_copy_dst, void *stream) {
    SRLX_REQUIRE(h && h->wf_planes, "qnet_refresh_fc1_planes: planes are not enabled on this handle (srlx_qnet_enable_fc1_planes)");
    SRLX_REQUIRE(d_src_wf || h->wf, "qnet_refresh_fc1_planes: no weight bound and none given");
    srlx::DeviceGuard guard(h->device);
    SRLX_TRY(srlx_fc1_planes_split_weight(h, d_src_wf ? d_src_wf : h->wf, d_copy_dst, (hipStream_t)stream));
    h->planes_valid = true;
    return SRLX_OK;
}

int srlx_qnet_invalidate_fc1_planes(srlx_qnet_t *h) {
    SRLX_REQUIRE(h, "qnet_invalidate_fc1_planes: NULL handle");
    h->planes_valid = false;
    return SRLX_OK;
}

int srlx_qnet_set_side_stream(srlx_qnet_t *h, void *stream) {
    SRLX_REQUIRE(h && stream, "qnet_set_side_stream: NULL argument");
    SRLX_REQUIRE(h->max_train > 0, "qnet_set_side_stream: call srlx_qnet_enable_training first");
    if (h->side && !h->side_external) (void)hipStreamDestroy(h->side);
    h->side = (hipStream_t)stream;
    h->side_external = true;
    return SRLX_OK;
}

int srlx_qnet_set_debug(srlx_qnet_t *h, void *d_phase_stamps) {
    SRLX_REQUIRE(h, "qnet_set_debug: NULL handle");
    h->fused_dbg = d_phase_stamps;
    return SRLX_OK;
}

int srlx_qnet_set_probe(srlx_qnet_t *h, void *ev_start, void *ev_end) {
    SRLX_REQUIRE(h, "qnet_set_probe: NULL handle");
    h->probe0 = (hipEvent_t)ev_start;
    h->probe1 = (hipEvent_t)ev_end;
    return SRLX_OK;
}

int srlx_qnet_set_fc1_span(srlx_qnet_t *h, uint64_t *d_span) {
    SRLX_REQUIRE(h, "qnet_set_fc1_span: NULL handle");
    h->fc1_span = d_span;
    return SRLX_OK;
}

int srlx_qnet_set_probe_fc1(srlx_qnet_t *h, void *ev_start, void *ev_end) {
    SRLX_REQUIRE(h, "qnet_set_probe_fc1: NULL handle");
    h->probe_fc0 = (hipEvent_t)ev_start;
    h->probe_fc1 = (hipEvent_t)ev_end;
    return SRLX_OK;
}

// ---- packed filters that outlive a forward, and the actors' published parameter sets (srlx_qnet_int.h) ----------------------------------------------------
int srlx_qnet_weights_changed(srlx_qnet_t *h) {
    SRLX_REQUIRE(h, "qnet_weights_changed: NULL handle");
    h->pack_valid = h->aset_cur >= 0;  // (a selected set is what it is: republish it)
    if (h->aset_cur < 0) h->planes_valid = false;
    return SRLX_OK;
}

int srlx_qnet_set_pack_sticky(srlx_qnet_t *h, int on) {
    SRLX_REQUIRE(h, "qnet_set_pack_sticky: NULL handle");
    h->pack_sticky = on != 0;
    if (!on && h->aset_cur < 0) h->pack_valid = false;
    return SRLX_OK;
}

int srlx_qnet_actor_sets_enable(srlx_qnet_t *h) {
    SRLX_REQUIRE(h, "qnet_actor_sets_enable: NULL handle");
    SRLX_REQUIRE(!h->eff[0], "qnet_actor_sets_enable: NoisyLinear layers draw new effective weights per forward: nothing to publish");
    SRLX_REQUIRE(h->H == 84 && h->W == 84 && h->Wn == 4 && h->F1 == 32, "qnet_actor_sets_enable: the published sets serve the fused convolution kernel's geometry (84 x 84 x 4, 32 filters)");
    srlx::DeviceGuard guard(h->device);
    SRLX_TRY(srlx_fc1_planes_alloc(h));
    const srlx_small_layout L = srlx_small_offsets(h);
    for (auto &st_ : h->aset) {
        if (st_.wpack) continue;
        SRLX_HIP(hipMalloc((void **)&st_.wpack, srlx_qnet_pack_bytes()));
        SRLX_HIP(hipMalloc((void **)&st_.wf_planes, srlx_fc1_planes_weight_bytes(h)));
        SRLX_HIP(hipMalloc((void **)&st_.small, (size_t)L.total * sizeof(float)));
        SRLX_HIP(hipMemset(st_.small, 0, (size_t)L.total * sizeof(float)));
    }
    return SRLX_OK;
}

int srlx_qnet_actor_set_planes(srlx_qnet_t *h, int set, void **d_planes) {
    SRLX_REQUIRE(h && d_planes && (set == 0 || set == 1) && h->aset[set].wf_planes, "qnet_actor_set_planes: bad argument (srlx_qnet_actor_sets_enable first)");
    *d_planes = h->aset[set].wf_planes;
    return SRLX_OK;
}

int srlx_qnet_actor_set_select(srlx_qnet_t *h, int set) {
    SRLX_REQUIRE(h && set >= -1 && set <= 1, "qnet_actor_set_select: bad argument");
    SRLX_REQUIRE(set < 0 || h->aset[set].wpack, "qnet_actor_set_select: srlx_qnet_actor_sets_enable first");
    if (h->aset_cur < 0 && set >= 0) h->wpack_own = h->wpack, h->wf_planes_own = h->wf_planes;
    if (set < 0) {
        if (h->aset_cur >= 0) {
            h->wpack = h->wpack_own, h->wf_planes = h->wf_planes_own;
            h->b1 = h->bound[1], h->b2 = h->bound[3], h->b3 = h->bound[5];
            h->bf = h->bound[7], h->v2w = h->bound[8], h->v2b = h->bound[9], h->a2w = h->bound[10], h->a2b = h->bound[11];
            h->uvfa.wx = h->uvfa.wx_bound;
            h->pack_valid = false, h->planes_valid = false;
        }
        h->aset_cur = -1;
        return SRLX_OK;
    }
    const srlx_small_layout L = srlx_small_offsets(h);
    const srlx_qnet::ActorSet &a = h->aset[set];
    h->wpack = a.wpack, h->wf_planes = a.wf_planes;
    h->b1 = a.small + L.b1, h->b2 = a.small + L.b2, h->b3 = a.small + L.b3;
    h->bf = a.small + L.bf, h->v2w = a.small + L.v2w, h->v2b = a.small + L.v2b, h->a2w = a.small + L.a2w, h->a2b = a.small + L.a2b;
    if (h->uvfa.X > 0) h->uvfa.wx = a.small + L.wx;
    h->pack_valid = true, h->planes_valid = true;
    h->aset_cur = set;
    return SRLX_OK;
}

// Packs `h_src`'s convolution filters for ITS next forwards (they then skip k_pack_filters until srlx_qnet_weights_changed / the next publish) and, with `h_actor`,
// publishes the network into that handle's set `set`: packed filters + small vectors in the same launch; `with_fc1`: also the first dense layer's weight as
// operand planes (one splitting pass -- the initial / out-of-band publish; an update publishes them from the fused Adam's epilogue, srlx_qnet_fuse_adam_fc1_planes).
int srlx_qnet_publish(srlx_qnet_t *h_src, srlx_qnet_t *h_actor, int set, int with_fc1, int64_t *d_bump, void *stream) {
    SRLX_REQUIRE(h_src && h_src->bound[0], "qnet_publish: no parameters bound on the source handle");
    SRLX_REQUIRE(!h_src->eff[0], "qnet_publish: NoisyLinear source");
    SRLX_REQUIRE(h_src->H == 84 && h_src->W == 84 && h_src->Wn == 4 && h_src->F1 == 32, "qnet_publish: the fused convolution kernel's geometry only");
    srlx::DeviceGuard guard(h_src->device);
    hipStream_t st = (hipStream_t)stream;
    if (!h_actor) {
        SRLX_TRY(srlx_qnet_pack_publish(h_src, nullptr, nullptr, st, d_bump, true));
        h_src->pack_valid = true;
        return SRLX_OK;
    }
    SRLX_REQUIRE((set == 0 || set == 1) && h_actor->aset[set].wpack, "qnet_publish: srlx_qnet_actor_sets_enable on the actor handle first");
    SRLX_REQUIRE(h_actor->hidden == h_src->hidden && h_actor->A == h_src->A && h_actor->flat == h_src->flat && h_actor->uvfa.X == h_src->uvfa.X,
                 "qnet_publish: the two handles describe different networks");
    const srlx_small_layout L = srlx_small_offsets(h_actor);
    SRLX_TRY(srlx_qnet_pack_publish(h_src, &h_actor->aset[set], &L, st, d_bump, true));
    if (h_src->aset_cur < 0) h_src->pack_valid = true;
    if (with_fc1) SRLX_TRY(srlx_fc1_planes_split_weight(h_actor, h_src->bound[6], nullptr, st, h_actor->aset[set].wf_planes));
    return SRLX_OK;
}

// splits > 0: the chip-filling first-dense-layer launches of this handle (operand planes) use half-CU workgroups (k_fc1_planes_h) with `splits` K splits -- for a
// handle whose passes run BESIDE a learner; 0: the CU-filling k_fc1_planes (fastest alone).
int srlx_qnet_set_fc1_neighbour(srlx_qnet_t *h, int splits) {
    SRLX_REQUIRE(h && splits >= 0 && splits <= 64, "qnet_set_fc1_neighbour: bad argument");
    srlx::DeviceGuard guard(h->device);
    const size_t need = (size_t)splits * h->max_batch * 2 * h->hidden;
    if (need > h->partial_floats) {  // split-K partial slabs of the chip-filling launch
        SRLX_REQUIRE(!h->partial_used, "qnet_set_fc1_neighbour: %d splits need a larger partial-sum buffer, and a forward (possibly captured in a graph) already uses the "
                                       "current one: choose the split count before the first forward", splits);
        SRLX_HIP(hipDeviceSynchronize());
        if (h->partial) SRLX_HIP(hipFree(h->partial));
        h->partial = nullptr;
        SRLX_HIP(hipMalloc((void **)&h->partial, need * sizeof(float)));
        h->partial_floats = need;
    }
    h->fc1_neighbour = splits;
    return SRLX_OK;
}

// A learner's handle on operand planes: `on` -- planes also for launches below 512 rows (its convolution kernel then writes float32 act3 AND planes, the first
// dense layer runs on the half-CU planes kernel with the split-K shape of the staging-split GEMM: bit-identical); d_weight_planes -- BORROWED planes of the bound
// weight for the next forwards (NULL: the handle's own, srlx_qnet_refresh_fc1_planes).  The caller vouches that the planes hold the bound weight.
int srlx_qnet_set_planes_small(srlx_qnet_t *h, int on, const void *d_weight_planes) {
    SRLX_REQUIRE(h, "qnet_set_planes_small: NULL handle");
    SRLX_REQUIRE(!on || h->a3_planes, "qnet_set_planes_small: srlx_qnet_enable_fc1_planes first");
    h->planes_small = on != 0;
    h->wf_planes_ext = d_weight_planes;
    if (on && h->fc1_neighbour <= 0) h->fc1_neighbour = 4;
    if (d_weight_planes) h->planes_valid = true;
    return SRLX_OK;
}

int srlx_qnet_set_priority_sink(srlx_qnet_t *h, srlx_per_t *per, int64_t n, const int64_t *d_indices, const void *d_priorities, int prio_kind) {
    SRLX_REQUIRE(h, "qnet_set_priority_sink: NULL handle");
    SRLX_REQUIRE(!per || (h->max_train > 0 && n > 0 && d_indices && d_priorities), "qnet_set_priority_sink: enable training first; n > 0 and device pointers");
    h->sink_per = per, h->sink_n = n, h->sink_idx = d_indices, h->sink_prio = d_priorities, h->sink_kind = prio_kind;
    return SRLX_OK;
}

int srlx_qnet_set_sink_done(srlx_qnet_t *h, void *event) {
    SRLX_REQUIRE(h, "qnet_set_sink_done: NULL handle");
    h->sink_done = (hipEvent_t)event;
    return SRLX_OK;
}

int srlx_qnet_set_main_first(srlx_qnet_t *h, int on) {
    SRLX_REQUIRE(h, "qnet_set_main_first: NULL handle");
    h->main_first = on != 0;
    return SRLX_OK;
}

int srlx_qnet_set_dgrad_split(srlx_qnet_t *h, int splits) {
    SRLX_REQUIRE(h && (splits == 1 || splits == 2), "qnet_set_dgrad_split: 1 or 2");
    h->dgrad_split = splits;
    return SRLX_OK;
}

int srlx_qnet_set_sink_stream(srlx_qnet_t *h, void *stream) {
    SRLX_REQUIRE(h, "qnet_set_sink_stream: NULL handle");
    h->sink_stream = (hipStream_t)stream;
    return SRLX_OK;
}

int srlx_qnet_set_sink_wait(srlx_qnet_t *h, void *event) {
    SRLX_REQUIRE(h, "qnet_set_sink_wait: NULL handle");
    h->sink_wait = (hipEvent_t)event;
    return SRLX_OK;
}

int srlx_qnet_set_fc1_branch(srlx_qnet_t *h, int order) {
    SRLX_REQUIRE(h && order >= 0 && order <= 2, "qnet_set_fc1_branch: order 0 (last on the weight-gradient branch), 1 (first on it) or 2 (a branch of its own)");
    h->fc1_order = order;
    return SRLX_OK;
}

int srlx_qnet_set_stamp_buffer(srlx_qnet_t *h, uint64_t *d_buf) {
    SRLX_REQUIRE(h, "qnet_set_stamp_buffer: NULL handle");
    h->stamp_buf = d_buf;
    return SRLX_OK;
}

int srlx_qnet_set_td_event(srlx_qnet_t *h, void *event) {
    SRLX_REQUIRE(h, "qnet_set_td_event: NULL handle");
    h->ev_td = (hipEvent_t)event;
    return SRLX_OK;
}

int srlx_qnet_fuse_adam_fc1_planes(srlx_qnet_t *h, void *d_planes_out) {
    SRLX_REQUIRE(h, "qnet_fuse_adam_fc1_planes: NULL handle");
    SRLX_REQUIRE(!d_planes_out || h->adam_m, "qnet_fuse_adam_fc1_planes: srlx_qnet_fuse_adam_fc1 first (the planes ride on the fused Adam epilogue)");
    h->adam_planes_out = d_planes_out;
    return SRLX_OK;
}

// ---- round 6: Agent57(_light)'s networks on the handle (srlx_qnet_int.h) ------------------------------------------------------------------------------------
int srlx_qnet_bind_uvfa(srlx_qnet_t *h, const float *d_wx, int n_cols, int col_ext, int col_int, int col_action, int n_action_in, int col_actor, int n_actor) {
    SRLX_REQUIRE(h && d_wx && n_cols > 0 && n_cols <= 256, "qnet_bind_uvfa: bad argument");
    SRLX_REQUIRE(h->uvfa.X == 0 || h->uvfa.X == n_cols, "qnet_bind_uvfa: the column count of a handle is fixed by its first binding");
    SRLX_REQUIRE(h->uvfa.X == n_cols || !h->aset[0].small, "qnet_bind_uvfa: bind before srlx_qnet_actor_sets_enable (the sets hold a copy of the columns)");
    SRLX_REQUIRE(!h->eff[0] && h->head_mode == 0, "qnet_bind_uvfa: plain dueling handles only");
    auto ok = [&](int c, int n) { return c < 0 || (n > 0 && c + n <= n_cols); };
    SRLX_REQUIRE(ok(col_ext, 1) && ok(col_int, 1) && ok(col_action, n_action_in) && ok(col_actor, n_actor), "qnet_bind_uvfa: a column range leaves the matrix");
    srlx_qnet::Uvfa &u = h->uvfa;
    u.wx_bound = d_wx;
    if (h->aset_cur < 0) u.wx = d_wx;
    u.X = n_cols, u.c_ext = col_ext, u.c_int = col_int, u.c_act = col_action, u.n_act_in = n_action_in, u.c_actor = col_actor, u.n_actor = n_actor;
    return SRLX_OK;
}

int srlx_qnet_set_uvfa_inputs(srlx_qnet_t *h, const float *d_r_ext, const float *d_r_int, const int32_t *d_action, const int32_t *d_actor) {
    SRLX_REQUIRE(h && h->uvfa.X > 0, "qnet_set_uvfa_inputs: srlx_qnet_bind_uvfa first");
    h->uvfa.r_ext = d_r_ext, h->uvfa.r_int = d_r_int, h->uvfa.action = d_action, h->uvfa.actor = d_actor;
    return SRLX_OK;
}

int srlx_qnet_fuse_adam_uvfa(srlx_qnet_t *h, float *d_grad_wx, float *d_exp_avg, float *d_exp_avg_sq) {
    SRLX_REQUIRE(h && h->uvfa.X > 0 && h->max_train > 0, "qnet_fuse_adam_uvfa: a training handle with UVFA columns");
    SRLX_REQUIRE(d_grad_wx && (!d_exp_avg || (d_exp_avg_sq && h->rest_on)), "qnet_fuse_adam_uvfa: the optimiser step rides on srlx_qnet_fuse_adam_rest's packing launch");
    h->uvfa.g_wx = d_grad_wx, h->uvfa.m_wx = d_exp_avg, h->uvfa.v_wx = d_exp_avg_sq;
    return SRLX_OK;
}

int srlx_qnet_set_td_extras(srlx_qnet_t *h, const float *d_discount_per_sample, float *d_td_signed) {
    SRLX_REQUIRE(h, "qnet_set_td_extras: NULL handle");
    h->td_disc_ps = d_discount_per_sample, h->td_signed = d_td_signed;
    return SRLX_OK;
}

int srlx_qnet_set_head_mode(srlx_qnet_t *h, int mode, int out_cols, const float *d_ln_w, const float *d_ln_b, double ln_eps) {
    SRLX_REQUIRE(h && (mode == 0 || mode == 1), "qnet_set_head_mode: mode 0 (dueling head) or 1 (hidden layer out)");
    SRLX_REQUIRE(mode == 0 || (out_cols > 0 && out_cols <= 2 * h->hidden && 2 * h->hidden <= 1024 && !h->eff[0] && h->uvfa.X == 0),
                 "qnet_set_head_mode: 1 <= out_cols <= 2 * hidden <= 1024, plain layers");
    SRLX_REQUIRE(!d_ln_w == !d_ln_b, "qnet_set_head_mode: LayerNorm needs weight and bias");
    h->head_mode = mode, h->out_cols = out_cols, h->ln_w = d_ln_w, h->ln_b = d_ln_b, h->ln_eps = (float)ln_eps;
    return SRLX_OK;
}

// conv1 -> conv2 -> conv3 from the uint8 ring into h->act3, then (d_q != NULL) the dense layers
static int forward_u8_impl(srlx_qnet_t *h, int64_t batch, const uint8_t *d_frame_base, const int64_t *d_frame_off, float *d_q, hipStream_t st) {
    static const bool no_fused = getenv("SRLX_NO_FUSED_CONV") && getenv("SRLX_NO_FUSED_CONV")[0] == '1';  // A/B switch for measurements
    if (!no_fused && h->H == 84 && h->W == 84 && h->Wn == 4 && h->F1 == 32) {
        // conv1 -> conv2 -> conv3 in one kernel, one workgroup per sample, activations in LDS (srlx_qnet_fused.hip)
        // (the probe events are recorded inside, right around k_convnet_fused: the filter-packing launch before it is not part of the timed kernel)
        h->want_planes_out = d_q != nullptr;
        SRLX_REQUIRE(srlx_qnet_fused_convs(h, batch, d_frame_base, d_frame_off, st), "qnet_forward_u8: launching the fused convolution kernel failed");
        h->probe0 = h->probe1 = nullptr;
        return d_q ? run_dense(h, batch, d_q, st) : SRLX_OK;
    }
    h->wt_from_forward = false;
    const size_t lds = (size_t)h->Wn * kC1Frame;
    if (h->F1 == 32 && h->Wn == 4 && 4 * (h->OH1 - 1) + 8 <= kC1Pad && 4 * (h->OW1 - 1) + 8 <= kC1Pad && h->W % 4 == 0) {
        // one workgroup per sample, frames + filters staged in LDS
        hipLaunchKernelGGL(k_conv1_u8, dim3((unsigned)batch, batch < 200 ? 2u : 1u), dim3(256), lds, st, d_frame_base, d_frame_off, h->Wn, h->H, h->W, h->OH1, h->OW1, h->w1, h->b1,
                           h->act1);
    } else {
        AU8 c1{d_frame_base, d_frame_off, h->Wn, h->H, h->W, 4, 3, h->OH1, h->OW1};
        launch_gemm<AU8, 32, true, false>(c1, h->w1, h->b1, h->act1, batch * h->OH1 * h->OW1, h->F1, h->Wn * 64, 1, st);
    }
    return run_tail(h, batch, d_q, st);
}

int srlx_qnet_forward_u8(srlx_qnet_t *h, int64_t batch, const uint8_t *d_frame_base, const int64_t *d_frame_off, float *d_q, void *stream) {
    SRLX_REQUIRE(h && d_frame_base && d_frame_off && d_q, "qnet_forward_u8: NULL argument");
    SRLX_REQUIRE(h->w1, "qnet_forward_u8: no parameters bound (srlx_qnet_bind)");
    SRLX_REQUIRE(batch > 0 && batch <= h->max_batch, "qnet_forward_u8: batch %lld exceeds max_batch %lld", (long long)batch, (long long)h->max_batch);
    srlx::DeviceGuard guard(h->device);
    return forward_u8_impl(h, batch, d_frame_base, d_frame_off, d_q, (hipStream_t)stream);
}

int srlx_qnet_forward_u8_policy(srlx_qnet_t *h, int64_t batch, const uint8_t *d_frame_base, const int64_t *d_frame_off, float *d_q, const float *d_eps, uint64_t seed,
                                const int64_t *d_counter, const uint8_t *d_invalid, int32_t *d_actions, float *d_q_copy, void *stream) {
    SRLX_REQUIRE(h && d_frame_base && d_frame_off && d_q && d_eps && d_counter && d_actions, "qnet_forward_u8_policy: NULL argument");
    SRLX_REQUIRE(h->bound[0], "qnet_forward_u8_policy: no parameters bound (srlx_qnet_bind)");
    SRLX_REQUIRE(batch > 0 && batch <= h->max_batch, "qnet_forward_u8_policy: batch %lld exceeds max_batch %lld", (long long)batch, (long long)h->max_batch);
    srlx::DeviceGuard guard(h->device);
    h->pol = srlx_qnet::Policy{d_eps, (unsigned long long)seed, d_counter, d_invalid, d_actions, d_q_copy};
    const int rc = forward_u8_impl(h, batch, d_frame_base, d_frame_off, d_q, (hipStream_t)stream);
    h->pol = srlx_qnet::Policy{};
    return rc;
}

int srlx_qnet_forward_convs_u8(srlx_qnet_t *h, int64_t batch, const uint8_t *d_frame_base, const int64_t *d_frame_off, float *d_features, void *stream) {
    SRLX_REQUIRE(h && d_frame_base && d_frame_off && d_features, "qnet_forward_convs_u8: NULL argument");
    SRLX_REQUIRE(h->w1, "qnet_forward_convs_u8: no parameters bound (srlx_qnet_bind)");
    SRLX_REQUIRE(batch > 0 && batch <= h->max_batch, "qnet_forward_convs_u8: batch %lld exceeds max_batch %lld", (long long)batch, (long long)h->max_batch);
    srlx::DeviceGuard guard(h->device);
    hipStream_t st = (hipStream_t)stream;
    SRLX_TRY(forward_u8_impl(h, batch, d_frame_base, d_frame_off, nullptr, st));
    SRLX_HIP(hipMemcpyAsync(d_features, h->act3, (size_t)batch * h->flat * sizeof(float), hipMemcpyDeviceToDevice, st));
    return SRLX_OK;
}

int srlx_qnet_forward_convs_multi_u8(srlx_qnet_t *const *hs, int n, int64_t batch, const uint8_t *d_frame_base, const int64_t *d_frame_off, void *stream) {
    SRLX_REQUIRE(hs && d_frame_base && d_frame_off, "qnet_forward_convs_multi_u8: NULL argument");
    srlx::DeviceGuard guard(hs[0]->device);
    return srlx_qnet_fused_convs_multi(hs, n, batch, d_frame_base, d_frame_off, (hipStream_t)stream);
}

int srlx_qnet_forward_dense_planes(srlx_qnet_t *h, int64_t batch, float *d_q, void *stream) {
    SRLX_REQUIRE(h && d_q, "qnet_forward_dense_planes: NULL argument");
    SRLX_REQUIRE(h->a3_planes_fresh && batch > 0 && batch <= h->max_batch, "qnet_forward_dense_planes: no fresh operand planes (srlx_qnet_forward_convs_multi_u8 first)");
    srlx::DeviceGuard guard(h->device);
    return run_dense(h, batch, d_q, (hipStream_t)stream);
}

int srlx_qnet_forward_f32(srlx_qnet_t *h, int64_t batch, const float *d_obs_nchw, float *d_q, void *stream) {
    SRLX_REQUIRE(h && d_obs_nchw && d_q, "qnet_forward_f32: NULL argument");
    SRLX_REQUIRE(h->w1, "qnet_forward_f32: no parameters bound (srlx_qnet_bind)");
    SRLX_REQUIRE(batch > 0 && batch <= h->max_batch, "qnet_forward_f32: batch %lld exceeds max_batch %lld", (long long)batch, (long long)h->max_batch);
    srlx::DeviceGuard guard(h->device);
    hipStream_t st = (hipStream_t)stream;
    h->wt_from_forward = false;
    ANchw c1{d_obs_nchw, h->Wn, h->H, h->W, 4, 3, h->OH1, h->OW1};
    launch_gemm<ANchw, 32, true, false>(c1, h->w1, h->b1, h->act1, batch * h->OH1 * h->OW1, h->F1, h->Wn * 64, 1, st);
    return run_tail(h, batch, d_q, st);
}

}  // extern "C"
