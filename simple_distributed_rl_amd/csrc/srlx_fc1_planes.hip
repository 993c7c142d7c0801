// srlx_fc1_planes.hip -- the first dense layer of the ACTORS' pass as a conversion-free GEMM on pre-split float16 operand planes.
//
// FC1 (srl/rl/torch_/blocks/dueling_network.py:8-59 behind rainbow/model_torch.py:55-67: [rows][7744] x [2*hidden][7744]^T) evaluates
// float32 x float32 as exact partial products on the 16-bit matrix pipe.  Rounds 3-5: six products of three bf16 parts.  Round 6: THREE products of two float16
// parts, x = hi + lo / 2048 with hi = f16(x), lo = f16((x - hi) * 2048) (derivation and error: srlx_qnet_fused.hip; k_gemm_s16<.., H16> in srlx_qnet.hip is the
// same arithmetic with the split done while staging).  For the chip-filling launch of the actors (1024 rows per lock-step) both operands arrive ALREADY split:
//   * the activations: the convolution kernel's conv3 epilogue writes act3 as planes (srlx_qnet_fused.hip, PLANES), and
//   * the weight: the update's fused Adam writes the planes of the published set (srlx_qnet_bwd.hip: k_fc1_wgrad<.., PLANES>); out of band: k_split_planes.
// Plane layout: activations [K/32 slabs][rows][4 k-groups][2 parts][8 f16], weight [K/32 slabs][rows][2 parts][4 k-groups][8 f16] (a part's 32 k of a row are
// 64 contiguous bytes: what a lane group of the weight-gradient kernel's Adam epilogue holds; the LDS image of a tile is the same for both, only the SOURCE chunk
// index of the DMA differs) -- the 16-byte chunk (k-group g, part p) IS the fragment a lane feeds v_mfma_f32_32x32x16_f16 (row i, k = 8 g .. 8 g + 7); a row's
// 32-deep K-slab is ONE 128-byte line (8 chunks: 4 bytes per value, the float32's own size -- the planes of rounds 3-5 were 6), and the slab-major order makes a
// 128-row operand tile of a K-slab one contiguous 16 KB run.
//
// The GEMM (k_fc1_planes_h): 128 x 128 output tile per workgroup of 4 waves (64 x 64 each: four accumulator pairs), K split over blockIdx.z like k_gemm_s16 (partials
// reduced by k_head), operand tiles fetched by LDS-DMA (global_load_lds_dwordx4: no VGPR round trip, no ds_write) into a three-slot LDS ring of whole K-slabs (96 KB:
// 64 KB of the CU stay free for the learner's kernels), one barrier per K-slab.  LDS image of a tile: row r = 8 chunk slots, chunk c = k-group * 2 + part stored at
// slot (c + ((r >> 1) & 7)) mod 8 -- the rotation makes the 16 rows a ds_read_b128 lane group touches ({0-3, 12-15, 20-27} / {4-11, 16-19, 28-31}:
// MI355X_MICROARCH.md, LDS) hit 16 different 16-byte bank groups ((8 r + slot) mod 16: r's parity picks the half, (r >> 1) mod 8 is distinct within each parity
// class of both groups); the rotation is applied on the SOURCE address of the DMA (its LDS side is lane-linear).
// Same K split, same k order and the same order of the three partial products as k_gemm_s16<.., H16>: the two paths are bit-identical.
// (Rounds 3-5 also had a CU-filling 512-thread variant, 83 us alone against this one's 97: with half the matrix work per K-slab the DMA ring is what either is
// bound by, and the variant is gone; srlx_qnet_set_fc1_neighbour(0) now only selects the generic K split.)
#include "srlx_qnet_int.h"

namespace {

using i64 = int64_t;
using f32x16 = __attribute__((ext_vector_type(16))) float;
using f16x8 = __attribute__((ext_vector_type(8))) _Float16;
typedef __attribute__((address_space(1))) const void gptr_t;
typedef __attribute__((address_space(3))) void lptr_t;

constexpr int kTM = 128, kTN = 128;         // output tile

// float32 [rows][K] -> planes [K/32][rows][8 chunks][8 f16] (+ an optional float32 copy: the actors' private weight and its planes in one pass);
// one thread per 8 consecutive k of a row.  hi = f16(x), lo = f16((x - hi) * 2048)
template <bool WEIGHT>  // WEIGHT: the weight's chunk order [part][k-group], else the activations' [k-group][part]
__global__ void __launch_bounds__(256) k_split_planes(const float *__restrict__ src, i64 rows, int K8, f16x8 *__restrict__ planes, float *__restrict__ copy) {
    const i64 q = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= rows * K8) return;
    const i64 row = q / K8;
    const int k8 = (int)(q % K8);
    const float4 x0 = reinterpret_cast<const float4 *>(src)[2 * q], x1 = reinterpret_cast<const float4 *>(src)[2 * q + 1];
    if (copy) {
        reinterpret_cast<float4 *>(copy)[2 * q] = x0;
        reinterpret_cast<float4 *>(copy)[2 * q + 1] = x1;
    }
    const float r[8] = {x0.x, x0.y, x0.z, x0.w, x1.x, x1.y, x1.z, x1.w};
    f16x8 hp, lp;
#pragma unroll
    for (int j = 0; j < 8; j++) {
        const _Float16 hi = (_Float16)r[j];
        hp[j] = hi, lp[j] = (_Float16)((r[j] - (float)hi) * 2048.0f);
    }
    f16x8 *dst = planes + ((i64)(k8 >> 2) * rows + row) * 8;
    dst[WEIGHT ? (k8 & 3) : (k8 & 3) * 2] = hp;
    dst[WEIGHT ? 4 + (k8 & 3) : (k8 & 3) * 2 + 1] = lp;
}

// ---- the GEMM ------------------------------------------------------------------------------------------------------------------------------------------------------
// 256 threads (one wave per SIMD), stages of one K-slab (32 k: 16 KB per operand tile), a three-slot ring.  In iteration st a wave multiplies stage st out of
// REGISTERS (fragments read during iteration st - 1), reads the fragments of stage st + 1 out of LDS, and issues the LDS-DMA of stage st + 3 into the slot stage st
// has just left -- the 16 reads and the 8 DMA pieces threaded between the 24 MFMAs.  An LDS-DMA is ordered for a reader only by the issuing wave's counted vmcnt
// followed by a barrier the reader has passed; the barrier is the raw instruction (__syncthreads() would drain vmcnt(0) and with it the stages that should keep
// travelling).  The split-K partial sums are added by k_head in split order, so the result depends on the NUMBER of splits only through float32 association.
constexpr int kHRow = 128;                     // bytes per row of a K-slab tile: 4 k-groups x 2 parts x 16 B
constexpr int kHTile = kTM * kHRow;            // 16 KB
constexpr int kHBuf = 2 * kHTile;              // A + B
constexpr int kHStages = 3;
constexpr size_t kHLds = kHStages * kHBuf;     // 96 KB

__global__ void __launch_bounds__(256) k_fc1_planes_h(const uint4 *__restrict__ A, const uint4 *__restrict__ W, float *__restrict__ C, int M, int N, int K8,
                                                      int slabs_per_split, unsigned long long *__restrict__ span) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    if (span && threadIdx.x == 0) atomicMin(&span[0], (unsigned long long)wall_clock64());  // srlx_qnet_set_fc1_span: first workgroup in
    const int t = threadIdx.x, wave = t >> 6, lane = t & 63, i = lane & 31, h = lane >> 5;
    unsigned bx = blockIdx.x, by = blockIdx.y, bz = blockIdx.z;
    {   // XCD-aware tile order: XCD x = linear id % 8 gets the x-th contiguous eighth of the (split, N tile, M tile) space
        const unsigned gx = gridDim.x, gy = gridDim.y, total = gx * gy * gridDim.z;
        if (total % 8 == 0) {
            const unsigned lin = bx + gx * (by + gy * bz), tile = (lin % 8) * (total / 8) + lin / 8;
            bx = tile % gx, by = (tile / gx) % gy, bz = tile / (gx * gy);
        }
    }
    const int m0 = bx * kTM, n0 = by * kTN;
    const int nsl_total = K8 / 4;
    const int s_beg = bz * slabs_per_split;
    const int s_end = s_beg + slabs_per_split < nsl_total ? s_beg + slabs_per_split : nsl_total;
    const int nst = s_end - s_beg;            // stages of this split: one 32-deep K-slab each
    const int wm = wave >> 1, wn = wave & 1;  // this wave's 64 x 64 block
    f32x16 acc[2][2], lo[2][2];               // the unscaled and the 2^11-scaled sums
#pragma unroll
    for (int a = 0; a < 2; a++)
#pragma unroll
        for (int b = 0; b < 2; b++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[a][b][r] = 0.f, lo[a][b][r] = 0.f;
    // ---- LDS-DMA addressing: per operand tile 1024 chunk slots = 16 wave-instructions, 4 per wave; instruction u of wave w fills slots (u * 4 + w) * 64 + lane.
    //      Slot sl = row r = sl / 8, position q = sl % 8 holds chunk c = (q - rot(r)) mod 8, c = k-group * 2 + part (the LDS order of BOTH operands)
    const uint4 *ga[4], *gb[4];
#pragma unroll
    for (int u = 0; u < 4; u++) {
        const int sl = (u * 4 + wave) * 64 + lane, r = sl >> 3, q = sl & 7;
        const int c = (q - ((r >> 1) & 7)) & 7;
        ga[u] = A + (((i64)s_beg * M + m0 + r) * 8 + c);                       // activations [slab][row][k-group 4][part 2]
        gb[u] = W + (((i64)s_beg * N + n0 + r) * 8 + (c & 1) * 4 + (c >> 1));  // weight      [slab][row][part 2][k-group 4]
    }
    // stage st = slab - s_beg: piece v = 0..7 -> (operand v & 1, instruction v >> 1)
    auto issue_piece = [&](int slot, int st, int v) __attribute__((always_inline)) {
        unsigned char *base = smem + slot * kHBuf + wave * 1024 + (v >> 1) * 4096;
        if (v & 1)
            __builtin_amdgcn_global_load_lds((gptr_t *)(gb[v >> 1] + (i64)st * N * 8), (lptr_t *)(base + kHTile), 16, 0, 0);
        else
            __builtin_amdgcn_global_load_lds((gptr_t *)(ga[v >> 1] + (i64)st * M * 8), (lptr_t *)base, 16, 0, 0);
    };
    // fragment addresses: row r, chunk c = (2 ks + h) * 2 + p at slot (c + rot) mod 8, rot = (r >> 1) & 7 = (i >> 1) & 7 (block row bases are multiples of 32)
    const int rot = (i >> 1) & 7;
    int fo[2][2];
#pragma unroll
    for (int ks = 0; ks < 2; ks++)
#pragma unroll
        for (int p = 0; p < 2; p++) fo[ks][p] = ((((2 * ks + h) * 2 + p) + rot) & 7) * 16;
    const int arow = (wm * 64 + i) * kHRow, brow = kHTile + (wn * 64 + i) * kHRow;
    // Fragments are double-buffered per K-STEP (16 k), not per stage: f0 always holds a stage's first k-step, f1 its second -- 64 registers instead of 128.  (With
    // whole-stage sets the kernel needed 256 + registers; the compiler parked accumulators in AGPRs and moved them in and out around every stage: 320 v_accvgpr
    // moves per two stages, and on gfx950 vector instructions take their issue slots from the matrix pipe: 8.5 VALU instructions per MFMA by the SQ counters.)
    struct Frags {
        f16x8 a[2][2], b[2][2];  // [tile][part]
    };
    // 4 of a k-step's 8 fragment reads: g = 0: the A parts of both row tiles, 1: the B parts of both column tiles
    auto read_half = [&](const unsigned char *buf, Frags &f, int ks, int g) __attribute__((always_inline)) {
#pragma unroll
        for (int x = 0; x < 2; x++)
#pragma unroll
            for (int p = 0; p < 2; p++) {
                if (g)
                    f.b[x][p] = *reinterpret_cast<const f16x8 *>(buf + brow + x * 32 * kHRow + fo[ks][p]);
                else
                    f.a[x][p] = *reinterpret_cast<const f16x8 *>(buf + arow + x * 32 * kHRow + fo[ks][p]);
            }
    };
    constexpr int pq[3][2] = {{1, 0}, {0, 1}, {0, 0}};  // (a part, b part): the two cross terms into `lo`, then hi x hi into `acc` (as k_gemm_s16<.., H16>)
    // k-step (st, ks): multiplies it out of `cur`; reads the NEXT k-step's fragments -- (st, 1) out of this stage's slot, or (st + 1, 0) out of the next one's -- into
    // `nxt`; a stage's second k-step starts with the stage's one barrier (stage st + 1 has landed for everybody; everybody has finished reading stage st: its slot
    // takes the DMA of stage st + 3, issued between this k-step's MFMAs)
    auto kstep = [&](int st, int ks, int slot_cur, int slot_next, const Frags &cur, Frags &nxt) __attribute__((always_inline)) {
        if (ks == 1) {
            if (st + 2 < nst)
                asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            else
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
        }
        const bool rd = ks == 0 || st + 1 < nst, dma = ks == 1 && st + 3 < nst;
        const unsigned char *buf = smem + (ks == 0 ? slot_cur : slot_next) * kHBuf;
#pragma unroll
        for (int c = 0; c < 3; c++) {  // three steps of four MFMAs (one partial product on the four accumulator pairs); behind them fragment reads / DMA pieces
#pragma unroll
            for (int ms = 0; ms < 2; ms++)
#pragma unroll
                for (int ns = 0; ns < 2; ns++) {
                    if (c < 2)
                        lo[ms][ns] = __builtin_amdgcn_mfma_f32_32x32x16_f16(cur.a[ms][pq[c][0]], cur.b[ns][pq[c][1]], lo[ms][ns], 0, 0, 0);
                    else
                        acc[ms][ns] = __builtin_amdgcn_mfma_f32_32x32x16_f16(cur.a[ms][pq[c][0]], cur.b[ns][pq[c][1]], acc[ms][ns], 0, 0, 0);
                }
            __builtin_amdgcn_sched_barrier(0);
            if (c < 2 && rd) read_half(buf, nxt, 1 - ks, c);
            if (dma) {
#pragma unroll
                for (int v = (c * 8) / 3; v < ((c + 1) * 8) / 3; v++) issue_piece(slot_cur, st + 3, v);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    if (nst > 0) {
        Frags f0, f1;
#pragma unroll
        for (int k = 0; k < 3; k++)
            if (k < nst)
#pragma unroll
                for (int v = 0; v < 8; v++) issue_piece(k, k, v);
        if (nst >= 3)
            asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
        else if (nst == 2)
            asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();  // stage 0 is in slot 0 for everybody
        read_half(smem, f0, 0, 0);
        read_half(smem, f0, 0, 1);
        int slot = 0;  // slot of stage st
        for (int st = 0; st < nst; st++) {
            const int s1 = slot == 2 ? 0 : slot + 1;
            kstep(st, 0, slot, s1, f0, f1);
            kstep(st, 1, slot, s1, f1, f0);
            slot = s1;
        }
    }
    float *Cz = C + (i64)bz * M * N;
#pragma unroll
    for (int ms = 0; ms < 2; ms++)
#pragma unroll
        for (int ns = 0; ns < 2; ns++) {
            const int n = n0 + wn * 64 + ns * 32 + i;
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const i64 m = m0 + wm * 64 + ms * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                Cz[m * N + n] = __builtin_fmaf(lo[ms][ns][r], 1.0f / 2048.0f, acc[ms][ns][r]);
            }
        }
    if (span && threadIdx.x == 0) atomicMax(&span[1], (unsigned long long)wall_clock64());
}

}  // namespace

// operand planes of this handle: allocated by srlx_qnet_enable_fc1_planes
int srlx_fc1_planes_alloc(srlx_qnet *h) {
    if (h->wf_planes) return SRLX_OK;
    const size_t N1 = 2 * (size_t)h->hidden;
    SRLX_REQUIRE(h->flat % 32 == 0 && N1 % kTN == 0, "fc1_planes: the layer must be a multiple of 32 wide (K) and of %d (units)", kTN);
    SRLX_HIP(hipMalloc((void **)&h->wf_planes, N1 * h->flat * 6));
    SRLX_HIP(hipMalloc((void **)&h->a3_planes, (size_t)((h->max_batch + 127) / 128 * 128) * h->flat * 6));
    return SRLX_OK;
}

// float32 weight [2 hidden][flat] (src, or the handle's bound weight) -> the handle's planes; `copy_dst` (optional) also receives the float32 values
size_t srlx_fc1_planes_weight_bytes(const srlx_qnet *h) { return 2 * (size_t)h->hidden * h->flat * 6; }

int srlx_fc1_planes_split_weight(srlx_qnet *h, const float *src, float *copy_dst, hipStream_t st, void *planes_dst) {
    const i64 rows = 2 * (i64)h->hidden, n8 = rows * h->flat / 8;
    hipLaunchKernelGGL(k_split_planes<true>, dim3((unsigned)((n8 + 255) / 256)), dim3(256), 0, st, src, rows, h->flat / 8,
                       (f16x8 *)(planes_dst ? planes_dst : h->wf_planes), copy_dst);
    SRLX_HIP(hipGetLastError());
    return SRLX_OK;
}

// float32 activations act3 [rows][flat] -> a3_planes (the path for geometries whose convolution kernel does not write planes itself)
int srlx_fc1_planes_split_act(srlx_qnet *h, int64_t rows, hipStream_t st) {
    const i64 n8 = rows * h->flat / 8;
    hipLaunchKernelGGL(k_split_planes<false>, dim3((unsigned)((n8 + 255) / 256)), dim3(256), 0, st, (const float *)h->act3, (i64)rows, h->flat / 8, (f16x8 *)h->a3_planes,
                       (float *)nullptr);
    SRLX_HIP(hipGetLastError());
    return SRLX_OK;
}

// chip-filling launches (>= 512 rows in multiples of the tile); with `planes_small` (srlx_qnet_set_planes_small: a learner's handle, half-CU kernel) any launch, rows
// padded to the tile
bool srlx_fc1_planes_applicable(const srlx_qnet *h, int64_t rows) {
    if (!h->wf_planes) return false;
    if (h->planes_small && h->fc1_neighbour > 0) return true;
    return rows % kTM == 0 && rows >= 512;
}

// partial[split][rows][2 hidden] = a3_planes x wf_planes^T over the split's K range; `splits` / `kps` (32-deep K-slabs per split) as k_gemm_s16's launch
int srlx_fc1_planes_gemm(srlx_qnet *h, int64_t rows, int splits, int kps, hipStream_t st) {
    static bool attr_h = false;
    if (!attr_h) {
        SRLX_HIP(hipFuncSetAttribute((const void *)k_fc1_planes_h, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kHLds));
        attr_h = true;
    }
    const int N1 = 2 * h->hidden;
    rows = (rows + kTM - 1) / kTM * kTM;  // (a small launch's pad rows multiply whatever the plane buffer holds: their partial sums are never read)
    const dim3 grid((unsigned)(rows / kTM), (unsigned)(N1 / kTN), (unsigned)splits);
    hipLaunchKernelGGL(k_fc1_planes_h, grid, dim3(256), kHLds, st, (const uint4 *)h->a3_planes, (const uint4 *)h->wf_planes, h->partial, (int)rows, N1, h->flat / 8, kps,
                       (unsigned long long *)h->fc1_span);
    h->fc1_span = nullptr;  // one launch only
    SRLX_HIP(hipGetLastError());
    return SRLX_OK;
}
