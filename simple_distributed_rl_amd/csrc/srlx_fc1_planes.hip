// srlx_fc1_planes.hip -- the first dense layer of the ACTORS' pass as a conversion-free GEMM on pre-split bf16 operand planes.
//
// FC1 (srl/rl/torch_/blocks/dueling_network.py:8-59 behind rainbow/model_torch.py:55-67: [rows][7744] x [2*hidden][7744]^T) evaluates
// float32 x float32 as six exact bf16 partial products (see k_gemm_s16 in srlx_qnet.hip, which splits both operands while staging:
// ~130 VALU instructions per 24 MFMAs, every K-slab, in every workgroup that touches the tile).  For the chip-filling launch of the
// actors (1024 rows per lock-step) both operands can arrive ALREADY split:
//   * the activations: the convolution kernel's conv3 epilogue writes act3 as planes (srlx_qnet_fused.hip, PLANES = true), and
//   * the weight: the actors act on a private copy of the online network that is refreshed once per lock-step
//     (device/rainbow.py:refresh_actor_copy) -- that copy now also emits the planes (k_split_planes, one pass over the 32 MB weight).
// Plane layout: activations [K/32 slabs][rows][4 k-groups][3 parts][8 bf16], weight [K/32 slabs][rows][3 parts][4 k-groups][8 bf16] (a part's 32 k of a row are
// 64 contiguous bytes: what a lane group of the weight-gradient kernel's Adam epilogue holds, srlx_qnet_bwd.hip: k_fc1_wgrad<.., PLANES>; the LDS image of a
// tile is the same for both, only the SOURCE chunk index of the DMA differs) -- the 16-byte chunk (k-group g, part p) IS the fragment a lane
// feeds v_mfma_f32_32x32x16_bf16 (row i, k = 8 g .. 8 g + 7); a row's 32-deep K-slab is 192 contiguous bytes (12 chunks), and the slab-major
// order makes a whole 128-row operand tile of a K-slab ONE contiguous 24 KB run: every 128-byte line the LDS-DMA touches is used in full (with
// row-major planes a row's slab straddles 2-3 lines of which 60 % is wanted, and the L2 -> CU path carries the rest for nothing).
//
// The GEMM: 128 x 128 output tile per workgroup (8 waves, 64 x 32 each), K split over blockIdx.z like k_gemm_s16 (partials reduced by
// k_head), operand tiles fetched by LDS-DMA (global_load_lds_dwordx4: no VGPR round trip, no ds_write) into a three-slot LDS ring, one barrier per
// K-slab.  LDS image of a tile: row r = 12 chunk slots, chunk j stored at slot (j + ((r >> 2) & 3)) mod 12 -- the rotation makes the 16 rows
// a ds_read_b128 lane group touches ({0-3, 12-15, 20-27} / {4-11, 16-19, 28-31}: MI355X_MICROARCH.md, LDS) hit 16 different 16-byte bank
// groups; the rotation is applied on the SOURCE address of the DMA (its LDS side is lane-linear).
// Same K split, same k order and the same order of the six partial products as k_gemm_s16: the two paths are bit-identical.
#include "srlx_qnet_int.h"

namespace {

using i64 = int64_t;
using f32x16 = __attribute__((ext_vector_type(16))) float;
using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
typedef __attribute__((address_space(1))) const void gptr_t;
typedef __attribute__((address_space(3))) void lptr_t;

constexpr int kTM = 128, kTN = 128;         // output tile
constexpr int kRowBytes = 192;              // one row's K-slab of 32: 4 k-groups x 3 parts x 16 B
constexpr int kTileBytes = kTM * kRowBytes;  // 24 KB per operand tile
constexpr int kBufBytes = 2 * kTileBytes;    // A + B
constexpr int kStages = 3;                   // LDS ring: two K-slabs travel while one is multiplied
constexpr size_t kLds = kStages * kBufBytes; // 144 KB

// float32 [rows][K] -> planes [K/32][rows][4][3][8] bf16 (+ an optional float32 copy: the actors' private weight and its planes in one pass);
// one thread per 8 consecutive k of a row
template <bool WEIGHT>  // WEIGHT: the weight's chunk order [part][k-group], else the activations' [k-group][part]
__global__ void __launch_bounds__(256) k_split_planes(const float *__restrict__ src, i64 rows, int K8, bf16x8 *__restrict__ planes, float *__restrict__ copy) {
    const i64 q = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= rows * K8) return;
    const i64 row = q / K8;
    const int k8 = (int)(q % K8);
    const float4 x0 = reinterpret_cast<const float4 *>(src)[2 * q], x1 = reinterpret_cast<const float4 *>(src)[2 * q + 1];
    if (copy) {
        reinterpret_cast<float4 *>(copy)[2 * q] = x0;
        reinterpret_cast<float4 *>(copy)[2 * q + 1] = x1;
    }
    float r[8] = {x0.x, x0.y, x0.z, x0.w, x1.x, x1.y, x1.z, x1.w};
    bf16x8 *dst = planes + ((i64)(k8 >> 2) * rows + row) * 12 + (WEIGHT ? (k8 & 3) : (k8 & 3) * 3);
#pragma unroll
    for (int p = 0; p < 3; p++) {
        bf16x8 part;
#pragma unroll
        for (int j = 0; j < 8; j++) {
            const __bf16 b = (__bf16)r[j];
            part[j] = b;
            r[j] -= (float)b;
        }
        dst[WEIGHT ? 4 * p : p] = part;
    }
}

__global__ void __launch_bounds__(512) k_fc1_planes(const uint4 *__restrict__ A, const uint4 *__restrict__ W, float *__restrict__ C, int M, int N, int K8,
                                                    int slabs_per_split, unsigned long long *__restrict__ span) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    if (span && threadIdx.x == 0) atomicMin(&span[0], (unsigned long long)wall_clock64());  // srlx_qnet_set_fc1_span: first workgroup in
    const int t = threadIdx.x, wave = t >> 6, lane = t & 63, i = lane & 31, h = lane >> 5;
    // XCD-aware tile order (as k_gemm): XCD x = linear id % 8 gets the x-th contiguous eighth of the (split, N tile, M tile) space -- one K range x
    // a share of the N tiles x all M tiles -- so that a weight tile is fetched by one XCD's L2 and an activation K-slice by as few as possible
    unsigned bx = blockIdx.x, by = blockIdx.y, bz = blockIdx.z;
    {
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
    const int wm = wave >> 2, wn = wave & 3;  // this wave's 64 x 32 block of the tile
    f32x16 acc[2];
#pragma unroll
    for (int a = 0; a < 2; a++)
#pragma unroll
        for (int r = 0; r < 16; r++) acc[a][r] = 0.f;

    // ---- LDS-DMA addressing: per operand tile 1536 chunks = 3 wave-instructions per wave; instruction u of wave w fills LDS chunk slots
    //      (u * 8 + w) * 64 + lane (lane-linear), each lane fetching the chunk that belongs there
    const uint4 *ga[3], *gb[3];
#pragma unroll
    for (int u = 0; u < 3; u++) {
        const int s = (u * 8 + wave) * 64 + lane, r = s / 12, q = s % 12;
        int j = q - ((r >> 2) & 3);
        j += j < 0 ? 12 : 0;
        ga[u] = A + (((i64)s_beg * M + m0 + r) * 12 + j);
        gb[u] = W + (((i64)s_beg * N + n0 + r) * 12 + (j % 3) * 4 + j / 3);  // chunk (k-group j / 3, part j % 3) in the weight's [part][k-group] order
    }
    // one piece (1 KB per wave) of the next slab: piece v = 0..5 -> (operand v & 1, instruction v >> 1)
    auto issue_piece = [&](int buf, int v) __attribute__((always_inline)) {
        unsigned char *base = smem + buf * kBufBytes + wave * 1024 + (v >> 1) * 8192;
        if (v & 1) {
            __builtin_amdgcn_global_load_lds((gptr_t *)gb[v >> 1], (lptr_t *)(base + kTileBytes), 16, 0, 0);
            gb[v >> 1] += (i64)N * 12;  // the next K-slab of the same rows
        } else {
            __builtin_amdgcn_global_load_lds((gptr_t *)ga[v >> 1], (lptr_t *)base, 16, 0, 0);
            ga[v >> 1] += (i64)M * 12;
        }
    };
    auto issue = [&](int buf) __attribute__((always_inline)) {
#pragma unroll
        for (int v = 0; v < 6; v++) issue_piece(buf, v);
    };
    // ---- fragment addresses: row r, chunk j = (2 ks + h) * 3 + p at slot (j + rot) mod 12, rot = (r >> 2) & 3 = (i >> 2) & 3 (tile row bases are multiples of 32)
    const int rot = (i >> 2) & 3;
    int fo[2][3];
#pragma unroll
    for (int ks = 0; ks < 2; ks++)
#pragma unroll
        for (int p = 0; p < 3; p++) {
            int j = (2 * ks + h) * 3 + p + rot;
            j -= j >= 12 ? 12 : 0;
            fo[ks][p] = j * 16;
        }
    const int arow = (wm * 64 + i) * kRowBytes, brow = kTileBytes + (wn * 32 + i) * kRowBytes;

    // Software pipeline over a three-slot LDS ring.  In iteration s a wave multiplies slab s out of REGISTERS (fragments read during iteration s - 1),
    // reads the fragments of slab s + 1 out of LDS, and issues the LDS-DMA of slab s + 3 into the slot slab s has just left -- the 18 reads and the 6 DMA
    // pieces threaded between the 24 MFMAs, so that nobody's matrix pipe waits for an LDS round trip or a DMA issue (ablation, 1024 rows: the DMA alone
    // 39 us, reads-then-MFMAs without any DMA 76 us against a 39 us MFMA floor: the exposed read phase of two barrier-synchronised waves per SIMD was the
    // loss).  An LDS-DMA is ordered for a reader only by the issuing wave's counted vmcnt followed by a barrier the reader has passed; the barrier is the
    // raw instruction (__syncthreads() would drain vmcnt(0) and with it the slabs that should keep travelling).
    struct Frags {
        bf16x8 b[2][3], a[2][2][3];
    };
    // 3 of a slab's 18 fragment reads: group g = 0..5 -> k-step g / 3; g % 3 = 0: the B parts, 1 / 2: the A parts of row tile 0 / 1
    auto read_group = [&](const unsigned char *buf, Frags &f, int g) __attribute__((always_inline)) {
        const int ks = g / 3, k = g % 3;
#pragma unroll
        for (int p = 0; p < 3; p++) {
            if (k == 0)
                f.b[ks][p] = *reinterpret_cast<const bf16x8 *>(buf + brow + fo[ks][p]);
            else
                f.a[ks][k - 1][p] = *reinterpret_cast<const bf16x8 *>(buf + arow + (k - 1) * 32 * kRowBytes + fo[ks][p]);
        }
    };
    constexpr int pq[6][2] = {{2, 0}, {0, 2}, {1, 1}, {1, 0}, {0, 1}, {0, 0}};  // smallest partial products first (as k_gemm_s16)
    // iteration s: `slot_next` holds slab s + 1, `slot_free` held slab s
    auto body = [&](int s, int slot_next, int slot_free, const Frags &cur, Frags &nxt) __attribute__((always_inline)) {
        // this wave's pieces of slab s + 1 have landed (of the DMA groups in flight only slab s + 2's may remain); its fragments of slab s are in registers
        if (s + 2 < s_end)
            asm volatile("s_waitcnt vmcnt(6) lgkmcnt(0)" ::: "memory");
        else
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();  // ... everybody's pieces; and every wave has finished READING slab s: its slot is free for slab s + 3
        const bool rd = s + 1 < s_end, dma = s + 3 < s_end;
        const unsigned char *buf = smem + slot_next * kBufBytes;
#pragma unroll
        for (int t = 0; t < 12; t++) {  // 12 steps of two MFMAs; behind them alternately three fragment reads / one DMA piece
            const int ks = t / 6, c = t % 6;
#pragma unroll
            for (int ms = 0; ms < 2; ms++) acc[ms] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(cur.a[ks][ms][pq[c][0]], cur.b[ks][pq[c][1]], acc[ms], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            if (t & 1) {
                if (dma) issue_piece(slot_free, t >> 1);
            } else if (rd) {
                read_group(buf, nxt, t >> 1);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    const int nsl = s_end - s_beg;
    if (nsl > 0) {
        Frags f0, f1;
#pragma unroll
        for (int k = 0; k < 3; k++)
            if (k < nsl) issue(k);  // slabs 0, 1, 2 -> slots 0, 1, 2
        if (nsl >= 3)
            asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
        else if (nsl == 2)
            asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        else
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();  // slab 0 is in slot 0 for everybody
#pragma unroll
        for (int g = 0; g < 6; g++) read_group(smem, f0, g);
        // slab s sits in slot s % 3; two iterations per trip so that the two fragment register sets swap roles statically
        int slot = 0;  // slot of slab s
        for (int s = s_beg; s < s_end; s += 2) {
            const int s1 = slot == 2 ? 0 : slot + 1, s2 = s1 == 2 ? 0 : s1 + 1;
            body(s, s1, slot, f0, f1);
            if (s + 1 < s_end) body(s + 1, s2, s1, f1, f0);
            slot = s2;
        }
    }
    float *Cz = C + (i64)bz * M * N;
    const int n = n0 + wn * 32 + i;
#pragma unroll
    for (int ms = 0; ms < 2; ms++)
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const i64 m = m0 + wm * 64 + ms * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
            Cz[m * N + n] = acc[ms][r];
        }
    if (span && threadIdx.x == 0) atomicMax(&span[1], (unsigned long long)wall_clock64());  // (last workgroup out; its stores are in flight, as at a kernel's end)
}

// ---- the same GEMM as a GOOD NEIGHBOUR (round 4) ---------------------------------------------------------------------------------------------------------
// k_fc1_planes above owns its CU: 512 threads whose registers fill the four SIMDs, 144 KB of LDS, one workgroup per CU for the whole launch (83 us).  Alone that
// is the fastest form; beside a learner it is the slowest -- the update is a chain of ~22 short dependent kernels, each of which then waits for a compute unit
// to come free (same-box A/B of the whole lock-step: 0.535 ms with it, 0.486 ms with the staging-split k_gemm_s16 that is 1.5x slower alone but leaves room).
// This variant keeps the conversion-free operand planes and the LDS-DMA ring and gives up the CU: 256 threads (one wave per SIMD, a 64 x 64 block of the
// 128 x 128 tile each: four accumulators, 12 fragment reads per 24 MFMAs instead of 18), stages of HALF a K-slab (16 k: 12 KB per operand tile), a
// three-slot ring = 72 KB -- two workgroups share a CU, or one workgroup and whatever the learner wants to run; with 8 K splits the launch is 512 workgroups
// that retire in a steady stream.  Same K order inside a split and the same order of the six partial products as k_fc1_planes / k_gemm_s16; the split-K
// partial sums are added by k_head in split order, so the result depends on the NUMBER of splits only through float32 association (bit-identical at equal splits).
// LDS image of a half-slab tile: row r = 6 chunk slots of 16 B (k-group hh = 0 / 1 of the half-slab x 3 parts: chunk c = hh * 3 + p), chunk c at slot
// (c + ((r >> 4) & 1)) mod 6: the 16 rows of a ds_read_b128 lane group ({0-3, 12-15, 20-27} / {4-11, 16-19, 28-31}) then hit 16 different 16-byte bank groups
// (6 r mod 16 takes the eight even values over r mod 8; the rotation moves rows 16-31 to the odd ones).
constexpr int kHRow = 96;                      // bytes per row of a half-slab tile
constexpr int kHTile = kTM * kHRow;            // 12 KB
constexpr int kHBuf = 2 * kHTile;              // A + B
constexpr int kHStages = 3;
constexpr size_t kHLds = kHStages * kHBuf;     // 72 KB

__global__ void __launch_bounds__(256, 2) k_fc1_planes_h(const uint4 *__restrict__ A, const uint4 *__restrict__ W, float *__restrict__ C, int M, int N, int K8,
                                                      int slabs_per_split, unsigned long long *__restrict__ span) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    if (span && threadIdx.x == 0) atomicMin(&span[0], (unsigned long long)wall_clock64());  // srlx_qnet_set_fc1_span: first workgroup in
    const int t = threadIdx.x, wave = t >> 6, lane = t & 63, i = lane & 31, h = lane >> 5;
    unsigned bx = blockIdx.x, by = blockIdx.y, bz = blockIdx.z;
    {   // XCD-aware tile order (as k_fc1_planes)
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
    const int nst = 2 * (s_end - s_beg);  // half-slab stages of this split
    const int wm = wave >> 1, wn = wave & 1;  // this wave's 64 x 64 block
    f32x16 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; a++)
#pragma unroll
        for (int b = 0; b < 2; b++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[a][b][r] = 0.f;
    // ---- LDS-DMA addressing: per operand tile 768 chunk slots = 12 wave-instructions, 3 per wave; instruction u of wave w fills slots (u * 4 + w) * 64 + lane
    const uint4 *ga[3], *gb[3];
#pragma unroll
    for (int u = 0; u < 3; u++) {
        const int sl = (u * 4 + wave) * 64 + lane, r = sl / 6, q = sl % 6;
        int c = q - ((r >> 4) & 1);
        c += c < 0 ? 6 : 0;
        const int hh = c / 3, p = c % 3;
        // activations [slab][row][k-group 4][part 3]: half-slab ks starts at chunk 6 ks; weight [slab][row][part 3][k-group 4]: chunk 4 p + 2 ks + hh
        ga[u] = A + (((i64)s_beg * M + m0 + r) * 12 + c);
        gb[u] = W + (((i64)s_beg * N + n0 + r) * 12 + 4 * p + hh);
    }
    // stage st = 2 (slab - s_beg) + ks: piece v = 0..5 -> (operand v & 1, instruction v >> 1)
    auto issue_piece = [&](int slot, int st, int v) __attribute__((always_inline)) {
        unsigned char *base = smem + slot * kHBuf + wave * 1024 + (v >> 1) * 4096;
        const int ks = st & 1;
        const i64 slab = st >> 1;
        if (v & 1)
            __builtin_amdgcn_global_load_lds((gptr_t *)(gb[v >> 1] + slab * N * 12 + 2 * ks), (lptr_t *)(base + kHTile), 16, 0, 0);
        else
            __builtin_amdgcn_global_load_lds((gptr_t *)(ga[v >> 1] + slab * M * 12 + 6 * ks), (lptr_t *)base, 16, 0, 0);
    };
    // fragment addresses: row r, chunk c = h * 3 + p at slot (c + rot) mod 6, rot = (r >> 4) & 1 = (i >> 4) & 1 (block row bases are multiples of 32)
    const int rot = (i >> 4) & 1;
    int fo[3];
#pragma unroll
    for (int p = 0; p < 3; p++) {
        int c = h * 3 + p + rot;
        c -= c >= 6 ? 6 : 0;
        fo[p] = c * 16;
    }
    const int arow = (wm * 64 + i) * kHRow, brow = kHTile + (wn * 64 + i) * kHRow;
    struct Frags {
        bf16x8 a[2][3], b[2][3];
    };
    // 3 of a stage's 12 fragment reads: group g = 0..3 -> g < 2: the A parts of row tile g, else the B parts of column tile g - 2
    auto read_group = [&](const unsigned char *buf, Frags &f, int g) __attribute__((always_inline)) {
#pragma unroll
        for (int p = 0; p < 3; p++) {
            if (g < 2)
                f.a[g][p] = *reinterpret_cast<const bf16x8 *>(buf + arow + g * 32 * kHRow + fo[p]);
            else
                f.b[g - 2][p] = *reinterpret_cast<const bf16x8 *>(buf + brow + (g - 2) * 32 * kHRow + fo[p]);
        }
    };
    constexpr int pq[6][2] = {{2, 0}, {0, 2}, {1, 1}, {1, 0}, {0, 1}, {0, 0}};  // smallest partial products first (as k_gemm_s16)
    auto body = [&](int st, int slot_next, int slot_free, const Frags &cur, Frags &nxt) __attribute__((always_inline)) {
        if (st + 2 < nst)
            asm volatile("s_waitcnt vmcnt(6) lgkmcnt(0)" ::: "memory");
        else
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();  // everybody's pieces of stage st + 1 have landed; every wave has finished READING stage st: its slot is free for stage st + 3
        const bool rd = st + 1 < nst, dma = st + 3 < nst;
        const unsigned char *buf = smem + slot_next * kHBuf;
#pragma unroll
        for (int x = 0; x < 6; x++) {  // six steps of four MFMAs (one partial product on the four accumulators); behind them two fragment-read groups or three DMA pieces
#pragma unroll
            for (int ms = 0; ms < 2; ms++)
#pragma unroll
                for (int ns = 0; ns < 2; ns++) acc[ms][ns] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(cur.a[ms][pq[x][0]], cur.b[ns][pq[x][1]], acc[ms][ns], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            if (x < 4) {
                if (rd) read_group(buf, nxt, x);
            } else if (dma) {
#pragma unroll
                for (int v = 0; v < 3; v++) issue_piece(slot_free, st + 3, (x - 4) * 3 + v);
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
                for (int v = 0; v < 6; v++) issue_piece(k, k, v);
        if (nst >= 3)
            asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
        else if (nst == 2)
            asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        else
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();  // stage 0 is in slot 0 for everybody
#pragma unroll
        for (int g = 0; g < 4; g++) read_group(smem, f0, g);
        int slot = 0;  // slot of stage st
        for (int st = 0; st < nst; st += 2) {
            const int s1 = slot == 2 ? 0 : slot + 1, s2 = s1 == 2 ? 0 : s1 + 1;
            body(st, s1, slot, f0, f1);
            if (st + 1 < nst) body(st + 1, s2, s1, f1, f0);
            slot = s2;
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
                Cz[m * N + n] = acc[ms][ns][r];
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
                       (bf16x8 *)(planes_dst ? planes_dst : h->wf_planes), copy_dst);
    SRLX_HIP(hipGetLastError());
    return SRLX_OK;
}

// float32 activations act3 [rows][flat] -> a3_planes (the path for geometries whose convolution kernel does not write planes itself)
int srlx_fc1_planes_split_act(srlx_qnet *h, int64_t rows, hipStream_t st) {
    const i64 n8 = rows * h->flat / 8;
    hipLaunchKernelGGL(k_split_planes<false>, dim3((unsigned)((n8 + 255) / 256)), dim3(256), 0, st, (const float *)h->act3, (i64)rows, h->flat / 8, (bf16x8 *)h->a3_planes,
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
    static bool attr_set = false;
    if (!attr_set) {
        SRLX_HIP(hipFuncSetAttribute((const void *)k_fc1_planes, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLds));
        attr_set = true;
    }
    const int N1 = 2 * h->hidden;
    rows = (rows + kTM - 1) / kTM * kTM;  // (a small launch's pad rows multiply whatever the plane buffer holds: their partial sums are never read)
    const dim3 grid((unsigned)(rows / kTM), (unsigned)(N1 / kTN), (unsigned)splits);
    if (h->fc1_neighbour) {  // half-CU workgroups (srlx_qnet_set_fc1_neighbour): the handle's passes run beside a learner
        static bool attr_h = false;
        if (!attr_h) {
            SRLX_HIP(hipFuncSetAttribute((const void *)k_fc1_planes_h, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kHLds));
            attr_h = true;
        }
        hipLaunchKernelGGL(k_fc1_planes_h, grid, dim3(256), kHLds, st, (const uint4 *)h->a3_planes, (const uint4 *)h->wf_planes, h->partial, (int)rows, N1, h->flat / 8, kps,
                           (unsigned long long *)h->fc1_span);
        h->fc1_span = nullptr;  // one launch only
        SRLX_HIP(hipGetLastError());
        return SRLX_OK;
    }
    hipLaunchKernelGGL(k_fc1_planes, grid, dim3(512), kLds, st, (const uint4 *)h->a3_planes, (const uint4 *)h->wf_planes, h->partial, (int)rows, N1, h->flat / 8, kps,
                       (unsigned long long *)h->fc1_span);
    h->fc1_span = nullptr;
    SRLX_HIP(hipGetLastError());
    return SRLX_OK;
}
