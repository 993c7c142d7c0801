// Probe: what does a scattered 64-byte fetch per draw cost by the SHAPE of the loads (the bottom groups of the bulk PER walk)?
//   a  every lane fetches its own half-line as four 16-byte loads (what k_descend_bulk did through round 4): 4 instructions x 64 lines
//   b  four lanes fetch one half-line together, one 16-byte load each (4 instructions x 16 lines of 64 contiguous bytes), pieces summed through DPP
//   c  every lane one 16-byte load of its own line (lower bound: one instruction x 64 lines)
//   d  every lane its own half-line as ONE 16-byte load + three 16-byte loads issued after it returned (dependent, L1-hit) -- the price of a hit
// build: hipcc --offload-arch=gfx950 -O3 tools/gather_probe.hip -o tools/_gather_probe ; run: tools/_gather_probe [log2_lines=17] [log2_draws=20]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef unsigned long long u64;
__device__ __forceinline__ u64 mix(u64 x) {
    x ^= x >> 33; x *= 0xff51afd7ed558ccdull; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ull; x ^= x >> 33;
    return x;
}
template <int MODE, int ILP>
__global__ void __launch_bounds__(1024) k_gather(const double *T, u64 line_mask, long long M, double *out) {
    const long long stride = (long long)gridDim.x * blockDim.x;
    double acc = 0.0;
    const int lane = threadIdx.x & 63, ql = lane & 3;
    for (long long base = (long long)blockIdx.x * blockDim.x + threadIdx.x; base < M; base += stride * ILP) {
        u64 line[ILP];
#pragma unroll
        for (int d = 0; d < ILP; d++) line[d] = mix((u64)(base + d * stride) * 0x9E3779B97F4A7C15ull + 12345) & line_mask;
        if (MODE == 0) {
            double2 v[ILP][4];
#pragma unroll
            for (int d = 0; d < ILP; d++)
#pragma unroll
                for (int k = 0; k < 4; k++) v[d][k] = reinterpret_cast<const double2 *>(T + line[d] * 16)[k];
#pragma unroll
            for (int d = 0; d < ILP; d++)
#pragma unroll
                for (int k = 0; k < 4; k++) acc += v[d][k].x + v[d][k].y;
        } else if (MODE == 1) {
            double2 v[ILP][4];
#pragma unroll
            for (int d = 0; d < ILP; d++)
#pragma unroll
                for (int o = 0; o < 4; o++) {
                    const u64 lo = __shfl(line[d], (lane & ~3) | o);  // the line of quad lane o
                    v[d][o] = reinterpret_cast<const double2 *>(T + lo * 16)[ql];
                }
            // lane ql keeps the sum of line (quad lane ql): piece p of it sits in lane p's v[d][ql]
#pragma unroll
            for (int d = 0; d < ILP; d++) {
                double mine = 0.0;
#pragma unroll
                for (int o = 0; o < 4; o++) {
                    double s = v[d][o].x + v[d][o].y;
                    s += __shfl_xor(s, 1);
                    s += __shfl_xor(s, 2);
                    mine = ql == o ? s : mine;
                }
                acc += mine;
            }
        } else if (MODE == 2) {
            double2 v[ILP];
#pragma unroll
            for (int d = 0; d < ILP; d++) v[d] = reinterpret_cast<const double2 *>(T + line[d] * 16)[0];
#pragma unroll
            for (int d = 0; d < ILP; d++) acc += v[d].x + v[d].y;
        } else {
            double2 v[ILP];
#pragma unroll
            for (int d = 0; d < ILP; d++) v[d] = reinterpret_cast<const double2 *>(T + line[d] * 16)[0];
            double2 w[ILP][3];
#pragma unroll
            for (int d = 0; d < ILP; d++) {
                const int sel = v[d].x > 2.0 ? 1 : 0;  // (never true: keeps the second round dependent on the first)
#pragma unroll
                for (int k = 0; k < 3; k++) w[d][k] = reinterpret_cast<const double2 *>(T + line[d] * 16)[1 + k + sel];
            }
#pragma unroll
            for (int d = 0; d < ILP; d++) {
                acc += v[d].x + v[d].y;
#pragma unroll
                for (int k = 0; k < 3; k++) acc += w[d][k].x + w[d][k].y;
            }
        }
    }
    out[(long long)blockIdx.x * blockDim.x + threadIdx.x] += acc;  // (one store per thread: an atomic on one address would be the whole kernel)
}
template <int MODE, int ILP>
static void run(const char *name, const double *T, u64 mask, long long M, double *out) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int w = 0; w < 3; w++) hipLaunchKernelGGL((k_gather<MODE, ILP>), dim3(256), dim3(1024), 0, 0, T, mask, M, out);
    hipMemset(out, 0, 256 * 1024 * 8);
    const int reps = 20;
    hipEventRecord(e0);
    for (int r = 0; r < reps; r++) hipLaunchKernelGGL((k_gather<MODE, ILP>), dim3(256), dim3(1024), 0, 0, T, mask, M, out);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    std::vector<double> hv(256 * 1024);
    hipMemcpy(hv.data(), out, hv.size() * 8, hipMemcpyDeviceToHost);
    double h[2] = {0, 0};
    for (double x : hv) h[1] += x;
    printf("%-28s ilp %d  %8.2f us per call  %7.1f GB/s of 64-B pieces  checksum %.6e\n", name, ILP, ms * 1000 / reps, M * 64.0 / (ms / reps * 1e-3) / 1e9, h[1] / reps);
}
int main(int argc, char **argv) {
    const int ll = argc > 1 ? atoi(argv[1]) : 17, ld = argc > 2 ? atoi(argv[2]) : 20;
    const u64 lines = 1ull << ll;
    const long long M = 1ll << ld;
    std::vector<double> h(lines * 16);
    for (size_t i = 0; i < h.size(); i++) h[i] = (double)((i * 2654435761u) % 1000) * 1e-3;
    double *T, *out;
    hipMalloc(&T, h.size() * 8);
    hipMalloc(&out, 256 * 1024 * 8);
    hipMemcpy(T, h.data(), h.size() * 8, hipMemcpyHostToDevice);
    printf("lines 2^%d (%.1f MB), draws 2^%d\n", ll, lines * 128.0 / 1e6, ld);
    run<0, 2>("a: 4 x 16 B per lane", T, lines - 1, M, out);
    run<1, 2>("b: quad-cooperative", T, lines - 1, M, out);
    run<2, 2>("c: 1 x 16 B per lane", T, lines - 1, M, out);
    run<3, 2>("d: 1 + 3 dependent", T, lines - 1, M, out);
    run<0, 4>("a: 4 x 16 B per lane", T, lines - 1, M, out);
    run<1, 4>("b: quad-cooperative", T, lines - 1, M, out);
    run<2, 4>("c: 1 x 16 B per lane", T, lines - 1, M, out);
    return 0;
}
