// Probe: how fast can ONE lane replay n dependent fp64 additions fed from LDS (the root of a PER add)?
// build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off tools/chain_probe.hip -o /tmp/chain_probe
#include <hip/hip_runtime.h>
#include <cstdio>
template <int U>
__global__ void k_chain(const double *in, int n, double *out, long long *cyc) {
    extern __shared__ double s[];
    for (int i = threadIdx.x; i < n; i += blockDim.x) s[i] = in[i];
    __syncthreads();
    if (threadIdx.x == 0) {
        double v = 1.0;
        long long t0 = clock64();
        for (int k = 0; k + U <= n; k += U) {
            double c[U];
#pragma unroll
            for (int u = 0; u < U; u++) c[u] = s[k + u];
#pragma unroll
            for (int u = 0; u < U; u++) v += c[u];
        }
        long long t1 = clock64();
        out[0] = v;
        cyc[0] = t1 - t0;
    }
}
// variant: values pre-loaded into registers by 64 lanes, chain walks lanes with readlane (no LDS in the chain)
__global__ void k_chain_readlane(const double *in, int n, double *out, long long *cyc) {
    const int lane = threadIdx.x;
    double v = 1.0;
    long long t0 = clock64();
    for (int base = 0; base < n; base += 64 * 4) {
        double r0 = in[base + lane * 4 + 0], r1 = in[base + lane * 4 + 1], r2 = in[base + lane * 4 + 2], r3 = in[base + lane * 4 + 3];
#pragma unroll
        for (int l = 0; l < 64; l++) {
            v += __shfl(r0, l);
            v += __shfl(r1, l);
            v += __shfl(r2, l);
            v += __shfl(r3, l);
        }
    }
    long long t1 = clock64();
    if (lane == 0) { out[0] = v; cyc[0] = t1 - t0; }
}
int main() {
    const int n = 8192;
    double *in, *out; long long *cyc;
    hipMalloc(&in, n * 8); hipMalloc(&out, 8); hipMalloc(&cyc, 8);
    double *h = new double[n]; for (int i = 0; i < n; i++) h[i] = 1e-3 * i;
    hipMemcpy(in, h, n * 8, hipMemcpyHostToDevice);
    long long c;
    for (int rep = 0; rep < 2; rep++) {
        hipLaunchKernelGGL(k_chain<8>, dim3(1), dim3(256), n * 8, 0, in, n, out, cyc); hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
        if (rep) printf("LDS unroll 8 : %.1f clock64 ticks per add\n", (double)c / n);
        hipLaunchKernelGGL(k_chain<32>, dim3(1), dim3(256), n * 8, 0, in, n, out, cyc); hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
        if (rep) printf("LDS unroll 32: %.1f ticks per add\n", (double)c / n);
        hipLaunchKernelGGL(k_chain<64>, dim3(1), dim3(256), n * 8, 0, in, n, out, cyc); hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
        if (rep) printf("LDS unroll 64: %.1f ticks per add\n", (double)c / n);
        hipLaunchKernelGGL(k_chain_readlane, dim3(1), dim3(64), 0, 0, in, n, out, cyc); hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
        if (rep) printf("readlane     : %.1f ticks per add\n", (double)c / n);
    }
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipEventRecord(a); for (int i = 0; i < 20; i++) hipLaunchKernelGGL(k_chain<32>, dim3(1), dim3(256), n * 8, 0, in, n, out, cyc); hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b); printf("k_chain<32> n=8192: %.1f us per launch\n", ms / 20 * 1e3);
    hipEventRecord(a); for (int i = 0; i < 20; i++) hipLaunchKernelGGL(k_chain_readlane, dim3(1), dim3(64), 0, 0, in, n, out, cyc); hipEventRecord(b); hipEventSynchronize(b);
    hipEventElapsedTime(&ms, a, b); printf("k_chain_readlane n=8192: %.1f us per launch\n", ms / 20 * 1e3);
    return 0;
}
