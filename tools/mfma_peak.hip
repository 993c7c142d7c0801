// Probe: what v_mfma_f32_32x32x2_f32 sustains on this box when nothing else limits it (clock included).
// hipcc --offload-arch=gfx950 -O3 tools/mfma_peak.hip -o /tmp/mfma_peak && /tmp/mfma_peak
#include <hip/hip_runtime.h>
#include <stdio.h>
using f32x16 = __attribute__((ext_vector_type(16))) float;
__global__ void __launch_bounds__(256) k(float *out, int iters) {
    f32x16 a0 = {0}, a1 = {0}, a2 = {0}, a3 = {0};
    float x = threadIdx.x * 1e-3f, y = blockIdx.x * 1e-3f;
    for (int i = 0; i < iters; i++) {
        a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a0, 0, 0, 0);
        a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a1, 0, 0, 0);
        a2 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a2, 0, 0, 0);
        a3 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a3, 0, 0, 0);
    }
    float s = 0;
    for (int r = 0; r < 16; r++) s += a0[r] + a1[r] + a2[r] + a3[r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
int main() {
    float *d;
    hipMalloc(&d, 4096 * 256 * 4);
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    for (int wgs : {256, 512, 1024, 2048}) {
        const int iters = 4096;
        hipLaunchKernelGGL(k, dim3(wgs), dim3(256), 0, 0, d, iters);
        hipDeviceSynchronize();
        hipEventRecord(a);
        for (int r = 0; r < 5; r++) hipLaunchKernelGGL(k, dim3(wgs), dim3(256), 0, 0, d, iters);
        hipEventRecord(b);
        hipEventSynchronize(b);
        float ms;
        hipEventElapsedTime(&ms, a, b);
        const double flops = 5.0 * wgs * 4 /*waves*/ * (double)iters * 4 /*mfma per iter*/ * 4096.0;
        printf("%4d workgroups: %.1f TFLOP/s  (%.3f ms per launch)\n", wgs, flops / (ms * 1e-3) / 1e12, ms / 5);
    }
    return 0;
}
