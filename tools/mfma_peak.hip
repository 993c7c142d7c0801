// Probe: what v_mfma_f32_32x32x2_f32 sustains on this box (clock included) as a function of the number of independent
// accumulator chains per wave, waves per SIMD, and VALU instructions threaded between the MFMAs.
// hipcc --offload-arch=gfx950 -O3 tools/mfma_peak.hip -o /tmp/mfma_peak && /tmp/mfma_peak
#include <hip/hip_runtime.h>
#include <stdio.h>
using f32x16 = __attribute__((ext_vector_type(16))) float;
template <int CHAINS, int VALU>
__global__ void __launch_bounds__(256) k(float *out, int iters) {
    f32x16 a[CHAINS];
    for (int c = 0; c < CHAINS; c++)
        for (int r = 0; r < 16; r++) a[c][r] = 0.f;
    float x = threadIdx.x * 1e-3f, y = blockIdx.x * 1e-3f, z = 1.0001f;
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int c = 0; c < CHAINS; c++) {
#pragma unroll
            for (int v = 0; v < VALU; v++) x = fmaf(x, z, 1e-7f);  // dependent VALU work feeding the next MFMA operand
            a[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a[c], 0, 0, 0);
        }
    }
    float s = 0;
    for (int c = 0; c < CHAINS; c++)
        for (int r = 0; r < 16; r++) s += a[c][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int CHAINS, int VALU>
void run(float *d, int wgs) {
    const int iters = 8192 / CHAINS;
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    hipLaunchKernelGGL((k<CHAINS, VALU>), dim3(wgs), dim3(256), 0, 0, d, iters);
    hipDeviceSynchronize();
    hipEventRecord(a);
    for (int r = 0; r < 5; r++) hipLaunchKernelGGL((k<CHAINS, VALU>), dim3(wgs), dim3(256), 0, 0, d, iters);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms;
    hipEventElapsedTime(&ms, a, b);
    const double flops = 5.0 * wgs * 4 * (double)iters * CHAINS * 4096.0;
    printf("chains %d  valu/mfma %d  waves/SIMD %d : %.1f TFLOP/s\n", CHAINS, VALU, wgs / 256, flops / (ms * 1e-3) / 1e12);
}
int main() {
    float *d;
    hipMalloc(&d, 4096 * 256 * 4);
    for (int wgs : {256, 512, 1024}) {
        run<1, 0>(d, wgs);
        run<2, 0>(d, wgs);
        run<4, 0>(d, wgs);
        run<1, 4>(d, wgs);
        run<2, 4>(d, wgs);
        run<1, 8>(d, wgs);
        run<2, 8>(d, wgs);
    }
    return 0;
}
