// Probe: what v_mfma_f32_32x32x16_bf16 sustains on this box (clock included) as a function of independent accumulator chains per wave and waves per SIMD,
// chip-wide (256 CUs) -- the practical ceiling behind `roofline.frac` of the split-bf16 kernels (nominal: 2.5 PFLOP/s at 2.4 GHz).
// hipcc --offload-arch=gfx950 -O3 tools/mfma_bf16_peak.hip -o /tmp/mfma_bf16_peak && /tmp/mfma_bf16_peak
#include <hip/hip_runtime.h>
#include <stdio.h>
using f32x16 = __attribute__((ext_vector_type(16))) float;
using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
template <int CHAINS>
__global__ void __launch_bounds__(256) k(float *out, int iters, unsigned long long *clk) {
    f32x16 a[CHAINS];
    for (int c = 0; c < CHAINS; c++)
        for (int r = 0; r < 16; r++) a[c][r] = 0.f;
    bf16x8 x, y;
    for (int j = 0; j < 8; j++) x[j] = (__bf16)(threadIdx.x * 1e-3f + j), y[j] = (__bf16)(blockIdx.x * 1e-3f + j);
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int c = 0; c < CHAINS; c++) a[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, a[c], 0, 0, 0);
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0;
    for (int c = 0; c < CHAINS; c++)
        for (int r = 0; r < 16; r++) s += a[c][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (blockIdx.x == 0 && threadIdx.x == 0) clk[0] = t1 - t0;
}
template <int CHAINS>
void run(float *d, unsigned long long *clk, int wgs) {
    const int iters = 16384 / CHAINS;
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    hipLaunchKernelGGL((k<CHAINS>), dim3(wgs), dim3(256), 0, 0, d, iters, clk);
    hipDeviceSynchronize();
    hipEventRecord(a);
    for (int r = 0; r < 5; r++) hipLaunchKernelGGL((k<CHAINS>), dim3(wgs), dim3(256), 0, 0, d, iters, clk);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms;
    hipEventElapsedTime(&ms, a, b);
    unsigned long long c;
    hipMemcpy(&c, clk, 8, hipMemcpyDeviceToHost);
    const double n = (double)iters * CHAINS;  // MFMAs per wave
    const double flops = 5.0 * wgs * 4 * n * 32768.0;
    printf("chains %d  waves/SIMD %d : %7.1f TFLOP/s = %.3f of 2500   %.1f shader clocks per MFMA and wave, %.1f per MFMA and SIMD, clock %.2f GHz\n", CHAINS, wgs / 256,
           flops / (ms * 1e-3) / 1e12, flops / (ms * 1e-3) / 2.5e15, c / n, c / n / (wgs / 256), c / (ms / 5 * 1e-3) / 1e9);
}
int main() {
    float *d;
    unsigned long long *clk;
    hipMalloc(&d, 4096 * 256 * 4);
    hipMalloc(&clk, 8);
    for (int wgs : {256, 512, 1024}) {
        run<1>(d, clk, wgs);
        run<2>(d, clk, wgs);
        run<4>(d, clk, wgs);
    }
    return 0;
}
