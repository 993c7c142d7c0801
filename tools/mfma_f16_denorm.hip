// Does v_mfma_f32_32x32x16_f16 keep float16 DENORMAL inputs?  (a uint8 pixel n zero-extended to 16 bits is the f16 denormal n * 2^-24: one v_perm_b32 per two pixels
// instead of a conversion per pixel, if the matrix pipe does not flush it).  Prints the product of A = n * 2^-24 (denormal bit patterns) with B = 1.0 summed over K = 16.
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_f16_denorm.hip -o tools/_build/mfma_f16_denorm && tools/_build/mfma_f16_denorm
#include <hip/hip_runtime.h>
#include <cstdio>
using f16x8 = __attribute__((ext_vector_type(8))) _Float16;
using f32x16 = __attribute__((ext_vector_type(16))) float;
__global__ void k(float *out) {
    const int lane = threadIdx.x;
    unsigned short bits[8];
    for (int j = 0; j < 8; j++) bits[j] = (unsigned short)(lane & 31);  // row i of A: the denormal i * 2^-24 in every k
    f16x8 a, b;
    __builtin_memcpy(&a, bits, 16);
    for (int j = 0; j < 8; j++) b[j] = (_Float16)1.0f;
    f32x16 acc;
    for (int r = 0; r < 16; r++) acc[r] = 0.f;
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc, 0, 0, 0);
    // C/D layout: col = lane & 31, row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5)
    for (int r = 0; r < 16; r++) out[((r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)) * 32 + (lane & 31)] = acc[r];
}
int main() {
    float *d, h[1024];
    hipMalloc(&d, sizeof(h));
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    int ok = 1;
    for (int i = 0; i < 32; i++) {
        const float want = 16.0f * i * 5.9604644775390625e-8f;
        if (h[i * 32] != want) ok = 0;
    }
    printf("row 1: %g (want %g), row 31: %g (want %g): denormal inputs %s\n", h[32], 16.0f * 5.9604644775390625e-8f, h[31 * 32], 16.0f * 31 * 5.9604644775390625e-8f,
           ok ? "KEPT" : "FLUSHED");
    return 0;
}
