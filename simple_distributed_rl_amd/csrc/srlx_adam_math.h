// srlx_adam_math.h -- the per-element Adam arithmetic shared by k_adam (srlx_train.hip) and the first dense layer's
// weight-gradient kernel that applies it in its epilogue (srlx_qnet_bwd.hip).  Order of operations follows torch's Adam
// (model_torch.py:71,109): step starts at 1; exp_avg = lerp(exp_avg, g, 1 - b1); bias corrections in double;
// denom = sqrt(v) / sqrt(bc2) + eps; p -= (lr / bc1) * m / denom, evaluated in float32.
#pragma once
#include <stdint.h>

#include <hip/hip_runtime.h>

namespace srlx {

struct AdamCoef {
    float w1, b2, w2, step_size, bc2_sqrt, eps;
};

// x^n for n >= 0 by repeated squaring (wave-uniform loop, about 2 log2(n) double multiplies, a few ulp from pow()): the
// library pow() costs every wave ~1000 instructions
__device__ __forceinline__ double powi(double x, int64_t n) {
    double r = 1.0;
    while (n > 0) {
        if (n & 1) r *= x;
        x *= x;
        n >>= 1;
    }
    return r;
}

// `steps_taken` = optimiser steps already applied (this is step steps_taken + 1)
__device__ __forceinline__ AdamCoef adam_coef(double lr, double beta1, double beta2, double eps, int64_t steps_taken) {
    const int64_t step = steps_taken + 1;
    const double bc1 = 1.0 - powi(beta1, step), bc2 = 1.0 - powi(beta2, step);
    return AdamCoef{(float)(1.0 - beta1), (float)beta2, (float)(1.0 - beta2), (float)(lr / bc1), (float)sqrt(bc2), (float)eps};
}

__device__ __forceinline__ void adam_one(float &p, float g, float &m, float &v, const AdamCoef &c) {
    m = m + c.w1 * (g - m);
    v = c.b2 * v + (c.w2 * g) * g;
    const float denom = sqrtf(v) / c.bc2_sqrt + c.eps;
    p = p - (c.step_size * m) / denom;
}

}  // namespace srlx
