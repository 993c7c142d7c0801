// srlx_image.hip -- frame preprocessing on the device: colour -> gray, trimming, resize, into uint8 (and/or normalised float32).
//
// Replaces srl/rl/processors/image_processor.py:104-151 (`cv2.cvtColor(COLOR_RGB2GRAY)`, the trimming slice, `cv2.resize(state, (w, h))`
// with the default INTER_LINEAR, `state.astype(float32) / 255`): in the reference every real frame (ALE: 210 x 160 x 3 uint8) goes through
// OpenCV on the host and reaches the network as a float32 array.  Here the raw frames of E environments are uploaded once (uint8) and
// ONE launch writes the 84 x 84 gray uint8 frames the ring stores -- no float32 round trip; a normalised float32 copy is optional (the
// single-frame drop-in `ImageProcessor.remap_observation`).
// Arithmetic = OpenCV's 8-bit fixed-point paths, restated in oracle/image_oracle.py (which cites them): 14-bit gray coefficients,
// 11-bit interpolation weights with round-half-even, the two-stage shift of VResizeLinear, the 2 x 2 INTER_AREA special case.  One
// thread per output pixel (x fastest: coalesced stores; the four source taps of neighbouring threads share cache lines).
#include "srlx_common.h"

namespace {
using i64 = int64_t;
using u8 = unsigned char;

struct ImgGeo {
    int H, W, C;              // source
    int top, left, th, tw;    // trimming window (th x tw)
    int oh, ow;               // output
    int to_gray;              // C == 3 -> one gray channel
    int area2;                // exact 2 x 2 down-scale
};

__device__ __forceinline__ int gray_of(const u8 *p) { return (p[0] * 4899 + p[1] * 9617 + p[2] * 1868 + 8192) >> 14; }

__device__ __forceinline__ int tap(const u8 *img, const ImgGeo &g, int y, int x, int c) {  // (y, x) inside the trimming window
    const u8 *p = img + ((i64)(g.top + y) * g.W + (g.left + x)) * g.C;
    return (g.to_gray && g.C == 3) ? gray_of(p) : p[c];
}

// first source index and the two int16 weights of one output coordinate (OpenCV resize.cpp: float32 coordinate, cvRound(w * 2048))
__device__ __forceinline__ void axis(int d, int src, int dst, int &s, int &w0, int &w1) {
    const double scale = (double)src / (double)dst;
    float f = (float)((d + 0.5) * scale - 0.5);
    int si = (int)floorf(f);
    f -= (float)si;
    if (si < 0) si = 0, f = 0.f;
    if (si >= src - 1) si = src - 1, f = 0.f;
    s = si;
    w0 = (int)rintf((1.0f - f) * 2048.0f);
    w1 = (int)rintf(f * 2048.0f);
}

__global__ void __launch_bounds__(256) k_image_preprocess(ImgGeo g, i64 n, const u8 *__restrict__ src, u8 *__restrict__ out_u8, float *__restrict__ out_f32,
                                                          int norm, float max_val) {
    const int oc = (g.to_gray || g.C == 1) ? 1 : g.C;
    const i64 per = (i64)g.oh * g.ow * oc;
    const i64 t = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n * per) return;
    const i64 b = t / per;
    const int r = (int)(t % per);
    const int c = r % oc, x = (r / oc) % g.ow, y = r / (oc * g.ow);
    const u8 *img = src + b * (i64)g.H * g.W * g.C;
    int v;
    if (g.oh == g.th && g.ow == g.tw) {
        v = tap(img, g, y, x, c);
    } else if (g.area2) {
        v = (tap(img, g, 2 * y, 2 * x, c) + tap(img, g, 2 * y, 2 * x + 1, c) + tap(img, g, 2 * y + 1, 2 * x, c) + tap(img, g, 2 * y + 1, 2 * x + 1, c) + 2) >> 2;
    } else {
        int sx, ax0, ax1, sy, ay0, ay1;
        axis(x, g.tw, g.ow, sx, ax0, ax1);
        axis(y, g.th, g.oh, sy, ay0, ay1);
        const int x1 = sx + 1 < g.tw ? sx + 1 : g.tw - 1, y1 = sy + 1 < g.th ? sy + 1 : g.th - 1;
        const int s0 = tap(img, g, sy, sx, c) * ax0 + tap(img, g, sy, x1, c) * ax1;  // horizontal pass, scale 2^11
        const int s1 = tap(img, g, y1, sx, c) * ax0 + tap(img, g, y1, x1, c) * ax1;
        v = (((ay0 * (s0 >> 4)) >> 16) + ((ay1 * (s1 >> 4)) >> 16) + 2) >> 2;
        v = v < 0 ? 0 : (v > 255 ? 255 : v);
    }
    if (out_u8) out_u8[t] = (u8)v;
    if (out_f32) {
        const float f = (float)v;
        out_f32[t] = norm == 1 ? __fdiv_rn(f, max_val) : (norm == 2 ? __fdiv_rn(f * 2.0f, max_val) - 1.0f : f);
    }
}
}  // namespace

extern "C" {

int srlx_image_preprocess(int64_t n_images, int src_h, int src_w, int src_channels, const uint8_t *d_src, int to_gray, int trim_top, int trim_left,
                          int trim_bottom, int trim_right, int out_h, int out_w, uint8_t *d_out_u8, float *d_out_f32, int normalize, double max_val, void *stream) {
    SRLX_REQUIRE(n_images > 0 && d_src && (d_out_u8 || d_out_f32), "image_preprocess: NULL argument");
    SRLX_REQUIRE(src_h > 0 && src_w > 0 && (src_channels == 1 || src_channels == 3), "image_preprocess: uint8 images of 1 or 3 channels");
    SRLX_REQUIRE(trim_top >= 0 && trim_left >= 0 && trim_bottom <= src_h && trim_right <= src_w && trim_top < trim_bottom && trim_left < trim_right,
                 "image_preprocess: bad trimming window");
    SRLX_REQUIRE(out_h > 0 && out_w > 0 && normalize >= 0 && normalize <= 2, "image_preprocess: bad output size / normalisation");
    ImgGeo g{src_h, src_w, src_channels, trim_top, trim_left, trim_bottom - trim_top, trim_right - trim_left, out_h, out_w, to_gray ? 1 : 0, 0};
    g.area2 = (g.th == 2 * out_h && g.tw == 2 * out_w) ? 1 : 0;
    const int oc = (g.to_gray || src_channels == 1) ? 1 : src_channels;
    const int64_t total = n_images * (int64_t)out_h * out_w * oc;
    hipLaunchKernelGGL(k_image_preprocess, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, g, n_images, d_src, d_out_u8, d_out_f32, normalize,
                       (float)max_val);
    SRLX_HIP(hipGetLastError());
    return SRLX_OK;
}

}  // extern "C"
