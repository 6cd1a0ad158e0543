// Input pipeline on the device: the transform chain of train_model.py:233-247,308-309 and
// eval_model.py:45-56 -- Resize (PIL bilinear with antialias), CenterCrop, ToTensor and the two Normalize maps -- for one
// decoded image (uint8, height x width x RGB) in ONE launch that writes up to three float planes sets: the [0,1] image, the
// classifier-normalised copy and the critic-normalised copy.
//
// The resize is PIL's ImagingResample for 8-bit images restated: a horizontal pass and a vertical pass, each a weighted sum
// over a window of source pixels with integer coefficients (22 fractional bits, rounded to nearest from the normalised double
// weights), an initial 1 << 21 for rounding, and a clip to 0..255 after EACH pass.  The coefficient tables are computed on the
// host exactly as PIL computes them (imageio.resample_table) and only for the rows / columns the crop keeps; the kernel does the
// integer arithmetic, so the result is bit-identical to PIL's.  ToTensor is a correctly rounded division by 255, Normalize a
// subtraction and a correctly rounded division (transforms.Normalize: sub_ then div_), as sempyr_normalize.
#include "common.h"

namespace {

constexpr int PRECISION_BITS = 32 - 8 - 2;

__device__ __forceinline__ int clip8(int v) {
    v >>= PRECISION_BITS;
    return v < 0 ? 0 : (v > 255 ? 255 : v);
}

struct NormArgs { float mean[3], std[3]; };

// htab / vtab: per output column / row {first source index, number of taps, coefficients[k]}.  One thread per output pixel;
// the horizontal pass of the (few) source rows a pixel needs is recomputed per pixel: 9..81 multiply-adds per channel, and the
// source bytes of neighbouring pixels overlap in L1/L2.
__global__ __launch_bounds__(256) void resize_crop_normalize_kernel(const uint8_t* __restrict__ src, int H, int W, int crop_h, int crop_w,
                                                                    const int* __restrict__ htab, int hk,
                                                                    const int* __restrict__ vtab, int vk,
                                                                    NormArgs na, float* __restrict__ out_a,
                                                                    NormArgs nb, float* __restrict__ out_b,
                                                                    float* __restrict__ out01)
{
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= crop_h * crop_w) return;
    const int oy = idx / crop_w, ox = idx - oy * crop_w;
    const int* hr = htab + (size_t)ox * (2 + hk);
    const int* vr = vtab + (size_t)oy * (2 + vk);
    const int xmin = hr[0], xn = hr[1], ymin = vr[0], yn = vr[1];
    int acc[3] = {1 << (PRECISION_BITS - 1), 1 << (PRECISION_BITS - 1), 1 << (PRECISION_BITS - 1)};
    for (int r = 0; r < yn; ++r) {
        const uint8_t* row = src + ((size_t)(ymin + r) * W + xmin) * 3;
        int h0 = 1 << (PRECISION_BITS - 1), h1 = h0, h2 = h0;
        for (int c = 0; c < xn; ++c) {
            const int k = hr[2 + c];
            h0 += k * row[3 * c]; h1 += k * row[3 * c + 1]; h2 += k * row[3 * c + 2];
        }
        const int kv = vr[2 + r];
        acc[0] += kv * clip8(h0); acc[1] += kv * clip8(h1); acc[2] += kv * clip8(h2);
    }
    const size_t plane = (size_t)crop_h * crop_w;
#pragma unroll
    for (int ch = 0; ch < 3; ++ch) {
        const float v = (float)clip8(acc[ch]) / 255.0f;
        if (out01) out01[ch * plane + idx] = v;
        if (out_a) out_a[ch * plane + idx] = (v - na.mean[ch]) / na.std[ch];
        if (out_b) out_b[ch * plane + idx] = (v - nb.mean[ch]) / nb.std[ch];
    }
}

}  // namespace

extern "C" int sempyr_resize_crop_normalize(const uint8_t* src, int H, int W, int crop_h, int crop_w,
                                            const int32_t* htab, int hk, const int32_t* vtab, int vk,
                                            const float* mean_a, const float* std_a, float* out_a,
                                            const float* mean_b, const float* std_b, float* out_b,
                                            float* out01, void* stream)
{
    if (!src || !htab || !vtab || H < 1 || W < 1 || crop_h < 1 || crop_w < 1 || hk < 1 || vk < 1)
        SEMPYR_FAIL("resize_crop_normalize: bad arguments");
    if ((out_a && (!mean_a || !std_a)) || (out_b && (!mean_b || !std_b)))
        SEMPYR_FAIL("resize_crop_normalize: a normalised output needs its mean / std (3 floats each, host memory)");
    if (!out_a && !out_b && !out01) SEMPYR_FAIL("resize_crop_normalize: no output");
    if ((long long)H * W * 3 >= (1ll << 31)) SEMPYR_FAIL("resize_crop_normalize: image too large");
    NormArgs na = {{0.f, 0.f, 0.f}, {1.f, 1.f, 1.f}}, nb = na;
    for (int i = 0; i < 3; ++i) {
        if (out_a) { na.mean[i] = mean_a[i]; na.std[i] = std_a[i]; }
        if (out_b) { nb.mean[i] = mean_b[i]; nb.std[i] = std_b[i]; }
    }
    hipLaunchKernelGGL(resize_crop_normalize_kernel, dim3(ceil_div((long long)crop_h * crop_w, 256)), dim3(256), 0, (hipStream_t)stream,
                       src, H, W, crop_h, crop_w, htab, hk, vtab, vk, na, out_a, nb, out_b, out01);
    SEMPYR_CHECK_LAUNCH();
    return 0;
}
