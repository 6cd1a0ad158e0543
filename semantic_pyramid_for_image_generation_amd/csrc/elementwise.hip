// HBM-bound elementwise kernels of the hot path (coalesced float4 where rows allow it):
// fused per-row affine + residual + activation (norm apply / eval BatchNorm / residual add,
// generator.py:27-29,56-57,87 ; discriminator.py:37-39,63-64), activation backward, feature*mask
// (generator.py:126-129), the WGAN-GP interpolation (train_model.py:51-55), the Normalize maps
// (train_model.py:240-245), 2x2 pooling / nearest upsample (discriminator.py:57,103 ;
// generator.py:44) and the ResNet stem max-pool (classifier.py:36).
#include "common.h"

namespace {

inline bool aligned16(const void* p) { return ((uintptr_t)p & 15) == 0; }

// y = act((x - sub[row]) * scale[row] + shift[row] + R)      (sub optional)
// R = res, or (res - rsub[row]) * rscale[row] + rshift[row] when the residual carries its own per-row affine (two norm
// layers summed: generator.py:54-57).
__global__ __launch_bounds__(256) void affine_act_kernel(const float* __restrict__ x, const float* __restrict__ sub,
                                                         const float* __restrict__ scale,
                                                         const float* __restrict__ shift, const float* __restrict__ res,
                                                         const float* __restrict__ rsub, const float* __restrict__ rscale,
                                                         const float* __restrict__ rshift,
                                                         float* __restrict__ y, long long total, int len, int vec,
                                                         int act, float slope, int res_w)
{
    // res_w > 0: `res` has HALF the resolution (rows of res_w/2 pixels, len/4 per plane) and is read through a nearest x2
    // upsample -- a 1x1 convolution commutes with nearest upsampling, so the skip branch of an upsampling residual block
    // is computed (and normalised) at the low resolution and only expanded here.
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (vec) {
        if (i * 4 >= total) return;
        const int row = (int)((i * 4) / len);
        float sc = 1.f, sh = 0.f, mu = 0.f;
        if (scale) { sc = scale[row]; sh = shift[row]; if (sub) mu = sub[row]; }
        float4 v = reinterpret_cast<const float4*>(x)[i];
        v.x = (v.x - mu) * sc + sh; v.y = (v.y - mu) * sc + sh; v.z = (v.z - mu) * sc + sh; v.w = (v.w - mu) * sc + sh;
        if (res) {
            float4 r;
            if (res_w > 0) {
                const int rem = (int)(i * 4 - (long long)row * len);
                const int yy = rem / res_w, xx = rem - yy * res_w;
                const float2 lo = *reinterpret_cast<const float2*>(res + (size_t)row * (len >> 2) + (yy >> 1) * (res_w >> 1) + (xx >> 1));
                r = make_float4(lo.x, lo.x, lo.y, lo.y);
            } else {
                r = reinterpret_cast<const float4*>(res)[i];
            }
            if (rscale) {
                const float rs = rscale[row], rh = rshift[row], rm = rsub ? rsub[row] : 0.f;
                r.x = (r.x - rm) * rs + rh; r.y = (r.y - rm) * rs + rh; r.z = (r.z - rm) * rs + rh; r.w = (r.w - rm) * rs + rh;
            }
            v.x += r.x; v.y += r.y; v.z += r.z; v.w += r.w;
        }
        v.x = act_apply(v.x, act, slope); v.y = act_apply(v.y, act, slope);
        v.z = act_apply(v.z, act, slope); v.w = act_apply(v.w, act, slope);
        reinterpret_cast<float4*>(y)[i] = v;
    } else {
        if (i >= total) return;
        const int row = (int)(i / len);
        float v = x[i];
        if (scale) v = (v - (sub ? sub[row] : 0.f)) * scale[row] + shift[row];
        if (res) {
            float r;
            if (res_w > 0) {
                const int rem = (int)(i - (long long)row * len);
                const int yy = rem / res_w, xx = rem - yy * res_w;
                r = res[(size_t)row * (len >> 2) + (yy >> 1) * (res_w >> 1) + (xx >> 1)];
            } else {
                r = res[i];
            }
            if (rscale) r = (r - (rsub ? rsub[row] : 0.f)) * rscale[row] + rshift[row];
            v += r;
        }
        y[i] = act_apply(v, act, slope);
    }
}

// g' = gy * act'(y);  gx = g' * scale[row] (scale NULL -> 1);  gres = g' (optional)
__global__ __launch_bounds__(256) void affine_act_bwd_kernel(const float* __restrict__ gy, const float* __restrict__ y,
                                                             const float* __restrict__ scale, float* __restrict__ gx,
                                                             float* __restrict__ gres, long long total, int len, int vec,
                                                             int act, float slope)
{
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (vec) {
        if (i * 4 >= total) return;
        float4 g = reinterpret_cast<const float4*>(gy)[i];
        if (act != SEMPYR_ACT_NONE) {
            const float4 yv = reinterpret_cast<const float4*>(y)[i];
            g.x *= act_deriv_y(yv.x, act, slope); g.y *= act_deriv_y(yv.y, act, slope);
            g.z *= act_deriv_y(yv.z, act, slope); g.w *= act_deriv_y(yv.w, act, slope);
        }
        if (gres) reinterpret_cast<float4*>(gres)[i] = g;
        if (gx) {
            if (scale) { const float sc = scale[(int)((i * 4) / len)]; g.x *= sc; g.y *= sc; g.z *= sc; g.w *= sc; }
            reinterpret_cast<float4*>(gx)[i] = g;
        }
    } else {
        if (i >= total) return;
        float g = gy[i];
        if (act != SEMPYR_ACT_NONE) g *= act_deriv_y(y[i], act, slope);
        if (gres) gres[i] = g;
        if (gx) gx[i] = scale ? g * scale[(int)(i / len)] : g;
    }
}

__global__ __launch_bounds__(256) void mul_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                  float* __restrict__ y, long long total, int vec)
{
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (vec) {
        if (i * 4 >= total) return;
        const float4 u = reinterpret_cast<const float4*>(a)[i], v = reinterpret_cast<const float4*>(b)[i];
        reinterpret_cast<float4*>(y)[i] = make_float4(u.x * v.x, u.y * v.y, u.z * v.z, u.w * v.w);
    } else {
        if (i >= total) return;
        y[i] = a[i] * b[i];
    }
}

// y = alpha * a + beta * b   (b may be NULL)
__global__ __launch_bounds__(256) void axpby_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                    float alpha, float beta, float* __restrict__ y, long long total, int vec)
{
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (vec) {
        if (i * 4 >= total) return;
        float4 u = reinterpret_cast<const float4*>(a)[i];
        u.x *= alpha; u.y *= alpha; u.z *= alpha; u.w *= alpha;
        if (b) { const float4 v = reinterpret_cast<const float4*>(b)[i]; u.x += beta * v.x; u.y += beta * v.y; u.z += beta * v.z; u.w += beta * v.w; }
        reinterpret_cast<float4*>(y)[i] = u;
    } else {
        if (i >= total) return;
        y[i] = alpha * a[i] + (b ? beta * b[i] : 0.f);
    }
}

// out = real + alpha[row] * (fake - real)
__global__ __launch_bounds__(256) void lerp_rows_kernel(const float* __restrict__ real, const float* __restrict__ fake,
                                                        const float* __restrict__ alpha, float* __restrict__ out,
                                                        long long total, int len, int vec)
{
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (vec) {
        if (i * 4 >= total) return;
        const float a = alpha[(int)((i * 4) / len)];
        const float4 r = reinterpret_cast<const float4*>(real)[i], f = reinterpret_cast<const float4*>(fake)[i];
        reinterpret_cast<float4*>(out)[i] = make_float4(r.x + a * (f.x - r.x), r.y + a * (f.y - r.y),
                                                        r.z + a * (f.z - r.z), r.w + a * (f.w - r.w));
    } else {
        if (i >= total) return;
        const float a = alpha[(int)(i / len)];
        out[i] = real[i] + a * (fake[i] - real[i]);
    }
}

// mode 0: y = (x - mean[c]) / std[c]   (true division, as transforms.Normalize's sub_/div_)
// mode 1: y = x / std[c]                (its backward)
__global__ __launch_bounds__(256) void normalize_kernel(const float* __restrict__ x, const float* __restrict__ mean,
                                                        const float* __restrict__ stdv, float* __restrict__ y,
                                                        long long total, int C, int HW, int vec, int mode)
{
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (vec) {
        if (i * 4 >= total) return;
        const int c = (int)(((i * 4) / HW) % C);
        const float m = mode == 0 ? mean[c] : 0.f, s = stdv[c];
        float4 v = reinterpret_cast<const float4*>(x)[i];
        v.x = (v.x - m) / s; v.y = (v.y - m) / s; v.z = (v.z - m) / s; v.w = (v.w - m) / s;
        reinterpret_cast<float4*>(y)[i] = v;
    } else {
        if (i >= total) return;
        const int c = (int)((i / HW) % C);
        y[i] = (x[i] - (mode == 0 ? mean[c] : 0.f)) / stdv[c];
    }
}

// y[r][h][w] = scale * sum_{a,b in 0..1} x[r][2h+a][2w+b] * act'(yref[r][2h+a][2w+b])      (H, W even; h = H/2)
// yref (optional) is the OUTPUT of a sign-preserving activation at the fine resolution: with it this is the backward of
// pool2_up_kernel's fused form below.
__global__ __launch_bounds__(256) void pool2_down_kernel(const float* __restrict__ x, const float* __restrict__ yref,
                                                         float* __restrict__ y, long long total_out, int h, int w,
                                                         float scale, int act, float slope)
{
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total_out) return;
    const int ox = (int)(i % w);
    const long long t = i / w;
    const int oy = (int)(t % h);
    const long long r = t / h;
    const long long off = (r * (2 * h) + 2 * oy) * (2 * (long long)w);
    const float2* p0 = reinterpret_cast<const float2*>(x + off) + ox;
    float2 a = *p0, b = *(p0 + w);
    if (yref) {
        const float2* q0 = reinterpret_cast<const float2*>(yref + off) + ox;
        const float2 ya = *q0, yb = *(q0 + w);
        a.x *= act_deriv_y(ya.x, act, slope); a.y *= act_deriv_y(ya.y, act, slope);
        b.x *= act_deriv_y(yb.x, act, slope); b.y *= act_deriv_y(yb.y, act, slope);
    }
    y[i] = scale * ((a.x + a.y) + (b.x + b.y));
}

// y[r][2h+a][2w+b] = scale * x[r][h][w] * act'(yref[r][2h+a][2w+b]): nearest upsample / avg-pool backward, optionally
// fused with the backward of the activation that fed the pool (critic stem: conv -> LeakyReLU -> AvgPool2d,
// discriminator.py:106-109), which saves one full-resolution read+write pass.
__global__ __launch_bounds__(256) void pool2_up_kernel(const float* __restrict__ x, const float* __restrict__ yref,
                                                       float* __restrict__ y, long long total_in, int h, int w,
                                                       float scale, int act, float slope)
{
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total_in) return;
    const int ix = (int)(i % w);
    const long long t = i / w;
    const int iy = (int)(t % h);
    const long long r = t / h;
    const float v = scale * x[i];
    const long long off = (r * (2 * h) + 2 * iy) * (2 * (long long)w);
    float2 a = make_float2(v, v), b = a;
    if (yref) {
        const float2* q0 = reinterpret_cast<const float2*>(yref + off) + ix;
        const float2 ya = *q0, yb = *(q0 + w);
        a.x *= act_deriv_y(ya.x, act, slope); a.y *= act_deriv_y(ya.y, act, slope);
        b.x *= act_deriv_y(yb.x, act, slope); b.y *= act_deriv_y(yb.y, act, slope);
    }
    float2* p0 = reinterpret_cast<float2*>(y + off) + ix;
    *p0 = a;
    *(p0 + w) = b;
}

// Two coarse pixels per thread (w even, 16-byte aligned tensors): every fine-resolution access is a float4
// `partials` (optional): sum of the workgroup's outputs, one float per workgroup (= [row][h*w/512] when a plane is a whole
// number of workgroups) -- the per-row sums of the result (a conv bias gradient) without reading it back.
// `bits` (instead of yref): the sign bits the fused stem convolution wrote (conv_igemm.hip, quad pixel order): the pair of
// coarse pixels of thread i owns byte i of the bit array -- bit 4c+2a+b is fine pixel (2h+a, 2w+b) of coarse pixel c.
__global__ __launch_bounds__(256) void pool2_up_v2_kernel(const float* __restrict__ x, const float* __restrict__ yref,
                                                          float* __restrict__ y, long long total_pairs, int h, int w2,
                                                          float scale, int act, float slope, float* __restrict__ partials,
                                                          const unsigned char* __restrict__ bits)
{
    __shared__ float sh[4];
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total_pairs) return;                       // (never taken with partials: the host checks total_pairs % 256 == 0)
    const int ix = (int)(i % w2);                       // pair index inside the coarse row
    const long long t = i / w2;                         // coarse row index (r * h + iy)
    const float2 xv = reinterpret_cast<const float2*>(x)[i];
    const float v0 = scale * xv.x, v1 = scale * xv.y;
    const long long off = t * (8 * (long long)w2) + 4 * ix;      // fine row 2*iy of plane r: (t * 2) * (4 * w2)
    float4 a = make_float4(v0, v0, v1, v1), b = a;
    if (yref) {
        const float4 ya = *reinterpret_cast<const float4*>(yref + off);
        const float4 yb = *reinterpret_cast<const float4*>(yref + off + 4 * w2);
        a.x *= act_deriv_y(ya.x, act, slope); a.y *= act_deriv_y(ya.y, act, slope);
        a.z *= act_deriv_y(ya.z, act, slope); a.w *= act_deriv_y(ya.w, act, slope);
        b.x *= act_deriv_y(yb.x, act, slope); b.y *= act_deriv_y(yb.y, act, slope);
        b.z *= act_deriv_y(yb.z, act, slope); b.w *= act_deriv_y(yb.w, act, slope);
    } else if (bits) {
        const unsigned m = bits[i];
        const float lo = (act == SEMPYR_ACT_LRELU) ? slope : 0.f;
        a.x *= (m & 1u) ? 1.f : lo;  a.y *= (m & 2u) ? 1.f : lo;  b.x *= (m & 4u) ? 1.f : lo;  b.y *= (m & 8u) ? 1.f : lo;
        a.z *= (m & 16u) ? 1.f : lo; a.w *= (m & 32u) ? 1.f : lo; b.z *= (m & 64u) ? 1.f : lo; b.w *= (m & 128u) ? 1.f : lo;
    }
    *reinterpret_cast<float4*>(y + off) = a;
    *reinterpret_cast<float4*>(y + off + 4 * w2) = b;
    if (partials) {
        const float s = block_sum_256(((a.x + a.y) + (a.z + a.w)) + ((b.x + b.y) + (b.z + b.w)), sh);
        if (threadIdx.x == 0) partials[blockIdx.x] = s;
    }
}

__global__ __launch_bounds__(256) void pool2_down_v2_kernel(const float* __restrict__ x, const float* __restrict__ yref,
                                                            float* __restrict__ y, long long total_pairs, int h, int w2,
                                                            float scale, int act, float slope,
                                                            const unsigned char* __restrict__ bits)
{
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total_pairs) return;
    const int ox = (int)(i % w2);
    const long long t = i / w2;
    const long long off = t * (8 * (long long)w2) + 4 * ox;
    float4 a = *reinterpret_cast<const float4*>(x + off);
    float4 b = *reinterpret_cast<const float4*>(x + off + 4 * w2);
    if (yref) {
        const float4 ya = *reinterpret_cast<const float4*>(yref + off);
        const float4 yb = *reinterpret_cast<const float4*>(yref + off + 4 * w2);
        a.x *= act_deriv_y(ya.x, act, slope); a.y *= act_deriv_y(ya.y, act, slope);
        a.z *= act_deriv_y(ya.z, act, slope); a.w *= act_deriv_y(ya.w, act, slope);
        b.x *= act_deriv_y(yb.x, act, slope); b.y *= act_deriv_y(yb.y, act, slope);
        b.z *= act_deriv_y(yb.z, act, slope); b.w *= act_deriv_y(yb.w, act, slope);
    } else if (bits) {                                  // sign bits of the fused stem convolution (see pool2_up_v2_kernel)
        const unsigned m = bits[i];
        const float lo = (act == SEMPYR_ACT_LRELU) ? slope : 0.f;
        a.x *= (m & 1u) ? 1.f : lo;  a.y *= (m & 2u) ? 1.f : lo;  b.x *= (m & 4u) ? 1.f : lo;  b.y *= (m & 8u) ? 1.f : lo;
        a.z *= (m & 16u) ? 1.f : lo; a.w *= (m & 32u) ? 1.f : lo; b.z *= (m & 64u) ? 1.f : lo; b.w *= (m & 128u) ? 1.f : lo;
    }
    reinterpret_cast<float2*>(y)[i] = make_float2(scale * ((a.x + a.y) + (b.x + b.y)), scale * ((a.z + a.w) + (b.z + b.w)));
}

// 3x3 stride-2 pad-1 max pool; idx = flat offset (iy*W+ix) of the first maximum in scan order
__global__ __launch_bounds__(256) void maxpool3s2_fwd_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                             int* __restrict__ idx, long long total_out,
                                                             int H, int W, int Ho, int Wo)
{
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total_out) return;
    const int ox = (int)(i % Wo);
    const long long t = i / Wo;
    const int oy = (int)(t % Ho);
    const long long r = t / Ho;
    const float* xr = x + r * H * W;
    float best = -INFINITY; int bi = -1;
    for (int a = 0; a < 3; ++a) {
        const int iy = 2 * oy - 1 + a;
        if (iy < 0 || iy >= H) continue;
        for (int b = 0; b < 3; ++b) {
            const int ix = 2 * ox - 1 + b;
            if (ix < 0 || ix >= W) continue;
            const float v = xr[iy * W + ix];
            if (bi < 0 || v > best || v != v) { best = v; bi = iy * W + ix; }
        }
    }
    y[i] = best; idx[i] = bi;
}

// gather form of the max-pool backward (deterministic: no atomics)
__global__ __launch_bounds__(256) void maxpool3s2_bwd_kernel(const float* __restrict__ gy, const int* __restrict__ idx,
                                                             float* __restrict__ gx, long long total_in,
                                                             int H, int W, int Ho, int Wo)
{
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total_in) return;
    const int ix = (int)(i % W);
    const long long t = i / W;
    const int iy = (int)(t % H);
    const long long r = t / H;
    const int self = iy * W + ix;
    const int oy0 = iy >> 1, oy1 = (iy + 1) >> 1;      // windows containing row iy: oy in {iy/2, (iy+1)/2}
    const int ox0 = ix >> 1, ox1 = (ix + 1) >> 1;
    float g = 0.f;
    const float* gr = gy + r * Ho * Wo; const int* ir = idx + r * Ho * Wo;
    for (int oy = oy0; oy <= oy1; ++oy) {
        if (oy >= Ho) continue;
        for (int ox = ox0; ox <= ox1; ++ox) {
            if (ox >= Wo) continue;
            if (ir[oy * Wo + ox] == self) g += gr[oy * Wo + ox];
        }
    }
    gx[i] = g;
}

// ---- small device-resident parameter blocks (inputs of captured HIP graphs) -------------------------------------------
// up to STORE_MAX 32-bit words travel in the kernel arguments and are written to device memory: a stream-ordered host ->
// device hand-over of a few scalars without a pinned staging buffer (nothing to keep alive, nothing to reuse too early)
constexpr int STORE_MAX = SEMPYR_STORE_MAX_WORDS;
struct StoreArgs { uint32_t w[STORE_MAX]; int n; };
__global__ __launch_bounds__(64) void store_words_kernel(const StoreArgs a, uint32_t* __restrict__ dst)
{
    const int i = threadIdx.x;
    if (i < a.n) dst[i] = a.w[i];
}

// the six feature masks of train_model.py:172-209 in one launch: mask t is `base` everywhere except the window
// rows [x0,x1) x cols [y0,y1) which hold `win`; the six-word spec of every mask is read from device memory
struct MaskArgs {
    float* ptr[SEMPYR_MASKS_MAX];
    long long first[SEMPYR_MASKS_MAX + 1];        // first thread of each mask
    int H[SEMPYR_MASKS_MAX], W[SEMPYR_MASKS_MAX], vec[SEMPYR_MASKS_MAX];
    int n;
};
__global__ __launch_bounds__(256) void build_masks_kernel(const MaskArgs a, const int32_t* __restrict__ spec)
{
    const long long tid = (long long)blockIdx.x * 256 + threadIdx.x;
    if (tid >= a.first[a.n]) return;
    int t = 0;
#pragma unroll
    for (int k = 1; k < SEMPYR_MASKS_MAX; ++k) if (k < a.n && tid >= a.first[k]) t = k;
    const int32_t* sp = spec + 6 * t;
    const float base = __int_as_float(sp[0]), win = __int_as_float(sp[1]);
    const int x0 = sp[2], x1 = sp[3], y0 = sp[4], y1 = sp[5];
    const int H = a.H[t], W = a.W[t];
    const long long local = tid - a.first[t];
    if (a.vec[t]) {
        const long long e = local * 4;
        const int wq = (int)(e % W);
        const int h = (int)((e / W) % H);
        const bool rowin = h >= x0 && h < x1;
        float4 v;
        v.x = (rowin && wq + 0 >= y0 && wq + 0 < y1) ? win : base;
        v.y = (rowin && wq + 1 >= y0 && wq + 1 < y1) ? win : base;
        v.z = (rowin && wq + 2 >= y0 && wq + 2 < y1) ? win : base;
        v.w = (rowin && wq + 3 >= y0 && wq + 3 < y1) ? win : base;
        *reinterpret_cast<float4*>(a.ptr[t] + e) = v;
    } else {
        const int wq = (int)(local % W);
        const int h = (int)((local / W) % H);
        a.ptr[t][local] = (h >= x0 && h < x1 && wq >= y0 && wq < y1) ? win : base;
    }
}

// ---- glue that replaces torch's own tiny kernels inside the loop body -------------------------------------------------
struct ConcatArgs { const float* src[SEMPYR_CONCAT_MAX]; long long first[SEMPYR_CONCAT_MAX + 1]; int n; };
// dst = [src0 | src1 | ...] (float4 granules; every part's length and start are multiples of 4 floats)
__global__ __launch_bounds__(256) void concat_kernel(const ConcatArgs a, float* __restrict__ dst)
{
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;          // float4 index in dst
    if (i >= a.first[a.n]) return;
    int t = 0;
#pragma unroll
    for (int k = 1; k < SEMPYR_CONCAT_MAX; ++k) if (k < a.n && i >= a.first[k]) t = k;
    reinterpret_cast<float4*>(dst)[i] = reinterpret_cast<const float4*>(a.src[t])[i - a.first[t]];
}
__global__ __launch_bounds__(256) void concat_scalar_kernel(const ConcatArgs a, float* __restrict__ dst)
{
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= a.first[a.n]) return;
    int t = 0;
#pragma unroll
    for (int k = 1; k < SEMPYR_CONCAT_MAX; ++k) if (k < a.n && i >= a.first[k]) t = k;
    dst[i] = a.src[t][i - a.first[t]];
}

__global__ __launch_bounds__(256) void fill_const_kernel(float* __restrict__ dst, float v, long long n)
{
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i < n) dst[i] = v;
}

// *out = ((w0*x0 + w1*x1 + ...) * s1) * s2 on device scalars, every product and sum rounded separately and summed left to
// right (what the chain of torch scalar ops it replaces computes)
struct LincombArgs { const float* x[SEMPYR_LINCOMB_MAX]; float w[SEMPYR_LINCOMB_MAX]; int n; float s1, s2; };
__global__ void lincomb_kernel(const LincombArgs a, float* __restrict__ out)
{
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    float acc = __fmul_rn(a.w[0], *a.x[0]);
    for (int i = 1; i < a.n; ++i) acc = __fadd_rn(acc, __fmul_rn(a.w[i], *a.x[i]));
    acc = __fmul_rn(acc, a.s1);
    acc = __fmul_rn(acc, a.s2);
    *out = acc;
}

// dst_i (+)= src_i for a table of small tensors (parameter-gradient accumulation of a whole backward pass in one launch);
// a workgroup owns one 1024-element chunk of one pair
struct AccumArgs { float* dst[SEMPYR_ACCUM_MAX]; const float* src[SEMPYR_ACCUM_MAX][SEMPYR_ACCUM_SRCS]; int numel[SEMPYR_ACCUM_MAX];
                   int first_chunk[SEMPYR_ACCUM_MAX + 1]; unsigned long long assign_mask; int n; };
__global__ __launch_bounds__(256) void accumulate_many_kernel(const AccumArgs a)
{
    const int chunk = blockIdx.x;
    int lo = 0, hi = a.n - 1;
    while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (a.first_chunk[mid] <= chunk) lo = mid; else hi = mid - 1; }
    const int t = lo;
    const bool assign = (a.assign_mask >> t) & 1ull;
    float* __restrict__ d = a.dst[t];
    const int base = (chunk - a.first_chunk[t]) * 1024;
    for (int i = base + threadIdx.x; i < a.numel[t] && i < base + 1024; i += 256) {
        float v = assign ? a.src[t][0][i] : d[i] + a.src[t][0][i];          // the sources are added in table order
#pragma unroll
        for (int k = 1; k < SEMPYR_ACCUM_SRCS; ++k) if (a.src[t][k]) v += a.src[t][k][i];
        d[i] = v;
    }
}

// y = a + act(beta[c]) and the running-statistics update of a training-mode BatchNorm2d whose input is all zeros (the
// feature branch of a generator level whose mask is all zero, generator.py:126-129 with quirk Q2): batch mean and variance
// are exactly 0, so BN(0) = beta and the buffers decay by (1 - momentum) -- exactly what the statistics pass over the zero
// tensor, the apply pass and the add would produce, in one launch.
__global__ __launch_bounds__(256) void add_channel_const_kernel(const float* __restrict__ a, const float* __restrict__ beta,
                                                                float* __restrict__ y, long long total, int len, int C, int vec,
                                                                int act, float slope,
                                                                float* running_mean, float* running_var, long long* nbt, float momentum)
{
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (blockIdx.x == 0 && running_mean) {
        for (int c = threadIdx.x; c < C; c += 256) {
            running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * 0.f;
            running_var[c] = (1.f - momentum) * running_var[c] + momentum * 0.f;
        }
        if (threadIdx.x == 0 && nbt) *nbt += 1;
    }
    if (vec) {
        if (i * 4 >= total) return;
        const int row = (int)((i * 4) / len);
        const float k = act_apply(beta[row % C], act, slope);
        float4 v = reinterpret_cast<const float4*>(a)[i];
        v.x += k; v.y += k; v.z += k; v.w += k;
        reinterpret_cast<float4*>(y)[i] = v;
    } else {
        if (i >= total) return;
        const int row = (int)(i / len);
        y[i] = a[i] + act_apply(beta[row % C], act, slope);
    }
}

} // namespace

#define LAUNCH_1D(kernel, nthreads, st, ...) do { \
    hipLaunchKernelGGL(kernel, dim3(ceil_div((nthreads), 256)), dim3(256), 0, (st), __VA_ARGS__); \
    SEMPYR_CHECK_LAUNCH(); } while (0)

static int affine_act_launch(const float* x, const float* sub_row, const float* scale_row, const float* shift_row,
                             const float* res, const float* rsub_row, const float* rscale_row, const float* rshift_row,
                             float* y, int rows, int len, int act, float slope, void* stream, int res_w = 0)
{
    if (!x || !y) SEMPYR_FAIL("affine_act: null pointer");
    if (res_w < 0 || (res_w > 0 && (!res || (res_w & 1) || len % res_w != 0 || ((len / res_w) & 1))))
        SEMPYR_FAIL("affine_act: a half-resolution residual needs even height and width (len %d, width %d)", len, res_w);
    if ((scale_row == nullptr) != (shift_row == nullptr)) SEMPYR_FAIL("affine_act: scale and shift must both be given");
    if (sub_row && !scale_row) SEMPYR_FAIL("affine_act: sub_row needs scale/shift");
    if ((rscale_row == nullptr) != (rshift_row == nullptr)) SEMPYR_FAIL("affine_act: residual scale and shift must both be given");
    if ((rscale_row || rsub_row) && (!res || !rscale_row)) SEMPYR_FAIL("affine_act: residual coefficients need res, scale and shift");
    if (rows < 1 || len < 1) SEMPYR_FAIL("affine_act: bad dims");
    if (act < 0 || act > 3) SEMPYR_FAIL("affine_act: bad activation %d", act);
    const long long total = (long long)rows * len;
    const int vec = (len % 4 == 0) && aligned16(x) && aligned16(y) &&
                    (!res || (res_w > 0 ? (res_w % 4 == 0 && (((uintptr_t)res) & 7) == 0) : aligned16(res)));
    HbmScope hs("affine_act_kernel", 4.0 * total * (2.0 + (res ? (res_w > 0 ? 0.25 : 1.0) : 0.0)), (hipStream_t)stream);
    LAUNCH_1D(affine_act_kernel, vec ? total / 4 : total, (hipStream_t)stream, x, sub_row, scale_row, shift_row, res,
              rsub_row, rscale_row, rshift_row, y, total, len, vec, act, slope, res_w);
    return 0;
}

extern "C" int sempyr_affine_act(const float* x, const float* sub_row, const float* scale_row, const float* shift_row,
                                 const float* res, float* y, int rows, int len, int act, float slope, void* stream)
{
    return affine_act_launch(x, sub_row, scale_row, shift_row, res, nullptr, nullptr, nullptr, y, rows, len, act, slope, stream);
}

extern "C" int sempyr_affine2_act(const float* x, const float* sub_row, const float* scale_row, const float* shift_row,
                                  const float* res, const float* rsub_row, const float* rscale_row, const float* rshift_row,
                                  float* y, int rows, int len, int act, float slope, int res_half_w, void* stream)
{
    if (!res || !rscale_row) SEMPYR_FAIL("affine2_act: the residual and its coefficients are required");
    return affine_act_launch(x, sub_row, scale_row, shift_row, res, rsub_row, rscale_row, rshift_row, y, rows, len, act, slope,
                             stream, res_half_w);
}

extern "C" int sempyr_affine_act_bwd(const float* gy, const float* y, const float* scale_row, float* gx, float* gres,
                                     int rows, int len, int act, float slope, void* stream)
{
    if (!gy || (!gx && !gres)) SEMPYR_FAIL("affine_act_bwd: null pointer");
    if (act != SEMPYR_ACT_NONE && !y) SEMPYR_FAIL("affine_act_bwd: activation needs y");
    if (rows < 1 || len < 1) SEMPYR_FAIL("affine_act_bwd: bad dims");
    if (act < 0 || act > 3) SEMPYR_FAIL("affine_act_bwd: bad activation %d", act);
    const long long total = (long long)rows * len;
    const int vec = (len % 4 == 0) && aligned16(gy) && (!y || aligned16(y)) && (!gx || aligned16(gx)) && (!gres || aligned16(gres));
    HbmScope hs("affine_act_bwd_kernel", 4.0 * total * (1.0 + (y ? 1 : 0) + (gx ? 1 : 0) + (gres ? 1 : 0)), (hipStream_t)stream);
    LAUNCH_1D(affine_act_bwd_kernel, vec ? total / 4 : total, (hipStream_t)stream, gy, y, scale_row, gx, gres, total, len, vec, act, slope);
    return 0;
}

extern "C" int sempyr_bn_zero_input_add(const float* a, const float* beta, float* y, int N, int C, int HW, int act, float slope,
                                        float* running_mean, float* running_var, int64_t* num_batches_tracked, float momentum,
                                        void* stream)
{
    if (!a || !beta || !y || N < 1 || C < 1 || HW < 1) SEMPYR_FAIL("bn_zero_input_add: bad arguments");
    if (act < SEMPYR_ACT_NONE || act > SEMPYR_ACT_LRELU) SEMPYR_FAIL("bn_zero_input_add: activation must be none / ReLU / LeakyReLU");
    if ((running_mean == nullptr) != (running_var == nullptr)) SEMPYR_FAIL("bn_zero_input_add: running_mean and running_var come together");
    const long long total = (long long)N * C * HW;
    const int vec = (HW % 4 == 0) && aligned16(a) && aligned16(y);
    LAUNCH_1D(add_channel_const_kernel, vec ? total / 4 : total, (hipStream_t)stream, a, beta, y, total, HW, C, vec, act, slope,
              running_mean, running_var, (long long*)num_batches_tracked, momentum);
    return 0;
}

extern "C" int sempyr_mul(const float* a, const float* b, float* y, int64_t n, void* stream)
{
    if (!a || !b || !y || n < 1) SEMPYR_FAIL("mul: bad arguments");
    const int vec = (n % 4 == 0) && aligned16(a) && aligned16(b) && aligned16(y);
    HbmScope hs("mul_kernel", 12.0 * n, (hipStream_t)stream);
    LAUNCH_1D(mul_kernel, vec ? n / 4 : n, (hipStream_t)stream, a, b, y, (long long)n, vec);
    return 0;
}

extern "C" int sempyr_axpby(const float* a, const float* b, float alpha, float beta, float* y, int64_t n, void* stream)
{
    if (!a || !y || n < 1) SEMPYR_FAIL("axpby: bad arguments");
    const int vec = (n % 4 == 0) && aligned16(a) && (!b || aligned16(b)) && aligned16(y);
    LAUNCH_1D(axpby_kernel, vec ? n / 4 : n, (hipStream_t)stream, a, b, alpha, beta, y, (long long)n, vec);
    return 0;
}

extern "C" int sempyr_lerp_rows(const float* real, const float* fake, const float* alpha_row, float* out,
                                int rows, int len, void* stream)
{
    if (!real || !fake || !alpha_row || !out || rows < 1 || len < 1) SEMPYR_FAIL("lerp_rows: bad arguments");
    const long long total = (long long)rows * len;
    const int vec = (len % 4 == 0) && aligned16(real) && aligned16(fake) && aligned16(out);
    HbmScope hs("lerp_rows_kernel", 12.0 * total, (hipStream_t)stream);
    LAUNCH_1D(lerp_rows_kernel, vec ? total / 4 : total, (hipStream_t)stream, real, fake, alpha_row, out, total, len, vec);
    return 0;
}

extern "C" int sempyr_normalize(const float* x, const float* mean_c, const float* std_c, float* y,
                                int N, int C, int HW, int backward, void* stream)
{
    if (!x || !std_c || !y || (!backward && !mean_c) || N < 1 || C < 1 || HW < 1) SEMPYR_FAIL("normalize: bad arguments");
    const long long total = (long long)N * C * HW;
    const int vec = (HW % 4 == 0) && aligned16(x) && aligned16(y);
    HbmScope hs("normalize_kernel", 8.0 * total, (hipStream_t)stream);
    LAUNCH_1D(normalize_kernel, vec ? total / 4 : total, (hipStream_t)stream, x, mean_c, std_c, y, total, C, HW, vec, backward ? 1 : 0);
    return 0;
}

extern "C" int sempyr_pool2_down(const float* x, const float* yref, const uint32_t* act_bits, float* y, int rows, int H, int W,
                                 float scale, int act, float slope, void* stream)
{
    if (act_bits && (yref || !(act == SEMPYR_ACT_RELU || act == SEMPYR_ACT_LRELU) || (W & 3) || ((long long)H * W) % 32 != 0))
        SEMPYR_FAIL("pool2_down: sign bits replace yref, for ReLU / LeakyReLU planes of a multiple of 32 elements and W %% 4 == 0");
    if (!x || !y || rows < 1 || H < 2 || W < 2 || (H & 1) || (W & 1)) SEMPYR_FAIL("pool2_down: bad arguments (H=%d W=%d)", H, W);
    if ((((uintptr_t)x | (uintptr_t)yref) & 7) != 0) SEMPYR_FAIL("pool2_down: inputs must be 8-byte aligned");
    if (act < 0 || act > 3) SEMPYR_FAIL("pool2_down: bad activation %d", act);
    const long long total = (long long)rows * (H / 2) * (W / 2);
    if ((W & 3) == 0 && ((((uintptr_t)x | (uintptr_t)yref) & 15) == 0) && (((uintptr_t)y & 7) == 0)) {
        HbmScope hs("pool2_down_v2_kernel", 4.0 * total * (5.0 + (yref ? 4.0 : act_bits ? 0.125 : 0.0)), (hipStream_t)stream);
        LAUNCH_1D(pool2_down_v2_kernel, total / 2, (hipStream_t)stream, x, yref, y, total / 2, H / 2, W / 4, scale, act, slope,
                  (const unsigned char*)act_bits);
        return 0;
    }
    if (act_bits) SEMPYR_FAIL("pool2_down: sign bits need 16-byte aligned tensors");
    LAUNCH_1D(pool2_down_kernel, total, (hipStream_t)stream, x, yref, y, total, H / 2, W / 2, scale, act, slope);
    return 0;
}

extern "C" int sempyr_pool2_up(const float* x, const float* yref, const uint32_t* act_bits, float* y, int rows, int h, int w,
                               float scale, int act, float slope, float* row_partials, void* stream)
{
    if (act_bits && (yref || !(act == SEMPYR_ACT_RELU || act == SEMPYR_ACT_LRELU) || (w & 1) || ((long long)h * w) % 8 != 0))
        SEMPYR_FAIL("pool2_up: sign bits replace yref, for ReLU / LeakyReLU planes of a multiple of 32 fine elements and even w");
    if (!x || !y || rows < 1 || h < 1 || w < 1) SEMPYR_FAIL("pool2_up: bad arguments");
    if ((((uintptr_t)y | (uintptr_t)yref) & 7) != 0) SEMPYR_FAIL("pool2_up: output / yref must be 8-byte aligned");
    if (act < 0 || act > 3) SEMPYR_FAIL("pool2_up: bad activation %d", act);
    const long long total = (long long)rows * h * w;
    if ((w & 1) == 0 && ((((uintptr_t)y | (uintptr_t)yref) & 15) == 0) && (((uintptr_t)x & 7) == 0)) {
        if (row_partials && ((long long)h * w) % 512 != 0) SEMPYR_FAIL("pool2_up: row partials need h*w to be a multiple of 512");
        LAUNCH_1D(pool2_up_v2_kernel, total / 2, (hipStream_t)stream, x, yref, y, total / 2, h, w / 2, scale, act, slope, row_partials,
                  (const unsigned char*)act_bits);
        return 0;
    }
    if (act_bits) SEMPYR_FAIL("pool2_up: sign bits need 16-byte aligned tensors");
    if (row_partials) SEMPYR_FAIL("pool2_up: row partials need an even width and 16-byte aligned tensors");
    LAUNCH_1D(pool2_up_kernel, total, (hipStream_t)stream, x, yref, y, total, h, w, scale, act, slope);
    return 0;
}

extern "C" int sempyr_maxpool3s2_fwd(const float* x, float* y, int32_t* idx, int rows, int H, int W, void* stream)
{
    if (!x || !y || !idx || rows < 1 || H < 1 || W < 1) SEMPYR_FAIL("maxpool3s2_fwd: bad arguments");
    const int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1;
    const long long total = (long long)rows * Ho * Wo;
    HbmScope hs("maxpool3s2_fwd_kernel", 4.0 * ((double)rows * H * W + 2.0 * total), (hipStream_t)stream);
    LAUNCH_1D(maxpool3s2_fwd_kernel, total, (hipStream_t)stream, x, y, idx, total, H, W, Ho, Wo);
    return 0;
}

extern "C" int sempyr_maxpool3s2_bwd(const float* gy, const int32_t* idx, float* gx, int rows, int H, int W, void* stream)
{
    if (!gy || !gx || !idx || rows < 1 || H < 1 || W < 1) SEMPYR_FAIL("maxpool3s2_bwd: bad arguments");
    const int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1;
    const long long total = (long long)rows * H * W;
    HbmScope hs("maxpool3s2_bwd_kernel", 4.0 * ((double)total + 2.0 * rows * Ho * Wo), (hipStream_t)stream);
    LAUNCH_1D(maxpool3s2_bwd_kernel, total, (hipStream_t)stream, gy, idx, gx, total, H, W, Ho, Wo);
    return 0;
}

extern "C" int sempyr_store_words(const uint32_t* words_host, int n, uint32_t* dst_dev, void* stream)
{
    if (!words_host || !dst_dev || n < 1 || n > STORE_MAX) SEMPYR_FAIL("store_words: 1..%d words expected (got %d)", STORE_MAX, n);
    StoreArgs a;
    for (int i = 0; i < n; ++i) a.w[i] = words_host[i];
    a.n = n;
    hipLaunchKernelGGL(store_words_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, a, dst_dev);
    SEMPYR_CHECK_LAUNCH();
    return 0;
}

extern "C" int sempyr_build_masks(const sempyr_mask_target* targets, int n, const int32_t* spec_dev, void* stream)
{
    if (!targets || !spec_dev || n < 1 || n > SEMPYR_MASKS_MAX) SEMPYR_FAIL("build_masks: 1..%d masks expected (got %d)", SEMPYR_MASKS_MAX, n);
    MaskArgs a;
    long long threads = 0;
    for (int i = 0; i < n; ++i) {
        const sempyr_mask_target& t = targets[i];
        if (!t.mask || t.rows < 1 || t.H < 1 || t.W < 1) SEMPYR_FAIL("build_masks: mask %d has a null pointer or an empty shape", i);
        const long long numel = (long long)t.rows * t.H * t.W;
        if (numel >= (1ll << 31)) SEMPYR_FAIL("build_masks: mask %d has %lld elements", i, numel);
        a.ptr[i] = t.mask; a.H[i] = t.H; a.W[i] = t.W;
        a.vec[i] = (t.W % 4 == 0) && aligned16(t.mask);
        a.first[i] = threads;
        threads += a.vec[i] ? numel / 4 : numel;
    }
    a.first[n] = threads; a.n = n;
    for (int i = n; i < SEMPYR_MASKS_MAX; ++i) { a.ptr[i] = nullptr; a.H[i] = a.W[i] = 1; a.vec[i] = 0; a.first[i + 1] = threads; }
    LAUNCH_1D(build_masks_kernel, threads, (hipStream_t)stream, a, spec_dev);
    return 0;
}

extern "C" int sempyr_concat(const float* const* srcs_host, const int64_t* numels_host, int n, float* dst, void* stream)
{
    if (!srcs_host || !numels_host || !dst || n < 1 || n > SEMPYR_CONCAT_MAX) SEMPYR_FAIL("concat: 1..%d sources expected (got %d)", SEMPYR_CONCAT_MAX, n);
    bool vec = aligned16(dst);
    for (int i = 0; i < n; ++i) {
        if (!srcs_host[i] || numels_host[i] < 1) SEMPYR_FAIL("concat: source %d is empty", i);
        vec = vec && aligned16(srcs_host[i]) && (numels_host[i] % 4 == 0);
    }
    ConcatArgs a;
    long long off = 0;
    for (int i = 0; i < n; ++i) { a.src[i] = srcs_host[i]; a.first[i] = off; off += vec ? numels_host[i] / 4 : numels_host[i]; }
    for (int i = n; i < SEMPYR_CONCAT_MAX; ++i) { a.src[i] = nullptr; a.first[i + 1] = off; }
    a.first[n] = off; a.n = n;
    if (vec) LAUNCH_1D(concat_kernel, off, (hipStream_t)stream, a, dst);
    else LAUNCH_1D(concat_scalar_kernel, off, (hipStream_t)stream, a, dst);
    return 0;
}

extern "C" int sempyr_fill_const(float* dst, float value, int64_t n, void* stream)
{
    if (!dst || n < 1) SEMPYR_FAIL("fill_const: bad arguments");
    LAUNCH_1D(fill_const_kernel, n, (hipStream_t)stream, dst, value, (long long)n);
    return 0;
}

extern "C" int sempyr_lincomb(const float* const* xs_host, const float* ws_host, int n, float s1, float s2, float* out, void* stream)
{
    if (!xs_host || !ws_host || !out || n < 1 || n > SEMPYR_LINCOMB_MAX) SEMPYR_FAIL("lincomb: 1..%d terms expected (got %d)", SEMPYR_LINCOMB_MAX, n);
    LincombArgs a;
    for (int i = 0; i < n; ++i) { if (!xs_host[i]) SEMPYR_FAIL("lincomb: term %d is null", i); a.x[i] = xs_host[i]; a.w[i] = ws_host[i]; }
    for (int i = n; i < SEMPYR_LINCOMB_MAX; ++i) { a.x[i] = nullptr; a.w[i] = 0.f; }
    a.n = n; a.s1 = s1; a.s2 = s2;
    hipLaunchKernelGGL(lincomb_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, a, out);
    SEMPYR_CHECK_LAUNCH();
    return 0;
}

extern "C" int sempyr_accumulate_many(const sempyr_accum_item* items_host, int n, void* stream)
{
    if (!items_host || n < 1) SEMPYR_FAIL("accumulate_many: empty table");
    for (int first = 0; first < n; first += SEMPYR_ACCUM_MAX) {
        const int m = n - first < SEMPYR_ACCUM_MAX ? n - first : SEMPYR_ACCUM_MAX;
        AccumArgs a;
        a.assign_mask = 0ull;
        long long chunks = 0;
        for (int i = 0; i < m; ++i) {
            const sempyr_accum_item& it = items_host[first + i];
            if (!it.dst || !it.src[0] || it.numel < 1 || it.numel >= (1ll << 31)) SEMPYR_FAIL("accumulate_many: bad item %d", first + i);
            for (int j = 0; j < first + i; ++j)
                if (items_host[j].dst == it.dst && j >= first) SEMPYR_FAIL("accumulate_many: destination of item %d repeats in one launch", first + i);
            a.dst[i] = it.dst; a.numel[i] = (int)it.numel; a.first_chunk[i] = (int)chunks;
            for (int k = 0; k < SEMPYR_ACCUM_SRCS; ++k) a.src[i][k] = it.src[k];
            if (it.assign) a.assign_mask |= 1ull << i;
            chunks += (it.numel + 1023) / 1024;
        }
        if (chunks >= (1ll << 31)) SEMPYR_FAIL("accumulate_many: too many elements");
        a.first_chunk[m] = (int)chunks; a.n = m;
        hipLaunchKernelGGL(accumulate_many_kernel, dim3((unsigned)chunks), dim3(256), 0, (hipStream_t)stream, a);
        SEMPYR_CHECK_LAUNCH();
    }
    return 0;
}
