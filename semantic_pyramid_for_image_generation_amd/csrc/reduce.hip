// Reductions of the loss path (deterministic two-stage sums, wavefront shuffle reductions):
// WGAN means (train_model.py:39-46), masked L1 feature loss (train_model.py:212-222 with nn.L1Loss,
// :270), diversity L1s (:109), per-sample L2 norm + (n-1)^2 penalty (:58-66), bias gradients.
#include "common.h"
#include <mutex>

namespace {

constexpr int MAX_PARTS = 1024;

inline bool aligned16(const void* p) { return ((uintptr_t)p & 15) == 0; }

// stage 1: partial[b] = sum over a grid-stride slice of f(i)
// OP 0: x ; OP 1: |a*m - b*m| (m optional) ; OP 2: x*x
template <int OP>
__global__ __launch_bounds__(256) void partial_sum_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                          const float* __restrict__ m, long long n, int vec,
                                                          float* __restrict__ partial)
{
    __shared__ float sh[4];
    float s = 0.f;
    const long long stride = (long long)gridDim.x * 256;
    if (vec) {
        const long long n4 = n / 4;
        for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride) {
            const float4 u = reinterpret_cast<const float4*>(a)[i];
            if (OP == 0) s += (u.x + u.y) + (u.z + u.w);
            else if (OP == 2) s += (u.x * u.x + u.y * u.y) + (u.z * u.z + u.w * u.w);
            else {
                const float4 v = reinterpret_cast<const float4*>(b)[i];
                if (m) {
                    const float4 k = reinterpret_cast<const float4*>(m)[i];
                    s += (fabsf(u.x * k.x - v.x * k.x) + fabsf(u.y * k.y - v.y * k.y)) +
                         (fabsf(u.z * k.z - v.z * k.z) + fabsf(u.w * k.w - v.w * k.w));
                } else {
                    s += (fabsf(u.x - v.x) + fabsf(u.y - v.y)) + (fabsf(u.z - v.z) + fabsf(u.w - v.w));
                }
            }
        }
    } else {
        for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
            if (OP == 0) s += a[i];
            else if (OP == 2) s += a[i] * a[i];
            else s += m ? fabsf(a[i] * m[i] - b[i] * m[i]) : fabsf(a[i] - b[i]);
        }
    }
    s = block_sum_256(s, sh);
    if (threadIdx.x == 0) partial[blockIdx.x] = s;
}

// stage 2: out = scale * sum partial[0..np)
__global__ __launch_bounds__(256) void final_sum_kernel(const float* __restrict__ partial, int np, float scale,
                                                        float* __restrict__ out)
{
    __shared__ float sh[4];
    float s = 0.f;
    for (int i = threadIdx.x; i < np; i += 256) s += partial[i];
    s = block_sum_256(s, sh);
    if (threadIdx.x == 0) *out = s * scale;
}

// y[i] = scale * (*g)
__global__ __launch_bounds__(256) void fill_scalar_kernel(const float* __restrict__ g, float scale, float* __restrict__ y, long long n)
{
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i < n) y[i] = scale * (*g);
}

// d = sign(a*m - b*m) * m * (*g) * scale ; ga = d ; gb = -d
__global__ __launch_bounds__(256) void l1_bwd_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                     const float* __restrict__ m, const float* __restrict__ g, float scale,
                                                     float* __restrict__ ga, float* __restrict__ gb, long long n)
{
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float k = m ? m[i] : 1.f;
    const float d = a[i] * k - b[i] * k;
    const float sg = d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f);
    const float v = sg * k * (*g) * scale;
    if (ga) ga[i] = v;
    if (gb) gb[i] = -v;
}

// per-row sum of squares -> L2 norm
template <int TPR>
__global__ __launch_bounds__(256) void row_l2norm_kernel(const float* __restrict__ x, int rows, int len, int vec,
                                                         float* __restrict__ norms, int take_sqrt)
{
    __shared__ float sh[4];
    constexpr int RPB = 256 / TPR;
    const int row = blockIdx.x * RPB + threadIdx.x / TPR;
    const int t = threadIdx.x % TPR;
    const bool live = row < rows;
    const float* xr = x + (size_t)(live ? row : 0) * len;
    float s = 0.f;
    if (live) {
        if (vec) {
            const float4* x4 = reinterpret_cast<const float4*>(xr);
            for (int i = t; i < len / 4; i += TPR) { const float4 v = x4[i]; s += (v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w); }
        } else {
            for (int i = t; i < len; i += TPR) s += xr[i] * xr[i];
        }
    }
    if (TPR == 256) s = block_sum_256(s, sh); else s = wave_sum(s);
    if (live && t == 0) norms[row] = take_sqrt ? sqrtf(s) : s;
}

// gp = mean_r (norm_r - 1)^2.  split > 1: `parts` holds `split` partial sums of squares per row (a 196 608-element row per
// sample would otherwise keep only `rows` = 16 workgroups busy: 54 us); the norms are finished here, in a fixed order.
__global__ __launch_bounds__(256) void gp_final_kernel(float* __restrict__ norms, const float* __restrict__ parts, int split,
                                                       int rows, float* __restrict__ gp)
{
    __shared__ float sh[4];
    float s = 0.f;
    for (int i = threadIdx.x; i < rows; i += 256) {
        float nr;
        if (split > 1) {
            float a = 0.f;
            for (int j = 0; j < split; ++j) a += parts[(size_t)i * split + j];
            nr = sqrtf(a);
            norms[i] = nr;
        } else {
            nr = norms[i];
        }
        const float d = nr - 1.f; s += d * d;
    }
    s = block_sum_256(s, sh);
    if (threadIdx.x == 0) *gp = s / (float)rows;
}

// out = grad * [ (*g) * (2/rows) * (norm_r - 1) / norm_r ]
__global__ __launch_bounds__(256) void gp_bwd_kernel(const float* __restrict__ grad, const float* __restrict__ norms,
                                                     const float* __restrict__ g, int rows, long long total, int len, int vec,
                                                     float* __restrict__ out)
{
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    const float gs = (*g) * 2.f / (float)rows;
    if (vec) {
        if (i * 4 >= total) return;
        const float nr = norms[(int)((i * 4) / len)];
        const float c = nr > 0.f ? gs * (nr - 1.f) / nr : 0.f;
        float4 v = reinterpret_cast<const float4*>(grad)[i];
        v.x *= c; v.y *= c; v.z *= c; v.w *= c;
        reinterpret_cast<float4*>(out)[i] = v;
    } else {
        if (i >= total) return;
        const float nr = norms[(int)(i / len)];
        out[i] = grad[i] * (nr > 0.f ? gs * (nr - 1.f) / nr : 0.f);
    }
}

// row sums (for bias gradients): rs[row] = sum x[row][:]
template <int TPR>
__global__ __launch_bounds__(256) void row_sum_kernel(const float* __restrict__ x, int rows, int len, int vec, float* __restrict__ rs)
{
    __shared__ float sh[4];
    constexpr int RPB = 256 / TPR;
    const int row = blockIdx.x * RPB + threadIdx.x / TPR;
    const int t = threadIdx.x % TPR;
    const bool live = row < rows;
    const float* xr = x + (size_t)(live ? row : 0) * len;
    float s = 0.f;
    if (live) {
        if (vec) {
            const float4* x4 = reinterpret_cast<const float4*>(xr);
            for (int i = t; i < len / 4; i += TPR) { const float4 v = x4[i]; s += (v.x + v.y) + (v.z + v.w); }
        } else {
            for (int i = t; i < len; i += TPR) s += xr[i];
        }
    }
    if (TPR == 256) s = block_sum_256(s, sh); else s = wave_sum(s);
    if (live && t == 0) rs[row] = s;
}

// out[c] = scale * sum_n rs[n*C + c].  16 lanes per channel (one thread per channel walking its N rows is a chain of N
// dependent L2 round trips: 7-10 us for 16-32 rows, 56 launches per cycle); fixed lane -> row map and shuffle tree: deterministic.
__global__ __launch_bounds__(256) void channel_fold_kernel(const float* __restrict__ rs, int N, int C, float scale,
                                                           float* __restrict__ out, int accumulate)
{
    const int t = blockIdx.x * 256 + threadIdx.x;
    const int c = t >> 4, l = t & 15;
    float a = 0.f;
    if (c < C)
        for (int n = l; n < N; n += 16) a += rs[n * C + c];
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) a += __shfl_xor(a, o, 64);
    if (c < C && l == 0) out[c] = accumulate ? out[c] + a * scale : a * scale;
}

template <int OP>
int two_stage(const float* a, const float* b, const float* m, long long n, float scale, float* out, float* scratch, hipStream_t st)
{
    const int vec = (n % 4 == 0) && aligned16(a) && (!b || aligned16(b)) && (!m || aligned16(m));
    long long work = vec ? n / 4 : n;
    int blocks = (int)((work + 256 * 8 - 1) / (256 * 8));
    if (blocks < 1) blocks = 1;
    if (blocks > MAX_PARTS) blocks = MAX_PARTS;
    hipLaunchKernelGGL(partial_sum_kernel<OP>, dim3(blocks), dim3(256), 0, st, a, b, m, n, vec, scratch);
    SEMPYR_CHECK_LAUNCH();
    hipLaunchKernelGGL(final_sum_kernel, dim3(1), dim3(256), 0, st, scratch, blocks, scale, out);
    SEMPYR_CHECK_LAUNCH();
    return 0;
}

} // namespace

extern "C" int sempyr_reduce_scratch_floats(void) { return MAX_PARTS; }

extern "C" int sempyr_mean(const float* x, int64_t n, float scale, float* out, float* scratch, void* stream)
{
    if (!x || !out || !scratch || n < 1) SEMPYR_FAIL("mean: bad arguments");
    return two_stage<0>(x, nullptr, nullptr, n, scale / (float)n, out, scratch, (hipStream_t)stream);
}

extern "C" int sempyr_fill_scalar(const float* g, float scale, float* y, int64_t n, void* stream)
{
    if (!g || !y || n < 1) SEMPYR_FAIL("fill_scalar: bad arguments");
    hipLaunchKernelGGL(fill_scalar_kernel, dim3(ceil_div(n, 256)), dim3(256), 0, (hipStream_t)stream, g, scale, y, (long long)n);
    SEMPYR_CHECK_LAUNCH();
    return 0;
}

extern "C" int sempyr_l1_mean(const float* a, const float* b, const float* mask, int64_t n, float* out,
                              float* scratch, void* stream)
{
    if (!a || !b || !out || !scratch || n < 1) SEMPYR_FAIL("l1_mean: bad arguments");
    return two_stage<1>(a, b, mask, n, 1.0f / (float)n, out, scratch, (hipStream_t)stream);
}

extern "C" int sempyr_l1_mean_bwd(const float* a, const float* b, const float* mask, const float* gout, int64_t n,
                                  float* ga, float* gb, void* stream)
{
    if (!a || !b || !gout || (!ga && !gb) || n < 1) SEMPYR_FAIL("l1_mean_bwd: bad arguments");
    hipLaunchKernelGGL(l1_bwd_kernel, dim3(ceil_div(n, 256)), dim3(256), 0, (hipStream_t)stream,
                       a, b, mask, gout, 1.0f / (float)n, ga, gb, (long long)n);
    SEMPYR_CHECK_LAUNCH();
    return 0;
}

extern "C" int sempyr_gp_fwd(const float* grad, int rows, int len, float* norms, float* gp, float* scratch, void* stream)
{
    if (!grad || !norms || !gp || rows < 1 || len < 1) SEMPYR_FAIL("gp_fwd: bad arguments");
    hipStream_t st = (hipStream_t)stream;
    const int vec = (len % 4 == 0) && aligned16(grad);
    int split = 1;
    if (scratch && rows < 256 && len % (SEMPYR_GP_SPLIT * 4) == 0 && len / SEMPYR_GP_SPLIT >= 2048) split = SEMPYR_GP_SPLIT;
    if (split > 1)          // `split` sub-rows per row, each reduced by its own workgroup
        hipLaunchKernelGGL(row_l2norm_kernel<256>, dim3(rows * split), dim3(256), 0, st, grad, rows * split, len / split, vec, scratch, 0);
    else if (len >= 2048) hipLaunchKernelGGL(row_l2norm_kernel<256>, dim3(rows), dim3(256), 0, st, grad, rows, len, vec, norms, 1);
    else hipLaunchKernelGGL(row_l2norm_kernel<64>, dim3(ceil_div(rows, 4)), dim3(256), 0, st, grad, rows, len, vec, norms, 1);
    SEMPYR_CHECK_LAUNCH();
    hipLaunchKernelGGL(gp_final_kernel, dim3(1), dim3(256), 0, st, norms, (const float*)scratch, split, rows, gp);
    SEMPYR_CHECK_LAUNCH();
    return 0;
}

// ---- the one entry point that allocates nothing and takes no workspace (include/sempyr.h: sempyr_gp_fwd_static) ----------
// The split partials live in a file-scope device array.  In host code `gp_static_scratch` is only a stand-in for the
// symbol: its device address is looked up once per device (hipGetSymbolAddress, with that device current) and from then on
// only that pointer is handed to kernels.  Every GPU holds its own copy of the array until the process ends.
static __device__ __align__(256) float gp_static_scratch[SEMPYR_GP_STATIC_MAX_ROWS * SEMPYR_GP_SPLIT];

namespace {
constexpr int GP_MAX_DEVICES = 64;
float* g_gp_scratch_ptr[GP_MAX_DEVICES];          // per device, set once (guarded by g_gp_scratch_lock)
std::mutex g_gp_scratch_lock;

int gp_static_scratch_ptr(float** out)
{
    int dev = -1;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) { sempyr_set_error("gp_fwd_static: hipGetDevice: %s", hipGetErrorString(e)); return (int)e; }
    if (dev < 0 || dev >= GP_MAX_DEVICES) SEMPYR_FAIL("gp_fwd_static: device index %d out of range", dev);
    std::lock_guard<std::mutex> lock(g_gp_scratch_lock);
    if (!g_gp_scratch_ptr[dev]) {
        float* p = nullptr;
        e = hipGetSymbolAddress((void**)&p, HIP_SYMBOL(gp_static_scratch));
        if (e != hipSuccess || !p) {
            sempyr_set_error("gp_fwd_static: hipGetSymbolAddress: %s", hipGetErrorString(e));
            return e != hipSuccess ? (int)e : SEMPYR_ERR_INVALID;
        }
        g_gp_scratch_ptr[dev] = p;
    }
    *out = g_gp_scratch_ptr[dev];
    return 0;
}
}  // namespace

extern "C" int sempyr_gp_fwd_static(const float* grad, int rows, int len, float* norms, float* gp, void* stream)
{
    if (!grad || !norms || !gp || rows < 1 || len < 1) SEMPYR_FAIL("gp_fwd_static: bad arguments");
    if (rows > SEMPYR_GP_STATIC_MAX_ROWS)
        SEMPYR_FAIL("gp_fwd_static: %d rows exceed the built-in scratch (SEMPYR_GP_STATIC_MAX_ROWS = %d); use sempyr_gp_fwd "
                    "with a caller-provided scratch buffer", rows, SEMPYR_GP_STATIC_MAX_ROWS);
    if (stream) {
        // the scratch array of the CURRENT device is used: a stream of another device would hand its kernels a foreign address
        hipDevice_t sdev = -1; int cur = -1;
        if (hipStreamGetDevice((hipStream_t)stream, &sdev) != hipSuccess || hipGetDevice(&cur) != hipSuccess) {
            (void)hipGetLastError();
            SEMPYR_FAIL("gp_fwd_static: cannot determine the device of the stream");
        }
        if ((int)sdev != cur)
            SEMPYR_FAIL("gp_fwd_static: the stream belongs to device %d but device %d is current; make the stream's device current",
                        (int)sdev, cur);
    }
    float* scratch = nullptr;
    const int rc = gp_static_scratch_ptr(&scratch);
    if (rc != 0) return rc;
    // every partial that gp_final_kernel reads is written by the row kernel of the same call: nothing has to start at zero
    return sempyr_gp_fwd(grad, rows, len, norms, gp, scratch, stream);
}

extern "C" int sempyr_gp_bwd(const float* grad, const float* norms, const float* gout, int rows, int len,
                             float* out, void* stream)
{
    if (!grad || !norms || !gout || !out || rows < 1 || len < 1) SEMPYR_FAIL("gp_bwd: bad arguments");
    const long long total = (long long)rows * len;
    const int vec = (len % 4 == 0) && aligned16(grad) && aligned16(out);
    hipLaunchKernelGGL(gp_bwd_kernel, dim3(ceil_div(vec ? total / 4 : total, 256)), dim3(256), 0, (hipStream_t)stream,
                       grad, norms, gout, rows, total, len, vec, out);
    SEMPYR_CHECK_LAUNCH();
    return 0;
}

extern "C" int sempyr_channel_sum(const float* x, int N, int C, int HW, float scale, float* out, int accumulate,
                                  float* scratch, void* stream)
{
    if (!x || !out || !scratch || N < 1 || C < 1 || HW < 1) SEMPYR_FAIL("channel_sum: bad arguments");
    hipStream_t st = (hipStream_t)stream;
    const int rows = N * C;
    const int vec = (HW % 4 == 0) && aligned16(x);
    {
        HbmScope hs(HW >= 2048 ? "row_sum_kernel<256>" : "row_sum_kernel<64>", 4.0 * rows * HW, st);
        if (HW >= 2048) hipLaunchKernelGGL(row_sum_kernel<256>, dim3(rows), dim3(256), 0, st, x, rows, HW, vec, scratch);
        else hipLaunchKernelGGL(row_sum_kernel<64>, dim3(ceil_div(rows, 4)), dim3(256), 0, st, x, rows, HW, vec, scratch);
    }
    SEMPYR_CHECK_LAUNCH();
    HbmScope hs_fold("channel_fold_kernel", 4.0 * (rows + C), st);
    hipLaunchKernelGGL(channel_fold_kernel, dim3(ceil_div((long long)C * 16, 256)), dim3(256), 0, st, scratch, N, C, scale, out, accumulate ? 1 : 0);
    SEMPYR_CHECK_LAUNCH();
    return 0;
}
