// Multi-tensor Adam step (optim.Adam(lr, betas=(.5,.999)) of train_model.py:273-274 and its
// .step() calls at :117,143).  All parameter tensors of a module are updated by ONE launch: the tensor table travels in the
// kernel arguments (no device-side table to upload), a workgroup owns one 4096-element chunk of one tensor and finds its
// tensor by a binary search over the per-tensor first-chunk index.  HBM-bound: 4 reads + 3 writes of 4 B per element.
#include "common.h"
#include <math.h>

namespace {

constexpr int ADAM_MAX = SEMPYR_ADAM_MAX_TENSORS;     // tensors per launch (4 pointers each: 2 KB of kernel arguments)
constexpr int ADAM_CHUNK = 4096;                       // elements per workgroup: 256 threads x 4 x float4

struct AdamArgs {
    float* p[ADAM_MAX];
    float* g[ADAM_MAX];
    float* m[ADAM_MAX];
    float* v[ADAM_MAX];
    int first_chunk[ADAM_MAX + 1];
    int numel[ADAM_MAX];
    int n;
    float w1;            // 1 - beta1 (the lerp weight)
    float beta2, w2;     // beta2, 1 - beta2
    float eps, neg_step_size, bc2_sqrt, grad_scale;
    int zero_grads;
    const float* dyn;    // optional: {neg_step_size, bc2_sqrt} in device memory (graph replay)
};

// the element update, in the operation order of torch's foreach Adam (lerp_, mul_ + addcmul_, sqrt / div / add, addcdiv_)
__device__ __forceinline__ void adam_elem(float& p, float& g, float& m, float& v, const AdamArgs& a, float neg_step_size,
                                          float bc2_sqrt)
{
    const float gs = g * a.grad_scale;
    const float d = gs - m;
    m = (a.w1 < 0.5f) ? m + a.w1 * d : gs - d * (1.f - a.w1);
    v = v * a.beta2 + (a.w2 * gs) * gs;
    const float denom = sqrtf(v) / bc2_sqrt + a.eps;
    p = p + neg_step_size * (m / denom);
}

__global__ __launch_bounds__(256) void adam_kernel(const AdamArgs a)
{
    const float nss = a.dyn ? a.dyn[0] : a.neg_step_size, bcs = a.dyn ? a.dyn[1] : a.bc2_sqrt;
    const int chunk = blockIdx.x;
    int lo = 0, hi = a.n - 1;                       // last tensor whose first chunk <= chunk (workgroup-uniform)
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (a.first_chunk[mid] <= chunk) lo = mid; else hi = mid - 1;
    }
    const int t = lo;
    const int n = a.numel[t];
    const int base = (chunk - a.first_chunk[t]) * ADAM_CHUNK;
    float* __restrict__ P = a.p[t]; float* __restrict__ G = a.g[t];
    float* __restrict__ M = a.m[t]; float* __restrict__ V = a.v[t];
    const bool vec = ((((uintptr_t)P | (uintptr_t)G | (uintptr_t)M | (uintptr_t)V) & 15) == 0) && (base + ADAM_CHUNK <= n);
    if (vec) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int i = base + (r * 256 + threadIdx.x) * 4;
            float4 p = *reinterpret_cast<const float4*>(P + i), g = *reinterpret_cast<const float4*>(G + i);
            float4 m = *reinterpret_cast<const float4*>(M + i), v = *reinterpret_cast<const float4*>(V + i);
            adam_elem(p.x, g.x, m.x, v.x, a, nss, bcs); adam_elem(p.y, g.y, m.y, v.y, a, nss, bcs);
            adam_elem(p.z, g.z, m.z, v.z, a, nss, bcs); adam_elem(p.w, g.w, m.w, v.w, a, nss, bcs);
            *reinterpret_cast<float4*>(P + i) = p; *reinterpret_cast<float4*>(M + i) = m; *reinterpret_cast<float4*>(V + i) = v;
            if (a.zero_grads) *reinterpret_cast<float4*>(G + i) = make_float4(0.f, 0.f, 0.f, 0.f);
        }
    } else {
        for (int i = base + threadIdx.x; i < n && i < base + ADAM_CHUNK; i += 256) {
            float p = P[i], g = G[i], m = M[i], v = V[i];
            adam_elem(p, g, m, v, a, nss, bcs);
            P[i] = p; M[i] = m; V[i] = v;
            if (a.zero_grads) G[i] = 0.f;
        }
    }
}

}  // namespace

static int adam_launch(const sempyr_adam_tensor* tensors, int n_tensors, int64_t step, const float* dyn, double lr, double beta1,
                       double beta2, double eps, double grad_scale, int zero_grads, void* stream)
{
    if (!tensors || n_tensors < 1) SEMPYR_FAIL("adam: empty tensor list");
    if (!dyn && step < 1) SEMPYR_FAIL("adam: step counts from 1 (got %lld)", (long long)step);
    if (!(lr >= 0.0) || !(eps >= 0.0) || !(beta1 >= 0.0 && beta1 < 1.0) || !(beta2 >= 0.0 && beta2 < 1.0))
        SEMPYR_FAIL("adam: bad hyper-parameters (lr %g, betas %g %g, eps %g)", lr, beta1, beta2, eps);
    for (int i = 0; i < n_tensors; ++i) {
        const sempyr_adam_tensor& t = tensors[i];
        if (!t.param || !t.grad || !t.exp_avg || !t.exp_avg_sq) SEMPYR_FAIL("adam: tensor %d has a null pointer", i);
        if (t.numel < 1 || t.numel >= (1ll << 31)) SEMPYR_FAIL("adam: tensor %d has %lld elements", i, (long long)t.numel);
    }
    // scalar preparation as torch.optim.Adam does it (Python doubles, then one rounding to fp32 per scalar)
    const double bc1 = 1.0 - pow(beta1, (double)(dyn ? 1 : step)), bc2 = 1.0 - pow(beta2, (double)(dyn ? 1 : step));
    AdamArgs a;
    a.dyn = dyn;
    a.w1 = (float)(1.0 - beta1); a.beta2 = (float)beta2; a.w2 = (float)(1.0 - beta2);
    a.eps = (float)eps; a.neg_step_size = (float)(-(lr / bc1)); a.bc2_sqrt = (float)sqrt(bc2);
    a.grad_scale = (float)grad_scale; a.zero_grads = zero_grads ? 1 : 0;
    for (int first = 0; first < n_tensors; first += ADAM_MAX) {
        const int n = n_tensors - first < ADAM_MAX ? n_tensors - first : ADAM_MAX;
        long long chunks = 0;
        for (int i = 0; i < n; ++i) {
            const sempyr_adam_tensor& t = tensors[first + i];
            a.p[i] = t.param; a.g[i] = t.grad; a.m[i] = t.exp_avg; a.v[i] = t.exp_avg_sq;
            a.numel[i] = (int)t.numel; a.first_chunk[i] = (int)chunks;
            chunks += (t.numel + ADAM_CHUNK - 1) / ADAM_CHUNK;
        }
        if (chunks >= (1ll << 31)) SEMPYR_FAIL("adam: too many elements for one launch");
        a.first_chunk[n] = (int)chunks; a.n = n;
        double numel = 0.0;
        for (int i = 0; i < n; ++i) numel += (double)a.numel[i];
        HbmScope hs("adam_kernel", 28.0 * numel, (hipStream_t)stream);
        hipLaunchKernelGGL(adam_kernel, dim3((unsigned)chunks), dim3(256), 0, (hipStream_t)stream, a);
        SEMPYR_CHECK_LAUNCH();
    }
    return 0;
}

extern "C" int sempyr_adam_step(const sempyr_adam_tensor* tensors, int n_tensors, int64_t step, double lr, double beta1,
                                double beta2, double eps, double grad_scale, int zero_grads, void* stream)
{
    return adam_launch(tensors, n_tensors, step, nullptr, lr, beta1, beta2, eps, grad_scale, zero_grads, stream);
}

extern "C" int sempyr_adam_scalars(int64_t step, double lr, double beta1, double beta2, float* out2_host)
{
    if (!out2_host || step < 1) SEMPYR_FAIL("adam_scalars: bad arguments");
    const double bc1 = 1.0 - pow(beta1, (double)step), bc2 = 1.0 - pow(beta2, (double)step);
    out2_host[0] = (float)(-(lr / bc1));
    out2_host[1] = (float)sqrt(bc2);
    return 0;
}

extern "C" int sempyr_adam_step_dev(const sempyr_adam_tensor* tensors, int n_tensors, const float* step_scalars_dev, double beta1,
                                    double beta2, double eps, double grad_scale, int zero_grads, void* stream)
{
    if (!step_scalars_dev) SEMPYR_FAIL("adam_step_dev: the device scalars are required");
    return adam_launch(tensors, n_tensors, 1, step_scalars_dev, 1.0, beta1, beta2, eps, grad_scale, zero_grads, stream);
}
