// Shared helpers for the sempyr HIP kernels (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include "../../include/sempyr.h"

// Error bookkeeping: every entry point returns 0 on success, <0 for an invalid
// descriptor / unsupported shape (never a silent fallback), >0 for a hipError_t.
void sempyr_set_error(const char* fmt, ...);

#define SEMPYR_FAIL(...) do { sempyr_set_error(__VA_ARGS__); return SEMPYR_ERR_INVALID; } while (0)
#define SEMPYR_CHECK_LAUNCH() do { hipError_t e_ = hipGetLastError(); \
    if (e_ != hipSuccess) { sempyr_set_error("%s:%d launch failed: %s", __FILE__, __LINE__, hipGetErrorString(e_)); return (int)e_; } } while (0)

static inline int ceil_div(long long a, long long b) { return (int)((a + b - 1) / b); }

// Measurement only (bench.py's roofline leg, never enabled by the product path): while sempyr_profile_begin/end bracket a
// region, an HbmScope around the launch of an HBM-bound kernel records a HIP event pair on the launch stream and the launch's
// ALGORITHMIC bytes (4 B x elements read once + written once).  sempyr_profile_hbm_report lists, per kernel name,
// launches, bytes and summed duration.  Off: one relaxed load and a branch.
struct HbmScope {
    int slot;
    hipStream_t st;
    HbmScope(const char* name, double bytes, hipStream_t stream);
    ~HbmScope();
};
void sempyr_hbm_profile_begin(int max_launches);
void sempyr_hbm_profile_end();

// wave64 sum reduction (all lanes get the result)
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// block (256 threads = 4 waves) sum; result valid in every thread. `sh` needs >= 4 floats.
__device__ __forceinline__ float block_sum_256(float v, float* sh) {
    v = wave_sum(v);
    const int w = threadIdx.x >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sh[w] = v;
    __syncthreads();
    return (sh[0] + sh[1]) + (sh[2] + sh[3]);
}

// activation of the fused epilogues / elementwise passes and its derivative expressed through the OUTPUT y
__device__ __forceinline__ float act_apply(float v, int act, float slope) {
    switch (act) {
        case SEMPYR_ACT_RELU: return v > 0.f ? v : 0.f;
        case SEMPYR_ACT_LRELU: return v > 0.f ? v : v * slope;
        case SEMPYR_ACT_TANH: return tanhf(v);
        default: return v;
    }
}
__device__ __forceinline__ float act_deriv_y(float y, int act, float slope) {
    switch (act) {
        case SEMPYR_ACT_RELU: return y > 0.f ? 1.f : 0.f;
        case SEMPYR_ACT_LRELU: return y > 0.f ? 1.f : slope;
        case SEMPYR_ACT_TANH: return 1.f - y * y;
        default: return 1.f;
    }
}
