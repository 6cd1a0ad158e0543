// Normalisation kernels: BatchNorm2d (train), InstanceNorm2d(affine), GroupNorm(1, C) —
// forward, backward and backward-of-backward (the WGAN-GP double backward,
// train_model.py:69-76 through discriminator.py:16-46).
//
// All three norms are one computation over a tensor seen as rows = N*C, len = H*W:
//   xhat = (x - mean_g) * rstd_g ;  y = act(xhat * gamma_c + beta_c)
// and differ only in which rows form a statistics group g:
//   BATCH: rows with equal c   INSTANCE: the row itself   LAYER: rows with equal n
// So the heavy passes are kind-agnostic, HBM-bound, coalesced float4 row kernels
// (wavefront shuffle reductions), and the per-kind logic lives in tiny "finalize" kernels that
// turn per-row sums into per-row affine coefficients.
#include "common.h"
#include <stdlib.h>

namespace {

__device__ __forceinline__ float act_deriv(float y, int act, float slope) {
    // derivative of relu / leaky-relu expressed through the OUTPUT y (sign-preserving activations)
    if (act == SEMPYR_ACT_RELU) return y > 0.f ? 1.f : 0.f;
    if (act == SEMPYR_ACT_LRELU) return y > 0.f ? 1.f : slope;
    return 1.f;
}

// Reduce NS per-thread sums over the TPR threads that share a row. TPR = 64: one wave per row;
// TPR = 256: the whole block. Result valid in all participating threads.
template <int TPR, int NS>
__device__ __forceinline__ void row_reduce(float (&v)[NS], float* sh) {
    if constexpr (TPR < 64) {                  // several rows per wave: reduce inside each aligned group of TPR lanes
#pragma unroll
        for (int i = 0; i < NS; ++i)
#pragma unroll
            for (int o = TPR / 2; o > 0; o >>= 1) v[i] += __shfl_xor(v[i], o, 64);
        return;
    }
#pragma unroll
    for (int i = 0; i < NS; ++i) v[i] = wave_sum(v[i]);
    if (TPR == 256) {
        const int w = threadIdx.x >> 6;
        __syncthreads();
        if ((threadIdx.x & 63) == 0) {
#pragma unroll
            for (int i = 0; i < NS; ++i) sh[w * NS + i] = v[i];
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < NS; ++i) v[i] = (sh[i] + sh[NS + i]) + (sh[2 * NS + i] + sh[3 * NS + i]);
    }
}

// per-row mean and M2 = sum (x - mean)^2.
// Rows served by one wave (TPR = 64, < 2048 elements) are read twice (the second pass hits L1/L2).  Long rows
// (TPR = 256: up to 256 KB per row, 268 MB per tensor -- beyond L2 and the Infinity Cache) are read ONCE, accumulating
// sum(x - p) and sum (x - p)^2 around a pivot p = mean of the row's first four elements; with p inside the data's range
// M2 = S2 - S1^2/n loses about one bit, not the catastrophic cancellation of raw sum / sum-of-squares.
template <int TPR>
__global__ __launch_bounds__(256) void row_moments_kernel(const float* __restrict__ x, int rows, int len, int vec,
                                                          float* __restrict__ mean_out, float* __restrict__ m2_out)
{
    __shared__ float sh[8];
    constexpr int RPB = 256 / TPR;
    const int row = blockIdx.x * RPB + threadIdx.x / TPR;
    const int t = threadIdx.x % TPR;
    const bool live = row < rows;
    const float* xr = x + (size_t)(live ? row : 0) * len;
    if constexpr (TPR == 256) {
        const float p = len >= 4 ? 0.25f * ((xr[0] + xr[1]) + (xr[2] + xr[3])) : xr[0];
        float s[2] = {0.f, 0.f};
        if (live) {
            if (vec) {
                const float4* x4 = reinterpret_cast<const float4*>(xr);
                for (int i = t; i < len / 4; i += TPR) {
                    const float4 v = x4[i];
                    const float a = v.x - p, b = v.y - p, c = v.z - p, d = v.w - p;
                    s[0] += (a + b) + (c + d);
                    s[1] += (a * a + b * b) + (c * c + d * d);
                }
            } else {
                for (int i = t; i < len; i += TPR) { const float a = xr[i] - p; s[0] += a; s[1] += a * a; }
            }
        }
        row_reduce<TPR, 2>(s, sh);
        if (live && t == 0) {
            const float dm = s[0] / (float)len;
            mean_out[row] = p + dm;
            m2_out[row] = fmaxf(s[1] - s[0] * dm, 0.f);
        }
        return;
    }
    float s[1] = {0.f};
    if (live) {
        if (vec) {
            const float4* x4 = reinterpret_cast<const float4*>(xr);
            for (int i = t; i < len / 4; i += TPR) { const float4 v = x4[i]; s[0] += (v.x + v.y) + (v.z + v.w); }
        } else {
            for (int i = t; i < len; i += TPR) s[0] += xr[i];
        }
    }
    row_reduce<TPR, 1>(s, sh);
    const float mean = s[0] / (float)len;
    float q[1] = {0.f};
    if (live) {
        if (vec) {
            const float4* x4 = reinterpret_cast<const float4*>(xr);
            for (int i = t; i < len / 4; i += TPR) {
                const float4 v = x4[i];
                const float a = v.x - mean, b = v.y - mean, c = v.z - mean, d = v.w - mean;
                q[0] += (a * a + b * b) + (c * c + d * d);
            }
        } else {
            for (int i = t; i < len; i += TPR) { const float a = xr[i] - mean; q[0] += a * a; }
        }
    }
    row_reduce<TPR, 1>(q, sh);
    if (live && t == 0) { mean_out[row] = mean; m2_out[row] = q[0]; }
}

__device__ __forceinline__ void group_rows(int kind, int g, int N, int C, int& start, int& stride, int& count) {
    if (kind == SEMPYR_NORM_BATCH) { start = g; stride = C; count = N; }
    else if (kind == SEMPYR_NORM_INSTANCE) { start = g; stride = 1; count = 1; }
    else { start = g * C; stride = 1; count = C; }
}
__host__ __device__ __forceinline__ int num_groups(int kind, int N, int C) {
    return kind == SEMPYR_NORM_BATCH ? C : (kind == SEMPYR_NORM_INSTANCE ? N * C : N);
}

// 16 lanes cooperate on one statistics group (a thread per group walking its rows one after the other spent ~13 us
// in back-to-back dependent L2 round trips for a 16-sample BatchNorm).
constexpr int FIN_LPG = 16;
__device__ __forceinline__ float lpg_sum(float v) {
#pragma unroll
    for (int o = FIN_LPG / 2; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

__global__ __launch_bounds__(256) void norm_finalize_kernel(int kind, const float* __restrict__ row_mean,
                                     const float* __restrict__ row_m2,
                                     int N, int C, int len, float eps,
                                     const float* __restrict__ gamma, const float* __restrict__ beta,
                                     float* running_mean, float* running_var, long long* nbt, float momentum, float rep,
                                     float* __restrict__ mean_row, float* __restrict__ rstd_row,
                                     float* __restrict__ scale_row, float* __restrict__ shift_row)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    const int g = t / FIN_LPG, l = t % FIN_LPG;
    const bool live = g < num_groups(kind, N, C);        // uniform per 16-lane group; dead groups still join the shuffles
    int start = 0, stride = 1, count = 0;
    if (live) group_rows(kind, g, N, C, start, stride, count);
    float msum = 0.f;
    for (int i = l; i < count; i += FIN_LPG) msum += row_mean[start + i * stride];
    const float mean = lpg_sum(msum) / (float)(count > 0 ? count : 1);
    float m2 = 0.f;
    for (int i = l; i < count; i += FIN_LPG) {
        const float d = row_mean[start + i * stride] - mean;
        m2 += row_m2[start + i * stride] + (float)len * d * d;
    }
    m2 = lpg_sum(m2);
    if (!live) return;
    const float m = (float)count * (float)len;
    const float var = m2 / m;
    const float rstd = 1.0f / sqrtf(var + eps);
    for (int i = l; i < count; i += FIN_LPG) {
        const int row = start + i * stride;
        const int c = row % C;
        const float sc = rstd * gamma[c];
        mean_row[row] = mean; rstd_row[row] = rstd;
        scale_row[row] = sc; shift_row[row] = beta[c];     // y = (x - mean_row) * scale_row + shift_row
    }
    if (l == 0 && kind == SEMPYR_NORM_BATCH && running_mean) {
        // rep > 1: the tensor stands for one in which every element occurs `rep` times (a nearest-upsampled map): mean and
        // biased variance are unchanged, the unbiased estimate counts rep * m elements
        const float unbiased = m * rep > 1.f ? (m2 * rep) / (m * rep - 1.f) : var;
        running_mean[g] = (1.f - momentum) * running_mean[g] + momentum * mean;
        running_var[g] = (1.f - momentum) * running_var[g] + momentum * unbiased;
        if (g == 0 && nbt) *nbt += 1;
    }
}

// BatchNorm statistics from the partial sums a convolution's epilogue left (sempyr_conv2d_fwd_stats): per (sample, channel)
// row G groups of `granule` pixels, each (sum, sum of squared deviations from the group's mean).  One workgroup per
// channel: mean of all N*G groups, then M2 = sum_g (M2_g + granule * (mean_g - mean)^2) (the pairwise-update formula, no
// cancellation), then exactly the coefficient step of norm_finalize_kernel.  Fixed thread -> group map and a fixed reduction
// tree: deterministic.
__global__ __launch_bounds__(256) void bn_tile_finalize_kernel(const float2* __restrict__ tiles, int N, int C, int G, int granule,
                                                               float eps, const float* __restrict__ gamma, const float* __restrict__ beta,
                                                               float* running_mean, float* running_var, long long* nbt,
                                                               float momentum, float rep,
                                                               float* __restrict__ mean_row, float* __restrict__ rstd_row,
                                                               float* __restrict__ scale_row, float* __restrict__ shift_row)
{
    __shared__ float sh[8];
    const int c = blockIdx.x, t = threadIdx.x;
    const int P = N * G;
    const float inv_gr = 1.f / (float)granule;
    // four independent loads in flight per thread and pass (a dependent chain of P / 256 L2 round trips took 27 us for the
    // 16384 groups of a 256x256 layer)
    auto at = [&](int i) -> const float2* { const int b = i / G; return tiles + ((size_t)b * C + c) * G + (i - b * G); };
    float s[1] = {0.f};
    {
        float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
        int i = t;
        for (; i + 768 < P; i += 1024) {
            const float a0 = at(i)->x, a1 = at(i + 256)->x, a2 = at(i + 512)->x, a3 = at(i + 768)->x;
            s0 += a0; s1 += a1; s2 += a2; s3 += a3;
        }
        for (; i < P; i += 256) s0 += at(i)->x;
        s[0] = (s0 + s1) + (s2 + s3);
    }
    row_reduce<256, 1>(s, sh);
    const float m = (float)P * (float)granule;
    const float mean = s[0] / m;
    __syncthreads();
    float q[1] = {0.f};
    {
        auto term = [&](float2 v) { const float d = v.x * inv_gr - mean; return v.y + (float)granule * d * d; };
        float q0 = 0.f, q1 = 0.f, q2 = 0.f, q3 = 0.f;
        int i = t;
        for (; i + 768 < P; i += 1024) {
            const float2 v0 = *at(i), v1 = *at(i + 256), v2 = *at(i + 512), v3 = *at(i + 768);
            q0 += term(v0); q1 += term(v1); q2 += term(v2); q3 += term(v3);
        }
        for (; i < P; i += 256) q0 += term(*at(i));
        q[0] = (q0 + q1) + (q2 + q3);
    }
    row_reduce<256, 1>(q, sh);
    const float m2 = q[0];
    const float var = m2 / m;
    const float rstd = 1.0f / sqrtf(var + eps);
    const float sc = rstd * gamma[c], bt = beta[c];
    for (int b = t; b < N; b += 256) {
        const int row = b * C + c;
        mean_row[row] = mean; rstd_row[row] = rstd; scale_row[row] = sc; shift_row[row] = bt;
    }
    if (t == 0 && running_mean) {
        const float unbiased = m * rep > 1.f ? (m2 * rep) / (m * rep - 1.f) : var;
        running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * mean;
        running_var[c] = (1.f - momentum) * running_var[c] + momentum * unbiased;
        if (c == 0 && nbt) *nbt += 1;
    }
}

__global__ void bn_eval_coeffs_kernel(const float* __restrict__ gamma, const float* __restrict__ beta,
                                      const float* __restrict__ rm, const float* __restrict__ rv, float eps,
                                      int N, int C, float* __restrict__ scale_row, float* __restrict__ shift_row)
{
    const int row = blockIdx.x * blockDim.x + threadIdx.x;
    if (row >= N * C) return;
    const int c = row % C;
    const float sc = gamma[c] / sqrtf(rv[c] + eps);
    scale_row[row] = sc; shift_row[row] = beta[c] - rm[c] * sc;
}

// Per-row sums for the backward, accumulated on PIVOT-SHIFTED data: with Kg = gy'[row][0],
//   out[0] = Kg, out[1] = sum (gy'-Kg), out[2] = sum xc, out[3] = sum (gy'-Kg)*xc      (gy' = gy*act'(y), xc = x-mean_g)
// In the WGAN-GP inner pass gy' has a mean far above its spread in some layers; summing the raw products in fp32
// and subtracting the mean terms afterwards loses 2-4 digits (measured 2e-2 on one critic layer). The finalize kernel
// undoes the shift in double precision.
template <int TPR>
__global__ __launch_bounds__(256) void norm_bwd_rowsums_kernel(const float* __restrict__ gy, const float* __restrict__ y,
                                                               const float* __restrict__ x,
                                                               const float* __restrict__ mean_row,
                                                               int rows, int len, int vec, int act, float slope,
                                                               float* __restrict__ out /* [4][rows] */)
{
    __shared__ float sh[12];
    constexpr int RPB = 256 / TPR;
    const int row = blockIdx.x * RPB + threadIdx.x / TPR;
    const int t = threadIdx.x % TPR;
    const bool live = row < rows;
    const size_t base = (size_t)(live ? row : 0) * len;
    const float mean = mean_row[live ? row : 0];
    float Kg = gy[base];
    if (act != SEMPYR_ACT_NONE) Kg *= act_deriv(y[base], act, slope);
    float s[3] = {0.f, 0.f, 0.f};
    if (live) {
        if (vec) {
            const float4* g4 = reinterpret_cast<const float4*>(gy + base);
            const float4* x4 = reinterpret_cast<const float4*>(x + base);
            const float4* y4 = reinterpret_cast<const float4*>(y + base);
            for (int i = t; i < len / 4; i += TPR) {
                float4 g = g4[i]; float4 xv = x4[i];
                if (act != SEMPYR_ACT_NONE) {
                    const float4 yv = y4[i];
                    g.x *= act_deriv(yv.x, act, slope); g.y *= act_deriv(yv.y, act, slope);
                    g.z *= act_deriv(yv.z, act, slope); g.w *= act_deriv(yv.w, act, slope);
                }
                g.x -= Kg; g.y -= Kg; g.z -= Kg; g.w -= Kg;
                xv.x -= mean; xv.y -= mean; xv.z -= mean; xv.w -= mean;
                s[0] += (g.x + g.y) + (g.z + g.w);
                s[1] += (xv.x + xv.y) + (xv.z + xv.w);
                s[2] += (g.x * xv.x + g.y * xv.y) + (g.z * xv.z + g.w * xv.w);
            }
        } else {
            for (int i = t; i < len; i += TPR) {
                float g = gy[base + i];
                if (act != SEMPYR_ACT_NONE) g *= act_deriv(y[base + i], act, slope);
                g -= Kg;
                const float xc = x[base + i] - mean;
                s[0] += g; s[1] += xc; s[2] += g * xc;
            }
        }
    }
    row_reduce<TPR, 3>(s, sh);
    if (live && t == 0) { out[row] = Kg; out[rows + row] = s[0]; out[2 * rows + row] = s[1]; out[3 * rows + row] = s[2]; }
}

// group sums -> per-row coefficients of gx = cA*gy' + cB*(x - mean) + cC ; channel sums -> ggamma, gbeta (double math)
__device__ __forceinline__ double lpg_sum_d(double v) {
#pragma unroll
    for (int o = FIN_LPG / 2; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

__global__ __launch_bounds__(256) void norm_bwd_finalize_kernel(int kind, const float* __restrict__ sums, int N, int C, int len,
                                         const float* __restrict__ gamma, const float* __restrict__ rstd_row,
                                         float* __restrict__ cA, float* __restrict__ cB, float* __restrict__ cC,
                                         float* __restrict__ ggamma, float* __restrict__ gbeta)
{
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    const int ng = num_groups(kind, N, C);
    const int rows = N * C;
    auto G1 = [&](int row) { return (double)sums[row] * len + (double)sums[rows + row]; };                        // sum gy'
    auto G2 = [&](int row) { return ((double)sums[3 * rows + row] + (double)sums[row] * (double)sums[2 * rows + row])
                                    * (double)rstd_row[row]; };                                                 // sum gy'*xhat
    if (idx < ng * FIN_LPG) {                  // FIN_LPG lanes per group (a LAYER-norm group has up to 512 rows)
        const int g = idx / FIN_LPG, l = idx % FIN_LPG;
        int start, stride, count;
        group_rows(kind, g, N, C, start, stride, count);
        double SA = 0.0, SB = 0.0;
        for (int i = l; i < count; i += FIN_LPG) {
            const int row = start + i * stride;
            const double gm = gamma[row % C];
            SA += gm * G1(row); SB += gm * G2(row);
        }
        SA = lpg_sum_d(SA); SB = lpg_sum_d(SB);
        const double m = (double)count * (double)len;
        const double abar = SA / m, sbar = SB / m;
        for (int i = l; i < count; i += FIN_LPG) {
            const int row = start + i * stride;
            const double r = rstd_row[row];
            cA[row] = (float)(r * gamma[row % C]);
            cB[row] = (float)(-r * r * sbar);
            cC[row] = (float)(-r * abar);
        }
    } else if (idx < (ng + C) * FIN_LPG) {         // FIN_LPG lanes per channel as well (ng * FIN_LPG is a multiple of 16:
        const int c = idx / FIN_LPG - ng, l = idx % FIN_LPG;      // a 16-lane group never straddles the two branches)
        double a = 0.0, b = 0.0;
        for (int n = l; n < N; n += FIN_LPG) { a += G1(n * C + c); b += G2(n * C + c); }
        a = lpg_sum_d(a); b = lpg_sum_d(b);
        if (l == 0) {
            if (gbeta) gbeta[c] = (float)a;
            if (ggamma) ggamma[c] = (float)b;
        }
    }
}

// gx = cA[row] * gy * act'(y) + cB[row] * (x - mean[row]) + cC[row]   (centred form: no cancellation against the mean)
__global__ __launch_bounds__(256) void norm_bwd_apply_kernel(const float* __restrict__ gy, const float* __restrict__ y,
                                                             const float* __restrict__ x, const float* __restrict__ mean_row,
                                                             const float* __restrict__ cA, const float* __restrict__ cB,
                                                             const float* __restrict__ cC,
                                                             long long total, int len, int vec, int act, float slope,
                                                             float* __restrict__ gx)
{
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (vec) {
        if (i * 4 >= total) return;
        const int row = (int)((i * 4) / len);
        const float a = cA[row], b = cB[row], c = cC[row], mu = mean_row[row];
        float4 g = reinterpret_cast<const float4*>(gy)[i];
        float4 xv = reinterpret_cast<const float4*>(x)[i];
        xv.x -= mu; xv.y -= mu; xv.z -= mu; xv.w -= mu;
        if (act != SEMPYR_ACT_NONE) {
            const float4 yv = reinterpret_cast<const float4*>(y)[i];
            g.x *= act_deriv(yv.x, act, slope); g.y *= act_deriv(yv.y, act, slope);
            g.z *= act_deriv(yv.z, act, slope); g.w *= act_deriv(yv.w, act, slope);
        }
        float4 o;
        o.x = a * g.x + b * xv.x + c; o.y = a * g.y + b * xv.y + c;
        o.z = a * g.z + b * xv.z + c; o.w = a * g.w + b * xv.w + c;
        reinterpret_cast<float4*>(gx)[i] = o;
    } else {
        if (i >= total) return;
        const int row = (int)(i / len);
        float g = gy[i];
        if (act != SEMPYR_ACT_NONE) g *= act_deriv(y[i], act, slope);
        gx[i] = cA[row] * g + cB[row] * (x[i] - mean_row[row]) + cC[row];
    }
}

// ---- backward of the backward ----
// Pivot-shifted per-row sums (Ku = u[row][0], Kg = gy'[row][0]); out layout [9][rows]:
//   Ku, Kg, sum(u-Ku), sum(gy'-Kg), sum xc, sum(u-Ku)*xc, sum(gy'-Kg)*xc, sum(u-Ku)*(gy'-Kg), (unused)
template <int TPR>
__global__ __launch_bounds__(256) void norm_bwdbwd_rowsums_kernel(const float* __restrict__ u, const float* __restrict__ gy,
                                                                  const float* __restrict__ y, const float* __restrict__ x,
                                                                  const float* __restrict__ mean_row,
                                                                  int rows, int len, int vec, int act, float slope,
                                                                  float* __restrict__ sums)
{
    __shared__ float sh[24];
    constexpr int RPB = 256 / TPR;
    const int row = blockIdx.x * RPB + threadIdx.x / TPR;
    const int t = threadIdx.x % TPR;
    const bool live = row < rows;
    const size_t base = (size_t)(live ? row : 0) * len;
    const float mean = mean_row[live ? row : 0];
    const float Ku = u[base];
    float Kg = gy[base];
    if (act != SEMPYR_ACT_NONE) Kg *= act_deriv(y[base], act, slope);
    float s[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (live) {
        if (vec) {
            const float4* u4 = reinterpret_cast<const float4*>(u + base);
            const float4* g4 = reinterpret_cast<const float4*>(gy + base);
            const float4* x4 = reinterpret_cast<const float4*>(x + base);
            const float4* y4 = reinterpret_cast<const float4*>(y + base);
            for (int i = t; i < len / 4; i += TPR) {
                float4 uv = u4[i]; float4 g = g4[i]; float4 xv = x4[i];
                if (act != SEMPYR_ACT_NONE) {
                    const float4 yv = y4[i];
                    g.x *= act_deriv(yv.x, act, slope); g.y *= act_deriv(yv.y, act, slope);
                    g.z *= act_deriv(yv.z, act, slope); g.w *= act_deriv(yv.w, act, slope);
                }
                uv.x -= Ku; uv.y -= Ku; uv.z -= Ku; uv.w -= Ku;
                g.x -= Kg; g.y -= Kg; g.z -= Kg; g.w -= Kg;
                xv.x -= mean; xv.y -= mean; xv.z -= mean; xv.w -= mean;
                s[0] += (uv.x + uv.y) + (uv.z + uv.w);
                s[1] += (g.x + g.y) + (g.z + g.w);
                s[2] += (xv.x + xv.y) + (xv.z + xv.w);
                s[3] += (uv.x * xv.x + uv.y * xv.y) + (uv.z * xv.z + uv.w * xv.w);
                s[4] += (g.x * xv.x + g.y * xv.y) + (g.z * xv.z + g.w * xv.w);
                s[5] += (uv.x * g.x + uv.y * g.y) + (uv.z * g.z + uv.w * g.w);
            }
        } else {
            for (int i = t; i < len; i += TPR) {
                const float uv = u[base + i] - Ku; float g = gy[base + i]; const float xc = x[base + i] - mean;
                if (act != SEMPYR_ACT_NONE) g *= act_deriv(y[base + i], act, slope);
                g -= Kg;
                s[0] += uv; s[1] += g; s[2] += xc; s[3] += uv * xc; s[4] += g * xc; s[5] += uv * g;
            }
        }
    }
    row_reduce<TPR, 6>(s, sh);
    if (live && t == 0) {
        sums[row] = Ku; sums[rows + row] = Kg;
#pragma unroll
        for (int i = 0; i < 6; ++i) sums[(2 + i) * rows + row] = s[i];
    }
}

// un-shifted row sums in double: r1 = sum u, r2 = sum u*xhat, r3 = sum gy', r4 = sum gy'*xhat, r5 = sum u*gy'
struct RowSums { double r1, r2, r3, r4, r5; };
__device__ __forceinline__ RowSums unshift(const float* __restrict__ sums, int rows, int row, int len, double rstd) {
    const double Ku = sums[row], Kg = sums[rows + row];
    const double Su = sums[2 * rows + row], Sg = sums[3 * rows + row], Sx = sums[4 * rows + row];
    const double Sux = sums[5 * rows + row], Sgx = sums[6 * rows + row], Sug = sums[7 * rows + row];
    RowSums o;
    o.r1 = Ku * len + Su;
    o.r3 = Kg * len + Sg;
    o.r2 = (Sux + Ku * Sx) * rstd;
    o.r4 = (Sgx + Kg * Sx) * rstd;
    o.r5 = Sug + Ku * Sg + Kg * Su + Ku * Kg * len;
    return o;
}

// coef layout [10][rows]: P1 P2 P3 Q1 Q2 Q3 Q4 ubar t (float) ; the group's ubar and t are also kept as double split into
// (hi, lo) floats -- hi in slots 7/8, lo in sums slot 8 and coef slot 9 -- for the gamma-gradient kernel (double math; see
// the derivation in DESIGN.md).  FIN_LPG lanes per group.
__global__ __launch_bounds__(256) void norm_bwdbwd_finalize_kernel(int kind, float* __restrict__ sums, int N, int C, int len,
                                            const float* __restrict__ gamma, const float* __restrict__ rstd_row,
                                            float* __restrict__ coef)
{
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    const int g = idx / FIN_LPG, l = idx % FIN_LPG;
    const bool live = g < num_groups(kind, N, C);
    const int rows = N * C;
    int start = 0, stride = 1, count = 0;
    if (live) group_rows(kind, g, N, C, start, stride, count);
    double U = 0, T = 0, SA = 0, SB = 0, UA = 0;
    for (int i = l; i < count; i += FIN_LPG) {
        const int row = start + i * stride;
        const double gm = gamma[row % C];
        const RowSums q = unshift(sums, rows, row, len, rstd_row[row]);
        U += q.r1; T += q.r2; SA += gm * q.r3; SB += gm * q.r4; UA += gm * q.r5;
    }
    U = lpg_sum_d(U); T = lpg_sum_d(T); SA = lpg_sum_d(SA); SB = lpg_sum_d(SB); UA = lpg_sum_d(UA);
    if (!live) return;
    const double m = (double)count * (double)len;
    const double ub = U / m, t = T / m, ab = SA / m, s = SB / m, mua = UA / m;
    const double w = mua - ub * ab - 3.0 * s * t;
    for (int i = l; i < count; i += FIN_LPG) {
        const int row = start + i * stride;
        const double gm = gamma[row % C];
        const double r = rstd_row[row];
        const double r2 = r * r, r3 = r2 * r;
        coef[row] = (float)(gm * r);                               // P1
        coef[rows + row] = (float)(-gm * r2 * t);                  // P2   (multiplies x - mean)
        coef[2 * rows + row] = (float)(-gm * r * ub);              // P3
        coef[3 * rows + row] = (float)(-r2 * s);                   // Q1
        coef[4 * rows + row] = (float)(-r2 * t * gm);              // Q2
        coef[5 * rows + row] = (float)(-r3 * w);                   // Q3   (multiplies x - mean)
        coef[6 * rows + row] = (float)(r2 * (s * ub + t * ab));    // Q4
        coef[7 * rows + row] = (float)ub;
        coef[8 * rows + row] = (float)t;
        sums[8 * rows + row] = (float)(ub - (double)(float)ub);
        coef[9 * rows + row] = (float)(t - (double)(float)t);
    }
}

// g_gamma[c] = sum_n rstd * (sum u*gy' - ubar*sum gy' - t*sum gy'*xhat), with the group's ubar and t from the finalize
// kernel (double math)
__global__ void norm_bwdbwd_ggamma_kernel(int kind, const float* __restrict__ sums, const float* __restrict__ coef,
                                          const float* __restrict__ rstd_row, int N, int C, int len,
                                          float* __restrict__ ggamma)
{
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const int rows = N * C;
    double a = 0.0;
    for (int n = 0; n < N; ++n) {
        const int row = n * C + c;
        const double ub = (double)coef[7 * rows + row] + (double)sums[8 * rows + row];
        const double t = (double)coef[8 * rows + row] + (double)coef[9 * rows + row];
        const RowSums q = unshift(sums, rows, row, len, rstd_row[row]);
        a += (double)rstd_row[row] * (q.r5 - ub * q.r3 - t * q.r4);
    }
    ggamma[c] = (float)a;
}

// g_gy = act'(y) * (P1*u + P2*x + P3) ;  g_x = Q1*u + Q2*gy*act'(y) + Q3*x + Q4
__global__ __launch_bounds__(256) void norm_bwdbwd_apply_kernel(const float* __restrict__ u, const float* __restrict__ gy,
                                                                const float* __restrict__ y, const float* __restrict__ x,
                                                                const float* __restrict__ mean_row,
                                                                const float* __restrict__ coef, int rows,
                                                                long long total, int len, int vec, int act, float slope,
                                                                float* __restrict__ g_gy, float* __restrict__ g_x)
{
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (vec) {
        if (i * 4 >= total) return;
        const int row = (int)((i * 4) / len);
        const float P1 = coef[row], P2 = coef[rows + row], P3 = coef[2 * rows + row];
        const float Q1 = coef[3 * rows + row], Q2 = coef[4 * rows + row], Q3 = coef[5 * rows + row], Q4 = coef[6 * rows + row];
        const float4 uv = reinterpret_cast<const float4*>(u)[i];
        const float4 gv = reinterpret_cast<const float4*>(gy)[i];
        float4 xv = reinterpret_cast<const float4*>(x)[i];
        const float mu = mean_row[row];
        xv.x -= mu; xv.y -= mu; xv.z -= mu; xv.w -= mu;
        float4 d = make_float4(1.f, 1.f, 1.f, 1.f);
        if (act != SEMPYR_ACT_NONE) {
            const float4 yv = reinterpret_cast<const float4*>(y)[i];
            d.x = act_deriv(yv.x, act, slope); d.y = act_deriv(yv.y, act, slope);
            d.z = act_deriv(yv.z, act, slope); d.w = act_deriv(yv.w, act, slope);
        }
        float4 a, b;
        a.x = d.x * (P1 * uv.x + P2 * xv.x + P3); a.y = d.y * (P1 * uv.y + P2 * xv.y + P3);
        a.z = d.z * (P1 * uv.z + P2 * xv.z + P3); a.w = d.w * (P1 * uv.w + P2 * xv.w + P3);
        b.x = Q1 * uv.x + Q2 * gv.x * d.x + Q3 * xv.x + Q4; b.y = Q1 * uv.y + Q2 * gv.y * d.y + Q3 * xv.y + Q4;
        b.z = Q1 * uv.z + Q2 * gv.z * d.z + Q3 * xv.z + Q4; b.w = Q1 * uv.w + Q2 * gv.w * d.w + Q3 * xv.w + Q4;
        reinterpret_cast<float4*>(g_gy)[i] = a;
        reinterpret_cast<float4*>(g_x)[i] = b;
    } else {
        if (i >= total) return;
        const int row = (int)(i / len);
        const float d = act != SEMPYR_ACT_NONE ? act_deriv(y[i], act, slope) : 1.f;
        const float xc = x[i] - mean_row[row];
        g_gy[i] = d * (coef[row] * u[i] + coef[rows + row] * xc + coef[2 * rows + row]);
        g_x[i] = coef[3 * rows + row] * u[i] + coef[4 * rows + row] * gy[i] * d + coef[5 * rows + row] * xc + coef[6 * rows + row];
    }
}

inline bool aligned16(const void* p) { return ((uintptr_t)p & 15) == 0; }

inline int check_norm_args(int kind, int N, int C, int HW, int act) {
    if (kind < 0 || kind > 2) SEMPYR_FAIL("norm: unknown kind %d", kind);
    if (N < 1 || C < 1 || HW < 1) SEMPYR_FAIL("norm: non-positive dims");
    if ((long long)N * C * HW >= (1ll << 31)) SEMPYR_FAIL("norm: tensor too large");
    if (act != SEMPYR_ACT_NONE && act != SEMPYR_ACT_RELU && act != SEMPYR_ACT_LRELU) SEMPYR_FAIL("norm: activation %d not fusable", act);
    return 0;
}

// ---- InstanceNorm2d fast paths: a row IS its statistics group, so statistics, coefficients and the apply pass need no
// cross-row step and run in ONE launch per direction.  A row (<= 64 KB) is read from HBM once; the second / third sweep of
// the same workgroup hits L1/L2.  TPR = 64: one wave per row (4 rows per workgroup); TPR = 256: one workgroup per row.
constexpr int INST_FUSED_MAX_LEN = 16384;
inline bool inst_fused_on() { static const bool on = !(getenv("SEMPYR_INST_FUSED") && getenv("SEMPYR_INST_FUSED")[0] == '0'); return on; }   // A/B aid

template <int TPR>
__global__ __launch_bounds__(256) void instnorm_fwd_kernel(const float* __restrict__ x, int rows, int C, int len, int vec,
                                                           float eps, const float* __restrict__ gamma,
                                                           const float* __restrict__ beta, int act, float slope,
                                                           float* __restrict__ y, float* __restrict__ rowbuf)
{
    __shared__ float sh[8];
    constexpr int RPB = 256 / TPR;
    const int row = blockIdx.x * RPB + threadIdx.x / TPR;
    const int t = threadIdx.x % TPR;
    const bool live = row < rows;
    const size_t base = (size_t)(live ? row : 0) * len;
    const float* xr = x + base;
    float s[1] = {0.f};
    if (live) {
        if (vec) {
            const float4* x4 = reinterpret_cast<const float4*>(xr);
            for (int i = t; i < len / 4; i += TPR) { const float4 v = x4[i]; s[0] += (v.x + v.y) + (v.z + v.w); }
        } else {
            for (int i = t; i < len; i += TPR) s[0] += xr[i];
        }
    }
    row_reduce<TPR, 1>(s, sh);
    const float mean = s[0] / (float)len;
    float q[1] = {0.f};
    if (live) {
        if (vec) {
            const float4* x4 = reinterpret_cast<const float4*>(xr);
            for (int i = t; i < len / 4; i += TPR) {
                const float4 v = x4[i];
                const float a = v.x - mean, b = v.y - mean, c = v.z - mean, d = v.w - mean;
                q[0] += (a * a + b * b) + (c * c + d * d);
            }
        } else {
            for (int i = t; i < len; i += TPR) { const float a = xr[i] - mean; q[0] += a * a; }
        }
    }
    row_reduce<TPR, 1>(q, sh);
    if (!live) return;
    const float var = q[0] / (float)len;
    const float rstd = 1.0f / sqrtf(var + eps);
    const int c = row % C;
    const float sc = rstd * gamma[c], sf = beta[c];
    if (t == 0) { rowbuf[row] = mean; rowbuf[rows + row] = rstd; rowbuf[2 * rows + row] = sc; rowbuf[3 * rows + row] = sf; }
    float* yr = y + base;
    if (vec) {
        const float4* x4 = reinterpret_cast<const float4*>(xr);
        float4* y4 = reinterpret_cast<float4*>(yr);
        for (int i = t; i < len / 4; i += TPR) {
            const float4 v = x4[i];
            float4 o;
            o.x = act_apply((v.x - mean) * sc + sf, act, slope); o.y = act_apply((v.y - mean) * sc + sf, act, slope);
            o.z = act_apply((v.z - mean) * sc + sf, act, slope); o.w = act_apply((v.w - mean) * sc + sf, act, slope);
            y4[i] = o;
        }
    } else {
        for (int i = t; i < len; i += TPR) yr[i] = act_apply((xr[i] - mean) * sc + sf, act, slope);
    }
}

// backward: pivot-shifted row sums as norm_bwd_rowsums_kernel, the coefficient step of norm_bwd_finalize_kernel for a
// one-row group (double math), then the apply sweep.  `sums` [4][rows] is still written: the channel kernel below turns it
// into ggamma / gbeta.
template <int TPR>
__global__ __launch_bounds__(256) void instnorm_bwd_kernel(const float* __restrict__ gy, const float* __restrict__ y,
                                                           const float* __restrict__ x, const float* __restrict__ gamma,
                                                           const float* __restrict__ mean_row, const float* __restrict__ rstd_row,
                                                           int rows, int C, int len, int vec, int act, float slope,
                                                           float* __restrict__ gx, float* __restrict__ sums)
{
    __shared__ float sh[12];
    constexpr int RPB = 256 / TPR;
    const int row = blockIdx.x * RPB + threadIdx.x / TPR;
    const int t = threadIdx.x % TPR;
    const bool live = row < rows;
    const size_t base = (size_t)(live ? row : 0) * len;
    const float mean = mean_row[live ? row : 0];
    float Kg = gy[base];
    if (act != SEMPYR_ACT_NONE) Kg *= act_deriv(y[base], act, slope);
    float s[3] = {0.f, 0.f, 0.f};
    if (live) {
        if (vec) {
            const float4* g4 = reinterpret_cast<const float4*>(gy + base);
            const float4* x4 = reinterpret_cast<const float4*>(x + base);
            const float4* y4 = reinterpret_cast<const float4*>(y + base);
            for (int i = t; i < len / 4; i += TPR) {
                float4 g = g4[i]; float4 xv = x4[i];
                if (act != SEMPYR_ACT_NONE) {
                    const float4 yv = y4[i];
                    g.x *= act_deriv(yv.x, act, slope); g.y *= act_deriv(yv.y, act, slope);
                    g.z *= act_deriv(yv.z, act, slope); g.w *= act_deriv(yv.w, act, slope);
                }
                g.x -= Kg; g.y -= Kg; g.z -= Kg; g.w -= Kg;
                xv.x -= mean; xv.y -= mean; xv.z -= mean; xv.w -= mean;
                s[0] += (g.x + g.y) + (g.z + g.w);
                s[1] += (xv.x + xv.y) + (xv.z + xv.w);
                s[2] += (g.x * xv.x + g.y * xv.y) + (g.z * xv.z + g.w * xv.w);
            }
        } else {
            for (int i = t; i < len; i += TPR) {
                float g = gy[base + i];
                if (act != SEMPYR_ACT_NONE) g *= act_deriv(y[base + i], act, slope);
                g -= Kg;
                const float xc = x[base + i] - mean;
                s[0] += g; s[1] += xc; s[2] += g * xc;
            }
        }
    }
    row_reduce<TPR, 3>(s, sh);
    if (!live) return;
    if (t == 0) { sums[row] = Kg; sums[rows + row] = s[0]; sums[2 * rows + row] = s[1]; sums[3 * rows + row] = s[2]; }
    const double r = rstd_row[row], gm = gamma[row % C];
    const double G1 = (double)Kg * len + (double)s[0];
    const double G2 = ((double)s[2] + (double)Kg * (double)s[1]) * r;
    const double m = (double)len;
    const float a = (float)(r * gm), b = (float)(-r * r * (gm * G2 / m)), c = (float)(-r * (gm * G1 / m));
    if (vec) {
        const float4* g4 = reinterpret_cast<const float4*>(gy + base);
        const float4* x4 = reinterpret_cast<const float4*>(x + base);
        const float4* y4 = reinterpret_cast<const float4*>(y + base);
        float4* o4 = reinterpret_cast<float4*>(gx + base);
        for (int i = t; i < len / 4; i += TPR) {
            float4 g = g4[i]; float4 xv = x4[i];
            xv.x -= mean; xv.y -= mean; xv.z -= mean; xv.w -= mean;
            if (act != SEMPYR_ACT_NONE) {
                const float4 yv = y4[i];
                g.x *= act_deriv(yv.x, act, slope); g.y *= act_deriv(yv.y, act, slope);
                g.z *= act_deriv(yv.z, act, slope); g.w *= act_deriv(yv.w, act, slope);
            }
            float4 o;
            o.x = a * g.x + b * xv.x + c; o.y = a * g.y + b * xv.y + c;
            o.z = a * g.z + b * xv.z + c; o.w = a * g.w + b * xv.w + c;
            o4[i] = o;
        }
    } else {
        for (int i = t; i < len; i += TPR) {
            float g = gy[base + i];
            if (act != SEMPYR_ACT_NONE) g *= act_deriv(y[base + i], act, slope);
            gx[base + i] = a * g + b * (x[base + i] - mean) + c;
        }
    }
}

// Register-resident forms: a row of TPR * 4 * NV4 elements is loaded ONCE (NV4 float4 per thread, all loads in flight
// together) and the two reductions and the apply step work on registers -- no second or third sweep.
template <int TPR, int NV4>
__global__ __launch_bounds__(256) void instnorm_fwd_reg_kernel(const float* __restrict__ x, int rows, int C, float eps,
                                                               const float* __restrict__ gamma, const float* __restrict__ beta,
                                                               int act, float slope, float* __restrict__ y,
                                                               float* __restrict__ rowbuf)
{
    __shared__ float sh[8];
    constexpr int RPB = 256 / TPR;
    constexpr int LEN = TPR * 4 * NV4;
    const int row = blockIdx.x * RPB + threadIdx.x / TPR;
    const int t = threadIdx.x % TPR;
    const bool live = row < rows;
    const float4* x4 = reinterpret_cast<const float4*>(x + (size_t)(live ? row : 0) * LEN);
    float4 v[NV4];
#pragma unroll
    for (int i = 0; i < NV4; ++i) v[i] = x4[t + i * TPR];
    float s[1] = {0.f};
#pragma unroll
    for (int i = 0; i < NV4; ++i) s[0] += (v[i].x + v[i].y) + (v[i].z + v[i].w);
    row_reduce<TPR, 1>(s, sh);
    const float mean = s[0] / (float)LEN;
    float q[1] = {0.f};
#pragma unroll
    for (int i = 0; i < NV4; ++i) {
        const float a = v[i].x - mean, b = v[i].y - mean, c = v[i].z - mean, d = v[i].w - mean;
        q[0] += (a * a + b * b) + (c * c + d * d);
    }
    row_reduce<TPR, 1>(q, sh);
    if (!live) return;
    const float rstd = 1.0f / sqrtf(q[0] / (float)LEN + eps);
    const int c = row % C;
    const float sc = rstd * gamma[c], sf = beta[c];
    if (t == 0) { rowbuf[row] = mean; rowbuf[rows + row] = rstd; rowbuf[2 * rows + row] = sc; rowbuf[3 * rows + row] = sf; }
    float4* y4 = reinterpret_cast<float4*>(y + (size_t)row * LEN);
#pragma unroll
    for (int i = 0; i < NV4; ++i) {
        float4 o;
        o.x = act_apply((v[i].x - mean) * sc + sf, act, slope); o.y = act_apply((v[i].y - mean) * sc + sf, act, slope);
        o.z = act_apply((v[i].z - mean) * sc + sf, act, slope); o.w = act_apply((v[i].w - mean) * sc + sf, act, slope);
        y4[t + i * TPR] = o;
    }
}

// Training-mode BatchNorm of a SMALL layer (N * HW <= 16384 elements per channel: the generator's 4^2 ... 32^2 blocks at
// batch 16) in ONE launch: a workgroup owns a channel, keeps its N rows in registers (NV4 float4 per thread, all loads in
// flight together), reduces mean and centred second moment over the whole channel, updates the running statistics and applies
// scale / shift / activation.  The three-launch form (row moments -> finalize -> apply) spends ~14 us of launch latency on
// tensors of 16 KB ... 1 MB, in a chain in front of the next convolution.
// APPLY = false: the statistics / coefficient step alone (residual tails apply two norms in one pass of their own); `rep`: the
// tensor stands for one in which every element occurs `rep` times (see norm_finalize_kernel).
template <int NV4, bool APPLY>
__global__ __launch_bounds__(256) void bn_fwd_small_kernel(const float* __restrict__ x, int N, int C, int HW, float eps,
                                                           const float* __restrict__ gamma, const float* __restrict__ beta,
                                                           float* running_mean, float* running_var, long long* nbt, float momentum,
                                                           float rep, int act, float slope, float* __restrict__ y, float* __restrict__ rowbuf)
{
    __shared__ float sh[8];
    const int c = blockIdx.x, t = threadIdx.x;
    const int q = HW >> 2;                          // float4 per row
    const int items = N * q;
    float4 v[NV4];
    size_t off[NV4];
#pragma unroll
    for (int i = 0; i < NV4; ++i) {
        const int it = t + i * 256;
        const bool live = it < items;
        const int n = live ? it / q : 0, e = live ? it - n * q : 0;
        off[i] = ((size_t)n * C + c) * HW + 4 * (size_t)e;
        v[i] = live ? *reinterpret_cast<const float4*>(x + off[i]) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    float s[1] = {0.f};
#pragma unroll
    for (int i = 0; i < NV4; ++i) s[0] += (v[i].x + v[i].y) + (v[i].z + v[i].w);
    row_reduce<256, 1>(s, sh);
    const float m = (float)N * (float)HW;
    const float mean = s[0] / m;
    __syncthreads();
    float r[1] = {0.f};
#pragma unroll
    for (int i = 0; i < NV4; ++i) {
        if (t + i * 256 < items) {
            const float a = v[i].x - mean, b = v[i].y - mean, cc = v[i].z - mean, d = v[i].w - mean;
            r[0] += (a * a + b * b) + (cc * cc + d * d);
        }
    }
    row_reduce<256, 1>(r, sh);
    const float m2 = r[0];
    const float var = m2 / m;
    const float rstd = 1.0f / sqrtf(var + eps);
    const float sc = rstd * gamma[c], sf = beta[c];
    const int rows = N * C;
    for (int b = t; b < N; b += 256) {
        const int row = b * C + c;
        rowbuf[row] = mean; rowbuf[rows + row] = rstd; rowbuf[2 * rows + row] = sc; rowbuf[3 * rows + row] = sf;
    }
    if (t == 0 && running_mean) {
        const float unbiased = m * rep > 1.f ? (m2 * rep) / (m * rep - 1.f) : var;
        running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * mean;
        running_var[c] = (1.f - momentum) * running_var[c] + momentum * unbiased;
        if (c == 0 && nbt) *nbt += 1;
    }
    if constexpr (!APPLY) return;
#pragma unroll
    for (int i = 0; i < NV4; ++i) {
        if (t + i * 256 < items) {
            float4 o;
            o.x = act_apply((v[i].x - mean) * sc + sf, act, slope); o.y = act_apply((v[i].y - mean) * sc + sf, act, slope);
            o.z = act_apply((v[i].z - mean) * sc + sf, act, slope); o.w = act_apply((v[i].w - mean) * sc + sf, act, slope);
            *reinterpret_cast<float4*>(y + off[i]) = o;
        }
    }
}

template <int TPR, int NV4>
__global__ __launch_bounds__(256) void instnorm_bwd_reg_kernel(const float* __restrict__ gy, const float* __restrict__ y,
                                                               const float* __restrict__ x, const float* __restrict__ gamma,
                                                               const float* __restrict__ mean_row, const float* __restrict__ rstd_row,
                                                               int rows, int C, int act, float slope,
                                                               float* __restrict__ gx, float* __restrict__ sums)
{
    __shared__ float sh[12];
    constexpr int RPB = 256 / TPR;
    constexpr int LEN = TPR * 4 * NV4;
    const int row = blockIdx.x * RPB + threadIdx.x / TPR;
    const int t = threadIdx.x % TPR;
    const bool live = row < rows;
    const size_t base = (size_t)(live ? row : 0) * LEN;
    const float mean = mean_row[live ? row : 0];
    const float4* g4 = reinterpret_cast<const float4*>(gy + base);
    const float4* x4 = reinterpret_cast<const float4*>(x + base);
    float4 g[NV4], xv[NV4];
#pragma unroll
    for (int i = 0; i < NV4; ++i) { g[i] = g4[t + i * TPR]; xv[i] = x4[t + i * TPR]; }
    float Kg = gy[base];
    if (act != SEMPYR_ACT_NONE) {
        const float4* y4 = reinterpret_cast<const float4*>(y + base);
        Kg *= act_deriv(y[base], act, slope);
#pragma unroll
        for (int i = 0; i < NV4; ++i) {
            const float4 yv = y4[t + i * TPR];
            g[i].x *= act_deriv(yv.x, act, slope); g[i].y *= act_deriv(yv.y, act, slope);
            g[i].z *= act_deriv(yv.z, act, slope); g[i].w *= act_deriv(yv.w, act, slope);
        }
    }
    float s[3] = {0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < NV4; ++i) {
        xv[i].x -= mean; xv[i].y -= mean; xv[i].z -= mean; xv[i].w -= mean;
        const float a = g[i].x - Kg, b = g[i].y - Kg, c = g[i].z - Kg, d = g[i].w - Kg;
        s[0] += (a + b) + (c + d);
        s[1] += (xv[i].x + xv[i].y) + (xv[i].z + xv[i].w);
        s[2] += (a * xv[i].x + b * xv[i].y) + (c * xv[i].z + d * xv[i].w);
    }
    row_reduce<TPR, 3>(s, sh);
    if (!live) return;
    if (t == 0) { sums[row] = Kg; sums[rows + row] = s[0]; sums[2 * rows + row] = s[1]; sums[3 * rows + row] = s[2]; }
    const double r = rstd_row[row], gm = gamma[row % C];
    const double G1 = (double)Kg * LEN + (double)s[0];
    const double G2 = ((double)s[2] + (double)Kg * (double)s[1]) * r;
    const float a = (float)(r * gm), b = (float)(-r * r * (gm * G2 / (double)LEN)), c = (float)(-r * (gm * G1 / (double)LEN));
    float4* o4 = reinterpret_cast<float4*>(gx + base);
#pragma unroll
    for (int i = 0; i < NV4; ++i) {
        float4 o;
        o.x = a * g[i].x + b * xv[i].x + c; o.y = a * g[i].y + b * xv[i].y + c;
        o.z = a * g[i].z + b * xv[i].z + c; o.w = a * g[i].w + b * xv[i].w + c;
        o4[t + i * TPR] = o;
    }
}

// host dispatch of the register-resident kernels: returns false when the row length has no instantiation
template <int TPR>
bool launch_instnorm_fwd_reg(int nv4, dim3 grid, hipStream_t st, const float* x, int rows, int C, float eps, const float* gamma,
                             const float* beta, int act, float slope, float* y, float* rowbuf)
{
    switch (nv4) {
#define SEMPYR_CASE(NV) case NV: hipLaunchKernelGGL((instnorm_fwd_reg_kernel<TPR, NV>), grid, dim3(256), 0, st, x, rows, C, eps, gamma, beta, act, slope, y, rowbuf); return true;
        SEMPYR_CASE(1) SEMPYR_CASE(2) SEMPYR_CASE(4) SEMPYR_CASE(8) SEMPYR_CASE(16)
#undef SEMPYR_CASE
        default: return false;
    }
}
template <int TPR>
bool launch_instnorm_bwd_reg(int nv4, dim3 grid, hipStream_t st, const float* gy, const float* y, const float* x,
                             const float* gamma, const float* mean_row, const float* rstd_row, int rows, int C, int act,
                             float slope, float* gx, float* sums)
{
    switch (nv4) {
#define SEMPYR_CASE(NV) case NV: hipLaunchKernelGGL((instnorm_bwd_reg_kernel<TPR, NV>), grid, dim3(256), 0, st, gy, y, x, gamma, mean_row, rstd_row, rows, C, act, slope, gx, sums); return true;
        SEMPYR_CASE(1) SEMPYR_CASE(2) SEMPYR_CASE(4) SEMPYR_CASE(8)
#undef SEMPYR_CASE
        default: return false;
    }
}

// ggamma[c] = sum_n sum gy'*xhat, gbeta[c] = sum_n sum gy' from the shifted row sums (double math, as the finalize kernel)
// FIN_LPG lanes per channel (one thread per channel walking its N rows was a chain of dependent L2 round trips: 9 us).
__global__ __launch_bounds__(256) void norm_bwd_channel_kernel(const float* __restrict__ sums, const float* __restrict__ rstd_row,
                                                               int N, int C, int len, float* __restrict__ ggamma,
                                                               float* __restrict__ gbeta)
{
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    const int c = idx / FIN_LPG, l = idx % FIN_LPG;
    const bool live = c < C;                       // uniform per 16-lane group; dead groups still join the shuffles
    const int rows = N * C;
    double a = 0.0, b = 0.0;
    if (live) {
        for (int n = l; n < N; n += FIN_LPG) {
            const int row = n * C + c;
            a += (double)sums[row] * len + (double)sums[rows + row];
            b += ((double)sums[3 * rows + row] + (double)sums[row] * (double)sums[2 * rows + row]) * (double)rstd_row[row];
        }
    }
    a = lpg_sum_d(a); b = lpg_sum_d(b);
    if (live && l == 0) {
        if (gbeta) gbeta[c] = (float)a;
        if (ggamma) ggamma[c] = (float)b;
    }
}

} // namespace

// one-launch training-mode BatchNorm of small layers (bn_fwd_small_kernel); SEMPYR_BN_SMALL=0: the multi-launch form (A/B aid)
inline bool bn_small_ok(int kind, int N, int HW, const float* x) {
    static const bool on = !(getenv("SEMPYR_BN_SMALL") && getenv("SEMPYR_BN_SMALL")[0] == '0');
    return on && kind == SEMPYR_NORM_BATCH && HW % 4 == 0 && (long long)N * HW <= 16384 && aligned16(x);
}
template <bool APPLY>
int launch_bn_small(const float* x, int N, int C, int HW, float eps, const float* gamma, const float* beta, float* running_mean,
                    float* running_var, int64_t* nbt, float momentum, int rep, int act, float slope, float* y, float* rowbuf,
                    hipStream_t st)
{
    const int items = N * (HW / 4);
    int nv4 = 1;
    while (nv4 * 256 < items) nv4 *= 2;
    switch (nv4) {
#define SEMPYR_CASE(NV) case NV: hipLaunchKernelGGL((bn_fwd_small_kernel<NV, APPLY>), dim3(C), dim3(256), 0, st, x, N, C, HW, eps, gamma, beta, \
                                                    running_mean, running_var, (long long*)nbt, momentum, (float)rep, act, slope, y, rowbuf); break;
        SEMPYR_CASE(1) SEMPYR_CASE(2) SEMPYR_CASE(4) SEMPYR_CASE(8) SEMPYR_CASE(16)
#undef SEMPYR_CASE
        default: SEMPYR_FAIL("norm: internal: %d float4 per thread", nv4);
    }
    SEMPYR_CHECK_LAUNCH();
    return 0;
}

extern "C" int sempyr_norm_stats(int kind, const float* x, int N, int C, int HW, float eps,
                                 const float* gamma, const float* beta,
                                 float* running_mean, float* running_var, int64_t* num_batches_tracked, float momentum,
                                 int rep, float* rowbuf, float* scratch, void* stream)
{
    if (int rc = check_norm_args(kind, N, C, HW, 0)) return rc;
    if (!x || !gamma || !beta || !rowbuf || !scratch) SEMPYR_FAIL("norm_stats: null pointer");
    if (rep < 1) SEMPYR_FAIL("norm_stats: replication factor %d", rep);
    hipStream_t st = (hipStream_t)stream;
    const int rows = N * C;
    if (bn_small_ok(kind, N, HW, x)) {
        HbmScope hs("bn_fwd_small_kernel", 4.0 * rows * HW, st);
        return launch_bn_small<false>(x, N, C, HW, eps, gamma, beta, running_mean, running_var, num_batches_tracked, momentum, rep,
                                      0, 0.f, nullptr, rowbuf, st);
    }
    const int vec = (HW % 4 == 0) && aligned16(x);
    float* rmean = scratch; float* rm2 = scratch + rows;
    {
        HbmScope hs(HW >= 2048 ? "row_moments_kernel<256>" : "row_moments_kernel<64>", 4.0 * rows * HW, st);
        if (HW >= 2048) hipLaunchKernelGGL(row_moments_kernel<256>, dim3(rows), dim3(256), 0, st, x, rows, HW, vec, rmean, rm2);
        else hipLaunchKernelGGL(row_moments_kernel<64>, dim3(ceil_div(rows, 4)), dim3(256), 0, st, x, rows, HW, vec, rmean, rm2);
    }
    SEMPYR_CHECK_LAUNCH();
    const int ng = num_groups(kind, N, C);
    HbmScope hs_fin("norm_finalize_kernel", 4.0 * 6.0 * rows, st);
    hipLaunchKernelGGL(norm_finalize_kernel, dim3(ceil_div((long long)ng * FIN_LPG, 256)), dim3(256), 0, st, kind, rmean, rm2, N, C, HW, eps,
                       gamma, beta, running_mean, running_var, (long long*)num_batches_tracked, momentum, (float)rep,
                       rowbuf, rowbuf + rows, rowbuf + 2 * rows, rowbuf + 3 * rows);
    SEMPYR_CHECK_LAUNCH();
    return 0;
}

extern "C" int sempyr_bn_stats_from_tiles(const float* tile_stats, int N, int C, int HW, int granule, float eps,
                                          const float* gamma, const float* beta,
                                          float* running_mean, float* running_var, int64_t* num_batches_tracked, float momentum,
                                          int rep, float* rowbuf, void* stream)
{
    if (int rc = check_norm_args(SEMPYR_NORM_BATCH, N, C, HW, 0)) return rc;
    if (!tile_stats || !gamma || !beta || !rowbuf) SEMPYR_FAIL("bn_stats_from_tiles: null pointer");
    if (granule < 1 || HW % granule != 0) SEMPYR_FAIL("bn_stats_from_tiles: %d pixels do not split into groups of %d", HW, granule);
    if (rep < 1) SEMPYR_FAIL("bn_stats_from_tiles: replication factor %d", rep);
    if (((uintptr_t)tile_stats & 7)) SEMPYR_FAIL("bn_stats_from_tiles: the statistics buffer must be 8-byte aligned");
    const int rows = N * C;
    HbmScope hs("bn_tile_finalize_kernel", 8.0 * rows * (HW / granule) + 16.0 * rows, (hipStream_t)stream);
    hipLaunchKernelGGL(bn_tile_finalize_kernel, dim3(C), dim3(256), 0, (hipStream_t)stream,
                       reinterpret_cast<const float2*>(tile_stats), N, C, HW / granule, granule, eps, gamma, beta,
                       running_mean, running_var, (long long*)num_batches_tracked, momentum, (float)rep,
                       rowbuf, rowbuf + rows, rowbuf + 2 * rows, rowbuf + 3 * rows);
    SEMPYR_CHECK_LAUNCH();
    return 0;
}

extern "C" int sempyr_norm_fwd(int kind, const float* x, int N, int C, int HW, float eps,
                               const float* gamma, const float* beta,
                               float* running_mean, float* running_var, int64_t* num_batches_tracked, float momentum,
                               int act, float slope, float* y, float* rowbuf, float* scratch, void* stream)
{
    if (int rc = check_norm_args(kind, N, C, HW, act)) return rc;
    if (!x || !gamma || !beta || !y || !rowbuf || !scratch) SEMPYR_FAIL("norm_fwd: null pointer");
    const int rows = N * C;
    if (kind == SEMPYR_NORM_INSTANCE && HW <= INST_FUSED_MAX_LEN && inst_fused_on()) {
        hipStream_t st = (hipStream_t)stream;
        const int vec = (HW % 4 == 0) && aligned16(x) && aligned16(y);
        bool done = false;
        HbmScope hs("instnorm_fwd_kernel", 8.0 * rows * HW, st);
        if (vec && HW >= 2048 && HW % 1024 == 0) done = launch_instnorm_fwd_reg<256>(HW / 1024, dim3(rows), st, x, rows, C, eps, gamma, beta, act, slope, y, rowbuf);
        else if (vec && HW < 2048 && HW % 256 == 0) done = launch_instnorm_fwd_reg<64>(HW / 256, dim3(ceil_div(rows, 4)), st, x, rows, C, eps, gamma, beta, act, slope, y, rowbuf);
        else if (vec && HW == 64) done = launch_instnorm_fwd_reg<16>(1, dim3(ceil_div(rows, 16)), st, x, rows, C, eps, gamma, beta, act, slope, y, rowbuf);
        else if (vec && HW == 16) done = launch_instnorm_fwd_reg<4>(1, dim3(ceil_div(rows, 64)), st, x, rows, C, eps, gamma, beta, act, slope, y, rowbuf);
        if (done) { SEMPYR_CHECK_LAUNCH(); return 0; }
        if (HW >= 2048) hipLaunchKernelGGL(instnorm_fwd_kernel<256>, dim3(rows), dim3(256), 0, st, x, rows, C, HW, vec, eps, gamma, beta, act, slope, y, rowbuf);
        else hipLaunchKernelGGL(instnorm_fwd_kernel<64>, dim3(ceil_div(rows, 4)), dim3(256), 0, st, x, rows, C, HW, vec, eps, gamma, beta, act, slope, y, rowbuf);
        SEMPYR_CHECK_LAUNCH();
        return 0;
    }
    if (bn_small_ok(kind, N, HW, x) && aligned16(y)) {
        HbmScope hs("bn_fwd_small_kernel", 8.0 * rows * HW, (hipStream_t)stream);
        return launch_bn_small<true>(x, N, C, HW, eps, gamma, beta, running_mean, running_var, num_batches_tracked, momentum, 1, act, slope,
                                     y, rowbuf, (hipStream_t)stream);
    }
    if (int rc = sempyr_norm_stats(kind, x, N, C, HW, eps, gamma, beta, running_mean, running_var, num_batches_tracked,
                                   momentum, 1, rowbuf, scratch, stream)) return rc;
    return sempyr_affine_act(x, rowbuf, rowbuf + 2 * rows, rowbuf + 3 * rows, nullptr, y, rows, HW, act, slope, stream);
}

extern "C" int sempyr_bn_eval_coeffs(const float* gamma, const float* beta, const float* running_mean,
                                     const float* running_var, float eps, int N, int C,
                                     float* scale_row, float* shift_row, void* stream)
{
    if (!gamma || !beta || !running_mean || !running_var || !scale_row || !shift_row) SEMPYR_FAIL("bn_eval_coeffs: null pointer");
    if (N < 1 || C < 1) SEMPYR_FAIL("bn_eval_coeffs: bad dims");
    hipLaunchKernelGGL(bn_eval_coeffs_kernel, dim3(ceil_div(N * C, 256)), dim3(256), 0, (hipStream_t)stream,
                       gamma, beta, running_mean, running_var, eps, N, C, scale_row, shift_row);
    SEMPYR_CHECK_LAUNCH();
    return 0;
}

extern "C" int sempyr_norm_bwd(int kind, const float* gy, const float* y, const float* x, const float* gamma,
                               const float* mean_row, const float* rstd_row, int N, int C, int HW,
                               int act, float slope, float* gx, float* ggamma, float* gbeta,
                               float* scratch, void* stream)
{
    if (int rc = check_norm_args(kind, N, C, HW, act)) return rc;
    if (!gy || !x || !gamma || !mean_row || !rstd_row || !gx || !scratch) SEMPYR_FAIL("norm_bwd: null pointer");
    if (act != SEMPYR_ACT_NONE && !y) SEMPYR_FAIL("norm_bwd: activation needs y");
    hipStream_t st = (hipStream_t)stream;
    const int rows = N * C;
    const int vec = (HW % 4 == 0) && aligned16(gy) && aligned16(x) && aligned16(gx) && (act == SEMPYR_ACT_NONE || aligned16(y));
    float* sums = scratch;
    if (kind == SEMPYR_NORM_INSTANCE && HW <= INST_FUSED_MAX_LEN && inst_fused_on()) {
        bool done = false;
        HbmScope* hs = new HbmScope("instnorm_bwd_kernel", 16.0 * rows * HW, st);
        if (vec && HW >= 2048 && HW % 1024 == 0) done = launch_instnorm_bwd_reg<256>(HW / 1024, dim3(rows), st, gy, y, x, gamma, mean_row, rstd_row, rows, C, act, slope, gx, sums);
        else if (vec && HW < 2048 && HW % 256 == 0) done = launch_instnorm_bwd_reg<64>(HW / 256, dim3(ceil_div(rows, 4)), st, gy, y, x, gamma, mean_row, rstd_row, rows, C, act, slope, gx, sums);
        else if (vec && HW == 64) done = launch_instnorm_bwd_reg<16>(1, dim3(ceil_div(rows, 16)), st, gy, y, x, gamma, mean_row, rstd_row, rows, C, act, slope, gx, sums);
        else if (vec && HW == 16) done = launch_instnorm_bwd_reg<4>(1, dim3(ceil_div(rows, 64)), st, gy, y, x, gamma, mean_row, rstd_row, rows, C, act, slope, gx, sums);
        if (!done) {
        if (HW >= 2048) hipLaunchKernelGGL(instnorm_bwd_kernel<256>, dim3(rows), dim3(256), 0, st, gy, y, x, gamma, mean_row, rstd_row, rows, C, HW, vec, act, slope, gx, sums);
        else hipLaunchKernelGGL(instnorm_bwd_kernel<64>, dim3(ceil_div(rows, 4)), dim3(256), 0, st, gy, y, x, gamma, mean_row, rstd_row, rows, C, HW, vec, act, slope, gx, sums);
        }
        delete hs;
        SEMPYR_CHECK_LAUNCH();
        if (ggamma || gbeta) {
            HbmScope hc("norm_bwd_channel_kernel", 4.0 * (5.0 * rows + 2.0 * C), st);
            hipLaunchKernelGGL(norm_bwd_channel_kernel, dim3(ceil_div((long long)C * FIN_LPG, 256)), dim3(256), 0, st, sums, rstd_row, N, C, HW, ggamma, gbeta);
            SEMPYR_CHECK_LAUNCH();
        }
        return 0;
    }
    float* cA = scratch + 4 * rows; float* cB = scratch + 5 * rows; float* cC = scratch + 6 * rows;
    {
        HbmScope hs(HW >= 2048 ? "norm_bwd_rowsums_kernel<256>" : "norm_bwd_rowsums_kernel<64>", 4.0 * rows * HW * (act == SEMPYR_ACT_NONE ? 2.0 : 3.0), st);
        if (HW >= 2048) hipLaunchKernelGGL(norm_bwd_rowsums_kernel<256>, dim3(rows), dim3(256), 0, st, gy, y, x, mean_row, rows, HW, vec, act, slope, sums);
        else hipLaunchKernelGGL(norm_bwd_rowsums_kernel<64>, dim3(ceil_div(rows, 4)), dim3(256), 0, st, gy, y, x, mean_row, rows, HW, vec, act, slope, sums);
    }
    SEMPYR_CHECK_LAUNCH();
    const int nthreads = (num_groups(kind, N, C) + C) * FIN_LPG;
    hipLaunchKernelGGL(norm_bwd_finalize_kernel, dim3(ceil_div(nthreads, 256)), dim3(256), 0, st, kind, sums, N, C, HW, gamma,
                       rstd_row, cA, cB, cC, ggamma, gbeta);
    SEMPYR_CHECK_LAUNCH();
    const long long total = (long long)rows * HW;
    const long long nthr = vec ? total / 4 : total;
    HbmScope hs_apply("norm_bwd_apply_kernel", 4.0 * total * (act == SEMPYR_ACT_NONE ? 3.0 : 4.0), st);
    hipLaunchKernelGGL(norm_bwd_apply_kernel, dim3(ceil_div(nthr, 256)), dim3(256), 0, st, gy, y, x, mean_row, cA, cB, cC, total, HW, vec, act, slope, gx);
    SEMPYR_CHECK_LAUNCH();
    return 0;
}

extern "C" int sempyr_norm_bwd_bwd(int kind, const float* u, const float* gy, const float* y, const float* x,
                                   const float* gamma, const float* mean_row, const float* rstd_row,
                                   int N, int C, int HW, int act, float slope,
                                   float* g_gy, float* g_x, float* g_gamma, float* scratch, void* stream)
{
    if (int rc = check_norm_args(kind, N, C, HW, act)) return rc;
    if (!u || !gy || !x || !gamma || !mean_row || !rstd_row || !g_gy || !g_x || !scratch) SEMPYR_FAIL("norm_bwd_bwd: null pointer");
    if (act != SEMPYR_ACT_NONE && !y) SEMPYR_FAIL("norm_bwd_bwd: activation needs y");
    hipStream_t st = (hipStream_t)stream;
    const int rows = N * C;
    const int vec = (HW % 4 == 0) && aligned16(u) && aligned16(gy) && aligned16(x) && aligned16(g_gy) && aligned16(g_x) &&
                    (act == SEMPYR_ACT_NONE || aligned16(y));
    float* sums = scratch; float* coef = scratch + 9 * rows;
    {
        HbmScope hs(HW >= 2048 ? "norm_bwdbwd_rowsums_kernel<256>" : "norm_bwdbwd_rowsums_kernel<64>", 4.0 * rows * HW * (act == SEMPYR_ACT_NONE ? 3.0 : 4.0), st);
        if (HW >= 2048) hipLaunchKernelGGL(norm_bwdbwd_rowsums_kernel<256>, dim3(rows), dim3(256), 0, st, u, gy, y, x, mean_row, rows, HW, vec, act, slope, sums);
        else hipLaunchKernelGGL(norm_bwdbwd_rowsums_kernel<64>, dim3(ceil_div(rows, 4)), dim3(256), 0, st, u, gy, y, x, mean_row, rows, HW, vec, act, slope, sums);
    }
    SEMPYR_CHECK_LAUNCH();
    hipLaunchKernelGGL(norm_bwdbwd_finalize_kernel, dim3(ceil_div((long long)num_groups(kind, N, C) * FIN_LPG, 256)), dim3(256), 0, st,
                       kind, sums, N, C, HW, gamma, rstd_row, coef);
    SEMPYR_CHECK_LAUNCH();
    if (g_gamma) {
        hipLaunchKernelGGL(norm_bwdbwd_ggamma_kernel, dim3(ceil_div(C, 64)), dim3(64), 0, st, kind, sums, coef, rstd_row, N, C, HW, g_gamma);
        SEMPYR_CHECK_LAUNCH();
    }
    const long long total = (long long)rows * HW;
    const long long nthr = vec ? total / 4 : total;
    HbmScope hs_apply("norm_bwdbwd_apply_kernel", 4.0 * total * (act == SEMPYR_ACT_NONE ? 5.0 : 6.0), st);
    hipLaunchKernelGGL(norm_bwdbwd_apply_kernel, dim3(ceil_div(nthr, 256)), dim3(256), 0, st, u, gy, y, x, mean_row, coef, rows, total, HW, vec, act, slope, g_gy, g_x);
    SEMPYR_CHECK_LAUNCH();
    return 0;
}
