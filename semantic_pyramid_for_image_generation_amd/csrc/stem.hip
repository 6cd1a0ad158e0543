// The critic stem of dis-type 'res' (discriminator.py:101-109): Conv2d(3 -> 64, 3x3, stride 1, pad 1) + bias -> LeakyReLU ->
// AvgPool2d(2), as ONE kernel that writes only the pooled map and one sign bit per full-resolution element (all its
// backward ever needs), and its double-backward twin (the gradient penalty, train_model.py:69-76): conv(u, w) * act'(bits)
// -> AvgPool2d(2).
//
// Why a kernel of its own: with Cin = 3 the GEMM reduction is K = 27.  The generic implicit-GEMM kernel decodes every
// gathered element separately (its uniform-tap path needs Cin % 16 == 0) and ran this layer at 0.18 of the fp32 MFMA peak
// (126 us for 16 images).  Here a workgroup owns two output rows x 256 columns of one image: the 3 x 4 x 258 input patch is
// staged in LDS once, the 64 x 27 weights sit in registers as MFMA A-fragments for the whole kernel, and each
// v_mfma_f32_32x32x2_f32 takes its B-fragment with one ds_read_b32 (per-lane pixel base + per-k-step tap offset).  Lanes are
// numbered in "quad" order (the four pixels of a pooling window are adjacent lanes), so the 2x2 average is two DPP permutes
// and a ballot over the wave gives the sign bits of 32 pixels of two channels at once.
//
// MFMA bound: 2 * 64 * 28 FLOP per pixel (K padded 27 -> 28) = 3.76 GFLOP per 16 images = 24 us at the 157.3 TFLOP/s peak;
// HBM: 12.6 MB in, 67 MB pooled + 8.4 MB bits out.
#include "common.h"
#include <stdlib.h>

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int ST_COLS = 256;                 // output columns per workgroup
constexpr int ST_PW = 272;                   // LDS row pitch (floats): 272 % 64 = 16 -> the two pixel rows of a lane group use disjoint banks
constexpr int ST_K = 27, ST_STEPS = 14;      // k = tap*3 + c (the packed FWD weight layout of conv_igemm.hip); 14 MFMA k-steps of 2

__host__ __device__ constexpr int st_tap_off(int k) {        // LDS offset of reduction index k relative to the pixel's (dy, dx) = (0, 0) tap
    return k >= ST_K ? 0 : (k % 3) * 4 * ST_PW + ((k / 3) / 3) * ST_PW + ((k / 3) % 3);
}

struct StemParams {
    const float* x;          // [B][3][H][W]
    const float* w;          // packed FWD weights [32][Mpad]: w[k][m], k = tap*3 + c, rows 27.. are zero
    int Mpad;
    const float* bias;       // [64] or null
    float* pooled;           // [B][64][H/2][W/2]
    uint32_t* bits;          // out (plain) / in (MASKED): [B][64][H*W/32], quad pixel order
    int B, H, W;
    float slope;             // LeakyReLU slope (0: ReLU)
};

constexpr int ST_PP = 132;                   // LDS pitch of the staged pooled tile (floats): the two accumulator halves (rows m, m + 4) hit disjoint banks

template <bool MASKED>
__global__ __launch_bounds__(256, 3) void stem3x3_pool_kernel(const StemParams p)
{
    __shared__ float patch[3 * 4 * ST_PW];
    __shared__ __attribute__((aligned(8))) uint32_t sbits[64 * 16];      // this tile's sign words: [word column][slot], slot = 2 (16 i + r) + half <-> channel m
    __shared__ __attribute__((aligned(16))) float spool[64 * ST_PP];       // pooled tile [channel][128 windows], stored coalesced at the end
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, half = lane >> 5;
    const int ctiles = p.W / ST_COLS;
    int id = blockIdx.x;
    const int ct = id % ctiles; id /= ctiles;
    const int yp = id % (p.H >> 1);
    const int b = id / (p.H >> 1);
    const int x0 = ct * ST_COLS, y0 = 2 * yp;
    const int Wh = p.W >> 1, Hh = p.H >> 1;
    const int wpp = (p.H * p.W) >> 5;                    // bit words per plane
    const int win0 = yp * Wh + (x0 >> 1);                // first pooling window of the tile (a multiple of 128)
    uint32_t* gbits = p.bits + (size_t)b * 64 * wpp + (win0 >> 3);          // 16 consecutive words per channel

    // ---- stage the 3 x 4 x 258 input patch (zero outside the image) ----
    const float* xb = p.x + (size_t)b * 3 * p.H * p.W;
    {
        const int gx = x0 - 1 + tid;                              // patch column tid of every (channel, row); columns 256, 257 by the first threads
        const bool in_x = gx >= 0 && gx < p.W;
        const int gx2 = x0 + 255 + (tid & 1);                     // patch columns 256 / 257
        const bool in_x2 = gx2 < p.W;
        float v[12], v2 = 0.f;
#pragma unroll
        for (int rr = 0; rr < 12; ++rr) {
            const int gy = y0 - 1 + (rr & 3);
            const bool in_y = gy >= 0 && gy < p.H;
            const float* row = xb + ((size_t)(rr >> 2) * p.H + (in_y ? gy : 0)) * p.W;
            v[rr] = (in_y && in_x) ? row[gx] : 0.f;
            if (tid >> 1 == rr) v2 = (in_y && in_x2) ? row[gx2] : 0.f;      // threads 2 rr, 2 rr + 1 fetch the two right halo columns of row rr
        }
#pragma unroll
        for (int rr = 0; rr < 12; ++rr) patch[rr * ST_PW + tid] = v[rr];
        if (tid < 24) patch[(tid >> 1) * ST_PW + 256 + (tid & 1)] = v2;
    }
    if (MASKED) {
#pragma unroll
        for (int e = tid; e < 64 * 16; e += 256) {               // e = m * 16 + word column (coalesced 64-B runs)
            const int m = e >> 4, i = m >> 5, q = m & 31;         // m = 32 i + (r & 3) + 8 (r >> 2) + 4 half
            const int r = (q & 3) + 4 * (q >> 3), hf = (q >> 2) & 1;
            sbits[(e & 15) * 64 + 2 * (16 * i + r) + hf] = gbits[(size_t)m * wpp + (e & 15)];
        }
    }
    // ---- weights as A-fragments: lane holds W[m = 32 i + l31][k = 2 s + half] (row 27 of the packed layout is zero padding) ----
    float a[ST_STEPS][2];
#pragma unroll
    for (int s = 0; s < ST_STEPS; ++s) {
        const int k = 2 * s + half;
#pragma unroll
        for (int i = 0; i < 2; ++i) a[s][i] = p.w[k * p.Mpad + 32 * i + l31];
    }
    // bias of the 32 accumulator rows of this lane: m = 32 i + (r & 3) + 8 (r >> 2) + 4 half
    float bias[2][16];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) bias[i][r] = (!MASKED && p.bias) ? p.bias[32 * i + (r & 3) + 8 * (r >> 2) + 4 * half] : 0.f;
    // per-lane pixel bases of the 4 n-tiles of this wave: quad order, window = l31 >> 2, q = l31 & 3 -> (row q >> 1, col & 1)
    int base[4];
#pragma unroll
    for (int j = 0; j < 4; ++j)
        base[j] = ((l31 >> 1) & 1) * ST_PW + 64 * wave + 16 * j + 2 * (l31 >> 2) + (l31 & 1);
    __syncthreads();

    f32x16 acc[2][4];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

#pragma unroll
    for (int s = 0; s < ST_STEPS; ++s) {
        const int off = half ? st_tap_off(2 * s + 1) : st_tap_off(2 * s);
        float bv[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) bv[j] = patch[base[j] + off];
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int i = 0; i < 2; ++i)
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s][i], bv[j], acc[i][j], 0, 0, 0);
    }

    // ---- epilogue in registers + LDS: accumulator register r of lane l is row m (above), column n = l31 of n-tile j ----
    const bool lead = (l31 & 3) == 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int wcol = 4 * wave + j;                    // word column of this lane group inside the tile
        const int pcol = 32 * wave + 8 * j + (l31 >> 2);  // pooled column of this lane's window inside the tile
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            float pv[16];
            unsigned long long bal[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float v = acc[i][j][r];
                if (MASKED) {
                    v *= ((sbits[wcol * 64 + 2 * (16 * i + r) + half] >> l31) & 1u) ? 1.f : p.slope;
                } else {
                    v += bias[i][r];
                    v = v > 0.f ? v : v * p.slope;
                    bal[r] = __builtin_amdgcn_ballot_w64(v > 0.f);            // low word: channel m (half 0), high word: m + 4 (half 1)
                }
                v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xf, 0xf, true));
                v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xf, 0xf, true));
                pv[r] = 0.25f * v;
            }
            if (lead) {
#pragma unroll
                for (int r = 0; r < 16; ++r) spool[(32 * i + (r & 3) + 8 * (r >> 2) + 4 * half) * ST_PP + pcol] = pv[r];
            }
            if (!MASKED && lane == 0) {
#pragma unroll
                for (int r = 0; r < 16; ++r) *reinterpret_cast<unsigned long long*>(&sbits[wcol * 64 + 2 * (16 * i + r)]) = bal[r];
            }
        }
    }
    __syncthreads();
    // ---- coalesced stores: 64 channels x 128 pooled floats (512 B runs), 64 x 16 bit words (64 B runs) ----
    float* gp = p.pooled + (size_t)b * 64 * Hh * Wh + win0;
#pragma unroll
    for (int e = tid; e < 64 * 32; e += 256) {
        const int m = e >> 5, c4 = e & 31;
        *reinterpret_cast<float4*>(gp + (size_t)m * Hh * Wh + 4 * c4) = *reinterpret_cast<const float4*>(&spool[m * ST_PP + 4 * c4]);
    }
    if (!MASKED) {
#pragma unroll
        for (int e = tid; e < 64 * 16; e += 256) {
            const int m = e >> 4, i = m >> 5, q = m & 31;
            const int r = (q & 3) + 4 * (q >> 3), hf = (q >> 2) & 1;
            gbits[(size_t)m * wpp + (e & 15)] = sbits[(e & 15) * 64 + 2 * (16 * i + r) + hf];
        }
    }
}

}  // namespace

// Entry used by sempyr_conv2d_fwd (conv_igemm.hip) when the descriptor is the stem's: returns 1 when it took the launch,
// 0 when the shape is not the stem's (the generic path then runs), or an error code.
int sempyr_stem_try_launch(const sempyr_conv_desc* d, const float* x, const float* w_packed, int Mpad, int Kpad, const float* bias,
                           int act, float slope, float* y_pooled, uint32_t* act_bits, const uint32_t* mask_bits, void* stream)
{
    static const bool enabled = [] { const char* e = getenv("SEMPYR_STEM_KERNEL"); return !(e && e[0] == '0'); }();
    if (!enabled) return 0;
    if (!(d->Cin == 3 && d->Cout == 64 && d->kh == 3 && d->kw == 3 && d->stride == 1 && d->pad == 1 && d->ups == 0)) return 0;
    if (!w_packed || Mpad < 64 || Kpad < 2 * ST_STEPS || !y_pooled || (!act_bits && !mask_bits)) return 0;
    if (d->Win % ST_COLS != 0 || (d->Hin & 1) || ((long long)d->Hin * d->Win) % 32 != 0) return 0;
    if (!mask_bits && !(act == SEMPYR_ACT_RELU || act == SEMPYR_ACT_LRELU)) return 0;
    StemParams p;
    p.x = x; p.w = w_packed; p.Mpad = Mpad; p.bias = mask_bits ? nullptr : bias; p.pooled = y_pooled;
    p.bits = mask_bits ? const_cast<uint32_t*>(mask_bits) : act_bits;
    p.B = d->B; p.H = d->Hin; p.W = d->Win;
    p.slope = (mask_bits || act == SEMPYR_ACT_LRELU) ? slope : 0.f;
    const long long blocks = (long long)d->B * (d->Hin / 2) * (d->Win / ST_COLS);
    if (blocks >= (1ll << 31)) SEMPYR_FAIL("stem: too many workgroups");
    if (mask_bits) hipLaunchKernelGGL(stem3x3_pool_kernel<true>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, p);
    else hipLaunchKernelGGL(stem3x3_pool_kernel<false>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, p);
    SEMPYR_CHECK_LAUNCH();
    return 1;
}
