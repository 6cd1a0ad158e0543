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
    // All loads are issued before the first use: out-of-image elements read a clamped address and are zeroed by the select
    // at the LDS store (a load under a divergent branch is followed by a full vmcnt wait at the join, one round trip per row).
    const float* xb = p.x + (size_t)b * 3 * p.H * p.W;
    {
        const int gx = x0 - 1 + tid;                              // patch column tid of every (channel, row)
        const bool in_x = gx >= 0 && gx < p.W;
        const int gxc = min(max(gx, 0), p.W - 1);
        float v[12];
#pragma unroll
        for (int rr = 0; rr < 12; ++rr) {
            const int gy = y0 - 1 + (rr & 3);
            v[rr] = xb[((size_t)(rr >> 2) * p.H + min(max(gy, 0), p.H - 1)) * p.W + gxc];
        }
        const int r2 = min(tid >> 1, 11);                         // threads 2 rr, 2 rr + 1 fetch patch columns 256 / 257 of row rr
        const int gx2 = x0 + 255 + (tid & 1);
        const int gy2 = y0 - 1 + (r2 & 3);
        const float v2 = xb[((size_t)(r2 >> 2) * p.H + min(max(gy2, 0), p.H - 1)) * p.W + min(gx2, p.W - 1)];
#pragma unroll
        for (int rr = 0; rr < 12; ++rr) {
            const int gy = y0 - 1 + (rr & 3);
            patch[rr * ST_PW + tid] = (in_x && gy >= 0 && gy < p.H) ? v[rr] : 0.f;
        }
        if (tid < 24) patch[r2 * ST_PW + 256 + (tid & 1)] = (gx2 < p.W && gy2 >= 0 && gy2 < p.H) ? v2 : 0.f;
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

// ------------------------------------------------------------------------------------------------------------------------
// Weight (+ bias) gradient of the stem from what its forward kept: gW[m][c][ky][kx] = sum_{b,pix} gz[b][m][pix] * x[b][c][pix + tap],
// gz = 0.25 * up2(gp) * act'(bits) built on the fly from the POOLED gradient gp and the sign bits -- the full-resolution
// gradient (268 MB for 16 images) is neither written nor read.  One v_mfma_f32_32x32x2_f32 covers 32 channels (M) x 32 "taps"
// (N: the 27 weights of a channel, column 27 = the bias gradient through a B-operand of ones) x 2 pixels (K); the reduction
// runs over the pixels of the tile in quad order, so the two pixels of a k-step share one pooled value.  A workgroup walks
// tiles t, t + grid, ..., keeps the 64 x 32 sums in registers, adds its four waves through LDS and writes one partial; a
// second kernel adds the partials in a fixed order (deterministic).
// MFMA bound: 2 MFMAs per pixel pair = 27 us per 16 images; HBM: 67 MB pooled gradient + 8.4 MB bits + 12.6 MB input.
// ------------------------------------------------------------------------------------------------------------------------
constexpr int SW_CS = 4 * ST_PW + 4;         // channel stride of the patch: the 27 taps x 2 pixels of a B-fragment read hit distinct banks
constexpr int SW_GP = 129;                   // pitch of the staged pooled-gradient tile [64][128]
constexpr int SW_BP = 17;                    // pitch of the staged bit words [64][16]
constexpr int SW_ONES = ST_PW + 2 * 128 + 2;   // every (row pp, column) offset a B-operand read can have
constexpr int SW_OUT = 64 * 32;              // sums per partial: [m][n], n = tap*3 + c (< 27), 27 = bias, 28..31 unused

struct StemWgradParams {
    const float* x;          // [B][3][H][W]
    const float* gp;         // [B][64][H/2][W/2]
    const uint32_t* bits;    // [B][64][H*W/32]
    float* partials;         // [grid][SW_OUT]
    int B, H, W, ntiles;
    float slope;
};

struct StemWgradTile {                       // one tile's global loads, held in registers while the previous tile is reduced
    float v[12], v2;
    float4 g4[8];
    uint32_t bw4[4];
};

__global__ __launch_bounds__(256, 2) void stem3x3_wgrad_kernel(const StemWgradParams p)
{
    __shared__ float patch[3 * SW_CS + SW_ONES];         // + a block of ones
    __shared__ float sgp[64 * SW_GP];                    // reused for the cross-wave reduction (4 x 2048 floats <= 64 x 129)
    __shared__ uint32_t sb[64 * SW_BP];
    static_assert(4 * SW_OUT <= 64 * SW_GP, "the reduction reuses the gradient tile");
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, half = lane >> 5;
    const int ctiles = p.W / ST_COLS;
    const int Wh = p.W >> 1, Hh = p.H >> 1;
    const int wpp = (p.H * p.W) >> 5;
    // B-operand: lane n = l31 reads the input at its tap; column 27 is the constant one, 28.. read tap 0 and are dropped
    const int n_c = l31 % 3, n_tap = l31 / 3;
    const int off = l31 < ST_K ? n_c * SW_CS + (n_tap / 3) * ST_PW + (n_tap % 3) : (l31 == ST_K ? 3 * SW_CS : 0);
    const float q1 = 0.25f, q0 = 0.25f * p.slope;

    auto fetch = [&](StemWgradTile& T, int t) {
        int id = t;
        const int ct = id % ctiles; id /= ctiles;
        const int yp = id % Hh;
        const int b = id / Hh;
        const int x0 = ct * ST_COLS, y0 = 2 * yp;
        const int win0 = yp * Wh + (x0 >> 1);
        const float* xb = p.x + (size_t)b * 3 * p.H * p.W;
        const float* gpt = p.gp + (size_t)b * 64 * Hh * Wh + win0;
        const uint32_t* gbits = p.bits + (size_t)b * 64 * wpp + (win0 >> 3);
        // branch-free and select-free: out-of-image elements load a clamped address and are zeroed when the tile is written
        // to LDS (a divergent load, or a select on the loaded value, would make the wave wait for memory right here)
        const int gxc = min(max(x0 - 1 + tid, 0), p.W - 1);
#pragma unroll
        for (int rr = 0; rr < 12; ++rr) {
            const int gy = y0 - 1 + (rr & 3);
            T.v[rr] = xb[((size_t)(rr >> 2) * p.H + min(max(gy, 0), p.H - 1)) * p.W + gxc];
        }
        {                                                // threads 2 rr, 2 rr + 1 fetch the two right halo columns of patch row rr
            const int rr = min(tid >> 1, 11);
            const int gx2 = x0 + 255 + (tid & 1);
            const int gy = y0 - 1 + (rr & 3);
            T.v2 = xb[((size_t)(rr >> 2) * p.H + min(max(gy, 0), p.H - 1)) * p.W + min(gx2, p.W - 1)];
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) {                    // element tid + 256 e: channel m = (tid >> 5) + 8 e, float4 column tid & 31
            const int m = (tid >> 5) + 8 * e;
            T.g4[e] = *reinterpret_cast<const float4*>(gpt + (size_t)m * Hh * Wh + 4 * (tid & 31));
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) T.bw4[e] = gbits[(size_t)((tid >> 4) + 16 * e) * wpp + (tid & 15)];
    };

    f32x16 acc[2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;

    struct WgOperands { float gp[2][8]; float b[16]; uint32_t bw[2]; };
    WgOperands ops[2];
    auto lds_fetch = [&](WgOperands& O, int wc) {        // word column wc of the tile: windows 8 wc .. 8 wc + 7
#pragma unroll
        for (int i = 0; i < 2; ++i) O.bw[i] = sb[(32 * i + l31) * SW_BP + wc];
#pragma unroll
        for (int win = 0; win < 8; ++win) {
#pragma unroll
            for (int i = 0; i < 2; ++i) O.gp[i][win] = sgp[(32 * i + l31) * SW_GP + 8 * wc + win];
#pragma unroll
            for (int pp = 0; pp < 2; ++pp) O.b[2 * win + pp] = patch[pp * ST_PW + 2 * (8 * wc + win) + half + off];
        }
    };
    for (int e = tid; e < SW_ONES; e += 256) patch[3 * SW_CS + e] = 1.f;       // what the bias column (lane 27) reads

    StemWgradTile T;
    if ((int)blockIdx.x < p.ntiles) fetch(T, blockIdx.x);
    for (int t = blockIdx.x; t < p.ntiles; t += gridDim.x) {
        __syncthreads();                                 // the previous tile's LDS reads are done
        {
            const int ct = t % ctiles, yp = (t / ctiles) % Hh;
            const int x0 = ct * ST_COLS, y0 = 2 * yp;
            const int gx = x0 - 1 + tid;
            const bool in_x = gx >= 0 && gx < p.W;
#pragma unroll
            for (int rr = 0; rr < 12; ++rr) {
                const int gy = y0 - 1 + (rr & 3);
                patch[(rr >> 2) * SW_CS + (rr & 3) * ST_PW + tid] = (in_x && gy >= 0 && gy < p.H) ? T.v[rr] : 0.f;
            }
            if (tid < 24) {
                const int gy = y0 - 1 + ((tid >> 1) & 3);
                const bool ok = x0 + 255 + (tid & 1) < p.W && gy >= 0 && gy < p.H;
                patch[(tid >> 3) * SW_CS + ((tid >> 1) & 3) * ST_PW + 256 + (tid & 1)] = ok ? T.v2 : 0.f;
            }
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            float* dst = &sgp[((tid >> 5) + 8 * e) * SW_GP + 4 * (tid & 31)];
            dst[0] = T.g4[e].x; dst[1] = T.g4[e].y; dst[2] = T.g4[e].z; dst[3] = T.g4[e].w;
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) sb[((tid >> 4) + 16 * e) * SW_BP + (tid & 15)] = T.bw4[e];
        __syncthreads();
        if (t + (int)gridDim.x < p.ntiles) fetch(T, t + gridDim.x);        // in flight while this tile is reduced

        // wave w: windows 32 w .. 32 w + 31 of the tile = 4 word columns j of 8 windows.  The LDS operands of column j + 1
        // are requested before the 32 MFMAs of column j are issued (the compiler otherwise sinks every ds_read to its use
        // and the wave sits out the LDS latency once per k-step).
        lds_fetch(ops[0], 4 * wave);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (j < 3) lds_fetch(ops[(j + 1) & 1], 4 * wave + j + 1);
            __builtin_amdgcn_sched_barrier(0);
            const WgOperands& O = ops[j & 1];
#pragma unroll
            for (int win = 0; win < 8; ++win) {
#pragma unroll
                for (int pp = 0; pp < 2; ++pp) {                              // pixel pair (row pp, columns 2 wi, 2 wi + 1): k = half
                    const int bit = 4 * win + 2 * pp + half;
                    const float a0 = O.gp[0][win] * (((O.bw[0] >> bit) & 1u) ? q1 : q0);
                    const float a1 = O.gp[1][win] * (((O.bw[1] >> bit) & 1u) ? q1 : q0);
                    acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, O.b[2 * win + pp], acc[0], 0, 0, 0);
                    acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, O.b[2 * win + pp], acc[1], 0, 0, 0);
                }
            }
        }
    }
    // ---- add the four waves (fixed order) and write this workgroup's partial ----
    __syncthreads();
    float* red = sgp;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r)
            red[wave * SW_OUT + (32 * i + (r & 3) + 8 * (r >> 2) + 4 * half) * 32 + l31] = acc[i][r];
    __syncthreads();
    float* out = p.partials + (size_t)blockIdx.x * SW_OUT;
#pragma unroll
    for (int e = tid; e < SW_OUT; e += 256)
        out[e] = (red[e] + red[SW_OUT + e]) + (red[2 * SW_OUT + e] + red[3 * SW_OUT + e]);
}

// out[o] (+)= sum_p partials[p][o]: 16 outputs per workgroup, 16 lanes per output walk the partials (four independent
// chains, so the loads overlap), fixed-order finish.
__global__ __launch_bounds__(256) void stem_wgrad_reduce_kernel(const float* __restrict__ partials, int nparts,
                                                                float* __restrict__ gw, int acc_w, float* __restrict__ gb, int acc_b)
{
    __shared__ float red[16][17];
    const int o16 = threadIdx.x & 15, pl = threadIdx.x >> 4;
    const int o = blockIdx.x * 16 + o16;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    int q = pl;
    for (; q + 48 < nparts; q += 64) {
        s0 += partials[(size_t)q * SW_OUT + o];
        s1 += partials[(size_t)(q + 16) * SW_OUT + o];
        s2 += partials[(size_t)(q + 32) * SW_OUT + o];
        s3 += partials[(size_t)(q + 48) * SW_OUT + o];
    }
    for (; q < nparts; q += 16) s0 += partials[(size_t)q * SW_OUT + o];
    red[pl][o16] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (threadIdx.x < 16) {
        float tot = 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) tot += red[k][o16];
        const int m = o >> 5, n = o & 31;
        if (n < ST_K) {
            float* d = gw + m * ST_K + (n % 3) * 9 + n / 3;              // torch layout [m][c][ky][kx]
            *d = acc_w ? *d + tot : tot;
        } else if (n == ST_K && gb) {
            gb[m] = acc_b ? gb[m] + tot : tot;
        }
    }
}

int stem_wgrad_grid(long long ntiles) { return (int)(ntiles < 512 ? ntiles : 512); }     // two resident workgroups per CU

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

static bool stem_wgrad_shape_ok(const sempyr_conv_desc* d)
{
    return d->Cin == 3 && d->Cout == 64 && d->kh == 3 && d->kw == 3 && d->stride == 1 && d->pad == 1 && d->ups == 0 &&
           d->Win % ST_COLS == 0 && !(d->Hin & 1) && ((long long)d->Hin * d->Win) % 32 == 0;
}

extern "C" int sempyr_stem_wgrad_workspace(const sempyr_conv_desc* d, size_t* bytes)
{
    if (!d || !bytes) SEMPYR_FAIL("stem wgrad: null pointer");
    if (!stem_wgrad_shape_ok(d)) SEMPYR_FAIL("stem wgrad: built for Conv2d(3 -> 64, 3x3, s1, p1) on rows that are a multiple of 256 pixels");
    const long long ntiles = (long long)d->B * (d->Hin / 2) * (d->Win / ST_COLS);
    *bytes = (size_t)stem_wgrad_grid(ntiles) * SW_OUT * sizeof(float);
    return 0;
}

extern "C" int sempyr_stem_wgrad(const sempyr_conv_desc* d, const float* g_pooled, const uint32_t* act_bits, float slope,
                                 const float* x, float* gw, int accumulate_w, float* gbias, int accumulate_b,
                                 void* ws, size_t ws_bytes, void* stream)
{
    if (!d || !stem_wgrad_shape_ok(d)) SEMPYR_FAIL("stem wgrad: built for Conv2d(3 -> 64, 3x3, s1, p1) on rows that are a multiple of 256 pixels");
    if (!g_pooled || !act_bits || !x || !gw || !ws) SEMPYR_FAIL("stem wgrad: null pointer");
    if (((uintptr_t)g_pooled & 15)) SEMPYR_FAIL("stem wgrad: the pooled gradient must be 16-byte aligned");
    const long long ntiles = (long long)d->B * (d->Hin / 2) * (d->Win / ST_COLS);
    if (ntiles <= 0 || ntiles >= (1ll << 31)) SEMPYR_FAIL("stem wgrad: bad tile count");
    const int grid = stem_wgrad_grid(ntiles);
    if (ws_bytes < (size_t)grid * SW_OUT * sizeof(float)) SEMPYR_FAIL("stem wgrad: workspace too small (see sempyr_stem_wgrad_workspace)");
    StemWgradParams p;
    p.x = x; p.gp = g_pooled; p.bits = act_bits; p.partials = (float*)ws;
    p.B = d->B; p.H = d->Hin; p.W = d->Win; p.ntiles = (int)ntiles; p.slope = slope;
    hipLaunchKernelGGL(stem3x3_wgrad_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, p);
    SEMPYR_CHECK_LAUNCH();
    hipLaunchKernelGGL(stem_wgrad_reduce_kernel, dim3(SW_OUT / 16), dim3(256), 0, (hipStream_t)stream,
                       (const float*)ws, grid, gw, accumulate_w, gbias, accumulate_b);
    SEMPYR_CHECK_LAUNCH();
    return 0;
}
