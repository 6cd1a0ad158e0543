// Implicit-GEMM convolution for gfx950 on the exact-fp32 matrix instruction
// v_mfma_f32_32x32x2_f32 (64-lane wavefronts, 4 waves per workgroup).
//
// One kernel template serves the three primitives every Conv2d / ConvTranspose2d of the
// reference decomposes into (generator.py:23-25, discriminator.py:37, classifier.py:33-43):
//
//   FWD    C[m=co][n=(b,oy,ox)]  = sum_k Wp[k=(tap,ci)][m] * x[b][ci][oy*s-p+ky][ox*s-p+kx]
//   DGRAD  C[m=ci][n=(b,qy,qx)]  = sum_k Wp[k=(tap',co)][m] * gy[b][co][qy+cy-j][qx+cx-i]   (per stride phase)
//   WGRAD  C[m=co][n=(tap,ci)]   = sum_{k=(b,oy,ox)} gy[b][co][oy][ox] * x[b][ci][oy*s-p+ky][ox*s-p+kx]
//
// The activation operand is gathered straight from the NCHW tensor through a small k-table
// (channel offset + packed dy/dx per reduction index), optionally through a fused nearest-x2
// upsample (coordinate >> 1). Weights are re-laid k-major once per optimizer step so the weight
// tile is a plain 2-D copy. Tiles are staged in LDS as [k][m] / [k][n]; a lane's MFMA operand is
// the dword at [k0 + lane/32][tile + lane%32], i.e. 32 consecutive banks per half-wave: no
// bank conflicts and no swizzle needed for the one-dword-per-lane f32 operands.
// Split-K partial tiles go to caller-provided slabs and are summed in a fixed order
// (deterministic: no float atomics anywhere).
#include "common.h"
#include <map>
#include <mutex>
#include <stdlib.h>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
// The gather table is read-only for the lifetime of a launch.  Reading it through the constant address space makes
// its wave-uniform reads scalar loads; as plain global loads hipcc turns every read after the first workgroup barrier
// into a per-lane vector load (a barrier counts as a possible clobber), i.e. a dependent memory round trip in front of
// the gathers of every k-chunk.
typedef const int __attribute__((address_space(4)))* ctab_ptr;
__device__ __forceinline__ int2 ctab_get(ctab_ptr t, int k) { return make_int2(t[2 * k], t[2 * k + 1]); }

namespace {

#ifndef IGEMM_OCC
#define IGEMM_OCC 3
#endif
constexpr int BK = 16;
constexpr int MODE_GATHER = 0;   // FWD and DGRAD: A = packed weights, B = gathered activations
constexpr int MODE_WGRAD = 1;    // A = gy, B = gathered activations, reduction over pixels

struct FastDiv {
    uint32_t mul, shr, one;
    __host__ void init(uint32_t d) {
        if (d <= 1) { mul = 0; shr = 0; one = 1; return; }
        one = 0;
        int l = 0; while ((1u << l) < d) ++l;           // ceil(log2 d)
        const int p = 31 + l;
        mul = (uint32_t)((((uint64_t)1 << p) + d - 1) / d);
        shr = (uint32_t)(p - 32);
    }
    __device__ __forceinline__ uint32_t div(uint32_t n) const {  // n < 2^31
        return one ? n : (__umulhi(n, mul) >> shr);
    }
};

// GEMM pixel index -> (sample, row, column) of the pixel grid.  `quad` order (used when the 2x2 average pool is fused into
// the epilogue) numbers the four pixels of each pooling window consecutively, so a window sits in 4 adjacent lanes of the
// accumulator layout and its sum is two DPP quad permutes.
__device__ __forceinline__ void decode_pixel(int n, int QHW, int QW, int quad, int& b, int& qy, int& qx) {
    if (quad) {
        const int q = n & 3, t = n >> 2;
        const int Wh = QW >> 1, HWh = QHW >> 2;
        b = t / HWh;
        const int r = t - b * HWh;
        const int yh = r / Wh;
        qy = 2 * yh + (q >> 1);
        qx = 2 * (r - yh * Wh) + (q & 1);
    } else {
        b = n / QHW;
        const int r = n - b * QHW;
        qy = r / QW;
        qx = r - qy * QW;
    }
}

// Tile geometry of the 4x4 / stride-2 patch kernels (conv_tpatch.h)
struct TpGeom {
    int R, CW;             // a tile is SPT samples x R rows x CW columns of the pixel grid (of one stride phase for DGRAD)
    int PR, PW;            // staged source rows / columns per sample
    int PWH, PSS;          // LDS row pitch / per-sample pitch in positions (padded so operand reads are conflict-free)
    int npos;              // staged source positions = SPT * PR * PW
    int slot_row[16];      // first row (channel 0) of each (phase, tap) slot in the k-major packed weights
    int slot_dydx[16];     // source offset of the slot's tap: dy << 16 | (dx & 0xffff)
    FastDiv dRC, dCW, dPRPW, dPW;
};

struct IgemmParams {
    const float* A;        // GATHER: packed weights [Kpad][Mpad]; WGRAD: gy [B][M][QH*QW]
    const float* X;        // gathered tensor
    float* Out;            // output tensor, or slab base when splits > 1 / WGRAD
    const float* scale;    // per-m scale applied before the bias (eval-mode BatchNorm folded into the epilogue), or null
    const float* res;      // residual in the output's layout, added before the activation, or null
    int res_half;          // 1: res has half the output's height and width and is read through a nearest x2 upsample
    const float* bias;     // per-m bias (only applied when splits == 1, GATHER mode)
    const int2* ktab;      // gather table: GATHER -> indexed by k ; WGRAD -> indexed by n
    int M, Mpad, N, K, Kpad;
    int XCHW;              // per-sample element stride of the gathered tensor
    int XW;                // physical row length of the gathered tensor
    int XHW;               // physical plane size of the gathered tensor
    int GH, GW;            // logical gather-space extent (physical << ups)
    int ups;
    int SY, SX;            // gather coordinate = q*S + d
    int QH, QW, QHW;       // pixel grid the GEMM pixel index runs over
    FastDiv dQHW, dQW;
    int OCHW, OHW, OW;     // GATHER output tensor geometry
    int OSY, OSX, OPY, OPX;
    int ldc;               // WGRAD slab row length
    int nchunks, chunks_per_split, nsplits;
    // stride phases of a DGRAD fused into one launch (nphases == 1 otherwise): per-phase overrides of A/ktab/N/K-chunks/
    // pixel grid/output offset
    int nphases;
    struct Phase { long long a_off; int tab_off, N, nchunks, QW, QHW, OPY, OPX; } ph[4];
    long long slab_stride; // elements between split slabs
    int C;                 // channel count of the gathered tensor
    int x_bytes, a_bytes;  // sizes of the X and A operands (buffer-load bounds)
    int act; float slope;  // GATHER epilogue activation (applied with the bias, only when splits == 1)
    int PR, PW, CW;        // patch kernel: patch rows / columns (source pixels), tile width in output pixels
    float* Pool; int quad; // fused AvgPool2d(2) of the (activated) output: pooled tensor, quad pixel order on
    // quad mode only. Bits (out): one bit per output element, 1 = activated value > 0, in quad pixel order
    // [b][m][OHW/32 words] -- all a ReLU-family backward needs of the full-resolution activation (1/32 of its bytes).
    // MaskBits (in): the accumulator is multiplied by act'(bit) before pooling (the double backward of that stem).
    uint32_t* Bits; const uint32_t* MaskBits;
    // rgb3x3 kernel, FUSED: the gathered tensor is not in memory but 0.25 * up2(GP) * act'(GBits): GP is the gradient of the
    // pooled stem output [B][C][GH/2][GW/2], GBits the stem's sign bits (quad order), gslope the LeakyReLU slope
    const float* GP; const unsigned char* GBits; float gslope;
    // patch3x3 kernel, single pass: per (sample, channel) row and group of TN*32 consecutive pixels, (sum, sum of squared
    // deviations from the group's mean) of the values written -- the BatchNorm statistics of the output without a pass over it
    float2* Stats;
    int math;              // SEMPYR_MATH_*: 1 = split-bf16 products where the launch's kernel has the variant
    TpGeom tp;
    int dbg;                      // tuning aid (SEMPYR_TP_DBG): 1 = no staging loads after the first group, 2 = no LDS stores after it
    unsigned long long* stamps;   // tuning aid (SEMPYR_TP_STAMPS): 8 wall-clock stamps per workgroup, or null
};

// UT ("uniform tap"): the channel count of the gathered tensor is a multiple of the k-chunk, so the 16 reduction
// indices of a GATHER chunk (resp. the consecutive columns a WGRAD thread owns) share one filter tap: the bounds test
// and the spatial offset are computed once per chunk and every load is base + per-channel constant. Without it
// (RGB stems, Cin = 3) each element is decoded separately.
template <int BM, int BN, int WM, int WN, int MODE, bool UT, int BKT = 16>
__global__ __launch_bounds__(256, IGEMM_OCC) void igemm_kernel(const IgemmParams p)
{
    // k-chunk per barrier: 16, or 32 for uniform-tap GATHER launches whose channel count allows it (half the barriers and
    // staging round trips per MFMA; the host passes chunk counts in units of BKT)
    constexpr int BK = BKT;
    static_assert(BKT == 16 || (MODE == MODE_GATHER && UT), "the 32-deep chunk is a uniform-tap GATHER form");
    constexpr int TM = BM / (WM * 32), TN = BN / (WN * 32);
    constexpr int PAD = (MODE == MODE_WGRAD) ? (UT ? 1 : 2) : 0;
    constexpr int LDA = BM + PAD, LDB = BN + PAD;
    constexpr int A_ELEMS = BK * LDA, B_ELEMS = BK * LDB;
    __shared__ __attribute__((aligned(16))) float smem[2 * (A_ELEMS + B_ELEMS)];
    float* const As = smem;
    float* const Bs = smem + 2 * A_ELEMS;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;

    // Workgroup -> (phase, tile, split). Workgroups are dealt round-robin over the 8 XCDs (ids b and b+8 share an XCD/L2):
    //  * no split-K: the m-tiles that re-read one activation tile (and the stride phases of a fused DGRAD) get ids 8
    //    apart -> same L2; consecutive n-tiles go to consecutive XCDs.  (Giving each XCD a contiguous band of pixel rows
    //    instead was measured: 1.5% faster on the 256x256 3x3 layers, 20% slower on the store-bound 1x1 / RGB-stem ones.)
    //  * split-K: the split index is the fastest-varying part of the id, so the splits of a tile (which share nothing)
    //    spread over all XCDs.  (With few tiles and the split in blockIdx.y, whole XCDs would receive no work.)
    const int MT = (p.Mpad + BM - 1) / BM;
    int id = blockIdx.x, split = 0, phase = 0, m_tile, n_tile;
    if (p.nsplits > 1) {
        phase = id % p.nphases; id /= p.nphases;
        split = id % p.nsplits; id /= p.nsplits;
        m_tile = id % MT; n_tile = id / MT;
    } else {
        const int xcd = id & 7; id >>= 3;
        phase = id % p.nphases; id /= p.nphases;
        m_tile = id % MT;
        n_tile = (id / MT) * 8 + xcd;
    }
    // per-launch or per-phase geometry
    const float* pA = p.A; const int2* pktab = p.ktab;
    int pN = p.N, pnchunks = p.nchunks, pcps = p.chunks_per_split, pQW = p.QW, pQHW = p.QHW, pOPY = p.OPY, pOPX = p.OPX;
    if constexpr (MODE == MODE_GATHER) {
        if (p.nphases > 1) {
            pA += p.ph[phase].a_off; pktab += p.ph[phase].tab_off;
            pN = p.ph[phase].N; pnchunks = p.ph[phase].nchunks; pQW = p.ph[phase].QW; pQHW = p.ph[phase].QHW;
            pOPY = p.ph[phase].OPY; pOPX = p.ph[phase].OPX;
            pcps = (pnchunks + p.nsplits - 1) / p.nsplits;
        }
    }
    const ctab_ptr ctab = (ctab_ptr)pktab;
    const int m0 = m_tile * BM, n0 = n_tile * BN;
    if (n0 >= pN) return;                                   // uniform for the whole workgroup

    const int c_begin = split * pcps;
    const int c_end = min(c_begin + pcps, pnchunks);

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // ---------------- per-thread load state ----------------
    // GATHER: A tile = BK x BM floats of packed weights (float4 copies); B tile: thread owns pixel column n_local,
    //         reduction rows krow0 + i*KSTEP.
    constexpr int A_V4 = BK * BM / 4;
    constexpr int A_PER = (A_V4 + 255) / 256;
    constexpr int KSTEP = (BN >= 256) ? 1 : 256 / BN;          // BN in {64, 128, 256}
    constexpr int B_PER = BK / KSTEP;
    // WGRAD: thread owns pixel (reduction row) k_local = tid % 16 and WA_PER rows of gy / WB_PER columns of x.
    constexpr int WA_PER = BM / 16, WB_PER = BN / 16;
    // UT WGRAD: a thread's WB_PER columns are NG groups of CPG consecutive columns (one filter tap per group). One group
    // of 16 needs channels % 16 == 0 only; the 192-wide tile (3 x 64 columns: no padding for Cin*9 = 576, 1152) uses 3
    // groups of 4 so that no group straddles a tap boundary.
    constexpr int CPG = (BN == 192) ? 4 : WB_PER, NG = WB_PER / CPG;
    constexpr int RA = (MODE == MODE_GATHER) ? A_PER * 4 : WA_PER;
    constexpr int RB = (MODE == MODE_GATHER) ? B_PER : WB_PER;
    float ra[RA], rb[RB];

    // Every global read of the loop is a buffer load: a 128-bit resource (base, size) in SGPRs + a per-lane unsigned byte
    // offset + a wave-uniform byte offset.  Per load that is one instruction and no address arithmetic (a flat load needs
    // a 64-bit VALU add), and an offset at or beyond the buffer size returns 0 in hardware: padding taps, ragged tile
    // edges and k-tails are expressed as the out-of-range offset OOB instead of selects on the loaded values.
    // (Host side checks that both operands are smaller than 2 GiB.)
    constexpr unsigned OOB = 0x80000000u;
    const __amdgpu_buffer_rsrc_t x_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.X), 0, p.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t a_rs =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(pA), 0, p.a_bytes - (int)((pA - p.A) * 4), 0x00020000);
    auto ld1 = [](__amdgpu_buffer_rsrc_t rs, unsigned voff, int soff) {
        return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, (int)voff, soff, 0));
    };

    // GATHER-mode column decode (fixed for the whole K loop)
    int g_ybase = 0, g_xbase = 0, g_off = 0, g_krow0 = 0;
    // weight tile: one float4 per thread and chunk through a flat pointer advanced by BK rows per chunk (this hipcc
    // lowers the 64/128-bit buffer-load builtins to a single dword, so only the dword gathers are buffer loads)
    const float* a_ptr[(MODE == MODE_GATHER) ? A_PER : 1];
    if constexpr (MODE == MODE_GATHER) {
        const int n_local = tid % BN;
        g_krow0 = __builtin_amdgcn_readfirstlane(tid / BN);
        const int n = n0 + n_local;
        const bool valid = n < pN;
        const int nn = valid ? n : 0;
        int b, qy, qx;
        decode_pixel(nn, pQHW, pQW, p.quad, b, qy, qx);
        g_ybase = valid ? qy * p.SY : -(1 << 20);               // invalid column never passes the bound test
        g_xbase = qx * p.SX;
        g_off = b * p.XCHW;
#pragma unroll
        for (int i = 0; i < A_PER; ++i) {
            const int idx = tid + i * 256;
            const int row = idx / (BM / 4), c4 = idx % (BM / 4);
            a_ptr[i] = pA + (size_t)(c_begin * BK + row) * p.Mpad + m0 + c4 * 4;      // Mpad % BM == 0 in GATHER mode
        }
    }
    // WGRAD-mode column decode: the gather-table entries of this thread's columns are loop-invariant -> registers
    constexpr int NTAB = (MODE == MODE_WGRAD) ? (UT ? NG : WB_PER) : 1;
    int2 w_tab[NTAB];
    unsigned w_rowoob[(MODE == MODE_WGRAD) ? WA_PER : 1];       // OOB for gy rows beyond M, else 0
    const int w_c0 = tid >> 4;
    if constexpr (MODE == MODE_WGRAD) {
#pragma unroll
        for (int i = 0; i < NTAB; ++i) {
            const int n = UT ? (n0 + i * 16 * CPG + w_c0 * CPG) : (n0 + w_c0 + 16 * i);
            w_tab[i] = (n < pN) ? pktab[n] : make_int2(0, (int)0x80008000);  // dy=dx=-32768: always out of range
        }
#pragma unroll
        for (int i = 0; i < WA_PER; ++i) {
            const int m = UT ? (m0 + w_c0 * WA_PER + i) : (m0 + w_c0 + 16 * i);
            w_rowoob[i] = m < p.M ? 0u : OOB;
        }
    }

    int2 e_nxt = make_int2(0, 0);      // UT GATHER: table entry of the next chunk to load, fetched one call ahead
    if constexpr (MODE == MODE_GATHER && UT) e_nxt = ctab_get(ctab, min(c_begin, pnchunks - 1) * BK);

    auto load_tiles = [&](int chunk) {
        const int k0 = chunk * BK;
        if constexpr (MODE == MODE_GATHER) {
#pragma unroll
            for (int i = 0; i < A_PER; ++i) {
                if (tid + i * 256 < A_V4) {                     // compile-time true except BM = 32 (uniform per wave)
                    const float4 v = *reinterpret_cast<const float4*>(a_ptr[i]);
                    ra[i * 4 + 0] = v.x; ra[i * 4 + 1] = v.y; ra[i * 4 + 2] = v.z; ra[i * 4 + 3] = v.w;
                    a_ptr[i] += BK * p.Mpad;
                }
            }
            if constexpr (UT) {
                // uniform: the whole chunk is one tap (e0.y) and its channels are consecutive planes from e0.x
                const int2 e0 = e_nxt;
                e_nxt = ctab_get(ctab, min(chunk + 1, pnchunks - 1) * BK);
                const int iy = g_ybase + (e0.y >> 16);
                const int ix = g_xbase + (int)(short)(e0.y & 0xffff);
                const bool ok = ((unsigned)iy < (unsigned)p.GH) & ((unsigned)ix < (unsigned)p.GW);
                const unsigned voff = ok ? 4u * (unsigned)(g_off + (iy >> p.ups) * p.XW + (ix >> p.ups)) : OOB;
                const int soff = (e0.x + g_krow0 * p.XHW) * 4;
#pragma unroll
                for (int i = 0; i < B_PER; ++i) rb[i] = ld1(x_rs, voff, soff + i * KSTEP * p.XHW * 4);
            } else {
#pragma unroll
                for (int i = 0; i < B_PER; ++i) {
                    const int2 e = ctab_get(ctab, k0 + g_krow0 + i * KSTEP);
                    const int iy = g_ybase + (e.y >> 16);
                    const int ix = g_xbase + (int)(short)(e.y & 0xffff);
                    const bool ok = ((unsigned)iy < (unsigned)p.GH) & ((unsigned)ix < (unsigned)p.GW);
                    rb[i] = ld1(x_rs, ok ? 4u * (unsigned)(g_off + e.x + (iy >> p.ups) * p.XW + (ix >> p.ups)) : OOB, 0);
                }
            }
        } else {
            const int k_local = tid & 15;
            const int k = k0 + k_local;
            const bool kv = k < p.K;
            const uint32_t kk = kv ? (uint32_t)k : 0u;
            const uint32_t b = p.dQHW.div(kk);
            const uint32_t r = kk - b * (uint32_t)pQHW;
            const uint32_t qy = p.dQW.div(r);
            const uint32_t qx = r - qy * (uint32_t)pQW;
            const int ybase = kv ? (int)qy * p.SY : -(1 << 20);
            const int xbase = (int)qx * p.SX;
            const int xoff = (int)b * p.XCHW;
            // A: gy[b][m][r]
            const unsigned a_base = kv ? 4u * (unsigned)(((int)b * p.M + m0 + (UT ? w_c0 * WA_PER : w_c0)) * pQHW + (int)r) : OOB;
#pragma unroll
            for (int i = 0; i < WA_PER; ++i) ra[i] = ld1(a_rs, a_base | w_rowoob[i], (UT ? i : 16 * i) * pQHW * 4);
            if constexpr (UT) {
                // B: x[b][ci0 + i][pixel shifted by this thread's tap]
#pragma unroll
                for (int g = 0; g < NG; ++g) {
                    const int2 e = w_tab[g];
                    const int iy = ybase + (e.y >> 16);
                    const int ix = xbase + (int)(short)(e.y & 0xffff);
                    const bool ok = ((unsigned)iy < (unsigned)p.GH) & ((unsigned)ix < (unsigned)p.GW);
                    const unsigned voff = ok ? 4u * (unsigned)(xoff + e.x + (iy >> p.ups) * p.XW + (ix >> p.ups)) : OOB;
#pragma unroll
                    for (int i = 0; i < CPG; ++i) rb[g * CPG + i] = ld1(x_rs, voff, i * p.XHW * 4);
                }
            } else {
#pragma unroll
                for (int i = 0; i < WB_PER; ++i) {
                    const int2 e = w_tab[i];
                    const int iy = ybase + (e.y >> 16);
                    const int ix = xbase + (int)(short)(e.y & 0xffff);
                    const bool ok = ((unsigned)iy < (unsigned)p.GH) & ((unsigned)ix < (unsigned)p.GW);
                    rb[i] = ld1(x_rs, ok ? 4u * (unsigned)(xoff + e.x + (iy >> p.ups) * p.XW + (ix >> p.ups)) : OOB, 0);
                }
            }
        }
    };

    auto store_tiles = [&](int buf) {
        float* as = As + buf * A_ELEMS;
        float* bs = Bs + buf * B_ELEMS;
        if constexpr (MODE == MODE_GATHER) {
#pragma unroll
            for (int i = 0; i < A_PER; ++i) {
                const int idx = tid + i * 256;
                if (idx < A_V4) {
                    const int row = idx / (BM / 4);
                    const int c4 = idx % (BM / 4);
                    *reinterpret_cast<float4*>(as + row * LDA + c4 * 4) =
                        make_float4(ra[i * 4 + 0], ra[i * 4 + 1], ra[i * 4 + 2], ra[i * 4 + 3]);
                }
            }
            const int n_local = tid % BN;
#pragma unroll
            for (int i = 0; i < B_PER; ++i) bs[(g_krow0 + i * KSTEP) * LDB + n_local] = rb[i];
        } else {
            const int k_local = tid & 15;
#pragma unroll
            for (int i = 0; i < WA_PER; ++i) as[k_local * LDA + (UT ? w_c0 * WA_PER + i : w_c0 + 16 * i)] = ra[i];
#pragma unroll
            for (int i = 0; i < WB_PER; ++i)
                bs[k_local * LDB + (UT ? (i / CPG) * 16 * CPG + w_c0 * CPG + (i % CPG) : w_c0 + 16 * i)] = rb[i];
        }
    };

    // ---------------- main loop: register-staged double buffer, one barrier per k-chunk ----------------
    if (c_begin < c_end) {
        load_tiles(c_begin);
        store_tiles(0);
        __syncthreads();
        const int kh = lane >> 5, l31 = lane & 31;
        for (int c = c_begin; c < c_end; ++c) {
            const int buf = (c - c_begin) & 1;
            const bool more = (c + 1) < c_end;
            // Nothing is in flight here (store_tiles consumed every load before the barrier). Saying so explicitly stops
            // hipcc from guarding the loop-carried staging registers with conservative vmcnt waits BETWEEN this chunk's
            // loads, which made the gather loads wait for the weight-tile load that had just been issued.
            __builtin_amdgcn_s_waitcnt(0x0F70);                // vmcnt(0) only
            if (more) load_tiles(c + 1);                       // global loads in flight under the MFMAs
            __builtin_amdgcn_sched_barrier(0);
            const float* as = As + buf * A_ELEMS + kh * LDA + wm * (TM * 32) + l31;
            const float* bs = Bs + buf * B_ELEMS + kh * LDB + wn * (TN * 32) + l31;
            // operands of k-step ks+1 are read from LDS before the MFMAs of k-step ks are issued (two register sets), so
            // the LDS latency sits under 4 MFMAs instead of between them
            float a[2][TM], b[2][TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) a[0][i] = as[i * 32];
#pragma unroll
            for (int j = 0; j < TN; ++j) b[0][j] = bs[j * 32];
#pragma unroll
            for (int ks = 0; ks < BK / 2; ++ks) {
                const int cur = ks & 1, nxt = cur ^ 1;
                if (ks + 1 < BK / 2) {
#pragma unroll
                    for (int i = 0; i < TM; ++i) a[nxt][i] = as[(ks + 1) * 2 * LDA + i * 32];
#pragma unroll
                    for (int j = 0; j < TN; ++j) b[nxt][j] = bs[(ks + 1) * 2 * LDB + j * 32];
                }
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[cur][i], b[cur][j], acc[i][j], 0, 0, 0);
                if (ks + 1 < BK / 2) {
                    __builtin_amdgcn_sched_group_barrier(0x100, TM + TN, 0);     // DS reads of the next k-step first ...
                    __builtin_amdgcn_sched_group_barrier(0x008, TM * TN, 0);     // ... then this k-step's MFMAs
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            if (more) store_tiles(buf ^ 1);
            __syncthreads();
        }
    }

    // ---------------- epilogue ----------------
    // accumulator register r of lane l: column n = l%32, row m = (r&3) + 8*(r>>2) + 4*(l/32)
    float* out = p.Out + (size_t)split * p.slab_stride;
    const int l31 = lane & 31, kh = lane >> 5;
    const bool full_m = (m0 + BM) <= p.M;                        // workgroup-uniform: no per-row guard needed
    if constexpr (MODE == MODE_GATHER) {
        if (p.scale) {                                           // uniform: y = act(scale[m] * conv + bias[m] + res)
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                float sv[16];
#pragma unroll
                for (int r16 = 0; r16 < 16; ++r16) {
                    const int m = m0 + wm * (TM * 32) + i * 32 + (r16 & 3) + 8 * (r16 >> 2) + 4 * kh;
                    sv[r16] = p.scale[m < p.M ? m : 0];
                }
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int r16 = 0; r16 < 16; ++r16) acc[i][j][r16] *= sv[r16];
            }
        }
        if (p.bias) {                                            // uniform; all bias loads issued before any use
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                float bv[16];
#pragma unroll
                for (int r16 = 0; r16 < 16; ++r16) {
                    const int m = m0 + wm * (TM * 32) + i * 32 + (r16 & 3) + 8 * (r16 >> 2) + 4 * kh;
                    bv[r16] = p.bias[m < p.M ? m : 0];
                }
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int r16 = 0; r16 < 16; ++r16) acc[i][j][r16] += bv[r16];
            }
        }
        if (p.res) {                                             // uniform
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int n = n0 + wn * (TN * 32) + j * 32 + l31;
                if (n >= pN) continue;
                int b, qy, qx;
                decode_pixel(n, pQHW, pQW, p.quad, b, qy, qx);
                const float* rp = p.res_half ? p.res + (size_t)b * (p.OCHW >> 2) + (qy >> 1) * (p.OW >> 1) + (qx >> 1)
                                             : p.res + (size_t)b * p.OCHW + (qy * p.OSY + pOPY) * p.OW + qx * p.OSX + pOPX;
                const int rstride = p.res_half ? (p.OHW >> 2) : p.OHW;
#pragma unroll
                for (int i = 0; i < TM; ++i) {
                    const int mb = m0 + wm * (TM * 32) + i * 32 + 4 * kh;
#pragma unroll
                    for (int r16 = 0; r16 < 16; ++r16) {
                        const int m = mb + (r16 & 3) + 8 * (r16 >> 2);
                        acc[i][j][r16] += rp[(size_t)(m < p.M ? m : 0) * rstride];
                    }
                }
            }
        }
        if (p.act != SEMPYR_ACT_NONE) {                          // uniform
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int r16 = 0; r16 < 16; ++r16) acc[i][j][r16] = act_apply(acc[i][j][r16], p.act, p.slope);
        }
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int n = n0 + wn * (TN * 32) + j * 32 + l31;
            if (n >= pN) continue;
            int b, qy, qx;
            decode_pixel(n, pQHW, pQW, p.quad, b, qy, qx);
            float* op = out + (size_t)b * p.OCHW + (qy * p.OSY + pOPY) * p.OW + qx * p.OSX + pOPX;
            if (p.quad) {                                       // uniform: 2x2 window = lanes 4q..4q+3 (N % 4 == 0)
                float* pp = p.Pool + (size_t)b * (p.OCHW >> 2) + (qy >> 1) * (p.OW >> 1) + (qx >> 1);
                const bool lead = (n & 3) == 0;
                const int wpp = p.OHW >> 5;                     // bit words per plane; this half-wave's 32 pixels are one word
                const size_t wbase = (size_t)b * p.M * wpp + ((n - b * pQHW) >> 5);
#pragma unroll
                for (int i = 0; i < TM; ++i) {
                    const int mb = m0 + wm * (TM * 32) + i * 32 + 4 * kh;
#pragma unroll
                    for (int r16 = 0; r16 < 16; ++r16) {
                        const int mrow = mb + (r16 & 3) + 8 * (r16 >> 2);
                        if (p.Bits) {                           // uniform
                            const unsigned long long bal = __builtin_amdgcn_ballot_w64(acc[i][j][r16] > 0.f);
                            if (l31 == 0 && mrow < p.M) p.Bits[wbase + (size_t)mrow * wpp] = kh ? (uint32_t)(bal >> 32) : (uint32_t)bal;
                        }
                        if (p.MaskBits) {                       // uniform
                            const uint32_t wd = p.MaskBits[wbase + (size_t)(mrow < p.M ? mrow : 0) * wpp];
                            acc[i][j][r16] *= ((wd >> l31) & 1u) ? 1.f : p.slope;
                        }
                        float v = acc[i][j][r16];
                        v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xf, 0xf, true));
                        v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xf, 0xf, true));
                        const int m = mb + (r16 & 3) + 8 * (r16 >> 2);
                        if (lead && m < p.M) pp[(size_t)m * (p.OHW >> 2)] = 0.25f * v;
                    }
                }
            }
            if (!p.Out) continue;                               // uniform: pooled output (+ sign bits) only
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const int mb = m0 + wm * (TM * 32) + i * 32 + 4 * kh;
                if (full_m) {
#pragma unroll
                    for (int r16 = 0; r16 < 16; ++r16)
                        op[(size_t)(mb + (r16 & 3) + 8 * (r16 >> 2)) * p.OHW] = acc[i][j][r16];
                } else {
#pragma unroll
                    for (int r16 = 0; r16 < 16; ++r16) {
                        const int m = mb + (r16 & 3) + 8 * (r16 >> 2);
                        if (m < p.M) op[(size_t)m * p.OHW] = acc[i][j][r16];
                    }
                }
            }
        }
    } else {
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int n = n0 + wn * (TN * 32) + j * 32 + l31;
            if (n >= pN) continue;
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const int mb = m0 + wm * (TM * 32) + i * 32 + 4 * kh;
                if (full_m) {
#pragma unroll
                    for (int r16 = 0; r16 < 16; ++r16)
                        out[(size_t)(mb + (r16 & 3) + 8 * (r16 >> 2)) * p.ldc + n] = acc[i][j][r16];
                } else {
#pragma unroll
                    for (int r16 = 0; r16 < 16; ++r16) {
                        const int m = mb + (r16 & 3) + 8 * (r16 >> 2);
                        if (m < p.M) out[(size_t)m * p.ldc + n] = acc[i][j][r16];
                    }
                }
            }
        }
    }
}

// Epilogue of the patch kernels (as the generic kernel's GATHER epilogue): eval-BatchNorm scale, bias, residual, activation,
// BatchNorm-statistics partials, store.  acc: the wave's TM x TN accumulator tiles of the 32x32 MFMA layout.
template <int BM, int BN, int WM, int WN>
__device__ __forceinline__ void patch_epilogue(const IgemmParams& p, f32x16 (&acc)[BM / (WM * 32)][BN / (WN * 32)], int m0, int tb,
                                               int gy0, int gx0, int split, int wm, int wn, int l31, int kh)
{
    constexpr int TM = BM / (WM * 32), TN = BN / (WN * 32);
    float* out = p.Out + (size_t)split * p.slab_stride;
    const bool full_m = (m0 + BM) <= p.M;
    if (p.scale) {
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            float sv[16];
#pragma unroll
            for (int r16 = 0; r16 < 16; ++r16) {
                const int m = m0 + wm * (TM * 32) + i * 32 + (r16 & 3) + 8 * (r16 >> 2) + 4 * kh;
                sv[r16] = p.scale[m < p.M ? m : 0];
            }
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int r16 = 0; r16 < 16; ++r16) acc[i][j][r16] *= sv[r16];
        }
    }
    if (p.bias) {
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            float bv[16];
#pragma unroll
            for (int r16 = 0; r16 < 16; ++r16) {
                const int m = m0 + wm * (TM * 32) + i * 32 + (r16 & 3) + 8 * (r16 >> 2) + 4 * kh;
                bv[r16] = p.bias[m < p.M ? m : 0];
            }
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int r16 = 0; r16 < 16; ++r16) acc[i][j][r16] += bv[r16];
        }
    }
    if (p.res) {
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int nl = wn * (TN * 32) + j * 32 + l31;
            const int ry = gy0 + nl / p.CW, rx = gx0 + nl % p.CW, rr = ry * p.QW + rx;
            const float* rp = p.res_half ? p.res + (size_t)tb * (p.OCHW >> 2) + (ry >> 1) * (p.OW >> 1) + (rx >> 1)
                                         : p.res + (size_t)tb * p.OCHW + rr;
            const int rstride = p.res_half ? (p.OHW >> 2) : p.OHW;
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const int mb = m0 + wm * (TM * 32) + i * 32 + 4 * kh;
#pragma unroll
                for (int r16 = 0; r16 < 16; ++r16) {
                    const int m = mb + (r16 & 3) + 8 * (r16 >> 2);
                    acc[i][j][r16] += rp[(size_t)(m < p.M ? m : 0) * rstride];
                }
            }
        }
    }
    if (p.act != SEMPYR_ACT_NONE) {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int r16 = 0; r16 < 16; ++r16) acc[i][j][r16] = act_apply(acc[i][j][r16], p.act, p.slope);
    }
    if (p.Stats) {
        // Two passes over the accumulators (sum, then squared deviations from the group's own mean: no cancellation), each
        // closed by a 32-lane butterfly: quad permutes, the two row mirrors and a swizzle across the 16-lane rows.
        constexpr int GR = TN * 32;
        const int G = p.QHW / GR;
        // the wave's TN*32 columns are GR consecutive pixels of one row of the block (CW % GR == 0)
        const int nl0 = wn * GR;
        const int g = ((gy0 + nl0 / p.CW) * p.QW + gx0 + nl0 % p.CW) / GR;
        auto sum32 = [](float v) {
            v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xf, 0xf, true));
            v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xf, 0xf, true));
            v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xf, 0xf, true));
            v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xf, 0xf, true));
            v += __builtin_bit_cast(float, __builtin_amdgcn_ds_swizzle(__builtin_bit_cast(int, v), 0x401F));   // lane ^ 16
            return v;
        };
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int mb = m0 + wm * (TM * 32) + i * 32 + 4 * kh;
#pragma unroll
            for (int r16 = 0; r16 < 16; ++r16) {
                float sm = acc[i][0][r16];
#pragma unroll
                for (int j = 1; j < TN; ++j) sm += acc[i][j][r16];
                sm = sum32(sm);
                const float mu = sm * (1.f / GR);
                float q = 0.f;
#pragma unroll
                for (int j = 0; j < TN; ++j) { const float dv = acc[i][j][r16] - mu; q = fmaf(dv, dv, q); }
                q = sum32(q);
                const int m = mb + (r16 & 3) + 8 * (r16 >> 2);
                if (l31 == 0 && m < p.M) p.Stats[((size_t)tb * p.M + m) * G + g] = make_float2(sm, q);
            }
        }
    }
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int nl = wn * (TN * 32) + j * 32 + l31;
        const int r = (gy0 + nl / p.CW) * p.QW + gx0 + nl % p.CW;        // the tile lies inside sample tb
        float* op = out + (size_t)tb * p.OCHW + r;           // stride-1 output: OHW == QHW, pixel r of the plane
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int mb = m0 + wm * (TM * 32) + i * 32 + 4 * kh;
            if (full_m) {
#pragma unroll
                for (int r16 = 0; r16 < 16; ++r16) op[(size_t)(mb + (r16 & 3) + 8 * (r16 >> 2)) * p.OHW] = acc[i][j][r16];
            } else {
#pragma unroll
                for (int r16 = 0; r16 < 16; ++r16) {
                    const int m = mb + (r16 & 3) + 8 * (r16 >> 2);
                    if (m < p.M) op[(size_t)m * p.OHW] = acc[i][j][r16];
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------------------------
// 3x3 / stride 1 / pad 1 convolutions (FWD incl. fused x2 upsample, and their DGRAD): the input patch of a tile is staged
// in LDS ONCE per group of PCG channels and the nine taps read it as shifted views, instead of gathering a fresh
// [16 x BN] operand tile per (tap, 16 channels).  A tile is BN consecutive output pixels = R full image rows (or a row
// segment), so its patch is (R + 2) x (CW + 2) source pixels ((R/2 + 2) x (CW/2 + 2) through the upsample); per 8 channels
// a thread issues ~24 gathers (8 with the upsample) where the generic kernel issues 72, and as many fewer LDS stores.
// MFMA operand reads stay conflict-free: the 32 lanes of an operand read consecutive patch dwords (pairs of lanes the
// same dword through the upsample = broadcast).  Per-lane patch offsets of the 9 taps are computed once per tile.
// ------------------------------------------------------------------------------------------------------------------
constexpr int PCG = 8;                                      // channels per staged patch

template <int BM, int BN, int WM, int WN, int NSP>
__global__ __launch_bounds__(256, IGEMM_OCC) void patch3x3_kernel(const IgemmParams p)
{
    constexpr int TM = BM / (WM * 32), TN = BN / (WN * 32);
    constexpr int A_FLOATS = 9 * PCG * BM;
    constexpr int A_V4 = A_FLOATS / 4, A_PER = (A_V4 + 255) / 256;
    extern __shared__ __attribute__((aligned(16))) float psm[];
    float* const As = psm;                                  // [tap][channel in group][m]
    float* const Ps = psm + A_FLOATS;                       // [channel in group][PITCH >= patch rows x cols]
    constexpr int PITCH = NSP * 256;                        // compile-time channel pitch: k-step offsets fold into the
                                                            // ds_read immediate

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int l31 = lane & 31, kh = lane >> 5;

    const int MT = p.Mpad / BM;
    int id = blockIdx.x, split = 0, m_tile, n_tile;
    if (p.nsplits > 1) {
        split = id % p.nsplits; id /= p.nsplits;
        m_tile = id % MT; n_tile = id / MT;
    } else {
        const int xcd = id & 7; id >>= 3;
        m_tile = id % MT;
        n_tile = (id / MT) * 8 + xcd;
    }
    const int m0 = m_tile * BM, n0 = n_tile * BN;
    if (n0 >= p.N) return;                                  // uniform (N % BN == 0: no ragged tile)

    // tile geometry (uniform): first output pixel, patch origin in source coordinates
    const int tb = n0 / p.QHW;
    // the tile is a block of BN / CW rows x CW columns of sample tb (CW == QW: whole rows; CW == BN: a row segment; in
    // between: a 2-D block, whose input patch -- (R+2) x (CW+2) -- is the smaller the squarer it is: 4 x 64 pixels stage
    // 396 source pixels per channel where 1 x 256 stages 774 and re-reads every input row three times)
    const int tt = (n0 - tb * p.QHW) / BN;                  // tile index inside the sample
    const int tpr = p.QW / p.CW;                            // tiles per row of blocks
    const int trow = tt / tpr, tcol = tt - trow * tpr;
    const int gy0 = trow * (BN / p.CW), gx0 = tcol * p.CW;
    const int s = p.ups;
    const int pr0 = (gy0 - 1) >> s, pc0 = (gx0 - 1) >> s;
    const int PW = p.PW, PRPW = p.PR * p.PW;
    const int XH = p.GH >> s;
    const ctab_ptr ctab = (ctab_ptr)p.ktab;
    const int C = p.C;

    // per-lane patch offsets (floats, < 2^16) of the 9 taps for the lane's TN operand columns (+ the k-half of the
    // lane), two columns per register
    static_assert(TN <= 2, "tap offsets are packed two per register");
    unsigned off[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) {
        const int pk = ctab[2 * (t * C) + 1];               // packed dy/dx of tap t (uniform)
        const int dy = pk >> 16, dx = (int)(short)(pk & 0xffff);
        unsigned packed = 0u;
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int nl = wn * (TN * 32) + j * 32 + l31;
            const int r = nl / p.CW, c = nl - r * p.CW;
            const int o = (((gy0 + r + dy) >> s) - pr0) * PW + (((gx0 + c + dx) >> s) - pc0) + kh * PITCH;
            packed |= (unsigned)o << (16 * j);
        }
        off[t] = packed;
    }

    // patch loads: thread owns NSP spatial patch positions, every channel of the group is one SGPR offset further
    constexpr unsigned OOB = 0x80000000u;
    const __amdgpu_buffer_rsrc_t x_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.X), 0, p.x_bytes, 0x00020000);
    unsigned pvoff[NSP];
#pragma unroll
    for (int i = 0; i < NSP; ++i) {
        const int pos = tid + 256 * i;
        const int pr = pos / PW, pc = pos - pr * PW;
        const int sy = pr0 + pr, sx = pc0 + pc;
        const bool ok = pos < PRPW && (unsigned)sy < (unsigned)XH && (unsigned)sx < (unsigned)p.XW;   // else: zero-filled
        pvoff[i] = ok ? 4u * (unsigned)(tb * p.XCHW + sy * p.XW + sx) : OOB;
    }
    int a_off[A_PER];
#pragma unroll
    for (int i = 0; i < A_PER; ++i) {
        const int idx = tid + i * 256;
        const int idc = idx < A_V4 ? idx : idx - A_V4;      // the surplus threads of the last slot re-read slot 0's data
        const int t = idc / (PCG * BM / 4), rem = idc - t * (PCG * BM / 4);
        const int kk = rem / (BM / 4), c4 = rem - kk * (BM / 4);
        a_off[i] = (t * C + kk) * p.Mpad + m0 + c4 * 4;
    }

    const int ngroups = C / PCG;
    const int gps = (ngroups + p.nsplits - 1) / p.nsplits;
    const int g_begin = split * gps, g_end = min(g_begin + gps, ngroups);

    float rp[NSP][PCG];
    f32x4 ra[A_PER];
    auto load_group = [&](int g) {
        const float* ab = p.A + (size_t)g * PCG * p.Mpad;
#pragma unroll
        for (int i = 0; i < A_PER; ++i) ra[i] = *reinterpret_cast<const f32x4*>(ab + a_off[i]);
        const int cb = g * PCG * p.XHW * 4;
#pragma unroll
        for (int i = 0; i < NSP; ++i)
#pragma unroll
            for (int ci = 0; ci < PCG; ++ci)
                rp[i][ci] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(x_rs, (int)pvoff[i], cb + ci * p.XHW * 4, 0));
    };
    auto store_group = [&]() {
#pragma unroll
        for (int i = 0; i < A_PER; ++i)
            if (tid + i * 256 < A_V4) *reinterpret_cast<f32x4*>(As + (tid + i * 256) * 4) = ra[i];
#pragma unroll
        for (int i = 0; i < NSP; ++i) {
            const int pos = tid + 256 * i;
            if (pos < PRPW) {
#pragma unroll
                for (int ci = 0; ci < PCG; ++ci) Ps[ci * PITCH + pos] = rp[i][ci];
            }
        }
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    if (g_begin < g_end) {
        load_group(g_begin);
        store_group();
        __syncthreads();
        const float* as = As + kh * BM + wm * (TM * 32) + l31;
        for (int g = g_begin; g < g_end; ++g) {
            const bool more = (g + 1) < g_end;
            if (more) load_group(g + 1);                    // in flight under the 144 MFMAs of this group
            __builtin_amdgcn_sched_barrier(0);
            // 36 k-steps (9 taps x 4); the operands of step st+1 are read before the MFMAs of step st issue
            constexpr int NST = 9 * (PCG / 2);
            float a[2][TM], b[2][TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) a[0][i] = as[i * 32];
#pragma unroll
            for (int j = 0; j < TN; ++j) b[0][j] = Ps[(off[0] >> (16 * j)) & 0xffffu];
#pragma unroll
            for (int st = 0; st < NST; ++st) {
                const int cur = st & 1, nxt = cur ^ 1;
                if (st + 1 < NST) {
                    const int t = (st + 1) / (PCG / 2), ks = (st + 1) % (PCG / 2);
#pragma unroll
                    for (int i = 0; i < TM; ++i) a[nxt][i] = as[(t * PCG + ks * 2) * BM + i * 32];
#pragma unroll
                    for (int j = 0; j < TN; ++j) b[nxt][j] = Ps[((off[t] >> (16 * j)) & 0xffffu) + ks * 2 * PITCH];
                }
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[cur][i], b[cur][j], acc[i][j], 0, 0, 0);
                if (st + 1 < NST) {
                    __builtin_amdgcn_sched_group_barrier(0x100, TM + TN, 0);
                    __builtin_amdgcn_sched_group_barrier(0x008, TM * TN, 0);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            __syncthreads();                                // every wave is done reading this group's patch
            if (more) { store_group(); __syncthreads(); }
        }
    }

    patch_epilogue<BM, BN, WM, WN>(p, acc, m0, tb, gy0, gx0, split, wm, wn, l31, kh);
}

// ------------------------------------------------------------------------------------------------------------------
// The same convolutions on the bf16 matrix pipe (sempyr_conv_desc.math = SEMPYR_MATH_BF16X3, opt-in): every fp32 operand is
// split into two bf16 values, x = hi + lo with hi = bf16(x) and lo = bf16(x - hi) (16 significant bits together), and a
// product is formed as hi*hi + hi*lo + lo*hi on v_mfma_f32_32x32x16_bf16 with fp32 accumulation -- three bf16 MFMAs of
// 32 cycles each per 16 reduction indices where the fp32 instruction needs eight of 64 cycles (5.3x fewer matrix-pipe
// cycles); the dropped lo*lo term and the split's own rounding leave a relative error of ~2^-17 per product.
// Staging: 16 channels per group.  The split happens ONCE per staged element (the 9 taps reuse it): patch positions and
// weight rows are written to LDS as 16-byte vectors of 8 consecutive channels, [k-half][position] resp. [tap][k-half][m], hi
// and lo images apart, so an MFMA operand is one ds_read_b128 and 16 lanes of a read group touch 64 distinct banks.
// ------------------------------------------------------------------------------------------------------------------
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
constexpr int BCG = 16;                                     // channels per staged group (one MFMA reduction block)

__device__ __forceinline__ void split_bf16x8(const float (&x)[8], bf16x8& hi, bf16x8& lo) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const __bf16 h = (__bf16)x[j];
        hi[j] = h;
        lo[j] = (__bf16)(x[j] - (float)h);
    }
}

template <int BM, int BN, int WM, int WN>
__global__ __launch_bounds__(256, 2) void patch3x3_bf16x3_kernel(const IgemmParams p)
{
    constexpr int TM = BM / (WM * 32), TN = BN / (WN * 32);
    constexpr int NSP = 2, PITCH = NSP * 256;               // patch positions per channel-half image
    constexpr int A_ITEMS = 9 * 2 * BM;                     // (tap, k-half, m) vectors of 8 channels
    constexpr int A_PER = (A_ITEMS + 255) / 256;
    extern __shared__ __attribute__((aligned(16))) float psm[];
    bf16x8* const Ahi = reinterpret_cast<bf16x8*>(psm);     // [tap][half][m]
    bf16x8* const Alo = Ahi + A_ITEMS;
    bf16x8* const Phi = Alo + A_ITEMS;                      // [half][position]
    bf16x8* const Plo = Phi + 2 * PITCH;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int l31 = lane & 31, kh = lane >> 5;

    const int MT = p.Mpad / BM;
    int id = blockIdx.x, split = 0, m_tile, n_tile;
    if (p.nsplits > 1) {
        split = id % p.nsplits; id /= p.nsplits;
        m_tile = id % MT; n_tile = id / MT;
    } else {
        const int xcd = id & 7; id >>= 3;
        m_tile = id % MT;
        n_tile = (id / MT) * 8 + xcd;
    }
    const int m0 = m_tile * BM, n0 = n_tile * BN;
    if (n0 >= p.N) return;                                  // uniform (N % BN == 0: no ragged tile)

    // tile geometry: exactly patch3x3_kernel's
    const int tb = n0 / p.QHW;
    const int tt = (n0 - tb * p.QHW) / BN;
    const int tpr = p.QW / p.CW;
    const int trow = tt / tpr, tcol = tt - trow * tpr;
    const int gy0 = trow * (BN / p.CW), gx0 = tcol * p.CW;
    const int s = p.ups;
    const int pr0 = (gy0 - 1) >> s, pc0 = (gx0 - 1) >> s;
    const int PW = p.PW, PRPW = p.PR * p.PW;
    const int XH = p.GH >> s;
    const ctab_ptr ctab = (ctab_ptr)p.ktab;
    const int C = p.C;

    // per-lane patch positions (+ the lane's k-half image) of the 9 taps for the lane's TN operand columns, two per register
    static_assert(TN <= 2, "tap offsets are packed two per register");
    unsigned off[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) {
        const int pk = ctab[2 * (t * C) + 1];
        const int dy = pk >> 16, dx = (int)(short)(pk & 0xffff);
        unsigned packed = 0u;
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int nl = wn * (TN * 32) + j * 32 + l31;
            const int r = nl / p.CW, c = nl - r * p.CW;
            const int o = (((gy0 + r + dy) >> s) - pr0) * PW + (((gx0 + c + dx) >> s) - pc0) + kh * PITCH;
            packed |= (unsigned)o << (16 * j);
        }
        off[t] = packed;
    }

    constexpr unsigned OOB = 0x80000000u;
    const __amdgpu_buffer_rsrc_t x_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.X), 0, p.x_bytes, 0x00020000);
    unsigned pvoff[NSP];
#pragma unroll
    for (int i = 0; i < NSP; ++i) {
        const int pos = tid + 256 * i;
        const int pr = pos / PW, pc = pos - pr * PW;
        const int sy = pr0 + pr, sx = pc0 + pc;
        const bool ok = pos < PRPW && (unsigned)sy < (unsigned)XH && (unsigned)sx < (unsigned)p.XW;   // else: zero-filled
        pvoff[i] = ok ? 4u * (unsigned)(tb * p.XCHW + sy * p.XW + sx) : OOB;
    }
    // weight items: item = tid + 256 r -> (tap, half, m); the 8 channels of an item are 8 rows of the k-major packed weights
    int a_off[A_PER];
#pragma unroll
    for (int i = 0; i < A_PER; ++i) {
        const int item = tid + i * 256;
        const int it = item < A_ITEMS ? item : item - A_ITEMS;      // surplus threads of the last round re-read round 0's data
        const int th = it / BM, m = it - th * BM;
        a_off[i] = ((th >> 1) * C + (th & 1) * 8) * p.Mpad + m0 + m;
    }

    const int ngroups = C / BCG;
    const int gps = (ngroups + p.nsplits - 1) / p.nsplits;
    const int g_begin = split * gps, g_end = min(g_begin + gps, ngroups);

    float rp[NSP][BCG];
    float ra[A_PER][8];
    auto load_group = [&](int g) {
        const float* ab = p.A + (size_t)g * BCG * p.Mpad;
#pragma unroll
        for (int i = 0; i < A_PER; ++i)
#pragma unroll
            for (int j = 0; j < 8; ++j) ra[i][j] = ab[a_off[i] + j * p.Mpad];
        const int cb = g * BCG * p.XHW * 4;
#pragma unroll
        for (int i = 0; i < NSP; ++i)
#pragma unroll
            for (int ci = 0; ci < BCG; ++ci)
                rp[i][ci] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(x_rs, (int)pvoff[i], cb + ci * p.XHW * 4, 0));
    };
    auto store_group = [&]() {
#pragma unroll
        for (int i = 0; i < A_PER; ++i) {
            const int item = tid + i * 256;
            if (item < A_ITEMS) {
                bf16x8 h, l;
                split_bf16x8(ra[i], h, l);
                Ahi[item] = h; Alo[item] = l;
            }
        }
#pragma unroll
        for (int i = 0; i < NSP; ++i) {
            const int pos = tid + 256 * i;
            if (pos < PRPW) {
#pragma unroll
                for (int hf = 0; hf < 2; ++hf) {
                    float x8[8];
#pragma unroll
                    for (int j = 0; j < 8; ++j) x8[j] = rp[i][hf * 8 + j];
                    bf16x8 h, l;
                    split_bf16x8(x8, h, l);
                    Phi[hf * PITCH + pos] = h; Plo[hf * PITCH + pos] = l;
                }
            }
        }
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    if (g_begin < g_end) {
        load_group(g_begin);
        store_group();
        __syncthreads();
        const int arow = kh * BM + wm * (TM * 32) + l31;    // + tap * 2 * BM + i * 32
        for (int g = g_begin; g < g_end; ++g) {
            const bool more = (g + 1) < g_end;
            if (more) load_group(g + 1);                    // in flight under the 9 x 3 x TM x TN MFMAs of this group
            __builtin_amdgcn_sched_barrier(0);
            bf16x8 ah[2][TM], al[2][TM], bh[2][TN], bl[2][TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) { ah[0][i] = Ahi[arow + i * 32]; al[0][i] = Alo[arow + i * 32]; }
#pragma unroll
            for (int j = 0; j < TN; ++j) { const int o = (off[0] >> (16 * j)) & 0xffffu; bh[0][j] = Phi[o]; bl[0][j] = Plo[o]; }
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                const int cur = t & 1, nxt = cur ^ 1;
                if (t + 1 < 9) {
#pragma unroll
                    for (int i = 0; i < TM; ++i) {
                        ah[nxt][i] = Ahi[(t + 1) * 2 * BM + arow + i * 32]; al[nxt][i] = Alo[(t + 1) * 2 * BM + arow + i * 32];
                    }
#pragma unroll
                    for (int j = 0; j < TN; ++j) {
                        const int o = (off[t + 1] >> (16 * j)) & 0xffffu;
                        bh[nxt][j] = Phi[o]; bl[nxt][j] = Plo[o];
                    }
                }
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j) {
                        // small terms first, then the leading product
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[cur][i], bh[cur][j], acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[cur][i], bl[cur][j], acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[cur][i], bh[cur][j], acc[i][j], 0, 0, 0);
                    }
            }
            __builtin_amdgcn_sched_barrier(0);
            __syncthreads();                                // every wave is done reading this group's images
            if (more) { store_group(); __syncthreads(); }
        }
    }
    patch_epilogue<BM, BN, WM, WN>(p, acc, m0, tb, gy0, gx0, split, wm, wn, l31, kh);
}

#include "conv_tpatch.h"

// GATHER-mode convolution with at most 4 output rows (RGB image heads: generator last conv, critic/classifier stem
// input gradients).  Padding M = 3 to a 32-row MFMA tile wastes 10x the matrix work, so this shape runs on the vector ALU:
// one output pixel per lane (64 consecutive pixels per wave -> 256-byte coalesced gathers), weights and k-table entries
// are wave-uniform scalar loads, 4 accumulators per lane.  HBM/L1-bound: each input element is read once per tap.
template <bool UT>
__global__ __launch_bounds__(256) void gather_small_m_kernel(const IgemmParams p)
{
    int id = blockIdx.x;
    const float* pA = p.A; const int2* pktab = p.ktab;
    int pN = p.N, pnchunks = p.nchunks, pQW = p.QW, pQHW = p.QHW, pOPY = p.OPY, pOPX = p.OPX;
    if (p.nphases > 1) {
        const int phase = id % p.nphases; id /= p.nphases;
        pA += p.ph[phase].a_off; pktab += p.ph[phase].tab_off;
        pN = p.ph[phase].N; pnchunks = p.ph[phase].nchunks; pQW = p.ph[phase].QW; pQHW = p.ph[phase].QHW;
        pOPY = p.ph[phase].OPY; pOPX = p.ph[phase].OPX;
    }
    if (id * 256 >= pN) return;
    const int n = id * 256 + threadIdx.x;
    const bool valid = n < pN;
    const int nn = valid ? n : 0;
    const int b = nn / pQHW;
    const int r = nn - b * pQHW;
    const int qy = r / pQW;
    const int qx = r - qy * pQW;
    const int ybase = valid ? qy * p.SY : -(1 << 20);
    const int xbase = qx * p.SX;
    const int xoff = b * p.XCHW;
    float acc0 = 0.f, acc1 = 0.f, acc2 = 0.f, acc3 = 0.f;
    constexpr int KS = 8;                                  // reduction indices per inner step (SGPR budget: 16 + 32)
    for (int k0 = 0; k0 < pnchunks * BK; k0 += KS) {
        // wave-uniform table entries and weights first, as one batch of scalar loads (a scalar load per element
        // interleaved with the vector loads would serialise the loop on scalar-memory latency)
        int2 ent[KS];
        float4 w[KS];
#pragma unroll
        for (int i = 0; i < KS; ++i) ent[i] = pktab[k0 + i];
#pragma unroll
        for (int i = 0; i < KS; ++i) w[i] = *reinterpret_cast<const float4*>(pA + (size_t)(k0 + i) * p.Mpad);
        __builtin_amdgcn_sched_barrier(0);
        float xv[KS];
        if constexpr (UT) {
            const int iy = ybase + (ent[0].y >> 16);            // KS divides the tap's channel run: one tap per step
            const int ix = xbase + (int)(short)(ent[0].y & 0xffff);
            const bool ok = ((unsigned)iy < (unsigned)p.GH) & ((unsigned)ix < (unsigned)p.GW);
            const int sp = ok ? xoff + (iy >> p.ups) * p.XW + (ix >> p.ups) : 0;
#pragma unroll
            for (int i = 0; i < KS; ++i) xv[i] = (p.X + ent[i].x)[sp];
#pragma unroll
            for (int i = 0; i < KS; ++i) xv[i] = ok ? xv[i] : 0.f;
        } else {
            unsigned okb = 0u;
#pragma unroll
            for (int i = 0; i < KS; ++i) {
                const int iy = ybase + (ent[i].y >> 16);
                const int ix = xbase + (int)(short)(ent[i].y & 0xffff);
                const bool ok = ((unsigned)iy < (unsigned)p.GH) & ((unsigned)ix < (unsigned)p.GW);
                const int off = xoff + ent[i].x + (iy >> p.ups) * p.XW + (ix >> p.ups);
                xv[i] = p.X[ok ? off : 0];
                okb |= (ok ? 1u : 0u) << i;
            }
#pragma unroll
            for (int i = 0; i < KS; ++i) xv[i] = ((okb >> i) & 1u) ? xv[i] : 0.f;
        }
#pragma unroll
        for (int i = 0; i < KS; ++i) {
            acc0 += xv[i] * w[i].x; acc1 += xv[i] * w[i].y; acc2 += xv[i] * w[i].z; acc3 += xv[i] * w[i].w;
        }
    }
    if (!valid) return;
    float* op = p.Out + (size_t)b * p.OCHW + (qy * p.OSY + pOPY) * p.OW + qx * p.OSX + pOPX;
    const float a[4] = {acc0, acc1, acc2, acc3};
#pragma unroll
    for (int m = 0; m < 4; ++m)
        if (m < p.M) op[(size_t)m * p.OHW] = act_apply(a[m] + (p.bias ? p.bias[m] : 0.f), p.act, p.slope);
}

// The same shapes with the loops the other way round (channel outer, tap inner) -- the classifier stem's input gradient
// (7x7 / stride 2, 64 -> 3 channels, resnet.py conv1; the generator step's feature loss reaches the image through it) and
// ConvTranspose image heads.  gather_small_m_kernel walks the packed reduction order (tap outer, channel inner): between two
// taps that touch the same rows of a channel plane lie the planes of all other channels, times every wave on the CU, so
// each of its loads went to L2 (measured 544 us for 4.9 GFLOP = 0.06 of the matrix peak, 9 TFLOP/s on the vector ALU).
// Here a lane keeps the NT tap offsets of its pixel in registers (out-of-image taps point beyond the buffer: zero fill),
// so a channel costs NT loads from the SAME few rows (L1 hits), no address arithmetic, and NT broadcast LDS reads of the
// phase's weights (staged once per workgroup as float4 [tap][channel]).
template <int NT>
__device__ __forceinline__ void small_m_cmajor_body(const IgemmParams& p, const float4* wl, const unsigned (&voff)[16], int C,
                                                    float& acc0, float& acc1, float& acc2, float& acc3)
{
    const __amdgpu_buffer_rsrc_t x_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.X), 0, p.x_bytes, 0x00020000);
    for (int c = 0; c < C; ++c) {
        const int so = c * p.XHW * 4;
        float xv[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t) xv[t] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(x_rs, (int)voff[t], so, 0));
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const float4 w = wl[t * C + c];
            acc0 += xv[t] * w.x; acc1 += xv[t] * w.y; acc2 += xv[t] * w.z; acc3 += xv[t] * w.w;
        }
    }
}

__global__ __launch_bounds__(256) void small_m_cmajor_kernel(const IgemmParams p)
{
    extern __shared__ __attribute__((aligned(16))) float4 wl[];       // [16][C]; taps beyond the phase's: zeros
    int id = blockIdx.x;
    const float* pA = p.A; const int2* pktab = p.ktab;
    int pN = p.N, pnchunks = p.nchunks, pQW = p.QW, pQHW = p.QHW, pOPY = p.OPY, pOPX = p.OPX;
    if (p.nphases > 1) {
        const int phase = id % p.nphases; id /= p.nphases;
        pA += p.ph[phase].a_off; pktab += p.ph[phase].tab_off;
        pN = p.ph[phase].N; pnchunks = p.ph[phase].nchunks; pQW = p.ph[phase].QW; pQHW = p.ph[phase].QHW;
        pOPY = p.ph[phase].OPY; pOPX = p.ph[phase].OPX;
    }
    if (id * 256 >= pN) return;
    const int C = p.C;
    const int ntaps = pnchunks * BK / C;                       // (the host checked: C % BK == 0, ntaps <= 16)
    for (int k = threadIdx.x; k < 16 * C; k += 256)
        wl[k] = k < ntaps * C ? *reinterpret_cast<const float4*>(pA + (size_t)k * p.Mpad) : make_float4(0.f, 0.f, 0.f, 0.f);
    const int n = id * 256 + threadIdx.x;
    const bool valid = n < pN;
    const int nn = valid ? n : 0;
    const int b = nn / pQHW;
    const int r = nn - b * pQHW;
    const int qy = r / pQW;
    const int qx = r - qy * pQW;
    const int ybase = qy * p.SY, xbase = qx * p.SX;
    unsigned voff[16];
#pragma unroll
    for (int t = 0; t < 16; ++t) {
        const int2 ent = pktab[(t < ntaps ? t : 0) * C];
        const int iy = ybase + (ent.y >> 16), ix = xbase + (int)(short)(ent.y & 0xffff);
        const bool ok = valid && t < ntaps && ((unsigned)iy < (unsigned)p.GH) & ((unsigned)ix < (unsigned)p.GW);
        voff[t] = ok ? 4u * (unsigned)(b * p.XCHW + iy * p.XW + ix) : 0x80000000u;
    }
    __syncthreads();
    float acc0 = 0.f, acc1 = 0.f, acc2 = 0.f, acc3 = 0.f;
    if (ntaps <= 4) small_m_cmajor_body<4>(p, wl, voff, C, acc0, acc1, acc2, acc3);
    else if (ntaps <= 9) small_m_cmajor_body<9>(p, wl, voff, C, acc0, acc1, acc2, acc3);
    else if (ntaps <= 12) small_m_cmajor_body<12>(p, wl, voff, C, acc0, acc1, acc2, acc3);
    else small_m_cmajor_body<16>(p, wl, voff, C, acc0, acc1, acc2, acc3);
    if (!valid) return;
    float* op = p.Out + (size_t)b * p.OCHW + (qy * p.OSY + pOPY) * p.OW + qx * p.OSX + pOPX;
    const float a[4] = {acc0, acc1, acc2, acc3};
#pragma unroll
    for (int m = 0; m < 4; ++m)
        if (m < p.M) op[(size_t)m * p.OHW] = act_apply(a[m] + (p.bias ? p.bias[m] : 0.f), p.act, p.slope);
}

// 3x3 / stride 1 / pad 1 convolution with at most 4 output rows on the vector ALU (generator image head
// generator.py:116, and the input gradient of the critic's 3x3 stem discriminator.py:95): 4 consecutive pixels per
// lane, so a channel costs 3 rows x (one float4 + the two edge pixels) = 9 loads for 108 FMAs; the 9 taps of a channel
// re-read the same three image rows back to back (L1 hits).  Weights sit in LDS as [channel][tap slot] float4 (uniform
// address -> broadcast reads); the slot of a reduction index comes from the gather table, so FWD and DGRAD (flipped
// taps) share the kernel.  The next channel's pixels are in flight while the current one is accumulated.
// WAVE_ROW (image width == 256 = 64 lanes x 4 pixels): a wave is exactly one image row, so the pixel left / right of a
// lane's quad is the neighbouring lane's (one DPP wave shift each, zero at the ends = the padding) and only the three
// aligned float4 loads per channel remain.
template <int NM, bool WAVE_ROW, bool FUSED = false>
__global__ __launch_bounds__(256) void rgb3x3_kernel(const IgemmParams p)
{
    static_assert(!FUSED || WAVE_ROW, "the fused un-pool path is built for 256-pixel rows");
    extern __shared__ float4 wl[];
    const int C = p.C;
    for (int k = threadIdx.x; k < 9 * C; k += 256) {
        const int tap = k / C, ci = k - tap * C;
        const int2 e = p.ktab[k];
        const int dy = e.y >> 16, dx = (int)(short)(e.y & 0xffff);
        wl[ci * 9 + (dy + 1) * 3 + (dx + 1)] = *reinterpret_cast<const float4*>(p.A + (size_t)k * p.Mpad);
    }
    __syncthreads();
    const int W = p.GW, H = p.GH, W4 = W >> 2;
    const int q = blockIdx.x * 256 + threadIdx.x;              // pixel quad
    const bool valid = q < (p.N >> 2);
    const int qq = valid ? q : 0;
    const int per_img = H * W4;
    const int b = qq / per_img;
    const int r = qq - b * per_img;
    const int y = r / W4;
    const int x0 = (r - y * W4) * 4;
    const float* xb = p.X + (size_t)b * p.XCHW;
    int ro[3]; bool rok[3];
    int gro[3], gsh[3];                                        // FUSED: offset of the coarse pixel pair / bit shift of the row parity
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        const int yy = y + d - 1;
        rok[d] = (unsigned)yy < (unsigned)H;
        const int yc = rok[d] ? yy : y;                        // out-of-image rows read row y and are zeroed below
        ro[d] = yc * W + x0;
        gro[d] = (yc >> 1) * (W >> 1) + (x0 >> 1);
        gsh[d] = 2 * (yc & 1);
    }
    const float* gpb = FUSED ? p.GP + (size_t)b * C * (p.XHW >> 2) : nullptr;
    const unsigned char* gbb = FUSED ? p.GBits + (size_t)b * C * (p.XHW >> 3) : nullptr;
    const float q1 = 0.25f, q0 = 0.25f * p.gslope;
    const bool lok = x0 > 0, rrok = x0 + 4 < W;
    const int xl = lok ? -1 : 0, xr = rrok ? 4 : 0;
    float acc[NM][4];
#pragma unroll
    for (int m = 0; m < NM; ++m)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[m][j] = 0.f;

    struct Stage { float4 c[3]; float l[3], r[3]; };
    auto load = [&](Stage& s, int ci) {
        if constexpr (FUSED) {
            // the lane's 4 pixels of row d are two pooling windows: one float2 of the pooled gradient + one byte of sign bits
            const float* gl = gpb + (size_t)ci * (p.XHW >> 2);
            const unsigned char* bl = gbb + (size_t)ci * (p.XHW >> 3);
#pragma unroll
            for (int d = 0; d < 3; ++d) {
                const float2 g2 = *reinterpret_cast<const float2*>(gl + gro[d]);
                const unsigned bt = (unsigned)bl[gro[d] >> 1] >> gsh[d];
                s.c[d] = make_float4(g2.x * ((bt & 1u) ? q1 : q0), g2.x * ((bt & 2u) ? q1 : q0),
                                     g2.y * ((bt & 16u) ? q1 : q0), g2.y * ((bt & 32u) ? q1 : q0));
            }
        } else {
            const float* pl = xb + (size_t)ci * p.XHW;
#pragma unroll
            for (int d = 0; d < 3; ++d) {
                s.c[d] = *reinterpret_cast<const float4*>(pl + ro[d]);
                if constexpr (!WAVE_ROW) { s.l[d] = pl[ro[d] + xl]; s.r[d] = pl[ro[d] + xr]; }
            }
        }
    };
    auto compute = [&](const Stage& s, int ci) {
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            float v[6];
            v[1] = rok[d] ? s.c[d].x : 0.f; v[2] = rok[d] ? s.c[d].y : 0.f;
            v[3] = rok[d] ? s.c[d].z : 0.f; v[4] = rok[d] ? s.c[d].w : 0.f;
            if constexpr (WAVE_ROW) {
                // wave_shr:1 / wave_shl:1 with bound_ctrl off and old = 0: lane 0 / lane 63 receive the zero padding
                v[0] = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v[4]), 0x138, 0xf, 0xf, false));
                v[5] = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v[1]), 0x130, 0xf, 0xf, false));
            } else {
                v[0] = (rok[d] && lok) ? s.l[d] : 0.f;
                v[5] = (rok[d] && rrok) ? s.r[d] : 0.f;
            }
#pragma unroll
            for (int t = 0; t < 3; ++t) {
                const float4 w4 = wl[ci * 9 + d * 3 + t];
                const float w[4] = {w4.x, w4.y, w4.z, w4.w};
#pragma unroll
                for (int m = 0; m < NM; ++m)
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[m][j] += w[m] * v[j + t];
            }
        }
    };
    Stage sa, sb;
    load(sa, 0);
    for (int ci = 0; ci < C; ci += 2) {
        load(sb, min(ci + 1, C - 1));
        compute(sa, ci);
        load(sa, min(ci + 2, C - 1));
        if (ci + 1 < C) compute(sb, ci + 1);
    }
    if (!valid) return;
    float* op = p.Out + (size_t)b * p.OCHW + y * W + x0;
#pragma unroll
    for (int m = 0; m < NM; ++m) {
        if (m < p.M) {
            const float bv = p.bias ? p.bias[m] : 0.f;
            *reinterpret_cast<float4*>(op + (size_t)m * p.OHW) =
                make_float4(act_apply(acc[m][0] + bv, p.act, p.slope), act_apply(acc[m][1] + bv, p.act, p.slope),
                            act_apply(acc[m][2] + bv, p.act, p.slope), act_apply(acc[m][3] + bv, p.act, p.slope));
        }
    }
}

// out[i] = sum_s slab[s][i] (+ bias[channel(i)])      -- GATHER-mode split-K combine
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float* __restrict__ slabs, float* __restrict__ out,
                                                            const float* __restrict__ bias, int splits,
                                                            long long slab_stride, int total4, int OHW, int OC,
                                                            int act, float slope, const float* __restrict__ scale,
                                                            const float* __restrict__ res, int res_w)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= total4) return;
    const float4* s = reinterpret_cast<const float4*>(slabs) + i;
    float4 a = s[0];
    int k = 1;
    for (; k + 3 < splits; k += 4) {                        // four loads in flight, added in slab order (same bits as a plain loop)
        float4 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = *(reinterpret_cast<const float4*>(slabs + (size_t)(k + u) * slab_stride) + i);
#pragma unroll
        for (int u = 0; u < 4; ++u) { a.x += v[u].x; a.y += v[u].y; a.z += v[u].z; a.w += v[u].w; }
    }
    for (; k < splits; ++k) {
        const float4 v = *(reinterpret_cast<const float4*>(slabs + (size_t)k * slab_stride) + i);
        a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
    }
    if (scale) {
        const int e = i * 4;
        a.x *= scale[(e / OHW) % OC]; a.y *= scale[((e + 1) / OHW) % OC];
        a.z *= scale[((e + 2) / OHW) % OC]; a.w *= scale[((e + 3) / OHW) % OC];
    }
    if (bias) {
        const int e = i * 4;
        // OHW may be < 4 (1x1 outputs): resolve the channel per element
        a.x += bias[(e / OHW) % OC]; a.y += bias[((e + 1) / OHW) % OC];
        a.z += bias[((e + 2) / OHW) % OC]; a.w += bias[((e + 3) / OHW) % OC];
    }
    if (res) {
        float4 r;
        if (res_w > 0) {                                  // half-resolution residual (OW % 4 == 0: the 4 elements share a row)
            const int e = i * 4, plane = e / OHW, rem = e - plane * OHW;
            const int yy = rem / res_w, xx = rem - yy * res_w;
            const float2 lo = *reinterpret_cast<const float2*>(res + (size_t)plane * (OHW >> 2) + (yy >> 1) * (res_w >> 1) + (xx >> 1));
            r = make_float4(lo.x, lo.x, lo.y, lo.y);
        } else {
            r = reinterpret_cast<const float4*>(res)[i];
        }
        a.x += r.x; a.y += r.y; a.z += r.z; a.w += r.w;
    }
    if (act != SEMPYR_ACT_NONE) {
        a.x = act_apply(a.x, act, slope); a.y = act_apply(a.y, act, slope);
        a.z = act_apply(a.z, act, slope); a.w = act_apply(a.w, act, slope);
    }
    reinterpret_cast<float4*>(out)[i] = a;
}

__global__ __launch_bounds__(256) void splitk_reduce_tail_kernel(const float* __restrict__ slabs, float* __restrict__ out,
                                                                 const float* __restrict__ bias, int splits,
                                                                 long long slab_stride, int begin, int total, int OHW, int OC,
                                                                 int act, float slope, const float* __restrict__ scale,
                                                                 const float* __restrict__ res, int res_w)
{
    const int i = begin + blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    float a = slabs[i];
    for (int k = 1; k < splits; ++k) a += slabs[(size_t)k * slab_stride + i];
    if (scale) a *= scale[(i / OHW) % OC];
    if (bias) a += bias[(i / OHW) % OC];
    if (res) {
        if (res_w > 0) {
            const int plane = i / OHW, rem = i - plane * OHW;
            const int yy = rem / res_w, xx = rem - yy * res_w;
            a += res[(size_t)plane * (OHW >> 2) + (yy >> 1) * (res_w >> 1) + (xx >> 1)];
        } else {
            a += res[i];
        }
    }
    out[i] = act_apply(a, act, slope);
}

// gw[co][ci][tap] = sum_s slab[s][co][n = tap*Cin + ci]   -- WGRAD combine + re-layout to torch order.
// Threads walk the slab layout (n fastest) so the `splits` reads are coalesced; the single write is the scattered one.
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* __restrict__ slabs, float* __restrict__ gw,
                                                           int splits, long long slab_stride, int M, int Cin, int T, int ldc,
                                                           int accumulate)
{
    const int N = Cin * T;
    const int i = blockIdx.x * 256 + threadIdx.x;        // index in GEMM layout: m * N + n
    if (i >= M * N) return;
    const int m = i / N;
    const int n = i - m * N;
    const int tap = n / Cin;
    const int ci = n - tap * Cin;
    const size_t src = (size_t)m * ldc + n;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;        // 4 interleaved partial sums (fixed order -> deterministic)
    int k = 0;
    for (; k + 3 < splits; k += 4) {
        a0 += slabs[(size_t)k * slab_stride + src];
        a1 += slabs[(size_t)(k + 1) * slab_stride + src];
        a2 += slabs[(size_t)(k + 2) * slab_stride + src];
        a3 += slabs[(size_t)(k + 3) * slab_stride + src];
    }
    for (; k < splits; ++k) a0 += slabs[(size_t)k * slab_stride + src];
    const size_t dst = (size_t)m * N + ci * T + tap;
    const float sum = (a0 + a1) + (a2 + a3);
    gw[dst] = accumulate ? gw[dst] + sum : sum;                // accumulate: gw += this launch's gradient
}

// The same through LDS for wide layers: a workgroup owns one output row m and 64 input channels, reads the slabs as runs of 64
// consecutive floats per tap (coalesced), sums them in the fixed split order, and writes its 64 x T block of gw as ONE
// contiguous run (the plain kernel's write is a 4-byte scatter with stride T: 512 -> 512 4x4 took 24 us for 16.8 MB).
__global__ __launch_bounds__(256) void wgrad_reduce_t_kernel(const float* __restrict__ slabs, float* __restrict__ gw,
                                                             int splits, long long slab_stride, int Cin, int T, int ldc,
                                                             int accumulate)
{
    extern __shared__ float tile[];                             // [64][T + 1]
    const int cblocks = (Cin + 63) / 64;
    const int m = blockIdx.x / cblocks, c0 = (blockIdx.x - m * cblocks) * 64;
    const int nc = min(64, Cin - c0);
    const int c = threadIdx.x & 63;
    const size_t src = (size_t)m * ldc + c0 + c;
    if (c < nc) {
        for (int t = threadIdx.x >> 6; t < T; t += 4) {
            const float* sp = slabs + src + (size_t)t * Cin;
            float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;       // the plain kernel's interleaved partial sums: same bits
            int k = 0;
            for (; k + 3 < splits; k += 4) {
                a0 += sp[(size_t)k * slab_stride];
                a1 += sp[(size_t)(k + 1) * slab_stride];
                a2 += sp[(size_t)(k + 2) * slab_stride];
                a3 += sp[(size_t)(k + 3) * slab_stride];
            }
            for (; k < splits; ++k) a0 += sp[(size_t)k * slab_stride];
            tile[c * (T + 1) + t] = (a0 + a1) + (a2 + a3);
        }
    }
    __syncthreads();
    float* dst = gw + ((size_t)m * Cin + c0) * T;
    for (int i = threadIdx.x; i < nc * T; i += 256) {
        const int cc = i / T, t = i - cc * T;
        const float v = tile[cc * (T + 1) + t];
        dst[i] = accumulate ? dst[i] + v : v;
    }
}

// The same for many slabs (thin problems reduce over 2^20 pixels in 500+ splits): 16 consecutive outputs (one 64-B
// segment per slab) x 16 split groups per workgroup, so a thread walks splits/16 slabs instead of all of them
// (1728 outputs x 512 slabs took 42 us as one dependent chain per thread). Fixed summation order: deterministic.
__global__ __launch_bounds__(256) void wgrad_reduce_many_kernel(const float* __restrict__ slabs, float* __restrict__ gw,
                                                                int splits, long long slab_stride, int M, int Cin, int T,
                                                                int ldc, int accumulate)
{
    __shared__ float sh[16][17];
    const int N = Cin * T;
    const int o = blockIdx.x * 16 + (threadIdx.x & 15);
    const int sg = threadIdx.x >> 4;
    const bool live = o < M * N;
    const int oo = live ? o : 0;
    const int m = oo / N;
    const int n = oo - m * N;
    const size_t src = (size_t)m * ldc + n;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    int k = sg;
    for (; k + 48 < splits; k += 64) {
        a0 += slabs[(size_t)k * slab_stride + src];
        a1 += slabs[(size_t)(k + 16) * slab_stride + src];
        a2 += slabs[(size_t)(k + 32) * slab_stride + src];
        a3 += slabs[(size_t)(k + 48) * slab_stride + src];
    }
    for (; k < splits; k += 16) a0 += slabs[(size_t)k * slab_stride + src];
    sh[sg][threadIdx.x & 15] = (a0 + a1) + (a2 + a3);
    __syncthreads();
    if (sg == 0 && live) {
        float t = 0.f;
#pragma unroll
        for (int g = 0; g < 16; ++g) t += sh[g][threadIdx.x];
        const int tap = n / Cin;
        const int ci = n - tap * Cin;
        const size_t dst = (size_t)m * N + ci * T + tap;
        gw[dst] = accumulate ? gw[dst] + t : t;
    }
}

// wp[k][m] = w[wsrc[k] + m*mstride]  (0 where wsrc[k] < 0 or m >= M)
__global__ __launch_bounds__(256) void pack_weights_kernel(const float* __restrict__ w, const int* __restrict__ wsrc,
                                                           float* __restrict__ wp, int Kpad_total, int Mpad, int M, int mstride)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= Kpad_total * Mpad) return;
    const int k = i / Mpad, m = i - k * Mpad;
    const int s = wsrc[k];
    wp[i] = (s >= 0 && m < M) ? w[(size_t)s + (size_t)m * mstride] : 0.f;
}


// Weight gradient of an RGB head (3x3 / stride 1 / pad 1, <= 4 output channels: the generator's last conv) on the vector
// ALU: padding M = 3 to a 32-row MFMA tile wastes 10x the matrix work (0.55 ms, 6.6 TFLOP/s).  A workgroup owns RGBW_ROWS image
// rows of one sample and RGBW_CG input channels; a thread owns one pixel column, walks down its rows with a rolling 3x3
// window per channel (3 new loads per channel and row) and keeps NM x 9 x RGBW_CG accumulators; the workgroup's sums go
// to `part` [strip][C][NM*9] and rgbw_reduce_kernel adds the strips in a fixed order.  Measured on 64 -> 3 at 256x256,
// batch 16: the per-workgroup reduction of the accumulators is the expensive part, so long strips win (rows 32 / 4 channels
// 291 us, 128 / 2 channels 146 us, 8 / 4 channels 627 us; MFMA path 577 us).
inline int rgbw_rows();
inline int rgbw_cg();
template <int NM, int RGBW_CG>
__global__ __launch_bounds__(256) void rgb_wgrad3x3_kernel(const float* __restrict__ gy, const float* __restrict__ x,
                                                           int C, int H, int W, int RGBW_ROWS, float* __restrict__ part)
{
    __shared__ float sh[4][NM * 9 * RGBW_CG];
    const int nstrip_img = (H + RGBW_ROWS - 1) / RGBW_ROWS;
    const int cgroups = C / RGBW_CG;
    int id = blockIdx.x;
    const int cg = id % cgroups; id /= cgroups;
    const int strip = id % nstrip_img; const int b = id / nstrip_img;
    const int rpp = 256 / W;                                   // image rows covered by one pass of the workgroup
    const int col = threadIdx.x % W, rsub = threadIdx.x / W;
    const int y0 = strip * RGBW_ROWS;
    const int yend = min(y0 + RGBW_ROWS, H);
    const float* xb = x + ((size_t)b * C + (size_t)cg * RGBW_CG) * H * W;
    const float* gb = gy + (size_t)b * NM * H * W;
    float acc[RGBW_CG][NM][9];
#pragma unroll
    for (int c = 0; c < RGBW_CG; ++c)
#pragma unroll
        for (int m = 0; m < NM; ++m)
#pragma unroll
            for (int t = 0; t < 9; ++t) acc[c][m][t] = 0.f;
    const bool lok = col > 0, rok = col < W - 1;
    auto load_row = [&](int c, int yy, float (&r)[3]) {        // x[c][yy][col-1 .. col+1], zero outside the image
        if (yy < 0 || yy >= H) { r[0] = r[1] = r[2] = 0.f; return; }
        const float* q = xb + (size_t)c * H * W + (size_t)yy * W + col;
        r[1] = q[0]; r[0] = lok ? q[-1] : 0.f; r[2] = rok ? q[1] : 0.f;
    };
    if (threadIdx.x < rpp * W) {
        float win[RGBW_CG][3][3];
        for (int y = y0 + rsub; y < yend; y += rpp) {
            if (rpp == 1 && y > y0) {                          // rolling window: rows y-1, y are already there
#pragma unroll
                for (int c = 0; c < RGBW_CG; ++c) {
#pragma unroll
                    for (int k = 0; k < 3; ++k) { win[c][0][k] = win[c][1][k]; win[c][1][k] = win[c][2][k]; }
                    load_row(c, y + 1, win[c][2]);
                }
            } else {
#pragma unroll
                for (int c = 0; c < RGBW_CG; ++c) { load_row(c, y - 1, win[c][0]); load_row(c, y, win[c][1]); load_row(c, y + 1, win[c][2]); }
            }
            float g[NM];
#pragma unroll
            for (int m = 0; m < NM; ++m) g[m] = gb[(size_t)m * H * W + (size_t)y * W + col];
#pragma unroll
            for (int c = 0; c < RGBW_CG; ++c)
#pragma unroll
                for (int m = 0; m < NM; ++m)
#pragma unroll
                    for (int t = 0; t < 9; ++t) acc[c][m][t] = fmaf(g[m], win[c][t / 3][t % 3], acc[c][m][t]);
        }
    }
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
#pragma unroll
    for (int c = 0; c < RGBW_CG; ++c)
#pragma unroll
        for (int m = 0; m < NM; ++m)
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                const float v = wave_sum(acc[c][m][t]);
                if (lane == 0) sh[wave][(c * NM + m) * 9 + t] = v;
            }
    __syncthreads();
    if (threadIdx.x < NM * 9 * RGBW_CG) {
        const int i = threadIdx.x;
        const float v = (sh[0][i] + sh[1][i]) + (sh[2][i] + sh[3][i]);
        const int c = i / (NM * 9), rem = i - c * (NM * 9);
        // part[(b * nstrip_img + strip)][channel][m * 9 + t]
        part[((size_t)(b * nstrip_img + strip) * C + (size_t)cg * RGBW_CG + c) * (NM * 9) + rem] = v;
    }
}

// gw[m][ci][t] (torch order) = sum over strips of part[strip][ci][m*9 + t]
__global__ __launch_bounds__(256) void rgbw_reduce_kernel(const float* __restrict__ part, int nstrips, int C, int NM,
                                                          float* __restrict__ gw, int accumulate)
{
    const int i = blockIdx.x * 256 + threadIdx.x;              // index into [ci][m*9 + t]
    const int per = NM * 9;
    if (i >= C * per) return;
    float a0 = 0.f, a1 = 0.f;
    int s = 0;
    for (; s + 1 < nstrips; s += 2) { a0 += part[(size_t)s * C * per + i]; a1 += part[(size_t)(s + 1) * C * per + i]; }
    if (s < nstrips) a0 += part[(size_t)s * C * per + i];
    const int ci = i / per, rem = i - ci * per;
    const int m = rem / 9, t = rem - m * 9;
    const size_t dst = ((size_t)m * C + ci) * 9 + t;
    const float sum = a0 + a1;
    gw[dst] = accumulate ? gw[dst] + sum : sum;
}

// The same for many weight tensors in ONE launch (all packed layouts of a module, refreshed right behind its optimizer
// step instead of one small launch in front of each convolution of the next pass).  The item table travels in the kernel
// arguments; a workgroup finds its item by a binary search over the items' first-workgroup index.
constexpr int PACK_MAX = SEMPYR_PACK_MAX_ITEMS;
struct PackManyArgs {
    const float* w[PACK_MAX];
    const int* wsrc[PACK_MAX];
    float* wp[PACK_MAX];
    int first_block[PACK_MAX + 1];
    int total[PACK_MAX], Mpad[PACK_MAX], M[PACK_MAX], mstride[PACK_MAX];
    int n;
};
__global__ __launch_bounds__(256) void pack_weights_many_kernel(const PackManyArgs a)
{
    int lo = 0, hi = a.n - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (a.first_block[mid] <= (int)blockIdx.x) lo = mid; else hi = mid - 1;
    }
    const int it = lo;
    const int i = ((int)blockIdx.x - a.first_block[it]) * 256 + threadIdx.x;
    if (i >= a.total[it]) return;
    const int Mpad = a.Mpad[it];
    const int k = i / Mpad, m = i - k * Mpad;
    const int s = a.wsrc[it][k];
    a.wp[it][i] = (s >= 0 && m < a.M[it]) ? a.w[it][(size_t)s + (size_t)m * a.mstride[it]] : 0.f;
}

// ------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------
enum { CFG_128x128 = 0, CFG_64x256 = 1, CFG_32x256 = 2, CFG_64x64 = 3, CFG_64x192 = 4, CFG_64x128 = 5,
       CFG_TPATCH = 6 /* 4x4 / stride-2 patch kernels: the tile comes from tp_pick() */,
       CFG_WPATCH = 7 /* row-ring weight-gradient kernel: the slabs come from wp_pick() */ };

// Tile shape for a GATHER launch over N pixels (all stride phases of a DGRAD together).  `kind`: the kernel the launch
// will run on — the choices below are measured (tools/bench_conv.py mid / small with SEMPYR_FORCE_CFG*).
enum { KIND_GENERIC_FWD = 0, KIND_PATCH3X3 = 1, KIND_DGRAD = 2, KIND_ANY = 3 };
inline int env_int(const char* name, int dflt);
inline int pick_cfg(int Mpad, long long N = 1ll << 40, int kind = KIND_ANY) {
    static const int small_tiles = env_int("SEMPYR_SMALL_TILES", 1);          // 0: the pre-tuning rules (A/B aid)
    if (!small_tiles) {
        if (Mpad % 128 == 0) return CFG_128x128;
        if (Mpad % 64 == 0) {
            if (kind == KIND_DGRAD || (kind == KIND_PATCH3X3 && N == (1ll << 40))) return CFG_64x256;
            const long long t256 = (Mpad / 64) * ((N + 255) / 256), t64 = (Mpad / 64) * ((N + 63) / 64);
            return (t256 < 512 && t64 >= 512) ? CFG_64x64 : CFG_64x256;
        }
        return CFG_32x256;
    }
    if (Mpad % 128 == 0) {
        // few 128x128 tiles mean deep split-K (slab writes + a combine pass per launch): 64x64 tiles need a quarter of the
        // splits. Measured at batch 16: 3x3 128->128 @32^2 0.060 -> 0.048 ms, 512->512 @8^2 0.062 -> 0.051, input gradient
        // of the critic's 512->512 4x4 s2 block 0.059 -> 0.041; the forward 4x4 s2 blocks with 32..128 tiles prefer 128x128.
        const long long t128 = (Mpad / 128) * ((N + 127) / 128);
        if (t128 < 512 && (kind == KIND_PATCH3X3 || kind == KIND_DGRAD)) return CFG_64x64;
        if (t128 < 16 && kind == KIND_GENERIC_FWD) return CFG_64x64;
        return CFG_128x128;
    }
    if (Mpad % 64 == 0) {
        // mid-size pixel counts (64 channels at 64x64 x batch 16): narrower tiles fill the chip without split-K slabs and
        // their combine pass (measured 0.121 -> 0.103 ms on the critic's 64->64 4x4 s2 block; the 3x3 patch kernel is
        // another 6-8 % faster on 64x128 tiles when those already give 512 workgroups)
        const long long t256 = (Mpad / 64) * ((N + 255) / 256), t128 = (Mpad / 64) * ((N + 127) / 128),
                        t64 = (Mpad / 64) * ((N + 63) / 64);
        if (t256 < 512 && t128 >= 512 && kind == KIND_PATCH3X3) return CFG_64x128;
        if (t256 < 512 && t64 >= 512) return CFG_64x64;
        return CFG_64x256;
    }
    return CFG_32x256;
}
inline int conv_kind(const sempyr_conv_desc* d, int mode) {
    if (d->kh == 3 && d->kw == 3 && d->stride == 1 && d->pad == 1 && !(mode == SEMPYR_CONV_DGRAD && d->ups) &&
        (mode == SEMPYR_CONV_FWD ? d->Cin : d->Cout) % 8 == 0)
        return KIND_PATCH3X3;
    return mode == SEMPYR_CONV_DGRAD ? KIND_DGRAD : KIND_GENERIC_FWD;
}
inline void cfg_dims(int cfg, int& bm, int& bn) {
    if (cfg == CFG_128x128) { bm = 128; bn = 128; }
    else if (cfg == CFG_64x256) { bm = 64; bn = 256; }
    else if (cfg == CFG_64x64) { bm = 64; bn = 64; }
    else if (cfg == CFG_64x192) { bm = 64; bn = 192; }
    else if (cfg == CFG_64x128) { bm = 64; bn = 128; }
    else { bm = 32; bn = 256; }
}
// WGRAD: the N dimension (Cin*taps) can be tiny (27 for an RGB stem, 64 for a 1x1 skip): use a narrow tile there
inline int pick_cfg_wgrad(int Mpad, int Nn, bool uniform_tap) {
    if (Nn <= 64 && Mpad % 64 == 0) return CFG_64x64;
    // 64-row problems whose column count is a multiple of 192 but pads by > 10 % in 256-wide tiles (576 -> 768)
    if (uniform_tap && Mpad % 128 != 0 && Mpad % 64 == 0 && Nn % 192 == 0 && (Nn + 255) / 256 * 256 * 10 > Nn * 11)
        return CFG_64x192;
    return pick_cfg(Mpad);
}

// split-K policy: aim at TARGET_BLOCKS workgroups (2 per CU) but keep at least MIN_CHUNKS k-chunks per split so the
// per-workgroup prologue / slab write does not dominate. (SEMPYR_TARGET_BLOCKS / SEMPYR_MIN_CHUNKS override: tuning aid.)
inline int env_int(const char* name, int dflt) { const char* v = getenv(name); return v ? atoi(v) : dflt; }
inline int target_blocks() { static const int v = env_int("SEMPYR_TARGET_BLOCKS", 512); return v; }
// WGRAD reduces over all pixels (K up to 2^20) into a handful of tiles: 3 workgroups per CU (768) balance better than 2
// (measured 0.98 -> 0.82 ms on the 64x576x2^20 problem), while the short-K forward / input-gradient launches prefer 512.
inline int target_blocks_wgrad() { static const int v = env_int("SEMPYR_TARGET_BLOCKS_WGRAD", 768); return v; }
inline int min_chunks() { static const int v = env_int("SEMPYR_MIN_CHUNKS", 8); return v; }

inline void pick_splits(int tiles, int nchunks, int& splits, int& cps, int target = 0) {
    if (target <= 0) target = target_blocks();
    splits = 1;
    if (tiles < target) {
        splits = (target + tiles - 1) / tiles;
        const int maxs = nchunks / min_chunks() > 0 ? nchunks / min_chunks() : 1;
        if (splits > maxs) splits = maxs;
        if (splits > 2048) splits = 2048;
    }
    cps = (nchunks + splits - 1) / splits;
    splits = (nchunks + cps - 1) / cps;                   // no empty splits
    if (splits < 1) splits = 1;
}

struct Geometry {
    int Ho, Wo, GHin, GWin, T;
};

inline int check_desc(const sempyr_conv_desc* d, Geometry& g) {
    if (!d) SEMPYR_FAIL("conv: null descriptor");
    if (d->B < 1 || d->Cin < 1 || d->Cout < 1 || d->Hin < 1 || d->Win < 1 || d->kh < 1 || d->kw < 1)
        SEMPYR_FAIL("conv: non-positive dimension");
    if (d->stride != 1 && d->stride != 2) SEMPYR_FAIL("conv: stride %d unsupported (1 or 2)", d->stride);
    if (d->pad < 0 || d->pad > 7) SEMPYR_FAIL("conv: pad %d unsupported", d->pad);
    if (d->ups != 0 && d->ups != 1) SEMPYR_FAIL("conv: ups must be 0/1");
    if (d->math != SEMPYR_MATH_F32 && d->math != SEMPYR_MATH_BF16X3) SEMPYR_FAIL("conv: unknown math mode %d", d->math);
    g.GHin = d->Hin << d->ups; g.GWin = d->Win << d->ups;
    if (g.GHin + 2 * d->pad < d->kh || g.GWin + 2 * d->pad < d->kw) SEMPYR_FAIL("conv: kernel larger than padded input");
    g.Ho = (g.GHin + 2 * d->pad - d->kh) / d->stride + 1;
    g.Wo = (g.GWin + 2 * d->pad - d->kw) / d->stride + 1;
    g.T = d->kh * d->kw;
    const long long lim = (1ll << 29) - 1;                       // 2 GiB of fp32: the buffer-load offset range
    if ((long long)d->B * d->Cin * g.GHin * g.GWin > lim || (long long)d->B * d->Cout * g.Ho * g.Wo > lim ||
        (long long)d->Cout * d->Cin * g.T > lim)
        SEMPYR_FAIL("conv: tensor larger than 2^29 elements");
    return 0;
}

inline int phase_taps(int kdim, int s, int p, int ph, int& r, int& c) {
    r = (ph + p) % s; c = (ph + p) / s;
    return r < kdim ? (kdim - r + s - 1) / s : 0;
}

// Opt-in kernel timing (bench.py's roofline leg): HIP events recorded on the launch stream around every
// implicit-GEMM launch, plus the launch's algorithmic FLOPs (2*M*N*K on the unpadded dims).
struct Profiler {
    std::mutex mu;
    bool on = false;
    std::vector<hipEvent_t> events;     // pairs (start, stop)
    struct Rec { int mode, bm, bn, M, N, K, splits, nphases; double flops; };
    std::vector<Rec> recs;
    size_t used = 0;
    double flops = 0.0;
};
Profiler g_prof;

// Dynamic LDS to ask for on top of `have` bytes so that at most SEMPYR_CONV_OCC_CAP workgroups of a convolution kernel fit on
// a CU (0 = no cap): the registers and LDS it leaves free are what lets the HBM-bound kernels of the other streams run
// beside an MFMA-bound convolution instead of queueing behind it.
inline size_t occ_cap_pad(size_t have) {
    static const int cap = env_int("SEMPYR_CONV_OCC_CAP", 0);
    if (cap < 1 || cap > 3) return 0;
    const size_t need = (size_t)(160 * 1024) / (cap + 1) + 256;
    return need > have ? need - have : 0;
}

template <int BM, int BN, int WM, int WN, int MODE, bool UT>
int launch_igemm(const IgemmParams& p, int splits, hipStream_t st) {
    const int MT = (p.Mpad + BM - 1) / BM;
    const int NT = (p.N + BN - 1) / BN;
    IgemmParams pp = p; pp.nsplits = splits;
    if (pp.nphases < 1) pp.nphases = 1;
    int NTm = NT; double flops = 2.0 * (double)p.M * (double)p.N * (double)p.K;
    if (pp.nphases > 1) {
        NTm = 0; flops = 0.0;
        for (int i = 0; i < pp.nphases; ++i) {
            const int nt = (pp.ph[i].N + BN - 1) / BN;
            if (nt > NTm) NTm = nt;
            flops += 2.0 * (double)p.M * (double)pp.ph[i].N * (double)pp.ph[i].nchunks * BK;
        }
    }
    const int NT8m = (NTm + 7) / 8 * 8;
    dim3 grid((splits > 1 ? MT * NTm * splits : MT * NT8m) * pp.nphases, 1, 1);
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (g_prof.on) {
        std::lock_guard<std::mutex> lk(g_prof.mu);
        if (g_prof.used + 2 <= g_prof.events.size()) {
            e0 = g_prof.events[g_prof.used]; e1 = g_prof.events[g_prof.used + 1];
            g_prof.used += 2;
            g_prof.flops += flops;
            g_prof.recs.push_back({MODE + (UT ? 0 : 2), BM, BN, p.M, p.N, p.K, splits, pp.nphases, flops});
        }
    }
    if (e0) (void)hipEventRecord(e0, st);
    bool deep = false;
    if constexpr (MODE == MODE_GATHER && UT && BM == 64 && (BN == 64 || BN == 128)) {
        // (64x256 tiles were measured too: 80 KB of LDS leave 2 workgroups per CU and the cycle got 0.3 ms slower)
        static const int mask = env_int("SEMPYR_BK32", 3);           // bit 0: 64x64 tiles, 1: 64x128
        const int bit = BN == 64 ? 1 : 2;
        deep = (mask & bit) && p.C % 32 == 0 && pp.nchunks % 2 == 0;
        for (int i = 0; deep && pp.nphases > 1 && i < pp.nphases; ++i) deep = pp.ph[i].nchunks % 2 == 0;
        if (deep) {
            pp.nchunks /= 2;
            pp.chunks_per_split = (pp.nchunks + splits - 1) / splits;
            for (int i = 0; i < pp.nphases && pp.nphases > 1; ++i) pp.ph[i].nchunks /= 2;
            hipLaunchKernelGGL((igemm_kernel<BM, BN, WM, WN, MODE, UT, 32>), grid, dim3(256), occ_cap_pad((size_t)2 * 32 * (BM + BN) * 4), st, pp);
        }
    }
    if (!deep) hipLaunchKernelGGL((igemm_kernel<BM, BN, WM, WN, MODE, UT>), grid, dim3(256), occ_cap_pad((size_t)2 * 16 * (BM + BN + 4) * 4), st, pp);
    if (e1) (void)hipEventRecord(e1, st);
    SEMPYR_CHECK_LAUNCH();
    return 0;
}

template <int MODE, bool UT>
int launch_cfg_ut(int cfg, const IgemmParams& p, int splits, hipStream_t st) {
    switch (cfg) {
        case CFG_128x128: return launch_igemm<128, 128, 2, 2, MODE, UT>(p, splits, st);
        case CFG_64x256:  return launch_igemm<64, 256, 1, 4, MODE, UT>(p, splits, st);
        case CFG_64x64:   return launch_igemm<64, 64, 2, 2, MODE, UT>(p, splits, st);
        case CFG_64x128:  return launch_igemm<64, 128, 2, 2, MODE, UT>(p, splits, st);
        case CFG_64x192:
            if constexpr (MODE == MODE_WGRAD && UT) return launch_igemm<64, 192, 2, 2, MODE, UT>(p, splits, st);
            else { sempyr_set_error("conv: the 64x192 tile is a uniform-tap WGRAD configuration"); return SEMPYR_ERR_INVALID; }
        default:          return launch_igemm<32, 256, 1, 4, MODE, UT>(p, splits, st);
    }
}
// `uniform_tap`: the gathered tensor's channel count is a multiple of BK and K needs no padding (see the kernel)
int launch_small_m(const IgemmParams& p, bool uniform_tap, hipStream_t st) {
    IgemmParams pp = p; pp.nsplits = 1;
    if (pp.nphases < 1) pp.nphases = 1;
    int maxN = p.N; double flops = 2.0 * (double)p.M * (double)p.N * (double)p.K;
    if (pp.nphases > 1) {
        maxN = 0; flops = 0.0;
        for (int i = 0; i < pp.nphases; ++i) {
            if (pp.ph[i].N > maxN) maxN = pp.ph[i].N;
            flops += 2.0 * (double)p.M * (double)pp.ph[i].N * (double)pp.ph[i].nchunks * BK;
        }
    }
    dim3 grid(ceil_div(maxN, 256) * pp.nphases, 1, 1);
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (g_prof.on) {
        std::lock_guard<std::mutex> lk(g_prof.mu);
        if (g_prof.used + 2 <= g_prof.events.size()) {
            e0 = g_prof.events[g_prof.used]; e1 = g_prof.events[g_prof.used + 1];
            g_prof.used += 2;
            g_prof.flops += flops;
            g_prof.recs.push_back({4 + (uniform_tap ? 0 : 2), 4, 256, p.M, p.N, p.K, 1, pp.nphases, flops});
        }
    }
    if (e0) (void)hipEventRecord(e0, st);
    // channel-major walk: whole taps of C channels each (uniform_tap and no K padding), at most 16 taps per phase, no fused
    // upsample, packed rows of at least 4 floats, and the gathered tensor inside one buffer descriptor
    static const int cmajor = env_int("SEMPYR_SMALL_M_CMAJOR", 1);
    bool cm = cmajor && uniform_tap && p.ups == 0 && p.C > 0 && p.C <= 256 && p.C % BK == 0 && p.Mpad >= 4 && !p.GP;
    if (cm) {
        if (pp.nphases > 1) { for (int i = 0; i < pp.nphases; ++i) if (pp.ph[i].N > 0 && (pp.ph[i].nchunks * BK % p.C != 0 || pp.ph[i].nchunks * BK / p.C > 16)) cm = false; }
        else if (p.nchunks * BK % p.C != 0 || p.nchunks * BK / p.C > 16) cm = false;
    }
    if (cm) hipLaunchKernelGGL(small_m_cmajor_kernel, grid, dim3(256), (size_t)16 * p.C * sizeof(float4), st, pp);
    else if (uniform_tap) hipLaunchKernelGGL(gather_small_m_kernel<true>, grid, dim3(256), 0, st, pp);
    else hipLaunchKernelGGL(gather_small_m_kernel<false>, grid, dim3(256), 0, st, pp);
    if (e1) (void)hipEventRecord(e1, st);
    SEMPYR_CHECK_LAUNCH();
    return 0;
}

// Shapes served by rgb3x3_kernel: 3x3 / stride 1 / pad 1, no fused upsample, <= 4 GEMM rows, image width % 4 == 0.
inline bool rgb3x3_ok(const sempyr_conv_desc* d, int mode) {
    const int M = mode == SEMPYR_CONV_FWD ? d->Cout : d->Cin;
    const int C = mode == SEMPYR_CONV_FWD ? d->Cin : d->Cout;
    return (mode == SEMPYR_CONV_FWD || mode == SEMPYR_CONV_DGRAD) && M <= 4 && C <= 256 && d->kh == 3 && d->kw == 3 &&
           d->stride == 1 && d->pad == 1 && d->ups == 0 && d->Win % 4 == 0;
}

int launch_rgb3x3(const IgemmParams& p, hipStream_t st) {
    if (((uintptr_t)p.X | (uintptr_t)p.Out) & 15) SEMPYR_FAIL("conv: rgb3x3 path needs 16-byte aligned tensors");
    if (p.GP && (p.GW != 256 || ((uintptr_t)p.GP & 7))) SEMPYR_FAIL("conv: the fused un-pool input gradient needs 256-pixel rows");
    dim3 grid(ceil_div(p.N / 4, 256), 1, 1);
    const size_t lds = (size_t)9 * p.C * sizeof(float4);
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (g_prof.on) {
        std::lock_guard<std::mutex> lk(g_prof.mu);
        if (g_prof.used + 2 <= g_prof.events.size()) {
            e0 = g_prof.events[g_prof.used]; e1 = g_prof.events[g_prof.used + 1];
            g_prof.used += 2;
            const double flops = 2.0 * (double)p.M * (double)p.N * (double)p.K;
            g_prof.flops += flops;
            g_prof.recs.push_back({8, 4, 1024, p.M, p.N, p.K, 1, 1, flops});
        }
    }
    if (e0) (void)hipEventRecord(e0, st);
    if (p.GP) {
        if (p.M <= 3) hipLaunchKernelGGL((rgb3x3_kernel<3, true, true>), grid, dim3(256), lds, st, p);
        else hipLaunchKernelGGL((rgb3x3_kernel<4, true, true>), grid, dim3(256), lds, st, p);
    } else if (p.GW == 256) {
        if (p.M <= 3) hipLaunchKernelGGL((rgb3x3_kernel<3, true>), grid, dim3(256), lds, st, p);
        else hipLaunchKernelGGL((rgb3x3_kernel<4, true>), grid, dim3(256), lds, st, p);
    } else {
        if (p.M <= 3) hipLaunchKernelGGL((rgb3x3_kernel<3, false>), grid, dim3(256), lds, st, p);
        else hipLaunchKernelGGL((rgb3x3_kernel<4, false>), grid, dim3(256), lds, st, p);
    }
    if (e1) (void)hipEventRecord(e1, st);
    SEMPYR_CHECK_LAUNCH();
    return 0;
}

// Patch-kernel eligibility for a GATHER launch (FWD or stride-1 DGRAD) with tile width bn.  On success fills PR/PW/CW.
inline bool patch3x3_ok(const sempyr_conv_desc* d, int mode, int bm, int bn, IgemmParams& p) {
    const int C = mode == SEMPYR_CONV_FWD ? d->Cin : d->Cout;
    const int M = mode == SEMPYR_CONV_FWD ? d->Cout : d->Cin;
    if (d->kh != 3 || d->kw != 3 || d->stride != 1 || d->pad != 1 || C % PCG != 0 || M <= 4) return false;
    if (mode == SEMPYR_CONV_DGRAD && d->ups) return false;
    if (p.Mpad % bm != 0 || p.N % bn != 0 || p.QHW % bn != 0) return false;      // tiles are whole and inside one sample
    int R, CW;
    static const int blk = env_int("SEMPYR_PATCH_BLOCK", 64);   // columns of a 2-D tile block (0: whole rows / row segments only)
    if (blk >= 64 && blk % 64 == 0 && p.QW > blk && p.QW % blk == 0 && bn % blk == 0 && p.QH % (bn / blk) == 0) { R = bn / blk; CW = blk; }
    else if (bn % p.QW == 0) { R = bn / p.QW; CW = p.QW; }
    else if (p.QW % bn == 0) { R = 1; CW = bn; }
    else return false;
    const int s = p.ups;
    if (s && ((CW & 1) || (R > 1 && (R & 1)))) return false;
    const int PR = s ? (R >= 2 ? R / 2 + 2 : 2) : R + 2;
    const int PW = s ? CW / 2 + 2 : CW + 2;
    if (PR * PW > 1024) return false;
    const int nsp = (PR * PW + 255) / 256 < 2 ? 2 : (PR * PW + 255) / 256;
    if ((9 * PCG * bm + PCG * nsp * 256) * 4 > 160 * 1024 / 3) return false;     // 3 workgroups per CU
    p.PR = PR; p.PW = PW; p.CW = CW;
    return true;
}

template <int BM, int BN, int WM, int WN>
int launch_patch(const IgemmParams& p, int splits, hipStream_t st) {
    const int MT = p.Mpad / BM, NT = p.N / BN;
    IgemmParams pp = p; pp.nsplits = splits; pp.nphases = 1;
    const int NT8 = (NT + 7) / 8 * 8;
    dim3 grid(splits > 1 ? MT * NT * splits : MT * NT8, 1, 1);
    const int nsp = (p.PR * p.PW + 255) / 256 < 2 ? 2 : (p.PR * p.PW + 255) / 256;
    size_t lds = (size_t)(9 * PCG * BM + PCG * nsp * 256) * sizeof(float);
    lds += occ_cap_pad(lds);
    const double flops = 2.0 * (double)p.M * (double)p.N * (double)p.K;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (g_prof.on) {
        std::lock_guard<std::mutex> lk(g_prof.mu);
        if (g_prof.used + 2 <= g_prof.events.size()) {
            e0 = g_prof.events[g_prof.used]; e1 = g_prof.events[g_prof.used + 1];
            g_prof.used += 2;
            g_prof.flops += flops;
            g_prof.recs.push_back({10, BM, BN, p.M, p.N, p.K, splits, 1, flops});
        }
    }
    if (e0) (void)hipEventRecord(e0, st);
    if constexpr (BM == 64) {
        // opt-in split-bf16 variant: 64-row tiles with at most 512 patch positions and 16-channel groups
        if (p.math == SEMPYR_MATH_BF16X3 && nsp == 2 && p.C % BCG == 0) {
            const size_t lds16 = (size_t)(2 * 9 * 2 * BM + 2 * 2 * 512) * 16;
            hipLaunchKernelGGL((patch3x3_bf16x3_kernel<BM, BN, WM, WN>), grid, dim3(256), lds16, st, pp);
            if (e1) (void)hipEventRecord(e1, st);
            SEMPYR_CHECK_LAUNCH();
            return 0;
        }
    }
    if (nsp == 2) hipLaunchKernelGGL((patch3x3_kernel<BM, BN, WM, WN, 2>), grid, dim3(256), lds, st, pp);
    else if (nsp == 3) hipLaunchKernelGGL((patch3x3_kernel<BM, BN, WM, WN, 3>), grid, dim3(256), lds, st, pp);
    else hipLaunchKernelGGL((patch3x3_kernel<BM, BN, WM, WN, 4>), grid, dim3(256), lds, st, pp);
    if (e1) (void)hipEventRecord(e1, st);
    SEMPYR_CHECK_LAUNCH();
    return 0;
}

// The tile configuration the patch kernel would run this problem with, or -1 when it does not take it (fills p.PR/PW/CW).
int patch_effective_cfg(const sempyr_conv_desc* d, int mode, int cfg, IgemmParams& p, int splits) {
    static const int enabled = env_int("SEMPYR_PATCH", 1);
    if (!enabled) return -1;
    if (cfg == CFG_64x256 && splits == 1) {                  // 64-row problems: narrower tiles when that evens out the rounds
        // 64x256 tiles run 3 per CU (768 at a time), 64x128 tiles 4 per CU (1024): take the shape whose last round is
        // fuller (N = 2^18: 1024 tiles = 1.33 rounds vs 2048 = 2.0 rounds; measured 0.204 -> 0.179 ms)
        static const int force = env_int("SEMPYR_PATCH_BN", 0);
        const long long t256 = (long long)(p.Mpad / 64) * (p.N / 256), t128 = 2 * t256;
        const double e256 = (double)t256 / (double)((t256 + 767) / 768 * 768);
        const double e128 = (double)t128 / (double)((t128 + 1023) / 1024 * 1024);
        if (force == 128 || (force == 0 && p.N % 256 == 0 && e128 > e256 + 0.15)) cfg = CFG_64x128;
    }
    int bm, bn; cfg_dims(cfg, bm, bn);
    if (cfg != CFG_128x128 && cfg != CFG_64x256 && cfg != CFG_64x64 && cfg != CFG_64x128) return -1;
    if (!patch3x3_ok(d, mode, bm, bn, p)) return -1;
    return cfg;
}

// pixels per statistics group of the patch kernel's BatchNorm-statistics epilogue (TN * 32 of the configuration)
inline int patch_stats_granule(int cfg) { return cfg == CFG_64x64 ? 32 : 64; }

// returns 1 if the launch was taken by the patch kernel, 0 if not eligible, < 0 / hipError on failure
int try_patch(const sempyr_conv_desc* d, int mode, int cfg, IgemmParams& p, int splits, hipStream_t st) {
    cfg = patch_effective_cfg(d, mode, cfg, p, splits);
    if (cfg < 0) return 0;
    int rc;
    if (cfg == CFG_128x128) rc = launch_patch<128, 128, 2, 2>(p, splits, st);
    else if (cfg == CFG_64x256) rc = launch_patch<64, 256, 1, 4>(p, splits, st);
    else if (cfg == CFG_64x128) rc = launch_patch<64, 128, 2, 2>(p, splits, st);
    else rc = launch_patch<64, 64, 2, 2>(p, splits, st);
    return rc ? rc : 1;
}

#include "conv_tpatch_host.h"
#include "conv_wpatch.h"

template <int MODE>
int launch_cfg(int cfg, const IgemmParams& p, int splits, bool uniform_tap, hipStream_t st) {
    if (MODE == MODE_GATHER && p.M <= 4 && splits == 1) return launch_small_m(p, uniform_tap, st);
    return uniform_tap ? launch_cfg_ut<MODE, true>(cfg, p, splits, st) : launch_cfg_ut<MODE, false>(cfg, p, splits, st);
}

} // namespace

namespace {
// Shapes served by rgb_wgrad3x3_kernel; rows per strip / channels per workgroup (SEMPYR_RGBW_ROWS / _CG: tuning aids)
inline int rgbw_rows() { static const int v = env_int("SEMPYR_RGBW_ROWS", 128); return v < 1 ? 1 : v; }
inline int rgbw_cg() { static const int v = env_int("SEMPYR_RGBW_CG", 2); return v == 4 ? 4 : 2; }
inline bool rgb_wgrad_ok(const sempyr_conv_desc* d) {
    static const int enabled = env_int("SEMPYR_RGB_WGRAD", 1);
    return enabled && d->kh == 3 && d->kw == 3 && d->stride == 1 && d->pad == 1 && d->ups == 0 && d->Cout <= 4 &&
           d->Cin % rgbw_cg() == 0 && d->Win <= 256 && 256 % d->Win == 0;
}
}  // namespace

extern "C" int sempyr_conv2d_plan(const sempyr_conv_desc* d, int mode, sempyr_conv_plan* plan)
{
    Geometry g;
    if (int rc = check_desc(d, g)) return rc;
    if (!plan) SEMPYR_FAIL("conv plan: null plan");
    if (mode < 0 || mode > 2) SEMPYR_FAIL("conv plan: bad mode %d", mode);
    *plan = sempyr_conv_plan{};
    plan->mode = mode; plan->Ho = g.Ho; plan->Wo = g.Wo;
    const long long in_numel = (long long)d->B * d->Cin * g.GHin * g.GWin;
    const long long out_numel = (long long)d->B * d->Cout * g.Ho * g.Wo;
    int bm, bn;
    if (mode == SEMPYR_CONV_FWD) {
        plan->nphases = 1;
        plan->K[0] = d->Cin * g.T; plan->Kpad[0] = (plan->K[0] + BK - 1) / BK * BK;
        plan->tab_off[0] = 0; plan->tab_entries = plan->Kpad[0];
        plan->Mpad = (d->Cout + 31) / 32 * 32;
        plan->cfg = pick_cfg(plan->Mpad, (long long)d->B * g.Ho * g.Wo, conv_kind(d, mode));
        { static const int f = env_int("SEMPYR_FORCE_CFG", -1); if (f >= 0 && plan->Mpad % (f == CFG_128x128 ? 128 : f == CFG_32x256 ? 32 : 64) == 0) plan->cfg = f; }
        cfg_dims(plan->cfg, bm, bn);
        plan->wpack_floats = (long long)plan->Kpad[0] * plan->Mpad;
        const long long N = (long long)d->B * g.Ho * g.Wo;
        int cps;
        pick_splits(ceil_div(plan->Mpad, bm) * ceil_div(N, bn), plan->Kpad[0] / BK, plan->splits[0], cps);
        if (rgb3x3_ok(d, mode) && N >= 4096) plan->splits[0] = 1;
        TpTile tp;
        if (tp_pick(d, mode, tp)) { plan->cfg = CFG_TPATCH; plan->splits[0] = tp_splits(tp); }
        plan->workspace_bytes = plan->splits[0] > 1 ? (long long)plan->splits[0] * out_numel * 4 : 0;
        if (plan->splits[0] == 1 && !rgb3x3_ok(d, mode) && N < (1ll << 31) && plan->cfg != CFG_TPATCH) {
            // would the patch kernel take it?  (the geometry fields its eligibility test reads)
            IgemmParams q{};
            q.Mpad = plan->Mpad; q.N = (int)N; q.QHW = g.Ho * g.Wo; q.QW = g.Wo; q.QH = g.Ho; q.ups = d->ups;
            const int eff = patch_effective_cfg(d, mode, plan->cfg, q, 1);
            if (eff >= 0 && (g.Ho * g.Wo) % patch_stats_granule(eff) == 0) plan->stats_granule = patch_stats_granule(eff);
        }
    } else if (mode == SEMPYR_CONV_DGRAD) {
        const int s = d->stride;
        plan->nphases = s * s;
        plan->Mpad = (d->Cin + 31) / 32 * 32;
        plan->cfg = pick_cfg(plan->Mpad, in_numel / d->Cin, conv_kind(d, mode));
        { static const int f = env_int("SEMPYR_FORCE_CFG_DGRAD", -1); if (f >= 0 && plan->Mpad % (f == CFG_128x128 ? 128 : f == CFG_32x256 ? 32 : 64) == 0) plan->cfg = f; }
        cfg_dims(plan->cfg, bm, bn);
        int off = 0, max_splits = 1;
        for (int py = 0; py < s; ++py) for (int px = 0; px < s; ++px) {
            const int ph = py * s + px;
            int ry, cy, rx, cx;
            const int nj = phase_taps(d->kh, s, d->pad, py, ry, cy);
            const int ni = phase_taps(d->kw, s, d->pad, px, rx, cx);
            plan->K[ph] = nj * ni * d->Cout;
            plan->Kpad[ph] = (plan->K[ph] + BK - 1) / BK * BK;
            plan->tab_off[ph] = off; off += plan->Kpad[ph];
            const int QH = (g.GHin - py + s - 1) / s, QW = (g.GWin - px + s - 1) / s;
            const long long N = (long long)d->B * QH * QW;
            int cps; plan->splits[ph] = 1;
            // the s*s phases run as one launch: count the workgroups of all of them against the target
            if (plan->K[ph] > 0 && N > 0)
                pick_splits(ceil_div(plan->Mpad, bm) * ceil_div(N, bn) * s * s, plan->Kpad[ph] / BK, plan->splits[ph], cps);
            if (rgb3x3_ok(d, mode) && N >= 4096) plan->splits[ph] = 1;
            if (plan->splits[ph] > max_splits) max_splits = plan->splits[ph];
        }
        TpTile tp;
        if (tp_pick(d, mode, tp)) {
            plan->cfg = CFG_TPATCH; max_splits = tp_splits(tp);
            for (int ph = 0; ph < s * s; ++ph) plan->splits[ph] = max_splits;
        }
        plan->tab_entries = off;
        plan->wpack_floats = (long long)off * plan->Mpad;
        plan->workspace_bytes = max_splits > 1 ? (long long)max_splits * in_numel * 4 : 0;
    } else {
        plan->nphases = 1;
        plan->K[0] = d->B * g.Ho * g.Wo;                        // reduction over pixels
        plan->Kpad[0] = (plan->K[0] + BK - 1) / BK * BK;
        const int Nn = (d->Cin * g.T + BK - 1) / BK * BK;       // = FWD Kpad: columns incl. padding
        plan->tab_off[0] = 0; plan->tab_entries = Nn;
        plan->Mpad = (d->Cout + 31) / 32 * 32;
        plan->cfg = pick_cfg_wgrad(plan->Mpad, Nn, d->Cin % BK == 0);
        // mid-size reductions (<= 2^17 pixels: the critic's blocks, the deep generator / classifier layers): 64x128 tiles
        // double the tile count, i.e. halve the split-K depth and the slabs the combine pass reads (measured 4-10 % on
        // the critic's 64->64 and 64->128 4x4 blocks, neutral elsewhere)
        static const int w128 = env_int("SEMPYR_WGRAD_64x128", 1);
        if (w128 && (plan->cfg == CFG_128x128 || plan->cfg == CFG_64x256) && plan->K[0] <= (1 << 17) && Nn % 128 == 0)
            plan->cfg = CFG_64x128;
        { static const int f = env_int("SEMPYR_FORCE_CFG_WGRAD", -1); if (f >= 0 && f != CFG_64x192 && plan->Mpad % (f == CFG_128x128 ? 128 : f == CFG_32x256 ? 32 : 64) == 0) plan->cfg = f; }
        cfg_dims(plan->cfg, bm, bn);
        plan->wpack_floats = 0;
        int cps;
        pick_splits(ceil_div(plan->Mpad, bm) * ceil_div(Nn, bn), plan->Kpad[0] / BK, plan->splits[0], cps, target_blocks_wgrad());
        WpPick wp;
        if (!rgb_wgrad_ok(d) && wp_pick(d, g.Ho, g.Wo, wp)) { plan->cfg = CFG_WPATCH; plan->splits[0] = wp.nslabs; }
        plan->workspace_bytes = (long long)plan->splits[0] * d->Cout * Nn * 4;
        if (rgb_wgrad_ok(d)) {
            const long long need = (long long)d->B * ceil_div(d->Hin, rgbw_rows()) * d->Cin * d->Cout * 9 * 4;
            if (need > plan->workspace_bytes) plan->workspace_bytes = need;
        }
    }
    return 0;
}

extern "C" int sempyr_conv2d_tables(const sempyr_conv_desc* d, int mode, int32_t* ktab, int32_t* wsrc)
{
    Geometry g;
    if (int rc = check_desc(d, g)) return rc;
    if (!ktab || !wsrc) SEMPYR_FAIL("conv tables: null output");
    const int INVALID = (int)0x80008000;                          // dy = dx = -32768
    const int T = g.T;
    if (mode == SEMPYR_CONV_FWD || mode == SEMPYR_CONV_WGRAD) {
        const int K = d->Cin * T, Kpad = (K + BK - 1) / BK * BK;
        for (int k = 0; k < Kpad; ++k) {
            if (k < K) {
                const int tap = k / d->Cin, ci = k % d->Cin;
                const int ky = tap / d->kw, kx = tap % d->kw;
                const int dy = ky - d->pad, dx = kx - d->pad;
                ktab[2 * k] = ci * d->Hin * d->Win;
                ktab[2 * k + 1] = (int)(((uint32_t)(dy & 0xffff) << 16) | (uint32_t)(dx & 0xffff));
                wsrc[k] = ci * T + tap;
            } else { ktab[2 * k] = 0; ktab[2 * k + 1] = INVALID; wsrc[k] = -1; }
        }
        return 0;
    }
    if (mode != SEMPYR_CONV_DGRAD) SEMPYR_FAIL("conv tables: bad mode %d", mode);
    const int s = d->stride;
    int off = 0;
    for (int py = 0; py < s; ++py) for (int px = 0; px < s; ++px) {
        int ry, cy, rx, cx;
        const int nj = phase_taps(d->kh, s, d->pad, py, ry, cy);
        const int ni = phase_taps(d->kw, s, d->pad, px, rx, cx);
        const int K = nj * ni * d->Cout, Kpad = (K + BK - 1) / BK * BK;
        for (int k = 0; k < Kpad; ++k) {
            const int e = off + k;
            if (k < K) {
                const int tp = k / d->Cout, co = k % d->Cout;
                const int j = tp / ni, i = tp % ni;
                const int ky = ry + s * j, kx = rx + s * i;
                const int dy = cy - j, dx = cx - i;
                ktab[2 * e] = co * g.Ho * g.Wo;
                ktab[2 * e + 1] = (int)(((uint32_t)(dy & 0xffff) << 16) | (uint32_t)(dx & 0xffff));
                wsrc[e] = co * d->Cin * T + ky * d->kw + kx;
            } else { ktab[2 * e] = 0; ktab[2 * e + 1] = INVALID; wsrc[e] = -1; }
        }
        off += Kpad;
    }
    return 0;
}

extern "C" int sempyr_conv2d_pack_weights(const sempyr_conv_desc* d, int mode, const float* w,
                                          const int32_t* wsrc_dev, float* wpacked, void* stream)
{
    sempyr_conv_plan pl;
    if (int rc = sempyr_conv2d_plan(d, mode, &pl)) return rc;
    if (mode == SEMPYR_CONV_WGRAD) SEMPYR_FAIL("pack_weights: WGRAD has no packed weights");
    if (!w || !wsrc_dev || !wpacked) SEMPYR_FAIL("pack_weights: null pointer");
    const int T = d->kh * d->kw;
    const int M = mode == SEMPYR_CONV_FWD ? d->Cout : d->Cin;
    const int mstride = mode == SEMPYR_CONV_FWD ? d->Cin * T : T;
    const long long total = (long long)pl.tab_entries * pl.Mpad;
    if (total == 0) return 0;
    hipLaunchKernelGGL(pack_weights_kernel, dim3(ceil_div(total, 256)), dim3(256), 0, (hipStream_t)stream,
                       w, wsrc_dev, wpacked, pl.tab_entries, pl.Mpad, M, mstride);
    SEMPYR_CHECK_LAUNCH();
    return 0;
}

extern "C" int sempyr_conv2d_pack_weights_many(const sempyr_pack_item* items, int n_items, void* stream)
{
    if (!items || n_items < 1) SEMPYR_FAIL("pack_weights_many: empty item list");
    PackManyArgs a;
    int n = 0; long long blocks = 0;
    auto flush = [&]() -> int {
        if (n == 0) return 0;
        a.first_block[n] = (int)blocks; a.n = n;
        HbmScope hs("pack_weights_many_kernel", 2048.0 * (double)blocks, (hipStream_t)stream);
        hipLaunchKernelGGL(pack_weights_many_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, a);
        SEMPYR_CHECK_LAUNCH();
        n = 0; blocks = 0;
        return 0;
    };
    for (int i = 0; i < n_items; ++i) {
        const sempyr_pack_item& it = items[i];
        sempyr_conv_plan pl;
        if (int rc = sempyr_conv2d_plan(&it.desc, it.mode, &pl)) return rc;
        if (it.mode == SEMPYR_CONV_WGRAD) SEMPYR_FAIL("pack_weights_many: WGRAD has no packed weights");
        if (!it.w || !it.wsrc_dev || !it.wpacked) SEMPYR_FAIL("pack_weights_many: item %d has a null pointer", i);
        const int T = it.desc.kh * it.desc.kw;
        const long long total = (long long)pl.tab_entries * pl.Mpad;
        if (total == 0) continue;
        if (total >= (1ll << 31)) SEMPYR_FAIL("pack_weights_many: item %d too large", i);
        a.w[n] = it.w; a.wsrc[n] = it.wsrc_dev; a.wp[n] = it.wpacked;
        a.first_block[n] = (int)blocks; a.total[n] = (int)total; a.Mpad[n] = pl.Mpad;
        a.M[n] = it.mode == SEMPYR_CONV_FWD ? it.desc.Cout : it.desc.Cin;
        a.mstride[n] = it.mode == SEMPYR_CONV_FWD ? it.desc.Cin * T : T;
        blocks += (total + 255) / 256; ++n;
        if (n == PACK_MAX || blocks > (1ll << 30)) { if (int rc = flush()) return rc; }
    }
    return flush();
}

static int run_splitk_reduce(const float* slabs, float* out, const float* bias, int splits, long long numel,
                             int OHW, int OC, int act, float slope, hipStream_t st,
                             const float* scale = nullptr, const float* res = nullptr, int res_w = 0)
{
    // float4 path only when every slab start (k * numel floats) stays 16-byte aligned
    const bool res_ok = res_w > 0 ? (res_w % 4 == 0 && ((uintptr_t)res & 7) == 0) : (((uintptr_t)res & 15) == 0);
    HbmScope hs("splitk_reduce_kernel", 4.0 * (double)numel * (splits + 1.0 + (res ? (res_w > 0 ? 0.25 : 1.0) : 0.0)), st);
    if (numel % 4 == 0 && (((uintptr_t)slabs | (uintptr_t)out) & 15) == 0 && res_ok) {
        const int total4 = (int)(numel / 4);
        hipLaunchKernelGGL(splitk_reduce_kernel, dim3(ceil_div(total4, 256)), dim3(256), 0, st,
                           slabs, out, bias, splits, numel, total4, OHW, OC, act, slope, scale, res, res_w);
    } else {
        hipLaunchKernelGGL(splitk_reduce_tail_kernel, dim3(ceil_div(numel, 256)), dim3(256), 0, st,
                           slabs, out, bias, splits, numel, 0, (int)numel, OHW, OC, act, slope, scale, res, res_w);
    }
    SEMPYR_CHECK_LAUNCH();
    return 0;
}

int sempyr_stem_try_launch(const sempyr_conv_desc* d, const float* x, const float* w_packed, int Mpad, int Kpad, const float* bias,
                           int act, float slope, float* y_pooled, uint32_t* act_bits, const uint32_t* mask_bits, void* stream);

static int conv2d_fwd_impl(const sempyr_conv_desc* d, const float* x, const float* wp,
                           const int32_t* ktab, const float* bias, const float* scale, const float* res,
                           int res_half, int act, float slope, float* y,
                           float* y_pooled, uint32_t* act_bits, const uint32_t* mask_bits, void* ws, size_t ws_bytes, void* stream,
                           float* tile_stats)
{
    sempyr_conv_plan pl;
    if (int rc = sempyr_conv2d_plan(d, SEMPYR_CONV_FWD, &pl)) return rc;
    if (!x || !wp || !ktab || (!y && !y_pooled)) SEMPYR_FAIL("conv fwd: null pointer");
    if (act < SEMPYR_ACT_NONE || act > SEMPYR_ACT_TANH) SEMPYR_FAIL("conv fwd: bad activation %d", act);
    if ((size_t)pl.workspace_bytes > ws_bytes || (pl.workspace_bytes && !ws))
        SEMPYR_FAIL("conv fwd: workspace too small (%zu < %lld)", ws_bytes, (long long)pl.workspace_bytes);
    hipStream_t st = (hipStream_t)stream;
    const long long out_numel = (long long)d->B * d->Cout * pl.Ho * pl.Wo;
    IgemmParams p{};
    p.math = d->math;
    p.A = wp; p.X = x; p.ktab = reinterpret_cast<const int2*>(ktab);
    p.M = d->Cout; p.Mpad = pl.Mpad; p.N = d->B * pl.Ho * pl.Wo; p.K = pl.K[0]; p.Kpad = pl.Kpad[0];
    p.XCHW = d->Cin * d->Hin * d->Win; p.XW = d->Win; p.XHW = d->Hin * d->Win;
    p.GH = d->Hin << d->ups; p.GW = d->Win << d->ups; p.ups = d->ups;
    p.SY = p.SX = d->stride;
    p.QH = pl.Ho; p.QW = pl.Wo; p.QHW = pl.Ho * pl.Wo; p.dQHW.init(p.QHW); p.dQW.init(p.QW);
    p.OCHW = d->Cout * pl.Ho * pl.Wo; p.OHW = pl.Ho * pl.Wo; p.OW = pl.Wo;
    p.OSY = p.OSX = 1; p.OPY = p.OPX = 0;
    p.nchunks = pl.Kpad[0] / BK;
    const int splits = pl.splits[0];
    p.chunks_per_split = ceil_div(p.nchunks, splits);
    p.C = d->Cin;
    p.x_bytes = d->B * p.XCHW * 4; p.a_bytes = (int)(pl.wpack_floats * 4);
    if ((scale || res) && (d->Cout <= 4 || y_pooled))
        SEMPYR_FAIL("conv fwd: the scale / residual epilogue is not available for RGB heads or pooled outputs");
    if (res_half && (!res || ((pl.Ho | pl.Wo) & 1))) SEMPYR_FAIL("conv fwd: a half-resolution residual needs even output height and width");
    if (splits == 1) { p.Out = y; p.bias = bias; p.scale = scale; p.res = res; p.res_half = res_half ? 1 : 0; p.slab_stride = 0; p.act = act; p.slope = slope; }
    else { p.Out = (float*)ws; p.bias = nullptr; p.slab_stride = out_numel; }
    if (y_pooled && ((pl.Ho | pl.Wo) & 1)) SEMPYR_FAIL("conv fwd: pooled output needs even output height and width");
    if (y_pooled && act == SEMPYR_ACT_TANH) SEMPYR_FAIL("conv fwd: pooled output is defined for ReLU-family activations");
    const bool fused_pool = y_pooled && splits == 1 && d->Cout > 4;          // MFMA single-pass launch: pool in the epilogue
    if (fused_pool) { p.Pool = y_pooled; p.quad = 1; }
    if (act_bits || mask_bits || !y) {
        if (!fused_pool || (pl.Ho * pl.Wo) % 32 != 0)
            SEMPYR_FAIL("conv fwd: sign bits / a pooled-only output need the fused pooled epilogue (single pass, Cout > 4, Ho*Wo %% 32 == 0); "
                        "check sempyr_conv2d_plan().splits");
        if (act_bits && !(act == SEMPYR_ACT_RELU || act == SEMPYR_ACT_LRELU)) SEMPYR_FAIL("conv fwd: sign bits are defined for ReLU / LeakyReLU");
        if (mask_bits && (act != SEMPYR_ACT_NONE || bias || act_bits)) SEMPYR_FAIL("conv fwd: mask bits replace bias / activation / sign-bit output");
        p.Bits = act_bits; p.MaskBits = mask_bits;
        if (mask_bits) p.slope = slope;
        if (!y) {                       // pooled map (+ bits) only: the critic stem has a kernel of its own (stem.hip)
            const int took = sempyr_stem_try_launch(d, x, wp, pl.Mpad, pl.Kpad[0], bias, act, slope, y_pooled, act_bits, mask_bits, stream);
            if (took) return took == 1 ? 0 : took;
        }
    }
    int rc = 0;
    if (tile_stats) {
        if (!pl.stats_granule || fused_pool || !y) SEMPYR_FAIL("conv fwd: this shape has no statistics epilogue (sempyr_conv2d_plan().stats_granule == 0)");
        p.Stats = reinterpret_cast<float2*>(tile_stats);
    }
    TpTile tp;
    int cfg = pl.cfg;
    if (cfg == CFG_TPATCH) {
        // epilogues the 4x4 patch kernel does not have fall back to the generic kernel (any tile works with the plan's splits)
        if (y && !y_pooled && !act_bits && !mask_bits && !tile_stats && !res_half && tp_pick(d, SEMPYR_CONV_FWD, tp)) {
            if ((rc = tp_launch(d, tp, p, nullptr, splits, st))) return rc;
            if (splits > 1)
                return run_splitk_reduce((const float*)ws, y, bias, splits, out_numel, pl.Ho * pl.Wo, d->Cout, act, slope, st, scale, res, 0);
            return 0;
        }
        cfg = pick_cfg(pl.Mpad, (long long)d->B * pl.Ho * pl.Wo, conv_kind(d, SEMPYR_CONV_FWD));
    }
    if (splits == 1 && rgb3x3_ok(d, SEMPYR_CONV_FWD)) rc = launch_rgb3x3(p, st);
    else {
        rc = fused_pool ? 0 : try_patch(d, SEMPYR_CONV_FWD, cfg, p, splits, st);
        if (rc == 0 && tile_stats) SEMPYR_FAIL("conv fwd: internal: the statistics epilogue was planned but the patch kernel declined");
        if (rc == 0) rc = launch_cfg<MODE_GATHER>(cfg, p, splits, d->Cin % BK == 0, st);
        else if (rc == 1) rc = 0;
    }
    if (rc) return rc;
    if (splits > 1)
        if ((rc = run_splitk_reduce((const float*)ws, y, bias, splits, out_numel, pl.Ho * pl.Wo, d->Cout, act, slope, st, scale, res, res_half ? pl.Wo : 0))) return rc;
    if (y_pooled && !fused_pool)
        return sempyr_pool2_down(y, nullptr, nullptr, y_pooled, d->B * d->Cout, pl.Ho, pl.Wo, 0.25f, SEMPYR_ACT_NONE, 0.f, stream);
    return 0;
}

extern "C" int sempyr_conv2d_fwd(const sempyr_conv_desc* d, const float* x, const float* wp,
                                 const int32_t* ktab, const float* bias, const float* scale, const float* res,
                                 int res_half, int act, float slope, float* y,
                                 float* y_pooled, uint32_t* act_bits, const uint32_t* mask_bits, void* ws, size_t ws_bytes, void* stream)
{
    return conv2d_fwd_impl(d, x, wp, ktab, bias, scale, res, res_half, act, slope, y, y_pooled, act_bits, mask_bits, ws, ws_bytes,
                           stream, nullptr);
}

extern "C" int sempyr_conv2d_fwd_stats(const sempyr_conv_desc* d, const float* x, const float* wp, const int32_t* ktab,
                                       float* y, float* tile_stats, void* stream)
{
    if (!tile_stats) SEMPYR_FAIL("conv fwd stats: null statistics buffer");
    return conv2d_fwd_impl(d, x, wp, ktab, nullptr, nullptr, nullptr, 0, SEMPYR_ACT_NONE, 0.f, y, nullptr, nullptr, nullptr,
                           nullptr, 0, stream, tile_stats);
}

static int conv2d_dgrad_impl(const sempyr_conv_desc* d, const float* gy, const float* wp,
                             const int32_t* ktab, const float* bias, float* gx,
                             void* ws, size_t ws_bytes, void* stream,
                             const float* g_pooled, const uint32_t* act_bits, float slope)
{
    sempyr_conv_plan pl;
    if (int rc = sempyr_conv2d_plan(d, SEMPYR_CONV_DGRAD, &pl)) return rc;
    if ((!gy && !g_pooled) || !wp || !ktab || !gx) SEMPYR_FAIL("conv dgrad: null pointer");
    if (g_pooled && (!act_bits || !rgb3x3_ok(d, SEMPYR_CONV_DGRAD) || pl.Wo != 256 || (pl.Ho & 1) || pl.splits[0] != 1))
        SEMPYR_FAIL("conv dgrad: the un-pooled form is built for the critic stem (3x3 s1 p1, <= 4 input channels, 256-pixel rows)");
    if ((size_t)pl.workspace_bytes > ws_bytes || (pl.workspace_bytes && !ws))
        SEMPYR_FAIL("conv dgrad: workspace too small (%zu < %lld)", ws_bytes, (long long)pl.workspace_bytes);
    hipStream_t st = (hipStream_t)stream;
    const int s = d->stride;
    const int GH = d->Hin << d->ups, GW = d->Win << d->ups;
    const long long in_numel = (long long)d->B * d->Cin * GH * GW;
    int splits = 1; bool empty_phase = false;
    for (int ph = 0; ph < pl.nphases; ++ph) {
        if (pl.splits[ph] > splits) splits = pl.splits[ph];
        if (pl.K[ph] == 0) empty_phase = true;
    }
    float* target = splits > 1 ? (float*)ws : gx;
    if (empty_phase) {
        hipError_t e = hipMemsetAsync(target, 0, (size_t)(splits > 1 ? splits : 1) * in_numel * 4, st);
        if (e != hipSuccess) { sempyr_set_error("conv dgrad: memset failed: %s", hipGetErrorString(e)); return (int)e; }
    }
    // all stride phases in ONE launch (they share M, the weight layout and the split count)
    IgemmParams p{};
    p.math = d->math;
    p.A = wp; p.X = gy; p.ktab = reinterpret_cast<const int2*>(ktab);
    p.M = d->Cin; p.Mpad = pl.Mpad;
    p.XCHW = d->Cout * pl.Ho * pl.Wo; p.XW = pl.Wo; p.XHW = pl.Ho * pl.Wo; p.GH = pl.Ho; p.GW = pl.Wo; p.ups = 0;
    p.SY = p.SX = 1;
    p.OCHW = d->Cin * GH * GW; p.OHW = GH * GW; p.OW = GW;
    p.OSY = p.OSX = s;
    p.Out = target; p.bias = splits == 1 ? bias : nullptr; p.slab_stride = splits > 1 ? in_numel : 0;
    p.nphases = pl.nphases;
    if (g_pooled) { p.GP = g_pooled; p.GBits = reinterpret_cast<const unsigned char*>(act_bits); p.gslope = slope; p.X = g_pooled; }
    int live = 0;
    for (int py = 0; py < s; ++py) for (int px = 0; px < s; ++px) {
        const int ph = py * s + px;
        const int QH = (GH - py + s - 1) / s, QW = (GW - px + s - 1) / s;
        IgemmParams::Phase& q = p.ph[ph];
        q.a_off = (long long)pl.tab_off[ph] * pl.Mpad;
        q.tab_off = pl.tab_off[ph];
        q.N = (pl.K[ph] == 0 || QH <= 0 || QW <= 0) ? 0 : d->B * QH * QW;     // N = 0: the phase's workgroups exit at once
        q.nchunks = pl.Kpad[ph] / BK;
        q.QW = QW > 0 ? QW : 1; q.QHW = (QH > 0 && QW > 0) ? QH * QW : 1;
        q.OPY = py; q.OPX = px;
        if (q.N > 0) {
            ++live;
            // phase 0 fields double as the single-phase view (and as the record for the profiler)
            if (live == 1 || q.N > p.N) { p.N = q.N; p.K = pl.K[ph]; p.Kpad = pl.Kpad[ph]; }
        }
    }
    if (live > 0) {
        if (p.nphases == 1) {                                      // stride 1: plain single-phase launch
            p.A = wp + p.ph[0].a_off; p.ktab += p.ph[0].tab_off;
            p.N = p.ph[0].N; p.nchunks = p.ph[0].nchunks; p.QW = p.ph[0].QW; p.QHW = p.ph[0].QHW;
            p.QH = p.QHW / p.QW; p.OPY = 0; p.OPX = 0;
            p.chunks_per_split = ceil_div(p.nchunks, splits);
        }
        p.C = d->Cout;
        p.x_bytes = d->B * p.XCHW * 4; p.a_bytes = (int)(pl.wpack_floats * 4);
        if (splits == 1 && rgb3x3_ok(d, SEMPYR_CONV_DGRAD)) return launch_rgb3x3(p, st);
        int cfg = pl.cfg;
        TpTile tp;
        if (cfg == CFG_TPATCH) {
            if (!g_pooled && tp_pick(d, SEMPYR_CONV_DGRAD, tp)) {
                if (int rc = tp_launch(d, tp, p, pl.tab_off, splits, st)) return rc;
                if (splits > 1) return run_splitk_reduce((const float*)ws, gx, bias, splits, in_numel, GH * GW, d->Cin, SEMPYR_ACT_NONE, 0.f, st);
                return 0;
            }
            cfg = pick_cfg(pl.Mpad, in_numel / d->Cin, conv_kind(d, SEMPYR_CONV_DGRAD));
        }
        int rc = p.nphases == 1 ? try_patch(d, SEMPYR_CONV_DGRAD, cfg, p, splits, st) : 0;
        if (rc == 0) rc = launch_cfg<MODE_GATHER>(cfg, p, splits, d->Cout % BK == 0, st);
        else if (rc == 1) rc = 0;
        if (rc) return rc;
    }
    if (splits > 1) return run_splitk_reduce((const float*)ws, gx, bias, splits, in_numel, GH * GW, d->Cin, SEMPYR_ACT_NONE, 0.f, st);
    return 0;
}

extern "C" int sempyr_conv2d_dgrad(const sempyr_conv_desc* d, const float* gy, const float* wp,
                                   const int32_t* ktab, const float* bias, float* gx,
                                   void* ws, size_t ws_bytes, void* stream)
{
    return conv2d_dgrad_impl(d, gy, wp, ktab, bias, gx, ws, ws_bytes, stream, nullptr, nullptr, 0.f);
}

extern "C" int sempyr_conv2d_dgrad_unpooled(const sempyr_conv_desc* d, const float* g_pooled, const uint32_t* act_bits, float slope,
                                            const float* wp, const int32_t* ktab, float* gx, void* stream)
{
    return conv2d_dgrad_impl(d, nullptr, wp, ktab, nullptr, gx, nullptr, 0, stream, g_pooled, act_bits, slope);
}

extern "C" int sempyr_conv2d_wgrad(const sempyr_conv_desc* d, const float* gy, const float* x,
                                   const int32_t* ktab, float* gw, int accumulate,
                                   void* ws, size_t ws_bytes, void* stream)
{
    sempyr_conv_plan pl;
    if (int rc = sempyr_conv2d_plan(d, SEMPYR_CONV_WGRAD, &pl)) return rc;
    if (!gy || !x || !ktab || !gw || !ws) SEMPYR_FAIL("conv wgrad: null pointer");
    if ((size_t)pl.workspace_bytes > ws_bytes)
        SEMPYR_FAIL("conv wgrad: workspace too small (%zu < %lld)", ws_bytes, (long long)pl.workspace_bytes);
    hipStream_t st = (hipStream_t)stream;
    const int T = d->kh * d->kw;
    const int Nn = pl.tab_entries;
    if (rgb_wgrad_ok(d)) {
        const int RGBW_ROWS = rgbw_rows(), RGBW_CG = rgbw_cg();
        const int nstrips = d->B * ceil_div(d->Hin, RGBW_ROWS);
        dim3 grid((unsigned)(nstrips * (d->Cin / RGBW_CG)));
        float* part = (float*)ws;
        hipEvent_t e0 = nullptr, e1 = nullptr;
        if (g_prof.on) {
            std::lock_guard<std::mutex> lk(g_prof.mu);
            if (g_prof.used + 2 <= g_prof.events.size()) {
                e0 = g_prof.events[g_prof.used]; e1 = g_prof.events[g_prof.used + 1];
                g_prof.used += 2;
                const double flops = 2.0 * (double)d->Cout * (double)(d->Cin * T) * (double)d->B * d->Hin * d->Win;
                g_prof.flops += flops;
                g_prof.recs.push_back({9, 4, 256, d->Cout, d->Cin * T, d->B * d->Hin * d->Win, 1, 1, flops});
            }
        }
        if (e0) (void)hipEventRecord(e0, st);
#define SEMPYR_RGBW(NM) do { \
            if (RGBW_CG == 2) hipLaunchKernelGGL((rgb_wgrad3x3_kernel<NM, 2>), grid, dim3(256), 0, st, gy, x, d->Cin, d->Hin, d->Win, RGBW_ROWS, part); \
            else hipLaunchKernelGGL((rgb_wgrad3x3_kernel<NM, 4>), grid, dim3(256), 0, st, gy, x, d->Cin, d->Hin, d->Win, RGBW_ROWS, part); } while (0)
        switch (d->Cout) {
            case 1: SEMPYR_RGBW(1); break;
            case 2: SEMPYR_RGBW(2); break;
            case 3: SEMPYR_RGBW(3); break;
            default: SEMPYR_RGBW(4); break;
        }
#undef SEMPYR_RGBW
        if (e1) (void)hipEventRecord(e1, st);
        SEMPYR_CHECK_LAUNCH();
        hipLaunchKernelGGL(rgbw_reduce_kernel, dim3(ceil_div(d->Cin * d->Cout * 9, 256)), dim3(256), 0, st,
                           (const float*)part, nstrips, d->Cin, d->Cout, gw, accumulate ? 1 : 0);
        SEMPYR_CHECK_LAUNCH();
        return 0;
    }
    const int splits = pl.splits[0];
    const long long slab_stride = (long long)d->Cout * Nn;
    if (pl.cfg == CFG_WPATCH) {
        WpPick wp;
        if (!wp_pick(d, pl.Ho, pl.Wo, wp) || wp.nslabs != splits) SEMPYR_FAIL("conv wgrad: plan and launch disagree");
        if (int rc = wp_launch(d, wp, pl.Ho, pl.Wo, gy, x, (float*)ws, st)) return rc;
    } else {
    IgemmParams p{};
    p.A = gy; p.X = x; p.ktab = reinterpret_cast<const int2*>(ktab);
    p.M = d->Cout; p.Mpad = pl.Mpad; p.N = Nn; p.K = pl.K[0]; p.Kpad = pl.Kpad[0];
    p.XCHW = d->Cin * d->Hin * d->Win; p.XW = d->Win; p.XHW = d->Hin * d->Win;
    p.GH = d->Hin << d->ups; p.GW = d->Win << d->ups; p.ups = d->ups;
    p.SY = p.SX = d->stride;
    p.QH = pl.Ho; p.QW = pl.Wo; p.QHW = pl.Ho * pl.Wo; p.dQHW.init(p.QHW); p.dQW.init(p.QW);
    p.ldc = Nn;
    p.nchunks = pl.Kpad[0] / BK;
    p.chunks_per_split = ceil_div(p.nchunks, splits);
    p.Out = (float*)ws; p.slab_stride = slab_stride;
    p.x_bytes = d->B * p.XCHW * 4; p.a_bytes = d->B * d->Cout * pl.Ho * pl.Wo * 4;
    if (int rc = launch_cfg<MODE_WGRAD>(pl.cfg, p, splits, d->Cin % BK == 0, st)) return rc;
    }
    const int total = d->Cout * d->Cin * T;
    static const int reduce_t = env_int("SEMPYR_WGRAD_REDUCE_T", 1);
    HbmScope hs("wgrad_reduce_kernel (all forms)", 4.0 * (double)total * (splits + 1.0 + (accumulate ? 1.0 : 0.0)), st);
    if (splits >= 64)
        hipLaunchKernelGGL(wgrad_reduce_many_kernel, dim3(ceil_div(total, 16)), dim3(256), 0, st,
                           (const float*)ws, gw, splits, slab_stride, d->Cout, d->Cin, T, Nn, accumulate ? 1 : 0);
    else if (reduce_t && T > 1 && d->Cin >= 32 && d->Cout * ceil_div(d->Cin, 64) >= 512)     // fewer workgroups: one thread per output wins (measured)
        hipLaunchKernelGGL(wgrad_reduce_t_kernel, dim3(d->Cout * ceil_div(d->Cin, 64)), dim3(256), (size_t)64 * (T + 1) * sizeof(float), st,
                           (const float*)ws, gw, splits, slab_stride, d->Cin, T, Nn, accumulate ? 1 : 0);
    else
        hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(ceil_div(total, 256)), dim3(256), 0, st,
                           (const float*)ws, gw, splits, slab_stride, d->Cout, d->Cin, T, Nn, accumulate ? 1 : 0);
    SEMPYR_CHECK_LAUNCH();
    return 0;
}

// ------------------------------------------------------------------------------------------
// kernel-timing hooks (measurement only; never enabled by the product path)
// ------------------------------------------------------------------------------------------
// a named do-nothing kernel: brackets the timed region in a rocprofv3 kernel trace (tools/roofline_table.py)
extern "C" __global__ void sempyr_marker_kernel(int tag) { (void)tag; }

extern "C" int sempyr_profile_marker(int tag, void* stream)
{
    hipLaunchKernelGGL(sempyr_marker_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, tag);
    SEMPYR_CHECK_LAUNCH();
    return 0;
}

extern "C" int sempyr_profile_begin(int max_launches)
{
    if (max_launches < 1) SEMPYR_FAIL("profile_begin: max_launches must be positive");
    std::lock_guard<std::mutex> lk(g_prof.mu);
    while (g_prof.events.size() < (size_t)max_launches * 2) {
        hipEvent_t e;
        hipError_t rc = hipEventCreate(&e);
        if (rc != hipSuccess) { sempyr_set_error("profile_begin: hipEventCreate: %s", hipGetErrorString(rc)); return (int)rc; }
        g_prof.events.push_back(e);
    }
    g_prof.used = 0; g_prof.flops = 0.0; g_prof.recs.clear(); g_prof.on = true;
    sempyr_hbm_profile_begin(max_launches);
    return 0;
}

extern "C" int sempyr_profile_end(double* kernel_ms, int64_t* launches, double* flops)
{
    sempyr_hbm_profile_end();
    std::lock_guard<std::mutex> lk(g_prof.mu);
    g_prof.on = false;
    double ms = 0.0;
    const char* dump = getenv("SEMPYR_PROFILE_DUMP");      // optional per-launch CSV (tuning aid)
    FILE* df = dump ? fopen(dump, "w") : nullptr;
    if (df) fprintf(df, "mode,bm,bn,M,N,K,splits,nphases,gflop,ms\n");
    for (size_t i = 0; i + 1 < g_prof.used; i += 2) {
        hipError_t rc = hipEventSynchronize(g_prof.events[i + 1]);
        if (rc != hipSuccess) { sempyr_set_error("profile_end: hipEventSynchronize: %s", hipGetErrorString(rc)); return (int)rc; }
        float t = 0.f;
        rc = hipEventElapsedTime(&t, g_prof.events[i], g_prof.events[i + 1]);
        if (rc != hipSuccess) { sempyr_set_error("profile_end: hipEventElapsedTime: %s", hipGetErrorString(rc)); return (int)rc; }
        ms += t;
        if (df && i / 2 < g_prof.recs.size()) {
            const Profiler::Rec& r = g_prof.recs[i / 2];
            fprintf(df, "%d,%d,%d,%d,%d,%d,%d,%d,%.4f,%.5f\n", r.mode, r.bm, r.bn, r.M, r.N, r.K, r.splits, r.nphases, r.flops * 1e-9, t);
        }
    }
    if (df) fclose(df);
    if (kernel_ms) *kernel_ms = ms;
    if (launches) *launches = (int64_t)(g_prof.used / 2);
    if (flops) *flops = g_prof.flops;
    return 0;
}
