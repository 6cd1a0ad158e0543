// Included by conv_igemm.hip inside its anonymous namespace.
//
// 4x4 / stride 2 / pad 1 convolutions (every block of the critic, discriminator.py:105-113, and the default generator's
// ConvTranspose2d blocks, generator.py:23) as LDS-patch kernels with wide operand reads:
//
//   FWD    y[co][oy][ox] = sum w[co][ci][ky][kx] * x[ci][2oy-1+ky][2ox-1+kx]     (NT = 16 taps, source stride S = 2)
//          a tile is SPT samples x R rows x CW columns of outputs; its (2R+2) x (2CW+2) source patch is staged once per
//          channel group, de-interleaved by column parity so that the 32 lanes of an operand read consecutive LDS words for
//          every tap.
//   DGRAD  each of the four stride phases of the input gradient is a 2x2 / stride-1 convolution over gy (NT = 4, S = 1):
//          the phase is part of the workgroup id, a tile is SPT x R x CW pixels of the phase grid and its patch (R+2) x (CW+2)
//          pixels of gy; the four phases of a tile have adjacent workgroup ids (same XCD: the patch is an L2 hit for three).
//
// Against the generic gather kernel: a source element is gathered ~1.2x (FWD) / ~1.3x (DGRAD) per tile pixel and
// channel instead of 16x / 4x, and the staged images are k-interleaved -- a lane's KI consecutive channels are one
// ds_read_b64 / ds_read_b128 -- so one LDS read per operand feeds KI k-steps of v_mfma_f32_32x32x2_f32 instead of one.
// The LDS images are double-buffered: the next channel group is stored between the MFMAs of the current one's last taps
// and there is ONE barrier per group.  Weights stay in the library's k-major packed layout: the A tile is transposed on
// the way into LDS.
template <int KI> struct TpVec;
template <> struct TpVec<2> { typedef float type __attribute__((ext_vector_type(2))); };
template <> struct TpVec<4> { typedef float type __attribute__((ext_vector_type(4))); };

constexpr int TP_SLOTS = 16;                                // (phase, tap) weight slots = the 16 taps of the 4x4 filter

template <int BN, int WM, int WN, int NT, int KI, int NSP, int OCC>
__global__ __launch_bounds__(256, OCC) void tpatch_kernel(const IgemmParams p)
{
    constexpr int BM = 64;
    constexpr int TM = BM / (WM * 32), TN = BN / (WN * 32);
    constexpr int PCG = 2 * KI;                             // channels per staged group (KI per k-half of the MFMA)
    constexpr int S = NT == 16 ? 2 : 1;                     // source stride
    constexpr int PITCH = NSP * 256;                        // LDS positions per k-half image
    constexpr int A_VECS = NT * 2 * BM;                     // [tap][k-half][m]
    constexpr int BUF_VECS = A_VECS + 2 * PITCH;            // one buffer: A image + [k-half][position] patch image
    constexpr int AI = (NT * 2 * (BM / 4) + 255) / 256;     // weight items per thread (item = tap, k-half, 4 rows m)
    constexpr bool A_ALL = (NT * 2 * (BM / 4)) % 256 == 0;
    static_assert(TN <= 2, "tap offsets are packed two per register");
    // Accumulator sets per 32x32 output block: consecutive MFMAs of a block go to different sets, which are added in the
    // epilogue.  A single set is one fp32 chain over the whole reduction (K = 1024 ... 8192 here): NACC sets cut the chain
    // (and its rounding error, which the gradient-parity tests see at the production batch) by NACC, and back-to-back MFMAs
    // no longer wait for each other's result.  Always 64 accumulator registers per lane.
    constexpr int NACC = 4 / (TM * TN);
    typedef typename TpVec<KI>::type vec;
    extern __shared__ __attribute__((aligned(16))) float tsm[];
    vec* const L = reinterpret_cast<vec*>(tsm);

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int l31 = lane & 31, kh = lane >> 5;
    auto stamp = [&](int k) {
        if (p.stamps && tid == 0) p.stamps[(size_t)blockIdx.x * 8 + k] = __builtin_amdgcn_s_memrealtime();
    };
    stamp(0);

    // workgroup -> (split, phase, m-tile, n-tile); the phases of a tile and the m-tiles re-reading one patch are 8 ids apart
    const int MT = p.Mpad / BM;
    int id = blockIdx.x, split = 0, phase, m_tile, n_tile;
    if (p.nsplits > 1) {
        split = id % p.nsplits; id /= p.nsplits;
        phase = id % p.nphases; id /= p.nphases;
        m_tile = id % MT; n_tile = id / MT;
    } else {
        const int xcd = id & 7; id >>= 3;
        phase = id % p.nphases; id /= p.nphases;
        m_tile = id % MT;
        n_tile = (id / MT) * 8 + xcd;
    }
    const int m0 = m_tile * BM, q0 = n_tile * BN;
    if (q0 >= p.N) return;                                  // uniform (N % BN == 0: no ragged tile)

    // tile geometry (uniform)
    const int R = p.tp.R, CW = p.tp.CW, PR = p.tp.PR, PW = p.tp.PW, PWH = p.tp.PWH, PSS = p.tp.PSS;
    const int RC = R * CW;
    const int tb0 = q0 / p.QHW;                             // first sample of the tile
    const int tt = (q0 - tb0 * p.QHW) / RC;                 // tile index inside the sample (0 when the tile is whole samples)
    const int tpr = p.QW / CW;
    const int trow = tt / tpr, tcol = tt - trow * tpr;
    const int gy0 = trow * R, gx0 = tcol * CW;
    const int sy0 = S * gy0 - 1, sx0 = S * gx0 - 1;         // source coordinates of the patch origin
    const int C = p.C;

    // LDS position of source (row pr, column pc) of sample smp of the tile
    auto lds_pos = [&](int smp, int pr, int pc) {
        return smp * PSS + (S == 2 ? ((pc & 1) * PR + pr) * PWH + (pc >> 1) : pr * PWH + pc);
    };

    // patch loads: a thread owns NSP source positions; every channel of the group is one SGPR offset further
    constexpr unsigned OOB = 0x80000000u;
    const __amdgpu_buffer_rsrc_t x_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.X), 0, p.x_bytes, 0x00020000);
    unsigned pvoff[NSP];
    int plds[NSP];
    const int XH = p.GH;
#pragma unroll
    for (int i = 0; i < NSP; ++i) {
        const int pos = tid + 256 * i;
        const int smp = (int)p.tp.dPRPW.div((uint32_t)pos), rem = pos - smp * (PR * PW);
        const int pr = (int)p.tp.dPW.div((uint32_t)rem), pc = rem - pr * PW;
        const int sy = sy0 + pr, sx = sx0 + pc;
        const bool live = pos < p.tp.npos;
        const bool ok = live && (unsigned)sy < (unsigned)XH && (unsigned)sx < (unsigned)p.XW;     // else: zero-filled
        pvoff[i] = ok ? 4u * (unsigned)((tb0 + smp) * p.XCHW + sy * p.XW + sx) : OOB;
        plds[i] = live ? A_VECS + lds_pos(smp, pr, pc) : -1;
    }
    // weight items: item = tid + 256 i -> (tap, k-half, 4 consecutive rows m); its KI channels are KI rows of the k-major
    // packed weights
    int a_off[AI];
#pragma unroll
    for (int i = 0; i < AI; ++i) {
        const int item = tid + 256 * i;
        const int t = (item >> 5) % NT, khi = (item >> 4) & 1, m4 = item & 15;
        int row = 0;
#pragma unroll
        for (int s = 0; s < NT; ++s) row = t == s ? p.tp.slot_row[phase * NT + s] : row;
        a_off[i] = (row + KI * khi) * p.Mpad + m0 + 4 * m4;
    }

    const int ngroups = C / PCG;
    const int gps = (ngroups + p.nsplits - 1) / p.nsplits;
    const int g_begin = split * gps, g_end = min(g_begin + gps, ngroups);

    float rp[NSP][PCG];
    f32x4 ra[AI][KI];
    auto load_group = [&](int g) {
        const float* ab = p.A + (size_t)g * PCG * p.Mpad;
#pragma unroll
        for (int i = 0; i < AI; ++i)
            if (A_ALL || tid + 256 * i < NT * 2 * (BM / 4)) {
#pragma unroll
                for (int r = 0; r < KI; ++r) ra[i][r] = *reinterpret_cast<const f32x4*>(ab + a_off[i] + r * p.Mpad);
            }
        const int cb = g * PCG * p.XHW * 4;
#pragma unroll
        for (int i = 0; i < NSP; ++i)
#pragma unroll
            for (int ci = 0; ci < PCG; ++ci)
                rp[i][ci] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(x_rs, (int)pvoff[i], cb + ci * p.XHW * 4, 0));
    };
    // the staged data as NPC pieces (weight item / one patch position), so that the stores can be spread between MFMAs
    constexpr int NPC = AI + NSP;
    auto store_piece = [&](vec* buf, int pc) {
        if (pc < AI) {
            const int item = tid + 256 * pc;
            if (A_ALL || item < NT * 2 * (BM / 4)) {
                vec* dst = buf + (item >> 4) * BM + 4 * (item & 15);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    vec v;
#pragma unroll
                    for (int r = 0; r < KI; ++r) v[r] = ra[pc][r][e];
                    dst[e] = v;
                }
            }
        } else {
            const int i = pc - AI;
            if (plds[i] >= 0) {
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    vec v;
#pragma unroll
                    for (int r = 0; r < KI; ++r) v[r] = rp[i][h * KI + r];
                    buf[h * PITCH + plds[i]] = v;
                }
            }
        }
    };

    f32x16 accs[NACC][TM][TN];
#pragma unroll
    for (int c = 0; c < NACC; ++c)
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) accs[c][i][j][r] = 0.f;

    // per-lane LDS positions (+ the lane's k-half image) of the NT taps for the lane's TN operand columns, two per register
    // (computed while the first group's loads are in flight)
    int lsmp[TN], lr[TN], lc[TN];
    unsigned off[NT];
    auto lane_offsets = [&]() {
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int ql = wn * (TN * 32) + j * 32 + l31;
            lsmp[j] = (int)p.tp.dRC.div((uint32_t)ql);
            const int rem = ql - lsmp[j] * RC;
            lr[j] = (int)p.tp.dCW.div((uint32_t)rem); lc[j] = rem - lr[j] * CW;
        }
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const int pk = p.tp.slot_dydx[phase * NT + t];  // uniform
            const int dy = pk >> 16, dx = (int)(short)(pk & 0xffff);
            unsigned packed = 0u;
#pragma unroll
            for (int j = 0; j < TN; ++j)
                packed |= (unsigned)(A_VECS + lds_pos(lsmp[j], S * lr[j] + dy + 1, S * lc[j] + dx + 1) + kh * PITCH) << (16 * j);
            off[t] = packed;
        }
    };

    if (g_begin < g_end) {
        stamp(1);
        load_group(g_begin);
        lane_offsets();
#pragma unroll
        for (int pc = 0; pc < NPC; ++pc) store_piece(L, pc);
        __syncthreads();
        stamp(2);
        const int a_lane = kh * BM + wm * (TM * 32) + l31;  // + t * 2 * BM + i * 32
        constexpr int T0 = NT - NT / 4;                     // the stores of the next group start behind tap T0 (late: the loads
                                                            // of all workgroups arrive as one burst)
        for (int g = g_begin; g < g_end; ++g) {
            const bool more = (g + 1) < g_end;
            vec* const cur_buf = L + ((g - g_begin) & 1) * BUF_VECS;
            vec* const nxt_buf = L + (((g - g_begin) & 1) ^ 1) * BUF_VECS;
            if (more && !(p.dbg & 1)) load_group(g + 1);    // in flight under this group's NT x KI x TM x TN MFMAs
            __builtin_amdgcn_sched_barrier(0);
            const vec* as = cur_buf + a_lane;
            vec a[2][TM], b[2][TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) a[0][i] = as[i * 32];
#pragma unroll
            for (int j = 0; j < TN; ++j) b[0][j] = cur_buf[(off[0] >> (16 * j)) & 0xffffu];
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const int cur = t & 1, nxt = cur ^ 1;
                if (t + 1 < NT) {                           // operands of the next tap are read before this tap's MFMAs issue
#pragma unroll
                    for (int i = 0; i < TM; ++i) a[nxt][i] = as[(t + 1) * 2 * BM + i * 32];
#pragma unroll
                    for (int j = 0; j < TN; ++j) b[nxt][j] = cur_buf[(off[t + 1] >> (16 * j)) & 0xffffu];
                }
#pragma unroll
                for (int s = 0; s < KI; ++s)
#pragma unroll
                    for (int i = 0; i < TM; ++i)
#pragma unroll
                        for (int j = 0; j < TN; ++j)
                            accs[(t * KI + s) % NACC][i][j] =
                                __builtin_amdgcn_mfma_f32_32x32x2f32(a[cur][i][s], b[cur][j][s], accs[(t * KI + s) % NACC][i][j], 0, 0, 0);
                if (t + 1 < NT) {
                    __builtin_amdgcn_sched_group_barrier(0x100, TM + TN, 0);
                    __builtin_amdgcn_sched_group_barrier(0x008, KI * TM * TN, 0);
                }
                if (t >= T0) {                              // this tap's share of the next group's stores (other buffer)
                    __builtin_amdgcn_sched_barrier(0);
                    if (more && !(p.dbg & 2)) {
#pragma unroll
                        for (int pc = NPC * (t - T0) / (NT - T0); pc < NPC * (t - T0 + 1) / (NT - T0); ++pc) store_piece(nxt_buf, pc);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            __syncthreads();                                // this group's reads and the next group's stores are done
        }
    }

    else lane_offsets();
    stamp(3);
    // ---------------- epilogue: eval-BatchNorm scale, bias, residual, activation, store ----------------
    // accumulator register r of lane l: column n = l%32, row m = (r&3) + 8*(r>>2) + 4*(l/32)
    f32x16 (&acc)[TM][TN] = accs[0];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            if constexpr (NACC == 4) acc[i][j] = (accs[0][i][j] + accs[1][i][j]) + (accs[2][i][j] + accs[3][i][j]);
            else if constexpr (NACC == 2) acc[i][j] = accs[0][i][j] + accs[1][i][j];
        }
    float* out = p.Out + (size_t)split * p.slab_stride;
    const bool full_m = (m0 + BM) <= p.M;
    const int py = phase >> 1, px = phase & 1;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int mb = m0 + wm * (TM * 32) + i * 32 + 4 * kh;
        if (p.scale) {                                      // uniform
            float sv[16];
#pragma unroll
            for (int r16 = 0; r16 < 16; ++r16) { const int m = mb + (r16 & 3) + 8 * (r16 >> 2); sv[r16] = p.scale[m < p.M ? m : 0]; }
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int r16 = 0; r16 < 16; ++r16) acc[i][j][r16] *= sv[r16];
        }
        if (p.bias) {                                       // uniform
            float bv[16];
#pragma unroll
            for (int r16 = 0; r16 < 16; ++r16) { const int m = mb + (r16 & 3) + 8 * (r16 >> 2); bv[r16] = p.bias[m < p.M ? m : 0]; }
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int r16 = 0; r16 < 16; ++r16) acc[i][j][r16] += bv[r16];
        }
    }
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        // output pixel of the lane's column: the tile pixel itself (FWD), or pixel (2q + phase) of the gradient (DGRAD)
        const size_t o = (size_t)(tb0 + lsmp[j]) * p.OCHW + ((gy0 + lr[j]) * p.OSY + py) * p.OW + (gx0 + lc[j]) * p.OSX + px;
        float* op = out + o;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int mb = m0 + wm * (TM * 32) + i * 32 + 4 * kh;
            if (p.res) {                                    // uniform
                const float* rp2 = p.res + o;
#pragma unroll
                for (int r16 = 0; r16 < 16; ++r16) {
                    const int m = mb + (r16 & 3) + 8 * (r16 >> 2);
                    acc[i][j][r16] += rp2[(size_t)(m < p.M ? m : 0) * p.OHW];
                }
            }
            if (p.act != SEMPYR_ACT_NONE) {                 // uniform
#pragma unroll
                for (int r16 = 0; r16 < 16; ++r16) acc[i][j][r16] = act_apply(acc[i][j][r16], p.act, p.slope);
            }
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
    __builtin_amdgcn_s_waitcnt(0);
    stamp(4);
}
