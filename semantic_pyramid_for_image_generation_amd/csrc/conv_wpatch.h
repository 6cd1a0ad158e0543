// Included by conv_igemm.hip inside its anonymous namespace.
//
// Weight gradient of 3x3 / stride 1 / pad 1 convolutions (optionally behind the fused nearest upsample) with a LONG
// reduction and few output blocks -- the generator's 64^2 ... 256^2 layers (generator.py:41-43) -- as an LDS row-ring
// kernel; a 4x4 / stride 2 / pad 1 instance (the critic's first block, discriminator.py:105) exists as an opt-in:
//
//     gw[co][ci][kh][kw] = sum over (n, oy, ox) of gy[n][co][oy][ox] * x[n][ci][S*oy - 1 + kh][S*ox - 1 + kw]
//
// GEMM view: M = Cout rows, N = Cin * T columns (T taps), K = pixels.  The generic gather kernel forms every one of the T
// tap operands with its own global gather, so an input element crosses the memory system T times (measured 5.08x the
// algorithmic bytes on 64 -> 64 @ 256^2).  Here a workgroup owns a 64 (co) x 64 (ci) x T block of the result and walks a
// "slab" of the reduction: consecutive output rows of one column strip (CW pixels wide) of one sample.  Per output row it
// stages ONE row segment of gy (64 co x CW) and only the S NEW source rows -- the other KH - S rows of the tap window are
// still in the ring from the previous steps.  A source row segment (columns cb - 1 ... cb + S*CW) is fetched as aligned
// float4 from column cb - 4 on (18 instead of 16.5 float4 per channel; the unused head and tail land in slack words of
// the LDS row), so a thread issues 5 wide loads per row instead of 17 dword loads.  All T tap operands of a k-step are
// shifted views of the ring:
//
//     wave (wco, wci) holds T accumulator blocks of 32 co x 32 ci; per pair of adjacent pixels: one ds_read_b32 of gy (the A
//     operand, shared by the T taps) + T ds_read_b32 of the ring feed T v_mfma_f32_32x32x2_f32.
//
// Rows are padded to odd pitches, so the 32 lanes of an operand (32 channels, same pixel) hit 32 different banks.  The next
// step's float4 are loaded one per pixel pair behind the first pairs' MFMAs and stored into the free ring slots / the other
// gy buffer behind the last pairs': one barrier per output row, and neither the vector-memory issue nor the LDS stores hold
// the matrix pipe up for long (a row step runs at 0.95 of its MFMA-only time).  Every slab writes its block of partial sums
// in the generic kernel's slab layout [slab][co][tap * Cin + ci]; the library's split-K combine kernels (fixed order)
// finish the job.  Measurements: DESIGN.md 5.12.
struct WpParams {
    const float* gy; const float* x; float* out;
    long long slab_stride;
    int Cout, Cin, ldc;
    int H, W;                    // stored source rows / columns (before the fused upsample)
    int OH, OW;
    int XHW, YHW;
    int strips, row_chunks, rows_per_slab, nslabs;
    int MT, CT;                  // 64-row (co) and 64-column (ci) blocks
    unsigned x_bytes, gy_bytes;
    unsigned long long* stamps;   // tuning aid: per-workgroup s_memrealtime stamps (start, prologue done, after steps 1..4, main loop done, end)
};

template <int KH, int KW, int S, bool UPS, int CW, int OCC>
__global__ __launch_bounds__(256, OCC) void wpatch_kernel(const WpParams p)
{
    constexpr int T = KH * KW;
    constexpr int PWC = UPS ? CW / 2 + 2 : CW * S + KW - S;     // staged columns of a source row: source columns cb - 1 ... cb + PWC - 2
    constexpr int NV = (PWC - 2) / 4 + 2;                        // ... fetched as NV aligned float4 from column cb - 4 on
    constexpr int SLACK = 3;                                     // the unused head of the first / tail of the last float4 land here
    constexpr int PP = (PWC + 2 * SLACK) | 1;                    // odd pitches: channel c of a lane group -> bank c
    constexpr int PA = CW + 1;
    constexpr int LIVE = UPS ? 2 : KH;                           // source rows one output row reads
    constexpr int NEWR = UPS ? 1 : S;                            // new source rows per step (at most)
    constexpr int NSLOT = LIVE + NEWR;
    constexpr int NX = (64 * NV + 255) / 256;                   // float4 per thread and source row
    constexpr int NA = 64 * (CW / 4) / 256;                     // float4 per thread and gy row
    constexpr int DUMP = 2 * 64 * PA + NSLOT * 64 * PP;         // 4 floats per thread: where the stores of idle items go
    static_assert((PWC - 2) % 4 == 0 && CW % 16 == 0, "rows are fetched as aligned float4");
    constexpr int NPP = CW / 2;
    static_assert(!UPS || (KH == 3 && KW == 3 && S == 1), "the fused upsample is a 3x3 / stride-1 feature");
    extern __shared__ __attribute__((aligned(16))) float wsm[];
    float* const Aimg = wsm;                                     // [2][64][PA]
    float* const ring = wsm + 2 * 64 * PA;                       // [NSLOT][64][PP]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wco = wave >> 1, wci = wave & 1;
    const int l31 = lane & 31, kh2 = lane >> 5;                  // kh2: the lane's k index inside the MFMA (pixel of the pair)
    auto stamp = [&](int k) {
        if (p.stamps && tid == 0) p.stamps[(size_t)blockIdx.x * 8 + k] = __builtin_amdgcn_s_memrealtime();
    };
    stamp(0);

    // workgroup -> (slab, co block, ci block); the blocks of one slab are 8 ids apart (same XCD: they read the same rows)
    const int ntiles = p.MT * p.CT;
    int id = blockIdx.x, slab, tile;
    if ((p.nslabs & 7) == 0) { const int lo = id & 7; id >>= 3; tile = id % ntiles; slab = (id / ntiles) * 8 + lo; }
    else { tile = id % ntiles; slab = id / ntiles; }
    const int co0 = (tile / p.CT) * 64, ci0 = (tile % p.CT) * 64;
    // slab -> (sample, row chunk, column strip).  The strips of one (sample, row chunk) are 8 slabs apart whenever that is
    // possible: workgroup ids 8 apart share an XCD, and the 128-byte lines at a strip boundary are wanted by both neighbours.
    int chunk, strip, n;
    if (((p.nslabs / p.strips) & 7) == 0) {
        const int hi = slab >> 3, g = (hi / p.strips) * 8 + (slab & 7);
        strip = hi % p.strips; chunk = g % p.row_chunks; n = g / p.row_chunks;
    } else {
        chunk = slab % p.row_chunks; const int rest = slab / p.row_chunks;
        strip = rest % p.strips; n = rest / p.strips;
    }
    const int r0 = chunk * p.rows_per_slab, r1 = min(r0 + p.rows_per_slab, p.OH);
    const int c0 = strip * CW;

    // first / last source row of output row oy (arithmetic shifts: row -1 is the zero padding)
    auto row_lo = [&](int oy) { return UPS ? ((oy - 1) >> 1) : S * oy - 1; };
    auto row_hi = [&](int oy) { return UPS ? ((oy + 1) >> 1) : S * oy - 1 + KH - 1; };
    auto slot_of = [&](int y) { return (y + 1) % NSLOT; };

    constexpr unsigned OOB = 0x80000000u;
    const __amdgpu_buffer_rsrc_t x_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.x), 0, p.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t g_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.gy), 0, p.gy_bytes, 0x00020000);
    unsigned xvoff[NX]; int xlds[NX];
    const int cb = UPS ? c0 / 2 : c0 * S;                       // (a multiple of 4: the float4 are aligned, and whole inside or outside the row)
#pragma unroll
    for (int j = 0; j < NX; ++j) {
        const int e = tid + 256 * j;
        const int ci = e / NV, v = e - ci * NV;
        const bool live = e < 64 * NV;
        const int col = cb - 4 + 4 * v;
        xvoff[j] = (live && (unsigned)col < (unsigned)p.W) ? 4u * (unsigned)(ci * p.XHW + col) : OOB;
        xlds[j] = live ? ci * PP + 4 * v : DUMP + 4 * tid;      // source column cb - 1 sits at SLACK
    }
    unsigned avoff[NA]; int alds[NA];
#pragma unroll
    for (int j = 0; j < NA; ++j) {
        const int e = tid + 256 * j;
        const int co = e / (CW / 4), v = e - co * (CW / 4);
        avoff[j] = 4u * (unsigned)(co * p.YHW + 4 * v);
        alds[j] = co * PA + 4 * v;
    }
    const int xs_base = (n * p.Cin + ci0) * p.XHW;              // + y * W
    const int gs_base = (n * p.Cout + co0) * p.YHW + c0;        // + oy * OW

    f32x4 rx[NEWR][NX], ra[NA];
    auto load_x_row = [&](int i, int y) {                      // (uniform) y outside the image: zeros, no loads
        if ((unsigned)y < (unsigned)p.H) {
            const int so = 4 * (xs_base + y * p.W);
#pragma unroll
            for (int j = 0; j < NX; ++j) rx[i][j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(x_rs, (int)xvoff[j], so, 0));
        } else {
#pragma unroll
            for (int j = 0; j < NX; ++j) rx[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
    };
    auto store_x_row = [&](int i, int y) {
        float* dst = ring + slot_of(y) * (64 * PP);
#pragma unroll
        for (int j = 0; j < NX; ++j)
#pragma unroll
            for (int k = 0; k < 4; ++k) dst[xlds[j] + k] = rx[i][j][k];
    };
    auto load_a = [&](int oy) {
        const int so = 4 * (gs_base + oy * p.OW);
#pragma unroll
        for (int j = 0; j < NA; ++j) ra[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(g_rs, (int)avoff[j], so, 0));
    };
    auto store_a = [&](int buf) {
        float* dst = Aimg + buf * (64 * PA);
#pragma unroll
        for (int j = 0; j < NA; ++j)
#pragma unroll
            for (int k = 0; k < 4; ++k) dst[alds[j] + k] = ra[j][k];
    };

    f32x16 acc[T];
#pragma unroll
    for (int t = 0; t < T; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

    if (r0 < r1) {
        // prologue: every source row of the first output row, NEWR at a time
        for (int y = row_lo(r0); y <= row_hi(r0); y += NEWR) {
#pragma unroll
            for (int i = 0; i < NEWR; ++i) if (y + i <= row_hi(r0)) load_x_row(i, y + i);
#pragma unroll
            for (int i = 0; i < NEWR; ++i) if (y + i <= row_hi(r0)) store_x_row(i, y + i);
        }
        load_a(r0);
        store_a(0);
        __syncthreads();
        stamp(1);

        // the lane's column offsets inside a staged row: pixel pair pp adds S * 2 pp (UPS: pp)
        int lofs[KW];
#pragma unroll
        for (int kw = 0; kw < KW; ++kw) lofs[kw] = SLACK + (UPS ? 1 + ((kh2 + kw - 1) >> 1) : kh2 * S + kw);
        const int b_lane = (wci * 32 + l31) * PP;
        const int a_lane = (wco * 32 + l31) * PA + kh2;

        // The next step's data as LD pieces (one float4 load each): piece q is loaded behind the MFMAs of pixel pair q and
        // stored (four LDS words) behind those of one of the last NPP / 4 pairs, so that neither the vector-memory issue nor the LDS
        // stores stand between two MFMAs for long.  The last step of a slab fetches one step too many (a row of the next
        // slab, or zeros) into slots nobody reads: cheaper than a branch around every piece.
        constexpr int LD = NEWR * NX + NA;
        constexpr int SP = NPP / 4;                             // the stores are spread over the last SP pairs (late: the loads need their latency)
        static_assert(LD <= NPP, "one piece per pixel pair");
        for (int oy = r0; oy < r1; ++oy) {
            const int cur = (oy - r0) & 1;
            const int y_new = row_hi(oy) + 1;                   // first source row this step does not have yet
            int x_so[NEWR]; bool x_ok[NEWR]; float* x_dst[NEWR];
#pragma unroll
            for (int i = 0; i < NEWR; ++i) {
                const int y = y_new + i;
                x_ok[i] = (unsigned)y < (unsigned)p.H;
                x_so[i] = 4 * (xs_base + (x_ok[i] ? y : 0) * p.W);
                x_dst[i] = ring + slot_of(y) * (64 * PP);
            }
            const int a_so = 4 * (gs_base + (oy + 1) * p.OW);
            float* const a_dst = Aimg + (cur ^ 1) * (64 * PA);
            auto load_piece = [&](int q) {
                if (q < NEWR * NX) {
                    const int i = q / NX, j = q % NX;
                    rx[i][j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(x_rs, (int)(x_ok[i] ? xvoff[j] : OOB), x_so[i], 0));
                } else {
                    const int j = q - NEWR * NX;
                    ra[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(g_rs, (int)avoff[j], a_so, 0));
                }
            };
            auto store_piece = [&](int q) {
                if (q < NEWR * NX) {
                    const int i = q / NX, j = q % NX;
#pragma unroll
                    for (int k = 0; k < 4; ++k) x_dst[i][xlds[j] + k] = rx[i][j][k];
                } else {
                    const int j = q - NEWR * NX;
#pragma unroll
                    for (int k = 0; k < 4; ++k) a_dst[alds[j] + k] = ra[j][k];
                }
            };
            const float* ap = Aimg + cur * (64 * PA) + a_lane;
            const float* bp[KH];
#pragma unroll
            for (int kh = 0; kh < KH; ++kh) {
                const int y = UPS ? ((oy + kh - 1) >> 1) : S * oy - 1 + kh;
                bp[kh] = ring + slot_of(y) * (64 * PP) + b_lane;
            }
            float a[2], b[2][T];
            auto read_ops = [&](int w, int pp) {
                a[w] = ap[2 * pp];
#pragma unroll
                for (int kh = 0; kh < KH; ++kh)
#pragma unroll
                    for (int kw = 0; kw < KW; ++kw) b[w][kh * KW + kw] = bp[kh][lofs[kw] + (UPS ? pp : 2 * S * pp)];
            };
            read_ops(0, 0);
#pragma unroll
            for (int pp = 0; pp < NPP; ++pp) {
                const int w = pp & 1;
                if (pp + 1 < NPP) read_ops(w ^ 1, pp + 1);      // the next pair's operands are read before this pair's MFMAs issue
#pragma unroll
                for (int t = 0; t < T; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[w], b[w][t], acc[t], 0, 0, 0);
                if (pp + 1 < NPP) {
                    __builtin_amdgcn_sched_group_barrier(0x100, T + 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x008, T, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
                if (pp < LD) load_piece(pp);
                if (pp >= NPP - SP) {
#pragma unroll
                    for (int q = LD * (pp - (NPP - SP)) / SP; q < LD * (pp - (NPP - SP) + 1) / SP; ++q) store_piece(q);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            __syncthreads();
            if (oy - r0 < 4) stamp(2 + oy - r0);
        }
    }

    stamp(6);
    // accumulator register r of lane l: column (ci) l%32, row (co) (r&3) + 8*(r>>2) + 4*(l/32)
    float* out = p.out + (size_t)slab * p.slab_stride + (size_t)(co0 + wco * 32 + 4 * kh2) * p.ldc + ci0 + wci * 32 + l31;
#pragma unroll
    for (int t = 0; t < T; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) out[(size_t)((r & 3) + 8 * (r >> 2)) * p.ldc + t * p.Cin] = acc[t][r];
    __builtin_amdgcn_s_waitcnt(0);
    stamp(7);
}

// ---------------------------------------------------------------------------------------------------------------------
// host side
struct WpPick { int kind, cw, occ, strips, row_chunks, rows_per_slab, nslabs, MT, CT; size_t lds; };

// kind: 0 = 3x3 / s1, 1 = 3x3 / s1 behind the fused upsample, 2 = 4x4 / s2.  SEMPYR_WPATCH: bit per kind (A/B aid).
inline bool wp_pick(const sempyr_conv_desc* d, int Ho, int Wo, WpPick& w) {
    // default: the two 3x3 kinds.  The 4x4 / stride-2 kind is an opt-in (bit 2): on the critic's first block it is SLOWER than
    // the generic kernel (measured, batch 16 / 32: 116 / 198 us against 99 / 182 us including the combine pass) although its
    // row steps run at 0.90 of the MFMA-only rate -- 256 slabs of only 8 / 16 rows pay the prologue, the epilogue and a 67 MB
    // combine pass that the generic kernel's 96 slabs do not.
    static const int enabled = env_int("SEMPYR_WPATCH", 3);
    static const int target = env_int("SEMPYR_WPATCH_BLOCKS", 256);
    static const int max_tiles = env_int("SEMPYR_WPATCH_MAX_TILES", 8);
    static const int min_rows = env_int("SEMPYR_WPATCH_MIN_ROWS", 4);
    // (whatever d->math says: weight gradients are always fp32 products, so that both arithmetic modes give the same bits)
    if (d->Cin % 64 != 0 || d->Cout % 64 != 0 || d->pad != 1 || d->kh != d->kw || d->Win % 4 != 0) return false;
    int kind;
    if (d->kh == 3 && d->stride == 1) kind = d->ups ? 1 : 0;
    else if (d->kh == 4 && d->stride == 2 && !d->ups && d->Hin % 2 == 0 && d->Win % 2 == 0) kind = 2;
    else return false;
    if (!((enabled >> kind) & 1)) return false;
    w.kind = kind;
    if (kind == 2) { if (Wo % 32 != 0) return false; w.cw = 32; }
    else { if (Wo % 64 != 0) return false; w.cw = 64; }
    w.occ = 1;
    w.MT = d->Cout / 64; w.CT = d->Cin / 64;
    const int ntiles = w.MT * w.CT;
    if (ntiles > max_tiles) return false;                       // many tiles: short reductions, the combine pass would dominate
    w.strips = Wo / w.cw;
    const int per_chunk = d->B * w.strips * ntiles;             // workgroups per row chunk
    int chunks = target / per_chunk;
    if (chunks < 1) chunks = 1;
    int rows = (Ho + chunks - 1) / chunks;
    if (rows < min_rows) rows = min_rows < Ho ? min_rows : Ho;
    w.rows_per_slab = rows;
    w.row_chunks = (Ho + rows - 1) / rows;
    w.nslabs = d->B * w.strips * w.row_chunks;
    if (w.nslabs < 8) return false;
    const int pwc = kind == 1 ? w.cw / 2 + 2 : w.cw * d->stride + d->kw - d->stride;
    const int live = kind == 1 ? 2 : d->kh, newr = kind == 1 ? 1 : d->stride;
    w.lds = (size_t)4 * (2 * 64 * (w.cw + 1) + (live + newr) * 64 * ((pwc + 6) | 1) + 4 * 256);
    return true;
}

template <int KH, int KW, int S, bool UPS, int CW, int OCC>
int wp_launch_t(const WpParams& p, const WpPick& w, hipStream_t st) {
    static bool attr_done = false;
    if (!attr_done) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(&wpatch_kernel<KH, KW, S, UPS, CW, OCC>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
            SEMPYR_FAIL("conv wgrad: cannot raise the dynamic LDS limit");
        attr_done = true;
    }
    hipLaunchKernelGGL((wpatch_kernel<KH, KW, S, UPS, CW, OCC>), dim3((unsigned)(w.nslabs * w.MT * w.CT)), dim3(256), w.lds, st, p);
    return 0;
}

int wp_launch(const sempyr_conv_desc* d, const WpPick& w, int Ho, int Wo, const float* gy, const float* x, float* slabs, hipStream_t st) {
    const int T = d->kh * d->kw;
    WpParams p{};
    p.gy = gy; p.x = x; p.out = slabs;
    p.Cout = d->Cout; p.Cin = d->Cin; p.ldc = d->Cin * T;
    p.slab_stride = (long long)d->Cout * p.ldc;
    p.H = d->Hin; p.W = d->Win; p.OH = Ho; p.OW = Wo;
    p.XHW = d->Hin * d->Win; p.YHW = Ho * Wo;
    p.strips = w.strips; p.row_chunks = w.row_chunks; p.rows_per_slab = w.rows_per_slab; p.nslabs = w.nslabs;
    p.MT = w.MT; p.CT = w.CT;
    const long long xb = 4ll * d->B * d->Cin * p.XHW, gb = 4ll * d->B * d->Cout * p.YHW;
    if (xb >= (1ll << 31) || gb >= (1ll << 31)) SEMPYR_FAIL("conv wgrad: tensor beyond 2 GiB");
    if (((uintptr_t)gy | (uintptr_t)x) & 15) SEMPYR_FAIL("conv wgrad: the row-ring kernel needs 16-byte aligned tensors");
    p.x_bytes = (unsigned)xb; p.gy_bytes = (unsigned)gb;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (g_prof.on) {
        std::lock_guard<std::mutex> lk(g_prof.mu);
        if (g_prof.used + 2 <= g_prof.events.size()) {
            e0 = g_prof.events[g_prof.used]; e1 = g_prof.events[g_prof.used + 1];
            g_prof.used += 2;
            const double flops = 2.0 * (double)d->Cout * (double)p.ldc * (double)d->B * Ho * Wo;
            g_prof.flops += flops;
            g_prof.recs.push_back({13, 64, 64 * T, d->Cout, p.ldc, d->B * Ho * Wo, w.nslabs, 1, flops});
        }
    }
    if (e0) (void)hipEventRecord(e0, st);
    static const char* stamp_path = getenv("SEMPYR_WP_STAMPS");     // tuning aid: per-workgroup timeline of every launch
    const size_t nwg = (size_t)w.nslabs * w.MT * w.CT;
    unsigned long long* stamps = nullptr;
    if (stamp_path && hipMalloc(&stamps, nwg * 64) == hipSuccess) { (void)hipMemset(stamps, 0, nwg * 64); p.stamps = stamps; }
    int rc;
    if (w.kind == 0) rc = wp_launch_t<3, 3, 1, false, 64, 1>(p, w, st);
    else if (w.kind == 1) rc = wp_launch_t<3, 3, 1, true, 64, 1>(p, w, st);
    else rc = wp_launch_t<4, 4, 2, false, 32, 1>(p, w, st);
    if (rc) return rc;
    if (e1) (void)hipEventRecord(e1, st);
    if (stamps) {
        (void)hipStreamSynchronize(st);
        std::vector<unsigned long long> h(nwg * 8);
        (void)hipMemcpy(h.data(), stamps, h.size() * 8, hipMemcpyDeviceToHost);
        (void)hipFree(stamps);
        if (FILE* f = fopen(stamp_path, "a")) {
            unsigned long long t0 = ~0ull;
            for (size_t i = 0; i < h.size(); i += 8) if (h[i] && h[i] < t0) t0 = h[i];
            fprintf(f, "# launch kind=%d Cout=%d Cin=%d OH=%d OW=%d slabs=%d rows=%d\n", w.kind, d->Cout, d->Cin, Ho, Wo, w.nslabs, w.rows_per_slab);
            for (size_t i = 0; i < h.size(); i += 8) {
                fprintf(f, "%zu", i / 8);
                for (int k = 0; k < 8; ++k) fprintf(f, " %lld", h[i + k] ? (long long)(h[i + k] - t0) : -1ll);
                fprintf(f, "\n");
            }
            fclose(f);
        }
    }
    SEMPYR_CHECK_LAUNCH();
    return 0;
}
