// Included by conv_igemm.hip inside its anonymous namespace (host side of conv_tpatch.h): which problems the 4x4 /
// stride-2 patch kernels take, their tile geometry, LDS pitches and launch.


struct TpTile {
    int nt;                // taps per workgroup: 16 = FWD, 4 = one stride phase of DGRAD
    int bn, ki, nsp, occ;  // pixels per tile, channels per k-half and group, staged positions per thread, workgroups per CU
    int spt;               // samples per tile
    long long tiles;       // workgroups without split-K: MT * n-tiles * phases
    int ngroups;           // channel groups of the reduction
    TpGeom g;
};

// Bank cost of one operand read (the 32 lanes of a k-half; the other half reads the same pattern one image further) for
// every 32-column block of a tile: sum over the instruction's lane groups of the largest number of distinct addresses on
// one bank (ds_read_b64: one group of 32 lanes, ds_read_b128: 16-lane groups {0-3,12-15,20-27} and {4-11,16-19,28-31};
// bank = dword address mod 64).  The best value is the number of groups.
inline int tp_bank_cost(int ki, int S, int R, int CW, int qb, int PWH, int PSS)
{
    static const int g128[2][16] = {{0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27},
                                    {4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31}};
    int cost = 0;
    for (int blk = 0; blk < qb; blk += 32) {
        int pos[32];
        for (int l = 0; l < 32; ++l) {
            const int ql = blk + l, smp = ql / (R * CW), rem = ql % (R * CW);
            pos[l] = smp * PSS + S * (rem / CW) * PWH + rem % CW;
        }
        const int ngroups = ki == 4 ? 2 : 1, glen = ki == 4 ? 16 : 32;
        for (int gi = 0; gi < ngroups; ++gi) {
            int cnt[64] = {0}, worst = 0;
            for (int e = 0; e < glen; ++e) {
                const int l = ki == 4 ? g128[gi][e] : e;
                bool dup = false;                            // identical addresses broadcast
                for (int f = 0; f < e && !dup; ++f) dup = pos[ki == 4 ? g128[gi][f] : f] == pos[l];
                if (dup) continue;
                for (int w = 0; w < ki; ++w) { const int bank = (pos[l] * ki + w) & 63; if (++cnt[bank] > worst) worst = cnt[bank]; }
            }
            cost += worst;
        }
    }
    return cost;
}

// LDS pitches: the smallest padding (within `maxpos` positions per k-half image) whose operand reads are conflict-free,
// else the least-conflicting one.  Cached per geometry (the search is a few thousand operations; conv launches ask per call).
inline bool tp_find_pitch(int ki, int S, int R, int CW, int spt, int qb, int PR, int PWHmin, int maxpos, int& PWH, int& PSS)
{
    static std::mutex mu;
    static std::map<unsigned long long, std::pair<int, int>> cache;
    const unsigned long long key = ((unsigned long long)ki << 60) ^ ((unsigned long long)S << 56) ^ ((unsigned long long)R << 46) ^
                                   ((unsigned long long)CW << 36) ^ ((unsigned long long)spt << 30) ^ ((unsigned long long)qb << 20) ^
                                   ((unsigned long long)PR << 12) ^ ((unsigned long long)maxpos);
    {
        std::lock_guard<std::mutex> lk(mu);
        auto it = cache.find(key);
        if (it != cache.end()) { PWH = it->second.first; PSS = it->second.second; return PWH > 0; }
    }
    const int planes = S == 2 ? 2 : 1;
    int best_cost = 1 << 30, best_size = 1 << 30, bw = 0, bs = 0;
    for (int w = PWHmin; w < PWHmin + 64; ++w) {
        const int base = planes * PR * w;
        for (int s = base; s < base + (spt > 1 ? 64 : 1); ++s) {
            if (spt * s > maxpos) break;
            const int c = tp_bank_cost(ki, S, R, CW, qb, w, s);
            if (c < best_cost || (c == best_cost && spt * s < best_size)) { best_cost = c; best_size = spt * s; bw = w; bs = s; }
        }
    }
    {
        std::lock_guard<std::mutex> lk(mu);
        cache[key] = std::make_pair(bw, bs);
    }
    PWH = bw; PSS = bs;
    return bw > 0;
}

// Tile geometry for `qb` pixels of a QH x QW grid per tile.
inline bool tp_geometry(int nt, int ki, int qb, int B, int QH, int QW, int nsp_max, TpTile& t)
{
    const int S = nt == 16 ? 2 : 1;
    const int QHW = QH * QW;
    int R, CW, spt = 1;
    if (QHW >= qb) {
        if (QHW % qb) return false;
        CW = (QW >= 32 && QW % 32 == 0 && qb % 32 == 0) ? 32 : QW;
        if (CW > qb || qb % CW) return false;
        R = qb / CW;
        if (QH % R) return false;
    } else {
        if (qb % QHW) return false;
        spt = qb / QHW;
        if (B % spt) return false;
        CW = QW; R = QH;
    }
    TpGeom& g = t.g;
    g.R = R; g.CW = CW;
    g.PR = S * R + 2; g.PW = S * CW + 2;
    g.npos = spt * g.PR * g.PW;
    const int PWHmin = S == 2 ? CW + 1 : CW + 2;
    if (g.npos > nsp_max * 256) return false;
    if (!tp_find_pitch(ki, S, R, CW, spt, qb, g.PR, PWHmin, nsp_max * 256, g.PWH, g.PSS)) return false;
    const int need = spt * g.PSS > g.npos ? spt * g.PSS : g.npos;
    t.nsp = (need + 255) / 256;
    t.bn = qb; t.spt = spt; t.nt = nt; t.ki = ki;
    return true;
}

inline size_t tp_lds_bytes(const TpTile& t) { return (size_t)2 * (t.nt * 2 * 64 + 2 * t.nsp * 256) * t.ki * sizeof(float); }

// Does a 4x4 / stride-2 patch kernel take this problem, and with which tile?  Depends on the descriptor only, so the plan
// (split count, workspace) and every launch agree.
inline bool tp_pick(const sempyr_conv_desc* d, int mode, TpTile& t)
{
    static const int enabled = env_int("SEMPYR_TPATCH", 3);          // bit 0: FWD, bit 1: DGRAD
    if (mode != SEMPYR_CONV_FWD && mode != SEMPYR_CONV_DGRAD) return false;
    if (!(enabled & (mode == SEMPYR_CONV_FWD ? 1 : 2))) return false;
    if (d->kh != 4 || d->kw != 4 || d->stride != 2 || d->pad != 1 || d->ups || d->math != SEMPYR_MATH_F32) return false;
    if ((d->Hin | d->Win) & 1) return false;
    const int M = mode == SEMPYR_CONV_FWD ? d->Cout : d->Cin, C = mode == SEMPYR_CONV_FWD ? d->Cin : d->Cout;
    if (M % 64 != 0 || C % 8 != 0) return false;
    // measured on the critic's blocks at batch 16 / 32 (tools/tp_prof.sh): the forward kernel wins everywhere (3 ... 18 %);
    // the per-phase input-gradient kernel wins 1 ... 6 % with two or more 64-row tiles and ties / loses 2 % with one
    static const int dgrad_min_m = env_int("SEMPYR_TP_DGRAD_MIN_M", 128);
    if (mode == SEMPYR_CONV_DGRAD && M < dgrad_min_m) return false;
    { static const int only_h = env_int("SEMPYR_TP_ONLY_H", 0), only_b = env_int("SEMPYR_TP_ONLY_B", 0);      // bisecting aid
      if ((only_h && d->Hin != only_h) || (only_b && d->B != only_b)) return false; }
    const int QH = d->Hin / 2, QW = d->Win / 2;
    const long long NQ = (long long)d->B * QH * QW;
    if (NQ >= (1ll << 31)) return false;
    const int MT = M / 64;
    const int nt = mode == SEMPYR_CONV_FWD ? 16 : 4, phases = mode == SEMPYR_CONV_FWD ? 1 : 4;
    const int ki = mode == SEMPYR_CONV_FWD ? 2 : 4;
    static const int force_bn_f = env_int("SEMPYR_TP_BN", 0), force_bn_d = env_int("SEMPYR_TP_BN_DGRAD", 0);
    const int force_bn = mode == SEMPYR_CONV_FWD ? force_bn_f : force_bn_d;
    // candidates from the largest tile down; take the one whose workgroups fill whole rounds of the chip best, preferring
    // larger tiles (fewer staged bytes per MFMA) when the difference is small
    static const int cand[3] = {256, 128, 64};
    static const int occ_of[3] = {2, 3, 3};                 // launch bounds of the instantiations (registers)
    static const double pref[3] = {1.0, 0.97, 0.90};
    double best_score = -1.0; bool have = false;
    for (int c = 0; c < 3; ++c) {
        if (force_bn && cand[c] != force_bn) continue;
        TpTile q{};
        if (!tp_geometry(nt, ki, cand[c], d->B, QH, QW, nt == 16 ? 5 : 2, q)) continue;
        q.tiles = (long long)MT * (NQ / q.bn) * phases;
        const int by_lds = (int)(160 * 1024 / tp_lds_bytes(q));
        q.occ = occ_of[c] < by_lds ? occ_of[c] : by_lds;
        if (q.occ < 1) continue;
        const long long slots = 256ll * q.occ;
        double score;
        if (q.tiles >= 384) score = pref[c] * (double)q.tiles / (double)((q.tiles + slots - 1) / slots * slots);
        else score = pref[c] * 0.5 * (double)q.tiles / 384.0;           // split-K territory: the more tiles the better
        if (score > best_score) { best_score = score; t = q; have = true; }
    }
    if (!have) return false;
    t.ngroups = C / (2 * t.ki);
    return true;
}

// split-K over channel groups: at least 128 reduction indices per split (as the generic kernel's 8 chunks of 16)
inline int tp_splits(const TpTile& t)
{
    const int k_per_group = 2 * t.ki * t.nt;
    const int min_groups = k_per_group >= 128 ? 1 : 128 / k_per_group;
    int splits = 1;
    const int target = target_blocks();
    if (t.tiles < target) {
        splits = (int)((target + t.tiles - 1) / t.tiles);
        const int maxs = t.ngroups / min_groups > 0 ? t.ngroups / min_groups : 1;
        if (splits > maxs) splits = maxs;
    }
    const int gps = (t.ngroups + splits - 1) / splits;
    splits = (t.ngroups + gps - 1) / gps;
    return splits < 1 ? 1 : splits;
}

template <int BN, int WM, int WN, int NT, int KI, int OCC>
int tp_launch_nsp(const IgemmParams& pp, const TpTile& t, dim3 grid, hipStream_t st)
{
    size_t lds = tp_lds_bytes(t);
    // An even share of workgroups per CU: the dispatcher fills a CU up to its occupancy limit before it moves on, so a launch
    // of 512 workgroups that fit 3 per CU leaves CUs with 3, 2 and 1 of them.  Asking for so much LDS that only
    // ceil(workgroups / 256) fit makes the placement even.
    lds += occ_cap_pad(lds);
    static const int even = env_int("SEMPYR_TP_EVEN", 0);      // measured: no effect on any critic shape (4 A/B runs), off
    const int want = (int)((grid.x + 255) / 256);
    if (even && want >= 1 && want < 4) {
        const size_t pad = (size_t)(160 * 1024) / (want + 1) + 256;
        if (pad > lds && pad <= (size_t)160 * 1024 / want) lds = pad;
    }
#define TP_CASE(NSP) case NSP: { \
        static const hipError_t attr = hipFuncSetAttribute((const void*)tpatch_kernel<BN, WM, WN, NT, KI, NSP, OCC>, \
                                                           hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); \
        (void)attr; \
        hipLaunchKernelGGL((tpatch_kernel<BN, WM, WN, NT, KI, NSP, OCC>), grid, dim3(256), lds, st, pp); return 0; }
    switch (t.nsp) {
        TP_CASE(1) TP_CASE(2)
        default: break;
    }
    if constexpr (NT == 16) {
        switch (t.nsp) {
            TP_CASE(3) TP_CASE(4) TP_CASE(5)
            default: break;
        }
    }
#undef TP_CASE
    sempyr_set_error("conv: internal: no 4x4 patch kernel for %d staged positions per thread", t.nsp);
    return SEMPYR_ERR_INVALID;
}

// p: the GATHER-mode parameters of the launch as conv2d_fwd_impl / conv2d_dgrad_impl fill them (A, X, ktab, M, Mpad, X
// geometry, output geometry, epilogue operands, x_bytes).  `tab_off`: first table row of each stride phase (DGRAD).
int tp_launch(const sempyr_conv_desc* d, const TpTile& t, const IgemmParams& p, const int* tab_off, int splits, hipStream_t st)
{
    IgemmParams pp = p;
    const bool fwd = t.nt == 16;
    const int QH = d->Hin / 2, QW = d->Win / 2;
    const int C = fwd ? d->Cin : d->Cout;
    pp.nsplits = splits; pp.nphases = fwd ? 1 : 4;
    pp.N = d->B * QH * QW; pp.QH = QH; pp.QW = QW; pp.QHW = QH * QW;
    pp.C = C;
    pp.tp = t.g;
    for (int s = 0; s < TP_SLOTS; ++s) {
        pp.tp.slot_row[s] = fwd ? s * C : tab_off[s >> 2] + (s & 3) * C;
        int dy, dx;
        if (fwd) { dy = s / 4 - 1; dx = s % 4 - 1; }       // tap (ky, kx) reads source (2 oy - 1 + ky, 2 ox - 1 + kx)
        else {                                              // phase (py, px), tap (j, i): gy pixel (qy + cy - j, qx + cx - i)
            int ry, cy, rx, cx;
            phase_taps(4, 2, 1, (s >> 2) >> 1, ry, cy); phase_taps(4, 2, 1, (s >> 2) & 1, rx, cx);
            dy = cy - ((s & 3) >> 1); dx = cx - (s & 1);
        }
        pp.tp.slot_dydx[s] = (int)(((uint32_t)(dy & 0xffff) << 16) | (uint32_t)(dx & 0xffff));
    }
    pp.tp.dRC.init(t.g.R * t.g.CW); pp.tp.dCW.init(t.g.CW); pp.tp.dPRPW.init(t.g.PR * t.g.PW); pp.tp.dPW.init(t.g.PW);
    const int MT = p.Mpad / 64, NT = pp.N / t.bn;
    const int NT8 = (NT + 7) / 8 * 8;
    dim3 grid((splits > 1 ? MT * NT * splits : MT * NT8) * pp.nphases, 1, 1);
    const double flops = 2.0 * (double)p.M * (double)pp.N * 16.0 * (double)C;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (g_prof.on) {
        std::lock_guard<std::mutex> lk(g_prof.mu);
        if (g_prof.used + 2 <= g_prof.events.size()) {
            e0 = g_prof.events[g_prof.used]; e1 = g_prof.events[g_prof.used + 1];
            g_prof.used += 2;
            g_prof.flops += flops;
            g_prof.recs.push_back({fwd ? 11 : 12, 64, t.bn, p.M, pp.N, fwd ? 16 * C : 4 * C, splits, pp.nphases, flops});
        }
    }
    if (e0) (void)hipEventRecord(e0, st);
    static const int dbg = env_int("SEMPYR_TP_DBG", 0);
    pp.dbg = dbg;
    static const char* stamp_path = getenv("SEMPYR_TP_STAMPS");     // tuning aid: per-workgroup timeline of every launch
    unsigned long long* stamps = nullptr;
    if (stamp_path && hipMalloc(&stamps, (size_t)grid.x * 64) == hipSuccess) { (void)hipMemset(stamps, 0, (size_t)grid.x * 64); pp.stamps = stamps; }
    int rc = SEMPYR_ERR_INVALID;
    if (fwd) {
        if (t.bn == 256) rc = tp_launch_nsp<256, 1, 4, 16, 2, 2>(pp, t, grid, st);
        else if (t.bn == 128) rc = tp_launch_nsp<128, 2, 2, 16, 2, 3>(pp, t, grid, st);
        else rc = tp_launch_nsp<64, 2, 2, 16, 2, 3>(pp, t, grid, st);
    } else {
        if (t.bn == 256) rc = tp_launch_nsp<256, 1, 4, 4, 4, 2>(pp, t, grid, st);
        else if (t.bn == 128) rc = tp_launch_nsp<128, 2, 2, 4, 4, 3>(pp, t, grid, st);
        else rc = tp_launch_nsp<64, 2, 2, 4, 4, 3>(pp, t, grid, st);
    }
    if (rc) return rc;
    if (e1) (void)hipEventRecord(e1, st);
    if (stamps) {
        (void)hipStreamSynchronize(st);
        std::vector<unsigned long long> h((size_t)grid.x * 8);
        (void)hipMemcpy(h.data(), stamps, h.size() * 8, hipMemcpyDeviceToHost);
        (void)hipFree(stamps);
        if (FILE* f = fopen(stamp_path, "a")) {
            unsigned long long t0 = ~0ull;
            for (size_t i = 0; i < h.size(); i += 8) if (h[i] && h[i] < t0) t0 = h[i];
            fprintf(f, "# launch nt=%d bn=%d M=%d N=%d C=%d splits=%d grid=%u\n", t.nt, t.bn, p.M, pp.N, C, splits, grid.x);
            for (size_t i = 0; i < h.size(); i += 8)
                if (h[i]) fprintf(f, "%zu %llu %llu %llu %llu %llu\n", i / 8, h[i] - t0, h[i + 1] - t0, h[i + 2] - t0, h[i + 3] - t0, h[i + 4] - t0);
            fclose(f);
        }
    }
    SEMPYR_CHECK_LAUNCH();
    return 0;
}
