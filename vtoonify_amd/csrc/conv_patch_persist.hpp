// Patch-resident 3x3 convolution, software-pipelined (conv_patch_pipe.hpp), with PERSISTENT workgroups: the K pipeline runs
// across tile boundaries (round 4).  Included by conv_igemm.hip inside its anonymous namespace after conv_patch_pipe.hpp.
//
// conv_patchp_kernel is one workgroup per tile.  Per round of 256 workgroups it pays (profiles/r04_epilogue.txt, the 256 x 128
// tiles, same box): ~3.6 us of prologue (address setup, the first patch and three taps of weights from a cold start, by all 256
// CUs at once) and, at its end, the acknowledgement of the tile's 64 KB of stores before the CU can take the next workgroup -- on
// launches of 2-4 rounds a fifth of the time.  Here a workgroup walks a contiguous range of tiles ([pixel tile][channel tile]
// order since round 5: the 2-4 channel tiles of one pixel tile run back to back on ONE workgroup, so their patch is fetched
// from HBM into one XCD's L2 once instead of once per channel tile on different XCDs) and the loader simply keeps going: in the last chunk of a tile the "next chunk" it prefetches (patch pieces spread over
// the taps, weights D taps ahead) is chunk 0 of the NEXT tile, so when the epilogue of a tile is done its successor's first patch
// and taps are in LDS, and the tile's stores drain under the successor's taps.  Same K order per tile, same lean epilogue: the
// same bits as conv_patchp_kernel.  Only the lean epilogue (conv_lean()) and whole K ranges (no split) are admitted.
// LDS: as conv_patchp_kernel; the patch pieces beyond the patch (dead 1 KB loads that keep every wave's operation count
// equal) all land on ONE slot behind the patch, which frees 6 KB at the end of the first patch buffer for the epilogue's fused-
// ToRGB exchange (the operand buffers are never idle here).
#pragma once

// LDS bytes the persistent kernel keeps for the epilogue tables of EVERY channel tile (bias / slope / gain: 8 bytes per channel),
// behind the patch: ONE constant for the kernel's static_assert and for the host's admission test in dispatch() (the
// two used to be 4096 and 6144 and agreed only for TH = 16, NW = 8)
constexpr int PQ_TAB_BYTES = 6144;

template <typename T, int TH, int BN, int WM, int WN, int NSTB = 4>
__global__ void __launch_bounds__(WM * WN * 64)
conv_patchq_kernel(const ConvArgs p, const GldsArgs g) {
    constexpr int TW = 16;
    constexpr int NW = WM * WN;
    constexpr int BM = TH * TW;
    constexpr int ESZ = (int)sizeof(T);
    constexpr int VEC = 16 / ESZ;
    constexpr int BK = 8 * VEC;
    constexpr int TM = BM / WM / 16, TN = BN / WN / 16;
    constexpr bool PERM = (TN % 2 == 0);
    constexpr int PH = TH + 2, PW = TW + 2, PROWS = PH * PW;
    constexpr int NPIECE = (PROWS + 7) / 8;                // live 1 KB pieces of a patch
    constexpr int PA = (NPIECE + NW - 1) / NW;             // pieces per wave per chunk (dead ones included)
    constexpr int LB = ((BN + 7) / 8 + NW - 1) / NW;
    constexpr int A_BYTES = PA * NW * 1024, B_BYTES = LB * NW * 1024;
    constexpr int D = NSTB - 1;
    constexpr int PTAPS = 9 - D;
    constexpr int PPT = (PA + PTAPS - 1) / PTAPS;
    constexpr int X_OFF = (NPIECE + 1) * 1024;             // epilogue scratch: behind the patch and the dead-piece slot
    static_assert(PERM, "lean epilogue: fragment pairs");
    static_assert(BM % (WM * 16) == 0 && BN % (WN * 16) == 0 && TM * WM == TH, "wave tiling");
    static_assert(NSTB >= 4 && NSTB <= 9, "ring depth");
    static_assert(2 * A_BYTES + NSTB * B_BYTES <= 160 * 1024, "LDS budget");
    static_assert((D - 2) * LB + PA < 64, "vmcnt is 6 bits");
    static_assert(WN == 1 || X_OFF + BM * WN * 12 <= A_BYTES, "room for the ToRGB exchange behind the patch");
    static_assert(A_BYTES - X_OFF >= PQ_TAB_BYTES, "room for the epilogue tables of every channel tile (the host admits tiles_n * BN * 8 <= PQ_TAB_BYTES)");

    __shared__ __attribute__((aligned(1024))) unsigned char smem[2 * A_BYTES + NSTB * B_BYTES];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = vt_uniform(tid >> 6) & (NW - 1);
    const int wm = wave / WN, wn = wave % WN;
    const int tiles_x = (p.Wo + TW - 1) / TW, tiles_y = (p.Ho + TH - 1) / TH;
    const int per_img = tiles_x * tiles_y;

    // units = tiles in [pixel tile][channel tile] order; workgroups in XCD-contiguous logical order (decode_block's) take
    // contiguous ranges
    int u0, u1;
    {
        const int nb = (int)gridDim.x, b = (int)blockIdx.x;
        const int qq = nb >> 3, rr = nb & 7, xcd = b & 7, idx = b >> 3;
        const int L = (xcd < rr ? xcd * (qq + 1) : rr * (qq + 1) + (xcd - rr) * qq) + idx;
        const int units = p.tiles_m * p.tiles_n;
        const int per = units / nb, rem = units - per * nb;
        u0 = L * per + (L < rem ? L : rem);
        u1 = u0 + per + (L < rem ? 1 : 0);
    }
    if (u0 >= u1) return;

    const int lrow = lane >> 3;
    const int jj = (lane & 7) ^ lrow;
    // Loader offsets are computed where they are used, not kept: 2 PA + LB registers per unit (the one being computed and its
    // successor) do not fit beside 64 accumulators and two fragment sets, and the dozen VALU operations per piece ride in the
    // shadow of the step's MFMAs.  Per unit only its scalars live: image, tile origin, channel tile.
    struct UnitPos {
        int im, ty0, tx0, tn;
    };
    auto unit_pos = [&](int u) {
        UnitPos r;
        const int tm = u / p.tiles_n;
        r.tn = u - tm * p.tiles_n;
        r.im = tm / per_img;
        const int tr = tm - r.im * per_img;
        r.ty0 = (tr / tiles_x) * TH, r.tx0 = (tr % tiles_x) * TW;
        return r;
    };
    UnitPos cur = unit_pos(u0), nxt = unit_pos(u0 + 1 < u1 ? u0 + 1 : u0);
    bool have_next = u0 + 1 < u1;

    const BufRsrc r0 = vt_make_rsrc(p.src0, g.nrec0);
    const BufRsrc r1 = vt_make_rsrc(p.src1 ? p.src1 : p.src0, p.src1 ? g.nrec1 : 0u);
    const BufRsrc rw = vt_make_rsrc(p.wgt, g.nrecw);
    const int nchunks = p.cin / BK;

    // offsets of patch piece `i` (pixel part, both sources) and of weight piece `i` of a unit
    auto patch_offset = [&](const UnitPos& up, int i, bool s1) -> uint32_t {
        const int pr = (i * NW + wave) * 8 + lrow;
        const int py = pr / PW, px = pr - py * PW;
        const int iy = up.ty0 - 1 + py, ix = up.tx0 - 1 + px;
        const bool in = pr < PROWS && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
        const uint32_t pix = (uint32_t)((up.im * p.H + iy) * p.W + ix);
        return in ? pix * (uint32_t)((s1 ? p.ld1 : p.ld0) * ESZ) + jj * 16 : GLDS_OOB;
    };
    auto weight_offset = [&](int tn, int i) -> uint32_t {
        const int row = (i * NW + wave) * 8 + lrow;
        const int n = tn * BN + tile_row_channel<PERM>(row);
        return (row < BN && n < p.coutT) ? (uint32_t)n * (uint32_t)(p.K * ESZ) + jj * 16 : GLDS_OOB;
    };
    // The unit being computed keeps its offsets in registers (the steady state of the K loop is conv_patchp_kernel's); they are
    // recomputed after every epilogue, so they are not live across it.  The successor's are computed where they are issued
    // (last chunk of a tile only): another 2 PA + LB registers do not fit beside 64 accumulators and two fragment sets.
    uint32_t pa0[PA], pa1[PA], woff[LB];
    auto load_offsets = [&]() {
#pragma unroll
        for (int i = 0; i < PA; ++i) pa0[i] = patch_offset(cur, i, false), pa1[i] = patch_offset(cur, i, true);
#pragma unroll
        for (int i = 0; i < LB; ++i) woff[i] = weight_offset(cur.tn, i);
    };
    load_offsets();
    // piece `i` of the patch of `chunk` (== nchunks: chunk 0 of the successor) -> patch buffer at byte offset `aoff`
    auto issue_a_piece = [&](int chunk, int aoff, int i) {
        const int piece = i * NW + wave;
        unsigned char* dst = smem + aoff + (piece < NPIECE ? piece : NPIECE) * 1024;
        if (chunk >= nchunks) {   // wave-uniform
            vt_glds16(r0, dst, have_next ? patch_offset(nxt, i, false) : GLDS_OOB, 0u);
        } else {
            const int kc = chunk * BK;
            const bool s1 = kc >= p.c0;
            vt_glds16(s1 ? r1 : r0, dst, s1 ? pa1[i] : pa0[i], (uint32_t)((s1 ? kc - p.c0 : kc) * ESZ));
        }
    };
    // weights of (chunk, tap) -> ring slot at byte offset `boff`
    auto issue_b = [&](int chunk, int tap, int boff) {
        const bool nx = chunk >= nchunks;
        const uint32_t so = (uint32_t)((tap * p.cin + (nx ? 0 : chunk * BK)) * ESZ);
#pragma unroll
        for (int i = 0; i < LB; ++i) {
            uint32_t vo = woff[i];
            if (nx) vo = have_next ? weight_offset(nxt.tn, i) : GLDS_OOB;
            vt_glds16(rw, smem + 2 * A_BYTES + boff + (i * NW + wave) * 1024, vo, so);
        }
    };

    f32x4 acc[TM][TN];
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int q = lane >> 4, l15 = lane & 15, l7 = lane & 7;
    uint32_t aswz[8][2];
#pragma unroll
    for (int cm = 0; cm < 8; ++cm)
#pragma unroll
        for (int sub = 0; sub < 2; ++sub)
            aswz[cm][sub] = (uint32_t)((wm * TM * PW + l15) * 128 + (((sub * 4 + q) ^ ((wm * TM * PW + l15 + cm) & 7)) << 4));
    uint32_t bfix[2];
#pragma unroll
    for (int sub = 0; sub < 2; ++sub)
        bfix[sub] = (uint32_t)(2 * A_BYTES + (wn * (TN * 16) + l15) * 128 + (((sub * 4 + q) ^ l7) << 4));

    u128 fa[2][TM], fb[2][TN];
    auto read_frags = [&](auto tapc, auto subc, u128 (&xa)[TM], u128 (&xb)[TN], int aoff, int boff) {
        constexpr int TAP = decltype(tapc)::value, SUB = decltype(subc)::value;
        constexpr int ky = TAP / 3, kx = TAP % 3;
        auto ra = [&](int a) {
            const int rowc = (a + ky) * PW + kx;
            xa[a] = ld128(smem + aoff + aswz[rowc & 7][SUB] + rowc * 128);
        };
        ra(0);
#pragma unroll
        for (int b = 0; b < TN; ++b) xb[b] = ld128(smem + boff + bfix[SUB] + b * 2048);
#pragma unroll
        for (int a = 1; a < TM; ++a) ra(a);
    };
    auto mma_all = [&](const u128 (&xa)[TM], const u128 (&xb)[TN]) {
#pragma unroll
        for (int a = 0; a < TM; ++a)
#pragma unroll
            for (int b = 0; b < TN; ++b) Mma<T>::run(acc[a][b], xb[b], xa[a]);
    };
    // epilogue tables of ALL channel tiles (bias, slope: tiles_n * BN floats each; the host admits at most 6 KB) in the free tail
    // of the SECOND patch buffer: read back per tile instead of 33 registers held through the K loop
    float* const ltab = reinterpret_cast<float*>(smem + A_BYTES + X_OFF);
    const int CT = p.tiles_n * BN;
    for (int i = tid; i < CT; i += NW * 64) {
        const bool ok = i < p.coutT;
        ltab[i] = (p.bias && ok) ? p.bias[i] : 0.0f;
        ltab[CT + i] = (p.slope_vec && ok) ? p.slope_vec[i] : p.slope;
    }   // (the prologue's barrier below publishes it)
    const float ga_all = p.gain_alpha * (p.alpha_dev ? p.alpha_dev[0] : 1.0f);

    // ---- prologue (once per workgroup): patch of the first chunk, weights of taps 0..D-1 -------------
#pragma unroll
    for (int i = 0; i < PA; ++i) issue_a_piece(0, 0, i);
    vt_static_for<D>([&](auto sc) {
        constexpr int s0 = decltype(sc)::value;
        issue_b(0, s0, s0 * B_BYTES);
    });
    vt_glds_wait_n<(D - 2) * LB>();
    vt_lds_barrier();
    read_frags(std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{}, fa[0], fb[0], 0, 0);

    int aoff = 0;
    int slot = 0;
    for (int u = u0; u < u1; ++u) {
        for (int chunk = 0; chunk < nchunks; ++chunk) {
            vt_static_for<9>([&](auto tc) {
                constexpr int t = decltype(tc)::value;
                constexpr int t3 = (t + D) % 9, c3 = (t + D) / 9;
                constexpr int t1 = (t + 1) % 9;
                constexpr int P0 = t * PPT < PA ? t * PPT : PA, P1 = (t + 1) * PPT < PA ? (t + 1) * PPT : PA;
                constexpr int TLO = t - (D - 2) > 0 ? t - (D - 2) : 0;
                constexpr int YOUNG = (P1 < PA ? P1 : PA) - (TLO * PPT < PA ? TLO * PPT : PA);
                static_assert(t >= PTAPS ? P0 == P1 : true, "patch pieces only in the first 9-D taps");
                const int boff = slot * B_BYTES;
                const int slot1 = slot + 1 == NSTB ? 0 : slot + 1, slotd = slot == 0 ? NSTB - 1 : slot - 1;
                const int boff1 = slot1 * B_BYTES, boff3 = slotd * B_BYTES;
                vt_sched_fence();
                read_frags(tc, std::integral_constant<int, 1>{}, fa[1], fb[1], aoff, boff);
                mma_all(fa[0], fb[0]);
                issue_b(chunk + c3, t3, boff3);
                vt_static_for<P1 - P0>([&](auto ic) { issue_a_piece(chunk + 1, aoff ^ A_BYTES, P0 + decltype(ic)::value); });
                read_frags(std::integral_constant<int, t1>{}, std::integral_constant<int, 0>{}, fa[0], fb[0],
                           t == 8 ? (aoff ^ A_BYTES) : aoff, boff1);
                mma_all(fa[1], fb[1]);
                vt_sched_fence();
                vt_glds_wait_n<(D - 2) * LB + YOUNG>();
#ifdef VT_EMU
                vt_lds_barrier();
#else
                __builtin_amdgcn_s_barrier();
#endif
                slot = slot1;
            });
            aoff ^= A_BYTES;
        }
        // ---- the tile is done: its successor's first patch and taps are landing; finish it from registers ----------------
        {
            EpiTables<TN> etab;
#pragma unroll
            for (int b = 0; b < TN; b += 2) {   // fragment pair (b, b+1) = channels 8q .. 8q+7 of a 32-channel group
                const int c0 = cur.tn * BN + wn * (TN * 16) + frag_channel<PERM>(b, q);
                unpack16<float>(ld128(ltab + c0), etab.bv[b]), unpack16<float>(ld128(ltab + c0 + 4), etab.bv[b + 1]);
                unpack16<float>(ld128(ltab + CT + c0), etab.sv[b]), unpack16<float>(ld128(ltab + CT + c0 + 4), etab.sv[b + 1]);
            }
            etab.ga = ga_all;
            conv_epilogue<T, BM, BN, WM, WN, 1>(p, acc, smem + X_OFF, PatchRows<TW>{cur.im, cur.ty0, cur.tx0, p.Ho, p.Wo},
                                                vt_opaque(cur.tn * BN), 0, u, etab);
        }
#pragma unroll
        for (int a = 0; a < TM; ++a)
#pragma unroll
            for (int b = 0; b < TN; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
        cur = nxt;
        have_next = u + 2 < u1;
        if (have_next) nxt = unit_pos(u + 2);
        load_offsets();
    }
}
