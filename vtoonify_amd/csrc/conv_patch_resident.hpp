// Patch-resident 3x3 convolution with the WEIGHTS resident too: single-chunk layers (Cin = one 128-byte chunk), round 4.
// Included by conv_igemm.hip inside its anonymous namespace, after conv_patch_pipe.hpp (vt_static_for).
//
// The 64 -> 64 conv of the 512^2 generator level (model/stylegan/model.py:417-430, `convs[...]` of the last-but-one
// resolution) has K = 9 x 64: one chunk.  On conv_patchp_kernel every one of its 4096 tiles (4 frames) is a workgroup
// that fetches its 41 KB patch AND the 72 KB of weights, runs 9 tap steps with a barrier each, and leaves: 140-205 us for
// 77 GFLOP / 268 MB -- the launch is ramps and drains, not steps; with 128 KB of LDS per workgroup no second one shares
// the CU to fill them.  Here a workgroup is PERSISTENT and its 8 waves are TWO GROUPS of 4 that alternate roles:
//   * the 9 taps of all BN output channels (72 KB at BN = 64) are fetched once per workgroup and stay in LDS;
//   * the workgroup walks tiles, even ones on group 0, odd ones on group 1.  A group owns ONE patch buffer and runs
//     "M" slots (the 9 taps of its tile: 64 x 64 per wave, no barrier inside) and "E" slots (LDS-DMA of its next patch
//     into the buffer it just finished with, then the epilogue of the tile from registers, then wait for the patch);
//     the groups are one slot apart, so on every SIMD one wave feeds the matrix pipe while the other does the vector /
//     memory work of an epilogue.  One workgroup barrier per slot couples them;
//   * same fragment layout, same K order ([tap][half]) as conv_patch_kernel / conv_patchp_kernel: bit-identical.
// A first form with all 8 waves on one tile and a double-buffered patch (one barrier per tile) measured 110 us against
// 140 on the pipelined tiles: with every wave in its epilogue at the same time the matrix pipe idled for it.
// LDS: 72 KB weights + 2 x 41 KB patches = 154 KB.  Tiles are handed out per XCD in contiguous ranges (neighbouring
// tiles share halo rows in that XCD's L2).
#pragma once

template <typename T, int TH, int BN>
__global__ void __launch_bounds__(512)
conv_patchw_kernel(const ConvArgs p, const GldsArgs g) {
    constexpr int TW = 16;
    constexpr int GW = 4;                                    // waves per group: one 16-pixel x TM-row block each
    constexpr int BM = TH * TW;
    constexpr int ESZ = (int)sizeof(T);
    constexpr int TM = BM / GW / 16, TN = BN / 16;
    constexpr bool PERM = (TN % 2 == 0);
    constexpr int PH = TH + 2, PW = TW + 2, PROWS = PH * PW;
    constexpr int NPIECE = (PROWS + 7) / 8;                  // 1 KB pieces of a patch
    constexpr int PA = (NPIECE + GW - 1) / GW;               // per wave of the group
    constexpr int WPIECE = (BN + 7) / 8;                     // 1 KB pieces of one tap's weights
    constexpr int LB = (WPIECE + 7) / 8;                     // per wave (all 8 fetch weights)
    constexpr int A_BYTES = NPIECE * 1024, W_TAP = WPIECE * 1024, W_BYTES = 9 * W_TAP;
    constexpr int A_OFF = W_BYTES, TAB_OFF = W_BYTES + 2 * A_BYTES;   // TAB: fp32 [bias | ToRGB w0 | w1 | w2][BN]
    static_assert(sizeof(T) == 2, "bf16 (the lean epilogue packs 8 channels into a 16-byte store)");
    static_assert(BM % (GW * 16) == 0 && BN % 16 == 0, "wave tiling");
    static_assert(TM * GW == TH, "one 16-pixel tile row per MFMA row block");
    static_assert(TAB_OFF + 4 * BN * 4 <= 160 * 1024, "LDS budget");
    static_assert(TM == 4 && TN == 4, "the epilogue below: 4 tile rows per wave (one per lane group), 2 x 32 channels");
    static_assert(9 * LB + PA < 64, "vmcnt is 6 bits");

    __shared__ __attribute__((aligned(1024))) unsigned char smem[TAB_OFF + 4 * BN * 4];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = vt_uniform(tid >> 6) & 7;
    const int grp = wave >> 2, wm = wave & 3;
    const int tiles_x = (p.Wo + TW - 1) / TW, tiles_y = (p.Ho + TH - 1) / TH;
    const int per_img = tiles_x * tiles_y;
    const int ntiles = p.tiles_m;

    // tiles of this workgroup: XCD x = blockIdx % 8 owns the contiguous range [x T8, (x+1) T8); its workgroups stride over it
    int tile0, tile_end, tile_step;
    {
        const int G = (int)gridDim.x, b = (int)blockIdx.x;
        if (G % 8 == 0) {
            const int t8 = (ntiles + 7) / 8, x = b & 7;
            tile0 = x * t8 + (b >> 3);
            tile_end = (x + 1) * t8 < ntiles ? (x + 1) * t8 : ntiles;
            tile_step = G >> 3;
        } else {
            tile0 = b, tile_end = ntiles, tile_step = G;
        }
    }
    if (tile0 >= tile_end) return;
    const int nmine = (tile_end - tile0 + tile_step - 1) / tile_step;   // tiles of the workgroup: k = 0 .. nmine-1
    const int ng = (nmine - grp + 1) / 2;                               // ... of this group: k = grp, grp + 2, ...
    auto tile_of = [&](int j) { return tile0 + (2 * j + grp) * tile_step; };

    const int lrow = lane >> 3;
    const int jj = (lane & 7) ^ lrow;
    const BufRsrc r0 = vt_make_rsrc(p.src0, g.nrec0);
    const BufRsrc rw = vt_make_rsrc(p.wgt, g.nrecw);

    // ---- weights: all 9 taps, once.  LDS row t*BN + r = tap t, tile row r (fragment order, tile_row_channel) ----------
#pragma unroll
    for (int i = 0; i < LB; ++i) {
        const int piece = i * 8 + wave;
        const int row = piece * 8 + lrow;
        const int n = tile_row_channel<PERM>(row);
        const uint32_t wo = (row < BN && n < p.coutT) ? (uint32_t)n * (uint32_t)(p.K * ESZ) + jj * 16 : GLDS_OOB;
        if (piece < WPIECE) {
#pragma unroll
            for (int t = 0; t < 9; ++t) vt_glds16(rw, smem + t * W_TAP + piece * 1024, wo, (uint32_t)(t * p.cin * ESZ));
        }
    }
    // patch of tile `tm` -> this group's patch buffer
    const int aoff = A_OFF + grp * A_BYTES;
    auto issue_patch = [&](int tm) {
        const int lrow = vt_opaque(lane >> 3);   // (the 2 x PA patch coordinates of a lane are loop-invariant: recomputed per tile, not kept)
        const int jj = (lane & 7) ^ lrow;
        const int im = tm / per_img, tr = tm - im * per_img;
        const int ty0 = (tr / tiles_x) * TH, tx0 = (tr % tiles_x) * TW;
#pragma unroll
        for (int i = 0; i < PA; ++i) {
            const int piece = i * GW + wm;
            const int pr = piece * 8 + lrow;
            const int py = pr / PW, px = pr - py * PW;
            const int iy = ty0 - 1 + py, ix = tx0 - 1 + px;
            const bool in = pr < PROWS && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
            const uint32_t pix = (uint32_t)((im * p.H + iy) * p.W + ix);
            const uint32_t vo = in ? pix * (uint32_t)(p.ld0 * ESZ) + jj * 16 : GLDS_OOB;
            if (piece < NPIECE) vt_glds16(r0, smem + aoff + piece * 1024, vo, 0u);
        }
    };
    if (ng > 0) issue_patch(tile_of(0));
    // epilogue tables -> LDS (read back per tile: 16 ds_read_b128 per lane instead of 64 registers held through the M slots)
    const bool rgbf = p.rgb_w != nullptr;
    if (tid < BN) {
        float* tab = reinterpret_cast<float*>(smem + TAB_OFF);
        tab[tid] = p.bias ? p.bias[tid] : 0.0f;
#pragma unroll
        for (int j = 0; j < 3; ++j) tab[(1 + j) * BN + tid] = rgbf ? to_f32(((const T*)p.rgb_w)[j * p.coutT + tid]) : 0.0f;
    }
    float rb[3] = {0.0f, 0.0f, 0.0f};
    if (rgbf && p.rgb_bias) rb[0] = p.rgb_bias[0], rb[1] = p.rgb_bias[1], rb[2] = p.rgb_bias[2];
    const int HoWo = p.Ho * p.Wo;
    // (the host checked that the three tensors are below 2 GB; a null skip reads as zeros)
    const BufRaw rout = vt_make_raw(p.out, (uint32_t)((int64_t)p.M * p.ld_out * ESZ));
    const BufRaw rskip = vt_make_raw(p.rgb_resid, (uint32_t)((int64_t)p.N * 3 * HoWo * 4));
    const BufRaw rimg = vt_make_raw(p.rgb_out, (uint32_t)((int64_t)p.N * 3 * HoWo * 4));

    const int q = lane >> 4, l15 = lane & 15, l7 = lane & 7;
    uint32_t aswz[8][2];
#pragma unroll
    for (int cm = 0; cm < 8; ++cm)
#pragma unroll
        for (int sub = 0; sub < 2; ++sub)
            aswz[cm][sub] = (uint32_t)(aoff + (wm * TM * PW + l15) * 128 + (((sub * 4 + q) ^ ((wm * TM * PW + l15 + cm) & 7)) << 4));
    uint32_t bfix[2];
#pragma unroll
    for (int sub = 0; sub < 2; ++sub) bfix[sub] = (uint32_t)(l15 * 128 + (((sub * 4 + q) ^ l7) << 4));

    u128 fa[2][TM], fb[2][TN];
    auto read_frags = [&](auto tapc, auto subc, u128 (&xa)[TM], u128 (&xb)[TN]) {
        constexpr int TAP = decltype(tapc)::value, SUB = decltype(subc)::value;
        constexpr int ky = TAP / 3, kx = TAP % 3;
        auto ra = [&](int a) {
            const int rowc = (a + ky) * PW + kx;
            xa[a] = ld128(smem + aswz[rowc & 7][SUB] + rowc * 128);
        };
        ra(0);
#pragma unroll
        for (int b = 0; b < TN; ++b) xb[b] = ld128(smem + TAP * W_TAP + bfix[SUB] + b * 2048);
#pragma unroll
        for (int a = 1; a < TM; ++a) ra(a);
    };

    // weights + first patches landed
    vt_glds_wait_n<0>();
    vt_lds_barrier();
    // Every group runs [M, barrier, E, barrier] per tile; group 1 starts one barrier late, so its M slots coincide with group
    // 0's E slots.  (M and E of a tile are straight-line code in one loop iteration: as two arms of a slot loop the
    // accumulators were loop-carried through both arms and the register allocator kept a second copy of all 64.)
    if (grp) vt_lds_barrier();
    for (int j = 0; j < ng; ++j) {
        // ---- M: 9 taps x 2 halves from the group's patch and the resident weights -------------------------------
        f32x4 acc[TM][TN];
#pragma unroll
        for (int a = 0; a < TM; ++a)
#pragma unroll
            for (int b = 0; b < TN; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
        read_frags(std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{}, fa[0], fb[0]);
        vt_static_for<9>([&](auto tc) {
            constexpr int t = decltype(tc)::value;
            // one wave per SIMD is in its M slot: nothing but its own next fragments hides an LDS round trip, so the
            // interleave is pinned -- the TM + TN reads of the next half-step spread over the TM x TN MFMAs of this one
            vt_sched_fence();
            read_frags(tc, std::integral_constant<int, 1>{}, fa[1], fb[1]);
#pragma unroll
            for (int a = 0; a < TM; ++a)
#pragma unroll
                for (int b = 0; b < TN; ++b) Mma<T>::run(acc[a][b], fb[0][b], fa[0][a]);
            vt_static_for<TM + TN>([&](auto) { vt_sched_group<0x100, 1>(); vt_sched_group<0x008, (TM * TN) / (TM + TN)>(); });
            vt_sched_fence();
            if constexpr (t < 8) read_frags(std::integral_constant<int, t + 1>{}, std::integral_constant<int, 0>{}, fa[0], fb[0]);
#pragma unroll
            for (int a = 0; a < TM; ++a)
#pragma unroll
                for (int b = 0; b < TN; ++b) Mma<T>::run(acc[a][b], fb[1][b], fa[1][a]);
            if constexpr (t < 8)
                vt_static_for<TM + TN>([&](auto) { vt_sched_group<0x100, 1>(); vt_sched_group<0x008, (TM * TN) / (TM + TN)>(); });
        });
        vt_sched_fence();
        vt_lds_barrier();
        // ---- E: next patch into the buffer the M slot is done with, lean epilogue from registers ----------------------
        // (bias + LeakyReLU * gain -> bf16 NHWC, optional fused ToRGB: the host admits nothing else.  The generic conv_epilogue
        // in this loop hoisted ~100 registers of per-lane invariants out of it and spilled them; every reload is a vmcnt(0)
        // that also drains the patch in flight: 154 us instead of 110 for the one-group form)
        // Every vector-memory operation of the slot is hidden from the compiler and COUNTED (vt_common.hpp): beside the LDS-DMA
        // in flight hipcc would wait vmcnt(0) at the first use of the skip pixels, and the slot would end by waiting for the
        // acknowledgement of its own stores.  Order: patch pieces, [3 skip loads], 8 activation stores, [3 image stores]; the
        // wave leaves the slot when everything older than its stores has landed -- the patch, which is what the next M needs.
        if (j + 1 < ng) issue_patch(tile_of(j + 1));
        {
            const int tm = tile_of(j);
            const int im = tm / per_img, tr = tm - im * per_img;
            const PatchRows<TW> rowmap{im, (tr / tiles_x) * TH, (tr % tiles_x) * TW, p.Ho, p.Wo};
            // lane (q, l15) finishes the ToRGB of tile row q of its wave (reduce-scatter below): its skip pixel is fetched now
            const int m_rgb = rowmap(wm * (TM * 16) + q * 16 + l15);
            uint32_t o_rgb = GLDS_OOB;
            if (m_rgb >= 0) {
                const int ii = m_rgb / HoWo;
                o_rgb = (uint32_t)(ii * 3 * HoWo + (m_rgb - ii * HoWo)) * 4u;
            }
            u128 rsd[3];
            if (rgbf) {
#pragma unroll
                for (int jc = 0; jc < 3; ++jc) vt_bload_hidden<1>(rsd[jc], rskip, m_rgb >= 0 ? o_rgb + (uint32_t)(jc * HoWo) * 4u : GLDS_OOB);
            }
            const float ga = p.gain_alpha;
            const float* tab = reinterpret_cast<const float*>(smem + TAB_OFF);
            // Fused ToRGB on the matrix cores (conv_igemm.hip::conv_epilogue has the derivation): the 8 values a lane finishes
            // for a fragment pair, packed to bf16, ARE the pixel operand of v_mfma_f32_16x16x32_bf16 for the pair's 32 channels;
            // the weight operand of fragment row a carries the three planes in rows 4a .. 4a + 2, so one accumulator collects
            // all four rows and lane (q, l15) ends with the plane values of pixel l15 of tile row q -- no shuffles.
            f32x4 racc = f32x4{0.f, 0.f, 0.f, 0.f};
            static_assert(TM <= 4 && sizeof(T) == 2, "ToRGB accumulator rows");
#pragma unroll
            for (int hp = 0; hp < 2; ++hp) {   // channels 32 hp + 8q .. + 7: fragments 2 hp (first four) and 2 hp + 1
                const int c0 = hp * 32 + q * 8;
                float bv[8];
                unpack16<float>(ld128(tab + c0), bv), unpack16<float>(ld128(tab + c0 + 4), bv + 4);
                u128 wfrag = u128{0u, 0u, 0u, 0u};
                if (rgbf) {   // lane (q, l15): plane l15 & 3 (none for 3), the table holds the bf16 weights widened to fp32
                    const int pl = (l15 & 3) < 3 ? (l15 & 3) : 0;
                    float wv[8];
                    unpack16<float>(ld128(tab + (1 + pl) * BN + c0), wv), unpack16<float>(ld128(tab + (1 + pl) * BN + c0 + 4), wv + 4);
                    const u128 wp = pack16<bf16_t>(wv);
                    const bool okw = (l15 & 3) < 3;
                    wfrag.x = okw ? wp.x : 0u, wfrag.y = okw ? wp.y : 0u, wfrag.z = okw ? wp.z : 0u, wfrag.w = okw ? wp.w : 0u;
                }
#pragma unroll
                for (int a = 0; a < TM; ++a) {
                    const int m = rowmap(wm * (TM * 16) + a * 16 + l15);
                    float f[8];
#pragma unroll
                    for (int i = 0; i < 8; ++i) {
                        float v = acc[a][2 * hp + (i >> 2)][i & 3] + bv[i];
                        if (p.act == VT_ACT_LRELU) v = (v > 0.0f) ? v : v * p.slope;
                        f[i] = v * ga;
                    }
                    const u128 fpk = pack16<T>(f);
                    if (rgbf) {
                        const bool mine = (l15 >> 2) == a;
                        u128 wa;
                        wa.x = mine ? wfrag.x : 0u, wa.y = mine ? wfrag.y : 0u, wa.z = mine ? wfrag.z : 0u, wa.w = mine ? wfrag.w : 0u;
                        Mma<bf16_t>::run(racc, wa, fpk);
                    }
                    vt_bstore_hidden<4>(rout, m >= 0 ? (uint32_t)(m * p.ld_out + c0) * (uint32_t)ESZ : GLDS_OOB, fpk);
                }
            }
            if (rgbf) {
                const float rr[3] = {racc[0], racc[1], racc[2]};   // lane (q, l15): pixel l15 of tile row q
                vt_vmcnt_fence<2 * TM>();   // the skip pixels (and with them the patch: older) have landed; 8 stores may fly
#pragma unroll
                for (int jc = 0; jc < 3; ++jc) {
                    u128 o;
                    o.x = vt_f2u((rr[jc] + rb[jc]) + vt_u2f(vt_settled(rsd[jc]).x)), o.y = o.z = o.w = 0u;
                    vt_bstore_hidden<1>(rimg, m_rgb >= 0 ? o_rgb + (uint32_t)(jc * HoWo) * 4u : GLDS_OOB, o);
                }
            } else {
                vt_vmcnt_fence<2 * TM>();   // the patch has landed; the 8 stores may fly
            }
        }
        vt_lds_barrier();
    }
    // both groups leave after the same number of barriers (nmine + 1)
    for (int k = grp + 2 * ng; k < nmine + 1; ++k) vt_lds_barrier();
}
