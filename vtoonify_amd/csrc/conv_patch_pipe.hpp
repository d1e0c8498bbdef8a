// Patch-resident 3x3 convolution, software-pipelined across the per-tap barrier (round 4).
// Included by conv_igemm.hip inside its anonymous namespace (uses ConvArgs, GldsArgs, Mma, conv_epilogue, PatchRows).
//
// conv_patch_kernel (conv_igemm.hip) runs every filter tap as: barrier -> issue LDS-DMA -> read the first fragments ->
// s_waitcnt lgkmcnt(0) -> MFMAs, so the matrix pipe of every SIMD idles for one LDS round trip (with all 8 waves
// of the workgroup queueing on the LDS at once) plus the LDS-DMA issue cost at the top of each of the 9 x chunks
// taps: measured MFMA-busy 0.36 on the 256 x 128 tiles (profiles/r03_pmc_mfma.json).  This form keeps the same tile,
// the same loader and the same K order ([chunk][tap][half], so the results are bit-identical) and removes that
// bubble:
//   * the weight ring is 4 deep and the barrier at the end of tap s publishes the weights of tap s+2, so tap s+1's
//     fragments may be read BEFORE the barrier that ends tap s;
//   * fragments are double-buffered in registers: the source reads the 8 fragments of the next half-step (the first half
//     of the NEXT tap in the second half of a tap) before the 16 MFMAs of the current one.  The interleave is hipcc's:
//     pinning it with sched_group_barrier (one read per MFMA, or per two) measured 0-5 % slower on every layer
//     (profiles/r04_patch_pipeline.txt) -- with two waves per SIMD the partner's MFMAs cover a wave's LDS waits, what
//     counts is that nothing waits right behind the barrier;
//   * the LDS-DMA for tap s+D is issued between the two halves, not at the top of the step, and the next chunk's patch
//     arrives a few 1 KB pieces per wave per tap (taps 0..8-D) instead of PA pieces at once;
//   * the 9 taps are unrolled at compile time: every LDS offset of a fragment read is an immediate, every counted
//     `s_waitcnt vmcnt` is a constant (no branch ladder in front of the barrier), and loads past the end of the K
//     range are issued with an out-of-range offset (the buffer unit zero-fills a dead slot) so that every wave
//     issues the same number of operations in every step.
// LDS: 2 patch buffers + 4 weight slots = all 160 KB for the 256-pixel x 128-channel tile (one workgroup per CU).
//
// UP = 1: conv_transpose2d(3x3, stride 2, pad 0) -- the first half of an up-sampling StyledConv (model/stylegan/model.py:273-
// 286) -- by OUTPUT PARITY on the same pipeline.  z[2I+a, 2J+b] += W[a][b] x[I, J]: tap (a, b) only reaches output pixels of
// parity class (a & 1, b & 1), and for the "quad" (I', J') -- the 2 x 2 output pixels (2I'+pa, 2J'+pb) -- it reads input pixel
// (I' - a/2, J' - b/2).  With the patch origin where the 3x3 conv has it, that is the ordinary tap (ky, kx) = (a == 2 ? 0 : 1,
// b == 2 ? 0 : 1): the 9 weight taps are 9 GEMM steps over the resident patch, each accumulating into ONE of four
// accumulator sets.  9 MACs per input pixel (the reference's count), no halo recompute, tiles of TH x 16 quads over the
// (H+1) x (W+1) quad grid; the four sets leave through the ordinary epilogue with the rows mapped to (2I'+pa, 2J'+pb).
// This is the transposed convolution of the operator surface (conv2d_gradfix.conv_transpose2d, what the reference's own
// StyledConv(upsample) calls in eager mode).  Measured as a replacement for conv_upblur_kernel on the deep generator
// levels of the frame (z to HBM + a streaming blur pass): 80 + 94 + 110 us against 81 + 103 + 119 us for the one-kernel form
// before the blur pass (23 + 39 + 63 us) -- tiles over the (H+1) x (W+1) quad grid waste 2.1 / 1.5 / 1.25x at 33^2 / 65^2 /
// 129^2 and 288 / 400 / 648 workgroups quantise badly to 256 CUs: the raw rate is 0.8-0.9 PFLOP/s, the useful one 0.24-0.41.
// The engine keeps conv_upblur_kernel (profiles/r04_up_by_parity.txt).
#pragma once

// (vt_static_for: conv_igemm.hip, in front of the kernel headers)

// UP: tile row -> z pixel (2I'+pa, 2J'+pb) of the (2H+1) x (2W+1) transposed-conv output, or -1.  Parity 0 exists for
// I' in [0, H], parity 1 for I' in [0, H-1] (same for columns).
template <int TW>
struct QuadRows {
    int img, y0, x0, pa, pb, H, W;
    __device__ __forceinline__ int operator()(int row) const {
        const int qi = y0 + row / TW, qj = x0 + row % TW;
        return (qi <= H - pa && qj <= W - pb) ? (img * (2 * H + 1) + 2 * qi + pa) * (2 * W + 1) + 2 * qj + pb : -1;
    }
};

// DIL = 2: the dilated 3x3 convs of the AdaResBlocks (model/vtoonify.py:201-207) on the 256-pixel x 32-channel tiles of a batch
// (round 5): the patch grows to 20 x 20 pixels (51 KB per chunk), two of them + a 6-deep ring of 8 KB slots are exactly 160 KB.
// Dilation 4 (24 x 24 pixels, 74 KB per chunk) does not fit and stays on the weight-stationary kernel.
template <typename T, int TH, int BN, int WM, int WN, int NSTB = 4, int UP = 0, int EPI = 0, int DIL = 1>
__global__ void __launch_bounds__(WM * WN * 64)
conv_patchp_kernel(const ConvArgs p, const GldsArgs g) {
    constexpr int TW = 16;
    constexpr int NW = WM * WN;                 // wavefronts
    constexpr int BM = TH * TW;
    constexpr int ESZ = (int)sizeof(T);
    constexpr int VEC = 16 / ESZ;
    constexpr int BK = 8 * VEC;                 // channels per chunk (128 B)
    constexpr int TM = BM / WM / 16, TN = BN / WN / 16;
    constexpr bool PERM = (TN % 2 == 0);        // weight rows in fragment order: see tile_row_channel
    static_assert(DIL == 1 || UP == 0, "the transposed form is not dilated");
    constexpr int PH = TH + 2 * DIL, PW = TW + 2 * DIL, PROWS = PH * PW;
    constexpr int PA = ((PROWS + 7) / 8 + NW - 1) / NW;   // patch pieces (1 KB loads) per wave per chunk
    constexpr int LB = ((BN + 7) / 8 + NW - 1) / NW;      // weight loads per wave per tap
    constexpr int A_BYTES = PA * NW * 1024, B_BYTES = LB * NW * 1024;
    // NSTB = weight ring depth: the weights of tap s+D (D = NSTB-1) are issued in tap s.  The barrier that ends tap s
    // publishes tap s+2 whatever the depth; a deeper ring only lets the loads fly longer (the 32-channel tiles: 4 KB per
    // tap, steps too short to cover an L2 round trip with two taps in flight).  A patch piece issued in tap t is older
    // than the weights of tap t+D+1, which are waited for at the end of tap t+D-1: pieces go out in taps 0..8-D, PPT each.
    constexpr int D = NSTB - 1;
    constexpr int PTAPS = 9 - D;                            // taps that may carry patch pieces
    constexpr int PPT = (PA + PTAPS - 1) / PTAPS;           // pieces per such tap
    static_assert(BM % (WM * 16) == 0 && BN % (WN * 16) == 0, "wave tiling");
    static_assert(TM * WM == TH, "one 16-pixel tile row per MFMA row block");
    static_assert(NSTB >= 4 && NSTB <= 9, "ring depth");
    static_assert(2 * A_BYTES + NSTB * B_BYTES <= 160 * 1024, "LDS budget");
    static_assert((D - 2) * LB + PA < 64, "vmcnt is 6 bits");

    __shared__ __attribute__((aligned(1024))) unsigned char smem[2 * A_BYTES + NSTB * B_BYTES];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = vt_uniform(tid >> 6) & (NW - 1);
    const int wm = wave / WN, wn = wave % WN;
    int tile_m, tile_n, split;
    decode_block_2d(p, tile_m, tile_n, split);
    // (UP: tiles of quads over (H+1) x (W+1); p.Ho x p.Wo = (2H+1) x (2W+1) is the z image)
    const int tiles_x = UP ? (p.W + 1 + TW - 1) / TW : (p.Wo + TW - 1) / TW;
    const int tiles_y = UP ? (p.H + 1 + TH - 1) / TH : (p.Ho + TH - 1) / TH;
    const int img = tile_m / (tiles_x * tiles_y);
    const int trem = tile_m - img * (tiles_x * tiles_y);
    const int y0 = (trem / tiles_x) * TH, x0 = (trem % tiles_x) * TW;
    const int n0 = tile_n * BN;

    // ---- loader state (fixed for the whole kernel: only the SGPR offset moves) -------------
    const int lrow = lane >> 3;
    const int jj = (lane & 7) ^ lrow;
    uint32_t pa0[PA], pa1[PA];
#pragma unroll
    for (int i = 0; i < PA; ++i) {
        const int pr = (i * NW + wave) * 8 + lrow;
        const int py = pr / PW, px = pr - py * PW;
        const int iy = y0 - DIL + py, ix = x0 - DIL + px;
        const bool in = pr < PROWS && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
        const uint32_t pix = (uint32_t)((img * p.H + iy) * p.W + ix);
        pa0[i] = in ? pix * (uint32_t)(p.ld0 * ESZ) + jj * 16 : GLDS_OOB;
        pa1[i] = in ? pix * (uint32_t)(p.ld1 * ESZ) + jj * 16 : GLDS_OOB;
    }
    uint32_t woff[LB];
#pragma unroll
    for (int i = 0; i < LB; ++i) {
        const int row = (i * NW + wave) * 8 + lrow;
        const int n = n0 + tile_row_channel<PERM>(row);
        woff[i] = (row < BN && n < p.coutT) ? (uint32_t)n * (uint32_t)(p.K * ESZ) + jj * 16 : GLDS_OOB;
    }
    const BufRsrc r0 = vt_make_rsrc(p.src0, g.nrec0);
    const BufRsrc r1 = vt_make_rsrc(p.src1 ? p.src1 : p.src0, p.src1 ? g.nrec1 : 0u);
    const BufRsrc rw = vt_make_rsrc(p.wgt, g.nrecw);

    // K slices are whole chunks: p.kps chunks per slice
    const int nchunks = p.cin / BK;
    const int ch0 = split * p.kps;
    const int ch1 = (ch0 + p.kps < nchunks) ? ch0 + p.kps : nchunks;

    // piece `i` of the patch of `chunk` -> patch buffer at byte offset `aoff`; chunks at or beyond ch1 fetch zeros
    auto issue_a_piece = [&](int chunk, int aoff, int i) {
        const bool live = chunk < ch1;
        const int kc = chunk * BK;
        const bool s1 = live && kc >= p.c0;
        const uint32_t so = live ? (uint32_t)((s1 ? kc - p.c0 : kc) * ESZ) : 0u;
        const uint32_t vo = live ? (s1 ? pa1[i] : pa0[i]) : GLDS_OOB;
        vt_glds16(s1 ? r1 : r0, smem + aoff + (i * NW + wave) * 1024, vo, so);
    };
    // weights of (chunk, tap) -> ring slot at byte offset `boff`
    auto issue_b = [&](int chunk, int tap, int boff) {
        const bool live = chunk < ch1;
        const uint32_t so = live ? (uint32_t)((tap * p.cin + chunk * BK) * ESZ) : 0u;
#pragma unroll
        for (int i = 0; i < LB; ++i)
            vt_glds16(rw, smem + 2 * A_BYTES + boff + (i * NW + wave) * 1024, live ? woff[i] : GLDS_OOB, so);
    };

    constexpr int NCLS = UP ? 4 : 1;   // accumulator sets: output parity classes of the transposed conv
    f32x4 acc[NCLS][TM][TN];
#pragma unroll
    for (int c = 0; c < NCLS; ++c)
#pragma unroll
        for (int a = 0; a < TM; ++a)
#pragma unroll
            for (int b = 0; b < TN; ++b) acc[c][a][b] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int q = lane >> 4, l15 = lane & 15, l7 = lane & 7;
    // per-lane parts of the fragment addresses.  A: patch row pr = rowconst + wm*TM*PW + l15 with rowconst a compile-time
    // function of (tap, a); its swizzle phase is (pr & 7) = (wm*TM*PW + l15 + (rowconst & 7)) & 7 -> one of 8 lane patterns per half
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
    // fragments of (tap TAP, half SUB) from the patch at `aoff` and the weight slot at `boff`, in the order the MFMAs
    // consume them: fa[0], fb[0..TN-1], fa[1..TM-1]
    auto read_frags = [&](auto tapc, auto subc, u128 (&xa)[TM], u128 (&xb)[TN], int aoff, int boff) {
        constexpr int TAP = decltype(tapc)::value, SUB = decltype(subc)::value;
        constexpr int ky = UP ? (TAP / 3 == 2 ? 0 : 1) : (TAP / 3) * DIL, kx = UP ? (TAP % 3 == 2 ? 0 : 1) : (TAP % 3) * DIL;
        auto ra = [&](int a) {
            const int rowc = (a + ky) * PW + kx;   // folds: a and TAP are compile-time after unrolling
            xa[a] = ld128(smem + aoff + aswz[rowc & 7][SUB] + rowc * 128);
        };
        ra(0);
#pragma unroll
        for (int b = 0; b < TN; ++b) xb[b] = ld128(smem + boff + bfix[SUB] + b * 2048);
#pragma unroll
        for (int a = 1; a < TM; ++a) ra(a);
    };
    auto mma_all = [&](auto tapc, const u128 (&xa)[TM], const u128 (&xb)[TN]) {
        constexpr int TAP = decltype(tapc)::value;
        constexpr int cls = UP ? ((TAP / 3) & 1) * 2 + ((TAP % 3) & 1) : 0;
#pragma unroll
        for (int a = 0; a < TM; ++a)
#pragma unroll
            for (int b = 0; b < TN; ++b) Mma<T>::run(acc[cls][a][b], xb[b], xa[a]);
    };
    // the epilogue's per-lane tables (bias, slope, gain): fetched here, used after the K loop -- their memory round trip was the
    // first of three at the end of every workgroup
    EpiTables<TN> etab;
    epi_tables<TN, PERM>(p, n0 + wn * (TN * 16), q, etab);
    // ---- prologue: patch of the first chunk, weights of taps 0..2; patch + taps 0, 1 landed -------------
#pragma unroll
    for (int i = 0; i < PA; ++i) issue_a_piece(ch0, 0, i);
    vt_static_for<D>([&](auto sc) {   // (D <= 8 taps: all of the first chunk)
        constexpr int s0 = decltype(sc)::value;
        issue_b(ch0, s0, s0 * B_BYTES);
    });
    vt_glds_wait_n<(D - 2) * LB>();
    vt_lds_barrier();
    read_frags(std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{}, fa[0], fb[0], 0, 0);

    int aoff = 0;        // patch buffer of the current chunk
    int slot = 0;        // ring slot of the current tap (step & 3)
    for (int chunk = ch0; chunk < ch1; ++chunk) {
        vt_static_for<9>([&](auto tc) {
            constexpr int t = decltype(tc)::value;
            constexpr int t3 = (t + D) % 9, c3 = (t + D) / 9;       // (chunk + c3, tap t3) is issued in this step
            constexpr int t1 = (t + 1) % 9;
            // patch pieces of this tap: [P0, P1); pieces younger than the weights of tap s+2 (issued in tap s+2-D): those of
            // taps t-(D-2) .. t of this chunk (earlier taps of the previous chunk carry none: they are >= PTAPS)
            constexpr int P0 = t * PPT < PA ? t * PPT : PA, P1 = (t + 1) * PPT < PA ? (t + 1) * PPT : PA;
            constexpr int TLO = t - (D - 2) > 0 ? t - (D - 2) : 0;
            constexpr int YOUNG = (P1 < PA ? P1 : PA) - (TLO * PPT < PA ? TLO * PPT : PA);
            static_assert(t >= PTAPS ? P0 == P1 : true, "patch pieces only in the first 9-D taps");
            const int boff = slot * B_BYTES;
            const int slot1 = slot + 1 == NSTB ? 0 : slot + 1, slotd = slot == 0 ? NSTB - 1 : slot - 1;
            const int boff1 = slot1 * B_BYTES, boff3 = slotd * B_BYTES;
            vt_sched_fence();
            // first half: MFMAs on the fragments read before the barrier; the second half's fragments arrive
            read_frags(tc, std::integral_constant<int, 1>{}, fa[1], fb[1], aoff, boff);
            mma_all(tc, fa[0], fb[0]);
            // LDS-DMA of this step: weights of tap s+3 into the slot tap s-1 used (every wave is past it: barrier of
            // step s-1), one piece of the next chunk's patch into the other patch buffer
            issue_b(chunk + c3, t3, boff3);
            vt_static_for<P1 - P0>([&](auto ic) { issue_a_piece(chunk + 1, aoff ^ A_BYTES, P0 + decltype(ic)::value); });
            // second half: MFMAs on the second half's fragments; the first half of tap s+1 arrives (its weights were
            // published by the barrier of step s-1, its patch -- at t == 8 the next chunk's -- by that of tap 7)
            read_frags(std::integral_constant<int, t1>{}, std::integral_constant<int, 0>{}, fa[0], fb[0],
                       t == 8 ? (aoff ^ A_BYTES) : aoff, boff1);
            mma_all(tc, fa[1], fb[1]);
            vt_sched_fence();
            // the weights of tap s+2 (issued in step s+2-D) must have landed: younger are the weights of taps s+3 .. s+D and the
            // patch pieces of the steps since (YOUNG)
            vt_glds_wait_n<(D - 2) * LB + YOUNG>();
#ifdef VT_EMU
            vt_lds_barrier();
#else
            __builtin_amdgcn_s_barrier();   // fragment reads in flight cross it: they read slots this barrier does not free
#endif
            slot = slot1;
        });
        aoff ^= A_BYTES;
    }
    __syncthreads();
    if constexpr (UP) {
        vt_static_for<4>([&](auto cc) {
            constexpr int c = decltype(cc)::value;
            conv_epilogue<T, BM, BN, WM, WN>(p, acc[c], smem, QuadRows<TW>{img, y0, x0, c >> 1, c & 1, p.H, p.W}, n0, split,
                                             tile_n * p.tiles_m + tile_m, etab);
        });
    } else {
        conv_epilogue<T, BM, BN, WM, WN, EPI>(p, acc[0], smem, PatchRows<TW>{img, y0, x0, p.Ho, p.Wo}, n0, split,
                                              tile_n * p.tiles_m + tile_m, etab);
    }
}
