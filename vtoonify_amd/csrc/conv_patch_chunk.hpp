// Patch-resident 3x3 convolution with ONE barrier per K chunk (round 6): the 32-channel tiles of the 32 x 32 trunk.
// Included by conv_igemm.hip inside its anonymous namespace (uses ConvArgs, GldsArgs, Mma, conv_epilogue, PatchRows).
//
// conv_patchp_kernel (conv_patch_pipe.hpp) ends every filter tap with a counted vmcnt wait + s_barrier, because its weight
// ring holds a few taps.  On the 256-pixel x 128-channel tiles a tap is 32 MFMAs per wave and the barrier is noise; on the
// 256 x 32 tiles of the trunk (one workgroup per CU at 4 frames: 4096 pixels x 512 channels = 256 tiles) a tap is 8 MFMAs per
// wave -- 128 matrix-pipe cycles between two rendezvous of 8 waves -- and the step measured 756 cycles against 256 of MFMA
// work (22.7 us per conv, 0.34 of the roof; profiles/r05_bench_kernels.txt).  The round-4 ablation of the big tile put
// "MFMA + barrier alone" at 1330 cycles per 1024-cycle step: ~300 cycles per rendezvous whatever the step holds.  And the
// per-CU L2 -> LDS rate the step was thought to be bound by is 33 B/clk for this access pattern, not 15 (tools/probe/
// ingest_probe.hip, profiles/r06_ingest_probe.txt): the step's 8.6 KB are 260 cycles of it.
//
// Here a 32-channel tile keeps ALL NINE taps of a chunk resident: patch (18 x 18 pixels x 128 B = 41.5 KB) + 9 weight slabs
// (9 x 4 KB) = 77.5 KB per chunk, two chunks = 155 KB of the 160 KB LDS.  Per chunk: every LDS-DMA piece of chunk c+1 is
// issued up front (it has the whole chunk to land), the 9 taps x 2 halves run back to back with register-double-buffered
// fragments, then ONE vmcnt(0) + barrier.  72 MFMAs per wave between rendezvous instead of 8.  Same tile, same loader
// addresses, same K order ([chunk][tap][half]) as conv_patchp_kernel<T,16,32,8,1,...>: the results are bit-identical
// (tests/test_ops.py::test_conv_patch_chunk_equals_pipelined).
//
// DIL = 2 (the dilation-2 convs of the AdaResBlocks, model/vtoonify.py:201-207): a 3x3 conv with dilation d and padding d is d*d
// independent DENSE 3x3 convs with padding 1 on the sub-images {(y, x): y % d = sy, x % d = sx}.  A tile is 16 x 16 pixels of ONE
// sub-image: its patch is the dense 18 x 18 patch of that sub-image (41.5 KB -- conv_patchp_kernel<.., DIL = 2> stages the 20 x 20
// patch of the full image, 51 KB, which does not leave room for two whole chunks), the loader strides the pixel address by d and
// the output row map strides it back.  K order [chunk][tap][half] as before: the same bits as the tap-granular dilated form.
#pragma once

// tile row -> output pixel of the sub-image (sy, sx) of a DIL-dilated conv (DIL = 1: PatchRows)
template <int TW, int DIL>
struct SubImageRows {
    int img, y0, x0, sy, sx, Ho, Wo;   // (y0, x0): the tile's origin in sub-image coordinates
    __device__ __forceinline__ int operator()(int row) const {
        const int oy = (y0 + row / TW) * DIL + sy, ox = (x0 + row % TW) * DIL + sx;
        return (oy < Ho && ox < Wo) ? (img * Ho + oy) * Wo + ox : -1;
    }
};

template <typename T, int TH, int BN, int WM, int WN, int EPI = 0, int DIL = 1>
__global__ void __launch_bounds__(WM * WN * 64)
conv_patchc_kernel(const ConvArgs p, const GldsArgs g) {
    constexpr int TW = 16;
    constexpr int NW = WM * WN;
    constexpr int BM = TH * TW;
    constexpr int ESZ = (int)sizeof(T);
    constexpr int VEC = 16 / ESZ;
    constexpr int BK = 8 * VEC;                 // channels per chunk (128 B)
    constexpr int TM = BM / WM / 16, TN = BN / WN / 16;
    constexpr bool PERM = (TN % 2 == 0);
    constexpr int PH = TH + 2, PW = TW + 2, PROWS = PH * PW;
    // LDS-DMA pieces (1 KB = 8 rows of 128 B per wave-instruction), packed tight: the patch is NPA pieces, a tap's slab NPB;
    // piece j of a kind goes to wave j % NW in round j / NW, and a wave skips the rounds whose piece does not exist (a
    // wave-uniform branch: a dead piece would zero-fill 1 KB of its neighbour's region)
    constexpr int NPA = (PROWS + 7) / 8, NPB = BN / 8;
    constexpr int PA = (NPA + NW - 1) / NW;               // patch rounds
    constexpr int LB = (9 * NPB + NW - 1) / NW;           // weight rounds: the nine slabs of a chunk are one run of 9 * NPB pieces
    constexpr int A_BYTES = NPA * 1024, B_BYTES = NPB * 1024;   // patch, one tap's slab
    constexpr int C_BYTES = A_BYTES + 9 * B_BYTES;        // one chunk
    static_assert(BM % (WM * 16) == 0 && BN % (WN * 16) == 0, "wave tiling");
    static_assert(TM * WM == TH, "one 16-pixel tile row per MFMA row block");
    static_assert(BN % 8 == 0 && NW % NPB == 0, "a wave's weight pieces are the same rows of every slab it loads");
    static_assert(2 * C_BYTES <= 160 * 1024, "LDS budget: two whole chunks");
    static_assert(PA + LB < 64, "vmcnt is 6 bits");

    __shared__ __attribute__((aligned(1024))) unsigned char smem[2 * C_BYTES];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = vt_uniform(tid >> 6) & (NW - 1);
    const int wm = wave / WN, wn = wave % WN;
    int tile_m, tile_n, split;
    decode_block_2d(p, tile_m, tile_n, split);
    // tiles of one image: [sub-image sy][sx][tile row][tile column] over sub-images of ceil(Ho / DIL) x ceil(Wo / DIL) pixels
    const int tiles_x = ((p.Wo + DIL - 1) / DIL + TW - 1) / TW;
    const int tiles_y = ((p.Ho + DIL - 1) / DIL + TH - 1) / TH;
    const int per_img = DIL * DIL * tiles_x * tiles_y;
    const int img = tile_m / per_img;
    int trem = tile_m - img * per_img;
    const int sub = trem / (tiles_x * tiles_y);
    trem -= sub * (tiles_x * tiles_y);
    const int sy = sub / DIL, sx = sub - sy * DIL;
    const int y0 = (trem / tiles_x) * TH, x0 = (trem % tiles_x) * TW;   // sub-image coordinates
    const int n0 = tile_n * BN;

    // ---- loader state: the addresses of conv_patchp_kernel (rows of 128 B, 16-byte columns XOR-swizzled by the row) ----
    const int lrow = lane >> 3;
    const int jj = (lane & 7) ^ lrow;
    uint32_t pa0[PA], pa1[PA];
#pragma unroll
    for (int i = 0; i < PA; ++i) {
        const int pr = (i * NW + wave) * 8 + lrow;
        const int py = pr / PW, px = pr - py * PW;
        const int iy = (y0 - 1 + py) * DIL + sy, ix = (x0 - 1 + px) * DIL + sx;   // (negative only through y0 - 1 + py = -1)
        const bool in = pr < PROWS && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
        const uint32_t pix = (uint32_t)((img * p.H + iy) * p.W + ix);
        pa0[i] = in ? pix * (uint32_t)(p.ld0 * ESZ) + jj * 16 : GLDS_OOB;
        pa1[i] = in ? pix * (uint32_t)(p.ld1 * ESZ) + jj * 16 : GLDS_OOB;
    }
    // weight piece w = i * NW + wave of the chunk's 9 * NPB: tap w / NPB, rows (w % NPB) * 8 + lrow -- and w % NPB = wave % NPB
    // whatever the round, so a wave's weight rows never change
    const int wsub = wave % NPB;
    uint32_t woff;
    {
        const int row = wsub * 8 + lrow;
        const int n = n0 + tile_row_channel<PERM>(row);
        woff = (n < p.coutT) ? (uint32_t)n * (uint32_t)(p.K * ESZ) + jj * 16 : GLDS_OOB;
    }
    const BufRsrc r0 = vt_make_rsrc(p.src0, g.nrec0);
    const BufRsrc r1 = vt_make_rsrc(p.src1 ? p.src1 : p.src0, p.src1 ? g.nrec1 : 0u);
    const BufRsrc rw = vt_make_rsrc(p.wgt, g.nrecw);

    const int nchunks = p.cin / BK;
    const int ch0 = split * p.kps;
    const int ch1 = (ch0 + p.kps < nchunks) ? ch0 + p.kps : nchunks;

    // the whole of `chunk` (patch + nine slabs) -> the chunk buffer at byte offset `coff`; beyond ch1: nothing is issued
    auto issue_chunk = [&](int chunk, int coff) {
        const int kc = chunk * BK;
        const bool s1 = kc >= p.c0;
        const uint32_t so = (uint32_t)((s1 ? kc - p.c0 : kc) * ESZ);
#pragma unroll
        for (int i = 0; i < PA; ++i)
            if ((i + 1) * NW <= NPA || i * NW + wave < NPA)   // (first clause: compile-time for the full rounds)
                vt_glds16(s1 ? r1 : r0, smem + coff + (i * NW + wave) * 1024, s1 ? pa1[i] : pa0[i], so);
#pragma unroll
        for (int i = 0; i < LB; ++i) {
            const int w = i * NW + wave;
            const int tap = w / NPB;
            if ((i + 1) * NW <= 9 * NPB || w < 9 * NPB)
                vt_glds16(rw, smem + coff + A_BYTES + w * 1024, woff, (uint32_t)((tap * p.cin + kc) * ESZ));
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
        bfix[sub] = (uint32_t)(A_BYTES + (wn * (TN * 16) + l15) * 128 + (((sub * 4 + q) ^ l7) << 4));

    // fragments: a ring of FD register buffers, read FD - 1 half-steps ahead of their MFMAs.  A half-step of this tile is 4
    // MFMAs per wave (64 matrix-pipe cycles, 128 with the SIMD's other wave): one half-step of lead does not cover an LDS round
    // trip with 8 waves reading (the tap-granular kernel and the first form of this one ran the K loop at ~490 cycles per tap
    // against 320 of MFMA issue whatever the barriers, the LDS bytes or the ring depth were -- profiles/r06_trunk_chunk.txt).
    // All 18 half-steps of a chunk are LDS-resident here, so the lead is a compile-time choice.
    constexpr int FD = 4;
    u128 fa[FD][TM], fb[FD][TN];
    auto read_frags = [&](auto hc, u128 (&xa)[TM], u128 (&xb)[TN], int coff) {   // half-step hc = 2 * tap + half
        constexpr int TAP = decltype(hc)::value / 2, SUB = decltype(hc)::value % 2;
        constexpr int ky = TAP / 3, kx = TAP % 3;
#pragma unroll
        for (int b = 0; b < TN; ++b) xb[b] = ld128(smem + coff + TAP * B_BYTES + bfix[SUB] + b * 2048);
#pragma unroll
        for (int a = 0; a < TM; ++a) {
            const int rowc = (a + ky) * PW + kx;
            xa[a] = ld128(smem + coff + aswz[rowc & 7][SUB] + rowc * 128);
        }
    };
    auto mma_all = [&](const u128 (&xa)[TM], const u128 (&xb)[TN]) {
#pragma unroll
        for (int a = 0; a < TM; ++a)
#pragma unroll
            for (int b = 0; b < TN; ++b) Mma<T>::run(acc[a][b], xb[b], xa[a]);
    };
    EpiTables<TN> etab;
    epi_tables<TN, PERM>(p, n0 + wn * (TN * 16), q, etab);

    // ---- prologue: the first chunk ----
    issue_chunk(ch0, 0);
    vt_glds_wait();
    vt_lds_barrier();

    int coff = 0;
    for (int chunk = ch0; chunk < ch1; ++chunk) {
        if (chunk + 1 < ch1) issue_chunk(chunk + 1, coff ^ C_BYTES);   // (wave-uniform branch; lands during this chunk's taps)
        vt_static_for<FD - 1>([&](auto hc) { read_frags(hc, fa[decltype(hc)::value], fb[decltype(hc)::value], coff); });
        vt_static_for<18>([&](auto hc) {
            constexpr int h = decltype(hc)::value;
            vt_sched_fence();
            if constexpr (h + FD - 1 < 18)
                read_frags(std::integral_constant<int, h + FD - 1>{}, fa[(h + FD - 1) % FD], fb[(h + FD - 1) % FD], coff);
            mma_all(fa[h % FD], fb[h % FD]);
        });
        vt_sched_fence();
        // the next chunk has landed (this wave's pieces: vmcnt(0); the others': the barrier), and every wave is done reading
        // the buffer the chunk after it will overwrite
        vt_glds_wait();
        vt_lds_barrier();
        coff ^= C_BYTES;
    }
    __syncthreads();
    conv_epilogue<T, BM, BN, WM, WN, EPI>(p, acc, smem, SubImageRows<TW, DIL>{img, y0, x0, sy, sx, p.Ho, p.Wo}, n0, split,
                                          tile_n * p.tiles_m + tile_m, etab);
}
