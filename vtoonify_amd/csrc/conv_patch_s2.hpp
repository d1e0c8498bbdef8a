// Stride-2 3x3 convolution (pad 1) on patch-resident tiles, by INPUT PARITY (round 6): the three down-sampling convs of the
// content encoder (model/vtoonify.py:167-176: nn.Conv2d(c, c', 3, 2, 1) + LeakyReLU).
// Included by conv_igemm.hip inside its anonymous namespace (uses ConvArgs, GldsArgs, Mma, conv_epilogue, PatchRows).
//
// Until round 6 these ran on the 1-D implicit-GEMM kernel (conv_igemm_glds_kernel, 64 x 64 / 64 x 128 tiles): every tap of every
// K step gathers its own 64 pixel rows, so a 64 x 64 tile ingests 16 KB per 32 MFMAs -- 8x what the matrix pipe needs -- and the
// two launches sat at 0.17 / 0.21 of the MFMA roof for three rounds.  A patch that stays in LDS for all its taps needs the taps
// to address it at unit stride; with stride 2 they do not.  They do per PARITY CLASS of the input pixel:
//
//     out[oy, ox] = sum_{ky, kx} W[ky][kx] . in[2 oy + ky - 1, 2 ox + kx - 1]
//     row 2 oy + ky - 1:  ky = 1 -> even row, index oy;   ky = 0 -> odd row, index oy - 1;   ky = 2 -> odd row, index oy
//
// so with E[i] = in[2 i], O[i] = in[2 i + 1] (rows; the same for columns) the conv is the sum of four DENSE stride-1 convs on the
// four quarter-size sub-images, with 1 (E,E), 2 (E,O), 2 (O,E) and 4 (O,O) taps at offsets {-1, 0}: 9 taps in all, no MAC wasted.
// A step of the K loop is one (64-channel chunk, class): the class's (TH+1) x (TW+1) sub-patch (37 KB) + its 1-4 weight slabs
// are LDS-resident, its taps run back to back (one barrier per step, the next step's pieces in flight meanwhile: the structure
// of conv_patch_chunk.hpp).  256 output pixels x 64 channels per workgroup, 8 waves of 64 x 32.
// Per chunk: 220 KB of LDS-DMA for 1152 MFMAs of a CU (33 B/clk is what a CU ingests from L2 in this access pattern,
// tools/probe/ingest_probe.hip) -- the 1-D form: 1.3 MB.
//
// K order: [chunk][class EE, EO, OE, OO][tap][half] -- not the 1-D kernel's, so the bits differ from it (both are fp32 sums of the
// same products; tests compare with the oracle at the usual tolerance and pin batch invariance).  H and W even, one source.
#pragma once

template <typename T, int TH, int BN, int WM, int WN, int EPI = 0>
__global__ void __launch_bounds__(WM * WN * 64)
conv_patchs2_kernel(const ConvArgs p, const GldsArgs g) {
    constexpr int TW = 16;
    constexpr int NW = WM * WN;
    constexpr int BM = TH * TW;
    constexpr int ESZ = (int)sizeof(T);
    constexpr int VEC = 16 / ESZ;
    constexpr int BK = 8 * VEC;                 // channels per chunk (128 B)
    constexpr int TM = BM / WM / 16, TN = BN / WN / 16;
    constexpr bool PERM = (TN % 2 == 0);
    constexpr int PH = TH + 1, PW = TW + 1, PROWS = PH * PW;   // sub-patch: sub-image rows y0-1 .. y0+TH-1
    constexpr int NPA = (PROWS + 7) / 8, NPB = BN / 8;          // 1 KB pieces: sub-patch, one tap's slab
    constexpr int PA = (NPA + NW - 1) / NW;
    constexpr int A_BYTES = NPA * 1024, B_BYTES = NPB * 1024;
    constexpr int S_BYTES = A_BYTES + 4 * B_BYTES;              // one step: sub-patch + up to four slabs
    static_assert(BM % (WM * 16) == 0 && BN % (WN * 16) == 0 && TM * WM == TH, "wave tiling");
    static_assert(NW % NPB == 0, "a wave's weight pieces are the same rows of every slab it loads");
    static_assert(2 * S_BYTES <= 160 * 1024, "LDS budget: two steps");

    __shared__ __attribute__((aligned(1024))) unsigned char smem[2 * S_BYTES];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = vt_uniform(tid >> 6) & (NW - 1);
    const int wm = wave / WN, wn = wave % WN;
    int tile_m, tile_n, split;
    decode_block_pixel_major(p, tile_m, tile_n);   // the channel tiles of a pixel tile share its patch in one XCD's L2
    split = 0;
    const int tiles_x = (p.Wo + TW - 1) / TW, tiles_y = (p.Ho + TH - 1) / TH;
    const int img = tile_m / (tiles_x * tiles_y);
    const int trem = tile_m - img * (tiles_x * tiles_y);
    const int y0 = (trem / tiles_x) * TH, x0 = (trem % tiles_x) * TW;
    const int n0 = tile_n * BN;

    // ---- loader: sub-patch pixel (py, px) = sub-image pixel (y0 - 1 + py, x0 - 1 + px) = input pixel (2 Y + pr, 2 X + pc);
    // H, W even: in-bounds does not depend on the class, which enters as a wave-uniform byte offset ----
    const int lrow = lane >> 3;
    const int jj = (lane & 7) ^ lrow;
    uint32_t pa0[PA];
#pragma unroll
    for (int i = 0; i < PA; ++i) {
        const int pr = (i * NW + wave) * 8 + lrow;
        const int py = pr / PW, px = pr - py * PW;
        const int Y = y0 - 1 + py, X = x0 - 1 + px;
        const bool in = pr < PROWS && (unsigned)(2 * Y) < (unsigned)p.H && (unsigned)(2 * X) < (unsigned)p.W;
        const uint32_t pix = (uint32_t)((img * p.H + 2 * Y) * p.W + 2 * X);
        pa0[i] = in ? pix * (uint32_t)(p.ld0 * ESZ) + jj * 16 : GLDS_OOB;
    }
    uint32_t woff;
    {
        const int row = (wave % NPB) * 8 + lrow;       // weight piece w = i * NW + wave of a class: slab w / NPB, rows (w % NPB) * 8 ..
        const int n = n0 + tile_row_channel<PERM>(row);
        woff = (n < p.coutT) ? (uint32_t)n * (uint32_t)(p.K * ESZ) + jj * 16 : GLDS_OOB;
    }
    const BufRsrc r0 = vt_make_rsrc(p.src0, g.nrec0);
    const BufRsrc rw = vt_make_rsrc(p.wgt, g.nrecw);
    const int nchunks = p.cin / BK;

    // class c = 2 * (row parity) + (column parity): its taps (ky, kx) in K order
    //   EE: (1,1)        EO: (1,0) (1,2)        OE: (0,1) (2,1)        OO: (0,0) (0,2) (2,0) (2,2)
    auto issue_step = [&](int chunk, auto cc, int soff) {
        constexpr int C = decltype(cc)::value;
        constexpr int PR = C >> 1, PC = C & 1;
        constexpr int NT = (PR ? 2 : 1) * (PC ? 2 : 1);
        const uint32_t so = (uint32_t)(((PR * p.W + PC) * p.ld0 + chunk * BK) * ESZ);
#pragma unroll
        for (int i = 0; i < PA; ++i)
            if ((i + 1) * NW <= NPA || i * NW + wave < NPA)
                vt_glds16(r0, smem + soff + (i * NW + wave) * 1024, pa0[i], so);
        constexpr int LBR = (NT * NPB + NW - 1) / NW;   // weight rounds of the class
#pragma unroll
        for (int i = 0; i < LBR; ++i) {
            const int w = i * NW + wave;                // piece of the class's NT * NPB
            const int j = w / NPB;                      // its tap (wave-uniform)
            const int ky = PR ? (j / (PC ? 2 : 1)) * 2 : 1, kx = PC ? (j % 2) * 2 : 1;
            if ((i + 1) * NW <= NT * NPB || w < NT * NPB)
                vt_glds16(rw, smem + soff + A_BYTES + w * 1024, woff, (uint32_t)(((ky * 3 + kx) * p.cin + chunk * BK) * ESZ));
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

    // half-step h of a class = 2 * tap + half; tap j reads the sub-patch at offset (dy, dx) in {-1, 0}^2:
    //   odd-parity axis: tap index along it 0 -> -1 (ky = 0), 1 -> 0 (ky = 2);  even-parity axis: 0
    constexpr int FD = 3;
    u128 fa[FD][TM], fb[FD][TN];
    auto read_frags = [&](auto cc, auto hc, u128 (&xa)[TM], u128 (&xb)[TN], int soff) {
        constexpr int C = decltype(cc)::value, H = decltype(hc)::value;
        constexpr int PR = C >> 1, PC = C & 1;
        constexpr int J = H / 2, SUB = H % 2;
        constexpr int dy = PR ? (J / (PC ? 2 : 1)) - 1 : 0, dx = PC ? (J % 2) - 1 : 0;
#pragma unroll
        for (int b = 0; b < TN; ++b) xb[b] = ld128(smem + soff + J * B_BYTES + bfix[SUB] + b * 2048);
#pragma unroll
        for (int a = 0; a < TM; ++a) {
            const int rowc = (a + 1 + dy) * PW + 1 + dx;
            xa[a] = ld128(smem + soff + aswz[rowc & 7][SUB] + rowc * 128);
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

    issue_step(0, std::integral_constant<int, 0>{}, 0);
    vt_glds_wait();
    vt_lds_barrier();

    int soff = 0;
    for (int chunk = 0; chunk < nchunks; ++chunk) {
        vt_static_for<4>([&](auto cc) {
            constexpr int C = decltype(cc)::value;
            constexpr int NH = 2 * ((C >> 1) ? 2 : 1) * ((C & 1) ? 2 : 1);   // half-steps of the class
            // the next step's pieces: the next class of this chunk, or class EE of the next chunk
            if constexpr (C < 3) {
                issue_step(chunk, std::integral_constant<int, C + 1>{}, soff ^ S_BYTES);
            } else {
                if (chunk + 1 < nchunks) issue_step(chunk + 1, std::integral_constant<int, 0>{}, soff ^ S_BYTES);
            }
            vt_static_for<(FD - 1 < NH ? FD - 1 : NH)>([&](auto hc) {
                read_frags(cc, hc, fa[decltype(hc)::value % FD], fb[decltype(hc)::value % FD], soff);
            });
            vt_static_for<NH>([&](auto hc) {
                constexpr int h = decltype(hc)::value;
                vt_sched_fence();
                if constexpr (h + FD - 1 < NH)
                    read_frags(cc, std::integral_constant<int, h + FD - 1>{}, fa[(h + FD - 1) % FD], fb[(h + FD - 1) % FD], soff);
                mma_all(fa[h % FD], fb[h % FD]);
            });
            vt_sched_fence();
            vt_glds_wait();
            vt_lds_barrier();
            soff ^= S_BYTES;
        });
    }
    __syncthreads();
    conv_epilogue<T, BM, BN, WM, WN, EPI>(p, acc, smem, PatchRows<TW>{img, y0, x0, p.Ho, p.Wo}, n0, split,
                                          tile_n * p.tiles_m + tile_m, etab);
}

template <typename T>
static bool patchs2_eligible(const ConvArgs& a, GldsArgs& g) {
    constexpr int ESZ = (int)sizeof(T);
    constexpr int BK = 8 * (16 / ESZ);
    if (a.force_generic || a.transposed || a.in_scale || a.src1 || a.c1 != 0 || a.x3) return false;
    if (a.stride != 2 || a.taps != 9 || a.kw != 3 || a.pad != 1 || a.pad_x != 1 || a.dil != 1 || a.phases != 1) return false;
    if (a.H % 2 || a.W % 2 || a.Ho != a.H / 2 || a.Wo != a.W / 2) return false;
    if (a.c0 % BK != 0 || a.cin != a.c0 || a.coutT % 8 != 0) return false;
    if (a.rgb_w || a.stats_part || a.tile_stats || a.in_tile_stats || a.up_fir) return false;
    const int64_t lim = ((int64_t)1 << 31) - 4096;
    const int64_t n0 = (int64_t)a.N * a.H * a.W * a.ld0 * ESZ, nw = (int64_t)a.coutT * a.K * ESZ;
    // largest scalar offset: class (O, O) + the last chunk
    if (n0 + ((int64_t)a.W + 1) * a.ld0 * ESZ >= lim || nw >= lim) return false;
    g.nrec0 = (uint32_t)n0;
    g.nrec1 = 0;
    g.nrecw = (uint32_t)nw;
    g.bias0 = g.bias1 = 0;
    return true;
}

// BN = 64: 8 waves of 64 pixels x 32 channels.  BN = 32 (8 waves of 32 x 32): twice the workgroups -- the deepest stage (64^2 ->
// 32^2, 16 pixel tiles at 4 frames) fills the GPU with it; its patches are read twice as often.
template <typename T, int BN = 64>
int launch_patchs2(const ConvArgs& a, const GldsArgs& g, vt_stream stream) {
    constexpr int TH = 16, WM = BN == 64 ? 4 : 8, WN = BN == 64 ? 2 : 1;
    ConvArgs args = a;
    args.slab_perm = ((BN / WN / 16) % 2 == 0) ? 1 : 0;
    args.tiles_n = vt_cdiv(a.coutT, BN);
    args.tiles_m = a.N * vt_cdiv(a.Ho, TH) * vt_cdiv(a.Wo, 16);
    args.splitk = 1;
    args.kps = a.cin / (8 * (16 / (int)sizeof(T)));
    args.tickets = nullptr;
    const int64_t blocks = (int64_t)args.tiles_m * args.tiles_n;
    if (blocks >= ((int64_t)1 << 31)) {
        vt_set_error("vt_conv2d: too many tiles");
        return VT_ERR_ARG;
    }
    if (conv_lean<T>(args)) {
        auto k = conv_patchs2_kernel<T, TH, BN, WM, WN, 1>;
        VT_LAUNCH(k, dim3((unsigned)blocks), dim3(WM * WN * 64), stream, args, g);
    } else {
        auto k = conv_patchs2_kernel<T, TH, BN, WM, WN, 0>;
        VT_LAUNCH(k, dim3((unsigned)blocks), dim3(WM * WN * 64), stream, args, g);
    }
    return vt_check_launch("vt_conv2d(patch, stride 2)");
}
