// The dilation-4 convs of the trunk's AdaResBlocks (model/vtoonify.py:201-207: 3x3, dilation 4, padding 4 on the H/8 x W/8 plane) on
// flat 8 x 8 blocks of their sub-images (round 6).  Included by conv_igemm.hip inside its anonymous namespace.
//
// A 3x3 conv with dilation d and padding d is d*d independent DENSE 3x3 convs with padding 1 on the sub-images {(y, x): y % d = sy,
// x % d = sx} (conv_patch_chunk.hpp runs dilation 2 that way on 16 x 16 tiles).  At d = 4 the sub-images of a 32 x 32 plane are
// 8 x 8 pixels: a 16 x 16 tile would be three quarters empty and the full-image patch of a 16 x 16 tile is 24 x 24 pixels (74 KB
// per 64-channel chunk), so these four convs stayed on the weight-stationary kernel (conv_fullkw.hpp: 31 us each at 4 frames, 0.25
// of the MFMA roof, against 21 us for the dense ones).  Here:
//
//   * a UNIT is an 8 x 8 block of one sub-image with its 10 x 10 patch, both FLAT with pitch 10 (conv_upblur_flat.hpp's trick):
//     output position f = 10 y + x reads patch slot f + 10 dy + dx for tap (dy, dx), so an MFMA fragment is 16 consecutive LDS rows
//     whatever row of the block it straddles.  80 positions = exactly 5 fragments, of which the x = 8, 9 columns are dropped
//     (64 of 80 useful);
//   * a workgroup = 4 waves = 4 units (one wave per SIMD, a unit per wave: 5 pixel fragments x 2 channel fragments), 32 output
//     channels; 4 frames of a 32 x 32 plane = 64 units = 16 tiles x 16 channel tiles = 256 workgroups;
//   * K in steps of 32 channels, 64-byte LDS rows, three 48 KB stages in a ring (4 x 7 patch pieces + 18 weight pieces + 2 out of
//     range per stage, 12 per wave: one counted vmcnt per step), the 9 taps of a step software-pipelined three deep: 7 fragment
//     reads per 10 MFMAs (the 8-wave tiles of conv_patch_chunk.hpp: 4 per 4).
//
// K order [32-channel step][tap]: not the bits of the whole-K kernels -- chosen by the batch rule that already chooses the patch
// tiles of the dense trunk convs (never under VT_BATCH_EXACT).
#pragma once

constexpr int F8_SIDE = 8, F8_PITCH = 10, F8_POS = 80, F8_SLOTS = 112;   // block side, flat pitch, positions and patch slots per unit

template <int DIL>
struct Flat8Rows {
    int img, oy0, ox0, Ho, Wo, live;   // this WAVE's unit: first output pixel (oy0, ox0), stride DIL; live = the unit exists
    int mod, base;                     // the epilogue runs on fragments 0..3 and on fragment 4: position = base + row % mod
    __device__ __forceinline__ int operator()(int row) const {
        const int f = base + row % mod;
        const int y = f / F8_PITCH, x = f - y * F8_PITCH;
        const int oy = oy0 + y * DIL, ox = ox0 + x * DIL;
        return (live && x < F8_SIDE && oy < Ho && ox < Wo) ? (img * Ho + oy) * Wo + ox : -1;
    }
};

struct Flat8Args {
    int nby, nbx;      // 8 x 8 blocks per sub-image
    int units;         // N * DIL * DIL * nby * nbx
};

template <typename T, int BN, int EPI, int DIL>
__global__ void __launch_bounds__(256, 1)
conv_flat8_kernel(const ConvArgs p, const GldsArgs g, const Flat8Args fa8) {
    static_assert(sizeof(T) == 2 && !is_x3<T>::value, "16-bit operands");
    constexpr int ESZ = 2;
    constexpr int NW = 4, UNITS = 4;
    constexpr int HK = 32;                              // input channels per K step
    constexpr int TM = F8_POS / 16, TN = BN / 16;       // 5 pixel fragments per wave, BN / 16 channel fragments
    constexpr bool PERM = (TN % 2 == 0);
    constexpr int UPIECES = F8_SLOTS / 16;              // 7 pieces of 16 patch slots x 64 B per unit
    constexpr int PPIECES = UNITS * UPIECES;
    constexpr int WROWS = 9 * BN, WPIECES = WROWS / 16;
    constexpr int PL = (PPIECES + WPIECES + NW - 1) / NW;
    constexpr int A_BYTES = PPIECES * 1024;
    constexpr int STAGE = PL * NW * 1024;
    constexpr int NST = 3, FD = 3;
    static_assert(TM == 5 && F8_POS % 16 == 0 && F8_SLOTS % 16 == 0 && F8_POS + 2 * F8_PITCH + 2 <= F8_SLOTS, "unit geometry");
    static_assert(WROWS % 16 == 0 && NST * STAGE <= 160 * 1024 && 2 * PL < 64, "stage");

    __shared__ __attribute__((aligned(1024))) unsigned char smem[NST * STAGE];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = vt_uniform(tid >> 6) & (NW - 1);
    const int q = lane >> 4, l15 = lane & 15;
    int tile_m, tile_n, split;
    decode_block_2d(p, tile_m, tile_n, split);
    const int n0 = tile_n * BN;
    const int per_sub = fa8.nby * fa8.nbx, per_img = DIL * DIL * per_sub;
    // unit U -> image, first pixel of its block in the image, or live = 0
    auto unit = [&](int U, int& img, int& oy0, int& ox0) -> int {
        const int ok = U < fa8.units;
        const int V = ok ? U : 0;
        img = V / per_img;
        const int r = V - img * per_img;
        const int sub = r / per_sub, blk = r - sub * per_sub;
        const int by = blk / fa8.nbx, bx = blk - by * fa8.nbx;
        const int sy = sub / DIL, sx = sub - sy * DIL;
        oy0 = by * F8_SIDE * DIL + sy;
        ox0 = bx * F8_SIDE * DIL + sx;
        return ok;
    };

    // ---- loader: lane l of a piece = row l >> 2 of its 16 rows, physical 16-byte slot l & 3; logical slot = (l & 3) ^ ((row >> 2) & 3)
    const int lrow = lane >> 2;
    const int jj = (lane & 3) ^ ((lane >> 4) & 3);
    uint32_t ldo[PL];
#pragma unroll
    for (int i = 0; i < PL; ++i) {
        const int pc = i * NW + wave;
        if (pc < PPIECES) {
            const int u = pc / UPIECES, s = (pc - u * UPIECES) * 16 + lrow;   // patch slot s of unit u: pixel (s / 10 - 1, s % 10 - 1) of its block
            int img, oy0, ox0;
            const int ok = unit(tile_m * UNITS + u, img, oy0, ox0);
            const int py = s / F8_PITCH, px = s - py * F8_PITCH;
            const int iy = oy0 + (py - 1) * DIL, ix = ox0 + (px - 1) * DIL;
            const bool in = ok && s < F8_PITCH * F8_PITCH && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
            const uint32_t pix = (uint32_t)((img * p.H + iy) * p.W + ix);
            ldo[i] = in ? pix * (uint32_t)(p.ld0 * ESZ) + jj * 16 : GLDS_OOB;
        } else {
            const int row = (pc - PPIECES) * 16 + lrow;   // weights of a step: LDS row tap * BN + r (fragment order)
            const int tap = row / BN, r = row - tap * BN;
            const int n = n0 + tile_row_channel<PERM>(r);
            ldo[i] = (row < WROWS && n < p.coutT) ? (uint32_t)n * (uint32_t)(p.K * ESZ) + (uint32_t)(tap * p.cin * ESZ) + jj * 16
                                                   : GLDS_OOB;
        }
    }
    const BufRsrc r0 = vt_make_rsrc(p.src0, g.nrec0);
    const BufRsrc rw = vt_make_rsrc(p.wgt, g.nrecw);
    const int nsteps = p.cin / HK;
    auto issue = [&](int step, int st) {
        const uint32_t so = (uint32_t)(step * HK * ESZ);
        unsigned char* base = smem + st * STAGE;
#pragma unroll
        for (int i = 0; i < PL; ++i) {
            const int pc = i * NW + wave;
            if (pc < PPIECES) vt_glds16(r0, base + pc * 1024, ldo[i], so);
            else vt_glds16(rw, base + pc * 1024, ldo[i], so);
        }
    };

    // fragment addresses inside a stage: this wave's unit, position fragment 0, per tap (fragment a: + a KiB); weights by lane
    uint32_t aoff[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) {
        const int pr = wave * F8_SLOTS + l15 + (t / 3) * F8_PITCH + (t % 3);
        aoff[t] = (uint32_t)(pr * 64 + ((q ^ ((pr >> 2) & 3)) << 4));
    }
    const uint32_t boff = (uint32_t)(A_BYTES + l15 * 64 + ((q ^ ((l15 >> 2) & 3)) << 4));

    f32x4 acc[TM][TN];
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
    EpiTables<TN> etab;
    epi_tables<TN, PERM>(p, n0, q, etab);

    issue(0, 0);
    if (nsteps > 1) issue(1, 1);
    int st = 0;
    for (int step = 0; step < nsteps; ++step) {
        if (step + 1 < nsteps) vt_glds_wait_n<PL>();   // the younger stage may still fly
        else vt_glds_wait_n<0>();
        vt_lds_barrier();                                // everybody's pieces of this stage; every wave is past stage st - 1
        if (step + 2 < nsteps) issue(step + 2, st == 0 ? NST - 1 : st - 1);
        const unsigned char* sbase = smem + st * STAGE;
        u128 fa[FD][TM], fb[FD][TN];
        auto read_tap = [&](auto tc) {
            constexpr int t = decltype(tc)::value;
#pragma unroll
            for (int a = 0; a < TM; ++a) fa[t % FD][a] = ld128(sbase + aoff[t] + a * 1024);
#pragma unroll
            for (int b = 0; b < TN; ++b) fb[t % FD][b] = ld128(sbase + boff + (t * BN + b * 16) * 64);
        };
        vt_static_for<FD - 1>([&](auto tc) { read_tap(tc); });
        vt_static_for<9>([&](auto tc) {
            constexpr int t = decltype(tc)::value;
            vt_sched_fence();
            if constexpr (t + FD - 1 < 9) read_tap(std::integral_constant<int, (t + FD - 1 < 9 ? t + FD - 1 : 0)>{});
#pragma unroll
            for (int a = 0; a < TM; ++a)
#pragma unroll
                for (int b = 0; b < TN; ++b) Mma<T>::run(acc[a][b], fb[t % FD][b], fa[t % FD][a]);
        });
        vt_sched_fence();
        st = st + 1 == NST ? 0 : st + 1;
    }
    __syncthreads();
    Flat8Rows<DIL> rows;
    rows.live = unit(tile_m * UNITS + wave, rows.img, rows.oy0, rows.ox0);
    rows.Ho = p.Ho, rows.Wo = p.Wo;
    // (conv_epilogue finishes at most four fragment rows per wave at a time: its fused-ToRGB accumulator has 16 rows)
    rows.mod = 64, rows.base = 0;
    conv_epilogue<T, NW * 64, BN, NW, 1, EPI>(p, reinterpret_cast<f32x4 (&)[4][TN]>(acc[0]), smem, rows, n0, 0,
                                              tile_n * p.tiles_m + tile_m, etab);
    rows.mod = 16, rows.base = 64;
    conv_epilogue<T, NW * 16, BN, NW, 1, EPI>(p, reinterpret_cast<f32x4 (&)[1][TN]>(acc[4]), smem, rows, n0, 0,
                                              tile_n * p.tiles_m + tile_m, etab);
}

// what the kernel takes: one 16-bit source, dilation 4, no K split, none of the statistics outputs
template <typename T>
static bool flat8_eligible(const ConvArgs& a) {
    if constexpr (sizeof(T) != 2 || is_x3<T>::value) return false;
    return a.dil == 4 && a.c1 == 0 && a.cin % 32 == 0 && !a.stats_part && !a.tile_stats && !a.in_tile_stats && !a.in_scale && !a.rgb_w &&
           !a.transposed && a.phases == 1;
}
