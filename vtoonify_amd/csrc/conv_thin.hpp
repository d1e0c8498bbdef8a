// Thin convolutions: 1x1 / 3x3, stride 1, cout <= 3, planar fp32 output -- the ToRGB convs below the fused levels,
// the Fusion mask conv (2C -> 1, vtoonify.py:111,126) and fusion_skip (C+3 -> 3, vtoonify.py:197-198,262) and the
// encoder's RGB head (vtoonify.py:176).  Included by conv_igemm.hip inside its anonymous namespace.
//
// Per frame these are 11 convs of 1-10 MFLOP over 2-33 MB of activations.  As GEMMs they have N = 1..3: the tile
// kernels ran them with 16 output columns, cut K into 4-16 slices to find some parallelism and paid a second
// (reduce) launch -- 10-18 us a conv, 160 us a frame.  Here the contraction is turned around ("scatter form"):
//
//     d[p][tap][co] = sum_c W[co][tap][c] * x[p][c]            one GEMM over the INPUT pixels p of the patch,
//                                                               N = taps * cout <= 27 virtual channels (2 MFMA
//                                                               fragments), every activation read once from
//                                                               global memory straight into MFMA fragments
//     out[q][co]    = sum_tap d[q + tap - 1][tap][co]           9-term stencil over the small d tile in LDS
//
// Zero padding is "pixels outside the image contribute d = 0".  A workgroup = one 8x8-pixel output tile (10x10
// patch = 7 fragments of 16 pixels; 8x8 = 4 fragments for 1x1), 4 wavefronts, each taking every 4th K-step of
// 4*VEC channels for all pixel fragments; the four partial d tiles meet in LDS and are summed in wave order
// (deterministic).  No split-K workspace, no second launch.
#pragma once

constexpr int TH_TW = 8;            // output tile edge
constexpr int TH_NW = 4;            // wavefronts = K-slices
constexpr int TH_MAXF = 7;          // pixel fragments of a 10x10 patch
constexpr int TH_COLS = 32;         // most virtual channels (taps * cout, padded): two 16-row weight fragments
constexpr int TH_PADC = 4;          // floats of padding per d row (LDS banks)

template <typename T>
static bool thin_eligible(const ConvArgs& a) {
    constexpr int KSTEP = 4 * (16 / (int)sizeof(T));
    if (a.force_generic || a.transposed || a.rgb_w || a.stats_part || a.tile_stats || a.in_tile_stats || a.up_fir || a.slope_vec)
        return false;
    if (a.in_scale && !a.in_shift) return false;
    if ((a.in_scale || a.in_absdiff) && !(a.taps == 9 && a.taps * a.coutT <= 16)) return false;   // the compiled prologue form
    if (a.out_layout != VT_OUT_NCHW || a.phases != 1 || a.stride != 1 || a.dil != 1) return false;
    if (!((a.taps == 9 && a.kw == 3 && a.pad == 1) || (a.taps == 1 && a.pad == 0))) return false;
    if (a.coutT < 1 || a.coutT > 3 || a.taps * a.coutT > TH_COLS) return false;
    // one source, or the Fusion gate's cat[x, |x - other|] (vt_conv_desc.in_absdiff: src1 = other, c1 = c0)
    if (a.in_absdiff ? (a.c1 != a.c0 || !a.src1 || a.ld1 % (16 / (int)sizeof(T)) != 0 || (uintptr_t)a.src1 % 16 != 0) : a.c1 != 0)
        return false;
    if (a.c0 % KSTEP != 0 || a.ld0 % (16 / (int)sizeof(T)) != 0) return false;
    if (a.Ho != a.H || a.Wo != a.W) return false;
    if ((uintptr_t)a.src0 % 16 != 0 || (uintptr_t)a.wgt % 16 != 0) return false;
    return (int64_t)a.N * a.H * a.W * a.ld0 * (int64_t)sizeof(T) < ((int64_t)1 << 40);
}

// PRO: input prologue in the loader (the Fusion gate, model/vtoonify.py:125-126): the K range is cat[x, |x - other|]
// (p.in_absdiff) and / or every in-image value goes through x' = x * in_scale[n][c] + in_shift[n][c], rounded to T like a
// stored tensor -- the vt_affine_apply launch and its 2C-channel normalised copy (268 MB per 4-frame step at the 256^2 level)
// fold into the conv that reads them.  Bit-identical to the two launches (same operations in the same order).
// Workgroup -> tile: consecutive workgroups go round the 8 XCDs, so tile = blockIdx gave horizontally adjacent tiles -- which share
// two of their 10 (18) patch columns -- to different L2s.  Here every XCD takes one contiguous run of tiles in row-major order (8
// rows of 16 tiles at the 256^2 level): the halos of a run meet in its L2.  Fabric bytes per 4-frame launch at that level
// (rocprofv3 --pmc, calls r06s -> r06ab): mask conv 258 -> 220 MB for 134 MB of input, fusion_skip 121 -> 107 MB; the time did
// not move (47 / 28 us): these kernels are not bound by the fabric.  (Neither by their branches: the same loop with range-checked
// buffer loads instead of `cond ? load : zero` -- one basic block -- measured 47 / 39 us, call r06ac, and was dropped.)
__device__ __forceinline__ int thin_tile_of_block() {
    const int nb = (int)gridDim.x, b = (int)blockIdx.x;
    const int q = nb >> 3, r = nb & 7;
    const int xcd = b & 7, idx = b >> 3;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
}

template <typename T, int KS, int NB, bool PRO = false>   // KS = 3 (3x3, pad 1) or 1 (1x1); NB = weight fragments (16 virtual channels each)
__global__ void __launch_bounds__(TH_NW * 64) conv_thin_kernel(const ConvArgs p) {
    constexpr int VEC = 16 / (int)sizeof(T);
    constexpr int KSTEP = 4 * VEC;                   // channels per MFMA K-step (lane group q owns VEC of them)
    constexpr int PW = TH_TW + KS - 1;               // patch edge: 10 or 8
    constexpr int NPIX = PW * PW;
    constexpr int NF = (NPIX + 15) / 16;             // pixel fragments: 7 or 4
    constexpr int TAPS = KS * KS;
    constexpr int UNR = KS == 3 ? 3 : 4;             // K-steps in flight per wave (NF * UNR 16-byte loads per lane)
    __shared__ __attribute__((aligned(16))) float dpart[TH_NW][NF * 16][16 * NB + TH_PADC];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = vt_uniform(tid >> 6) & (TH_NW - 1);
    const int q = lane >> 4, l15 = lane & 15;
    const int tiles_x = (p.W + TH_TW - 1) / TH_TW, tiles_y = (p.H + TH_TW - 1) / TH_TW;
    const int tile = thin_tile_of_block();
    const int img = tile / (tiles_x * tiles_y);
    const int trem = tile - img * (tiles_x * tiles_y);
    const int y0 = (trem / tiles_x) * TH_TW, x0 = (trem % tiles_x) * TH_TW;
    const int ncols = TAPS * p.coutT;

    // this lane's pixel of every fragment (patch pixel f*16 + l15) -> element offset of its channel group, or -1
    const T* src = (const T*)p.src0;
    const T* oth = (const T*)p.src1;
    int64_t poff[NF], qoff[PRO ? NF : 1];
#pragma unroll
    for (int f = 0; f < NF; ++f) {
        const int pp = f * 16 + l15;
        const int py = pp / PW, px = pp - py * PW;
        const int iy = y0 + py - (KS / 2), ix = x0 + px - (KS / 2);
        const bool in = pp < NPIX && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
        poff[f] = in ? ((int64_t)(img * p.H + iy) * p.W + ix) * p.ld0 + q * VEC : -1;
        if (PRO) qoff[f] = in ? ((int64_t)(img * p.H + iy) * p.W + ix) * p.ld1 + q * VEC : -1;
    }
    // this lane's weight row of both fragments: virtual channel v = b*16 + l15 = tap * cout + co
    const T* wg = (const T*)p.wgt;
    int64_t woff[NB];
#pragma unroll
    for (int b = 0; b < NB; ++b) {
        const int v = b * 16 + l15;
        const int tap = v / p.coutT, co = v - tap * p.coutT;
        woff[b] = v < ncols ? ((int64_t)co * TAPS + tap) * p.cin + q * VEC : -1;
    }

    f32x4 acc[NF][NB];
#pragma unroll
    for (int f = 0; f < NF; ++f)
#pragma unroll
        for (int b = 0; b < NB; ++b) acc[f][b] = f32x4{0.f, 0.f, 0.f, 0.f};

    const u128 zero = u128{0u, 0u, 0u, 0u};
    const int nk = p.cin / KSTEP;
    if constexpr (!PRO) {
        for (int k0 = wave; k0 < nk; k0 += TH_NW * UNR) {
            u128 fa[UNR][NF], fw[UNR][NB];
#pragma unroll
            for (int u = 0; u < UNR; ++u) {
                const int ks = k0 + u * TH_NW;
                const bool live = ks < nk;
                const int kb = (live ? ks : k0) * KSTEP;
                // (predicated loads are right HERE: hipcc batches them -- all NF * UNR are issued before the first wait -- and
                // dead lanes fetch nothing; the unconditional-at-a-clamped-offset form measured 124 -> 145 us over the 7 launches)
#pragma unroll
                for (int f = 0; f < NF; ++f) fa[u][f] = (live && poff[f] >= 0) ? ld128(src + poff[f] + kb) : zero;
#pragma unroll
                for (int b = 0; b < NB; ++b) fw[u][b] = (live && woff[b] >= 0) ? ld128(wg + woff[b] + kb) : zero;
            }
#pragma unroll
            for (int u = 0; u < UNR; ++u) {
#pragma unroll
                for (int f = 0; f < NF; ++f)
#pragma unroll
                    for (int b = 0; b < NB; ++b) Mma<T>::run(acc[f][b], fw[u][b], fa[u][f]);
            }
        }
    } else {
        // Prologue form.  A wave still takes K-steps wave, wave + 4, ... in increasing order (the order of the plain form: same
        // bits), but in two loops: the x half of the K range with 2 steps in flight, the |x - other| half -- two operands per
        // fragment -- with one: the registers of the larger loop (~170), not of both at once (330 as one loop: one wave per
        // SIMD and 110 us at the 256^2 level).
        const int nk0 = p.in_absdiff ? p.c0 / KSTEP : nk;   // K-steps of the first half
        auto affine = [&](u128& frag, bool ok, const float* sc, const float* sh, const u128* og) {
            if (!ok) return;                                  // padding stays zero: it pads the NORMALISED tensor
            float v[VEC];
            unpack16<T>(frag, v);
            if (og) {
                float g[VEC];
                unpack16<T>(*og, g);
#pragma unroll
                for (int i = 0; i < VEC; ++i) v[i] = fabsf(v[i] - g[i]);
            }
            if (p.in_scale) {                                 // vt_affine_apply's arithmetic: mul, then add, then round
#pragma unroll
                for (int i = 0; i < VEC; ++i) v[i] = v[i] * sc[i] + sh[i];
            }
            frag = pack16<T>(v);
        };
        auto table = [&](int ks, float* sc, float* sh) {      // this lane's VEC channels of the K-step: one row serves all fragments
            if (!p.in_scale) return;
            const int so = img * p.cin + ks * KSTEP + q * VEC;
#pragma unroll
            for (int i = 0; i < VEC; i += 4) {
                unpack16<float>(ld128(p.in_scale + so + i), sc + i);
                unpack16<float>(ld128(p.in_shift + so + i), sh + i);
            }
        };
        constexpr int U1 = 2;
        int k0 = wave;
        for (; k0 < nk0; k0 += TH_NW * U1) {
            u128 fa[U1][NF], fw[U1][NB];
            float scv[U1][VEC], shv[U1][VEC];
#pragma unroll
            for (int u = 0; u < U1; ++u) {
                const int ks = k0 + u * TH_NW;
                const bool live = ks < nk0;
                const int kb = (live ? ks : k0) * KSTEP;
#pragma unroll
                for (int f = 0; f < NF; ++f) fa[u][f] = (live && poff[f] >= 0) ? ld128(src + poff[f] + kb) : zero;
                table(live ? ks : k0, scv[u], shv[u]);
#pragma unroll
                for (int b = 0; b < NB; ++b) fw[u][b] = (live && woff[b] >= 0) ? ld128(wg + woff[b] + kb) : zero;
            }
#pragma unroll
            for (int u = 0; u < U1; ++u) {
                const bool live = k0 + u * TH_NW < nk0;
#pragma unroll
                for (int f = 0; f < NF; ++f) affine(fa[u][f], live && poff[f] >= 0, scv[u], shv[u], nullptr);
#pragma unroll
                for (int f = 0; f < NF; ++f)
#pragma unroll
                    for (int b = 0; b < NB; ++b) Mma<T>::run(acc[f][b], fw[u][b], fa[u][f]);
            }
        }
        // this wave's first K-step of the second half: the smallest ks >= nk0 with ks = wave (mod 4)
        for (int ks = nk0 + ((wave - nk0) % TH_NW + TH_NW) % TH_NW; ks < nk; ks += TH_NW) {
            u128 fa[NF], fo[NF], fw[NB];
            float scv[VEC], shv[VEC];
            const int kb = ks * KSTEP, kc = kb - p.c0;
#pragma unroll
            for (int f = 0; f < NF; ++f) {
                fa[f] = poff[f] >= 0 ? ld128(src + poff[f] + kc) : zero;
                fo[f] = qoff[f] >= 0 ? ld128(oth + qoff[f] + kc) : zero;
            }
            table(ks, scv, shv);
#pragma unroll
            for (int b = 0; b < NB; ++b) fw[b] = woff[b] >= 0 ? ld128(wg + woff[b] + kb) : zero;
#pragma unroll
            for (int f = 0; f < NF; ++f) affine(fa[f], poff[f] >= 0, scv, shv, &fo[f]);
#pragma unroll
            for (int f = 0; f < NF; ++f)
#pragma unroll
                for (int b = 0; b < NB; ++b) Mma<T>::run(acc[f][b], fw[b], fa[f]);
        }
    }
    // partial d tile of this wave: pixel f*16 + l15, virtual channels b*16 + 4q .. +3
#pragma unroll
    for (int f = 0; f < NF; ++f)
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            float v4[4] = {acc[f][b][0], acc[f][b][1], acc[f][b][2], acc[f][b][3]};
            st128(&dpart[wave][f * 16 + l15][b * 16 + q * 4], pack16<float>(v4));
        }
    __syncthreads();

    // stencil + epilogue: thread = (output pixel, output channel)
    const int opix = tid & 63, co = tid >> 6;
    const int oy = opix / TH_TW, ox = opix - oy * TH_TW;
    const int gy = y0 + oy, gx = x0 + ox;
    if (co >= p.coutT || gy >= p.H || gx >= p.W) return;
    float s = 0.0f;
#pragma unroll
    for (int tap = 0; tap < TAPS; ++tap) {
        const int ky = tap / KS, kx = tap - ky * KS;
        const int pp = (oy + ky) * PW + ox + kx;        // patch pixel under this tap
        float t = 0.0f;
#pragma unroll
        for (int w = 0; w < TH_NW; ++w) t += dpart[w][pp][tap * p.coutT + co];   // K-slices in wave order
        s += t;
    }
    const float ga = p.gain_alpha * (p.alpha_dev ? p.alpha_dev[0] : 1.0f);
    const float v = conv_finish(p, s, p.bias ? p.bias[co] : 0.0f, ga, p.slope);
    const int64_t HoWo = (int64_t)p.H * p.W;
    const int64_t off = ((int64_t)img * p.cout + co) * HoWo + (int64_t)gy * p.W + gx;
    const float* rs = (const float*)p.resid;
    ((float*)p.out)[off] = post_act(p, v + (rs ? p.beta * rs[off] : 0.0f));
}

// ---------------------------------------------------------------------------------------
// The same contraction on 16 x 16-pixel tiles with the PIXELS split over the wavefronts (round 6): the thin convs of the two
// large fusion levels (128^2, 256^2: mask conv 2C -> 1 with the gate prologue, fusion_skip C + 64 -> 3).
//
// With several steps in flight the frame rate is set by what every kernel takes of the GPU, not by its latency
// (profiles/r06_lanes.txt), and at those levels the 8 x 8 tiles above take the whole GPU for 63 + 41 + 29 + 15 us per step at
// 2.0-2.5 TB/s: every tile reads a 10 x 10 patch for 8 x 8 outputs (1.56 x), each wave walks its quarter of the K steps with one
// or two loads in flight and the four partial tiles meet in 36 KB of LDS.  Here a workgroup owns 16 x 16 outputs (18 x 18 patch:
// 1.27 x), a wave owns every fourth pixel fragment for ALL K steps (no partial tiles: the d tile is written once), and the loads of
// two K steps are in flight per wave.  K order: 0 .. nk-1 in one accumulator chain per output -- not the order of the 8 x 8 kernel
// (K steps interleaved over four waves), so the choice between the two is by the per-image geometry only (launch_thin).
// ---------------------------------------------------------------------------------------
constexpr int TH16 = 16;
constexpr int TH16_NW = 4;           // wavefronts of the 16 x 16 form: 6 pixel fragments each (21 of an 18 x 18 patch).  (8 waves of 3
                                    // fragments with two K steps in flight in the gate half measured 50 / 31 us against 47 / 27.)

template <typename T, int KS, int NB, bool PRO = false>
__global__ void __launch_bounds__(TH16_NW * 64) conv_thin16_kernel(const ConvArgs p) {
    constexpr int VEC = 16 / (int)sizeof(T);
    constexpr int KSTEP = 4 * VEC;
    constexpr int PW = TH16 + KS - 1;                // 18 or 16
    constexpr int NPIX = PW * PW;
    constexpr int NF = (NPIX + 15) / 16;             // 21 or 16 pixel fragments
    constexpr int FPW = (NF + TH16_NW - 1) / TH16_NW;   // fragments per wave: 6 or 4
    constexpr int TAPS = KS * KS;
    constexpr int UNR = 2;                           // K steps in flight per wave
    __shared__ __attribute__((aligned(16))) float dt[NF * 16][16 * NB + TH_PADC];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = vt_uniform(tid >> 6) & (TH16_NW - 1);
    const int q = lane >> 4, l15 = lane & 15;
    const int tiles_x = (p.W + TH16 - 1) / TH16, tiles_y = (p.H + TH16 - 1) / TH16;
    const int tile = thin_tile_of_block();
    const int img = tile / (tiles_x * tiles_y);
    const int trem = tile - img * (tiles_x * tiles_y);
    const int y0 = (trem / tiles_x) * TH16, x0 = (trem % tiles_x) * TH16;
    const int ncols = TAPS * p.coutT;

    const T* src = (const T*)p.src0;
    const T* oth = (const T*)p.src1;
    int64_t poff[FPW], qoff[PRO ? FPW : 1];
#pragma unroll
    for (int i = 0; i < FPW; ++i) {
        const int f = wave + TH16_NW * i;            // this wave's i-th fragment
        const int pp = f * 16 + l15;
        const int py = pp / PW, px = pp - py * PW;
        const int iy = y0 + py - (KS / 2), ix = x0 + px - (KS / 2);
        const bool in = f < NF && pp < NPIX && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
        poff[i] = in ? ((int64_t)(img * p.H + iy) * p.W + ix) * p.ld0 + q * VEC : -1;
        if (PRO) qoff[i] = in ? ((int64_t)(img * p.H + iy) * p.W + ix) * p.ld1 + q * VEC : -1;
    }
    const T* wg = (const T*)p.wgt;
    int64_t woff[NB];
#pragma unroll
    for (int b = 0; b < NB; ++b) {
        const int v = b * 16 + l15;
        const int tap = v / p.coutT, co = v - tap * p.coutT;
        woff[b] = v < ncols ? ((int64_t)co * TAPS + tap) * p.cin + q * VEC : -1;
    }
    f32x4 acc[FPW][NB];
#pragma unroll
    for (int i = 0; i < FPW; ++i)
#pragma unroll
        for (int b = 0; b < NB; ++b) acc[i][b] = f32x4{0.f, 0.f, 0.f, 0.f};

    const u128 zero = u128{0u, 0u, 0u, 0u};
    const int nk = p.cin / KSTEP;
    auto affine = [&](u128& frag, bool ok, const float* sc, const float* sh, const u128* og) {   // (the gate prologue of the 8 x 8 kernel)
        if (!ok) return;
        float v[VEC];
        unpack16<T>(frag, v);
        if (og) {
            float g[VEC];
            unpack16<T>(*og, g);
#pragma unroll
            for (int i = 0; i < VEC; ++i) v[i] = fabsf(v[i] - g[i]);
        }
        if (p.in_scale) {
#pragma unroll
            for (int i = 0; i < VEC; ++i) v[i] = v[i] * sc[i] + sh[i];
        }
        frag = pack16<T>(v);
    };
    auto table = [&](int ks, float* sc, float* sh) {
        if (!p.in_scale) return;
        const int so = img * p.cin + ks * KSTEP + q * VEC;
#pragma unroll
        for (int i = 0; i < VEC; i += 4) {
            unpack16<float>(ld128(p.in_scale + so + i), sc + i);
            unpack16<float>(ld128(p.in_shift + so + i), sh + i);
        }
    };
    const int nk0 = (PRO && p.in_absdiff) ? p.c0 / KSTEP : nk;   // K steps that read src0 as it is
    // ---- K steps [0, nk0): one source ----
    for (int k0 = 0; k0 < nk0; k0 += UNR) {
        u128 fa[UNR][FPW], fw[UNR][NB];
        float scv[UNR][PRO ? VEC : 1], shv[UNR][PRO ? VEC : 1];
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            const bool live = k0 + u < nk0;
            const int kb = (live ? k0 + u : k0) * KSTEP;
#pragma unroll
            for (int i = 0; i < FPW; ++i) fa[u][i] = (live && poff[i] >= 0) ? ld128(src + poff[i] + kb) : zero;
            if (PRO) table(live ? k0 + u : k0, scv[u], shv[u]);
#pragma unroll
            for (int b = 0; b < NB; ++b) fw[u][b] = (live && woff[b] >= 0) ? ld128(wg + woff[b] + kb) : zero;
        }
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            if (PRO) {
                const bool live = k0 + u < nk0;
#pragma unroll
                for (int i = 0; i < FPW; ++i) affine(fa[u][i], live && poff[i] >= 0, scv[u], shv[u], nullptr);
            }
#pragma unroll
            for (int i = 0; i < FPW; ++i)
#pragma unroll
                for (int b = 0; b < NB; ++b) Mma<T>::run(acc[i][b], fw[u][b], fa[u][i]);
        }
    }
    // ---- K steps [nk0, nk): |src0 - src1| (the Fusion gate's second half) ----
    if constexpr (PRO) {
        constexpr int UNR2 = FPW <= 3 ? 2 : 1;   // two operands per fragment: one K step in flight at 6 fragments per wave (registers)
        for (int k0 = nk0; k0 < nk; k0 += UNR2) {
            u128 fa[UNR2][FPW], fo[UNR2][FPW], fw[UNR2][NB];
            float scv[UNR2][VEC], shv[UNR2][VEC];
#pragma unroll
            for (int u = 0; u < UNR2; ++u) {
                const bool live = k0 + u < nk;
                const int ks = live ? k0 + u : k0;
                const int kb = ks * KSTEP, kc = kb - p.c0;
#pragma unroll
                for (int i = 0; i < FPW; ++i) {
                    fa[u][i] = (live && poff[i] >= 0) ? ld128(src + poff[i] + kc) : zero;
                    fo[u][i] = (live && qoff[i] >= 0) ? ld128(oth + qoff[i] + kc) : zero;
                }
                table(ks, scv[u], shv[u]);
#pragma unroll
                for (int b = 0; b < NB; ++b) fw[u][b] = (live && woff[b] >= 0) ? ld128(wg + woff[b] + kb) : zero;
            }
#pragma unroll
            for (int u = 0; u < UNR2; ++u) {
                const bool live = k0 + u < nk;
#pragma unroll
                for (int i = 0; i < FPW; ++i) affine(fa[u][i], live && poff[i] >= 0, scv[u], shv[u], &fo[u][i]);
#pragma unroll
                for (int i = 0; i < FPW; ++i)
#pragma unroll
                    for (int b = 0; b < NB; ++b) Mma<T>::run(acc[i][b], fw[u][b], fa[u][i]);
            }
        }
    }
    // the d tile: pixel f*16 + l15, virtual channels b*16 + 4q .. +3 -- written once, by the wave that owns the fragment
#pragma unroll
    for (int i = 0; i < FPW; ++i) {
        const int f = wave + TH16_NW * i;
        if (f < NF) {
#pragma unroll
            for (int b = 0; b < NB; ++b) {
                float v4[4] = {acc[i][b][0], acc[i][b][1], acc[i][b][2], acc[i][b][3]};
                st128(&dt[f * 16 + l15][b * 16 + q * 4], pack16<float>(v4));
            }
        }
    }
    __syncthreads();
    // stencil + epilogue: thread = output pixel, all of its (<= 3) channels
    if (tid >= TH16 * TH16) return;
    const int oy = tid / TH16, ox = tid - oy * TH16;
    const int gy = y0 + oy, gx = x0 + ox;
    if (gy >= p.H || gx >= p.W) return;
    const float ga = p.gain_alpha * (p.alpha_dev ? p.alpha_dev[0] : 1.0f);
    const int64_t HoWo = (int64_t)p.H * p.W;
    const float* rs = (const float*)p.resid;
    for (int co = 0; co < p.coutT; ++co) {
        float s = 0.0f;
#pragma unroll
        for (int tap = 0; tap < TAPS; ++tap) {
            const int ky = tap / KS, kx = tap - ky * KS;
            s += dt[(oy + ky) * PW + ox + kx][tap * p.coutT + co];
        }
        const float v = conv_finish(p, s, p.bias ? p.bias[co] : 0.0f, ga, p.slope);
        const int64_t off = ((int64_t)img * p.cout + co) * HoWo + (int64_t)gy * p.W + gx;
        ((float*)p.out)[off] = post_act(p, v + (rs ? p.beta * rs[off] : 0.0f));
    }
}

// 16 x 16 tiles from 256 tiles per IMAGE up (the 256^2 level of a 256^2 frame; at 128^2 the two forms measured the same, 29 / 15 us
// against 31 / 16): by geometry only -- the two kernels sum K in different orders -- and only for the 3 x 3 forms
static bool thin16_wanted(const ConvArgs& a) {
    return a.taps == 9 && (int64_t)vt_cdiv(a.H, TH16) * vt_cdiv(a.W, TH16) >= 256;
}

template <typename T>
int launch_thin(const ConvArgs& a, vt_stream stream) {
    ConvArgs args = a;
    args.splitk = 1;
    args.kps = 0;
    const int64_t blocks = (int64_t)a.N * vt_cdiv(a.H, TH_TW) * vt_cdiv(a.W, TH_TW);
    if (blocks >= ((int64_t)1 << 31)) {
        vt_set_error("vt_conv2d: too many tiles");
        return VT_ERR_ARG;
    }
    const bool two = a.taps * a.coutT > 16;
    if ((a.in_scale || a.in_absdiff) && (a.taps != 9 || two)) {   // the loader prologue: only the Fusion gate's shape is compiled
        vt_set_error("vt_conv2d: in_scale / in_absdiff on a thin conv need a 3x3 kernel with 9 * cout <= 16");
        return VT_ERR_UNSUPPORTED;
    }
    if (thin16_wanted(a)) {
        const unsigned b16 = (unsigned)((int64_t)a.N * vt_cdiv(a.H, TH16) * vt_cdiv(a.W, TH16));
        if (a.in_scale || a.in_absdiff) {
            auto k = conv_thin16_kernel<T, 3, 1, true>;
            VT_LAUNCH(k, dim3(b16), dim3(TH16_NW * 64), stream, args);
        } else if (two) {
            auto k = conv_thin16_kernel<T, 3, 2>;
            VT_LAUNCH(k, dim3(b16), dim3(TH16_NW * 64), stream, args);
        } else {
            auto k = conv_thin16_kernel<T, 3, 1>;
            VT_LAUNCH(k, dim3(b16), dim3(TH16_NW * 64), stream, args);
        }
        return vt_check_launch("vt_conv2d(thin, 16x16)");
    }
    if (a.in_scale || a.in_absdiff) {
        auto k = conv_thin_kernel<T, 3, 1, true>;
        VT_LAUNCH(k, dim3((unsigned)blocks), dim3(TH_NW * 64), stream, args);
    } else if (a.taps == 9 && two) {
        auto k = conv_thin_kernel<T, 3, 2>;
        VT_LAUNCH(k, dim3((unsigned)blocks), dim3(TH_NW * 64), stream, args);
    } else if (a.taps == 9) {
        auto k = conv_thin_kernel<T, 3, 1>;
        VT_LAUNCH(k, dim3((unsigned)blocks), dim3(TH_NW * 64), stream, args);
    } else {
        auto k = conv_thin_kernel<T, 1, 1>;
        VT_LAUNCH(k, dim3((unsigned)blocks), dim3(TH_NW * 64), stream, args);
    }
    return vt_check_launch("vt_conv2d(thin)");
}
