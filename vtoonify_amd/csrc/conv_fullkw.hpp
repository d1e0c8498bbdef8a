// Weight-stationary whole-K 3x3 convolution: conv_fullk.hpp's kernel for a BATCH of frames (round 3).
// Included by conv_igemm.hip after conv_fullk.hpp (shares its constants, FullkArgs-style helpers, ConvArgs,
// Mma<T>, conv_finish, store_out4, decode_block).
//
// Why: the whole-K kernel streams the 295 KB weight slice of its 32 output channels from L2 once per 8x8-pixel
// tile, and that stream -- not MFMA, not HBM -- is what bounds it (profiles/r02_fullk_stream_scaling.txt: ~70 GB/s
// of operand ingest per CU; removing every MFMA changes nothing).  At batch 4 the same conv therefore cost exactly
// 4x batch 1 (39.9 vs 10.5 us, profiles/r03a_convbench_trunk_baseline.txt): four frames streamed the same weights
// four times.  The reference runs the video loop at --batch_size 4 (style_transfer.py:35,176) through these layers
// (model/vtoonify.py:92-104,235-239; model/dualstylegan.py:38-45).
//
// Here the WEIGHTS ARE THE RESIDENT OPERAND.  The biggest on-chip memory of a CU is its register file (512 KB);
// the 295 KB slice fits it exactly as it is already split: wavefront w owns input channels [64w, 64w+64) and
// keeps the 18 sub-steps x 2 fragments x 4 registers = 144 VGPRs of weights for the life of the workgroup (one
// L2 -> VGPR stream per workgroup, loads hidden from the compiler like in conv_fullk.hpp).  The PIXELS stream:
// a workgroup walks G consecutive 8x8 tiles of one image as 2G half-tile steps (4 rows x 8 pixels = two MFMA
// pixel fragments), each step's 6x10-pixel patch chunk arriving by LDS-DMA in a wave-private double buffer while
// the previous step computes.  With G = 4 a 4-frame trunk conv is still 256 workgroups, but each amortises its
// weight stream over 256 pixels instead of 64: operand ingest per output drops 2.2x.
// Measured (profiles/r03_fullkw_ablation.txt): 512 -> 512 @ 32x32, batch 4: 39.9 -> 29.3 us.  What bounds it now is not
// the ingest but the step structure -- MFMA + fragment reads, patch LDS-DMA, sum + epilogue and barriers run as four
// serial phases of lock-step waves, ~3.4 us per step against ~1.4 of LDS-DMA alone.
//
// Per step (s):   wait patch s  ->  [AdaIN rewrite of the patch in LDS]  ->  9 sub-steps of MFMA
//                 ->  cross-wave sum + epilogue (+ tile statistics) of step s-1  ->  9 sub-steps of MFMA
//                 ->  issue patch s+2 into the buffer just consumed  ->  barrier  ->  park the partial tile
//                 ->  barrier.
// The K split across the 8 wavefronts, the order of the 18 sub-steps, the wave-order sum, the epilogue and the
// two-pass tile statistics are those of conv_fullk_kernel, value for value: a frame convolved here is BIT-IDENTICAL
// to the same frame on conv_fullk_kernel for every G (tests/test_ops.py::test_conv_weight_stationary_equals_whole_k),
// so the choice between the two may depend on the batch without breaking "batched == frame by frame".
//
// LDS (all 160 KB): per wave 2 patch slots of 60 rows x 128 B (the 8th wave-load of a slot is aimed at rows 52..59
// and re-fetches four rows instead of spilling past the slot; any 16-byte LDS base is legal for buffer_load ... lds,
// profiles/r03_glds_probe2.txt) + 512 B of AdaIN scale / shift; 8 x 4 KB for the partial tiles of one step (the 2 KB
// statistics exchange overlays them); 2 x 2 KB for the residual values of the step ahead (4-byte LDS-DMA).
#pragma once

struct FullkwArgs {
    uint32_t nrec0, nrec1;    // byte sizes of the two sources (buffer range check = zero padding)
    const void* wstream;      // vt_conv_weight_stream image of the weights
    int tiles_y, tiles_x;     // 8x8-pixel tiles per dilation phase
    int group;                // G: consecutive tiles of one image per workgroup
    int groups_per_img;       // ceil(d*d*tiles_y*tiles_x / G)
    int xm, xn;               // XCD grid (xm * xn = 8): XCD (i, j) owns the i-th 1/xm of the pixel groups and the j-th
                              // 1/xn of the channel tiles; 0 = decode_block's order
};

constexpr int FW_HR = 4;                                 // output rows per step (half an 8x8 tile)
constexpr int FW_PH = FW_HR + 2;                         // patch rows
constexpr int FW_PROWS = FW_PH * FK_PW;                  // 60 pixel rows of 128 B
constexpr int FW_SLOT = FW_PROWS * 128;                  // 7680 B
constexpr int FW_WAVE = 2 * FW_SLOT + 512;               // two slots + [scale 64 | shift 64] floats
constexpr int FW_RED = FK_NW * FW_WAVE;                  // partial tiles: 8 waves x 32 pixels x 128 B
constexpr int FW_RES = FW_RED + FK_NW * 4096;            // residual values of the step ahead: 2 parities x 4 waves x 512 B
constexpr int FW_LDS = FW_RES + 2 * 4 * 512;
static_assert(FW_LDS <= 163840, "LDS of one CU");        // (the statistics exchange, 2 KB, overlays the partial tiles)

// vt_vmcnt_fence<2 * k + 8>() for a value k that is a compile-time constant after unrolling (0 <= k <= MAXK)
template <int MAXK>
__device__ __forceinline__ void fkw_wait_pairs(int k) {
    if (k >= MAXK) vt_vmcnt_fence<2 * MAXK + 8>();
    else fkw_wait_pairs<MAXK - 1>(k);
}
template <>
__device__ __forceinline__ void fkw_wait_pairs<0>(int) { vt_vmcnt_fence<8>(); }

// the MFMA sub-steps [LO, HI) of one step over the wave's patch slot; FIRST = the step whose weights are still
// landing (hand-counted waits: after sub-step st's pair come the pairs of the 17 - st later sub-steps and the 8
// LDS-DMA loads of patch 1)
template <typename T, bool FIRST, int LO, int HI, int ABL = 0>
__device__ __forceinline__ void fkw_mma(f32x4 (&acc)[2][2], const u128 (&wr)[18][2], const unsigned char* slot,
                                        const uint32_t (&abase)[3][2]) {
    u128 fa[2][2];
    auto read_a = [&](u128 (&f)[2], int st) {
        const int tap = st >> 1, sub = st & 1;
        const int ky = tap / 3, kx = tap - ky * 3;
#pragma unroll
        for (int a = 0; a < 2; ++a) f[a] = ld128(slot + abase[kx][sub] + ((2 * a + ky) * FK_PW + kx) * 128);
    };
    read_a(fa[LO & 1], LO);
    if constexpr (ABL == 35) read_a(fa[(LO + 1) & 1], LO + 1);   // ablation: MFMAs on two resident fragment pairs, no further LDS reads
    // The interleave is PINNED (sched_group_barrier): the two fragment reads of sub-step st+1, then the four MFMAs of
    // sub-step st.  Left alone, hipcc (241 registers in use) re-used ONE fragment register quad and emitted
    // ds_read -> lgkmcnt(0) -> 2 MFMAs, 36 times per step: every MFMA pair behind a fresh LDS round trip.
    constexpr bool PIN = !FIRST && ABL == 0;
    if constexpr (PIN) vt_sched_group<0x100, 2>();
#pragma unroll
    for (int st = LO; st < HI; ++st) {
        if constexpr (FIRST) fkw_wait_pairs<17>(17 - st);
        if (st + 1 < HI && ABL != 35) read_a(fa[(st + 1) & 1], st + 1);
#pragma unroll
        for (int a = 0; a < 2; ++a) {
            if constexpr (ABL == 36) {   // ablation: LDS reads only
                acc[a][0][0] += vt_u2f(fa[st & 1][a].x ^ wr[st][0].x);
                continue;
            }
            Mma<T>::run(acc[a][0], wr[st][0], fa[st & 1][a]);
            Mma<T>::run(acc[a][1], wr[st][1], fa[st & 1][a]);
        }
        if constexpr (PIN) {
            if (st + 1 < HI) vt_sched_group<0x100, 2>();
            vt_sched_group<0x008, 4>();
        }
    }
}

// SAFE = 1: every counted wait becomes vmcnt(0) (GPU bisection aid; same results, no overlap)
// ABL: ablations of tools/conv_bench.py (VT_FULLKW_ABLATE; compile-time so that the code shape of the hot loop is the
// product's: 31 no MFMA / fragment reads, 32 no patch loads after the first two, 33 no sum / epilogue / store / park,
// 35 MFMAs on resident fragments (no further LDS reads), 36 fragment reads only, 37 no barriers.  Results are wrong.
template <typename T, int SAFE, int ABL = 0>
__global__ void __launch_bounds__(FK_NW * 64)
conv_fullkw_kernel(const ConvArgs p, const FullkwArgs g) {
    constexpr int ESZ = (int)sizeof(T);
    constexpr int VEC = 16 / ESZ;
    constexpr int BK = 8 * VEC;                 // channels per 128-byte row
    constexpr int NSUB = 18;
    __shared__ __attribute__((aligned(1024))) unsigned char smem[FW_LDS];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = vt_uniform(tid >> 6) & (FK_NW - 1);
    const int q = lane >> 4, l15 = lane & 15;
    const int hi = l15 >> 3, lo = l15 & 7;
    int tile_m, tile_n, split;
    if (g.xm > 0) {
        // XCD-aware placement for a BATCH.  Workgroup b runs on XCD b % 8, each with its own 4 MiB L2.  decode_block gives an
        // XCD two channel tiles and EVERY pixel group: at 4 frames that is 4 MB of activations streamed through each L2
        // twice with a reuse distance of the whole tensor -- every patch read misses (measured: 31 us per conv = 125 MB of
        // patch reads at the fabric's ~4 TB/s).  Here XCD (i, j) owns a sub-grid [pixel groups / xm] x [channel tiles / xn]
        // whose weights + activations fit its L2: every weight slice is fetched by xm L2s, every patch by xn.
        const int b = blockIdx.x;
        const int xcd = b & 7, idx = b >> 3;
        const int mper = p.tiles_m / g.xm;
        const int nl = idx / mper, ml = idx - nl * mper;
        tile_m = (xcd % g.xm) * mper + ml;
        tile_n = (xcd / g.xm) * (p.tiles_n / g.xn) + nl;
        split = 0;
    } else {
        decode_block(p, tile_m, tile_n, split);
    }
    const int d = p.dil;
    const int per_phase = g.tiles_y * g.tiles_x;
    const int per_img = per_phase * d * d;
    const int img = tile_m / g.groups_per_img;
    const int t_first = (tile_m - img * g.groups_per_img) * g.group;
    const int ntl = (per_img - t_first) < g.group ? (per_img - t_first) : g.group;
    const int nsteps = 2 * ntl;
    const int n0 = tile_n * FK_BN;

    unsigned char* my = smem + wave * FW_WAVE;
    float* tab = reinterpret_cast<float*>(my + 2 * FW_SLOT);   // [scale BK | shift BK]
    const BufRsrc r0 = vt_make_rsrc(p.src0, g.nrec0);
    const BufRsrc r1 = vt_make_rsrc(p.src1 ? p.src1 : p.src0, p.src1 ? g.nrec1 : 0u);
    const int kc = wave * BK;                   // this wave's input channels (single round: cin = 8 * BK)
    const bool s1 = kc >= p.c0;
    const uint32_t so = (uint32_t)((s1 ? kc - p.c0 : kc) * ESZ);
    const uint32_t ldb = (uint32_t)((s1 ? p.ld1 : p.ld0) * ESZ);

    // image position of pixel (0, 0) of tile t (tile order of conv_fullk_kernel: phase, tile row, tile column); scalar
    // arithmetic.  (Rolling the origins of the three tiles a step touches through scalar registers instead of
    // re-deriving them was measured: the extra live values spilled and the conv went from 31 to 41 us.)
    auto tile_origin = [&](int t, int& y0, int& x0) {
        const int ph = t / per_phase, rem = t - ph * per_phase;
        const int fy = ph / d, fx = ph - fy * d;
        const int ty0 = rem / g.tiles_x, tx0 = rem - ty0 * g.tiles_x;
        y0 = fy + ty0 * FK_TH * d;
        x0 = fx + tx0 * FK_TW * d;
    };
    // patch chunk of step s -> slot s & 1 (8 wave-loads of 8 rows x 128 B; the last one covers rows 52..59)
    auto issue_patch = [&](int s) {
        int y0, x0;
        tile_origin(t_first + (s >> 1), y0, x0);
        y0 += (s & 1) * FW_HR * d;
        unsigned char* slot = my + (s & 1) * FW_SLOT;
        const int ln = vt_opaque(lane);   // per-row constants recomputed per step, not hoisted and kept live (spills)
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int rbase = i < 7 ? i * 8 : FW_PROWS - 8;
            const int row = rbase + (ln >> 3);
            const int py = row / FK_PW, px = row - py * FK_PW;
            const int iy = y0 + (py - 1) * d, ix = x0 + (px - 1) * d;
            const bool in = (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
            const uint32_t pix = (uint32_t)((img * p.H + iy) * p.W + ix);
            const uint32_t off = in ? pix * ldb + (uint32_t)((((ln & 7) ^ fk_swz(px))) << 4) : GLDS_OOB;
            if (s1) vt_glds16(r1, slot + rbase * 128, off, so);
            else vt_glds16(r0, slot + rbase * 128, off, so);
        }
    };

    // ---- epilogue constants, loaded ONCE and first (hidden: the counted wait of step 0 retires them): four
    // dependent bias loads per step, each waited for with vmcnt(0) behind the LDS-DMA in flight, cost 4.5 us per
    // step in the first version of this kernel ----
    const int c4 = tid & 7;
    const int n = n0 + 4 * c4;
    const bool n_ok = n < p.coutT;               // coutT % 8 == 0: a thread's 4 channels are all in or all out
    u128 bias_v, alpha_v;
    vt_bload_hidden<4>(bias_v, vt_make_raw(p.bias, (uint32_t)p.coutT * 4u), n_ok ? (uint32_t)n * 4u : GLDS_OOB);
    vt_bload_hidden<1>(alpha_v, vt_make_raw(p.alpha_dev, 4u), 0u);
    // the tensors the steps read and write through range-checked descriptors (a lane with nothing to do passes an
    // out-of-range offset; every wave issues the same vector-memory operations, which is what the waits count)
    const BufRsrc rres = vt_make_rsrc(p.resid ? p.resid : p.out,
                                      p.resid ? (uint32_t)((int64_t)p.N * p.H * p.W * p.ld_res * ESZ) : 0u);
    const BufRaw rout = vt_make_raw(p.out, (uint32_t)((int64_t)p.N * p.H * p.W * p.ld_out * ESZ));

    const int epx = tid >> 3;                    // 0..63, the tile pixel as in conv_fullk_kernel
    const int epy_d = (epx >> 3) * d, epx_d = (epx & 7) * d;
    // byte offset of this thread's 4 channels of the pixel it finishes in step s (or out of range), per `ld`
    auto ep_offset = [&](int s, int ld) -> uint32_t {
        int y0, x0;
        tile_origin(t_first + (s >> 1), y0, x0);
        const int oy = y0 + epy_d, ox = x0 + epx_d;
        const bool live = (wave >> 2) == (s & 1) && oy < p.H && ox < p.W && n_ok;
        return live ? (uint32_t)((((img * p.H + oy) * p.W + ox) * ld + n) * ESZ) : GLDS_OOB;
    };
    // residual of step s (bf16: 8 bytes per thread) -> LDS by two 4-byte LDS-DMA loads of the owner waves, issued at the
    // end of step s-1 BEFORE patch s+1, so that the counted wait for that patch (start of step s+1, the step that
    // finishes step s) retires it: no wait in the middle of a step, no register targeted by a load in flight across
    // the loop back-edge
    const bool own0 = (wave >> 2) == 0;          // this wave finishes the even steps
    auto owns = [&](int s) { return ((s & 1) == 0) == own0; };
    auto issue_resid = [&](int s) {
        if (!p.resid || !owns(s)) return;
        const uint32_t off = ep_offset(s, p.ld_res);
        unsigned char* dst = smem + FW_RES + (s & 1) * 2048 + (wave & 3) * 512;
        vt_glds4(rres, dst, off, 0u);
        vt_glds4(rres, dst + 256, off, 4u);
    };

    // ---- prologue: residual of step 0, patch 0, the whole weight slice of this wave (18 pairs, hidden), patch 1 ----
    vt_sched_fence();
    issue_resid(0);
    issue_patch(0);
    vt_sched_fence();   // the counted waits below assume this issue order: patch 0, 36 weight loads, patch 1
    const uint32_t wlane = (uint32_t)lane * 16;
    const unsigned char* wcur = (const unsigned char*)g.wstream + (size_t)((tile_n * FK_NW + wave) * (NSUB * 2)) * 1024;
    u128 wr[NSUB][2];
#pragma unroll
    for (int st = 0; st < NSUB; ++st) vt_gload16_pair_hidden<0>(wr[st][0], wr[st][1], wcur + st * 2048, wlane);
    vt_sched_fence();
    issue_patch(1);
    vt_sched_fence();

    // AdaIN prologue (model/dualstylegan.py:16-21), conv_fullk_kernel's merge verbatim: one image per workgroup, so
    // scale / shift are computed once (lane = channel) and parked in the wave's table
    if (p.in_tile_stats && ABL == 45) {   // ablation: no merge, identity table
        if (lane < BK) tab[lane] = 1.0f, tab[BK + lane] = 0.0f;
        vt_wave_sync();
    } else if (p.in_tile_stats) {
        const int cl = lane & (BK - 1);
        const int d2 = p.in_stats_dil;
        const int nt = d2 * d2 * vt_cdiv_dev(vt_cdiv_dev(p.H, d2), FK_TH) * vt_cdiv_dev(vt_cdiv_dev(p.W, d2), FK_TW);
        const float* rec = p.in_tile_stats + ((size_t)img * nt * p.cin + kc + cl) * 2;
        const float* cnt = p.in_tile_stats + (size_t)p.N * nt * p.cin * 2 + (size_t)img * nt;
        const float x0 = rec[0];
        double s1d = 0.0, s2d = 0.0;
        for (int t0 = 0; t0 < nt; t0 += 16) {
            float mv[16], qv[16], cv[16];
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                const int t = (t0 + k < nt) ? t0 + k : nt - 1;   // clamped: loads stay unconditional
                const u64v rv = *reinterpret_cast<const u64v*>(rec + (size_t)t * p.cin * 2);
                mv[k] = vt_u2f(rv.x);
                qv[k] = vt_u2f(rv.y);
                const float cload = cnt[t];                // unconditional (t is clamped): `cond ? cnt[t] : 0` is a branch
                    cv[k] = (t0 + k < nt) ? cload : 0.0f;       // + a dependent round trip per record
            }
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                const double dm = (double)mv[k] - (double)x0, n = (double)cv[k];
                s1d += n * dm;
                s2d += (cv[k] > 0.0f ? (double)qv[k] : 0.0) + n * dm * dm;
            }
        }
        const double hw = (double)p.H * (double)p.W;
        const double mean = (double)x0 + s1d / hw;
        double var = (s2d - s1d * s1d / hw) / hw;   // biased, as F.instance_norm
        if (var < 0.0) var = 0.0;
        const float rstd = (float)(1.0 / sqrt(var + (double)FK_IN_EPS));
        float gamma = 1.0f, beta = 0.0f;
        if (p.in_gb) {
            gamma = p.in_gb[(size_t)img * p.in_ld_gb + kc + cl];
            beta = p.in_gb[(size_t)img * p.in_ld_gb + p.cin + kc + cl];
        }
        if (lane < BK) {
            tab[cl] = gamma * rstd;
            tab[BK + cl] = beta - gamma * rstd * (float)mean;
        }
        vt_wave_sync();
    }
    // x' = x * scale[c] + shift[c] on the landed patch of step s, pixels inside the image only (the conv's zero
    // padding applies to the NORMALISED tensor), rounded to T like a stored tensor
    auto adain_rewrite = [&](int s) {
        int y0, x0;
        tile_origin(t_first + (s >> 1), y0, x0);
        y0 += (s & 1) * FW_HR * d;
        unsigned char* slot = my + (s & 1) * FW_SLOT;
        float sc[VEC], sh[VEC];
        const int ln = vt_opaque(lane);
        const int jj = ln & 7;
#pragma unroll
        for (int k = 0; k < VEC; k += 4) {
            unpack16<float>(ld128(tab + jj * VEC + k), sc + k);
            unpack16<float>(ld128(tab + BK + jj * VEC + k), sh + k);
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int row = i * 8 + (ln >> 3);
            const int py = row / FK_PW, px = row - py * FK_PW;
            const int iy = y0 + (py - 1) * d, ix = x0 + (px - 1) * d;
            const bool in = row < FW_PROWS && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
            if (in) {
                unsigned char* at = slot + row * 128 + ((jj ^ fk_swz(px)) << 4);
                float f[VEC];
                unpack16<T>(ld128(at), f);
#pragma unroll
                for (int k = 0; k < VEC; ++k) f[k] = fmaf(f[k], sc[k], sh[k]);
                st128(at, pack16<T>(f));
            }
        }
        vt_wave_sync();
    };

    // per-lane LDS read bases: rows (hi, lo) of the fragment, one per (kx, half row)
    uint32_t abase[3][2];
#pragma unroll
    for (int kx = 0; kx < 3; ++kx)
#pragma unroll
        for (int sub = 0; sub < 2; ++sub)
            abase[kx][sub] = (uint32_t)((hi * FK_PW + lo) * 128 + (((sub * 4 + q) ^ fk_swz(kx + lo)) << 4));

    f32x4 acc[2][2];
    auto zero_acc = [&]() {
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
    };
    // park this wave's partial 32-pixel x 32-channel tile: row = pixel (128 B), 16-byte slot s at s ^ (pixel & 7)
    unsigned char* red = smem + FW_RED;
    auto park = [&]() {
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                const int px = a * 16 + l15;
                float f[4] = {acc[a][b][0], acc[a][b][1], acc[a][b][2], acc[a][b][3]};
                st128(red + wave * 4096 + px * 128 + (((2 * q + b) ^ (px & 7)) << 4), pack16<float>(f));
            }
    };

    // ---- sum + epilogue of step s: the half of the workgroup that owned these rows in conv_fullk_kernel
    // (waves 0-3 rows 0-3, waves 4-7 rows 4-7; thread = (pixel, 4 channels)); values kept for the statistics.
    // Lean form of conv_fullk_kernel's epilogue: the host only sends NHWC outputs in the compute dtype with vector
    // stores and a scalar slope here.  The residual read (issue_resid, one step ahead of its use) and the output
    // write are hidden buffer operations: the compiler sees no vector-memory result in the loop, so it never drains
    // the patch in flight; the waits are counted by hand (see the step loop). ----
    float fst[4] = {0.f, 0.f, 0.f, 0.f};
    bool live_st = false;
    float* xs = reinterpret_cast<float*>(smem + FW_RED);   // overlays the partial tiles: one barrier before its first write
    auto finish_step = [&](int s) {
        u128 rv = zero128();
        if (p.resid && owns(s)) {
            const unsigned char* src = smem + FW_RES + (s & 1) * 2048 + (wave & 3) * 512 + lane * 4;
            rv.x = *reinterpret_cast<const uint32_t*>(src);
            rv.y = *reinterpret_cast<const uint32_t*>(src + 256);
        }
        const int h = s & 1;
        const int t = t_first + (s >> 1);
        {
            const uint32_t off = ep_offset(s, p.ld_out);
            const bool live = off != GLDS_OOB;
            const int pxo = epx & 31;            // pixel of this half
            float f[4] = {0.f, 0.f, 0.f, 0.f};
            if ((wave >> 2) == h) {
#pragma unroll
                for (int w = 0; w < FK_NW; ++w) {
                    float gv[4];
                    unpack16<float>(ld128(red + w * 4096 + pxo * 128 + ((c4 ^ (pxo & 7)) << 4)), gv);
#pragma unroll
                    for (int i = 0; i < 4; ++i) f[i] += gv[i];
                }
                const float ga = p.gain_alpha * (p.alpha_dev ? vt_u2f(alpha_v.x) : 1.0f);
                const float b4[4] = {vt_u2f(bias_v.x), vt_u2f(bias_v.y), vt_u2f(bias_v.z), vt_u2f(bias_v.w)};
#pragma unroll
                for (int i = 0; i < 4; ++i) {   // conv_finish for the two activations the host sends here (none, LeakyReLU)
                    float v = f[i] + b4[i];
                    if (p.act == VT_ACT_LRELU) v = (v > 0.0f) ? v : v * p.slope;
                    f[i] = v * ga;
                }
                // the values as stored (conv_fullk_kernel, store_out4): residual, post-activation, rounding to T
                if (live) {
                    if (p.resid) {
                        float g4[4];
                        if (sizeof(T) == 2) {
                            g4[0] = vt_u2f(rv.x << 16); g4[1] = vt_u2f(rv.x & 0xffff0000u);
                            g4[2] = vt_u2f(rv.y << 16); g4[3] = vt_u2f(rv.y & 0xffff0000u);
                        } else {
                            unpack16<float>(rv, g4);
                        }
#pragma unroll
                        for (int i = 0; i < 4; ++i) f[i] += p.beta * g4[i];
                    }
                    post_act_n<4>(p, f);
                } else {
                    f[0] = f[1] = f[2] = f[3] = 0.0f;
                }
            }
            u128 ov = zero128();
            if (sizeof(T) == 2) {
                ov.x = pack_bf16x2(f[0], f[1]);
                ov.y = pack_bf16x2(f[2], f[3]);
                f[0] = vt_u2f(ov.x << 16); f[1] = vt_u2f(ov.x & 0xffff0000u);
                f[2] = vt_u2f(ov.y << 16); f[3] = vt_u2f(ov.y & 0xffff0000u);
            } else {
                ov = pack16<float>(f);
            }
            // every wave issues it: non-owners and dead lanes are out of range (ABL 38: no store; 41..43: cache policy)
            if constexpr (ABL != 38) vt_bstore_hidden<ESZ == 2 ? 2 : 4, (ABL >= 41 && ABL <= 43) ? ABL - 40 : 0>(rout, off, ov);
            if (p.tile_stats && (wave >> 2) == h) {
#pragma unroll
                for (int i = 0; i < 4; ++i) fst[i] = f[i];
                live_st = live;
            }
        }
        if (!p.tile_stats || !h || ABL == 46) return;
        // ---- {mean, M2} record of the 8x8 tile whose second half just finished: two passes, lanes 8 apart hold the
        // 8 pixels of one tile row, the 8 wavefronts the 8 rows; fixed shuffle tree + wave order 0..7 ----
        const int tcount = fk_tile_count(t, d, g.tiles_y, g.tiles_x, p.H, p.W);
        vt_lds_barrier();   // every owner is done reading the partial tiles the exchange buffers overlay
        float s4[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            float v = fst[i];
            v += __shfl_xor(v, 8, 64);
            v += __shfl_xor(v, 16, 64);
            v += __shfl_xor(v, 32, 64);
            s4[i] = v;
        }
        if (lane < 8) {
#pragma unroll
            for (int i = 0; i < 4; ++i) xs[wave * 32 + lane * 4 + i] = s4[i];
        }
        vt_lds_barrier();
        float mean4[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            float v = 0.0f;
#pragma unroll
            for (int w = 0; w < FK_NW; ++w) v += xs[w * 32 + c4 * 4 + i];
            mean4[i] = tcount > 0 ? v / (float)tcount : 0.0f;
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float dv = live_st ? fst[i] - mean4[i] : 0.0f;
            float v = dv * dv;
            v += __shfl_xor(v, 8, 64);
            v += __shfl_xor(v, 16, 64);
            v += __shfl_xor(v, 32, 64);
            s4[i] = v;
        }
        if (lane < 8) {
#pragma unroll
            for (int i = 0; i < 4; ++i) xs[256 + wave * 32 + lane * 4 + i] = s4[i];
        }
        vt_lds_barrier();
        const int tile_g = img * per_img + t;
        if (tid == 0 && tile_n == 0)     // pixel count of this tile, after the records of all images
            p.tile_stats[(size_t)p.N * per_img * p.coutT * 2 + tile_g] = (float)tcount;
        if (tid < 8 && n < p.coutT) {   // thread c4 writes the records of channels 4*c4 .. 4*c4+3
            float* rec = p.tile_stats + ((size_t)tile_g * p.coutT + n) * 2;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                if (n + i >= p.coutT) break;
                float v = 0.0f;
#pragma unroll
                for (int w = 0; w < FK_NW; ++w) v += xs[256 + w * 32 + c4 * 4 + i];
                rec[2 * i] = mean4[i];
                rec[2 * i + 1] = v;
            }
        }
    };

    // ---- the steps.  Vector-memory operations of a wave in issue order (P = the 8 LDS-DMA loads of a patch, W = 36
    // weight loads, R = the 2 residual LDS-DMA loads of an owner wave, S = output write; B, A = bias / d_s reads):
    //     B A R0 P0 W P1 | R1 P2 | S0 R2 P3 | S1 R3 P4 | ... | S(s-1) R(s+1) P(s+2) | ... | S(last)
    // Vector memory retires in issue order, so "at most N outstanding" retires everything but the N youngest.  The only
    // wait of a step is at its start, for patch s -- which also retires R(s-1), issued before it (step s finishes step
    // s-1); what may stay in flight is everything younger: S(s-2), R(s), P(s+1).  No wait in the middle of a step (the
    // first version had one there and ran every step at the latency of the patch issued half a step earlier), none on
    // a store.  R(j) lives in LDS from the end of step j-1 to the middle of step j+1: two buffers, by parity.
    // (SAFE: every wait is vmcnt(0)). ----
    auto wait_patch = [&](int s) {   // s >= 1
        // the younger LOADS only: S(s-2) is not counted, so the wait holds whether or not a store may retire ahead of
        // older loads (it then also waits for that store's acknowledgement, half a step old)
        const int n = ((p.resid && owns(s)) ? 2 : 0) + (s + 1 < nsteps ? 8 : 0);
        if (SAFE || n == 0) vt_vmcnt_fence<0>();
        else if (n == 2) vt_vmcnt_fence<2>();
        else if (n == 8) vt_vmcnt_fence<8>();
        else vt_vmcnt_fence<10>();
    };
    if (SAFE) vt_vmcnt_fence<0>();
    else vt_vmcnt_fence<2 * NSUB + 8>();     // B, A, R0, patch 0 landed (36 weight loads + patch 1 may be in flight)
    if (p.in_tile_stats && ABL != 44) adain_rewrite(0);
    zero_acc();
    fkw_mma<T, true, 0, NSUB>(acc, wr, my, abase);
    vt_sched_fence();
    issue_resid(1);
    if (2 < nsteps) issue_patch(2);
    vt_sched_fence();
    park();
    vt_lds_barrier();
    for (int s = 1; s < nsteps; ++s) {
        const unsigned char* slot = my + (s & 1) * FW_SLOT;
        wait_patch(s);                       // patch s (and the residual of step s-1) landed
        if (p.in_tile_stats && ABL != 44) adain_rewrite(s);
        zero_acc();
        if constexpr (ABL != 31) fkw_mma<T, false, 0, 9, ABL>(acc, wr, slot, abase);
        if constexpr (ABL != 33) finish_step(s - 1);
        if constexpr (ABL != 31) fkw_mma<T, false, 9, NSUB, ABL>(acc, wr, slot, abase);
        vt_sched_fence();
        if (s + 1 < nsteps) issue_resid(s + 1);
        if (s + 2 < nsteps && ABL != 32) issue_patch(s + 2);
        vt_sched_fence();
        if constexpr (ABL != 37) vt_lds_barrier();   // every owner is done reading the partial tiles of step s-1
        if constexpr (ABL != 33) park();
        if constexpr (ABL != 37) vt_lds_barrier();
    }
    vt_vmcnt_fence<0>();                     // the residual of the last step
    finish_step(nsteps - 1);
}

template <typename T>
static bool fullkw_eligible(const ConvArgs& a, const void* wstream, FullkwArgs& g) {
    FullkArgs fg;
    if (sizeof(T) != 2) return false;   // (the residual exchange is sized for 8-byte vectors; fp32 trunks are two rounds anyway)
    if (!fullk_eligible<T>(a, wstream, fg) || fg.rounds != 1) return false;
    // lean epilogue: NHWC in the compute dtype, 8 / 16-byte vector stores, whole 4-channel groups
    if (a.out_layout != VT_OUT_NHWC || !a.vec_store || a.out_f32 != (sizeof(T) == 4) || a.coutT % 8 != 0 || a.phases != 1 ||
        a.slope_vec || (a.act != VT_ACT_NONE && a.act != VT_ACT_LRELU))
        return false;
    g.nrec0 = fg.nrec0;
    g.nrec1 = fg.nrec1;
    g.wstream = wstream;
    g.tiles_y = fg.tiles_y;
    g.tiles_x = fg.tiles_x;
    g.group = 1;
    g.groups_per_img = a.dil * a.dil * fg.tiles_y * fg.tiles_x;
    g.xm = g.xn = 0;
    return true;
}

// G for a batch: the power of two that minimises rounds-of-256-workgroups x (weight stream + G tiles), in units
// of one tile's compute (the weight stream of a workgroup costs about two).  1 = conv_fullk_kernel's schedule.
static int fullkw_group(int n_img, int per_img, int tiles_n) {
    // VT_FULLKW_G: force G (A/B and tests; any value gives the same results).  Read per call: tests flip it.
    const char* fe = getenv("VT_FULLKW_G");
    const int forced = fe ? atoi(fe) : 0;
    if (forced > 0) return forced > 8 ? 8 : forced;
    int best = 1;
    double best_cost = 1e30;
    for (int G = 1; G <= 8; G *= 2) {
        const int64_t wgs = (int64_t)n_img * vt_cdiv(per_img, G) * tiles_n;
        const double cost = (double)vt_cdiv(wgs, 256) * (2.0 + G);
        if (cost < best_cost - 1e-9) best_cost = cost, best = G;
    }
    return best;
}

template <typename T>
int launch_fullkw(const ConvArgs& a, FullkwArgs g, int group, vt_stream stream) {
    ConvArgs args = a;
    args.splitk = 1;
    args.kps = 0;
    args.slab_perm = 0;
    const int per_img = a.dil * a.dil * g.tiles_y * g.tiles_x;
    g.group = group;
    g.groups_per_img = vt_cdiv(per_img, group);
    args.tiles_n = vt_cdiv(a.coutT, FK_BN);
    args.tiles_m = a.N * g.groups_per_img;
    const int64_t blocks = (int64_t)args.tiles_m * args.tiles_n;
    if (blocks >= ((int64_t)1 << 31)) {
        vt_set_error("vt_conv2d: too many tiles");
        return VT_ERR_ARG;
    }
    // XCD grid: minimise the bytes the 8 L2s fetch between them, weights x xm + activations (with halo) x xn, over the
    // splits that divide the grid.
    {
        constexpr int ESZ = (int)sizeof(T);
        const double wbytes = (double)a.coutT * a.K * ESZ;
        const double abytes = 1.9 * (double)a.N * a.H * a.W * a.cin * ESZ;
        double best = 1e300;
        g.xm = g.xn = 0;
        for (int xm = 1; xm <= 8; xm *= 2) {
            const int xn = 8 / xm;
            if (args.tiles_m % xm || args.tiles_n % xn) continue;
            // a sub-grid whose operands do not fit the L2 streams them: price that like a miss per reader
            const bool fits = wbytes / xn + abytes / xm <= 3.5e6;
            const double cost = (wbytes * xm + abytes * xn) * (fits ? 1.0 : 4.0);
            if (cost < best) best = cost, g.xm = xm, g.xn = xn;
        }
    }
    auto k = conv_fullkw_kernel<T, 0>;
    VT_LAUNCH(k, dim3((unsigned)blocks), dim3(FK_NW * 64), stream, args, g);
    return vt_check_launch("vt_conv2d(fullkw)");
}
