// Whole-K 3x3 convolution for the small-plane, wide-channel layers -- the 32x32-pixel trunk of the
// frame (encoder.4.* VToonifyResBlocks, model/vtoonify.py:92-104,235-239; the AdaResBlocks of
// model/dualstylegan.py:38-45; the Fusion conv, vtoonify.py:125-127).  Included by conv_igemm.hip
// inside its anonymous namespace (shares ConvArgs, Mma<T>, conv_finish, store_out4, decode_block).
//
// Why another kernel: these layers are M = 1024 pixels x N = 512 channels x K = 4608 (4.8 GFLOP,
// 1.9 us of MFMA on 256 CUs).  The patch kernel fills the chip by cutting K into 8 slices per
// 128x128 tile; every slice writes a 64 KB fp32 slab (16.8 MB per conv) that a second kernel
// re-reads -- 4.6-6.7x the algorithmic HBM bytes and a second launch per conv (round-1 PMC).
// Here K is split ACROSS THE 8 WAVEFRONTS OF ONE WORKGROUP instead and summed through LDS:
//
//   * workgroup = 8x8 output pixels x 32 output channels, 512 threads; 1024 x 512 outputs = 256
//     workgroups = one per CU, no split-K workspace, no reduce pass, bf16 written once;
//   * wavefront w owns the input channels [w*BK, (w+1)*BK) of every round of 8*BK channels
//     (BK = 64 bf16 / 32 fp32 = one 128-byte LDS row) and computes the whole 64x32 tile over them:
//     its 10x10-pixel input patch chunk is fetched ONCE into a wave-private LDS region
//     (buffer_load ... lds, zero fill = the conv's zero padding) and serves all 9 taps;
//   * the weights are read exactly once per workgroup and used by one wave only, so they never
//     touch LDS: they stream from L2 straight into VGPRs, pre-packed in MFMA-fragment order
//     (vt_conv_weight_stream: one wave-instruction = one fully coalesced 1 KB line), DEPTH
//     sub-steps ahead;
//   * no workgroup barrier in the main loop (nothing is shared between waves until the end);
//     the 8 partial tiles are exchanged through the (now idle) patch regions and summed in wave
//     order 0..7 -- deterministic;
//   * dilation d is d*d interleaved dense problems: a tile takes every d-th pixel of one phase
//     (y % d, x % d), so the patch is 10x10 pixels for every dilation (a dense 8x8 tile at d = 4
//     would need a 16x16 patch); only the address arithmetic knows about d.
//
// LDS image of a patch chunk: row r = py*10 + px (128 B = BK channels), 16-byte slot s stored at
// slot s ^ 2*((px >> 1) & 3).  An A fragment is 16 pixels = two 8-pixel tile rows; with that XOR
// every ds_read_b128 lane group of gfx950 ({0-3,12-15,20-27}, ...) hits 16 distinct slots of the
// 256-byte bank row for every tap (checked exhaustively, tools/lds_bank_check.py).
#pragma once

struct FullkArgs {
    uint32_t nrec0, nrec1;    // byte sizes of the two sources (buffer range check = zero padding)
    const void* wstream;      // vt_conv_weight_stream image of the weights
    int rounds;               // cin / (8 * BK)
    int tiles_y, tiles_x;     // 8x8-pixel tiles per dilation phase
};

constexpr int FK_TH = 8, FK_TW = 8, FK_BN = 32, FK_NW = 8;
constexpr int FK_PW = FK_TW + 2, FK_PH = FK_TH + 2, FK_PROWS = FK_PW * FK_PH;   // 10 x 10 patch
constexpr int FK_PA = (FK_PROWS + 7) / 8;                                      // 13 loads of 8 rows
constexpr int FK_ABYTES = FK_PA * 1024;                                        // 13 KB per wave
constexpr int FK_DEPTH_DEFAULT = 6;                                            // weight sub-steps in flight

__device__ __forceinline__ int fk_swz(int px) { return ((px >> 1) & 3) << 1; }
constexpr float FK_IN_EPS = 1e-5f;   // nn.InstanceNorm2d default (model/dualstylegan.py:10)

// tile index within an image -> number of its pixels that lie inside the H x W image, for a tiling of dilation d
__device__ __forceinline__ int fk_tile_count(int t, int d, int tiles_y, int tiles_x, int H, int W) {
    const int per_phase = tiles_y * tiles_x;
    const int ph = t / per_phase, rem = t - ph * per_phase;
    const int fy = ph / d, fx = ph - fy * d;
    const int ty = rem / tiles_x, tx = rem - ty * tiles_x;
    int ny = (H - fy + d - 1) / d - ty * FK_TH;   // rows of this phase at or below the tile's first row
    int nx = (W - fx + d - 1) / d - tx * FK_TW;
    ny = ny < 0 ? 0 : ny > FK_TH ? FK_TH : ny;
    nx = nx < 0 ? 0 : nx > FK_TW ? FK_TW : nx;
    return ny * nx;
}

// vt_vmcnt_fence<2 * n>() for a value n that is a compile-time constant after unrolling (0 <= n <= MAXN)
template <int MAXN>
__device__ __forceinline__ void fk_wait_pairs(int n) {
    if (n >= MAXN) vt_vmcnt_fence<2 * MAXN>();
    else fk_wait_pairs<MAXN - 1>(n);
}
template <>
__device__ __forceinline__ void fk_wait_pairs<0>(int) { vt_vmcnt_fence<0>(); }

template <typename T, int FK_DEPTH, int FK_AUX = 0>
__global__ void __launch_bounds__(FK_NW * 64)
conv_fullk_kernel(const ConvArgs p, const FullkArgs g) {
    constexpr int ESZ = (int)sizeof(T);
    constexpr int VEC = 16 / ESZ;
    constexpr int BK = 8 * VEC;                 // channels per 128-byte row
    constexpr int NSUB = 18;                    // sub-steps per chunk: 9 taps x 2 half rows
    static_assert(NSUB % FK_DEPTH == 0, "static ring slots");
    __shared__ __attribute__((aligned(1024))) unsigned char smem[FK_NW * FK_ABYTES];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = vt_uniform(tid >> 6) & (FK_NW - 1);
    const int q = lane >> 4, l15 = lane & 15;
    const int hi = l15 >> 3, lo = l15 & 7;
    int tile_m, tile_n, split;
    decode_block(p, tile_m, tile_n, split);
    const int d = p.dil;
    // tile_m -> (image, phase_y, phase_x, tile_y, tile_x)
    const int per_phase = g.tiles_y * g.tiles_x;
    const int per_img = per_phase * d * d;
    const int img = tile_m / per_img;
    int rem = tile_m - img * per_img;
    const int ph = rem / per_phase;
    rem -= ph * per_phase;
    const int fy = ph / d, fx = ph - fy * d;
    const int ty0 = rem / g.tiles_x, tx0 = rem - ty0 * g.tiles_x;
    const int y0 = fy + ty0 * FK_TH * d, x0 = fx + tx0 * FK_TW * d;   // image position of tile pixel (0, 0)
    const int n0 = tile_n * FK_BN;

    unsigned char* my = smem + wave * FK_ABYTES;
    const BufRsrc r0 = vt_make_rsrc(p.src0, g.nrec0);
    const BufRsrc r1 = vt_make_rsrc(p.src1 ? p.src1 : p.src0, p.src1 ? g.nrec1 : 0u);
    const int nchunks = p.cin / BK;

    // patch chunk of round r -> this wave's LDS region (13 wave-loads of 8 rows x 128 B)
    auto issue_patch = [&](int r) {
        const int kc = (r * FK_NW + wave) * BK;
        const bool s1 = kc >= p.c0;
        const uint32_t so = (uint32_t)((s1 ? kc - p.c0 : kc) * ESZ);
        const uint32_t ldb = (uint32_t)((s1 ? p.ld1 : p.ld0) * ESZ);
#pragma unroll
        for (int i = 0; i < FK_PA; ++i) {
            const int row = i * 8 + (lane >> 3);
            const int py = row / FK_PW, px = row - py * FK_PW;
            const int iy = y0 + (py - 1) * d, ix = x0 + (px - 1) * d;
            const bool in = row < FK_PROWS && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
            const uint32_t pix = (uint32_t)((img * p.H + iy) * p.W + ix);
            const uint32_t off = in ? pix * ldb + (uint32_t)((((lane & 7) ^ fk_swz(px))) << 4) : GLDS_OOB;
            if (s1) vt_glds16(r1, my + i * 1024, off, so);
            else vt_glds16(r0, my + i * 1024, off, so);
        }
    };
    // weight stream: [tile_n][chunk][tap][half][fragment b][lane] x 16 B; the wave-uniform base goes through
    // SGPRs, the lane offset is fixed
    const uint32_t wlane = (uint32_t)lane * 16;
    auto wround = [&](int r) -> const unsigned char* {
        return (const unsigned char*)g.wstream + (size_t)((tile_n * nchunks + r * FK_NW + wave) * (NSUB * 2)) * 1024;
    };

    // Epilogue constants of this thread (it finishes pixel tid >> 3, channels n0 + 4 * (tid & 7) ..+3), fetched BEFORE
    // the main loop and in ONE branch.  The epilogue used to select "bias or 0" per element on a run-time condition:
    // hipcc then branches around each load and waits vmcnt(0) per element -- four dependent L2 round trips (~2 us of a
    // 10 us launch) at the very end of the kernel, where nothing overlaps them (round 3, found on conv_fullkw.hpp).
    float ep_bias[4] = {0.f, 0.f, 0.f, 0.f}, ep_slope[4] = {p.slope, p.slope, p.slope, p.slope};
    float ep_alpha = 1.0f;
    {
        const int en = n0 + 4 * (tid & 7);
        if (en < p.coutT) {   // coutT % 8 == 0 (fullk_eligible): the four channels are all in or all out
            if (p.bias) {
#pragma unroll
                for (int i = 0; i < 4; ++i) ep_bias[i] = p.bias[en + i];
            }
            if (p.slope_vec) {
#pragma unroll
                for (int i = 0; i < 4; ++i) ep_slope[i] = p.slope_vec[en + i];
            }
        }
        if (p.alpha_dev) ep_alpha = p.alpha_dev[0];
    }

    f32x4 acc[4][2];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};

    // per-lane LDS read bases: rows (hi, lo) of the fragment, one per (kx, half)
    uint32_t abase[3][2];
#pragma unroll
    for (int kx = 0; kx < 3; ++kx)
#pragma unroll
        for (int sub = 0; sub < 2; ++sub)
            abase[kx][sub] = (uint32_t)((hi * FK_PW + lo) * 128 + (((sub * 4 + q) ^ fk_swz(kx + lo)) << 4));
    auto read_a = [&](u128 (&fa)[4], int st) {   // the four pixel fragments of sub-step st (compile-time st)
        const int tap = st >> 1, sub = st & 1;
        const int ky = tap / 3, kx = tap - ky * 3;
#pragma unroll
        for (int a = 0; a < 4; ++a) fa[a] = ld128(my + abase[kx][sub] + ((2 * a + ky) * FK_PW + kx) * 128);
    };

    // Every round starts like a prologue (patch + the first DEPTH weight sub-steps issued together, one memory
    // latency) and drains its ring at the end: no register that a hidden load targets is live across the loop
    // back-edge.  The trunk's 512-channel bf16 convs are a single round.
    for (int r = 0; r < g.rounds; ++r) {
        issue_patch(r);   // (r > 0: every ds_read of the previous round has been consumed by an MFMA)
        const unsigned char* wcur = wround(r);
        u128 wr[FK_DEPTH][2];
#pragma unroll
        for (int s = 0; s < FK_DEPTH; ++s) vt_gload16_pair_hidden<FK_AUX>(wr[s][0], wr[s][1], wcur + s * 2048, wlane);
        // AdaIN prologue, part 1 (model/dualstylegan.py:16-21 ahead of every AdaResBlock conv): while the patch
        // and the first weights are in flight, merge the producer's per-tile {mean, M2} records of THIS wave's
        // channels (lane = channel) into scale / shift.  One pass, fp64, tile order (the same in every workgroup
        // and for every batch size), shifted by the first tile's mean:
        //   S1 = sum n_t (m_t - x0),  S2 = sum [M2_t + n_t (m_t - x0)^2];  mean = x0 + S1/N,  M2 = S2 - S1^2/N
        // tile pixel counts n_t come from the producer (no index arithmetic here), loads are issued 16 at a time.
        float ad_scale = 1.0f, ad_shift = 0.0f;
        const int cl = lane & (BK - 1);
        if (p.in_tile_stats) {
            const int kc = (r * FK_NW + wave) * BK;
            const int d2 = p.in_stats_dil;
            const int nt = d2 * d2 * vt_cdiv_dev(vt_cdiv_dev(p.H, d2), FK_TH) * vt_cdiv_dev(vt_cdiv_dev(p.W, d2), FK_TW);
            const float* rec = p.in_tile_stats + ((size_t)img * nt * p.cin + kc + cl) * 2;
            const float* cnt = p.in_tile_stats + (size_t)p.N * nt * p.cin * 2 + (size_t)img * nt;
            const float x0 = rec[0];
            double s1 = 0.0, s2 = 0.0;
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
                    s1 += n * dm;
                    s2 += (cv[k] > 0.0f ? (double)qv[k] : 0.0) + n * dm * dm;
                }
            }
            const double hw = (double)p.H * (double)p.W;
            const double mean = (double)x0 + s1 / hw;
            double var = (s2 - s1 * s1 / hw) / hw;   // biased, as F.instance_norm
            if (var < 0.0) var = 0.0;
            const float rstd = (float)(1.0 / sqrt(var + (double)FK_IN_EPS));
            float gamma = 1.0f, beta = 0.0f;
            if (p.in_gb) {
                gamma = p.in_gb[(size_t)img * p.in_ld_gb + kc + cl];
                beta = p.in_gb[(size_t)img * p.in_ld_gb + p.cin + kc + cl];
            }
            ad_scale = gamma * rstd;
            ad_shift = beta - gamma * rstd * (float)mean;
        }
        vt_vmcnt_fence<0>();   // patch + first weights landed (the compiler's own wait for the LDS-DMA drains both anyway)
        if (p.in_tile_stats) {
            // part 2: park scale / shift in the 4 spare rows of the patch region, then rewrite the landed patch in
            // place: x' = x * scale[c] + shift[c] for pixels inside the image (the zero padding of the conv applies
            // to the NORMALISED tensor), rounded to T like a stored tensor.
            float* tab = reinterpret_cast<float*>(my + FK_PROWS * 128);   // rows 100..103: [scale BK | shift BK]
            if (lane < BK) {
                tab[cl] = ad_scale;
                tab[BK + cl] = ad_shift;
            }
            vt_wave_sync();
            // every lane owns the LOGICAL 16-byte chunk (lane & 7) of its pixel rows: scale / shift of its VEC channels
            // are read once (the first version looked the physical slot's chunk up per row: 4 table reads x 13 rows)
            float sc[VEC], sh[VEC];
            const int jj = lane & 7;
#pragma unroll
            for (int k = 0; k < VEC; k += 4) {
                unpack16<float>(ld128(tab + jj * VEC + k), sc + k);
                unpack16<float>(ld128(tab + BK + jj * VEC + k), sh + k);
            }
#pragma unroll
            for (int i = 0; i < FK_PA; ++i) {
                const int row = i * 8 + (lane >> 3);
                const int py = row / FK_PW, px = row - py * FK_PW;
                const int iy = y0 + (py - 1) * d, ix = x0 + (px - 1) * d;
                const bool in = row < FK_PROWS && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
                if (in) {
                    unsigned char* at = my + row * 128 + ((jj ^ fk_swz(px)) << 4);   // physical slot of the chunk
                    float f[VEC];
                    unpack16<T>(ld128(at), f);
#pragma unroll
                    for (int k = 0; k < VEC; ++k) f[k] = fmaf(f[k], sc[k], sh[k]);
                    st128(at, pack16<T>(f));
                }
            }
            vt_wave_sync();
        }
        u128 fa[2][4];
        read_a(fa[0], 0);
        u128 wk0 = zero128(), wk1 = zero128();   // f32x3: the even half-step's weight fragments, kept for the odd one
#pragma unroll
        for (int st = 0; st < NSUB; ++st) {
            // loads issued after sub-step st's pair: the refills of the following min(DEPTH-1, NSUB-1-st) sub-steps
            constexpr int AHEAD = FK_DEPTH - 1;
            if (st >= FK_DEPTH) fk_wait_pairs<AHEAD>(NSUB - 1 - st < AHEAD ? NSUB - 1 - st : AHEAD);
            // (f32x3 runs VALU work on the pair: vt_settled ties it to the wait above; the MFMA-only forms keep the
            // instruction stream they were tuned and validated with)
            const u128 w0 = is_x3<T>::value ? vt_settled(wr[st % FK_DEPTH][0]) : wr[st % FK_DEPTH][0];
            const u128 w1 = is_x3<T>::value ? vt_settled(wr[st % FK_DEPTH][1]) : wr[st % FK_DEPTH][1];
            if (st + FK_DEPTH < NSUB)
                vt_gload16_pair_hidden<FK_AUX>(wr[st % FK_DEPTH][0], wr[st % FK_DEPTH][1], wcur + (st + FK_DEPTH) * 2048,
                                               wlane);
            if constexpr (is_x3<T>::value) {
                // f32x3 (conv_igemm.hip): a tap's two half-steps carry the 16-byte chunks q and 4+q of the 32-channel row, for
                // pixels (fa[0], fa[1]) and weights (even / odd stream pair) alike: one bf16 fragment after the split
                if ((st & 1) == 0) {
                    wk0 = w0, wk1 = w1;
                    read_a(fa[1], st + 1);
                } else {
                    u128 ah[4], al[4], bh[2], bl[2];
#pragma unroll
                    for (int a = 0; a < 4; ++a) x3_split(fa[0][a], fa[1][a], ah[a], al[a]);
                    x3_split(wk0, w0, bh[0], bl[0]);
                    x3_split(wk1, w1, bh[1], bl[1]);
                    if (st + 1 < NSUB) read_a(fa[0], st + 1);
#pragma unroll
                    for (int a = 0; a < 4; ++a) {
                        Mma<f32x3_t>::run3(acc[a][0], bh[0], bl[0], ah[a], al[a]);
                        Mma<f32x3_t>::run3(acc[a][1], bh[1], bl[1], ah[a], al[a]);
                    }
                }
            } else {
                if (st + 1 < NSUB) read_a(fa[(st + 1) & 1], st + 1);
#pragma unroll
                for (int a = 0; a < 4; ++a) {
                    Mma<T>::run(acc[a][0], w0, fa[st & 1][a]);
                    Mma<T>::run(acc[a][1], w1, fa[st & 1][a]);
                }
            }
        }
    }
    // ---- sum the 8 partial tiles through LDS (each wave parks its tile in its own patch region) ----
    // scratch image: row = tile pixel (128 B = 32 fp32 channels), 16-byte slot s at s ^ (pixel & 7)
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            // fragment b of lane group q holds channels 8q + 4b .. +3 (weight rows are packed in that order)
            const int px = a * 16 + l15;
            float f[4] = {acc[a][b][0], acc[a][b][1], acc[a][b][2], acc[a][b][3]};
            st128(my + px * 128 + (((2 * q + b) ^ (px & 7)) << 4), pack16<float>(f));
        }
    __syncthreads();
    const int px = tid >> 3, c4 = tid & 7;
    float f[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int w = 0; w < FK_NW; ++w) {
        float gv[4];
        unpack16<float>(ld128(smem + w * FK_ABYTES + px * 128 + ((c4 ^ (px & 7)) << 4)), gv);
#pragma unroll
        for (int i = 0; i < 4; ++i) f[i] += gv[i];
    }
    const int oy = y0 + (px >> 3) * d, ox = x0 + (px & 7) * d;
    const int n = n0 + 4 * c4;
    const bool live = oy < p.H && ox < p.W && n < p.coutT;
    const float ga = p.gain_alpha * ep_alpha;
#pragma unroll
    for (int i = 0; i < 4; ++i) f[i] = conv_finish(p, f[i], ep_bias[i], ga, ep_slope[i]);
    const int m = (img * p.H + oy) * p.W + ox;
    if (!p.tile_stats) {
        if (live) store_out4(p, m, n, f);
        return;
    }
    // ---- output + its InstanceNorm tile record (host guarantees NHWC, T-typed, 8-byte aligned vector stores) ----
    // the statistics are those of the ROUNDED values as stored: a later pass over the tensor would see the same
    if (live) {
        if (p.resid) {
            float g4[4];
            if (sizeof(T) == 2) {
                const u64v rv = *reinterpret_cast<const u64v*>((const bf16_t*)p.resid + (int64_t)m * p.ld_res + n);
                g4[0] = vt_u2f(rv.x << 16); g4[1] = vt_u2f(rv.x & 0xffff0000u);
                g4[2] = vt_u2f(rv.y << 16); g4[3] = vt_u2f(rv.y & 0xffff0000u);
            } else {
                unpack16<float>(ld128((const float*)p.resid + (int64_t)m * p.ld_res + n), g4);
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) f[i] += p.beta * g4[i];
        }
        post_act_n<4>(p, f);
        if (sizeof(T) == 2) {
            u64v v;
            v.x = pack_bf16x2(f[0], f[1]);
            v.y = pack_bf16x2(f[2], f[3]);
            *reinterpret_cast<u64v*>((bf16_t*)p.out + (int64_t)m * p.ld_out + n) = v;
            f[0] = vt_u2f(v.x << 16); f[1] = vt_u2f(v.x & 0xffff0000u);
            f[2] = vt_u2f(v.y << 16); f[3] = vt_u2f(v.y & 0xffff0000u);
        } else {
            st128((float*)p.out + (int64_t)m * p.ld_out + n, pack16<float>(f));
        }
    } else {
        f[0] = f[1] = f[2] = f[3] = 0.0f;
    }
    // two passes over the tile's <= 64 pixels per channel: lanes 8 apart hold the 8 pixels of one tile row (same
    // c4), the 8 wavefronts hold the 8 rows; fixed shuffle tree + wave order 0..7 => deterministic
    const int tcount = fk_tile_count(tile_m - img * per_img, d, g.tiles_y, g.tiles_x, p.H, p.W);
    float* xs = reinterpret_cast<float*>(smem);   // 8 waves x 32 channels
    __syncthreads();   // every wave is done reading the partial tiles
    float s4[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        float v = f[i];
        v += __shfl_xor(v, 8, 64);
        v += __shfl_xor(v, 16, 64);
        v += __shfl_xor(v, 32, 64);
        s4[i] = v;
    }
    if (lane < 8) {
#pragma unroll
        for (int i = 0; i < 4; ++i) xs[wave * 32 + lane * 4 + i] = s4[i];
    }
    __syncthreads();
    float mean4[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        float v = 0.0f;
#pragma unroll
        for (int w = 0; w < FK_NW; ++w) v += xs[w * 32 + c4 * 4 + i];
        mean4[i] = tcount > 0 ? v / (float)tcount : 0.0f;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const float dv = live ? f[i] - mean4[i] : 0.0f;
        float v = dv * dv;
        v += __shfl_xor(v, 8, 64);
        v += __shfl_xor(v, 16, 64);
        v += __shfl_xor(v, 32, 64);
        s4[i] = v;
    }
    if (lane < 8) {
#pragma unroll
        for (int i = 0; i < 4; ++i) xs[wave * 32 + lane * 4 + i] = s4[i];
    }
    __syncthreads();
    if (tid == 0 && tile_n == 0)     // pixel count of this tile, after the records of all images
        p.tile_stats[(size_t)p.N * per_img * p.coutT * 2 + tile_m] = (float)tcount;
    if (tid < 8 && n < p.coutT) {   // thread c4 writes the records of channels 4*c4 .. 4*c4+3
        float* rec = p.tile_stats + ((size_t)tile_m * p.coutT + n) * 2;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if (n + i >= p.coutT) break;
            float v = 0.0f;
#pragma unroll
            for (int w = 0; w < FK_NW; ++w) v += xs[w * 32 + c4 * 4 + i];
            rec[2 * i] = mean4[i];
            rec[2 * i + 1] = v;
        }
    }
}

// fragment-stream image of packed weights [cout][taps][cin] (vt_conv_weight_stream)
template <typename T>
__global__ void __launch_bounds__(256)
weight_stream_kernel(T* __restrict__ out, const T* __restrict__ w, int cout, int taps, int cin, int64_t total16) {
    constexpr int VEC = 16 / (int)sizeof(T);
    constexpr int BK = 8 * VEC;
    const int nchunks = cin / BK;
    for (int64_t v = (int64_t)blockIdx.x * 256 + threadIdx.x; v < total16; v += (int64_t)gridDim.x * 256) {
        // v = ((((tile_n * nchunks + chunk) * taps + tap) * 2 + sub) * 2 + b) * 64 + lane
        const int lane = (int)(v & 63);
        int64_t t = v >> 6;
        const int b = (int)(t & 1); t >>= 1;
        const int sub = (int)(t & 1); t >>= 1;
        const int tap = (int)(t % taps); t /= taps;
        const int chunk = (int)(t % nchunks);
        const int tn = (int)(t / nchunks);
        const int q = lane >> 4, l15 = lane & 15;
        const int n = tn * FK_BN + 8 * (l15 >> 2) + 4 * b + (l15 & 3);
        const int k = chunk * BK + sub * (BK / 2) + q * VEC;
        u128 val = zero128();
        if (n < cout) val = ld128(w + ((int64_t)n * taps + tap) * cin + k);
        st128(out + v * VEC, val);
    }
}

template <typename T>
static bool fullk_eligible(const ConvArgs& a, const void* wstream, FullkArgs& g) {
    constexpr int ESZ = (int)sizeof(T);
    constexpr int BK = 8 * (16 / ESZ);
    if (!wstream || a.force_generic || a.transposed || a.in_scale || a.rgb_w) return false;
    if (a.taps != 9 || a.kw != 3 || a.stride != 1 || a.pad != a.dil || a.dil < 1 || a.dil > 8) return false;
    if (a.Ho != a.H || a.Wo != a.W || a.phases != 1) return false;
    if (a.cin % (FK_NW * BK) != 0 || a.c0 % BK != 0 || a.coutT % 8 != 0) return false;
    const int64_t lim = ((int64_t)1 << 31) - 4096;
    const int64_t px = (int64_t)a.N * a.H * a.W;
    const int64_t n0 = px * a.ld0 * ESZ, n1 = px * a.ld1 * ESZ;
    if (n0 >= lim || n1 >= lim) return false;
    g.nrec0 = (uint32_t)n0;
    g.nrec1 = (uint32_t)n1;
    g.wstream = wstream;
    g.rounds = a.cin / (FK_NW * BK);
    g.tiles_y = vt_cdiv(vt_cdiv(a.H, a.dil), FK_TH);
    g.tiles_x = vt_cdiv(vt_cdiv(a.W, a.dil), FK_TW);
    return true;
}

template <typename T>
int launch_fullk(const ConvArgs& a, const FullkArgs& g, vt_stream stream) {
    ConvArgs args = a;
    args.splitk = 1;
    args.kps = 0;
    args.slab_perm = 0;
    args.tiles_n = vt_cdiv(a.coutT, FK_BN);
    args.tiles_m = a.N * a.dil * a.dil * g.tiles_y * g.tiles_x;
    const int64_t blocks = (int64_t)args.tiles_m * args.tiles_n;
    if (blocks >= ((int64_t)1 << 31)) {
        vt_set_error("vt_conv2d: too many tiles");
        return VT_ERR_ARG;
    }
    if constexpr (sizeof(T) == 4 && !is_x3<T>::value) {
        if (a.x3) {   // f32x3 instance (conv_igemm.hip, "f32x3")
            auto k = conv_fullk_kernel<f32x3_t, FK_DEPTH_DEFAULT>;
            VT_LAUNCH(k, dim3((unsigned)blocks), dim3(FK_NW * 64), stream, args, g);
            return vt_check_launch("vt_conv2d(whole-K, f32x3)");
        }
    }
    auto k = conv_fullk_kernel<T, FK_DEPTH_DEFAULT>;
    VT_LAUNCH(k, dim3((unsigned)blocks), dim3(FK_NW * 64), stream, args, g);
    return vt_check_launch("vt_conv2d(fullk)");
}
