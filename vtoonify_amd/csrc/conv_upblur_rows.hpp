// Up-sampling StyledConv of the two top levels (Cin = 64 / 128, bf16) with the blur ON THE MATRIX CORES: conv_transpose2d(3x3,
// stride 2) + 4x4 FIR + bias + LeakyReLU (model/stylegan/model.py:273-286, 74-90, 364-370) without the z tile ever visiting LDS.
// Included by conv_igemm.hip inside its anonymous namespace (round 5).
//
// conv_upblur.hpp parks the pre-blur tile z in LDS and filters it on the vector ALUs: 25 vector instructions per output element
// (address arithmetic, unpacking, 8 separable taps, activation), a phase in which the matrix cores idle and which the SQ
// counters show as 56-64 % "VALU active" on the two top levels (profiles/r05_pmc_upblur_sq.txt) -- those levels are bound by it.
// Here one WAVE owns a strip of 16 input columns and marches down the rows of its block; per input row y:
//
//   1. transposed conv, the 9 taps as in conv_upblur.hpp but with the operands the other way round: A = 16 input pixels of the
//      row (fragments straight from global memory: a lane's 16 bytes are 8 channels of its pixel, no LDS patch), B = the
//      weights out of LDS (all 9 taps x 32 output channels resident for the life of the workgroup).  Four accumulator sets =
//      the parity classes (pa, pb): z[2y + pa][2x + pb]; a lane (q, l15) ends with pixels 4q..4q+3 of channel l15.
//   2. the blur, BOTH directions, as MFMAs on the finished accumulators: the lane's 8 values of a z row (both column parities of
//      its 4 pixels = the 8 z columns 8q..8q+7 of the strip, rounded to bf16 like the z tile of conv_upblur.hpp) ARE the A
//      operand of a product with a constant banded matrix (B operand: FIR row a spread on the band of output column o), and
//      the accumulator is the running sum of the output row the z row feeds with FIR row a:
//          out[Y][c][o] = sum_a sum_j z[Y-1+a][c][j] K[a][j-o-1]        (4 MFMAs per output row and 16 columns, one per z row)
//      The result lands transposed: a lane holds 4 CHANNELS of one output pixel, and with the weight rows loaded in the permuted
//      order of conv_igemm.hip (tile_row_channel) its two channel fragments are 8 consecutive channels = one 16-byte store.
//      The bias is the C operand of the MFMA that starts an output row.  [1,3,3,1]-type kernels are exact in bf16; any other
//      FIR runs the general copy of the loop (a second MFMA per product with the taps' remainders: fp32-grade taps).
//   3. what is left for the vector ALUs: 8 conversions per z row, activation (3 instructions per element), the pack and one
//      range-checked 16-byte store per 16 columns -- about 7 vector instructions per output element instead of 25, no LDS
//      round trip, no barrier: waves are independent after the weights have landed.
//
// Strips overlap by 2 input columns (32 z columns give 28 complete output columns: 1.14x recompute, like conv_upblur.hpp's tiles
// in x), row blocks by 2 input rows (1.05x at the 1024^2 level of a 4-frame step).  Bits: independent of the batch and of the
// block partition (every output element sees the same operations in the same order); not by construction the bits of
// conv_upblur.hpp (the MFMA accumulates the 16 taps in its own order) -- on the frames measured the two agree bit for bit,
// the sums of 16 products of bf16 numbers with 1/16, 3/16, 9/16 being exact in fp32 -- so the choice between the two depends
// on the layer's shape only, never on the batch.
//
// Measured (MI355X, 4 frames, conv_bench, same box): 128 -> 64 @256^2 -> 512^2  109 -> 71-78 us; 64 -> 32 @512^2 -> 1024^2
// 171 -> 101-107 us.  What the kernel waits for now (profiles/r05_uprows.txt): MFMA busy 40 %, VALU 33 %; with the input loads
// out of range 71 us, without the stores 80 us at the top level -- the rest is the serial chain conv -> blur -> store of a wave
// with one partner on its SIMD (245 registers).
#pragma once
#ifndef UR_EXP
#define UR_EXP 0   // experiment builds (python -m vtoonify_amd.build --variant ... -DUR_EXP=bits): 1 no stores, 2 weight fragments
#endif             // not re-read from LDS, 4 no scheduling fences inside the conv, 8 no input loads

struct UprowsArgs {
    uint32_t nrec0, nrecw, nreco;
    int strips;      // 28-column output strips per row
    int rblocks;     // row blocks per image
    int R;           // input rows per block (2R output rows)
    int steps;       // input-row steps per unit: R + 2, rounded up to a multiple of 6
    int units;       // N * rblocks * strips, per channel tile
    int slots;       // workgroups per channel tile
    int xcd_group;   // gridDim.x % (8 * tiles_n) == 0: the channel tiles of one slot run on the same XCD
};

constexpr int UR_OW = 28;   // output columns per strip

template <int CIN>
__global__ void __launch_bounds__(512, 1) conv_upblur_rows_kernel(const ConvArgs p, const UprowsArgs g) {
    using T = bf16_t;
    constexpr int ESZ = 2, BK = 64, CN = 32, TN = 2, NW = 8;
    constexpr int KS = CIN / 32;                     // MFMA K-steps per tap
    constexpr int NCH = CIN / BK;                    // 128-byte chunks of a weight row
    constexpr int WROWS = 9 * CN;                    // LDS weight rows of a chunk: [tap][channel in fragment order]
    constexpr int LBC = (WROWS / 8 + NW - 1) / NW;   // weight loads per wave per chunk
    constexpr int B_BYTES = LBC * NW * 1024;
    static_assert(CIN % BK == 0 && NCH * B_BYTES + 16 * 1024 <= 160 * 1024, "weights resident in LDS");
    constexpr int W_BYTES = NCH * B_BYTES;
    // + the 16 bands of the general form (4 FIR rows x 2 column halves x {head, remainder}, 1 KB each: lane-linear u128)
    __shared__ __attribute__((aligned(1024))) unsigned char smem[W_BYTES + 16 * 1024];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = vt_uniform(tid >> 6) & (NW - 1);
    const int q = lane >> 4, l15 = lane & 15, l7 = lane & 7;
    int tile_n, slot;
    if (g.xcd_group) {
        tile_n = ((int)blockIdx.x >> 3) % p.tiles_n;
        slot = ((int)blockIdx.x / (8 * p.tiles_n)) * 8 + ((int)blockIdx.x & 7);
    } else {
        tile_n = (int)blockIdx.x % p.tiles_n;
        slot = (int)blockIdx.x / p.tiles_n;
    }
    const int n0 = tile_n * CN;
    const int OH = 2 * p.H, OW = 2 * p.W;

    // ---- weights: all 9 taps of this channel tile, once (the loader of conv_upblur.hpp) ---------------------------------
    {
        const int lrow = lane >> 3, jj = l7 ^ lrow;
        const BufRsrc rw = vt_make_rsrc(p.wgt, g.nrecw);
#pragma unroll
        for (int i = 0; i < LBC; ++i) {
            const int row = (i * NW + wave) * 8 + lrow;
            const int tap = row / CN, r = row - tap * CN;
            const int n = n0 + tile_row_channel<true>(r);
            const uint32_t woff = (row < WROWS && n < p.coutT)
                                      ? (uint32_t)n * (uint32_t)(p.K * ESZ) + (uint32_t)(tap * p.cin * ESZ) + jj * 16 : GLDS_OOB;
#pragma unroll
            for (int c = 0; c < NCH; ++c) vt_glds16(rw, smem + c * B_BYTES + (i * NW + wave) * 1024, woff, (uint32_t)(c * BK * ESZ));
        }
    }

    // ---- constants of the wave: the blur matrix, bias / activation of the lane's 8 channels -----------------------------
    float bv[8], gneg[8], ga;
    // B operands of the blur: tk[a][nb] = the band of FIR row a for output columns 16 nb .. 16 nb + 15 of the strip (bf16 heads;
    // tr: the remainders, used only when some tap is not a bf16 number -- [1,3,3,1]-type kernels are exact)
    u128 tk[2][2];              // FIR rows 0 (= 3) and 1 (= 2) of the fast form
    bool fast;                  // exact taps, FIR rows 0 = 3 and 1 = 2, one activation slope: the copy of the loop without remainder bands,
                                // with two bands and a scalar slope (24 registers less)
    float kx[4], ky[4];
    bool exact = true;
    {
        const float* fir = p.up_fir;
        float ksum = 0.0f;
#pragma unroll
        for (int i = 0; i < 4; ++i) kx[i] = ky[i] = 0.0f;
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                const float t = fir[(3 - a) * 4 + (3 - b)];   // upfirdn2d applies the flipped kernel
                ky[a] += t;
                kx[b] += t;
                ksum += t;
            }
        const float inv = 1.0f / ksum;
#pragma unroll
        for (int i = 0; i < 4; ++i) ky[i] *= inv;             // ky[a] kx[b] = the tap (a, b) of the separable FIR (conv_upblur.hpp)
        // K slot t of lane group q = z column j = 2 (4q + (t & 3)) + (t >> 2) of the strip; output column o = 16 nb + l15 of the
        // strip is image column 2 (xs + 1) + o = z column o + 2 of the strip:  out[Y][o] = sum_a sum_b ky[a] kx[b] z[Y-1+a][o+1+b]
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float tap = ky[a] * kx[i];
                exact = exact && (bf16_bits_to_f32(f32_to_bf16_bits(tap)) == tap);
            }
        ga = p.gain_alpha * (p.alpha_dev ? p.alpha_dev[0] : 1.0f);
        // lane's channels: fragment n, element e <-> channel n0 + 8q + 4n + e (frag_channel<true>)
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int nch = n0 + 8 * q + k;
            const int nc = nch < p.coutT ? nch : 0;           // clamped: loads stay unconditional
            const float b = p.bias ? p.bias[nc] : 0.0f;
            const float sl = p.slope_vec ? p.slope_vec[nc] : p.slope;
            bv[k] = nch < p.coutT ? b : 0.0f;
            gneg[k] = (p.act == VT_ACT_LRELU) ? ga * (nch < p.coutT ? sl : p.slope) : ga;
        }
    }
    // (the band's position is computed once per nb; hardware conversions: pack16 rounds to nearest even like f32_to_bf16_bits)
    float kb[2][8];             // kx on the band: K slot t of this lane against output column 16 nb + l15
#pragma unroll
    for (int nb = 0; nb < 2; ++nb) {
        const int o = nb * 16 + l15;
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            const int b = 2 * (4 * q + (t & 3)) + (t >> 2) - o - 1;
            float v = 0.0f;
#pragma unroll
            for (int i = 0; i < 4; ++i) v = (b == i) ? kx[i] : v;
            kb[nb][t] = (o < UR_OW) ? v : 0.0f;
        }
    }
    auto band = [&](int a, int nb, bool remainder) -> u128 {
        float c[8];
#pragma unroll
        for (int t = 0; t < 8; ++t) c[t] = ky[a] * kb[nb][t];
        const u128 hd = pack16<T>(c);
        if (!remainder) return hd;
        float h[8];
        unpack16<T>(hd, h);
#pragma unroll
        for (int t = 0; t < 8; ++t) c[t] -= h[t];
        return pack16<T>(c);
    };
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int nb = 0; nb < 2; ++nb) tk[a][nb] = band(a, nb, false);
    if (wave == 0) {            // the general form reads its bands from LDS (56 registers it does not have at Cin = 128)
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int nb = 0; nb < 2; ++nb) {
                st128(smem + W_BYTES + (a * 2 + nb) * 1024 + lane * 16, band(a, nb, false));
                st128(smem + W_BYTES + (8 + a * 2 + nb) * 1024 + lane * 16, band(a, nb, true));
            }
    }
    fast = exact && ky[0] == ky[3] && ky[1] == ky[2] && !p.slope_vec;
    const bool full = n0 + 8 * q + 8 <= p.coutT;     // (coutT % 8 == 0: a lane's 8 channels are all inside or all outside)
    const BufRsrc r0 = vt_make_rsrc(p.src0, g.nrec0);
    const BufRsrc ro = vt_make_rsrc(p.out, g.nreco);
    const uint32_t pxb = (uint32_t)(p.ld0 * ESZ);

    vt_glds_wait_n<0>();
    __syncthreads();          // the only barrier: from here on the waves are independent

    // the whole unit loop exists twice -- `fast` / the general form with remainder bands -- so that the remainder bands, FIR rows 2
    // and 3 and the per-channel slopes (56 registers) and the constants they are built from are live in the second copy only
    auto run = [&](auto ex) {
    for (int u = slot * NW + wave; u < g.units; u += g.slots * NW) {
        const int s = u % g.strips;
        const int t2 = u / g.strips;
        const int rb = t2 % g.rblocks, img = t2 / g.rblocks;
        const int xs = 14 * s - 1;                       // first input column of the strip (fragment row 0)
        const int yb = rb * g.R - 1;                     // first input row of the march
        const int Ylo = 2 * g.R * rb;
        const int Yhi = Ylo + 2 * g.R < OH ? Ylo + 2 * g.R : OH;
        const int X0 = UR_OW * s;
        int xc[2];
        bool cin_ok[2];
        uint32_t ooff[2];      // byte offset of the lane's 16-byte store in row 0 of its image, or out of range
#pragma unroll
        for (int dj = 0; dj < 2; ++dj) {
            xc[dj] = xs - dj + l15;
            cin_ok[dj] = (unsigned)xc[dj] < (unsigned)p.W;
        }
#pragma unroll
        for (int nb = 0; nb < 2; ++nb) {
            const int X = X0 + nb * 16 + l15;
            const bool ok = full && nb * 16 + l15 < UR_OW && X < OW;
            ooff[nb] = (ok && !(UR_EXP & 1)) ? (uint32_t)(((img * OH) * OW + X) * p.ld_out + n0 + 8 * q) * ESZ : GLDS_OOB;
        }
        const uint32_t orow = (uint32_t)(OW * p.ld_out * ESZ);

        // fragments of input row y: [dj][K-step]; rows / columns outside the image read as zero (= the zero padding of the blur
        // and the missing taps of the border pixels)
        auto loadA = [&](int y, u128 (&fa)[2][KS]) {
            const bool rowok = (unsigned)y < (unsigned)p.H;
#pragma unroll
            for (int dj = 0; dj < 2; ++dj) {
                const uint32_t off = (rowok && cin_ok[dj] && !(UR_EXP & 8)) ? (uint32_t)((img * p.H + y) * p.W + xc[dj]) * pxb + q * 16 : GLDS_OOB;
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) fa[dj][ks] = vt_bload16(r0, off + ks * 64);
            }
        };
        auto conv_row = [&](const u128 (&pv)[2][KS], const u128 (&cu)[2][KS], f32x4 (&acc)[4][TN]) {
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const unsigned char* pb = smem + (ks >> 1) * B_BYTES;
                const int sl = (ks & 1) * 4 + q;
#pragma unroll
                for (int sh = 0; sh < 4; ++sh) {       // taps grouped by input shift (a/2, b/2), conv_upblur.hpp
                    const int di = sh >> 1, dj = sh & 1;
                    const u128& fa = di ? pv[dj][ks] : cu[dj][ks];
#pragma unroll
                    for (int ta = 2 * di; ta < (di ? 3 : 2); ++ta)
#pragma unroll
                        for (int tb = 2 * dj; tb < (dj ? 3 : 2); ++tb) {
                            const int tap = ta * 3 + tb, cls = (ta & 1) * 2 + (tb & 1);
#pragma unroll
                            for (int n = 0; n < TN; ++n) {
                                const u128 fb = (UR_EXP & 2) ? tk[tap & 1][n] : ld128(pb + (tap * CN + n * 16 + l15) * 128 + ((sl ^ l7) << 4));
                                // (input shift (0,0) holds exactly one tap of every parity class: its MFMAs start the accumulators)
                                if (ks == 0 && sh == 0) acc[cls][n] = f32x4{0.f, 0.f, 0.f, 0.f};
                                Mma<T>::run(acc[cls][n], fa, fb);      // rows = pixels, columns = channels
                            }
                        }
                }
            }
            // the LDS reads run PD fragments ahead of the MFMAs that consume them (left alone, hipcc re-uses ONE fragment quad:
            // ds_read -> lgkmcnt(0) -> MFMA, 36 LDS round trips in a row)
            constexpr int NM = 9 * KS * TN, PD = 4;
            vt_sched_group<0x100, PD>();
            vt_static_for<NM - PD>([&](auto) { vt_sched_group<0x008, 1>(); vt_sched_group<0x100, 1>(); });
            vt_sched_group<0x008, PD>();
        };
        // z row i (parity pa of the finished accumulators) enters the four output rows i-2 .. i+1 it feeds -- BOTH blur directions
        // as MFMAs: the lane's 8 values of the z row (both column parities of its 4 pixels, rounded to the compute dtype like the z
        // tile of conv_upblur.hpp) are the A operand, FIR row a spread on the band of tk[a] the B operand, and the running sum of
        // the output row the accumulator: out[Y] = sum_a z[Y-1+a] (x) tk[a].  Rows = channels, columns = output pixels.  The sum
        // that receives FIR row 3 is complete (row i - 2): activation, gain, one 16-byte store per 16 columns; its registers
        // restart as bias + z (x) tk[0] (the bias is the MFMA's C operand).  Sums rotate their roles, conv_upblur.hpp.
        f32x4 bias4[TN];
#pragma unroll
        for (int n = 0; n < TN; ++n) bias4[n] = f32x4{bv[n * 4], bv[n * 4 + 1], bv[n * 4 + 2], bv[n * 4 + 3]};
        auto zrow = [&](auto ex, f32x4 (&o1)[2][TN], f32x4 (&o2)[2][TN], f32x4 (&o3)[2][TN], const f32x4 (&acc)[4][TN], int pa, int i) {
            constexpr bool EX = decltype(ex)::value;
            const int Y = i - 2;
            const bool rowst = Y >= Ylo && Y < Yhi;
            u128 zp[TN];
            // acc += z (x) FIR row a on the band of column half nb (general form: head + remainder, both out of LDS)
            auto blur = [&](f32x4& o, const u128& z, int a, int nb) {
                if constexpr (EX) {
                    Mma<T>::run(o, z, tk[a < 2 ? a : 3 - a][nb]);
                } else {
                    Mma<T>::run(o, z, ld128(smem + W_BYTES + (a * 2 + nb) * 1024 + lane * 16));
                    Mma<T>::run(o, z, ld128(smem + W_BYTES + (8 + a * 2 + nb) * 1024 + lane * 16));
                }
            };
#pragma unroll
            for (int n = 0; n < TN; ++n) {
                const float f[8] = {acc[pa * 2][n][0], acc[pa * 2][n][1], acc[pa * 2][n][2], acc[pa * 2][n][3],
                                    acc[pa * 2 + 1][n][0], acc[pa * 2 + 1][n][1], acc[pa * 2 + 1][n][2], acc[pa * 2 + 1][n][3]};
                zp[n] = pack16<T>(f);
            }
#pragma unroll
            for (int nb = 0; nb < 2; ++nb)
#pragma unroll
                for (int n = 0; n < TN; ++n) {
                    blur(o3[nb][n], zp[n], 3, nb);
                }
#pragma unroll
            for (int nb = 0; nb < 2; ++nb)
#pragma unroll
                for (int n = 0; n < TN; ++n) {
                    blur(o2[nb][n], zp[n], 2, nb);
                    blur(o1[nb][n], zp[n], 1, nb);
                }
#pragma unroll
            for (int nb = 0; nb < 2; ++nb) {
                float f[8];
#pragma unroll
                for (int n = 0; n < TN; ++n)
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float v = o3[nb][n][e];
                        f[n * 4 + e] = v * (v > 0.0f ? ga : (EX ? gneg[0] : gneg[n * 4 + e]));
                    }
                vt_bstore16(ro, rowst ? ooff[nb] + (uint32_t)Y * orow : GLDS_OOB, pack16<T>(f));
#pragma unroll
                for (int n = 0; n < TN; ++n) {
                    o3[nb][n] = bias4[n];
                    blur(o3[nb][n], zp[n], 0, nb);
                }
            }
        };

        u128 A0[2][KS], A1[2][KS];
        f32x4 oa[2][TN], ob[2][TN], oc[2][TN];
#pragma unroll
        for (int nb = 0; nb < 2; ++nb)
#pragma unroll
            for (int n = 0; n < TN; ++n) oa[nb][n] = ob[nb][n] = oc[nb][n] = f32x4{0.f, 0.f, 0.f, 0.f};
        loadA(yb - 1, A0);
        loadA(yb, A1);
        // One input row y: transposed conv of (row y - 1, row y), then the fragments of row y + 1 are requested INTO THE SET OF ROW
        // y - 1, which the conv has just released (they are read one step later: the two z rows of this step and the other wave
        // of the SIMD cover the latency; a third set requested a whole step ahead measured the same or worse, 107-110 against
        // 101-108 us at the 1024^2 level, and does not fit the registers at Cin = 128).  The kernel must stay clear of scratch in
        // this loop: a spilled store offset costs more than its reload -- the scratch load shares vmcnt with the prefetch and
        // turned every store's wait into vmcnt(0), i.e. into the full latency of the loads just issued.
        // Fragment sets alternate (period 2), the running sums rotate per z row (period 3 rows): six rows per trip of the loop.
        auto step = [&](auto cc, auto ex, int y) {
            constexpr int c = decltype(cc)::value;
            u128 (&pv)[2][KS] = c % 2 == 0 ? A0 : A1;   // row y - 1
            u128 (&cu)[2][KS] = c % 2 == 0 ? A1 : A0;   // row y
            vt_sched_fence();
            f32x4 acc[4][TN];
            conv_row(pv, cu, acc);
            vt_sched_fence();
            loadA(y + 1, pv);
            if constexpr ((2 * c) % 3 == 0) zrow(ex, oa, ob, oc, acc, 0, 2 * y);
            else if constexpr ((2 * c) % 3 == 1) zrow(ex, oc, oa, ob, acc, 0, 2 * y);
            else zrow(ex, ob, oc, oa, acc, 0, 2 * y);
            vt_sched_fence();
            if constexpr ((2 * c + 1) % 3 == 0) zrow(ex, oa, ob, oc, acc, 1, 2 * y + 1);
            else if constexpr ((2 * c + 1) % 3 == 1) zrow(ex, oc, oa, ob, acc, 1, 2 * y + 1);
            else zrow(ex, ob, oc, oa, acc, 1, 2 * y + 1);
            vt_sched_fence();
        };
#pragma unroll 1
        for (int t = 0; t < g.steps; t += 6)
            vt_static_for<6>([&](auto cc) { step(cc, ex, yb + t + decltype(cc)::value); });
    }
    };
    if (fast) run(std::true_type{});
    else run(std::false_type{});
}

// Host side.  `force` (VT_UPBLUR_ROWS=1, tests) ignores the size threshold; the choice never looks at the batch.
// (Not by the tile width of the plan either: that one follows the batch -- 16-channel tiles for few-tile launches -- and the
// tile kernels give the same bits at both widths; this kernel always works on 32 channels per workgroup.)
template <typename T>
static bool uprows_wanted(const ConvArgs& a) {
    if constexpr (sizeof(T) != 2) {
        return false;
    } else {
        const char* e = getenv("VT_UPBLUR_ROWS");
        if (e && e[0] == '0') return false;
        if (a.cin != 64 && a.cin != 128) return false;
        if (a.ld0 % 8 != 0 || (uintptr_t)a.src0 % 16 != 0 || a.ld_out % 8 != 0 || (uintptr_t)a.out % 16 != 0) return false;
        // range-checked 32-bit stores: ONE image must fit the window; a batch that does not is cut into groups of images by
        // launch_uprows (the choice of kernel -- hence the bits of a frame -- must not depend on the batch)
        if ((int64_t)4 * a.H * a.W * a.ld_out * 2 >= ((int64_t)1 << 31) - 4096) return false;
        return (e && e[0] == '1') || (int64_t)a.H * a.W >= 128 * 128;
    }
}

template <typename T>
int launch_uprows(const ConvArgs& a, vt_stream stream) {
    UpblurArgs ub;
    UprowsArgs g;
    {   // batches beyond the 32-bit range of the counted loads / stores: groups of images, each its own launch on offset
        // pointers (only src0 and out are per image).  A frame's bits do not depend on the group it rides in.
        const int64_t lim = ((int64_t)1 << 31) - 4096;
        const int64_t in_img = (int64_t)a.H * a.W * a.ld0 * 2, out_img = (int64_t)4 * a.H * a.W * a.ld_out * 2;
        const int64_t per = in_img > out_img ? in_img : out_img;
        if (per > 0 && (int64_t)a.N * per >= lim && a.N > 1) {
            const int gn = (int)((lim - 1) / per) > 0 ? (int)((lim - 1) / per) : 1;
            for (int n0 = 0; n0 < a.N; n0 += gn) {
                ConvArgs sub = a;
                sub.N = a.N - n0 < gn ? a.N - n0 : gn;
                sub.src0 = (const char*)a.src0 + (int64_t)n0 * in_img;
                sub.out = (char*)a.out + (int64_t)n0 * out_img;
                sub.M = sub.N * a.Ho * a.Wo;
                const int rc = launch_uprows<T>(sub, stream);
                if (rc != VT_OK) return rc;
            }
            return VT_OK;
        }
    }
    if (!upblur_eligible<T>(a, ub, 2, UR_OW)) {
        vt_set_error("vt_conv2d: up_fir (conv_transpose + blur) form not supported for this convolution");
        return VT_ERR_UNSUPPORTED;
    }
    ConvArgs args = a;
    args.splitk = 1;
    args.kps = 0;
    args.slab_perm = 0;
    args.tiles_n = vt_cdiv(a.coutT, 32);
    g.nrec0 = ub.nrec0;
    g.nrecw = ub.nrecw;
    g.nreco = (uint32_t)((int64_t)a.N * 4 * a.H * a.W * a.ld_out * 2);
    g.strips = vt_cdiv(2 * a.W, UR_OW);
    const char* e = getenv("VT_UPBLUR_WGS");   // tests: few workgroups = several units per wave
    int wgs = e && atoi(e) > 0 ? atoi(e) : device_cus();
    int slots = wgs / args.tiles_n;
    if (slots >= 8) slots &= ~7;
    if (slots < 1) slots = 1;
    // row blocks: one unit per wave where the image allows it (>= 8 input rows per block: the 2 halo rows stay <= 25 %)
    const int64_t waves = (int64_t)slots * 8;
    int rblocks = (int)(waves / ((int64_t)a.N * g.strips));
    if (rblocks < 1) rblocks = 1;
    int R = vt_cdiv(a.H, rblocks);
    if (R < 8) R = a.H < 8 ? a.H : 8;
    g.R = R;
    g.rblocks = vt_cdiv(a.H, R);
    g.steps = (R + 2 + 5) / 6 * 6;
    const int64_t units = (int64_t)a.N * g.rblocks * g.strips;
    if (units >= ((int64_t)1 << 31)) {
        vt_set_error("vt_conv2d: too many tiles");
        return VT_ERR_ARG;
    }
    g.units = (int)units;
    if ((int64_t)slots * 8 > units) slots = (int)((units + 7) / 8);
    g.slots = slots;
    const int blocks = slots * args.tiles_n;
    g.xcd_group = (slots % 8 == 0) ? 1 : 0;
    if (a.cin == 64) {
        auto k = conv_upblur_rows_kernel<64>;
        VT_LAUNCH(k, dim3((unsigned)blocks), dim3(512), stream, args, g);
    } else {
        auto k = conv_upblur_rows_kernel<128>;
        VT_LAUNCH(k, dim3((unsigned)blocks), dim3(512), stream, args, g);
    }
    return vt_check_launch("vt_conv2d(upblur rows)");
}
