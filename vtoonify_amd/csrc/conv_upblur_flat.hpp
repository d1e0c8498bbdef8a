// The DEEP up-sampling StyledConvs (Cin >= 256: 32^2 -> 64^2, 64^2 -> 128^2, 128^2 -> 256^2) on flattened quad tiles (round 6).
// Same three phases and the same arithmetic as conv_upblur_kernel (conv_upblur.hpp: conv_transpose2d(3x3, stride 2) by
// parity class on the matrix cores, z tile parked in LDS, separable 4-tap blur + bias + LeakyReLU on the vector ALUs;
// model/stylegan/model.py:273-286, 74-90, 364-370) -- and the same K order [64-channel chunk][32-channel half][input shift]
// [tap], so the output equals that kernel's bit for bit (tests/test_ops.py).  What changes is the tile and the K pipeline:
//
//   * conv_upblur's tiles are QY rows of exactly one 16-quad MFMA fragment: 12 x 16 (or 24 x 16) quads give 20 x 28 (44 x 28)
//     output pixels, and a 64^2 output needs 4 x 3 of them -- 2304 quads are computed per image where 1024 input pixels exist
//     (2.25x; 1.4x at 128^2).  Here a tile is R = 10 rows of PW = 34 quads, a FLAT run of 340 quads cut into 16-quad
//     fragments that wrap around the tile's rows; the patch sits in LDS with the same pitch (34 pixels a row), so that the
//     pixel a quad reads for input shift (di, dj) is its own flat index plus a constant and a fragment is 16 consecutive
//     LDS rows whatever row boundary it straddles.  (The wrapped-around neighbours feed only the first quad of a row, whose
//     even-column class lies outside the z tile.)  Output tile = 16 x 64 pixels: it divides 64 / 128 / 256, the 64^2 level
//     is ONE tile wide (no horizontal halo at all), 1.33x quads per input pixel instead of 2.25x / 1.41x / 1.41x.
//   * K in steps of 32 channels (one MFMA K), 64-byte LDS rows, three stages of 48 KB in a ring: the loads of step s + 2
//     are issued behind the barrier of step s, every wave issues the same six 1-KiB LDS-DMA pieces per stage (patch 24,
//     weights 18, the rest out of range = zeros) so that ONE counted vmcnt per step says "stage s has landed".
//     conv_upblur's single-stage tall form waits for a whole 92 KB chunk (vmcnt 0) with nothing else resident on the CU.
//   * 8 waves, fragments dealt round robin (22 fragments of quads: the third fragment of waves 6 and 7 lies past the tile --
//     it is computed on whatever the stage holds there and dropped; a SIMD runs two waves and waits for 6 fragments either way).
//
// z LDS image, blur and store are conv_upblur's (128-byte lines of LP pixels, slot s of line l at s ^ (l & 7)).
// Included by conv_igemm.hip inside its anonymous namespace, after conv_upblur.hpp (UpblurArgs, upblur_eligible).
#pragma once

constexpr int UF_PW = 34;   // quads (= patch pixels) per tile row: 2 (PW - 2) = 64 output columns
// tap u of a K step: row a, column b of the 3x3 filter and its input shift 2 (a / 2) + b / 2
constexpr int UF_TA[9] = {0, 0, 1, 1, 0, 1, 2, 2, 2};
constexpr int UF_TB[9] = {0, 1, 0, 1, 2, 2, 0, 1, 2};
constexpr int UF_SH[9] = {0, 0, 0, 0, 1, 1, 2, 2, 3};

template <typename T, int CN, int R>
__global__ void __launch_bounds__(512, 1)
conv_upflat_kernel(const ConvArgs p, const UpblurArgs g) {
    static_assert(sizeof(T) == 2 && !is_x3<T>::value, "16-bit operands");
    constexpr int ESZ = 2, VEC = 8;
    constexpr int NW = 8, NT = NW * 64;
    constexpr int HK = 32;                          // input channels per K step
    constexpr int PW = UF_PW;
    constexpr int TY = 2 * (R - 2), TX = 2 * (PW - 2);
    constexpr int NQ = R * PW;                      // quads of the tile, flat index f = r * PW + c
    constexpr int NFRAG = (NQ + 15) / 16;
    constexpr int MF = (NFRAG + NW - 1) / NW;       // fragments per wave: fragment m * NW + wave
    constexpr int TN = CN / 16;
    constexpr bool PERM = (TN % 2 == 0);
    // patch pixel pr (flat, pitch PW, one leading slot): pr = 1 + py * PW + k  <->  input pixel (I0 - 2 + py, J0 - 1 + k);
    // quad f reads pr = f + 1 + (1 - di) * PW - dj
    constexpr int NPIX = (R + 1) * PW + 1;
    constexpr int PPIECES = (NPIX + 15) / 16;       // 1-KiB pieces of 16 pixel rows x 64 B
    constexpr int WROWS = 9 * CN, WPIECES = WROWS / 16;
    constexpr int PL = (PPIECES + WPIECES + NW - 1) / NW;   // pieces per wave per stage (uniform: counted vmcnt)
    constexpr int A_BYTES = PPIECES * 1024;
    constexpr int STAGE = PL * NW * 1024;
    constexpr int NST = 3;
    constexpr int FD = 3;                           // fragment pipeline: taps in flight
    constexpr int ZH = 2 * R - 1, ZW = 2 * PW - 1;
    constexpr int PXB = CN * ESZ;
    constexpr int LP = 128 / PXB;
    constexpr int ZLINES = (ZW + LP - 1) / LP;
    constexpr int Z_BYTES = ZH * ZLINES * 128;
    constexpr int SMEM = NST * STAGE > Z_BYTES ? NST * STAGE : Z_BYTES;
    static_assert(WROWS % 16 == 0 && CN % 16 == 0 && 128 % PXB == 0, "tile shape");
    static_assert(16 * MF * NW + PW + 1 <= STAGE / 64, "fragment reads stay inside the stage");
    static_assert(SMEM <= 160 * 1024, "LDS budget");
    static_assert(2 * PL < 64, "vmcnt is 6 bits");

    __shared__ __attribute__((aligned(1024))) unsigned char smem[SMEM];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = vt_uniform(tid >> 6) & (NW - 1);
    const int q = lane >> 4, l15 = lane & 15;
    int tile_m, tile_n;
    decode_block_pixel_major(p, tile_m, tile_n);
    if (tile_m >= p.tiles_m) return;
    const int per_img = g.tiles_y * g.tiles_x;
    const int n0 = tile_n * CN;
    const int OH = 2 * p.H, OW = 2 * p.W;
    const int img = tile_m / per_img;
    const int trem = tile_m - img * per_img;
    const int u0 = (trem / g.tiles_x) * TY, v0 = (trem % g.tiles_x) * TX;   // first output pixel of the tile
    const int I0 = u0 / 2, J0 = v0 / 2;

    // ---- loader: lane l of a piece = row l >> 2 of its 16 rows, physical 16-byte slot l & 3 of the 64-byte row; the logical
    // slot (channels 8 s .. 8 s + 7 of the step) of physical slot ps of row r is ps ^ 2 ((r >> 2) & 1) -- conv3x3_c32_kernel's
    // swizzle G = {0, 2, 0, 2}: conflict-free for gfx950's ds_read_b128 lane groups ({0-3, 12-15, 20-27}, {4-11, 16-19, 28-31})
    // over 16 consecutive rows from ANY start (tools/lds_bank_check.py).  The first build used ps ^ ((r >> 2) & 3), which is
    // conflict-free for CONTIGUOUS 16-lane groups and 2-way for the real ones: every fragment read took two LDS passes.
    const int lrow = lane >> 2;
    const int jj = (lane & 3) ^ (((lane >> 4) & 1) << 1);
    uint32_t ldo[PL];
#pragma unroll
    for (int i = 0; i < PL; ++i) {
        const int pc = i * NW + wave;
        if (pc < PPIECES) {
            const int pr = pc * 16 + lrow;
            const int e = pr - 1;
            const int py = e / PW, k = e - py * PW;
            const int iy = I0 - 2 + py, ix = J0 - 1 + k;
            const bool in = pr >= 1 && pr < NPIX && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
            const uint32_t pix = (uint32_t)((img * p.H + iy) * p.W + ix);
            ldo[i] = in ? pix * (uint32_t)(p.ld0 * ESZ) + jj * 16 : GLDS_OOB;
        } else {
            // weights of a step: LDS row tap * CN + r = tap, tile row r (fragment order, see tile_row_channel)
            const int row = (pc - PPIECES) * 16 + lrow;
            const int tap = row / CN, r = row - tap * CN;
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

    // ---- fragment addresses inside a stage: pixel rows by (fragment, shift), weight rows by lane ----
    uint32_t aoff[MF][4];
#pragma unroll
    for (int m = 0; m < MF; ++m)
#pragma unroll
        for (int sh = 0; sh < 4; ++sh) {
            const int di = sh >> 1, dj = sh & 1;
            const int pr = (m * NW + wave) * 16 + l15 + 1 + (1 - di) * PW - dj;
            aoff[m][sh] = (uint32_t)(pr * 64 + ((q ^ (((pr >> 2) & 1) << 1)) << 4));
        }
    const uint32_t boff = (uint32_t)(A_BYTES + l15 * 64 + ((q ^ (((l15 >> 2) & 1) << 1)) << 4));

    f32x4 acc[4][MF][TN];
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int m = 0; m < MF; ++m)
#pragma unroll
            for (int n = 0; n < TN; ++n) acc[c][m][n] = f32x4{0.f, 0.f, 0.f, 0.f};

    // ---- 1. transposed conv: 9 taps per 32-channel step, accumulator set = parity class of the tap ----------------------
    issue(0, 0);
    if (nsteps > 1) issue(1, 1);
    int st = 0;
    for (int step = 0; step < nsteps; ++step) {
        if (step + 1 < nsteps) vt_glds_wait_n<PL>();   // the younger stage may still fly
        else vt_glds_wait_n<0>();
        vt_lds_barrier();                                // ... everybody's pieces of this stage; every wave is past stage st - 1
        if (step + 2 < nsteps) issue(step + 2, st == 0 ? NST - 1 : st - 1);
        const unsigned char* sbase = smem + st * STAGE;
        // the 9 taps in conv_upblur's order -- grouped by input shift (a/2, b/2): the taps (0|1, 0|1) of a shift share one set
        // of pixel fragments -- as a software pipeline: the fragments of tap u + FD - 1 are requested before the MFMAs of tap u
        // (the compiler's own schedule read a tap's weights and waited for them, profiles/r06_upflat.txt)
        u128 fa[2][MF], fb[FD][TN];
        auto read_unit = [&](auto uc) {
            constexpr int u = decltype(uc)::value;
            constexpr int sh = UF_SH[u], tap = UF_TA[u] * 3 + UF_TB[u];
            if constexpr (u == 0 || UF_SH[u] != UF_SH[u > 0 ? u - 1 : 0]) {
#pragma unroll
                for (int m = 0; m < MF; ++m) fa[sh & 1][m] = ld128(sbase + aoff[m][sh]);
            }
#pragma unroll
            for (int n = 0; n < TN; ++n) fb[u % FD][n] = ld128(sbase + boff + (tap * CN + n * 16) * 64);
        };
        vt_static_for<FD - 1>([&](auto uc) { read_unit(uc); });
        vt_static_for<9>([&](auto uc) {
            constexpr int u = decltype(uc)::value;
            constexpr int cls = (UF_TA[u] & 1) * 2 + (UF_TB[u] & 1), aset = UF_SH[u] & 1;
            vt_sched_fence();
            if constexpr (u + FD - 1 < 9) read_unit(std::integral_constant<int, (u + FD - 1 < 9 ? u + FD - 1 : 0)>{});
#pragma unroll
            for (int m = 0; m < MF; ++m)
#pragma unroll
                for (int n = 0; n < TN; ++n) Mma<T>::run(acc[cls][m][n], fb[u % FD][n], fa[aset][m]);
        });
        vt_sched_fence();
        st = st + 1 == NST ? 0 : st + 1;
    }
    __syncthreads();   // the stages are dead

    // ---- 2. z tile -> LDS.  quad (r, c) class (pa, pb) = z pixel (2r + pa - 1, 2c + pb - 1) of the tile ----------------
#pragma unroll
    for (int m = 0; m < MF; ++m) {
        const int f = (m * NW + wave) * 16 + l15;
        const int r = f / PW, c = f - r * PW;
        if (f >= NQ) continue;
#pragma unroll
        for (int cl = 0; cl < 4; ++cl) {
            const int zy = 2 * r + (cl >> 1) - 1, zx = 2 * c + (cl & 1) - 1;
            if (zy < 0 || zx < 0) continue;
            const int line = zx / LP, subp = zx - line * LP;
            unsigned char* zl = smem + (zy * ZLINES + line) * 128;
#pragma unroll
            for (int n = 0; n < TN; ++n) {
                const int ch = frag_channel<PERM>(n, q);                // first of this lane's 4 channels
                const int b0 = subp * PXB + ch * ESZ;                   // byte offset inside the line
                const int phys = ((b0 >> 4) ^ (line & 7)) << 4;
                u64v v;
                v.x = pack_bf16x2(acc[cl][m][n][0], acc[cl][m][n][1]);
                v.y = pack_bf16x2(acc[cl][m][n][2], acc[cl][m][n][3]);
                *reinterpret_cast<u64v*>(zl + phys + (b0 & 15)) = v;
            }
        }
    }
    __syncthreads();

    // ---- 3. blur + bias + activation: conv_upblur_kernel's phase 3 (same operands in the same order) --------------------
    constexpr int NV = CN / VEC;                 // 16-byte channel vectors per pixel
    constexpr int GROUPS = NT / (TX * NV);
    constexpr int ROWS = TY / GROUPS;
    static_assert(GROUPS >= 1 && TX * NV * GROUPS == NT && TY % GROUPS == 0, "one thread per (column, channel vector, row group)");
    float kx[4], ky[4], bv[VEC], gpos[VEC], gneg[VEC];   // act(v) * gain = v * (v > 0 ? gpos : gneg)
    const int qv = tid % NV, col = (tid / NV) % TX, grp = tid / (NV * TX);
    const int nch = n0 + qv * VEC;
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
        for (int i = 0; i < 4; ++i) ky[i] *= inv;
        const float ga = p.gain_alpha * (p.alpha_dev ? p.alpha_dev[0] : 1.0f);
        float slv[VEC];
#pragma unroll
        for (int k = 0; k < VEC; ++k) bv[k] = 0.0f, slv[k] = p.slope;
        if (p.bias) {
#pragma unroll
            for (int k = 0; k < VEC; ++k) bv[k] = p.bias[nch + k < p.coutT ? nch + k : 0];
        }
        if (p.slope_vec) {
#pragma unroll
            for (int k = 0; k < VEC; ++k) slv[k] = p.slope_vec[nch + k < p.coutT ? nch + k : 0];
        }
#pragma unroll
        for (int k = 0; k < VEC; ++k) {
            if (nch + k >= p.coutT) bv[k] = 0.0f, slv[k] = p.slope;
            gpos[k] = ga;
            gneg[k] = (p.act == VT_ACT_LRELU) ? ga * slv[k] : ga;
        }
    }
    const int r0w = grp * ROWS;
    const unsigned char* zt[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const int zx = col + t;
        const int line = zx / LP, subp = zx - line * LP;
        const int s = (subp * PXB + qv * 16) >> 4;
        zt[t] = smem + (r0w * ZLINES + line) * 128 + ((s ^ (line & 7)) << 4);
    }
    const int64_t ostep = (int64_t)OW * p.ld_out;
    const bool full = nch + VEC <= p.coutT;
    {
        const int ov = v0 + col;
        float oa[VEC], ob[VEC], oc[VEC];
#pragma unroll
        for (int k = 0; k < VEC; ++k) oa[k] = ob[k] = oc[k] = 0.0f;
        T* o = (T*)p.out + (((int64_t)img * OH + u0 + r0w - 3) * OW + ov) * p.ld_out + nch;   // row (rr - 3) of this thread
        const bool colok = ov < OW;
        const unsigned char *z0 = zt[0], *z1 = zt[1], *z2 = zt[2], *z3 = zt[3];
        int rr = 0;
        auto row = [&](float (&o1)[VEC], float (&o2)[VEC], float (&o3)[VEC]) {
            vt_sched_fence();   // one row's unpacked pixels live at a time
            const u128 c0 = ld128(z0), c1 = ld128(z1), c2 = ld128(z2), c3 = ld128(z3);
            z0 += ZLINES * 128; z1 += ZLINES * 128; z2 += ZLINES * 128; z3 += ZLINES * 128;
            float f0[VEC], f1[VEC], f2[VEC], f3[VEC], h[VEC];
            unpack16<T>(c0, f0);
            unpack16<T>(c1, f1);
            unpack16<T>(c2, f2);
            unpack16<T>(c3, f3);
            float f[VEC];
#pragma unroll
            for (int k = 0; k < VEC; ++k) {
                h[k] = fmaf(f1[k], kx[1], f0[k] * kx[0]) + fmaf(f3[k], kx[3], f2[k] * kx[2]);
                const float v = fmaf(h[k], ky[3], o3[k]);     // completes output row rr - 3
                f[k] = v * (v > 0.0f ? gpos[k] : gneg[k]);
                o2[k] = fmaf(h[k], ky[2], o2[k]);             // -> the next row's o3
                o1[k] = fmaf(h[k], ky[1], o1[k]);             // -> the next row's o2
                o3[k] = fmaf(h[k], ky[0], bv[k]);             // -> the next row's o1
            }
            const int u = rr - 3;
            if (u >= 0 && u0 + r0w + u < OH && colok) {
                if (full) {
                    st128(o, pack16<T>(f));
                } else {
                    for (int k = 0; k < VEC && nch + k < p.coutT; ++k) o[k] = from_f32<T>(f[k]);
                }
            }
            o += ostep;
            ++rr;
        };
        auto step3 = [&](auto cc) {
            constexpr int c = decltype(cc)::value;
            if constexpr (c % 3 == 0) row(oa, ob, oc);
            else if constexpr (c % 3 == 1) row(oc, oa, ob);
            else row(ob, oc, oa);
        };
        constexpr int NROW = ROWS + 3;
#pragma unroll 1
        for (int t = 0; t < NROW / 3; ++t) vt_static_for<3>([&](auto cc) { step3(cc); });
        vt_static_for<NROW % 3>([&](auto cc) { step3(cc); });
    }
}

// the flat tiles pay wherever the layer has enough K steps for the ring to run in (Cin >= 256: the three deep levels): same-box
// A/B at 1 / 2 / 4 / 16 frames (profiles/r06_upflat.txt): 32^2 -> 64^2  30 / 46 / 70 / 215 -> 28 / 31 / 35 / 130 us,
// 64^2 -> 128^2  45 / 66 / 74 / 270 -> 30 / 34 / 67 / 240,  128^2 -> 256^2  45 / 46 / 91 / 354 -> 23 / 41 / 82 / 317.
// VT_UPBLUR_FLAT = 0 / 1: never / whenever eligible (tests)
template <typename T>
static bool upflat_wanted(const ConvArgs& a) {
    if constexpr (sizeof(T) != 2 || is_x3<T>::value) return false;
    const char* e = getenv("VT_UPBLUR_FLAT");
    if (e && e[0] == '0') return false;
    if (a.cin % 64 != 0) return false;
    if (e && e[0] == '1') return true;
    return a.cin >= 256;
}
// 16-channel tiles while 32-channel ones leave half of the CUs without a workgroup (one or two frames at the deepest levels).
// Same bits (the K order does not depend on the tile width; tests/test_ops.py).  VT_UPBLUR_FLAT_CN = 16 / 32 forces (tests)
static int upflat_cn(const ConvArgs& a) {
    const char* e = getenv("VT_UPBLUR_FLAT_CN");
    if (e && atoi(e) > 0) return atoi(e) == 16 ? 16 : 32;
    const int64_t wgs32 = (int64_t)a.N * vt_cdiv(2 * a.H, 16) * vt_cdiv(2 * a.W, 64) * vt_cdiv(a.coutT, 32);
    return wgs32 <= 128 ? 16 : 32;
}

template <typename T, int CN>
int launch_upflat_cn(const ConvArgs& a, vt_stream stream) {
    constexpr int R = 10;
    UpblurArgs g;
    if (!upblur_eligible<T>(a, g, 2 * (R - 2), 2 * (UF_PW - 2))) {
        vt_set_error("vt_conv2d: up_fir (conv_transpose + blur) form not supported for this convolution");
        return VT_ERR_UNSUPPORTED;
    }
    ConvArgs args = a;
    args.splitk = 1;
    args.kps = 0;
    args.slab_perm = 0;
    args.tiles_n = vt_cdiv(a.coutT, CN);
    args.tiles_m = a.N * g.tiles_y * g.tiles_x;
    const int64_t blocks = (int64_t)args.tiles_m * args.tiles_n;
    if (blocks >= ((int64_t)1 << 31)) {
        vt_set_error("vt_conv2d: too many tiles");
        return VT_ERR_ARG;
    }
    if constexpr (sizeof(T) == 2 && !is_x3<T>::value) {
        auto k = conv_upflat_kernel<T, CN, R>;
        VT_LAUNCH(k, dim3((unsigned)blocks), dim3(512), stream, args, g);
        return vt_check_launch("vt_conv2d(upblur, flat tiles)");
    }
    return VT_ERR_UNSUPPORTED;
}

template <typename T>
int launch_upflat(const ConvArgs& a, vt_stream stream) {
    return upflat_cn(a) == 16 ? launch_upflat_cn<T, 16>(a, stream) : launch_upflat_cn<T, 32>(a, stream);
}
