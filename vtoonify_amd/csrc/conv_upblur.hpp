// Up-sampling StyledConv in ONE kernel at the reference's MAC count: conv_transpose2d(3x3, stride 2) followed by
// the 4x4 FIR blur (model/stylegan/model.py:273-286, Blur / upfirdn2d model.py:74-90), + bias + LeakyReLU * sqrt(2)
// (FusedLeakyReLU of StyledConv, model.py:364-370).  Included by conv_igemm.hip inside its anonymous namespace.
//
// The polyphase form this replaces folds the blur into the weights: four 3x3 filters per output channel,
// 36 MACs per input pixel and channel pair where the reference spends 9 (+ a channel-wise blur) -- the five
// up-sampling convs of a frame issued 193 GFLOP for 48 GFLOP of algorithmic work.  Here the blur stays a
// channel-wise stencil and runs on the vector ALUs out of LDS:
//
//   1. transposed conv on the matrix cores, 9 MACs: output pixel z[2I+pa, 2J+pb] of parity class (pa, pb) only
//      receives the taps W[a][b] with a = pa (mod 2), b = pb (mod 2):  tap (a, b) reads input pixel
//      (I - a/2, J - b/2).  So the 9 taps of W are 9 ordinary GEMM steps over a shifted input patch, each
//      accumulating into ONE of four accumulator sets (classes) -- a "quad" (I, J) yields 2x2 z pixels.
//      M = 16-wide rows of quads (one MFMA fragment each), N = CN output channels, K = Cin per tap; the patch of a
//      64-channel chunk is LDS-resident for its 9 taps, the weights stream per tap through a 3-stage ring
//      (same loader / counted-vmcnt pipeline as conv_patch_kernel).
//   2. the z tile ((2QY-1) x 31 pixels x CN channels, compute dtype) is parked in LDS over the dead patch / ring
//      buffers -- it never exists in HBM (134 MB fp32 per frame at the top level in the reference);
//   3. separable 4-tap blur per channel from LDS (the FIR is an outer product, make_kernel of a 1-D list,
//      model.py:21-29): every thread walks a column of the output tile with a sliding window of horizontally
//      filtered rows, adds the bias, LeakyReLU * gain, and stores 16-byte NHWC vectors.
//
// Output tile = 2(QY-2) x 28 pixels (a 1-quad halo each side feeds the blur: 1.37x recompute at QY = 12).
// z LDS image: 128-byte lines of LP = 128 / (CN * sizeof(T)) pixels, 16-byte slot s of line l stored at
// s ^ (l & 7): the fragment stores of step 2 and the column reads of step 3 are spread over the banks.
#pragma once

struct UpblurArgs {
    uint32_t nrec0, nrecw;
    int tiles_y, tiles_x;     // output tiles per image
};

// NW = wavefronts (3 quad rows each at QY = 3 NW).  4: 12 x 16 quads, two workgroups per CU.  8 (QY = 24): 24 x 16 quads,
// one workgroup per CU -- the 36 KB of weights a chunk brings in serve twice the pixels and the blur halo is recomputed
// for 1.25x instead of 1.37x of the output: the deep levels are bound by L2 -> LDS bytes (609 MB at 4.4 TB/s = the 137 us
// of the 512 -> 256 level at 4 frames), so bytes per MAC is what counts there.
template <typename T, int CN, int QY, int DB, int PERSIST, int LB2, int NW = 4>
__global__ void __launch_bounds__(NW * 64, (LB2 && NW == 4) ? 2 : 1)   // LB2: cap at 256 registers so that two 4-wave workgroups share a CU
conv_upblur_kernel(const ConvArgs p, const UpblurArgs g) {
    constexpr int ESZ = (int)sizeof(T);
    constexpr int VEC = 16 / ESZ;
    constexpr int BK = 8 * VEC;
    constexpr int NT = NW * 64;                   // threads
    constexpr int QX = 16, TY = 2 * (QY - 2), TX = 2 * (QX - 2);
    constexpr int MF = QY / NW;                   // quad rows (MFMA fragments) per wave
    constexpr int TN = CN / 16;                   // channel fragments per class
    constexpr bool PERM = (TN % 2 == 0);
    constexpr int PH = QY + 1, PW = QX + 1, PROWS = PH * PW;
    constexpr int PA = ((PROWS + 7) / 8 + NW - 1) / NW;      // patch loads per wave per chunk
    constexpr int WROWS = 9 * CN;                            // weight rows of a chunk: [tap][channel]
    constexpr int LBC = (WROWS / 8 + NW - 1) / NW;           // weight loads per wave per chunk
    constexpr int A_BYTES = PA * NW * 1024, B_BYTES = LBC * NW * 1024;
    constexpr int STAGE = A_BYTES + B_BYTES;
    constexpr int ZH = 2 * QY - 1, ZW = 2 * QX - 1;
    constexpr int PXB = CN * ESZ;                 // bytes of one z pixel
    constexpr int LP = 128 / PXB;                 // z pixels per 128-byte line
    constexpr int ZLINES = (ZW + LP - 1) / LP;
    constexpr int Z_BYTES = ZH * ZLINES * 128;
    constexpr int K_BYTES = (DB ? 2 : 1) * STAGE;
    // PERSIST (single-chunk layers, Cin = BK): the workgroup walks several tiles; the weights stay in LDS, the next
    // tile's patch is fetched during the blur of the current one, so the z tile gets its own region
    constexpr int Z_OFF = PERSIST ? STAGE : 0;
    constexpr int SMEM = PERSIST ? STAGE + Z_BYTES : (K_BYTES > Z_BYTES ? K_BYTES : Z_BYTES);
    static_assert(QY % NW == 0 && CN % 16 == 0 && 128 % PXB == 0 && CN % 8 == 0, "tile shape");
    static_assert(!(PERSIST && DB), "persistent form is single-chunk");
    static_assert(SMEM <= 160 * 1024, "LDS budget");
    static_assert(PA + LBC < 64, "vmcnt is 6 bits");

    __shared__ __attribute__((aligned(1024))) unsigned char smem[SMEM];
    auto sA = [&](int b) -> unsigned char* { return smem + b * STAGE; };
    auto sB = [&](int b) -> unsigned char* { return smem + b * STAGE + A_BYTES; };
    unsigned char* const zbase = smem + Z_OFF;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = vt_uniform(tid >> 6) & (NW - 1);
    const int q = lane >> 4, l15 = lane & 15, l7 = lane & 7;
    int tile_m, tile_n, split, tile_step;
    if (PERSIST) {
        tile_n = (int)(blockIdx.x % (unsigned)p.tiles_n);
        tile_m = (int)(blockIdx.x / (unsigned)p.tiles_n);
        tile_step = (int)(gridDim.x / (unsigned)p.tiles_n);
    } else {
        decode_block_pixel_major(p, tile_m, tile_n);   // (no K split in this kernel)
        split = 0;
        tile_step = 0x40000000;
    }
    const int per_img = g.tiles_y * g.tiles_x;
    const int n0 = tile_n * CN;
    const int OH = 2 * p.H, OW = 2 * p.W;

    // ---- loader state: patch pixel (py, px) = input pixel (I0 - 2 + py, J0 - 2 + px) --------------------------
    const int lrow = lane >> 3;
    const int jj = l7 ^ lrow;
    uint32_t pa0[PA];
    auto set_patch = [&](int tm) {
        const int im = tm / per_img, tr = tm - im * per_img;
        const int I0 = ((tr / g.tiles_x) * TY) / 2, J0 = ((tr % g.tiles_x) * TX) / 2;
#pragma unroll
        for (int i = 0; i < PA; ++i) {
            const int pr = (i * NW + wave) * 8 + lrow;
            const int py = pr / PW, px = pr - py * PW;
            const int iy = I0 - 2 + py, ix = J0 - 2 + px;
            const bool in = pr < PROWS && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
            const uint32_t pix = (uint32_t)((im * p.H + iy) * p.W + ix);
            pa0[i] = in ? pix * (uint32_t)(p.ld0 * ESZ) + jj * 16 : GLDS_OOB;
        }
    };
    // weights of a chunk: LDS row t*CN + r = tap t, tile row r (fragment order, see tile_row_channel)
    uint32_t woff[LBC];
#pragma unroll
    for (int i = 0; i < LBC; ++i) {
        const int row = (i * NW + wave) * 8 + lrow;
        const int tap = row / CN, r = row - tap * CN;
        const int n = n0 + tile_row_channel<PERM>(r);
        woff[i] = (row < WROWS && n < p.coutT) ? (uint32_t)n * (uint32_t)(p.K * ESZ) + (uint32_t)(tap * p.cin * ESZ) + jj * 16
                                                : GLDS_OOB;
    }
    const BufRsrc r0 = vt_make_rsrc(p.src0, g.nrec0);
    const BufRsrc rw = vt_make_rsrc(p.wgt, g.nrecw);
    const int nchunks = p.cin / BK;

    // one chunk = BK input channels: its patch AND all 9 taps of its weights are LDS-resident, so the 9 taps run
    // without a barrier (a 3-stage per-tap weight ring measured ~1 us per tap step: two taps in flight do not
    // cover the L2 latency).  DB: the next chunk loads into the other stage while this one computes (one
    // workgroup per CU, the deep layers); !DB: single stage, two workgroups per CU.
    auto issue = [&](int chunk, int st, bool with_weights) {
        const uint32_t so = (uint32_t)(chunk * BK * ESZ);
#pragma unroll
        for (int i = 0; i < PA; ++i) vt_glds16(r0, sA(st) + (i * NW + wave) * 1024, pa0[i], so);
        if (with_weights) {
#pragma unroll
            for (int i = 0; i < LBC; ++i) vt_glds16(rw, sB(st) + (i * NW + wave) * 1024, woff[i], so);
        }
    };

    // ---- blur constants (phase 3), fixed for the workgroup ----------------------------------------------------
    constexpr int NV = CN / VEC;                 // 16-byte channel vectors per pixel
    // (two row groups of 14 rows instead of four of 7 on the 8-wave forms -- 17 instead of 20 row iterations per SIMD -- measured
    // SLOWER: 227 against 190 us at the top level, profiles/r05_upblur_rows.txt: the phase is bound by the latency of its
    // LDS round trips, which fewer resident waves hide worse, not by its instruction count)
    constexpr int GROUPS = NT / (TX * NV) >= 1 ? NT / (TX * NV) : 1;
    constexpr int ROWS = TY / GROUPS;
    static_assert(TX * NV <= NT && TY % GROUPS == 0, "one thread per (column, channel vector, row group)");
    float kx[4], ky[4], bv[VEC], gpos[VEC], gneg[VEC];   // act(v) * gain = v * (v > 0 ? gpos : gneg)
    const int qv = tid % NV, col = (tid / NV) % TX, grp = tid / (NV * TX);
    const int nch = n0 + qv * VEC;
    // (computed AFTER the accumulation loop unless the workgroup is persistent: ~50 registers that would
    // otherwise be live across the MFMA loop and push the two-workgroups-per-CU form over its 256-register cap)
    auto blur_consts = [&]() {
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
        // (one wave-uniform branch per table, unconditional loads at clamped indices: a per-element "pointer or 0"
        // select costs a dependent L2 round trip per element)
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
    };
    if (PERSIST) blur_consts();
    const int r0w = grp * ROWS;
    // byte address of z pixel (row r0w, column col + t), channel vector qv: one register per tap, rows by immediates
    const unsigned char* zt[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const int zx = col + t;
        const int line = zx / LP, subp = zx - line * LP;
        const int s = (subp * PXB + qv * 16) >> 4;
        zt[t] = zbase + (r0w * ZLINES + line) * 128 + ((s ^ (line & 7)) << 4);
    }
    const int64_t ostep = (int64_t)OW * p.ld_out;
    const bool full = nch + VEC <= p.coutT;

    if (tile_m >= p.tiles_m) return;
    set_patch(tile_m);
    issue(0, 0, true);
    // (not a `for` over tiles: with !PERSIST the body runs exactly once and the compiler must SEE that -- as a loop
    // it hoisted every loop-invariant LDS address out of it and needed 480 registers)
    for (;;) {
        const int img = tile_m / per_img;
        const int trem = tile_m - img * per_img;
        const int u0 = (trem / g.tiles_x) * TY, v0 = (trem % g.tiles_x) * TX;   // first output pixel of the tile

        f32x4 acc[4][MF][TN];
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int m = 0; m < MF; ++m)
#pragma unroll
                for (int n = 0; n < TN; ++n) acc[c][m][n] = f32x4{0.f, 0.f, 0.f, 0.f};

        // ---- 1. transposed conv: 9 taps per chunk, accumulator set = parity class of the tap -------------------
        for (int chunk = 0; chunk < nchunks; ++chunk) {
            const int st = DB ? (chunk & 1) : 0;
            // (DB: stage st^1 was read by chunk - 1: every wave is past it once it has reached this barrier)
            vt_glds_wait_n<0>();
            vt_lds_barrier();
            if (DB && chunk + 1 < nchunks) issue(chunk + 1, st ^ 1, true);
            const unsigned char* pa = sA(st);
            const unsigned char* pb = sB(st);
            if constexpr (is_x3<T>::value) {
                // f32x3 (conv_igemm.hip): both 16-byte halves of a lane group's row at once -- chunks q and 4+q of pixels and
                // weights alike -- split into bf16 head + remainder, three bf16 MFMAs per product
#pragma unroll
                for (int sh = 0; sh < 4; ++sh) {
                    const int di = sh >> 1, dj = sh & 1;
                    vt_sched_fence();
                    u128 ah[MF], al[MF];
#pragma unroll
                    for (int m = 0; m < MF; ++m) {
                        const int pr = (wave * MF + m + 1 - di) * PW + (1 - dj) + l15;
                        x3_split(ld128(pa + pr * 128 + ((q ^ (pr & 7)) << 4)), ld128(pa + pr * 128 + (((4 + q) ^ (pr & 7)) << 4)),
                                 ah[m], al[m]);
                    }
#pragma unroll
                    for (int ta = 2 * di; ta < (di ? 3 : 2); ++ta)
#pragma unroll
                        for (int tb = 2 * dj; tb < (dj ? 3 : 2); ++tb) {
                            const int tap = ta * 3 + tb, cls = (ta & 1) * 2 + (tb & 1);
                            u128 bh[TN], bl[TN];
#pragma unroll
                            for (int n = 0; n < TN; ++n) {
                                const unsigned char* row = pb + (tap * CN + n * 16 + l15) * 128;
                                x3_split(ld128(row + ((q ^ l7) << 4)), ld128(row + (((4 + q) ^ l7) << 4)), bh[n], bl[n]);
                            }
#pragma unroll
                            for (int m = 0; m < MF; ++m)
#pragma unroll
                                for (int n = 0; n < TN; ++n) Mma<f32x3_t>::run3(acc[cls][m][n], bh[n], bl[n], ah[m], al[m]);
                        }
                }
            } else {
#pragma unroll
            for (int sub = 0; sub < 2; ++sub) {
                const int slot = sub * 4 + q;
                // taps grouped by input shift (a/2, b/2): the four taps (0|1, 0|1) share one set of pixel fragments
#pragma unroll
                for (int sh = 0; sh < 4; ++sh) {
                    const int di = sh >> 1, dj = sh & 1;
                    vt_sched_fence();   // one shift group's fragments live at a time (the scheduler otherwise hoists
                                        // every LDS read of the chunk: > 256 registers, spills)
                    u128 fa[MF];
#pragma unroll
                    for (int m = 0; m < MF; ++m) {
                        const int pr = (wave * MF + m + 1 - di) * PW + (1 - dj) + l15;
                        fa[m] = ld128(pa + pr * 128 + ((slot ^ (pr & 7)) << 4));
                    }
#pragma unroll
                    for (int ta = 2 * di; ta < (di ? 3 : 2); ++ta)
#pragma unroll
                        for (int tb = 2 * dj; tb < (dj ? 3 : 2); ++tb) {
                            const int tap = ta * 3 + tb, cls = (ta & 1) * 2 + (tb & 1);
                            u128 fb[TN];
#pragma unroll
                            for (int n = 0; n < TN; ++n)
                                fb[n] = ld128(pb + (tap * CN + n * 16 + l15) * 128 + ((slot ^ l7) << 4));
#pragma unroll
                            for (int m = 0; m < MF; ++m)
#pragma unroll
                                for (int n = 0; n < TN; ++n) Mma<T>::run(acc[cls][m][n], fb[n], fa[m]);
                        }
                }
            }
            }
            if (!DB && chunk + 1 < nchunks) {
                vt_lds_barrier();   // every wave is done reading the stage
                issue(chunk + 1, 0, true);
            }
        }
        __syncthreads();   // the patch (and, unless PERSIST, the whole stage) is dead
        if (PERSIST && tile_m + tile_step < p.tiles_m) {   // next tile's patch flies during the z tile + blur
            set_patch(tile_m + tile_step);
            issue(0, 0, false);
        }
        // ---- 2. z tile -> LDS.  quad (qy, l15) class (pa, pb) = z pixel (2qy + pa - 1, 2 l15 + pb - 1) of the tile --
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int zpa = c >> 1, zpb = c & 1;
#pragma unroll
            for (int m = 0; m < MF; ++m) {
                const int zy = 2 * (wave * MF + m) + zpa - 1, zx = 2 * l15 + zpb - 1;
                if (zy < 0 || zx < 0) continue;
                const int line = zx / LP, subp = zx - line * LP;
                unsigned char* zl = zbase + (zy * ZLINES + line) * 128;
#pragma unroll
                for (int n = 0; n < TN; ++n) {
                    const int ch = frag_channel<PERM>(n, q);                // first of this lane's 4 channels
                    const int b0 = subp * PXB + ch * ESZ;                   // byte offset inside the line
                    const int phys = ((b0 >> 4) ^ (line & 7)) << 4;
                    float f[4] = {acc[c][m][n][0], acc[c][m][n][1], acc[c][m][n][2], acc[c][m][n][3]};
                    if (ESZ == 2) {
                        u64v v;
                        v.x = pack_bf16x2(f[0], f[1]);
                        v.y = pack_bf16x2(f[2], f[3]);
                        *reinterpret_cast<u64v*>(zl + phys + (b0 & 15)) = v;
                    } else {
                        st128(zl + phys, pack16<float>(f));
                    }
                }
            }
        }
        __syncthreads();

        // ---- 3. blur + bias + activation.  out(u, v) = sum_{p,q} z[u + p][v + q] * ky[p] * kx[q] / S in tile coordinates
        // (upfirdn2d pad (1,1): z row u + p - 1 of the image; the tile's z rows start at image row u0 - 1).
        // Pure vector-ALU work (70 % of the kernel before it was trimmed): explicit fmaf (the build runs
        // -ffp-contract=off), addresses as one register per tap + immediates, bias folded into the vertical sum,
        // activation as one select.  Row rr of the horizontally filtered tile feeds the four output rows
        // rr-3 .. rr: four running sums per channel (ring slot = output row & 3), so no arithmetic chain is longer
        // than three dependent operations (dependent v_pk_fma chains ran at 5-6 cycles per instruction).  The row
        // loop is a REAL loop of 4-row bodies (static ring slots inside): fully unrolled the compiler hoisted
        // every LDS read and needed 472 registers.
        if (!PERSIST) blur_consts();
        if (grp < GROUPS) {
            const int ov = v0 + col;
            // running sums by age: o1 / o2 / o3 have received 1 / 2 / 3 vertical taps; the row that completes o3
            // finishes output row rr - 3.  The three sums keep their REGISTERS and rotate their ROLES (round 5): a row adds its
            // taps in place (o2 += h ky2 becomes the new o3, o1 += h ky1 the new o2, the finished o3 restarts as h ky0 + bias),
            // and three rows later every variable is back in its role -- the one-row loop body moved all 24 sums every row (26
            // v_mov per 170 instructions; unrolled 4 rows with static slots it kept four rows of unpacked pixels live and
            // spilled).  Same operands in the same order: the same bits.  Worth 1 % of the kernel (same box: 571 -> 565 us over
            // the five levels): the phase waits for LDS, it does not run out of issue slots.
            float oa[VEC], ob[VEC], oc[VEC];
#pragma unroll
            for (int k = 0; k < VEC; ++k) oa[k] = ob[k] = oc[k] = 0.0f;
            T* o = (T*)p.out + (((int64_t)img * OH + u0 + r0w - 3) * OW + ov) * p.ld_out + nch;   // row (rr - 3) of this thread
            const bool colok = ov < OW;
            const unsigned char *z0 = zt[0], *z1 = zt[1], *z2 = zt[2], *z3 = zt[3];
            int rr = 0;
            // ... and in the persistent 8-wave form (the 1024^2 level) the four LDS reads of a row are issued ONE ROW AHEAD, into
            // the other of two register sets (roles again: a call consumes `cur`, fills `nxt`): 183 -> 171 us at 4 frames, same
            // box; the 4-wave and multi-chunk forms lose 1-3 us to it (two resident workgroups / a register-hungrier K loop
            // hide the round trip already) and read their row where they use it (profiles/r05_upblur_rows.txt).
            constexpr bool PF = PERSIST != 0;
            u128 ra[4], rb[4];
            if constexpr (PF) ra[0] = ld128(z0), ra[1] = ld128(z1), ra[2] = ld128(z2), ra[3] = ld128(z3);
            auto row = [&](float (&o1)[VEC], float (&o2)[VEC], float (&o3)[VEC], u128 (&cur)[4], u128 (&nxt)[4], bool more) {
                vt_sched_fence();   // one row's unpacked pixels live at a time
                if constexpr (!PF) cur[0] = ld128(z0), cur[1] = ld128(z1), cur[2] = ld128(z2), cur[3] = ld128(z3);
                z0 += ZLINES * 128; z1 += ZLINES * 128; z2 += ZLINES * 128; z3 += ZLINES * 128;
                if constexpr (PF) {
                    if (more) nxt[0] = ld128(z0), nxt[1] = ld128(z1), nxt[2] = ld128(z2), nxt[3] = ld128(z3);
                }
                float f0[VEC], f1[VEC], f2[VEC], f3[VEC], h[VEC];
                unpack16<T>(cur[0], f0);
                unpack16<T>(cur[1], f1);
                unpack16<T>(cur[2], f2);
                unpack16<T>(cur[3], f3);
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
            // call number c of a period of six: sums in role rotation c % 3, register sets alternating
            auto step = [&](auto cc, bool more) {
                constexpr int c = decltype(cc)::value;
                if constexpr (c % 6 == 0) row(oa, ob, oc, ra, rb, more);
                else if constexpr (c % 6 == 1) row(oc, oa, ob, rb, ra, more);
                else if constexpr (c % 6 == 2) row(ob, oc, oa, ra, rb, more);
                else if constexpr (c % 6 == 3) row(oa, ob, oc, rb, ra, more);
                else if constexpr (c % 6 == 4) row(oc, oa, ob, ra, rb, more);
                else row(ob, oc, oa, rb, ra, more);
            };
            constexpr int NROW = ROWS + 3;
#pragma unroll 1
            for (int t = 0; t < NROW / 6; ++t)
                vt_static_for<6>([&](auto cc) { step(cc, t * 6 + decltype(cc)::value + 1 < NROW); });
            vt_static_for<NROW % 6>([&](auto cc) { step(cc, (NROW / 6) * 6 + decltype(cc)::value + 1 < NROW); });
        }
        // (PERSIST: the wait + barrier at the top of the next tile's chunk loop also fences the z tile)
        if (!PERSIST) break;
        tile_m += tile_step;
        if (tile_m >= p.tiles_m) break;
    }
}

template <typename T>
static bool upblur_eligible(const ConvArgs& a, UpblurArgs& g, int ty, int tx) {
    constexpr int ESZ = (int)sizeof(T);
    constexpr int BK = 8 * (16 / ESZ);
    if (!a.up_fir || a.transposed || a.in_scale || a.rgb_w || a.resid || a.c1 != 0 || a.post_relu) return false;
    if (a.taps != 9 || a.kw != 3 || a.phases != 1 || a.Ho != 2 * a.H || a.Wo != 2 * a.W) return false;
    if (a.cin % BK != 0 || a.coutT % 8 != 0) return false;
    if (a.act != VT_ACT_NONE && a.act != VT_ACT_LRELU) return false;
    if (a.out_layout != VT_OUT_NHWC || a.out_f32 != (ESZ == 4) || !a.vec_store) return false;
    const int64_t lim = ((int64_t)1 << 31) - 4096;
    const int64_t n0 = (int64_t)a.N * a.H * a.W * a.ld0 * ESZ, nw = (int64_t)a.coutT * a.K * ESZ;
    if (n0 >= lim || nw >= lim) return false;
    g.nrec0 = (uint32_t)n0;
    g.nrecw = (uint32_t)nw;
    g.tiles_y = vt_cdiv(2 * a.H, ty);
    g.tiles_x = vt_cdiv(2 * a.W, tx);
    return true;
}

template <typename T, int CN, int QY, int DB, int PERSIST, int LB2, int NW = 4>
int launch_upblur(const ConvArgs& a, vt_stream stream) {
    UpblurArgs g;
    if (!upblur_eligible<T>(a, g, 2 * (QY - 2), 28)) {
        vt_set_error("vt_conv2d: up_fir (conv_transpose + blur) form not supported for this convolution");
        return VT_ERR_UNSUPPORTED;
    }
    ConvArgs args = a;
    args.splitk = 1;
    args.kps = 0;
    args.slab_perm = 0;
    args.tiles_n = vt_cdiv(a.coutT, CN);
    args.tiles_m = a.N * g.tiles_y * g.tiles_x;
    int64_t blocks = (int64_t)args.tiles_m * args.tiles_n;
    if (blocks >= ((int64_t)1 << 31)) {
        vt_set_error("vt_conv2d: too many tiles");
        return VT_ERR_ARG;
    }
    if (PERSIST) {   // one workgroup per CU (256 CUs), each walks tiles_m / (grid / tiles_n) tiles of one channel tile
        const char* e = getenv("VT_UPBLUR_WGS");   // tests: few workgroups = several tiles each
        const int pwg = e && atoi(e) > 0 ? atoi(e) : 256;
        int per_n = pwg / args.tiles_n;
        if (per_n < 1) per_n = 1;
        if (per_n > args.tiles_m) per_n = args.tiles_m;
        blocks = (int64_t)per_n * args.tiles_n;
    }
    if constexpr (sizeof(T) == 4 && !is_x3<T>::value) {
        if (a.x3) {   // f32x3 instance of the same tile
            auto k = conv_upblur_kernel<f32x3_t, CN, QY, DB, PERSIST, LB2, NW>;
            VT_LAUNCH(k, dim3((unsigned)blocks), dim3(NW * 64), stream, args, g);
            return vt_check_launch("vt_conv2d(upblur, f32x3)");
        }
    }
    auto k = conv_upblur_kernel<T, CN, QY, DB, PERSIST, LB2, NW>;
    VT_LAUNCH(k, dim3((unsigned)blocks), dim3(NW * 64), stream, args, g);
    return vt_check_launch("vt_conv2d(upblur)");
}
