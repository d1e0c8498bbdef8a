// 3x3, Cin = Cout = 64, stride 1: the 512x512 level of the generator (convs.13, model/stylegan/model.py:364-370 on the
// channel map :422-432) -- 19 GFLOP on 67 MB of activations.  Included by conv_igemm.hip inside its anonymous namespace.
//
// The patch-resident tile kernel ran this layer at 250 TFLOP/s (72-77 us): K = 576 is nine K-steps, so every workgroup
// was mostly prologue (patch + weight latency), barrier-per-tap and epilogue.  This is the 1024x1024 level's kernel
// (conv3x3_c32_kernel) widened to 64 channels:
//   * 8 wavefronts per workgroup = 4 pixel quarters (4 rows of the 16x16-pixel tile) x 2 output-channel halves; a
//     wave's 32 x 9 x 64 weights (36 MFMA fragments, 144 registers) stay in REGISTERS for the life of the workgroup;
//   * workgroups are PERSISTENT and loop over tiles; the next tile's 18x18-pixel patch (128-byte pixel rows,
//     direct-to-LDS, zero fill = padding) is in flight while the current tile runs its 9 taps x 2 K-halves;
//   * LDS image: 24-pixel row pitch, slot ^ (pixel & 7) -- conflict-free for gfx950's ds_read_b128 lane groups when 16 consecutive
//     pixels are read from any start (tools/lds_bank_check.py);
//   * lean epilogue from registers: bias + LeakyReLU * gain -> bf16 NHWC (one 16-byte store per lane), optional fused
//     ToRGB (model.py:383-392): each wave reduces its 32 channels, the two channel halves meet in LDS.
#pragma once

template <typename T>
static bool c64_eligible(const ConvArgs& a, GldsArgs& g) {
    if (sizeof(T) != 2 || a.force_generic || a.transposed || a.in_scale) return false;
    if (a.taps != 9 || a.kw != 3 || a.stride != 1 || a.pad != 1 || a.dil != 1) return false;
    if (a.c0 != 64 || a.c1 != 0 || a.cout != 64 || a.phases != 1 || a.Ho != a.H || a.Wo != a.W) return false;
    if (a.splitk > 1 || a.stats_part || a.tile_stats || a.in_tile_stats || a.up_fir) return false;
    if (a.out_layout != VT_OUT_NHWC || a.out_f32 || !a.vec_store || a.resid || a.slope_vec || a.alpha_dev || a.post_relu) return false;
    if (a.act != VT_ACT_NONE && a.act != VT_ACT_LRELU) return false;
    const int64_t n0 = (int64_t)a.N * a.H * a.W * a.ld0 * 2;
    if (n0 >= (((int64_t)1 << 31) - 4096)) return false;
    g.nrec0 = (uint32_t)n0;
    g.nrec1 = g.nrecw = g.bias0 = g.bias1 = 0;
    return true;
}

// NWM = pixel-row groups of a workgroup (4 tile rows each): 4 -> 16x16-pixel tiles, 8 waves, one workgroup per CU;
// 2 -> 8x16-pixel tiles, 4 waves, TWO workgroups per CU whose load / compute / store phases interleave (default)
// PIPE = prefetch depth of the tap loop (see there)
template <typename T, int NWM, int PIPE>
__global__ void __launch_bounds__(NWM * 128, 2)   // 2 waves per SIMD (<= 256 registers)
conv3x3_c64_kernel(const ConvArgs p, const GldsArgs g) {
    static_assert(sizeof(T) == 2, "bf16 only (128-byte pixel rows)");
    constexpr int TH = 4 * NWM, TW = 16, NWAVES = 2 * NWM;
    constexpr int TM = 4, TN = 2;                                // per wave: 4 tile rows x 16 pixels, 2 x 16 channels
    // 18x18-pixel patch stored with a 24-pixel row pitch: every (tile row, tap row) offset is then a multiple of 8
    // pixels, the swizzle term (pixel & 7) depends on the lane and kx only, and the 72 fragment addresses of a tile
    // are 6 registers + immediate offsets (with an 18-pixel pitch LLVM kept 36+ loop-invariant addresses live next
    // to the 144 weight registers and spilled inside the tap loop)
    constexpr int PH = TH + 2, PW = TW + 2, PITCH = 24;
    constexpr int NLOAD = PH * (PITCH / 8);                      // 8-pixel (1 KB) wave loads per tile: 54 / 30
    constexpr int PA = (NLOAD + NWAVES - 1) / NWAVES;            // per wave: 7 / 8
    constexpr int A_BYTES = PA * NWAVES * 1024;                  // 56 / 32 KB per buffer
    __shared__ __attribute__((aligned(1024))) unsigned char smem[2 * A_BYTES];
    __shared__ float rgbx[NWM][TM][16][3];                       // ToRGB partials of the upper channel half
    __shared__ __attribute__((aligned(16))) float epc[4][64];    // epilogue constants: bias, 3 rows of ToRGB weights

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = vt_uniform(tid >> 6) & (NWAVES - 1);
    const int wm = wave >> 1, wn = wave & 1;
    const int q = lane >> 4, l15 = lane & 15;

    // ---- weights -> registers: fragment (tap, K-half, b): row n = wn*32 + perm(b*16 + l15), k = kh*32 + q*8 .. +7
    u128 wreg[9][2][TN];
    {
        const T* wg = (const T*)p.wgt;
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
            for (int kh = 0; kh < 2; ++kh)
#pragma unroll
                for (int b = 0; b < TN; ++b)   // fragment order: (b, lane group q) <-> channels 8q + 4b .. +3 of the half
                    wreg[t][kh][b] = ld128(wg + (int64_t)(wn * 32 + tile_row_channel<true>(b * 16 + l15)) * p.K + t * 64 +
                                           kh * 32 + q * 8);
    }
    const BufRsrc r0 = vt_make_rsrc(p.src0, g.nrec0);
    const int tiles_x = (p.Wo + TW - 1) / TW, tiles_y = (p.Ho + TH - 1) / TH;
    const int ntiles = p.N * tiles_x * tiles_y;
    const int lpix = lane >> 3, lslot = lane & 7;
    struct Rows {   // tile pixel (row-major, 16 wide) -> global GEMM row m, or -1
        int img, y0, x0, Ho, Wo;
        __device__ __forceinline__ int operator()(int row) const {
            const int oy = y0 + (row >> 4), ox = x0 + (row & 15);
            return (oy < Ho && ox < Wo) ? (img * Ho + oy) * Wo + ox : -1;
        }
    };

    auto issue = [&](int tile, int buf) {
        const int img = tile / (tiles_x * tiles_y);
        const int trem = tile - img * (tiles_x * tiles_y);
        const int y0 = (trem / tiles_x) * TH, x0 = (trem % tiles_x) * TW;
#pragma unroll
        for (int i = 0; i < PA; ++i) {
            const int li = i * NWAVES + wave;
            const int pr = li * 8 + lpix;             // linear patch pixel at the 24-pixel pitch
            const int py = pr / PITCH, px = pr - py * PITCH;
            const int iy = y0 - 1 + py, ix = x0 - 1 + px;
            const bool in = py < PH && px < PW && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
            const uint32_t pix = (uint32_t)((img * p.H + iy) * p.W + ix);
            // the lane that lands in physical slot s fetches the logical 16-byte chunk s ^ (pixel & 7)
            const uint32_t off = in ? pix * (uint32_t)(p.ld0 * 2) + ((uint32_t)(lslot ^ (pr & 7)) << 4) : GLDS_OOB;
            if (p.dbg != 33) vt_glds16(r0, smem + buf * A_BYTES + li * 1024, off, 0u);   // (33: ablation, no patch loads)
        }
    };

    // Bias and ToRGB weights (32 values per lane) cannot stay in registers next to the 144 weight registers (held
    // across the tap loop they spilled 67 dwords).  Round 2 re-read them from global memory per tile, each behind a
    // per-element "pointer or 0" select: hipcc branches around every such load and waits vmcnt(0) for it -- 32
    // dependent L2 round trips per TILE (the "+14-26 us of the fused ToRGB epilogue").  They are staged in LDS once per
    // workgroup instead and read back per tile with eight 16-byte LDS reads.
    const bool rgbf = p.rgb_w != nullptr;
    if (tid < 256) {
        const int row = tid >> 6, c = tid & 63;
        float v = 0.0f;
        if (row == 0) {
            if (p.bias) v = p.bias[c];
        } else if (rgbf) {
            v = to_f32(((const T*)p.rgb_w)[(row - 1) * 64 + c]);
        }
        epc[row][c] = v;   // (read after the tile loop's first barrier)
    }
    float rb0 = 0.0f, rb1 = 0.0f, rb2 = 0.0f;
    if (rgbf && p.rgb_bias) rb0 = p.rgb_bias[0], rb1 = p.rgb_bias[1], rb2 = p.rgb_bias[2];

    int tile = blockIdx.x;
    if (tile >= ntiles) return;
    issue(tile, 0);
    int buf = 0;
    for (; tile < ntiles; tile += gridDim.x) {
        const int next = tile + gridDim.x;
        if (next < ntiles) {
            issue(next, buf ^ 1);
            vt_glds_wait_n<PA>();      // this tile's patch has landed, the next one may be in flight
        } else {
            vt_glds_wait_n<0>();
        }
        vt_lds_barrier();
        const int img = tile / (tiles_x * tiles_y);
        const int trem = tile - img * (tiles_x * tiles_y);
        const int y0 = (trem / tiles_x) * TH, x0 = (trem % tiles_x) * TW;
        const Rows rowmap{img, y0, x0, p.Ho, p.Wo};
        const int HoWo = p.Ho * p.Wo;
        f32x4 acc[TM][TN];
#pragma unroll
        for (int a = 0; a < TM; ++a)
#pragma unroll
            for (int b = 0; b < TN; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
        const unsigned char* sa = smem + buf * A_BYTES;
        // software pipeline: the 4 pixel fragments of K-half h+1 are read while the 8 MFMAs of K-half h issue
        // (with the reads and MFMAs of one K-half fenced together the LDS latency was exposed 18 times per tile:
        // the tap loop measured 3.5 us per tile, 14 of the kernel's 44 us)
        auto read_a = [&](u128 (&fa)[TM], int h) {   // h = tap * 2 + kh, compile-time after unrolling
            const int t = h >> 1, kh = h & 1;
            const int ky = t / 3, kx = t - ky * 3;
#pragma unroll
            for (int a = 0; a < TM; ++a) {
                const int pr = (wm * TM + a + ky) * PITCH + kx + l15;
                fa[a] = ld128(sa + pr * 128 + (((kh * 4 + q) ^ ((kx + l15) & 7)) << 4));
            }
        };
        u128 fa[2][TM];
        if (PIPE) read_a(fa[0], 0);
#pragma unroll
        for (int h = 0; h < 18; ++h) {
            if (p.dbg == 32) break;   // ablation (tools/conv_bench.py, VT_RGB_ABLATE): no tap loop
            if (PIPE) {
                if (h + 1 < 18) read_a(fa[(h + 1) & 1], h + 1);
            } else {
                read_a(fa[h & 1], h);
            }
#pragma unroll
            for (int a = 0; a < TM; ++a)
#pragma unroll
                for (int b = 0; b < TN; ++b) Mma<T>::run(acc[a][b], wreg[h >> 1][h & 1][b], fa[h & 1][a]);
            vt_sched_fence();   // at most two K-halves' fragments live (the scheduler otherwise hoists every read)
        }
        // lean epilogue: bias + LeakyReLU * gain -> bf16 NHWC, optional fused ToRGB
        float r[TM][3];
        float rsd[TM][3];   // the up-sampled skip this tile adds to (lower channel half's lanes only)
#pragma unroll
        for (int a = 0; a < TM; ++a) rsd[a][0] = rsd[a][1] = rsd[a][2] = 0.0f;
        if (rgbf && p.rgb_resid && q == 0 && wn == 0) {
#pragma unroll
            for (int a = 0; a < TM; ++a) {
                const int m = rowmap(wm * (TM * 16) + a * 16 + l15);
                const int mm = m < 0 ? 0 : m;
                const int im = mm / HoWo;
                const int64_t o0 = (int64_t)im * 3 * HoWo + (mm - im * HoWo);
#pragma unroll
                for (int j = 0; j < 3; ++j) rsd[a][j] = p.rgb_resid[o0 + (int64_t)j * HoWo];
            }
        }
        {
            const float ga = p.gain_alpha;
            // this lane's 8 consecutive channels wn*32 + 8q .. +7: fragment 0 holds the first four, fragment 1 the rest
            float rwt[3][TN][4], bvr[TN][4];
            {
                const int ch = wn * 32 + q * 8;
                unpack16<float>(ld128(&epc[0][ch]), bvr[0]);
                unpack16<float>(ld128(&epc[0][ch + 4]), bvr[1]);
#pragma unroll
                for (int j = 0; j < 3; ++j) {
                    unpack16<float>(ld128(&epc[1 + j][ch]), rwt[j][0]);
                    unpack16<float>(ld128(&epc[1 + j][ch + 4]), rwt[j][1]);
                }
            }
#pragma unroll
            for (int a = 0; a < TM; ++a) {
                const int m = rowmap(wm * (TM * 16) + a * 16 + l15);
                float r0 = 0.f, r1 = 0.f, r2 = 0.f;
                float f[8];   // channels wn*32 + 8q .. +7 of this pixel: fragment 0 holds 8q..+3, fragment 1 8q+4..+7
#pragma unroll
                for (int b = 0; b < TN; ++b) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        float v = acc[a][b][i] + bvr[b][i];
                        if (p.act == VT_ACT_LRELU) v = (v > 0.0f) ? v : v * p.slope;
                        f[4 * b + i] = v * ga;
                    }
                    if (rgbf && p.dbg != 35) {   // (35: ablation, no dot products)
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            r0 += f[4 * b + i] * rwt[0][b][i];
                            r1 += f[4 * b + i] * rwt[1][b][i];
                            r2 += f[4 * b + i] * rwt[2][b][i];
                        }
                    }
                }
                if (m >= 0 && !(p.dbg == 31 && f[0] != 123.456f))   // (31: ablation, no activation stores)
                    st128((bf16_t*)p.out + (int64_t)m * p.ld_out + wn * 32 + q * 8, pack16<bf16_t>(f));
                if (rgbf) {
                    r0 += __shfl_xor(r0, 16, 64); r0 += __shfl_xor(r0, 32, 64);
                    r1 += __shfl_xor(r1, 16, 64); r1 += __shfl_xor(r1, 32, 64);
                    r2 += __shfl_xor(r2, 16, 64); r2 += __shfl_xor(r2, 32, 64);
                }
                r[a][0] = r0; r[a][1] = r1; r[a][2] = r2;
            }
        }
        if (rgbf) {   // the two channel halves meet in LDS: lower half + upper half, in that order
            if (wn == 1 && q == 0) {
#pragma unroll
                for (int a = 0; a < TM; ++a) {
                    rgbx[wm][a][l15][0] = r[a][0];
                    rgbx[wm][a][l15][1] = r[a][1];
                    rgbx[wm][a][l15][2] = r[a][2];
                }
            }
            if (p.dbg != 34) vt_lds_barrier();   // (34: ablation, no exchange barrier)
            if (wn == 0 && q == 0) {
#pragma unroll
                for (int a = 0; a < TM; ++a) {
                    const int m = rowmap(wm * (TM * 16) + a * 16 + l15);
                    if (m >= 0) {
                        const int im = m / HoWo;
                        const int64_t o0 = (int64_t)im * 3 * HoWo + (m - im * HoWo);
                        p.rgb_out[o0] = (r[a][0] + rgbx[wm][a][l15][0]) + rb0 + rsd[a][0];
                        p.rgb_out[o0 + HoWo] = (r[a][1] + rgbx[wm][a][l15][1]) + rb1 + rsd[a][1];
                        p.rgb_out[o0 + 2 * (int64_t)HoWo] = (r[a][2] + rgbx[wm][a][l15][2]) + rb2 + rsd[a][2];
                    }
                }
            }
        }
        vt_lds_barrier();   // every wave is done reading `buf` (and rgbx) before the next issue overwrites it
        buf ^= 1;
    }
}

template <typename T>
int launch_c64(const ConvArgs& a, const GldsArgs& g, vt_stream stream) {
    ConvArgs args = a;
    args.slab_perm = 0;
    args.splitk = 1;
    args.tiles_n = 1;
    static const int rows = [] {   // VT_C64_ROWS=16: the 16x16-pixel / 8-wave form (A/B)
        const char* e = getenv("VT_C64_ROWS");
        return e && atoi(e) == 16 ? 16 : 8;
    }();
    args.tiles_m = a.N * vt_cdiv(a.Ho, rows) * vt_cdiv(a.Wo, 16);
    const int cap = rows == 16 ? 256 : 512;                 // persistent: one 8-wave / two 4-wave workgroups per CU
    int blocks = args.tiles_m < cap ? args.tiles_m : cap;
    if (const char* e = getenv("VT_C32_BLOCKS")) {          // tests: force several tiles per workgroup
        const int v = atoi(e);
        if (v > 0 && v < blocks) blocks = v;
    }
    (void)sizeof(T);
    static const int pipe = [] {   // VT_C64_PIPE=1: software-pipelined fragment reads (A/B)
        const char* e = getenv("VT_C64_PIPE");
        return e ? atoi(e) : 0;
    }();
    if (rows == 16) {
        auto k = conv3x3_c64_kernel<bf16_t, 4, 0>;
        VT_LAUNCH(k, dim3((unsigned)blocks), dim3(512), stream, args, g);
    } else if (pipe) {
        auto k = conv3x3_c64_kernel<bf16_t, 2, 1>;
        VT_LAUNCH(k, dim3((unsigned)blocks), dim3(256), stream, args, g);
    } else {
        auto k = conv3x3_c64_kernel<bf16_t, 2, 0>;
        VT_LAUNCH(k, dim3((unsigned)blocks), dim3(256), stream, args, g);
    }
    return vt_check_launch("vt_conv2d(c64)");
}
