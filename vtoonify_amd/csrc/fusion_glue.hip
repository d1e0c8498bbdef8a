// Fusion.forward's gate in ONE launch (model/vtoonify.py:122-128, VToonify-D):
//     m_E  = tanh(relu(conv2_{2C->1, 3x3}(AdaIN(cat[f_G, |f_G - f_E|]))))        -> mask (n,1,h,w) fp32
//     fem  = [skip(3) | zeros | f_E * m_E]                                        -> the operand of the fusion conv
//                                                                                    (cat[f_G, f_E * m_E]) and of
//                                                                                    fusion_skip (cat[skip, f_E * m_E])
// given the AdaIN scale / shift of vt_instnorm_stats.  Before, this was vt_affine_apply (writes the normalised 2C
// tensor) -> thin conv -> vt_fusion_pack: three launches of 4-14 us each per level, and the 2C-channel normalised
// tensor went to HBM and back.
//
// The mask conv runs in the scatter form of conv_thin.hpp: d[p][tap] = sum_c W[tap][c] x'[p][c] over the 10x10 input
// patch of an 8x8-pixel tile on the MFMA, out[q] = sum_tap d[q + tap - 1][tap].  In that form the AdaIN affine is exact
// algebra on the operands: x' = s x + t  =>  d[p][tap] = sum_c (W[tap][c] s_c) x[p][c] + T[tap],  T[tap] = sum_c W[tap][c] t_c
// for pixels p inside the image (zero padding pads the NORMALISED tensor: outside pixels contribute nothing).  The
// weight fragment is scaled in registers, |f_G - f_E| is formed in registers from the two tensors, T is 9 dot products
// per workgroup.  The same workgroup then writes its 64 pixels of `fem`.
#include "vt_common.hpp"

namespace {

#ifdef VT_EMU
typedef emu_f32x4 f32x4;
#else
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
#endif

constexpr int FG_TW = 8, FG_PW = 10, FG_NPIX = 100, FG_NF = 7, FG_DLD = 20;
// 16 wavefronts per workgroup = 4 K-slices x 4 pixel-fragment groups (2, 2, 2, 1 fragments): at 32x32 / 64x64 pixels
// there are only 16 / 64 tiles, so the latency chain INSIDE a workgroup is the launch time -- with 4 waves (8 K-steps and
// 18 pack vectors per lane in sequence) the fused kernel took 32 us where the three separate launches took 21
constexpr int FG_KG = 4, FG_PG = 4, FG_FPG = 2, FG_NW = FG_KG * FG_PG;

struct FusionGateArgs {
    float* mask;
    void* fem;
    const void* f_g;
    const void* f_e;
    const float* scale;   // (n, 2c)
    const float* shift;
    const void* wgt;      // packed [1][9][2c]
    const float* bias;    // (1) or NULL
    const float* skip;    // (n,3,h,w)
    int ld_fem, ld_g, ld_e, n, h, w, c;
};

template <typename T>
struct Mma1;
template <>
struct Mma1<bf16_t> {
    static __device__ __forceinline__ void run(f32x4& acc, const u128& a, const u128& b) {
#ifdef VT_EMU
        emu_bf16x8 av, bv;
        memcpy(&av, &a, 16);
        memcpy(&bv, &b, 16);
        acc = emu_mfma_f32_16x16x32_bf16(av, bv, acc);
#else
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), acc, 0, 0, 0);
#endif
    }
};
template <>
struct Mma1<float> {
    static __device__ __forceinline__ void run(f32x4& acc, const u128& a, const u128& b) {
#ifdef VT_EMU
        acc = emu_mfma_f32_16x16x4f32(vt_u2f(a.x), vt_u2f(b.x), acc);
        acc = emu_mfma_f32_16x16x4f32(vt_u2f(a.y), vt_u2f(b.y), acc);
        acc = emu_mfma_f32_16x16x4f32(vt_u2f(a.z), vt_u2f(b.z), acc);
        acc = emu_mfma_f32_16x16x4f32(vt_u2f(a.w), vt_u2f(b.w), acc);
#else
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(vt_u2f(a.x), vt_u2f(b.x), acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(vt_u2f(a.y), vt_u2f(b.y), acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(vt_u2f(a.z), vt_u2f(b.z), acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(vt_u2f(a.w), vt_u2f(b.w), acc, 0, 0, 0);
#endif
    }
};

template <typename T>
__global__ void __launch_bounds__(FG_NW * 64) fusion_gate_kernel(const FusionGateArgs p) {
    constexpr int VEC = 16 / (int)sizeof(T);
    constexpr int KSTEP = 4 * VEC;
    __shared__ __attribute__((aligned(16))) float dpart[FG_KG][FG_NF * 16][FG_DLD];
    __shared__ float tpart[9][FG_NW];
    __shared__ float tsum[9];
    __shared__ float mtile[64];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = vt_uniform(tid >> 6) & (FG_NW - 1);
    const int kg = wave & (FG_KG - 1), pgp = wave >> 2;      // K-slice, pixel-fragment group
    const int q = lane >> 4, l15 = lane & 15;
    const int tiles_x = (p.w + FG_TW - 1) / FG_TW, tiles_y = (p.h + FG_TW - 1) / FG_TW;
    const int img = blockIdx.x / (tiles_x * tiles_y);
    const int trem = blockIdx.x - img * (tiles_x * tiles_y);
    const int y0 = (trem / tiles_x) * FG_TW, x0 = (trem % tiles_x) * FG_TW;
    const int C = p.c, C2 = 2 * p.c;
    const T* fg = (const T*)p.f_g;
    const T* fe = (const T*)p.f_e;
    const T* wg = (const T*)p.wgt;
    const float* sc = p.scale + (int64_t)img * C2;
    const float* sh = p.shift + (int64_t)img * C2;

    // T[tap] = sum_c W[tap][c] * shift[c]: every thread a strided share of the 2C channels, xor tree per wave, then
    // the four wave sums in wave order
    {
        float part[9];
#pragma unroll
        for (int t = 0; t < 9; ++t) part[t] = 0.0f;
        for (int ch = tid; ch < C2; ch += FG_NW * 64) {
            const float s = sh[ch];
#pragma unroll
            for (int t = 0; t < 9; ++t) part[t] += to_f32(wg[(int64_t)t * C2 + ch]) * s;
        }
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const float ws = wave_sum(part[t]);
            if (lane == 0) tpart[t][wave] = ws;
        }
    }

    // this lane's patch pixel of every fragment -> pixel index (or -1), for both tensors
    int64_t pix[FG_FPG];
#pragma unroll
    for (int f = 0; f < FG_FPG; ++f) {
        const int pp = (pgp * FG_FPG + f) * 16 + l15;   // (fragment 7 of the last group does not exist: pp >= 100)
        const int py = pp / FG_PW, px = pp - py * FG_PW;
        const int iy = y0 + py - 1, ix = x0 + px - 1;
        const bool in = pp < FG_NPIX && (unsigned)iy < (unsigned)p.h && (unsigned)ix < (unsigned)p.w;
        pix[f] = in ? (int64_t)(img * p.h + iy) * p.w + ix : -1;
    }
    f32x4 acc[FG_FPG];
#pragma unroll
    for (int f = 0; f < FG_FPG; ++f) acc[f] = f32x4{0.f, 0.f, 0.f, 0.f};
    const u128 zero = u128{0u, 0u, 0u, 0u};
    const int nk = C2 / KSTEP;
    constexpr int UNR = 3;   // K-steps whose loads are in flight together (4: 72 bytes of scratch under the 128-register cap)
    for (int k0 = kg; k0 < nk; k0 += FG_KG * UNR) {
        u128 ra[UNR][FG_FPG], rb[UNR][FG_FPG], rw[UNR];
        float rs[UNR][VEC];
        bool second[UNR], livek[UNR];
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            const int ks = k0 + u * FG_KG;
            livek[u] = ks < nk;
            const int kb = (livek[u] ? ks : k0) * KSTEP + q * VEC;   // first of this lane's VEC channels of cat[f_G, |f_G - f_E|]
            second[u] = kb >= C;
            const int kc = second[u] ? kb - C : kb;                  // channel in f_G / f_E
            rw[u] = (livek[u] && l15 < 9) ? ld128(wg + (int64_t)l15 * C2 + kb) : zero;   // row l15 = tap; rows 9..15 zero
#pragma unroll
            for (int k = 0; k < VEC; k += 4) unpack16<float>(ld128(sc + kb + k), rs[u] + k);
#pragma unroll
            for (int f = 0; f < FG_FPG; ++f) {
                const bool on = livek[u] && pix[f] >= 0;
                ra[u][f] = on ? ld128(fg + pix[f] * p.ld_g + kc) : zero;
                rb[u][f] = (on && second[u]) ? ld128(fe + pix[f] * p.ld_e + kc) : zero;
            }
        }
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            float wv[VEC];
            unpack16<T>(rw[u], wv);
#pragma unroll
            for (int k = 0; k < VEC; ++k) wv[k] *= rs[u][k];          // the AdaIN scale of the fragment's channels
            const u128 fw = pack16<T>(wv);
#pragma unroll
            for (int f = 0; f < FG_FPG; ++f) {
                u128 fa = ra[u][f];
                if (second[u]) {
                    float a[VEC], b[VEC];
                    unpack16<T>(ra[u][f], a);
                    unpack16<T>(rb[u][f], b);
#pragma unroll
                    for (int k = 0; k < VEC; ++k) a[k] = fabsf(a[k] - b[k]);
                    fa = pack16<T>(a);
                }
                Mma1<T>::run(acc[f], fw, fa);
            }
        }
    }
#pragma unroll
    for (int f = 0; f < FG_FPG; ++f) {
        const int fr = pgp * FG_FPG + f;
        if (fr < FG_NF) {
            float v4[4] = {acc[f][0], acc[f][1], acc[f][2], acc[f][3]};
            st128(&dpart[kg][fr * 16 + l15][q * 4], pack16<float>(v4));
        }
    }
    __syncthreads();
    if (tid < 9) {   // finish T[tap] in thread order (deterministic)
        float s = 0.0f;
#pragma unroll
        for (int i = 0; i < FG_NW; ++i) s += tpart[tid][i];
        tsum[tid] = s;
    }
    __syncthreads();
    const int oy = (tid & 63) / FG_TW, ox = (tid & 63) - oy * FG_TW;
    const int gy = y0 + oy, gx = x0 + ox;
    const bool live = gy < p.h && gx < p.w;
    if (tid < 64) {
        float s = 0.0f;
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int ky = tap / 3, kx = tap - ky * 3;
            const int pp = (oy + ky) * FG_PW + ox + kx;
            const int iy = gy + ky - 1, ix = gx + kx - 1;
            float t = 0.0f;
#pragma unroll
            for (int w = 0; w < FG_KG; ++w) t += dpart[w][pp][tap];
            if ((unsigned)iy < (unsigned)p.h && (unsigned)ix < (unsigned)p.w) t += tsum[tap];
            s += t;
        }
        const float m = tanhf(fmaxf(s + (p.bias ? p.bias[0] : 0.0f), 0.0f));
        mtile[tid] = m;
        if (live) p.mask[(int64_t)(img * p.h + gy) * p.w + gx] = m;
    }
    __syncthreads();
    if (!p.fem) return;
    // fem rows of the tile's 64 pixels: [skip(3) | zeros ... | f_E * m]
    const int hdr = p.ld_fem - C;
    const int hv = hdr / VEC, per_px = hv + C / VEC;
    T* fem = (T*)p.fem;
    const int64_t hw = (int64_t)p.h * p.w;
    constexpr int PB = 5;    // vectors per thread per round trip (a load -> store chain per vector serialised them)
    for (int i0 = tid; i0 < 64 * per_px; i0 += FG_NW * 64 * PB) {
        u128 val[PB];
        int64_t dst[PB];
#pragma unroll
        for (int b = 0; b < PB; ++b) {
            const int i = i0 + b * FG_NW * 64;
            dst[b] = -1;
            val[b] = zero;
            if (i >= 64 * per_px) continue;
            const int opix = i / per_px, v = i - opix * per_px;
            const int py = opix / FG_TW, px = opix - py * FG_TW;
            if (y0 + py >= p.h || x0 + px >= p.w) continue;
            const int64_t pg = (int64_t)(img * p.h + y0 + py) * p.w + x0 + px;
            if (v < hv) {
                const int64_t pl = (int64_t)(y0 + py) * p.w + x0 + px;
                float f[VEC];
#pragma unroll
                for (int k = 0; k < VEC; ++k) {
                    const int ch = v * VEC + k;
                    f[k] = (ch < 3) ? p.skip[((int64_t)img * 3 + ch) * hw + pl] : 0.0f;
                }
                val[b] = pack16<T>(f);
                dst[b] = pg * p.ld_fem + v * VEC;
            } else {
                val[b] = ld128(fe + pg * p.ld_e + (v - hv) * VEC);
                dst[b] = pg * p.ld_fem + hdr + (v - hv) * VEC;
            }
        }
#pragma unroll
        for (int b = 0; b < PB; ++b) {
            if (dst[b] < 0) continue;
            const int i = i0 + b * FG_NW * 64;
            const int opix = i / per_px, v = i - opix * per_px;
            if (v >= hv) {
                float f[VEC];
                unpack16<T>(val[b], f);
                const float m = mtile[opix];
#pragma unroll
                for (int k = 0; k < VEC; ++k) f[k] *= m;
                val[b] = pack16<T>(f);
            }
            st128(fem + dst[b], val[b]);
        }
    }
}

}  // namespace

extern "C" int vt_fusion_gate(float* mask, void* fem, int ld_fem, const void* f_g, int ld_g, const void* f_e, int ld_e,
                              const float* scale, const float* shift, const void* weight, const float* bias,
                              const float* skip, int n, int h, int w, int c, int dtype, vt_stream stream) {
    VT_REQUIRE(mask && f_g && f_e && scale && shift && weight, "vt_fusion_gate: null tensor");
    VT_REQUIRE(dtype == VT_F32 || dtype == VT_BF16, "vt_fusion_gate: dtype");
    const int vec = dtype == VT_F32 ? 4 : 8;
    VT_REQUIRE(n > 0 && h > 0 && w > 0 && c > 0 && (2 * c) % (4 * vec) == 0 && c % (4 * vec) == 0,
               "vt_fusion_gate: c must be a multiple of the MFMA K-step (%d)", 4 * vec);
    VT_REQUIRE(ld_g % vec == 0 && ld_e % vec == 0 && ld_g >= c && ld_e >= c, "vt_fusion_gate: pixel strides");
    VT_REQUIRE(!fem || (skip && ld_fem >= c + vec && (ld_fem - c) % vec == 0 && ld_fem % vec == 0),
               "vt_fusion_gate: fem needs skip and ld_fem = header + c (header a multiple of %d)", vec);
    VT_REQUIRE((uintptr_t)f_g % 16 == 0 && (uintptr_t)f_e % 16 == 0 && (uintptr_t)weight % 16 == 0 &&
               (!fem || (uintptr_t)fem % 16 == 0), "vt_fusion_gate: 16-byte aligned tensors");
    FusionGateArgs a;
    a.mask = mask; a.fem = fem; a.f_g = f_g; a.f_e = f_e; a.scale = scale; a.shift = shift; a.wgt = weight;
    a.bias = bias; a.skip = skip; a.ld_fem = ld_fem; a.ld_g = ld_g; a.ld_e = ld_e; a.n = n; a.h = h; a.w = w; a.c = c;
    const int64_t blocks = (int64_t)n * vt_cdiv(h, FG_TW) * vt_cdiv(w, FG_TW);
    VT_REQUIRE(blocks < ((int64_t)1 << 31), "vt_fusion_gate: too many tiles");
    if (dtype == VT_F32) {
        auto k = fusion_gate_kernel<float>;
        VT_LAUNCH(k, dim3((unsigned)blocks), dim3(FG_NW * 64), stream, a);
    } else {
        auto k = fusion_gate_kernel<bf16_t>;
        VT_LAUNCH(k, dim3((unsigned)blocks), dim3(FG_NW * 64), stream, a);
    }
    return vt_check_launch("vt_fusion_gate");
}
