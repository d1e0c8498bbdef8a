// Implicit-GEMM convolution on the gfx950 matrix cores -- every dense contraction of the
// VToonify frame (reference: F.conv2d / F.conv_transpose2d behind
// model/stylegan/op/conv2d_gradfix.py:22-75 and the nn.Conv2d modules of
// model/vtoonify.py:96-97,111-113,162-198; cuDNN in the reference).
//
// GEMM view (NHWC activations, weights packed [cout][tap][cin]):
//     D[m, n] = sum_k A[m, k] * B[n, k]
//     m = output pixel (n_img, oy, ox)          M = N*Ho*Wo
//     n = output channel (x4 polyphase filters) Ncols = phases*cout
//     k = (tap, cin)                            K = kh*kw*(c0+c1)
// Both operands are K-contiguous, so every lane feeds the MFMA straight from 16-byte LDS
// reads: bf16 uses v_mfma_f32_16x16x32_bf16 (8 consecutive k per lane), fp32 uses
// v_mfma_f32_16x16x4_f32 x4 on the same 16 bytes (4 consecutive k per lane; the k-slot
// permutation is identical for A and B so the dot product is unchanged).  fp32 mode is
// exact fp32 (fmaf chain) and exists for parity against the fp32 reference.
//
// Workgroup = 256 threads = 4 wavefronts computing a BM x BN tile; K advances in steps of
// 128 bytes per row (64 bf16 / 32 fp32).  Tiles are register-staged (global -> VGPR ->
// LDS), double buffered, one barrier per K-step: the global loads of step t+1 are issued
// before the MFMAs of step t.  LDS rows are 128 B with the 16-byte slot XOR-swizzled by
// (row & 7): conflict-free for the ds_read_b128 lane groups and for the 8-lane
// ds_write_b128 groups (MI355X_MICROARCH.md, LDS table).
//
// The im2col gather, zero padding, channel concatenation of two sources (torch.cat in
// Fusion, model/vtoonify.py:125-127), the optional per-(n,cin) AdaIN affine
// (model/dualstylegan.py:16-21) and the K tail are all resolved in the loader; the
// epilogue (bias, LeakyReLU / relu-tanh, gain, style-degree scaling, residual add,
// pixel-shuffle for the polyphase up-sampling conv, NHWC or planar NCHW stores) runs
// from an LDS-staged fp32 tile so every global store is a full 16-byte vector.
#include <stdlib.h>
#include <utility>

#include "vt_common.hpp"

namespace {

#ifdef VT_EMU
typedef emu_f32x4 f32x4;
#else
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
#endif

// fp32 slab row length: whole 32-channel fragment groups, so that fragment-order columns stay in the row
static inline int slab_ld(int cout_total) { return (cout_total + 31) / 32 * 32; }
constexpr int64_t VT_TICKET_BYTES = 16384;  // split-K arrival counters at the head of the workspace: 4096 tiles

struct ConvArgs {
    const void* src0;
    const void* src1;
    const void* wgt;
    const float* bias;
    const float* slope_vec;
    const void* rgb_w;       // fused ToRGB: packed [3][cout] weights (compute dtype) or NULL
    const float* rgb_bias;
    const float* rgb_resid;
    float* rgb_out;
    const float* alpha_dev;
    const float* in_scale;
    const float* in_shift;
    const void* resid;
    void* out;
    int c0, c1, ld0, ld1, cin;
    int N, H, W, Ho, Wo;
    int coutT, cout, K, taps, kw;
    int stride, pad, pad_x, dil, transposed, phases;   // pad = vertical, pad_x = horizontal (vt_conv_desc.pad_w_p1)
    int act;
    float slope, gain_alpha, beta;
    int ld_res, ld_out, out_layout, out_f32, vec_store;
    int M, tiles_n, tiles_m;
    // split-K: each tile's K range is cut into `splitk` slices of `kps` K-steps; slice s
    // writes raw fp32 accumulators to partial[s][m][ldp] and conv_splitk_reduce finishes.
    int splitk, kps, ldp;
    int slab_perm;      // slab columns are in fragment order (tile_row_channel), set by the launcher
    float* partial;
    int* tickets;       // per-tile arrival counters (zero between launches) or NULL = two-pass split-K
    int phase;          // 0 = slices + reduce, 1 = slices only, 2 = reduce only (two-pass split-K)
    int force_generic;  // tile_hint flag: run the register-staged kernel even when glds applies
    StatRec* stats_part;  // InstanceNorm chunk records of the output (vt_conv_desc.stats_part) or NULL
    int post_relu;      // vt_conv_desc.post_relu: max(., 0) after the residual add
    const void* wstream;  // vt_conv_desc.weight_stream: fragment-stream image of the weights (whole-K kernel) or NULL
    float* tile_stats;          // whole-K kernel: {mean, M2} per (image, tile, channel) of the output, or NULL
    const float* in_tile_stats; // whole-K kernel: records of the input tensor -> AdaIN prologue, or NULL
    int in_stats_dil;
    const float* in_gb;
    int in_ld_gb;
    const float* up_fir;        // conv_transpose2d(stride 2) + blur form (conv_upblur.hpp) or NULL
    int rgb_only;               // vt_conv_desc.rgb_only: the C-channel output is not stored (fused ToRGB only)
    int in_absdiff;             // vt_conv_desc.in_absdiff: input = cat[src0, |src0 - src1|] (thin kernel)
    int x3;             // vt_conv_desc.dtype == VT_F32X3: fp32 tensors, products as three bf16 MFMAs where the instance exists
    int blk_pm, blk_cn;  // decode_block_2d: pixel tiles x channel tiles of the block of tiles one XCD owns (0 = channel-major order)
};

template <typename T>
struct Mma;
template <>
struct Mma<bf16_t> {
    static __device__ __forceinline__ void run(f32x4& acc, const u128& a, const u128& b) {
#ifdef VT_EMU
        emu_bf16x8 av, bv;
        memcpy(&av, &a, 16);
        memcpy(&bv, &b, 16);
        acc = emu_mfma_f32_16x16x32_bf16(av, bv, acc);
#else
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a),
                                                      __builtin_bit_cast(bf16x8_t, b), acc, 0, 0, 0);
#endif
    }
};
template <>
struct Mma<float> {
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

// ---------------------------------------------------------------------------------------
// f32x3: fp32 operands on the bf16 matrix cores (round 4).  The reference computes fp32
// (model/stylegan/op/upfirdn2d_kernel.cu:311, fused_bias_act_kernel.cu:96, cuDNN fp32 convs); the exact-fp32 MFMA
// (v_mfma_f32_16x16x4_f32) runs at 1/16 of the bf16 rate.  Here every fp32 operand x is split ON THE FLY, in the
// fragment registers, into its bf16 head h = rne_bf16(x) and the bf16-rounded remainder l = rne_bf16(x - h)
// (|x - h - l| <= 2^-17 |x|), and a product a*b becomes ah*bh + al*bh + ah*bl on v_mfma_f32_16x16x32_bf16 with fp32
// accumulation (the dropped al*bl is <= 2^-16 |a*b|): 3 matrix instructions of 16 cycles per 32-channel row instead
// of 8 of 32 cycles.  Tensors and weights stay plain fp32 in memory -- the mode is a property of the launch
// (vt_conv_desc.dtype = VT_F32X3), any conv whose kernel instance has no f32x3 form simply runs exact fp32.
// Lane group q of a fragment takes the 16-byte chunks q and 4+q of a 128-byte row (channels 4q..4q+3 and
// 16+4q..16+4q+3) for BOTH operands, so the K positions pair up whatever the order.
// End-to-end against the fp32 oracle: 4e-5 of max|y| (bar 1e-4; exact-fp32 MFMA: 5e-6) -- tests/test_engine.py.
// ---------------------------------------------------------------------------------------
template <typename T>
struct is_x3 {
    static constexpr bool value = false;
};
template <>
struct is_x3<f32x3_t> {
    static constexpr bool value = true;
};
__device__ __forceinline__ void x3_split(const u128& x, const u128& y, u128& hi, u128& lo) {
    auto pair = [](uint32_t a, uint32_t b, uint32_t& h, uint32_t& l) {
        const float fa = vt_u2f(a), fb = vt_u2f(b);
        h = pack_bf16x2(fa, fb);
        l = pack_bf16x2(fa - vt_u2f(h << 16), fb - vt_u2f(h & 0xffff0000u));
    };
    pair(x.x, x.y, hi.x, lo.x);
    pair(x.z, x.w, hi.y, lo.y);
    pair(y.x, y.y, hi.z, lo.z);
    pair(y.z, y.w, hi.w, lo.w);
}
template <>
struct Mma<f32x3_t> {   // on already split operands: (weights head, weights remainder) x (pixels head, pixels remainder)
    static __device__ __forceinline__ void run3(f32x4& acc, const u128& wh, const u128& wl, const u128& ah, const u128& al) {
        Mma<bf16_t>::run(acc, wh, ah);
        Mma<bf16_t>::run(acc, wh, al);
        Mma<bf16_t>::run(acc, wl, ah);
    }
};
// One 128-byte K row of a wave tile: acc[a][b] += W_b . A_a over the row's channels.  `pa(a, sub)` / `pb(b, sub)` give
// the LDS address of the 16-byte chunk (sub * 4 + q) of pixel fragment a / weight fragment b.  bf16 / fp32: the two
// half-row steps of the kernels' original loops, verbatim; f32x3: both halves, split, three products.
template <typename T, int TM, int TN, typename PA, typename PB>
__device__ __forceinline__ void mma_row(f32x4 (&acc)[TM][TN], PA&& pa, PB&& pb) {
    if constexpr (is_x3<T>::value) {
        u128 ah[TM], al[TM], bh[TN], bl[TN];
#pragma unroll
        for (int a = 0; a < TM; ++a) x3_split(ld128(pa(a, 0)), ld128(pa(a, 1)), ah[a], al[a]);
#pragma unroll
        for (int b = 0; b < TN; ++b) x3_split(ld128(pb(b, 0)), ld128(pb(b, 1)), bh[b], bl[b]);
#pragma unroll
        for (int a = 0; a < TM; ++a)
#pragma unroll
            for (int b = 0; b < TN; ++b) Mma<f32x3_t>::run3(acc[a][b], bh[b], bl[b], ah[a], al[a]);
    } else {
#pragma unroll
        for (int sub = 0; sub < 2; ++sub) {
            u128 fa[TM], fb[TN];
#pragma unroll
            for (int a = 0; a < TM; ++a) fa[a] = ld128(pa(a, sub));
#pragma unroll
            for (int b = 0; b < TN; ++b) fb[b] = ld128(pb(b, sub));
#pragma unroll
            for (int a = 0; a < TM; ++a)
#pragma unroll
                for (int b = 0; b < TN; ++b) Mma<T>::run(acc[a][b], fb[b], fa[a]);
        }
    }
}

// x' = x * scale + shift on one 16-byte vector (AdaIN prologue)
template <typename T>
__device__ __forceinline__ u128 affine16(u128 v, const float* sc, const float* sh) {
    constexpr int VEC = 16 / sizeof(T);
    float f[VEC];
    unpack16<T>(v, f);
#pragma unroll
    for (int i = 0; i < VEC; ++i) f[i] = f[i] * sc[i] + sh[i];
    return pack16<T>(f);
}


// ReLU after the residual add (ResNet BasicBlock, model/bisenet/resnet.py:36-48): out = max(v + beta*resid, 0)
__device__ __forceinline__ float post_act(const ConvArgs& p, float x) { return p.post_relu ? fmaxf(x, 0.0f) : x; }
template <int N>
__device__ __forceinline__ void post_act_n(const ConvArgs& p, float* f) {
    if (p.post_relu) {
#pragma unroll
        for (int i = 0; i < N; ++i) f[i] = fmaxf(f[i], 0.0f);
    }
}

// Finished values (bias/activation/gain applied) of 8 consecutive output columns n..n+7 of
// GEMM row m -> NHWC store with the optional residual add and the polyphase pixel shuffle.
__device__ __forceinline__ void store_nhwc8(const ConvArgs& p, int m, int n, float* f) {
    const int HoWo = p.Ho * p.Wo;
    int64_t opix = m;
    int co = n;
    if (p.phases > 1) {
        const int ph = n / p.cout;
        co = n - ph * p.cout;
        const int img = m / HoWo;
        const int rem = m - img * HoWo;
        const int oy = rem / p.Wo, ox = rem - oy * p.Wo;
        opix = ((int64_t)img * (2 * p.Ho) + 2 * oy + (ph >> 1)) * (2 * p.Wo) + 2 * ox + (ph & 1);
    }
    const int lim = (p.phases > 1) ? p.cout : p.coutT;
    const int nvalid = (lim - co) < 8 ? (lim - co) : 8;
    if (p.out_f32) {
        float* o = (float*)p.out + opix * p.ld_out + co;
        const float* rs = p.resid ? (const float*)p.resid + opix * p.ld_res + co : nullptr;
        if (p.vec_store && nvalid == 8) {
            if (rs) {
                float g[8];
                unpack16<float>(ld128(rs), g);
                unpack16<float>(ld128(rs + 4), g + 4);
#pragma unroll
                for (int i = 0; i < 8; ++i) f[i] += p.beta * g[i];
            }
            post_act_n<8>(p, f);
            st128(o, pack16<float>(f));
            st128(o + 4, pack16<float>(f + 4));
        } else {
            for (int i = 0; i < nvalid; ++i) o[i] = post_act(p, f[i] + (rs ? p.beta * rs[i] : 0.0f));
        }
    } else {
        bf16_t* o = (bf16_t*)p.out + opix * p.ld_out + co;
        const bf16_t* rs = p.resid ? (const bf16_t*)p.resid + opix * p.ld_res + co : nullptr;
        if (p.vec_store && nvalid == 8) {
            if (rs) {
                float g[8];
                unpack16<bf16_t>(ld128(rs), g);
#pragma unroll
                for (int i = 0; i < 8; ++i) f[i] += p.beta * g[i];
            }
            post_act_n<8>(p, f);
            st128(o, pack16<bf16_t>(f));
        } else {
            for (int i = 0; i < nvalid; ++i)
                o[i] = from_f32<bf16_t>(post_act(p, f[i] + (rs ? p.beta * to_f32(rs[i]) : 0.0f)));
        }
    }
}

__device__ __forceinline__ float conv_finish(const ConvArgs& p, float v, float bias, float ga, float slope) {
    v += bias;
    if (p.act == VT_ACT_LRELU) v = (v > 0.0f) ? v : v * slope;   // slope: scalar or per-channel (PReLU)
    else if (p.act >= VT_ACT_RELU_TANH) {
        // ONE tanhf expansion for the three saturating activations (sigmoid(v) = 0.5 + 0.5 tanh(v / 2)): separate
        // expf / tanhf branches in this unrolled epilogue spilled 272 bytes per lane in the 128-channel tile kernels
        // and cost the 256x128 instance 35 % (41 -> 55 us per launch, measured)
        const float a = p.act == VT_ACT_RELU_TANH ? fmaxf(v, 0.0f) : (p.act == VT_ACT_SIGMOID ? 0.5f * v : v);
        const float t = tanhf(a);
        v = p.act == VT_ACT_SIGMOID ? 0.5f + 0.5f * t : t;
    }
    return v * ga;
}

// XCD-aware block -> (tile_m, tile_n, K-slice) mapping.  Workgroup b is dispatched to XCD b % 8
// (observed, MI355X_MICROARCH.md); each XCD has its own 4 MiB L2.  Blocks are renumbered so that
// XCD x owns one contiguous range of the logical order [tile_n][tile_m][slice]: all tiles that
// read the same weight rows (tile_n) -- and neighbouring pixel tiles, which share halo rows --
// run on the same XCD, so a conv's weight matrix is fetched from HBM/MALL once, not once per
// XCD.  Pure speed: any placement gives the same results.  Bijective for every grid size.
// The same logical order with the CHANNEL tile innermost: the workgroups that run together on an XCD (consecutive L) share a
// pixel tile and fetch its patch into that XCD's L2 once; with the channel tile outermost every XCD owns one or two channel
// tiles and all eight read the whole input (conv_upblur at the 64^2 -> 128^2 level: 192 MB of fabric traffic per launch for
// 77 MB of operands, profiles/r05_pmc_traffic.json).  Which workgroup computes a tile does not change its bits.
__device__ __forceinline__ void decode_block_pixel_major(const ConvArgs& p, int& tile_m, int& tile_n) {
    const int nb = gridDim.x, b = blockIdx.x;
    const int q = nb >> 3, r = nb & 7;
    const int xcd = b & 7, idx = b >> 3;
    const int L = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    tile_m = L / p.tiles_n;
    tile_n = L - tile_m * p.tiles_n;
}
// ... and with a BLOCK of blk_pm pixel tiles x blk_cn channel tiles per XCD (one-round launches without a K split whose tile
// counts divide: the host picks the block that minimises patch + weight bytes per XCD, xcd_block()): the 32 x 32 trunk of a
// batch on 256 x 32 tiles reads 3.7 MB per XCD instead of 5.9 (every XCD used to read the whole input).
__device__ __forceinline__ void decode_block_2d(const ConvArgs& p, int& tile_m, int& tile_n, int& split) {
    if (p.blk_cn == 0) {
        const int nb = gridDim.x, b = blockIdx.x;
        const int q = nb >> 3, r = nb & 7;
        const int xcd = b & 7, idx = b >> 3;
        const int L = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
        const int per_n = p.tiles_m * p.splitk;
        tile_n = L / per_n;
        const int rem = L - tile_n * per_n;
        tile_m = rem / p.splitk;
        split = rem - tile_m * p.splitk;
        return;
    }
    const int per = (int)gridDim.x >> 3;                 // (gridDim.x % 8 == 0 here)
    const int xcd = blockIdx.x & 7, w = blockIdx.x >> 3;  // block `xcd`, tile w of it
    const int nsbm = p.tiles_m / p.blk_pm;
    const int sbm = xcd % nsbm, sbn = xcd / nsbm;
    (void)per;
    tile_m = sbm * p.blk_pm + w / p.blk_cn;
    tile_n = sbn * p.blk_cn + w % p.blk_cn;
    split = 0;
}
__device__ __forceinline__ void decode_block(const ConvArgs& p, int& tile_m, int& tile_n, int& split) {
    const int nb = gridDim.x, b = blockIdx.x;
    const int q = nb >> 3, r = nb & 7;
    const int xcd = b & 7, idx = b >> 3;
    const int L = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    const int per_n = p.tiles_m * p.splitk;
    tile_n = L / per_n;
    const int rem = L - tile_n * per_n;
    tile_m = rem / p.splitk;
    split = rem - tile_m * p.splitk;
}

// GEMM row of a tile -> global GEMM row m (or -1 when the row is padding)
struct LinearRows {   // 1-D tiles: BM consecutive pixels of the flattened (n, oy, ox) index
    int m0, M;
    __device__ __forceinline__ int operator()(int row) const {
        const int m = m0 + row;
        return m < M ? m : -1;
    }
};
template <int TW>
struct PatchRows {    // 2-D tiles of TW-pixel rows (patch-resident kernel)
    int img, y0, x0, Ho, Wo;
    __device__ __forceinline__ int operator()(int row) const {
        const int oy = y0 + row / TW, ox = x0 + row % TW;
        return (oy < Ho && ox < Wo) ? (img * Ho + oy) * Wo + ox : -1;
    }
};

// Four consecutive output columns n..n+3 of GEMM row m, finished (bias/activation/gain applied):
// residual add, pixel shuffle of the polyphase form, NHWC (8/16-byte store) or planar NCHW.
__device__ __forceinline__ void store_out4(const ConvArgs& p, int m, int n, float* f) {
    const int HoWo = p.Ho * p.Wo;
    if (p.out_layout == VT_OUT_NHWC) {
        int64_t opix = m;
        int co = n;
        if (p.phases > 1) {
            const int ph = n / p.cout;
            co = n - ph * p.cout;
            const int img = m / HoWo;
            const int rem = m - img * HoWo;
            const int oy = rem / p.Wo, ox = rem - oy * p.Wo;
            opix = ((int64_t)img * (2 * p.Ho) + 2 * oy + (ph >> 1)) * (2 * p.Wo) + 2 * ox + (ph & 1);
        }
        const int lim = (p.phases > 1) ? p.cout : p.coutT;
        const int nvalid = (lim - co) < 4 ? (lim - co) : 4;
        if (p.out_f32) {
            float* o = (float*)p.out + opix * p.ld_out + co;
            const float* rs = p.resid ? (const float*)p.resid + opix * p.ld_res + co : nullptr;
            if (p.vec_store && nvalid == 4) {
                if (rs) {
                    float g[4];
                    unpack16<float>(ld128(rs), g);
#pragma unroll
                    for (int i = 0; i < 4; ++i) f[i] += p.beta * g[i];
                }
                post_act_n<4>(p, f);
                st128(o, pack16<float>(f));
            } else {
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    if (i < nvalid) o[i] = post_act(p, f[i] + (rs ? p.beta * rs[i] : 0.0f));
            }
        } else {
            bf16_t* o = (bf16_t*)p.out + opix * p.ld_out + co;
            const bf16_t* rs = p.resid ? (const bf16_t*)p.resid + opix * p.ld_res + co : nullptr;
            if (p.vec_store && nvalid == 4) {
                if (rs) {
                    const u64v r = *reinterpret_cast<const u64v*>(rs);
                    f[0] += p.beta * vt_u2f(r.x << 16);
                    f[1] += p.beta * vt_u2f(r.x & 0xffff0000u);
                    f[2] += p.beta * vt_u2f(r.y << 16);
                    f[3] += p.beta * vt_u2f(r.y & 0xffff0000u);
                }
                post_act_n<4>(p, f);
                u64v v;
                v.x = pack_bf16x2(f[0], f[1]);
                v.y = pack_bf16x2(f[2], f[3]);
                *reinterpret_cast<u64v*>(o) = v;
            } else {
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    if (i < nvalid) o[i] = from_f32<bf16_t>(post_act(p, f[i] + (rs ? p.beta * to_f32(rs[i]) : 0.0f)));
            }
        }
    } else {
        // planar NCHW fp32 (small cout: ToRGB, fusion_skip, masks; generic op surface)
        float* o = (float*)p.out;
        const float* rs = (const float*)p.resid;
        const int img = m / HoWo;
        const int rem = m - img * HoWo;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if (n + i >= p.coutT) break;
            const int64_t off = ((int64_t)img * p.cout + n + i) * HoWo + rem;
            o[off] = post_act(p, f[i] + (rs ? p.beta * rs[off] : 0.0f));
        }
    }
}

// Fragment order of the output channels.  With the weights as the MFMA "A" operand, lane group q
// of fragment b holds rows 4q..4q+3 of that fragment.  When a wave owns an even number of
// fragments, the weight rows are loaded into the LDS tile in the order
//     tile row 32j + 16h + 4q + r   <-   channel 32j + 8q + 4h + r        (h = fragment parity)
// so that lane group q holds EIGHT consecutive channels across the fragment pair (2j, 2j+1): one
// 16-byte bf16 store per pixel and pair instead of two 8-byte ones -- a wave store covers 16 pixels
// x 64 contiguous bytes.  (The 8-byte form ran the epilogues at ~1 TB/s: 32-byte pieces of
// 128-byte lines.)  Pure relabeling: LDS layout, swizzle and fragment reads do not change.
template <bool PERM>
__device__ __forceinline__ int tile_row_channel(int row) {   // LDS weight-tile row -> channel offset in the tile
    if (!PERM) return row;
    return (row & ~31) | (((row >> 2) & 3) << 3) | (((row >> 4) & 1) << 2) | (row & 3);
}
template <bool PERM>
__device__ __forceinline__ int frag_channel(int b, int q) {  // first of the 4 channels of (fragment b, lane group q)
    if (!PERM) return b * 16 + q * 4;
    return (b >> 1) * 32 + q * 8 + (b & 1) * 4;
}

// Eight consecutive output columns n..n+7 of GEMM row m (bf16 NHWC, vector path): one 16-byte store.
__device__ __forceinline__ bool store_out8_bf16(const ConvArgs& p, int m, int n, float* f, bool have_rpre = false,
                                                u128 rpre = u128{0u, 0u, 0u, 0u}) {
    // rpre (have_rpre): the residual vector of this store, fetched by the caller ahead of ALL its stores
    if (p.out_layout != VT_OUT_NHWC || p.out_f32 || !p.vec_store) return false;
    const int HoWo = p.Ho * p.Wo;
    int64_t opix = m;
    int co = n;
    if (p.phases > 1) {
        const int ph = n / p.cout;
        co = n - ph * p.cout;
        const int img = m / HoWo;
        const int rem = m - img * HoWo;
        const int oy = rem / p.Wo, ox = rem - oy * p.Wo;
        opix = ((int64_t)img * (2 * p.Ho) + 2 * oy + (ph >> 1)) * (2 * p.Wo) + 2 * ox + (ph & 1);
    }
    const int lim = (p.phases > 1) ? p.cout : p.coutT;
    if (co + 8 > lim || ((p.ld_out | co) & 7) || (p.resid && (p.ld_res & 7))) return false;
    bf16_t* o = (bf16_t*)p.out + opix * p.ld_out + co;
    if (p.resid) {
        float g[8];
        unpack16<bf16_t>(have_rpre ? rpre : ld128((const bf16_t*)p.resid + opix * p.ld_res + co), g);
#pragma unroll
        for (int i = 0; i < 8; ++i) f[i] += p.beta * g[i];
    }
    post_act_n<8>(p, f);
    st128(o, pack16<bf16_t>(f));
    return true;
}

// fused-ToRGB weights of the 8 channels a lane holds in fragment pair (b0, b0 + 1): [h][rgb][i], zero beyond coutT
template <typename T, bool PERM>
__device__ __forceinline__ void rgb_pair_weights(const ConvArgs& p, int nbase, int b0, int q, float (&w)[2][3][4]) {
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int n = nbase + frag_channel<PERM>(b0 + h, q);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int nc = (n + i < p.coutT) ? n + i : p.coutT - 1;   // clamped: loads stay unconditional
            const float live = (n + i < p.coutT) ? 1.0f : 0.0f;
#pragma unroll
            for (int j = 0; j < 3; ++j) w[h][j][i] = to_f32(((const T*)p.rgb_w)[j * p.coutT + nc]) * live;
        }
    }
}

// Per-lane epilogue tables: bias and activation slope of the lane's 4 channels of every fragment column, and the gain.
// ALL of them are fetched in one batch (one memory round trip) -- and, in the kernels that can afford the registers, before
// the K loop, so that the round trip is hidden by it.  Fetched per fragment pair inside the finishing loop they were three
// dependent round trips at the end of every workgroup (alpha, tables of the first pair, tables of the second pair), each
// behind the acknowledgement of the stores issued before it: measured on the 256 x 128 patch tiles, same box, the epilogue
// cost 9-14 us per round of 256 workgroups (profiles/r04_epilogue.txt).
template <int TN>
struct EpiTables {
    float bv[TN][4], sv[TN][4];
    float ga;
};
template <int TN, bool PERM>
__device__ __forceinline__ void epi_tables(const ConvArgs& p, int nbase, int q, EpiTables<TN>& t) {
    // nbase = n0 + wn * (TN * 16): first channel of the wave's fragment columns
    int coi[TN][4];
    bool okc[TN][4];
#pragma unroll
    for (int b = 0; b < TN; ++b)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int nn = nbase + frag_channel<PERM>(b, q) + i;
            okc[b][i] = nn < p.coutT;
            const int nc = okc[b][i] ? nn : 0;
            coi[b][i] = (p.phases > 1) ? nc % p.cout : nc;
        }
    // ONE wave-uniform branch per table and unconditional loads at clamped indices inside it.  (The per-element form
    // `(ptr && nn < coutT) ? ptr[co] : 0` makes hipcc branch around every load and wait vmcnt(0) for it.)
    float al = 1.0f;
    if (p.alpha_dev) al = p.alpha_dev[0];
    if (p.bias) {
#pragma unroll
        for (int b = 0; b < TN; ++b)
#pragma unroll
            for (int i = 0; i < 4; ++i) t.bv[b][i] = p.bias[coi[b][i]];
    } else {
#pragma unroll
        for (int b = 0; b < TN; ++b)
#pragma unroll
            for (int i = 0; i < 4; ++i) t.bv[b][i] = 0.0f;
    }
    if (p.slope_vec) {
#pragma unroll
        for (int b = 0; b < TN; ++b)
#pragma unroll
            for (int i = 0; i < 4; ++i) t.sv[b][i] = p.slope_vec[coi[b][i]];
    } else {
#pragma unroll
        for (int b = 0; b < TN; ++b)
#pragma unroll
            for (int i = 0; i < 4; ++i) t.sv[b][i] = p.slope;
    }
#pragma unroll
    for (int b = 0; b < TN; ++b)
#pragma unroll
        for (int i = 0; i < 4; ++i)
            if (!okc[b][i]) t.bv[b][i] = 0.0f, t.sv[b][i] = p.slope;
    t.ga = p.gain_alpha * al;
}

// Epilogue straight from the accumulators.  The kernels issue the MFMAs with the operands
// swapped (weights as the "A" matrix, pixels as "B"), so the C/D fragment of lane (q, l15) holds
// FOUR CONSECUTIVE OUTPUT CHANNELS (4q..4q+3) of ONE pixel (l15): bias, activation, residual and
// the store are done in registers -- 8-byte bf16 / 16-byte fp32 vectors per lane, no LDS staging
// pass, no barriers.  (The LDS-staged epilogue this replaces cost 12 us of a 34 us launch.)
// EPI = 1: ONLY the lean path is compiled in (the host checked conv_lean(): see the lean path below, no split-K).  The general
// epilogue is 20 000+ lines of ISA per 64 x 64 wave tile; with it in the kernel the waves of a workgroup fetch their way
// through ~10 KB of cold code once per tile: the arithmetic of the lean path measured 4.5-6 us per workgroup, four times
// its instruction count (profiles/r04_epilogue.txt).
template <typename T, int BM, int BN, int WM, int WN, int EPI = 0, typename RowMap>
__device__ __forceinline__ void conv_epilogue(const ConvArgs& p, f32x4 (&acc)[BM / WM / 16][BN / WN / 16],
                                              unsigned char* smem, const RowMap rowmap, int n0, int split,
                                              int tile_id, const EpiTables<BN / WN / 16>& tab) {
    static_assert(EPI == 0 || (BN / WN / 16) % 2 == 0, "the lean path finishes fragment pairs");
    // tab: the lane's bias / slope / gain tables (epi_tables) -- fetched by the caller before its K loop where it can afford
    // the registers, else by the overload below right here
    constexpr int TM = BM / WM / 16, TN = BN / WN / 16;
    constexpr bool PERM = (TN % 2 == 0);
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int q = lane >> 4, l15 = lane & 15;
    // ---- split-K: every slice writes its raw fp32 accumulators to the workspace ----------
    if (EPI == 0 && p.splitk > 1) {
        const int64_t slab = (int64_t)p.M * p.ldp;
        float* part = p.partial + (int64_t)split * slab;
#pragma unroll
        for (int a = 0; a < TM; ++a) {
            const int m = rowmap(wm * (TM * 16) + a * 16 + l15);
#pragma unroll
            for (int b = 0; b < TN; ++b) {
                // slab columns in FRAGMENT order (the reduce pass undoes it): the four lane groups
                // of a fragment write 64 contiguous bytes per pixel
                const int n = n0 + wn * (TN * 16) + b * 16 + q * 4;
                if (m >= 0 && n < p.ldp) {
                    float f[4] = {acc[a][b][0], acc[a][b][1], acc[a][b][2], acc[a][b][3]};
                    st128(part + (int64_t)m * p.ldp + n, pack16<float>(f));
                }
            }
        }
        if (!p.tickets) return;  // two-pass mode: conv_splitk_reduce_kernel finishes
        // In-launch reduction: the slice that arrives LAST at the tile's ticket counter sums all
        // slabs (in slice order 0..S-1 -> deterministic whichever slice that is) and runs the
        // normal epilogue.  Publish: plain stores -> drain -> barrier -> one agent-scope release
        // -> relaxed ticket; consume: agent-scope acquire by one lane -> barrier -> plain loads.
        vt_drain_vmem();
        __syncthreads();
        int* flag = reinterpret_cast<int*>(smem);  // the K loop is done with the tile buffers
        int* ticket = p.tickets + tile_id;
        if (tid == 0) {
            vt_release_agent();
            const int t = vt_ticket_add(ticket, 1);
            const int last = (t == p.splitk - 1);
            if (last) {
                vt_acquire_agent();
                vt_ticket_store(ticket, 0);  // re-arm for the next launch (all slices have arrived)
            }
            *flag = last;
        }
        __syncthreads();
        if (*flag == 0) return;
#pragma unroll
        for (int a = 0; a < TM; ++a) {
            const int m = rowmap(wm * (TM * 16) + a * 16 + l15);
#pragma unroll
            for (int b = 0; b < TN; ++b) {
                const int n = n0 + wn * (TN * 16) + b * 16 + q * 4;   // fragment-order slab column
                float f[4] = {0.f, 0.f, 0.f, 0.f};
                if (m >= 0 && n < p.ldp) {
                    const float* src = p.partial + (int64_t)m * p.ldp + n;
                    unpack16<float>(ld128(src), f);
                    constexpr int SG = 8;   // slabs per memory round trip (summed in slice order all the same)
                    for (int s0 = 1; s0 < p.splitk; s0 += SG) {
                        float g[SG][4];
#pragma unroll
                        for (int k = 0; k < SG; ++k)
                            unpack16<float>(ld128(src + ((s0 + k < p.splitk) ? s0 + k : 0) * slab), g[k]);
#pragma unroll
                        for (int k = 0; k < SG; ++k) {
                            if (s0 + k < p.splitk) {
#pragma unroll
                                for (int i = 0; i < 4; ++i) f[i] += g[k][i];
                            }
                        }
                    }
                }
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[a][b][i] = f[i];
            }
        }
        // fall through to the fused epilogue below
    }
    const float ga = tab.ga;
    // Fused ToRGB (model/stylegan/model.py:383-392): when this tile holds ALL output channels of its
    // pixels, the 1x1 modulated conv C -> 3 that follows a same-resolution StyledConv is three dot
    // products over values that are already in registers -- the C-channel activation (67 MB at the
    // 1024^2 level) is not read back from HBM by a separate ToRGB launch.
    const bool rgbf = p.rgb_w != nullptr;
    float rp[TM][3];
#pragma unroll
    for (int a = 0; a < TM; ++a) rp[a][0] = rp[a][1] = rp[a][2] = 0.0f;
    constexpr int BS = PERM ? 2 : 1;   // fragments finished together: under PERM a pair = 8 consecutive channels
    // residual vectors of every 16-byte bf16 store, ahead of the first store (loads at clamped addresses: no branch per
    // element).  Fetched inside store_out8_bf16 each was a round trip behind the acknowledgement of the store before it.
    constexpr int NPAIR = (TN + BS - 1) / BS;
    u128 rpre[NPAIR][TM];
    const bool rvec = EPI == 1 ? (sizeof(T) == 2 && p.resid != nullptr)
                               : (BS == 2 && sizeof(T) == 2 && p.resid && p.phases == 1 && p.out_layout == VT_OUT_NHWC && !p.out_f32 &&
                                  p.vec_store && !((p.ld_out | p.ld_res) & 7));
    if (rvec) {
#pragma unroll
        for (int b0 = 0; b0 < TN; b0 += BS) {
            const int nn = n0 + wn * (TN * 16) + frag_channel<PERM>(b0, q);
#pragma unroll
            for (int a = 0; a < TM; ++a) {
                const int m = rowmap(wm * (TM * 16) + a * 16 + l15);
                const bool ok = m >= 0 && nn + 8 <= p.coutT;
                rpre[b0 / BS][a] = ld128((const bf16_t*)p.resid + (int64_t)(ok ? m : 0) * p.ld_res + (ok ? nn : 0));
            }
        }
    }
    // Lean path: (Leaky)ReLU or no activation, bf16 NHWC rows of whole 16-byte vectors -- every 3x3 conv of the frame's hot
    // kernels.  Branch-free per value (the activation is a select against a slope of 1 where there is none).  The general
    // loop below runs conv_finish / store_out8_bf16 per value and per store: wave-uniform, but ~10 scalar branches per
    // value, 20 000 lines of ISA for a 64 x 64 wave tile -- the epilogue of the 256 x 128 patch tiles cost 9-14 us per round
    // of 256 workgroups (37 of 102 us on the 128 -> 128 conv at 256^2, profiles/r04_epilogue.txt), and it was neither
    // its table loads nor its stores.
    bool lean = false;
    if constexpr (BS == 2) {
        constexpr bool H = sizeof(T) == 2;   // bf16 rows (one 16-byte store per 8 channels) or fp32 rows (two)
        lean = EPI == 1 ||   // (the same predicate on the host: conv_lean())
               ((p.act == VT_ACT_NONE || p.act == VT_ACT_LRELU) && p.phases == 1 && p.out_layout == VT_OUT_NHWC &&
                (H ? !p.out_f32 : p.out_f32 != 0) && p.vec_store && !p.post_relu && !(p.coutT & 7) &&
                (H ? (!p.resid || rvec) && !(p.ld_out & 7) : !(p.ld_out & 3) && (!p.resid || !(p.ld_res & 3))));
        // (with the fused ToRGB too.  Round 4 kept those convs on the general path after a wrong-image-row defect; the cause was
        // not this code but what hipcc's SLP vectoriser made of the partial sums below -- `v_pk_mul_f32 / v_pk_add_f32 ...
        // op_sel:[0,1]`, whose low half reads src1's high register as ZERO on gfx950 while another wave's MFMA shares the SIMD
        // (tools/probe/pk_war_probe.hip).  The library is built with -fno-slp-vectorize and tests/test_isa_lint.py forbids
        // the instruction form)
        if (lean) {
            const bool lrelu = p.act == VT_ACT_LRELU;
            // bf16: the fused ToRGB on the matrix cores.  Under PERM the 8 values a lane finishes for a fragment pair are 8
            // CONSECUTIVE channels of one pixel -- packed to bf16 (the 16 bytes the store writes) they are exactly the pixel
            // operand of v_mfma_f32_16x16x32_bf16 for this pair's 32 channels, and with the ToRGB weights as the other operand
            // (row j < 3 of the fragment = image plane j, rows 3..15 zero) ONE instruction per 16-pixel fragment row and pair
            // replaces 48 multiply-adds per lane, the weight unpacking and the 24 cross-lane shuffles of the sums below: lanes
            // q == 0 end up with the three plane values of their pixel.  The products are those of the un-fused ToRGB launch
            // (bf16 activations as stored x bf16 weights, fp32 accumulate); round 4's lean fused epilogue cost +20 us on the
            // 128 -> 128 conv at 256^2 in vector instructions (profiles/r05_torgb_fused.txt).
            // The weight operand of fragment row a carries its three rows at 4a .. 4a + 2, so the products of the TM <= 4
            // fragment rows land in DIFFERENT rows of one accumulator: lane (q, l15) ends up with the three plane values of
            // pixel l15 of fragment row q -- the reduce-scatter the vector form needed 9 shuffles per plane triple for.
            constexpr bool RGB_MMA = sizeof(T) == 2;
            static_assert(!RGB_MMA || TM <= 4, "four fragment rows share the 16 rows of the ToRGB accumulator");
            f32x4 racc = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int b0 = 0; b0 < TN; b0 += 2) {
                const int nn = n0 + wn * (TN * 16) + frag_channel<PERM>(b0, q);
                float bb[8], se[8], wq[2][3][4];
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    bb[k] = tab.bv[b0 + (k >> 2)][k & 3];
                    se[k] = lrelu ? tab.sv[b0 + (k >> 2)][k & 3] : 1.0f;
                }
                u128 wfrag = u128{0u, 0u, 0u, 0u};
                if (rgbf) {
                    if constexpr (RGB_MMA) {
                        // weight operand: lane (q, l15) = plane l15 & 3 (row l15 of the operand of fragment row l15 >> 2), channels
                        // nn .. nn + 7 (unconditional load at a clamped address)
                        const bool okw = (l15 & 3) < 3 && nn + 8 <= p.coutT;
                        const u128 wv = ld128((const bf16_t*)p.rgb_w + (okw ? (l15 & 3) * p.coutT + nn : 0));
                        wfrag.x = okw ? wv.x : 0u, wfrag.y = okw ? wv.y : 0u, wfrag.z = okw ? wv.z : 0u, wfrag.w = okw ? wv.w : 0u;
                    } else {
                        rgb_pair_weights<T, PERM>(p, n0 + wn * (TN * 16), b0, q, wq);
                    }
                }
#pragma unroll
                for (int a = 0; a < TM; ++a) {
                    const int m = rowmap(wm * (TM * 16) + a * 16 + l15);
                    float f[8];
#pragma unroll
                    for (int k = 0; k < 8; ++k) {
                        float v = acc[a][b0 + (k >> 2)][k & 3] + bb[k];
                        v = (v > 0.0f) ? v : v * se[k];
                        f[k] = v * ga;
                    }
                    if (rgbf) {
                        if constexpr (RGB_MMA) {
                            const bool mine = (l15 >> 2) == a;
                            u128 wa;
                            wa.x = mine ? wfrag.x : 0u, wa.y = mine ? wfrag.y : 0u, wa.z = mine ? wfrag.z : 0u, wa.w = mine ? wfrag.w : 0u;
                            Mma<bf16_t>::run(racc, wa, pack16<bf16_t>(f));
                        } else {
#pragma unroll
                            for (int h = 0; h < 2; ++h)
#pragma unroll
                                for (int j = 0; j < 3; ++j)
                                    rp[a][j] += (f[4 * h] * wq[h][j][0] + f[4 * h + 1] * wq[h][j][1]) +
                                                (f[4 * h + 2] * wq[h][j][2] + f[4 * h + 3] * wq[h][j][3]);
                        }
                    }
                    const bool live = m >= 0 && nn + 8 <= p.coutT;
                    if constexpr (H) {
                        if (rvec) {
                            float g[8];
                            unpack16<bf16_t>(rpre[b0 / 2][a], g);
#pragma unroll
                            for (int k = 0; k < 8; ++k) f[k] += p.beta * g[k];
                        }
                        if (live) st128((bf16_t*)p.out + (int64_t)m * p.ld_out + nn, pack16<bf16_t>(f));
                    } else {
                        if (p.resid) {   // (fp32: the residual of a pixel is fetched here, at a clamped address)
                            const float* rs = (const float*)p.resid + (int64_t)(live ? m : 0) * p.ld_res + (live ? nn : 0);
                            float g[8];
                            unpack16<float>(ld128(rs), g), unpack16<float>(ld128(rs + 4), g + 4);
#pragma unroll
                            for (int k = 0; k < 8; ++k) f[k] += p.beta * g[k];
                        }
                        if (live) {
                            float* o = (float*)p.out + (int64_t)m * p.ld_out + nn;
                            st128(o, pack16<float>(f)), st128(o + 4, pack16<float>(f + 4));
                        }
                    }
                }
            }
            if constexpr (RGB_MMA) {
                if (rgbf) {
                    // ---- tail of the matrix form: lane (q, l15) owns pixel l15 of fragment row q -------------------------------
                    const int HoWo = p.Ho * p.Wo;
                    const int m = q < TM ? rowmap(wm * (TM * 16) + q * 16 + l15) : -1;
                    const int mc = m < 0 ? 0 : m;
                    const int img = mc / HoWo;
                    const int64_t off = (int64_t)img * 3 * HoWo + (mc - img * HoWo);   // (clamped: the loads stay unconditional)
                    float rsd[3] = {0.0f, 0.0f, 0.0f}, rb[3] = {0.0f, 0.0f, 0.0f};
                    const bool writer = wn == 0 && m >= 0;
                    if (p.rgb_resid) {
#pragma unroll
                        for (int j = 0; j < 3; ++j) rsd[j] = p.rgb_resid[off + (int64_t)j * HoWo];
                    }
                    if (p.rgb_bias) rb[0] = p.rgb_bias[0], rb[1] = p.rgb_bias[1], rb[2] = p.rgb_bias[2];
                    float v[3] = {racc[0], racc[1], racc[2]};
                    if (WN > 1) {   // ... summed over the WN wavefronts that split the channels (through LDS, fixed order)
                        float* xs = reinterpret_cast<float*>(smem);
                        __syncthreads();   // every wave is done with the tile buffers
                        if (wn != 0) {
#pragma unroll
                            for (int j = 0; j < 3; ++j) xs[(((wn - 1) * WM + wm) * 64 + lane) * 3 + j] = v[j];
                        }
                        __syncthreads();
                        if (wn == 0) {
                            for (int w2 = 1; w2 < WN; ++w2)
#pragma unroll
                                for (int j = 0; j < 3; ++j) v[j] += xs[(((w2 - 1) * WM + wm) * 64 + lane) * 3 + j];
                        }
                    }
                    if (writer) {
#pragma unroll
                        for (int j = 0; j < 3; ++j) p.rgb_out[off + (int64_t)j * HoWo] = (v[j] + rb[j]) + rsd[j];
                    }
                    return;
                }
            }
        }
    }
    if (EPI == 0 && !lean) {
#pragma unroll
    for (int b0 = 0; b0 < TN; b0 += BS) {
        int nh[BS];
        float bv[BS][4], sv[BS][4], wr[BS][3][4];
#pragma unroll
        for (int h = 0; h < BS; ++h) {
            nh[h] = n0 + wn * (TN * 16) + frag_channel<PERM>(b0 + h, q);
#pragma unroll
            for (int i = 0; i < 4; ++i) bv[h][i] = tab.bv[b0 + h][i], sv[h][i] = tab.sv[b0 + h][i];
        }
        if constexpr (BS == 2) {
            if (rgbf) rgb_pair_weights<T, PERM>(p, n0 + wn * (TN * 16), b0, q, wr);
        } else {
            if (rgbf) {   // wave-uniform: convs without the fusion pay one scalar branch per fragment column
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int nc = (nh[0] + i < p.coutT) ? nh[0] + i : p.coutT - 1;   // clamped: loads stay unconditional
                    const float live = (nh[0] + i < p.coutT) ? 1.0f : 0.0f;
#pragma unroll
                    for (int j = 0; j < 3; ++j) wr[0][j][i] = to_f32(((const T*)p.rgb_w)[j * p.coutT + nc]) * live;
                }
            }
        }
#pragma unroll
        for (int a = 0; a < TM; ++a) {
            const int m = rowmap(wm * (TM * 16) + a * 16 + l15);
            if (m < 0 || nh[0] >= p.coutT) continue;
            float f[4 * BS];
#pragma unroll
            for (int h = 0; h < BS; ++h) {
#pragma unroll
                for (int i = 0; i < 4; ++i) f[4 * h + i] = conv_finish(p, acc[a][b0 + h][i], bv[h][i], ga, sv[h][i]);
                if (rgbf && nh[h] < p.coutT) {
                    // the ToRGB conv reads the activation as it is STORED (rounded to T): the un-fused launch does, and so do the
                    // MFMA forms of the lean / c32 / resident epilogues -- every fused form multiplies the same operands
                    float fr[4];
#pragma unroll
                    for (int i = 0; i < 4; ++i) fr[i] = to_f32(from_f32<T>(f[4 * h + i]));
#pragma unroll
                    for (int j = 0; j < 3; ++j)
                        rp[a][j] += (fr[0] * wr[h][j][0] + fr[1] * wr[h][j][1]) + (fr[2] * wr[h][j][2] + fr[3] * wr[h][j][3]);
                }
            }
            if (BS == 2 && nh[1] < p.coutT && store_out8_bf16(p, m, nh[0], f, rvec, rpre[b0 / BS][a])) continue;
#pragma unroll
            for (int h = 0; h < BS; ++h)
                if (nh[h] < p.coutT) store_out4(p, m, nh[h], f + 4 * h);
        }
    }
    }   // !lean
    if (!rgbf) return;
    // up-sampled skip + bias of the pixels this lane writes: ALL loads issued here, unconditional at clamped offsets under
    // wave-uniform branches, so that they fly during the shuffles / the LDS exchange below.  (`if (p.rgb_resid) v +=
    // p.rgb_resid[off]` per element was a branch + load + vmcnt(0) each: 3 x TM dependent HBM round trips at the end of
    // every tile -- the 128 -> 128 @256^2 conv spent more time there than in its MFMA loop.)
    const int HoWo = p.Ho * p.Wo;
    const bool writer = wn == 0 && q == 0;
    int64_t roff[TM];
    float rsd[TM][3], rb[3] = {0.0f, 0.0f, 0.0f};
#pragma unroll
    for (int a = 0; a < TM; ++a) {
        const int m = rowmap(wm * (TM * 16) + a * 16 + l15);
        const int mc = m < 0 ? 0 : m;
        const int img = mc / HoWo;
        roff[a] = (m < 0 || !writer) ? (int64_t)-1 : (int64_t)img * 3 * HoWo + (mc - img * HoWo);
    }
    if (p.rgb_resid) {
#pragma unroll
        for (int a = 0; a < TM; ++a)
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                rsd[a][j] = p.rgb_resid[(roff[a] < 0 ? 0 : roff[a]) + (int64_t)j * HoWo];
            }
    } else {
#pragma unroll
        for (int a = 0; a < TM; ++a) rsd[a][0] = rsd[a][1] = rsd[a][2] = 0.0f;
    }
    if (p.rgb_bias) rb[0] = p.rgb_bias[0], rb[1] = p.rgb_bias[1], rb[2] = p.rgb_bias[2];
    // sum the partial dot products over the 4 lane groups (channels 4q..4q+3 of every fragment) ...
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            float v = rp[a][j];
            v += __shfl_xor(v, 16, 64);
            v += __shfl_xor(v, 32, 64);
            rp[a][j] = v;
        }
    // ... and over the WN wavefronts that split the channels (through LDS, fixed order)
    if (WN > 1) {
        float* xs = reinterpret_cast<float*>(smem);
        __syncthreads();   // every wave is done with the tile buffers
        if (q == 0) {
#pragma unroll
            for (int a = 0; a < TM; ++a)
#pragma unroll
                for (int j = 0; j < 3; ++j) xs[((wn * WM + wm) * (TM * 16) + a * 16 + l15) * 3 + j] = rp[a][j];
        }
        __syncthreads();
        if (writer) {
#pragma unroll
            for (int a = 0; a < TM; ++a)
#pragma unroll
                for (int j = 0; j < 3; ++j) {
                    float v = rp[a][j];
                    for (int w2 = 1; w2 < WN; ++w2) v += xs[((w2 * WM + wm) * (TM * 16) + a * 16 + l15) * 3 + j];
                    rp[a][j] = v;
                }
        }
    }
#pragma unroll
    for (int a = 0; a < TM; ++a) {
        if (roff[a] < 0) continue;
#pragma unroll
        for (int j = 0; j < 3; ++j) p.rgb_out[roff[a] + (int64_t)j * HoWo] = (rp[a][j] + rb[j]) + rsd[a][j];
    }
}

// the same with the tables fetched here (one batch)
template <typename T, int BM, int BN, int WM, int WN, typename RowMap>
__device__ __forceinline__ void conv_epilogue(const ConvArgs& p, f32x4 (&acc)[BM / WM / 16][BN / WN / 16],
                                              unsigned char* smem, const RowMap rowmap, int n0, int split, int tile_id) {
    constexpr int TN = BN / WN / 16;
    EpiTables<TN> tab;
    const int tid = threadIdx.x;
    epi_tables<TN, (TN % 2 == 0)>(p, n0 + ((tid >> 6) % WN) * (TN * 16), (tid & 63) >> 4, tab);
    conv_epilogue<T, BM, BN, WM, WN>(p, acc, smem, rowmap, n0, split, tile_id, tab);
}

template <typename T, int BM, int BN, int WM, int WN>
__global__ void __launch_bounds__(256)
conv_igemm_kernel(const ConvArgs p) {
    static_assert(WM * WN == 4, "4 wavefronts per workgroup");
    constexpr int VEC = 16 / sizeof(T);  // elements per 16-byte vector
    constexpr int BK = 8 * VEC;          // elements per 128-byte tile row
    constexpr int TM = BM / WM / 16, TN = BN / WN / 16;
    constexpr bool PERM = (TN % 2 == 0);   // weight rows in fragment order: see tile_row_channel
    static_assert(TM >= 1 && TN >= 1, "wave tile too small");
    constexpr int AI = (BM + 31) / 32, BI = (BN + 31) / 32;
    constexpr int A_BYTES = BM * 128, B_BYTES = BN * 128;
    constexpr int SROWS = WM * 16, SLD = BN + 4;
    static_assert(SROWS * SLD * 4 <= 2 * (A_BYTES + B_BYTES), "epilogue staging must fit");

    __shared__ __attribute__((aligned(16))) unsigned char smem[2 * (A_BYTES + B_BYTES)];
    // buffer b lives at smem + b*(A_BYTES+B_BYTES): [A tile | B tile]
    auto sA = [&](int b) -> unsigned char* { return smem + b * (A_BYTES + B_BYTES); };
    auto sB = [&](int b) -> unsigned char* { return smem + b * (A_BYTES + B_BYTES) + A_BYTES; };

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    int tile_m, tile_n, split;
    decode_block(p, tile_m, tile_n, split);
    const int m0 = tile_m * BM, n0 = tile_n * BN;

    // ---- loader state -----------------------------------------------------------
    const int j = tid & 7, rbase = tid >> 3;
    int a_pix[AI], a_y[AI], a_x[AI], a_img[AI];
    bool a_ok[AI];
    const int HoWo = p.Ho * p.Wo;
#pragma unroll
    for (int i = 0; i < AI; ++i) {
        const int row = rbase + 32 * i;
        const int m = m0 + row;
        a_ok[i] = (row < BM) && (m < p.M);
        const int mm = a_ok[i] ? m : 0;
        const int img = mm / HoWo;
        const int rem = mm - img * HoWo;
        const int oy = rem / p.Wo, ox = rem - oy * p.Wo;
        a_img[i] = img;
        a_pix[i] = img * p.H * p.W;
        a_y[i] = p.transposed ? oy + p.pad : oy * p.stride - p.pad;
        a_x[i] = p.transposed ? ox + p.pad : ox * p.stride - p.pad_x;
    }
    const int nk_all = (p.K + BK - 1) / BK;
    const int kt0 = split * p.kps;
    const int kt1 = (kt0 + p.kps < nk_all) ? kt0 + p.kps : nk_all;
    int tap = (kt0 * BK + j * VEC) / p.cin;
    int kc = (kt0 * BK + j * VEC) - tap * p.cin;

    const T* src0 = (const T*)p.src0;
    const T* src1 = (const T*)p.src1;
    const T* wgt = (const T*)p.wgt;

    u128 ra[AI], rb[BI];
    auto load_tiles = [&]() {
        const bool tap_ok = tap < p.taps;
        const int ky = tap / p.kw, kx = tap - ky * p.kw;
        const int dy = ky * p.dil, dx = kx * p.dil;
        const T* sp;
        int ld, cc;
        if (kc < p.c0) {
            sp = src0; ld = p.ld0; cc = kc;
        } else {
            sp = src1; ld = p.ld1; cc = kc - p.c0;
        }
#pragma unroll
        for (int i = 0; i < AI; ++i) {
            u128 v = zero128();
            if (a_ok[i] && tap_ok) {
                int iy, ix;
                bool in;
                if (!p.transposed) {
                    iy = a_y[i] + dy;
                    ix = a_x[i] + dx;
                    in = (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
                } else {
                    const int ty = a_y[i] - dy, tx = a_x[i] - dx;
                    in = ty >= 0 && tx >= 0 && (ty % p.stride) == 0 && (tx % p.stride) == 0;
                    iy = ty / p.stride;
                    ix = tx / p.stride;
                    in = in && iy < p.H && ix < p.W;
                }
                if (in) {
                    v = ld128(sp + (int64_t)(a_pix[i] + iy * p.W + ix) * ld + cc);
                    if (p.in_scale) {
                        const int so = a_img[i] * p.cin + kc;
                        v = affine16<T>(v, p.in_scale + so, p.in_shift + so);
                    }
                }
            }
            ra[i] = v;
        }
        const int k_elem = tap * p.cin + kc;
#pragma unroll
        for (int i = 0; i < BI; ++i) {
            const int row = rbase + 32 * i;
            const int n = n0 + tile_row_channel<PERM>(row);
            u128 v = zero128();
            if (row < BN && n < p.coutT && tap_ok) v = ld128(wgt + (int64_t)n * p.K + k_elem);
            rb[i] = v;
        }
    };
    auto advance_k = [&]() {
        kc += BK;
        while (kc >= p.cin) {
            kc -= p.cin;
            ++tap;
        }
    };
    auto store_tiles = [&](int buf) {
#pragma unroll
        for (int i = 0; i < AI; ++i) {
            const int row = rbase + 32 * i;
            if (row < BM) st128(sA(buf) + row * 128 + ((j ^ (row & 7)) << 4), ra[i]);
        }
#pragma unroll
        for (int i = 0; i < BI; ++i) {
            const int row = rbase + 32 * i;
            if (row < BN) st128(sB(buf) + row * 128 + ((j ^ (row & 7)) << 4), rb[i]);
        }
    };

    f32x4 acc[TM][TN];
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int nk = kt1 - kt0;
    const int q = lane >> 4, l15 = lane & 15, l7 = lane & 7;
    const int a_row0 = wm * (TM * 16) + l15, b_row0 = wn * (TN * 16) + l15;

    load_tiles();
    store_tiles(0);
    __syncthreads();

    for (int kt = 0; kt < nk; ++kt) {
        const int buf = kt & 1;
        if (kt + 1 < nk) {
            advance_k();
            load_tiles();
        }
#pragma unroll
        for (int sub = 0; sub < 2; ++sub) {
            const int phys = ((sub * 4 + q) ^ l7) << 4;
            u128 fa[TM], fb[TN];
#pragma unroll
            for (int a = 0; a < TM; ++a) fa[a] = ld128(sA(buf) + (a_row0 + a * 16) * 128 + phys);
#pragma unroll
            for (int b = 0; b < TN; ++b) fb[b] = ld128(sB(buf) + (b_row0 + b * 16) * 128 + phys);
#pragma unroll
            for (int a = 0; a < TM; ++a)
#pragma unroll
                for (int b = 0; b < TN; ++b) Mma<T>::run(acc[a][b], fb[b], fa[a]);
        }
        if (kt + 1 < nk) store_tiles(buf ^ 1);
        __syncthreads();
    }

    conv_epilogue<T, BM, BN, WM, WN>(p, acc, smem, LinearRows{m0, p.M}, n0, split, tile_n * p.tiles_m + tile_m);
}

// ---------------------------------------------------------------------------------------
// Fast path: direct-to-LDS loader (buffer_load_dwordx4 ... lds).
//
// When every K-step lies inside ONE filter tap and ONE concat source (channel counts are
// multiples of the K-step) the im2col gather needs no per-step vector arithmetic at all:
//   address = buffer base (SGPR descriptor)
//           + voffset  per-lane byte offset of the output pixel's "centre" input pixel and of
//                      the lane's 16-byte chunk; replaced by an out-of-range sentinel when the
//                      tap falls outside the image for that pixel -- the buffer unit then writes
//                      ZEROS to LDS, which is exactly the conv's zero padding (probed on gfx950,
//                      tools/probe/glds_probe.hip: soffset takes part in the range check)
//           + soffset  wave-uniform byte offset of (tap, channel chunk)  (SGPR)
// and the data lands in LDS without touching VGPRs.  One wave-instruction fills 8 tile rows of
// 128 B (lane l -> row l/8, physical 16-byte slot l%8); the XOR swizzle of the register-staged
// kernel is kept by permuting which chunk of its row a lane fetches: chunk (l%8) ^ (row&7).
// Per K-step a wave issues (BM+BN)/32 loads, 2*(TM+TN) ds_read_b128 and 2*TM*TN MFMAs; the
// validity masks / effective offsets are recomputed only when the tap (or source) changes.
// Two LDS buffers: the loads of step t+1 are in flight during the MFMAs of step t.
// ---------------------------------------------------------------------------------------
constexpr uint32_t GLDS_OOB = 0x80000000u;  // >= any num_records we create (< 2^31)

struct GldsArgs {
    uint32_t nrec0, nrec1, nrecw;   // buffer sizes in bytes (incl. the tap bias for sources)
    uint32_t bias0, bias1;          // bytes the source bases are moved back by ((pad*W+pad) pixels)
};

template <typename T, int BM, int BN, int WM, int WN, int NST>
__global__ void __launch_bounds__(256)
conv_igemm_glds_kernel(const ConvArgs p, const GldsArgs g) {
    static_assert(WM * WN == 4, "4 wavefronts per workgroup");
    static_assert(NST >= 2 && NST <= 4, "2..4 LDS stages");
    constexpr int ESZ = (int)sizeof(T);
    constexpr int VEC = 16 / ESZ;
    constexpr int BK = 8 * VEC;
    constexpr int TM = BM / WM / 16, TN = BN / WN / 16;
    constexpr bool PERM = (TN % 2 == 0);   // weight rows in fragment order: see tile_row_channel
    // every wave issues the same number of loads per K-step (counted vmcnt below): tiles are
    // padded to a multiple of 32 rows in LDS; rows beyond the tile fetch the zero sentinel
    constexpr int AI = (BM + 31) / 32, BI = (BN + 31) / 32;
    constexpr int LOADS = AI + BI;
    constexpr int A_BYTES = AI * 32 * 128, B_BYTES = BI * 32 * 128;
    constexpr int STAGE = A_BYTES + B_BYTES;
    constexpr int SROWS = WM * 16, SLD = BN + 4;
    static_assert(SROWS * SLD * 4 <= NST * STAGE, "epilogue staging must fit");

    __shared__ __attribute__((aligned(1024))) unsigned char smem[NST * STAGE];
    auto sA = [&](int b) -> unsigned char* { return smem + b * STAGE; };
    auto sB = [&](int b) -> unsigned char* { return smem + b * STAGE + A_BYTES; };

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = vt_uniform(tid >> 6) & 3;
    const int wm = wave / WN, wn = wave % WN;
    int tile_m, tile_n, split;
    if (p.blk_pm < 0) {   // pixel-major order (launch_cfg): the channel tiles of one pixel tile run together on one XCD
        decode_block_pixel_major(p, tile_m, tile_n);
        split = 0;
    } else {
        decode_block(p, tile_m, tile_n, split);
    }
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    const int HoWo = p.Ho * p.Wo;

    // ---- loader state ---------------------------------------------------------------
    const int lrow = lane >> 3;                  // row inside an 8-row group
    const int jj = (lane & 7) ^ lrow;            // logical 16-byte chunk this lane fetches
    uint32_t ctr0[AI], ctr1[AI], eff[AI];
    int cy[AI], cx[AI];
#pragma unroll
    for (int i = 0; i < AI; ++i) {
        const int row = (i * 4 + wave) * 8 + lrow;
        const int m = m0 + row;
        const bool ok = row < BM && m < p.M;
        const int mm = ok ? m : 0;
        const int img = mm / HoWo;
        const int rem = mm - img * HoWo;
        const int oy = rem / p.Wo, ox = rem - oy * p.Wo;
        const uint32_t pix = (uint32_t)((img * p.H + oy * p.stride) * p.W + ox * p.stride);
        ctr0[i] = pix * (uint32_t)(p.ld0 * ESZ) + jj * 16;
        ctr1[i] = pix * (uint32_t)(p.ld1 * ESZ) + jj * 16;
        cy[i] = ok ? oy * p.stride - p.pad : -0x40000000;  // never valid
        cx[i] = ox * p.stride - p.pad;
        eff[i] = GLDS_OOB;
    }
    uint32_t woff[BI];
#pragma unroll
    for (int i = 0; i < BI; ++i) {
        const int row = (i * 4 + wave) * 8 + lrow;
        const int n = n0 + tile_row_channel<PERM>(row);   // weight rows in fragment order (see tile_row_channel)
        woff[i] = (row < BN && n < p.coutT) ? (uint32_t)n * (uint32_t)(p.K * ESZ) + jj * 16 : GLDS_OOB;
    }
    const BufRsrc r0 = vt_make_rsrc((const char*)p.src0 - g.bias0, g.nrec0);
    const BufRsrc r1 = vt_make_rsrc(p.src1 ? (const char*)p.src1 - g.bias1 : (const char*)p.src0, p.src1 ? g.nrec1 : 0u);
    const BufRsrc rw = vt_make_rsrc(p.wgt, g.nrecw);

    // K-step state (all wave-uniform); `tap`/`kc` describe the NEXT step to be issued
    const int nk_all = p.K / BK;
    const int kt0 = split * p.kps;
    const int kt1 = (kt0 + p.kps < nk_all) ? kt0 + p.kps : nk_all;
    int tap = (kt0 * BK) / p.cin;
    int kc = kt0 * BK - tap * p.cin;   // channel offset inside the concatenated input
    int seg_tap = -1, seg_src = -1;
    uint32_t soff_a = 0;               // byte offset of (tap, first channel of the source)

    auto issue = [&](int kt, int buf) {
        const int src = (kc >= p.c0) ? 1 : 0;
        if (tap != seg_tap || src != seg_src) {   // uniform: new tap or new source
            seg_tap = tap;
            seg_src = src;
            const int ky = tap / p.kw, kx = tap - ky * p.kw;
            const int dy = ky * p.dil, dx = kx * p.dil;
            soff_a = (uint32_t)(dy * p.W + dx) * (uint32_t)((src ? p.ld1 : p.ld0) * ESZ);
#pragma unroll
            for (int i = 0; i < AI; ++i) {
                const bool in = (unsigned)(cy[i] + dy) < (unsigned)p.H && (unsigned)(cx[i] + dx) < (unsigned)p.W;
                eff[i] = in ? (src ? ctr1[i] : ctr0[i]) : GLDS_OOB;
            }
        }
        const uint32_t sa = soff_a + (uint32_t)((src ? kc - p.c0 : kc) * ESZ);
        const BufRsrc& ra = src ? r1 : r0;
#pragma unroll
        for (int i = 0; i < AI; ++i) vt_glds16(ra, sA(buf) + (i * 4 + wave) * 1024, eff[i], sa);
        const uint32_t sw = (uint32_t)(kt * BK * ESZ);
#pragma unroll
        for (int i = 0; i < BI; ++i) vt_glds16(rw, sB(buf) + (i * 4 + wave) * 1024, woff[i], sw);
        // advance to the following step
        kc += BK;
        if (kc >= p.cin) {
            kc = 0;
            ++tap;
        }
    };

    f32x4 acc[TM][TN];
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int q = lane >> 4, l15 = lane & 15, l7 = lane & 7;
    const int a_row0 = wm * (TM * 16) + l15, b_row0 = wn * (TN * 16) + l15;

    // prologue: NST-1 steps in flight, the first one landed
    int issued = kt0;
#pragma unroll
    for (int s = 0; s < NST - 1; ++s)
        if (issued < kt1) {
            issue(issued, s);
            ++issued;
        }
    // Outstanding loads allowed while step `kt` is consumed: those of the steps issued after it.
    auto wait_for = [&](int kt) {
        const int ahead = issued - 1 - kt;  // steps issued beyond kt
        if (NST >= 4 && ahead >= 2) vt_glds_wait_n<2 * LOADS>();
        else if (NST >= 3 && ahead >= 1) vt_glds_wait_n<LOADS>();
        else vt_glds_wait_n<0>();
    };
    wait_for(kt0);
    vt_lds_barrier();

    int buf = 0, nbuf = NST - 1;  // buffer of step kt / buffer the next issued step goes to
    for (int kt = kt0; kt < kt1; ++kt) {
        if (issued < kt1) {
            issue(issued, nbuf);
            ++issued;
        }
        if constexpr (is_x3<T>::value) {
            mma_row<T, TM, TN>(acc,
                               [&](int a, int sub) { return sA(buf) + (a_row0 + a * 16) * 128 + (((sub * 4 + q) ^ l7) << 4); },
                               [&](int b, int sub) { return sB(buf) + (b_row0 + b * 16) * 128 + (((sub * 4 + q) ^ l7) << 4); });
        } else {
#pragma unroll
        for (int sub = 0; sub < 2; ++sub) {
            const int phys = ((sub * 4 + q) ^ l7) << 4;
            u128 fa[TM], fb[TN];
#pragma unroll
            for (int a = 0; a < TM; ++a) fa[a] = ld128(sA(buf) + (a_row0 + a * 16) * 128 + phys);
#pragma unroll
            for (int b = 0; b < TN; ++b) fb[b] = ld128(sB(buf) + (b_row0 + b * 16) * 128 + phys);
#pragma unroll
            for (int a = 0; a < TM; ++a)
#pragma unroll
                for (int b = 0; b < TN; ++b) Mma<T>::run(acc[a][b], fb[b], fa[a]);
        }
        }
        // step kt+1 must have landed (in every wave) before anyone reads it; step kt's buffer may
        // be overwritten by the next issue once every wave has finished reading it
        wait_for(kt + 1);
        vt_lds_barrier();
        buf = (buf + 1 == NST) ? 0 : buf + 1;
        nbuf = (nbuf + 1 == NST) ? 0 : nbuf + 1;
    }
    __syncthreads();
    conv_epilogue<T, BM, BN, WM, WN>(p, acc, smem, LinearRows{m0, p.M}, n0, split, tile_n * p.tiles_m + tile_m);
}

// ---------------------------------------------------------------------------------------
// Patch-resident 3x3 kernel (stride 1, pad == dilation): the direct-to-LDS GEMM above re-fetches
// every input pixel once per filter tap (9x) and every weight once per pixel tile, and runs out of
// L2->LDS bandwidth (~10 TB/s measured) long before the MFMA pipe.  Here a workgroup owns a 2-D
// tile of TH x 16 output pixels; per 64-channel chunk it fetches the input PATCH (tile + halo,
// (TH+2d) x (16+2d) pixels) ONCE and runs all 9 taps out of it -- an A fragment for tap (ky,kx)
// is just the 16 patch pixels shifted by (ky*d, kx*d), i.e. 16 consecutive LDS rows, so the XOR
// swizzle stays conflict-free.  Only the weights stream per tap (3-stage ring).  L2->LDS bytes
// per MAC drop ~3x (A: 9x fewer, B: halved again by the 256-pixel tile of the 8-wave variant).
//   K order: [chunk][tap]  (same products as [tap][chunk], different fp32 summation order)
//   split-K: slices are whole chunks.
// ---------------------------------------------------------------------------------------
// counted-vmcnt selector: `ahead` weight steps (LB loads each) and optionally the next patch (PA
// loads) may stay in flight
template <int K, int LB, int PA>
struct PatchWait {
    static __device__ __forceinline__ void run(int ahead, bool a_out) {
        if (ahead >= K) {
            if (a_out) vt_glds_wait_n<K * LB + PA>();
            else vt_glds_wait_n<K * LB>();
        } else {
            PatchWait<K - 1, LB, PA>::run(ahead, a_out);
        }
    }
};
template <int LB, int PA>
struct PatchWait<0, LB, PA> {
    static __device__ __forceinline__ void run(int, bool a_out) {
        if (a_out) vt_glds_wait_n<PA>();
        else vt_glds_wait_n<0>();
    }
};

// NSTB = weight ring depth (NSTB-1 taps of weights in flight).  With a deep ring and one channel
// chunk per K-slice (the 32x32-pixel trunk under split-K) a workgroup issues its patch and most of
// its 9 weight tiles up front and waits for memory ONCE instead of once per tap.
// ABUF = patch buffers: 2 = next chunk's patch prefetched during the current chunk; 1 = the slice
// must be a single chunk (host-checked).
template <typename T, int TH, int BN, int WM, int WN, int DIL, int NSTB, int ABUF>
__global__ void __launch_bounds__(WM * WN * 64)
conv_patch_kernel(const ConvArgs p, const GldsArgs g) {
    constexpr int TW = 16;
    constexpr int NW = WM * WN;                 // wavefronts
    constexpr int BM = TH * TW;
    constexpr int ESZ = (int)sizeof(T);
    constexpr int VEC = 16 / ESZ;
    constexpr int BK = 8 * VEC;                 // channels per chunk (128 B)
    constexpr int TM = BM / WM / 16, TN = BN / WN / 16;
    constexpr bool PERM = (TN % 2 == 0);   // weight rows in fragment order: see tile_row_channel
    constexpr int PH = TH + 2 * DIL, PW = TW + 2 * DIL, PROWS = PH * PW;
    constexpr int PA = ((PROWS + 7) / 8 + NW - 1) / NW;   // patch loads per wave per chunk
    constexpr int LB = ((BN + 7) / 8 + NW - 1) / NW;      // weight loads per wave per tap
    constexpr int A_BYTES = PA * NW * 1024, B_BYTES = LB * NW * 1024;
    constexpr int JA = 3;
    static_assert(BM % (WM * 16) == 0 && BN % (WN * 16) == 0, "wave tiling");
    static_assert(TM * WM == TH, "one 16-pixel tile row per MFMA row block");
    static_assert(NSTB >= 3 && (ABUF == 1 || ABUF == 2), "ring depth / patch buffers");
    static_assert(ABUF * A_BYTES + NSTB * B_BYTES <= 160 * 1024, "LDS budget");
    static_assert((NSTB - 2) * LB + PA < 64, "vmcnt is 6 bits");

    __shared__ __attribute__((aligned(1024))) unsigned char smem[ABUF * A_BYTES + NSTB * B_BYTES];
    auto sA = [&](int b) -> unsigned char* { return smem + b * A_BYTES; };
    auto sB = [&](int b) -> unsigned char* { return smem + ABUF * A_BYTES + b * B_BYTES; };

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = vt_uniform(tid >> 6) & (NW - 1);
    const int wm = wave / WN, wn = wave % WN;
    int tile_m, tile_n, split;
    decode_block(p, tile_m, tile_n, split);
    const int tiles_x = (p.Wo + TW - 1) / TW, tiles_y = (p.Ho + TH - 1) / TH;
    const int img = tile_m / (tiles_x * tiles_y);
    const int trem = tile_m - img * (tiles_x * tiles_y);
    const int y0 = (trem / tiles_x) * TH, x0 = (trem % tiles_x) * TW;
    const int n0 = tile_n * BN;

    // ---- loader state (fixed for the whole kernel: only the SGPR offset moves) -------------
    const int lrow = lane >> 3;
    const int jj = (lane & 7) ^ lrow;
    uint32_t pa0[PA], pa1[PA];
#pragma unroll
    for (int i = 0; i < PA; ++i) {
        const int pr = (i * NW + wave) * 8 + lrow;
        const int py = pr / PW, px = pr - py * PW;
        const int iy = y0 - DIL + py, ix = x0 - DIL + px;
        const bool in = pr < PROWS && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
        const uint32_t pix = (uint32_t)((img * p.H + iy) * p.W + ix);
        pa0[i] = in ? pix * (uint32_t)(p.ld0 * ESZ) + jj * 16 : GLDS_OOB;
        pa1[i] = in ? pix * (uint32_t)(p.ld1 * ESZ) + jj * 16 : GLDS_OOB;
    }
    uint32_t woff[LB];
#pragma unroll
    for (int i = 0; i < LB; ++i) {
        const int row = (i * NW + wave) * 8 + lrow;
        const int n = n0 + tile_row_channel<PERM>(row);   // weight rows in fragment order (see tile_row_channel)
        woff[i] = (row < BN && n < p.coutT) ? (uint32_t)n * (uint32_t)(p.K * ESZ) + jj * 16 : GLDS_OOB;
    }
    const BufRsrc r0 = vt_make_rsrc(p.src0, g.nrec0);
    const BufRsrc r1 = vt_make_rsrc(p.src1 ? p.src1 : p.src0, p.src1 ? g.nrec1 : 0u);
    const BufRsrc rw = vt_make_rsrc(p.wgt, g.nrecw);

    // K slices are whole chunks: p.kps chunks per slice
    const int nchunks = p.cin / BK;
    const int ch0 = split * p.kps;
    const int ch1 = (ch0 + p.kps < nchunks) ? ch0 + p.kps : nchunks;
    const int nsteps = (ch1 - ch0) * 9;

    auto issue_a = [&](int chunk, int abuf) {
        const int kc = chunk * BK;
        const bool s1 = kc >= p.c0;
        const uint32_t so = (uint32_t)((s1 ? kc - p.c0 : kc) * ESZ);
        if (s1) {
#pragma unroll
            for (int i = 0; i < PA; ++i) vt_glds16(r1, sA(abuf) + (i * NW + wave) * 1024, pa1[i], so);
        } else {
#pragma unroll
            for (int i = 0; i < PA; ++i) vt_glds16(r0, sA(abuf) + (i * NW + wave) * 1024, pa0[i], so);
        }
    };
    auto issue_b = [&](int step, int bbuf) {   // step = (chunk - ch0) * 9 + tap
        const int cl = step / 9, tap = step - cl * 9;
        const uint32_t so = (uint32_t)((tap * p.cin + (ch0 + cl) * BK) * ESZ);
#pragma unroll
        for (int i = 0; i < LB; ++i) vt_glds16(rw, sB(bbuf) + (i * NW + wave) * 1024, woff[i], so);
    };

    f32x4 acc[TM][TN];
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int q = lane >> 4, l15 = lane & 15, l7 = lane & 7;
    const int b_row0 = wn * (TN * 16) + l15;

    // prologue: patch of the first chunk + NSTB-1 weight steps in flight; patch and step 0 landed
    issue_a(ch0, 0);
    int issued = 0;   // weight steps issued so far
#pragma unroll
    for (int s = 0; s < NSTB - 1; ++s)
        if (issued < nsteps) {
            issue_b(issued, s);
            ++issued;
        }
    PatchWait<NSTB - 2, LB, PA>::run(issued - 1, false);
    vt_lds_barrier();

    int abuf = 0, bbuf = 0, nbbuf = NSTB - 1, tap = 0, chunk = ch0;
    int a_age = 99;  // steps since the next chunk's patch was issued (99 = none in flight)
    for (int s = 0; s < nsteps; ++s) {
        if (issued < nsteps) {
            issue_b(issued, nbbuf);
            ++issued;
        }
        if (ABUF == 2 && tap == JA && chunk + 1 < ch1) {
            issue_a(chunk + 1, abuf ^ 1);
            a_age = 0;
        }
        const int ky = tap / 3, kx = tap - ky * 3;
        if constexpr (is_x3<T>::value) {
            mma_row<T, TM, TN>(acc,
                               [&](int a, int sub) {
                                   const int pr = (wm * TM + a + ky * DIL) * PW + kx * DIL + l15;
                                   return sA(abuf) + pr * 128 + (((sub * 4 + q) ^ (pr & 7)) << 4);
                               },
                               [&](int b, int sub) { return sB(bbuf) + (b_row0 + b * 16) * 128 + (((sub * 4 + q) ^ l7) << 4); });
        } else {
#pragma unroll
        for (int sub = 0; sub < 2; ++sub) {
            const int slot = sub * 4 + q;
            u128 fa[TM], fb[TN];
#pragma unroll
            for (int a = 0; a < TM; ++a) {
                const int pr = (wm * TM + a + ky * DIL) * PW + kx * DIL + l15;
                fa[a] = ld128(sA(abuf) + pr * 128 + ((slot ^ (pr & 7)) << 4));
            }
#pragma unroll
            for (int b = 0; b < TN; ++b) fb[b] = ld128(sB(bbuf) + (b_row0 + b * 16) * 128 + ((slot ^ l7) << 4));
#pragma unroll
            for (int a = 0; a < TM; ++a)
#pragma unroll
                for (int b = 0; b < TN; ++b) Mma<T>::run(acc[a][b], fb[b], fa[a]);
        }
        }
        // Before step s+1 is read its weights must have landed (and, at a chunk boundary, the next
        // patch).  Loads younger than B(s+1): B(s+2..issued-1) and a patch issued in step s-1 or s
        // (it is issued AFTER that step's weights).  Once the patch is two steps old it sits
        // behind nothing that may stay outstanding, so the count below forces it to completion --
        // JA + 2 < 9, i.e. always before the chunk ends.
        PatchWait<NSTB - 2, LB, PA>::run(issued - 2 - s, a_age <= 1);
        vt_lds_barrier();
        if (a_age < 99) ++a_age;
        bbuf = (bbuf + 1 == NSTB) ? 0 : bbuf + 1;
        nbbuf = (nbbuf + 1 == NSTB) ? 0 : nbbuf + 1;
        if (++tap == 9) {
            tap = 0;
            ++chunk;
            if (ABUF == 2) abuf ^= 1;
            a_age = 99;
        }
    }
    __syncthreads();
    conv_epilogue<T, BM, BN, WM, WN>(p, acc, smem, PatchRows<TW>{img, y0, x0, p.Ho, p.Wo}, n0, split, tile_n * p.tiles_m + tile_m);
}

// ---------------------------------------------------------------------------------------
// 3x3, Cin = Cout = 32, stride 1: the 1024x1024 level of the generator (convs.15) -- 19 GFLOP on
// 134 MB of activations, HBM-bound (144 flop/B < the 312 flop/B ridge).  Everything that is not
// the activation stream is taken off the memory path:
//   * the whole weight tensor (32 x 9 x 32 bf16 = 18 KB) lives in REGISTERS for the life of the
//     workgroup (18 MFMA B-fragments per lane, loaded once);
//   * workgroups are PERSISTENT: each loops over 16x16-pixel tiles, the next tile's input patch
//     (18x18 pixels x 64 B, direct-to-LDS with zero fill at the image border) is in flight while
//     the current tile runs its 9 taps -- no per-tile prologue, one barrier pair per tile;
//   * a pixel's 32 channels are one 64-byte LDS row; a tap's A fragment is one ds_read_b128 per
//     16 pixels (4x4 XOR swizzle of the 16-byte slots), feeding 2 MFMAs (16x16x32);
//   * epilogue (bias, LeakyReLU, fused ToRGB) from registers.
// ---------------------------------------------------------------------------------------
template <typename T>
__global__ void __launch_bounds__(256, 2)   // 2 waves per SIMD = 2 workgroups per CU (<= 256 registers)
conv3x3_c32_kernel(const ConvArgs p, const GldsArgs g) {
    static_assert(sizeof(T) == 2, "bf16 only (64-byte pixel rows)");
    constexpr int TH = 16, TW = 16, BM = 256, BN = 32, WM = 4, WN = 1;
    constexpr int TM = 4, TN = 2;
    constexpr int PH = TH + 2, PW = TW + 2, PROWS = PH * PW;   // 324 patch pixels
    constexpr int PA = ((PROWS + 15) / 16 + 3) / 4;              // 16-pixel loads per wave per tile: 6
    constexpr int A_BYTES = PA * 4 * 1024;                       // 24 KB per buffer
    constexpr int NBUF = 3;                                      // patches in flight: this tile's + two ahead (72 KB: two workgroups per CU)
    __shared__ __attribute__((aligned(1024))) unsigned char smem[NBUF * A_BYTES];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = vt_uniform(tid >> 6) & 3;
    const int wm = wave;
    const int q = lane >> 4, l15 = lane & 15;
    // swizzle of the four 16-byte slots of a 64-byte row: slot' = slot ^ G[(pixel >> 2) & 3], G = {0,2,0,2}.
    // Conflict-free for gfx950's ds_read_b128 lane groups ({0-3,12-15,20-27}, ...) when 16 consecutive pixels are
    // read from ANY starting pixel (tools/lds_bank_check.py; round 1's G = {0,2,3,1} was 2-way for 35 of 40
    // starts -- it assumed contiguous 16-lane groups)
    auto swz = [](int pr) -> int { return ((pr >> 2) & 1) << 1; };

    // ---- weights -> registers: fragment (tap, b): row n = b*16 + l15, k = q*8 .. q*8+7 ----------
    // cout > 32 (the encoder's 32 -> 128 conv): blockIdx.y = the group of 32 output channels this workgroup owns -- its
    // own 18 KB of weights in registers, the same patches (L2 / MALL hits after the first group), a 64-byte slice of
    // every output pixel
    const int cg = (int)blockIdx.y * 32;
    u128 wreg[9][TN];
    {
        const T* wg = (const T*)p.wgt;
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
            for (int b = 0; b < TN; ++b)   // fragment order: (b, lane group q) <-> channels 8q + 4b .. +3
                wreg[t][b] = ld128(wg + (int64_t)(cg + tile_row_channel<true>(b * 16 + l15)) * p.K + t * 32 + q * 8);
    }
    const BufRsrc r0 = vt_make_rsrc(p.src0, g.nrec0);
    const int tiles_x = (p.Wo + TW - 1) / TW, tiles_y = (p.Ho + TH - 1) / TH;
    const int ntiles = p.N * tiles_x * tiles_y;
    const int lpix = lane >> 2, lchunk = lane & 3;

    auto issue = [&](int tile, int buf) {
        const int img = tile / (tiles_x * tiles_y);
        const int trem = tile - img * (tiles_x * tiles_y);
        const int y0 = (trem / tiles_x) * TH, x0 = (trem % tiles_x) * TW;
#pragma unroll
        for (int i = 0; i < PA; ++i) {
            const int pr = (i * 4 + wave) * 16 + lpix;
            const int py = pr / PW, px = pr - py * PW;
            const int iy = y0 - 1 + py, ix = x0 - 1 + px;
            const bool in = pr < PROWS && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
            const uint32_t pix = (uint32_t)((img * p.H + iy) * p.W + ix);
            const uint32_t off = in ? pix * (uint32_t)(p.ld0 * 2) + ((lchunk ^ swz(pr)) << 4) : GLDS_OOB;
            vt_glds16(r0, smem + buf * A_BYTES + (i * 4 + wave) * 1024, off, 0u);
        }
    };

    // fused-ToRGB constants in registers for the life of the workgroup
    const bool rgbf = p.rgb_w != nullptr;
    // (one wave-uniform branch per table with all of its loads inside: the per-element "pointer ? load : 0" form is a
    // dependent L2 round trip per element -- 35 of them ahead of the first tile of every persistent workgroup)
    // Fused ToRGB on the matrix cores (conv_epilogue has the derivation): a lane's 8 finished values of a pixel, packed to
    // bf16, are the pixel operand of v_mfma_f32_16x16x32_bf16 over the 32 channels; the weight operand holds plane l15 & 3 in
    // row l15 (used for fragment row l15 >> 2), so ONE accumulator collects the four tile rows and lane (q, l15) ends with
    // the three plane values of pixel l15 of row q.  4 registers of weights instead of 24, 4 MFMAs instead of 96 multiply-adds
    // and 9 shuffles per tile.
    float bvr[TN][4];
#pragma unroll
    for (int b = 0; b < TN; ++b)
#pragma unroll
        for (int i = 0; i < 4; ++i) bvr[b][i] = 0.0f;
    u128 rwf = u128{0u, 0u, 0u, 0u};
    if (rgbf) {
        const bool okw = (l15 & 3) < 3;
        const u128 wv = ld128((const bf16_t*)p.rgb_w + (okw ? (l15 & 3) * 32 + q * 8 : 0));
        rwf.x = okw ? wv.x : 0u, rwf.y = okw ? wv.y : 0u, rwf.z = okw ? wv.z : 0u, rwf.w = okw ? wv.w : 0u;
    }
    if (p.bias) {
#pragma unroll
        for (int b = 0; b < TN; ++b)
#pragma unroll
            for (int i = 0; i < 4; ++i) bvr[b][i] = p.bias[cg + frag_channel<true>(b, q) + i];
    }
    float rb0 = 0.0f, rb1 = 0.0f, rb2 = 0.0f;
    if (rgbf && p.rgb_bias) rb0 = p.rgb_bias[0], rb1 = p.rgb_bias[1], rb2 = p.rgb_bias[2];

    int tile = blockIdx.x;
    if (tile >= ntiles) return;
    issue(tile, 0);
    if (tile + (int)gridDim.x < ntiles) issue(tile + (int)gridDim.x, 1);
    int buf = 0;
    for (; tile < ntiles; tile += gridDim.x) {
        // Round 6: TWO patches ahead (a CU streams from HBM at ~12 B/clk, profiles/r06_ingest_probe.txt: one 21 KB patch per
        // workgroup in flight does not cover the latency at that rate).  The patch of tile t + 2 goes into the buffer tile t - 1
        // was read from (every wave is past the barrier that ended t - 1).  vmcnt counts everything in order -- patches, the
        // skip loads, the stores: the 2 PA newest operations are the patch just issued and >= PA of what followed the patch of
        // t + 1 (itself, the stores and skip loads of t - 1), so this tile's patch is older than all of them.
        const int next2 = tile + 2 * (int)gridDim.x;
        if (next2 < ntiles) {
            issue(next2, buf == 0 ? 2 : buf - 1);
            vt_glds_wait_n<2 * PA>();
        } else if (tile + (int)gridDim.x < ntiles) {
            vt_glds_wait_n<PA>();      // (the last but one: only the next patch is younger)
        } else {
            vt_glds_wait_n<0>();
        }
        vt_lds_barrier();
        const int img = tile / (tiles_x * tiles_y);
        const int trem = tile - img * (tiles_x * tiles_y);
        const int y0 = (trem / tiles_x) * TH, x0 = (trem % tiles_x) * TW;
        const PatchRows<TW> rowmap{img, y0, x0, p.Ho, p.Wo};
        const int HoWo = p.Ho * p.Wo;
        // the up-sampled skip this tile adds to: fetched NOW so the loads fly during the 9 taps.  Lane (q, l15) finishes
        // the ToRGB of tile row q of its wave (TM == 4 rows, 4 lane groups; see the reduce-scatter in the epilogue): three
        // 256-byte loads and stores per wave and tile instead of twelve 64-byte ones
        static_assert(TM == 4, "one tile row per lane group");
        float rsd[3] = {0.0f, 0.0f, 0.0f};
        const int m_rgb = rowmap(wm * (TM * 16) + q * 16 + l15);
        int64_t o_rgb = 0;
        {
            const int mm = m_rgb < 0 ? 0 : m_rgb;
            const int im = mm / HoWo;
            o_rgb = (int64_t)im * 3 * HoWo + (mm - im * HoWo);
        }
        if (rgbf && p.rgb_resid) {
#pragma unroll
            for (int j = 0; j < 3; ++j) rsd[j] = p.rgb_resid[o_rgb + (int64_t)j * HoWo];
        }
        f32x4 acc[TM][TN];
#pragma unroll
        for (int a = 0; a < TM; ++a)
#pragma unroll
            for (int b = 0; b < TN; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
        const unsigned char* sa = smem + buf * A_BYTES;
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const int ky = t / 3, kx = t - ky * 3;
            u128 fa[TM];
#pragma unroll
            for (int a = 0; a < TM; ++a) {
                const int pr = (wm * TM + a + ky) * PW + kx + l15;
                fa[a] = ld128(sa + pr * 64 + ((q ^ swz(pr)) << 4));
            }
#pragma unroll
            for (int a = 0; a < TM; ++a)
#pragma unroll
                for (int b = 0; b < TN; ++b) Mma<T>::run(acc[a][b], wreg[t][b], fa[a]);
            vt_sched_fence();   // keep one tap's fragments live at a time (the scheduler otherwise
                                // hoists all 36 reads: 376 registers, occupancy 1)
        }
        // lean epilogue (the generic one costs ~200 registers next to the resident weights):
        // bias + LeakyReLU * gain -> bf16 NHWC (8-byte stores), optional fused ToRGB
        {
            const float ga = p.gain_alpha;
            f32x4 racc = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int a = 0; a < TM; ++a) {
                const int m = rowmap(wm * (TM * 16) + a * 16 + l15);
                float f[8];   // channels 8q .. 8q+7 of this pixel: fragment 0 holds 8q..+3, fragment 1 8q+4..+7
#pragma unroll
                for (int b = 0; b < TN; ++b) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        float v = acc[a][b][i] + bvr[b][i];
                        if (p.act == VT_ACT_LRELU) v = (v > 0.0f) ? v : v * p.slope;
                        f[4 * b + i] = v * ga;
                    }
                }
                const u128 fpk = pack16<bf16_t>(f);
                if (rgbf) {
                    const bool mine = (l15 >> 2) == a;
                    u128 wa;
                    wa.x = mine ? rwf.x : 0u, wa.y = mine ? rwf.y : 0u, wa.z = mine ? rwf.z : 0u, wa.w = mine ? rwf.w : 0u;
                    Mma<bf16_t>::run(racc, wa, fpk);
                }
                // one 16-byte store per lane: the four lane groups write the pixel's 64 bytes
                if (m >= 0 && !p.rgb_only) st128((bf16_t*)p.out + (int64_t)m * p.ld_out + cg + q * 8, fpk);
            }
            if (rgbf) {
                const float rr[3] = {racc[0], racc[1], racc[2]};   // lane (q, l15): pixel l15 of tile row q
                if (m_rgb >= 0) {
                    p.rgb_out[o_rgb] = rr[0] + rb0 + rsd[0];
                    p.rgb_out[o_rgb + HoWo] = rr[1] + rb1 + rsd[1];
                    p.rgb_out[o_rgb + 2 * (int64_t)HoWo] = rr[2] + rb2 + rsd[2];
                }
            }
        }
        vt_lds_barrier();   // every wave is done reading `buf` before the next issue overwrites it
        buf = buf + 1 == NBUF ? 0 : buf + 1;
    }
}

// compile-time loop: f(std::integral_constant<int, 0>{}), ..., f(std::integral_constant<int, N - 1>{})
template <int... I, typename F>
__device__ __forceinline__ void vt_static_for_impl(std::integer_sequence<int, I...>, F&& f) {
    (f(std::integral_constant<int, I>{}), ...);
}
template <int N, typename F>
__device__ __forceinline__ void vt_static_for(F&& f) {
    vt_static_for_impl(std::make_integer_sequence<int, N>{}, static_cast<F&&>(f));
}

#include "conv_fullk.hpp"
#include "conv_fullkw.hpp"
#include "conv_upblur.hpp"
static int device_cus();
#include "conv_upblur_rows.hpp"
#include "conv_upblur_flat.hpp"
#include "conv_thin.hpp"
#include "conv_patch_pipe.hpp"
#include "conv_patch_chunk.hpp"
#include "conv_patch_s2.hpp"
#include "conv_patch_resident.hpp"
#include "conv_patch_persist.hpp"

// slab column of the 4-channel group starting at channel n (n % 4 == 0): identity, or the
// fragment order the slices wrote (tile row 32j + 16h + 4q + r  <-  channel 32j + 8q + 4h + r)
__device__ __forceinline__ int slab_col4(const ConvArgs& p, int n) {
    return p.slab_perm ? (n & ~31) + ((n >> 2) & 1) * 16 + ((n >> 3) & 3) * 4 : n;
}

// Second pass of a split-K convolution: sum the K-slices in slice order (deterministic),
// then the same bias / activation / gain / residual / layout epilogue as the fused kernel.
// One thread per (GEMM row, 8 output columns).
__global__ void __launch_bounds__(256) conv_splitk_reduce_kernel(const ConvArgs p) {
    const int nv = p.ldp / 8;
    const int64_t total = (int64_t)p.M * nv;
    const int64_t slab = (int64_t)p.M * p.ldp;
    const float ga = p.gain_alpha * (p.alpha_dev ? p.alpha_dev[0] : 1.0f);
    const int HoWo = p.Ho * p.Wo;
    for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
        int m, v8;
        if (p.out_layout == VT_OUT_NHWC) {
            m = (int)(idx / nv);
            v8 = (int)(idx - (int64_t)m * nv);
        } else {  // planar output: lanes walk pixels
            v8 = (int)(idx / p.M);
            m = (int)(idx - (int64_t)v8 * p.M);
        }
        const int n = v8 * 8;
        if (n >= p.coutT) continue;
        // channels n..n+3 and n+4..n+7 sit 16 columns apart when the slab is in fragment order
        const int c0 = slab_col4(p, n);
        const int dc = slab_col4(p, n + 4) - c0;
        const float* src = p.partial + (int64_t)m * p.ldp + c0;
        float f[8];
        unpack16<float>(ld128(src), f);
        unpack16<float>(ld128(src + dc), f + 4);
        // slices in order (deterministic), SG of them per memory round trip: the thin convs (masks, ToRGB,
        // fusion_skip: 9-16 slices, a few KB of output) spent one L2 latency PER SLICE here (5-10 us a launch)
        constexpr int SG = 8;
        for (int s0 = 1; s0 < p.splitk; s0 += SG) {
            float g[SG][8];
#pragma unroll
            for (int k = 0; k < SG; ++k) {
                const int s = (s0 + k < p.splitk) ? s0 + k : 0;   // clamped: loads stay unconditional
                unpack16<float>(ld128(src + s * slab), g[k]);
                unpack16<float>(ld128(src + s * slab + dc), g[k] + 4);
            }
#pragma unroll
            for (int k = 0; k < SG; ++k) {
                if (s0 + k < p.splitk) {
#pragma unroll
                    for (int i = 0; i < 8; ++i) f[i] += g[k][i];
                }
            }
        }
        {
            float bv8[8], sv8[8];
            int co8[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int nc = (n + i < p.coutT) ? n + i : 0;
                co8[i] = (p.phases > 1) ? nc % p.cout : nc;
                bv8[i] = 0.0f;
                sv8[i] = p.slope;
            }
            if (p.bias) {
#pragma unroll
                for (int i = 0; i < 8; ++i) bv8[i] = p.bias[co8[i]];
            }
            if (p.slope_vec) {
#pragma unroll
                for (int i = 0; i < 8; ++i) sv8[i] = p.slope_vec[co8[i]];
            }
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const bool okc = n + i < p.coutT;
                f[i] = conv_finish(p, f[i], okc ? bv8[i] : 0.0f, ga, okc ? sv8[i] : p.slope);
            }
        }
        if (p.out_layout == VT_OUT_NHWC) {
            store_nhwc8(p, m, n, f);
        } else {
            float* o = (float*)p.out;
            const float* rs = (const float*)p.resid;
            const int img = m / HoWo;
            const int rem = m - img * HoWo;
            for (int i = 0; i < 8 && n + i < p.coutT; ++i) {
                const int64_t off = ((int64_t)img * p.cout + n + i) * HoWo + rem;
                o[off] = post_act(p, f[i] + (rs ? p.beta * rs[off] : 0.0f));
            }
        }
    }
}

// Split-K reduce pass that ALSO emits the InstanceNorm chunk records of the tensor it writes (the
// 32x32-pixel trunk: every conv feeds an AdaIN, and a separate statistics launch re-reads the tensor
// for 5 us of latency).  One workgroup per (image, statistics chunk, group of CG channel vectors):
// threads = CG channel vectors x 256/CG pixel rows, so at the trunk's 16-pixel chunks every thread
// owns ONE pixel and fetches its S slab vectors + the residual in a single round trip.  The
// finished, ROUNDED vectors go to the output and to LDS; one thread per channel then folds the
// chunk in exactly the order instnorm_partial_kernel uses (pixel rows r = 0..rowsP-1, each row's
// pixels in order, rows summed in order) -- the records are bit-identical to what the stand-alone
// statistics pass produces from the stored tensor.  chunk_px <= 64 (planes up to 16384 pixels).
template <typename T>
__global__ void __launch_bounds__(256)
conv_splitk_reduce_stats_kernel(const ConvArgs p, int chunk_px, int chunks, int cgroups) {
    constexpr int VEC = 16 / sizeof(T);
    constexpr int CG = 16;            // channel vectors per workgroup
    constexpr int ROWS = 256 / CG;    // pixel rows per workgroup
    constexpr int SG = 8;             // slabs fetched per round trip
    constexpr int MAXPX = 64;
    __shared__ float vals[MAXPX][CG * VEC + 1];
    const int tid = threadIdx.x;
    const int hw = p.Ho * p.Wo;
    const int cg = blockIdx.x % cgroups;
    const int chunk = (blockIdx.x / cgroups) % chunks, img = blockIdx.x / (cgroups * chunks);
    const int p_lo = chunk * chunk_px;
    const int p_hi = (p_lo + chunk_px < hw) ? p_lo + chunk_px : hw;
    const int c = p.coutT;
    const int cvn = c / VEC;
    const int cvl = tid % CG, prow = tid / CG;
    const int cv = cg * CG + cvl;
    const int n = cv * VEC;
    const int64_t slab = (int64_t)p.M * p.ldp;
    const float ga = p.gain_alpha * (p.alpha_dev ? p.alpha_dev[0] : 1.0f);
    if (cv < cvn) {
        float bv[VEC], sv[VEC];
#pragma unroll
        for (int i = 0; i < VEC; ++i) bv[i] = 0.0f, sv[i] = p.slope;
        if (p.bias) {
#pragma unroll
            for (int i = 0; i < VEC; ++i) bv[i] = p.bias[n + i];
        }
        if (p.slope_vec) {
#pragma unroll
            for (int i = 0; i < VEC; ++i) sv[i] = p.slope_vec[n + i];
        }
        for (int px = p_lo + prow; px < p_hi; px += ROWS) {
            const int64_t m = (int64_t)img * hw + px;
            const float* src = p.partial + m * p.ldp;
            int col[VEC / 4];
#pragma unroll
            for (int v = 0; v < VEC; v += 4) col[v / 4] = slab_col4(p, n + v);
            float f[VEC], r[VEC];
#pragma unroll
            for (int v = 0; v < VEC; v += 4) unpack16<float>(ld128(src + col[v / 4]), f + v);
            if (p.resid) unpack16<T>(ld128((const T*)p.resid + m * p.ld_res + n), r);
            for (int s0 = 1; s0 < p.splitk; s0 += SG) {   // slice order 0..S-1: deterministic
                float g[SG][VEC];
#pragma unroll
                for (int k = 0; k < SG; ++k) {
                    const int s = (s0 + k < p.splitk) ? s0 + k : 0;   // clamped: loads stay unconditional
#pragma unroll
                    for (int v = 0; v < VEC; v += 4) unpack16<float>(ld128(src + s * slab + col[v / 4]), g[k] + v);
                }
#pragma unroll
                for (int k = 0; k < SG; ++k) {
                    if (s0 + k < p.splitk) {
#pragma unroll
                        for (int i = 0; i < VEC; ++i) f[i] += g[k][i];
                    }
                }
            }
#pragma unroll
            for (int i = 0; i < VEC; ++i) {
                f[i] = conv_finish(p, f[i], bv[i], ga, sv[i]);
                if (p.resid) f[i] += p.beta * r[i];
                f[i] = post_act(p, f[i]);
            }
            const u128 packed = pack16<T>(f);
            st128((T*)p.out + m * p.ld_out + n, packed);
            unpack16<T>(packed, f);
#pragma unroll
            for (int i = 0; i < VEC; ++i) vals[px - p_lo][cvl * VEC + i] = f[i];
        }
    }
    __syncthreads();
    // fold: thread t owns channel cg*CG*VEC + t of this chunk
    const int ch = cg * CG * VEC + tid;
    if (tid < CG * VEC && ch < c) {
        const int cparP = cvn < 256 ? cvn : 256;
        const int rowsP = 256 / cparP;            // instnorm_partial_kernel's pixel rows
        const float x0 = vals[0][tid];
        float a1 = 0.0f, a2 = 0.0f;
        for (int r = 0; r < rowsP; ++r) {
            float t1 = 0.0f, t2 = 0.0f;
            for (int px = p_lo + r; px < p_hi; px += rowsP) {
                const float d = vals[px - p_lo][tid] - x0;
                t1 += d;
                t2 += d * d;
            }
            a1 += t1;
            a2 += t2;
        }
        StatRec rec;
        rec.x0 = x0;
        rec.s1 = a1;
        rec.s2 = a2;
        p.stats_part[((int64_t)img * chunks + chunk) * c + ch] = rec;
    }
}

// set by launch_reduce when the reduce pass wrote the chunk records (vt_conv2d then skips the
// stand-alone statistics launch)
static thread_local bool g_stats_emitted = false;

// the tensor this conv writes can carry InstanceNorm records: NHWC, compute dtype, vector stores
static bool stats_fusable(const ConvArgs& a, int esz) {
    const int vec = 16 / esz;
    return a.stats_part && a.out_layout == VT_OUT_NHWC && a.phases == 1 && a.vec_store &&
           a.out_f32 == (esz == 4) && a.coutT % vec == 0 && a.ld_out % vec == 0 && a.ldp % vec == 0 &&
           (!a.resid || a.ld_res % vec == 0) && stat_chunk_pixels(a.Ho * a.Wo) <= 64;
}

// second pass of the two-pass split-K
template <typename T>
static int launch_reduce(const ConvArgs& args, vt_stream stream) {
    if (stats_fusable(args, (int)sizeof(T))) {
        const int hw = args.Ho * args.Wo;
        const int cpx = stat_chunk_pixels(hw);
        const int chunks = (hw + cpx - 1) / cpx;
        const int cgroups = vt_cdiv(args.coutT / (16 / (int)sizeof(T)), 16);
        auto k = conv_splitk_reduce_stats_kernel<T>;
        VT_LAUNCH(k, dim3((unsigned)(args.N * chunks * cgroups)), dim3(256), stream, args, cpx, chunks, cgroups);
        g_stats_emitted = true;
        return vt_check_launch("vt_conv2d(split-K reduce + statistics)");
    }
    int64_t blocks = ((int64_t)args.M * (args.ldp / 8) + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    VT_LAUNCH(conv_splitk_reduce_kernel, dim3((unsigned)blocks), dim3(256), stream, args);
    return vt_check_launch("vt_conv2d(split-K reduce)");
}

// The direct-to-LDS loader applies when every K-step stays inside one tap and one source and
// all byte offsets fit the 31-bit buffer range; everything else runs the register-staged kernel.
template <typename T>
static bool glds_eligible(const ConvArgs& a, GldsArgs& g) {
    constexpr int ESZ = (int)sizeof(T);
    constexpr int BK = 8 * (16 / ESZ);
    if (a.force_generic || a.transposed || a.in_scale) return false;
    if (a.c0 % BK != 0 || a.c1 % BK != 0) return false;
    const int64_t lim = ((int64_t)1 << 31) - 4096;
    const int64_t px = (int64_t)a.N * a.H * a.W;
    const int64_t b0 = ((int64_t)a.pad * a.W + a.pad) * a.ld0 * ESZ;
    const int64_t b1 = ((int64_t)a.pad * a.W + a.pad) * a.ld1 * ESZ;
    const int64_t n0 = px * a.ld0 * ESZ + b0, n1 = px * a.ld1 * ESZ + b1;
    const int64_t nw = (int64_t)a.coutT * a.K * ESZ;
    // largest soffset: last tap + all channels
    const int64_t tapmax = ((int64_t)(a.taps / a.kw - 1) * a.dil * a.W + (int64_t)(a.kw - 1) * a.dil);
    if (n0 + tapmax * a.ld0 * ESZ >= lim || n1 + tapmax * a.ld1 * ESZ >= lim || nw >= lim) return false;
    g.nrec0 = (uint32_t)n0;
    g.nrec1 = (uint32_t)n1;
    g.nrecw = (uint32_t)nw;
    g.bias0 = (uint32_t)b0;
    g.bias1 = (uint32_t)b1;
    return true;
}

// split-K finishing mode.  Two forms, both deterministic (slabs are summed in slice order):
//   * two-pass (default): a separate conv_splitk_reduce_kernel;
//   * in-launch (VT_SPLITK_IN_LAUNCH=1, A/B runs): the slice that arrives last at the tile's ticket reduces.
//     Measured slower on MI355X for EVERY shape of the frame: 2x for the wide convs (64-128 KB of slabs per
//     tile; the per-workgroup agent-scope release writes back the XCD's dirty L2 lines, ~6 us each), and for the
//     thin ones too (masks / ToRGB / fusion_skip, a few KB of slabs per tile: 17.6 -> 19.4, 9.9 -> 12.6,
//     17.9 -> 27.6 us per conv against slices + reduce kernel; whole frame -3 %, same-box A/B, round 2).
// An explicit two-pass call (vt_conv_desc.splitk_phase 1 / 2) never uses tickets.  vt_conv2d_splitk_mode()
// reports the choice.
static bool in_launch_rule(const ConvArgs& a, int64_t ntile) {
    if (a.phase != 0 || ntile * 4 > VT_TICKET_BYTES) return false;
    const char* e = getenv("VT_SPLITK_IN_LAUNCH");   // read per call: tests flip it at run time
    return e && e[0] == '1';
}
static void split_mode(ConvArgs& args) {
    if (args.splitk <= 1 || !in_launch_rule(args, (int64_t)args.tiles_m * args.tiles_n)) args.tickets = nullptr;
}

template <typename T, int BM, int BN, int WM, int WN>
int launch_cfg(const ConvArgs& a, vt_stream stream) {
    constexpr int BK = 8 * (16 / (int)sizeof(T));
    ConvArgs args = a;
    args.slab_perm = ((BN / WN / 16) % 2 == 0) ? 1 : 0;   // matches PERM of the kernel instance
    args.tiles_n = vt_cdiv(a.coutT, BN);
    args.tiles_m = vt_cdiv(a.M, BM);
    const int nk = vt_cdiv(a.K, BK);
    if (args.splitk > 1) {
        args.kps = vt_cdiv(nk, args.splitk);
        args.splitk = vt_cdiv(nk, args.kps);  // no empty slices
    }
    if (args.splitk <= 1) {
        args.splitk = 1;
        args.kps = nk;
    }
    split_mode(args);
    const int64_t tiles = (int64_t)vt_cdiv(a.M, BM) * args.tiles_n * args.splitk;
    if (tiles >= ((int64_t)1 << 31)) {
        vt_set_error("vt_conv2d: too many tiles");
        return VT_ERR_ARG;
    }
    GldsArgs g;
    if (args.phase == 2) {
        // reduce pass only (the slices were launched by an earlier vt_conv2d with phase 1)
    } else if (glds_eligible<T>(args, g)) {
        // Which operand every XCD re-reads.  decode_block's order is channel-major: XCD x owns a range of channel tiles and
        // ALL pixels, so the input crosses the fabric once per XCD -- 331 MB per launch for the stride-2 encoder convs against
        // 39 MB of operands (profiles/r05_pmc_traffic.json).  Pixel-major (an XCD owns a range of pixel tiles and every channel
        // tile of them) makes the WEIGHTS the re-read operand instead: chosen when they are the smaller one.  No K split only
        // (the slices of a tile are consecutive in decode_block's order).  Which workgroup computes a tile never changes its bits.
        if (args.splitk == 1 && args.tiles_n > 1 &&
            (int64_t)a.N * a.H * a.W * a.cin > (int64_t)a.coutT * a.K)
            args.blk_pm = -1;
        // As many LDS stages (K-steps of loads in flight) as still let TWO workgroups share a
        // CU's 160 KiB: measured on MI355X, occupancy 2 with 2 stages beats occupancy 1 with 3
        // (128x128: 2 stages; 128x64 / 64x128: 3; 64x64 and smaller: 4)
        constexpr int STAGE = ((BM + 31) / 32 + (BN + 31) / 32) * 32 * 128;
        constexpr int NST = STAGE * 4 <= 80 * 1024 ? 4 : STAGE * 3 <= 80 * 1024 ? 3 : 2;
        bool done = false;
        if constexpr (sizeof(T) == 4 && !is_x3<T>::value) {
            if (a.x3) {   // f32x3 instance of the same tile (conv_igemm.hip, "f32x3")
                auto k = conv_igemm_glds_kernel<f32x3_t, BM, BN, WM, WN, NST>;
                VT_LAUNCH(k, dim3((unsigned)tiles), dim3(256), stream, args, g);
                done = true;
            }
        }
        if (!done) {
            auto k = conv_igemm_glds_kernel<T, BM, BN, WM, WN, NST>;
            VT_LAUNCH(k, dim3((unsigned)tiles), dim3(256), stream, args, g);
        }
    } else {
        auto k = conv_igemm_kernel<T, BM, BN, WM, WN>;
        VT_LAUNCH(k, dim3((unsigned)tiles), dim3(256), stream, args);
    }
    int rc = vt_check_launch("vt_conv2d");
    if (rc != VT_OK || args.splitk == 1 || args.tickets || args.phase == 1) return rc;
    return launch_reduce<T>(args, stream);
}

// ---------------------------------------------------------------------------------------
// Kernel / tile / split-K selection, shared by vt_conv2d, vt_conv2d_tile and
// vt_conv2d_ws_bytes.  Every choice is a function of the PER-IMAGE geometry (Ho*Wo, cout, K)
// only -- never of the batch -- so each output element sees the same K-chunking whether its
// frame is processed alone or in a batch (bit-identical results; tests/test_engine.py).
//
// tile code (tile_hint / vt_conv2d_tile):  G*1e9 + P*1e8 + S*1e6 + BM*1e3 + BN
//   G 1 = force the register-staged loader;  P 1 = patch-resident kernel (BM = 16*TH),
//   P 2 (hint only) = never use the patch kernel;  S = split-K slices (0 = auto in a hint)
// ---------------------------------------------------------------------------------------
struct TilePlan {
    int kind;  // 0 = 1-D tile GEMM kernels, 1 = patch-resident 3x3 kernel, 3 = persistent 32->32, 4 = whole-K (conv_fullk.hpp),
               // 5 = conv_transpose + blur (conv_upblur.hpp), 6 = thin outputs (conv_thin.hpp), 7 = stride-2 3x3 by input parity
               // (conv_patch_s2.hpp; rounds 2-3: the retired persistent 64->64 kernel),
               // 8 = whole-K, weight-stationary over G tiles (conv_fullkw.hpp; bm = 64 * G)
    int bm, bn, splitk;
};

template <typename T>
static bool patch_eligible(const ConvArgs& a, GldsArgs& g) {
    constexpr int ESZ = (int)sizeof(T);
    constexpr int BK = 8 * (16 / ESZ);
    if (a.force_generic || a.transposed || a.in_scale) return false;
    if (a.taps != 9 || a.kw != 3 || a.stride != 1 || a.pad != a.dil) return false;
    if (a.dil != 1 && a.dil != 2 && a.dil != 4) return false;
    if (a.Ho != a.H || a.Wo != a.W) return false;
    if (a.c0 % BK != 0 || a.c1 % BK != 0) return false;
    const int64_t lim = ((int64_t)1 << 31) - 4096;
    const int64_t px = (int64_t)a.N * a.H * a.W;
    const int64_t n0 = px * a.ld0 * ESZ, n1 = px * a.ld1 * ESZ, nw = (int64_t)a.coutT * a.K * ESZ;
    if (n0 >= lim || n1 >= lim || nw >= lim) return false;
    g.nrec0 = (uint32_t)n0;
    g.nrec1 = (uint32_t)n1;
    g.nrecw = (uint32_t)nw;
    g.bias0 = g.bias1 = 0;
    return true;
}

// conv_transpose2d(3x3, stride 2, pad 0) by output parity on the pipelined patch tiles (conv_patch_pipe.hpp, UP = 1)
template <typename T>
static bool up_eligible(const ConvArgs& a, GldsArgs& g) {
    constexpr int ESZ = (int)sizeof(T);
    constexpr int BK = 8 * (16 / ESZ);
    if (a.force_generic || !a.transposed || a.in_scale || a.rgb_w || a.stats_part || a.tile_stats || a.in_tile_stats) return false;
    if (a.taps != 9 || a.kw != 3 || a.stride != 2 || a.pad != 0 || a.pad_x != 0 || a.dil != 1 || a.phases != 1) return false;
    if (a.Ho != 2 * a.H + 1 || a.Wo != 2 * a.W + 1 || a.c1 != 0 || a.c0 % BK != 0 || a.coutT % 8 != 0) return false;
    if (a.out_layout != VT_OUT_NHWC) return false;
    const int64_t lim = ((int64_t)1 << 31) - 4096;
    const int64_t n0 = (int64_t)a.N * a.H * a.W * a.ld0 * ESZ, nw = (int64_t)a.coutT * a.K * ESZ;
    if (n0 >= lim || nw >= lim) return false;
    g.nrec0 = (uint32_t)n0;
    g.nrec1 = 0;
    g.nrecw = (uint32_t)nw;
    g.bias0 = g.bias1 = 0;
    return true;
}

template <typename T>
static bool c32_eligible(const ConvArgs& a, GldsArgs& g) {
    if (sizeof(T) != 2 || a.force_generic || a.transposed || a.in_scale) return false;
    if (a.taps != 9 || a.kw != 3 || a.stride != 1 || a.pad != 1 || a.dil != 1) return false;
    if (a.c0 != 32 || a.c1 != 0 || a.cout % 32 != 0 || a.cout > 256 || a.phases != 1 || a.Ho != a.H || a.Wo != a.W) return false;
    if (a.cout != 32 && (a.rgb_w || a.rgb_only)) return false;   // the fused ToRGB needs all channels in one workgroup
    if (a.splitk > 1) return false;
    // lean epilogue: bf16 NHWC vector stores, bias + (Leaky)ReLU * gain, optional fused ToRGB
    if (a.out_layout != VT_OUT_NHWC || a.out_f32 || !a.vec_store || a.resid || a.slope_vec || a.alpha_dev || a.post_relu) return false;
    if (a.act != VT_ACT_NONE && a.act != VT_ACT_LRELU) return false;
    const int64_t n0 = (int64_t)a.N * a.H * a.W * a.ld0 * 2;
    if (n0 >= (((int64_t)1 << 31) - 4096)) return false;
    g.nrec0 = (uint32_t)n0;
    g.nrec1 = g.nrecw = g.bias0 = g.bias1 = 0;
    return true;
}

// VT_BATCH_EXACT=1: no plan choice that makes a frame inside a batch differ (in rounding) from the frame alone
static bool batch_exact() {
    const char* be = getenv("VT_BATCH_EXACT");   // read per call: tests flip it
    return be && be[0] == '1';
}

// persistent workgroups of the weights-resident form: one per CU.  VT_PATCHW_WGS (read per call): tests use a few workgroups
// so that small convolutions walk several tiles each
static int device_cus() {   // compute units of the current device (256 on MI355X), read once per device
#ifdef VT_EMU
    return 256;
#else
    static int cus[16] = {0};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return 256;
    if (!cus[dev]) {
        int n = 0;
        cus[dev] = (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && n > 0) ? n : 256;
    }
    return cus[dev];
#endif
}
static int patchw_wgs() {   // persistent workgroups: one per compute unit (not the constant 256)
    const char* e = getenv("VT_PATCHW_WGS");
    return e && atoi(e) > 0 ? atoi(e) : device_cus();
}

template <typename T>
static TilePlan choose_plan(const ConvArgs& a, int hint, int64_t ws_floats_avail) {
    constexpr int BK = 8 * (16 / (int)sizeof(T));
    constexpr int wg_target = 256;   // workgroups a split-K launch aims for: one per CU
    TilePlan t;
    t.kind = 0;
    t.bm = t.bn = 0;
    t.splitk = 0;
    if (a.up_fir) {   // conv_transpose2d + blur: its own kernel family; 16-channel tiles when 32 would leave CUs idle
        t.kind = 5;
        t.bm = 20 * 28;
        const int64_t tiles = (int64_t)vt_cdiv(2 * a.H, 20) * vt_cdiv(2 * a.W, 28);
        // (the batch counts: tile width changes which workgroup computes a channel, not the order of its sum -- results are
        // bit-identical, tests/test_ops.py -- and at 4 frames the deepest level fills the GPU with 32-channel tiles, which
        // read every patch half as often: 99 -> 81 us)
        t.bn = (hint % 1000 == 16 || (hint % 1000 != 32 && a.N * tiles * vt_cdiv(a.coutT, 32) < 256 && a.coutT >= 32)) ? 16 : 32;
        t.splitk = 1;
        return t;
    }
    const int hp = (hint / 100000000) % 10;
    const int hs = (hint / 1000000) % 100, hbm = (hint / 1000) % 1000, hbn = hint % 1000;
    {
        // conv_transpose2d(3x3, stride 2) with wide channels: by output parity on the pipelined patch tiles
        // (256 quads x 64 channels x 4 parity classes per workgroup); anything else transposed: the gather form below
        GldsArgs gu;
        if (hp != 2 && hbm == 0 && hs == 0 && a.coutT >= 64 && a.cin >= 128 && up_eligible<T>(a, gu)) {
            t.kind = 1;
            t.bm = 256, t.bn = 64;
            t.splitk = 1;
            return t;
        }
    }
    {
        // thin outputs (cout <= 3, planar): one launch, K over the wavefronts of a workgroup, no slabs
        if ((hp == 6 || (hp == 0 && hbm == 0 && hs == 0)) && thin_eligible<T>(a)) {
            t.kind = 6;
            t.bm = thin16_wanted(a) ? TH16 * TH16 : TH_TW * TH_TW;   // (16 x 16-pixel tiles at the large levels, conv_thin.hpp)
            t.bn = a.taps * a.coutT > 16 ? 32 : 16;
            t.splitk = 1;
            return t;
        }
    }
    const int m1 = a.Ho * a.Wo;  // rows of one image
    auto tiles = [&](int m, int n) { return (int64_t)vt_cdiv(m1, m) * vt_cdiv(a.coutT, n); };
    auto ptiles = [&](int th, int n) {
        return (int64_t)vt_cdiv(a.Ho, th) * vt_cdiv(a.Wo, 16) * vt_cdiv(a.coutT, n);
    };
    GldsArgs g;
    if constexpr (sizeof(T) == 2) {
        // stride-2 3x3 convs (the encoder's down-sampling convs) by input parity on patch-resident tiles (conv_patch_s2.hpp):
        // from 128 tiles up -- per IMAGE under VT_BATCH_EXACT (its K order is not the 1-D kernel's, so the choice must then not
        // look at the batch), per launch otherwise (like the other batch-aware plans, section 4.1h of DESIGN.md): the 64^2 ->
        // 32^2 conv of the deepest stage has 32 tiles per image and takes this kernel from 4 frames up, the 1-D tiles with a
        // K split below
        GldsArgs gs;
        if ((hp == 7 || (hp == 0 && hbm == 0 && hs == 0 && (batch_exact() ? 1 : a.N) * ptiles(16, 64) >= 128)) &&
            patchs2_eligible<T>(a, gs)) {
            t.kind = 7;
            // 32-channel tiles where 64-channel ones leave CUs idle (same K order: the same bits at both widths)
            t.bm = 256, t.bn = (hbn == 32 || (hbn != 64 && (batch_exact() ? 1 : a.N) * ptiles(16, 64) < 256)) ? 32 : 64;
            t.splitk = 1;
            return t;
        }
    }
    if (hp != 2 && hbm == 0 && c32_eligible<T>(a, g)) {   // the 1024^2 level: persistent register-weight kernel
        t.kind = 3;
        t.bm = 256;
        t.bn = 32;
        t.splitk = 1;
        return t;
    }
    {
        // whole-K kernel (conv_fullk.hpp): the layers the heuristics below would cut along K into fp32 slabs --
        // few pixels per image, wide channels.  Per-image geometry only (batch-invariant like every plan).
        FullkArgs fg;
        const bool hinted = hp == 4;
        if (hp != 2 && (hbm == 0 || hinted) && fullk_eligible<T>(a, a.wstream, fg)) {
            const int64_t wgs = (int64_t)a.dil * a.dil * fg.tiles_y * fg.tiles_x * vt_cdiv(a.coutT, FK_BN);   // per image
            constexpr int fk_max_wgs = 1024;   // largest per-image grid the heuristic gives the whole-K kernel
            // A batch that fills the GPU with 256-pixel x 128-channel patch tiles (one 8-wave workgroup per CU, no K split) is
            // better served by them: 512 -> 512 @64^2 111 -> 82 us, 1024 -> 512 @64^2 250 -> 161 us at 4 frames
            // (profiles/r03_batch_tiles.txt).  The patch kernel sums K in another order than the whole-K kernels, so a frame
            // inside such a batch equals the frame alone to rounding, not to the bit: VT_BATCH_EXACT=1 keeps the per-image
            // choice (read per call).
            const bool batch_patch = !hinted && hbm == 0 && hp == 0 && a.N > 1 && a.dil == 1 && a.coutT >= 128 && !a.tile_stats &&
                                     !a.in_tile_stats && !a.stats_part && (int64_t)a.N * ptiles(16, 128) >= 256 &&
                                     !batch_exact() && patch_eligible<T>(a, g);
            if (batch_patch) {
                t.kind = 1;
                t.bm = 256, t.bn = 128;
                t.splitk = 1;
                return t;
            }
            // Round 4: the same for the 32 x 32 trunk, where 256 x 128 tiles would leave three CUs in four idle: 256-pixel x
            // 32-channel tiles of the pipelined patch kernel (conv_patch_pipe.hpp; one workgroup per CU at 4 frames, weights
            // 8 taps ahead).  A workgroup then ingests 77 KB per 64-channel chunk -- the 41 KB patch once, not the 61 KB per
            // 32-pixel step of the weight-stationary kernel: 28.7 -> 24.5 us per conv before tuning, 66 -> 44 us for the
            // 1024 -> 512 fusion conv (profiles/r04_patch_pipeline.txt).  bf16 only (fp32 is MFMA-bound either way and keeps
            // the per-image plans of the parity mode); same rule as above for VT_BATCH_EXACT.
            // (planes of at most 48 x 48 pixels -- the trunk of frames up to 1536^2: the measured case; two frames of a
            // 64 x 64-pixel level would qualify by tile count, but were never measured against the weight-stationary kernel)
            // (round 5: the dilation-2 convs of the AdaResBlocks too -- conv_patch_pipe.hpp, DIL = 2: 32.3 -> 2x us at 4 frames;
            // only in the pipelined form, so not under VT_PATCH_PIPE=0)
            const char* ppe = getenv("VT_PATCH_PIPE");
            // (round 6: dilation 4 on flat 8 x 8 blocks of its 16 sub-images was built and measured -- 30.3 us against the 29.1 of the
            // weight-stationary kernel, same box, profiles/r06_ab_flat8_c32.txt -- and removed: DESIGN.md 4.1u)
            const bool dil_ok = a.dil == 1 || (a.dil == 2 && !(ppe && ppe[0] == '0') && !a.x3);
            const bool batch_patch32 = sizeof(T) == 2 && !hinted && hbm == 0 && hp == 0 && a.N > 1 && dil_ok && m1 <= 2304 &&
                                       a.coutT >= 128 && !a.tile_stats && !a.in_tile_stats && !a.stats_part &&
                                       (int64_t)a.N * ptiles(16, 32) >= 256 && !batch_exact() && patch_eligible<T>(a, g);
            if (batch_patch32) {
                t.kind = 1;
                t.bm = 256, t.bn = 32;
                t.splitk = 1;
                return t;
            }
            if (hinted || (a.coutT >= 128 && wgs <= fk_max_wgs)) {
                t.kind = 4;
                t.bm = FK_TH * FK_TW;
                t.bn = FK_BN;
                t.splitk = 1;
                // a batch of frames: the weight-stationary form (conv_fullkw.hpp), G tiles of one image per workgroup.
                // The ONE plan choice that looks at the batch -- allowed because the two kernels are bit-identical per
                // frame for every G (tests/test_ops.py::test_conv_weight_stationary_equals_whole_k).
                // VT_FULLKW=0 disables it; VT_FULLKW_MIN_G: smallest G that uses it (read per call: tests flip them)
                int fkw_min_g = 2;
                {
                    const char* e = getenv("VT_FULLKW");
                    const char* m = getenv("VT_FULLKW_MIN_G");
                    if (m && atoi(m) > 0) fkw_min_g = atoi(m);
                    if (e && e[0] == '0') fkw_min_g = 1 << 30;
                }
                FullkwArgs wg;
                if (fullkw_eligible<T>(a, a.wstream, wg)) {
                    const int G = fullkw_group(a.N, wg.groups_per_img, vt_cdiv(a.coutT, FK_BN));
                    if (G >= fkw_min_g) t.kind = 8, t.bm = FK_TH * FK_TW * (G > 8 ? 8 : G);
                }
                return t;
            }
        }
    }
    const bool can_patch = hp != 2 && hp != 4 && patch_eligible<T>(a, g);
    int units = 0;  // K units that can be split: K-steps (1-D) or channel chunks (patch)
    if (hbm > 0 && hp != 4) {
        t.kind = hp == 1 ? 1 : 0;
        t.bm = hbm;
        t.bn = hbn;
        t.splitk = hs;
        if (t.kind == 1 && !can_patch) t.kind = 0, t.bm = 128, t.bn = hbn >= 64 ? hbn : 64;
    } else if (can_patch) {
        // measured on MI355X (tools/conv_bench.py): 256-pixel tiles with 8 waves when they still
        // give every CU a workgroup; otherwise 128-pixel tiles.  Few tiles + wide N (the 32x32 /
        // 64x64-pixel layers): 128x128 tiles with the deep weight ring and split-K over channel
        // chunks, so a workgroup waits for memory once, not once per tap.
        t.kind = 1;
        const int units_p = a.cin / BK;
        if (a.dil == 1 && a.coutT <= 16) {
            t.bm = 128, t.bn = 16;
        } else if (a.dil == 1 && a.coutT >= 128 && ptiles(16, 128) >= 192) {
            t.bm = 256, t.bn = 128;
        } else if (a.dil == 1 && a.coutT >= 128 && (int64_t)a.N * ptiles(16, 128) >= 256 &&
                   ((ptiles(8, 128) >= 192 && ptiles(8, 64) >= 384 && !a.rgb_w) || !batch_exact())) {
            // a batch fills the GPU with the 256 x 128 tiles where one frame would not: 256 -> 256 @128^2 117 -> 92 us,
            // 512 -> 256 @128^2 220 -> 169 us at 4 frames.  Where one frame alone takes the 128 x 64 tiles without a K split
            // (the first pair of conditions) the sum order is the same and so are the bits (tests/test_ops.py); elsewhere this
            // is a choice only a batch gets unless VT_BATCH_EXACT=1.  (A conv asked to fuse its ToRGB is "elsewhere": whether the
            // tile holds all channels decides whether the caller fuses at all, and the fused epilogue reads the activations
            // before they are rounded to the storage type.)
            t.bm = 256, t.bn = 128;
            t.splitk = 1;
        } else if (a.dil == 1 && a.coutT == 64 && ptiles(16, 64) >= 192) {
            t.bm = 256, t.bn = 64;
        } else if (a.coutT >= 128 && ptiles(8, 128) < 192 &&
                   (a.dil == 1 || ptiles(8, 128) * units_p <= 1024)) {
            t.bm = 128, t.bn = 128;
            int64_t sk = (wg_target + ptiles(8, 128) - 1) / ptiles(8, 128);
            if (sk > units_p || a.dil != 1) sk = units_p;   // dilated instances: one chunk per slice
            if (sk > 32) sk = 32;
            t.splitk = (int)sk;
            // the dilated instances exist only in the one-chunk-per-slice form: they need the full
            // split, i.e. a workspace that can hold it -- otherwise run the 1-D kernel
            const int64_t need_full = (int64_t)units_p * a.M * slab_ld(a.coutT);
            if (a.dil != 1 && (sk != units_p || (units_p > 1 && (!a.partial || need_full > ws_floats_avail))))
                t.kind = 0, t.bm = 0, t.splitk = 0;
        } else if (a.dil == 1 && a.coutT >= 64) {
            t.bm = 128, t.bn = 64;
        } else {
            t.kind = 0;  // no patch instance for this shape: use the 1-D kernels
        }
    }
    if (t.kind == 0 && t.bm == 0) {
        // 1-D tiles: keep >= ~2 workgroups per CU (256 CUs) where the problem allows it
        t.bn = a.coutT <= 16 ? 16 : a.coutT <= 32 ? 32 : a.coutT <= 64 ? 64 : 128;
        t.bm = 128;
        if (t.bn == 128) {
            if (tiles(128, 128) < 512) {
                t.bm = 64;
                if (tiles(64, 128) < 512 && tiles(64, 64) >= 128) t.bn = 64;
            }
        } else if (t.bn == 64) {
            if (tiles(128, 64) < 512) t.bm = 64;
        }
    }
    int64_t ntiles;
    if (t.kind == 1) {
        units = a.cin / BK;
        ntiles = ptiles(t.bm / 16, t.bn);
    } else {
        units = vt_cdiv(a.K, BK);
        ntiles = tiles(t.bm, t.bn);
    }
    if (t.splitk == 0) {
        // too few tiles to fill 256 CUs: cut K so that ~512 workgroups exist
        t.splitk = 1;
        const int min_units = t.kind == 1 ? 1 : 2;
        const int64_t enough = (t.kind == 1 && t.bm == 256 ? 192 : 384) * wg_target / 256;  // 8-wave tiles: 1 WG fills a CU
        if (ntiles < enough && units >= 2 * min_units) {
            int64_t sk = (2 * wg_target + ntiles - 1) / ntiles;
            if (sk > units / min_units) sk = units / min_units;
            if (sk > 32) sk = 32;
            t.splitk = (int)sk;
        }
    }
    if (t.splitk > units) t.splitk = units;
    if (t.splitk > 1) {
        const int64_t need = (int64_t)t.splitk * a.M * slab_ld(a.coutT);
        if (!a.partial || need > ws_floats_avail) t.splitk = 1;  // no workspace: single pass
    }
    if (t.splitk < 1) t.splitk = 1;
    // normalise so that no slice is empty (what the launch will use)
    const int per = vt_cdiv(units, t.splitk);
    t.splitk = vt_cdiv(units, per);
    return t;
}

template <typename T, int TH, int BN, int WM, int WN, int DIL, int NSTB, int ABUF>
int launch_patch(const ConvArgs& a, const GldsArgs& g, vt_stream stream) {
    constexpr int BK = 8 * (16 / (int)sizeof(T));
    ConvArgs args = a;
    args.slab_perm = ((BN / WN / 16) % 2 == 0) ? 1 : 0;   // matches PERM of the kernel instance
    args.tiles_n = vt_cdiv(a.coutT, BN);
    args.tiles_m = a.N * vt_cdiv(a.Ho, TH) * vt_cdiv(a.Wo, 16);
    const int units = a.cin / BK;
    args.kps = vt_cdiv(units, args.splitk);
    args.splitk = vt_cdiv(units, args.kps);
    split_mode(args);
    const int64_t blocks = (int64_t)args.tiles_m * args.tiles_n * args.splitk;
    if (blocks >= ((int64_t)1 << 31)) {
        vt_set_error("vt_conv2d: too many tiles");
        return VT_ERR_ARG;
    }
    if (ABUF == 1 && args.kps != 1) {
        vt_set_error("vt_conv2d: single-buffer patch instance needs one chunk per K slice");
        return VT_ERR_UNSUPPORTED;
    }
    if (args.phase != 2) {
        bool done = false;
        if constexpr (sizeof(T) == 4 && !is_x3<T>::value && DIL == 1) {
            if (a.x3) {   // f32x3 instance of the same tile
                auto k = conv_patch_kernel<f32x3_t, TH, BN, WM, WN, DIL, NSTB, ABUF>;
                VT_LAUNCH(k, dim3((unsigned)blocks), dim3(WM * WN * 64), stream, args, g);
                done = true;
            }
        }
        if (!done) {
            auto k = conv_patch_kernel<T, TH, BN, WM, WN, DIL, NSTB, ABUF>;
            VT_LAUNCH(k, dim3((unsigned)blocks), dim3(WM * WN * 64), stream, args, g);
        }
    }
    int rc = vt_check_launch("vt_conv2d(patch)");
    if (rc != VT_OK || args.splitk == 1 || args.tickets || args.phase == 1) return rc;
    return launch_reduce<T>(args, stream);
}

// the convolutions whose epilogue is the lean path of conv_epilogue (same predicate as there): kernels instantiated with EPI = 1
// carry nothing else
template <typename T>
static bool conv_lean(const ConvArgs& a) {
    constexpr bool H = sizeof(T) == 2;
    return (a.act == VT_ACT_NONE || a.act == VT_ACT_LRELU) && a.phases == 1 && a.out_layout == VT_OUT_NHWC &&
           (H ? !a.out_f32 : a.out_f32 != 0) && a.vec_store && !a.post_relu && !(a.coutT & 7) &&
           (H ? (!a.resid || !(a.ld_res & 7)) && !(a.ld_out & 7) : !(a.ld_out & 3) && (!a.resid || !(a.ld_res & 3)));
}

// persistent form of the pipelined patch tiles (conv_patch_persist.hpp): launches of more tiles than CUs, whole K, lean epilogue
template <typename T, int TH, int BN, int WM, int WN, int NSTB = 4>
int launch_patchq(const ConvArgs& a, const GldsArgs& g, vt_stream stream) {
    ConvArgs args = a;
    args.slab_perm = 1;
    args.tiles_n = vt_cdiv(a.coutT, BN);
    args.tiles_m = a.N * vt_cdiv(a.Ho, TH) * vt_cdiv(a.Wo, 16);
    args.kps = a.cin / (8 * (16 / (int)sizeof(T)));
    args.splitk = 1;
    split_mode(args);
    const int64_t units = (int64_t)args.tiles_m * args.tiles_n;
    const int wgs = units < patchw_wgs() ? (int)units : patchw_wgs();
    auto k = conv_patchq_kernel<T, TH, BN, WM, WN, NSTB>;
    VT_LAUNCH(k, dim3((unsigned)wgs), dim3(WM * WN * 64), stream, args, g);
    return vt_check_launch("vt_conv2d(patch, pipelined, persistent)");
}

// block of tiles per XCD for decode_block_2d: exactly 8 blocks of pm x cn tiles, minimal patch + weight bytes per XCD
static void xcd_block(ConvArgs& args, int64_t blocks, int64_t patch_bytes, int64_t wtile_bytes) {
    args.blk_pm = args.blk_cn = 0;
    if (args.splitk != 1 || blocks < 8 || blocks % 8 != 0) return;
    const int per = (int)(blocks / 8);
    int64_t best = -1;
    for (int cn = 1; cn <= args.tiles_n; ++cn) {
        if (args.tiles_n % cn != 0 || per % cn != 0) continue;
        const int pm = per / cn;
        if (pm < 1 || args.tiles_m % pm != 0) continue;
        if ((int64_t)(args.tiles_m / pm) * (args.tiles_n / cn) != 8) continue;
        const int64_t cost = pm * patch_bytes + cn * wtile_bytes;
        if (best < 0 || cost < best) best = cost, args.blk_pm = pm, args.blk_cn = cn;
    }
}

template <typename T, int TH, int BN, int WM, int WN, int NSTB = 4, int UP = 0, int DIL = 1>
int launch_patchp(const ConvArgs& a, const GldsArgs& g, vt_stream stream) {
    constexpr int BK = 8 * (16 / (int)sizeof(T));
    ConvArgs args = a;
    args.slab_perm = ((BN / WN / 16) % 2 == 0) ? 1 : 0;   // matches PERM of the kernel instance
    args.tiles_n = vt_cdiv(a.coutT, BN);
    args.tiles_m = UP ? a.N * vt_cdiv(a.H + 1, TH) * vt_cdiv(a.W + 1, 16) : a.N * vt_cdiv(a.Ho, TH) * vt_cdiv(a.Wo, 16);
    const int units = a.cin / BK;
    args.kps = vt_cdiv(units, args.splitk);
    args.splitk = vt_cdiv(units, args.kps);
    split_mode(args);
    const int64_t blocks = (int64_t)args.tiles_m * args.tiles_n * args.splitk;
    if (blocks >= ((int64_t)1 << 31)) {
        vt_set_error("vt_conv2d: too many tiles");
        return VT_ERR_ARG;
    }
    if (UP == 0)
        xcd_block(args, blocks, (int64_t)(TH + 2 * DIL) * (16 + 2 * DIL) * a.cin * (int)sizeof(T), (int64_t)BN * a.K * (int)sizeof(T));
    if (args.phase != 2) {
        bool done = false;
        if constexpr (UP == 0 && (BN / WN / 16) % 2 == 0) {
            if (args.splitk == 1 && conv_lean<T>(args)) {   // the lean-epilogue instance: a fifth of the code
                auto k = conv_patchp_kernel<T, TH, BN, WM, WN, NSTB, UP, 1, DIL>;
                VT_LAUNCH(k, dim3((unsigned)blocks), dim3(WM * WN * 64), stream, args, g);
                done = true;
            }
        }
        if (!done) {
            auto k = conv_patchp_kernel<T, TH, BN, WM, WN, NSTB, UP, 0, DIL>;
            VT_LAUNCH(k, dim3((unsigned)blocks), dim3(WM * WN * 64), stream, args, g);
        }
    }
    int rc = vt_check_launch("vt_conv2d(patch, pipelined)");
    if (rc != VT_OK || args.splitk == 1 || args.tickets || args.phase == 1) return rc;
    return launch_reduce<T>(args, stream);
}

// one barrier per K chunk (conv_patch_chunk.hpp): the 32-channel tiles of the trunk.  Same tiles, same split rules, same bits as
// launch_patchp<T, 16, 32, 8, 1, ...>
template <typename T, int DIL = 1>
int launch_patchc(const ConvArgs& a, const GldsArgs& g, vt_stream stream) {
    constexpr int TH = 16, BN = 32, WM = 8, WN = 1;
    constexpr int BK = 8 * (16 / (int)sizeof(T));
    ConvArgs args = a;
    args.slab_perm = ((BN / WN / 16) % 2 == 0) ? 1 : 0;
    args.tiles_n = vt_cdiv(a.coutT, BN);
    // (DIL > 1: tiles over the DIL x DIL sub-images of every image, conv_patch_chunk.hpp)
    args.tiles_m = a.N * DIL * DIL * vt_cdiv(vt_cdiv(a.Ho, DIL), TH) * vt_cdiv(vt_cdiv(a.Wo, DIL), 16);
    const int units = a.cin / BK;
    args.kps = vt_cdiv(units, args.splitk);
    args.splitk = vt_cdiv(units, args.kps);
    split_mode(args);
    const int64_t blocks = (int64_t)args.tiles_m * args.tiles_n * args.splitk;
    if (blocks >= ((int64_t)1 << 31)) {
        vt_set_error("vt_conv2d: too many tiles");
        return VT_ERR_ARG;
    }
    xcd_block(args, blocks, (int64_t)(TH + 2) * (16 + 2) * a.cin * (int)sizeof(T), (int64_t)BN * a.K * (int)sizeof(T));
    if (args.phase != 2) {
        if (args.splitk == 1 && conv_lean<T>(args)) {
            auto k = conv_patchc_kernel<T, TH, BN, WM, WN, 1, DIL>;
            VT_LAUNCH(k, dim3((unsigned)blocks), dim3(WM * WN * 64), stream, args, g);
        } else {
            auto k = conv_patchc_kernel<T, TH, BN, WM, WN, 0, DIL>;
            VT_LAUNCH(k, dim3((unsigned)blocks), dim3(WM * WN * 64), stream, args, g);
        }
    }
    int rc = vt_check_launch("vt_conv2d(patch, chunk barriers)");
    if (rc != VT_OK || args.splitk == 1 || args.tickets || args.phase == 1) return rc;
    return launch_reduce<T>(args, stream);
}

// weights-resident persistent form (conv_patch_resident.hpp): one chunk of K, one channel tile, no split
template <typename T, int TH, int BN>
int launch_patchw(const ConvArgs& a, const GldsArgs& g, vt_stream stream) {
    ConvArgs args = a;
    args.slab_perm = ((BN / 16) % 2 == 0) ? 1 : 0;
    args.tiles_n = 1;
    args.tiles_m = a.N * vt_cdiv(a.Ho, TH) * vt_cdiv(a.Wo, 16);
    args.kps = 1;
    args.splitk = 1;
    split_mode(args);
    int wgs = patchw_wgs();
    if (wgs > args.tiles_m) wgs = args.tiles_m;
    if (wgs >= 8) wgs &= ~7;                    // whole XCD rounds: the kernel hands out tiles per XCD
    auto k = conv_patchw_kernel<T, TH, BN>;
    VT_LAUNCH(k, dim3((unsigned)wgs), dim3(512), stream, args, g);
    return vt_check_launch("vt_conv2d(patch, weights resident)");
}

template <typename T>
int launch_c32(const ConvArgs& a, const GldsArgs& g, vt_stream stream) {
    ConvArgs args = a;
    args.slab_perm = 0;
    args.splitk = 1;
    args.tiles_n = 1;
    args.tiles_m = a.N * vt_cdiv(a.Ho, 16) * vt_cdiv(a.Wo, 16);
    const int groups = a.cout / 32;                          // blockIdx.y: 32 output channels each
    const int per_group = 512 / groups > 0 ? 512 / groups : 1;
    int blocks = args.tiles_m < per_group ? args.tiles_m : per_group;   // persistent: 2 workgroups per CU in all
    if (const char* e = getenv("VT_C32_BLOCKS")) {          // tests: force several tiles per workgroup
        const int v = atoi(e);
        if (v > 0 && v < blocks) blocks = v;
    }
#ifdef VT_EMU
    auto k = conv3x3_c32_kernel<bf16_t>;
#else
    auto k = conv3x3_c32_kernel<bf16_t>;
#endif
    (void)sizeof(T);
    VT_LAUNCH(k, dim3((unsigned)blocks, (unsigned)groups), dim3(256), stream, args, g);
    return vt_check_launch("vt_conv2d(c32)");
}

template <typename T>
int dispatch(const ConvArgs& a0, int hint, int64_t ws_floats, vt_stream stream) {
    ConvArgs a = a0;
    a.force_generic = hint >= 1000000000 || a.pad_x != a.pad;
    const TilePlan t = choose_plan<T>(a, hint % 1000000000, ws_floats);
    a.splitk = t.splitk;
    a.ldp = slab_ld(a.coutT);
    if (a.rgb_w && (t.bn < a.coutT || t.splitk > 1)) {
        vt_set_error("vt_conv2d: fused ToRGB needs all %d output channels in one tile (plan %dx%d, split %d)",
                     a.coutT, t.bm, t.bn, t.splitk);
        return VT_ERR_UNSUPPORTED;
    }
    if (a.in_absdiff && t.kind != 6) {
        vt_set_error("vt_conv2d: in_absdiff needs the thin-output kernel (plan kind %d)", t.kind);
        return VT_ERR_UNSUPPORTED;
    }
    if (a.rgb_only && t.kind != 3) {
        vt_set_error("vt_conv2d: rgb_only needs the persistent 32 -> 32 kernel (plan kind %d)", t.kind);
        return VT_ERR_UNSUPPORTED;
    }
    if (t.kind == 5) {
        // deep layers (>= 4 channel chunks): double-buffered chunks, one workgroup per CU; shallow ones: single
        // stage, two workgroups per CU
        const int chunks = a.cin / (8 * (16 / (int)sizeof(T)));
        // double-buffered chunks (one workgroup per CU) only pay on the 16-channel tiles of the deepest layer
        // (33 vs 37 us); the 32-channel tiles run single-stage with two workgroups per CU (44 vs 51 us at 512->256)
        const char* de = getenv("VT_UPBLUR_DB");    // A/B: minimum chunk count of the double-buffered form
        // ... and on 32-channel tiles of launches of at most three rounds of one workgroup per CU (the deepest level at 4 frames:
        // 768 workgroups, 81 vs 85 us)
        const int64_t wgs_all = (int64_t)a.N * vt_cdiv(2 * a.H, 20) * vt_cdiv(2 * a.W, 28) * vt_cdiv(a.coutT, t.bn);
        const bool db = de ? chunks >= atoi(de) : (chunks >= 4 && (t.bn == 16 || wgs_all <= 768));
        // (single-stage 32-channel forms are capped at 256 registers -- 2 workgroups per CU, a few cold values spilled: 40 vs
        // 51 us and 54 vs 77 us on the 256^2 / 512^2-pixel levels)
        if constexpr (sizeof(T) == 2) {
            // the two top levels (Cin <= 128, >= 128^2 input pixels): one wave per strip, horizontal blur on the matrix cores,
            // no z tile (conv_upblur_rows.hpp).  Other bits than the tile kernels below, so the choice is by shape only.
            if (uprows_wanted<T>(a)) return launch_uprows<T>(a, stream);
            // the deep levels (Cin >= 256): flattened 10 x 34-quad tiles, K ring of 32-channel steps (conv_upblur_flat.hpp).
            // The bits of the tile kernels below (same K order, same blur), so the choice may depend on the batch.
            if (upflat_wanted<T>(a)) return launch_upflat<T>(a, stream);
        }
        if constexpr (sizeof(T) == 2) {
            // single-chunk layers (the 1024^2 level) with >= 4 tiles per CU: persistent 8-wave workgroups on 16 x 16-quad tiles
            // -- the 36 KB of weights stay in LDS instead of being re-fetched by every tile (more than the tile's 28 KB
            // patch), the next patch flies during the blur, 8 waves keep the CU as busy as two 4-wave workgroups did.
            // 264 -> 218 us at 4 frames (the 4-wave persistent form of round 2 lost to two plain workgroups per CU: 277).
            // Same bits (tile shape only; tests/test_ops.py).  VT_UPBLUR_P8=0 / 1: never / always
            const char* p8 = getenv("VT_UPBLUR_P8");
            const int64_t tiles8 = (int64_t)a.N * vt_cdiv(2 * a.H, 28) * vt_cdiv(2 * a.W, 28) * vt_cdiv(a.coutT, 32);
            const bool on = p8 ? p8[0] == '1' : tiles8 >= 1024;
            if (on && t.bn == 32 && chunks == 1) return launch_upblur<T, 32, 16, 0, 1, 0, 8>(a, stream);
        }
        {
            // tall tiles (24 x 16 quads, 8 waves, one workgroup per CU) where the layer is bound by L2 -> LDS bytes -- four or
            // more channel chunks -- and still gives every CU ~2 workgroups: 125 -> 100 us (512 -> 256) and 132 -> 119 us
            // (256 -> 128) at 4 frames; 128 -> 64 loses (152 -> 162: two resident workgroups hide its blur phase better).
            // Same bits as the 12-row tiles (tests/test_ops.py).
            // VT_UPBLUR_TALL = minimum workgroup count of the tall form (0 = never; the tests pass 1 and a 2-chunk layer)
            const char* te = getenv("VT_UPBLUR_TALL");
            const int64_t tall_min = te ? atoll(te) : 448;
            const int64_t wgs_tall = (int64_t)a.N * vt_cdiv(2 * a.H, 44) * vt_cdiv(2 * a.W, 28) * vt_cdiv(a.coutT, 32);
            if constexpr (sizeof(T) == 2) {   // (the fp32 z tile of 47 x 31 pixels does not fit the LDS)
                if (t.bn == 32 && !db && chunks >= (te ? 2 : 4) && tall_min > 0 && wgs_tall >= tall_min)
                    return launch_upblur<T, 32, 24, 0, 0, 0, 8>(a, stream);
            }
        }
        if (t.bn == 16) return db ? launch_upblur<T, 16, 12, 1, 0, 0>(a, stream) : launch_upblur<T, 16, 12, 0, 0, 0>(a, stream);
        if (db) return launch_upblur<T, 32, 12, 1, 0, 0>(a, stream);
        return launch_upblur<T, 32, 12, 0, 0, 1>(a, stream);
    }
    if ((a.tile_stats || a.in_tile_stats) && t.kind != 4 && t.kind != 8) {
        vt_set_error("vt_conv2d: tile_stats / in_tile_stats need the whole-K kernel (plan kind %d)", t.kind);
        return VT_ERR_UNSUPPORTED;
    }
    if (a.tile_stats && !(a.out_layout == VT_OUT_NHWC && a.vec_store && a.out_f32 == (sizeof(T) == 4) && a.coutT % 8 == 0)) {
        vt_set_error("vt_conv2d: tile_stats needs an NHWC output in the compute dtype with 16-byte aligned pixels");
        return VT_ERR_UNSUPPORTED;
    }
    if (a.in_tile_stats && a.c1 != 0) {
        vt_set_error("vt_conv2d: in_tile_stats (AdaIN prologue) supports a single source");
        return VT_ERR_UNSUPPORTED;
    }
    if (t.kind == 6) return launch_thin<T>(a, stream);
    if (t.kind == 7) {
        GldsArgs gs;
        if constexpr (sizeof(T) == 2) {
            if (patchs2_eligible<T>(a, gs)) return t.bn == 32 ? launch_patchs2<T, 32>(a, gs, stream) : launch_patchs2<T, 64>(a, gs, stream);
        }
        vt_set_error("vt_conv2d: stride-2 patch kernel requested for an ineligible convolution");
        return VT_ERR_UNSUPPORTED;
    }
    if (t.kind == 8) {
        FullkwArgs wg;
        if (!fullkw_eligible<T>(a, a.wstream, wg)) {
            vt_set_error("vt_conv2d: weight-stationary whole-K kernel requested for an ineligible convolution");
            return VT_ERR_UNSUPPORTED;
        }
        return launch_fullkw<T>(a, wg, t.bm / (FK_TH * FK_TW), stream);
    }
    if (t.kind == 4) {
        FullkArgs fg;
        if (!fullk_eligible<T>(a, a.wstream, fg)) {
            vt_set_error("vt_conv2d: whole-K kernel requested for an ineligible convolution");
            return VT_ERR_UNSUPPORTED;
        }
        return launch_fullk<T>(a, fg, stream);
    }
    if (t.kind == 3) {
        GldsArgs g;
        if (!c32_eligible<T>(a, g)) {
            vt_set_error("vt_conv2d: c32 kernel requested for an ineligible convolution");
            return VT_ERR_UNSUPPORTED;
        }
        return launch_c32<T>(a, g, stream);
    }
    if (t.kind == 1 && a.transposed) {
        GldsArgs g;
        if (!up_eligible<T>(a, g)) {
            vt_set_error("vt_conv2d: transposed patch plan requested for an ineligible convolution");
            return VT_ERR_UNSUPPORTED;
        }
        return launch_patchp<T, 16, 64, 4, 2, 8, 1>(a, g, stream);
    }
    if (t.kind == 1) {
        GldsArgs g;
        if (!patch_eligible<T>(a, g)) {
            vt_set_error("vt_conv2d: patch kernel requested for an ineligible convolution");
            return VT_ERR_UNSUPPORTED;
        }
        const int units = a.cin / (8 * (16 / (int)sizeof(T)));
        const bool one_chunk = vt_cdiv(units, a.splitk) == 1;
#define VT_PATCH(TH_, BN_, WM_, WN_, DIL_, NSTB_, ABUF_, COND_) \
    if (t.bm == TH_ * 16 && t.bn == BN_ && a.dil == DIL_ && (COND_))  \
        return launch_patch<T, TH_, BN_, WM_, WN_, DIL_, NSTB_, ABUF_>(a, g, stream);
        {
            // software-pipelined form of the 256-pixel tiles (conv_patch_pipe.hpp; same K order, same bits).
            // VT_PATCH_PIPE=0: the per-tap form below (A/B; read per call: tests flip it)
            const char* e = getenv("VT_PATCH_PIPE");
            // (f32x3 runs the per-tap form: a tap-granular pipelined f32x3 step -- split, LDS-DMA, next tap's raw fragments, 3 MFMAs
            // per product -- measured SLOWER than it on the 64-channel tiles, 0.81 vs 0.51 ms per 4 frames, and 149 us per trunk
            // conv on 32-channel tiles against 100 us on the whole-K kernel: with both waves of a SIMD splitting at the same
            // time right after the barrier the VALU burst is exposed; profiles/r04_f32x3.txt)
            const bool pipe = !(e && e[0] == '0') && !a.x3;
            // one chunk of K, one channel tile, several tiles per CU: weights resident, persistent workgroups (VT_PATCH_PIPE=1:
            // the plain pipelined form, A/B)
            if constexpr (sizeof(T) == 2) {
                // (lean epilogue: bf16 NHWC vector stores of all 64 channels, bias + (Leaky)ReLU * gain, optional fused ToRGB)
                const bool lean = a.coutT == 64 && a.phases == 1 && a.out_layout == VT_OUT_NHWC && !a.out_f32 && a.vec_store &&
                                  a.ld_out % 8 == 0 && !a.resid && !a.slope_vec && !a.alpha_dev && !a.post_relu && !a.stats_part &&
                                  (a.act == VT_ACT_NONE || a.act == VT_ACT_LRELU) &&
                                  (int64_t)a.N * a.Ho * a.Wo * a.ld_out * 2 < (((int64_t)1 << 31) - 4096) &&   // counted buffer stores
                                  (int64_t)a.N * a.Ho * a.Wo * 12 < (((int64_t)1 << 31) - 4096);
                if (pipe && !(e && e[0] == '1') && a.dil == 1 && t.bm == 256 && t.bn == 64 && units == 1 && lean && !a.src1 &&
                    a.splitk <= 1 &&
                    (int64_t)(batch_exact() ? 1 : a.N) * vt_cdiv(a.Ho, 16) * vt_cdiv(a.Wo, 16) >= 2 * patchw_wgs())
                    return launch_patchw<T, 16, 64>(a, g, stream);
            }
            // more tiles than CUs, whole K, lean epilogue: persistent workgroups, the pipeline runs across tile boundaries
            // (VT_PATCH_PIPE=1: one workgroup per tile, A/B).  VT_BATCH_EXACT or not: the same bits either way.
            if constexpr (sizeof(T) == 2) {
                if (pipe && !(e && e[0] == '1') && a.dil == 1 && t.bm == 256 && (t.bn == 128 || t.bn == 64) && a.splitk <= 1 &&
                    conv_lean<T>(a) && vt_cdiv(a.coutT, t.bn) * t.bn * 8 <= PQ_TAB_BYTES &&   // (the tables of every channel tile, conv_patch_persist.hpp)
                    (int64_t)a.N * vt_cdiv(a.Ho, 16) * vt_cdiv(a.Wo, 16) * vt_cdiv(a.coutT, t.bn) > patchw_wgs()) {
                    if (t.bn == 128) return launch_patchq<T, 16, 128, 4, 2, 4>(a, g, stream);
                    return launch_patchq<T, 16, 64, 4, 2, 4>(a, g, stream);
                }
            }
            if (pipe && a.dil == 1 && t.bm == 256 && t.bn == 128) return launch_patchp<T, 16, 128, 4, 2>(a, g, stream);
            if (pipe && a.dil == 1 && t.bm == 256 && t.bn == 64) return launch_patchp<T, 16, 64, 4, 2, 4>(a, g, stream);
            // 32-channel tiles: all nine taps of a chunk resident, one barrier per chunk (conv_patch_chunk.hpp; VT_PATCH_PIPE=1:
            // the tap-granular pipeline, A/B -- same bits)
            if constexpr (sizeof(T) == 2) {
                if (pipe && !(e && e[0] == '1') && a.dil == 1 && t.bm == 256 && t.bn == 32) return launch_patchc<T>(a, g, stream);
                if (pipe && !(e && e[0] == '1') && a.dil == 2 && t.bm == 256 && t.bn == 32 && !a.stats_part)
                    return launch_patchc<T, 2>(a, g, stream);
            }
            if (pipe && a.dil == 1 && t.bm == 256 && t.bn == 32) return launch_patchp<T, 16, 32, 8, 1, 8>(a, g, stream);
            if (pipe && a.dil == 2 && t.bm == 256 && t.bn == 32) return launch_patchp<T, 16, 32, 8, 1, 6, 0, 2>(a, g, stream);
        }
        VT_PATCH(16, 128, 4, 2, 1, 3, 2, true)
        VT_PATCH(16, 64, 4, 2, 1, 3, 2, true)
        VT_PATCH(8, 64, 2, 2, 1, 3, 2, true)
        VT_PATCH(8, 16, 4, 1, 1, 3, 2, true)
        // one chunk per slice = single patch buffer.  Short ring (71 KB: TWO workgroups per CU, the
        // prologue / epilogue of one overlaps the taps of the other) when the launch runs several
        // rounds of workgroups; deep ring (5 taps in flight) for the latency-bound single round.
        VT_PATCH(8, 128, 2, 2, 1, 6, 1, one_chunk)
        VT_PATCH(8, 128, 2, 2, 2, 6, 1, one_chunk)
        VT_PATCH(8, 128, 2, 2, 4, 6, 1, one_chunk)
        VT_PATCH(8, 128, 2, 2, 1, 6, 2, !one_chunk)
#undef VT_PATCH
        vt_set_error("vt_conv2d: no compiled patch tile %dx%d dil %d", t.bm, t.bn, a.dil);
        return VT_ERR_UNSUPPORTED;
    }
    const int bm = t.bm, bn = t.bn;
#define VT_CFG(M_, N_, WM_, WN_) \
    if (bm == M_ && bn == N_) return launch_cfg<T, M_, N_, WM_, WN_>(a, stream);
    VT_CFG(128, 128, 2, 2)
    VT_CFG(128, 64, 2, 2)
    VT_CFG(128, 32, 4, 1)
    VT_CFG(128, 16, 4, 1)
    VT_CFG(64, 64, 2, 2)
    VT_CFG(64, 128, 2, 2)
    VT_CFG(32, 64, 2, 2)
#undef VT_CFG
    vt_set_error("vt_conv2d: no compiled tile %dx%d", bm, bn);
    return VT_ERR_UNSUPPORTED;
}

}  // namespace

static int fill_args(const vt_conv_desc* d, ConvArgs& a) {
    VT_REQUIRE(d, "vt_conv2d: null descriptor");
    VT_REQUIRE(d->src0 && d->weight && d->out, "vt_conv2d: null tensor");
    VT_REQUIRE(d->dtype == VT_F32 || d->dtype == VT_BF16 || d->dtype == VT_F32X3, "vt_conv2d: dtype must be fp32, bf16 or f32x3");
    const int cdt = d->dtype == VT_F32X3 ? VT_F32 : d->dtype;   // storage type of src*, weight
    VT_REQUIRE(d->c0 > 0 && d->c0 % 8 == 0 && d->c1 >= 0 && d->c1 % 8 == 0,
               "vt_conv2d: channel counts must be multiples of 8 (got %d,%d)", d->c0, d->c1);
    VT_REQUIRE(d->c1 == 0 || d->src1, "vt_conv2d: c1 > 0 but src1 is null");
    VT_REQUIRE(d->ld0 >= d->c0 && (d->c1 == 0 || d->ld1 >= d->c1), "vt_conv2d: pixel stride < channels");
    const int esz = cdt == VT_F32 ? 4 : 2;
    VT_REQUIRE(((uintptr_t)d->src0 % 16 == 0) && ((int64_t)d->ld0 * esz % 16 == 0) &&
                   ((uintptr_t)d->weight % 16 == 0),
               "vt_conv2d: src0/weight must be 16-byte aligned with 16-byte pixel stride");
    VT_REQUIRE(d->c1 == 0 || (((uintptr_t)d->src1 % 16 == 0) && ((int64_t)d->ld1 * esz % 16 == 0)),
               "vt_conv2d: src1 must be 16-byte aligned with 16-byte pixel stride");
    VT_REQUIRE(d->n > 0 && d->h > 0 && d->w > 0 && d->out_h > 0 && d->out_w > 0, "vt_conv2d: bad sizes");
    VT_REQUIRE(d->kh > 0 && d->kw > 0 && d->stride > 0 && d->dil > 0 && d->cout > 0, "vt_conv2d: bad conv params");
    VT_REQUIRE(d->phases == 1 || d->phases == 4, "vt_conv2d: phases must be 1 or 4");
    VT_REQUIRE(d->phases == 1 || (d->cout % 8 == 0 && d->out_layout == VT_OUT_NHWC && !d->transposed),
               "vt_conv2d: polyphase form needs cout %% 8 == 0 and NHWC output");
    VT_REQUIRE(d->out_layout == VT_OUT_NHWC || d->out_dtype == VT_F32, "vt_conv2d: NCHW output is fp32 only");
    VT_REQUIRE(d->out_dtype == VT_F32 || d->out_dtype == VT_BF16, "vt_conv2d: bad out dtype");
    VT_REQUIRE((int64_t)d->n * d->h * d->w < ((int64_t)1 << 31) &&
                   (int64_t)d->n * d->out_h * d->out_w * (d->phases == 4 ? 4 : 1) < ((int64_t)1 << 31),
               "vt_conv2d: tensor too large for 32-bit pixel indices");

    VT_REQUIRE(!d->stats_part || (d->phases == 1 && d->out_layout == VT_OUT_NHWC && d->out_dtype == cdt &&
                                  d->cout % 8 == 0 && (cdt == VT_BF16 || cdt == VT_F32)),
               "vt_conv2d: stats_part needs phases == 1, NHWC output in the compute dtype, cout %% 8 == 0");
    VT_REQUIRE(!d->post_relu || !d->rgb_weight, "vt_conv2d: post_relu cannot be combined with the fused ToRGB");
    VT_REQUIRE(!d->rgb_weight || (d->rgb_out && d->phases == 1 && d->out_layout == VT_OUT_NHWC && !d->transposed),
               "vt_conv2d: fused ToRGB needs rgb_out, phases == 1 and NHWC output");
    memset(&a, 0, sizeof(a));
    a.src0 = d->src0;
    a.src1 = d->src1;
    a.wgt = d->weight;
    a.bias = d->bias;
    a.slope_vec = d->slope_vec;
    a.rgb_w = d->rgb_weight;
    a.rgb_bias = d->rgb_bias;
    a.rgb_resid = d->rgb_resid;
    a.rgb_out = d->rgb_out;
    a.stats_part = (StatRec*)d->stats_part;
    a.x3 = d->dtype == VT_F32X3 ? 1 : 0;
    a.alpha_dev = d->alpha_dev;
    a.post_relu = d->post_relu ? 1 : 0;
    a.wstream = d->weight_stream;
    a.rgb_only = (d->rgb_only && d->rgb_weight) ? 1 : 0;
    a.in_absdiff = d->in_absdiff ? 1 : 0;
    VT_REQUIRE(!a.in_absdiff || (d->src1 && d->c1 == d->c0), "vt_conv2d: in_absdiff needs src1 with c1 == c0");
    a.tile_stats = (float*)d->tile_stats;
    a.in_tile_stats = (const float*)d->in_tile_stats;
    a.in_stats_dil = d->in_stats_dil > 0 ? d->in_stats_dil : 1;
    a.in_gb = d->in_gb;
    a.in_ld_gb = d->in_ld_gb;
    a.up_fir = d->up_fir;
    a.in_scale = d->in_scale;
    a.in_shift = d->in_shift;
    a.resid = d->resid;
    a.out = d->out;
    a.c0 = d->c0;
    a.c1 = d->c1;
    a.ld0 = d->ld0;
    a.ld1 = d->ld1;
    a.cin = d->c0 + d->c1;
    a.N = d->n;
    a.H = d->h;
    a.W = d->w;
    a.Ho = d->out_h;
    a.Wo = d->out_w;
    a.cout = d->cout;
    a.coutT = d->cout * d->phases;
    a.taps = d->kh * d->kw;
    a.kw = d->kw;
    a.K = a.taps * a.cin;
    a.stride = d->stride;
    a.pad = d->pad;
    a.pad_x = d->pad_w_p1 > 0 ? d->pad_w_p1 - 1 : d->pad;
    a.dil = d->dil;
    a.transposed = d->transposed;
    a.phases = d->phases;
    a.act = d->act;
    a.slope = d->slope;
    a.gain_alpha = d->gain * d->alpha;
    a.beta = d->beta;
    a.ld_res = d->ld_res;
    a.ld_out = d->ld_out;
    a.out_layout = d->out_layout;
    a.out_f32 = d->out_dtype == VT_F32;
    a.M = d->n * d->out_h * d->out_w;
    const int osz = a.out_f32 ? 4 : 2;
    // workspace layout: [0, VT_TICKET_BYTES) per-tile arrival counters (must be zero when a launch
    // starts; every launch leaves them zero), fp32 slabs after
    a.phase = d->splitk_phase;
    a.tickets = d->splitk_ws ? (int*)d->splitk_ws : nullptr;
    a.partial = d->splitk_ws ? (float*)((char*)d->splitk_ws + VT_TICKET_BYTES) : nullptr;
    a.vec_store = (d->out_layout == VT_OUT_NHWC) && ((uintptr_t)d->out % 16 == 0) &&
                  ((int64_t)d->ld_out * osz % 16 == 0) && (d->cout % 8 == 0) &&
                  (!d->resid || (((uintptr_t)d->resid % 16 == 0) && ((int64_t)d->ld_res * osz % 16 == 0)));
    return VT_OK;
}

extern "C" int vt_conv2d(const vt_conv_desc* d, vt_stream stream) {
    ConvArgs a;
    const int rc = fill_args(d, a);
    if (rc != VT_OK) return rc;
    const int64_t wsf = (d->splitk_ws && d->splitk_ws_bytes > VT_TICKET_BYTES) ? (d->splitk_ws_bytes - VT_TICKET_BYTES) / 4 : 0;
    g_stats_emitted = false;
    a.x3 = d->dtype == VT_F32X3;
    const int rcl = d->dtype == VT_BF16 ? dispatch<bf16_t>(a, d->tile_hint, wsf, stream)
                                        : dispatch<float>(a, d->tile_hint, wsf, stream);
    if (rcl != VT_OK || !a.stats_part || g_stats_emitted || d->splitk_phase == 1) return rcl;
    // the plan had no reduce pass to carry the statistics: append the stand-alone launch
    return vt_internal_instnorm_partial(a.stats_part, d->out, d->ld_out, d->n, d->out_h * d->out_w, d->cout,
                                        d->dtype == VT_F32X3 ? VT_F32 : d->dtype, stream);
}

extern "C" int vt_conv2d_tile(const vt_conv_desc* d) {
    ConvArgs a;
    if (fill_args(d, a) != VT_OK) return -1;
    a.force_generic = d->tile_hint >= 1000000000 || a.pad_x != a.pad;
    const int64_t wsf = (d->splitk_ws && d->splitk_ws_bytes > VT_TICKET_BYTES) ? (d->splitk_ws_bytes - VT_TICKET_BYTES) / 4 : 0;
    const TilePlan t = d->dtype == VT_BF16 ? choose_plan<bf16_t>(a, d->tile_hint % 1000000000, wsf)
                                           : choose_plan<float>(a, d->tile_hint % 1000000000, wsf);
    int kind = t.kind;
    if (kind == 0) {  // report which 1-D loader the launch will use: 2 = direct-to-LDS, 0 = register-staged
        GldsArgs g;
        const bool glds = d->dtype == VT_BF16 ? glds_eligible<bf16_t>(a, g) : glds_eligible<float>(a, g);
        kind = glds ? 2 : 0;
    }
    int bm = t.bm, bn = t.bn;
    if (kind == 5 && d->dtype == VT_BF16) {   // 9 = the strip-marching form of the top up-sampling convs (conv_upblur_rows.hpp)
        UpblurArgs ub;
        if (uprows_wanted<bf16_t>(a) && upblur_eligible<bf16_t>(a, ub, 2, UR_OW)) kind = 9, bm = UR_OW, bn = 32;
        else if (upflat_wanted<bf16_t>(a) && upblur_eligible<bf16_t>(a, ub, 16, 64)) kind = 10, bm = 10 * UF_PW, bn = upflat_cn(a);   // conv_upblur_flat.hpp
    }
    return kind * 100000000 + t.splitk * 1000000 + bm * 1000 + bn;
}

extern "C" int vt_conv2d_splitk_mode(const vt_conv_desc* d) {
    ConvArgs a;
    if (fill_args(d, a) != VT_OK) return -1;
    a.force_generic = d->tile_hint >= 1000000000 || a.pad_x != a.pad;
    a.phase = d->splitk_phase;
    const int64_t wsf = (d->splitk_ws && d->splitk_ws_bytes > VT_TICKET_BYTES) ? (d->splitk_ws_bytes - VT_TICKET_BYTES) / 4 : 0;
    const TilePlan t = d->dtype == VT_BF16 ? choose_plan<bf16_t>(a, d->tile_hint % 1000000000, wsf)
                                           : choose_plan<float>(a, d->tile_hint % 1000000000, wsf);
    if (t.splitk <= 1) return 0;
    // the tile grid of the launch (launch_cfg / launch_patch): 1-D tiles of bm pixels, or bm/16 x 16-pixel tiles
    const int64_t tiles_m = t.kind == 1 ? (int64_t)a.N * vt_cdiv(a.Ho, t.bm / 16) * vt_cdiv(a.Wo, 16) : vt_cdiv(a.M, t.bm);
    return in_launch_rule(a, tiles_m * vt_cdiv(a.coutT, t.bn)) ? 1 : 2;
}

extern "C" int64_t vt_conv2d_ws_bytes(const vt_conv_desc* d) {
    ConvArgs a;
    if (fill_args(d, a) != VT_OK) return -1;
    a.force_generic = d->tile_hint >= 1000000000 || a.pad_x != a.pad;
    float dummy;
    a.partial = &dummy;  // "a workspace of any size exists": report what the heuristic would use
    const int64_t big = (int64_t)1 << 40;
    const TilePlan t = d->dtype == VT_BF16 ? choose_plan<bf16_t>(a, d->tile_hint % 1000000000, big)
                                           : choose_plan<float>(a, d->tile_hint % 1000000000, big);
    if (t.splitk <= 1) return 0;
    return VT_TICKET_BYTES + (int64_t)t.splitk * a.M * slab_ld(a.coutT) * 4;
}

extern "C" int64_t vt_conv_weight_stream_bytes(int cout, int taps, int cin, int dtype) {
    if (cout <= 0 || taps <= 0 || cin <= 0 || (dtype != VT_F32 && dtype != VT_BF16)) return -1;
    const int esz = dtype == VT_F32 ? 4 : 2;
    const int bk = 128 / esz;
    if (cin % bk != 0) return -1;
    return (int64_t)vt_cdiv(cout, FK_BN) * (cin / bk) * taps * 4 * 1024;
}

extern "C" int64_t vt_conv_tile_stats_bytes(int n, int h, int w, int dil, int c) {
    if (n <= 0 || h <= 0 || w <= 0 || dil <= 0 || c <= 0) return -1;
    const int64_t tiles = (int64_t)n * dil * dil * vt_cdiv(vt_cdiv(h, dil), FK_TH) * vt_cdiv(vt_cdiv(w, dil), FK_TW);
    return tiles * c * 8 + tiles * 4;   // {mean, M2} per (tile, channel), then the pixel count of every tile
}

extern "C" int vt_conv_weight_stream(void* out, const void* packed, int cout, int taps, int cin, int dtype,
                                     vt_stream stream) {
    VT_REQUIRE(out && packed, "vt_conv_weight_stream: null tensor");
    const int64_t bytes = vt_conv_weight_stream_bytes(cout, taps, cin, dtype);
    VT_REQUIRE(bytes > 0, "vt_conv_weight_stream: unsupported shape/dtype (cout %d, taps %d, cin %d, dtype %d)", cout, taps,
               cin, dtype);
    VT_REQUIRE(((uintptr_t)out % 16 == 0) && ((uintptr_t)packed % 16 == 0), "vt_conv_weight_stream: 16-byte alignment");
    const int64_t total16 = bytes / 16;
    int64_t blocks = (total16 + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    if (dtype == VT_BF16) {
        auto k = weight_stream_kernel<bf16_t>;
        VT_LAUNCH(k, dim3((unsigned)blocks), dim3(256), stream, (bf16_t*)out, (const bf16_t*)packed, cout, taps, cin, total16);
    } else {
        auto k = weight_stream_kernel<float>;
        VT_LAUNCH(k, dim3((unsigned)blocks), dim3(256), stream, (float*)out, (const float*)packed, cout, taps, cin, total16);
    }
    return vt_check_launch("vt_conv_weight_stream");
}

// ---------------------------------------------------------------------------------
// MFMA lane-map self test: one wavefront computes C(16x16) = A(16xKK) * B(16xKK)^T with
// the same fragment addressing as the conv kernel (KK = 32 bf16 / 16 fp32 = one sub-step).
// ---------------------------------------------------------------------------------
template <typename T>
__global__ void __launch_bounds__(64) mfma_selftest_kernel(float* c, const T* a, const T* b) {
    constexpr int VEC = 16 / sizeof(T);
    const int lane = threadIdx.x & 63, q = lane >> 4, l15 = lane & 15;
    const u128 fa = ld128(a + l15 * (4 * VEC) + q * VEC);
    const u128 fb = ld128(b + l15 * (4 * VEC) + q * VEC);
    f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
    Mma<T>::run(acc, fa, fb);
    for (int r = 0; r < 4; ++r) c[(q * 4 + r) * 16 + l15] = acc[r];
}

extern "C" int vt_mfma_selftest(float* c, const void* a, const void* b, int dtype, vt_stream stream) {
    VT_REQUIRE(c && a && b, "vt_mfma_selftest: null tensor");
    if (dtype == VT_BF16) {
        auto k = mfma_selftest_kernel<bf16_t>;
        VT_LAUNCH(k, dim3(1), dim3(64), stream, c, (const bf16_t*)a, (const bf16_t*)b);
    } else if (dtype == VT_F32) {
        auto k = mfma_selftest_kernel<float>;
        VT_LAUNCH(k, dim3(1), dim3(64), stream, c, (const float*)a, (const float*)b);
    } else {
        vt_set_error("vt_mfma_selftest: dtype");
        return VT_ERR_UNSUPPORTED;
    }
    return vt_check_launch("vt_mfma_selftest");
}
