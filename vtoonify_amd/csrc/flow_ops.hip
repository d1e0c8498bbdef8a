// Flow warp and the temporal fusion of parsing maps -- the loop body of the reference's flicker-reduction pre-pass
// (smooth_parsing_map.py:37-75 `warp`, :143-167 the window fusion).  fp32, NCHW planes like the
// reference's tensors.
//
//   vt_flow_warp     out[b,c,y,x] = mask[b,y,x] * bilinear_zero(x[b,c], (x + flo[b,0,y,x], y + flo[b,1,y,x]))
//                    mask = 1 where the bilinear weights of the in-image corners sum to >= 0.9999, else 0
//                    (grid_sample(align_corners=True) of x and of ones: smooth_parsing_map.py:61-69)
//   vt_parsing_fuse  for one centre frame and a window of 2w+1 neighbours (smooth_parsing_map.py:155-166):
//                      aligned_I_j, aligned_P_j = warp(cat(image2_j, P_j), flow_j)
//                      ws_j = exp(-mean_c (aligned_I_j - image1)^2 / (2 sigma^2)) * mask_j ;  ws_centre = 1
//                      aligned_P_centre = P_centre
//                      fused = sum_j aligned_P_j * (ws_j wt_j) / sum_j (ws_j wt_j)
//                    one pass: every lane owns one pixel, walks the window, keeps the 19 class sums in registers;
//                    the warped frames / weights / normalised weights (5 full-size intermediates per window frame in
//                    the reference) never exist in memory.
// HBM-bound gathers: algorithmic bytes per output pixel = (2w+1) * (3 + CP + 2) * 4 read + CP * 4 written.
//
// Sample coordinates follow torch's grid_sampler arithmetic so that the 0.9999 mask threshold falls on the same
// pixels: v = 2 (x + f) / max(W-1, 1) - 1 (smooth_parsing_map.py:58-59), ix = (v + 1) / 2 * (W - 1)
// (ATen GridSampler.h grid_sampler_unnormalize, align_corners = true).
#include "vt_common.hpp"

namespace {

constexpr int FUSE_MAX_CP = 32;   // parsing classes held in registers (BiSeNet: 19)

struct Bilin {
    int x0, y0;          // top-left corner
    float w00, w01, w10, w11;   // weights of (y0,x0) (y0,x0+1) (y0+1,x0) (y0+1,x0+1); 0 where the corner is outside
    float msum;          // sum of the in-image weights (grid_sample of a plane of ones)
};

__device__ __forceinline__ Bilin bilin_setup(int x, int y, float fx, float fy, int H, int W) {
    const float wm = (float)(W - 1 > 1 ? W - 1 : 1), hm = (float)(H - 1 > 1 ? H - 1 : 1);
    const float vx = 2.0f * ((float)x + fx) / wm - 1.0f;
    const float vy = 2.0f * ((float)y + fy) / hm - 1.0f;
    const float ix = ((vx + 1.0f) / 2.0f) * (float)(W - 1);
    const float iy = ((vy + 1.0f) / 2.0f) * (float)(H - 1);
    const float fx0 = floorf(ix), fy0 = floorf(iy);
    Bilin b;
    // far-away targets: clamp the integer corner (all four corners are outside either way)
    b.x0 = (int)fminf(fmaxf(fx0, -2.0f), (float)W + 1.0f);
    b.y0 = (int)fminf(fmaxf(fy0, -2.0f), (float)H + 1.0f);
    const float tx = ix - fx0, ty = iy - fy0;          // weights as ATen: nw = (ix_se - ix)(iy_se - iy), ...
    const float ax = (fx0 + 1.0f) - ix, ay = (fy0 + 1.0f) - iy;
    const bool xin0 = (unsigned)b.x0 < (unsigned)W, xin1 = (unsigned)(b.x0 + 1) < (unsigned)W;
    const bool yin0 = (unsigned)b.y0 < (unsigned)H, yin1 = (unsigned)(b.y0 + 1) < (unsigned)H;
    b.w00 = (xin0 && yin0) ? ax * ay : 0.0f;
    b.w01 = (xin1 && yin0) ? tx * ay : 0.0f;
    b.w10 = (xin0 && yin1) ? ax * ty : 0.0f;
    b.w11 = (xin1 && yin1) ? tx * ty : 0.0f;
    b.msum = ((b.w00 + b.w01) + b.w10) + b.w11;        // ATen accumulation order nw, ne, sw, se
    return b;
}

__device__ __forceinline__ float bilin_sample(const float* plane, const Bilin& b, int W) {
    // corners with zero weight are never dereferenced
    float v = 0.0f;
    const float* p = plane + (int64_t)b.y0 * W + b.x0;
    if (b.w00 != 0.0f) v += p[0] * b.w00;
    if (b.w01 != 0.0f) v += p[1] * b.w01;
    if (b.w10 != 0.0f) v += p[W] * b.w10;
    if (b.w11 != 0.0f) v += p[W + 1] * b.w11;
    return v;
}

__global__ void __launch_bounds__(256)
flow_warp_kernel(float* __restrict__ out, float* __restrict__ mask, const float* __restrict__ x,
                 const float* __restrict__ flo, int B, int C, int H, int W) {
    const int64_t hw = (int64_t)H * W;
    const int64_t total = (int64_t)B * hw;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int b = (int)(i / hw);
        const int64_t rem = i - (int64_t)b * hw;
        const int py = (int)(rem / W), px = (int)(rem - (int64_t)py * W);
        const Bilin bl = bilin_setup(px, py, flo[((int64_t)b * 2 + 0) * hw + rem], flo[((int64_t)b * 2 + 1) * hw + rem], H, W);
        const float m = bl.msum < 0.9999f ? 0.0f : 1.0f;      // mask[mask < 0.9999] = 0; mask[mask > 0] = 1
        if (mask) mask[i] = m;
        for (int c = 0; c < C; ++c) {
            const int64_t o = ((int64_t)b * C + c) * hw;
            out[o + rem] = m != 0.0f ? bilin_sample(x + o, bl, W) : 0.0f;
        }
    }
}

__global__ void __launch_bounds__(256)
parsing_fuse_kernel(float* __restrict__ fused, const float* __restrict__ frames, const float* __restrict__ center,
                    const float* __restrict__ parsing, const float* __restrict__ flow, const float* __restrict__ wt,
                    int wn, int ci, int cp, int H, int W, float inv2s2) {
    const int64_t hw = (int64_t)H * W;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < hw; i += (int64_t)gridDim.x * 256) {
        const int py = (int)(i / W), px = (int)(i - (int64_t)py * W);
        const float c0 = center[i], c1 = center[hw + i], c2 = center[2 * hw + i];
        float acc[FUSE_MAX_CP];
#pragma unroll
        for (int c = 0; c < FUSE_MAX_CP; ++c) acc[c] = 0.0f;
        float wsum = 0.0f;
        for (int j = 0; j < wn; ++j) {      // window order (the reference's sum over dim 0)
            const float* pj = parsing + (int64_t)j * cp * hw;
            if (j == ci) {                  // aligned_Ps[window] = Ps_[i]; ws[window] = 1
                const float wj = wt[j];
                wsum += wj;
#pragma unroll
                for (int c = 0; c < FUSE_MAX_CP; ++c)
                    if (c < cp) acc[c] += pj[(int64_t)c * hw + i] * wj;
                continue;
            }
            const Bilin bl = bilin_setup(px, py, flow[((int64_t)j * 2 + 0) * hw + i], flow[((int64_t)j * 2 + 1) * hw + i], H, W);
            if (bl.msum < 0.9999f) continue;   // mask = 0: weight 0, contributes nothing
            const float* fj = frames + (int64_t)j * 3 * hw;
            const float d0 = bilin_sample(fj, bl, W) - c0, d1 = bilin_sample(fj + hw, bl, W) - c1,
                        d2 = bilin_sample(fj + 2 * hw, bl, W) - c2;
            const float mse = ((d0 * d0 + d1 * d1) + d2 * d2) / 3.0f;
            const float wj = expf(-mse * inv2s2) * wt[j];
            wsum += wj;
#pragma unroll
            for (int c = 0; c < FUSE_MAX_CP; ++c)
                if (c < cp) acc[c] += bilin_sample(pj + (int64_t)c * hw, bl, W) * wj;
        }
        const float inv = 1.0f / wsum;       // the centre frame always contributes wt[ci] > 0
#pragma unroll
        for (int c = 0; c < FUSE_MAX_CP; ++c)
            if (c < cp) fused[(int64_t)c * hw + i] = acc[c] * inv;
    }
}

// ---- RAFT glue (model/raft/core/update.py:44-56 SepConvGRU gates, raft.py:72-84 convex up-sampling, :58-66 coords) ----
// out[r][c] = a op b on rows of `c` channels with independent row strides: 0 mul (r * h), 1 add, 2 relu(add)
template <typename T>
__global__ void __launch_bounds__(256)
eltwise2_kernel(T* __restrict__ out, int ld_out, const T* __restrict__ a, int ld_a, const T* __restrict__ b, int ld_b,
                int64_t rows, int c, int op) {
    const int64_t total = rows * c;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t r = i / c;
        const int ch = (int)(i - r * c);
        const float x = to_f32(a[r * ld_a + ch]), y = to_f32(b[r * ld_b + ch]);
        float v = op == 0 ? x * y : x + y;
        if (op == 2) v = fmaxf(v, 0.0f);
        out[r * ld_out + ch] = from_f32<T>(v);
    }
}

// h = (1 - z) * h + z * q in place (update.py:49,55)
template <typename T>
__global__ void __launch_bounds__(256)
gru_blend_kernel(T* __restrict__ h, int ld_h, const T* __restrict__ z, const T* __restrict__ q, int64_t rows, int c) {
    const int64_t total = rows * c;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t r = i / c;
        const int ch = (int)(i - r * c);
        const float zz = to_f32(z[i]), hh = to_f32(h[r * ld_h + ch]), qq = to_f32(q[i]);
        h[r * ld_h + ch] = from_f32<T>((1.0f - zz) * hh + zz * qq);
    }
}

// coords[b, 0, y, x] = (x + flow[b,0,y,x], y + flow[b,1,y,x]): coords1 of raft.py:58-66,121 as the lookup wants them
__global__ void __launch_bounds__(256)
coords_from_flow_kernel(float* __restrict__ coords, const float* __restrict__ flow, int n, int h, int w) {
    const int64_t hw = (int64_t)h * w, total = (int64_t)n * hw;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int b = (int)(i / hw);
        const int64_t rem = i - (int64_t)b * hw;
        const int y = (int)(rem / w), x = (int)(rem - (int64_t)y * w);
        coords[i * 2 + 0] = (float)x + flow[((int64_t)b * 2 + 0) * hw + rem];
        coords[i * 2 + 1] = (float)y + flow[((int64_t)b * 2 + 1) * hw + rem];
    }
}

// RAFT.upsample_flow (raft.py:72-84): softmax over the 9 neighbours of mask (n, 9*64, h, w), convex combination of
// the 3x3 neighbourhood of 8 * flow (zero outside), pixel shuffle to (n, 2, 8h, 8w).  One lane per fine pixel.
__global__ void __launch_bounds__(256)
convex_upsample_kernel(float* __restrict__ out, const float* __restrict__ flow, const float* __restrict__ mask,
                       int n, int h, int w) {
    const int64_t hw = (int64_t)h * w;
    const int H8 = 8 * h, W8 = 8 * w;
    const int64_t total = (int64_t)n * H8 * W8;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int b = (int)(i / ((int64_t)H8 * W8));
        const int64_t rem = i - (int64_t)b * H8 * W8;
        const int Y = (int)(rem / W8), X = (int)(rem - (int64_t)Y * W8);
        const int y = Y >> 3, ii = Y & 7, x = X >> 3, jj = X & 7;
        const float* mp = mask + ((int64_t)b * 576 + ii * 8 + jj) * hw + (int64_t)y * w + x;
        float m[9], mx = -INFINITY;
#pragma unroll
        for (int k = 0; k < 9; ++k) {
            m[k] = mp[(int64_t)k * 64 * hw];
            mx = fmaxf(mx, m[k]);
        }
        float den = 0.0f;
#pragma unroll
        for (int k = 0; k < 9; ++k) {
            m[k] = expf(m[k] - mx);
            den += m[k];
        }
        float u0 = 0.0f, u1 = 0.0f;
#pragma unroll
        for (int k = 0; k < 9; ++k) {
            const int yy = y + k / 3 - 1, xx = x + k % 3 - 1;
            if ((unsigned)yy < (unsigned)h && (unsigned)xx < (unsigned)w) {
                const float wk = m[k] / den;
                const int64_t o = (int64_t)yy * w + xx;
                u0 += wk * (8.0f * flow[((int64_t)b * 2 + 0) * hw + o]);
                u1 += wk * (8.0f * flow[((int64_t)b * 2 + 1) * hw + o]);
            }
        }
        out[((int64_t)b * 2 + 0) * H8 * W8 + rem] = u0;
        out[((int64_t)b * 2 + 1) * H8 * W8 + rem] = u1;
    }
}

}  // namespace

extern "C" int vt_flow_warp(float* out, float* mask, const float* x, const float* flo, int n, int c, int h, int w,
                            vt_stream stream) {
    VT_REQUIRE(out && x && flo, "vt_flow_warp: null tensor");
    VT_REQUIRE(n > 0 && c > 0 && h > 0 && w > 0, "vt_flow_warp: bad sizes");
    VT_REQUIRE((int64_t)n * c * h * w < ((int64_t)1 << 40), "vt_flow_warp: tensor too large");
    int64_t blocks = ((int64_t)n * h * w + 255) / 256;
    if (blocks > 65536) blocks = 65536;
    VT_LAUNCH(flow_warp_kernel, dim3((unsigned)blocks), dim3(256), stream, out, mask, x, flo, n, c, h, w);
    return vt_check_launch("vt_flow_warp");
}

extern "C" int vt_parsing_fuse(float* fused, const float* frames, const float* center, const float* parsing,
                               const float* flow, const float* wt, int wn, int center_index, int cp, int h, int w,
                               float sigma, vt_stream stream) {
    VT_REQUIRE(fused && frames && center && parsing && flow && wt, "vt_parsing_fuse: null tensor");
    VT_REQUIRE(wn > 0 && center_index >= 0 && center_index < wn, "vt_parsing_fuse: the centre frame must be inside the window");
    VT_REQUIRE(cp > 0 && cp <= FUSE_MAX_CP, "vt_parsing_fuse: 1..%d parsing classes", FUSE_MAX_CP);
    VT_REQUIRE(h > 0 && w > 0 && sigma > 0.0f, "vt_parsing_fuse: bad sizes");
    int64_t blocks = ((int64_t)h * w + 255) / 256;
    if (blocks > 65536) blocks = 65536;
    VT_LAUNCH(parsing_fuse_kernel, dim3((unsigned)blocks), dim3(256), stream, fused, frames, center, parsing, flow, wt,
              wn, center_index, cp, h, w, 1.0f / (2.0f * sigma * sigma));
    return vt_check_launch("vt_parsing_fuse");
}

extern "C" int vt_eltwise2(void* out, int ld_out, const void* a, int ld_a, const void* b, int ld_b, int64_t rows,
                           int c, int op, int dtype, vt_stream stream) {
    VT_REQUIRE(out && a && b, "vt_eltwise2: null tensor");
    VT_REQUIRE(rows > 0 && c > 0 && ld_out >= c && ld_a >= c && ld_b >= c && op >= 0 && op <= 2, "vt_eltwise2: bad arguments");
    int64_t blocks = (rows * c + 255) / 256;
    if (blocks > 65536) blocks = 65536;
    if (dtype == VT_F32) {
        auto k = eltwise2_kernel<float>;
        VT_LAUNCH(k, dim3((unsigned)blocks), dim3(256), stream, (float*)out, ld_out, (const float*)a, ld_a, (const float*)b, ld_b, rows, c, op);
    } else if (dtype == VT_BF16) {
        auto k = eltwise2_kernel<bf16_t>;
        VT_LAUNCH(k, dim3((unsigned)blocks), dim3(256), stream, (bf16_t*)out, ld_out, (const bf16_t*)a, ld_a, (const bf16_t*)b, ld_b, rows, c, op);
    } else {
        vt_set_error("vt_eltwise2: dtype");
        return VT_ERR_UNSUPPORTED;
    }
    return vt_check_launch("vt_eltwise2");
}

extern "C" int vt_gru_blend(void* h, int ld_h, const void* z, const void* q, int64_t rows, int c, int dtype,
                            vt_stream stream) {
    VT_REQUIRE(h && z && q, "vt_gru_blend: null tensor");
    VT_REQUIRE(rows > 0 && c > 0 && ld_h >= c, "vt_gru_blend: bad sizes");
    int64_t blocks = (rows * c + 255) / 256;
    if (blocks > 65536) blocks = 65536;
    if (dtype == VT_F32) {
        auto k = gru_blend_kernel<float>;
        VT_LAUNCH(k, dim3((unsigned)blocks), dim3(256), stream, (float*)h, ld_h, (const float*)z, (const float*)q, rows, c);
    } else if (dtype == VT_BF16) {
        auto k = gru_blend_kernel<bf16_t>;
        VT_LAUNCH(k, dim3((unsigned)blocks), dim3(256), stream, (bf16_t*)h, ld_h, (const bf16_t*)z, (const bf16_t*)q, rows, c);
    } else {
        vt_set_error("vt_gru_blend: dtype");
        return VT_ERR_UNSUPPORTED;
    }
    return vt_check_launch("vt_gru_blend");
}

extern "C" int vt_coords_from_flow(float* coords, const float* flow, int n, int h, int w, vt_stream stream) {
    VT_REQUIRE(coords && flow && n > 0 && h > 0 && w > 0, "vt_coords_from_flow: bad arguments");
    int64_t blocks = ((int64_t)n * h * w + 255) / 256;
    if (blocks > 65536) blocks = 65536;
    VT_LAUNCH(coords_from_flow_kernel, dim3((unsigned)blocks), dim3(256), stream, coords, flow, n, h, w);
    return vt_check_launch("vt_coords_from_flow");
}

extern "C" int vt_convex_upsample(float* out, const float* flow, const float* mask, int n, int h, int w,
                                  vt_stream stream) {
    VT_REQUIRE(out && flow && mask && n > 0 && h > 0 && w > 0, "vt_convex_upsample: bad arguments");
    int64_t blocks = ((int64_t)n * h * w * 64 + 255) / 256;
    if (blocks > 65536) blocks = 65536;
    VT_LAUNCH(convex_upsample_kernel, dim3((unsigned)blocks), dim3(256), stream, out, flow, mask, n, h, w);
    return vt_check_launch("vt_convex_upsample");
}
