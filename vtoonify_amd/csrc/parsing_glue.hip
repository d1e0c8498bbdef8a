// Glue kernels of the face-parsing network (BiSeNet, model/bisenet/model.py:92-254 and
// model/bisenet/resnet.py:58-80) around the MFMA convolutions, all HBM-streaming (gfx950):
//   vt_maxpool2d          nn.MaxPool2d(3, 2, 1) of the ResNet18 stem (resnet.py:63,71), NHWC
//   vt_gate_add_nearest   AttentionRefinementModule's feat * atten (model.py:78-85) fused with the
//                         "+ avg_up" / "+ feat32_up" add and the nearest F.interpolate that follows
//                         it in ContextPath.forward (model.py:108-121)
//   vt_resize_bilinear    F.interpolate(mode='bilinear') on planar fp32 images, either corner
//                         convention: the x2 up-sampling of the frame in front of the parsing net
//                         (style_transfer.py:171, align_corners=False) and the up-sampling of the
//                         class maps to the input size (model.py:252-254, align_corners=True); an
//                         output step > 1 evaluates only every step-th pixel, which is what the
//                         nearest F.interpolate(scale_factor=0.5) of style_transfer.py:171-172 keeps.
#include "vt_common.hpp"

namespace {

inline unsigned pg_grid(int64_t total) {
    int64_t b = (total + 255) / 256;
    if (b < 1) b = 1;
    if (b > 262144) b = 262144;
    return (unsigned)b;
}

template <typename T>
__global__ void __launch_bounds__(256)
maxpool_kernel(T* __restrict__ out, const T* __restrict__ x, int n, int h, int w, int c, int oh, int ow,
               int k, int stride, int pad) {
    constexpr int VEC = 16 / sizeof(T);
    const int cvn = c / VEC;
    const int64_t total = (int64_t)n * oh * ow * cvn;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int cv = (int)(i % cvn);
        const int64_t pix = i / cvn;
        const int ox = (int)(pix % ow);
        const int64_t t = pix / ow;
        const int oy = (int)(t % oh), img = (int)(t / oh);
        float m[VEC];
#pragma unroll
        for (int e = 0; e < VEC; ++e) m[e] = -INFINITY;
        for (int ky = 0; ky < k; ++ky) {
            const int iy = oy * stride - pad + ky;
            if (iy < 0 || iy >= h) continue;
            for (int kx = 0; kx < k; ++kx) {
                const int ix = ox * stride - pad + kx;
                if (ix < 0 || ix >= w) continue;
                float f[VEC];
                unpack16<T>(ld128(x + (((int64_t)img * h + iy) * w + ix) * c + cv * VEC), f);
#pragma unroll
                for (int e = 0; e < VEC; ++e) m[e] = fmaxf(m[e], f[e]);
            }
        }
        st128(out + pix * c + cv * VEC, pack16<T>(m));
    }
}

// F.interpolate(mode='nearest') source index (aten upsample_nearest: floor(dst * in/out), clamped)
__device__ __forceinline__ int nearest_src(int dst, float scale, int in_size) {
    const int s = (int)floorf((float)dst * scale);
    return s < in_size - 1 ? s : in_size - 1;
}

template <typename T>
__global__ void __launch_bounds__(256)
gate_add_nearest_kernel(T* __restrict__ out, const T* __restrict__ res, const float* __restrict__ gate,
                        const float* __restrict__ add_vec, const T* __restrict__ add, int n, int h, int w,
                        int c, int oh, int ow) {
    constexpr int VEC = 16 / sizeof(T);
    const int cvn = c / VEC;
    const int64_t total = (int64_t)n * oh * ow * cvn;
    const float sy = (float)h / (float)oh, sx = (float)w / (float)ow;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int cv = (int)(i % cvn);
        const int64_t pix = i / cvn;
        const int X = (int)(pix % ow);
        const int64_t t = pix / ow;
        const int Y = (int)(t % oh), img = (int)(t / oh);
        const int y = (oh == h) ? Y : nearest_src(Y, sy, h);
        const int x = (ow == w) ? X : nearest_src(X, sx, w);
        const int64_t spix = ((int64_t)img * h + y) * w + x;
        float f[VEC];
        unpack16<T>(ld128(res + spix * c + cv * VEC), f);
        const float* gt = gate + (int64_t)img * c + cv * VEC;
#pragma unroll
        for (int e = 0; e < VEC; ++e) f[e] *= gt[e];
        if (add_vec) {
            const float* av = add_vec + (int64_t)img * c + cv * VEC;
#pragma unroll
            for (int e = 0; e < VEC; ++e) f[e] += av[e];
        }
        if (add) {
            float g[VEC];
            unpack16<T>(ld128(add + spix * c + cv * VEC), g);
#pragma unroll
            for (int e = 0; e < VEC; ++e) f[e] += g[e];
        }
        st128(out + pix * c + cv * VEC, pack16<T>(f));
    }
}

// aten upsample_bilinear2d: source coordinate, the two taps and their weights along one axis
__device__ __forceinline__ void bilinear_tap(int dst, float scale, int align, int in_size, int& i0, int& i1,
                                             float& l0, float& l1) {
    float src;
    if (align) {
        src = scale * (float)dst;
    } else {
        src = scale * ((float)dst + 0.5f) - 0.5f;
        if (src < 0.0f) src = 0.0f;
    }
    i0 = (int)src;
    if (i0 > in_size - 1) i0 = in_size - 1;
    i1 = i0 + (i0 < in_size - 1 ? 1 : 0);
    l1 = src - (float)i0;
    l0 = 1.0f - l1;
}

template <typename T>
__global__ void __launch_bounds__(256)
resize_bilinear_kernel(T* __restrict__ out, int nhwc, int ld_out, const float* __restrict__ in, int n, int c,
                       int h, int w, float sy, float sx, int align, int step, int oh, int ow, float mul) {
    const int64_t total = (int64_t)n * c * oh * ow;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        // planar output: lanes walk X; NHWC output: lanes walk channels of one pixel
        int X, Y, ch, img;
        if (nhwc) {
            ch = (int)(i % c);
            int64_t t = i / c;
            X = (int)(t % ow); t /= ow;
            Y = (int)(t % oh); img = (int)(t / oh);
        } else {
            X = (int)(i % ow);
            int64_t t = i / ow;
            Y = (int)(t % oh); t /= oh;
            ch = (int)(t % c); img = (int)(t / c);
        }
        int y0, y1, x0, x1;
        float ly0, ly1, lx0, lx1;
        bilinear_tap(Y * step, sy, align, h, y0, y1, ly0, ly1);
        bilinear_tap(X * step, sx, align, w, x0, x1, lx0, lx1);
        const float* p = in + ((int64_t)img * c + ch) * h * w;
        const float v = ly0 * (lx0 * p[(int64_t)y0 * w + x0] + lx1 * p[(int64_t)y0 * w + x1]) +
                        ly1 * (lx0 * p[(int64_t)y1 * w + x0] + lx1 * p[(int64_t)y1 * w + x1]);
        const int64_t o = nhwc ? (((int64_t)img * oh + Y) * ow + X) * ld_out + ch
                               : (((int64_t)img * c + ch) * oh + Y) * ow + X;
        out[o] = from_f32<T>(v * mul);
    }
}

}  // namespace

extern "C" int vt_maxpool2d(void* out, const void* x, int n, int h, int w, int c, int k, int stride, int pad,
                            int dtype, vt_stream stream) {
    VT_REQUIRE(out && x, "vt_maxpool2d: null tensor");
    VT_REQUIRE(n > 0 && h > 0 && w > 0 && c > 0 && c % 8 == 0 && k > 0 && stride > 0 && pad >= 0 && 2 * pad <= k,
               "vt_maxpool2d: bad sizes (c must be a multiple of 8, pad <= k/2)");
    const int oh = (h + 2 * pad - k) / stride + 1, ow = (w + 2 * pad - k) / stride + 1;
    VT_REQUIRE(oh > 0 && ow > 0, "vt_maxpool2d: empty output");
    if (dtype == VT_F32) {
        auto kfn = maxpool_kernel<float>;
        VT_LAUNCH(kfn, dim3(pg_grid((int64_t)n * oh * ow * (c / 4))), dim3(256), stream, (float*)out, (const float*)x,
                  n, h, w, c, oh, ow, k, stride, pad);
    } else if (dtype == VT_BF16) {
        auto kfn = maxpool_kernel<bf16_t>;
        VT_LAUNCH(kfn, dim3(pg_grid((int64_t)n * oh * ow * (c / 8))), dim3(256), stream, (bf16_t*)out,
                  (const bf16_t*)x, n, h, w, c, oh, ow, k, stride, pad);
    } else {
        vt_set_error("vt_maxpool2d: dtype");
        return VT_ERR_UNSUPPORTED;
    }
    return vt_check_launch("vt_maxpool2d");
}

extern "C" int vt_gate_add_nearest(void* out, const void* res, const float* gate, const float* add_vec,
                                   const void* add, int n, int h, int w, int c, int out_h, int out_w, int dtype,
                                   vt_stream stream) {
    VT_REQUIRE(out && res && gate, "vt_gate_add_nearest: null tensor");
    VT_REQUIRE(n > 0 && h > 0 && w > 0 && out_h > 0 && out_w > 0 && c > 0 && c % 8 == 0,
               "vt_gate_add_nearest: bad sizes (c must be a multiple of 8)");
    if (dtype == VT_F32) {
        auto kfn = gate_add_nearest_kernel<float>;
        VT_LAUNCH(kfn, dim3(pg_grid((int64_t)n * out_h * out_w * (c / 4))), dim3(256), stream, (float*)out,
                  (const float*)res, gate, add_vec, (const float*)add, n, h, w, c, out_h, out_w);
    } else if (dtype == VT_BF16) {
        auto kfn = gate_add_nearest_kernel<bf16_t>;
        VT_LAUNCH(kfn, dim3(pg_grid((int64_t)n * out_h * out_w * (c / 8))), dim3(256), stream, (bf16_t*)out,
                  (const bf16_t*)res, gate, add_vec, (const bf16_t*)add, n, h, w, c, out_h, out_w);
    } else {
        vt_set_error("vt_gate_add_nearest: dtype");
        return VT_ERR_UNSUPPORTED;
    }
    return vt_check_launch("vt_gate_add_nearest");
}

extern "C" int vt_resize_bilinear(void* out, int out_layout, int ld_out, int out_dtype, const float* in, int n,
                                  int c, int h, int w, int virt_h, int virt_w, int align_corners, int step,
                                  int out_h, int out_w, float mul, vt_stream stream) {
    VT_REQUIRE(out && in, "vt_resize_bilinear: null tensor");
    VT_REQUIRE(n > 0 && c > 0 && h > 0 && w > 0 && virt_h > 0 && virt_w > 0 && step >= 1 && out_h > 0 && out_w > 0,
               "vt_resize_bilinear: bad sizes");
    VT_REQUIRE((out_h - 1) * step < virt_h && (out_w - 1) * step < virt_w,
               "vt_resize_bilinear: output grid exceeds the virtual image");
    VT_REQUIRE(out_layout == VT_OUT_NCHW || (out_layout == VT_OUT_NHWC && ld_out >= c),
               "vt_resize_bilinear: NHWC output needs ld_out >= c");
    // aten area_pixel_compute_scale: align_corners ? (in-1)/(out-1) : in/out
    float sy, sx;
    if (align_corners) {
        sy = virt_h > 1 ? (float)(h - 1) / (float)(virt_h - 1) : 0.0f;
        sx = virt_w > 1 ? (float)(w - 1) / (float)(virt_w - 1) : 0.0f;
    } else {
        sy = (float)h / (float)virt_h;
        sx = (float)w / (float)virt_w;
    }
    const int nhwc = out_layout == VT_OUT_NHWC;
    const int64_t total = (int64_t)n * c * out_h * out_w;
    if (out_dtype == VT_F32) {
        auto kfn = resize_bilinear_kernel<float>;
        VT_LAUNCH(kfn, dim3(pg_grid(total)), dim3(256), stream, (float*)out, nhwc, ld_out, in, n, c, h, w, sy, sx,
                  align_corners ? 1 : 0, step, out_h, out_w, mul);
    } else if (out_dtype == VT_BF16) {
        auto kfn = resize_bilinear_kernel<bf16_t>;
        VT_LAUNCH(kfn, dim3(pg_grid(total)), dim3(256), stream, (bf16_t*)out, nhwc, ld_out, in, n, c, h, w, sy, sx,
                  align_corners ? 1 : 0, step, out_h, out_w, mul);
    } else {
        vt_set_error("vt_resize_bilinear: out dtype");
        return VT_ERR_UNSUPPORTED;
    }
    return vt_check_launch("vt_resize_bilinear");
}
