// fused bias + activation for gfx950 -- the operator behind FusedLeakyReLU /
// fused_leaky_relu (reference surface: model/stylegan/op/fused_act.py:87-119, native op
// model/stylegan/op/fused_bias_act_kernel.cu:18-105).
//
//   y[i] = act(x[i] + b[(i / step_b) % size_b]) * scale
//
// Pure HBM streaming: 2 * numel * sizeof(T) + size_b * sizeof(T) algorithmic bytes.
// Design for CDNA4:
//   * "plane" kernel (spatial tensors, step_b >= 64): a workgroup owns a slice of ONE
//     (n, c) plane, so the channel index and bias are wave-uniform (one scalar load per
//     workgroup instead of an integer divide + gather per element as in the reference
//     kernel), and each lane moves 16-byte vectors, 4 in flight.
//   * "flat" kernel for (N, C) inputs of EqualLinear (step_b == 1) and unaligned planes:
//     per-element index arithmetic, still coalesced.
// fp32 arithmetic is add, select-multiply, multiply in that order -- the same two/three
// roundings as the reference kernel, so fp32 results are bit-identical to op_cpu.
#include "vt_common.hpp"

namespace {

__device__ __forceinline__ float act_apply(float x, float ref, int use_ref, int mode, float alpha) {
    // mode = act*10+grad of fused_bias_act_kernel.cu:40-61
    float y;
    switch (mode) {
        case 12:
        case 32:
            y = 0.0f;
            break;
        case 30:
            y = (x > 0.0f) ? x : x * alpha;
            break;
        case 31:
            y = ((use_ref ? ref : 0.0f) > 0.0f) ? x : x * alpha;
            break;
        default:  // 10, 11: linear
            y = x;
            break;
    }
    return y;
}

// One workgroup = 256 threads = a contiguous chunk of one plane.
template <typename T, int VEC>
__global__ void __launch_bounds__(256)
fba_plane_kernel(T* __restrict__ out, const T* __restrict__ x, const T* __restrict__ bias,
                 const T* __restrict__ refer, int64_t step_b, int size_b, int chunks_per_plane,
                 int mode, float alpha, float scale) {
    constexpr int ITER = 4;
    const int64_t blk = blockIdx.x;
    const int64_t plane = blk / chunks_per_plane;
    const int chunk = (int)(blk - plane * chunks_per_plane);
    const float b = bias ? to_f32(bias[plane % size_b]) : 0.0f;
    const int64_t base = plane * step_b;
    const int64_t start = (int64_t)chunk * (256 * VEC * ITER);
#pragma unroll
    for (int it = 0; it < ITER; ++it) {
        const int64_t off = start + ((int64_t)it * 256 + threadIdx.x) * VEC;
        if (off >= step_b) break;
        if (VEC > 1 && off + VEC <= step_b) {
            // 16-byte path
            u128 vx = ld128(x + base + off);
            u128 vr = refer ? ld128(refer + base + off) : zero128();
            const T* ex = reinterpret_cast<const T*>(&vx);
            const T* er = reinterpret_cast<const T*>(&vr);
            u128 vo;
            T* eo = reinterpret_cast<T*>(&vo);
#pragma unroll
            for (int j = 0; j < VEC; ++j) {
                float v = to_f32(ex[j]) + b;
                v = act_apply(v, to_f32(er[j]), refer != nullptr, mode, alpha);
                eo[j] = from_f32<T>(v * scale);
            }
            st128(out + base + off, vo);
        } else {
            for (int j = 0; j < VEC && off + j < step_b; ++j) {
                float v = to_f32(x[base + off + j]) + b;
                float r = refer ? to_f32(refer[base + off + j]) : 0.0f;
                v = act_apply(v, r, refer != nullptr, mode, alpha);
                out[base + off + j] = from_f32<T>(v * scale);
            }
        }
    }
}

template <typename T>
__global__ void __launch_bounds__(256)
fba_flat_kernel(T* __restrict__ out, const T* __restrict__ x, const T* __restrict__ bias,
                const T* __restrict__ refer, int64_t numel, int64_t step_b, int size_b, int mode,
                float alpha, float scale) {
    const int64_t stride = (int64_t)gridDim.x * 256;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < numel; i += stride) {
        float v = to_f32(x[i]);
        if (bias) v += to_f32(bias[(i / step_b) % size_b]);
        float r = refer ? to_f32(refer[i]) : 0.0f;
        v = act_apply(v, r, refer != nullptr, mode, alpha);
        out[i] = from_f32<T>(v * scale);
    }
}

// VT_F64 (fused_bias_act_kernel.cu:96 dispatches double too): double tensors, double arithmetic, same operation order.
__global__ void __launch_bounds__(256)
fba_flat_f64_kernel(double* __restrict__ out, const double* __restrict__ x, const double* __restrict__ bias,
                    const double* __restrict__ refer, int64_t numel, int64_t step_b, int size_b, int mode,
                    double alpha, double scale) {
    const int64_t stride = (int64_t)gridDim.x * 256;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < numel; i += stride) {
        double v = x[i];
        if (bias) v += bias[(i / step_b) % size_b];
        const double r = refer ? refer[i] : 0.0;
        double y;
        switch (mode) {
            case 12: case 32: y = 0.0; break;
            case 30: y = (v > 0.0) ? v : v * alpha; break;
            case 31: y = (r > 0.0) ? v : v * alpha; break;
            default: y = v; break;
        }
        out[i] = y * scale;
    }
}

template <typename T>
int launch_fba(void* out, const void* x, const void* bias, const void* refer, int64_t numel,
               int64_t step_b, int size_b, int mode, float alpha, float scale, vt_stream stream) {
    constexpr int VEC = 16 / sizeof(T);
    const bool aligned = ((uintptr_t)out % 16 == 0) && ((uintptr_t)x % 16 == 0) &&
                         (!refer || (uintptr_t)refer % 16 == 0) && (step_b % VEC == 0);
    if (step_b >= 64 && numel % step_b == 0) {
        const int64_t planes = numel / step_b;
        if (aligned) {
            const int cpp = vt_cdiv(step_b, 256 * VEC * 4);
            const int64_t blocks = planes * cpp;
            if (blocks < (int64_t)1 << 31) {
                auto k = fba_plane_kernel<T, VEC>;
                VT_LAUNCH(k, dim3((unsigned)blocks), dim3(256), stream, (T*)out, (const T*)x,
                          (const T*)bias, (const T*)refer, step_b, size_b, cpp, mode, alpha, scale);
                return vt_check_launch("fused_bias_act(plane)");
            }
        } else {
            const int cpp = vt_cdiv(step_b, 256 * 4);
            const int64_t blocks = planes * cpp;
            if (blocks < (int64_t)1 << 31) {
                auto k = fba_plane_kernel<T, 1>;
                VT_LAUNCH(k, dim3((unsigned)blocks), dim3(256), stream, (T*)out, (const T*)x,
                          (const T*)bias, (const T*)refer, step_b, size_b, cpp, mode, alpha, scale);
                return vt_check_launch("fused_bias_act(plane,scalar)");
            }
        }
    }
    int64_t blocks = (numel + 255) / 256;
    if (blocks > 256 * 32) blocks = 256 * 32;  // grid-stride beyond 8192 workgroups
    auto k = fba_flat_kernel<T>;
    VT_LAUNCH(k, dim3((unsigned)blocks), dim3(256), stream, (T*)out, (const T*)x, (const T*)bias,
              (const T*)refer, numel, step_b, size_b, mode, alpha, scale);
    return vt_check_launch("fused_bias_act(flat)");
}

}  // namespace

extern "C" int vt_fused_bias_act(void* out, const void* x, const void* bias, const void* refer,
                                 int64_t numel, int64_t step_b, int size_b, int act, int grad,
                                 float alpha, float scale, int dtype, vt_stream stream) {
    VT_REQUIRE(numel >= 0, "vt_fused_bias_act: negative numel");
    if (numel == 0) return VT_OK;
    VT_REQUIRE(out && x, "vt_fused_bias_act: null tensor");
    VT_REQUIRE(step_b >= 1, "vt_fused_bias_act: step_b must be >= 1");
    VT_REQUIRE(!bias || size_b >= 1, "vt_fused_bias_act: bias given but size_b < 1");
    if (!bias) size_b = 1;
    const int mode = act * 10 + grad;
    switch (dtype) {
        case VT_F32:
            return launch_fba<float>(out, x, bias, refer, numel, step_b, size_b, mode, alpha, scale, stream);
        case VT_BF16:
            return launch_fba<bf16_t>(out, x, bias, refer, numel, step_b, size_b, mode, alpha, scale, stream);
        case VT_F16:
            return launch_fba<f16_t>(out, x, bias, refer, numel, step_b, size_b, mode, alpha, scale, stream);
        case VT_F64: {
            int64_t blocks = (numel + 255) / 256;
            if (blocks > 256 * 32) blocks = 256 * 32;
            auto k = fba_flat_f64_kernel;
            VT_LAUNCH(k, dim3((unsigned)blocks), dim3(256), stream, (double*)out, (const double*)x, (const double*)bias,
                      (const double*)refer, numel, step_b, size_b, mode, (double)alpha, (double)scale);
            return vt_check_launch("fused_bias_act(f64)");
        }
    }
    vt_set_error("vt_fused_bias_act: unsupported dtype %d", dtype);
    return VT_ERR_UNSUPPORTED;
}
