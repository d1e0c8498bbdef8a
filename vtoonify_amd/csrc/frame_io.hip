// Frame packing / unpacking either side of VToonify.forward (gfx950) -- the per-frame host work of
// the reference's video loop moved onto the GPU:
//
//   vt_frame_pack    uint8 HWC frame (+ optional fp32 parsing map) -> the fp32 NCHW network input
//                      x[c] = ((u8 / 255) - 0.5) / 0.5                (transforms.ToTensor + Normalize,
//                                                                      style_transfer.py:57-60,160)
//                      x[3+j] = parsing[j] * parsing_scale            (x_p / 16., style_transfer.py:174)
//                    with the BGR->RGB swap of cv2.cvtColor (style_transfer.py:114) folded in.
//   vt_frame_unpack  fp32 NCHW image -> uint8 HWC frame
//                      u8 = (uint8)((clamp(y, -1, 1) + 1.0) * 127.5)  (torch.clamp style_transfer.py:177,
//                                                                      tensor2cv2 util.py:190-192)
//                    with the RGB->BGR swap of tensor2cv2 folded in.
//
// Both are pure streaming kernels: 4 pixels per lane (12 payload bytes <-> three 16-byte plane
// vectors), every plane access a coalesced 16-byte vector.  The fp32 arithmetic is the reference's op
// sequence (div, sub, div / add, mul, truncate), unfused, so results are BIT-exact against the
// numpy / torch formulas (tests/test_video.py).
// Algorithmic bytes: pack  n*h*w*(3 + 4*(3+pc) + 4*pc);  unpack  n*H*W*(12 + 3).
#include "vt_common.hpp"

namespace {

__device__ __forceinline__ float norm_u8(unsigned v) { return ((float)v / 255.0f - 0.5f) / 0.5f; }

__device__ __forceinline__ unsigned quant_u8(float y) {
    y = fminf(fmaxf(y, -1.0f), 1.0f);
    return (unsigned)(int)((y + 1.0f) * 127.5f);   // truncation, as ndarray.astype(np.uint8)
}

// one thread = 4 consecutive pixels of one image (hw % 4 == 0) or 1 pixel (G = 1)
template <int G>
__global__ void __launch_bounds__(256)
frame_pack_kernel(float* __restrict__ x, const unsigned char* __restrict__ frames,
                  const float* __restrict__ parsing, int n, int hw, int pc, float pscale, int swap_rb) {
    const int groups = hw / G;
    const int64_t total = (int64_t)n * groups;
    const int ct = 3 + pc;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int img = (int)(i / groups);
        const int p0 = (int)(i - (int64_t)img * groups) * G;
        const unsigned char* src = frames + ((int64_t)img * hw + p0) * 3;
        float* dst = x + (int64_t)img * ct * hw + p0;
        unsigned char b[3 * G];
        if (G == 4) {
            const uint32_t* s32 = reinterpret_cast<const uint32_t*>(src);   // 12-byte aligned group
            const uint32_t w0 = s32[0], w1 = s32[1], w2 = s32[2];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                b[k] = (w0 >> (8 * k)) & 255;
                b[4 + k] = (w1 >> (8 * k)) & 255;
                b[8 + k] = (w2 >> (8 * k)) & 255;
            }
        } else {
#pragma unroll
            for (int k = 0; k < 3 * G; ++k) b[k] = src[k];
        }
#pragma unroll
        for (int ch = 0; ch < 3; ++ch) {
            const int sc = swap_rb ? 2 - ch : ch;
            float v[G];
#pragma unroll
            for (int k = 0; k < G; ++k) v[k] = norm_u8(b[3 * k + sc]);
            if (G == 4) {
                st128(dst + (int64_t)ch * hw, pack16<float>(v));
            } else {
                dst[(int64_t)ch * hw] = v[0];
            }
        }
        for (int j = 0; j < pc; ++j) {
            const float* ps = parsing + ((int64_t)img * pc + j) * hw + p0;
            float* pd = dst + (int64_t)(3 + j) * hw;
            if (G == 4) {
                float q[4];
                unpack16<float>(ld128(ps), q);
#pragma unroll
                for (int k = 0; k < 4; ++k) q[k] *= pscale;
                st128(pd, pack16<float>(q));
            } else {
                pd[0] = ps[0] * pscale;
            }
        }
    }
}

template <int G>
__global__ void __launch_bounds__(256)
frame_unpack_kernel(unsigned char* __restrict__ frames, const float* __restrict__ image, int n, int hw,
                    int swap_rb) {
    const int groups = hw / G;
    const int64_t total = (int64_t)n * groups;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int img = (int)(i / groups);
        const int p0 = (int)(i - (int64_t)img * groups) * G;
        const float* src = image + (int64_t)img * 3 * hw + p0;
        unsigned char* dst = frames + ((int64_t)img * hw + p0) * 3;
        unsigned b[3 * G];
#pragma unroll
        for (int ch = 0; ch < 3; ++ch) {
            const int dc = swap_rb ? 2 - ch : ch;
            float v[G];
            if (G == 4) {
                unpack16<float>(ld128(src + (int64_t)ch * hw), v);
            } else {
                v[0] = src[(int64_t)ch * hw];
            }
#pragma unroll
            for (int k = 0; k < G; ++k) b[3 * k + dc] = quant_u8(v[k]);
        }
        if (G == 4) {
            uint32_t* d32 = reinterpret_cast<uint32_t*>(dst);
            d32[0] = b[0] | (b[1] << 8) | (b[2] << 16) | (b[3] << 24);
            d32[1] = b[4] | (b[5] << 8) | (b[6] << 16) | (b[7] << 24);
            d32[2] = b[8] | (b[9] << 8) | (b[10] << 16) | (b[11] << 24);
        } else {
#pragma unroll
            for (int k = 0; k < 3 * G; ++k) dst[k] = (unsigned char)b[k];
        }
    }
}

inline unsigned grid_for(int64_t total) {
    int64_t b = (total + 255) / 256;
    if (b > 65536) b = 65536;
    if (b < 1) b = 1;
    return (unsigned)b;
}

}  // namespace

extern "C" int vt_frame_pack(float* x, const uint8_t* frames, int swap_rb, const float* parsing, int parsing_channels,
                             float parsing_scale, int n, int h, int w, vt_stream stream) {
    VT_REQUIRE(x && frames, "vt_frame_pack: null tensor");
    VT_REQUIRE(n > 0 && h > 0 && w > 0 && (int64_t)h * w < ((int64_t)1 << 31), "vt_frame_pack: bad sizes");
    VT_REQUIRE(parsing_channels >= 0 && (parsing_channels == 0 || parsing), "vt_frame_pack: parsing map missing");
    const int hw = h * w;
    const bool vec = hw % 4 == 0 && (uintptr_t)x % 16 == 0 && (uintptr_t)frames % 4 == 0 &&
                     (parsing_channels == 0 || (uintptr_t)parsing % 16 == 0);
    if (vec) {
        auto k = frame_pack_kernel<4>;
        VT_LAUNCH(k, dim3(grid_for((int64_t)n * hw / 4)), dim3(256), stream, x, (const unsigned char*)frames, parsing,
                  n, hw, parsing_channels, parsing_scale, swap_rb);
    } else {
        auto k = frame_pack_kernel<1>;
        VT_LAUNCH(k, dim3(grid_for((int64_t)n * hw)), dim3(256), stream, x, (const unsigned char*)frames, parsing, n,
                  hw, parsing_channels, parsing_scale, swap_rb);
    }
    return vt_check_launch("vt_frame_pack");
}

extern "C" int vt_frame_unpack(uint8_t* frames, const float* image, int swap_rb, int n, int h, int w,
                               vt_stream stream) {
    VT_REQUIRE(frames && image, "vt_frame_unpack: null tensor");
    VT_REQUIRE(n > 0 && h > 0 && w > 0 && (int64_t)h * w < ((int64_t)1 << 31), "vt_frame_unpack: bad sizes");
    const int hw = h * w;
    const bool vec = hw % 4 == 0 && (uintptr_t)image % 16 == 0 && (uintptr_t)frames % 4 == 0;
    if (vec) {
        auto k = frame_unpack_kernel<4>;
        VT_LAUNCH(k, dim3(grid_for((int64_t)n * hw / 4)), dim3(256), stream, (unsigned char*)frames, image, n, hw, swap_rb);
    } else {
        auto k = frame_unpack_kernel<1>;
        VT_LAUNCH(k, dim3(grid_for((int64_t)n * hw)), dim3(256), stream, (unsigned char*)frames, image, n, hw, swap_rb);
    }
    return vt_check_launch("vt_frame_unpack");
}
