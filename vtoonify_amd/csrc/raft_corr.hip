// Correlation lookup of RAFT's AlternateCorrBlock -- the reference's third native extension
// (model/raft/alt_cuda_corr/correlation_kernel.cu:19-120, bound as alt_cuda_corr.forward in
// correlation.cpp:24-34; caller model/raft/core/corr.py:63-91).  For every pixel p of frame 1 with
// target coordinate (x, y) in frame 2 and every offset (a, b) in [0, 2r]^2:
//     corr[b_, 0, a + (2r+1) * b, p] = bilinear_{(y - r + a, x - r + b)} < fmap1[p, :], fmap2[., ., :] >
// (zero outside fmap2), i.e. the (2r+2)^2 integer-grid dot products around floor(x, y) blended
// with the fractional part -- the all-pairs correlation volume is never materialised.
//
// gfx950 mapping: one wavefront per output pixel.  fmap1[p, :] is staged in LDS once and read back as
// a broadcast; every lane owns grid points (lane, lane + 64, ...) of the (2r+2)^2 window and walks
// the channels of "its" fmap2 pixel with 16-byte loads (neighbouring lanes touch neighbouring 4C-byte
// pixel rows; windows of neighbouring output pixels overlap, so the reads are L2 hits), the window
// sums go through LDS, and lanes 0..(2r+1)^2-1 blend and store.  No atomics: the reference's
// read-modify-write of corr (`+=` per channel stride) becomes a register accumulation.
//   vt_avgpool2x2  F.avg_pool2d(x, 2, stride=2) on NHWC fp32 (the fmap2 pyramid, corr.py:68-71)
#include "vt_common.hpp"

namespace {

constexpr int CORR_WAVES = 4;          // output pixels per workgroup
constexpr int CORR_MAX_GRID = 324;     // (2r+2)^2 for r <= 8
constexpr int CORR_MAX_C = 512;

__global__ void __launch_bounds__(CORR_WAVES * 64)
corr_lookup_kernel(float* __restrict__ corr, const float* __restrict__ fmap1, const float* __restrict__ fmap2,
                   const float* __restrict__ coords, int B, int H1, int W1, int H2, int W2, int C, int r,
                   float scale, float coord_scale) {
    __shared__ __attribute__((aligned(16))) float f1s[CORR_WAVES][CORR_MAX_C];
    __shared__ float sgrid[CORR_WAVES][CORR_MAX_GRID];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int64_t npix = (int64_t)B * H1 * W1;
    const int64_t p = (int64_t)blockIdx.x * CORR_WAVES + wave;
    const bool live = p < npix;
    const int rd = 2 * r + 1, g = rd + 1, ng = g * g;
    float x = 0.0f, y = 0.0f;
    int b = 0;
    if (live) {
        b = (int)(p / ((int64_t)H1 * W1));
        x = coords[p * 2 + 0] * coord_scale;   // coords / 2**level (corr.py:84): exact for powers of two
        y = coords[p * 2 + 1] * coord_scale;
        for (int c = lane * 4; c < C; c += 256) st128(&f1s[wave][c], ld128(fmap1 + p * C + c));
    }
    __syncthreads();
    const float fx = floorf(x), fy = floorf(y);
    const float dx = x - fx, dy = y - fy;
    const int bx = (int)fx - r, by = (int)fy - r;
    if (live) {
        for (int gi = lane; gi < ng; gi += 64) {
            const int iy = gi / g, ix = gi - iy * g;
            const int h2 = by + iy, w2 = bx + ix;
            float s = 0.0f;
            if (h2 >= 0 && h2 < H2 && w2 >= 0 && w2 < W2) {
                const float* q = fmap2 + (((int64_t)b * H2 + h2) * W2 + w2) * C;
                for (int c = 0; c < C; c += 4) {
                    float a4[4], b4[4];
                    unpack16<float>(ld128(&f1s[wave][c]), a4);
                    unpack16<float>(ld128(q + c), b4);
                    s += a4[0] * b4[0];
                    s += a4[1] * b4[1];
                    s += a4[2] * b4[2];
                    s += a4[3] * b4[3];
                }
            }
            sgrid[wave][gi] = s;
        }
    }
    __syncthreads();
    if (live) {
        const int hw = H1 * W1;
        const int64_t pin = p - (int64_t)b * hw;
        float* out = corr + (int64_t)b * rd * rd * hw + pin;
        for (int o = lane; o < rd * rd; o += 64) {
            const int a = o % rd, bb = o / rd;   // channel = a (y offset) + rd * b (x offset), kernel.cu:92-95
            const float* sg = sgrid[wave];
            const float v = (1.0f - dy) * (1.0f - dx) * sg[a * g + bb] + (1.0f - dy) * dx * sg[a * g + bb + 1] +
                            dy * (1.0f - dx) * sg[(a + 1) * g + bb] + dy * dx * sg[(a + 1) * g + bb + 1];
            out[(int64_t)o * hw] = v * scale;
        }
    }
}

__global__ void __launch_bounds__(256)
avgpool2x2_kernel(float* __restrict__ out, const float* __restrict__ x, int n, int h, int w, int c) {
    const int oh = h / 2, ow = w / 2, cv = c / 4;
    const int64_t total = (int64_t)n * oh * ow * cv;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int v = (int)(i % cv);
        const int64_t pix = i / cv;
        const int ox = (int)(pix % ow);
        const int64_t t = pix / ow;
        const int oy = (int)(t % oh), img = (int)(t / oh);
        const float* p = x + (((int64_t)img * h + 2 * oy) * w + 2 * ox) * c + v * 4;
        float a[4], b[4], cc[4], d[4], r[4];
        unpack16<float>(ld128(p), a);
        unpack16<float>(ld128(p + c), b);
        unpack16<float>(ld128(p + (int64_t)w * c), cc);
        unpack16<float>(ld128(p + (int64_t)w * c + c), d);
#pragma unroll
        for (int k = 0; k < 4; ++k) r[k] = (((a[k] + b[k]) + cc[k]) + d[k]) * 0.25f;
        st128(out + pix * c + v * 4, pack16<float>(r));
    }
}

}  // namespace

extern "C" int vt_corr_lookup(float* corr, const float* fmap1, const float* fmap2, const float* coords, int batch,
                              int h1, int w1, int h2, int w2, int c, int radius, float scale, float coord_scale,
                              vt_stream stream) {
    VT_REQUIRE(corr && fmap1 && fmap2 && coords, "vt_corr_lookup: null tensor");
    VT_REQUIRE(batch > 0 && h1 > 0 && w1 > 0 && h2 > 0 && w2 > 0, "vt_corr_lookup: bad sizes");
    VT_REQUIRE(c > 0 && c % 4 == 0 && c <= CORR_MAX_C, "vt_corr_lookup: channels must be a multiple of 4, <= 512");
    VT_REQUIRE(radius >= 0 && (2 * radius + 2) * (2 * radius + 2) <= CORR_MAX_GRID, "vt_corr_lookup: radius must be <= 8");
    const int64_t npix = (int64_t)batch * h1 * w1;
    VT_REQUIRE(npix < ((int64_t)1 << 31), "vt_corr_lookup: too many pixels");
    const unsigned blocks = (unsigned)((npix + CORR_WAVES - 1) / CORR_WAVES);
    VT_LAUNCH(corr_lookup_kernel, dim3(blocks), dim3(CORR_WAVES * 64), stream, corr, fmap1, fmap2, coords, batch, h1, w1,
              h2, w2, c, radius, scale, coord_scale);
    return vt_check_launch("vt_corr_lookup");
}

extern "C" int vt_avgpool2x2(float* out, const float* x, int n, int h, int w, int c, vt_stream stream) {
    VT_REQUIRE(out && x, "vt_avgpool2x2: null tensor");
    VT_REQUIRE(n > 0 && h >= 2 && w >= 2 && c > 0 && c % 4 == 0, "vt_avgpool2x2: needs h, w >= 2 and c % 4 == 0");
    int64_t total = (int64_t)n * (h / 2) * (w / 2) * (c / 4);
    int64_t blocks = (total + 255) / 256;
    if (blocks > 65536) blocks = 65536;
    VT_LAUNCH(avgpool2x2_kernel, dim3((unsigned)blocks), dim3(256), stream, out, x, n, h, w, c);
    return vt_check_launch("vt_avgpool2x2");
}
