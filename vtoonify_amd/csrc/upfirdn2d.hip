// upfirdn2d for gfx950: zero-insert upsample -> pad/crop -> 2-D FIR -> decimate.
// Reference surface: model/stylegan/op/upfirdn2d.py:149-165; native op
// model/stylegan/op/upfirdn2d_kernel.cu:49-369; CPU twin op_cpu/upfirdn2d.py:20-60.
//
//   out[p, oy, ox] = sum_{ky,kx} Z[oy*down_y + ky, ox*down_x + kx] * fir[kh-1-ky, kw-1-kx]
//   Z[Y, X] = in[p, (Y-pad_y0)/up_y, (X-pad_x0)/up_x]  when both quotients are exact and
//             inside the image, else 0                          (negative pads crop)
//
// HBM-bound streaming op: algorithmic bytes = planes*(in_h*in_w + out_h*out_w)*sizeof(T).
//
// Kernels (written for CDNA4, not derived from the reference's CUDA tiling):
//   * upfirdn2d_tile<T,UP,DOWN>: up, down in {1,2} (same on both axes), FIR <= 4x4 -- the
//     three parameterisations the network uses (Blur, Upsample, Downsample:
//     model/stylegan/model.py:32-90).  A 256-thread workgroup produces a 32x64 output
//     tile of one plane.  The input window (with halo) and the flipped FIR taps are
//     staged in LDS once; every lane owns one output COLUMN and 8 consecutive rows, so
//     each store instruction of a wavefront writes 64 consecutive elements of one row
//     (one 256-byte fp32 / 128-byte bf16 segment) and LDS reads are lane-consecutive
//     (conflict free).  For up == 1 a vertical sliding window re-uses each LDS row read
//     for up to 4 output rows (44 reads / 8 outputs instead of 128).
//   * upfirdn2d_generic<T>: any up/down/pad/FIR shape (the separable 1x12 / 12x1 filters
//     with asymmetric factors of model/simple_augment.py:413-439); one output per lane,
//     only the taps that hit a real sample are visited.
// Accumulation is fp32 in (ky, kx) ascending order in both kernels.
#include "vt_common.hpp"

namespace {

__device__ __forceinline__ int posmod(int a, int m) {
    int r = a % m;
    return r < 0 ? r + m : r;
}
__device__ __forceinline__ int floordiv(int a, int b) {  // b > 0
    int q = a / b;
    return (a % b != 0 && a < 0) ? q - 1 : q;
}
__device__ __forceinline__ int ceildiv_s(int a, int b) { return -floordiv(-a, b); }

constexpr int KMAX = 4;
constexpr int TILE_W = 64;
constexpr int ROWS_PER_THREAD = 8;
constexpr int TILE_H = 4 * ROWS_PER_THREAD;  // 256 threads = 4 thread-rows x 64 columns

template <int UP, int DOWN>
struct TileGeom {
    static constexpr int IH = ((TILE_H - 1) * DOWN + KMAX - 1) / UP + 2;
    static constexpr int IW = ((TILE_W - 1) * DOWN + KMAX - 1) / UP + 2;
    static constexpr int LDW = IW | 1;  // odd row stride: column walks stay conflict free
};

template <typename T, int UP, int DOWN>
__global__ void __launch_bounds__(256)
upfirdn2d_tile(T* __restrict__ out, const T* __restrict__ in, const float* __restrict__ fir,
               int in_h, int in_w, int kh, int kw, int pad_x0, int pad_y0, int out_h, int out_w,
               int tiles_x, int tiles_y) {
    using G = TileGeom<UP, DOWN>;
    __shared__ float s_in[G::IH * G::LDW];
    __shared__ float s_k[KMAX * KMAX];

    const int tid = threadIdx.x;
    int64_t b = blockIdx.x;
    const int tile_x = (int)(b % tiles_x);
    b /= tiles_x;
    const int tile_y = (int)(b % tiles_y);
    const int64_t plane = b / tiles_y;

    const int oy0 = tile_y * TILE_H, ox0 = tile_x * TILE_W;
    // window of real input samples this tile can touch
    const int iy_lo = ceildiv_s(oy0 * DOWN - pad_y0, UP);
    const int ix_lo = ceildiv_s(ox0 * DOWN - pad_x0, UP);

    if (tid < KMAX * KMAX) {
        const int ky = tid / KMAX, kx = tid % KMAX;
        // flipped taps, zero beyond the real FIR size (upfirdn2d_kernel.cu:137 flips too)
        s_k[tid] = (ky < kh && kx < kw) ? fir[(kh - 1 - ky) * kw + (kw - 1 - kx)] : 0.0f;
    }
    const T* src = in + plane * (int64_t)in_h * in_w;
    {
        // all of a lane's window loads are issued before the first LDS write (independent loads in
        // flight together instead of a load -> wait -> ds_write chain per element)
        constexpr int NLD = (G::IH * G::IW + 255) / 256;
        float stg[NLD];
#pragma unroll
        for (int j = 0; j < NLD; ++j) {
            const int i = tid + 256 * j;
            const int r = i / G::IW, c = i - r * G::IW;
            const int iy = iy_lo + r, ix = ix_lo + c;
            const bool ok = i < G::IH * G::IW && iy >= 0 && iy < in_h && ix >= 0 && ix < in_w;
            // unconditional load at a clamped (always legal) address, masked afterwards: `ok ? load : 0` is a branch +
            // a vmcnt(0) wait per element
            const float v = to_f32(src[(int64_t)(ok ? iy : 0) * in_w + (ok ? ix : 0)]);
            stg[j] = ok ? v : 0.0f;
        }
#pragma unroll
        for (int j = 0; j < NLD; ++j) {
            const int i = tid + 256 * j;
            const int r = i / G::IW, c = i - r * G::IW;
            if (i < G::IH * G::IW) s_in[r * G::LDW + c] = stg[j];
        }
    }
    __syncthreads();

    const int tx = tid & 63, ty = tid >> 6;
    const int ox = ox0 + tx;
    const int oyb = oy0 + ty * ROWS_PER_THREAD;
    float acc[ROWS_PER_THREAD];
#pragma unroll
    for (int r = 0; r < ROWS_PER_THREAD; ++r) acc[r] = 0.0f;

    if constexpr (UP == 1) {
        // sliding window over the (ROWS-1)*DOWN + KMAX input rows this lane needs
        float kreg[KMAX * KMAX];
#pragma unroll
        for (int i = 0; i < KMAX * KMAX; ++i) kreg[i] = s_k[i];
        const int lx = ox * DOWN - pad_x0 - ix_lo;          // >= 0 by construction
        const int ly0 = oyb * DOWN - pad_y0 - iy_lo;
        constexpr int NROWS = (ROWS_PER_THREAD - 1) * DOWN + KMAX;
        // accumulate in (ky, kx) ascending order per output: iterate ky outer via row walk
#pragma unroll
        for (int ir = 0; ir < NROWS; ++ir) {
            float v[KMAX];
#pragma unroll
            for (int kx = 0; kx < KMAX; ++kx) v[kx] = s_in[(ly0 + ir) * G::LDW + lx + kx];
#pragma unroll
            for (int r = 0; r < ROWS_PER_THREAD; ++r) {
                const int ky = ir - r * DOWN;
                if (ky >= 0 && ky < KMAX) {
#pragma unroll
                    for (int kx = 0; kx < KMAX; ++kx) acc[r] = fmaf(v[kx], kreg[ky * KMAX + kx], acc[r]);
                }
            }
        }
    } else {
        // UP == 2 (DOWN == 1): output (Y, X) only sees the taps ky = Y0 (mod 2), kx = X0 (mod 2) of the zero-inserted
        // image -- 2 x 2 of the 4 x 4 FIR.  The column pattern is fixed per lane, the row pattern alternates with the
        // output row: both parities' four weights and LDS bases are formed ONCE, the 8 rows then cost 4 LDS reads +
        // 4 FMAs each in (ky, kx) ascending order.  (Round 1 re-derived posmod / floor divisions per output: ~40
        // integer instructions per element; 1.4 TB/s bf16 on a 32 x 512^2 -> 1024^2 tensor.)
        static_assert(UP == 2 && DOWN == 1, "up-sampling branch");
        const int X0 = ox - pad_x0;
        const int kx0 = posmod(-X0, 2);
        const int lx0 = floordiv(X0 + kx0, 2) - ix_lo;          // input column of tap kx0; tap kx0 + 2 reads lx0 + 1
        float wgt[2][2][2];
        int lyb[2];
#pragma unroll
        for (int par = 0; par < 2; ++par) {
            const int Y0 = oyb + par - pad_y0;
            const int ky0 = posmod(-Y0, 2);
            lyb[par] = floordiv(Y0 + ky0, 2) - iy_lo;           // input row of tap ky0 for output row oyb + par
#pragma unroll
            for (int jy = 0; jy < 2; ++jy)
#pragma unroll
                for (int jx = 0; jx < 2; ++jx) wgt[par][jy][jx] = s_k[(ky0 + 2 * jy) * KMAX + kx0 + 2 * jx];
        }
#pragma unroll
        for (int r = 0; r < ROWS_PER_THREAD; ++r) {
            const int par = r & 1;
            const float* row0 = s_in + (lyb[par] + (r >> 1)) * G::LDW + lx0;   // output row oyb + r: one input row lower every 2 rows
            float a = 0.0f;
            a = fmaf(row0[0], wgt[par][0][0], a);
            a = fmaf(row0[1], wgt[par][0][1], a);
            a = fmaf(row0[G::LDW], wgt[par][1][0], a);
            a = fmaf(row0[G::LDW + 1], wgt[par][1][1], a);
            acc[r] = a;
        }
    }

    if (ox < out_w) {
        T* dst = out + plane * (int64_t)out_h * out_w;
#pragma unroll
        for (int r = 0; r < ROWS_PER_THREAD; ++r) {
            const int oy = oyb + r;
            if (oy < out_h) dst[(int64_t)oy * out_w + ox] = from_f32<T>(acc[r]);
        }
    }
}

// ---------------------------------------------------------------------------------------
// Wide-plane FIR (up == down == 1, the Blur of model/stylegan/model.py:74-90 -- 96 % of the
// op's bytes in a frame).  HBM-bound: the point is to spend as few instructions per byte as
// possible.  A 256-thread workgroup produces a 16 x 256 output tile; every lane owns a 4 x 4
// block of outputs, so one row of its 7 x 7 input footprint is two 16-byte LDS reads
// (ds_read_b128, lane-consecutive => conflict-free) and each value read is used by up to 16
// FMAs.  Global loads are issued 5 rows x 5 column passes per lane back to back (all in flight
// together, 128 contiguous bytes per wavefront pass), stores are 4 consecutive elements per lane
// (8 B bf16 / 16 B fp32) => a wavefront writes 512 B / 1 KiB contiguous per row.
// Accumulation order per output is (ky, kx) ascending, as in the other kernels.
// ---------------------------------------------------------------------------------------
constexpr int W_TW = 256, W_TH = 16;                 // output tile
constexpr int W_IH = W_TH + KMAX - 1;                // 19 input rows
constexpr int W_IW = W_TW + 8;                       // 264 floats per LDS row (259 used; 16-byte rows)

template <typename T>
__global__ void __launch_bounds__(256)
upfirdn2d_wide(T* __restrict__ out, const T* __restrict__ in, const float* __restrict__ fir,
               int in_h, int in_w, int kh, int kw, int pad_x0, int pad_y0, int out_h, int out_w,
               int tiles_x, int tiles_y, int vec_store) {
    __shared__ __attribute__((aligned(16))) float s_in[20 * W_IW];   // 5 x 4 staged rows (19 used)
    __shared__ float s_k[KMAX * KMAX];
    const int tid = threadIdx.x;
    const int lx = tid & 63, ly = tid >> 6;
    int64_t b = blockIdx.x;
    const int tile_x = (int)(b % tiles_x);
    b /= tiles_x;
    const int tile_y = (int)(b % tiles_y);
    const int64_t plane = b / tiles_y;
    const int oy0 = tile_y * W_TH, ox0 = tile_x * W_TW;
    const int iy_lo = oy0 - pad_y0, ix_lo = ox0 - pad_x0;   // input coords of window (0,0)

    if (tid < KMAX * KMAX) {
        const int ky = tid / KMAX, kx = tid % KMAX;
        s_k[tid] = (ky < kh && kx < kw) ? fir[(kh - 1 - ky) * kw + (kw - 1 - kx)] : 0.0f;
    }
    const T* src = in + plane * (int64_t)in_h * in_w;
    // stage the window: wave `ly` takes rows ly, ly+4, ...; lanes sweep the columns
    // Branch-free: every load is issued unconditionally at a clamped (always legal) address and
    // masked afterwards -- per-load exec-mask branches cost ~20 scalar instructions each.
    float stg[5][5];
    int cidx[5];
    bool cmask[5];
#pragma unroll
    for (int cp = 0; cp < 5; ++cp) {
        const int c = lx + 64 * cp;
        const int ix = ix_lo + c;
        cmask[cp] = c < W_IW && ix >= 0 && ix < in_w;
        cidx[cp] = ix < 0 ? 0 : (ix >= in_w ? in_w - 1 : ix);
    }
#pragma unroll
    for (int rr = 0; rr < 5; ++rr) {
        const int r = ly + 4 * rr;
        const int iy = iy_lo + r;
        const bool rmask = r < W_IH && iy >= 0 && iy < in_h;
        const int iyc = iy < 0 ? 0 : (iy >= in_h ? in_h - 1 : iy);
        const T* rowp = src + (int64_t)iyc * in_w;
#pragma unroll
        for (int cp = 0; cp < 5; ++cp) {
            const float v = to_f32(rowp[cidx[cp]]);
            stg[rr][cp] = (rmask && cmask[cp]) ? v : 0.0f;   // select, not multiply: padding is exactly 0
        }
    }
#pragma unroll
    for (int rr = 0; rr < 5; ++rr) {
        const int r = ly + 4 * rr;
#pragma unroll
        for (int cp = 0; cp < 4; ++cp) s_in[r * W_IW + lx + 64 * cp] = stg[rr][cp];
        if (lx < W_IW - 256) s_in[r * W_IW + lx + 256] = stg[rr][4];
    }
    __syncthreads();
    float kreg[KMAX * KMAX];
#pragma unroll
    for (int i = 0; i < KMAX * KMAX; ++i) kreg[i] = s_k[i];
    float acc[4][4];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[j][i] = 0.0f;
#pragma unroll
    for (int r = 0; r < 7; ++r) {
        float v[8];
        const float* rowp = s_in + (ly * 4 + r) * W_IW + lx * 4;
        unpack16<float>(ld128(rowp), v);
        unpack16<float>(ld128(rowp + 4), v + 4);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int ky = r - j;
            if (ky >= 0 && ky < KMAX) {
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int kx = 0; kx < KMAX; ++kx) acc[j][i] = fmaf(v[i + kx], kreg[ky * KMAX + kx], acc[j][i]);
            }
        }
    }
    const int ox = ox0 + lx * 4;
    T* dst = out + plane * (int64_t)out_h * out_w;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int oy = oy0 + ly * 4 + j;
        if (oy >= out_h || ox >= out_w) continue;
        T* o = dst + (int64_t)oy * out_w + ox;
        if (vec_store && ox + 4 <= out_w) {
            if (sizeof(T) == 4) {
                st128(o, pack16<float>(acc[j]));
            } else {
                u64v pv;
                T e[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) e[i] = from_f32<T>(acc[j][i]);
                memcpy(&pv, e, 8);
                *reinterpret_cast<u64v*>(o) = pv;
            }
        } else {
            for (int i = 0; i < 4 && ox + i < out_w; ++i) o[i] = from_f32<T>(acc[j][i]);
        }
    }
}

template <typename T>
__global__ void __launch_bounds__(256)
upfirdn2d_generic(T* __restrict__ out, const T* __restrict__ in, const float* __restrict__ fir,
                  int64_t total, int in_h, int in_w, int kh, int kw, int up_x, int up_y,
                  int down_x, int down_y, int pad_x0, int pad_y0, int out_h, int out_w) {
    const int64_t stride = (int64_t)gridDim.x * 256;
    for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += stride) {
        const int ox = (int)(idx % out_w);
        const int64_t t = idx / out_w;
        const int oy = (int)(t % out_h);
        const int64_t plane = t / out_h;
        const T* src = in + plane * (int64_t)in_h * in_w;
        const int Y0 = oy * down_y - pad_y0, X0 = ox * down_x - pad_x0;
        float acc = 0.0f;
        for (int ky = posmod(-Y0, up_y); ky < kh; ky += up_y) {
            const int Y = Y0 + ky;
            if (Y < 0) continue;
            const int iy = Y / up_y;
            if (iy >= in_h) break;
            for (int kx = posmod(-X0, up_x); kx < kw; kx += up_x) {
                const int X = X0 + kx;
                if (X < 0) continue;
                const int ix = X / up_x;
                if (ix >= in_w) break;
                acc = fmaf(to_f32(src[(int64_t)iy * in_w + ix]), fir[(kh - 1 - ky) * kw + (kw - 1 - kx)], acc);
            }
        }
        out[idx] = from_f32<T>(acc);
    }
}

// VT_F64: the reference's AT_DISPATCH_FLOATING_TYPES_AND_HALF (upfirdn2d_kernel.cu:311) includes double -- tensors, taps and
// the accumulation are double there.  Off every path of the network; one output per lane, taps in (ky, kx) ascending order.
__global__ void __launch_bounds__(256)
upfirdn2d_generic_f64(double* __restrict__ out, const double* __restrict__ in, const double* __restrict__ fir,
                      int64_t total, int in_h, int in_w, int kh, int kw, int up_x, int up_y,
                      int down_x, int down_y, int pad_x0, int pad_y0, int out_h, int out_w) {
    const int64_t stride = (int64_t)gridDim.x * 256;
    for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += stride) {
        const int ox = (int)(idx % out_w);
        const int64_t t = idx / out_w;
        const int oy = (int)(t % out_h);
        const int64_t plane = t / out_h;
        const double* src = in + plane * (int64_t)in_h * in_w;
        const int Y0 = oy * down_y - pad_y0, X0 = ox * down_x - pad_x0;
        double acc = 0.0;
        for (int ky = posmod(-Y0, up_y); ky < kh; ky += up_y) {
            const int Y = Y0 + ky;
            if (Y < 0) continue;
            const int iy = Y / up_y;
            if (iy >= in_h) break;
            for (int kx = posmod(-X0, up_x); kx < kw; kx += up_x) {
                const int X = X0 + kx;
                if (X < 0) continue;
                const int ix = X / up_x;
                if (ix >= in_w) break;
                acc = fma(src[(int64_t)iy * in_w + ix], fir[(kh - 1 - ky) * kw + (kw - 1 - kx)], acc);
            }
        }
        out[idx] = acc;
    }
}

template <typename T>
int launch_upfirdn2d(void* out, const void* in, const float* fir, int64_t planes, int in_h,
                     int in_w, int kh, int kw, int up_x, int up_y, int down_x, int down_y,
                     int pad_x0, int pad_y0, int out_h, int out_w, vt_stream stream) {
    const bool tileable = up_x == up_y && down_x == down_y && (up_x == 1 || up_x == 2) &&
                          (down_x == 1 || down_x == 2) && !(up_x == 2 && down_x == 2) &&
                          kh <= KMAX && kw <= KMAX;
    if (tileable && up_x == 1 && down_x == 1 && out_w >= 192) {
        const int tiles_x = vt_cdiv(out_w, W_TW), tiles_y = vt_cdiv(out_h, W_TH);
        const int64_t blocks = planes * tiles_x * tiles_y;
        if (blocks < ((int64_t)1 << 31)) {
            const int esz = (int)sizeof(T);
            const int vec = ((uintptr_t)out % (4 * esz) == 0) && (out_w % 4 == 0);
            auto k = upfirdn2d_wide<T>;
            VT_LAUNCH(k, dim3((unsigned)blocks), dim3(256), stream, (T*)out, (const T*)in, fir, in_h, in_w, kh, kw,
                      pad_x0, pad_y0, out_h, out_w, tiles_x, tiles_y, vec);
            return vt_check_launch("upfirdn2d(wide)");
        }
    }
    if (tileable) {
        const int tiles_x = vt_cdiv(out_w, TILE_W), tiles_y = vt_cdiv(out_h, TILE_H);
        const int64_t blocks = planes * tiles_x * tiles_y;
        if (blocks < ((int64_t)1 << 31)) {
            dim3 grid((unsigned)blocks), block(256);
            if (up_x == 1 && down_x == 1) {
                auto k = upfirdn2d_tile<T, 1, 1>;
                VT_LAUNCH(k, grid, block, stream, (T*)out, (const T*)in, fir, in_h, in_w, kh, kw,
                          pad_x0, pad_y0, out_h, out_w, tiles_x, tiles_y);
            } else if (up_x == 2) {
                auto k = upfirdn2d_tile<T, 2, 1>;
                VT_LAUNCH(k, grid, block, stream, (T*)out, (const T*)in, fir, in_h, in_w, kh, kw,
                          pad_x0, pad_y0, out_h, out_w, tiles_x, tiles_y);
            } else {
                auto k = upfirdn2d_tile<T, 1, 2>;
                VT_LAUNCH(k, grid, block, stream, (T*)out, (const T*)in, fir, in_h, in_w, kh, kw,
                          pad_x0, pad_y0, out_h, out_w, tiles_x, tiles_y);
            }
            return vt_check_launch("upfirdn2d(tile)");
        }
    }
    const int64_t total = planes * out_h * out_w;
    int64_t blocks = (total + 255) / 256;
    if (blocks > 256 * 32) blocks = 256 * 32;
    auto k = upfirdn2d_generic<T>;
    VT_LAUNCH(k, dim3((unsigned)blocks), dim3(256), stream, (T*)out, (const T*)in, fir, total, in_h,
              in_w, kh, kw, up_x, up_y, down_x, down_y, pad_x0, pad_y0, out_h, out_w);
    return vt_check_launch("upfirdn2d(generic)");
}

}  // namespace

extern "C" int vt_upfirdn2d_out_size(int in_h, int in_w, int kh, int kw, int up_x, int up_y,
                                     int down_x, int down_y, int pad_x0, int pad_x1, int pad_y0,
                                     int pad_y1, int* out_h, int* out_w) {
    VT_REQUIRE(up_x >= 1 && up_y >= 1 && down_x >= 1 && down_y >= 1,
               "vt_upfirdn2d: up/down factors must be >= 1");
    VT_REQUIRE(kh >= 1 && kw >= 1, "vt_upfirdn2d: empty FIR kernel");
    // op/upfirdn2d.py:104-105 (Python floor division)
    const int nh = in_h * up_y + pad_y0 + pad_y1 - kh + down_y;
    const int nw = in_w * up_x + pad_x0 + pad_x1 - kw + down_x;
    *out_h = nh >= 0 ? nh / down_y : -((-nh + down_y - 1) / down_y);
    *out_w = nw >= 0 ? nw / down_x : -((-nw + down_x - 1) / down_x);
    return VT_OK;
}

extern "C" int vt_upfirdn2d(void* out, const void* in, const float* fir, int64_t planes, int in_h,
                            int in_w, int kh, int kw, int up_x, int up_y, int down_x, int down_y,
                            int pad_x0, int pad_x1, int pad_y0, int pad_y1, int dtype,
                            vt_stream stream) {
    int out_h = 0, out_w = 0;
    int rc = vt_upfirdn2d_out_size(in_h, in_w, kh, kw, up_x, up_y, down_x, down_y, pad_x0, pad_x1,
                                   pad_y0, pad_y1, &out_h, &out_w);
    if (rc != VT_OK) return rc;
    VT_REQUIRE(planes >= 0 && in_h >= 0 && in_w >= 0, "vt_upfirdn2d: negative size");
    VT_REQUIRE(out_h > 0 && out_w > 0,
               "vt_upfirdn2d: empty output (%d x %d): pads crop away the whole image", out_h, out_w);
    if (planes == 0) return VT_OK;
    VT_REQUIRE(out && in && fir, "vt_upfirdn2d: null tensor");
    VT_REQUIRE((int64_t)in_h * up_y < (1 << 30) && (int64_t)in_w * up_x < (1 << 30),
               "vt_upfirdn2d: image too large for 32-bit index arithmetic");
    switch (dtype) {
        case VT_F32:
            return launch_upfirdn2d<float>(out, in, fir, planes, in_h, in_w, kh, kw, up_x, up_y,
                                           down_x, down_y, pad_x0, pad_y0, out_h, out_w, stream);
        case VT_BF16:
            return launch_upfirdn2d<bf16_t>(out, in, fir, planes, in_h, in_w, kh, kw, up_x, up_y,
                                            down_x, down_y, pad_x0, pad_y0, out_h, out_w, stream);
        case VT_F16:
            return launch_upfirdn2d<f16_t>(out, in, fir, planes, in_h, in_w, kh, kw, up_x, up_y,
                                           down_x, down_y, pad_x0, pad_y0, out_h, out_w, stream);
        case VT_F64: {
            const int64_t total = planes * out_h * out_w;
            int64_t blocks = (total + 255) / 256;
            if (blocks > 256 * 32) blocks = 256 * 32;
            auto k = upfirdn2d_generic_f64;
            VT_LAUNCH(k, dim3((unsigned)blocks), dim3(256), stream, (double*)out, (const double*)in,
                      reinterpret_cast<const double*>(fir), total, in_h, in_w, kh, kw, up_x, up_y, down_x, down_y,
                      pad_x0, pad_y0, out_h, out_w);
            return vt_check_launch("upfirdn2d(f64)");
        }
    }
    vt_set_error("vt_upfirdn2d: unsupported dtype %d", dtype);
    return VT_ERR_UNSUPPORTED;
}
