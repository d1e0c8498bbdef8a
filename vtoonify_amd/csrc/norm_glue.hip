// Normalisation and fusion glue on NHWC activations (gfx950), all HBM-streaming:
//   vt_instnorm_stats  nn.InstanceNorm2d statistics folded with the AdaIN affine
//                      (model/dualstylegan.py:6-21) into per-(n,c) scale/shift
//   vt_affine_apply    x*scale+shift, optionally on cat[x, |x-other|] (model/vtoonify.py:125)
//   vt_fusion_pack     [skip | f_E * m_E] operand of fusion_out.conv / fusion_skip
//                      (model/vtoonify.py:127, 259-262)
//   vt_nchw_to_nhwc / vt_nhwc_to_nchw   layout change at the model boundary
//
// Statistics are deterministic (no float atomics): every workgroup reduces a chunk of
// pixels to per-channel {x0, sum(x-x0), sum((x-x0)^2)} with the chunk's first pixel as the
// shift x0 (kills the E[x^2]-E[x]^2 cancellation), and a second kernel merges the chunks
// in fixed order with Chan's parallel-variance update in fp64.
#include "vt_common.hpp"

namespace {

constexpr float IN_EPS = 1e-5f;  // nn.InstanceNorm2d default (dualstylegan.py:10)

// One workgroup reduces a chunk of pixels for ALL channels in a single pass: x (and `other`) are
// read once, the statistics of x and of |x - other| are accumulated together, and 4 pixels per
// thread are in flight (independent 16-byte loads, accumulated in pixel order).
template <typename T, bool HAS_OTHER>
__global__ void __launch_bounds__(256)
instnorm_partial_kernel(StatRec* __restrict__ part, const T* __restrict__ x, int ld_x,
                        const T* __restrict__ other, int ld_o, int hw, int c, int chunk_px,
                        int chunks) {
    constexpr int VEC = 16 / sizeof(T);
    constexpr int UNR = 4;
    constexpr int HALVES = HAS_OTHER ? 2 : 1;
    __shared__ float red[256 * VEC * 2];
    const int tid = threadIdx.x;
    const int chunk = blockIdx.x % chunks, img = blockIdx.x / chunks;
    const int p_lo = chunk * chunk_px;
    const int p_hi = (p_lo + chunk_px < hw) ? p_lo + chunk_px : hw;
    const int cvn = c / VEC;
    const int cpar = cvn < 256 ? cvn : 256;  // channel-vectors handled in parallel
    const int rows = 256 / cpar;             // pixel rows handled in parallel
    const int cv0 = tid % cpar, prow = tid / cpar;
    const bool active = prow < rows;
    const int ctot = c * HALVES;
    const T* xb = x + (int64_t)img * hw * ld_x;
    const T* ob = HAS_OTHER ? other + (int64_t)img * hw * ld_o : nullptr;

    for (int cbase = 0; cbase < cvn; cbase += cpar) {
        const int cv = cbase + cv0;
        const bool on = active && cv < cvn;
        float x0a[VEC], s1a[VEC], s2a[VEC], x0b[VEC], s1b[VEC], s2b[VEC];
#pragma unroll
        for (int i = 0; i < VEC; ++i) x0a[i] = s1a[i] = s2a[i] = x0b[i] = s1b[i] = s2b[i] = 0.0f;
        if (on) {
            {   // shift = the chunk's first pixel (kills the E[x^2]-E[x]^2 cancellation)
                float f[VEC];
                unpack16<T>(ld128(xb + (int64_t)p_lo * ld_x + cv * VEC), f);
#pragma unroll
                for (int i = 0; i < VEC; ++i) x0a[i] = f[i];
                if (HAS_OTHER) {
                    float g[VEC];
                    unpack16<T>(ld128(ob + (int64_t)p_lo * ld_o + cv * VEC), g);
#pragma unroll
                    for (int i = 0; i < VEC; ++i) x0b[i] = fabsf(f[i] - g[i]);
                }
            }
            for (int px = p_lo + prow; px < p_hi; px += rows * UNR) {
                u128 vx[UNR], vo[UNR];
#pragma unroll
                for (int u = 0; u < UNR; ++u) {
                    const int q = px + u * rows;
                    const int qq = q < p_hi ? q : p_lo;  // clamp: keep every load unconditional
                    vx[u] = ld128(xb + (int64_t)qq * ld_x + cv * VEC);
                    if (HAS_OTHER) vo[u] = ld128(ob + (int64_t)qq * ld_o + cv * VEC);
                }
#pragma unroll
                for (int u = 0; u < UNR; ++u) {
                    const float live = (px + u * rows < p_hi) ? 1.0f : 0.0f;
                    float f[VEC], g[VEC];
                    unpack16<T>(vx[u], f);
                    if (HAS_OTHER) unpack16<T>(vo[u], g);
#pragma unroll
                    for (int i = 0; i < VEC; ++i) {
                        const float d = (f[i] - x0a[i]) * live;
                        s1a[i] += d;
                        s2a[i] += d * d;
                        if (HAS_OTHER) {
                            const float e = (fabsf(f[i] - g[i]) - x0b[i]) * live;
                            s1b[i] += e;
                            s2b[i] += e * e;
                        }
                    }
                }
            }
        }
#pragma unroll
        for (int half = 0; half < HALVES; ++half) {
            __syncthreads();
#pragma unroll
            for (int i = 0; i < VEC; ++i) {
                red[(tid * VEC + i) * 2 + 0] = half ? s1b[i] : s1a[i];
                red[(tid * VEC + i) * 2 + 1] = half ? s2b[i] : s2a[i];
            }
            __syncthreads();
            // fold the pixel rows in order r = 0 .. rows - 1, one (channel, sum) per thread: cpar * VEC * 2 sums over the 256
            // threads.  (Round 6: the thread of pixel row 0 used to fold all 2 * VEC sums of its channel vector by itself --
            // 2 * VEC * rows dependent LDS reads on 1 thread in 16, 8-16 us of tail per chunk at the 128 / 256-channel levels.)
            // The same additions in the same order: the same records.
            StatRec* rec0 = part + ((int64_t)img * chunks + chunk) * ctot + half * c + cbase * VEC;
            const int nsum = cpar * VEC * 2;
            for (int o = tid; o < nsum; o += 256) {
                const int which = o & 1, ch = o >> 1;             // ch = cv0' * VEC + i inside this channel block
                if (cbase * VEC + ch < c) {
                    float a = 0.0f;
                    for (int r = 0; r < rows; ++r) a += red[((r * cpar) * VEC + ch) * 2 + which];
                    (which ? rec0[ch].s2 : rec0[ch].s1) = a;
                }
            }
            if (on && prow == 0) {
#pragma unroll
                for (int i = 0; i < VEC; ++i) rec0[cv0 * VEC + i].x0 = half ? x0b[i] : x0a[i];
            }
        }
    }
}

// Merge the chunk records of 16 channels per workgroup: 16 chunk-lanes per channel each fold every
// 16th chunk, the lanes are combined through LDS in lane order.  Two passes over the records, fp64,
// no divisions in the loops (the Chan-update form this replaces spent 10 us per launch in fp64
// divides):   mean = sum_k (x0_k n_k + s1_k) / hw
//             M2   = sum_k [ s2_k - 2 d_k s1_k + n_k d_k^2 ],  d_k = mean - x0_k
// The order is a function of `chunks` only => deterministic and batch-independent.
__global__ void __launch_bounds__(256)
instnorm_finalize_kernel(float* __restrict__ scale, float* __restrict__ shift,
                         const StatRec* __restrict__ part, int n, int hw, int ctot, int chunk_px,
                         int chunks, const float* __restrict__ style_gb, int ld_gb) {
    __shared__ double red[16][17];
    const int chl = threadIdx.x & 15, cl = threadIdx.x >> 4;
    const int groups = ctot / 16;  // ctot is a multiple of 8; host guarantees 16 here
    const int img = blockIdx.x / groups, ch = (blockIdx.x % groups) * 16 + chl;
    const StatRec* pc = part + (int64_t)img * chunks * ctot + ch;
    double sum = 0.0;
    for (int k = cl; k < chunks; k += 16) {
        const StatRec r = pc[(int64_t)k * ctot];
        int npx = hw - k * chunk_px;
        if (npx > chunk_px) npx = chunk_px;
        sum += (double)r.x0 * (double)npx + (double)r.s1;
    }
    red[cl][chl] = sum;
    __syncthreads();
    double tot = 0.0;
#pragma unroll
    for (int l = 0; l < 16; ++l) tot += red[l][chl];   // same fixed order in every thread
    const double mean = tot / (double)hw;
    __syncthreads();
    double m2 = 0.0;
    for (int k = cl; k < chunks; k += 16) {
        const StatRec r = pc[(int64_t)k * ctot];
        int npx = hw - k * chunk_px;
        if (npx > chunk_px) npx = chunk_px;
        const double d = mean - (double)r.x0;
        m2 += (double)r.s2 - 2.0 * d * (double)r.s1 + (double)npx * d * d;
    }
    red[cl][chl] = m2;
    __syncthreads();
    if (cl != 0) return;
    double t2 = 0.0;
#pragma unroll
    for (int l = 0; l < 16; ++l) t2 += red[l][chl];
    double var = t2 / (double)hw;  // biased, as F.instance_norm
    if (var < 0.0) var = 0.0;
    const float rstd = (float)(1.0 / sqrt(var + (double)IN_EPS));
    float gamma = 1.0f, beta = 0.0f;
    if (style_gb) {
        gamma = style_gb[(int64_t)img * ld_gb + ch];
        beta = style_gb[(int64_t)img * ld_gb + ctot + ch];
    }
    const int idx = img * ctot + ch;
    scale[idx] = gamma * rstd;
    shift[idx] = beta - gamma * rstd * (float)mean;
}

// finalize + apply in one launch for SMALL tensors (the 32x32-pixel trunk: 12 AdaINs per frame,
// each 5-6 us of pure launch latency as two kernels).  One workgroup per (image, 16-byte channel
// vector): it folds the chunk records of ITS channels only (a few KB -- letting every workgroup
// re-derive all channels made the kernel L2-bound at 30-50 us), then walks all pixels of the plane.
// Merge (fp64, fixed order, no divisions in the loop):
//   mean = sum_k (x0_k n_k + s1_k) / hw ;  M2 = sum_k [ s2_k - 2 d_k s1_k + n_k d_k^2 ], d_k = mean - x0_k
template <typename T>
__global__ void __launch_bounds__(256)
instnorm_apply_small_kernel(T* __restrict__ out, int ld_out, const T* __restrict__ x, int ld_x,
                            const StatRec* __restrict__ part, int hw, int c, int chunk_px, int chunks,
                            const float* __restrict__ style_gb, int ld_gb) {
    constexpr int VEC = 16 / sizeof(T);
    constexpr int PARTS = 256 / VEC;          // threads cooperating per channel
    __shared__ double s_red[256];
    __shared__ float s_aff[2 * VEC];
    const int cvn = c / VEC;
    const int img = blockIdx.x / cvn, cv = blockIdx.x - img * cvn;
    const int tid = threadIdx.x;
    const int chl = tid % VEC, pt = tid / VEC;
    const int ch = cv * VEC + chl;
    const StatRec* pc = part + (int64_t)img * chunks * c + ch;
    // pass 1: mean
    double sum = 0.0;
    for (int k = pt; k < chunks; k += PARTS) {
        const StatRec r = pc[(int64_t)k * c];
        int npx = hw - k * chunk_px;
        if (npx > chunk_px) npx = chunk_px;
        sum += (double)r.x0 * (double)npx + (double)r.s1;
    }
    s_red[tid] = sum;
    __syncthreads();
    double tot = 0.0;
    for (int j = 0; j < PARTS; ++j) tot += s_red[j * VEC + chl];   // same fixed order in every thread
    const double mean = tot / (double)hw;
    __syncthreads();
    // pass 2: M2 around that mean
    double m2 = 0.0;
    for (int k = pt; k < chunks; k += PARTS) {
        const StatRec r = pc[(int64_t)k * c];
        int npx = hw - k * chunk_px;
        if (npx > chunk_px) npx = chunk_px;
        const double d = mean - (double)r.x0;
        m2 += (double)r.s2 - 2.0 * d * (double)r.s1 + (double)npx * d * d;
    }
    s_red[tid] = m2;
    __syncthreads();
    if (pt == 0) {
        double t2 = 0.0;
        for (int j = 0; j < PARTS; ++j) t2 += s_red[j * VEC + chl];
        double var = t2 / (double)hw;
        if (var < 0.0) var = 0.0;
        const float rstd = (float)(1.0 / sqrt(var + (double)IN_EPS));
        float gamma = 1.0f, beta = 0.0f;
        if (style_gb) {
            gamma = style_gb[(int64_t)img * ld_gb + ch];
            beta = style_gb[(int64_t)img * ld_gb + c + ch];
        }
        s_aff[chl] = gamma * rstd;
        s_aff[VEC + chl] = beta - gamma * rstd * (float)mean;
    }
    __syncthreads();
    float sc[VEC], sh[VEC];
#pragma unroll
    for (int k = 0; k < VEC; ++k) {
        sc[k] = s_aff[k];
        sh[k] = s_aff[VEC + k];
    }
    for (int p = tid; p < hw; p += 256) {
        const int64_t pix = (int64_t)img * hw + p;
        float f[VEC];
        unpack16<T>(ld128(x + pix * ld_x + cv * VEC), f);
#pragma unroll
        for (int k = 0; k < VEC; ++k) f[k] = f[k] * sc[k] + sh[k];
        st128(out + pix * ld_out + cv * VEC, pack16<T>(f));
    }
}

// AdaIN of a SMALL plane in ONE launch, statistics included (the 32x32-pixel trunk, 12 per frame).  One workgroup per
// (image, VPW 16-byte channel vectors): its slice of the plane is read ONCE into registers, mean and the centred second
// moment are reduced over the workgroup (fp32, fixed shuffle tree + wave order: deterministic, and a function of the image
// alone, so batch-independent), the affine is applied from the registers.  No records, no second pass over memory, and
// in-place operation is safe (a workgroup reads all it will write before writing).  VPW adjacent lanes hold adjacent
// vectors of one pixel: with VPW = 4 a workgroup reads and writes 64 contiguous bytes per pixel (VPW = 1, 16 bytes per
// pixel at a 1 KB stride, made every 128-byte line of the output the target of 8 partial writes from 8 workgroups:
// 16.7 us per launch for 8 MB of traffic).
// Round 3 measurement (profiles/r03_adain_ab.txt): folding the same AdaIN into the consumer conv (tile records from
// the producer, merge + LDS patch rewrite in conv_fullkw_kernel) costs 6.5 us in the producer and 17 us in the
// consumer, on 256 workgroups that own their CUs outright.
template <typename T, int PPT, int VPW, bool HAS_OTHER>
__global__ void __launch_bounds__(256)
instnorm_plane_kernel(T* __restrict__ out, int ld_out, const T* __restrict__ x, int ld_x,
                      const T* __restrict__ other, int ld_o, int hw, int c,
                      const float* __restrict__ style_gb, int ld_gb) {
    constexpr int VEC = 16 / sizeof(T);
    constexpr int HALVES = HAS_OTHER ? 2 : 1;   // second half: |x - other| (Fusion.forward, vtoonify.py:125) -> out[.., c + ch]
    constexpr int PPP = 256 / VPW;              // pixels per pass of the workgroup
    __shared__ float s_red[2][HALVES][4][VPW][VEC];
    const int cgn = c / (VEC * VPW);
    const int img = blockIdx.x / cgn, cg = blockIdx.x - img * cgn;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int vl = tid % VPW, pl = tid / VPW;   // vector of the group, pixel of the pass
    const int ch0 = (cg * VPW + vl) * VEC;      // first channel of this thread
    const T* xb = x + (int64_t)img * hw * ld_x + ch0;
    const T* ob = HAS_OTHER ? other + (int64_t)img * hw * ld_o + ch0 : nullptr;
    // the style affine first: its loads fly with the plane's (issued after the statistics they were a third dependent
    // memory round trip of a kernel that is nothing but a latency chain)
    const int ctot = c * HALVES;
    float sc[HALVES][VEC], sh[HALVES][VEC];
    if (style_gb) {   // one wave-uniform branch, unconditional loads inside
#pragma unroll
        for (int h = 0; h < HALVES; ++h)
#pragma unroll
            for (int k = 0; k < VEC; ++k) {
                sc[h][k] = style_gb[(int64_t)img * ld_gb + h * c + ch0 + k];            // gamma
                sh[h][k] = style_gb[(int64_t)img * ld_gb + ctot + h * c + ch0 + k];     // beta
            }
    } else {
#pragma unroll
        for (int h = 0; h < HALVES; ++h)
#pragma unroll
            for (int k = 0; k < VEC; ++k) sc[h][k] = 1.0f, sh[h][k] = 0.0f;
    }
    u128 raw[HALVES][PPT];
#pragma unroll
    for (int i = 0; i < PPT; ++i) {
        const int px = i * PPP + pl;
        const int pc = px < hw ? px : hw - 1;   // clamped: every load unconditional
        raw[0][i] = ld128(xb + (int64_t)pc * ld_x);
        if (HAS_OTHER) raw[HALVES - 1][i] = ld128(ob + (int64_t)pc * ld_o);
    }
    // the VEC values of pixel i in each half, from the registers
    auto values = [&](int i, float (&f)[HALVES][VEC]) {
        unpack16<T>(raw[0][i], f[0]);
        if (HAS_OTHER) {
            float g[VEC];
            unpack16<T>(raw[HALVES - 1][i], g);
#pragma unroll
            for (int k = 0; k < VEC; ++k) f[HALVES - 1][k] = fabsf(f[0][k] - g[k]);
        }
    };
    // sum over the workgroup's pixels: lanes VPW apart hold the same channels
    auto reduce = [&](float (&acc)[HALVES][VEC], int slot) {
#pragma unroll
        for (int h = 0; h < HALVES; ++h)
#pragma unroll
            for (int k = 0; k < VEC; ++k) {
                float v = acc[h][k];
#pragma unroll
                for (int off = 32; off >= VPW; off >>= 1) v += __shfl_xor(v, off, 64);
                if (lane < VPW) s_red[slot][h][wave][vl][k] = v;
            }
        __syncthreads();
#pragma unroll
        for (int h = 0; h < HALVES; ++h)
#pragma unroll
            for (int k = 0; k < VEC; ++k)
                acc[h][k] = ((s_red[slot][h][0][vl][k] + s_red[slot][h][1][vl][k]) + s_red[slot][h][2][vl][k]) +
                            s_red[slot][h][3][vl][k];
    };
    float acc[HALVES][VEC], mean[HALVES][VEC];
#pragma unroll
    for (int h = 0; h < HALVES; ++h)
#pragma unroll
        for (int k = 0; k < VEC; ++k) acc[h][k] = 0.0f;
#pragma unroll
    for (int i = 0; i < PPT; ++i) {
        float f[HALVES][VEC];
        values(i, f);
        const float live = (i * PPP + pl < hw) ? 1.0f : 0.0f;
#pragma unroll
        for (int h = 0; h < HALVES; ++h)
#pragma unroll
            for (int k = 0; k < VEC; ++k) acc[h][k] += f[h][k] * live;
    }
    reduce(acc, 0);
    const float inv = 1.0f / (float)hw;
#pragma unroll
    for (int h = 0; h < HALVES; ++h)
#pragma unroll
        for (int k = 0; k < VEC; ++k) {
            mean[h][k] = acc[h][k] * inv;
            acc[h][k] = 0.0f;
        }
#pragma unroll
    for (int i = 0; i < PPT; ++i) {
        float f[HALVES][VEC];
        values(i, f);
        const float live = (i * PPP + pl < hw) ? 1.0f : 0.0f;
#pragma unroll
        for (int h = 0; h < HALVES; ++h)
#pragma unroll
            for (int k = 0; k < VEC; ++k) {
                const float dv = (f[h][k] - mean[h][k]) * live;
                acc[h][k] += dv * dv;
            }
    }
    reduce(acc, 1);
#pragma unroll
    for (int h = 0; h < HALVES; ++h)
#pragma unroll
        for (int k = 0; k < VEC; ++k) {
            const float var = acc[h][k] * inv;   // biased, as F.instance_norm
            const float rstd = 1.0f / sqrtf(var + IN_EPS);
            const float gamma = sc[h][k];
            sc[h][k] = gamma * rstd;
            sh[h][k] = sh[h][k] - gamma * rstd * mean[h][k];
        }
    T* outb = out + (int64_t)img * hw * ld_out + ch0;
#pragma unroll
    for (int i = 0; i < PPT; ++i) {
        const int px = i * PPP + pl;
        if (px >= hw) break;
        float f[HALVES][VEC];
        values(i, f);
#pragma unroll
        for (int h = 0; h < HALVES; ++h) {
#pragma unroll
            for (int k = 0; k < VEC; ++k) f[h][k] = fmaf(f[h][k], sc[h][k], sh[h][k]);
            st128(outb + (int64_t)px * ld_out + h * c, pack16<T>(f[h]));
        }
    }
}

template <typename T>
__global__ void __launch_bounds__(256)
affine_apply_kernel(T* __restrict__ out, int ld_out, const T* __restrict__ x, int ld_x,
                    const T* __restrict__ other, int ld_o, const float* __restrict__ scale,
                    const float* __restrict__ shift, int n, int hw, int c) {
    constexpr int VEC = 16 / sizeof(T);
    const int cvn = c / VEC;
    const int halves = other ? 2 : 1;
    const int ctot = c * halves;
    const int64_t total = (int64_t)n * hw * cvn * halves;
    const int64_t stride = (int64_t)gridDim.x * 256;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += stride) {
        const int cvh = (int)(i % (cvn * halves));
        const int64_t pix = i / (cvn * halves);  // global pixel index n*hw + p
        const int half = cvh / cvn, cv = cvh - half * cvn;
        const int img = (int)(pix / hw);
        float f[VEC];
        unpack16<T>(ld128(x + pix * ld_x + cv * VEC), f);
        if (half == 1) {
            float g[VEC];
            unpack16<T>(ld128(other + pix * ld_o + cv * VEC), g);
#pragma unroll
            for (int k = 0; k < VEC; ++k) f[k] = fabsf(f[k] - g[k]);
        }
        const int so = img * ctot + half * c + cv * VEC;
#pragma unroll
        for (int k = 0; k < VEC; ++k) f[k] = f[k] * scale[so + k] + shift[so + k];
        st128(out + pix * ld_out + half * c + cv * VEC, pack16<T>(f));
    }
}

template <typename T>
__global__ void __launch_bounds__(256)
fusion_pack_kernel(T* __restrict__ out, int ld_out, const T* __restrict__ f_e, int ld_e,
                   const float* __restrict__ mask, const float* __restrict__ skip, int n, int hw,
                   int c) {
    constexpr int VEC = 16 / sizeof(T);
    const int cvn = c / VEC;
    const int hdr = ld_out - c;   // header channels [skip(3) | zeros]: 8, or 64 so that the consumer's
                                  // channel count is a multiple of the direct-to-LDS K-step
    const int hv = hdr / VEC;
    const int per_px = cvn + hv;
    const int64_t total = (int64_t)n * hw * per_px;
    const int64_t stride = (int64_t)gridDim.x * 256;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += stride) {
        const int v = (int)(i % per_px);
        const int64_t pix = i / per_px;
        T* o = out + pix * ld_out;
        if (v < hv) {
            const int img = (int)(pix / hw);
            const int p = (int)(pix - (int64_t)img * hw);
            float f[VEC];
#pragma unroll
            for (int k = 0; k < VEC; ++k) {
                const int ch = v * VEC + k;
                f[k] = (ch < 3) ? skip[((int64_t)img * 3 + ch) * hw + p] : 0.0f;
            }
            st128(o + v * VEC, pack16<T>(f));
        } else {
            const int cv = v - hv;
            const float m = mask ? mask[pix] : 1.0f;
            float f[VEC];
            unpack16<T>(ld128(f_e + pix * ld_e + cv * VEC), f);
#pragma unroll
            for (int k = 0; k < VEC; ++k) f[k] *= m;
            st128(o + hdr + cv * VEC, pack16<T>(f));
        }
    }
}

template <typename TI, typename TO>
__global__ void __launch_bounds__(256)
nchw_to_nhwc_kernel(TO* __restrict__ out, int ld_out, const TI* __restrict__ in, int n, int c,
                    int hw, int cpad) {
    // one thread per (pixel, group of 8 channels); lanes walk pixels so the strided
    // plane reads are coalesced
    const int groups = cpad / 8;
    const int64_t total = (int64_t)n * groups * hw;
    const int64_t stride = (int64_t)gridDim.x * 256;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += stride) {
        const int p = (int)(i % hw);
        const int64_t t = i / hw;
        const int g = (int)(t % groups);
        const int img = (int)(t / groups);
        TO* o = out + ((int64_t)img * hw + p) * ld_out + g * 8;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int ch = g * 8 + k;
            const float v = ch < c ? to_f32(in[((int64_t)img * c + ch) * hw + p]) : 0.0f;
            o[k] = from_f32<TO>(v);
        }
    }
}

// Few channels (the 22-channel frame at the model boundary, vtoonify.py:226): one thread per PIXEL.  A lane reads its pixel
// from every plane (adjacent lanes, adjacent pixels: 256-byte segments per plane) and writes the pixel's whole padded row with
// 16-byte stores, so a wavefront writes 64 complete, consecutive pixel rows -- the (pixel, 8-channel group) form above has three
// far-apart threads write the three 16-byte thirds of every 64-byte row (34.6 us for the 4 x 22 x 256 x 256 frame batch of
// bench.py, 1.15 TB/s: 1.4 % of the step in a kernel nobody had looked at; profiles/r05_rocprofv3_kernel_stats_lanes1.txt).
template <typename TI, typename TO, int CPAD>
__global__ void __launch_bounds__(256)
nchw_to_nhwc_pixel_kernel(TO* __restrict__ out, int ld_out, const TI* __restrict__ in, int n, int c, int hw) {
    constexpr int VEC = 16 / (int)sizeof(TO);
    const int64_t total = (int64_t)n * hw;
    const int64_t stride = (int64_t)gridDim.x * 256;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += stride) {
        const int img = (int)(i / hw);
        const int p = (int)(i - (int64_t)img * hw);
        const TI* src = in + (int64_t)img * c * hw + p;
        float f[CPAD];
#pragma unroll
        for (int ch = 0; ch < CPAD; ++ch) f[ch] = ch < c ? to_f32(src[(int64_t)ch * hw]) : 0.0f;   // (c is uniform: no divergence)
        TO* o = out + i * ld_out;
#pragma unroll
        for (int v = 0; v < CPAD / VEC; ++v) st128(o + v * VEC, pack16<TO>(f + v * VEC));
    }
}

template <typename TI, typename TO>
__global__ void __launch_bounds__(256)
nhwc_to_nchw_kernel(TO* __restrict__ out, const TI* __restrict__ in, int ld_in, int n, int c,
                    int hw) {
    const int64_t total = (int64_t)n * c * hw;
    const int64_t stride = (int64_t)gridDim.x * 256;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += stride) {
        const int p = (int)(i % hw);
        const int64_t t = i / hw;
        const int ch = (int)(t % c);
        const int img = (int)(t / c);
        out[i] = from_f32<TO>(to_f32(in[((int64_t)img * hw + p) * ld_in + ch]));
    }
}

// Wide tensors (the stand-alone convolutions of conv2d_gradfix, RAFT's correlation features): a 64-pixel x 64-channel tile per
// workgroup goes through LDS, so that BOTH sides of the copy move whole segments -- a wavefront reads 64 consecutive pixels of a
// plane (128 / 256 bytes) and 8 lanes write the 64 channels of one pixel (128 / 256 bytes).  The one-thread-per-(pixel, 8 channels)
// forms above and below leave one side at 16 bytes per 256-byte row (nchw_to_nhwc_kernel's stores) or at one element per row
// (nhwc_to_nchw_kernel's loads): profiles/r05_grad_bench.json, where a 128-channel 4 x 256^2 conv of the operator surface spent
// more time in its two layout changes than in its contraction.  Exact copies (or the same round-to-nearest-even as those forms).
constexpr int TR_P = 64, TR_C = 64;

template <typename TI, typename TO>
__global__ void __launch_bounds__(256)
nchw_to_nhwc_tile_kernel(TO* __restrict__ out, int ld_out, const TI* __restrict__ in, int n, int c, int hw, int cpad) {
    constexpr int VEC = 16 / (int)sizeof(TO);
    __shared__ float tile[TR_C][TR_P + 1];
    const int ptiles = (hw + TR_P - 1) / TR_P, ctiles = (cpad + TR_C - 1) / TR_C;
    const int64_t total = (int64_t)n * ptiles * ctiles;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, g = threadIdx.x & 7;
    for (int64_t t = blockIdx.x; t < total; t += gridDim.x) {
        const int ct = (int)(t % ctiles);
        const int64_t r = t / ctiles;
        const int pt = (int)(r % ptiles), img = (int)(r / ptiles);
        const int p0 = pt * TR_P, c0 = ct * TR_C;
        for (int k = wave; k < TR_C; k += 4) {   // planes -> LDS (channels >= c read as zero: the padding of the rows)
            const int ch = c0 + k, p = p0 + lane;
            tile[k][lane] = (ch < c && p < hw) ? to_f32(in[((int64_t)img * c + ch) * hw + p]) : 0.0f;
        }
        __syncthreads();
        for (int q = threadIdx.x >> 3; q < TR_P; q += 32) {   // LDS -> pixel rows
            const int p = p0 + q, ch = c0 + g * 8;
            if (p < hw && ch < cpad) {
                float f[8];
#pragma unroll
                for (int k = 0; k < 8; ++k) f[k] = tile[g * 8 + k][q];
                TO* o = out + ((int64_t)img * hw + p) * ld_out + ch;
#pragma unroll
                for (int v = 0; v < 8 / VEC; ++v) st128(o + v * VEC, pack16<TO>(f + v * VEC));
            }
        }
        __syncthreads();
    }
}

template <typename TI, typename TO>
__global__ void __launch_bounds__(256)
nhwc_to_nchw_tile_kernel(TO* __restrict__ out, const TI* __restrict__ in, int ld_in, int n, int c, int hw) {
    constexpr int VEC = 16 / (int)sizeof(TI);
    __shared__ float tile[TR_C][TR_P + 1];
    const int ptiles = (hw + TR_P - 1) / TR_P, ctiles = (c + TR_C - 1) / TR_C;
    const int64_t total = (int64_t)n * ptiles * ctiles;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, g = threadIdx.x & 7;
    for (int64_t t = blockIdx.x; t < total; t += gridDim.x) {
        const int ct = (int)(t % ctiles);
        const int64_t r = t / ctiles;
        const int pt = (int)(r % ptiles), img = (int)(r / ptiles);
        const int p0 = pt * TR_P, c0 = ct * TR_C;
        for (int q = threadIdx.x >> 3; q < TR_P; q += 32) {   // pixel rows -> LDS (c % 8 == 0: whole 8-channel groups)
            const int p = p0 + q, ch = c0 + g * 8;
            if (p < hw && ch < c) {
                float f[8];
                const TI* src = in + ((int64_t)img * hw + p) * ld_in + ch;
#pragma unroll
                for (int v = 0; v < 8 / VEC; ++v) unpack16<TI>(ld128(src + v * VEC), f + v * VEC);
#pragma unroll
                for (int k = 0; k < 8; ++k) tile[g * 8 + k][q] = f[k];
            }
        }
        __syncthreads();
        for (int k = wave; k < TR_C; k += 4) {   // LDS -> planes
            const int ch = c0 + k, p = p0 + lane;
            if (ch < c && p < hw) out[((int64_t)img * c + ch) * hw + p] = from_f32<TO>(tile[k][lane]);
        }
        __syncthreads();
    }
}

inline unsigned tile_grid(int n, int hw, int cc) {
    int64_t b = (int64_t)n * ((hw + TR_P - 1) / TR_P) * ((cc + TR_C - 1) / TR_C);
    if (b > 16384) b = 16384;
    return (unsigned)(b < 1 ? 1 : b);
}

// ---------------------------------------------------------------------------------
// pSp encoder glue (model/encoder/encoders/helpers.py:53-119, psp_encoders.py:71-88)
// ---------------------------------------------------------------------------------
// AdaptiveAvgPool2d(1) from the chunk records of instnorm_partial_kernel: mean[n][c].  One workgroup
// per (image, 16 channels): 16 chunk lanes per channel each sum every 16th chunk in index order, the
// 16 partial sums are added in lane order (fixed order -> deterministic).  (One thread per (n, c)
// walking all chunks serially took 24-85 us on the 64x64 .. 128x128-pixel maps of BiSeNet / pSp.)
__global__ void __launch_bounds__(256)
channel_mean_kernel(float* __restrict__ mean, const StatRec* __restrict__ part, int n, int hw, int c,
                    int chunk_px, int chunks) {
    __shared__ double red[16][17];
    const int cgroups = (c + 15) / 16;
    const int cg = blockIdx.x % cgroups, img = blockIdx.x / cgroups;
    const int cl = threadIdx.x & 15, kl = threadIdx.x >> 4;
    const int ch = cg * 16 + cl;
    double sum = 0.0;
    if (ch < c) {
        for (int k = kl; k < chunks; k += 16) {
            const StatRec r = part[((int64_t)img * chunks + k) * c + ch];
            int npx = hw - k * chunk_px;
            if (npx > chunk_px) npx = chunk_px;
            sum += (double)r.x0 * npx + (double)r.s1;
        }
    }
    red[kl][cl] = sum;
    __syncthreads();
    if (kl == 0 && ch < c) {
        double t = 0.0;
        for (int j = 0; j < 16; ++j) t += red[j][cl];
        mean[(int64_t)img * c + ch] = (float)(t / (double)hw);
    }
}

// SE gate + residual of bottleneck_IR_SE:  out[n,oy,ox,c] = res[n,oy,ox,c] * gate[n][c]
//                                                           + sc[n, oy*sc_stride, ox*sc_stride, c]
// (sc_stride > 1 is MaxPool2d(1, stride) of the block input, helpers.py:100-101).
template <typename T>
__global__ void __launch_bounds__(256)
se_apply_kernel(T* __restrict__ out, const T* __restrict__ res, const float* __restrict__ gate,
                const T* __restrict__ sc, int n, int oh, int ow, int c, int sc_h, int sc_w, int sc_stride) {
    constexpr int VEC = 16 / sizeof(T);
    const int cvn = c / VEC;
    const int64_t total = (int64_t)n * oh * ow * cvn;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int cv = (int)(i % cvn);
        const int64_t pix = i / cvn;
        const int ox = (int)(pix % ow);
        const int64_t t = pix / ow;
        const int oy = (int)(t % oh), img = (int)(t / oh);
        float f[VEC], g[VEC];
        unpack16<T>(ld128(res + pix * c + cv * VEC), f);
        const int64_t spix = ((int64_t)img * sc_h + (int64_t)oy * sc_stride) * sc_w + (int64_t)ox * sc_stride;
        unpack16<T>(ld128(sc + spix * c + cv * VEC), g);
        const float* gt = gate + (int64_t)img * c + cv * VEC;
#pragma unroll
        for (int k = 0; k < VEC; ++k) f[k] = f[k] * gt[k] + g[k];
        st128(out + pix * c + cv * VEC, pack16<T>(f));
    }
}

// F.interpolate(x, size=(H,W), mode='bilinear', align_corners=True) + y   (psp_encoders.py:71-88)
template <typename T>
__global__ void __launch_bounds__(256)
upsample_add_kernel(T* __restrict__ out, const T* __restrict__ x, const T* __restrict__ y, int n, int h,
                    int w, int H, int W, int c) {
    constexpr int VEC = 16 / sizeof(T);
    const int cvn = c / VEC;
    const int64_t total = (int64_t)n * H * W * cvn;
    const float sy = H > 1 ? (float)(h - 1) / (float)(H - 1) : 0.0f;
    const float sx = W > 1 ? (float)(w - 1) / (float)(W - 1) : 0.0f;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int cv = (int)(i % cvn);
        const int64_t pix = i / cvn;
        const int X = (int)(pix % W);
        const int64_t t = pix / W;
        const int Y = (int)(t % H), img = (int)(t / H);
        const float fy = sy * (float)Y, fx = sx * (float)X;
        int y0 = (int)fy, x0 = (int)fx;
        if (y0 > h - 1) y0 = h - 1;
        if (x0 > w - 1) x0 = w - 1;
        const int y1 = y0 + 1 < h ? y0 + 1 : h - 1, x1 = x0 + 1 < w ? x0 + 1 : w - 1;
        const float ly = fy - (float)y0, lx = fx - (float)x0;
        const T* xb = x + (int64_t)img * h * w * c + cv * VEC;
        float a[VEC], b[VEC], cc[VEC], d[VEC], r[VEC];
        unpack16<T>(ld128(xb + ((int64_t)y0 * w + x0) * c), a);
        unpack16<T>(ld128(xb + ((int64_t)y0 * w + x1) * c), b);
        unpack16<T>(ld128(xb + ((int64_t)y1 * w + x0) * c), cc);
        unpack16<T>(ld128(xb + ((int64_t)y1 * w + x1) * c), d);
        unpack16<T>(ld128(y + pix * c + cv * VEC), r);
#pragma unroll
        for (int k = 0; k < VEC; ++k) {
            const float top = a[k] + (b[k] - a[k]) * lx, bot = cc[k] + (d[k] - cc[k]) * lx;
            r[k] += top + (bot - top) * ly;
        }
        st128(out + pix * c + cv * VEC, pack16<T>(r));
    }
}

inline unsigned grid_for(int64_t total) {
    int64_t b = (total + 255) / 256;
    if (b > 8192) b = 8192;
    if (b < 1) b = 1;
    return (unsigned)b;
}

}  // namespace

extern "C" int64_t vt_instnorm_ws_bytes(int n, int hw, int c_total) {
    if (n <= 0 || hw <= 0 || c_total <= 0) return 0;
    const int cpx = stat_chunk_pixels(hw);
    const int chunks = (hw + cpx - 1) / cpx;
    return (int64_t)n * chunks * c_total * (int64_t)sizeof(StatRec);
}

extern "C" int vt_instnorm_stats(float* scale, float* shift, const void* x, int ld_x,
                                 const void* absdiff_other, int ld_other, int n, int hw, int c,
                                 const float* style_gb, int ld_gb, void* partials, int dtype,
                                 vt_stream stream) {
    VT_REQUIRE(scale && shift && x && partials, "vt_instnorm_stats: null tensor");
    VT_REQUIRE(n > 0 && hw > 0 && c > 0 && c % 16 == 0, "vt_instnorm_stats: c must be a positive multiple of 16");
    VT_REQUIRE(dtype == VT_F32 || dtype == VT_BF16, "vt_instnorm_stats: dtype");
    const int cpx = stat_chunk_pixels(hw);
    const int chunks = (hw + cpx - 1) / cpx;
    const int ctot = absdiff_other ? 2 * c : c;
    dim3 grid((unsigned)(n * chunks)), block(256);
#define VT_IN_LAUNCH(TT, HO)                                                                        \
    {                                                                                               \
        auto k = instnorm_partial_kernel<TT, HO>;                                                   \
        VT_LAUNCH(k, grid, block, stream, (StatRec*)partials, (const TT*)x, ld_x,                   \
                  (const TT*)absdiff_other, ld_other, hw, c, cpx, chunks);                          \
    }
    if (dtype == VT_F32) {
        if (absdiff_other) VT_IN_LAUNCH(float, true) else VT_IN_LAUNCH(float, false)
    } else {
        if (absdiff_other) VT_IN_LAUNCH(bf16_t, true) else VT_IN_LAUNCH(bf16_t, false)
    }
#undef VT_IN_LAUNCH
    int rc = vt_check_launch("vt_instnorm_stats(partial)");
    if (rc) return rc;
    VT_LAUNCH(instnorm_finalize_kernel, dim3((unsigned)(n * (ctot / 16))), dim3(256), stream,
              scale, shift, (const StatRec*)partials, n, hw, ctot, cpx, chunks, style_gb, ld_gb);
    return vt_check_launch("vt_instnorm_stats(finalize)");
}

extern "C" int vt_affine_apply(void* out, int ld_out, const void* x, int ld_x,
                               const void* absdiff_other, int ld_other, const float* scale,
                               const float* shift, int n, int hw, int c, int dtype,
                               vt_stream stream) {
    VT_REQUIRE(out && x && scale && shift, "vt_affine_apply: null tensor");
    VT_REQUIRE(n > 0 && hw > 0 && c > 0 && c % 8 == 0, "vt_affine_apply: c must be a positive multiple of 8");
    const int halves = absdiff_other ? 2 : 1;
    if (dtype == VT_F32) {
        const int64_t total = (int64_t)n * hw * (c / 4) * halves;
        auto k = affine_apply_kernel<float>;
        VT_LAUNCH(k, dim3(grid_for(total)), dim3(256), stream, (float*)out, ld_out, (const float*)x, ld_x,
                  (const float*)absdiff_other, ld_other, scale, shift, n, hw, c);
    } else if (dtype == VT_BF16) {
        const int64_t total = (int64_t)n * hw * (c / 8) * halves;
        auto k = affine_apply_kernel<bf16_t>;
        VT_LAUNCH(k, dim3(grid_for(total)), dim3(256), stream, (bf16_t*)out, ld_out, (const bf16_t*)x, ld_x,
                  (const bf16_t*)absdiff_other, ld_other, scale, shift, n, hw, c);
    } else {
        vt_set_error("vt_affine_apply: dtype");
        return VT_ERR_UNSUPPORTED;
    }
    return vt_check_launch("vt_affine_apply");
}

extern "C" int vt_fusion_pack(void* out, int ld_out, const void* f_e, int ld_e, const float* mask,
                              const float* skip, int n, int hw, int c, int dtype, vt_stream stream) {
    VT_REQUIRE(out && f_e && skip, "vt_fusion_pack: null tensor");
    VT_REQUIRE(n > 0 && hw > 0 && c > 0 && c % 8 == 0 && ld_out >= c + 8 && (ld_out - c) % 8 == 0,
               "vt_fusion_pack: bad sizes (ld_out = header + c, header a multiple of 8, >= 8)");
    if (dtype == VT_F32) {
        const int64_t total = (int64_t)n * hw * (ld_out / 4);
        auto k = fusion_pack_kernel<float>;
        VT_LAUNCH(k, dim3(grid_for(total)), dim3(256), stream, (float*)out, ld_out, (const float*)f_e, ld_e,
                  mask, skip, n, hw, c);
    } else if (dtype == VT_BF16) {
        const int64_t total = (int64_t)n * hw * (ld_out / 8);
        auto k = fusion_pack_kernel<bf16_t>;
        VT_LAUNCH(k, dim3(grid_for(total)), dim3(256), stream, (bf16_t*)out, ld_out, (const bf16_t*)f_e, ld_e,
                  mask, skip, n, hw, c);
    } else {
        vt_set_error("vt_fusion_pack: dtype");
        return VT_ERR_UNSUPPORTED;
    }
    return vt_check_launch("vt_fusion_pack");
}

template <typename TI>
static int nchw_to_nhwc_out(void* out, int ld_out, const TI* in, int n, int c, int hw, int out_dtype,
                            vt_stream stream) {
    const int cpad = (c + 7) / 8 * 8;
    const int64_t total = (int64_t)n * (cpad / 8) * hw;
    // one thread per pixel for the narrow tensors of the model boundary (16-byte aligned pixel rows)
    const int osz = out_dtype == VT_F32 ? 4 : 2;
    if ((cpad == 8 || cpad == 16 || cpad == 24 || cpad == 32) && (out_dtype == VT_F32 || out_dtype == VT_BF16) &&
        ((int64_t)ld_out * osz) % 16 == 0 && (uintptr_t)out % 16 == 0) {
        const unsigned grid = grid_for((int64_t)n * hw);
#define VT_PIX(TO_, CP_)                                                                                        \
    {                                                                                                           \
        auto k = nchw_to_nhwc_pixel_kernel<TI, TO_, CP_>;                                                       \
        VT_LAUNCH(k, dim3(grid), dim3(256), stream, (TO_*)out, ld_out, in, n, c, hw);                          \
    }
#define VT_PIX_C(TO_) \
    if (cpad == 8) VT_PIX(TO_, 8) else if (cpad == 16) VT_PIX(TO_, 16) else if (cpad == 24) VT_PIX(TO_, 24) else VT_PIX(TO_, 32)
        if (out_dtype == VT_F32) { VT_PIX_C(float) } else { VT_PIX_C(bf16_t) }
#undef VT_PIX_C
#undef VT_PIX
        return vt_check_launch("vt_nchw_to_nhwc");
    }
    if (cpad > 32 && (out_dtype == VT_F32 || out_dtype == VT_BF16) && ((int64_t)ld_out * osz) % 16 == 0 &&
        (uintptr_t)out % 16 == 0) {   // wide tensors: tiles through LDS
        const unsigned grid = tile_grid(n, hw, cpad);
        if (out_dtype == VT_F32) {
            auto k = nchw_to_nhwc_tile_kernel<TI, float>;
            VT_LAUNCH(k, dim3(grid), dim3(256), stream, (float*)out, ld_out, in, n, c, hw, cpad);
        } else {
            auto k = nchw_to_nhwc_tile_kernel<TI, bf16_t>;
            VT_LAUNCH(k, dim3(grid), dim3(256), stream, (bf16_t*)out, ld_out, in, n, c, hw, cpad);
        }
        return vt_check_launch("vt_nchw_to_nhwc");
    }
    if (out_dtype == VT_F32) {
        auto k = nchw_to_nhwc_kernel<TI, float>;
        VT_LAUNCH(k, dim3(grid_for(total)), dim3(256), stream, (float*)out, ld_out, in, n, c, hw, cpad);
    } else if (out_dtype == VT_BF16) {
        auto k = nchw_to_nhwc_kernel<TI, bf16_t>;
        VT_LAUNCH(k, dim3(grid_for(total)), dim3(256), stream, (bf16_t*)out, ld_out, in, n, c, hw, cpad);
    } else {
        vt_set_error("vt_nchw_to_nhwc: out dtype");
        return VT_ERR_UNSUPPORTED;
    }
    return vt_check_launch("vt_nchw_to_nhwc");
}

extern "C" int vt_nchw_to_nhwc(void* out, int ld_out, const void* in, int n, int c, int hw,
                               int in_dtype, int out_dtype, vt_stream stream) {
    VT_REQUIRE(out && in && n > 0 && c > 0 && hw > 0, "vt_nchw_to_nhwc: bad arguments");
    VT_REQUIRE(ld_out >= (c + 7) / 8 * 8, "vt_nchw_to_nhwc: ld_out must cover c rounded up to 8");
    if (in_dtype == VT_F32) return nchw_to_nhwc_out<float>(out, ld_out, (const float*)in, n, c, hw, out_dtype, stream);
    if (in_dtype == VT_BF16) return nchw_to_nhwc_out<bf16_t>(out, ld_out, (const bf16_t*)in, n, c, hw, out_dtype, stream);
    if (in_dtype == VT_F16) return nchw_to_nhwc_out<f16_t>(out, ld_out, (const f16_t*)in, n, c, hw, out_dtype, stream);
    vt_set_error("vt_nchw_to_nhwc: in dtype");
    return VT_ERR_UNSUPPORTED;
}

template <typename TI>
static int nhwc_to_nchw_out(void* out, const TI* in, int ld_in, int n, int c, int hw, int out_dtype,
                            vt_stream stream) {
    const int64_t total = (int64_t)n * c * hw;
    if (c >= 32 && c % 8 == 0 && (out_dtype == VT_F32 || out_dtype == VT_BF16) &&
        ((int64_t)ld_in * (int64_t)sizeof(TI)) % 16 == 0 && (uintptr_t)in % 16 == 0) {   // wide tensors: tiles through LDS
        const unsigned grid = tile_grid(n, hw, c);
        if (out_dtype == VT_F32) {
            auto k = nhwc_to_nchw_tile_kernel<TI, float>;
            VT_LAUNCH(k, dim3(grid), dim3(256), stream, (float*)out, in, ld_in, n, c, hw);
        } else {
            auto k = nhwc_to_nchw_tile_kernel<TI, bf16_t>;
            VT_LAUNCH(k, dim3(grid), dim3(256), stream, (bf16_t*)out, in, ld_in, n, c, hw);
        }
        return vt_check_launch("vt_nhwc_to_nchw");
    }
    if (out_dtype == VT_F32) {
        auto k = nhwc_to_nchw_kernel<TI, float>;
        VT_LAUNCH(k, dim3(grid_for(total)), dim3(256), stream, (float*)out, in, ld_in, n, c, hw);
    } else if (out_dtype == VT_BF16) {
        auto k = nhwc_to_nchw_kernel<TI, bf16_t>;
        VT_LAUNCH(k, dim3(grid_for(total)), dim3(256), stream, (bf16_t*)out, in, ld_in, n, c, hw);
    } else if (out_dtype == VT_F16) {
        auto k = nhwc_to_nchw_kernel<TI, f16_t>;
        VT_LAUNCH(k, dim3(grid_for(total)), dim3(256), stream, (f16_t*)out, in, ld_in, n, c, hw);
    } else {
        vt_set_error("vt_nhwc_to_nchw: out dtype");
        return VT_ERR_UNSUPPORTED;
    }
    return vt_check_launch("vt_nhwc_to_nchw");
}

extern "C" int vt_nhwc_to_nchw(void* out, const void* in, int ld_in, int n, int c, int hw,
                               int in_dtype, int out_dtype, vt_stream stream) {
    VT_REQUIRE(out && in && n > 0 && c > 0 && hw > 0 && ld_in >= c, "vt_nhwc_to_nchw: bad arguments");
    if (in_dtype == VT_F32) return nhwc_to_nchw_out<float>(out, (const float*)in, ld_in, n, c, hw, out_dtype, stream);
    if (in_dtype == VT_BF16) return nhwc_to_nchw_out<bf16_t>(out, (const bf16_t*)in, ld_in, n, c, hw, out_dtype, stream);
    vt_set_error("vt_nhwc_to_nchw: in dtype");
    return VT_ERR_UNSUPPORTED;
}

extern "C" int vt_channel_mean(float* mean, const void* x, int ld_x, int n, int hw, int c, void* partials,
                               int dtype, vt_stream stream) {
    VT_REQUIRE(mean && x && partials, "vt_channel_mean: null tensor");
    VT_REQUIRE(n > 0 && hw > 0 && c > 0 && c % 8 == 0, "vt_channel_mean: c must be a positive multiple of 8");
    VT_REQUIRE(dtype == VT_F32 || dtype == VT_BF16, "vt_channel_mean: dtype");
    const int cpx = stat_chunk_pixels(hw);
    const int chunks = (hw + cpx - 1) / cpx;
    dim3 grid((unsigned)(n * chunks)), block(256);
    if (dtype == VT_F32) {
        auto k = instnorm_partial_kernel<float, false>;
        VT_LAUNCH(k, grid, block, stream, (StatRec*)partials, (const float*)x, ld_x, (const float*)nullptr, 0, hw, c, cpx, chunks);
    } else {
        auto k = instnorm_partial_kernel<bf16_t, false>;
        VT_LAUNCH(k, grid, block, stream, (StatRec*)partials, (const bf16_t*)x, ld_x, (const bf16_t*)nullptr, 0, hw, c, cpx, chunks);
    }
    int rc = vt_check_launch("vt_channel_mean(partial)");
    if (rc) return rc;
    VT_LAUNCH(channel_mean_kernel, dim3((unsigned)(n * ((c + 15) / 16))), dim3(256), stream, mean,
              (const StatRec*)partials, n, hw, c, cpx, chunks);
    return vt_check_launch("vt_channel_mean");
}

extern "C" int vt_se_apply(void* out, const void* res, const float* gate, const void* shortcut, int n,
                           int oh, int ow, int c, int sc_h, int sc_w, int sc_stride, int dtype,
                           vt_stream stream) {
    VT_REQUIRE(out && res && gate && shortcut, "vt_se_apply: null tensor");
    VT_REQUIRE(n > 0 && oh > 0 && ow > 0 && c > 0 && c % 8 == 0 && sc_stride >= 1, "vt_se_apply: bad sizes");
    VT_REQUIRE((oh - 1) * sc_stride < sc_h && (ow - 1) * sc_stride < sc_w, "vt_se_apply: shortcut too small");
    if (dtype == VT_F32) {
        auto k = se_apply_kernel<float>;
        VT_LAUNCH(k, dim3(grid_for((int64_t)n * oh * ow * (c / 4))), dim3(256), stream, (float*)out, (const float*)res,
                  gate, (const float*)shortcut, n, oh, ow, c, sc_h, sc_w, sc_stride);
    } else if (dtype == VT_BF16) {
        auto k = se_apply_kernel<bf16_t>;
        VT_LAUNCH(k, dim3(grid_for((int64_t)n * oh * ow * (c / 8))), dim3(256), stream, (bf16_t*)out,
                  (const bf16_t*)res, gate, (const bf16_t*)shortcut, n, oh, ow, c, sc_h, sc_w, sc_stride);
    } else {
        vt_set_error("vt_se_apply: dtype");
        return VT_ERR_UNSUPPORTED;
    }
    return vt_check_launch("vt_se_apply");
}

extern "C" int vt_upsample_bilinear_add(void* out, const void* x, const void* y, int n, int h, int w, int H,
                                        int W, int c, int dtype, vt_stream stream) {
    VT_REQUIRE(out && x && y, "vt_upsample_bilinear_add: null tensor");
    VT_REQUIRE(n > 0 && h > 0 && w > 0 && H > 0 && W > 0 && c > 0 && c % 8 == 0, "vt_upsample_bilinear_add: bad sizes");
    if (dtype == VT_F32) {
        auto k = upsample_add_kernel<float>;
        VT_LAUNCH(k, dim3(grid_for((int64_t)n * H * W * (c / 4))), dim3(256), stream, (float*)out, (const float*)x,
                  (const float*)y, n, h, w, H, W, c);
    } else if (dtype == VT_BF16) {
        auto k = upsample_add_kernel<bf16_t>;
        VT_LAUNCH(k, dim3(grid_for((int64_t)n * H * W * (c / 8))), dim3(256), stream, (bf16_t*)out, (const bf16_t*)x,
                  (const bf16_t*)y, n, h, w, H, W, c);
    } else {
        vt_set_error("vt_upsample_bilinear_add: dtype");
        return VT_ERR_UNSUPPORTED;
    }
    return vt_check_launch("vt_upsample_bilinear_add");
}

int vt_internal_instnorm_partial(void* partials, const void* x, int ld_x, int n, int hw, int c, int dtype,
                                 vt_stream stream) {
    const int cpx = stat_chunk_pixels(hw);
    const int chunks = (hw + cpx - 1) / cpx;
    dim3 grid((unsigned)(n * chunks)), block(256);
    if (dtype == VT_F32) {
        auto k = instnorm_partial_kernel<float, false>;
        VT_LAUNCH(k, grid, block, stream, (StatRec*)partials, (const float*)x, ld_x, (const float*)nullptr, 0, hw, c, cpx, chunks);
    } else {
        auto k = instnorm_partial_kernel<bf16_t, false>;
        VT_LAUNCH(k, grid, block, stream, (StatRec*)partials, (const bf16_t*)x, ld_x, (const bf16_t*)nullptr, 0, hw, c, cpx, chunks);
    }
    return vt_check_launch("instnorm statistics");
}

// The finalize/apply half of vt_instnorm_apply alone: `partials` already hold the chunk records of
// x -- written by the conv that produced x (vt_conv_desc.stats_part).
extern "C" int vt_instnorm_apply_stats(void* out, int ld_out, const void* x, int ld_x, int n, int hw, int c,
                                       const float* style_gb, int ld_gb, const void* partials, int dtype,
                                       vt_stream stream) {
    VT_REQUIRE(out && x && partials, "vt_instnorm_apply_stats: null tensor");
    VT_REQUIRE(n > 0 && hw > 0 && c > 0 && c % 16 == 0, "vt_instnorm_apply_stats: c must be a positive multiple of 16");
    VT_REQUIRE(dtype == VT_F32 || dtype == VT_BF16, "vt_instnorm_apply_stats: dtype");
    if ((int64_t)hw > 16384) {
        vt_set_error("vt_instnorm_apply_stats: tensor too large for the fused form");
        return VT_ERR_UNSUPPORTED;
    }
    const int cpx = stat_chunk_pixels(hw);
    const int chunks = (hw + cpx - 1) / cpx;
    const unsigned nblk = (unsigned)(n * (c / (dtype == VT_F32 ? 4 : 8)));
    if (dtype == VT_F32) {
        auto k = instnorm_apply_small_kernel<float>;
        VT_LAUNCH(k, dim3(nblk), dim3(256), stream, (float*)out, ld_out, (const float*)x, ld_x,
                  (const StatRec*)partials, hw, c, cpx, chunks, style_gb, ld_gb);
    } else {
        auto k = instnorm_apply_small_kernel<bf16_t>;
        VT_LAUNCH(k, dim3(nblk), dim3(256), stream, (bf16_t*)out, ld_out, (const bf16_t*)x, ld_x,
                  (const StatRec*)partials, hw, c, cpx, chunks, style_gb, ld_gb);
    }
    return vt_check_launch("vt_instnorm_apply_stats");
}

// AdaIN of a small plane in one launch (instnorm_plane_kernel); hw <= 4096, else VT_ERR_UNSUPPORTED.
extern "C" int vt_instnorm_plane(void* out, int ld_out, const void* x, int ld_x, const void* absdiff_other, int ld_other,
                                 int n, int hw, int c, const float* style_gb, int ld_gb, int dtype, vt_stream stream) {
    VT_REQUIRE(out && x, "vt_instnorm_plane: null tensor");
    VT_REQUIRE(n > 0 && hw > 0 && c > 0 && c % 8 == 0, "vt_instnorm_plane: c must be a positive multiple of 8");
    VT_REQUIRE(dtype == VT_F32 || dtype == VT_BF16, "vt_instnorm_plane: dtype");
    VT_REQUIRE(!absdiff_other || out != x, "vt_instnorm_plane: the cat[x, |x - other|] form cannot run in place");
    if (hw > 4096) {
        vt_set_error("vt_instnorm_plane: plane too large for the register-resident form (hw <= 4096)");
        return VT_ERR_UNSUPPORTED;
    }
    // vectors per workgroup x pixels per thread: the widest rows (64 contiguous bytes per pixel) the plane's size and the
    // channel count allow -- (4 x 4) 256 pixels, (4 x 16) 1024, (2 x 16) 2048, (1 x 16) 4096
    const int vec = dtype == VT_F32 ? 4 : 8;
    int vpw = hw <= 1024 ? 4 : hw <= 2048 ? 2 : 1;
    while (vpw > 1 && c % (vec * vpw) != 0) vpw >>= 1;
    const int ppt = (hw + 256 / vpw - 1) / (256 / vpw);
    if (ppt > 16) vpw = 1;   // (a channel count that forced narrower rows than the plane's size wanted)
    const unsigned nblk = (unsigned)(n * (c / (vec * vpw)));
#define VT_PLANE(TT, P_, V_, HO_)                                                                     \
    {                                                                                                 \
        auto k = instnorm_plane_kernel<TT, P_, V_, HO_>;                                              \
        VT_LAUNCH(k, dim3(nblk), dim3(256), stream, (TT*)out, ld_out, (const TT*)x, ld_x,             \
                  (const TT*)absdiff_other, ld_other, hw, c, style_gb, ld_gb);                        \
    }
#define VT_PLANE_P(TT, HO_)                                                                           \
    if (vpw == 4 && hw <= 256) VT_PLANE(TT, 4, 4, HO_) else if (vpw == 4) VT_PLANE(TT, 16, 4, HO_)    \
    else if (vpw == 2) VT_PLANE(TT, 16, 2, HO_) else VT_PLANE(TT, 16, 1, HO_)
    if (dtype == VT_F32) {
        if (absdiff_other) { VT_PLANE_P(float, true) } else { VT_PLANE_P(float, false) }
    } else {
        if (absdiff_other) { VT_PLANE_P(bf16_t, true) } else { VT_PLANE_P(bf16_t, false) }
    }
#undef VT_PLANE_P
#undef VT_PLANE
    return vt_check_launch("vt_instnorm_plane");
}

// AdaIN in two launches (statistics + fused finalize/apply) when the tensor is small enough for
// every workgroup to re-derive all channel statistics; returns VT_ERR_UNSUPPORTED otherwise (use
// vt_instnorm_stats + vt_affine_apply).
extern "C" int vt_instnorm_apply(void* out, int ld_out, const void* x, int ld_x, int n, int hw, int c,
                                 const float* style_gb, int ld_gb, void* partials, int dtype,
                                 vt_stream stream) {
    VT_REQUIRE(out && x && partials, "vt_instnorm_apply: null tensor");
    VT_REQUIRE(n > 0 && hw > 0 && c > 0 && c % 16 == 0, "vt_instnorm_apply: c must be a positive multiple of 16");
    VT_REQUIRE(dtype == VT_F32 || dtype == VT_BF16, "vt_instnorm_apply: dtype");
    const int cpx = stat_chunk_pixels(hw);
    const int chunks = (hw + cpx - 1) / cpx;
    if ((int64_t)hw > 16384) {   // one workgroup walks a whole plane: small planes only
        vt_set_error("vt_instnorm_apply: tensor too large for the fused form");
        return VT_ERR_UNSUPPORTED;
    }
    dim3 grid((unsigned)(n * chunks)), block(256);
    if (dtype == VT_F32) {
        auto k = instnorm_partial_kernel<float, false>;
        VT_LAUNCH(k, grid, block, stream, (StatRec*)partials, (const float*)x, ld_x, (const float*)nullptr, 0, hw, c, cpx, chunks);
    } else {
        auto k = instnorm_partial_kernel<bf16_t, false>;
        VT_LAUNCH(k, grid, block, stream, (StatRec*)partials, (const bf16_t*)x, ld_x, (const bf16_t*)nullptr, 0, hw, c, cpx, chunks);
    }
    int rc = vt_check_launch("vt_instnorm_apply(partial)");
    if (rc) return rc;
    const unsigned nblk = (unsigned)(n * (c / (dtype == VT_F32 ? 4 : 8)));
    if (dtype == VT_F32) {
        auto k = instnorm_apply_small_kernel<float>;
        VT_LAUNCH(k, dim3(nblk), block, stream, (float*)out, ld_out, (const float*)x, ld_x,
                  (const StatRec*)partials, hw, c, cpx, chunks, style_gb, ld_gb);
    } else {
        auto k = instnorm_apply_small_kernel<bf16_t>;
        VT_LAUNCH(k, dim3(nblk), block, stream, (bf16_t*)out, ld_out, (const bf16_t*)x, ld_x,
                  (const StatRec*)partials, hw, c, cpx, chunks, style_gb, ld_gb);
    }
    return vt_check_launch("vt_instnorm_apply");
}
