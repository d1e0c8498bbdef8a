// Style path of the synthesis network on gfx950 (frame-invariant per (style, d_s)):
//   vt_linear          EqualLinear / nn.Linear            model/stylegan/model.py:152-162
//   vt_pixel_norm      PixelNorm                          model/stylegan/model.py:17-18
//   vt_modulate_weight ModulatedConv2d modulate + demod   model/stylegan/model.py:259-267
//                      (+ polyphase fold of conv_transpose2d + Blur, model.py:273-286)
//   vt_pack_conv_weight  plain conv weights -> implicit-GEMM layout
//
// All reductions (dot products, sum of squares over cin*k*k <= 4608 elements) are done
// by ONE 64-lane wavefront per output with xor-shuffle butterflies -- no LDS, no
// atomics, deterministic.
#include "vt_common.hpp"

namespace {

// ---------------------------------------------------------------------------------
// y[r,o] = act(dot(x[r,:], W[o,:]) * w_scale + b[o] * b_scale)
// The reference scales the weight matrix (W*scale) before the GEMV; scaling the dot
// product instead differs only by fp32 reassociation.
// ---------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
linear_kernel(float* __restrict__ y, int ld_y, const float* __restrict__ x, int ld_x,
              const float* __restrict__ W, const float* __restrict__ b, int rows, int in_dim,
              int out_dim, float w_scale, float b_scale, int act, float slope, float gain) {
    const int lane = threadIdx.x & 63;
    const int64_t wave = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int64_t total = (int64_t)rows * out_dim;
    // no early return before the shuffles: clamp and mask the store instead
    const int64_t wi = wave < total ? wave : total - 1;
    const int r = (int)(wi / out_dim), o = (int)(wi % out_dim);
    const float* xr = x + (int64_t)r * ld_x;
    const float* wr = W + (int64_t)o * in_dim;
    float acc = 0.0f;
    for (int i = lane; i < in_dim; i += 64) acc += xr[i] * wr[i];
    acc = wave_sum(acc);
    if (lane == 0 && wave < total) {
        float v = acc * w_scale;
        if (b) v += b[o] * b_scale;
        if (act == VT_ACT_LRELU) v = ((v > 0.0f) ? v : v * slope) * gain;
        else if (act == VT_ACT_SIGMOID) v = 1.0f / (1.0f + expf(-v));
        y[(int64_t)r * ld_y + o] = v;
    }
}

__global__ void __launch_bounds__(256)
pixel_norm_kernel(float* __restrict__ y, const float* __restrict__ x, int rows, int dim, const int* __restrict__ gate) {
    if (gate && gate[0] == 0) return;   // style gate: the inputs of the style path did not change (vt_style_gate)
    const int lane = threadIdx.x & 63;
    const int wave = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int r = wave < rows ? wave : rows - 1;
    const float* xr = x + (int64_t)r * dim;
    float ss = 0.0f;
    for (int i = lane; i < dim; i += 64) ss += xr[i] * xr[i];
    ss = wave_sum(ss);
    const float inv = rsqrtf(ss / (float)dim + 1e-8f);
    if (wave < rows)
        for (int i = lane; i < dim; i += 64) y[(int64_t)r * dim + i] = xr[i] * inv;
}

// ---------------------------------------------------------------------------------
// Modulated conv weight.  One wavefront per output channel `co`.
//   pass 1: sumsq over (ci, tap) of (scale * w * s[ci])^2  -> demod
//   pass 2: write packed [co][tap][ci]                      (fir == nullptr)
//        or the four 3x3 polyphase filters of convT(stride 2) followed by the 4x4 blur:
//           Weff[p=(py,px)][co][ky,kx][ci] =
//               sum_{a,b} w'[co,ci,a,b] * K[4 - 2ky - a + py][4 - 2kx - b + px]
//           (K indices outside 0..3 contribute 0), packed [p*cout + co][ky*3+kx][ci].
//     Derivation: DESIGN.md "Up-sampling StyledConv as one polyphase GEMM".
// ---------------------------------------------------------------------------------
template <typename T>
__global__ void __launch_bounds__(256)
modulate_weight_kernel(T* __restrict__ out, const float* __restrict__ w, const float* __restrict__ s,
                       int cout, int cin, int k, float scale, int demodulate,
                       const float* __restrict__ fir) {
    const int lane = threadIdx.x & 63;
    const int wave = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int co = wave < cout ? wave : cout - 1;
    const bool live = wave < cout;
    const int taps = k * k;
    const int n = cin * taps;
    const float* wc = w + (int64_t)co * n;  // [ci][a][b]
    float demod = 1.0f;
    if (demodulate) {
        float ss = 0.0f;
        for (int i = lane; i < n; i += 64) {
            const float v = scale * wc[i] * s[i / taps];
            ss += v * v;
        }
        ss = wave_sum(ss);
        demod = rsqrtf(ss + 1e-8f);
    }
    if (!live) return;
    if (fir == nullptr) {
        T* oc = out + (int64_t)co * n;  // [tap][ci]
        for (int i = lane; i < n; i += 64) {
            const int tap = i / cin, ci = i - tap * cin;
            const float v = scale * wc[ci * taps + tap] * s[ci] * demod;
            oc[i] = from_f32<T>(v);
        }
    } else {
        // k == 3, fir 4x4
        float K[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) K[i] = fir[i];
        for (int ci = lane; ci < cin; ci += 64) {
            float wm[9];
#pragma unroll
            for (int t = 0; t < 9; ++t) wm[t] = scale * wc[ci * 9 + t] * s[ci] * demod;
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                const int py = p >> 1, px = p & 1;
#pragma unroll
                for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                    for (int kx = 0; kx < 3; ++kx) {
                        float acc = 0.0f;
#pragma unroll
                        for (int a = 0; a < 3; ++a) {
                            const int u = 4 - 2 * ky - a + py;
                            if (u < 0 || u > 3) continue;
#pragma unroll
                            for (int b = 0; b < 3; ++b) {
                                const int v = 4 - 2 * kx - b + px;
                                if (v < 0 || v > 3) continue;
                                acc += wm[a * 3 + b] * K[u * 4 + v];
                            }
                        }
                        out[((int64_t)(p * cout + co) * 9 + (ky * 3 + kx)) * cin + ci] = from_f32<T>(acc);
                    }
            }
        }
    }
}

// ---------------------------------------------------------------------------------
// Batched forms: the style path is ~50 tiny independent GEMVs and 15 weight modulations per
// frame; launching them one by one is pure launch latency (5-20 us each).  The tables travel
// BY VALUE in the kernel argument block (no device-side table to keep alive, graph-capture
// safe); a workgroup finds its item with a scalar scan of the wave/row prefix.
// ---------------------------------------------------------------------------------
constexpr int LIN_BATCH = 24, MOD_BATCH = 16;

struct LinearTable {
    vt_linear_item it[LIN_BATCH];
    const int* gate;   // NULL, or the style gate's flag: 0 = the whole launch is skipped
    int32_t first_wave[LIN_BATCH + 1];
    int32_t n;
};

// One wavefront per (item, output o): the weight row W[o, :] is read ONCE (16-byte loads when the row allows it)
// and dotted with every input row of the item, RB rows at a time -- the style MLP applies the same 512x512
// matrix to 18 latent rows, and one wave per (row, output) re-read it 18 times in 4-byte pieces (35 us for the
// first dependency level of a frame; the 12 MB of fp32 weights are 2-3 us of HBM).
// (Measured and dropped, round 2: FOUR output columns per wavefront -- a quarter of the waves, four weight rows in
//  flight per lane -- is 2x slower: 52 + 45 + 20 us against 18 + 17 + 25 us for the three launches of a frame.)
__global__ void __launch_bounds__(256) linear_batch_kernel(const LinearTable t) {
    if (t.gate && t.gate[0] == 0) return;
    constexpr int RB = 6;
    const int lane = threadIdx.x & 63;
    const int wave = blockIdx.x * 4 + (threadIdx.x >> 6);
    // the prefix table goes through LDS once per workgroup: a dependent chain of up to 24 scalar
    // loads from the argument block per wave cost ~10 us (measured), a 5-step LDS search nothing
    __shared__ int pre[LIN_BATCH + 1];
    if (threadIdx.x <= (unsigned)t.n) pre[threadIdx.x] = t.first_wave[threadIdx.x];
    __syncthreads();
    const int total = pre[t.n];
    const int wv = wave < total ? wave : total - 1;
    int lo = 0, hi = t.n;  // invariant: pre[lo] <= wv < pre[hi]
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (wv >= pre[mid]) lo = mid;
        else hi = mid;
    }
    const int k = lo;
    const vt_linear_item& L = t.it[k];
    const int o = wv - t.first_wave[k];
    const float* wr = L.W + (int64_t)o * L.in_dim;
    // the summation order depends on the SHAPE only (groups of 4 consecutive inputs per lane when in_dim % 4 == 0),
    // never on pointer alignment: an engine whose weights are views into a broadcast bucket (4-byte aligned) and
    // one with separately allocated parameters must produce the same bits
    const bool quad = (L.in_dim % 4 == 0);
    const bool w16 = quad && ((uintptr_t)L.W % 16 == 0);
    const bool x16 = quad && (L.ld_x % 4 == 0) && ((uintptr_t)L.x % 16 == 0);
    for (int r0 = 0; r0 < L.rows; r0 += RB) {
        float acc[RB];
#pragma unroll
        for (int j = 0; j < RB; ++j) acc[j] = 0.0f;
        if (quad) {
            for (int i = lane * 4; i < L.in_dim; i += 256) {
                float w4[4];
                if (w16) {
                    unpack16<float>(ld128(wr + i), w4);
                } else {
                    w4[0] = wr[i]; w4[1] = wr[i + 1]; w4[2] = wr[i + 2]; w4[3] = wr[i + 3];
                }
#pragma unroll
                for (int j = 0; j < RB; ++j) {
                    const int r = (r0 + j < L.rows) ? r0 + j : L.rows - 1;   // clamped: loads stay unconditional
                    const float* xp = L.x + (int64_t)r * L.ld_x + i;
                    float x4[4];
                    if (x16) {
                        unpack16<float>(ld128(xp), x4);
                    } else {
                        x4[0] = xp[0]; x4[1] = xp[1]; x4[2] = xp[2]; x4[3] = xp[3];
                    }
                    acc[j] += (x4[0] * w4[0] + x4[1] * w4[1]) + (x4[2] * w4[2] + x4[3] * w4[3]);
                }
            }
        } else {
            for (int i = lane; i < L.in_dim; i += 64) {
                const float w = wr[i];
#pragma unroll
                for (int j = 0; j < RB; ++j) {
                    const int r = (r0 + j < L.rows) ? r0 + j : L.rows - 1;
                    acc[j] += L.x[(int64_t)r * L.ld_x + i] * w;
                }
            }
        }
#pragma unroll
        for (int j = 0; j < RB; ++j) acc[j] = wave_sum(acc[j]);
        if (lane == 0 && wave < total) {
#pragma unroll
            for (int j = 0; j < RB; ++j) {
                if (r0 + j >= L.rows) break;
                float v = acc[j] * L.w_scale;
                if (L.b) v += L.b[o] * L.b_scale;
                if (L.act == VT_ACT_LRELU) v = ((v > 0.0f) ? v : v * L.slope) * L.gain;
                else if (L.act == VT_ACT_SIGMOID) v = 1.0f / (1.0f + expf(-v));
                L.y[(int64_t)(r0 + j) * L.ld_y + o] = v;
            }
        }
    }
}

struct ModTable {
    vt_modulate_item it[MOD_BATCH];
    const int* gate;   // NULL, or the style gate's flag
    int32_t first_row[MOD_BATCH + 1];  // prefix of cout
    int32_t n;
};

// One WORKGROUP (4 wavefronts) per output channel: the sum of squares over cin*k*k is reduced
// by the four waves through LDS in a fixed order, then all 256 lanes write the packed row(s).
template <typename T>
__global__ void __launch_bounds__(256) modulate_batch_kernel(const ModTable t) {
    if (t.gate && t.gate[0] == 0) return;
    __shared__ float red[4];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    int lo = 0, hi = t.n;  // binary search of the row prefix (scalar loads, <= 4 steps)
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if ((int)blockIdx.x >= t.first_row[mid]) lo = mid;
        else hi = mid;
    }
    const int k = lo;
    const vt_modulate_item& M = t.it[k];
    const int co = blockIdx.x - t.first_row[k];
    const int cout = M.cout, cin = M.cin, taps = M.k * M.k;
    const int n = cin * taps;
    const float* w = (const float*)M.weight;
    const float* s = M.s;
    const float* wc = w + (int64_t)co * n;  // [ci][a][b]
    const float scale = M.scale;
    float demod = 1.0f;
    if (M.demodulate) {
        float ss = 0.0f;
        for (int i = tid; i < n; i += 256) {
            const float v = scale * wc[i] * s[i / taps];
            ss += v * v;
        }
        ss = wave_sum(ss);
        if (lane == 0) red[wv] = ss;
        __syncthreads();
        demod = rsqrtf(((red[0] + red[1]) + (red[2] + red[3])) + 1e-8f);
    }
    T* out = (T*)M.out;
    if (M.fir == nullptr) {
        T* oc = out + (int64_t)co * n;  // [tap][ci]
        for (int i = tid; i < n; i += 256) {
            const int tap = i / cin, ci = i - tap * cin;
            oc[i] = from_f32<T>(scale * wc[ci * taps + tap] * s[ci] * demod);
        }
    } else {
        float K[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) K[i] = M.fir[i];
        for (int ci = tid; ci < cin; ci += 256) {
            float wm[9];
#pragma unroll
            for (int q = 0; q < 9; ++q) wm[q] = scale * wc[ci * 9 + q] * s[ci] * demod;
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                const int py = p >> 1, px = p & 1;
#pragma unroll
                for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                    for (int kx = 0; kx < 3; ++kx) {
                        float acc = 0.0f;
#pragma unroll
                        for (int a = 0; a < 3; ++a) {
                            const int u = 4 - 2 * ky - a + py;
                            if (u < 0 || u > 3) continue;
#pragma unroll
                            for (int b = 0; b < 3; ++b) {
                                const int v = 4 - 2 * kx - b + px;
                                if (v < 0 || v > 3) continue;
                                acc += wm[a * 3 + b] * K[u * 4 + v];
                            }
                        }
                        out[((int64_t)(p * cout + co) * 9 + (ky * 3 + kx)) * cin + ci] = from_f32<T>(acc);
                    }
            }
        }
    }
}

// out[co][tap][cd] = scale * w[co][map[cd]][tap]   (or 0 when map[cd] < 0)
template <typename T>
__global__ void __launch_bounds__(256)
pack_weight_kernel(T* __restrict__ out, const float* __restrict__ w, int cout, int cin_src, int kh,
                   int kw, int cin_dst, const int32_t* __restrict__ chan_map, float scale,
                   int src_transposed) {
    const int taps = kh * kw;
    const int64_t total = (int64_t)cout * taps * cin_dst;
    const int64_t stride = (int64_t)gridDim.x * 256;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += stride) {
        const int cd = (int)(i % cin_dst);
        const int64_t t = i / cin_dst;
        const int tap = (int)(t % taps);
        const int co = (int)(t / taps);
        const int cs = chan_map ? chan_map[cd] : (cd < cin_src ? cd : -1);
        float v = 0.0f;
        if (cs >= 0) {
            if (!src_transposed) {
                v = w[((int64_t)co * cin_src + cs) * taps + tap];
            } else {
                // source (cin, cout, kh, kw); the gather form of conv_transpose2d visits
                // tap (ky,kx) at input offset -(ky,kx)*dil, so taps keep their index.
                v = w[((int64_t)cs * cout + co) * taps + tap];
            }
        }
        out[i] = from_f32<T>(v * scale);
    }
}

}  // namespace

extern "C" int vt_linear(float* y, int ld_y, const float* x, int ld_x, const float* W,
                         const float* b, int rows, int in_dim, int out_dim, float w_scale,
                         float b_scale, int act, float slope, float gain, vt_stream stream) {
    VT_REQUIRE(y && x && W, "vt_linear: null tensor");
    VT_REQUIRE(rows >= 0 && in_dim > 0 && out_dim > 0, "vt_linear: bad sizes");
    VT_REQUIRE(act == VT_ACT_NONE || act == VT_ACT_LRELU || act == VT_ACT_SIGMOID, "vt_linear: unsupported act %d", act);
    if (rows == 0) return VT_OK;
    const int64_t waves = (int64_t)rows * out_dim;
    VT_LAUNCH(linear_kernel, dim3((unsigned)((waves + 3) / 4)), dim3(256), stream, y, ld_y, x, ld_x,
              W, b, rows, in_dim, out_dim, w_scale, b_scale, act, slope, gain);
    return vt_check_launch("vt_linear");
}

extern "C" int vt_pixel_norm(float* y, const float* x, int rows, int dim, vt_stream stream) {
    VT_REQUIRE(y && x && rows >= 0 && dim > 0, "vt_pixel_norm: bad arguments");
    if (rows == 0) return VT_OK;
    VT_LAUNCH(pixel_norm_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), stream, y, x, rows, dim, (const int*)nullptr);
    return vt_check_launch("vt_pixel_norm");
}

extern "C" int vt_modulate_weight(void* out, const float* weight, const float* s, int cout, int cin,
                                  int k, float scale, int demodulate, const float* fir,
                                  int out_dtype, vt_stream stream) {
    VT_REQUIRE(out && weight && s, "vt_modulate_weight: null tensor");
    VT_REQUIRE(cout > 0 && cin > 0 && (k == 1 || k == 3), "vt_modulate_weight: bad shape");
    VT_REQUIRE(!fir || k == 3, "vt_modulate_weight: polyphase fold needs a 3x3 kernel");
    dim3 grid((unsigned)((cout + 3) / 4)), block(256);
    if (out_dtype == VT_F32) {
        auto kf = modulate_weight_kernel<float>;
        VT_LAUNCH(kf, grid, block, stream, (float*)out, weight, s, cout, cin, k, scale, demodulate, fir);
    } else if (out_dtype == VT_BF16) {
        auto kf = modulate_weight_kernel<bf16_t>;
        VT_LAUNCH(kf, grid, block, stream, (bf16_t*)out, weight, s, cout, cin, k, scale, demodulate, fir);
    } else {
        vt_set_error("vt_modulate_weight: unsupported dtype %d", out_dtype);
        return VT_ERR_UNSUPPORTED;
    }
    return vt_check_launch("vt_modulate_weight");
}

extern "C" int vt_pack_conv_weight(void* out, const float* w, int cout, int cin_src, int kh, int kw,
                                   int cin_dst, const int32_t* chan_map, float scale,
                                   int src_transposed, int out_dtype, vt_stream stream) {
    VT_REQUIRE(out && w, "vt_pack_conv_weight: null tensor");
    VT_REQUIRE(cout > 0 && cin_src > 0 && cin_dst > 0 && kh > 0 && kw > 0, "vt_pack_conv_weight: bad shape");
    const int64_t total = (int64_t)cout * kh * kw * cin_dst;
    int64_t blocks = (total + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    dim3 grid((unsigned)blocks), block(256);
    if (out_dtype == VT_F32) {
        auto kf = pack_weight_kernel<float>;
        VT_LAUNCH(kf, grid, block, stream, (float*)out, w, cout, cin_src, kh, kw, cin_dst, chan_map, scale, src_transposed);
    } else if (out_dtype == VT_BF16) {
        auto kf = pack_weight_kernel<bf16_t>;
        VT_LAUNCH(kf, grid, block, stream, (bf16_t*)out, w, cout, cin_src, kh, kw, cin_dst, chan_map, scale, src_transposed);
    } else {
        vt_set_error("vt_pack_conv_weight: unsupported dtype %d", out_dtype);
        return VT_ERR_UNSUPPORTED;
    }
    return vt_check_launch("vt_pack_conv_weight");
}

static int linear_batch_impl(const vt_linear_item* items, int n_items, const int* gate, vt_stream stream);
extern "C" int vt_linear_batch(const vt_linear_item* items, int n_items, vt_stream stream) {
    return linear_batch_impl(items, n_items, nullptr, stream);
}
extern "C" int vt_linear_batch_gated(const vt_linear_item* items, int n_items, const int* gate, vt_stream stream) {
    return linear_batch_impl(items, n_items, gate, stream);
}
static int linear_batch_impl(const vt_linear_item* items, int n_items, const int* gate, vt_stream stream) {
    VT_REQUIRE(items && n_items >= 0, "vt_linear_batch: bad arguments");
    for (int base = 0; base < n_items; base += LIN_BATCH) {
        LinearTable t;
        memset(&t, 0, sizeof(t));
        t.gate = gate;
        t.n = (n_items - base < LIN_BATCH) ? n_items - base : LIN_BATCH;
        int64_t waves = 0;
        for (int i = 0; i < t.n; ++i) {
            const vt_linear_item& L = items[base + i];
            VT_REQUIRE(L.y && L.x && L.W && L.rows > 0 && L.in_dim > 0 && L.out_dim > 0,
                       "vt_linear_batch: item %d: bad tensor/sizes", base + i);
            VT_REQUIRE(L.act == VT_ACT_NONE || L.act == VT_ACT_LRELU || L.act == VT_ACT_SIGMOID,
                       "vt_linear_batch: unsupported act %d", L.act);
            t.it[i] = L;
            t.first_wave[i] = (int32_t)waves;
            waves += (int64_t)L.out_dim;   // one wavefront per output column (all rows of the item)
        }
        VT_REQUIRE(waves < ((int64_t)1 << 30), "vt_linear_batch: too many outputs");
        t.first_wave[t.n] = (int32_t)waves;
        if (waves == 0) continue;
        VT_LAUNCH(linear_batch_kernel, dim3((unsigned)((waves + 3) / 4)), dim3(256), stream, t);
        const int rc = vt_check_launch("vt_linear_batch");
        if (rc) return rc;
    }
    return VT_OK;
}

static int modulate_batch_impl(const vt_modulate_item* items, int n_items, int out_dtype, const int* gate, vt_stream stream);
extern "C" int vt_modulate_weight_batch(const vt_modulate_item* items, int n_items, int out_dtype,
                                        vt_stream stream) {
    return modulate_batch_impl(items, n_items, out_dtype, nullptr, stream);
}
extern "C" int vt_modulate_weight_batch_gated(const vt_modulate_item* items, int n_items, int out_dtype, const int* gate,
                                              vt_stream stream) {
    return modulate_batch_impl(items, n_items, out_dtype, gate, stream);
}
static int modulate_batch_impl(const vt_modulate_item* items, int n_items, int out_dtype, const int* gate, vt_stream stream) {
    VT_REQUIRE(items && n_items >= 0, "vt_modulate_weight_batch: bad arguments");
    VT_REQUIRE(out_dtype == VT_F32 || out_dtype == VT_BF16, "vt_modulate_weight_batch: unsupported dtype %d", out_dtype);
    for (int base = 0; base < n_items; base += MOD_BATCH) {
        ModTable t;
        memset(&t, 0, sizeof(t));
        t.gate = gate;
        t.n = (n_items - base < MOD_BATCH) ? n_items - base : MOD_BATCH;
        int64_t rows = 0;
        for (int i = 0; i < t.n; ++i) {
            const vt_modulate_item& M = items[base + i];
            VT_REQUIRE(M.out && M.weight && M.s && M.cout > 0 && M.cin > 0 && (M.k == 1 || M.k == 3),
                       "vt_modulate_weight_batch: item %d: bad tensor/shape", base + i);
            VT_REQUIRE(!M.fir || M.k == 3, "vt_modulate_weight_batch: polyphase fold needs a 3x3 kernel");
            t.it[i] = M;
            t.first_row[i] = (int32_t)rows;
            rows += M.cout;
        }
        t.first_row[t.n] = (int32_t)rows;
        if (rows == 0) continue;
        if (out_dtype == VT_F32) {
            auto kf = modulate_batch_kernel<float>;
            VT_LAUNCH(kf, dim3((unsigned)rows), dim3(256), stream, t);
        } else {
            auto kf = modulate_batch_kernel<bf16_t>;
            VT_LAUNCH(kf, dim3((unsigned)rows), dim3(256), stream, t);
        }
        const int rc = vt_check_launch("vt_modulate_weight_batch");
        if (rc) return rc;
    }
    return VT_OK;
}


// ---------------------------------------------------------------------------------
// Style gate (round 3): the style path (mapping linears, modulation + demodulation of 15 conv weights, AdaIN gamma /
// beta) depends only on (W+ rows, d_s), which a video keeps for all of its frames (style_transfer.py:138-150,176 passes
// `s_w.repeat(B,1,1)` -- a NEW tensor with the same content -- on every call).  vt_style_gate compares the caller's rows
// with the ones the plan's style products were computed from, ON THE DEVICE (no host sync): flag[0] = 1 and the rows are
// adopted when anything differs (or flag[1], the host's "force" word, is set: a new plan, another d_s), else flag[0] = 0
// and the *_gated launches of the style path return at once.  Bitwise comparison: -0.0 vs 0.0 or a NaN payload count as
// a change, which only costs a recomputation.
// ---------------------------------------------------------------------------------
namespace {
__global__ void __launch_bounds__(256) style_gate_kernel(int* __restrict__ flag, uint32_t* __restrict__ cached,
                                                          const uint32_t* __restrict__ fresh, int n) {
    __shared__ int diff;
    if (threadIdx.x == 0) diff = flag[1];
    __syncthreads();
    int d = 0;
    for (int i = threadIdx.x; i < n; i += 256) d |= (cached[i] != fresh[i]) ? 1 : 0;
    if (d) diff = 1;   // benign race: every writer stores 1
    __syncthreads();
    const int changed = diff;
    if (changed) {
        for (int i = threadIdx.x; i < n; i += 256) cached[i] = fresh[i];
    }
    if (threadIdx.x == 0) {
        flag[0] = changed;
        flag[1] = 0;
    }
}
}  // namespace

extern "C" int vt_style_gate(int* flag, float* cached, const float* fresh, int n, vt_stream stream) {
    VT_REQUIRE(flag && cached && fresh && n > 0, "vt_style_gate: bad arguments");
    VT_LAUNCH(style_gate_kernel, dim3(1), dim3(256), stream, flag, (uint32_t*)cached, (const uint32_t*)fresh, n);
    return vt_check_launch("vt_style_gate");
}

extern "C" int vt_pixel_norm_gated(float* y, const float* x, int rows, int dim, const int* gate, vt_stream stream) {
    VT_REQUIRE(y && x && rows > 0 && dim > 0, "vt_pixel_norm: bad arguments");
    VT_LAUNCH(pixel_norm_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), stream, y, x, rows, dim, gate);
    return vt_check_launch("vt_pixel_norm");
}
