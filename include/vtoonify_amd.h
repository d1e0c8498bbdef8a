/*
 * vtoonify_amd.h -- C ABI of libvtoonify_amd.so (gfx950 / MI355X).
 *
 * This is the drop-in boundary for VToonify's per-frame inference hot path
 * (williamyang1991/VToonify, model/vtoonify.py:210-277).  The reference reaches its
 * native code through two pybind11 modules JIT-built from CUDA sources:
 *
 *   upfirdn2d_op.upfirdn2d(input, kernel, up_x, up_y, down_x, down_y,
 *                          pad_x0, pad_x1, pad_y0, pad_y1)   model/stylegan/op/upfirdn2d.cpp:17-31
 *   fused.fused_bias_act(input, bias, refer, act, grad, alpha, scale)
 *                                                           model/stylegan/op/fused_bias_act.cpp:18-32
 *
 * and through cuDNN (F.conv2d / F.conv_transpose2d, model/stylegan/op/conv2d_gradfix.py:22-75).
 * The entry points below replace exactly those, plus the frame-invariant style maths
 * (ModulatedConv2d weight modulation, model/stylegan/model.py:259-267) and the
 * normalisation glue that the reference runs as eager aten ops.
 *
 * Conventions
 *   - plain C, no torch types: raw DEVICE pointers, sizes, a hipStream_t passed as void*
 *   - every call is asynchronous on `stream`, never synchronises, never allocates
 *   - inputs are borrowed, outputs are caller-allocated (the Python layer allocates
 *     them, mirroring at::empty in upfirdn2d_kernel.cu:242 / fused_bias_act_kernel.cu:94)
 *   - return 0 on success, a VT_ERR_* code otherwise (vt_last_error() has the text);
 *     nothing throws
 *   - dtype codes: activations/weights may be fp32, bf16 or fp16 where noted; all
 *     accumulation is fp32
 */
#ifndef VTOONIFY_AMD_H
#define VTOONIFY_AMD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VT_ABI_VERSION 5

enum { VT_F32 = 0, VT_BF16 = 1, VT_F16 = 2,
       /* vt_conv_desc.dtype only (ABI 5): fp32 tensors and weights in memory exactly as VT_F32, but the contraction may run on
        * the bf16 matrix cores as three terms per product (bf16 head / remainder split of both operands in registers, fp32
        * accumulate; ~4e-5 of max|y| end to end against fp32).  The reference's fp32 convolutions (cuDNN behind
        * op/conv2d_gradfix.py:34-42, F.conv2d in model/vtoonify.py:92-128) at matrix-core speed; kernel instances without
        * that form run exact fp32.  out_dtype stays VT_F32 / VT_BF16. */
       VT_F32X3 = 3,
       /* vt_upfirdn2d / vt_fused_bias_act only: double tensors AND double arithmetic -- the reference's native ops are
        * dispatched over AT_DISPATCH_FLOATING_TYPES_AND_HALF (upfirdn2d_kernel.cu:311, fused_bias_act_kernel.cu:96), which
        * includes double.  The path never uses it; one plain kernel per op, no tiling.  With VT_F64 the `fir` argument of
        * vt_upfirdn2d points at DOUBLE taps (the reference hands kernel.data_ptr<scalar_t>()). */
       VT_F64 = 4 };
enum { VT_OK = 0, VT_ERR_ARG = 1, VT_ERR_UNSUPPORTED = 2, VT_ERR_LAUNCH = 3 };

/* activation codes of the fused epilogues */
enum { VT_ACT_NONE = 0, VT_ACT_LRELU = 1, VT_ACT_RELU_TANH = 2, VT_ACT_SIGMOID = 3, VT_ACT_TANH = 4 /* conv epilogues: all; vt_linear: 0, 1, 3 */ };
/* output layouts of vt_conv2d */
enum { VT_OUT_NHWC = 0, VT_OUT_NCHW = 1 };

typedef void* vt_stream; /* hipStream_t */

int vt_abi_version(void);
const char* vt_last_error(void);
/* "gfx950" for the product build */
const char* vt_build_target(void);

/* ---------------------------------------------------------------------------------
 * upfirdn2d -- replaces upfirdn2d_op.upfirdn2d (op/upfirdn2d.cpp:17-31,
 * upfirdn2d_kernel.cu:209-369).  `in` is `planes` contiguous (in_h, in_w) images (the
 * Python side folds N*C into planes exactly like op/upfirdn2d.py:100 with minor=1);
 * `fir` is the un-flipped (kh, kw) fp32 kernel (the flip of upfirdn2d_kernel.cu:137 is
 * applied inside).  out_h/out_w follow op/upfirdn2d.py:104-105 and are returned by
 * vt_upfirdn2d_out_size.  dtype: VT_F32 / VT_BF16 / VT_F16 (fp32 accumulate), VT_F64 (double taps, double accumulate).
 * --------------------------------------------------------------------------------- */
int vt_upfirdn2d_out_size(int in_h, int in_w, int kh, int kw, int up_x, int up_y,
                          int down_x, int down_y, int pad_x0, int pad_x1, int pad_y0,
                          int pad_y1, int* out_h, int* out_w);
int vt_upfirdn2d(void* out, const void* in, const float* fir, int64_t planes, int in_h,
                 int in_w, int kh, int kw, int up_x, int up_y, int down_x, int down_y,
                 int pad_x0, int pad_x1, int pad_y0, int pad_y1, int dtype,
                 vt_stream stream);

/* ---------------------------------------------------------------------------------
 * fused_bias_act -- replaces fused.fused_bias_act (op/fused_bias_act.cpp:18-32,
 * fused_bias_act_kernel.cu:18-105):  y = act(x + b[(i / step_b) % size_b]) * scale with
 * act*10+grad in {10,11,12,30,31,32} (kernel.cu:40-61).  `bias` / `refer` may be NULL
 * ("empty tensor" in the reference).  bias has the dtype of x.
 * --------------------------------------------------------------------------------- */
int vt_fused_bias_act(void* out, const void* x, const void* bias, const void* refer,
                      int64_t numel, int64_t step_b, int size_b, int act, int grad,
                      float alpha, float scale, int dtype, vt_stream stream);

/* ---------------------------------------------------------------------------------
 * Dense contraction -- replaces every F.conv2d / F.conv_transpose2d / nn.Conv2d on the
 * path (op/conv2d_gradfix.py:34-42,66-75; model/vtoonify.py:96-97,111-113,162-198) with
 * one implicit-GEMM MFMA kernel family.
 *
 * Activations are NHWC.  The input may be the channel concatenation of two tensors
 * (torch.cat of model/vtoonify.py:125,127,259) given as two (pointer, channels, pixel
 * stride) triples.  Weights are packed [cout_total][kh*kw][cin] in the compute dtype
 * (vt_pack_conv_weight / vt_modulate_weight produce that layout).
 *
 *   acc   = sum_{tap,c} in[n, oy*stride + ky*dil - pad, ox*stride + kx*dil - pad, c] * w[co,tap,c]
 *   v     = act(acc + bias[co]) * gain * alpha * (alpha_dev ? *alpha_dev : 1)
 *   out   = v + beta * resid
 *
 * phases == 4: "up-sampling StyledConv" form.  cout_total = 4*cout, output column
 * p*cout+co of input-grid pixel (y,x) is stored at pixel (2y + p/2, 2x + p%2) of a
 * (2h, 2w) image: conv_transpose2d(stride 2) followed by the 4x4 FIR blur
 * (model/stylegan/model.py:273-286) collapses into four 3x3 polyphase filters.
 * transposed != 0: gather form of F.conv_transpose2d (used by the generic
 * conv2d_gradfix.conv_transpose2d entry point only).
 * --------------------------------------------------------------------------------- */
typedef struct vt_conv_desc {
    const void* src0;      /* NHWC, dtype `dtype` */
    const void* src1;      /* second concat source or NULL */
    int32_t c0, c1;        /* channels read from src0 / src1 (multiples of 8) */
    int32_t ld0, ld1;      /* per-pixel stride of each source, in elements */
    int32_t n, h, w;       /* input batch / height / width */
    int32_t out_h, out_w;  /* conv output size (before the x2 of phases==4) */
    const void* weight;    /* [phases*cout][kh*kw][c0+c1], dtype `dtype` */
    int32_t cout;
    int32_t kh, kw, stride, pad, dil;
    int32_t phases;        /* 1 or 4 */
    int32_t transposed;    /* 0 or 1 */
    const float* in_scale; /* optional per-(n, cin) affine prologue (AdaIN):   */
    const float* in_shift; /*   x' = x * in_scale[n][c] + in_shift[n][c]; NULL = off */
    const float* bias;     /* [cout] fp32 or NULL */
    int32_t act;           /* VT_ACT_* */
    float slope, gain, alpha, beta;
    const float* alpha_dev; /* optional device scalar (style degree d_s) */
    const void* resid;     /* same layout/dtype as out, or NULL */
    int32_t ld_res;
    void* out;
    int32_t ld_out;        /* NHWC: per-pixel stride in elements; NCHW: ignored */
    int32_t out_layout;    /* VT_OUT_NHWC / VT_OUT_NCHW */
    int32_t out_dtype;     /* VT_F32 / VT_BF16 (NCHW output is always fp32) */
    int32_t dtype;         /* VT_F32 / VT_BF16: dtype of src*, weight; VT_F32X3: fp32 tensors, bf16 x 3 products (above) */
    int32_t tile_hint;     /* 0 = auto; otherwise SPLITK*1000000 + BM*1000 + BN of a compiled tile
                              (SPLITK 0 = auto); +1000000000 forces the register-staged
                              loader where the direct-to-LDS one would apply */
    void* splitk_ws;       /* optional workspace for split-K (NULL: never split).  Its first 16 KiB are
                              per-tile arrival counters and MUST be zero before the first launch
                              (every launch leaves them zero).  Small-M /                            */
    int64_t splitk_ws_bytes; /* small-cout convs with a deep K are cut along K into slices that run
                              as separate workgroups and are summed in slice order (deterministic) by a
                              second kernel (or by the last slice to arrive, VT_SPLITK_IN_LAUNCH=1: slower on MI355X,
                              see vt_conv2d_splitk_mode()); vt_conv2d_ws_bytes() says how much the heuristic wants */
    const float* slope_vec; /* optional per-output-channel negative slope [cout] for VT_ACT_LRELU
                              (nn.PReLU of the pSp encoder, model/encoder/encoders/helpers.py:97-119);
                              NULL = the scalar `slope` */
    /* Fused ToRGB (model/stylegan/model.py:383-392) for a same-resolution StyledConv whose tile holds
     * all `cout` channels (vt_conv2d_tile: BN >= cout, no split-K; VT_ERR_UNSUPPORTED otherwise):
     *   rgb_out[n][j][p] = sum_c rgb_weight[j][c] * out_fp32[n][p][c] + rgb_bias[j] + rgb_resid[n][j][p]
     * rgb_weight: packed [3][1][cout] in `dtype` (vt_modulate_weight of the ToRGB layer, no demod);
     * rgb_out / rgb_resid: planar fp32 (n,3,out_h,out_w), may alias.  NULL rgb_weight = off. */
    const void* rgb_weight;
    const float* rgb_bias;
    const float* rgb_resid;
    float* rgb_out;
    int32_t splitk_phase;  /* two-pass split-K only: 0 = slices + reduce (default), 1 = launch the K
                              slices only, 2 = launch the reduce pass only (lets a caller time or
                              schedule the two kernels separately) */
    void* stats_part;      /* NULL, or vt_instnorm_ws_bytes(n, out_h*out_w, cout) bytes that receive the
                              InstanceNorm chunk records of the tensor this conv writes (NHWC output in
                              the compute dtype, phases == 1): the split-K reduce pass emits them while
                              it writes the output, any other plan appends the statistics launch.
                              Consumer: vt_instnorm_apply_stats. */
    int32_t post_relu;     /* != 0: out = max(v + beta * resid, 0) -- the ReLU that FOLLOWS the shortcut add
                              of a ResNet BasicBlock (model/bisenet/resnet.py:36-48); 0 = off */
    const void* weight_stream; /* NULL, or the vt_conv_weight_stream image of `weight` (same values in MFMA-fragment
                              order).  Enables the whole-K kernel for 3x3 stride-1 pad == dil convs whose channel
                              count is a multiple of 512 (bf16) / 256 (fp32): few-pixel, wide-channel layers run
                              without split-K slabs and without a reduce pass (KIND 4 of vt_conv2d_tile) */
    /* AdaIN / InstanceNorm folded into whole-K convolutions (model/dualstylegan.py:6-21,38-45: every conv of an
     * AdaResBlock is preceded by AdaIN of the previous conv's output).  Both ends must run the whole-K kernel
     * (vt_conv2d_tile KIND 4), VT_ERR_UNSUPPORTED otherwise:
     *   tile_stats     out: per (image, 8x8-pixel tile of THIS conv, channel) {mean, M2} fp32 records of the
     *                  tensor this conv writes (the rounded values as stored), followed by the pixel count of
     *                  every tile; vt_conv_tile_stats_bytes() bytes in all;
     *   in_tile_stats  in: such records of the tensor read through src0, written by a conv of dilation
     *                  in_stats_dil; the kernel merges them in tile order (fp64) into mean / biased variance per
     *                  (image, channel), eps 1e-5, and convolves AdaIN(x) = x*gamma*rstd + (beta - gamma*rstd*mean)
     *                  (rounded to `dtype` like a stored tensor; zero padding stays zero) instead of x;
     *   in_gb          [n or 1][2*c0] fp32, gamma first (the style Linear of AdaptiveInstanceNorm), row stride
     *                  in_ld_gb (0: one style for the batch); NULL = plain InstanceNorm (gamma 1, beta 0). */
    void* tile_stats;
    const void* in_tile_stats;
    int32_t in_stats_dil;
    const float* in_gb;
    int32_t in_ld_gb;
    /* Up-sampling StyledConv at the reference's MAC count (model/stylegan/model.py:273-286): up_fir != NULL makes
     * this descriptor   out = act(upfirdn2d(conv_transpose2d(src0, W, stride 2, padding 0), up_fir, pad (1,1)) + bias) * gain
     * with `weight` the PLAIN modulated 3x3 filters [cout][9][cin] (vt_modulate_weight with fir == NULL, tap (a, b) at
     * a*3+b as conv_transpose2d indexes them), h x w the INPUT size and out_h = 2h, out_w = 2w.  up_fir: (4,4) fp32
     * device taps, an outer product (make_kernel of a 1-D list, model.py:21-29).  The transposed conv runs on the
     * matrix cores (9 MACs per input pixel instead of the 36 of the phases == 4 form), the blur on the vector
     * ALUs out of LDS; the (2h+1)^2 intermediate never reaches HBM.  KIND 5 of vt_conv2d_tile (KIND 9: bf16 layers of
     * 64 / 128 input channels and >= 128^2 input pixels, where the blur runs on the matrix cores too).  kh = kw = 3,
     * phases = 1, single source, NHWC output in the compute dtype, no residual. */
    const float* up_fir;
    /* ABI 3: horizontal padding + 1 when it differs from `pad` (0 = same as `pad`): the (1,5) / (5,1) convs of
     * RAFT's SepConvGRU (model/raft/core/update.py:37-42: padding (0,2) / (2,0)); `pad` is then the vertical one.
     * Such convs run on the register-staged kernel. */
    int32_t pad_w_p1;
    /* ABI 4: != 0 together with rgb_weight: only the fused ToRGB image (rgb_out) is written, the C-channel activation
     * is NOT stored (`out` is ignored).  The last synthesis level's StyledConv output feeds nothing but its ToRGB
     * (model/stylegan/model.py:364-392, model/vtoonify.py:269-272): 67 MB per 1024^2 frame that the reference writes
     * and nobody reads.  Supported by the persistent 32 -> 32 kernel (KIND 3); VT_ERR_UNSUPPORTED elsewhere. */
    int32_t rgb_only;
    /* ABI 4: != 0: the input is cat[src0, |src0 - src1|] (src1 = "other", c1 == c0): the operand of the Fusion gate's mask
     * conv (model/vtoonify.py:125-126), formed in the loader; with in_scale / in_shift ([n][2 c0], e.g. from
     * vt_instnorm_stats) the AdaIN affine is applied there too, so vt_affine_apply and its 2C-channel normalised copy are not
     * needed.  Bit-identical to vt_affine_apply -> vt_conv2d.  Supported by the thin-output kernel (KIND 6: 3x3, pad 1,
     * cout == 1); VT_ERR_UNSUPPORTED elsewhere. */
    int32_t in_absdiff;
} vt_conv_desc;

int vt_conv2d(const vt_conv_desc* desc, vt_stream stream);
/* The workgroup tile vt_conv2d would run `desc` on, as KIND*100000000+SPLITK*1000000+BM*1000+BN (KIND 0 register-staged, 1 patch-resident, 2 direct-to-LDS, 3 persistent 32->32 kernel, 4 whole-K kernel, 5 conv_transpose+blur kernel, 6 thin-output kernel, 7 persistent 64->64 kernel; -1: invalid descriptor).
 * Host-only query (no launch); lets a profiler name the kernel instance of each launch. */
int vt_conv2d_tile(const vt_conv_desc* desc);
/* Bytes of split-K workspace vt_conv2d would like for `desc` (0: it would not split). */
int64_t vt_conv2d_ws_bytes(const vt_conv_desc* desc);
/* How vt_conv2d would finish a split-K launch of `desc`: 0 = it does not split, 1 = in the same launch (the last
 * slice to arrive at a tile's ticket sums the slabs), 2 = a second kernel (what splitk_phase 1 / 2 lets a caller
 * issue as two launches of its own).  -1: invalid descriptor.  Host-only query. */
int vt_conv2d_splitk_mode(const vt_conv_desc* desc);

/* Plain conv weight (cout, cin_src, kh, kw) fp32 -> packed [cout][kh*kw][cin_dst],
 * multiplied by `scale` (EqualConv2d's 1/sqrt(fan_in), model/stylegan/model.py:101,117).
 * chan_map[cin_dst] (device int32) gives the source channel of each packed channel, -1
 * for zero padding; NULL = identity.  src_transposed: source is (cin_src, cout, kh, kw)
 * as F.conv_transpose2d expects; spatial taps are flipped for the gather form. */
int vt_pack_conv_weight(void* out, const float* w, int cout, int cin_src, int kh, int kw,
                        int cin_dst, const int32_t* chan_map, float scale,
                        int src_transposed, int out_dtype, vt_stream stream);

/* Fragment-stream image of packed weights for the whole-K kernel (vt_conv_desc.weight_stream):
 *   packed [cout][taps][cin] (vt_pack_conv_weight layout, dtype VT_F32 / VT_BF16, cin a multiple of the
 *   128-byte K-step: 64 bf16 / 32 fp32)  ->  [ceil(cout/32)][cin/BK][taps][2][2][64 lanes] x 16 bytes,
 * i.e. for every (32-channel tile, channel chunk, tap, half chunk, 16-row fragment) the 1 KB a wavefront
 * feeds to the MFMA as its weight operand, in lane order; rows beyond cout are zero.
 * vt_conv_weight_stream_bytes gives the size of `out` (> 0) or -1 for unsupported shapes. */
int64_t vt_conv_weight_stream_bytes(int cout, int taps, int cin, int dtype);
/* Bytes of a vt_conv_desc.tile_stats buffer for an (n, h, w, c) output written by a conv of dilation `dil`. */
int64_t vt_conv_tile_stats_bytes(int n, int h, int w, int dil, int c);
int vt_conv_weight_stream(void* out, const void* packed, int cout, int taps, int cin, int dtype,
                          vt_stream stream);

/* ModulatedConv2d weight path (model/stylegan/model.py:259-267):
 *   w'[co,ci,a,b] = scale * weight[co,ci,a,b] * s[ci];  demod: w' *= rsqrt(sum w'^2 + 1e-8)
 * (one wavefront per output channel, shuffle reduction over cin*k*k).
 * fir == NULL : packed [cout][k*k][cin]
 * fir != NULL : (4,4) blur taps; k must be 3; packed [4*cout][9][cin] polyphase weights
 *               of conv_transpose2d(stride 2) + Blur(pad (1,1)) (model.py:273-286). */
int vt_modulate_weight(void* out, const float* weight, const float* s, int cout, int cin,
                       int k, float scale, int demodulate, const float* fir,
                       int out_dtype, vt_stream stream);

/* y[r, o] = act(sum_i x[r,i] * W[o,i] * w_scale + b[o] * b_scale)
 * EqualLinear (model/stylegan/model.py:152-162), nn.Linear (dualstylegan.py:11,
 * vtoonify.py:114-119).  act: VT_ACT_NONE or VT_ACT_LRELU with (slope, gain).  fp32.
 * One wavefront per output, shuffle reduction. */
int vt_linear(float* y, int ld_y, const float* x, int ld_x, const float* W, const float* b,
              int rows, int in_dim, int out_dim, float w_scale, float b_scale, int act,
              float slope, float gain, vt_stream stream);

/* Batched forms of vt_linear / vt_modulate_weight: `items` is a HOST array, copied by value
 * into the launch (nothing to keep alive, hipGraph-capture safe).  All items of one call are
 * independent (no item may read another item's output).  The style path of a frame is ~50
 * GEMVs + 15 modulations; batching turns them into 5 launches. */
typedef struct vt_linear_item {
    float* y; const float* x; const float* W; const float* b; /* b may be NULL */
    int32_t ld_y, ld_x, rows, in_dim, out_dim, act;
    float w_scale, b_scale, slope, gain;
} vt_linear_item;
int vt_linear_batch(const vt_linear_item* items, int n_items, vt_stream stream);

typedef struct vt_modulate_item {
    void* out; const float* weight; const float* s; const float* fir; /* fir may be NULL */
    int32_t cout, cin, k, demodulate;
    float scale;
    int32_t reserved;
} vt_modulate_item;
int vt_modulate_weight_batch(const vt_modulate_item* items, int n_items, int out_dtype,
                             vt_stream stream);

/* PixelNorm (model/stylegan/model.py:17-18) on (rows, dim) fp32. */
int vt_pixel_norm(float* y, const float* x, int rows, int dim, vt_stream stream);

/* Style gate (ABI 4).  The style path of VToonify.forward (model/vtoonify.py:212-224; model/stylegan/model.py:259-267)
 * depends only on (W+ rows, d_s), which the video loop keeps for every frame but re-materialises per call
 * (`s_w.repeat(B,1,1)`, style_transfer.py:176).  vt_style_gate compares `fresh` with `cached` (n fp32 words, bitwise) on
 * the device -- no host synchronisation: flag[0] <- 1 and cached <- fresh when they differ or flag[1] (the host's "force"
 * word: new plan, another d_s) is set, else flag[0] <- 0; flag[1] <- 0.  The *_gated forms of the three style launches
 * take that flag word and return immediately when it is 0 (gate == NULL: always run). */
int vt_style_gate(int* flag, float* cached, const float* fresh, int n, vt_stream stream);
int vt_linear_batch_gated(const vt_linear_item* items, int n_items, const int* gate, vt_stream stream);
int vt_modulate_weight_batch_gated(const vt_modulate_item* items, int n_items, int out_dtype, const int* gate,
                                   vt_stream stream);
int vt_pixel_norm_gated(float* y, const float* x, int rows, int dim, const int* gate, vt_stream stream);

/* ---------------------------------------------------------------------------------
 * InstanceNorm / AdaIN (model/dualstylegan.py:6-21) on NHWC activations.
 * vt_instnorm_stats accumulates deterministic per-chunk (count, mean, M2) partials and
 * reduces them to scale/shift such that AdaIN(x) = x * scale[n][c] + shift[n][c]:
 *   scale = gamma * rstd,  shift = beta - gamma * rstd * mean   (eps 1e-5, biased var)
 * gamma/beta come from `style_gb` = Linear(style) laid out [n][2*C] (gamma first).
 * If `absdiff_other` != NULL the tensor normalised is cat[x, |x - other|] (Fusion,
 * model/vtoonify.py:125): C counts the channels of x, stats cover 2*C channels.
 * `partials` is caller workspace of vt_instnorm_ws_bytes(...) bytes.
 * --------------------------------------------------------------------------------- */
int64_t vt_instnorm_ws_bytes(int n, int hw, int c_total);
int vt_instnorm_stats(float* scale, float* shift, const void* x, int ld_x,
                      const void* absdiff_other, int ld_other, int n, int hw, int c,
                      const float* style_gb, int ld_gb, void* partials, int dtype,
                      vt_stream stream);
/* AdaIN(x) in two launches for SMALL planes (hw <= 16384 -- the 32x32-pixel trunk): statistics,
 * then one kernel whose workgroups each own one 16-byte channel vector: fold its chunk records,
 * normalise every pixel of the plane.  VT_ERR_UNSUPPORTED for larger tensors (use vt_instnorm_stats +
 * vt_affine_apply). */
int vt_instnorm_apply(void* out, int ld_out, const void* x, int ld_x, int n, int hw, int c,
                      const float* style_gb, int ld_gb, void* partials, int dtype, vt_stream stream);
/* The second launch of vt_instnorm_apply alone, on chunk records a conv already wrote
 * (vt_conv_desc.stats_part). */
int vt_instnorm_apply_stats(void* out, int ld_out, const void* x, int ld_x, int n, int hw, int c,
                            const float* style_gb, int ld_gb, const void* partials, int dtype,
                            vt_stream stream);
/* AdaIN(x) of a small plane (hw <= 4096: the trunk of frames up to 512x512 inputs) in ONE launch, statistics
 * included: one workgroup per (image, 16-byte channel vector) holds its slice in registers.  nn.InstanceNorm2d
 * (biased variance, eps 1e-5) + the style affine (model/stylegan/dualstylegan.py:6-21).  `out` may alias `x`.
 * absdiff_other != NULL: the AdaIN of cat[x, |x - other|] (Fusion.forward, model/vtoonify.py:125) into the 2c channels
 * of `out` (style_gb = [gamma 2c | beta 2c]; not in place).  VT_ERR_UNSUPPORTED for larger planes. */
int vt_instnorm_plane(void* out, int ld_out, const void* x, int ld_x, const void* absdiff_other, int ld_other,
                      int n, int hw, int c, const float* style_gb, int ld_gb, int dtype, vt_stream stream);
/* out[p][c] = x*scale+shift  (and the |x-other| half when absdiff_other != NULL). */
int vt_affine_apply(void* out, int ld_out, const void* x, int ld_x,
                    const void* absdiff_other, int ld_other, const float* scale,
                    const float* shift, int n, int hw, int c, int dtype, vt_stream stream);

/* Fusion glue (model/vtoonify.py:127, 259): out[p] = [skip(3) | zeros | f_e[p][:] * m[p]] with
 * per-pixel stride ld_out = header + c, header = 8, 16, ... channels (64 makes the consumer's
 * channel count a multiple of the direct-to-LDS K-step).  skip is NCHW fp32 (n,3,h,w); mask (n,h,w) fp32
 * (NULL = 1, the Toonify backbone, vtoonify.py:262). */
int vt_fusion_pack(void* out, int ld_out, const void* f_e, int ld_e, const float* mask,
                   const float* skip, int n, int hw, int c, int dtype, vt_stream stream);

/* ---------------------------------------------------------------------------------
 * Frame packing either side of VToonify.forward: the per-frame host work of the reference's
 * video loop (style_transfer.py:99-183) as two streaming kernels.
 *   vt_frame_pack    frames (n,h,w,3) uint8 [+ parsing (n,pc,h,w) fp32] -> x (n,3+pc,h,w) fp32:
 *                    x[c] = ((u8/255) - 0.5) / 0.5  (transforms.ToTensor + Normalize(0.5,0.5),
 *                    style_transfer.py:57-60,160), x[3+j] = parsing[j] * parsing_scale (x_p/16.,
 *                    style_transfer.py:174).  swap_rb != 0: frames are BGR as cv2 delivers them
 *                    (replaces cv2.cvtColor(frame, COLOR_BGR2RGB), style_transfer.py:114).
 *   vt_frame_unpack  image (n,3,H,W) fp32 -> frames (n,H,W,3) uint8:
 *                    u8 = (uint8)((clamp(y,-1,1) + 1.0) * 127.5)  (style_transfer.py:177 +
 *                    tensor2cv2, util.py:190-192); swap_rb != 0 writes BGR (tensor2cv2's cvtColor).
 * fp32 op order is the reference's; results are bit-exact against the numpy/torch formulas.
 * --------------------------------------------------------------------------------- */
int vt_frame_pack(float* x, const uint8_t* frames, int swap_rb, const float* parsing, int parsing_channels,
                  float parsing_scale, int n, int h, int w, vt_stream stream);
int vt_frame_unpack(uint8_t* frames, const float* image, int swap_rb, int n, int h, int w, vt_stream stream);

/* ---------------------------------------------------------------------------------
 * pSp style-encoder glue (GradualStyleEncoder, model/encoder/encoders/psp_encoders.py:35-116,
 * bottleneck_IR_SE / SEModule, helpers.py:53-119), NHWC activations.
 *   vt_channel_mean           AdaptiveAvgPool2d(1): mean[n][c] fp32 (deterministic two-stage
 *                             reduction; `partials` = vt_instnorm_ws_bytes(n, hw, c) bytes)
 *   vt_se_apply               out = res * gate[n][c] + shortcut[n, oy*s, ox*s, c]  (s > 1 is the
 *                             MaxPool2d(1, stride) shortcut, helpers.py:100-101)
 *   vt_upsample_bilinear_add  F.interpolate(x, (H,W), bilinear, align_corners=True) + y
 *                             (_upsample_add, psp_encoders.py:71-88)
 * ReLU / sigmoid of the SE bottleneck run as vt_linear activations (ReLU = VT_ACT_LRELU with
 * slope 0, gain 1; VT_ACT_SIGMOID).
 * --------------------------------------------------------------------------------- */
int vt_channel_mean(float* mean, const void* x, int ld_x, int n, int hw, int c, void* partials,
                    int dtype, vt_stream stream);
int vt_se_apply(void* out, const void* res, const float* gate, const void* shortcut, int n, int oh,
                int ow, int c, int sc_h, int sc_w, int sc_stride, int dtype, vt_stream stream);
int vt_upsample_bilinear_add(void* out, const void* x, const void* y, int n, int h, int w, int H,
                             int W, int c, int dtype, vt_stream stream);

/* ---------------------------------------------------------------------------------
 * Face-parsing network glue (BiSeNet, model/bisenet/model.py:92-254, resnet.py:58-80; the caller
 * of the hot path that produces 19 of its 22 input channels, style_transfer.py:171-174), NHWC.
 *   vt_maxpool2d         nn.MaxPool2d(k, stride, pad) (resnet.py:63: 3, 2, 1); out (n,oh,ow,c),
 *                        oh = (h + 2*pad - k)/stride + 1
 *   vt_gate_add_nearest  out[n,Y,X,c] = res[n,y,x,c] * gate[n][c] + add_vec[n][c] + add[n,y,x,c],
 *                        (y,x) = nearest-neighbour source of (Y,X) for an (out_h,out_w) output
 *                        (F.interpolate(mode='nearest'): floor(dst * in/out)); add_vec / add may be
 *                        NULL.  ARM attention product + "+ avg_up" / "+ feat32_up" + up-sampling
 *                        (model.py:78-85, 108-121); with gate = 1 + atten it is also the
 *                        FeatureFusionModule output feat * atten + feat (model.py:205-207)
 *   vt_resize_bilinear   F.interpolate(in, (virt_h, virt_w), mode='bilinear', align_corners) * mul on
 *                        planar fp32 input (n,c,h,w), evaluated at every `step`-th pixel:
 *                        out[Y][X] = resized[Y*step][X*step], (out_h,out_w) <= ceil(virt/step);
 *                        output planar fp32/bf16 (VT_OUT_NCHW) or NHWC with pixel stride ld_out
 *                        (channels >= c are left untouched).  aten's source-index arithmetic
 *                        (area_pixel_compute_source_index) in fp32.
 * --------------------------------------------------------------------------------- */
int vt_maxpool2d(void* out, const void* x, int n, int h, int w, int c, int k, int stride, int pad,
                 int dtype, vt_stream stream);
int vt_gate_add_nearest(void* out, const void* res, const float* gate, const float* add_vec,
                        const void* add, int n, int h, int w, int c, int out_h, int out_w, int dtype,
                        vt_stream stream);
int vt_resize_bilinear(void* out, int out_layout, int ld_out, int out_dtype, const float* in, int n,
                       int c, int h, int w, int virt_h, int virt_w, int align_corners, int step,
                       int out_h, int out_w, float mul, vt_stream stream);

/* ---------------------------------------------------------------------------------
 * RAFT correlation lookup -- replaces alt_cuda_corr.forward (model/raft/alt_cuda_corr/correlation.cpp:24-34,
 * correlation_kernel.cu:19-120; caller AlternateCorrBlock, model/raft/core/corr.py:63-91).  fp32.
 *   fmap1 (batch,h1,w1,c), fmap2 (batch,h2,w2,c) NHWC contiguous; coords (batch,1,h1,w1,2) = (x, y)
 *   target positions in fmap2; corr (batch,1,(2r+1)^2,h1,w1):
 *     corr[b,0,a+(2r+1)*k,p] = scale * bilinear_{(y-r+a, x-r+k)} <fmap1[b,p,:], fmap2[b,.,.,:]>, zero outside
 *   evaluated at coords * coord_scale (the pyramid level's coords / 2**i, corr.py:84) (the reference applies the
 *   1/sqrt(c) afterwards, corr.py:91: scale = 1, coord_scale = 1 give alt_cuda_corr.forward's exact output).
 *   vt_avgpool2x2: F.avg_pool2d(x, 2, stride=2) on NHWC fp32 (the feature pyramid, corr.py:68-71).
 * --------------------------------------------------------------------------------- */
int vt_corr_lookup(float* corr, const float* fmap1, const float* fmap2, const float* coords, int batch,
                   int h1, int w1, int h2, int w2, int c, int radius, float scale, float coord_scale,
                   vt_stream stream);
int vt_avgpool2x2(float* out, const float* x, int n, int h, int w, int c, vt_stream stream);

/* ---------------------------------------------------------------------------------
 * Flow warp and temporal fusion of parsing maps -- the loop body of the flicker-reduction pre-pass
 * (smooth_parsing_map.py:37-75 `warp`, :143-167).  fp32 NCHW planes, like the reference's tensors.
 *   vt_flow_warp: x (n,c,h,w), flo (n,2,h,w) = (dx, dy) -> out (n,c,h,w) = mask * grid_sample(x, grid + flo,
 *     bilinear, zeros, align_corners=True), mask (n,h,w) or NULL = 1 where grid_sample(ones) >= 0.9999 else 0
 *     (smooth_parsing_map.py:58-73; the reference returns the mask broadcast over c).
 *   vt_parsing_fuse: one centre frame against a window of `wn` frames (wn = 2*window+1, :155-166):
 *     frames (wn,3,h,w) = image2, center (3,h,w) = image1, parsing (wn,cp,h,w), flow (wn,2,h,w) = RAFT's flow_up,
 *     wt (wn) temporal weights (:140) -> fused (cp,h,w) = sum_j aligned_P_j w_j / sum_j w_j with
 *     w_j = wt_j * exp(-mean_c (aligned_I_j - image1)^2 / (2 sigma^2)) * mask_j, and for j = center_index
 *     aligned_P = parsing[center_index], w = wt_j (:161-163).  sigma = 0.2 in the reference.  cp <= 32.
 *     The caller finishes with Downsample = vt_upfirdn2d(down 2, pad (1,1)) (:108,167).
 * --------------------------------------------------------------------------------- */
int vt_flow_warp(float* out, float* mask, const float* x, const float* flo, int n, int c, int h, int w,
                 vt_stream stream);
int vt_parsing_fuse(float* fused, const float* frames, const float* center, const float* parsing,
                    const float* flow, const float* wt, int wn, int center_index, int cp, int h, int w,
                    float sigma, vt_stream stream);

/* ---------------------------------------------------------------------------------
 * RAFT glue (model/raft/core): everything between the convolutions of the optical-flow network.
 *   vt_eltwise2: out = a op b on `rows` pixel rows of `c` channels with independent row strides (elements);
 *     op 0 = a * b (r * h, update.py:48,54), 1 = a + b, 2 = relu(a + b) (ResidualBlock, extractor.py:53-60).
 *   vt_gru_blend: h = (1 - z) * h + z * q in place (update.py:49,55); z, q contiguous, h with row stride ld_h.
 *   vt_coords_from_flow: coords (n,1,h,w,2) = pixel grid + flow (n,2,h,w) (raft.py:58-66,121,127): the argument of
 *     vt_corr_lookup.
 *   vt_convex_upsample: RAFT.upsample_flow (raft.py:72-84): flow (n,2,h,w), mask (n,576,h,w) -> out (n,2,8h,8w).
 * --------------------------------------------------------------------------------- */
int vt_eltwise2(void* out, int ld_out, const void* a, int ld_a, const void* b, int ld_b, int64_t rows, int c,
                int op, int dtype, vt_stream stream);
int vt_gru_blend(void* h, int ld_h, const void* z, const void* q, int64_t rows, int c, int dtype, vt_stream stream);
int vt_coords_from_flow(float* coords, const float* flow, int n, int h, int w, vt_stream stream);
int vt_convex_upsample(float* out, const float* flow, const float* mask, int n, int h, int w, vt_stream stream);

/* Layout converters at the boundary (frames arrive NCHW fp32, model/vtoonify.py:210). */
int vt_nchw_to_nhwc(void* out, int ld_out, const void* in, int n, int c, int hw,
                    int in_dtype, int out_dtype, vt_stream stream);
int vt_nhwc_to_nchw(void* out, const void* in, int ld_in, int n, int c, int hw,
                    int in_dtype, int out_dtype, vt_stream stream);

/* MFMA fragment-layout self test: C = A(16xK) * B(KxN)^T on one wavefront, used by the
 * GPU test-suite to pin the 16x16x32 bf16 / 16x16x4 f32 lane maps on real hardware. */
int vt_mfma_selftest(float* c, const void* a, const void* b, int dtype, vt_stream stream);

#ifdef __cplusplus
}
#endif
#endif /* VTOONIFY_AMD_H */
