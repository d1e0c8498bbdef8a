// Minimal reproducer probe for the round-4 wrong-image-row defect: does a VALU write of a VGPR that the
// PRECEDING packed-fp32 instruction reads as a source ever reach that instruction's result on gfx950?
//
//   hipcc --offload-arch=gfx950 -O2 -o tools/probe/bin/pk_war_probe tools/probe/pk_war_probe.hip ; tools/probe/bin/pk_war_probe
//
// In the failing kernel (conv_patch_kernel<bf16,8,64,2,2,1,3,2>, lean epilogue, SLP-packed ToRGB sums) every wrong value was the
// LOW half of a `v_pk_add_f32 vD, v[A:A+1], v[B:B+1]` whose register vA was overwritten by the next VALU instruction of the
// wave (`v_mov_b32 vA, ...` right behind it, or behind two scalar instructions and a taken branch) -- never a high half, never
// a sum left as scalar v_add_f32, and only while another kernel shared the CU.  The victim kernels below run that instruction
// pair (and controls: wait states in between, the write aimed at other operands, a scalar add) in a loop and count results
// that differ from the architectural value; the aggressor kernels keep the same SIMDs busy from a second stream.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

// SEQ: the instructions between the packed add and the read-back
//  0: v_pk_add ; v_mov src0.lo ; v_mov src0.hi                      (the failing shape)
//  1: v_pk_add ; s_nop 0 ; v_mov src0.lo ; v_mov src0.hi
//  2: v_pk_add ; s_nop 1 ; v_mov ...
//  3: v_pk_add ; s_and_b64 ; s_cbranch (not taken) ; v_mov ...       (the a=1 / a=3 shape: scalar instructions in between)
//  4: v_pk_add ; v_mov src1.lo ; v_mov src1.hi                      (write aimed at the other source)
//  5: v_pk_mul ; v_mov src0.lo ; v_mov src0.hi
//  6: v_add_f32 ; v_mov src0 (scalar control)
//  7: v_pk_add with op_sel:[0,1] op_sel_hi:[1,0] ; v_mov src1.hi ; v_mov src1.lo
//  8: v_pk_add ; v_pk_mov_b32 over src0 (packed overwrite)
//  9: v_pk_fma ; v_mov src0.lo ; v_mov src0.hi
template <int SEQ>
__global__ void __launch_bounds__(256) victim(unsigned long long* err, int iters, float* sample = nullptr) {
    const int lane = threadIdx.x & 63;
    unsigned long long e_lo = 0, e_hi = 0;
    for (int it = 0; it < iters; ++it) {
        const float a = (float)(lane + (it & 1023)), b = a + 0.25f, c = 0.5f + (float)(it & 7), d = c + 1.0f;
        const float poison = -12345.0f - (float)lane;
        float r0, r1;
        float x0 = a + c, x1 = b + d;   // architectural results
#define HEAD "v_mov_b32 v10, %2\n\tv_mov_b32 v11, %3\n\tv_mov_b32 v12, %4\n\tv_mov_b32 v13, %5\n\tv_mov_b32 v16, %6\n\tv_mov_b32 v17, %6\n\tv_mov_b32 v18, 1.0\n\tv_mov_b32 v19, 1.0\n\ts_nop 7\n\t"
#define TAIL "s_nop 7\n\tv_mov_b32 %0, v14\n\tv_mov_b32 %1, v15\n\t"
#define OPS : "=v"(r0), "=v"(r1) : "v"(a), "v"(b), "v"(c), "v"(d), "v"(poison) : "v10", "v11", "v12", "v13", "v14", "v15", "v16", "v17", "v18", "v19", "vcc"
        if constexpr (SEQ == 0) {
            asm volatile(HEAD "v_pk_add_f32 v[14:15], v[10:11], v[12:13]\n\tv_mov_b32 v10, v16\n\tv_mov_b32 v11, v17\n\t" TAIL OPS);
        } else if constexpr (SEQ == 1) {
            asm volatile(HEAD "v_pk_add_f32 v[14:15], v[10:11], v[12:13]\n\ts_nop 0\n\tv_mov_b32 v10, v16\n\tv_mov_b32 v11, v17\n\t" TAIL OPS);
        } else if constexpr (SEQ == 2) {
            asm volatile(HEAD "v_pk_add_f32 v[14:15], v[10:11], v[12:13]\n\ts_nop 1\n\tv_mov_b32 v10, v16\n\tv_mov_b32 v11, v17\n\t" TAIL OPS);
        } else if constexpr (SEQ == 3) {
            asm volatile(HEAD "v_pk_add_f32 v[14:15], v[10:11], v[12:13]\n\ts_and_b64 vcc, exec, 0\n\ts_cbranch_vccnz 0\n\t"
                              "v_mov_b32 v10, v16\n\tv_mov_b32 v11, v17\n\t" TAIL OPS);
        } else if constexpr (SEQ == 4) {
            asm volatile(HEAD "v_pk_add_f32 v[14:15], v[10:11], v[12:13]\n\tv_mov_b32 v12, v16\n\tv_mov_b32 v13, v17\n\t" TAIL OPS);
        } else if constexpr (SEQ == 5) {
            x0 = a * c, x1 = b * d;
            asm volatile(HEAD "v_pk_mul_f32 v[14:15], v[10:11], v[12:13]\n\tv_mov_b32 v10, v16\n\tv_mov_b32 v11, v17\n\t" TAIL OPS);
        } else if constexpr (SEQ == 6) {
            x1 = 0.0f;
            asm volatile(HEAD "v_add_f32 v14, v10, v12\n\tv_mov_b32 v10, v16\n\tv_mov_b32 v15, 0\n\t" TAIL OPS);
        } else if constexpr (SEQ == 7) {
            x0 = a + d, x1 = b + c;
            asm volatile(HEAD "v_pk_add_f32 v[14:15], v[10:11], v[12:13] op_sel:[0,1] op_sel_hi:[1,0]\n\tv_mov_b32 v13, v16\n\tv_mov_b32 v12, v17\n\t" TAIL OPS);
        } else if constexpr (SEQ == 8) {
            asm volatile(HEAD "v_pk_add_f32 v[14:15], v[10:11], v[12:13]\n\tv_pk_mov_b32 v[10:11], v[16:17], v[16:17]\n\t" TAIL OPS);
        } else if constexpr (SEQ == 9) {
            x0 = __builtin_fmaf(a, c, c), x1 = __builtin_fmaf(b, d, d);
            asm volatile(HEAD "v_pk_fma_f32 v[14:15], v[10:11], v[12:13], v[12:13]\n\tv_mov_b32 v10, v16\n\tv_mov_b32 v11, v17\n\t" TAIL OPS);
        } else if constexpr (SEQ == 10) {   // the packed add waits for its own src0 (produced by the packed multiply in front of it)
            asm volatile(HEAD "v_pk_mul_f32 v[10:11], v[10:11], v[18:19]\n\tv_pk_add_f32 v[14:15], v[10:11], v[12:13]\n\t"
                              "v_mov_b32 v10, v16\n\tv_mov_b32 v11, v17\n\t" TAIL OPS);
        } else if constexpr (SEQ == 11) {   // ... for its src1
            asm volatile(HEAD "v_pk_mul_f32 v[12:13], v[12:13], v[18:19]\n\tv_pk_add_f32 v[14:15], v[10:11], v[12:13]\n\t"
                              "v_mov_b32 v10, v16\n\tv_mov_b32 v11, v17\n\t" TAIL OPS);
        } else if constexpr (SEQ == 12) {   // the failing kernel's own tail: add, mov over the OTHER add's src1.lo, add, mov over src0.lo
            asm volatile(HEAD "v_pk_mul_f32 v[12:13], v[12:13], v[18:19]\n\tv_pk_add_f32 v[12:13], v[12:13], 0 op_sel_hi:[1,0]\n\t"
                              "v_mov_b32 v18, v16\n\tv_pk_add_f32 v[14:15], v[10:11], v[12:13]\n\t"
                              "v_mov_b32 v10, v16\n\tv_mov_b32 v11, v17\n\t" TAIL OPS);
        // ---- crossed lane selects: the LOW result takes the HIGH half of src1 (op_sel:[0,1]), the high result its low half ----
#define XADD "v_pk_add_f32 v[14:15], v[10:11], v[12:13] op_sel:[0,1] op_sel_hi:[1,0]\n\t"
        } else if constexpr (SEQ == 13) {   // nothing behind it
            x0 = a + d, x1 = b + c;
            asm volatile(HEAD XADD TAIL OPS);
        } else if constexpr (SEQ == 14) {   // the overwrites in the other order (src1.lo first)
            x0 = a + d, x1 = b + c;
            asm volatile(HEAD XADD "v_mov_b32 v12, v17\n\tv_mov_b32 v13, v16\n\t" TAIL OPS);
        } else if constexpr (SEQ == 15) {
            x0 = a + d, x1 = b + c;
            asm volatile(HEAD XADD "s_nop 0\n\tv_mov_b32 v13, v16\n\tv_mov_b32 v12, v17\n\t" TAIL OPS);
        } else if constexpr (SEQ == 16) {
            x0 = a + d, x1 = b + c;
            asm volatile(HEAD XADD "s_nop 1\n\tv_mov_b32 v13, v16\n\tv_mov_b32 v12, v17\n\t" TAIL OPS);
        } else if constexpr (SEQ == 17) {   // crossed on src0
            x0 = b + c, x1 = a + d;
            asm volatile(HEAD "v_pk_add_f32 v[14:15], v[10:11], v[12:13] op_sel:[1,0] op_sel_hi:[0,1]\n\tv_mov_b32 v11, v16\n\tv_mov_b32 v10, v17\n\t" TAIL OPS);
        } else if constexpr (SEQ == 18) {   // crossed multiply
            x0 = a * d, x1 = b * c;
            asm volatile(HEAD "v_pk_mul_f32 v[14:15], v[10:11], v[12:13] op_sel:[0,1] op_sel_hi:[1,0]\n\tv_mov_b32 v13, v16\n\tv_mov_b32 v12, v17\n\t" TAIL OPS);
        } else if constexpr (SEQ == 19) {   // another kind of writer behind it
            x0 = a + d, x1 = b + c;
            asm volatile(HEAD XADD "v_add_f32 v13, v16, v17\n\tv_add_f32 v12, v16, v17\n\t" TAIL OPS);
        } else if constexpr (SEQ == 20) {   // the crossed operand written by the instruction IN FRONT (read-after-write), nothing behind
            x0 = a + d, x1 = b + c;
            asm volatile(HEAD "v_mov_b32 v13, %5\n\t" XADD TAIL OPS);
        } else if constexpr (SEQ == 21) {   // broadcast select (both halves take src1.lo), src1.lo overwritten behind it
            x0 = a + c, x1 = b + c;
            asm volatile(HEAD "v_pk_add_f32 v[14:15], v[10:11], v[12:13] op_sel_hi:[1,0]\n\tv_mov_b32 v12, v16\n\tv_mov_b32 v13, v17\n\t" TAIL OPS);
        } else {                            // crossed add, an unrelated VALU instruction behind it (no overwrite)
            x0 = a + d, x1 = b + c;
            asm volatile(HEAD XADD "v_mov_b32 v18, v16\n\tv_mov_b32 v19, v17\n\t" TAIL OPS);
        }
        if (sample && r0 != x0 && e_lo == 0 && atomicCAS((int*)&sample[8], 0, 1) == 0) {
            sample[0] = r0, sample[1] = x0, sample[2] = a, sample[3] = b, sample[4] = c, sample[5] = d, sample[6] = poison, sample[7] = (float)it;
        }
        e_lo += (r0 != x0);
        e_hi += (r1 != x1);
    }
    if (e_lo) atomicAdd(&err[0], e_lo);
    if (e_hi) atomicAdd(&err[1], e_hi);
}

// aggressors: keep the SIMDs of every CU busy with one class of instruction from another stream
__global__ void __launch_bounds__(256) agg_valu(float* out, int iters) {
    float x = threadIdx.x * 1e-3f, y = 1.0001f;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int k = 0; k < 16; ++k) x = __builtin_fmaf(x, y, 1e-7f);
    }
    if (x == 123.456f) out[0] = x;
}
__global__ void __launch_bounds__(256) agg_pk(float* out, int iters) {
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 x = {threadIdx.x * 1e-3f, 0.5f}, y = {1.0001f, 0.9999f}, z = {1e-7f, 1e-7f};
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int k = 0; k < 16; ++k) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(x) : "v"(y), "v"(z));
    }
    if (x.x == 123.456f) out[0] = x.x + x.y;
}
typedef float f32x4v __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8v __attribute__((ext_vector_type(8)));
__global__ void __launch_bounds__(256) agg_mfma(float* out, int iters) {
    f32x4v acc = {0, 0, 0, 0};
    bf16x8v a, b;
    for (int k = 0; k < 8; ++k) a[k] = (__bf16)(threadIdx.x * 1e-3f), b[k] = (__bf16)1.0f;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int k = 0; k < 8; ++k) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc, 0, 0, 0);
    }
    if (acc[0] == 123.456f) out[0] = acc[0];
}
__global__ void __launch_bounds__(256) agg_mfma4(float* out, int iters) {   // four independent accumulators: back-to-back issue
    f32x4v acc[4];
    for (int k = 0; k < 4; ++k) acc[k] = f32x4v{0, 0, 0, 0};
    bf16x8v a, b;
    for (int k = 0; k < 8; ++k) a[k] = (__bf16)(threadIdx.x * 1e-3f), b[k] = (__bf16)1.0f;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int k = 0; k < 8; ++k) acc[k & 3] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[k & 3], 0, 0, 0);
    }
    if (acc[0][0] + acc[1][0] + acc[2][0] + acc[3][0] == 123.456f) out[0] = acc[0][0];
}
typedef float f32x16v __attribute__((ext_vector_type(16)));
__global__ void __launch_bounds__(256) agg_mfma32(float* out, int iters) {
    f32x16v acc;
    for (int k = 0; k < 16; ++k) acc[k] = 0.f;
    bf16x8v a, b;
    for (int k = 0; k < 8; ++k) a[k] = (__bf16)(threadIdx.x * 1e-3f), b[k] = (__bf16)1.0f;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int k = 0; k < 8; ++k) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
    }
    if (acc[0] == 123.456f) out[0] = acc[0];
}
__global__ void __launch_bounds__(256) agg_lds(float* out, int iters) {
    __shared__ float s[4096];
    for (int i = threadIdx.x; i < 4096; i += 256) s[i] = i;
    __syncthreads();
    float x = 0;
    int idx = threadIdx.x;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            x += s[(idx + k * 67) & 4095];
            x += __shfl_xor(x, 16, 64);
        }
        idx = (idx * 5 + 1) & 4095;
    }
    if (x == 123.456f) out[0] = x;
}
__global__ void __launch_bounds__(256) agg_mem(float* out, const float* in, size_t n, int iters) {
    float x = 0;
    size_t i0 = (size_t)blockIdx.x * 256 + threadIdx.x;
    for (int i = 0; i < iters; ++i)
        for (int k = 0; k < 8; ++k) x += in[(i0 + (size_t)(i * 8 + k) * 1048583u) % n];
    if (x == 123.456f) out[0] = x;
}

int main(int argc, char** argv) {
    const int reps = argc > 1 ? atoi(argv[1]) : 3;
    const bool only_x = argc > 2;   // second argument: only the crossed-select sequences and the matrix aggressors
    hipStream_t sa, sb;
    CK(hipStreamCreate(&sa));
    CK(hipStreamCreate(&sb));
    unsigned long long* err;
    CK(hipMalloc(&err, 16));
    float* out;
    CK(hipMalloc(&out, 64));
    float* smp;
    CK(hipMalloc(&smp, 64));
    const size_t n = (size_t)64 << 20;
    float* big;
    CK(hipMalloc(&big, n * 4));
    CK(hipMemset(big, 0, n * 4));
    const int vit = 200000;   // victim iterations per lane
    const char* agn[] = {"none", "valu", "pk_f32", "mfma16x16x32", "lds+bpermute", "memory", "victim(seq0)", "mfma32x32x16", "mfma16 x4 indep"};
    const char* sqn[] = {"pk_add; mov s0.lo; mov s0.hi", "pk_add; s_nop 0; mov..", "pk_add; s_nop 1; mov..", "pk_add; s_and; s_cbranch; mov..",
                         "pk_add; mov s1.lo; mov s1.hi", "pk_mul; mov s0.lo; mov s0.hi", "v_add_f32; mov s0 (scalar)",
                         "pk_add op_sel; mov s1.hi; mov s1.lo", "pk_add; v_pk_mov over s0", "pk_fma; mov s0.lo; mov s0.hi",
                         "pk_mul->s0; pk_add; mov s0.lo; mov s0.hi", "pk_mul->s1; pk_add; mov s0.lo; mov s0.hi",
                         "pk_mul; pk_add 0; mov; pk_add; mov s0..",
                         "Xadd (lo<-s1.hi); nothing", "Xadd; mov s1.lo; mov s1.hi", "Xadd; s_nop 0; mov s1.hi; mov s1.lo",
                         "Xadd; s_nop 1; mov s1.hi; mov s1.lo", "Xadd on s0; mov s0.hi; mov s0.lo", "Xmul; mov s1.hi; mov s1.lo",
                         "Xadd; v_add->s1.hi; v_add->s1.lo", "mov->s1.hi; Xadd; nothing", "add bcast s1.lo; mov s1.lo; mov s1.hi",
                         "Xadd; unrelated movs"};
    printf("# victim: 512 workgroups x 256 threads, %d iterations per lane; errors = results != architectural value (lo half, hi half)\n", vit);
    for (int ag = 0; ag < 9; ++ag) {
        if (only_x && ag != 0 && ag != 3 && ag != 7 && ag != 8) continue;
        for (int sq = 0; sq < 23; ++sq) {
            if (only_x && sq != 0 && sq != 7 && sq < 13) continue;
            unsigned long long tot[2] = {0, 0};
            for (int r = 0; r < reps; ++r) {
                CK(hipMemsetAsync(err, 0, 16, sa));
                CK(hipMemsetAsync(smp, 0, 64, sa));
                CK(hipStreamSynchronize(sa));
                const int ait = 400000;
                switch (ag) {
                    case 1: agg_valu<<<1024, 256, 0, sb>>>(out, ait / 4); break;
                    case 2: agg_pk<<<1024, 256, 0, sb>>>(out, ait / 4); break;
                    case 3: agg_mfma<<<1024, 256, 0, sb>>>(out, ait / 4); break;
                    case 4: agg_lds<<<1024, 256, 0, sb>>>(out, ait / 16); break;
                    case 5: agg_mem<<<1024, 256, 0, sb>>>(out, big, n, ait / 64); break;
                    case 6: victim<0><<<512, 256, 0, sb>>>(err + 0, vit); break;
                    case 7: agg_mfma32<<<1024, 256, 0, sb>>>(out, ait / 8); break;
                    case 8: agg_mfma4<<<1024, 256, 0, sb>>>(out, ait / 4); break;
                    default: break;
                }
#define V(S) case S: victim<S><<<512, 256, 0, sa>>>(err, vit, smp); break;
                switch (sq) { V(0) V(1) V(2) V(3) V(4) V(5) V(6) V(7) V(8) V(9) V(10) V(11) V(12) V(13) V(14) V(15) V(16) V(17) V(18) V(19) V(20) V(21) V(22) }
                CK(hipDeviceSynchronize());
                unsigned long long h[2];
                CK(hipMemcpy(h, err, 16, hipMemcpyDeviceToHost));
                tot[0] += h[0], tot[1] += h[1];
            }
            printf("aggressor %-16s | %-40s | lo errors %10llu  hi errors %10llu", agn[ag], sqn[sq], tot[0], tot[1]);
            if (tot[0]) {
                float h[8];
                CK(hipMemcpy(h, smp, 32, hipMemcpyDeviceToHost));
                if (h[1] != 0.0f) printf("   e.g. got %g want %g (a %g b %g c %g d %g poison %g, iteration %g)", h[0], h[1], h[2], h[3], h[4], h[5], h[6], h[7]);
            }
            printf("\n");
        }
    }
    return 0;
}
