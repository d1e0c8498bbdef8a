// Shared device helpers for the gfx950 kernels of libvtoonify_amd.so.
//
// Two build modes:
//   (default)  hipcc --offload-arch=gfx950        the product
//   -DVT_EMU   host clang++ with tests/emu/hip_emu.hpp force-included: the same kernel
//              sources run as coroutines on the CPU so index arithmetic, LDS tiling and
//              epilogues can be debugged in the GPU-less authoring container.  The
//              emulator is test infrastructure only; the Python loader never picks it.
#pragma once

#ifndef VT_EMU
#include <hip/hip_runtime.h>
#endif
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "../../include/vtoonify_amd.h"

// ---------------------------------------------------------------------------------
// error plumbing (host)
// ---------------------------------------------------------------------------------
void vt_set_error(const char* fmt, ...);
int vt_check_launch(const char* what);

#define VT_REQUIRE(cond, ...)            \
    do {                                 \
        if (!(cond)) {                   \
            vt_set_error(__VA_ARGS__);   \
            return VT_ERR_ARG;           \
        }                                \
    } while (0)

#ifdef VT_EMU
#define VT_LAUNCH(kern, grid, block, stream, ...) \
    emu::launch((grid), (block), [=]() { kern(__VA_ARGS__); })
#else
#define VT_LAUNCH(kern, grid, block, stream, ...) \
    hipLaunchKernelGGL(kern, (grid), (block), 0, (hipStream_t)(stream), __VA_ARGS__)
#endif

// ---------------------------------------------------------------------------------
// element types
// ---------------------------------------------------------------------------------
struct bf16_t {
    uint16_t v;
};
#ifdef VT_EMU
struct f16_t {
    uint16_t v;
};
#else
struct f16_t {
    _Float16 v;
};
#endif

__device__ __forceinline__ float vt_u2f(uint32_t u) {
    float f;
    memcpy(&f, &u, 4);
    return f;
}
__device__ __forceinline__ uint32_t vt_f2u(float f) {
    uint32_t u;
    memcpy(&u, &f, 4);
    return u;
}

__device__ __forceinline__ float bf16_bits_to_f32(uint32_t h) { return vt_u2f(h << 16); }
// round-to-nearest-even, NaN preserved
__device__ __forceinline__ uint32_t f32_to_bf16_bits(float f) {
    uint32_t u = vt_f2u(f);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (u >> 16) | 0x40u;
    u += 0x7fffu + ((u >> 16) & 1u);
    return u >> 16;
}

__device__ __forceinline__ float to_f32(float x) { return x; }
__device__ __forceinline__ float to_f32(bf16_t x) { return bf16_bits_to_f32(x.v); }
#ifdef VT_EMU
float emu_half_to_float(uint16_t h);
uint16_t emu_float_to_half(float f);
__device__ __forceinline__ float to_f32(f16_t x) { return emu_half_to_float(x.v); }
#else
__device__ __forceinline__ float to_f32(f16_t x) { return (float)x.v; }
#endif

template <typename T>
__device__ __forceinline__ T from_f32(float f);
template <>
__device__ __forceinline__ float from_f32<float>(float f) {
    return f;
}
template <>
__device__ __forceinline__ bf16_t from_f32<bf16_t>(float f) {
    bf16_t r;
#ifdef VT_EMU
    r.v = (uint16_t)f32_to_bf16_bits(f);
#else
    // v_cvt_pk_bf16_f32 (round-to-nearest-even, as pack_bf16x2 below): ONE instruction where the bit arithmetic of
    // f32_to_bf16_bits is 6 -- round 6 found 10 000 such sequences in the conv epilogues' element-wise paths and 8 per lane-vector
    // in the stand-alone fused_bias_act / upfirdn2d kernels, whose bf16 forms were slower than their fp32 forms on the same
    // element count (profiles/r06_op_bench_bf16.json).  Same bits for every finite value; a NaN comes out as the canonical quiet NaN.
    r.v = __builtin_bit_cast(uint16_t, (__bf16)f);
#endif
    return r;
}
template <>
__device__ __forceinline__ f16_t from_f32<f16_t>(float f) {
    f16_t r;
#ifdef VT_EMU
    r.v = emu_float_to_half(f);
#else
    r.v = (_Float16)f;
#endif
    return r;
}

// fp32 tensors whose products run on the bf16 matrix cores as three terms (hi*hi + hi*lo + lo*hi of the bf16 head /
// remainder split of both operands, fp32 accumulate): element type tag of the conv kernels' "f32x3" instances
// (vt_conv_desc.dtype == VT_F32X3).  In memory it IS a float.
struct f32x3_t {
    float v;
};
__device__ __forceinline__ float to_f32(f32x3_t x) { return x.v; }
template <>
__device__ __forceinline__ f32x3_t from_f32<f32x3_t>(float f) {
    f32x3_t r;
    r.v = f;
    return r;
}

// 16-byte vector of T
template <typename T>
struct Vec16 {
    static constexpr int N = 16 / sizeof(T);
    T e[N];
};

struct __attribute__((aligned(16))) u128 {
    uint32_t x, y, z, w;
};
struct __attribute__((aligned(8))) u64v {
    uint32_t x, y;
};

__device__ __forceinline__ u128 ld128(const void* p) { return *reinterpret_cast<const u128*>(p); }
__device__ __forceinline__ void st128(void* p, u128 v) { *reinterpret_cast<u128*>(p) = v; }
__device__ __forceinline__ u128 zero128() {
    u128 z;
    z.x = z.y = z.z = z.w = 0u;
    return z;
}

// unpack a 16-byte vector into 16/sizeof(T) floats
template <typename T>
__device__ __forceinline__ void unpack16(u128 v, float* f);
template <>
__device__ __forceinline__ void unpack16<float>(u128 v, float* f) {
    f[0] = vt_u2f(v.x);
    f[1] = vt_u2f(v.y);
    f[2] = vt_u2f(v.z);
    f[3] = vt_u2f(v.w);
}
template <>
__device__ __forceinline__ void unpack16<f32x3_t>(u128 v, float* f) {
    unpack16<float>(v, f);
}
template <>
__device__ __forceinline__ void unpack16<bf16_t>(u128 v, float* f) {
    f[0] = vt_u2f(v.x << 16);
    f[1] = vt_u2f(v.x & 0xffff0000u);
    f[2] = vt_u2f(v.y << 16);
    f[3] = vt_u2f(v.y & 0xffff0000u);
    f[4] = vt_u2f(v.z << 16);
    f[5] = vt_u2f(v.z & 0xffff0000u);
    f[6] = vt_u2f(v.w << 16);
    f[7] = vt_u2f(v.w & 0xffff0000u);
}
template <>
__device__ __forceinline__ void unpack16<f16_t>(u128 v, float* f) {
    f16_t h[8];
    memcpy(h, &v, 16);
#pragma unroll
    for (int i = 0; i < 8; ++i) f[i] = to_f32(h[i]);
}
template <typename T>
__device__ __forceinline__ u128 pack16(const float* f);
template <>
__device__ __forceinline__ u128 pack16<float>(const float* f) {
    u128 v;
    v.x = vt_f2u(f[0]);
    v.y = vt_f2u(f[1]);
    v.z = vt_f2u(f[2]);
    v.w = vt_f2u(f[3]);
    return v;
}
template <>
__device__ __forceinline__ u128 pack16<f32x3_t>(const float* f) {
    return pack16<float>(f);
}
// two fp32 -> packed bf16 pair, round-to-nearest-even (one v_cvt_pk_bf16_f32 on gfx950)
__device__ __forceinline__ uint32_t pack_bf16x2(float lo, float hi) {
#ifdef VT_EMU
    return f32_to_bf16_bits(lo) | (f32_to_bf16_bits(hi) << 16);
#else
    typedef __bf16 vt_bf16x2 __attribute__((ext_vector_type(2)));
    typedef float vt_f32x2 __attribute__((ext_vector_type(2)));
    const vt_f32x2 v = {lo, hi};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, vt_bf16x2));
#endif
}
template <>
__device__ __forceinline__ u128 pack16<bf16_t>(const float* f) {
    u128 v;
    v.x = pack_bf16x2(f[0], f[1]);
    v.y = pack_bf16x2(f[2], f[3]);
    v.z = pack_bf16x2(f[4], f[5]);
    v.w = pack_bf16x2(f[6], f[7]);
    return v;
}

// ---------------------------------------------------------------------------------
// buffer descriptor + direct-to-LDS 16-byte load (buffer_load_dwordx4 ... offen lds)
//   every lane fetches 16 bytes at base + voff + soff and the wave writes them lane-linear
//   to lds_wave_base + lane*16; bytes at or beyond `nrec` read as ZERO (range check is on
//   voff + soff).  The emulation build reproduces exactly that.
// ---------------------------------------------------------------------------------
// ---------------------------------------------------------------------------------
// inter-workgroup hand-off (split-K arrival counter): agent-scope release / acquire + a relaxed
// agent-scope ticket, exactly the placement-independent protocol of cdna_hip_programming.md
// section 5 ("in-launch split-K reduction") / section 6 Guideline 16.
// ---------------------------------------------------------------------------------
#ifdef VT_EMU
static inline void vt_release_agent() { __atomic_thread_fence(__ATOMIC_RELEASE); }
static inline void vt_acquire_agent() { __atomic_thread_fence(__ATOMIC_ACQUIRE); }
static inline void vt_drain_vmem() {}
static inline int vt_ticket_add(int* p, int v) { return __atomic_fetch_add(p, v, __ATOMIC_ACQ_REL); }
static inline void vt_ticket_store(int* p, int v) { __atomic_store_n(p, v, __ATOMIC_RELEASE); }
#else
__device__ __forceinline__ void vt_drain_vmem() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
__device__ __forceinline__ void vt_release_agent() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    vt_drain_vmem();  // restate the post-write-back wait where the compiler cannot drop it (ROCm 7.2)
}
__device__ __forceinline__ void vt_acquire_agent() { __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent"); }
__device__ __forceinline__ int vt_ticket_add(int* p, int v) {
    return __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void vt_ticket_store(int* p, int v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
#endif

#ifdef VT_EMU
static inline void vt_sched_fence() {}
template <int MASK, int N>
static inline void vt_sched_group() {}
#else
// scheduling group: the next N instructions of class MASK (0x008 MFMA, 0x100 LDS read, 0x200 LDS write, 0x020 VMEM
// read, 0x002 VALU) come next in the region's schedule -- a sequence of these pins the interleave of an unrolled loop
template <int MASK, int N>
__device__ __forceinline__ void vt_sched_group() { __builtin_amdgcn_sched_group_barrier(MASK, N, 0); }
// scheduling fence: the compiler may not move instructions across it
__device__ __forceinline__ void vt_sched_fence() { __builtin_amdgcn_sched_barrier(0); }
#endif

#ifdef VT_EMU
struct BufRsrc {
    const char* base;
    uint32_t nrec;
};
static inline BufRsrc vt_make_rsrc(const void* base, uint32_t nrec) { return BufRsrc{(const char*)base, nrec}; }
static inline void vt_glds16(const BufRsrc& r, void* lds_wave_base, uint32_t voff, uint32_t soff) {
    unsigned char* dst = (unsigned char*)lds_wave_base + (size_t)emu::cur()->lane * 16;
    const uint64_t off = (uint64_t)voff + soff;
    for (int d = 0; d < 4; ++d) {
        if (off + 4 * d + 4 <= r.nrec) memcpy(dst + 4 * d, r.base + off + 4 * d, 4);
        else memset(dst + 4 * d, 0, 4);
    }
}
// 4 bytes per lane: lane l's dword lands at lds_wave_base + 4 l
static inline void vt_glds4(const BufRsrc& r, void* lds_wave_base, uint32_t voff, uint32_t soff) {
    unsigned char* dst = (unsigned char*)lds_wave_base + (size_t)emu::cur()->lane * 4;
    const uint64_t off = (uint64_t)voff + soff;
    if (off + 4 <= r.nrec) memcpy(dst, r.base + off, 4);
    else memset(dst, 0, 4);
}
static inline void vt_glds_wait() {}
template <int N>
static inline void vt_glds_wait_n() {}
static inline void vt_lds_barrier() { emu::syncthreads(); }
static inline int vt_uniform(int v) { return v; }
#else
struct BufRsrc {
    __amdgpu_buffer_rsrc_t r;
};
__device__ __forceinline__ BufRsrc vt_make_rsrc(const void* base, uint32_t nrec) {
    BufRsrc b;
    // word3 0x00020000: raw buffer, 32-bit data format bits as used by the compiler's own buffer ops
    b.r = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, nrec, 0x00020000);
    return b;
}
__device__ __forceinline__ void vt_glds16(const BufRsrc& r, void* lds_wave_base, uint32_t voff, uint32_t soff) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r.r, (__attribute__((address_space(3))) void*)lds_wave_base, 16,
                                             voff, soff, 0, 0);
}
__device__ __forceinline__ void vt_glds4(const BufRsrc& r, void* lds_wave_base, uint32_t voff, uint32_t soff) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r.r, (__attribute__((address_space(3))) void*)lds_wave_base, 4,
                                             voff, soff, 0, 0);
}
__device__ __forceinline__ void vt_glds_wait() { __builtin_amdgcn_s_waitcnt(0x0f70); /* vmcnt(0), expcnt/lgkmcnt untouched */ }
// wait until at most N of this wave's vector-memory operations (LDS-DMA loads included) are
// outstanding; expcnt / lgkmcnt untouched.  gfx9 encoding: vmcnt = simm16[15:14]:[3:0].
template <int N>
__device__ __forceinline__ void vt_glds_wait_n() {
    static_assert(N >= 0 && N < 64, "vmcnt is 6 bits");
    __builtin_amdgcn_s_waitcnt((N & 15) | 0x0070 | 0x0f00 | ((N >> 4) << 14));
}
// workgroup barrier that does NOT drain in-flight LDS-DMA loads (a plain __syncthreads() waits
// vmcnt(0)): LDS reads of this wave are retired first, vector memory is left to the counted waits
__device__ __forceinline__ void vt_lds_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
}
__device__ __forceinline__ int vt_uniform(int v) { return __builtin_amdgcn_readfirstlane(v); }
#endif

// ---------------------------------------------------------------------------------
// Register-destination global loads the COMPILER DOES NOT SEE (inline asm), for operands that are
// streamed many sub-steps ahead of their use: hipcc sinks an ordinary load down to its first use and
// waits vmcnt(0) there (observed on the whole-K conv kernel: the 6-deep weight ring collapsed into
// load -> wait -> MFMA).  Contract (cdna_hip_programming.md 5.7, form iii): the destination registers
// count as written at the statement, so nothing may read them before vt_vmcnt_fence<N>() has run with
// N = the number of vector-memory operations this wave issued AFTER the load; the fence also stops the
// scheduler from hoisting register-only consumers (MFMAs) above the wait.
//   base: wave-uniform pointer (SGPR pair), voff: per-lane byte offset; loads 16 B at +0 and at +1024.
// ---------------------------------------------------------------------------------
#ifdef VT_EMU
template <int AUX = 0>
static inline void vt_gload16_pair_hidden(u128& a, u128& b, const void* base, uint32_t voff) {
    a = ld128((const unsigned char*)base + voff);
    b = ld128((const unsigned char*)base + voff + 1024);
}
template <int N>
static inline void vt_vmcnt_fence() {
    // lanes are coroutines here: the rendezvous stands for "the whole wave's loads have landed" (a wave's own
    // LDS-DMA data is read back by OTHER lanes of the wave without a workgroup barrier in between)
    const int dummy = 0;
    (void)emu::exchange(&dummy, (int)sizeof(dummy));
}
#else
typedef uint32_t vt_u32x4 __attribute__((ext_vector_type(4)));
// AUX = cache policy of the two loads (A/B knob of the whole-K conv): 0 default, 1 nt, 2 sc1, 3 sc0 sc1
template <int AUX = 0>
__device__ __forceinline__ void vt_gload16_pair_hidden(u128& a, u128& b, const void* base, uint32_t voff) {
    vt_u32x4 x, y;
    // s_nop 4: the base may have just come from a v_readfirstlane (VALU write of an SGPR -> VMEM read of it)
    if constexpr (AUX == 1)
        asm volatile("s_nop 4\n\tglobal_load_dwordx4 %0, %2, %3 nt\n\tglobal_load_dwordx4 %1, %2, %3 offset:1024 nt"
                     : "=&v"(x), "=&v"(y)
                     : "v"(voff), "s"(base));
    else if constexpr (AUX == 2)
        asm volatile("s_nop 4\n\tglobal_load_dwordx4 %0, %2, %3 sc1\n\tglobal_load_dwordx4 %1, %2, %3 offset:1024 sc1"
                     : "=&v"(x), "=&v"(y)
                     : "v"(voff), "s"(base));
    else if constexpr (AUX == 3)
        asm volatile("s_nop 4\n\tglobal_load_dwordx4 %0, %2, %3 sc0 sc1\n\tglobal_load_dwordx4 %1, %2, %3 offset:1024 sc0 sc1"
                     : "=&v"(x), "=&v"(y)
                     : "v"(voff), "s"(base));
    else
        asm volatile("s_nop 4\n\tglobal_load_dwordx4 %0, %2, %3\n\tglobal_load_dwordx4 %1, %2, %3 offset:1024"
                     : "=&v"(x), "=&v"(y)
                     : "v"(voff), "s"(base));
    a = __builtin_bit_cast(u128, x);
    b = __builtin_bit_cast(u128, y);
}
template <int N>
__device__ __forceinline__ void vt_vmcnt_fence() {
    static_assert(N >= 0 && N < 64, "vmcnt is 6 bits");
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N));
    __builtin_amdgcn_sched_barrier(0);
}
#endif
// The value of a hidden load's destination AFTER its vt_vmcnt_fence: an empty volatile asm that "rewrites" the registers.
// Volatile asm statements keep their order, so every consumer of the result is data-dependent on a statement that comes
// after the wait -- without it only the machine scheduler's barrier holds register-only consumers below the wait, and
// instruction selection may already have emitted pure VALU work on the registers above it (seen on the f32x3 whole-K
// kernel: its bf16 splits of the next weight pair moved above the counted wait and read registers the load had not
// written yet -- NaNs on a cold first launch, clean reruns).
#ifdef VT_EMU
static inline u128 vt_settled(const u128& v) { return v; }
#else
__device__ __forceinline__ u128 vt_settled(const u128& v) {
    vt_u32x4 x = __builtin_bit_cast(vt_u32x4, v);
    asm volatile("" : "+v"(x));
    return __builtin_bit_cast(u128, x);
}
#endif

// Identity the optimiser cannot see through: values derived from the result are recomputed where they are used
// instead of being hoisted out of a loop and kept live (LLVM hoists every loop-invariant address / predicate; in a
// kernel whose registers are full of resident operands those copies spill -- and a scratch reload is a VMEM round
// trip behind the LDS-DMA in flight).
#ifdef VT_EMU
static inline int vt_opaque(int v) { return v; }
#else
__device__ __forceinline__ int vt_opaque(int v) {
    asm volatile("" : "+v"(v));
    return v;
}
#endif

// 16-byte range-checked load into registers (compiler-visible): bytes at or beyond the descriptor's size -- and
// "negative" offsets, which wrap to huge unsigned ones -- read as zero, so a kernel may fetch whole aligned 16-byte
// chunks around a row whose first and last elements are not 16-byte aligned without touching foreign memory.
#ifdef VT_EMU
static inline u128 vt_bload16(const BufRsrc& r, uint32_t voff) {
    uint32_t w[4] = {0u, 0u, 0u, 0u};
    for (int d = 0; d < 4; ++d)
        if ((uint64_t)voff + 4 * d + 4 <= r.nrec) memcpy(&w[d], r.base + voff + 4 * d, 4);
    u128 v;
    v.x = w[0], v.y = w[1], v.z = w[2], v.w = w[3];
    return v;
}
#else
__device__ __forceinline__ u128 vt_bload16(const BufRsrc& r, uint32_t voff) {
    return __builtin_bit_cast(u128, __builtin_amdgcn_raw_buffer_load_b128(r.r, voff, 0, 0));
}
#endif
// ... and the range-checked 16-byte store: an offset at or beyond the descriptor's size (GLDS_OOB) writes nothing -- a
// predicated store without a branch, so a kernel's row step stays ONE basic block for the scheduler (conv_upblur_rows.hpp)
#ifdef VT_EMU
static inline void vt_bstore16(const BufRsrc& r, uint32_t voff, const u128& v) {
    if ((uint64_t)voff + 16 <= r.nrec) memcpy(const_cast<char*>(r.base) + voff, &v, 16);
}
#else
__device__ __forceinline__ void vt_bstore16(const BufRsrc& r, uint32_t voff, const u128& v) {
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(__attribute__((__vector_size__(4 * sizeof(unsigned int)))) unsigned int, v),
                                           r.r, voff, 0, 0);
}
#endif

// ---------------------------------------------------------------------------------
// Buffer loads / stores the COMPILER DOES NOT SEE (inline asm), for the per-step residual read and output write
// of a kernel that keeps LDS-DMA loads in flight across steps: beside a pending `buffer_load ... lds` hipcc waits
// vmcnt(0) at the first use of any ordinary load result (cdna_hip_programming.md, "pipelining across barriers"),
// which drains the prefetch every step.  Same contract as vt_gload16_pair_hidden: the caller counts every
// vector-memory operation of the wave and places vt_vmcnt_fence<N>() before the first read of a destination.
// Offsets are range-checked by the descriptor (raw buffer: a dword at or beyond `nrec` reads as zero / is not
// written), so a lane that has nothing to do passes GLDS-style out-of-range offsets and every wave issues the
// same number of operations.  DW = dwords per lane (1, 2 or 4).
// ---------------------------------------------------------------------------------
#ifdef VT_EMU
struct BufRaw {
    char* base;
    uint32_t nrec;
};
static inline BufRaw vt_make_raw(const void* base, uint32_t nrec) { return BufRaw{(char*)base, base ? nrec : 0u}; }
template <int DW>
static inline void vt_bload_hidden(u128& v, const BufRaw& r, uint32_t voff) {
    uint32_t w[4] = {0u, 0u, 0u, 0u};
    for (int d = 0; d < DW; ++d)
        if ((uint64_t)voff + 4 * d + 4 <= r.nrec) memcpy(&w[d], r.base + voff + 4 * d, 4);
    v.x = w[0], v.y = w[1], v.z = w[2], v.w = w[3];
}
template <int DW, int AUX = 0>
static inline void vt_bstore_hidden(const BufRaw& r, uint32_t voff, const u128& v) {
    const uint32_t w[4] = {v.x, v.y, v.z, v.w};
    for (int d = 0; d < DW; ++d)
        if ((uint64_t)voff + 4 * d + 4 <= r.nrec) memcpy(r.base + voff + 4 * d, &w[d], 4);
}
#else
struct BufRaw {
    vt_u32x4 v;   // V#: base[47:0], stride 0, num_records, raw 32-bit format (the words vt_make_rsrc builds)
};
__device__ __forceinline__ BufRaw vt_make_raw(const void* base, uint32_t nrec) {
    const uint64_t a = (uint64_t)base;
    BufRaw b;
    b.v = vt_u32x4{(uint32_t)a, (uint32_t)(a >> 32) & 0xffffu, base ? nrec : 0u, 0x00020000u};
    return b;
}
template <int DW>
__device__ __forceinline__ void vt_bload_hidden(u128& v, const BufRaw& r, uint32_t voff) {
    static_assert(DW == 1 || DW == 2 || DW == 4, "dwords per lane");
    if constexpr (DW == 4) {
        vt_u32x4 x;
        asm volatile("s_nop 4\n\tbuffer_load_dwordx4 %0, %1, %2, 0 offen" : "=&v"(x) : "v"(voff), "s"(r.v));
        v = __builtin_bit_cast(u128, x);
    } else if constexpr (DW == 2) {
        typedef uint32_t vt_u32x2 __attribute__((ext_vector_type(2)));
        vt_u32x2 x;
        asm volatile("s_nop 4\n\tbuffer_load_dwordx2 %0, %1, %2, 0 offen" : "=&v"(x) : "v"(voff), "s"(r.v));
        v.x = x.x, v.y = x.y, v.z = 0u, v.w = 0u;
    } else {
        uint32_t x;
        asm volatile("s_nop 4\n\tbuffer_load_dword %0, %1, %2, 0 offen" : "=&v"(x) : "v"(voff), "s"(r.v));
        v.x = x, v.y = 0u, v.z = 0u, v.w = 0u;
    }
}
// AUX = cache policy of the store (A/B knob): 0 default, 1 nt, 2 sc1, 3 sc0 sc1
template <int DW, int AUX = 0>
__device__ __forceinline__ void vt_bstore_hidden(const BufRaw& r, uint32_t voff, const u128& v) {
    static_assert(DW == 1 || DW == 2 || DW == 4, "dwords per lane");
#define VT_BST(OP_, X_)                                                                                                   \
    if constexpr (AUX == 1) asm volatile("s_nop 4\n\t" OP_ " %0, %1, %2, 0 offen nt" ::"v"(X_), "v"(voff), "s"(r.v) : "memory");          \
    else if constexpr (AUX == 2) asm volatile("s_nop 4\n\t" OP_ " %0, %1, %2, 0 offen sc1" ::"v"(X_), "v"(voff), "s"(r.v) : "memory");    \
    else if constexpr (AUX == 3) asm volatile("s_nop 4\n\t" OP_ " %0, %1, %2, 0 offen sc0 sc1" ::"v"(X_), "v"(voff), "s"(r.v) : "memory"); \
    else asm volatile("s_nop 4\n\t" OP_ " %0, %1, %2, 0 offen" ::"v"(X_), "v"(voff), "s"(r.v) : "memory");
    if constexpr (DW == 4) {
        const vt_u32x4 x = __builtin_bit_cast(vt_u32x4, v);
        VT_BST("buffer_store_dwordx4", x)
    } else if constexpr (DW == 2) {
        typedef uint32_t vt_u32x2 __attribute__((ext_vector_type(2)));
        const vt_u32x2 x = {v.x, v.y};
        VT_BST("buffer_store_dwordx2", x)
    } else {
        const uint32_t x = v.x;
        VT_BST("buffer_store_dword", x)
    }
#undef VT_BST
}
#endif

// wavefront (64 lanes) all-reduce sum via xor shuffles
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

static inline int vt_cdiv(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }
__device__ __forceinline__ int vt_cdiv_dev(int a, int b) { return (a + b - 1) / b; }

// Lanes of ONE wavefront exchange data through LDS (a wave-private region): the hardware runs the wave's LDS
// operations in order, so this is only a compiler scheduling fence; the host emulation, where lanes are
// coroutines, needs a real rendezvous.
#ifdef VT_EMU
static inline void vt_wave_sync() {
    const int dummy = 0;
    (void)emu::exchange(&dummy, (int)sizeof(dummy));
}
#else
__device__ __forceinline__ void vt_wave_sync() { __builtin_amdgcn_wave_barrier(); }
#endif

// ---- InstanceNorm chunk records, shared by norm_glue.hip and the conv kernels that emit them ----
struct StatRec {
    float x0, s1, s2;   // shift (the chunk's first pixel), sum(x-x0), sum((x-x0)^2)
};
// chunk geometry shared by host and device.  It depends on the plane size ONLY (never on the
// batch), so a frame's statistics -- hence its output bits -- are the same whether it is
// processed alone or inside a batch.
__host__ __device__ inline int stat_chunk_pixels(int hw) {
    // ~256 chunks per image, 64..4096 pixels each.  (Round 6: the floor was 16 -- a 64 x 64-pixel level of 1024 channels wrote and
    // re-read 12.6 MB of 12-byte records per 4-frame step for 33 MB of input, in workgroups of 16 pixels.)
    int px = (hw + 255) / 256;
    if (px < 64) px = 64;
    if (px > 4096) px = 4096;
    return px;
}
// statistics pass alone (norm_glue.hip); library-internal
__attribute__((visibility("hidden"))) int vt_internal_instnorm_partial(void* partials, const void* x, int ld_x,
                                                                       int n, int hw, int c, int dtype,
                                                                       vt_stream stream);
