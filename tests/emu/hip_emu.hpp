// Minimal single-header HIP *emulation* for the authoring container (no GPU).
// TEST INFRASTRUCTURE ONLY -- never loaded by the product (vtoonify_amd/_lib.py).
//
// The kernel sources under vtoonify_amd/csrc/*.hip are compiled as plain C++ with this
// header force-included and -DVT_EMU.  Every HIP thread of a workgroup is a stackful
// coroutine (hand-written x86-64 context switch); __syncthreads() and the wavefront
// collectives (__shfl_xor, MFMA) are rendezvous points between coroutines.  Workgroups
// run in parallel on host threads; `__shared__` maps to `static thread_local`.
//
// MFMA builtins are emulated from the gfx950 lane maps documented in
// /opt/skills/guides/cdna_hip_programming.md section 3; the real lane maps are pinned on
// hardware by vt_mfma_selftest (tests/test_gpu_ops.py).
#pragma once
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include <functional>

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline __attribute__((always_inline))
#define __launch_bounds__(...)
#define __shared__ static thread_local

typedef void* hipStream_t;

struct dim3 {
    unsigned x, y, z;
    dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};

namespace emu {
struct ThreadInfo {
    dim3 tid, bid, bdim, gdim;
    int lane, wave;
};
ThreadInfo* cur();
void launch(dim3 grid, dim3 block, std::function<void()> body);
void syncthreads();
float shfl_xor_f(float v, int mask);
int shfl_xor_i(int v, int mask);
// wave-collective exchange: every live lane deposits `bytes` bytes; after the call
// `all` points to a [64][bytes] array valid until this lane's next collective.
const unsigned char* exchange(const void* mine, int bytes);
constexpr int kLaneStride = 64;  // bytes between lanes in the exchange buffer
template <typename F>
static inline const F& lane_frag(const unsigned char* base, int lane) {
    return *reinterpret_cast<const F*>(base + (size_t)lane * kLaneStride);
}
}  // namespace emu

#define threadIdx (emu::cur()->tid)
#define blockIdx (emu::cur()->bid)
#define blockDim (emu::cur()->bdim)
#define gridDim (emu::cur()->gdim)

static inline void __syncthreads() { emu::syncthreads(); }
static inline float __shfl_xor(float v, int mask, int width = 64) { (void)width; return emu::shfl_xor_f(v, mask); }
static inline int __shfl_xor(int v, int mask, int width = 64) { (void)width; return emu::shfl_xor_i(v, mask); }
static inline float rsqrtf(float x) { return 1.0f / sqrtf(x); }

template <typename T>
static inline T atomicAdd(T* p, T v) {
    // fp32 / int atomics on "global" memory shared by concurrently running workgroups
    T old = __atomic_load_n(reinterpret_cast<volatile T*>(p), __ATOMIC_RELAXED), des;
    do {
        des = old + v;
    } while (!__atomic_compare_exchange(p, &old, &des, false, __ATOMIC_RELAXED, __ATOMIC_RELAXED));
    return old;
}
template <>
inline float atomicAdd<float>(float* p, float v) {
    uint32_t* u = reinterpret_cast<uint32_t*>(p);
    uint32_t old = __atomic_load_n(u, __ATOMIC_RELAXED), des;
    float f;
    do {
        memcpy(&f, &old, 4);
        f += v;
        memcpy(&des, &f, 4);
    } while (!__atomic_compare_exchange_n(u, &old, des, false, __ATOMIC_RELAXED, __ATOMIC_RELAXED));
    memcpy(&f, &old, 4);
    return f;
}

// ---- MFMA emulation -------------------------------------------------------------
typedef float emu_f32x4 __attribute__((ext_vector_type(4)));
typedef short emu_bf16x8 __attribute__((ext_vector_type(8)));

// D = A(16x32 bf16) * B(32x16 bf16) + C.  lane l holds A[l&15][8*(l>>4)+j], B[8*(l>>4)+j][l&15];
// D/C: col = l&15, row = 4*(l>>4)+r.
static inline emu_f32x4 emu_mfma_f32_16x16x32_bf16(emu_bf16x8 a, emu_bf16x8 b, emu_f32x4 c) {
    struct Frag {
        emu_bf16x8 a, b;
    } mine{a, b};
    const unsigned char* all = emu::exchange(&mine, sizeof(Frag));
    const int lane = emu::cur()->lane;
    const int col = lane & 15;
    emu_f32x4 d = c;
    for (int r = 0; r < 4; ++r) {
        const int row = 4 * (lane >> 4) + r;
        float acc = d[r];
        for (int q = 0; q < 4; ++q)
            for (int j = 0; j < 8; ++j) {
                uint32_t ua = ((uint32_t)(uint16_t)emu::lane_frag<Frag>(all, row + 16 * q).a[j]) << 16;
                uint32_t ub = ((uint32_t)(uint16_t)emu::lane_frag<Frag>(all, col + 16 * q).b[j]) << 16;
                float fa, fb;
                memcpy(&fa, &ua, 4);
                memcpy(&fb, &ub, 4);
                acc += fa * fb;
            }
        d[r] = acc;
    }
    return d;
}
// D = A(16x4 f32) * B(4x16 f32) + C.  lane l holds A[l&15][l>>4], B[l>>4][l&15].
static inline emu_f32x4 emu_mfma_f32_16x16x4f32(float a, float b, emu_f32x4 c) {
    struct Frag {
        float a, b;
    } mine{a, b};
    const unsigned char* all = emu::exchange(&mine, sizeof(Frag));
    const int lane = emu::cur()->lane;
    const int col = lane & 15;
    emu_f32x4 d = c;
    for (int r = 0; r < 4; ++r) {
        const int row = 4 * (lane >> 4) + r;
        float acc = d[r];
        for (int q = 0; q < 4; ++q)
            acc = fmaf(emu::lane_frag<Frag>(all, row + 16 * q).a, emu::lane_frag<Frag>(all, col + 16 * q).b, acc);
        d[r] = acc;
    }
    return d;
}
