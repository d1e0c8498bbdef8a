// How fast can ONE compute unit pull L2-resident bytes into its LDS?  (round 6)
//
// Every conv kernel of the frame that is not at the MFMA roof is described in DESIGN.md as "bound by the per-CU L2->LDS
// rate" with figures between 15 and 29 B/clk inferred from whole kernels.  This probe measures the rate itself: one
// workgroup per CU, NW waves, each wave keeps DEPTH 1-KiB LDS-DMA loads (buffer_load_dwordx4 ... lds) in flight over a
// source region that all workgroups of the launch share (so it is served by the L2s after the first pass), no compute.
//   pattern 0: 1 KiB contiguous per wave-instruction (a packed weight slab)
//   pattern 1: 8 rows of 128 B at a `stride`-byte pitch per wave-instruction (NHWC pixels / K-major weight rows: what the
//              conv loaders issue)
//   pattern 2: as 1, but into VGPRs (buffer_load_dwordx4 without lds): the same TA/TCP path without the LDS write
// Output: one line per configuration with GB/s per CU and B/clk at the measured shader clock (s_memtime ticks of wave 0
// of workgroup 0 against the wall time of the launch).
//
//   hipcc --offload-arch=gfx950 -O3 -o tools/probe/bin/ingest_probe tools/probe/ingest_probe.hip
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int N>
__device__ __forceinline__ void wait_vm() {
    __builtin_amdgcn_s_waitcnt((N & 15) | 0x0070 | 0x0f00 | ((N >> 4) << 14));
}

// region: `bytes_per_wg` of source per workgroup pass, the SAME for every workgroup whose index is equal mod `share`
// (share = 1: all workgroups read one region; share = 32: 32 distinct regions, one per CU of an XCD...)
template <int NW, int DEPTH, int PATTERN>
__global__ void __launch_bounds__(NW * 64)
ingest_kernel(const char* __restrict__ src, uint32_t region_bytes, int share, int iters, int stride, uint64_t* __restrict__ ticks,
              float* __restrict__ sink) {
    extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const char* base = src + (size_t)(blockIdx.x % share) * region_bytes;
    __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(base), 0, region_bytes, 0x00020000);
    // per-lane offset inside one 1-KiB piece
    uint32_t voff;
    if (PATTERN == 0) voff = lane * 16;
    else voff = (uint32_t)(lane >> 3) * (uint32_t)stride + (lane & 7) * 16;
    const uint32_t piece_span = PATTERN == 0 ? 1024u : 8u * (uint32_t)stride;   // source bytes covered by one instruction's rows
    const uint32_t pieces = region_bytes / piece_span;                          // per-region pieces (rows x 128 B each for pattern 1)
    uint64_t t0 = 0;
    if (blockIdx.x == 0 && tid == 0) t0 = __builtin_readcyclecounter();
    float acc = 0.f;
    uint32_t p = wave;   // piece index; waves interleave
    constexpr int SLOTS = 8;    // LDS ring of 1-KiB slots per wave (a slot may be rewritten while an older load to it is in flight: the data is not used)
    for (int it = 0; it < iters; it += DEPTH) {
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) {
            uint32_t off;
            if (PATTERN == 0) off = (p % pieces) * 1024u;
            else {   // the 128-byte column of the rows advances every `pieces` pieces
                const uint32_t col = (p / pieces) % ((uint32_t)stride / 128u);
                off = (p % pieces) * piece_span + col * 128u;
            }
            if (PATTERN == 2) {
                typedef __attribute__((__vector_size__(4 * sizeof(unsigned int)))) unsigned int u4;
                u4 v = __builtin_amdgcn_raw_buffer_load_b128(r, voff, off, 0);
                acc += __builtin_bit_cast(float, v[0] ^ v[3]);
            } else {
                __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)(smem + (wave * SLOTS + ((it + d) % SLOTS)) * 1024),
                                                         16, voff, off, 0, 0);
            }
            p += NW;
        }
        if (PATTERN != 2) wait_vm<DEPTH / 2>();   // keep at least DEPTH/2 in flight
    }
    if (PATTERN != 2) wait_vm<0>();
    __syncthreads();
    if (PATTERN != 2) acc = ((float*)smem)[tid];
    if (blockIdx.x == 0 && tid == 0) ticks[0] = __builtin_readcyclecounter() - t0;
    if (acc == 12345.678f) sink[blockIdx.x * blockDim.x + tid] = acc;
}

template <int NW, int DEPTH, int PATTERN>
int run(const char* src, uint32_t region, int share, int stride, int ncu, uint64_t* dticks, float* dsink, const char* label) {
    const int iters = 4096 / NW * 4;   // pieces per wave: 16 MiB per workgroup in total at any NW
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto k = ingest_kernel<NW, DEPTH, PATTERN>;
    CK(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, NW * 8 * 1024));
    for (int rep = 0; rep < 2; ++rep) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(k, dim3(ncu), dim3(NW * 64), NW * 8 * 1024, 0, src, region, share, iters, stride, dticks, dsink);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
    }
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    uint64_t ticks = 0;
    CK(hipMemcpy(&ticks, dticks, 8, hipMemcpyDeviceToHost));
    const double bytes = (double)iters * NW * 1024.0;
    const double gbs = bytes / (ms * 1e-3) / 1e9;
    // s_memtime / readcyclecounter ticks at a fixed 100 MHz on gfx9: report both the wall-derived rate and B/clk at 2.4 GHz
    printf("%-34s NW=%d depth=%2d region=%5u KiB share=%3d stride=%5d : %7.2f us  %6.1f GB/s per CU  %5.1f B/clk @2.4GHz  (chip %5.2f TB/s)  ticks=%llu\n",
           label, NW, DEPTH, region >> 10, share, stride, ms * 1e3, gbs, gbs / 2.4, gbs * ncu / 1e3, (unsigned long long)ticks);
    return 0;
}

int main(int argc, char** argv) {
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    const int ncu = prop.multiProcessorCount;
    printf("device %s, %d CUs, clock %d kHz\n", prop.gcnArchName, ncu, prop.clockRate);
    const size_t total = 256u << 20;
    char* src;
    uint64_t* dticks;
    float* dsink;
    CK(hipMalloc(&src, total));
    CK(hipMemset(src, 1, total));
    CK(hipMalloc(&dticks, 64));
    CK(hipMalloc(&dsink, (size_t)ncu * 512 * 4));
    // 1. contiguous slabs, shared by every workgroup (L2-resident after the first touch): waves x depth
#define R(NW, D, P, region, share, stride, label) if (run<NW, D, P>(src, region, share, stride, ncu, dticks, dsink, label)) return 1;
    R(1, 8, 0, 512u << 10, 1, 1024, "contig shared");
    R(2, 8, 0, 512u << 10, 1, 1024, "contig shared");
    R(4, 8, 0, 512u << 10, 1, 1024, "contig shared");
    R(8, 4, 0, 512u << 10, 1, 1024, "contig shared");
    R(8, 8, 0, 512u << 10, 1, 1024, "contig shared");
    R(8, 16, 0, 512u << 10, 1, 1024, "contig shared");
    R(16, 8, 0, 512u << 10, 1, 1024, "contig shared");
    // 2. the conv loaders' pattern: 8 rows x 128 B per instruction at a 1 KiB (512-channel pixel) / 9216 B (K-major weight row) pitch
    R(4, 8, 1, 512u << 10, 1, 1024, "rows128 shared");
    R(8, 8, 1, 512u << 10, 1, 1024, "rows128 shared");
    R(8, 16, 1, 512u << 10, 1, 1024, "rows128 shared");
    R(8, 8, 1, 4608u << 10, 1, 9216, "rows128 pitch 9216 shared");
    R(8, 8, 1, 512u << 10, 1, 256, "rows128 pitch 256 (128-ch pixel)");
    // 3. the same into VGPRs (no LDS write)
    R(8, 8, 2, 512u << 10, 1, 1024, "rows128 -> VGPR shared");
    R(8, 8, 2, 512u << 10, 1, 256, "rows128 -> VGPR pitch 256");
    // 4. what sharing buys: per-XCD-distinct regions (8), per-CU-distinct (256: HBM / MALL streaming)
    R(8, 8, 0, 512u << 10, 8, 1024, "contig 8 regions");
    R(8, 8, 0, 512u << 10, 256, 1024, "contig 256 regions (128 MiB)");
    R(8, 8, 1, 512u << 10, 16, 1024, "rows128 16 regions");
    // 5. a trunk conv's working set: 16 pixel tiles x 16 channel tiles, each CU its patch (331 KiB, shared by 16) + weights (295 KiB, shared by 16)
    R(8, 8, 1, 640u << 10, 16, 1024, "rows128 640 KiB x 16 regions");
    return 0;
}
