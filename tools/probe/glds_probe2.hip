// Hardware probe, round 3 (weight-stationary whole-K conv, conv_fullkw.hpp): what may the LDS base of a
// buffer_load ... lds be, and does the DMA honour EXEC?
//   hipcc --offload-arch=gfx950 -O3 glds_probe2.hip -o /tmp/glds_probe2 && /tmp/glds_probe2
//   A  LDS base at 16-byte granularity (6656 = row 52 of a 128-byte-row patch, 7680 = a 60-row slot, 7696)
//   B  the same load under a partial EXEC mask (lanes < 32 only): are the other lanes' 16-byte slots written?
//   C  a kernel that declares all 160 KiB of LDS statically launches
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <vector>

__global__ void probe(const uint32_t* src, uint32_t* out, unsigned base_bytes, int half) {
    __shared__ __attribute__((aligned(1024))) uint32_t lds[8192];
    const int tid = threadIdx.x;
    for (int i = tid; i < 8192; i += 64) lds[i] = 0xdeadbeefu;
    __syncthreads();
    __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, 1u << 20, 0x00020000);
    unsigned char* b = reinterpret_cast<unsigned char*>(lds) + base_bytes;
    if (!half || tid < 32)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)b, 16, tid * 16, 0, 0, 0);
    __builtin_amdgcn_s_waitcnt(0);
    __syncthreads();
    for (int i = tid; i < 8192; i += 64) out[i] = lds[i];
}

__global__ void __launch_bounds__(512) big_lds(uint32_t* out) {
    __shared__ __attribute__((aligned(1024))) uint32_t lds[163840 / 4];
    for (int i = threadIdx.x; i < 163840 / 4; i += 512) lds[i] = i;
    __syncthreads();
    uint32_t s = 0;
    for (int i = threadIdx.x; i < 163840 / 4; i += 512) s += lds[163840 / 4 - 1 - i];
    out[blockIdx.x * 512 + threadIdx.x] = s;
}

int main() {
    const int N = 1 << 18;
    std::vector<uint32_t> h(N), res(8192);
    for (int i = 0; i < N; ++i) h[i] = i;
    uint32_t *d, *o;
    hipMalloc(&d, N * 4);
    hipMalloc(&o, 1 << 20);
    hipMemcpy(d, h.data(), N * 4, hipMemcpyHostToDevice);
    for (int half = 0; half < 2; ++half)
        for (unsigned base : {1024u, 6656u, 7680u, 7696u}) {
            probe<<<1, 64>>>(d, o, base, half);
            hipError_t e = hipDeviceSynchronize();
            hipMemcpy(res.data(), o, 8192 * 4, hipMemcpyDeviceToHost);
            int good = 0, bad_in = 0, touched_out = 0, masked_written = 0;
            const int lanes = 64;
            for (int i = 0; i < 8192; ++i) {
                const int rel = i - (int)base / 4;
                if (rel >= 0 && rel < lanes * 4) {
                    const int lane = rel / 4;
                    if (half && lane >= 32) {
                        if (res[i] != 0xdeadbeefu) masked_written++;
                    } else if (res[i] == (uint32_t)rel) good++;
                    else bad_in++;
                } else if (res[i] != 0xdeadbeefu) touched_out++;
            }
            printf("half=%d base=%5u: rc=%d lane-linear dwords ok=%d wrong=%d outside-window-modified=%d masked-lanes-written=%d\n",
                   half, base, (int)e, good, bad_in, touched_out, masked_written);
        }
    big_lds<<<256, 512>>>(o);
    hipError_t e = hipDeviceSynchronize();
    printf("160 KiB static LDS launch: %s (last error %s)\n", e == hipSuccess ? "ok" : "FAILED", hipGetErrorString(hipGetLastError()));
    return 0;
}
