// Hardware probe for buffer_load ... lds semantics on gfx950 (used to design conv_igemm v2).
//   hipcc --offload-arch=gfx950 -O3 glds_probe.hip -o glds_probe && ./glds_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <vector>

__global__ void probe(const uint32_t* src, uint32_t* out, const uint32_t* voffs, unsigned soff, unsigned nrec) {
    __shared__ __attribute__((aligned(16))) uint32_t lds[2048];
    const int tid = threadIdx.x;
    for (int i = tid; i < 2048; i += 64) lds[i] = 0xdeadbeefu;
    __syncthreads();
    __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, nrec, 0x00020000);
    // one wave: 64 lanes x 16 B -> lds[256 .. 512) dwords expected lane-linear
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)(lds + 256), 16, voffs[tid], soff, 0, 0);
    __builtin_amdgcn_s_waitcnt(0);
    __syncthreads();
    for (int i = tid; i < 2048; i += 64) out[i] = lds[i];
}

int main() {
    const int N = 1 << 16;
    std::vector<uint32_t> h(N);
    for (int i = 0; i < N; ++i) h[i] = i;  // dword index as value
    uint32_t *d, *o, *v;
    hipMalloc(&d, N * 4); hipMalloc(&o, 2048 * 4); hipMalloc(&v, 64 * 4);
    hipMemcpy(d, h.data(), N * 4, hipMemcpyHostToDevice);
    std::vector<uint32_t> vo(64), res(2048);
    // lanes: reversed order of 16-B chunks; lanes 5, 17 out of range (sentinel); lane 40 just past num_records
    const unsigned nrec = 4096;  // bytes
    for (int l = 0; l < 64; ++l) vo[l] = (63 - l) * 16;
    vo[5] = 0x80000000u; vo[17] = 0x80000000u; vo[40] = nrec; vo[41] = nrec - 16; vo[42] = nrec - 8;
    hipMemcpy(v, vo.data(), 256, hipMemcpyHostToDevice);
    for (unsigned soff : {0u, 64u, 8192u}) {
        probe<<<1, 64>>>(d, o, v, soff, nrec);
        hipMemcpy(res.data(), o, 2048 * 4, hipMemcpyDeviceToHost);
        printf("soff=%u\n", soff);
        int untouched_bad = 0;
        for (int i = 0; i < 2048; ++i) if ((i < 256 || i >= 512) && res[i] != 0xdeadbeefu) untouched_bad++;
        printf("  outside-window modified: %d\n", untouched_bad);
        for (int l : {0, 1, 5, 17, 40, 41, 42, 63}) {
            printf("  lane %2d voff=%08x -> lds dwords %08x %08x %08x %08x (expect src dword %u if in range)\n", l, vo[l],
                   res[256 + l * 4], res[256 + l * 4 + 1], res[256 + l * 4 + 2], res[256 + l * 4 + 3], (vo[l] + soff) / 4);
        }
    }
    return 0;
}
