// probe: buffer_load_dwordx4 ... offen lds -- lane-linear destination, out-of-range lanes
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <vector>
__global__ void k(const char* src, unsigned nbytes, char* dst, const unsigned* offs, unsigned soff) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    ((uint4*)smem)[threadIdx.x] = make_uint4(0xAAAAAAAAu, 0xAAAAAAAAu, 0xAAAAAAAAu, 0xAAAAAAAAu);
    __syncthreads();
    __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, nbytes, 0x00020000);
    unsigned voff = offs[threadIdx.x];
    int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)(smem + wave * 1024), 16, voff, soff, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    ((uint4*)dst)[threadIdx.x] = ((uint4*)smem)[threadIdx.x];
}
int main() {
    const int n = 128, nbytes = 4096;
    std::vector<unsigned> src(nbytes / 4), offs(n), out(n * 4);
    for (int i = 0; i < nbytes / 4; ++i) src[i] = i;
    for (int i = 0; i < n; ++i) offs[i] = ((i * 37) % 256) * 16;
    offs[3] = nbytes;            // first byte out of range
    offs[5] = 0x80000000u;       // far out of range
    offs[7] = nbytes - 8;        // straddles the end
    offs[9] = 0xFFFFFFF0u;       // wraps with soffset?
    char *dsrc, *ddst; unsigned* doffs;
    hipMalloc(&dsrc, nbytes * 2); hipMalloc(&ddst, n * 16); hipMalloc(&doffs, n * 4);
    hipMemset(dsrc, 0x55, nbytes * 2);
    hipMemcpy(dsrc, src.data(), nbytes, hipMemcpyHostToDevice);
    hipMemcpy(doffs, offs.data(), n * 4, hipMemcpyHostToDevice);
    for (unsigned soff : {0u, 64u}) {
        hipLaunchKernelGGL(k, dim3(1), dim3(n), 4096, 0, dsrc, (unsigned)nbytes, ddst, doffs, soff);
        hipMemcpy(out.data(), ddst, n * 16, hipMemcpyDeviceToHost);
        int bad = 0;
        for (int i = 0; i < n; ++i) {
            if (i == 3 || i == 5 || i == 7 || i == 9) { printf("soff %u lane %d off %08x -> %08x %08x %08x %08x\n", soff, i, offs[i], out[4*i], out[4*i+1], out[4*i+2], out[4*i+3]); continue; }
            for (int e = 0; e < 4; ++e) if (out[4 * i + e] != (offs[i] + soff) / 4 + e) ++bad;
        }
        printf("soff %u: in-range mismatches %d\n", soff, bad);
    }
    return 0;
}
