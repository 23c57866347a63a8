// LDS cycles of ds_read_b64_tr_b16 for the per-lane address patterns of wgrad16_kernel (32-byte pixel pitch) and
// wgrad_kernel (64-byte pitch), as a function of the byte shift a tap adds.  Prints s_memtime cycles per read (one wave
// and four waves per CU issuing back to back).   hipcc --offload-arch=gfx950 -O3 tr_conflict.hip -o tr_conflict
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#define NREAD 64
__global__ void k(const int* __restrict__ lane_addr, int shift, long long* out, int reps) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    for (int i = threadIdx.x; i < 32768 / 4; i += blockDim.x) ((unsigned*)lds)[i] = i;
    __syncthreads();
    const unsigned a = (unsigned)(uintptr_t)lds + (unsigned)lane_addr[threadIdx.x & 63] + (unsigned)shift;
    uint2 acc = make_uint2(0u, 0u);
    long long t0 = __builtin_readcyclecounter();
    for (int r = 0; r < reps; ++r) {
#pragma unroll
        for (int i = 0; i < NREAD / 8; ++i) {
            uint2 v0, v1, v2, v3, v4, v5, v6, v7;       // 8 reads in flight, ONE wait: throughput, not latency
            asm volatile("ds_read_b64_tr_b16 %0, %8\n\tds_read_b64_tr_b16 %1, %8 offset:1024\n\tds_read_b64_tr_b16 %2, %8 offset:2048\n\t"
                         "ds_read_b64_tr_b16 %3, %8 offset:3072\n\tds_read_b64_tr_b16 %4, %8 offset:4096\n\tds_read_b64_tr_b16 %5, %8 offset:5120\n\t"
                         "ds_read_b64_tr_b16 %6, %8 offset:6144\n\tds_read_b64_tr_b16 %7, %8 offset:7168\n\ts_waitcnt lgkmcnt(0)"
                         : "=&v"(v0), "=&v"(v1), "=&v"(v2), "=&v"(v3), "=&v"(v4), "=&v"(v5), "=&v"(v6), "=&v"(v7) : "v"(a));
            acc.x ^= v0.x ^ v1.x ^ v2.x ^ v3.x ^ v4.x ^ v5.x ^ v6.x ^ v7.x;
            acc.y ^= v0.y ^ v1.y ^ v2.y ^ v3.y ^ v4.y ^ v5.y ^ v6.y ^ v7.y;
        }
    }
    long long t1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
    if (acc.x == 0x12345678u) out[1000] = acc.y;
}
int main() {
    int h16[64], h64[64];
    for (int l = 0; l < 64; ++l) {
        const int g4 = l >> 4, s16 = l & 15;
        h16[l] = (4 * g4 + (s16 >> 2)) * 32 + (s16 & 3) * 8;                                  // wgrad16: krow * 32 + colb
        h64[l] = (8 * (g4 >> 1) + (s16 >> 2)) * 64 + (16 * (g4 & 1) + 4 * (s16 & 3)) * 2;      // wgrad: krow * 64 + chan_off
    }
    int *d16, *d64; long long* dout;
    hipMalloc(&d16, 256); hipMalloc(&d64, 256); hipMalloc(&dout, 8192 * 8);
    hipMemcpy(d16, h16, 256, hipMemcpyHostToDevice); hipMemcpy(d64, h64, 256, hipMemcpyHostToDevice);
    const int reps = 20;
    // one dispatch per (pattern, shift): under `rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS` the CSV rows are in this order
    for (int pat = 0; pat < 2; ++pat) {
        for (int s = 0; s < 24; ++s) {
            const int shift = s * (pat ? 64 : 32);
            k<<<1, 512, 65536>>>(pat ? d64 : d16, shift, dout, reps);
            hipDeviceSynchronize();
            long long t; hipMemcpy(&t, dout, 8, hipMemcpyDeviceToHost);
            printf("pitch %d shift %2d pixels: %5.2f cycles per wave-read (8 waves)\n", pat ? 64 : 32, s, (double)t / (reps * NREAD));
        }
    }
    return 0;
}
