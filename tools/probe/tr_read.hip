// Probe the lane/element mapping of ds_read_b64_tr_b16 on gfx950.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
__global__ void k(unsigned short* out, int pitch) {
    __shared__ __attribute__((aligned(16))) unsigned short lds[64 * 64];
    for (int i = threadIdx.x; i < 64 * 64; i += 64) lds[i] = (unsigned short)i;
    __syncthreads();
    const int l = threadIdx.x;
    // lane l of each 16-lane group g reads 8 bytes at row (l&15)/4 ... let every lane give: row = (l&15), col0 = 4*(l>>4)
    // (address fully per lane; we learn what each lane receives)
    const int row = l & 15, col0 = (l >> 4) * 4;
    unsigned addr = (unsigned)(uintptr_t)(lds) + (row * pitch + col0) * 2;
    uint2 v;
    asm volatile("ds_read_b64_tr_b16 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr) : "memory");
    out[l * 4 + 0] = v.x & 0xffff; out[l * 4 + 1] = v.x >> 16; out[l * 4 + 2] = v.y & 0xffff; out[l * 4 + 3] = v.y >> 16;
}
int main() {
    unsigned short* d; hipMalloc(&d, 64 * 4 * 2);
    const int pitch = 64;
    k<<<1, 64>>>(d, pitch);
    unsigned short h[256]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    printf("lane l supplied address (row=l&15, col0=4*(l>>4)) in a [64][%d] image whose value = row*%d+col\n", pitch, pitch);
    for (int l = 0; l < 64; ++l) {
        printf("lane %2d:", l);
        for (int e = 0; e < 4; ++e) printf("  (r%2d,c%2d)", h[l * 4 + e] / pitch, h[l * 4 + e] % pitch);
        printf("\n");
    }
    return 0;
}
