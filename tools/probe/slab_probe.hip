// probe (round 5): what bounds the 96 -> 96 5x5 conv -- the per-tap barrier, or the LDS-DMA stream that feeds it?
//
// conv_mfma_kernel<3,3,false> moves, per 256-pixel tile, 2 x 44.8 KB of patch and 50 x 10.7 KB of weight slabs through LDS-DMA: 627 KB
// per 118 MFLOP, 7.5 GB per B = 64 launch = 6.5 TB/s at 1.15 ms -- about what MI355X_MICROARCH.md gives an all-LDS-DMA stream
// (12-13 B/clk/CU).  The 48-channel kernel moves the same bytes per FLOP.  This probe runs the production tap loop (weights + pixels
// from LDS, double-buffered slab, one barrier per tap, real geometry, random operands, no epilogue) with the DMA streams switched on
// one by one, for the production workgroup (4 waves, 256 pixels, two per CU) and for an 8-wave workgroup that shares every slab
// between 512 pixels (16 x 32 tile, 20 x 36 patch, one per CU: half the slab bytes per FLOP).
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/slab_probe tools/probe/slab_probe.hip && /tmp/slab_probe
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));
typedef __attribute__((address_space(3))) void* lds_ptr_t;

constexpr int PSTRIDE = 112, TAPS = 25, CHUNKS = 2, KS = 3;
constexpr int SLAB = 11 * 1024;                      // BBYTES of the production kernel: (96 rows x 7 pieces + 63) / 64 KB-instructions
constexpr int WINSTR = 11;

__device__ __forceinline__ h8 lds_frag(const char* p) { return __builtin_bit_cast(h8, *(const uint4*)p); }

// DMA: 0 none (static slabs), 1 weight slabs, 2 weight slabs + the patch of every chunk
template <int WAVES, int DMA>
__global__ __launch_bounds__(WAVES * 64) void probe(const char* __restrict__ wgt, const char* __restrict__ act, unsigned act_bytes,
                                                    const uint4* __restrict__ fill, float* out, int tiles_per_wg) {
    constexpr int TW = WAVES == 4 ? 16 : 32, PW = TW + 4, NPIX = 20 * PW;
    constexpr int PATCH = NPIX * PSTRIDE;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, lhi = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    for (int i = tid; i < (PATCH + 2 * SLAB) / 16; i += WAVES * 64) ((uint4*)smem)[i] = fill[(i + blockIdx.x * 7) % 4096];
    __syncthreads();
    const char* patch = smem;
    char* slab = smem + PATCH;
    int abase[2];
    for (int mt = 0; mt < 2; ++mt) {
        const int m = wave * 64 + mt * 32 + l31, i = m / TW, j = m % TW;
        abase[mt] = (i * PW + j) * PSTRIDE + lhi * 16;
    }
    const int boff = l31 * PSTRIDE + lhi * 16;
    const __amdgpu_buffer_rsrc_t w_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)wgt, 0, (unsigned)(CHUNKS * TAPS * SLAB), 0x00020000);
    const __amdgpu_buffer_rsrc_t a_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)act, 0, act_bytes, 0x00020000);
    auto dma_slab = [&](const int buf, const int ct) {       // the 11 KB slab of (chunk, tap) ct, instruction i by wave i % WAVES
        for (int i = wave; i < WINSTR; i += WAVES)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(w_rsrc, (lds_ptr_t)(slab + buf * SLAB + i * 1024), 16, (unsigned)(lane * 16),
                                                     (unsigned)(ct * SLAB + i * 1024), 0, 0);
    };
    float sum = 0.f;
    for (int t = 0; t < tiles_per_wg; ++t) {
        f16v acc[2][3];
        for (int a = 0; a < 2; ++a) for (int b = 0; b < 3; ++b) for (int e = 0; e < 16; ++e) acc[a][b][e] = 0.f;
        const unsigned tile_off = (unsigned)(((blockIdx.x * tiles_per_wg + t) * 2) % 4000) * 65536u;      // a different 44-80 KB span per tile and chunk
        for (int cc = 0; cc < CHUNKS; ++cc) {
            if (cc || t) __syncthreads();
            if constexpr (DMA >= 2) {
                constexpr int PINSTR = (PATCH + 1023) / 1024;
                for (int i = wave; i < PINSTR; i += WAVES)
                    if (i * 1024 + lane * 16 < PATCH)
                        __builtin_amdgcn_raw_ptr_buffer_load_lds(a_rsrc, (lds_ptr_t)(smem + i * 1024), 16, (unsigned)(lane * 16),
                                                                 tile_off + (unsigned)(cc * 65536 / 2 + i * 1024), 0, 0);
            }
            if constexpr (DMA >= 1) dma_slab(0, cc * TAPS);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            for (int tap = 0; tap < TAPS; ++tap) {
                if constexpr (DMA >= 1) { if (tap + 1 < TAPS) dma_slab((tap + 1) & 1, cc * TAPS + tap + 1); }
                const int toff = ((tap / 5) * PW + tap % 5) * PSTRIDE;
                const char* bp = slab + (tap & 1) * SLAB + boff;
#pragma unroll
                for (int kk = 0; kk < KS; ++kk) {
                    h8 w[3], x[2];
#pragma unroll
                    for (int nt = 0; nt < 3; ++nt) w[nt] = lds_frag(bp + nt * 32 * PSTRIDE + kk * 32);
                    x[0] = lds_frag(patch + abase[0] + toff + kk * 32);
                    x[1] = lds_frag(patch + abase[1] + toff + kk * 32);
#pragma unroll
                    for (int nt = 0; nt < 3; ++nt) {
                        acc[0][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w[nt], x[0], acc[0][nt], 0, 0, 0);
                        acc[1][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w[nt], x[1], acc[1][nt], 0, 0, 0);
                    }
                }
                if constexpr (DMA >= 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();
            }
        }
        for (int a = 0; a < 2; ++a) for (int b = 0; b < 3; ++b) sum += acc[a][b][0] + acc[a][b][7];
    }
    if (sum == 12345.678f) out[blockIdx.x * blockDim.x + tid] = sum;
}

// RING: three slab buffers packed tightly (96 rows x 7 pieces = 10 752 B each), the DMA of tap t + 2 issued at the start of tap t, and NO
// workgroup-wide barrier inside the tap loop: every wave publishes two monotonic counters in LDS -- landed[w] = taps whose share of the
// slab DMA wave w has seen land, done[w] = taps wave w has finished reading -- and checks the minimum over the four waves before it
// reads a slab (all shares landed) and before it overwrites one (everybody is done with the tap that used it).  A wave may run one
// whole tap ahead of the slowest one; with a barrier it may not run ahead at all.
constexpr int SLABT = 96 * 7 * 16;                  // 10 752 B
template <int DMA>
__global__ __launch_bounds__(256) void probe_ring(const char* __restrict__ wgt, const char* __restrict__ act, unsigned act_bytes,
                                                  const uint4* __restrict__ fill, float* out, int tiles_per_wg) {
    constexpr int TW = 16, PW = 20, NPIX = 400, PATCH = NPIX * PSTRIDE;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, lhi = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    for (int i = tid; i < (PATCH + 3 * SLABT) / 16; i += 256) ((uint4*)smem)[i] = fill[(i + blockIdx.x * 7) % 4096];
    volatile int* flags = (volatile int*)(smem + PATCH + 3 * SLABT);      // landed[4], done[4]
    if (tid < 8) flags[tid] = 0;
    __syncthreads();
    const char* patch = smem;
    char* slab = smem + PATCH;
    int abase[2];
    for (int mt = 0; mt < 2; ++mt) {
        const int m = wave * 64 + mt * 32 + l31, i = m / TW, j = m % TW;
        abase[mt] = (i * PW + j) * PSTRIDE + lhi * 16;
    }
    const int boff = l31 * PSTRIDE + lhi * 16;
    const __amdgpu_buffer_rsrc_t w_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)wgt, 0, (unsigned)(CHUNKS * TAPS * SLAB), 0x00020000);
    const __amdgpu_buffer_rsrc_t a_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)act, 0, act_bytes, 0x00020000);
    // 11 DMA instructions per slab (the last one half full): wave w issues instructions w, w + 4, w + 8
    const int nshare = wave < 3 ? 3 : 2;
    auto dma_slab = [&](const int buf, const int ct) {
        for (int i = wave; i < WINSTR; i += 4)
            if (i * 1024 + lane * 16 < SLABT)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(w_rsrc, (lds_ptr_t)(slab + buf * SLABT + i * 1024), 16, (unsigned)(lane * 16),
                                                         (unsigned)(ct * SLAB + i * 1024), 0, 0);
    };
    auto wait_min = [&](const int base, const int want) {          // spin until min(flags[base .. base + 3]) >= want
        for (;;) {
            const int f0 = flags[base], f1 = flags[base + 1], f2 = flags[base + 2], f3 = flags[base + 3];
            if (min(min(f0, f1), min(f2, f3)) >= want) break;
            __builtin_amdgcn_s_sleep(1);
        }
    };
    float sum = 0.f;
    int g0 = 0;                                         // global tap counter of this workgroup (monotonic over chunks and tiles)
    for (int t = 0; t < tiles_per_wg; ++t) {
        f16v acc[2][3];
        for (int a = 0; a < 2; ++a) for (int b = 0; b < 3; ++b) for (int e = 0; e < 16; ++e) acc[a][b][e] = 0.f;
        const unsigned tile_off = (unsigned)(((blockIdx.x * tiles_per_wg + t) * 2) % 4000) * 65536u;
        for (int cc = 0; cc < CHUNKS; ++cc, g0 += TAPS) {
            __syncthreads();                            // everybody is done with the previous chunk's patch and slabs
            if constexpr (DMA >= 2) {
                constexpr int PINSTR = (PATCH + 1023) / 1024;
                for (int i = wave; i < PINSTR; i += 4)
                    if (i * 1024 + lane * 16 < PATCH)
                        __builtin_amdgcn_raw_ptr_buffer_load_lds(a_rsrc, (lds_ptr_t)(smem + i * 1024), 16, (unsigned)(lane * 16),
                                                                 tile_off + (unsigned)(cc * 65536 / 2 + i * 1024), 0, 0);
            }
            if constexpr (DMA >= 1) { dma_slab(g0 % 3, cc * TAPS); dma_slab((g0 + 1) % 3, cc * TAPS + 1); }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (lane == 0) { flags[wave] = g0 + 2; flags[4 + wave] = g0; }
            __syncthreads();
            for (int tap = 0; tap < TAPS; ++tap) {
                const int g = g0 + tap;
                if constexpr (DMA >= 1) {
                    if (tap >= 2) {
                        // my share of slab g (issued during tap g - 2) has landed when at most the share of slab g + 1 is outstanding
                        if (tap + 1 < TAPS) { if (nshare == 3) asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); }
                        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                        if (lane == 0) flags[wave] = g + 1;
                        wait_min(0, g + 1);             // every wave's share of slab g has landed
                    }
                    if (tap + 2 < TAPS) {
                        if (tap >= 1) wait_min(4, g);   // every wave has finished tap g - 1: its buffer may be overwritten
                        dma_slab((g + 2) % 3, cc * TAPS + tap + 2);
                    }
                }
                const int toff = ((tap / 5) * PW + tap % 5) * PSTRIDE;
                const char* bp = slab + (g % 3) * SLABT + boff;
#pragma unroll
                for (int kk = 0; kk < KS; ++kk) {
                    h8 w[3], x[2];
#pragma unroll
                    for (int nt = 0; nt < 3; ++nt) w[nt] = lds_frag(bp + nt * 32 * PSTRIDE + kk * 32);
                    x[0] = lds_frag(patch + abase[0] + toff + kk * 32);
                    x[1] = lds_frag(patch + abase[1] + toff + kk * 32);
#pragma unroll
                    for (int nt = 0; nt < 3; ++nt) {
                        acc[0][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w[nt], x[0], acc[0][nt], 0, 0, 0);
                        acc[1][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w[nt], x[1], acc[1][nt], 0, 0, 0);
                    }
                }
                if constexpr (DMA >= 1) {
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       // my fragment reads of this tap are done
                    if (lane == 0) flags[4 + wave] = g + 1;
                }
            }
        }
        for (int a = 0; a < 2; ++a) for (int b = 0; b < 3; ++b) sum += acc[a][b][0] + acc[a][b][7];
    }
    if (sum == 12345.678f) out[blockIdx.x * blockDim.x + tid] = sum;
}

template <int DMA>
static void run_ring(const char* name, const char* w, const char* act, unsigned act_bytes, const uint4* fill, float* out) {
    const int lds = 400 * PSTRIDE + 3 * SLABT + 64;
    hipFuncSetAttribute((const void*)probe_ring<DMA>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    const int nwg = 512, tiles = 12288 / nwg;
    for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL((probe_ring<DMA>), dim3(nwg), dim3(256), lds, 0, w, act, act_bytes, fill, out, tiles);
    hipDeviceSynchronize();
    hipEvent_t s, e;
    hipEventCreate(&s); hipEventCreate(&e);
    const int iters = 10;
    hipEventRecord(s);
    for (int rep = 0; rep < iters; ++rep) hipLaunchKernelGGL((probe_ring<DMA>), dim3(nwg), dim3(256), lds, 0, w, act, act_bytes, fill, out, tiles);
    hipEventRecord(e);
    hipEventSynchronize(e);
    float ms = 0.f;
    hipEventElapsedTime(&ms, s, e);
    ms /= iters;
    const double flops = 2.0 * 12288.0 * 256 * 96 * 96 * 25;
    printf("%-58s 2 WG/CU LDS %3d KB  %7.3f ms  %.3f of 2.5 PF\n", name, lds / 1024, ms, flops / ms / 1e9 / 2500.0);
}

template <int WAVES, int DMA>
static void run(const char* name, const char* w, const char* act, unsigned act_bytes, const uint4* fill, float* out, int wgs_per_cu) {
    constexpr int TW = WAVES == 4 ? 16 : 32, NPIX = 20 * (TW + 4);
    const int lds = NPIX * PSTRIDE + 2 * SLAB + 1024;
    hipFuncSetAttribute((const void*)probe<WAVES, DMA>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    const int nwg = 256 * wgs_per_cu, total_tiles = 12288 * 256 / (64 * WAVES), tiles = total_tiles / nwg;
    for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL((probe<WAVES, DMA>), dim3(nwg), dim3(WAVES * 64), lds, 0, w, act, act_bytes, fill, out, tiles);
    hipDeviceSynchronize();
    hipEvent_t s, e;
    hipEventCreate(&s); hipEventCreate(&e);
    const int iters = 10;
    hipEventRecord(s);
    for (int rep = 0; rep < iters; ++rep) hipLaunchKernelGGL((probe<WAVES, DMA>), dim3(nwg), dim3(WAVES * 64), lds, 0, w, act, act_bytes, fill, out, tiles);
    hipEventRecord(e);
    hipEventSynchronize(e);
    float ms = 0.f;
    hipEventElapsedTime(&ms, s, e);
    ms /= iters;
    const double flops = 2.0 * (double)(nwg * tiles) * (64 * WAVES) * 96 * 96 * 25;
    const double dma = (double)(nwg * tiles) * CHUNKS * ((DMA >= 1 ? (double)TAPS * SLAB : 0.0) + (DMA >= 2 ? (double)NPIX * PSTRIDE : 0.0));
    printf("%-58s %d WG/CU LDS %3d KB  %7.3f ms  %.3f of 2.5 PF  LDS-DMA %5.2f GB = %5.2f TB/s\n", name, wgs_per_cu, lds / 1024, ms,
           flops / ms / 1e9 / 2500.0, dma / 1e9, dma / ms / 1e9);
}

int main() {
    std::vector<uint16_t> h(4096 * 8);
    srand(1);
    for (auto& v : h) { _Float16 x = (_Float16)((rand() / (float)RAND_MAX) * 2.f - 1.f); v = *(uint16_t*)&x; }
    const size_t wbytes = (size_t)CHUNKS * TAPS * SLAB, abytes = (size_t)4096 * 65536;        // 550 KB of weights; 256 MB of activations
    std::vector<uint16_t> hw(wbytes / 2);
    for (auto& v : hw) { _Float16 x = (_Float16)(((rand() / (float)RAND_MAX) * 2.f - 1.f) * 0.05f); v = *(uint16_t*)&x; }
    uint4* dfill; char *dw, *da; float* dout;
    hipMalloc(&dfill, h.size() * 2); hipMalloc(&dw, wbytes); hipMalloc(&da, abytes); hipMalloc(&dout, 1 << 22);
    hipMemcpy(dfill, h.data(), h.size() * 2, hipMemcpyHostToDevice);
    hipMemcpy(dw, hw.data(), wbytes, hipMemcpyHostToDevice);
    for (size_t o = 0; o < abytes; o += h.size() * 2) hipMemcpy(da + o, h.data(), h.size() * 2, hipMemcpyHostToDevice);
    run<4, 0>("4 waves / 256 px, static slabs (barrier only)", dw, da, (unsigned)abytes, dfill, dout, 2);
    run<4, 1>("4 waves / 256 px, + weight-slab DMA", dw, da, (unsigned)abytes, dfill, dout, 2);
    run<4, 2>("4 waves / 256 px, + weight-slab and patch DMA", dw, da, (unsigned)abytes, dfill, dout, 2);
    run_ring<1>("4 waves / 256 px, RING of 3 slabs, flag sync, slab DMA", dw, da, (unsigned)abytes, dfill, dout);
    run_ring<2>("4 waves / 256 px, RING of 3 slabs, flag sync, slab+patch DMA", dw, da, (unsigned)abytes, dfill, dout);
    run<8, 0>("8 waves / 512 px, static slabs (barrier only)", dw, da, (unsigned)abytes, dfill, dout, 1);
    run<8, 1>("8 waves / 512 px, + weight-slab DMA", dw, da, (unsigned)abytes, dfill, dout, 1);
    run<8, 2>("8 waves / 512 px, + weight-slab and patch DMA", dw, da, (unsigned)abytes, dfill, dout, 1);
    return 0;
}
