// bn.hip -- training-mode BatchNorm2d + activation on bf16 NHWC activations (gfx950, HBM-bound).
//
// Reference: nn.BatchNorm2d inside Conv2dBlock / ConvBlock / DownConvBlock / UpConvBlock
// (M1/networks.py:38-39, M2/networks.py:38-39,107-108,137-138), torch defaults eps=1e-5,
// momentum=0.1: normalise with the biased batch variance, update running_var with the unbiased
// one.  The conv kernel writes the raw (pre-BN) output; these kernels do
//   stats (two-stage, deterministic)  ->  finalize  ->  apply (+ReLU/PReLU).
// Each thread moves 16-byte channel runs (8 bf16), so every pass streams the tensor once at
// full line width.
#include "sos_common.h"
#include <stdlib.h>

#ifndef SOS_BN_BWD_ALIGNED
#define SOS_BN_BWD_ALIGNED 0      // 1: backward passes with pixel lanes spanning whole 128-byte lines like the forward ones -- measured (round 3,
                                  // tools/probe/archive/bn_ab.sh): 96 ch 570 vs 562 us, 48 ch 275 vs 282 us for reduce + apply: inside the noise, not adopted
#endif
#ifndef SOS_BN_RED_U
#define SOS_BN_RED_U 2            // 16-byte loads in flight per thread and operand in bn_bwd_reduce (4: 0-7 % slower)
#endif

struct View {
    bf16_t* ptr;
    long long npix;
    int row, c_off, C, x3;
    long long third;
};

static inline View to_view(const sos_view* v) {
    View o;
    o.ptr = (bf16_t*)v->ptr; o.npix = v->npix; o.row = v->row; o.c_off = v->c_off; o.C = v->C; o.x3 = v->x3;
    o.third = v->third;
    return o;
}

// pixel lanes of a 256-thread workgroup for CG 8-channel groups: the largest count whose pixels span whole 128-byte lines
// (96 channels: 20 pixels x 192 B = 30 lines instead of 21 x 192 B = 31.5 -- neighbouring workgroups then never share a
// line), never fewer than 3/4 of the lanes the threads allow
__host__ __device__ static inline int bn_pl(int CG, int row_elems) {
    const int PL = 256 / CG;
    for (int q = PL; q >= 1 && 4 * q >= 3 * PL; --q)
        if ((q * row_elems * 2) % 128 == 0) return q;
    return PL;
}

static int check_view(const sos_view* v, const char* what) {
    if (!v || !v->ptr || v->npix < 1 || v->C < 1 || v->C > 2048 || v->row % 8 || v->c_off % 8 || (v->x3 && v->third % 8)) {
        sos_set_error("%s: bad view", what);
        return SOS_EINVAL;
    }
    return SOS_OK;
}

typedef unsigned bn_u32x4 __attribute__((ext_vector_type(4)));
// streaming accesses (every byte is touched once per pass and the tensors are far larger than L2): the
// non-temporal hint is worth 5-8 % of the HBM rate of these kernels (bn_bwd 4.7 -> 5.1 TB/s)
__device__ __forceinline__ uint4 ld16(const bf16_t* p) {
    const bn_u32x4 v = __builtin_nontemporal_load((const bn_u32x4*)p);
    return make_uint4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ void st16(bf16_t* p, const uint4 v) {
    const bn_u32x4 w = {v.x, v.y, v.z, v.w};
    __builtin_nontemporal_store(w, (bn_u32x4*)p);
}

__device__ __forceinline__ void load8(const View& v, long long pix, int c8, float (&f)[8]) {
    const bf16_t* p = v.ptr + pix * v.row + v.c_off + c8;
    const uint4 h = ld16(p);
    const unsigned hw[4] = {h.x, h.y, h.z, h.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        f[2 * i] = sos_lo2f(hw[i]);
        f[2 * i + 1] = sos_hi2f(hw[i]);
    }
    if (v.x3) {
        const uint4 l = *(const uint4*)(p + 2 * v.third);
        const unsigned lw[4] = {l.x, l.y, l.z, l.w};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            f[2 * i] += sos_lo2f(lw[i]);
            f[2 * i + 1] += sos_hi2f(lw[i]);
        }
    }
}

// U pixels (pix, pix + step, ...) of one 8-channel group: all loads are issued before the first conversion, so a
// thread keeps U (2U in hi|hi|lo mode) 16-byte requests in flight.  Pixels past the end are clamped (callers skip them).
template <int BN_U>
__device__ __forceinline__ void load8u(const View& v, long long pix, long long step, int c8, float (&f)[BN_U][8]) {
    uint4 h[BN_U], l[BN_U];
#pragma unroll
    for (int u = 0; u < BN_U; ++u) {
        const long long pp = pix + u * step < v.npix ? pix + u * step : v.npix - 1;
        const bf16_t* p = v.ptr + pp * v.row + v.c_off + c8;
        h[u] = ld16(p);
        if (v.x3) l[u] = *(const uint4*)(p + 2 * v.third);
    }
#pragma unroll
    for (int u = 0; u < BN_U; ++u) {
        const unsigned hw[4] = {h[u].x, h[u].y, h[u].z, h[u].w};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            f[u][2 * i] = sos_lo2f(hw[i]);
            f[u][2 * i + 1] = sos_hi2f(hw[i]);
        }
        if (v.x3) {
            const unsigned lw[4] = {l[u].x, l[u].y, l[u].z, l[u].w};
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                f[u][2 * i] += sos_lo2f(lw[i]);
                f[u][2 * i + 1] += sos_hi2f(lw[i]);
            }
        }
    }
}

__device__ __forceinline__ void store8(const View& v, long long pix, int c8, const float (&f)[8]) {
    bf16_t* p = v.ptr + pix * v.row + v.c_off + c8;
    unsigned hw[4], lw[4];
    if (!v.x3) {        // v_cvt_pk_bf16_f32: one instruction per pair instead of the integer rounding sequence
        st16(p, make_uint4(pack2bf(f[0], f[1]), pack2bf(f[2], f[3]), pack2bf(f[4], f[5]), pack2bf(f[6], f[7])));
        return;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const bf16_t h0 = f2bf(f[2 * i]), h1 = f2bf(f[2 * i + 1]);
        hw[i] = (unsigned)h0 | ((unsigned)h1 << 16);
        if (v.x3) lw[i] = (unsigned)f2bf(f[2 * i] - bf2f(h0)) | ((unsigned)f2bf(f[2 * i + 1] - bf2f(h1)) << 16);
    }
    const uint4 hv = make_uint4(hw[0], hw[1], hw[2], hw[3]);
    st16(p, hv);
    if (v.x3) {
        *(uint4*)(p + v.third) = hv;
        *(uint4*)(p + 2 * v.third) = make_uint4(lw[0], lw[1], lw[2], lw[3]);
    }
}

#define BN_MAX_BLOCKS 2048

extern "C" int sos_bn_stats_blocks(int64_t npix) {
    int64_t b = (npix + 255) / 256;
    if (b > BN_MAX_BLOCKS) b = BN_MAX_BLOCKS;      // (768 / 1024 / 1536 measured: no better)
    if (b < 1) b = 1;
    return (int)b;
}

// thread = (pixel lane, 8-channel group); per-thread sums, then an LDS tree over the pixel lanes.
__global__ __launch_bounds__(256) void bn_stats_kernel(View x, float* __restrict__ partial) {
    constexpr int BN_U = 4;
    __shared__ float red[256 * 8];          // (8 KB: the two sums go through it one after the other, see bn_bwd_reduce_kernel)
    const int CG = (x.C + 7) / 8;           // <= 256 (C <= 2048)
    const int PL = bn_pl(CG, x.row);
    const int tid = threadIdx.x;
    const int cg = tid % CG, pl = tid / CG;
    float s[8], q[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) { s[i] = 0.f; q[i] = 0.f; }
    if (pl < PL) {
        const long long step = (long long)gridDim.x * PL;
        for (long long pix = (long long)blockIdx.x * PL + pl; pix < x.npix; pix += BN_U * step) {
            float f[BN_U][8];
            load8u<BN_U>(x, pix, step, cg * 8, f);
#pragma unroll
            for (int u = 0; u < BN_U; ++u) {
                if (pix + u * step >= x.npix) break;
#pragma unroll
                for (int i = 0; i < 8; ++i) { s[i] += f[u][i]; q[i] = fmaf(f[u][i], f[u][i], q[i]); }
            }
        }
    }
    // one thread per channel: add over the pixel lanes in a fixed order (sum, then sum of squares)
#pragma unroll
    for (int which = 0; which < 2; ++which) {
        if (which) __syncthreads();
#pragma unroll
        for (int i = 0; i < 8; ++i) red[tid * 8 + i] = which ? q[i] : s[i];
        __syncthreads();
        for (int c = tid; c < x.C; c += 256) {
            const int g = c >> 3, e = c & 7;
            float acc = 0.f;
            for (int l = 0; l < PL; ++l) acc += red[(l * CG + g) * 8 + e];
            partial[((size_t)which * x.C + c) * gridDim.x + blockIdx.x] = acc;       // [2][C][blocks]
        }
    }
}

extern "C" int sos_bn_stats(const sos_view* x, float* partial, sos_stream_t stream) {
    int rc = check_view(x, "sos_bn_stats");
    if (rc) return rc;
    if (!partial) { sos_set_error("sos_bn_stats: null partial"); return SOS_EINVAL; }
    hipLaunchKernelGGL(bn_stats_kernel, dim3(sos_bn_stats_blocks(x->npix)), dim3(256), 0, (hipStream_t)stream,
                       to_view(x), partial);
    return sos_check_launch("sos_bn_stats");
}

// one workgroup per channel: 256 threads stride over the channel's row of per-workgroup partial sums ([2][C][blocks]:
// contiguous, so the 12 288 tile sums of a conv epilogue are read coalesced; fixed order -> deterministic), double
// accumulation, LDS tree.
// NTH threads per channel: 256, or 1024 for the 12 288 tile sums of a full-resolution conv epilogue (round 4: a channel's row
// is then two rounds of 8 loads per thread instead of six -- 20 -> ~6 us per launch of a kernel that runs with the chip to itself)
template <int NTH>
__global__ __launch_bounds__(NTH) void bn_finalize_kernel(const float* __restrict__ partial, int nblk, int C, double count,
                                   const float* __restrict__ gamma, const float* __restrict__ beta, float eps,
                                   float momentum, float* __restrict__ rmean, float* __restrict__ rvar,
                                   long long* __restrict__ nbt, float* __restrict__ scale, float* __restrict__ shift,
                                   float* __restrict__ save_mean, float* __restrict__ save_invstd) {
    __shared__ double rs[NTH], rq[NTH];
    const int c = blockIdx.x, tid = threadIdx.x;
    double s = 0.0, q = 0.0;
    const float* ps = partial + ((size_t)0 * C + c) * nblk;
    const float* pq = partial + ((size_t)1 * C + c) * nblk;
    int b = tid;
    for (; b + 3 * NTH < nblk; b += 4 * NTH) {   // 8 loads in flight per thread; fixed order
        const float a0 = ps[b], a1 = ps[b + NTH], a2 = ps[b + 2 * NTH], a3 = ps[b + 3 * NTH];
        const float c0 = pq[b], c1 = pq[b + NTH], c2 = pq[b + 2 * NTH], c3 = pq[b + 3 * NTH];
        s += (double)a0; s += (double)a1; s += (double)a2; s += (double)a3;
        q += (double)c0; q += (double)c1; q += (double)c2; q += (double)c3;
    }
    for (; b < nblk; b += NTH) { s += (double)ps[b]; q += (double)pq[b]; }
    rs[tid] = s; rq[tid] = q;
    __syncthreads();
    for (int st = NTH / 2; st > 0; st >>= 1) {
        if (tid < st) { rs[tid] += rs[tid + st]; rq[tid] += rq[tid + st]; }
        __syncthreads();
    }
    if (tid != 0) return;
    if (c == 0 && nbt) nbt[0] += 1;
    s = rs[0]; q = rq[0];
    const double mean = s / count;
    double var = q / count - mean * mean;
    if (var < 0.0) var = 0.0;
    const float invstd = (float)(1.0 / sqrt(var + (double)eps));
    const float g = gamma ? gamma[c] : 1.f, bt = beta ? beta[c] : 0.f;
    const float sc = g * invstd;
    scale[c] = sc;
    shift[c] = bt - (float)mean * sc;
    if (save_mean) save_mean[c] = (float)mean;
    if (save_invstd) save_invstd[c] = invstd;
    if (rmean) rmean[c] = (1.f - momentum) * rmean[c] + momentum * (float)mean;
    if (rvar) rvar[c] = (1.f - momentum) * rvar[c] + momentum * (float)(var * count / (count > 1.0 ? count - 1.0 : 1.0));
}

extern "C" int sos_bn_finalize(const float* partial, int nblk, int C, int64_t count, const float* gamma,
                               const float* beta, float eps, float momentum, float* running_mean, float* running_var,
                               int64_t* num_batches_tracked, float* scale, float* shift, float* save_mean,
                               float* save_invstd, sos_stream_t stream) {
    if (!partial || !scale || !shift || nblk < 1 || C < 1 || count < 1) { sos_set_error("sos_bn_finalize: bad args"); return SOS_EINVAL; }
    if (nblk >= 4096)
        hipLaunchKernelGGL(bn_finalize_kernel<1024>, dim3(C), dim3(1024), 0, (hipStream_t)stream, partial, nblk, C,
                           (double)count, gamma, beta, eps, momentum, running_mean, running_var,
                           (long long*)num_batches_tracked, scale, shift, save_mean, save_invstd);
    else
        hipLaunchKernelGGL(bn_finalize_kernel<256>, dim3(C), dim3(256), 0, (hipStream_t)stream, partial, nblk, C,
                           (double)count, gamma, beta, eps, momentum, running_mean, running_var,
                           (long long*)num_batches_tracked, scale, shift, save_mean, save_invstd);
    return sos_check_launch("sos_bn_finalize");
}

__device__ __forceinline__ float apply_act(float z, int act, float slope) {
    if (act == SOS_ACT_RELU) return fmaxf(z, 0.f);
    if (act == SOS_ACT_PRELU) return z >= 0.f ? z : slope * z;
    if (act == SOS_ACT_SIGMOID) return 1.0f / (1.0f + expf(-z));
    return z;
}

// Elementwise passes: a thread keeps ONE 8-channel group for its whole pixel walk, so the per-channel
// coefficients are loaded once into registers (per-element scalar loads made these passes
// issue-bound at ~1.3 TB/s).
__global__ __launch_bounds__(256) void bn_apply_kernel(View x, const float* __restrict__ scale,
                                                       const float* __restrict__ shift, int act,
                                                       const float* __restrict__ slope_p, View y) {
    constexpr int BN_U = 4;
    const int CG = (x.C + 7) / 8;
    const int PL = bn_pl(CG, x.row);
    const int cg = threadIdx.x % CG, pl = threadIdx.x / CG;
    if (pl >= PL) return;
    const float slope = slope_p ? slope_p[0] : 0.f;
    float sc[8], sh[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int c = min(cg * 8 + e, x.C - 1);
        sc[e] = scale[c]; sh[e] = shift[c];
    }
    const long long step = (long long)gridDim.x * PL;
    for (long long pix = (long long)blockIdx.x * PL + pl; pix < x.npix; pix += BN_U * step) {
        float f[BN_U][8];
        load8u<BN_U>(x, pix, step, cg * 8, f);
#pragma unroll
        for (int u = 0; u < BN_U; ++u) {
            if (pix + u * step >= x.npix) break;
#pragma unroll
            for (int e = 0; e < 8; ++e) f[u][e] = (cg * 8 + e < x.C) ? apply_act(fmaf(f[u][e], sc[e], sh[e]), act, slope) : 0.f;
            store8(y, pix + u * step, cg * 8, f[u]);
        }
    }
}

// feature form: one thread per (b, w', h), all C (<= 16) channels
__global__ __launch_bounds__(256) void bn_apply_feat_kernel(View x, const float* __restrict__ scale,
                                                            const float* __restrict__ shift, int act,
                                                            const float* __restrict__ slope_p, View y, int H, int W,
                                                            int Wo, const int* __restrict__ gather, long long total) {
    const float slope = slope_p ? slope_p[0] : 0.f;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int h = (int)(i % H);
        const long long r = i / H;
        const int wo = (int)(r % Wo);
        const long long b = r / Wo;
        const int w = gather ? gather[wo] : wo;
        const long long pix = (b * H + h) * W + w;
        for (int c0 = 0; c0 < x.C; c0 += 8) {
            float f[8];
            load8(x, pix, c0, f);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int c = c0 + e;
                if (c >= x.C) break;
                const float v = apply_act(fmaf(f[e], scale[c], shift[c]), act, slope);
                bf16_t* o = y.ptr + (b * Wo + wo) * y.row + (long long)(y.c_off + c) * H + h;
                const bf16_t hi = f2bf(v);
                o[0] = hi;
                if (y.x3) { o[y.third] = hi; o[2 * y.third] = f2bf(v - bf2f(hi)); }
            }
        }
    }
}

static inline unsigned grid_for(long long total) {
    long long g = (total + 255) / 256;
    static const char* cap_env = getenv("SOS_BN_GRID");
    // 768 = three workgroups per CU: measured optimum of the streaming passes (512 / 640 / 896 / 1024 / 1536 are 1-2 % slower
    // on the training step; the stride between a thread's consecutive pixels depends on it)
    const long long cap = cap_env ? atoll(cap_env) : 768;
    if (g > cap) g = cap;
    if (g < 1) g = 1;
    return (unsigned)g;
}

extern "C" int sos_bn_act_apply(const sos_view* x, const float* scale, const float* shift, int act,
                                const float* slope, const sos_view* y, int feat_H, int feat_W, int feat_Wo,
                                const int32_t* gather, sos_stream_t stream) {
    int rc = check_view(x, "sos_bn_act_apply");
    if (rc) return rc;
    if (!y || !y->ptr || !scale || !shift) { sos_set_error("sos_bn_act_apply: null pointer"); return SOS_EINVAL; }
    if (feat_H > 0) {
        if (feat_W < 1 || feat_Wo < 1 || x->npix % ((long long)feat_H * feat_W)) { sos_set_error("sos_bn_act_apply: bad feature geometry"); return SOS_EINVAL; }
        const long long B = x->npix / ((long long)feat_H * feat_W);
        const long long total = B * feat_Wo * feat_H;
        hipLaunchKernelGGL(bn_apply_feat_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, to_view(x),
                           scale, shift, act, slope, to_view(y), feat_H, feat_W, feat_Wo, gather, total);
        return sos_check_launch("sos_bn_act_apply(feat)");
    }
    rc = check_view(y, "sos_bn_act_apply");
    if (rc) return rc;
    if (y->npix != x->npix || y->C < x->C) { sos_set_error("sos_bn_act_apply: view mismatch"); return SOS_EINVAL; }
    const int PLh = bn_pl((x->C + 7) / 8, x->row);
    hipLaunchKernelGGL(bn_apply_kernel, dim3(grid_for((x->npix + PLh - 1) / PLh * 256)), dim3(256), 0, (hipStream_t)stream,
                       to_view(x), scale, shift, act, slope, to_view(y));
    return sos_check_launch("sos_bn_act_apply");
}


// ------------------------------------------------------------------------- BatchNorm backward
// dz = dy * act'(z), z = x*scale + shift (x = raw conv output); xhat = (x - mean) * invstd.
//   reduce  : S1 = sum dz, S2 = sum dz*xhat, S3 = sum_{z<0} dy*z (PReLU slope gradient)
//   finalize: dgamma = S2, dbeta = S1, dslope = sum_c S3;  dx = a*dz + b*xhat + c with
//             a = gamma*invstd, b = -a*S2/N, c = -a*S1/N
//   apply   : writes dx (grad of the raw conv output) as bf16 / bf16x3.
// With mean == NULL the layer has no BatchNorm (bias + activation only): dx = dz, S1 = dbias.
__device__ __forceinline__ float act_grad(float z, int act, float slope) {
    if (act == SOS_ACT_RELU) return z > 0.f ? 1.f : 0.f;
    if (act == SOS_ACT_PRELU) return z >= 0.f ? 1.f : slope;
    if (act == SOS_ACT_SIGMOID) { const float y = 1.0f / (1.0f + expf(-z)); return y * (1.f - y); }
    return 1.f;
}

// RELU_ONLY (round 4): the ReLU blocks -- 42 of the 60 BatchNorm layers, all the full-resolution ones -- need neither the PReLU slope
// sum S3 nor the general activation derivative: dz = z > 0 ? dy : 0 (three VALU per element less in a pass that runs ~20 % above its
// HBM time; S3's row of the partial buffer is written as zeros so that the finalize stays one code path)
template <bool RELU_ONLY>
__global__ __launch_bounds__(256) void bn_bwd_reduce_kernel(View dy, View x, const float* __restrict__ scale,
                                                            const float* __restrict__ shift,
                                                            const float* __restrict__ mean,
                                                            const float* __restrict__ invstd, int act,
                                                            const float* __restrict__ slope_p,
                                                            float* __restrict__ partial) {
    constexpr int BN_U = SOS_BN_RED_U;
    __shared__ float red[256 * 8];
    const int CG = (x.C + 7) / 8;
    const int PL = SOS_BN_BWD_ALIGNED ? bn_pl(CG, x.row) : 256 / CG;      // pixel lanes spanning whole 128-byte lines (see bn_pl)
    const int tid = threadIdx.x;
    const int cg = tid % CG, pl = tid / CG;
    const float slope = slope_p ? slope_p[0] : 0.f;
    float s1[8], s2[8], s3[8], sc[8], sh[8], mu[8], is[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        s1[i] = s2[i] = s3[i] = 0.f;
        const int c = min(cg * 8 + i, x.C - 1);
        sc[i] = scale[c]; sh[i] = shift[c];
        mu[i] = mean ? mean[c] : 0.f; is[i] = invstd ? invstd[c] : 1.f;
    }
    if (pl < PL) {
        const long long step = (long long)gridDim.x * PL;
        for (long long pix = (long long)blockIdx.x * PL + pl; pix < x.npix; pix += BN_U * step) {
            float fx[BN_U][8], fg[BN_U][8];
            load8u<BN_U>(x, pix, step, cg * 8, fx);
            load8u<BN_U>(dy, pix, step, cg * 8, fg);
#pragma unroll
            for (int u = 0; u < BN_U; ++u) {
                if (pix + u * step >= x.npix) break;
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const float z = fmaf(fx[u][i], sc[i], sh[i]);
                    if constexpr (RELU_ONLY) {
                        const float dz = z > 0.f ? fg[u][i] : 0.f;
                        s1[i] += dz;
                        s2[i] = fmaf(dz, (fx[u][i] - mu[i]) * is[i], s2[i]);
                    } else {
                        const float dz = fg[u][i] * act_grad(z, act, slope);
                        s1[i] += dz;
                        s2[i] = fmaf(dz, (fx[u][i] - mu[i]) * is[i], s2[i]);
                        if (z < 0.f) s3[i] = fmaf(fg[u][i], z, s3[i]);
                    }
                }
            }
        }
    }
    // Fixed-order sum over the pixel lanes, ONE of the three sums at a time through an 8 KB buffer (round 5; rounds 1-4 held all
    // three at once in 24 KB).  Same order of additions, bit-identical partials -- but a workgroup now fits beside two resident
    // workgroups of the 96-channel conv kernel (2 x 72.6 KB of a CU's 160 KB leave 18 KB): under the concurrent training schedule
    // this pass sat at 32.7 ms of kernel time per step against 12.3 ms alone, waiting for LDS that the other streams' MFMA
    // workgroups held (profiles/r05_steady_families.md of the first refresh); bn_stats (16 KB) and the apply passes (0) always fitted.
#pragma unroll
    for (int which = 0; which < 3; ++which) {
        if (which) __syncthreads();
        if (RELU_ONLY && which == 2) {
            for (int c = tid; c < x.C; c += 256) partial[((size_t)2 * x.C + c) * gridDim.x + blockIdx.x] = 0.f;
            break;
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) red[tid * 8 + i] = which == 0 ? s1[i] : (which == 1 ? s2[i] : s3[i]);
        __syncthreads();
        for (int c = tid; c < x.C; c += 256) {
            const int g = c >> 3, e = c & 7;
            float acc = 0.f;
            for (int l = 0; l < PL; ++l) acc += red[(l * CG + g) * 8 + e];
            partial[((size_t)which * x.C + c) * gridDim.x + blockIdx.x] = acc;      // [3][C][blocks]: the finalize reads a channel's row contiguously
        }
    }
}

__global__ __launch_bounds__(256) void bn_bwd_finalize_kernel(const float* __restrict__ partial, int nblk, int C, double count,
                                       const float* __restrict__ gamma, const float* __restrict__ invstd,
                                       float* __restrict__ dgamma, float* __restrict__ dbeta, float* __restrict__ s3out,
                                       float* __restrict__ ca, float* __restrict__ cb, float* __restrict__ cc,
                                       const float* __restrict__ out_scale) {
    __shared__ double r1[256], r2[256], r3[256];
    const int c = blockIdx.x, tid = threadIdx.x;
    double s1 = 0.0, s2 = 0.0, s3 = 0.0;
    const float* p1 = partial + ((size_t)0 * C + c) * nblk;
    const float* p2 = partial + ((size_t)1 * C + c) * nblk;
    const float* p3 = partial + ((size_t)2 * C + c) * nblk;
    int b = tid;
    for (; b + 256 < nblk; b += 512) {           // 6 loads in flight per thread; fixed order
        const float a0 = p1[b], a1 = p1[b + 256], c0 = p2[b], c1 = p2[b + 256], e0 = p3[b], e1 = p3[b + 256];
        s1 += (double)a0; s1 += (double)a1; s2 += (double)c0; s2 += (double)c1; s3 += (double)e0; s3 += (double)e1;
    }
    for (; b < nblk; b += 256) { s1 += (double)p1[b]; s2 += (double)p2[b]; s3 += (double)p3[b]; }
    r1[tid] = s1; r2[tid] = s2; r3[tid] = s3;
    __syncthreads();
    for (int st = 128; st > 0; st >>= 1) {
        if (tid < st) { r1[tid] += r1[tid + st]; r2[tid] += r2[tid + st]; r3[tid] += r3[tid + st]; }
        __syncthreads();
    }
    if (tid != 0) return;
    s1 = r1[0]; s2 = r2[0]; s3 = r3[0];
    // parameter gradients leave in the caller's units: out_scale = 1 / (loss scale the activation gradients carry);
    // dx below stays in the scaled units
    const double os = out_scale ? (double)out_scale[0] : 1.0;
    if (dgamma) dgamma[c] = (float)(s2 * os);
    if (dbeta) dbeta[c] = (float)(s1 * os);
    if (s3out) s3out[c] = (float)(s3 * os);     // per-channel PReLU slope partials, summed by the next kernel
    if (invstd) {
        const double a = (double)(gamma ? gamma[c] : 1.f) * (double)invstd[c];
        ca[c] = (float)a;
        cb[c] = (float)(-a * s2 / count);
        cc[c] = (float)(-a * s1 / count);
    } else {
        ca[c] = 1.f; cb[c] = 0.f; cc[c] = 0.f;
    }
}

__global__ __launch_bounds__(256) void slope_sum_kernel(const float* __restrict__ s3, int C, float* __restrict__ dslope) {
    __shared__ double r[256];
    double t = 0.0;
    for (int c = threadIdx.x; c < C; c += 256) t += (double)s3[c];
    r[threadIdx.x] = t;
    __syncthreads();
    for (int st = 128; st > 0; st >>= 1) {
        if ((int)threadIdx.x < st) r[threadIdx.x] += r[threadIdx.x + st];
        __syncthreads();
    }
    if (threadIdx.x == 0) dslope[0] = (float)r[0];
}

// ---- round 5: the ReLU reduce as per-wave streams.  bn_bwd_reduce_kernel above reads its two tensors at 4.6-4.9 TB/s where the
// one-tensor bn_stats reaches 6.1 and the two-tensor thin weight-gradient kernel (wgrad.hip) 6.0: its 2 048 workgroups walk the
// tensor block-interleaved and every thread issues its next 4 loads only after the sums of the previous 4 are done.  Here a WAVE
// owns a contiguous run of pixels (lane = (pixel lane, 8-channel group) as above, 64 / CG pixels per instruction) and keeps TWO
// register sets of U loads per tensor: the loads of iteration k + 1 are in flight while iteration k is summed.  Same per-element
// arithmetic; the partial sums are added in a different (fixed) order.  partial: [3][C][gridDim.x] like the kernel above.
template <int U, bool RELU_ONLY>
__global__ __launch_bounds__(256) void bn_bwd_reduce_stream_kernel(View dy, View x, const float* __restrict__ scale,
                                                                   const float* __restrict__ shift,
                                                                   const float* __restrict__ mean,
                                                                   const float* __restrict__ invstd, int act,
                                                                   const float* __restrict__ slope_p,
                                                                   float* __restrict__ partial) {
    __shared__ float red[256 * 8];
    const int CG = (x.C + 7) / 8;
    const int PLW = 64 / CG;                            // pixels per wave instruction (launcher: CG <= 64)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int cg = lane % CG, pw = lane / CG;
    const bool live = pw < PLW;
    const float slope = slope_p ? slope_p[0] : 0.f;
    float s1[8], s2[8], s3[8], sc[8], sh[8], mu[8], is[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        s1[i] = s2[i] = s3[i] = 0.f;
        const int c = min(cg * 8 + i, x.C - 1);
        sc[i] = scale[c]; sh[i] = shift[c];
        mu[i] = mean ? mean[c] : 0.f; is[i] = invstd ? invstd[c] : 1.f;
    }
    const long long nw = (long long)gridDim.x * 4, wid = (long long)blockIdx.x * 4 + wave;
    const int IT = U * PLW;                             // pixels per wave iteration
    long long per = (x.npix + nw - 1) / nw;
    per = (per + IT - 1) / IT * IT;
    const long long p0 = wid * per, p1 = p0 + per < x.npix ? p0 + per : x.npix;
    // addresses relative to the wave's first pixel (32-bit); a pixel past the tensor's end is clamped to the last one and skipped
    // by the sums.  The fetches are UNCONDITIONAL so that the compiler's wait-count model of the two register sets is exact
    // (a fetch under a branch made it drain every load at the top of the loop).
    const bf16_t* xb = x.ptr + x.c_off + cg * 8 + p0 * x.row;
    const bf16_t* gb = dy.ptr + dy.c_off + cg * 8 + p0 * dy.row;
    const int rmax = (int)(x.npix - 1 - p0), rend = (int)(p1 - p0);
    auto fetch = [&](int r0, uint4 (&hx)[U], uint4 (&hg)[U]) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int r = min(r0 + u * PLW + pw, rmax);
            hx[u] = ld16(xb + (size_t)((unsigned)r * (unsigned)x.row));
            hg[u] = ld16(gb + (size_t)((unsigned)r * (unsigned)dy.row));
        }
    };
    auto sum = [&](int r0, const uint4 (&hx)[U], const uint4 (&hg)[U]) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const bool ok = r0 + u * PLW + pw < rend;
            const unsigned xw[4] = {hx[u].x, hx[u].y, hx[u].z, hx[u].w}, gw[4] = {hg[u].x, hg[u].y, hg[u].z, hg[u].w};
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const float xv = (i & 1) ? sos_hi2f(xw[i >> 1]) : sos_lo2f(xw[i >> 1]);
                const float gv = (i & 1) ? sos_hi2f(gw[i >> 1]) : sos_lo2f(gw[i >> 1]);
                const float z = fmaf(xv, sc[i], sh[i]);
                if constexpr (RELU_ONLY) {
                    const float dz = (ok && z > 0.f) ? gv : 0.f;
                    s1[i] += dz;
                    s2[i] = fmaf(dz, (xv - mu[i]) * is[i], s2[i]);
                } else {
                    const float g0 = ok ? gv : 0.f;
                    const float dz = g0 * act_grad(z, act, slope);
                    s1[i] += dz;
                    s2[i] = fmaf(dz, (xv - mu[i]) * is[i], s2[i]);
                    if (z < 0.f) s3[i] = fmaf(g0, z, s3[i]);
                }
            }
        }
    };
    if (live && p0 < p1) {
        uint4 ax[U], ag[U], bx[U], bg[U];
        fetch(0, ax, ag);
        for (int r = 0; r < rend; r += 2 * IT) {
            fetch(r + IT, bx, bg);
            sum(r, ax, ag);
            fetch(r + 2 * IT, ax, ag);
            sum(r + IT, bx, bg);
        }
    }
#pragma unroll
    for (int which = 0; which < 3; ++which) {
        if (which) __syncthreads();
        if (RELU_ONLY && which == 2) {
            for (int c = tid; c < x.C; c += 256) partial[((size_t)2 * x.C + c) * gridDim.x + blockIdx.x] = 0.f;
            break;
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) red[tid * 8 + i] = which == 0 ? s1[i] : (which == 1 ? s2[i] : s3[i]);
        __syncthreads();
        for (int c = tid; c < x.C; c += 256) {
            const int g = c >> 3, e = c & 7;
            float acc = 0.f;
            for (int w = 0; w < 4; ++w)
                for (int l = 0; l < PLW; ++l) acc += red[(w * 64 + l * CG + g) * 8 + e];
            partial[((size_t)which * x.C + c) * gridDim.x + blockIdx.x] = acc;
        }
    }
}

// (The apply pass -- 2 reads + 1 write per element -- was written the same way and measured: 315-330 us against 316 us for the
// block-interleaved kernel below at 96 channels, 157-167 against 162 at 48: whatever keeps it at 5.1-5.3 TB/s (torch's add, also 2R + 1W,
// reaches 6.0) is not the loads in flight.  Removed.)
template <bool RELU_ONLY>
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(View dy, View x, const float* __restrict__ scale,
                                                           const float* __restrict__ shift,
                                                           const float* __restrict__ mean,
                                                           const float* __restrict__ invstd, int act,
                                                           const float* __restrict__ slope_p,
                                                           const float* __restrict__ ca, const float* __restrict__ cb,
                                                           const float* __restrict__ cc, View dx) {
    constexpr int BN_U = 2;
    const int CG = (x.C + 7) / 8;
    const int PL = SOS_BN_BWD_ALIGNED ? bn_pl(CG, x.row) : 256 / CG;
    const int cg = threadIdx.x % CG, pl = threadIdx.x / CG;
    if (pl >= PL) return;
    const float slope = slope_p ? slope_p[0] : 0.f;
    float sc[8], sh[8], mu[8], is[8], a[8], b[8], c0[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int c = min(cg * 8 + e, x.C - 1);
        sc[e] = scale[c]; sh[e] = shift[c];
        mu[e] = mean ? mean[c] : 0.f; is[e] = mean ? invstd[c] : 0.f;
        a[e] = ca[c]; b[e] = cb[c]; c0[e] = cc[c];
    }
    const long long step = (long long)gridDim.x * PL;
    for (long long pix = (long long)blockIdx.x * PL + pl; pix < x.npix; pix += BN_U * step) {
        float fx[BN_U][8], fg[BN_U][8];
        load8u<BN_U>(x, pix, step, cg * 8, fx);
        load8u<BN_U>(dy, pix, step, cg * 8, fg);
#pragma unroll
        for (int u = 0; u < BN_U; ++u) {
            if (pix + u * step >= x.npix) break;
            float o[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float z = fmaf(fx[u][e], sc[e], sh[e]);
                const float dz = RELU_ONLY ? (z > 0.f ? fg[u][e] : 0.f) : fg[u][e] * act_grad(z, act, slope);
                const float xh = (fx[u][e] - mu[e]) * is[e];
                o[e] = (cg * 8 + e < x.C) ? fmaf(a[e], dz, fmaf(b[e], xh, c0[e])) : 0.f;
            }
            store8(dx, pix + u * step, cg * 8, o);
        }
    }
}

extern "C" int sos_bn_bwd(const sos_view* dy, const sos_view* x, const float* scale, const float* shift,
                          const float* mean, const float* invstd, const float* gamma, int act, const float* slope,
                          float* partial, float* coef /* [4][C] */, float* dgamma, float* dbeta, float* dslope,
                          const sos_view* dx, const float* out_scale, sos_stream_t stream) {
    int rc = check_view(dy, "sos_bn_bwd");
    if (!rc) rc = check_view(x, "sos_bn_bwd");
    if (!rc) rc = check_view(dx, "sos_bn_bwd");
    if (rc) return rc;
    if (!scale || !shift || !partial || !coef || dy->npix != x->npix || dx->npix != x->npix || dy->C < x->C || dx->C < x->C ||
        (mean == nullptr) != (invstd == nullptr)) {
        sos_set_error("sos_bn_bwd: bad args");
        return SOS_EINVAL;
    }
    hipStream_t s = (hipStream_t)stream;
    int nblk = sos_bn_stats_blocks(x->npix);
    const int C = x->C;
    // SOS_BN_STREAM=<workgroups> (default 512; 0 = the block-interleaved kernel everywhere): the per-wave streaming reduce for the
    // ReLU blocks at full resolution.  Alone on the chip 17-23 % faster (96 channels 219 -> 184 us = 6.1 TB/s, 48: 120 -> 92), in the
    // training step 25.5 -> 23.0 ms of this pass per 4 steps: 543.4 -> 545.3 utt/s concurrent, 513.0 -> 515.3 serial (four / two
    // alternations; a second box: 558.2 -> 561.1, 521.4 -> 524.0).  A first version that also took the U-Net's PReLU layers was 0.998 of the step: see below.
    static const int stream_wgs = [] { const char* e = getenv("SOS_BN_STREAM"); return e ? atoi(e) : 512; }();
    // (ReLU blocks at full resolution only: on the U-Net's PReLU layers -- 64-256 channels, 0.18-0.74 M pixels -- the streaming
    // variant took 140 us where the block-interleaved kernel takes 59-117: runs of ~100 pixels per wave are all prologue)
    const bool use_stream = stream_wgs > 0 && act == SOS_ACT_RELU && !x->x3 && !dy->x3 && (C + 7) / 8 <= 64 && x->npix >= 1 << 20;
    if (use_stream) {
        // (the partial buffer is sized for sos_bn_stats_blocks workgroups: fewer rows are used, with their own pitch)
        if (nblk > stream_wgs) nblk = stream_wgs;
        hipLaunchKernelGGL((bn_bwd_reduce_stream_kernel<4, true>), dim3(nblk), dim3(256), 0, s, to_view(dy), to_view(x), scale, shift,
                           mean, invstd, act, slope, partial);
    } else if (act == SOS_ACT_RELU)
        hipLaunchKernelGGL(bn_bwd_reduce_kernel<true>, dim3(nblk), dim3(256), 0, s, to_view(dy), to_view(x), scale, shift, mean,
                           invstd, act, slope, partial);
    else
        hipLaunchKernelGGL(bn_bwd_reduce_kernel<false>, dim3(nblk), dim3(256), 0, s, to_view(dy), to_view(x), scale, shift, mean,
                           invstd, act, slope, partial);
    // partial[0 .. C) of block 0's S1 row is dead after the finalize read it: reuse the head of the
    // (nblk*3*C) partial buffer?  No -- keep it simple: the slope partials go to the tail of `coef`
    // (caller sizes coef as [4][C]).
    hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3(C), dim3(256), 0, s, partial, nblk, C, (double)x->npix, gamma, invstd,
                       dgamma, dbeta, dslope ? coef + 3 * C : nullptr, coef, coef + C, coef + 2 * C, out_scale);
    if (dslope) hipLaunchKernelGGL(slope_sum_kernel, dim3(1), dim3(256), 0, s, coef + 3 * C, C, dslope);
    const int PLh = SOS_BN_BWD_ALIGNED ? bn_pl((C + 7) / 8, x->row) : 256 / ((C + 7) / 8);
    if (act == SOS_ACT_RELU)
        hipLaunchKernelGGL(bn_bwd_apply_kernel<true>, dim3(grid_for((x->npix + PLh - 1) / PLh * 256)), dim3(256), 0, s, to_view(dy),
                           to_view(x), scale, shift, mean, invstd, act, slope, coef, coef + C, coef + 2 * C, to_view(dx));
    else
        hipLaunchKernelGGL(bn_bwd_apply_kernel<false>, dim3(grid_for((x->npix + PLh - 1) / PLh * 256)), dim3(256), 0, s, to_view(dy),
                           to_view(x), scale, shift, mean, invstd, act, slope, coef, coef + C, coef + 2 * C, to_view(dx));
    return sos_check_launch("sos_bn_bwd");
}

// dz = dy * act'(y) for layers whose pre-activation is not kept (nn.Linear + ReLU / Sigmoid heads):
// ReLU: y > 0, Sigmoid: y(1-y).
__global__ __launch_bounds__(256) void act_bwd_from_y_kernel(View dy, View y, int act, View dz) {
    const int CG = (y.C + 7) / 8;
    const long long total = y.npix * CG;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const long long pix = i / CG;
        const int cg = (int)(i - pix * CG);
        float fy[8], fg[8], o[8];
        load8(y, pix, cg * 8, fy);
        load8(dy, pix, cg * 8, fg);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            float d = fg[e];
            if (act == SOS_ACT_RELU) d = fy[e] > 0.f ? d : 0.f;
            else if (act == SOS_ACT_SIGMOID) d = d * fy[e] * (1.f - fy[e]);
            o[e] = (cg * 8 + e < y.C) ? d : 0.f;
        }
        store8(dz, pix, cg * 8, o);
    }
}

extern "C" int sos_act_bwd_from_y(const sos_view* dy, const sos_view* y, int act, const sos_view* dz,
                                  sos_stream_t stream) {
    int rc = check_view(dy, "sos_act_bwd_from_y");
    if (!rc) rc = check_view(y, "sos_act_bwd_from_y");
    if (!rc) rc = check_view(dz, "sos_act_bwd_from_y");
    if (rc) return rc;
    if (dy->npix != y->npix || dz->npix != y->npix) { sos_set_error("sos_act_bwd_from_y: view mismatch"); return SOS_EINVAL; }
    const long long total = y->npix * ((y->C + 7) / 8);
    hipLaunchKernelGGL(act_bwd_from_y_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, to_view(dy),
                       to_view(y), act, to_view(dz));
    return sos_check_launch("sos_act_bwd_from_y");
}

// f32 strided gradient (+ optional sigmoid' from the f32 output y) -> bf16 / bf16x3 rows.
// element (outer o, inner t, channel c) is read at g[o*so + t*st + c*sc] and written to
// out row (o*inner + t), channel c.
__global__ __launch_bounds__(256) void pack_grad_kernel(const float* __restrict__ g, const float* __restrict__ y, int act,
                                                        long long outer, long long inner, int C, long long so,
                                                        long long st, long long sc, View out, const float* __restrict__ mul_p) {
    const long long total = outer * inner * C;
    const float mul = mul_p ? mul_p[0] : 1.f;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        // t fastest so that reads along the (usually unit-stride) inner axis coalesce
        const long long t = i % inner;
        const long long r = i / inner;
        const int c = (int)(r % C);
        const long long o = r / C;
        const long long src = o * so + t * st + c * sc;
        float v = g[src] * mul;
        if (act == SOS_ACT_SIGMOID) { const float yy = y[src]; v *= yy * (1.f - yy); }
        bf16_t* dst = out.ptr + (o * inner + t) * out.row + out.c_off + c;
        const bf16_t hi = f2bf(v);
        dst[0] = hi;
        if (out.x3) { dst[out.third] = hi; dst[2 * out.third] = f2bf(v - bf2f(hi)); }
    }
}

// channel-contiguous rows (sc == 1: the LSTM gate gradients), round 4: a thread converts 8 consecutive channels of a row -- two
// 16-byte loads, one 16-byte store, 32-bit index arithmetic (the element-wise kernel above pays three 64-bit divisions per VALUE)
__global__ __launch_bounds__(256) void pack_grad_rows_kernel(const float* __restrict__ g, int inner, int CG, long long so, long long st,
                                                             View out, const float* __restrict__ mul_p, int total) {
    const float mul = mul_p ? mul_p[0] : 1.f;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < total; i += gridDim.x * 256) {
        const int row = i / CG, cg = i - row * CG;
        const int o = row / inner, t = row - o * inner;
        const float4* src = (const float4*)(g + o * so + t * st + cg * 8);
        const float4 a = src[0], b = src[1];
        float f[8] = {a.x * mul, a.y * mul, a.z * mul, a.w * mul, b.x * mul, b.y * mul, b.z * mul, b.w * mul};
        store8(out, row, cg * 8, f);
    }
}

extern "C" int sos_pack_grad_f32(const float* g, const float* y, int act, int64_t outer, int64_t inner, int C,
                                 int64_t so, int64_t st, int64_t sc, const sos_view* out, const float* mul, sos_stream_t stream) {
    if (!g || !out || !out->ptr || outer < 1 || inner < 1 || C < 1 || (act == SOS_ACT_SIGMOID && !y)) {
        sos_set_error("sos_pack_grad_f32: bad args");
        return SOS_EINVAL;
    }
    const long long total = outer * inner * C;
    if (sc == 1 && act != SOS_ACT_SIGMOID && C % 8 == 0 && total / 8 < 0x7fffffffLL && so % 4 == 0 && st % 4 == 0 &&
        ((uintptr_t)g & 15) == 0 && out->c_off % 8 == 0 && out->row % 8 == 0 && out->C >= C) {
        const int n8 = (int)(total / 8);
        hipLaunchKernelGGL(pack_grad_rows_kernel, dim3(grid_for((long long)n8)), dim3(256), 0, (hipStream_t)stream, g, (int)inner, C / 8,
                           (long long)so, (long long)st, to_view(out), mul, n8);
        return sos_check_launch("sos_pack_grad_f32(rows)");
    }
    hipLaunchKernelGGL(pack_grad_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, g, y, act, outer, inner, C,
                       so, st, sc, to_view(out), mul);
    return sos_check_launch("sos_pack_grad_f32");
}

// gradient of the feature form back to NHWC: dy[b][h][w][c] = sum_{i in [lo[w], hi[w])} dfeat[b][i][(c_off+c)*H + h]
// (lo/hi NULL: identity, i == w).  Inverse of sos_bn_act_apply's feature write (+ nearest-resize gather).
__global__ __launch_bounds__(256) void feat_to_nhwc_kernel(View f, int H, int W, int Wo, const int* __restrict__ lo,
                                                           const int* __restrict__ hi, View out, long long total) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        int h, w;
        long long b;
        if (total < 0x7fffffffll) {
            const unsigned iu = (unsigned)i, ru = iu / (unsigned)H;
            h = (int)(iu - ru * (unsigned)H);
            b = ru / (unsigned)W;
            w = (int)(ru - (unsigned)b * (unsigned)W);
        } else {
            h = (int)(i % H);
            const long long r = i / H;
            w = (int)(r % W);
            b = r / W;
        }
        const int i0 = lo ? lo[w] : w, i1 = hi ? hi[w] : w + 1;
        const long long pix = (b * H + h) * W + w;
        if (out.C <= 8 && out.c_off % 8 == 0 && out.c_off + 8 <= out.row) {    // (the run's padding channels are written as zeros)
            // (round 4) the heads feeding the BiLSTM have 8 / 4 channels: gather them and write the pixel's 16-byte run at once
            // instead of one 2-byte store per channel at a 32-byte pitch across the wave
            float acc[8];
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                acc[c] = 0.f;
                if (c < out.C) {
                    for (int k = i0; k < i1; ++k) {
                        const bf16_t* p = f.ptr + (b * Wo + k) * f.row + (long long)(f.c_off + c) * H + h;
                        acc[c] += bf2f(p[0]);
                        if (f.x3) acc[c] += bf2f(p[2 * f.third]);
                    }
                }
            }
            store8(out, pix, 0, acc);
            continue;
        }
        for (int c = 0; c < out.C; ++c) {
            float acc = 0.f;
            for (int k = i0; k < i1; ++k) {
                const bf16_t* p = f.ptr + (b * Wo + k) * f.row + (long long)(f.c_off + c) * H + h;
                acc += bf2f(p[0]);
                if (f.x3) acc += bf2f(p[2 * f.third]);
            }
            bf16_t* d = out.ptr + pix * out.row + out.c_off + c;
            const bf16_t hv = f2bf(acc);
            d[0] = hv;
            if (out.x3) { d[out.third] = hv; d[2 * out.third] = f2bf(acc - bf2f(hv)); }
        }
    }
}

// the plain case of the above (no pooling ranges, <= 8 channels, 16-bit single storage, W == Wo) through an LDS tile: the
// kernel above reads along h (coalesced) and writes one 16-byte pixel run per thread at a pitch of W pixels -- 0.9 TB/s.  Here a
// workgroup owns (clip, 64 rows, 8 columns): it reads the 8 x 8 (column, channel) runs of 64 h-contiguous values (128 B each),
// transposes through LDS and writes 8 neighbouring pixels' runs per row.
__global__ __launch_bounds__(256) void feat_to_nhwc_tile_kernel(View f, int H, int W, View out, int tiles_h, int tiles_w) {
    __shared__ unsigned short tile[8][64][8 + 2];          // [w][h][c] (+2: the transposing writes spread over the banks)
    const int tid = threadIdx.x;
    int t = blockIdx.x;
    const int tw = t % tiles_w; t /= tiles_w;
    const int th = t % tiles_h;
    const long long b = t / tiles_h;
    const int h0 = th * 64, w0 = tw * 8;
    // read: run q = (w, c) of 64 values; 4 runs per pass of 256 threads
    for (int q = tid >> 6; q < 64; q += 4) {
        const int w = q >> 3, c = q & 7, h = h0 + (tid & 63);
        unsigned short v = 0;
        if (c < out.C && w0 + w < W && h < H) v = f.ptr[(b * W + w0 + w) * f.row + (long long)(f.c_off + c) * H + h];
        tile[w][tid & 63][c] = v;
    }
    __syncthreads();
    // write: 512 pixels, 2 per thread, w fastest
    for (int p = tid; p < 512; p += 256) {
        const int w = p & 7, hh = p >> 3;
        if (w0 + w >= W || h0 + hh >= H) continue;
        const unsigned short* r = tile[w][hh];
        const uint4 v = make_uint4(r[0] | ((unsigned)r[1] << 16), r[2] | ((unsigned)r[3] << 16), r[4] | ((unsigned)r[5] << 16),
                                   r[6] | ((unsigned)r[7] << 16));
        *(uint4*)(out.ptr + ((b * H + h0 + hh) * W + w0 + w) * out.row + out.c_off) = v;
    }
}

extern "C" int sos_feat_to_nhwc(const sos_view* feat, int B, int H, int W, int Wo, const int32_t* lo, const int32_t* hi,
                                const sos_view* out, sos_stream_t stream) {
    if (!feat || !feat->ptr || !out || !out->ptr || B < 1 || H < 1 || W < 1 || Wo < 1 || (lo == nullptr) != (hi == nullptr)) {
        sos_set_error("sos_feat_to_nhwc: bad args");
        return SOS_EINVAL;
    }
    const long long total = (long long)B * W * H;
    const int tiles_h = (H + 63) / 64, tiles_w = (W + 7) / 8;
    if (!lo && W == Wo && !feat->x3 && !out->x3 && out->C <= 8 && out->c_off % 8 == 0 && out->c_off + 8 <= out->row &&
        (long long)B * tiles_h * tiles_w < 0x7fffffffll && !getenv("SOS_FEAT_NO_TILE")) {
        hipLaunchKernelGGL(feat_to_nhwc_tile_kernel, dim3((unsigned)(B * tiles_h * tiles_w)), dim3(256), 0, (hipStream_t)stream,
                           to_view(feat), H, W, to_view(out), tiles_h, tiles_w);
        return sos_check_launch("sos_feat_to_nhwc");
    }
    hipLaunchKernelGGL(feat_to_nhwc_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, to_view(feat), H, W, Wo,
                       lo, hi, to_view(out), total);
    return sos_check_launch("sos_feat_to_nhwc");
}


// ------------------------------------------------------------------- reflect-pad gradient fold
// Backward of ReflectionPad2d(pad) (DownConvBlock, M2/networks.py:105): the gradient w.r.t. the
// padded tensor [B][H+2p][W+2p] is folded back, every border cell adding to the interior cell it
// mirrors.  out (+)= fold(padded).
// border_only: the interior cell (h + pad, w + pad) has already been written to `out` by the convolution (sos_conv_desc.fold_pad);
// only pixels with a mirrored border cell are touched, and those cells are ADDED to what is there.
template <bool BORDER_ONLY>
__global__ __launch_bounds__(256) void reflect_fold_kernel(View pd, int H, int W, int pad, View out, int accumulate,
                                                           long long total) {
    const int CG = (out.C + 7) / 8;
    const int Wp = W + 2 * pad, Hp = H + 2 * pad;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int cg = (int)(i % CG);
        long long r = i / CG;
        const int w = (int)(r % W); r /= W;
        const int h = (int)(r % H);
        const long long b = r / H;
        int us[3], vs[3], nu = 0, nv = 0;
        us[nu++] = h + pad;
        if (h >= 1 && h <= pad) us[nu++] = pad - h;
        if (h >= H - 1 - pad && h <= H - 2) us[nu++] = 2 * (H - 1) - h + pad;
        vs[nv++] = w + pad;
        if (w >= 1 && w <= pad) vs[nv++] = pad - w;
        if (w >= W - 1 - pad && w <= W - 2) vs[nv++] = 2 * (W - 1) - w + pad;
        if (BORDER_ONLY && nu == 1 && nv == 1) continue;
        float acc[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[e] = 0.f;
        for (int a = 0; a < nu; ++a)
            for (int c = 0; c < nv; ++c) {
                if (BORDER_ONLY && a == 0 && c == 0) continue;
                float f[8];
                load8(pd, (b * Hp + us[a]) * Wp + vs[c], cg * 8, f);
#pragma unroll
                for (int e = 0; e < 8; ++e) acc[e] += f[e];
            }
        const long long opix = (b * H + h) * W + w;
        if (accumulate) {
            float f[8];
            load8(out, opix, cg * 8, f);
#pragma unroll
            for (int e = 0; e < 8; ++e) acc[e] += f[e];
        }
        store8(out, opix, cg * 8, acc);
    }
}

// Border-only fold (round 4): the generic kernel above walks EVERY (pixel, 8-channel group) of the image with three 64-bit
// divisions each and skips the interior ones -- 120 iterations per thread on a 256 x 178 x 64-channel gradient of which 4 %
// touch a border cell (2.2 ms per training step for 16 launches of what is a few MB of traffic).  This one enumerates only the
// pixels that HAVE a mirrored border cell: set A = the rows 1..pad and H-1-pad..H-2 (all columns), set B = the columns
// 1..pad and W-1-pad..W-2 of the remaining rows; blockIdx.y = image, 32-bit arithmetic.  Same sums in the same order per pixel.
struct FoldGeom { int nrows, rows_all, ncols, cols_all, nA, n; };
__host__ __device__ __forceinline__ FoldGeom fold_geom(int H, int W, int pad) {
    FoldGeom g;
    g.rows_all = H - 1 - pad <= pad + 1;                  // the two row bands meet: every row 1..H-2 has a mirrored cell
    g.nrows = g.rows_all ? (H - 2 > 0 ? H - 2 : 0) : 2 * pad;
    g.cols_all = W - 1 - pad <= pad + 1;
    g.ncols = g.cols_all ? (W - 2 > 0 ? W - 2 : 0) : 2 * pad;
    g.nA = g.nrows * W;
    g.n = g.nA + (H - g.nrows) * g.ncols;
    return g;
}
__global__ __launch_bounds__(256) void reflect_fold_border_kernel(View pd, int H, int W, int pad, View out, FoldGeom g) {
    const int CG = (out.C + 7) / 8;
    const int Wp = W + 2 * pad, Hp = H + 2 * pad;
    const long long b = blockIdx.y;
    const int total = g.n * CG;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < total; i += gridDim.x * 256) {
        const int j = i / CG, cg = i - j * CG;
        int h, w;
        if (j < g.nA) {                                   // set A: a banded row, any column
            const int slot = j / W;
            w = j - slot * W;
            h = g.rows_all ? 1 + slot : (slot < pad ? 1 + slot : H - 1 - pad + (slot - pad));
        } else {                                          // set B: another row, a banded column
            const int k = j - g.nA, rslot = k / g.ncols, cslot = k - rslot * g.ncols;
            // rows outside the bands: 0, pad+1 .. H-2-pad, H-1 (only 0 and H-1 when the bands cover 1..H-2)
            h = rslot == 0 ? 0 : (g.rows_all ? H - 1 : (rslot <= H - 2 - 2 * pad ? pad + rslot : H - 1));
            w = g.cols_all ? 1 + cslot : (cslot < pad ? 1 + cslot : W - 1 - pad + (cslot - pad));
        }
        int us[3], vs[3], nu = 0, nv = 0;
        us[nu++] = h + pad;
        if (h >= 1 && h <= pad) us[nu++] = pad - h;
        if (h >= H - 1 - pad && h <= H - 2) us[nu++] = 2 * (H - 1) - h + pad;
        vs[nv++] = w + pad;
        if (w >= 1 && w <= pad) vs[nv++] = pad - w;
        if (w >= W - 1 - pad && w <= W - 2) vs[nv++] = 2 * (W - 1) - w + pad;
        float acc[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[e] = 0.f;
        for (int a = 0; a < nu; ++a)
            for (int c = 0; c < nv; ++c) {
                if (a == 0 && c == 0) continue;
                float f[8];
                load8(pd, (b * Hp + us[a]) * Wp + vs[c], cg * 8, f);
#pragma unroll
                for (int e = 0; e < 8; ++e) acc[e] += f[e];
            }
        const long long opix = (b * H + h) * W + w;
        float f[8];
        load8(out, opix, cg * 8, f);
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[e] += f[e];
        store8(out, opix, cg * 8, acc);
    }
}

extern "C" int sos_reflect_fold(const sos_view* padded, int H, int W, int pad, const sos_view* out, int accumulate,
                                sos_stream_t stream) {
    int rc = check_view(padded, "sos_reflect_fold");
    if (!rc) rc = check_view(out, "sos_reflect_fold");
    if (rc) return rc;
    if (H < 1 || W < 1 || pad < 0 || pad >= H || pad >= W || out->npix % ((long long)H * W) ||
        padded->npix != out->npix / ((long long)H * W) * (H + 2 * pad) * (W + 2 * pad) || padded->C < out->C) {
        sos_set_error("sos_reflect_fold: bad geometry");
        return SOS_EINVAL;
    }
    const long long total = out->npix * ((out->C + 7) / 8);
    hipLaunchKernelGGL(reflect_fold_kernel<false>, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, to_view(padded), H, W,
                       pad, to_view(out), accumulate, total);
    return sos_check_launch("sos_reflect_fold");
}

extern "C" int sos_reflect_fold_border(const sos_view* padded, int H, int W, int pad, const sos_view* out, sos_stream_t stream) {
    int rc = check_view(padded, "sos_reflect_fold_border");
    if (!rc) rc = check_view(out, "sos_reflect_fold_border");
    if (rc) return rc;
    if (H < 1 || W < 1 || pad < 1 || pad >= H || pad >= W || out->npix % ((long long)H * W) ||
        padded->npix != out->npix / ((long long)H * W) * (H + 2 * pad) * (W + 2 * pad) || padded->C < out->C) {
        sos_set_error("sos_reflect_fold_border: bad geometry");
        return SOS_EINVAL;
    }
    const long long B = out->npix / ((long long)H * W);
    const FoldGeom g = fold_geom(H, W, pad);
    const long long per_image = (long long)g.n * ((out->C + 7) / 8);
    static const char* old_path = getenv("SOS_FOLD_BORDER_WALK");        // =1: the all-pixel walk of round 3 (A/B switch)
    // (pad <= min(H, W) - 2: both bands lie inside rows 1..H-2 / columns 1..W-2; a pad of H - 1 also mirrors onto rows 0 and H - 1)
    if ((old_path && atoi(old_path)) || B > 65535 || per_image >= 0x7fffffffLL || pad > H - 2 || pad > W - 2) {
        const long long total = out->npix * ((out->C + 7) / 8);
        hipLaunchKernelGGL(reflect_fold_kernel<true>, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, to_view(padded), H, W,
                           pad, to_view(out), 1, total);
        return sos_check_launch("sos_reflect_fold_border");
    }
    if (per_image == 0) return SOS_OK;
    long long gx = (per_image + 255) / 256;
    const long long cap = (768 + B - 1) / B;                               // ~three workgroups per CU in all
    if (gx > cap) gx = cap;
    if (gx < 1) gx = 1;
    hipLaunchKernelGGL(reflect_fold_border_kernel, dim3((unsigned)gx, (unsigned)B), dim3(256), 0, (hipStream_t)stream, to_view(padded),
                       H, W, pad, to_view(out), g);
    return sos_check_launch("sos_reflect_fold_border");
}

// copy a channel slice between two NHWC pixel grids of different size: the overlap
// [min(Hs,Hd)] x [min(Ws,Wd)] is copied, the rest of dst is zero filled (crop of the
// ConvTranspose2d output to the skip tensor's size in the forward pass, zero-pad back in backward).
__global__ __launch_bounds__(256) void copy_crop_kernel(View src, int Hs, int Ws, View dst, int Hd, int Wd, long long total) {
    const int CG = (dst.C + 7) / 8;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        int cg, w, h;
        long long b;
        if (total < 0x7fffffffll) {        // 32-bit divisions (the 64-bit ones cost more than the 16-byte copy they index)
            unsigned r = (unsigned)i;
            cg = (int)(r % (unsigned)CG); r /= (unsigned)CG;
            w = (int)(r % (unsigned)Wd); r /= (unsigned)Wd;
            h = (int)(r % (unsigned)Hd);
            b = r / (unsigned)Hd;
        } else {
            cg = (int)(i % CG);
            long long r = i / CG;
            w = (int)(r % Wd); r /= Wd;
            h = (int)(r % Hd);
            b = r / Hd;
        }
        float f[8];
        if (h < Hs && w < Ws) {
            load8(src, (b * Hs + h) * Ws + w, cg * 8, f);
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) f[e] = 0.f;
        }
        store8(dst, (b * Hd + h) * Wd + w, cg * 8, f);
    }
}

extern "C" int sos_copy_crop(const sos_view* src, int Hs, int Ws, const sos_view* dst, int Hd, int Wd, sos_stream_t stream) {
    int rc = check_view(src, "sos_copy_crop");
    if (!rc) rc = check_view(dst, "sos_copy_crop");
    if (rc) return rc;
    if (Hs < 1 || Ws < 1 || Hd < 1 || Wd < 1 || src->npix % ((long long)Hs * Ws) || dst->npix % ((long long)Hd * Wd) ||
        src->npix / ((long long)Hs * Ws) != dst->npix / ((long long)Hd * Wd) || src->C < dst->C) {
        sos_set_error("sos_copy_crop: bad geometry");
        return SOS_EINVAL;
    }
    const long long total = dst->npix * ((dst->C + 7) / 8);
    hipLaunchKernelGGL(copy_crop_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, to_view(src), Hs, Ws,
                       to_view(dst), Hd, Wd, total);
    return sos_check_launch("sos_copy_crop");
}
