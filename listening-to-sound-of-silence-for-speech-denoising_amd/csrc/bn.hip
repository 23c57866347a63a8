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

static int check_view(const sos_view* v, const char* what) {
    if (!v || !v->ptr || v->npix < 1 || v->C < 1 || v->C > 256 || v->row % 8 || v->c_off % 8 || (v->x3 && v->third % 8)) {
        sos_set_error("%s: bad view", what);
        return SOS_EINVAL;
    }
    return SOS_OK;
}

__device__ __forceinline__ void load8(const View& v, long long pix, int c8, float (&f)[8]) {
    const bf16_t* p = v.ptr + pix * v.row + v.c_off + c8;
    const uint4 h = *(const uint4*)p;
    const unsigned hw[4] = {h.x, h.y, h.z, h.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        f[2 * i] = __uint_as_float(hw[i] << 16);
        f[2 * i + 1] = __uint_as_float(hw[i] & 0xffff0000u);
    }
    if (v.x3) {
        const uint4 l = *(const uint4*)(p + 2 * v.third);
        const unsigned lw[4] = {l.x, l.y, l.z, l.w};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            f[2 * i] += __uint_as_float(lw[i] << 16);
            f[2 * i + 1] += __uint_as_float(lw[i] & 0xffff0000u);
        }
    }
}

__device__ __forceinline__ void store8(const View& v, long long pix, int c8, const float (&f)[8]) {
    bf16_t* p = v.ptr + pix * v.row + v.c_off + c8;
    unsigned hw[4], lw[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const bf16_t h0 = f2bf(f[2 * i]), h1 = f2bf(f[2 * i + 1]);
        hw[i] = (unsigned)h0 | ((unsigned)h1 << 16);
        if (v.x3) lw[i] = (unsigned)f2bf(f[2 * i] - bf2f(h0)) | ((unsigned)f2bf(f[2 * i + 1] - bf2f(h1)) << 16);
    }
    const uint4 hv = make_uint4(hw[0], hw[1], hw[2], hw[3]);
    *(uint4*)p = hv;
    if (v.x3) {
        *(uint4*)(p + v.third) = hv;
        *(uint4*)(p + 2 * v.third) = make_uint4(lw[0], lw[1], lw[2], lw[3]);
    }
}

#define BN_MAX_BLOCKS 2048

extern "C" int sos_bn_stats_blocks(int64_t npix) {
    int64_t b = (npix + 255) / 256;
    if (b > BN_MAX_BLOCKS) b = BN_MAX_BLOCKS;
    if (b < 1) b = 1;
    return (int)b;
}

// thread = (pixel lane, 8-channel group); per-thread sums, then an LDS tree over the pixel lanes.
__global__ __launch_bounds__(256) void bn_stats_kernel(View x, float* __restrict__ partial) {
    __shared__ float red[256 * 16];
    const int CG = (x.C + 7) / 8;
    const int PL = 256 / CG;
    const int tid = threadIdx.x;
    const int cg = tid % CG, pl = tid / CG;
    float s[8], q[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) { s[i] = 0.f; q[i] = 0.f; }
    if (pl < PL) {
        for (long long pix = (long long)blockIdx.x * PL + pl; pix < x.npix; pix += (long long)gridDim.x * PL) {
            float f[8];
            load8(x, pix, cg * 8, f);
#pragma unroll
            for (int i = 0; i < 8; ++i) { s[i] += f[i]; q[i] = fmaf(f[i], f[i], q[i]); }
        }
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) { red[tid * 16 + i] = s[i]; red[tid * 16 + 8 + i] = q[i]; }
    __syncthreads();
    // one thread per (channel, sum|sq): add over the pixel lanes in a fixed order
    for (int o = tid; o < 2 * x.C; o += 256) {
        const int which = o / x.C, c = o - which * x.C;
        const int g = c >> 3, e = c & 7;
        float acc = 0.f;
        for (int l = 0; l < PL; ++l) acc += red[(l * CG + g) * 16 + which * 8 + e];
        partial[((size_t)blockIdx.x * 2 + which) * x.C + c] = acc;
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

__global__ void bn_finalize_kernel(const float* __restrict__ partial, int nblk, int C, double count,
                                   const float* __restrict__ gamma, const float* __restrict__ beta, float eps,
                                   float momentum, float* __restrict__ rmean, float* __restrict__ rvar,
                                   long long* __restrict__ nbt, float* __restrict__ scale, float* __restrict__ shift,
                                   float* __restrict__ save_mean, float* __restrict__ save_invstd) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c == 0 && nbt) nbt[0] += 1;
    if (c >= C) return;
    double s = 0.0, q = 0.0;
    for (int b = 0; b < nblk; ++b) {
        s += (double)partial[((size_t)b * 2 + 0) * C + c];
        q += (double)partial[((size_t)b * 2 + 1) * C + c];
    }
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
    hipLaunchKernelGGL(bn_finalize_kernel, dim3((C + 63) / 64), dim3(64), 0, (hipStream_t)stream, partial, nblk, C,
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

__global__ __launch_bounds__(256) void bn_apply_kernel(View x, const float* __restrict__ scale,
                                                       const float* __restrict__ shift, int act,
                                                       const float* __restrict__ slope_p, View y) {
    const int CG = (x.C + 7) / 8;
    const float slope = slope_p ? slope_p[0] : 0.f;
    const long long total = x.npix * CG;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const long long pix = i / CG;
        const int cg = (int)(i - pix * CG);
        float f[8];
        load8(x, pix, cg * 8, f);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int c = cg * 8 + e;
            f[e] = c < x.C ? apply_act(fmaf(f[e], scale[c], shift[c]), act, slope) : 0.f;
        }
        store8(y, pix, cg * 8, f);
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
    if (g > 8192) g = 8192;
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
    const long long total = x->npix * ((x->C + 7) / 8);
    hipLaunchKernelGGL(bn_apply_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, to_view(x), scale,
                       shift, act, slope, to_view(y));
    return sos_check_launch("sos_bn_act_apply");
}
