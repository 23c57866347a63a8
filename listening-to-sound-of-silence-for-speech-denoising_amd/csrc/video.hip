// video.hip -- glue kernels of the audio-visual variant's video branch (reference classes
// Conv3dBlock M1/networks.py:54-77 and make_video_branch :110-118, configuration :87-89, fusion :135-142).
//
// A Conv3d(kt x kh x kw, temporal stride 1, zero padding (kt-1)/2) over frames [B][T] is run by the 2-D implicit-GEMM
// kernel (conv.hip) on a TIME-STACKED input: for every frame the kt neighbouring frames' channels side by side,
// channel index dt*C + c (zeros where t + dt - pt falls outside the clip), so the temporal taps become part of the
// contraction axis.  time_stack_kernel builds that tensor (HBM-bound: reads kt x, writes kt x the activation);
// spatial_mean_kernel is torch.mean(f_v, dim=(-2,-1)) written straight into the BiLSTM feature matrix next to the
// audio features (the channel concat of the reference).
#include "sos_common.h"

// in  [B*T][HW][in_row]  (in_row = nseg * in_cs; C real channels at the start of every third)
// out [B*T][HW][nseg * out_cs], out channel dt*C + c; out_cs >= kt*C (rest zero)
__global__ void time_stack_kernel(const bf16_t* __restrict__ in, int T, long long HW, int C, int in_cs, int nseg, int kt,
                                  bf16_t* __restrict__ out, int out_cs, long long total) {
    const int pt = (kt - 1) / 2;
    const int groups = out_cs / 8;                       // 8-channel output pieces per third
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int g = (int)(i % groups);
        long long r = i / groups;
        const int third = (int)(r % nseg); r /= nseg;
        const long long pix = r % HW;
        const long long bt = r / HW;
        const int t = (int)(bt % T);
        bf16_t v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int oc = g * 8 + e;
            const int dt = oc / C, c = oc - dt * C;
            const int ts = t + dt - pt;
            v[e] = (dt < kt && ts >= 0 && ts < T) ? in[((bt + dt - pt) * HW + pix) * ((long long)nseg * in_cs) + third * in_cs + c] : (bf16_t)0;
        }
        uint4 o;
        o.x = v[0] | ((unsigned)v[1] << 16); o.y = v[2] | ((unsigned)v[3] << 16);
        o.z = v[4] | ((unsigned)v[5] << 16); o.w = v[6] | ((unsigned)v[7] << 16);
        *(uint4*)(out + ((bt * HW + pix) * nseg + third) * out_cs + g * 8) = o;
    }
}

extern "C" int sos_time_stack(const void* in, int64_t B, int T, int64_t HW, int C, int in_cs, int nseg, int kt, void* out,
                              int out_cs, sos_stream_t stream) {
    if (!in || !out || B < 1 || T < 1 || HW < 1 || C < 1 || C > in_cs || (nseg != 1 && nseg != 3) || kt < 1 || !(kt & 1) ||
        out_cs % 8 || out_cs < kt * C) {
        sos_set_error("sos_time_stack: bad args (T=%d C=%d in_cs=%d nseg=%d kt=%d out_cs=%d)", T, C, in_cs, nseg, kt, out_cs);
        return SOS_EINVAL;
    }
    const long long total = (long long)B * T * HW * nseg * (out_cs / 8);
    long long grid = (total + 255) / 256;
    if (grid > 65536) grid = 65536;
    hipLaunchKernelGGL(time_stack_kernel, dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)in, T,
                       (long long)HW, C, in_cs, nseg, kt, (bf16_t*)out, out_cs, total);
    return sos_check_launch("sos_time_stack");
}

// out[n * out_row + third * out_third + c_off + c] = mean over the HW pixels of image n (hi + lo in bf16x3 mode,
// re-split into hi | hi | lo)
__global__ void spatial_mean_kernel(const bf16_t* __restrict__ in, long long N, long long HW, int C, int in_cs, int nseg,
                                    bf16_t* __restrict__ out, long long out_row, int out_third, int c_off) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N * C) return;
    const int c = (int)(i % C);
    const long long n = i / C;
    const long long row = (long long)nseg * in_cs;
    float acc = 0.f;
    for (long long p = 0; p < HW; ++p) {
        const bf16_t* px = in + (n * HW + p) * row;
        acc += bf2f(px[c]);
        if (nseg == 3) acc += bf2f(px[2 * in_cs + c]);
    }
    acc /= (float)HW;
    const bf16_t hi = f2bf(acc);
    bf16_t* o = out + n * out_row + c_off + c;
    o[0] = hi;
    if (nseg == 3) { o[out_third] = hi; o[2 * out_third] = f2bf(acc - bf2f(hi)); }
}

extern "C" int sos_spatial_mean(const void* in, int64_t N, int64_t HW, int C, int in_cs, int nseg, void* out, int64_t out_row,
                                int out_third, int out_c_off, sos_stream_t stream) {
    if (!in || !out || N < 1 || HW < 1 || C < 1 || C > in_cs || (nseg != 1 && nseg != 3) || out_c_off < 0) {
        sos_set_error("sos_spatial_mean: bad args");
        return SOS_EINVAL;
    }
    const long long total = (long long)N * C;
    hipLaunchKernelGGL(spatial_mean_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       (const bf16_t*)in, (long long)N, (long long)HW, C, in_cs, nseg, (bf16_t*)out, (long long)out_row,
                       out_third, out_c_off);
    return sos_check_launch("sos_spatial_mean");
}

// ---- backward of the two glue kernels (training of the variant)
// d_in[bt][pix][c] = sum_dt d_st[bt - (dt - pt)][pix][dt*C + c] over the frames of the same clip (the transpose of
// time_stack).  Gradients carry hi|hi|lo thirds in bf16x3 mode: terms are summed as hi + lo and re-split.
__global__ void time_unstack_kernel(const bf16_t* __restrict__ dst_, int T, long long HW, int C, int st_cs, int nseg, int kt,
                                    bf16_t* __restrict__ din, int in_cs, long long total) {
    const int pt = (kt - 1) / 2;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(i % in_cs);
        long long r = i / in_cs;
        const long long pix = r % HW;
        const long long bt = r / HW;
        const int t = (int)(bt % T);
        float acc = 0.f;
        if (c < C) {
            for (int dt = 0; dt < kt; ++dt) {
                const int tf = t - dt + pt;                       // frame whose stacked copy holds this frame at tap dt
                if (tf < 0 || tf >= T) continue;
                const bf16_t* px = dst_ + ((bt - dt + pt) * HW + pix) * ((long long)nseg * st_cs) + dt * C + c;
                acc += bf2f(px[0]);
                if (nseg == 3) acc += bf2f(px[2 * st_cs]);
            }
        }
        bf16_t* o = din + (bt * HW + pix) * ((long long)nseg * in_cs) + c;
        const bf16_t hi = f2bf(acc);
        o[0] = hi;
        if (nseg == 3) { o[in_cs] = hi; o[2 * in_cs] = f2bf(acc - bf2f(hi)); }
    }
}

extern "C" int sos_time_unstack(const void* d_stacked, int64_t B, int T, int64_t HW, int C, int st_cs, int nseg, int kt,
                                void* d_in, int in_cs, sos_stream_t stream) {
    if (!d_stacked || !d_in || B < 1 || T < 1 || HW < 1 || C < 1 || C > in_cs || (nseg != 1 && nseg != 3) || kt < 1 ||
        !(kt & 1) || st_cs < kt * C) {
        sos_set_error("sos_time_unstack: bad args");
        return SOS_EINVAL;
    }
    const long long total = (long long)B * T * HW * in_cs;
    long long grid = (total + 255) / 256;
    if (grid > 65536) grid = 65536;
    hipLaunchKernelGGL(time_unstack_kernel, dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)d_stacked, T,
                       (long long)HW, C, st_cs, nseg, kt, (bf16_t*)d_in, in_cs, total);
    return sos_check_launch("sos_time_unstack");
}

// dy[n][pix][c] = dfeat[n * f_row + f_c_off + c] / HW  (dfeat thirds f_third apart in bf16x3 mode); channels >= C zero
__global__ void spatial_mean_bwd_kernel(const bf16_t* __restrict__ dfeat, long long N, long long HW, int C, long long f_row,
                                        int f_third, int f_c_off, int nseg, bf16_t* __restrict__ dy, int cs) {
    const long long total = N * HW * cs;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(i % cs);
        const long long r = i / cs;
        const long long n = r / HW;
        float v = 0.f;
        if (c < C) {
            const bf16_t* f = dfeat + n * f_row + f_c_off + c;
            v = bf2f(f[0]);
            if (nseg == 3) v += bf2f(f[2 * f_third]);
            v /= (float)HW;
        }
        bf16_t* o = dy + r * ((long long)nseg * cs) + c;
        const bf16_t hi = f2bf(v);
        o[0] = hi;
        if (nseg == 3) { o[cs] = hi; o[2 * cs] = f2bf(v - bf2f(hi)); }
    }
}

extern "C" int sos_spatial_mean_bwd(const void* dfeat, int64_t N, int64_t HW, int C, int64_t f_row, int f_third, int f_c_off,
                                    int nseg, void* dy, int cs, sos_stream_t stream) {
    if (!dfeat || !dy || N < 1 || HW < 1 || C < 1 || C > cs || (nseg != 1 && nseg != 3)) {
        sos_set_error("sos_spatial_mean_bwd: bad args");
        return SOS_EINVAL;
    }
    const long long total = (long long)N * HW * cs;
    long long grid = (total + 255) / 256;
    if (grid > 65536) grid = 65536;
    hipLaunchKernelGGL(spatial_mean_bwd_kernel, dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)dfeat,
                       (long long)N, (long long)HW, C, (long long)f_row, f_third, f_c_off, nseg, (bf16_t*)dy, cs);
    return sos_check_launch("sos_spatial_mean_bwd");
}
