// wave_io.hip -- the front door of both models: decoded PCM -> mono f32 -> band-limited resampling to the
// model rate.
//
// Reference call sites: `librosa.load(path, sr=14000)` (M1/dataset.py:226, M2/predict.py:288,297,303,
// M2/dataset.py) = soundfile decode (integer PCM / 2^(bits-1)) -> to_mono (mean over channels) ->
// resampy.resample(filter='kaiser_best') -> zero-pad to ceil(n * ratio).  resampy is a third-party dependency
// that is absent here (requirements.txt: librosa==0.7.1 -> resampy>=0.2.2); these kernels follow its published
// algorithm (Smith's band-limited interpolation: a Kaiser-windowed sinc tabulated at 512 points per zero
// crossing, linearly interpolated, left wing + right wing per output sample).
//
// resample_kernel: one output sample per thread; the filter half-window (32 769 + 1 floats = 128 KB) lives
// in LDS, so the ~2 x 202 taps of an output at 44.1 -> 14 kHz cost one ds_read2_b32 (w[o], w[o+1]) and one
// L1-resident input read each.  A workgroup walks RS_PER_WG consecutive outputs to amortise the table load.
#include "sos_common.h"

#define RS_THREADS 256
#define RS_PER_WG 4096

template <typename T> struct Pcm;
template <> struct Pcm<int16_t> { static __device__ __forceinline__ float cvt(int16_t v) { return (float)v * (1.0f / 32768.0f); } };
template <> struct Pcm<int32_t> { static __device__ __forceinline__ float cvt(int32_t v) { return (float)((double)v * (1.0 / 2147483648.0)); } };
template <> struct Pcm<uint8_t> { static __device__ __forceinline__ float cvt(uint8_t v) { return ((float)v - 128.0f) * (1.0f / 128.0f); } };
template <> struct Pcm<float> { static __device__ __forceinline__ float cvt(float v) { return v; } };

// interleaved [n_frames][channels] -> mono f32 (np.mean over channels of the f32 samples)
template <typename T>
__global__ void pcm_to_mono_kernel(const T* __restrict__ pcm, int channels, int64_t n_frames, float* __restrict__ out) {
    const float inv = 1.0f / (float)channels;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_frames; i += (int64_t)gridDim.x * blockDim.x) {
        float s = 0.f;
        for (int c = 0; c < channels; ++c) s += Pcm<T>::cvt(pcm[i * channels + c]);
        out[i] = channels == 1 ? s : s * inv;
    }
}

extern "C" int sos_pcm_to_mono_f32(const void* pcm, int format, int channels, int64_t n_frames, float* out,
                                   sos_stream_t stream) {
    if (!pcm || !out || channels < 1 || channels > 64 || n_frames < 1) {
        sos_set_error("sos_pcm_to_mono_f32: bad args (channels=%d, n_frames=%lld)", channels, (long long)n_frames);
        return SOS_EINVAL;
    }
    const unsigned grid = (unsigned)((n_frames + RS_THREADS - 1) / RS_THREADS < 4096 ? (n_frames + RS_THREADS - 1) / RS_THREADS : 4096);
    hipStream_t st = (hipStream_t)stream;
    switch (format) {
    case SOS_PCM_S16: hipLaunchKernelGGL(pcm_to_mono_kernel<int16_t>, dim3(grid), dim3(RS_THREADS), 0, st, (const int16_t*)pcm, channels, n_frames, out); break;
    case SOS_PCM_S32: hipLaunchKernelGGL(pcm_to_mono_kernel<int32_t>, dim3(grid), dim3(RS_THREADS), 0, st, (const int32_t*)pcm, channels, n_frames, out); break;
    case SOS_PCM_U8: hipLaunchKernelGGL(pcm_to_mono_kernel<uint8_t>, dim3(grid), dim3(RS_THREADS), 0, st, (const uint8_t*)pcm, channels, n_frames, out); break;
    case SOS_PCM_F32: hipLaunchKernelGGL(pcm_to_mono_kernel<float>, dim3(grid), dim3(RS_THREADS), 0, st, (const float*)pcm, channels, n_frames, out); break;
    default: sos_set_error("sos_pcm_to_mono_f32: unknown sample format %d", format); return SOS_EINVAL;
    }
    return sos_check_launch("sos_pcm_to_mono_f32");
}

// resampy keeps the source time of output t in a RUNNING f64 sum (time_register += 1/ratio), and because it
// walks the filter table with the truncated step int(scale * num_table), its result is not continuous where
// that time crosses an integer -- which the exact time t * 441/140 does at every 140th output of 44.1 -> 14 kHz.
// Matching it therefore needs the running sum bit for bit.  A sequential f64 sum is piecewise linear: while the
// register and register + inc share a binade [2^e, 2^(e+1)) every addition rounds on the same grid and adds the
// same d = rn_grid(inc).  The host builds that piecewise description (a few entries per binade: one explicit
// step across each binade boundary, one where a round-to-even tie settles, then a run) and the kernel evaluates
// time(t) = fma(t - k0, d, s0) exactly.
#define RS_MAXSEG 120
struct RsSegs {
    int n;
    long long k0[RS_MAXSEG];
    double s0[RS_MAXSEG], d[RS_MAXSEG];
};

static int rs_build_segments(double inc, long long n_out, RsSegs* g) {
    long long k = 0;
    double s = 0.0;
    g->n = 0;
    while (k < n_out) {
        if (g->n == RS_MAXSEG) return -1;
        const int i = g->n++;
        volatile double s1v = s + inc;                   // one literal step of the running sum
        const double s1 = s1v;
        g->k0[i] = k; g->s0[i] = s; g->d[i] = s1 - s;
        int e;
        const double limit = s > 0.0 ? ldexp(1.0, (frexp(s, &e), e)) : 0.0;      // top of s's binade
        volatile double s2v = s1 + inc;
        const double s2 = s2v;
        if (!(s > 0.0) || !(s2 < limit) || (s2 - s1) != (s1 - s)) { k += 1; s = s1; continue; }   // explicit single step
        const double d = s1 - s;
        // largest M with s + M*d + inc < limit: steps k .. k+M+1 all add d
        long long M = (long long)((limit - inc - s) / d);
        if (M < 1) M = 1;
        for (;;) { volatile double v = fma((double)M, d, s) + inc; if (v < limit || M == 1) break; --M; }
        for (;;) { volatile double v = fma((double)(M + 1), d, s) + inc; if (!(v < limit)) break; ++M; }
        k += M + 1;
        s = fma((double)(M + 1), d, s);
    }
    return 0;
}

extern "C" int sos_resample_time_segments(double ratio, int64_t n_out, int64_t* k0, double* s0, double* d, int capacity) {
    RsSegs g;
    if (!(ratio > 0.0) || n_out < 1 || !k0 || !s0 || !d) { sos_set_error("sos_resample_time_segments: bad args"); return SOS_EINVAL; }
    if (rs_build_segments(1.0 / ratio, n_out, &g) != 0 || g.n > capacity) {
        sos_set_error("sos_resample_time_segments: more than %d segments", capacity < RS_MAXSEG ? capacity : RS_MAXSEG);
        return SOS_ENOSPC;
    }
    for (int i = 0; i < g.n; ++i) { k0[i] = g.k0[i]; s0[i] = g.s0[i]; d[i] = g.d[i]; }
    return g.n;
}

// win: half window already scaled by min(1, ratio), nwin entries.  Output t:
//   time = resampy's running sum (RsSegs); n = floor(time); frac = scale * (time - n)
//   left wing : sum_i (w[o + i*step] + eta * (w[o + i*step + 1] - w[o + i*step])) * x[n - i],    o = int(frac * num_table)
//   right wing: the same with frac' = scale - frac over x[n + 1 + k]
// where the difference table has a zero last entry (resampy: interp_delta[-1] = 0): LDS holds w[nwin] = w[nwin-1].
__global__ __launch_bounds__(RS_THREADS) void resample_kernel(const float* __restrict__ x, int64_t n_in, RsSegs seg,
                                                              double scale, const float* __restrict__ win, int nwin,
                                                              int num_table, int index_step, float* __restrict__ out,
                                                              int64_t n_out, int64_t n_valid) {
    extern __shared__ float w[];
    for (int i = threadIdx.x; i <= nwin; i += RS_THREADS) w[i] = win[i < nwin ? i : nwin - 1];
    __syncthreads();
    const int64_t t0 = (int64_t)blockIdx.x * RS_PER_WG;
    for (int64_t t = t0 + threadIdx.x; t < t0 + RS_PER_WG && t < n_out; t += RS_THREADS) {
        if (t >= n_valid) { out[t] = 0.f; continue; }    // librosa fix_length: zero padding up to ceil(n * ratio)
        int sg = seg.n - 1;
        while (sg > 0 && seg.k0[sg] > t) --sg;
        const double time = fma((double)(t - seg.k0[sg]), seg.d[sg], seg.s0[sg]);
        const int64_t n = (int64_t)time;
        double frac = scale * (time - (double)n);
        double index_frac = frac * (double)num_table;
        int o = (int)index_frac;
        float eta = (float)(index_frac - (double)o);
        float acc = 0.f;
        {
            const int64_t lim = (int64_t)((nwin - o) / index_step);
            const int i_max = (int)(n + 1 < lim ? n + 1 : lim);
            const float* xp = x + n;
            for (int i = 0; i < i_max; ++i, o += index_step) {
                const float a = w[o], b = w[o + 1];
                acc = fmaf(fmaf(eta, b - a, a), xp[-i], acc);
            }
        }
        frac = scale - frac;
        index_frac = frac * (double)num_table;
        o = (int)index_frac;
        eta = (float)(index_frac - (double)o);
        {
            const int64_t lim = (int64_t)((nwin - o) / index_step);
            const int64_t rem = n_in - n - 1;
            const int k_max = (int)(rem < lim ? rem : lim);
            const float* xp = x + n + 1;
            for (int k = 0; k < k_max; ++k, o += index_step) {
                const float a = w[o], b = w[o + 1];
                acc = fmaf(fmaf(eta, b - a, a), xp[k], acc);
            }
        }
        out[t] = acc;
    }
}

extern "C" int sos_resample_f32(const float* x, int64_t n_in, double ratio, const float* win, int nwin, int num_table,
                                float* out, int64_t n_out, sos_stream_t stream) {
    if (!x || !win || !out || n_in < 1 || n_out < 1 || !(ratio > 0.0) || nwin < 2 || num_table < 1 ||
        (size_t)(nwin + 1) * sizeof(float) > 160 * 1024) {
        sos_set_error("sos_resample_f32: bad args (n_in=%lld, n_out=%lld, ratio=%g, nwin=%d)", (long long)n_in,
                      (long long)n_out, ratio, nwin);
        return SOS_EINVAL;
    }
    const double scale = ratio < 1.0 ? ratio : 1.0;
    const int index_step = (int)(scale * (double)num_table);
    if (index_step < 1) { sos_set_error("sos_resample_f32: ratio %g too small for a %d-point table", ratio, num_table); return SOS_EINVAL; }
    int64_t n_valid = (int64_t)((double)n_in * ratio);      // resampy's output length; the rest is librosa's padding
    if (n_valid > n_out) n_valid = n_out;
    const size_t lds = (size_t)(nwin + 1) * sizeof(float);
    static sos_device_once attr_once;
    if (sos_per_device_once(attr_once, [] {
            if (hipFuncSetAttribute((const void*)resample_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) {
                sos_set_error("sos_resample_f32: cannot raise the dynamic LDS limit");
                return (int)SOS_ELAUNCH;
            }
            return (int)SOS_OK;
        }))
        return SOS_ELAUNCH;
    RsSegs seg;
    if (rs_build_segments(1.0 / ratio, n_valid > 0 ? n_valid : 1, &seg) != 0) {
        sos_set_error("sos_resample_f32: time register needs more than %d segments", RS_MAXSEG);
        return SOS_ENOSPC;
    }
    const unsigned grid = (unsigned)((n_out + RS_PER_WG - 1) / RS_PER_WG);
    hipLaunchKernelGGL(resample_kernel, dim3(grid), dim3(RS_THREADS), lds, (hipStream_t)stream, x, n_in, seg, scale,
                       win, nwin, num_table, index_step, out, n_out, n_valid);
    return sos_check_launch("sos_resample_f32");
}
