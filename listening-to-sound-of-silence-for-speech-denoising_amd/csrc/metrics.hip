// metrics.hip -- per-sample / per-frame work of the objective measures of M2/metrics.py:
// segmental SNR family (metrics_ssnr :86-130, _shift :132-176, _exclude_silence :178-244), llr :561-623 with
// lpcoeff :626-681, wss :404-558, metrics_L1 :40-45.  The kernels reduce the signals to per-frame quantities (frame
// energies, LPC log-likelihood ratios, weighted spectral slope distances); the few-thousand-element finalisation
// (log10 / clamp / mean / trimmed mean / composite formulas) is host code in sos_amd/metrics.py.
// Frames: start = f * skip, `winlength` samples, window w[i] = 0.5 (1 - cos(2 pi (i+1) / (winlength+1))).
#include "sos_common.h"

#define MT 256

__device__ __forceinline__ double block_sum(double v, double* red) {
    const int tid = threadIdx.x;
    red[tid] = v;
    __syncthreads();
    for (int s = MT / 2; s > 0; s >>= 1) {
        if (tid < s) red[tid] += red[tid + s];
        __syncthreads();
    }
    const double r = red[0];
    __syncthreads();
    return r;
}

// out[0] = sum ref^2, out[1] = sum (ref - deg)^2, out[2] = max |ref|   (single workgroup: evaluation sizes)
__global__ __launch_bounds__(MT) void metric_totals_kernel(const float* __restrict__ ref, const float* __restrict__ deg,
                                                            long long n, double* __restrict__ out) {
    __shared__ double red[MT];
    double a = 0, b = 0, m = 0;
    for (long long i = threadIdx.x; i < n; i += MT) {
        const double r = ref[i], d = r - (double)deg[i];
        a += r * r; b += d * d;
        m = fmax(m, fabs(r));
    }
    const double sa = block_sum(a, red), sb = block_sum(b, red);
    red[threadIdx.x] = m;
    __syncthreads();
    for (int s = MT / 2; s > 0; s >>= 1) {
        if (threadIdx.x < s) red[threadIdx.x] = fmax(red[threadIdx.x], red[threadIdx.x + s]);
        __syncthreads();
    }
    if (threadIdx.x == 0) { out[0] = sa; out[1] = sb; out[2] = red[0]; }
}

// one workgroup per frame: out[f][0] = sum (w c)^2, out[f][1] = sum (w c - w p)^2
__global__ __launch_bounds__(MT) void metric_frame_energy_kernel(const float* __restrict__ ref, const float* __restrict__ deg,
                                                                  int winlength, int skip, const double* __restrict__ window,
                                                                  double* __restrict__ out) {
    __shared__ double red[MT];
    const long long start = (long long)blockIdx.x * skip;
    double a = 0, b = 0;
    for (int i = threadIdx.x; i < winlength; i += MT) {
        const double c = (double)ref[start + i] * window[i], p = (double)deg[start + i] * window[i];
        a += c * c; b += (c - p) * (c - p);
    }
    const double sa = block_sum(a, red), sb = block_sum(b, red);
    if (threadIdx.x == 0) { out[2 * blockIdx.x] = sa; out[2 * blockIdx.x + 1] = sb; }
}

// order-preserving compaction of the samples with |clean| >= thr (single workgroup, chunked prefix scan)
__global__ __launch_bounds__(MT) void metric_compact_kernel(const float* __restrict__ clean, const float* __restrict__ proc,
                                                             long long n, float thr, float* __restrict__ oc,
                                                             float* __restrict__ op, long long* __restrict__ count) {
    __shared__ int scan[MT];
    __shared__ long long base;
    if (threadIdx.x == 0) base = 0;
    __syncthreads();
    for (long long c0 = 0; c0 < n; c0 += MT) {
        const long long i = c0 + threadIdx.x;
        const int keep = (i < n && !(fabsf(clean[i]) < thr)) ? 1 : 0;
        scan[threadIdx.x] = keep;
        __syncthreads();
        for (int s = 1; s < MT; s <<= 1) {
            const int v = threadIdx.x >= s ? scan[threadIdx.x - s] : 0;
            __syncthreads();
            scan[threadIdx.x] += v;
            __syncthreads();
        }
        if (keep) { const long long o = base + scan[threadIdx.x] - 1; oc[o] = clean[i]; op[o] = proc[i]; }
        __syncthreads();
        if (threadIdx.x == 0) base += scan[MT - 1];
        __syncthreads();
    }
    if (threadIdx.x == 0) *count = base;
}

// LLR of one frame per workgroup: autocorrelation lags 0..P of both windowed frames (f64), Levinson-Durbin (lane 0),
// then -- like the reference, which casts R and the LPC vectors to float32 first -- the two quadratic forms in f32.
#define LLR_MAXP 16
__global__ __launch_bounds__(MT) void metric_llr_kernel(const float* __restrict__ ref, const float* __restrict__ deg, int winlength,
                                                         int skip, const double* __restrict__ window, int P,
                                                         float* __restrict__ out) {
    extern __shared__ double fr[];                      // [2][winlength]
    __shared__ double red[MT];
    __shared__ double R[2][LLR_MAXP + 1];
    const long long start = (long long)blockIdx.x * skip;
    for (int i = threadIdx.x; i < winlength; i += MT) {
        fr[i] = (double)ref[start + i] * window[i];
        fr[winlength + i] = (double)deg[start + i] * window[i];
    }
    __syncthreads();
    for (int s = 0; s < 2; ++s)
        for (int k = 0; k <= P; ++k) {
            double a = 0;
            for (int i = threadIdx.x; i < winlength - k; i += MT) a += fr[s * winlength + i] * fr[s * winlength + i + k];
            const double v = block_sum(a, red);
            if (threadIdx.x == 0) R[s][k] = v;
        }
    __syncthreads();
    if (threadIdx.x == 0) {
        float A[2][LLR_MAXP + 1];
        for (int s = 0; s < 2; ++s) {
            double a[LLR_MAXP], ap[LLR_MAXP], E = R[s][0];
            for (int i = 0; i < P; ++i) a[i] = 1.0;
            for (int i = 0; i < P; ++i) {
                double sum = 0;
                for (int j = 0; j < i; ++j) { ap[j] = a[j]; sum += a[j] * R[s][i - j]; }
                const double rc = (R[s][i + 1] - sum) / E;
                a[i] = rc;
                for (int j = 0; j < i; ++j) a[j] = ap[j] - rc * ap[i - 1 - j];
                E = (1.0 - rc * rc) * E;
            }
            A[s][0] = 1.f;
            for (int i = 0; i < P; ++i) A[s][i + 1] = (float)(-a[i]);
        }
        float Rc[LLR_MAXP + 1];
        for (int k = 0; k <= P; ++k) Rc[k] = (float)R[0][k];
        float num = 0.f, den = 0.f;
        for (int i = 0; i <= P; ++i) {                   // row vector . toeplitz(Rc), then . column vector
            float tn = 0.f, td = 0.f;
            for (int j = 0; j <= P; ++j) {
                const float r = Rc[i > j ? i - j : j - i];
                tn += A[1][j] * r; td += A[0][j] * r;
            }
            num += tn * A[1][i]; den += td * A[0][i];
        }
        out[blockIdx.x] = logf(num / den);
    }
}

// WSS of one frame per workgroup: |DFT|^2 of both windowed frames on bins 0..n_fft/2-1 (direct DFT, twiddles in LDS),
// 25 critical-band energies -> dB -> slopes -> weighted distance (sequential part on lane 0).
#define WSS_NCRIT 25
__global__ __launch_bounds__(MT) void metric_wss_kernel(const float* __restrict__ ref, const float* __restrict__ deg, int winlength,
                                                         int skip, const double* __restrict__ window, int n_fft,
                                                         const float* __restrict__ crit, double eps, float* __restrict__ out) {
    extern __shared__ float sm[];                       // frames [2][winlength], twiddle cos/sin [n_fft] each, spectra [2][n_fft/2]
    float* fc = sm;
    float* fp = sm + winlength;
    float* tc = fp + winlength;
    float* ts = tc + n_fft;
    float* sp = ts + n_fft;
    __shared__ double red[MT];
    __shared__ double en[2][WSS_NCRIT];
    const int half = n_fft / 2;
    const long long start = (long long)blockIdx.x * skip;
    for (int i = threadIdx.x; i < winlength; i += MT) {
        fc[i] = (float)((double)ref[start + i] * window[i]);
        fp[i] = (float)((double)deg[start + i] * window[i]);
    }
    for (int i = threadIdx.x; i < n_fft; i += MT) {
        double s, c;
        sincos(-2.0 * 3.14159265358979323846 * (double)i / (double)n_fft, &s, &c);
        tc[i] = (float)c; ts[i] = (float)s;
    }
    __syncthreads();
    for (int k = threadIdx.x; k < half; k += MT) {
        float cr = 0.f, ci = 0.f, pr = 0.f, pi = 0.f;
        int ph = 0;
        for (int i = 0; i < winlength; ++i) {
            const float c = tc[ph], s = ts[ph];
            cr = fmaf(fc[i], c, cr); ci = fmaf(fc[i], s, ci);
            pr = fmaf(fp[i], c, pr); pi = fmaf(fp[i], s, pi);
            ph += k; if (ph >= n_fft) ph -= n_fft;
        }
        sp[k] = cr * cr + ci * ci;
        sp[half + k] = pr * pr + pi * pi;
    }
    __syncthreads();
    for (int s = 0; s < 2; ++s)
        for (int b = 0; b < WSS_NCRIT; ++b) {
            double a = 0;
            for (int k = threadIdx.x; k < half; k += MT) a += (double)sp[s * half + k] * (double)crit[b * half + k];
            const double v = block_sum(a, red);
            if (threadIdx.x == 0) en[s][b] = 10.0 * log10(fmax(v, eps));
        }
    __syncthreads();
    if (threadIdx.x == 0) {
        const int NC = WSS_NCRIT;
        double slope[2][WSS_NCRIT - 1], peak[2][WSS_NCRIT - 1], dbmax[2];
        for (int s = 0; s < 2; ++s) {
            dbmax[s] = en[s][0];
            for (int b = 1; b < NC; ++b) dbmax[s] = fmax(dbmax[s], en[s][b]);
            for (int b = 0; b < NC - 1; ++b) slope[s][b] = en[s][b + 1] - en[s][b];
            for (int i = 0; i < NC - 1; ++i) {
                int n = i;
                if (slope[s][i] > 0) {
                    while (n < NC - 1 && slope[s][n] > 0) ++n;
                    peak[s][i] = en[s][n - 1];
                } else {
                    while (n >= 0 && slope[s][n] <= 0) --n;
                    peak[s][i] = en[s][n + 1];
                }
            }
        }
        double num = 0, den = 0;
        for (int b = 0; b < NC - 1; ++b) {
            double W = 0;
            for (int s = 0; s < 2; ++s)
                W += (20.0 / (20.0 + dbmax[s] - en[s][b])) * (1.0 / (1.0 + peak[s][b] - en[s][b]));
            W *= 0.5;
            const double d = slope[0][b] - slope[1][b];
            num += W * d * d; den += W;
        }
        out[blockIdx.x] = (float)(num / den);
    }
}

// mean | lerp(output)(steps) - target |, steps = linspace(0, n_out - 1, n_t)   (scipy interp1d + np.linspace)
__global__ __launch_bounds__(MT) void metric_l1_kernel(const float* __restrict__ outp, long long n_out, const float* __restrict__ target,
                                                        long long n_t, double* __restrict__ res) {
    __shared__ double red[MT];
    double a = 0;
    const double step = n_t > 1 ? (double)(n_out - 1) / (double)(n_t - 1) : 0.0;
    for (long long j = threadIdx.x; j < n_t; j += MT) {
        double x = (double)j * step;
        if (j == n_t - 1) x = (double)(n_out - 1);
        long long i0 = (long long)x;
        if (i0 > n_out - 2) i0 = n_out - 2 > 0 ? n_out - 2 : 0;
        const double f = x - (double)i0;
        const double v = n_out > 1 ? (double)outp[i0] + f * ((double)outp[i0 + 1] - (double)outp[i0]) : (double)outp[0];
        a += fabs(v - (double)target[j]);
    }
    const double s = block_sum(a, red);
    if (threadIdx.x == 0) res[0] = s / (double)n_t;
}

#define MCHK(c, name) if (!(c)) { sos_set_error(name ": bad args"); return SOS_EINVAL; }

extern "C" int sos_metric_totals(const float* ref, const float* deg, int64_t n, double* out3, sos_stream_t stream) {
    MCHK(ref && deg && out3 && n > 0, "sos_metric_totals")
    hipLaunchKernelGGL(metric_totals_kernel, dim3(1), dim3(MT), 0, (hipStream_t)stream, ref, deg, (long long)n, out3);
    return sos_check_launch("sos_metric_totals");
}
extern "C" int sos_metric_frame_energy(const float* ref, const float* deg, int64_t n, int winlength, int skip, int64_t num_frames,
                                       const double* window, double* out, sos_stream_t stream) {
    MCHK(ref && deg && window && out && winlength > 0 && skip > 0 && num_frames > 0 && (num_frames - 1) * skip + winlength <= n,
         "sos_metric_frame_energy")
    hipLaunchKernelGGL(metric_frame_energy_kernel, dim3((unsigned)num_frames), dim3(MT), 0, (hipStream_t)stream, ref, deg, winlength,
                       skip, window, out);
    return sos_check_launch("sos_metric_frame_energy");
}
extern "C" int sos_metric_compact(const float* clean, const float* proc, int64_t n, float thr, float* out_clean, float* out_proc,
                                  int64_t* count, sos_stream_t stream) {
    MCHK(clean && proc && out_clean && out_proc && count && n > 0, "sos_metric_compact")
    hipLaunchKernelGGL(metric_compact_kernel, dim3(1), dim3(MT), 0, (hipStream_t)stream, clean, proc, (long long)n, thr, out_clean,
                       out_proc, (long long*)count);
    return sos_check_launch("sos_metric_compact");
}
extern "C" int sos_metric_llr(const float* ref, const float* deg, int64_t n, int winlength, int skip, int64_t num_frames,
                              const double* window, int P, float* out, sos_stream_t stream) {
    MCHK(ref && deg && window && out && winlength > 0 && skip > 0 && num_frames > 0 && P > 0 && P <= LLR_MAXP && P < winlength &&
         (num_frames - 1) * skip + winlength <= n && (size_t)winlength * 16 <= 64 * 1024, "sos_metric_llr")
    hipLaunchKernelGGL(metric_llr_kernel, dim3((unsigned)num_frames), dim3(MT), (size_t)winlength * 16, (hipStream_t)stream, ref, deg,
                       winlength, skip, window, P, out);
    return sos_check_launch("sos_metric_llr");
}
extern "C" int sos_metric_wss(const float* ref, const float* deg, int64_t n, int winlength, int skip, int64_t num_frames,
                              const double* window, int n_fft, const float* crit_filter, double eps, float* out, sos_stream_t stream) {
    const size_t lds = ((size_t)2 * winlength + 2 * (size_t)n_fft + (size_t)n_fft) * 4;
    MCHK(ref && deg && window && crit_filter && out && winlength > 0 && skip > 0 && num_frames > 0 && n_fft >= winlength &&
         (n_fft & (n_fft - 1)) == 0 && (num_frames - 1) * skip + winlength <= n && lds <= 60 * 1024, "sos_metric_wss")
    hipLaunchKernelGGL(metric_wss_kernel, dim3((unsigned)num_frames), dim3(MT), lds, (hipStream_t)stream, ref, deg, winlength, skip,
                       window, n_fft, crit_filter, eps, out);
    return sos_check_launch("sos_metric_wss");
}
extern "C" int sos_metric_l1(const float* output, int64_t n_out, const float* target, int64_t n_t, double* result, sos_stream_t stream) {
    MCHK(output && target && result && n_out > 0 && n_t > 0, "sos_metric_l1")
    hipLaunchKernelGGL(metric_l1_kernel, dim3(1), dim3(MT), 0, (hipStream_t)stream, output, (long long)n_out, target, (long long)n_t, result);
    return sos_check_launch("sos_metric_l1");
}
