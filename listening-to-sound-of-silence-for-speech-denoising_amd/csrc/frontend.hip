// frontend.hip -- complex-ratio-mask / add_signals / bits->sample-mask / layout kernels (gfx950); the STFT and ISTFT
// live in stft_mfma.hip.
//
// Reference semantics (paths relative to /root/reference, M1 = model_1_silent_interval_detection/
// audioonly_model, M2 = model_2_audio_denoising/audio_denoising_model):
//   fast_stft  M1/transform.py:188-193   fast_istft M1/transform.py:196-202
//   batch_fast_icRM_sigmoid M1/transform.py:156-169   fast_cRM_sigmoid :130-138
//   convert_bitstreammask_to_audiomask M2/tools.py:340-362
#include "sos_common.h"
#include <stdlib.h>

// ------------------------------------------------------------------- complex ratio mask ops
__device__ __forceinline__ float crm_recover(float c, float inv_a, float b) {
    // 1/a * (log(c / (1 - c + 1e-8) + 1e-10) + b), same fp32 operation order as the torch code
    return inv_a * (logf(c / (1.0f - c + 1e-8f) + 1e-10f) + b);
}

__global__ void crm_apply_kernel(const float* __restrict__ Y, const float* __restrict__ crm, float* __restrict__ rec,
                                 int64_t total, int64_t plane, float inv_a, float b) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t bb = i / plane, r = i - bb * plane;
        const int64_t o = bb * 2 * plane + r;
        const float yr = Y[o], yi = Y[o + plane];
        const float mr = crm_recover(crm[o], inv_a, b), mi = crm_recover(crm[o + plane], inv_a, b);
        rec[o] = mr * yr - mi * yi;
        rec[o + plane] = mr * yi + mi * yr;
    }
}

__global__ void crm_apply_bwd_kernel(const float* __restrict__ Y, const float* __restrict__ crm,
                                     const float* __restrict__ g, float* __restrict__ gc, int64_t total,
                                     int64_t plane, float inv_a) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t bb = i / plane, r = i - bb * plane;
        const int64_t o = bb * 2 * plane + r;
        const float yr = Y[o], yi = Y[o + plane];
        const float gr = g[o], gi = g[o + plane];
        const float dmr = gr * yr + gi * yi;
        const float dmi = gi * yr - gr * yi;
        const float c0 = crm[o], c1 = crm[o + plane];
        const float d0 = 1.0f - c0 + 1e-8f, d1 = 1.0f - c1 + 1e-8f;
        // dM/dc = (1/a) * (1+1e-8) / (den^2 * (c/den + 1e-10))
        gc[o] = dmr * inv_a * (1.0f + 1e-8f) / (d0 * d0 * (c0 / d0 + 1e-10f));
        gc[o + plane] = dmi * inv_a * (1.0f + 1e-8f) / (d1 * d1 * (c1 / d1 + 1e-10f));
    }
}

__global__ void crm_target_kernel(const float* __restrict__ S, const float* __restrict__ Y, float* __restrict__ out,
                                  int64_t total, int64_t plane, float a, float b) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t bb = i / plane, r = i - bb * plane;
        const int64_t o = bb * 2 * plane + r;
        const float yr = Y[o], yi = Y[o + plane], sr = S[o], si = S[o + plane];
        const float den = yr * yr + yi * yi + 1e-8f;
        const float mr = (yr * sr + yi * si) / den;
        const float mi = (yr * si - yi * sr) / den;
        out[o] = 1.0f / (1.0f + expf(-a * mr + b));
        out[o + plane] = 1.0f / (1.0f + expf(-a * mi + b));
    }
}

static inline unsigned ew_grid(int64_t total) {
    int64_t g = (total + 255) / 256;
    if (g > 4096) g = 4096;
    if (g < 1) g = 1;
    return (unsigned)g;
}

extern "C" int sos_crm_apply_f32(const float* Y, const float* crm, float* rec, int64_t batch, int64_t plane,
                                 float a, float b, sos_stream_t stream) {
    if (!Y || !crm || !rec || batch < 1 || plane < 1 || a == 0.f) { sos_set_error("sos_crm_apply_f32: bad args"); return SOS_EINVAL; }
    const int64_t total = batch * plane;
    hipLaunchKernelGGL(crm_apply_kernel, dim3(ew_grid(total)), dim3(256), 0, (hipStream_t)stream, Y, crm, rec, total,
                       plane, 1.0f / a, b);
    return sos_check_launch("sos_crm_apply_f32");
}

extern "C" int sos_crm_apply_bwd_f32(const float* Y, const float* crm, const float* grad_rec, float* grad_crm,
                                     int64_t batch, int64_t plane, float a, sos_stream_t stream) {
    if (!Y || !crm || !grad_rec || !grad_crm || batch < 1 || plane < 1 || a == 0.f) {
        sos_set_error("sos_crm_apply_bwd_f32: bad args");
        return SOS_EINVAL;
    }
    const int64_t total = batch * plane;
    hipLaunchKernelGGL(crm_apply_bwd_kernel, dim3(ew_grid(total)), dim3(256), 0, (hipStream_t)stream, Y, crm,
                       grad_rec, grad_crm, total, plane, 1.0f / a);
    return sos_check_launch("sos_crm_apply_bwd_f32");
}

extern "C" int sos_crm_target_f32(const float* clean, const float* mix, float* out, int64_t batch, int64_t plane,
                                  float a, float b, sos_stream_t stream) {
    if (!clean || !mix || !out || batch < 1 || plane < 1) { sos_set_error("sos_crm_target_f32: bad args"); return SOS_EINVAL; }
    const int64_t total = batch * plane;
    hipLaunchKernelGGL(crm_target_kernel, dim3(ew_grid(total)), dim3(256), 0, (hipStream_t)stream, clean, mix, out,
                       total, plane, a, b);
    return sos_check_launch("sos_crm_target_f32");
}

// ---------------------------------------------------------- a16: mix speech and noise at an SNR
// One workgroup per clip.  Pass 1: energies of the signal and of each noise (f64 sums: the clip is 28 000 .. 1e6
// samples); pass 2: each noise is rescaled so that E_signal / E_noise = 10^(snr/10) (left as it is when the signal or
// the noise is silent) and added, the peak of the sum is taken; pass 3: mixed, signal and noises are divided by
// peak / norm (norm == 0 or a silent sum: no normalisation).  M2/tools.py:217-276.
__global__ __launch_bounds__(256) void add_signals_kernel(const float* __restrict__ sig, const float* __restrict__ noi,
                                                          const float* __restrict__ snr_db, int K, long long n, float norm,
                                                          float* __restrict__ mixed, float* __restrict__ sig_out,
                                                          float* __restrict__ noi_out) {
    __shared__ double red[256];
    __shared__ float gain[9];                 // [k < 8] noise gains, [8] final 1/scale
    const long long b = blockIdx.x;
    const int tid = threadIdx.x;
    const float* s = sig + b * n;
    auto block_sum = [&](double v) {
        red[tid] = v;
        __syncthreads();
        for (int st = 128; st > 0; st >>= 1) {
            if (tid < st) red[tid] += red[tid + st];
            __syncthreads();
        }
        const double r = red[0];
        __syncthreads();
        return r;
    };
    double e = 0.0;
    for (long long i = tid; i < n; i += 256) e += (double)s[i] * (double)s[i];
    const double e_sig = block_sum(e);
    const double target = e_sig / pow(10.0, (double)snr_db[b] / 10.0);        // wanted noise energy
    for (int k = 0; k < K; ++k) {
        const float* nz = noi + (b * K + k) * n;
        double en = 0.0;
        for (long long i = tid; i < n; i += 256) en += (double)nz[i] * (double)nz[i];
        en = block_sum(en);
        if (tid == 0) gain[k] = (e_sig == 0.0 || en == 0.0) ? 1.f : (float)(sqrt(target) / sqrt(en));
    }
    __syncthreads();
    float peak = 0.f;
    for (long long i = tid; i < n; i += 256) {
        float m = s[i];
        for (int k = 0; k < K; ++k) m += noi[(b * K + k) * n + i] * gain[k];
        mixed[b * n + i] = m;
        peak = fmaxf(peak, fabsf(m));
    }
    red[tid] = (double)peak;
    __syncthreads();
    for (int st = 128; st > 0; st >>= 1) {
        if (tid < st) red[tid] = fmax(red[tid], red[tid + st]);
        __syncthreads();
    }
    if (tid == 0) gain[8] = (norm != 0.f && red[0] != 0.0) ? (float)((double)norm / red[0]) : 1.f;
    __syncthreads();
    const float inv = gain[8];
    for (long long i = tid; i < n; i += 256) {
        mixed[b * n + i] *= inv;
        sig_out[b * n + i] = s[i] * inv;
        for (int k = 0; k < K; ++k) noi_out[(b * K + k) * n + i] = noi[(b * K + k) * n + i] * gain[k] * inv;
    }
}

extern "C" int sos_add_signals_f32(const float* signal, const float* noises, const float* snr_db, int64_t batch, int n_noises,
                                   int64_t n, float norm, float* mixed, float* signal_out, float* noises_out,
                                   sos_stream_t stream) {
    if (!signal || !noises || !snr_db || !mixed || !signal_out || !noises_out || batch < 1 || n < 1 || n_noises < 1 ||
        n_noises > 8) {
        sos_set_error("sos_add_signals_f32: bad args (1..8 noises per clip)");
        return SOS_EINVAL;
    }
    hipLaunchKernelGGL(add_signals_kernel, dim3((unsigned)batch), dim3(256), 0, (hipStream_t)stream, signal, noises, snr_db,
                       n_noises, (long long)n, norm, mixed, signal_out, noises_out);
    return sos_check_launch("sos_add_signals_f32");
}

// ------------------------------------------------------------ power-law companding of a waveform
// power_law, M1/transform.py:178-185: sign(x) * |x|^p (p = 0.3 before the STFT, 1/0.3 after the ISTFT when the
// reference's `power=True` switch is on; its callers leave it off).
__global__ void power_law_kernel(const float* __restrict__ x, long long n, float p, float* __restrict__ out) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const float v = x[i];
        const float m = powf(fabsf(v), p);
        out[i] = v >= 0.f ? m : -m;
    }
}
extern "C" int sos_power_law_f32(const float* x, int64_t n, float power, float* out, sos_stream_t stream) {
    if (!x || !out || n < 1 || !(power > 0.f)) { sos_set_error("sos_power_law_f32: bad args"); return SOS_EINVAL; }
    hipLaunchKernelGGL(power_law_kernel, dim3(ew_grid(n)), dim3(256), 0, (hipStream_t)stream, x, (long long)n, power, out);
    return sos_check_launch("sos_power_law_f32");
}

// --------------------------------------------------------------------- bits -> sample mask
// Pre-flip mask value of sample j: 1 if j lies in [int(i*r), int((i+1)*r - 1)) of a silent
// frame i (bit 0), else 0.  All index arithmetic in IEEE double with explicit (un-fused)
// multiply/add so it reproduces Python's float64 evaluation bit for bit.
__device__ __forceinline__ int premask(const uint8_t* bits, int64_t n_frames, double ratio, int64_t j) {
    int64_t i0 = (int64_t)((double)j / ratio);
    for (int64_t i = i0 - 1; i <= i0 + 1; ++i) {
        if (i < 0 || i >= n_frames) continue;
        const int64_t lo = (int64_t)__dmul_rn((double)i, ratio);
        const int64_t hi = (int64_t)__dadd_rn(__dmul_rn((double)(i + 1), ratio), -1.0);
        if (j >= lo && j < hi) return bits[i] == 0 ? 1 : 0;
    }
    return 0;
}

__global__ void bits_to_mask_kernel(const uint8_t* __restrict__ bits, int64_t n_frames, double ratio,
                                    int64_t n_samples, float* __restrict__ mask, const float* __restrict__ sig,
                                    float* __restrict__ masked, const int* __restrict__ fr_tab,
                                    const int* __restrict__ ns_tab) {
    const int64_t b = blockIdx.y;
    const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t pitch_s = n_samples, pitch_f = n_frames;       // ragged batch: row pitches stay the batch maxima
    if (fr_tab) { n_frames = fr_tab[b]; n_samples = ns_tab[b]; }
    if (j >= n_samples) return;
    const uint8_t* bb = bits + b * pitch_f;
    if (ratio >= 16.0) {
        // Frames are longer than the 9-sample neighbourhood: j - 4 .. j + 4 can only lie in the frames i0 - 1 .. i0 + 1 of
        // sample j, whose [lo, hi) are computed ONCE (same un-fused double arithmetic as premask, so the values are the
        // same bit for bit); the per-neighbour double division + three interval evaluations made this kernel ALU bound
        // (37 us for 64 clips against ~5 us of HBM time).
        const int64_t i0 = (int64_t)((double)j / ratio);
        int64_t lo[3], hi[3];
        int val[3];
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            const int64_t i = i0 - 1 + q;
            const bool ok = i >= 0 && i < n_frames;
            lo[q] = ok ? (int64_t)__dmul_rn((double)i, ratio) : 0;
            hi[q] = ok ? (int64_t)__dadd_rn(__dmul_rn((double)(i + 1), ratio), -1.0) : 0;      // empty interval when !ok
            val[q] = ok && bb[i] == 0 ? 1 : 0;
        }
        auto pm = [&](const int64_t jj) {
#pragma unroll
            for (int q = 0; q < 3; ++q)
                if (jj >= lo[q] && jj < hi[q]) return val[q];
            return 0;
        };
        const int v = pm(j);
        int len = 1;
        for (int d = 1; d <= 4 && j - d >= 0; ++d) {
            if (pm(j - d) != v) break;
            ++len;
        }
        for (int d = 1; d <= 4 && j + d < n_samples && len < 5; ++d) {
            if (pm(j + d) != v) break;
            ++len;
        }
        const float m = (float)(len < 5 ? 1 - v : v);
        mask[b * pitch_s + j] = m;
        if (masked) masked[b * pitch_s + j] = sig[b * pitch_s + j] * m;
        return;
    }
    const int v = premask(bb, n_frames, ratio, j);
    // length of the ORIGINAL run containing j (capped): the reference flips every run shorter
    // than 5 samples in one pass over the original runs (groupby never sees its own writes).
    int len = 1;
    for (int d = 1; d <= 4 && j - d >= 0; ++d) {
        if (premask(bb, n_frames, ratio, j - d) != v) break;
        ++len;
    }
    for (int d = 1; d <= 4 && j + d < n_samples && len < 5; ++d) {
        if (premask(bb, n_frames, ratio, j + d) != v) break;
        ++len;
    }
    const float m = (float)(len < 5 ? 1 - v : v);
    mask[b * pitch_s + j] = m;
    if (masked) masked[b * pitch_s + j] = sig[b * pitch_s + j] * m;
}

extern "C" int sos_bits_to_mask(const uint8_t* bits, int64_t batch, int64_t n_frames, double ratio,
                                int64_t n_samples, float* mask, const float* sig, float* masked,
                                const int32_t* clip_frames, const int32_t* clip_samples, sos_stream_t stream) {
    if (!bits || !mask || batch < 1 || batch > 65535 || n_frames < 1 || n_samples < 1 || !(ratio > 1.0) ||
        (masked && !sig) || ((clip_frames == nullptr) != (clip_samples == nullptr))) {
        sos_set_error("sos_bits_to_mask: bad args");
        return SOS_EINVAL;
    }
    dim3 grid((unsigned)((n_samples + 255) / 256), (unsigned)batch);
    hipLaunchKernelGGL(bits_to_mask_kernel, grid, dim3(256), 0, (hipStream_t)stream, bits, n_frames, ratio,
                       n_samples, mask, sig, masked, clip_frames, clip_samples);
    return sos_check_launch("sos_bits_to_mask");
}

__global__ void threshold_kernel(const float* __restrict__ logits, int64_t n, float thr, uint8_t* __restrict__ bits,
                                 float* __restrict__ conf) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float s = 1.0f / (1.0f + expf(-logits[i]));
    bits[i] = s >= thr ? 1 : 0;
    if (conf) conf[i] = s;
}

extern "C" int sos_threshold_bits(const float* logits, int64_t n, float threshold, uint8_t* bits, float* conf,
                                  sos_stream_t stream) {
    if (!logits || !bits || n < 1) { sos_set_error("sos_threshold_bits: bad args"); return SOS_EINVAL; }
    hipLaunchKernelGGL(threshold_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, logits, n,
                       threshold, bits, conf);
    return sos_check_launch("sos_threshold_bits");
}

// ---- two-pass detector of the 'mixed' pipeline (round 5): which clips have a frame whose 16-bit logit lies within `band_rel`
// x max(1, max_t |logit|) of the threshold (logit 0 <=> sigmoid 0.5)?  One workgroup per clip; fixed-order LDS tree for the
// maximum.  Marked clips keep their rows of the geometry tables (tabs_out = tabs_in), unmarked clips get width 0 in every table:
// the second (parity-precision) detector pass then runs through the ragged per-clip geometry of the kernels and every tile of an
// unmarked clip exits at once -- the selection never reaches the host.  count[0] += marked clips, count[1] += clips.
__global__ void logit_band_mark_kernel(const float* __restrict__ logits, const float* __restrict__ scale, int64_t n,
                                       const int32_t* __restrict__ n_valid,
                                       float band_rel, const int32_t* __restrict__ tabs_in, int32_t* __restrict__ tabs_out,
                                       int ntab, int64_t B, int32_t* __restrict__ mark, int32_t* __restrict__ count) {
    __shared__ float red[64];
    const int64_t b = blockIdx.x;
    const int nv = n_valid ? min((int)n, n_valid[b]) : (int)n;
    const float* lp = logits + b * n;
    float m = 0.f;
    for (int t = threadIdx.x; t < nv; t += 64) m = fmaxf(m, fabsf(lp[t]));
    // scale (ABI 10, optional): the magnitude the 16-bit pass's logit error is relative to -- sum_i |w_i| |a_i| + |bias| of the
    // last layer's dot product, >= |logit| -- so that a clip whose logits hover around 0 by CANCELLATION still gets a band as
    // wide as its terms are large (NaN counts as a hit below: fmaxf drops it here)
    if (scale) { const float* sp = scale + b * n; for (int t = threadIdx.x; t < nv; t += 64) m = fmaxf(m, fabsf(sp[t])); }
    red[threadIdx.x] = m;
    __syncthreads();
    for (int s = 32; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) red[threadIdx.x] = fmaxf(red[threadIdx.x], red[threadIdx.x + s]);
        __syncthreads();
    }
    const float band = band_rel * fmaxf(1.f, red[0]);
    __syncthreads();
    float hit = 0.f;
    for (int t = threadIdx.x; t < nv; t += 64) {
        const float v = fabsf(lp[t]);
        if (!(v >= band)) hit = 1.f;              // (NaN counts as a hit: the parity pass decides)
    }
    red[threadIdx.x] = hit;
    __syncthreads();
    for (int s = 32; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) red[threadIdx.x] = fmaxf(red[threadIdx.x], red[threadIdx.x + s]);
        __syncthreads();
    }
    const int mk = red[0] != 0.f ? 1 : 0;
    if (threadIdx.x == 0) {
        mark[b] = mk;
        if (count) { atomicAdd(count, mk); atomicAdd(count + 1, 1); }
    }
    for (int k = threadIdx.x; k < ntab; k += 64) tabs_out[(int64_t)k * B + b] = mk ? tabs_in[(int64_t)k * B + b] : 0;
}

extern "C" int sos_logit_band_mark(const float* logits, const float* scale, int64_t batch, int64_t n, const int32_t* n_valid,
                                   float band_rel, const int32_t* tabs_in, int32_t* tabs_out, int ntab, int32_t* mark,
                                   int32_t* count, sos_stream_t stream) {
    if (!logits || !mark || batch < 1 || batch > 0x7fffffff || n < 1 || !(band_rel >= 0.f) || ntab < 0 ||
        (ntab > 0 && (!tabs_in || !tabs_out))) {
        sos_set_error("sos_logit_band_mark: bad args");
        return SOS_EINVAL;
    }
    hipLaunchKernelGGL(logit_band_mark_kernel, dim3((unsigned)batch), dim3(64), 0, (hipStream_t)stream, logits, scale, n, n_valid,
                       band_rel, tabs_in, tabs_out, ntab, batch, mark, count);
    return sos_check_launch("sos_logit_band_mark");
}

// ----------------------------------------------------------------- NCHW f32 -> NHWC bf16 pack
__global__ void pack_kernel(const float* __restrict__ in, int C, int64_t HW, int64_t total, bf16_t* __restrict__ out,
                            int cs, int x3, const float* __restrict__ mul_p) {
    const int third = x3 ? cs / 3 : cs;
    const float mul = mul_p ? mul_p[0] : 1.f;
    if (!x3 && (cs & 7) == 0) {          // common case: a pixel's channel run as whole 16-byte stores
        for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
            const int64_t b = i / HW, r = i - b * HW;
            const float* ip = in + b * C * HW + r;
            for (int c0 = 0; c0 < cs; c0 += 8) {
                unsigned w4[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int c = c0 + 2 * e;
                    const float v0 = c < C ? ip[(int64_t)c * HW] : 0.f, v1 = c + 1 < C ? ip[(int64_t)(c + 1) * HW] : 0.f;
                    w4[e] = pack2bf(v0 * mul, v1 * mul);
                }
                *(uint4*)(out + i * cs + c0) = make_uint4(w4[0], w4[1], w4[2], w4[3]);
            }
        }
        return;
    }
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t b = i / HW, r = i - b * HW;
        bf16_t* o = out + i * cs;
        for (int c = 0; c < third; ++c) {
            float v = c < C ? in[(b * C + c) * HW + r] * mul : 0.f;
            const bf16_t hi = f2bf(v);
            o[c] = hi;
            if (x3) {
                o[c + third] = hi;
                o[c + 2 * third] = f2bf(v - bf2f(hi));
            }
        }
    }
}

extern "C" int sos_pack_nchw_to_nhwc(const float* in, int64_t B, int C, int64_t H, int64_t W, void* out, int cs,
                                     int dtype, const float* mul, sos_stream_t stream) {
    const int x3 = dtype == SOS_DT_BF16X3;
    if (!in || !out || B < 1 || C < 1 || (dtype != SOS_DT_BF16 && !x3) || (x3 && cs % 3) || (x3 ? cs / 3 : cs) < C) {
        sos_set_error("sos_pack_nchw_to_nhwc: bad args");
        return SOS_EINVAL;
    }
    const int64_t total = B * H * W;
    hipLaunchKernelGGL(pack_kernel, dim3(ew_grid(total)), dim3(256), 0, (hipStream_t)stream, in, C, H * W, total,
                       (bf16_t*)out, cs, x3, mul);
    return sos_check_launch("sos_pack_nchw_to_nhwc");
}

// ------------------------------------------- NCHW f32 -> NHWC with the horizontal taps of the first conv layer on the channel axis
// The first block of every encoder / of the U-Net reads 2 channels (M1/networks.py:120-128, M2/networks.py:72-80,158,165):
// on the MFMA kernels a tap contracts 16 stored channels of which 2 are real.  Folding the kw horizontal taps into the
// channel axis HERE (channel t*C + c of pixel (h, w) = in[c][h][w + t - pad_left], zero or reflected outside the clip) costs
// no extra byte -- a pixel is 16 stored channels either way -- and turns the kh x kw layer into a kh x 1 layer over kw*C
// real channels (1x7: ONE tap with 14 of 16 channels real; 5x5: 5 taps with 10 of 16), for the forward conv and for its
// weight gradient alike.
// CT / KWT > 0: the channel and tap counts as compile-time constants (2 x 7 and 2 x 5, the module inputs of the encoders and of the
// U-Net): the tap / channel of a stored channel is then a constant of the unrolled loop instead of two divisions per element (the
// generic form ran at 1.4 TB/s, VALU-bound: ~500 instructions per pixel)
template <int CT, int KWT>
__global__ void pack_wtaps_kernel(const float* __restrict__ in, int C_, int H, int W, int64_t total, bf16_t* __restrict__ out,
                                  int cs, int x3, int kw_, int pad_l, int reflect, const int* __restrict__ clip_w,
                                  const float* __restrict__ mul_p) {
    const int C = CT > 0 ? CT : C_, kw = KWT > 0 ? KWT : kw_;
    const int third = x3 ? cs / 3 : cs;                  // multiple of 8, >= kw * C
    const float mul = mul_p ? mul_p[0] : 1.f;
    const int64_t HW = (int64_t)H * W;
    const int kc = kw * C;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        int64_t b;
        int h, w;
        if (total < 0x7fffffffll) {                      // (32-bit divisions: the 64-bit ones were a quarter of the kernel)
            const unsigned iu = (unsigned)i, bu = iu / (unsigned)HW, ru = iu - bu * (unsigned)HW;
            b = bu; h = (int)(ru / (unsigned)W); w = (int)(ru - (unsigned)h * (unsigned)W);
        } else {
            b = i / HW;
            const int64_t r = i - b * HW;
            h = (int)(r / W); w = (int)(r - (int64_t)h * W);
        }
        const int Wc = clip_w ? clip_w[b] : W;
        const float* ip = in + b * C * HW + (int64_t)h * W;
        bf16_t* o = out + i * cs;
        for (int c0 = 0; c0 < third; c0 += 8) {
            unsigned hw4[4], lw4[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float v[2];
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const int k = c0 + 2 * e + u;
                    float x = 0.f;
                    if (k < kc && w < Wc) {
                        const int t = k / C, c = k - t * C;
                        int ws = w + t - pad_l;
                        if (reflect) ws = reflect_index(ws, Wc);
                        if (ws >= 0 && ws < Wc) x = ip[(int64_t)c * HW + ws] * mul;
                    }
                    v[u] = x;
                }
                hw4[e] = pack2bf(v[0], v[1]);
                lw4[e] = pack2bf(v[0] - sos_lo2f(hw4[e]), v[1] - sos_hi2f(hw4[e]));
            }
            const uint4 hv = make_uint4(hw4[0], hw4[1], hw4[2], hw4[3]);
            *(uint4*)(o + c0) = hv;
            if (x3) {
                *(uint4*)(o + c0 + third) = hv;
                *(uint4*)(o + c0 + 2 * third) = make_uint4(lw4[0], lw4[1], lw4[2], lw4[3]);
            }
        }
    }
}

extern "C" int sos_pack_nchw_wtaps(const float* in, int64_t B, int C, int64_t H, int64_t W, int kw, int pad_left, int pad_mode,
                                   const int32_t* clip_w, void* out, int cs, int dtype, const float* mul, sos_stream_t stream) {
    const int x3 = dtype == SOS_DT_BF16X3;
    const int third = x3 ? cs / 3 : cs;
    if (!in || !out || B < 1 || C < 1 || H < 1 || W < 1 || kw < 1 || pad_left < 0 || pad_left >= kw || (dtype != SOS_DT_BF16 && !x3) ||
        (x3 && cs % 3) || third % 8 || third < kw * C || H * W >= 0x7fffffffll ||
        (pad_mode != SOS_PAD_ZERO && pad_mode != SOS_PAD_REFLECT) || (pad_mode == SOS_PAD_REFLECT && pad_left >= W)) {
        sos_set_error("sos_pack_nchw_wtaps: bad args (C=%d kw=%d pad=%d cs=%d)", C, kw, pad_left, cs);
        return SOS_EINVAL;
    }
    const int64_t total = B * H * W;
#define SOS_PACK_WTAPS(CT, KWT)                                                                                                  \
    hipLaunchKernelGGL((pack_wtaps_kernel<CT, KWT>), dim3(ew_grid(total)), dim3(256), 0, (hipStream_t)stream, in, C, (int)H, (int)W, \
                       total, (bf16_t*)out, cs, x3, kw, pad_left, pad_mode == SOS_PAD_REFLECT ? 1 : 0, clip_w, mul)
    if (C == 2 && kw == 7) SOS_PACK_WTAPS(2, 7);
    else if (C == 2 && kw == 5) SOS_PACK_WTAPS(2, 5);
    else SOS_PACK_WTAPS(0, 0);
#undef SOS_PACK_WTAPS
    return sos_check_launch("sos_pack_nchw_wtaps");
}
