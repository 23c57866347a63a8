// stft_mfma.hip -- STFT / ISTFT as windowed-DFT GEMMs on the matrix cores (gfx950).
//
// Reference: fast_stft M1/transform.py:188-193 (librosa.stft(data, 510, 158, 400): periodic hann of 400 centred in
// n_fft = 510, center=True, reflect padding) fused with real_imag_expand (:10-17) and the caller's transpose to
// [2][F][T] (M2/dataset.py:255); fast_istft :196-202 (librosa.istft(S, 158, 400): inverse real DFT, synthesis
// window, overlap-add, division by the window-sum-square, centre trim).
//
// Both transforms are dense matrix products with a constant matrix, so they run on MFMA instead of the vector ALU
// (the round-1 direct DFT was VALU-bound at 1-4 % of the HBM roofline):
//   STFT : C[j][t] = sum_n  D[j][n] * x[t*hop + n + off]        j = (re|im, bin) 2*nbins rows, K = win samples
//   ISTFT: Y[n][t] = sum_k  E[n][k] * S[k][t]                   k = (re|im, bin), n = sample inside the frame
// with the analysis / synthesis window (and the irfft weights 1,2,..,2,1 / n_fft) folded into D / E.  The parity bar
// is 1e-5 against an f64 oracle, so the products are NOT taken in plain half precision: every operand is split into
// hi + lo halves (22 significand bits together) and contracted as hi*hi + hi*lo + lo*hi on
// v_mfma_f32_32x32x16_f16 with fp32 accumulation (3 passes at the 2.5 PF rate instead of one at the 157 TF fp32-matrix
// rate; operands pre-scaled by powers of two so that the lo halves stay out of the subnormal range).
//
// Memory behaviour: the STFT's column operand is the clip itself -- frame t is the sample window starting at t*hop,
// so a workgroup stages ONE contiguous span of samples in LDS (coalesced, reflect-indexed at the clip's own ends) and
// every fragment is a 16-byte slice of it (row pitch = hop samples = 79 dwords: conflict free); D / E are packed once
// on the host in MFMA fragment order (one coalesced 1 KB load per fragment, L2 resident, < 1 MB).  Accumulator
// columns are frames, so the STFT's stores are 128-byte runs along t of the planar [B][2][F][T] output; the ISTFT
// stages its column operand through LDS (coalesced reads along t, transposed 16-byte writes), writes the frame
// signals to LDS and overlap-adds them in a fixed order (deterministic, no atomics) with the window-sum-square
// computed from the same <= 3 frames.
#include "sos_common.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <type_traits>

typedef _Float16 fe_h8 __attribute__((ext_vector_type(8)));
typedef float fe_f32x16 __attribute__((ext_vector_type(16)));

#define FE_COLS 64              // frames (accumulator columns) per workgroup
#define FE_SX 16.0f             // STFT: sample scale          (exact powers of two, undone in the epilogue)
#define FE_SD 16.0f             // STFT: DFT-matrix scale
#define FE_IS 0.25f             // ISTFT: spectrogram scale
#define FE_IE 4096.0f           // ISTFT: synthesis-matrix scale
#define FE_IADV 61              // ISTFT: frames a workgroup advances (FE_COLS minus the ceil(win/hop) = 3 halo frames)
#define FE_IKC 64               // ISTFT: K (re|im, bin) values staged per chunk
#define FE_RING 4               // matrix-fragment buffers in flight per wave (= k-steps of an ISTFT chunk)
#define FE_SWAVES 4              // STFT: waves per workgroup
#define FE_SRS 4                 // STFT: workgroups that share a frame tile, each computing 16 / FE_SRS of the row tiles
#define FE_SLD 42               // STFT: sample loads in flight per thread while staging
#define FE_IWAVES 16            // ISTFT: waves per workgroup (two per SIMD: the kernel is latency bound, one workgroup per CU)

__device__ __forceinline__ fe_h8 fe_frag16(const uint4 v) { return __builtin_bit_cast(fe_h8, v); }

// ------------------------------------------------------------------------------------------------ STFT
struct StftParams {
    const float* wave;
    long long wave_stride, n_samples, T;
    const uint4* dhi;           // [ksteps][rtiles][64 lanes] 16-byte A fragments (8 consecutive k of row lane&31)
    const uint4* dlo;
    int n_fft, hop, win, nbins, ksteps, rtiles, span;
    float* out;
    const int* n_tab;
};

// One workgroup = FE_COLS consecutive frames of one clip x all 2*nbins output rows; wave w owns the row tiles
// {w, w+4, ...} (4 of the 16 for n_fft = 510) and both 32-frame column tiles.
__global__ __launch_bounds__(FE_SWAVES * 64) void stft_mfma_kernel(StftParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    _Float16* shi = (_Float16*)smem;
    _Float16* slo = shi + p.span;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const long long b = blockIdx.y, t0 = (long long)blockIdx.x * FE_COLS;
    long long ns = p.n_samples, Tc = p.T;
    if (p.n_tab) { ns = p.n_tab[b]; Tc = 1 + ns / p.hop; }     // ragged batch: the clip's own length / frame count
    if (t0 >= Tc) return;
    // sample m of the span is clip index i0 + m (frame t, window sample n -> t*hop + n + lpad - n_fft/2), reflected at
    // the clip's own ends; frames of this tile past Tc are computed on clamped indices and never stored
    const long long i0 = t0 * p.hop + (p.n_fft - p.win) / 2 - p.n_fft / 2;
    const float* wv = p.wave + b * p.wave_stride;
    // FE_SLD independent loads in flight per thread: the whole span of the reference geometry (10 360 samples, 41 per
    // thread of a 4-wave workgroup) is one round trip to HBM (a plain loop issues them one round trip at a time: 40 us of 55; batches of 8
    // were three round trips, ~5 us of 28)
    constexpr int NTHR = FE_SWAVES * 64;
    for (int m0 = tid; m0 < p.span; m0 += FE_SLD * NTHR) {
        float v[FE_SLD];
#pragma unroll
        for (int u = 0; u < FE_SLD; ++u) {
            long long i = i0 + m0 + u * NTHR;
            if (i < 0) i = -i;
            if (i >= ns) i = 2 * (ns - 1) - i;
            i = i < 0 ? 0 : (i >= ns ? ns - 1 : i);
            v[u] = wv[i];
        }
#pragma unroll
        for (int u = 0; u < FE_SLD; ++u) {
            const int m = m0 + u * NTHR;
            if (m < p.span) {
                const float x = v[u] * FE_SX;
                const _Float16 h = (_Float16)x;
                shi[m] = h;
                slo[m] = (_Float16)(x - (float)h);
            }
        }
    }
    __syncthreads();

    constexpr int RT = 16 / (FE_SWAVES * FE_SRS);    // row tiles per wave
    static_assert(RT >= 1, "16 row-tile slots over FE_SRS workgroups of FE_SWAVES waves");
    fe_f32x16 acc[RT][2];
#pragma unroll
    for (int r = 0; r < RT; ++r)
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[r][c][e] = 0.f;
    const int l31 = lane & 31, g = lane >> 5;
    // column operand: 8 consecutive window samples k0 + 8g .. of frame (c*32 + l31): a 4-byte aligned 16-byte slice
    const int boff = (l31 * p.hop + 8 * g) * 2;
    const int cstride = 32 * p.hop * 2;
    int rt[RT];
    bool rok[RT];
#pragma unroll
    for (int r = 0; r < RT; ++r) { rt[r] = (int)blockIdx.z * (FE_SWAVES * RT) + wave + FE_SWAVES * r; rok[r] = rt[r] < p.rtiles; if (!rok[r]) rt[r] = p.rtiles - 1; }
    // matrix fragments: a ring of FE_RING buffers, loaded FE_RING-1 k-steps ahead (an L2 round trip is ~3 k-steps of
    // this wave's MFMAs; with one k-step of lead every k-step stalled on it: 56 us instead of ~15 for B = 64)
    uint4 ahi[FE_RING][RT], alo[FE_RING][RT];
    auto load_a = [&](auto buf_tag, const int ks) {
        constexpr int buf = decltype(buf_tag)::value;
        if (ks >= p.ksteps) return;
#pragma unroll
        for (int r = 0; r < RT; ++r) {
            const size_t idx = ((size_t)ks * p.rtiles + rt[r]) * 64 + lane;
            ahi[buf][r] = p.dhi[idx];
            alo[buf][r] = p.dlo[idx];
        }
    };
    auto read_b = [&](const _Float16* plane, const int c, const int ks) {
        const char* q = (const char*)plane + boff + c * cstride + ks * 32;
        const unsigned w0 = *(const unsigned*)q, w1 = *(const unsigned*)(q + 4), w2 = *(const unsigned*)(q + 8),
                       w3 = *(const unsigned*)(q + 12);
        return fe_frag16(make_uint4(w0, w1, w2, w3));
    };
    // the two fragment buffers alternate with COMPILE-TIME indices (a run-time index would demote them to scratch)
    auto kstep = [&](auto cur_tag, const int ks) {
        constexpr int cur = decltype(cur_tag)::value;
        load_a(std::integral_constant<int, (cur + FE_RING - 1) % FE_RING>{}, ks + FE_RING - 1);
        fe_h8 bh[2], bl[2];
#pragma unroll
        for (int c = 0; c < 2; ++c) { bh[c] = read_b(shi, c, ks); bl[c] = read_b(slo, c, ks); }
        // the three partial products of a tile go to the SAME accumulator: issue them a full round of the 8 tiles apart
        // (back-to-back dependent MFMAs wait for each other's write-back)
#pragma unroll
        for (int pass = 0; pass < 3; ++pass)
#pragma unroll
            for (int r = 0; r < RT; ++r)
#pragma unroll
                for (int c = 0; c < 2; ++c)
                    acc[r][c] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fe_frag16(pass == 2 ? alo[cur][r] : ahi[cur][r]),
                                                                       pass == 1 ? bl[c] : bh[c], acc[r][c], 0, 0, 0);
    };
    load_a(std::integral_constant<int, 0>{}, 0);
    load_a(std::integral_constant<int, 1>{}, 1);
    load_a(std::integral_constant<int, 2>{}, 2);
    int ks = 0;
    for (; ks + 3 < p.ksteps; ks += 4) {
        kstep(std::integral_constant<int, 0>{}, ks);
        kstep(std::integral_constant<int, 1>{}, ks + 1);
        kstep(std::integral_constant<int, 2>{}, ks + 2);
        kstep(std::integral_constant<int, 3>{}, ks + 3);
    }
    if (ks < p.ksteps) kstep(std::integral_constant<int, 0>{}, ks);
    if (ks + 1 < p.ksteps) kstep(std::integral_constant<int, 1>{}, ks + 1);
    if (ks + 2 < p.ksteps) kstep(std::integral_constant<int, 2>{}, ks + 2);
    // accumulator (row = (reg&3) + 8*(reg>>2) + 4*(lane>>5), column = lane&31): for a fixed register the 32 lanes of a
    // half wave hold 32 consecutive frames of one (re|im, bin) row -> 128-byte runs of the planar output
    const float scale = 1.0f / (FE_SX * FE_SD);
#pragma unroll
    for (int r = 0; r < RT; ++r) {
        if (!rok[r]) continue;
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            const long long t = t0 + c * 32 + l31;
            if (t >= Tc) continue;
            // row j = (re|im plane cc, bin f) lives at plane-major offset (cc*nbins + f) = j: no division needed
            float* o = p.out + (b * 2 * p.nbins + rt[r] * 32 + 4 * g) * p.T + t;
#pragma unroll
            for (int e = 0; e < 16; ++e) o[(size_t)((e & 3) + 8 * (e >> 2)) * p.T] = acc[r][c][e] * scale;
        }
    }
}

// ----------------------------------------------------------------------------------------------- ISTFT
struct IstftParams {
    const float* spec;
    long long T;                // frames per clip (row pitch of the spectrogram)
    const uint4* ehi;           // [ksteps][rtiles][64] A fragments of the synthesis matrix E[n][k]
    const uint4* elo;
    const float* win2;          // [win] squared synthesis window (librosa.filters.window_sumsquare terms)
    int n_fft, hop, win, nbins, ksteps, rtiles;
    float* out;
    long long out_stride;
    const int* t_tab;
};

// One workgroup = FE_IADV*hop consecutive output samples of one clip: the (<= FE_COLS) frames that overlap them are
// synthesised by the GEMM (rows = sample inside the frame, columns = frames), written to LDS and overlap-added.
__global__ __launch_bounds__(FE_IWAVES * 64) void istft_mfma_kernel(IstftParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int RT = 16 / FE_IWAVES, NTHR = FE_IWAVES * 64;          // row tiles per wave
    constexpr int SW = FE_IWAVES < 8 ? FE_IWAVES : 8, SH = 8 / SW;     // waves that stage the chunk; K groups per staging thread
    static_assert(FE_IWAVES == 4 || FE_IWAVES == 8 || FE_IWAVES == 16, "16 row-tile slots and 8 K groups per chunk are split over the waves");
    constexpr int BPITCH = FE_IKC * 2 + 16;                    // bytes per frame row of a staged K chunk (padded: banks)
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const long long b = blockIdx.y;
    long long Tc = p.T;
    if (p.t_tab) Tc = p.t_tab[b];
    const long long n_out = (long long)p.hop * (Tc - 1);
    const long long j0 = (long long)blockIdx.x * FE_IADV * p.hop;
    if (j0 >= n_out) return;
    const int lpad = (p.n_fft - p.win) / 2, half = p.n_fft / 2;
    const long long p0 = j0 + half;                            // padded-signal coordinate of the first sample
    long long tlo = p0 - lpad - p.win + 1;                     // frames t with t*hop + lpad <= p < t*hop + lpad + win
    tlo = tlo <= 0 ? 0 : (tlo + p.hop - 1) / p.hop;
    long long thi = (p0 + (long long)FE_IADV * p.hop - 1 - lpad) / p.hop;
    if (thi > Tc - 1) thi = Tc - 1;
    const int nfr = (int)(thi - tlo + 1);                      // <= FE_COLS by construction

    char* bhi = smem;                                          // [FE_COLS][BPITCH] hi halves of the chunk, then lo
    char* blo = smem + FE_COLS * BPITCH;
    fe_f32x16 acc[RT][2];
#pragma unroll
    for (int r = 0; r < RT; ++r)
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[r][c][e] = 0.f;
    const int l31 = lane & 31, g = lane >> 5;
    int rt[RT];
    bool rok[RT];
#pragma unroll
    for (int r = 0; r < RT; ++r) { rt[r] = wave + FE_IWAVES * r; rok[r] = rt[r] < p.rtiles; if (!rok[r]) rt[r] = p.rtiles - 1; }

    // staging: thread = (frame column tid & 63, 8-value K group tid >> 6 (and + 4 with 4 waves)): coalesced reads along t
    const int scol = tid & 63, sgrp = tid >> 6;
    const bool col_ok = scol < nfr && sgrp < SW;
    const float* sp = p.spec + (size_t)b * 2 * p.nbins * p.T + tlo + scol;
    float stage[SH][8];
    auto load_chunk = [&](const int ch) {
#pragma unroll
        for (int h = 0; h < SH; ++h) {
            const int k0 = ch * FE_IKC + ((sgrp & (SW - 1)) + SW * h) * 8;                       // row (re|im, bin) of the spectrogram
#pragma unroll
            for (int e = 0; e < 8; ++e) stage[h][e] = col_ok ? sp[(size_t)(k0 + e) * p.T] * FE_IS : 0.f;
        }
    };
    auto store_chunk = [&]() {
        if (sgrp >= SW) return;
#pragma unroll
        for (int h = 0; h < SH; ++h) {
            fe_h8 hv, lv;
#pragma unroll
            for (int e = 0; e < 8; ++e) { hv[e] = (_Float16)stage[h][e]; lv[e] = (_Float16)(stage[h][e] - (float)hv[e]); }
            const int off = scol * BPITCH + (sgrp + SW * h) * 16;
            *(uint4*)(bhi + off) = __builtin_bit_cast(uint4, hv);
            *(uint4*)(blo + off) = __builtin_bit_cast(uint4, lv);
        }
    };
    const int nchunks = 2 * p.nbins / FE_IKC;
    static_assert(FE_IKC / 16 == FE_RING, "one chunk = FE_RING k-steps: the ring index is the k-step inside the chunk");
    // synthesis-matrix fragments: ring of FE_RING buffers, FE_RING-1 k-steps ahead (see the STFT)
    uint4 ehi[FE_RING][RT], elo[FE_RING][RT];
    auto load_e = [&](auto buf_tag, const int ks) {
        constexpr int buf = decltype(buf_tag)::value;
        if (ks >= p.ksteps) return;
#pragma unroll
        for (int r = 0; r < RT; ++r) {
            const size_t idx = ((size_t)ks * p.rtiles + rt[r]) * 64 + lane;
            ehi[buf][r] = p.ehi[idx];
            elo[buf][r] = p.elo[idx];
        }
    };
    auto kstep = [&](auto kk_tag, const int ch) {
        constexpr int kk = decltype(kk_tag)::value;
        const int ks = ch * FE_RING + kk;
        load_e(std::integral_constant<int, (kk + FE_RING - 1) % FE_RING>{}, ks + FE_RING - 1);
        // the next chunk's spectrogram values go out AFTER this k-step's fragment prefetch: vector-memory waits complete in
        // order, so issued the other way round the fragments needed three k-steps from now would wait for all 16 of them
        if (kk == 0 && ch + 1 < nchunks) load_chunk(ch + 1);
        fe_h8 bh[2], bl[2];
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            const int off = (c * 32 + l31) * BPITCH + kk * 32 + g * 16;
            bh[c] = fe_frag16(*(const uint4*)(bhi + off));
            bl[c] = fe_frag16(*(const uint4*)(blo + off));
        }
#pragma unroll
        for (int pass = 0; pass < 3; ++pass)                   // same accumulator a full round of tiles apart (see the STFT)
#pragma unroll
            for (int r = 0; r < RT; ++r) {
                if (!rok[r]) continue;                         // wave-uniform
#pragma unroll
                for (int c = 0; c < 2; ++c)
                    acc[r][c] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fe_frag16(pass == 2 ? elo[kk][r] : ehi[kk][r]),
                                                                       pass == 1 ? bl[c] : bh[c], acc[r][c], 0, 0, 0);
            }
    };
    load_chunk(0);
    load_e(std::integral_constant<int, 0>{}, 0);
    load_e(std::integral_constant<int, 1>{}, 1);
    load_e(std::integral_constant<int, 2>{}, 2);
    for (int ch = 0; ch < nchunks; ++ch) {
        __syncthreads();                                       // the previous chunk's fragments have been read
        store_chunk();
        __syncthreads();
        kstep(std::integral_constant<int, 0>{}, ch);           // (also issues the next chunk's loads: in flight during the MFMAs)
        kstep(std::integral_constant<int, 1>{}, ch);
        kstep(std::integral_constant<int, 2>{}, ch);
        kstep(std::integral_constant<int, 3>{}, ch);
    }
    __syncthreads();
    // frame signals to LDS: yf[frame][n], pitch odd (in dwords) so that the 32 frames of a store hit 32 banks
    const int ypitch = p.rtiles * 32 + 1;
    float* yf = (float*)smem;
    const float scale = 1.0f / (FE_IS * FE_IE);
#pragma unroll
    for (int r = 0; r < RT; ++r) {
        if (!rok[r]) continue;
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int n = rt[r] * 32 + (e & 3) + 8 * (e >> 2) + 4 * g;
                yf[(c * 32 + l31) * ypitch + n] = acc[r][c][e] * scale;
            }
    }
    __syncthreads();
    // overlap-add in a fixed order (newest frame first, like the round-1 kernel) + window-sum-square normalisation.
    // A sample is covered by at most 3 frames (host check): the three terms are independent, predicated loads from LDS
    // (the squared window sits behind the frame signals), and 4 samples are in flight per thread -- the first version's
    // data-dependent loop with a global window read per term serialised ~100 round trips per thread (15 of 75 us).
    float* w2 = yf + FE_COLS * ypitch;
    for (int n = tid; n < p.win; n += NTHR) w2[n] = p.win2[n];
    __syncthreads();
    const long long jend = j0 + (long long)FE_IADV * p.hop < n_out ? j0 + (long long)FE_IADV * p.hop : n_out;
    const int ihop = p.hop;
    for (long long jb = j0 + tid; jb < jend; jb += 4 * NTHR) {
        float y[4], wss[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const long long j = jb + u * NTHR;
            const long long pp = j + half;
            long long tmax = (pp - lpad) / ihop;
            if (tmax > Tc - 1) tmax = Tc - 1;
            const int n0 = (int)(pp - tmax * ihop) - lpad;               // sample index inside the newest covering frame
            const int f0 = (int)(tmax - tlo);
            y[u] = 0.f; wss[u] = 0.f;
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                const int n = n0 + q * ihop, f = f0 - q;
                const bool ok = j < jend && f >= 0 && n < p.win;
                const int nn = ok ? n : 0, ff = ok ? f : 0;
                const float v = yf[ff * ypitch + nn], w = w2[nn];
                y[u] += ok ? v : 0.f;
                wss[u] += ok ? w : 0.f;
            }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const long long j = jb + u * NTHR;
            if (j < jend) p.out[b * p.out_stride + j] = wss[u] > 1.17549435e-38f ? y[u] / wss[u] : y[u];
        }
    }
}

// ------------------------------------------------------------------------------------------- host side
static inline uint16_t fe_f2h(float f) { const _Float16 h = (_Float16)f; uint16_t u; memcpy(&u, &h, 2); return u; }
static inline float fe_h2f(uint16_t u) { _Float16 h; memcpy(&h, &u, 2); return (float)h; }

static int fe_geometry_ok(int n_fft, int hop, int win, const char* who) {
    if (n_fft < 2 || (n_fft & 1) || win < 16 || win > n_fft || (win & 15) || hop < 2 || (hop & 1) || ((n_fft + 2) & 31) ||
        (win + hop - 1) / hop > FE_COLS - FE_IADV) {
        sos_set_error("%s: unsupported geometry n_fft=%d hop=%d win=%d (need n_fft+2 %% 32 == 0, win %% 16 == 0, even hop, "
                      "ceil(win/hop) <= %d; the reference uses 510/158/400)", who, n_fft, hop, win, FE_COLS - FE_IADV);
        return SOS_EINVAL;
    }
    return SOS_OK;
}

// hann(win, periodic) -- scipy.signal.get_window('hann', win, fftbins=True), what librosa.stft / istft use
static inline double fe_hann(int n, int win) { return 0.5 - 0.5 * cos(2.0 * M_PI * (double)n / (double)win); }

// Packed matrices (HOST buffers; upload them once per device).  Fragment order of the MFMA row operand:
//   element [(ks*rtiles + rt)*64 + lane][e] = M[rt*32 + (lane&31)][ks*16 + 8*(lane>>5) + e], hi and lo halves.
extern "C" int64_t sos_stft_matrix_bytes(int n_fft, int hop, int win_length) {
    if (fe_geometry_ok(n_fft, hop, win_length, "sos_stft_matrix_bytes")) return -1;
    return (int64_t)(win_length / 16) * ((n_fft + 2) / 32) * 64 * 16;
}
extern "C" int sos_stft_pack_matrix(int n_fft, int hop, int win_length, void* hi, void* lo) {
    if (fe_geometry_ok(n_fft, hop, win_length, "sos_stft_pack_matrix")) return SOS_EINVAL;
    if (!hi || !lo) { sos_set_error("sos_stft_pack_matrix: null pointer"); return SOS_EINVAL; }
    const int nbins = n_fft / 2 + 1, rtiles = 2 * nbins / 32, ksteps = win_length / 16, lpad = (n_fft - win_length) / 2;
    uint16_t* H = (uint16_t*)hi;
    uint16_t* Lo = (uint16_t*)lo;
    for (int ks = 0; ks < ksteps; ++ks)
        for (int rt = 0; rt < rtiles; ++rt)
            for (int lane = 0; lane < 64; ++lane)
                for (int e = 0; e < 8; ++e) {
                    const int j = rt * 32 + (lane & 31), n = ks * 16 + 8 * (lane >> 5) + e;
                    const int c = j / nbins, f = j - c * nbins;
                    // X[f] = sum_n w[n] x[n + lpad'] exp(-2 pi i f (n + lpad) / n_fft): re = +cos, im = -sin
                    const long long ph = ((long long)f * (n + lpad)) % n_fft;
                    const double ang = 2.0 * M_PI * (double)ph / (double)n_fft;
                    const double v = fe_hann(n, win_length) * (c == 0 ? cos(ang) : -sin(ang)) * (double)FE_SD;
                    const uint16_t h = fe_f2h((float)v);
                    const size_t o = (((size_t)ks * rtiles + rt) * 64 + lane) * 8 + e;
                    H[o] = h;
                    Lo[o] = fe_f2h((float)(v - (double)fe_h2f(h)));
                }
    return SOS_OK;
}
extern "C" int64_t sos_istft_matrix_bytes(int n_fft, int hop, int win_length) {
    if (fe_geometry_ok(n_fft, hop, win_length, "sos_istft_matrix_bytes")) return -1;
    return (int64_t)((n_fft + 2) / 16) * ((win_length + 31) / 32) * 64 * 16;
}
extern "C" int sos_istft_pack_matrix(int n_fft, int hop, int win_length, void* hi, void* lo, float* win_sq) {
    if (fe_geometry_ok(n_fft, hop, win_length, "sos_istft_pack_matrix")) return SOS_EINVAL;
    if (!hi || !lo || !win_sq) { sos_set_error("sos_istft_pack_matrix: null pointer"); return SOS_EINVAL; }
    const int nbins = n_fft / 2 + 1, rtiles = (win_length + 31) / 32, ksteps = 2 * nbins / 16, lpad = (n_fft - win_length) / 2;
    uint16_t* H = (uint16_t*)hi;
    uint16_t* Lo = (uint16_t*)lo;
    for (int n = 0; n < win_length; ++n) { const float w = (float)fe_hann(n, win_length); win_sq[n] = w * w; }
    for (int ks = 0; ks < ksteps; ++ks)
        for (int rt = 0; rt < rtiles; ++rt)
            for (int lane = 0; lane < 64; ++lane)
                for (int e = 0; e < 8; ++e) {
                    const int n = rt * 32 + (lane & 31), k = ks * 16 + 8 * (lane >> 5) + e;
                    const int c = k / nbins, f = k - c * nbins;
                    double v = 0.0;
                    if (n < win_length) {
                        // irfft: y[m] = (1/N) (Re X0 + (-1)^m Re X_{N/2} + 2 sum_{0<f<N/2} Re X_f cos - Im X_f sin), m = n + lpad
                        const long long ph = ((long long)f * (n + lpad)) % n_fft;
                        const double ang = 2.0 * M_PI * (double)ph / (double)n_fft;
                        const double wgt = (f == 0 || f == nbins - 1) ? 1.0 : 2.0;
                        v = fe_hann(n, win_length) * wgt / (double)n_fft * (c == 0 ? cos(ang) : -sin(ang)) * (double)FE_IE;
                    }
                    const uint16_t h = fe_f2h((float)v);
                    const size_t o = (((size_t)ks * rtiles + rt) * 64 + lane) * 8 + e;
                    H[o] = h;
                    Lo[o] = fe_f2h((float)(v - (double)fe_h2f(h)));
                }
    return SOS_OK;
}

extern "C" int sos_stft_f32(const float* wave, int64_t batch, int64_t n_samples, int64_t wave_stride, const void* mat_hi,
                            const void* mat_lo, int n_fft, int hop, int win_length, float* out, int64_t n_frames,
                            const int32_t* clip_samples, sos_stream_t stream) {
    if (!wave || !mat_hi || !mat_lo || !out) { sos_set_error("sos_stft_f32: null pointer"); return SOS_EINVAL; }
    int rc = fe_geometry_ok(n_fft, hop, win_length, "sos_stft_f32");
    if (rc) return rc;
    if (n_samples <= n_fft / 2 || n_frames != 1 + n_samples / hop || batch < 1 || batch > 65535 || wave_stride < n_samples) {
        sos_set_error("sos_stft_f32: bad geometry n_fft=%d hop=%d win=%d n=%lld T=%lld", n_fft, hop, win_length,
                      (long long)n_samples, (long long)n_frames);
        return SOS_EINVAL;
    }
    StftParams p;
    p.wave = wave; p.wave_stride = wave_stride; p.n_samples = n_samples; p.T = n_frames;
    p.dhi = (const uint4*)mat_hi; p.dlo = (const uint4*)mat_lo;
    p.n_fft = n_fft; p.hop = hop; p.win = win_length; p.nbins = n_fft / 2 + 1;
    p.ksteps = win_length / 16; p.rtiles = 2 * p.nbins / 32;
    p.span = ((FE_COLS - 1) * hop + win_length + 7) & ~7;
    p.out = out; p.n_tab = clip_samples;
    if (p.rtiles > 16) { sos_set_error("sos_stft_f32: n_fft too large for one workgroup's row tiles"); return SOS_ENOSPC; }
    const size_t lds = (size_t)p.span * 2 * 2;
    if (lds > 160 * 1024) { sos_set_error("sos_stft_f32: hop/window too long for LDS staging"); return SOS_ENOSPC; }
    static sos_device_once once;
    (void)sos_per_device_once(once, [] {
        (void)hipFuncSetAttribute((const void*)stft_mfma_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        return (int)SOS_OK;
    });
    // (Round 6, built and removed -- profiles/r06_stft_resident.txt: a PERSISTENT workgroup with its four row tiles' matrix
    // fragments resident in registers (25 k-steps x (hi, lo) = 200 registers per wave, one wave per SIMD) walking (clip, frame tile)
    // items with the next item's samples prefetched through registers -- no per-tile matrix stream from L2 (157 MB per B = 64
    // launch here).  Bit-identical, and SLOWER: 33.6 vs 25.8 us at B = 64, 110 vs 94 us at B = 256 -- one wave per SIMD exposes the
    // dependent-MFMA latency of its two accumulators and every LDS fragment read that three co-resident workgroups hide here.  The
    // floor of this formulation is the MFMA time of the three hi / lo passes: 7.5 us at B = 64 = 0.51 of the HBM roofline.)
    // 64 frames x all rows per workgroup would be 192 workgroups for 64 two-second clips (a quarter of the CUs idle, no
    // co-resident workgroup to cover staging / stores): the rows are split over FE_SRS workgroups that stage the same span
    dim3 grid((unsigned)((n_frames + FE_COLS - 1) / FE_COLS), (unsigned)batch, FE_SRS);
    hipLaunchKernelGGL(stft_mfma_kernel, grid, dim3(FE_SWAVES * 64), lds, (hipStream_t)stream, p);
    return sos_check_launch("sos_stft_f32");
}

extern "C" int sos_istft_f32(const float* spec, int64_t batch, int64_t n_frames, const void* mat_hi, const void* mat_lo,
                             const float* win_sq, int n_fft, int hop, int win_length, float* out, int64_t out_stride,
                             const int32_t* clip_frames, sos_stream_t stream) {
    if (!spec || !mat_hi || !mat_lo || !win_sq || !out) { sos_set_error("sos_istft_f32: null pointer"); return SOS_EINVAL; }
    int rc = fe_geometry_ok(n_fft, hop, win_length, "sos_istft_f32");
    if (rc) return rc;
    const int64_t n_out = (int64_t)hop * (n_frames - 1);
    if (n_frames < 2 || batch < 1 || batch > 65535 || out_stride < n_out) {
        sos_set_error("sos_istft_f32: bad geometry n_fft=%d hop=%d win=%d T=%lld", n_fft, hop, win_length, (long long)n_frames);
        return SOS_EINVAL;
    }
    IstftParams p;
    p.spec = spec; p.T = n_frames; p.ehi = (const uint4*)mat_hi; p.elo = (const uint4*)mat_lo; p.win2 = win_sq;
    p.n_fft = n_fft; p.hop = hop; p.win = win_length; p.nbins = n_fft / 2 + 1;
    p.ksteps = 2 * p.nbins / 16; p.rtiles = (win_length + 31) / 32;
    p.out = out; p.out_stride = out_stride; p.t_tab = clip_frames;
    if (p.rtiles > 16 || (2 * p.nbins) % FE_IKC) { sos_set_error("sos_istft_f32: window / n_fft not supported by the tile"); return SOS_ENOSPC; }
    size_t lds = (size_t)2 * FE_COLS * (FE_IKC * 2 + 16);
    const size_t ylds = (size_t)FE_COLS * (p.rtiles * 32 + 1) * 4 + (size_t)win_length * 4;      // frame signals + squared window
    if (ylds > lds) lds = ylds;
    if (lds > 160 * 1024) { sos_set_error("sos_istft_f32: window too long for LDS staging"); return SOS_ENOSPC; }
    static sos_device_once once;
    (void)sos_per_device_once(once, [] {
        (void)hipFuncSetAttribute((const void*)istft_mfma_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        return (int)SOS_OK;
    });
    dim3 grid((unsigned)((n_out + (int64_t)FE_IADV * hop - 1) / ((int64_t)FE_IADV * hop)), (unsigned)batch);
    hipLaunchKernelGGL(istft_mfma_kernel, grid, dim3(FE_IWAVES * 64), lds, (hipStream_t)stream, p);
    return sos_check_launch("sos_istft_f32");
}
