// wgrad.hip -- weight gradient of Conv2d / ConvTranspose2d / Linear on bf16 MFMA (gfx950).
//
// Backward of the conv blocks of M1/networks.py:28-51 and M2/networks.py:28-51,97-149 with
// respect to their weights (autograd of F.conv2d / F.conv_transpose2d / F.linear in the reference's
// `loss.backward()`, M1/agent.py:106-111, M2/agent.py:101-106):
//     dW[m][n][a][b] = sum over pixels p of  G[p][m] * X[p*stride + (a,b)*dil - pad][n]
// For a conv, G = grad of the raw conv output (m = cout) and X = the layer input (n = cin); for the
// transposed conv the roles swap (G = layer input, X = output grad, stride 2) and the result is
// already in ConvTranspose2d's (Cin, Cout, kh, kw) layout.
//
// The contraction runs over PIXELS while activations are stored channel-contiguous (NHWC), so the
// MFMA operands (8 consecutive k per lane) are produced by gfx950's LDS transpose read
// ds_read_b64_tr_b16 straight from the pixel-major LDS images -- tap shifts are whole-row address
// offsets, so no alignment problem and no explicit transpose pass.  One workgroup owns 32 rows (m) x
// up to NTB*32 columns (n) x all taps of dW in registers (<= 16 accumulator tiles per wave) and
// walks a slice of the pixels (split-K); partial sums go to a [ksplit][taps][M][N] fp32 buffer
// that a deterministic reduce kernel folds into the OIHW gradient (no atomics).
#include "sos_common.h"
#include <stdlib.h>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define WG_MAXT 4             // accumulator tiles per wave (64 registers: two 8-wave workgroups per CU,
                              // so one workgroup's operand staging overlaps the other's MFMAs)
#define WG_WAVES 8            // waves per workgroup (512 threads)
#define WG_THREADS (WG_WAVES * 64)
// pixel tile: NC residue classes x TH x TW (dilation-strided coordinates, like the forward kernel),
// 256 pixels per k-step

struct WgParams {
    const bf16_t* g;          // [B][Hg][Wg][g_cs]   (tile side, m channels)
    const bf16_t* x;          // [B][Hx][Wx][x_cs]   (patch side, n channels)
    float* partial;           // [ksplit][taps][Mp][Np]
    int B, Hg, Wg, g_cs, g_off, Hx, Wx, x_cs, x_off;
    int M, N, Mp, Np;         // logical / padded-to-32 dims
    int kh, kw, stride, dh, dw, pad_t, pad_l, pad_mode;
    int NTB;                  // n-tiles (of 32) per workgroup
    int tiles_h, tiles_w, ngw; // pixel tiles per (image, residue class group)
    int steps_per_split, nsteps, ksplit;
    int NC, logTH, logTW, PH, PW, npix;
};

__device__ __forceinline__ uint2 lds_tr(unsigned addr) {
    uint2 v;
    asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(v) : "v"(addr));
    return v;
}

template <int NTB>
__global__ __launch_bounds__(WG_THREADS) void wgrad_kernel(WgParams p) {
    constexpr int XC = NTB * 32;                // patch channels held in LDS
    // Row pitches chosen for the ds_read_b64_tr_b16 access: a 32-lane half reads 4 pixel rows x 64 B;
    // pitch/4 = 16 or 48 (mod 64) dwords puts the 4 rows on disjoint 16-bank windows (conflict free).
    constexpr int XSTRIDE = NTB == 1 ? 64 : (NTB == 2 ? 192 : 320);   // bytes per patch pixel
    constexpr int GSTRIDE = 64;                 // bytes per tile pixel of G (32 channels, no padding)
    constexpr int XCPR = XC / 8;                // 16-byte pieces per patch pixel
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* gimg = smem;                                   // [256][GSTRIDE]
    char* ximg = smem + 256 * GSTRIDE;                   // [npix][XSTRIDE]
    const unsigned gbase = (unsigned)(uintptr_t)gimg, xbase = (unsigned)(uintptr_t)ximg;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int split = blockIdx.x, mt = blockIdx.y, ng = blockIdx.z;
    const int m0 = mt * 32, n0 = ng * XC;
    const int taps = p.kh * p.kw;
    const int ntl = taps * NTB;                          // 32x32 output tiles of this workgroup
    // per-lane geometry of the transpose read: 16-lane group g4, in-group lane s
    const int g4 = lane >> 4, s = lane & 15;
    const int chan_off = (16 * (g4 & 1) + 4 * (s & 3)) * 2;      // byte offset of the 4-channel run
    const int krow = 8 * (g4 >> 1) + (s >> 2);                   // pixel (k) row inside a 16-pixel k-step; +4 for 2nd read

    f32x16 acc[WG_MAXT];
#pragma unroll
    for (int t = 0; t < WG_MAXT; ++t)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[t][e] = 0.f;

    const int step0 = split * p.steps_per_split;
    const int step1 = min(step0 + p.steps_per_split, p.nsteps);
    const int TH = 1 << p.logTH, TW = 1 << p.logTW;
    const int TWm = TW - 1, THm = TH - 1, lsh = p.logTW + p.logTH;
    const bool reflect = p.pad_mode == SOS_PAD_REFLECT;
    constexpr int GPF = 1024 / WG_THREADS;        // G pieces per thread (256 px x 4 pieces)
    constexpr int XPF = 8;                        // X pieces per thread fetched ahead (rest: synchronous)

    // global -> register fetch of one k-step's operands (issued one step ahead: the loads land while
    // the MFMAs of the current step run; the LDS images are rewritten after the step's barrier)
    auto fetch = [&](int step, uint4 (&gq)[GPF], uint4 (&xq)[XPF], int first_piece, bool sync_rest) {
        int t = step;
        const int tj = t % p.tiles_w; t /= p.tiles_w;
        const int ti = t % p.tiles_h; t /= p.tiles_h;
        const int gw = t % p.ngw; t /= p.ngw;
        const int rh = t % p.dh; t /= p.dh;
        const int b = t;
        const int rw0 = gw * p.NC;
        const int ho_base = rh + ti * TH * p.dh, wo_base = rw0 + tj * TW * p.dw;
        const int hin0 = ho_base * p.stride - p.pad_t, win0 = wo_base * p.stride - p.pad_l;
        if (!sync_rest) {
#pragma unroll
            for (int u = 0; u < GPF; ++u) {
                const int m = (tid >> 2) + (WG_THREADS / 4) * u, q = tid & 3;
                const int j = m & TWm, i = (m >> p.logTW) & THm, cls = m >> lsh;
                const int h = ho_base + i * p.dh, w = wo_base + cls + j * p.dw;
                const int ch = m0 + q * 8;      // channels past M inside a stored 8-run are the producer's zero padding
                const bool ok = h < p.Hg && w < p.Wg && (cls == 0 || rw0 + cls < p.dw) && ch < p.M && p.g_off + ch + 8 <= p.g_cs;
                const int hc = min(h, p.Hg - 1), wc = min(w, p.Wg - 1);
                const int cc = min(p.g_off + ch, p.g_cs - 8);
                uint4 v = *(const uint4*)(p.g + (((long long)b * p.Hg + hc) * p.Wg + wc) * p.g_cs + cc);
                gq[u] = ok ? v : make_uint4(0u, 0u, 0u, 0u);
            }
        }
        const int npieces = p.npix * XCPR;
#pragma unroll
        for (int u = 0; u < XPF; ++u) {
            const int piece = first_piece + tid + WG_THREADS * u;
            const int pc = min(piece, npieces - 1);
            const int pix = pc / XCPR, cl = pc - pix * XCPR;
            const int c = pix % p.PW, rr = pix / p.PW;
            const int r = rr % p.PH, cls = rr / p.PH;
            int h = hin0 + r * p.dh, w = win0 + cls * p.stride + c * p.dw;
            bool ok = reflect || (h >= 0 && h < p.Hx && w >= 0 && w < p.Wx);
            h = reflect ? reflect_index(h, p.Hx) : min(max(h, 0), p.Hx - 1);
            w = reflect ? reflect_index(w, p.Wx) : min(max(w, 0), p.Wx - 1);
            const int ch = n0 + cl * 8;
            ok = ok && ch < p.N && p.x_off + ch + 8 <= p.x_cs;
            const int cc = min(p.x_off + ch, p.x_cs - 8);
            const uint4 v = *(const uint4*)(p.x + (((long long)b * p.Hx + h) * p.Wx + w) * p.x_cs + cc);
            xq[u] = ok ? v : make_uint4(0u, 0u, 0u, 0u);
        }
    };
    auto put = [&](const uint4 (&gq)[GPF], const uint4 (&xq)[XPF], int first_piece, bool with_g) {
        if (with_g) {
#pragma unroll
            for (int u = 0; u < GPF; ++u) {
                const int m = (tid >> 2) + (WG_THREADS / 4) * u, q = tid & 3;
                *(uint4*)(gimg + m * GSTRIDE + q * 16) = gq[u];
            }
        }
        const int npieces = p.npix * XCPR;
#pragma unroll
        for (int u = 0; u < XPF; ++u) {
            const int piece = first_piece + tid + WG_THREADS * u;
            if (piece < npieces) {
                const int pix = piece / XCPR, cl = piece - pix * XCPR;
                *(uint4*)(ximg + pix * XSTRIDE + cl * 16) = xq[u];
            }
        }
    };

    for (int step = step0; step < step1; ++step) {
        __syncthreads();                              // previous step's LDS reads are done
        {
            uint4 gq[GPF], xq[XPF];
            fetch(step, gq, xq, 0, false);
            put(gq, xq, 0, true);
        }
        // patches larger than XPF pieces per thread: the remainder is fetched synchronously
        for (int fp = WG_THREADS * XPF; fp < p.npix * XCPR; fp += WG_THREADS * XPF) {
            uint4 g2[GPF], x2[XPF];
            fetch(step, g2, x2, fp, true);
            put(g2, x2, fp, false);
        }
        __syncthreads();
        // ---- 16 k-steps of 16 pixels; every wave walks its own list of (tap, n-tile) output tiles
        unsigned toff[1][4];
#pragma unroll
        for (int g = 0; g < 1; ++g)
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int tl = min(wave + WG_WAVES * (4 * g + u), ntl - 1);
                const int tap = tl / NTB, nt = tl - tap * NTB;
                const int ta = tap / p.kw, tb = tap - ta * p.kw;
                toff[g][u] = (unsigned)((ta * p.PW + tb) * XSTRIDE + nt * 64);
            }
#pragma unroll 1
        for (int ks = 0; ks < 16; ++ks) {
            const int k0 = ks * 16 + krow, k1 = k0 + 4;
            const unsigned ga = gbase + k0 * GSTRIDE + chan_off;
            uint2 a0 = lds_tr(ga), a1 = lds_tr(ga + 4 * GSTRIDE);
            const int pp0 = (((k0 >> lsh) * p.PH + ((k0 >> p.logTW) & THm) * p.stride) * p.PW + (k0 & TWm) * p.stride);
            const int pp1 = (((k1 >> lsh) * p.PH + ((k1 >> p.logTW) & THm) * p.stride) * p.PW + (k1 & TWm) * p.stride);
            const unsigned xa = xbase + pp0 * XSTRIDE + chan_off;
            const unsigned xs = (unsigned)((pp1 - pp0) * XSTRIDE);
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a0), "+v"(a1));
            const bf16x8 af = __builtin_bit_cast(bf16x8, make_uint4(a0.x, a0.y, a1.x, a1.y));
#pragma unroll
            for (int g = 0; g < WG_MAXT / 4; ++g) {
                if (wave + 4 * WG_WAVES * g >= ntl) break;          // wave-uniform
                uint2 b0[4], b1[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    b0[u] = lds_tr(xa + toff[g][u]);
                    b1[u] = lds_tr(xa + toff[g][u] + xs);
                }
                // the wait names its registers so that every consumer is ordered behind it
                asm volatile("s_waitcnt lgkmcnt(0)"
                             : "+v"(b0[0]), "+v"(b1[0]), "+v"(b0[1]), "+v"(b1[1]), "+v"(b0[2]), "+v"(b1[2]), "+v"(b0[3]), "+v"(b1[3])
                             );
#pragma unroll
                for (int u = 0; u < 4; ++u)
                    if (wave + WG_WAVES * (4 * g + u) < ntl) {
                        const bf16x8 bfr = __builtin_bit_cast(bf16x8, make_uint4(b0[u].x, b0[u].y, b1[u].x, b1[u].y));
                        acc[4 * g + u] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, bfr, acc[4 * g + u], 0, 0, 0);
                    }
            }
        }
    }
    // ---- write this split's partial tiles: D[row = m][col = n], row = (reg&3)+8*(reg>>2)+4*(lane>>5), col = lane&31
    float* out = p.partial + (size_t)split * taps * p.Mp * p.Np;
#pragma unroll
    for (int tt = 0; tt < WG_MAXT; ++tt) {
        const int tl = wave + WG_WAVES * tt;
        if (tl >= ntl) continue;
        const int tap = tl / NTB, nt = tl - tap * NTB;
        const int n = n0 + nt * 32 + (lane & 31);
        if (n >= p.Np) continue;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = m0 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
            out[((size_t)tap * p.Mp + m) * p.Np + n] = acc[tt][r];
        }
    }
}

// dW[m][n][tap] (+)= sum over splits of partial[s][tap][m][n]; also used for 1x1 / Linear weights.
__global__ void wgrad_reduce_kernel(const float* __restrict__ partial, int ksplit, int taps, int M, int N, int Mp,
                                    int Np, float* __restrict__ dw, int accumulate, float scale) {
    const long long total = (long long)M * N * taps;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int tap = (int)(i % taps);
        const long long r = i / taps;
        const int n = (int)(r % N), m = (int)(r / N);
        float acc = 0.f;
        for (int s = 0; s < ksplit; ++s) acc += partial[(((size_t)s * taps + tap) * Mp + m) * Np + n];
        acc *= scale;
        dw[i] = accumulate ? dw[i] + acc : acc;
    }
}

extern "C" int64_t sos_wgrad_workspace_bytes(const sos_wgrad_desc* d) {
    if (!d) return -1;
    const int64_t Mp = (d->M + 31) / 32 * 32, Np = (d->N + 31) / 32 * 32;
    return (int64_t)d->ksplit * d->kh * d->kw * Mp * Np * 4;
}

extern "C" int sos_conv2d_wgrad(const sos_wgrad_desc* d, sos_stream_t stream) {
    if (!d || !d->g || !d->x || !d->partial || !d->dw) { sos_set_error("sos_conv2d_wgrad: null pointer"); return SOS_EINVAL; }
    if (d->M < 1 || d->N < 1 || d->g_cs % 8 || d->x_cs % 8 || d->g_off % 8 || d->x_off % 8 || d->kh < 1 || d->kw < 1 ||
        d->stride < 1 || d->dil_h < 1 || d->dil_w < 1 || (d->stride > 1 && (d->dil_h > 1 || d->dil_w > 1)) ||
        d->ksplit < 1 || d->B < 1) {
        sos_set_error("sos_conv2d_wgrad: bad descriptor");
        return SOS_EINVAL;
    }
    WgParams p;
    p.g = (const bf16_t*)d->g; p.x = (const bf16_t*)d->x; p.partial = d->partial;
    p.B = d->B; p.Hg = d->Hg; p.Wg = d->Wg; p.g_cs = d->g_cs; p.g_off = d->g_off;
    p.Hx = d->Hx; p.Wx = d->Wx; p.x_cs = d->x_cs; p.x_off = d->x_off;
    p.M = d->M; p.N = d->N; p.Mp = (d->M + 31) / 32 * 32; p.Np = (d->N + 31) / 32 * 32;
    p.kh = d->kh; p.kw = d->kw; p.stride = d->stride; p.dh = d->dil_h; p.dw = d->dil_w;
    p.pad_t = d->pad_top; p.pad_l = d->pad_left; p.pad_mode = d->pad_mode;
    const int taps = d->kh * d->kw;
    int ntb = WG_MAXT * WG_WAVES / taps;          // tiles per workgroup <= 64
    if (ntb > 4) ntb = 4;
    if (ntb < 1) ntb = 1;
    if (ntb == 3) ntb = 2;
    const int ntiles_n = p.Np / 32;
    if (ntb > ntiles_n) ntb = ntiles_n >= 4 ? 4 : (ntiles_n >= 2 ? 2 : 1);
    if (taps * ntb > WG_MAXT * WG_WAVES) { sos_set_error("sos_conv2d_wgrad: kernel with %d taps not supported", taps); return SOS_ENOSPC; }
    p.NTB = ntb;
    const int Hc = (d->Hg + d->dil_h - 1) / d->dil_h, Wc = (d->Wg + d->dil_w - 1) / d->dil_w;
    // pixel tile (NC x TH x TW = 256): fewest k-steps (best utilisation of the 256 lanes-worth of pixels)
    // among the shapes whose patch fits LDS with the current channel tile; shrink the channel tile if none
    size_t lds = 0;
    for (;;) {
        double best = 1e300;
        int bnc = 0, bth = 0, btw = 0;
        for (int lnc = 0; lnc <= 6; ++lnc) {
            const int NC = 1 << lnc;
            if (NC > 1 && (d->stride > 1 || NC > d->dil_w)) break;
            for (int lth = 0; lth + lnc <= 8; ++lth) {
                const int ltw = 8 - lnc - lth;
                if (ltw < 2) continue;                      // a transpose-read quad (4 pixels) stays inside one tile row
                const int TH = 1 << lth, TW = 1 << ltw;
                const int PH = (TH - 1) * d->stride + d->kh, PW = (TW - 1) * d->stride + d->kw;
                const size_t need = (size_t)256 * 64 + (size_t)NC * PH * PW * (ntb == 1 ? 64 : (ntb == 2 ? 192 : 320));
                if (need > 160 * 1024) continue;
                const double steps = (double)((Hc + TH - 1) / TH) * ((Wc + TW - 1) / TW) * ((d->dil_w + NC - 1) / NC);
                const double cost = steps * (256.0 * taps + 4.0 * NC * PH * PW);
                if (cost < best) { best = cost; bnc = NC; bth = lth; btw = ltw; lds = need; }
            }
        }
        if (bnc) { p.NC = bnc; p.logTH = bth; p.logTW = btw; break; }
        if (ntb == 1) { sos_set_error("sos_conv2d_wgrad: patch does not fit LDS"); return SOS_ENOSPC; }
        ntb >>= 1;
    }
    p.NTB = ntb;
    {
        const int TH = 1 << p.logTH, TW = 1 << p.logTW;
        p.tiles_h = (Hc + TH - 1) / TH; p.tiles_w = (Wc + TW - 1) / TW; p.ngw = (d->dil_w + p.NC - 1) / p.NC;
        p.PH = (TH - 1) * d->stride + d->kh; p.PW = (TW - 1) * d->stride + d->kw;
        p.npix = p.NC * p.PH * p.PW;
    }
    p.nsteps = d->B * d->dil_h * p.ngw * p.tiles_h * p.tiles_w;
    p.ksplit = d->ksplit;
    p.steps_per_split = (p.nsteps + d->ksplit - 1) / d->ksplit;
    dim3 grid((unsigned)d->ksplit, (unsigned)(p.Mp / 32), (unsigned)((ntiles_n + ntb - 1) / ntb));
    hipStream_t s = (hipStream_t)stream;
    static bool attr[5] = {false, false, false, false, false};
#define SOS_WG_LAUNCH(NTBV)                                                                                         \
    {                                                                                                               \
        if (!attr[NTBV]) {                                                                                          \
            (void)hipFuncSetAttribute((const void*)wgrad_kernel<NTBV>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); \
            attr[NTBV] = true;                                                                                      \
        }                                                                                                           \
        hipLaunchKernelGGL(wgrad_kernel<NTBV>, grid, dim3(WG_THREADS), lds, s, p);                                         \
    }
    if (ntb == 1) SOS_WG_LAUNCH(1) else if (ntb == 2) SOS_WG_LAUNCH(2) else SOS_WG_LAUNCH(4)
#undef SOS_WG_LAUNCH
    int rc = sos_check_launch("sos_conv2d_wgrad");
    if (rc) return rc;
    const long long total = (long long)d->M * d->N * taps;
    long long gb = (total + 255) / 256;
    if (gb > 4096) gb = 4096;
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)gb), dim3(256), 0, s, d->partial, d->ksplit, taps, d->M, d->N,
                       p.Mp, p.Np, d->dw, d->accumulate, d->scale);
    return sos_check_launch("sos_conv2d_wgrad(reduce)");
}
