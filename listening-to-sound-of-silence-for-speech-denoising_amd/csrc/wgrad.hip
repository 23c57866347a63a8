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

#define WG_WAVES 8            // waves per workgroup (512 threads, two per SIMD)
#define WG_THREADS (WG_WAVES * 64)
#define WG_PAIRS 4            // (tap, n-tile) pairs per wave; x MT m-tiles = accumulator tiles per wave

struct WgParams {
    const bf16_t* g;          // [B][Hg][Wg][g_cs]   (tile side, m channels)
    const bf16_t* x;          // [B][Hx][Wx][x_cs]   (patch side, n channels)
    float* partial;           // [ksplit][taps][Mp][Np]
    int B, Hg, Wg, g_cs, g_off, Hx, Wx, x_cs, x_off;
    int M, N, Mp, Np;         // logical / padded-to-32 dims
    int kh, kw, stride, dh, dw, pad_t, pad_l, pad_mode;
    int tiles_h, tiles_w, ngw; // pixel tiles per (image, residue class group)
    int steps_per_split, nsteps, ksplit;
    int NC, logTH, logTW, PH, PW, npix;
    int dbuf, ppk;            // LDS double buffering on/off; staging pieces per thread per k-step
    int dbg;                  // SOS_WGRAD_DBG ablation mask (0 in production)
    unsigned magic_pw, magic_ph; // ceil(2^32 / PW), ceil(2^32 / PH): exact division of small indices by mulhi
};

__device__ __forceinline__ uint2 lds_tr(unsigned addr) {
    uint2 v;
    asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(v) : "v"(addr));
    return v;
}

// Workgroup = MT m-tiles (32 rows of dW each) x NTB n-tiles x all taps, 8 waves.  Wave w owns the
// (tap, n-tile) pairs {w, w+8, w+16, w+24} for ALL MT m-tiles, so every transposed X fragment feeds
// MT MFMAs and every G fragment WG_PAIRS of them (LDS reads per MFMA ~1.2 instead of 2.5).
// The operands of the NEXT pixel tile are fetched 1/16th per k-step into the other LDS buffer while
// the MFMAs of the current tile run.
template <int MT, int NTB>
__global__ __launch_bounds__(WG_THREADS) void wgrad_kernel(WgParams p) {
    constexpr int XC = NTB * 32, GC = MT * 32;
    // Row pitches chosen for ds_read_b64_tr_b16: a 32-lane half reads 4 pixel rows x 64 B; pitch/4 = 16 or
    // 48 (mod 64) dwords puts the 4 rows on disjoint 16-bank windows (conflict free).
    constexpr int XSTRIDE = NTB == 1 ? 64 : (NTB == 2 ? 192 : 320);
    constexpr int GSTRIDE = MT == 1 ? 64 : 192;
    constexpr int XCPR = XC / 8, GCPR = GC / 8;          // 16-byte pieces per pixel
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int gbytes = 256 * GSTRIDE;
    const int bufbytes = gbytes + p.npix * XSTRIDE;
    const unsigned sbase = (unsigned)(uintptr_t)smem;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int split = blockIdx.x;
    const int m0 = blockIdx.y * GC, n0 = blockIdx.z * XC;
    const int taps = p.kh * p.kw;
    const int npairs = taps * NTB;
    const int g4 = lane >> 4, s = lane & 15;
    const int chan_off = (16 * (g4 & 1) + 4 * (s & 3)) * 2;      // byte offset of this lane's 4-channel run
    const int krow = 8 * (g4 >> 1) + (s >> 2);                   // pixel (k) inside a 16-pixel k-step; +4 for 2nd read
    const int TH = 1 << p.logTH, TW = 1 << p.logTW;
    const int TWm = TW - 1, THm = TH - 1, lsh = p.logTW + p.logTH;
    const bool reflect = p.pad_mode == SOS_PAD_REFLECT;
    const int ngp = 256 * GCPR, npieces = ngp + p.npix * XCPR;

    f32x16 acc[MT][WG_PAIRS];
#pragma unroll
    for (int a = 0; a < MT; ++a)
#pragma unroll
        for (int t = 0; t < WG_PAIRS; ++t)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[a][t][e] = 0.f;

    // per-wave offsets of its pairs inside the X image (tap shift + n-tile)
    unsigned toff[WG_PAIRS];
#pragma unroll
    for (int u = 0; u < WG_PAIRS; ++u) {
        const int pr = min(wave + WG_WAVES * u, npairs - 1);
        const int tap = pr / NTB, nt = pr - tap * NTB;
        const int ta = tap / p.kw, tb = tap - ta * p.kw;
        toff[u] = (unsigned)((ta * p.PW + tb) * XSTRIDE + nt * 64);
    }

    // tile origin of pixel tile `step` (wave-uniform, computed once per tile)
    struct Origin { int b, rw0, ho_base, wo_base, hin0, win0; };
    auto origin_of = [&](int step) {
        int t = step;
        const int tj = t % p.tiles_w; t /= p.tiles_w;
        const int ti = t % p.tiles_h; t /= p.tiles_h;
        const int gw = t % p.ngw; t /= p.ngw;
        const int rh = t % p.dh; t /= p.dh;
        Origin o;
        o.b = t; o.rw0 = gw * p.NC;
        o.ho_base = rh + ti * TH * p.dh; o.wo_base = o.rw0 + tj * TW * p.dw;
        o.hin0 = o.ho_base * p.stride - p.pad_t; o.win0 = o.wo_base * p.stride - p.pad_l;
        return o;
    };
    // one 16-byte staging piece of a pixel tile: value + LDS byte offset inside a buffer.  Index
    // decoding uses shifts / constant divisors / mulhi magic numbers only (it runs once per k-step
    // next to the MFMAs and must stay a handful of VALU instructions).
    auto fetch_piece = [&](const Origin& o, int id, uint4& v, int& dst) {
        const int idc = min(id, npieces - 1);
        if (idc < ngp) {
            const int m = idc / GCPR, q = idc - m * GCPR;
            const int j = m & TWm, i = (m >> p.logTW) & THm, cls = m >> lsh;
            const int h = o.ho_base + i * p.dh, w = o.wo_base + cls + j * p.dw;
            const int ch = m0 + q * 8;          // channels past M inside a stored 8-run are the producer's zero padding
            const bool ok = h < p.Hg && w < p.Wg && (cls == 0 || o.rw0 + cls < p.dw) && ch < p.M && p.g_off + ch + 8 <= p.g_cs;
            const int hc = min(h, p.Hg - 1), wc = min(w, p.Wg - 1);
            const int cc = min(p.g_off + ch, p.g_cs - 8);
            const uint4 r = *(const uint4*)(p.g + (((long long)o.b * p.Hg + hc) * p.Wg + wc) * p.g_cs + cc);
            v = ok ? r : make_uint4(0u, 0u, 0u, 0u);
            dst = m * GSTRIDE + q * 16;
        } else {
            const int pc = idc - ngp;
            const int pix = pc / XCPR, cl = pc - pix * XCPR;
            const int rr = (int)__umulhi((unsigned)pix, p.magic_pw), c = pix - rr * p.PW;
            const int cls = (int)__umulhi((unsigned)rr, p.magic_ph), r = rr - cls * p.PH;
            int h = o.hin0 + r * p.dh, w = o.win0 + cls * p.stride + c * p.dw;
            bool ok = reflect || (h >= 0 && h < p.Hx && w >= 0 && w < p.Wx);
            h = reflect ? reflect_index(h, p.Hx) : min(max(h, 0), p.Hx - 1);
            w = reflect ? reflect_index(w, p.Wx) : min(max(w, 0), p.Wx - 1);
            const int ch = n0 + cl * 8;
            ok = ok && ch < p.N && p.x_off + ch + 8 <= p.x_cs;
            const int cc = min(p.x_off + ch, p.x_cs - 8);
            const uint4 rv = *(const uint4*)(p.x + (((long long)o.b * p.Hx + h) * p.Wx + w) * p.x_cs + cc);
            v = ok ? rv : make_uint4(0u, 0u, 0u, 0u);
            dst = gbytes + pix * XSTRIDE + cl * 16;
        }
    };
    // stage a whole pixel tile synchronously into buffer `buf`
    auto stage_all = [&](int step, int buf) {
        const Origin o = origin_of(step);
        for (int id0 = 0; id0 < npieces; id0 += WG_THREADS * 4) {
            uint4 v[4];
            int d[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) fetch_piece(o, id0 + tid + WG_THREADS * u, v[u], d[u]);
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (id0 + tid + WG_THREADS * u < npieces) *(uint4*)(smem + buf * bufbytes + d[u]) = v[u];
        }
    };

    const int step0 = split * p.steps_per_split;
    const int step1 = min(step0 + p.steps_per_split, p.nsteps);
    int cur = 0;
    if (step0 < step1 && p.dbuf) stage_all(step0, 0);
    for (int step = step0; step < step1; ++step) {
        if (!p.dbuf) {
            __syncthreads();                       // previous tile's LDS reads are done
            stage_all(step, 0);
        }
        __syncthreads();                           // tile `step` is complete in buffer `cur`
        const bool more = p.dbuf && step + 1 < step1 && !(p.dbg & 1);
        const Origin onext = origin_of(more ? step + 1 : step);
        const unsigned gb = sbase + cur * bufbytes, xb = gb + gbytes;
#pragma unroll 1
        for (int ks = 0; ks < 16; ++ks) {
            // 1/16th of the next tile: global loads now, LDS stores after this k-step's MFMAs
            uint4 pv[2];
            int pd[2];
            if (more) {
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    if (j < p.ppk) fetch_piece(onext, (ks * p.ppk + j) * WG_THREADS + tid, pv[j], pd[j]);
            }
            const int k0 = ks * 16 + krow, k1 = k0 + 4;
            uint2 a0[MT], a1[MT];
            const bool rd = !(p.dbg & 2) || ks == 0;
#pragma unroll
            for (int a = 0; a < MT; ++a) {
                const unsigned ga = gb + k0 * GSTRIDE + a * 64 + chan_off;
                if (rd) { a0[a] = lds_tr(ga); a1[a] = lds_tr(ga + 4 * GSTRIDE); } else { a0[a] = make_uint2(ks, k0); a1[a] = a0[a]; }
            }
            const int pp0 = (((k0 >> lsh) * p.PH + ((k0 >> p.logTW) & THm) * p.stride) * p.PW + (k0 & TWm) * p.stride);
            const int pp1 = (((k1 >> lsh) * p.PH + ((k1 >> p.logTW) & THm) * p.stride) * p.PW + (k1 & TWm) * p.stride);
            const unsigned xa = xb + pp0 * XSTRIDE + chan_off;
            const unsigned xs = (unsigned)((pp1 - pp0) * XSTRIDE);
            uint2 b0[WG_PAIRS], b1[WG_PAIRS];
#pragma unroll
            for (int u = 0; u < WG_PAIRS; ++u) {
                if (rd) { b0[u] = lds_tr(xa + toff[u]); b1[u] = lds_tr(xa + toff[u] + xs); } else { b0[u] = make_uint2(xa, u); b1[u] = b0[u]; }
            }
            // the waits name their registers so that every consumer is ordered behind them
            if constexpr (MT == 1) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a0[0]), "+v"(a1[0]));
            if constexpr (MT == 2) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a0[0]), "+v"(a1[0]), "+v"(a0[1]), "+v"(a1[1]));
            if constexpr (MT == 3)
                asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a0[0]), "+v"(a1[0]), "+v"(a0[1]), "+v"(a1[1]), "+v"(a0[2]), "+v"(a1[2]));
            asm volatile("s_waitcnt lgkmcnt(0)"
                         : "+v"(b0[0]), "+v"(b1[0]), "+v"(b0[1]), "+v"(b1[1]), "+v"(b0[2]), "+v"(b1[2]), "+v"(b0[3]), "+v"(b1[3]));
#pragma unroll
            for (int u = 0; u < WG_PAIRS; ++u) {
                if (wave + WG_WAVES * u < npairs) {
                    const bf16x8 bfr = __builtin_bit_cast(bf16x8, make_uint4(b0[u].x, b0[u].y, b1[u].x, b1[u].y));
#pragma unroll
                    for (int a = 0; a < MT; ++a) {
                        const bf16x8 af = __builtin_bit_cast(bf16x8, make_uint4(a0[a].x, a0[a].y, a1[a].x, a1[a].y));
                        acc[a][u] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, bfr, acc[a][u], 0, 0, 0);
                    }
                }
            }
            if (more) {
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    if (j < p.ppk && (ks * p.ppk + j) * WG_THREADS + tid < npieces)
                        *(uint4*)(smem + (cur ^ 1) * bufbytes + pd[j]) = pv[j];
            }
        }
        if (p.dbuf) cur ^= 1;
    }
    // ---- write this split's partial tiles: D[row = m][col = n], row = (reg&3)+8*(reg>>2)+4*(lane>>5), col = lane&31
    float* out = p.partial + (size_t)split * taps * p.Mp * p.Np;
#pragma unroll
    for (int u = 0; u < WG_PAIRS; ++u) {
        const int pr = wave + WG_WAVES * u;
        if (pr >= npairs) continue;
        const int tap = pr / NTB, nt = pr - tap * NTB;
        const int n = n0 + nt * 32 + (lane & 31);
        if (n >= p.Np) continue;
#pragma unroll
        for (int a = 0; a < MT; ++a) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + a * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                if (m < p.Mp) out[((size_t)tap * p.Mp + m) * p.Np + n] = acc[a][u][r];
            }
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
    if (taps > WG_WAVES * WG_PAIRS) { sos_set_error("sos_conv2d_wgrad: kernel with %d taps not supported", taps); return SOS_ENOSPC; }
    int ntb = WG_WAVES * WG_PAIRS / taps;         // (tap, n-tile) pairs per workgroup <= 32
    ntb = ntb >= 4 ? 4 : (ntb >= 2 ? 2 : 1);
    const int ntiles_n = p.Np / 32, ntiles_m = p.Mp / 32;
    if (ntb > ntiles_n) ntb = ntiles_n >= 4 ? 4 : (ntiles_n >= 2 ? 2 : 1);
    const int mgroups = (ntiles_m + 2) / 3;
    const int mt = (ntiles_m + mgroups - 1) / mgroups;          // 1..3 m-tiles per workgroup, balanced
    const int gstride = mt == 1 ? 64 : 192;
    const int Hc = (d->Hg + d->dil_h - 1) / d->dil_h, Wc = (d->Wg + d->dil_w - 1) / d->dil_w;
    // pixel tile (NC x TH x TW = 256): fewest k-steps among the shapes whose operands fit LDS (double
    // buffered if possible); shrink the channel tile if none fits
    size_t lds = 0;
    for (;;) {
        const int xstride = ntb == 1 ? 64 : (ntb == 2 ? 192 : 320);
        double best = 1e300;
        int bnc = 0, bth = 0, btw = 0, bdb = 0;
        for (int lnc = 0; lnc <= 6; ++lnc) {
            const int NC = 1 << lnc;
            if (NC > 1 && (d->stride > 1 || NC > d->dil_w)) break;
            for (int lth = 0; lth + lnc <= 8; ++lth) {
                const int ltw = 8 - lnc - lth;
                if (ltw < 2) continue;
                const int TH = 1 << lth, TW = 1 << ltw;
                const int PH = (TH - 1) * d->stride + d->kh, PW = (TW - 1) * d->stride + d->kw;
                const size_t one = (size_t)256 * gstride + (size_t)NC * PH * PW * xstride;
                if (one > 160 * 1024) continue;
                const int db = 2 * one <= 160 * 1024;
                const double steps = (double)((Hc + TH - 1) / TH) * ((Wc + TW - 1) / TW) * ((d->dil_w + NC - 1) / NC);
                const double cost = steps * (256.0 * taps + (db ? 1.0 : 6.0) * NC * PH * PW);
                if (cost < best) { best = cost; bnc = NC; bth = lth; btw = ltw; bdb = db; lds = db ? 2 * one : one; }
            }
        }
        if (bnc) { p.NC = bnc; p.logTH = bth; p.logTW = btw; p.dbuf = bdb; break; }
        if (ntb == 1) { sos_set_error("sos_conv2d_wgrad: patch does not fit LDS"); return SOS_ENOSPC; }
        ntb >>= 1;
    }
    {
        const int TH = 1 << p.logTH, TW = 1 << p.logTW;
        p.tiles_h = (Hc + TH - 1) / TH; p.tiles_w = (Wc + TW - 1) / TW; p.ngw = (d->dil_w + p.NC - 1) / p.NC;
        p.PH = (TH - 1) * d->stride + d->kh; p.PW = (TW - 1) * d->stride + d->kw;
        p.npix = p.NC * p.PH * p.PW;
    }
    {
        const int np = 256 * mt * 4 + p.npix * ntb * 4;
        p.ppk = (np + WG_THREADS * 16 - 1) / (WG_THREADS * 16);
        if (p.ppk > 2) { p.dbuf = 0; lds = (size_t)256 * gstride + (size_t)p.npix * (ntb == 1 ? 64 : (ntb == 2 ? 192 : 320)); }
    }
    { const char* e = getenv("SOS_WGRAD_DBG"); p.dbg = e ? atoi(e) : 0; }
    p.magic_pw = (unsigned)((0x100000000ULL + p.PW - 1) / p.PW);
    p.magic_ph = (unsigned)((0x100000000ULL + p.PH - 1) / p.PH);
    p.nsteps = d->B * d->dil_h * p.ngw * p.tiles_h * p.tiles_w;
    p.ksplit = d->ksplit;
    p.steps_per_split = (p.nsteps + d->ksplit - 1) / d->ksplit;
    dim3 grid((unsigned)d->ksplit, (unsigned)mgroups, (unsigned)((ntiles_n + ntb - 1) / ntb));
    hipStream_t s = (hipStream_t)stream;
    static bool attr_done = false;
#define SOS_WG_CASE(MTV, NTBV)                                                                                       \
    if (mt == MTV && ntb == NTBV) {                                                                                  \
        hipLaunchKernelGGL((wgrad_kernel<MTV, NTBV>), grid, dim3(WG_THREADS), lds, s, p);                            \
    }
    if (!attr_done) {       // every instantiation may use the full 160 KB of LDS
        (void)hipFuncSetAttribute((const void*)wgrad_kernel<1, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute((const void*)wgrad_kernel<1, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute((const void*)wgrad_kernel<1, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute((const void*)wgrad_kernel<2, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute((const void*)wgrad_kernel<2, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute((const void*)wgrad_kernel<2, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute((const void*)wgrad_kernel<3, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute((const void*)wgrad_kernel<3, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute((const void*)wgrad_kernel<3, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_done = true;
    }
    SOS_WG_CASE(1, 1) SOS_WG_CASE(1, 2) SOS_WG_CASE(1, 4) SOS_WG_CASE(2, 1) SOS_WG_CASE(2, 2) SOS_WG_CASE(2, 4)
    SOS_WG_CASE(3, 1) SOS_WG_CASE(3, 2) SOS_WG_CASE(3, 4)
#undef SOS_WG_CASE
    int rc = sos_check_launch("sos_conv2d_wgrad");
    if (rc) return rc;
    const long long total = (long long)d->M * d->N * taps;
    long long gb = (total + 255) / 256;
    if (gb > 4096) gb = 4096;
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)gb), dim3(256), 0, s, d->partial, d->ksplit, taps, d->M, d->N,
                       p.Mp, p.Np, d->dw, d->accumulate, d->scale);
    return sos_check_launch("sos_conv2d_wgrad(reduce)");
}
