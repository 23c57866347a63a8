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
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>
#include <algorithm>
#include <map>
#include <vector>
#include <mutex>
#include <type_traits>

typedef sos_half_t bf16x8 __attribute__((ext_vector_type(8)));   // 8 storage-type (bf16, or fp16 in the SOS_F16 build) MFMA operands
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

#define WG_WAVES 8            // waves per workgroup (512 threads, two per SIMD)
#define WG_THREADS (WG_WAVES * 64)
#ifndef SOS_WGRAD_PIPE
#define SOS_WGRAD_PIPE 0      // 1: register double-buffered k-step pipeline of the 25-tap kernels (measured SLOWER)
#endif
#define WG_PAIRS 4            // (tap, n-tile) pairs per wave; x MT m-tiles = accumulator tiles per wave

// Ablation switches (SOS_WGRAD_DBG bit mask: 1 no prefetch DMA, 8 DMA lanes all out of range, 16 ... all offset 0)
// exist only in `make ABLATE=1` builds.
#ifdef SOS_ABLATE
#define WDBG(bit) (p.dbg & (bit))
#else
#define WDBG(bit) 0
#endif

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
    int npixp;                // npix rounded up to 16 (one wave's DMA never straddles two sub-images)
    int bufbytes;             // bytes of one LDS operand buffer (multiple of 1 KB)
    int dbuf;                 // two operand buffers: the next tile is fetched while this one is multiplied
    int dbg;                  // SOS_WGRAD_DBG ablation mask (0 in production)
    int ny, nz, xcdmap;       // m-groups / n-groups of the 1-D grid (wgrad_kernel), XCD-aware id mapping on/off
    int tT, tcin, tpad;       // temporal taps (tT = 0: off): frames per clip, channels per frame, temporal padding
    int ntg, taps_all;        // kernels with more taps than one workgroup holds (7x7): ntg workgroups own kh tap ROWS each
                              // (kh / kw above are then the group's), taps_all = taps of the whole kernel
    int kord;                 // order of the 256 tile pixels along the contraction: 0 = (class, row, column), 1 = (class, column,
                              // row).  A k-step (16 / 32 consecutive tile pixels) none of whose pixels lies inside the image is
                              // skipped, so the order decides WHICH border pixels cost nothing: whole rows below the image (0) or
                              // whole columns right of it (1); the host counts both and keeps the cheaper
};

// tile pixel e (k order) -> (class, row i, column j)
__device__ __host__ __forceinline__ void wg_decode(int e, int logTH, int logTW, int kord, int& cls, int& i, int& j) {
    cls = e >> (logTH + logTW);
    if (kord) { i = e & ((1 << logTH) - 1); j = (e >> logTH) & ((1 << logTW) - 1); }
    else { j = e & ((1 << logTW) - 1); i = (e >> logTW) & ((1 << logTH) - 1); }
}

typedef __attribute__((address_space(3))) void* lds_ptr_t;

__device__ __forceinline__ uint2 lds_tr(unsigned addr) {
    uint2 v;
    asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(v) : "v"(addr));
    return v;
}
template <int OFF>
__device__ __forceinline__ uint2 lds_tr_off(unsigned addr) {      // same with an immediate byte offset
    uint2 v;
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
    return v;
}
__device__ __forceinline__ uint2 lds_read64(unsigned addr) {     // opaque to hipcc's LDS-DMA alias tracking
    uint2 v;
    asm volatile("ds_read_b64 %0, %1" : "=v"(v) : "v"(addr));
    return v;
}

#if __HIP_DEVICE_COMPILE__     // buffer-resource builtins exist in the device pass only; the host pass needs just the stubs
// origin of a pixel tile (wave-uniform, computed once per tile)
struct WgTile {
    __amdgpu_buffer_rsrc_t rg, rx;
    int gh0, gw0, xh0, xw0;
    unsigned gorg, xorg;
    bool xslow, gfast, xfast;   // reflect border tile; every pixel of the G tile / X patch is inside its image
};

// Operand staging shared by the MFMA kernels below: LDS-DMA (buffer_load_dwordx4 ... lds, no registers, no
// ds_write).  The LDS images are sub-images of CW = 8 << LP channels with a pixel pitch of 16 << LP bytes:
//     G: [GSUB sub-images][256 tile pixels][CW channels]     X: [XSUB sub-images][npixp patch pixels][CW channels]
// A DMA instruction moves 64 consecutive 16-byte pieces (1 KB of LDS) per wave.  The source offset of a pixel
// relative to the tile origin is the same for every tile, so it is decoded ONCE per kernel into an LDS table
// {byte offset, row | column << 16}; per tile a lane's piece costs one table read, an add, two range checks
// (out-of-range lanes get an offset beyond the buffer: the hardware writes zeros) and the DMA issue.
template <int LP, int XSUB>
struct WgStage {
    static constexpr int CW = 8 << LP;            // channels per sub-image
    const WgParams& p;
    char* smem;
    unsigned tab;                                 // LDS address of the pixel table [256 + npixp] x {rel, crd}
    int lane, wave, lanepix, q8;
    int ngp, nx, npieces, ninstr;                 // pieces of the G image / of one X sub-image / total; DMA instructions
    int glim, xlim;
    unsigned gq, xq;
    int TH, TW, xspan_h, xspan_w;
    unsigned gimg_bytes, ximg_bytes;
    bool reflect;
    int dtoff;                                    // temporal taps: frame offset of this workgroup's X columns
    int xrow;                                     // tap-row group: input-row offset of this workgroup's first tap row

    __device__ __forceinline__ WgStage(const WgParams& p_, char* smem_, int tid, int gsub, int xsub, int m0, int n0, int tg = 0)
        : p(p_), smem(smem_) {
        xrow = tg * p_.kh * p_.dh;
        lane = tid & 63;
        wave = __builtin_amdgcn_readfirstlane(tid >> 6);
        lanepix = lane >> LP;
        q8 = (lane & ((1 << LP) - 1)) * 8;        // this lane's channel run inside a pixel's sub-image row
        ngp = gsub * (256 << LP);
        nx = p.npixp << LP;
        npieces = ngp + xsub * nx;
        ninstr = (npieces + 63) >> 6;
        tab = (unsigned)(uintptr_t)smem + (unsigned)p.bufbytes * (p.dbuf ? 2 : 1);
        glim = min(p.M - m0, p.g_cs - p.g_off - m0 - 7);
        // temporal taps: column n = dt * tcin + c is channel c of the frame dt - tpad away (an n-group never straddles two
        // frames: tcin is a multiple of the widest n-group)
        const int nc = p.tT ? n0 % p.tcin : n0;
        dtoff = p.tT ? n0 / p.tcin - p.tpad : 0;
        xlim = p.tT ? min(p.tcin - nc, p.x_cs - p.x_off - nc - 7) : min(p.N - n0, p.x_cs - p.x_off - n0 - 7);
        gq = (unsigned)(p.g_off + m0 + q8) * 2;
        xq = (unsigned)(p.x_off + nc + q8) * 2;
        TH = 1 << p.logTH; TW = 1 << p.logTW;
        xspan_h = (p.PH - 1) * p.dh; xspan_w = (p.NC - 1) * p.stride + (p.PW - 1) * p.dw;
        gimg_bytes = (unsigned)p.Hg * p.Wg * p.g_cs * 2; ximg_bytes = (unsigned)p.Hx * p.Wx * p.x_cs * 2;
        reflect = p.pad_mode == SOS_PAD_REFLECT;
        // ---- tile-invariant pixel table
        for (int e = tid; e < 256 + p.npixp; e += WG_THREADS) {
            unsigned r = 0u, c = 0x7fff7fffu;                     // invalid: always out of range
            if (e < 256) {
                int j, i, cls;
                wg_decode(e, p.logTH, p.logTW, p.kord, cls, i, j);
                const int hrel = i * p.dh, wrel = cls + j * p.dw;
                r = (unsigned)((hrel * p.Wg + wrel) * p.g_cs * 2);
                c = (unsigned)hrel | ((unsigned)wrel << 16);
            } else if (e - 256 < p.npix) {
                const int pix = e - 256;
                const int rr = pix / p.PW, cc = pix - rr * p.PW;
                const int cls = rr / p.PH, row = rr - cls * p.PH;
                const int hrel = row * p.dh, wrel = cls * p.stride + cc * p.dw;
                r = (unsigned)((hrel * p.Wx + wrel) * p.x_cs * 2);
                c = (unsigned)hrel | ((unsigned)wrel << 16);
            }
            *(uint2*)(smem + (tab - (unsigned)(uintptr_t)smem) + e * 8) = make_uint2(r, c);
        }
    }
    // Position of a pixel tile in the walk (tile column fastest, then tile row, class group, row class, image).  The kernels
    // decompose their FIRST step with divisions and then advance by carries: origin_of(step) cost four scalar divisions
    // (~150 SALU) per tile and wave, right behind the tile's barrier (SQ counters: 4.5 SALU per MFMA in wgrad_kernel).
    struct Walk { int tj, ti, gw, rh, t; };
    __device__ __forceinline__ Walk walk_init(int step) const {
        Walk w;
        int t = step;
        w.tj = t % p.tiles_w; t /= p.tiles_w;
        w.ti = t % p.tiles_h; t /= p.tiles_h;
        w.gw = t % p.ngw; t /= p.ngw;
        w.rh = t % p.dh; t /= p.dh;
        w.t = t;
        return w;
    }
    __device__ __forceinline__ void walk_next(Walk& w) const {
        if (++w.tj < p.tiles_w) return;
        w.tj = 0;
        if (++w.ti < p.tiles_h) return;
        w.ti = 0;
        if (++w.gw < p.ngw) return;
        w.gw = 0;
        if (++w.rh < p.dh) return;
        w.rh = 0;
        ++w.t;
    }
    __device__ __forceinline__ WgTile origin_of(int step) const { return origin_at(walk_init(step)); }
    __device__ __forceinline__ WgTile origin_at(const Walk& wk) const {
        const int tj = wk.tj, ti = wk.ti, gw = wk.gw, rh = wk.rh, t = wk.t;
        WgTile o;
        o.gh0 = rh + ti * TH * p.dh; o.gw0 = gw * p.NC + tj * TW * p.dw;
        o.xh0 = o.gh0 * p.stride - p.pad_t + xrow; o.xw0 = o.gw0 * p.stride - p.pad_l;
        o.gorg = (unsigned)((o.gh0 * p.Wg + o.gw0) * p.g_cs * 2);
        o.xorg = (unsigned)((o.xh0 * p.Wx + o.xw0) * p.x_cs * 2);      // may be "negative": wraps, valid lanes land in range
        o.xfast = o.xh0 >= 0 && o.xw0 >= 0 && o.xh0 + xspan_h < p.Hx && o.xw0 + xspan_w < p.Wx;
        o.xslow = reflect && !o.xfast;
        o.gfast = o.gh0 + (TH - 1) * p.dh < p.Hg && o.gw0 + (p.NC - 1) + (TW - 1) * p.dw < p.Wg;
        o.rg = __builtin_amdgcn_make_buffer_rsrc((void*)(p.g + (size_t)t * p.Hg * p.Wg * p.g_cs), 0, gimg_bytes, 0x00020000);
        int xi = t;
        unsigned xbytes = ximg_bytes;
        if (p.tT) {                      // the frame dtoff away, inside the same clip; outside: an empty resource (all zeros)
            const int fr = t % p.tT + dtoff;
            if (fr < 0 || fr >= p.tT) xbytes = 0u; else xi = t + dtoff;
        }
        o.rx = __builtin_amdgcn_make_buffer_rsrc((void*)(p.x + (size_t)xi * p.Hx * p.Wx * p.x_cs), 0, xbytes, 0x00020000);
        return o;
    }
    // sub-image index of X piece l (wave-uniform)
    __device__ __forceinline__ int xsub_of(int l) const {
        if constexpr (XSUB == 1) return 0;
        else if constexpr (XSUB == 2) return l >= nx;
        else if constexpr (XSUB == 3) return (l >= nx) + (l >= 2 * nx);
        else return (l >= nx) + (l >= 2 * nx) + (l >= 3 * nx);
    }
    // LDS address of this lane's pixel-table entry for DMA instruction i (pieces [64 i, 64 i + 64), i wave-uniform)
    __device__ __forceinline__ unsigned entry_addr(int i) const {
        const int L0 = i * 64;
        int e = ((L0 >> LP) & 255) + lanepix;
        if (L0 >= ngp) {
            const int l = L0 - ngp;
            e = 256 + ((l - xsub_of(l) * nx) >> LP) + lanepix;
        }
        return tab + (unsigned)e * 8;
    }
    // issue DMA instruction i of tile `o` into LDS buffer `buf`; ent = that lane's table entry
    __device__ __forceinline__ void issue(int i, const uint2 ent, const WgTile& o, int buf) const {
        const int L0 = i * 64;                                     // wave-uniform
        if (i >= ninstr) return;
        lds_ptr_t dst = (lds_ptr_t)(smem + buf * p.bufbytes + L0 * 16);
        if (L0 < ngp) {
            const int a = L0 >> (8 + LP);    // channels past M inside a stored 8-run are the producer's zero padding
            const bool ok = a * CW + q8 < glim && (unsigned)(o.gh0 + (int)(ent.y & 0xffffu)) < (unsigned)p.Hg &&
                            (unsigned)(o.gw0 + (int)(ent.y >> 16)) < (unsigned)p.Wg;
            const unsigned voff = ok && !WDBG(8) ? ent.x + gq + (o.gorg + (unsigned)a * (2u * CW)) : (WDBG(16) ? 0u : 0xffffffffu);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(o.rg, dst, 16, voff, 0, 0, 0);
        } else {
            const int nt = xsub_of(L0 - ngp);
            const int h = o.xh0 + (int)(ent.y & 0xffffu), w = o.xw0 + (int)(ent.y >> 16);
            bool ok = nt * CW + q8 < xlim && ent.y != 0x7fff7fffu;
            unsigned voff = ent.x + xq + (o.xorg + (unsigned)nt * (2u * CW));
            if (o.xslow) {     // ReflectionPad2d border tile: mirror the coordinates
                const int hr = reflect_index(h, p.Hx), wr = reflect_index(w, p.Wx);
                voff = (unsigned)((hr * p.Wx + wr) * p.x_cs * 2) + xq + (unsigned)nt * (2u * CW);
                ok = ok && (unsigned)hr < (unsigned)p.Hx && (unsigned)wr < (unsigned)p.Wx;
            } else {           // zero padding (and interior tiles): plain range check
                ok = ok && (unsigned)h < (unsigned)p.Hx && (unsigned)w < (unsigned)p.Wx;
            }
            voff = ok && !WDBG(8) ? voff : (WDBG(16) ? 0u : 0xffffffffu);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(o.rx, dst, 16, voff, 0, 0, 0);
        }
    }
    __device__ __forceinline__ void issue_all(const WgTile& o, int buf) const {
        for (int i = wave; i < ninstr; i += WG_WAVES) {
            uint2 ent = lds_read64(entry_addr(i));
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(ent));
            issue(i, ent, o, buf);
        }
    }
};
#endif

// Workgroup = MT m-tiles (32 rows of dW each) x NTB n-tiles x all taps, 8 waves.  Wave w owns the
// (tap, n-tile) pairs {w, w+8, w+16, w+24} for ALL MT m-tiles, so every transposed X fragment feeds
// MT MFMAs and every G fragment WG_PAIRS of them.  Operands: WgStage<2> (32-channel sub-images, 64-byte
// pixel pitch: conflict free for the transpose reads); with double buffering the DMA instructions of the
// next tile are issued one per k-step by the "light" waves while the MFMAs of the current tile run.
// BAL (25 taps, NTB = 1): 25 pairs on 8 waves would give wave 0 a fourth pair (12 accumulator tiles, its SIMD 21 MFMAs
// per k-step against 18 on the others).  Instead every wave owns three pairs and the 25th tap is split by m-tile
// over the waves 0..MT-1 (three different SIMDs): 10 accumulator tiles, at most 19 MFMAs per SIMD and k-step.
template <int MT, int NTB, bool BAL = false>
__global__ __launch_bounds__(WG_THREADS) void wgrad_kernel(WgParams p) {
#if __HIP_DEVICE_COMPILE__
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int gbytes = 256 * 64 * MT;
    const int ximg = p.npixp * 64;                                // bytes of one X sub-image
    const unsigned sbase = (unsigned)(uintptr_t)smem;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // 1-D grid, XCD aware: workgroups go to the 8 XCDs round robin by their linear id, and every XCD has its own L2.
    // The (m-group, n-group) workgroups of ONE pixel split read the same G / X tiles at the same time, so they are
    // given ids that are congruent mod 8: the tile is fetched from HBM once per XCD instead of once per workgroup
    // (96->96: G was read three times, 2.55 GB -> 1.43 GB per launch).
    // (used when the splits are a multiple of 8 and an XCD's 32 CUs hold all its workgroups at once: p.xcdmap)
    const int nyz = p.ny * p.nz;
    const int lin = blockIdx.x;
    int split, yz;
    if (p.xcdmap) {
        const int xcd = lin & 7, slot = lin >> 3;
        yz = slot % nyz;
        split = (slot / nyz) * 8 + xcd;
    } else {
        split = lin % p.ksplit;
        yz = lin / p.ksplit;
    }
    const int zz = yz % p.nz, tg = zz % p.ntg;                     // p.nz = n-groups x tap-row groups
    const int m0 = (yz / p.nz) * (MT * 32), n0 = (zz / p.ntg) * (NTB * 32);
    const int taps = p.kh * p.kw;                                  // of this workgroup's tap-row group
    const int npairs = taps * NTB;
    constexpr int NP = BAL ? 3 : WG_PAIRS;                      // full (tap, n-tile) pairs per wave
    const bool hasx = BAL && wave < MT;                          // this wave also owns m-tile `wave` of the last tap
    const int g4 = lane >> 4, s16 = lane & 15;
    const int chan_off = (16 * (g4 & 1) + 4 * (s16 & 3)) * 2;    // byte offset of this lane's 4-channel run
    const int krow = 8 * (g4 >> 1) + (s16 >> 2);                 // pixel (k) inside a 16-pixel k-step; +4 for 2nd read

    const WgStage<2, NTB> st(p, smem, tid, MT, NTB, m0, n0, tg);
    const int ninstr = st.ninstr;

    f32x16 acc[MT][NP], accx;
#pragma unroll
    for (int a = 0; a < MT; ++a)
#pragma unroll
        for (int t = 0; t < NP; ++t)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[a][t][e] = 0.f;
#pragma unroll
    for (int e = 0; e < 16; ++e) accx[e] = 0.f;

    // per-wave offsets of its pairs inside the X image (tap shift + n-tile sub-image)
    unsigned toff[NP];
#pragma unroll
    for (int u = 0; u < NP; ++u) {
        const int pr = min(wave + WG_WAVES * u, npairs - 1);
        const int tap = pr / NTB, nt = pr - tap * NTB;
        const int ta = tap / p.kw, tb = tap - ta * p.kw;
        toff[u] = (unsigned)((ta * p.PW + tb) * 64 + nt * ximg);
    }
    const unsigned toffx = (unsigned)((((taps - 1) / p.kw) * p.PW + (taps - 1) % p.kw) * 64);   // BAL: the last tap
    // While a tile is multiplied, the next one is fetched by the "light" waves: those that own one (tap, n-tile)
    // pair fewer than the others (25 taps on 8 waves: wave 0 has 4 pairs, waves 1..7 have 3).
    int heavy = BAL ? 0 : (npairs & (WG_WAVES - 1));
    if ((WG_WAVES - heavy) * 16 < ninstr) heavy = 0;
    const int nlight = WG_WAVES - heavy, lw = wave - heavy;       // lw < 0: this wave issues no DMA in the k-loop

    // patch pixel of tile pixel k (k-order = (class, row, column) of the 256-pixel tile)
    auto pp_of = [&](int k) {
        int c, i, j;
        wg_decode(k, p.logTH, p.logTW, p.kord, c, i, j);
        return (c * p.PH + i * p.stride) * p.PW + j * p.stride;
    };
    const unsigned glane = (unsigned)(krow * 64 + chan_off);
    const unsigned xlane0 = (unsigned)(pp_of(krow) * 64 + chan_off), xlane1 = (unsigned)(pp_of(krow + 4) * 64 + chan_off);

    const unsigned ppk = (unsigned)pp_of((lane & 15) * 16) * 64u;      // lane ks: X-image byte offset of k-step ks
    // lane ks: image-relative coordinates of the FIRST pixel of k-step ks.  Validity falls monotonically along rows, columns
    // and classes and a k-step is an aligned block of the pixel index, so a k-step holds a pixel inside the image exactly
    // when its first pixel is inside: one compare + ballot per tile gives the mask of the k-steps worth multiplying
    // (the others would add G = 0 rows: skipping them changes no bit of the result).
    int kfh, kfw;
    {
        int c, i, j;
        wg_decode((lane & 15) * 16, p.logTH, p.logTW, p.kord, c, i, j);
        kfh = i * p.dh; kfw = c + j * p.dw;
    }

    const int step0 = split * p.steps_per_split;
    const int step1 = min(step0 + p.steps_per_split, p.nsteps);
    int cur = 0;
    __syncthreads();                               // pixel table complete
    int cgh0 = 0, cgw0 = 0;                        // origin of the tile being multiplied
    auto wk = st.walk_init(step0 < step1 ? step0 : 0);
    if (step0 < step1) {
        const WgTile o0 = st.origin_at(wk);
        cgh0 = o0.gh0; cgw0 = o0.gw0;
        st.issue_all(o0, 0);
    }
    for (int step = step0; step < step1; ++step) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();                           // tile `step` has landed in buffer `cur`; buffer cur^1 is free
        const bool more = step + 1 < step1 && !WDBG(1);
        if (more) st.walk_next(wk);
        const WgTile onext = st.origin_at(wk);
        const bool prefetch = p.dbuf && more;
        const unsigned gb = sbase + cur * p.bufbytes, xb = gb + gbytes;
        unsigned kmask = (unsigned)__builtin_amdgcn_ballot_w64(cgh0 + kfh < p.Hg && cgw0 + kfw < p.Wg) & 0xffffu;   // wave-uniform
        if (WDBG(32)) kmask = 0xffffu;
        cgh0 = onext.gh0; cgw0 = onext.gw0;
        if constexpr (BAL && SOS_WGRAD_PIPE) {
            // ---- software pipeline over the k-steps (round 3).  The counted waits of the loop below do not survive hipcc's
            // scheduler (the MFMAs sink below them: every k-step waited for ALL of its reads before its first MFMA, and the
            // in-order MFMA issue of a k-step -- 9..10 x 32 cycles -- then kept the wave from issuing the next reads: LDS
            // latency and MFMA issue were serial, the MFMA pipe 57 % busy).  Here the fragments are double buffered in
            // registers: the reads of k-step ks + 1 are issued BEFORE the MFMAs of ks, whose operands landed during the
            // MFMAs of ks - 1; one wait per k-step, pinned with scheduling barriers.
            struct Frag { u32x4 av[MT], bv[NP]; };
            Frag F0, F1;
            u32x4 avx, bvx;               // the wave's part of the 25th tap: read at the top of ITS k-step, used last (8 registers, not 16)
            auto rd2 = [&](const unsigned a0addr, const unsigned a1addr) {
                const uint2 lo = lds_tr(a0addr), hi = lds_tr(a1addr);
                return u32x4{lo.x, lo.y, hi.x, hi.y};
            };
            auto issue_reads = [&](const int ks, Frag& F) {
                const unsigned ga = (gb + (unsigned)ks * 1024u) + glane;
                const unsigned xk = xb + (unsigned)__builtin_amdgcn_readlane((int)ppk, ks);
                { const uint2 lo = lds_tr(ga), hi = lds_tr_off<256>(ga); F.av[0] = u32x4{lo.x, lo.y, hi.x, hi.y}; }
                if constexpr (MT >= 2) { const uint2 lo = lds_tr_off<16384>(ga), hi = lds_tr_off<16384 + 256>(ga); F.av[1] = u32x4{lo.x, lo.y, hi.x, hi.y}; }
                if constexpr (MT >= 3) { const uint2 lo = lds_tr_off<32768>(ga), hi = lds_tr_off<32768 + 256>(ga); F.av[2] = u32x4{lo.x, lo.y, hi.x, hi.y}; }
#pragma unroll
                for (int u = 0; u < NP; ++u) F.bv[u] = rd2(xlane0 + (xk + toff[u]), xlane1 + (xk + toff[u]));
            };
            // hx / last are compile-time tags: the k-loop below exists twice (waves with / without a part of the 25th tap) and
            // only the last k-step is peeled (two full copies of the loop, one per wave kind, made hipcc spill 483 registers)
            auto kstep = [&](auto last, const int ks, Frag& C, Frag& N) {
                constexpr bool LAST = decltype(last)::value;
                uint2 ent = make_uint2(0u, 0u);
                const int di = ks * nlight + lw;                      // this wave's DMA instruction in this k-step
                const bool dma = prefetch && lw >= 0 && di < ninstr;
                if (dma) ent = lds_read64(st.entry_addr(di));          // ahead of the next reads: it returns before them
                if (hasx) {
                    const unsigned gax = (gb + (unsigned)ks * 1024u) + glane + (unsigned)wave * 16384u;
                    const unsigned xk = xb + (unsigned)__builtin_amdgcn_readlane((int)ppk, ks);
                    const uint2 lo = lds_tr(gax), hi = lds_tr_off<256>(gax);
                    avx = u32x4{lo.x, lo.y, hi.x, hi.y};
                    bvx = rd2(xlane0 + (xk + toffx), xlane1 + (xk + toffx));
                }
                if constexpr (!LAST) issue_reads(ks + 1, N);
                __builtin_amdgcn_sched_barrier(0);
                // LDS returns in order: when no more than the reads issued in THIS k-step are outstanding, `ent` and C (issued
                // a k-step ago) have landed.  lgkmcnt is a 4-bit counter: 15 = "at most 15 outstanding".
                constexpr int NOUT = LAST ? 0 : 2 * (MT + NP);
                if (hasx) asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(ent) : "n"(NOUT + 4 > 15 ? 15 : NOUT + 4));
                else asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(ent) : "n"(NOUT));
#pragma unroll
                for (int a = 0; a < MT; ++a) asm volatile("" : "+v"(C.av[a]));
#pragma unroll
                for (int u = 0; u < NP; ++u) asm volatile("" : "+v"(C.bv[u]));
                __builtin_amdgcn_sched_barrier(0);
                if (dma) st.issue(di, ent, onext, cur ^ 1);
#pragma unroll
                for (int u = 0; u < NP; ++u) {
                    const bf16x8 bfr = __builtin_bit_cast(bf16x8, C.bv[u]);
#pragma unroll
                    for (int a = 0; a < MT; ++a)
                        acc[a][u] = SOS_MFMA_32x32x16(__builtin_bit_cast(bf16x8, C.av[a]), bfr, acc[a][u], 0, 0, 0);
                }
                if (hasx) {          // its four reads were the first of this k-step: landed once only the next k-step's are outstanding
                    __builtin_amdgcn_sched_barrier(0);
                    asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(avx), "+v"(bvx) : "n"(LAST ? 0 : 2 * (MT + NP)));
                    accx = SOS_MFMA_32x32x16(__builtin_bit_cast(bf16x8, avx), __builtin_bit_cast(bf16x8, bvx), accx, 0, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
            };
            {
                constexpr std::false_type more_t{};
                constexpr std::true_type last_t{};
                issue_reads(0, F0);
#pragma unroll 1
                for (int ks = 0; ks < 14; ks += 2) {
                    kstep(more_t, ks, F0, F1);
                    kstep(more_t, ks + 1, F1, F0);
                }
                kstep(more_t, 14, F0, F1);
                kstep(last_t, 15, F1, F0);
            }
        } else {
        int it = 0;                                 // k-steps done: the DMA slots of the next tile are counted in these
#pragma unroll 1
        for (; kmask; ++it) {
            const int ks = __builtin_ctz(kmask);
            kmask &= kmask - 1;
            uint2 ent = make_uint2(0u, 0u);
            const int di = it * nlight + lw;                      // this wave's DMA instruction in this k-step
            const bool dma = prefetch && lw >= 0 && di < ninstr;
            if (dma) ent = lds_read64(st.entry_addr(di));
            // k = 16 ks + krow: the bits of 16 ks and of krow (< 16) are disjoint, so the patch pixel of k is
            // pp(16 ks) + pp(krow) -- a scalar term per k-step plus two per-lane constants.
            const unsigned ga = (gb + (unsigned)ks * 1024u) + glane;
            const unsigned xk = xb + (unsigned)__builtin_amdgcn_readlane((int)ppk, ks);   // wave-uniform (tile invariant: lane ks of ppk)
            // LDS reads in the order the MFMAs consume them (LDS returns in order): every MFMA group waits only
            // for its own fragments, so the first MFMAs start while the later fragments are still in flight.
            // The waits name their registers so that every consumer is ordered behind them.
            // (the two 64-bit halves of an operand are assembled into one 128-bit value first: no register moves)
            u32x4 av[MT], bv[NP], avx, bvx;
            auto rd2 = [&](const unsigned a0addr, const unsigned a1addr) {
                const uint2 lo = lds_tr(a0addr), hi = lds_tr(a1addr);
                return u32x4{lo.x, lo.y, hi.x, hi.y};
            };
            if constexpr (BAL) {                   // issued first: LDS returns in order, so the counted waits below do not change
                if (hasx) {
                    const unsigned gax = ga + (unsigned)wave * 16384u;
                    const uint2 lo = lds_tr(gax), hi = lds_tr_off<256>(gax);
                    avx = u32x4{lo.x, lo.y, hi.x, hi.y};
                    bvx = rd2(xlane0 + (xk + toffx), xlane1 + (xk + toffx));
                }
            }
            { const uint2 lo = lds_tr(ga), hi = lds_tr_off<256>(ga); av[0] = u32x4{lo.x, lo.y, hi.x, hi.y}; }
            bv[0] = rd2(xlane0 + (xk + toff[0]), xlane1 + (xk + toff[0]));
            if constexpr (MT >= 2) { const uint2 lo = lds_tr_off<16384>(ga), hi = lds_tr_off<16384 + 256>(ga); av[1] = u32x4{lo.x, lo.y, hi.x, hi.y}; }
            if constexpr (MT >= 3) { const uint2 lo = lds_tr_off<32768>(ga), hi = lds_tr_off<32768 + 256>(ga); av[2] = u32x4{lo.x, lo.y, hi.x, hi.y}; }
#pragma unroll
            for (int u = 1; u < NP; ++u) bv[u] = rd2(xlane0 + (xk + toff[u]), xlane1 + (xk + toff[u]));
            constexpr int REST = 2 * (NP - 1);              // reads behind pair 0 / m-tile a
            asm volatile("s_waitcnt lgkmcnt(%3)" : "+v"(ent), "+v"(av[0]), "+v"(bv[0]) : "n"(REST + 2 * (MT - 1)));
            if constexpr (BAL) { if (hasx) asm volatile("" : "+v"(avx), "+v"(bvx)); }   // ordered behind the wait above
            if (dma) st.issue(di, ent, onext, cur ^ 1);
            const bool on0 = BAL || wave < npairs;      // BAL: 25 pairs, every wave owns pair 0 (no per-MFMA branches)
            {
                const bf16x8 bfr = __builtin_bit_cast(bf16x8, bv[0]);
                if (on0) acc[0][0] = SOS_MFMA_32x32x16(__builtin_bit_cast(bf16x8, av[0]), bfr, acc[0][0], 0, 0, 0);
                if constexpr (MT >= 2) {
                    asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(av[1]) : "n"(REST + 2 * (MT - 2)));
                    if (on0) acc[1][0] = SOS_MFMA_32x32x16(__builtin_bit_cast(bf16x8, av[1]), bfr, acc[1][0], 0, 0, 0);
                }
                if constexpr (MT >= 3) {
                    asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(av[2]) : "n"(REST));
                    if (on0) acc[2][0] = SOS_MFMA_32x32x16(__builtin_bit_cast(bf16x8, av[2]), bfr, acc[2][0], 0, 0, 0);
                }
            }
#pragma unroll
            for (int u = 1; u < NP; ++u) {
                if constexpr (NP == 4) {
                    if (u == 1) asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(bv[1]));
                    if (u == 2) asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(bv[2]));
                    if (u == 3) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(bv[3]));
                } else {
                    if (u == 1) asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(bv[1]));
                    if (u == 2) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(bv[2]));
                }
                if (BAL || wave + WG_WAVES * u < npairs) {
                    const bf16x8 bfr = __builtin_bit_cast(bf16x8, bv[u]);
#pragma unroll
                    for (int a = 0; a < MT; ++a)
                        acc[a][u] = SOS_MFMA_32x32x16(__builtin_bit_cast(bf16x8, av[a]), bfr, acc[a][u], 0, 0, 0);
                }
            }
            if constexpr (BAL) {
                if (hasx) accx = SOS_MFMA_32x32x16(__builtin_bit_cast(bf16x8, avx), __builtin_bit_cast(bf16x8, bvx), accx, 0, 0, 0);
            }
        }
        // border tile with skipped k-steps: the DMA slots those k-steps would have carried
        if (prefetch && lw >= 0) {
            for (int di = it * nlight + lw; di < ninstr; di += nlight) {
                uint2 ent = lds_read64(st.entry_addr(di));
                asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(ent));
                st.issue(di, ent, onext, cur ^ 1);
            }
        }
        }
        if (p.dbuf) {
            cur ^= 1;
        } else if (more) {
            __syncthreads();                       // every wave is done reading the single buffer
            st.issue_all(onext, 0);
        }
    }
    // ---- write this split's partial tiles: D[row = m][col = n], row = (reg&3)+8*(reg>>2)+4*(lane>>5), col = lane&31
    float* out = p.partial + ((size_t)split * p.taps_all + (size_t)tg * taps) * p.Mp * p.Np;
#pragma unroll
    for (int u = 0; u < NP; ++u) {
        const int pr = wave + WG_WAVES * u;
        if (pr >= npairs) continue;
        const int tap = pr / NTB, nt = pr - tap * NTB;
        const int n = n0 + nt * 32 + (lane & 31);
        if (n >= p.Np || tg * taps + tap >= p.taps_all) continue;      // (phantom tap of the last tap-row group)
#pragma unroll
        for (int a = 0; a < MT; ++a) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + a * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                if (m < p.Mp) out[((size_t)tap * p.Mp + m) * p.Np + n] = acc[a][u][r];
            }
        }
    }
    if constexpr (BAL) {
        const int n = n0 + (lane & 31);
        if (hasx && n < p.Np) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                if (m < p.Mp) out[((size_t)(taps - 1) * p.Mp + m) * p.Np + n] = accx[r];
            }
        }
    }
#endif
}


// ---- small channel counts (the 48-channel context layers): v_mfma_f32_16x16x32_bf16.
// With 32x32 tiles a 48 x 48 gradient pays for 64 x 64; here one workgroup owns ALL of dW: M16 x N16 tiles of
// 16 x 16 per tap.  Wave w owns the taps {w, w+8, ...} (FT = taps / 8 of them) for every (m, n) tile, and the one
// remainder tap (25 = 3*8 + 1, 9 = 8 + 1) is split into M16 single-m-tile units for the waves 0..M16-1.
// Operands: WgStage<1> (16-channel sub-images, 32-byte pixel pitch).  A k-step is 32 tile pixels; MFMA k-slot
// (lane group g, element e) is tile pixel 4 g + e (e < 4) or 16 + 4 g + (e - 4): the first transpose read of a
// wave then covers 16 consecutive pixels x 32 B = 128 consecutive dwords (conflict free), the second the next 16.
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int M16, int N16, int FT>
__global__ __launch_bounds__(WG_THREADS) void wgrad16_kernel(WgParams p) {
#if __HIP_DEVICE_COMPILE__
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int gbytes = 256 * 32 * M16;
    const int ximg = p.npixp * 32;                                // bytes of one X sub-image
    const unsigned sbase = (unsigned)(uintptr_t)smem;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int split = blockIdx.x;
    const int taps = p.kh * p.kw;
    const int rem = taps - FT * WG_WAVES;                         // remainder taps (launcher: rem * M16 <= 8)
    const bool has_e = wave < rem * M16;
    const int tap_e = min(FT * WG_WAVES + wave / M16, taps - 1), mt_e = wave % M16;
    const int g4 = lane >> 4, s16 = lane & 15;
    const int krow = 4 * g4 + (s16 >> 2);                         // tile pixel of the first read inside the k-step; +16 second
    const int colb = (s16 & 3) * 8;

    const WgStage<1, N16> st(p, smem, tid, M16, N16, 0, 0);
    const int ninstr = st.ninstr;

    f32x4 acc[FT][M16][N16], acce[N16];
#pragma unroll
    for (int f = 0; f < FT; ++f)
#pragma unroll
        for (int a = 0; a < M16; ++a)
#pragma unroll
            for (int n = 0; n < N16; ++n) acc[f][a][n] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int n = 0; n < N16; ++n) acce[n] = f32x4{0.f, 0.f, 0.f, 0.f};

    unsigned toff[FT];
#pragma unroll
    for (int f = 0; f < FT; ++f) {
        const int tap = wave + WG_WAVES * f;
        const int ta = tap / p.kw, tb = tap - ta * p.kw;
        toff[f] = (unsigned)((ta * p.PW + tb) * 32);
    }
    const unsigned toffe = (unsigned)(((tap_e / p.kw) * p.PW + (tap_e % p.kw)) * 32);
    auto pp_of = [&](int k) {
        int c, i, j;
        wg_decode(k, p.logTH, p.logTW, p.kord, c, i, j);
        return (c * p.PH + i * p.stride) * p.PW + j * p.stride;
    };
    const unsigned glane = (unsigned)(krow * 32 + colb);
    const unsigned xlane0 = (unsigned)(pp_of(krow) * 32 + colb), xlane1 = (unsigned)(pp_of(krow + 16) * 32 + colb);
    const unsigned ppk = (unsigned)pp_of((lane & 7) * 32) * 32u;        // lane ks: X-image byte offset of k-step ks
    int kfh, kfw;                                                       // lane ks: first pixel of k-step ks (see wgrad_kernel)
    {
        int c, i, j;
        wg_decode((lane & 7) * 32, p.logTH, p.logTW, p.kord, c, i, j);
        kfh = i * p.dh; kfw = c + j * p.dw;
    }

    const int step0 = split * p.steps_per_split;
    const int step1 = min(step0 + p.steps_per_split, p.nsteps);
    int cur = 0;
    __syncthreads();                               // pixel table complete
    int cgh0 = 0, cgw0 = 0;                        // origin of the tile being multiplied
    auto wk = st.walk_init(step0 < step1 ? step0 : 0);
    if (step0 < step1) {
        const WgTile o0 = st.origin_at(wk);
        cgh0 = o0.gh0; cgw0 = o0.gw0;
        st.issue_all(o0, 0);
    }
    for (int step = step0; step < step1; ++step) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();                           // tile `step` has landed in buffer `cur`; buffer cur^1 is free
        const bool more = step + 1 < step1 && !WDBG(1);
        if (more) st.walk_next(wk);
        const WgTile onext = st.origin_at(wk);
        const bool prefetch = p.dbuf && more;
        const unsigned gb = sbase + cur * p.bufbytes, xb = gb + gbytes;
        unsigned kmask = (unsigned)__builtin_amdgcn_ballot_w64(cgh0 + kfh < p.Hg && cgw0 + kfw < p.Wg) & 0xffu;   // wave-uniform
        if (WDBG(32)) kmask = 0xffu;
        cgh0 = onext.gh0; cgw0 = onext.gw0;
        int it = 0;
#pragma unroll 1
        for (; kmask; ++it) {
            const int ks = __builtin_ctz(kmask);
            kmask &= kmask - 1;
            if (prefetch) {                        // the next tile: 8 waves x 8 k-steps DMA slots (more rounds if the patch is big)
                for (int di = it * WG_WAVES + wave; di < ninstr; di += 8 * WG_WAVES) {
                    uint2 ent = lds_read64(st.entry_addr(di));
                    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(ent));
                    st.issue(di, ent, onext, cur ^ 1);
                }
            }
            const unsigned ga = (gb + (unsigned)ks * 1024u) + glane;
            const unsigned xk = xb + (unsigned)__builtin_amdgcn_readlane((int)ppk, ks);   // wave-uniform (lane ks of ppk)
            // reads in consumption order (LDS returns in order): remainder unit, G, then tap by tap.  The two
            // 64-bit halves of an operand are assembled into ONE 128-bit value before the counted wait names it,
            // so the register coalescer lets each transpose read write its half in place (no v_mov).
            auto rd2 = [&](const unsigned a0addr, const unsigned a1addr) {
                const uint2 lo = lds_tr(a0addr), hi = lds_tr(a1addr);
                return u32x4{lo.x, lo.y, hi.x, hi.y};
            };
            u32x4 ae, be[N16];                                   // only defined (and used) when has_e
            if (has_e) {
                ae = rd2(ga + (unsigned)mt_e * 8192u, ga + (unsigned)mt_e * 8192u + 512u);
#pragma unroll
                for (int n = 0; n < N16; ++n)
                    be[n] = rd2(xlane0 + (xk + toffe + (unsigned)(n * ximg)), xlane1 + (xk + toffe + (unsigned)(n * ximg)));
            }
            u32x4 av[M16], bv[FT][N16];
            {
                const uint2 lo = lds_tr(ga), hi = lds_tr_off<512>(ga);
                av[0] = u32x4{lo.x, lo.y, hi.x, hi.y};
            }
            if constexpr (M16 >= 2) { const uint2 lo = lds_tr_off<8192>(ga), hi = lds_tr_off<8192 + 512>(ga); av[1] = u32x4{lo.x, lo.y, hi.x, hi.y}; }
            if constexpr (M16 >= 3) { const uint2 lo = lds_tr_off<16384>(ga), hi = lds_tr_off<16384 + 512>(ga); av[2] = u32x4{lo.x, lo.y, hi.x, hi.y}; }
#pragma unroll
            for (int f = 0; f < FT; ++f)
#pragma unroll
                for (int n = 0; n < N16; ++n)
                    bv[f][n] = rd2(xlane0 + (xk + toff[f] + (unsigned)(n * ximg)), xlane1 + (xk + toff[f] + (unsigned)(n * ximg)));
#pragma unroll
            for (int f = 0; f < FT; ++f) {
                // everything up to tap f has landed when 2 N16 (FT - 1 - f) reads are still outstanding
                if constexpr (N16 == 3 && M16 == 3) {
                    if (f == 0)
                        asm volatile("s_waitcnt lgkmcnt(%6)" : "+v"(av[0]), "+v"(av[1]), "+v"(av[2]), "+v"(bv[0][0]), "+v"(bv[0][1]), "+v"(bv[0][2])
                                     : "n"(6 * (FT - 1)));
                    else
                        asm volatile("s_waitcnt lgkmcnt(%3)" : "+v"(bv[f][0]), "+v"(bv[f][1]), "+v"(bv[f][2]) : "n"(6 * (FT - 1 - f)));
                }
#pragma unroll
                for (int n = 0; n < N16; ++n) {
#pragma unroll
                    for (int a = 0; a < M16; ++a)
                        acc[f][a][n] = SOS_MFMA_16x16x32(__builtin_bit_cast(bf16x8, av[a]),
                                                                               __builtin_bit_cast(bf16x8, bv[f][n]), acc[f][a][n], 0, 0, 0);
                }
            }
            if (has_e) {
                if constexpr (N16 == 3) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(ae), "+v"(be[0]), "+v"(be[1]), "+v"(be[2]));
#pragma unroll
                for (int n = 0; n < N16; ++n)
                    acce[n] = SOS_MFMA_16x16x32(__builtin_bit_cast(bf16x8, ae), __builtin_bit_cast(bf16x8, be[n]),
                                                                      acce[n], 0, 0, 0);
            }
        }
        if (prefetch) {                            // the DMA slots of skipped k-steps
            for (; it < 8; ++it)
                for (int di = it * WG_WAVES + wave; di < ninstr; di += 8 * WG_WAVES) {
                    uint2 ent = lds_read64(st.entry_addr(di));
                    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(ent));
                    st.issue(di, ent, onext, cur ^ 1);
                }
        }
        if (p.dbuf) {
            cur ^= 1;
        } else if (more) {
            __syncthreads();                       // every wave is done reading the single buffer
            st.issue_all(onext, 0);
        }
    }
    // ---- partial tiles: D[m][n] of v_mfma_f32_16x16x32: n = lane & 15, m = 4 (lane >> 4) + reg
    float* out = p.partial + (size_t)split * taps * p.Mp * p.Np;
#pragma unroll
    for (int f = 0; f < FT; ++f) {
        const int tap = wave + WG_WAVES * f;
        if (tap >= taps) continue;                 // (7-tap kernels: the eighth wave multiplied rows below the patch)
#pragma unroll
        for (int a = 0; a < M16; ++a)
#pragma unroll
            for (int n = 0; n < N16; ++n)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    out[((size_t)tap * p.Mp + a * 16 + 4 * g4 + r) * p.Np + n * 16 + s16] = acc[f][a][n][r];
    }
    if (has_e) {
#pragma unroll
        for (int n = 0; n < N16; ++n)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                out[((size_t)tap_e * p.Mp + mt_e * 16 + 4 * g4 + r) * p.Np + n * 16 + s16] = acce[n][r];
    }
#endif
}

// dW[m][n][tap] (+)= sum over splits of partial[s][tap][m][n]; also used for 1x1 / Linear weights.
// A workgroup owns 64 consecutive (tap, m, n) elements with n fastest -- the partial buffers' own order, so a wave's
// read of one split is a coalesced 256-byte row -- and its four waves sum the splits s = w, w + 4, ... (four loads
// in flight per thread); the four partial sums are added in a fixed order (deterministic).  Only the single
// write per element is strided (by taps).  (One thread per element walking all <= 256 splits ran at 0.6 TB/s.)
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* __restrict__ partial, int ksplit, int taps, int M, int N,
                                                           int Mp, int Np, float* __restrict__ dw, int accumulate, float scale_h,
                                                           const float* __restrict__ scale_dev) {
    __shared__ float part[4][64];
    const float scale = scale_dev ? scale_h * scale_dev[0] : scale_h;
    const long long total = (long long)taps * M * N;
    const size_t sstride = (size_t)taps * Mp * Np;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    for (long long i0 = (long long)blockIdx.x * 64; i0 < total; i0 += (long long)gridDim.x * 64) {
        const long long i = i0 + lane;
        const bool live = i < total;
        const int n = (int)(i % N);
        const long long r = i / N;
        const int m = (int)(r % M), tap = (int)(r / M);
        float acc = 0.f;
        if (live) {
            const float* pp = partial + ((size_t)tap * Mp + m) * Np + n;
            int sidx = w;
            for (; sidx + 12 < ksplit; sidx += 16) {
                const float a = pp[(size_t)sidx * sstride], b = pp[(size_t)(sidx + 4) * sstride];
                const float c = pp[(size_t)(sidx + 8) * sstride], d = pp[(size_t)(sidx + 12) * sstride];
                acc += a; acc += b; acc += c; acc += d;
            }
            for (; sidx < ksplit; sidx += 4) acc += pp[(size_t)sidx * sstride];
        }
        part[w][lane] = acc;
        __syncthreads();
        if (w == 0 && live) {
            const float v = (((part[0][lane] + part[1][lane]) + part[2][lane]) + part[3][lane]) * scale;
            const size_t o = ((size_t)m * N + n) * taps + tap;
            dw[o] = accumulate ? dw[o] + v : v;
        }
        __syncthreads();
    }
}

// ---- 1x1 gradients with many channels on both sides (Linear layers, the LSTM input projections: dW_ih [1600][3072] over
// 11 392 pixels): a plain "TN" GEMM  dW[m][n] = sum_k G[k][m] X[k][n].  The pixel-tile kernel above gives such a shape
// 96 x 128 tiles with half of its waves idle (one tap: 4 (tap, n-tile) pairs for 8 waves) and re-reads X once per 96
// rows (0.059 of peak).  Here: 128 x 128 tiles, 4 waves of 64 x 64 (2 x 2 MFMA tiles), 64-pixel stages double buffered
// by LDS-DMA, the same [sub-image of 32 channels][pixel][64 B] LDS images and ds_read_b64_tr_b16 operand reads as
// wgrad_kernel, split-K partial sums into the same [ksplit][1][Mp][Np] buffer for wgrad_reduce_kernel.
// Channels past M / N read the neighbouring channels (or zeros past the buffer end): they only reach rows / columns that
// are not stored.  Pixels past the split's end lie beyond the buffer resource's range: zeros.
struct WgGemmParams {
    const bf16_t* g; const bf16_t* x; float* partial;
    int K, g_cs, x_cs, Mp, Np, tiles_n, ntiles, ksplit, kper;
};
#define WGG_KT 64
#define WGG_BUF 32768          // one stage: 16 KB of G + 16 KB of X

__global__ __launch_bounds__(256, 2) void wgrad_gemm_kernel(WgGemmParams p) {
#if __HIP_DEVICE_COMPILE__
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const unsigned sbase = (unsigned)(uintptr_t)smem;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // XCD-aware id mapping: an XCD's L2 serves a contiguous run of (split, tile) pairs (n fastest: they share the G tile)
    int bid = blockIdx.x;
    {
        const int nblk = gridDim.x, q = nblk / 8, r = nblk % 8;
        const int xcd = bid % 8, loc = bid / 8;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
    }
    const int split = bid / p.ntiles, tile = bid - split * p.ntiles;
    const int m0 = (tile / p.tiles_n) * 128, n0 = (tile % p.tiles_n) * 128;
    const int kbeg = split * p.kper, kend = min(p.K, kbeg + p.kper);
    const __amdgpu_buffer_rsrc_t rg = __builtin_amdgcn_make_buffer_rsrc((void*)p.g, 0, (unsigned)((long long)kend * p.g_cs * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, (unsigned)((long long)kend * p.x_cs * 2), 0x00020000);
    // DMA: instruction i of a stage fills 1 KB = 16 pixels x 64 B of sub-image (i & 15) >> 2 of G (i < 16) or X
    const unsigned lane_g = (unsigned)(((lane >> 2) * p.g_cs + (lane & 3) * 8) * 2);
    const unsigned lane_x = (unsigned)(((lane >> 2) * p.x_cs + (lane & 3) * 8) * 2);
    auto issue_stage = [&](const int k0, const int buf) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int i = wave + 4 * u;                 // wave-uniform
            const int j = i & 15, sub = j >> 2, pb = j & 3;
            const unsigned dst = (unsigned)(buf * WGG_BUF + i * 1024);
            if (i < 16) {
                const unsigned so = (unsigned)(((long long)(k0 + pb * 16) * p.g_cs + m0 + sub * 32) * 2);
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rg, (lds_ptr_t)(smem + dst), 16, lane_g + so, 0, 0, 0);
            } else {
                const unsigned so = (unsigned)(((long long)(k0 + pb * 16) * p.x_cs + n0 + sub * 32) * 2);
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, (lds_ptr_t)(smem + dst), 16, lane_x + so, 0, 0, 0);
            }
        }
    };
    const int g4 = lane >> 4, s16 = lane & 15;
    const int chan_off = (16 * (g4 & 1) + 4 * (s16 & 3)) * 2;    // byte offset of this lane's 4-channel run
    const int krow = 8 * (g4 >> 1) + (s16 >> 2);                 // pixel inside a 16-pixel k-step; +4 for the 2nd read
    const unsigned glane = (unsigned)(krow * 64 + chan_off);
    const int wm = wave >> 1, wn = wave & 1;
    const unsigned aoff = (unsigned)(wm * 2 * 4096) + glane, boff = (unsigned)(16384 + wn * 2 * 4096) + glane;

    f32x16 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[a][b][e] = 0.f;

    int cur = 0;
    if (kbeg < kend) issue_stage(kbeg, 0);
    for (int k0 = kbeg; k0 < kend; k0 += WGG_KT) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();                           // stage k0 has landed in buffer `cur`; buffer cur^1 is free
        if (k0 + WGG_KT < kend) issue_stage(k0 + WGG_KT, cur ^ 1);
        const unsigned ab = sbase + (unsigned)(cur * WGG_BUF) + aoff, bb = sbase + (unsigned)(cur * WGG_BUF) + boff;
        u32x4 av[2][2], bv[2][2];
        auto rd = [&](const unsigned addr) {
            const uint2 lo = lds_tr(addr), hi = lds_tr_off<256>(addr);
            return u32x4{lo.x, lo.y, hi.x, hi.y};
        };
        auto read_step = [&](const int ks, const int set) {
            av[set][0] = rd(ab + ks * 1024); av[set][1] = rd(ab + 4096 + ks * 1024);
            bv[set][0] = rd(bb + ks * 1024); bv[set][1] = rd(bb + 4096 + ks * 1024);
        };
        read_step(0, 0);
#pragma unroll
        for (int ks = 0; ks < WGG_KT / 16; ++ks) {
            const int set = ks & 1;
            if (ks + 1 < WGG_KT / 16) {
                read_step(ks + 1, set ^ 1);
                asm volatile("s_waitcnt lgkmcnt(8)" : "+v"(av[set][0]), "+v"(av[set][1]), "+v"(bv[set][0]), "+v"(bv[set][1]));
            } else {
                asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(av[set][0]), "+v"(av[set][1]), "+v"(bv[set][0]), "+v"(bv[set][1]));
            }
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < 2; ++b)
                    acc[a][b] = SOS_MFMA_32x32x16(__builtin_bit_cast(bf16x8, av[set][a]), __builtin_bit_cast(bf16x8, bv[set][b]),
                                                  acc[a][b], 0, 0, 0);
        }
        cur ^= 1;
    }
    // D[row = m][col = n], row = (reg&3) + 8*(reg>>2) + 4*(lane>>5), col = lane&31
    float* out = p.partial + (size_t)split * p.Mp * p.Np;
#pragma unroll
    for (int b = 0; b < 2; ++b) {
        const int n = n0 + (wn * 2 + b) * 32 + (lane & 31);
        if (n >= p.Np) continue;
#pragma unroll
        for (int a = 0; a < 2; ++a) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + (wm * 2 + a) * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                if (m < p.Mp) out[(size_t)m * p.Np + n] = acc[a][b][r];
            }
        }
    }
#endif
}

// split-K of the GEMM path: the fewest "rounds x stages" over 512 workgroup slots (2 per CU) plus the cost of writing
// and folding ksplit partial planes; never more planes than the workspace holds
static int wg_gemm_split(int ntiles, int K, int64_t plane_bytes, int cap) {
    int best = 1;
    double bt = 1e300;
    for (int ks = 1; ks <= cap && ks <= 32; ++ks) {
        const int kper = ((K + ks - 1) / ks + WGG_KT - 1) / WGG_KT * WGG_KT;
        if ((int64_t)(ks - 1) * kper >= K) continue;             // an empty split
        const double rounds = (double)(((int64_t)ntiles * ks + 511) / 512);
        const double t = rounds * (kper / WGG_KT) * 0.45 + (double)ks * plane_bytes * 2 / 4.0e6;
        if (t < bt) { bt = t; best = ks; }
    }
    return best;
}

// ---- 1x1 gradients with FEW channels on one side (the first / last layers of the encoders: 14 -> 48 / 96 with the horizontal
// taps folded into the channel axis, 48 / 96 -> 4 / 8): no FLOPs to speak of (<= 6 MFMAs per 32 pixels), two tensors to stream.
// The tiled kernel walks them at 1.6 TB/s (one 256-pixel tile in flight per CU behind a workgroup barrier).  Here every WAVE
// is its own stream: it owns a contiguous run of pixels, fetches 32 of them at a time -- M16 + N16 sub-images of 16 channels,
// one 1 KB LDS-DMA instruction each -- into a private ring of NBUF slots (NBUF - 1 stages in flight per wave: four waves x
// 3 x 7 KB = 84 KB per CU for 96 + 16 channels) and multiplies the slot that has landed with v_mfma_f32_16x16x32 through the
// same transposing reads as wgrad16_kernel.  No barrier until the four waves' sums are added (fixed order, through LDS) into
// the workgroup's plane of the [ksplit][1][Mp][Np] buffer that wgrad_reduce_kernel folds.  Pixels past a wave's run lie
// beyond its buffer resources' range (zeros); channels past M / N read neighbouring channels into rows / columns that are
// never stored.
struct WgThinParams {
    const bf16_t* g; const bf16_t* x; float* partial;
    int K, g_cs, x_cs, Mp, Np, kper;
};
#define WGT_NBUF 4

template <int M16, int N16>
__global__ __launch_bounds__(256) void wgrad_thin_kernel(WgThinParams p) {
#if __HIP_DEVICE_COMPILE__
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int SUBS = M16 + N16, STAGE = SUBS * 1024;
    const unsigned sbase = (unsigned)(uintptr_t)smem;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const long long kbeg = (long long)(blockIdx.x * 4 + wave) * p.kper;
    const long long kend = kbeg + p.kper < (long long)p.K ? kbeg + p.kper : (long long)p.K;
    const long long klim = kend > kbeg ? kend : 0;                 // an empty run reads nothing but zeros
    const __amdgpu_buffer_rsrc_t rg = __builtin_amdgcn_make_buffer_rsrc((void*)p.g, 0, (unsigned)(klim * p.g_cs * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, (unsigned)(klim * p.x_cs * 2), 0x00020000);
    // a DMA instruction fills 1 KB = 32 pixels x 32 B: lane l = pixel l >> 1, channels 8 (l & 1) .. + 7 of the sub-image
    const unsigned lane_g = (unsigned)(((lane >> 1) * p.g_cs + (lane & 1) * 8) * 2);
    const unsigned lane_x = (unsigned)(((lane >> 1) * p.x_cs + (lane & 1) * 8) * 2);
    const unsigned wbase = (unsigned)(wave * WGT_NBUF * STAGE);
    auto issue = [&](const long long k0, const int slot) {
        const unsigned dst = wbase + (unsigned)(slot * STAGE);
        const unsigned og = (unsigned)(k0 * p.g_cs * 2), ox = (unsigned)(k0 * p.x_cs * 2);
#pragma unroll
        for (int a = 0; a < M16; ++a)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rg, (lds_ptr_t)(smem + dst + a * 1024), 16, lane_g + og + a * 32, 0, 0, 0);
#pragma unroll
        for (int n = 0; n < N16; ++n)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, (lds_ptr_t)(smem + dst + (M16 + n) * 1024), 16, lane_x + ox + n * 32, 0, 0, 0);
    };
    const int g4 = lane >> 4, s16 = lane & 15;
    const unsigned rlane = (unsigned)((4 * g4 + (s16 >> 2)) * 32 + (s16 & 3) * 8);     // see wgrad16_kernel: pixel 4 g + e / 16 + 4 g + e

    f32x4 acc[M16][N16];
#pragma unroll
    for (int a = 0; a < M16; ++a)
#pragma unroll
        for (int n = 0; n < N16; ++n) acc[a][n] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int nst = kend > kbeg ? (int)((kend - kbeg + 31) / 32) : 0;
#pragma unroll
    for (int s = 0; s < WGT_NBUF - 1; ++s) issue(kbeg + 32ll * s, s);             // (stages past the run: zeros, never multiplied)
    int slot = 0;
    for (int s = 0; s < nst; ++s) {
        // stage s has landed when only the (NBUF - 2) younger stages' instructions are outstanding
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"((WGT_NBUF - 2) * SUBS) : "memory");
        const unsigned b = sbase + wbase + (unsigned)(slot * STAGE) + rlane;
        uint2 lo[SUBS], hi[SUBS];
#pragma unroll
        for (int i = 0; i < SUBS; ++i) { lo[i] = lds_tr(b + i * 1024); hi[i] = lds_tr(b + i * 1024 + 512); }
        // every half is named by a wait before anything reads it (the compiler does not know these reads are asynchronous)
#pragma unroll
        for (int i = 0; i < SUBS; ++i) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(lo[i]), "+v"(hi[i]));
        u32x4 av[M16], bv[N16];
#pragma unroll
        for (int a = 0; a < M16; ++a) av[a] = u32x4{lo[a].x, lo[a].y, hi[a].x, hi[a].y};
#pragma unroll
        for (int n = 0; n < N16; ++n) bv[n] = u32x4{lo[M16 + n].x, lo[M16 + n].y, hi[M16 + n].x, hi[M16 + n].y};
        // the slot that stage s - 1 used is free (its reads were waited for before its MFMAs): stage s + NBUF - 1 goes there
        issue(kbeg + 32ll * (s + WGT_NBUF - 1), slot == 0 ? WGT_NBUF - 1 : slot - 1);
#pragma unroll
        for (int a = 0; a < M16; ++a)
#pragma unroll
            for (int n = 0; n < N16; ++n)
                acc[a][n] = SOS_MFMA_16x16x32(__builtin_bit_cast(bf16x8, av[a]), __builtin_bit_cast(bf16x8, bv[n]), acc[a][n], 0, 0, 0);
        slot = slot + 1 == WGT_NBUF ? 0 : slot + 1;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();                               // every wave's ring is idle: the sums go through the same LDS
    float* red = (float*)smem;                     // [wave][M16 * N16 * 4][64]
#pragma unroll
    for (int a = 0; a < M16; ++a)
#pragma unroll
        for (int n = 0; n < N16; ++n)
#pragma unroll
            for (int r = 0; r < 4; ++r) red[(wave * (M16 * N16 * 4) + (a * N16 + n) * 4 + r) * 64 + lane] = acc[a][n][r];
    __syncthreads();
    float* out = p.partial + (size_t)blockIdx.x * p.Mp * p.Np;
    for (int idx = tid; idx < M16 * N16 * 4 * 64; idx += 256) {
        const int l = idx & 63, q = idx >> 6, r = q & 3, an = q >> 2, a = an / N16, n = an - a * N16;
        const float v = ((red[(0 * (M16 * N16 * 4) + q) * 64 + l] + red[(1 * (M16 * N16 * 4) + q) * 64 + l]) +
                         red[(2 * (M16 * N16 * 4) + q) * 64 + l]) + red[(3 * (M16 * N16 * 4) + q) * 64 + l];
        // D[m][n] of v_mfma_f32_16x16x32: n = lane & 15, m = 4 (lane >> 4) + reg
        const int m = a * 16 + 4 * (l >> 4) + r, nn = n * 16 + (l & 15);
        if (m < p.Mp && nn < p.Np) out[(size_t)m * p.Np + nn] = v;
    }
#endif
}

// ---- the same for a THIN INPUT under a kh x 1 kernel (the U-Net's first block with its horizontal taps folded into the channel
// axis: 10 of 16 stored channels, 5 vertical taps, reflection padding): tap a of pixel (b, h, w) multiplies X at row
// reflect(h + a dil - pad) of the same column -- one more 1 KB DMA instruction per tap and stage (X is the thin side: 32 B per
// pixel) whose per-lane source offset follows the lane's row; G is fetched once.  M16 x KH accumulator tiles per wave.
struct WgThinTapParams {
    const bf16_t* g; const bf16_t* x; float* partial;
    int K, H, W, g_cs, x_cs, Mp, Np, kper, dil, pad, reflect;
};

template <int M16, int KH>
__global__ __launch_bounds__(256) void wgrad_thin_taps_kernel(WgThinTapParams p) {
#if __HIP_DEVICE_COMPILE__
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int SUBS = M16 + KH, STAGE = SUBS * 1024;
    const unsigned sbase = (unsigned)(uintptr_t)smem;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const long long kbeg = (long long)(blockIdx.x * 4 + wave) * p.kper;
    const long long kend = kbeg + p.kper < (long long)p.K ? kbeg + p.kper : (long long)p.K;
    const long long klim = kend > kbeg ? kend : 0;
    const __amdgpu_buffer_rsrc_t rg = __builtin_amdgcn_make_buffer_rsrc((void*)p.g, 0, (unsigned)(klim * p.g_cs * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, (unsigned)((long long)p.K * p.x_cs * 2), 0x00020000);
    const unsigned lane_g = (unsigned)(((lane >> 1) * p.g_cs + (lane & 1) * 8) * 2);
    const unsigned xhalf = (unsigned)((lane & 1) * 16);
    const unsigned wbase = (unsigned)(wave * WGT_NBUF * STAGE);
    // this lane's pixel of the NEXT stage to be issued: flat index, row, column (advanced by 32 pixels per stage)
    long long lk = kbeg + (lane >> 1);
    int lw = (int)(lk % p.W), lh = (int)((lk / p.W) % p.H);
    auto issue = [&](const long long k0, const int slot) {
        const unsigned dst = wbase + (unsigned)(slot * STAGE);
        const unsigned og = (unsigned)(k0 * p.g_cs * 2);
#pragma unroll
        for (int a = 0; a < M16; ++a)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rg, (lds_ptr_t)(smem + dst + a * 1024), 16, lane_g + og + a * 32, 0, 0, 0);
#pragma unroll
        for (int t = 0; t < KH; ++t) {
            int hh = lh + t * p.dil - p.pad;
            if (p.reflect) hh = reflect_index(hh, p.H);
            const bool ok = (unsigned)hh < (unsigned)p.H && lk < kend;
            const unsigned vo = ok ? (unsigned)((lk + (long long)(hh - lh) * p.W) * p.x_cs * 2) + xhalf : 0xffffffffu;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, (lds_ptr_t)(smem + dst + (M16 + t) * 1024), 16, vo, 0, 0, 0);
        }
        lk += 32; lw += 32;
        while (lw >= p.W) { lw -= p.W; if (++lh == p.H) lh = 0; }
    };
    const int g4 = lane >> 4, s16 = lane & 15;
    const unsigned rlane = (unsigned)((4 * g4 + (s16 >> 2)) * 32 + (s16 & 3) * 8);

    f32x4 acc[KH][M16];
#pragma unroll
    for (int t = 0; t < KH; ++t)
#pragma unroll
        for (int a = 0; a < M16; ++a) acc[t][a] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int nst = kend > kbeg ? (int)((kend - kbeg + 31) / 32) : 0;
#pragma unroll
    for (int s = 0; s < WGT_NBUF - 1; ++s) issue(kbeg + 32ll * s, s);
    int slot = 0;
    for (int s = 0; s < nst; ++s) {
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"((WGT_NBUF - 2) * SUBS) : "memory");
        const unsigned b = sbase + wbase + (unsigned)(slot * STAGE) + rlane;
        uint2 lo[SUBS], hi[SUBS];
#pragma unroll
        for (int i = 0; i < SUBS; ++i) { lo[i] = lds_tr(b + i * 1024); hi[i] = lds_tr(b + i * 1024 + 512); }
#pragma unroll
        for (int i = 0; i < SUBS; ++i) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(lo[i]), "+v"(hi[i]));
        issue(kbeg + 32ll * (s + WGT_NBUF - 1), slot == 0 ? WGT_NBUF - 1 : slot - 1);
#pragma unroll
        for (int t = 0; t < KH; ++t) {
            const u32x4 bv = u32x4{lo[M16 + t].x, lo[M16 + t].y, hi[M16 + t].x, hi[M16 + t].y};
#pragma unroll
            for (int a = 0; a < M16; ++a) {
                const u32x4 av = u32x4{lo[a].x, lo[a].y, hi[a].x, hi[a].y};
                acc[t][a] = SOS_MFMA_16x16x32(__builtin_bit_cast(bf16x8, av), __builtin_bit_cast(bf16x8, bv), acc[t][a], 0, 0, 0);
            }
        }
        slot = slot + 1 == WGT_NBUF ? 0 : slot + 1;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    float* red = (float*)smem;                     // [wave][KH * M16 * 4][64]
    constexpr int NQ = KH * M16 * 4;
#pragma unroll
    for (int t = 0; t < KH; ++t)
#pragma unroll
        for (int a = 0; a < M16; ++a)
#pragma unroll
            for (int r = 0; r < 4; ++r) red[(wave * NQ + (t * M16 + a) * 4 + r) * 64 + lane] = acc[t][a][r];
    __syncthreads();
    float* out = p.partial + (size_t)blockIdx.x * KH * p.Mp * p.Np;
    for (int idx = tid; idx < NQ * 64; idx += 256) {
        const int l = idx & 63, q = idx >> 6, r = q & 3, ta = q >> 2, t = ta / M16, a = ta - t * M16;
        const float v = ((red[(0 * NQ + q) * 64 + l] + red[(1 * NQ + q) * 64 + l]) + red[(2 * NQ + q) * 64 + l]) + red[(3 * NQ + q) * 64 + l];
        const int m = a * 16 + 4 * (l >> 4) + r, nn = l & 15;
        if (m < p.Mp && nn < p.Np) out[((size_t)t * p.Mp + m) * p.Np + nn] = v;
    }
#endif
}

// which 1x1 gradients take the streaming kernel, and with how many workgroups (= partial planes)
static bool wg_thin_shape(const sos_wgrad_desc* d, bool is_flat, int* m16, int* n16) {
    if (!is_flat || getenv("SOS_WGRAD_NO_THIN")) return false;
    const int a = (d->M + 15) / 16, b = (d->N + 15) / 16;
    if (!((a == 1 && b >= 1 && b <= 6) || (b == 1 && a >= 1 && a <= 6))) return false;
    if (a == 5 || b == 5) return false;                       // (no instance)
    // the kernel's buffer resources and offsets are 32-bit ((unsigned)(k * cs * 2)); a descriptor that ARRIVES flat
    // (B = Hg = 1, a huge Wg) has not been through the flatten step's range check -- and the 16-channel sub-images the kernel
    // fetches (16 m16 / 16 n16 channels from g_off / x_off) must lie inside a pixel's channel run.  Anything else: tiled kernel.
    const uint64_t npx = (uint64_t)d->B * d->Hg * d->Wg;
    if (npx * (uint64_t)d->g_cs * 2 >= 0xfff00000ull || npx * (uint64_t)d->x_cs * 2 >= 0xfff00000ull) return false;
    if (d->g_off + 16 * a > d->g_cs || d->x_off + 16 * b > d->x_cs) return false;
    *m16 = a; *n16 = b;
    return true;
}

// ksplit <= 0 in the descriptor = automatic: one workgroup per CU (MI355X: 256) over (pixel split, m-group, n-group),
// bounded by 256 MB of partial sums.
static const int WG_NCU = 256;
static const int WG_MAXSPLIT = 1024;      // several workgroups per CU for gradients whose tiles are short (see sos_conv2d_wgrad)
static int wg_max_split(const sos_wgrad_desc* d) {
    const int64_t Mp = (d->M + 31) / 32 * 32, Np = (d->N + 31) / 32 * 32;
    const int64_t per = (int64_t)d->kh * d->kw * Mp * Np * 4;
    const int64_t cap = ((int64_t)256 << 20) / per;
    return (int)(cap < 1 ? 1 : (cap > WG_MAXSPLIT ? WG_MAXSPLIT : cap));
}
// the kh x 1 gradients of a thin input that take wgrad_thin_taps_kernel (instance: 49..64 output channels, 5 taps)
static bool wg_thin_taps_shape(const sos_wgrad_desc* d, bool temporal) {
    if (temporal || getenv("SOS_WGRAD_NO_THIN") || d->kw != 1 || d->kh != 5 || d->stride != 1 || d->N > 16 || (d->M + 15) / 16 != 4) return false;
    if (d->Hg != d->Hx || d->Wg != d->Wx || d->pad_left != 0 || d->Wg < 32) return false;
    if (d->pad_mode == SOS_PAD_REFLECT && d->pad_top >= d->Hg) return false;
    const uint64_t npx = (uint64_t)d->B * d->Hg * d->Wg;
    return npx * (uint64_t)d->g_cs * 2 < 0xfff00000ull && npx * (uint64_t)d->x_cs * 2 < 0xfff00000ull;
}
// workgroups (= partial planes) of wgrad_thin_kernel: one per CU, at least 16 stages per wave
static int wg_thin_split(const sos_wgrad_desc* d, int subs) {
    int occ = 1;                                   // measured (48 + 16 channels, 2.9 M pixels): 1 -> 73 us, 2 -> 81, 3 -> 90
    { const char* e = getenv("SOS_WGT_OCC"); if (e && atoi(e) >= 1 && atoi(e) <= 8) occ = atoi(e); }
    int n = WG_NCU * occ;
    const int cap = d->ksplit > 0 ? d->ksplit : wg_max_split(d);
    if (n > cap) n = cap;
    const int by_work = d->Wg / (4 * 32 * 16);
    if (n > by_work) n = by_work;
    return n < 1 ? 1 : n;
}


// ---------------------------------------------------------------------------------------------------------------------------
// Host side of wgrad_kernel / wgrad16_kernel: a PLAN = the workgroup's channel tile (MT m-tiles x NTB n-tiles of 32), its pixel
// tile (NC residue classes x 2^lth x 2^ltw = 256 pixels), the order of those pixels along the contraction and the number of
// workgroups per CU.  Plans come from (1) the measured table (round 4: sos_wgrad_tune / sos_wgrad_tune_load, the shipped
// wgrad_table_gfx950.txt -- every process and every rank then runs the same plans, hence the same summation order), else (2) the
// calibrated cost model below.
struct WgPlan { int mt, ntb, nc, lth, ltw, kord, occ, db, bufbytes, npixp, pw; double cost; };
struct WgCtx {
    bool temporal, is_flat, use16;
    int khg, ntg, taps, taps_all, ntiles_m, ntiles_n, m16, n16, Hc, Wc;
};
struct WgKey {
    int v[12];
    bool operator<(const WgKey& o) const { return memcmp(v, o.v, sizeof(v)) < 0; }
};
static std::mutex& wg_mu() { static std::mutex m; return m; }
static std::map<WgKey, WgPlan>& wg_table() { static std::map<WgKey, WgPlan> t; return t; }      // measured plans
static std::map<WgKey, WgPlan>& wg_model_cache() { static std::map<WgKey, WgPlan> t; return t; } // the cost model's picks

static WgCtx wg_ctx(const sos_wgrad_desc* d, bool temporal, bool is_flat) {
    WgCtx c;
    c.temporal = temporal; c.is_flat = is_flat;
    // more taps than one workgroup's 32 (tap, n-tile) pairs (7x7): the tap ROWS are divided over ntg workgroups that read the
    // same G tile at the same time on the same XCD (one launch, G fetched from HBM once instead of once per row)
    // (the last group may own fewer real rows: its phantom taps are computed on the rows below and never stored)
    int khg = d->kh;
    if (khg * d->kw > WG_WAVES * WG_PAIRS) khg = WG_WAVES * WG_PAIRS / d->kw < 1 ? 1 : WG_WAVES * WG_PAIRS / d->kw;
    c.khg = khg; c.ntg = (d->kh + khg - 1) / khg;
    c.taps_all = d->kh * d->kw; c.taps = khg * d->kw;
    c.ntiles_m = (d->M + 31) / 32; c.ntiles_n = (d->N + 31) / 32;
    c.m16 = (d->M + 15) / 16; c.n16 = (d->N + 15) / 16;
    // small channel counts: the 16x16x32 kernel owns all of dW in one workgroup (no padding to 32)
    c.use16 = !temporal && c.ntg == 1 && c.m16 == 3 && c.n16 == 3 && (c.taps == 25 || c.taps == 9 || (c.taps == 7 && !getenv("SOS_WGRAD_NO16_7"))) &&
              !getenv("SOS_WGRAD_NO16");      // (7 taps: one per wave, the eighth wave only stages)
    c.Hc = (d->Hg + d->dil_h - 1) / d->dil_h; c.Wc = (d->Wg + d->dil_w - 1) / d->dil_w;
    return c;
}
static WgKey wg_key(const sos_wgrad_desc* d, const WgCtx& c) {
    return WgKey{{d->Hg, d->Wg, d->kh, d->kw, d->stride, d->dil_h, d->dil_w, d->M, d->N, c.use16 ? 1 : 0, c.temporal ? d->t_taps : 0,
                  c.is_flat ? 1 : 0}};
}

// Completes a plan (patch pitch, LDS bytes, double buffering) for the given channel / pixel tile and prices it with the cost
// model; false: the tile is not legal for this shape (dilation classes, coordinate range, LDS).
static bool wg_make_plan(const sos_wgrad_desc* d, const WgCtx& c, int mt, int ntb, int lnc, int lth, int kord, int occ, WgPlan* out) {
    const size_t lds_max = 160 * 1024;
    const int NC = 1 << lnc, ltw = 8 - lnc - lth;
    if (ltw < 2 || lth < 0) return false;
    if (NC > 1 && (d->stride > 1 || NC > d->dil_w || d->dil_w % NC)) return false;
    const int TH = 1 << lth, TW = 1 << ltw;
    const int PH = (TH - 1) * d->stride + c.khg, PW = (TW - 1) * d->stride + d->kw;
    if ((TH - 1 + d->kh) * d->dil_h >= 0x7fff || (d->stride * NC + (PW - 1) * d->dil_w) >= 0x7fff) return false;
    const int tiles_h = (c.Hc + TH - 1) / TH, tiles_w = (c.Wc + TW - 1) / TW, ngw = (d->dil_w + NC - 1) / NC;
    const double ntiles = (double)d->dil_h * ngw * tiles_h * tiles_w;
    // Column-major order: the pixels a transposed read gathers are consecutive ROWS of the patch, PWl pixel
    // pitches apart; with an even PWl their 64-byte (32-byte) runs land on the same banks (PW = 12: 768 B = 0 mod
    // 256 -- SQ_LDS_BANK_CONFLICT was 59 % of the LDS-active cycles of the 96 -> 96 gradient), with an odd one
    // they tile the 64 banks.  One more (never multiplied) patch column buys that.
    const int PWl = (kord == 1 && (PW & 1) == 0 && !getenv("SOS_WGRAD_EVEN_PITCH")) ? PW + 1 : PW;
    const bool use16 = c.use16;
    const int npixp = use16 ? (NC * PH * PWl + 31) / 32 * 32 : (NC * PH * PWl + 15) / 16 * 16;
    const size_t one = use16 ? ((size_t)256 * 32 * c.m16 + (size_t)npixp * 32 * c.n16 + 1023) / 1024 * 1024
                             : ((size_t)256 * 64 * mt + (size_t)npixp * 64 * ntb + 1023) / 1024 * 1024;
    const size_t tabb = (size_t)(256 + npixp) * 8;
    if (one + tabb > lds_max / occ) return false;
    const int db = 2 * one + tabb <= lds_max / occ;
    // Cost of one image in k-step units, calibrated on MI355X with tools/probe/wgrad_tile_sweep.py (round 3: every
    // tile x order of the 96- and 48-channel layers timed; this model's pick is within 1.1 % of the best measured
    // one on each).  A tile multiplies its k-steps that hold a pixel of the image (first pixel of the k-step
    // inside); meanwhile the DMA of the NEXT tile runs, `kb` k-steps per KB of operand image whether its pixels
    // are inside or not -- so a border tile that keeps 2 of its k-steps still takes its successor's fetch time
    // (double buffered: the longer of the two; single buffered: a fraction of the fetch is exposed) -- plus a
    // fixed barrier / drain cost per tile.  The 16x16x32 kernel's k-step is 32 pixels x 9 tiles per tap and
    // fetches 48-channel operands: its DMA term weighs more (the former bank-conflict term of that kernel is
    // subsumed: the sweep ranks its tiles correctly without it).
    const double kb = use16 ? 0.15 : 0.04, sbf = use16 ? 0.3 : 0.5, fixed = use16 ? 1.5 : 1.0;
    const double dma = (double)one / 1024.0 * kb;
    const int kpix = use16 ? 32 : 16;              // tile pixels per k-step
    double cost = 0.0;
    const int nk = 256 / kpix;
    int kh_rel[16], kw_rel[16];                       // first pixel of every k-step, relative to the tile origin
    for (int ks = 0; ks < nk; ++ks) {
        int cc, i, j;
        wg_decode(ks * kpix, lth, ltw, kord, cc, i, j);
        kh_rel[ks] = i * d->dil_h; kw_rel[ks] = cc + j * d->dil_w;
    }
    for (int rh = 0; rh < d->dil_h; ++rh)
        for (int gw = 0; gw < ngw; ++gw)
            for (int ti = 0; ti < tiles_h; ++ti)
                for (int tj = 0; tj < tiles_w; ++tj) {
                    const int gh0 = rh + ti * TH * d->dil_h, gw0 = gw * NC + tj * TW * d->dil_w;
                    int nks = 0;
                    for (int ks = 0; ks < nk; ++ks) nks += (gh0 + kh_rel[ks] < d->Hg) & (gw0 + kw_rel[ks] < d->Wg);
                    cost += db ? (nks > dma ? nks : dma) : nks + sbf * dma;
                }
    cost += fixed * ntiles;
    *out = WgPlan{mt, ntb, NC, lth, ltw, kord, occ, db, (int)one, npixp, PWl, cost};
    return true;
}

// best pixel tile of the cost model for a channel tile (mt, ntb); false: no tile fits LDS
static bool wg_best_tile(const sos_wgrad_desc* d, const WgCtx& c, int mt, int ntb, int occ, WgPlan* out) {
    int fnc = -1, fth = -1, ftw = -1, fko = -1;
    const char* force = getenv("SOS_WGRAD_TILE");  // experiments: "nc,lth,ltw,kord"
    if (force) sscanf(force, "%d,%d,%d,%d", &fnc, &fth, &ftw, &fko);
    bool any = false;
    for (int lnc = 0; lnc <= 6; ++lnc)
        for (int lth = 0; lth + lnc <= 8; ++lth)
            for (int kord = 0; kord < 2; ++kord) {
                if (force && fnc > 0 && ((1 << lnc) != fnc || lth != fth || 8 - lnc - lth != ftw)) continue;
                if (force && fko >= 0 && kord != fko) continue;
                WgPlan pl;
                if (!wg_make_plan(d, c, mt, ntb, lnc, lth, kord, occ, &pl)) continue;
                if (!any || pl.cost < out->cost) { *out = pl; any = true; }
            }
    return any;
}

// SIMD load of `pairs` (tap, n-tile) pairs dealt round robin to 8 waves, two waves per SIMD
static int wg_simd_max(int pairs) {
    int simd[4] = {0, 0, 0, 0};
    for (int w = 0; w < WG_WAVES; ++w) simd[w & 3] += pairs / WG_WAVES + (w < pairs % WG_WAVES ? 1 : 0);
    return std::max(std::max(simd[0], simd[1]), std::max(simd[2], simd[3]));
}

// The cost model's plan.
static int wg_model_plan(const sos_wgrad_desc* d, const WgCtx& c, WgPlan* out) {
    int ntb = WG_WAVES * WG_PAIRS / c.taps;         // (tap, n-tile) pairs per workgroup <= 32
    ntb = ntb >= 4 ? 4 : (ntb >= 2 ? 2 : 1);
    if (ntb > c.ntiles_n) ntb = c.ntiles_n >= 4 ? 4 : (c.ntiles_n >= 2 ? 2 : 1);
    int mgroups = (c.ntiles_m + 2) / 3;
    int mt = (c.ntiles_m + mgroups - 1) / mgroups;              // 1..3 m-tiles per workgroup, balanced
    bool forced_ntb = false;
    {   // experiments (tools/probe/wgrad_cfg_sweep.py): force the workgroup's channel tile
        const char* e = getenv("SOS_WGRAD_MT");
        if (e && atoi(e) >= 1 && atoi(e) <= 3) { mt = std::min(atoi(e), c.ntiles_m); }
        e = getenv("SOS_WGRAD_NTB");
        if (e && atoi(e) >= 1 && atoi(e) <= 4 && atoi(e) * c.taps <= WG_WAVES * WG_PAIRS) { ntb = std::min(atoi(e), c.ntiles_n); forced_ntb = true; }
    }
    // workgroups per CU: one (its own double-buffered DMA pipeline covers the fetch of the next tile) unless a tile is so
    // short that the fetch latency of a tile exceeds its MFMA time -- then several co-resident workgroups cover each other
    // (measured: the 96 -> 8 / 48 -> 4 1x1 heads, two MFMAs per k-step and workgroup: 0.456 -> 0.338 / 0.239 -> 0.185 ms with
    // two workgroups per CU; the thin 5x5 / 1x7 first layers, whose double buffers no longer fit then, get slower)
    // (round 4: only when the M side is the thin one -- 96 -> 8: 0.42 -> 0.32 ms, 48 -> 4: 0.22 -> 0.18 with two; a thin N side with
    // two m-tiles -- the first layers with their taps folded, 14 -> 48 -- is FASTER with one: 0.235 vs 0.285 ms)
    int occ = c.is_flat && mt == 1 && ntb <= 2 ? 2 : 1;
    { const char* e = getenv("SOS_WGRAD_OCC"); if (e && atoi(e) >= 1 && atoi(e) <= 4) occ = atoi(e); }
    if (c.use16) { if (!wg_best_tile(d, c, mt, ntb, 1, out)) { sos_set_error("sos_conv2d_wgrad: patch does not fit LDS"); return SOS_ENOSPC; } return SOS_OK; }
    // Few-tap kernels (3x3, 7x1; round 4): the (tap, n-tile) pairs of NTB n-tiles go round robin over 8 waves, two waves per
    // SIMD, so NTB decides (a) how much of the last n-group is padding, (b) how evenly the pairs load the four SIMDs, (c) how
    // many fragment reads an MFMA costs ((MT + pairs per wave) reads feed MT x pairs MFMAs) and (d) which pixel tiles still fit
    // LDS, double buffered or not.  Time model: tile cost x n-groups x busiest SIMD's pairs x (1 + 0.8 reads per MFMA),
    // fitted to tools/probe/wgrad_cfg_sweep.py on MI355X: 96 -> 96 7x1 takes NTB = 3 (21 pairs, one n-group: 0.73 -> 0.47 ms),
    // 256 -> 256 3x3 NTB = 3 (0.292 -> 0.280), its dilation-16 sibling and 128 -> 256 / 64 -> 128 3x3 keep NTB = 2.
    if (!forced_ntb && c.taps >= 5 && c.taps * 2 <= WG_WAVES * WG_PAIRS) {
        double best = 1e300;
        bool any = false;
        for (int cn = 1; cn <= 4 && cn <= c.ntiles_n && cn * c.taps <= WG_WAVES * WG_PAIRS; ++cn) {
            WgPlan pl;
            if (!wg_best_tile(d, c, mt, cn, 1, &pl)) continue;
            const int pairs = cn * c.taps, groups = (c.ntiles_n + cn - 1) / cn;
            const double ppw = (double)pairs / WG_WAVES, reads = (mt + ppw) / (mt * ppw);
            const double t = pl.cost * groups * wg_simd_max(pairs) * (1.0 + 0.8 * reads);
            if (t < best) { best = t; *out = pl; any = true; }
        }
        if (any) return SOS_OK;
    }
    for (int ntb_try = ntb;;) {
        int o = occ;
        while (o >= 1 && !wg_best_tile(d, c, mt, ntb_try, o, out)) --o;     // (a second workgroup per CU only if the buffers allow it)
        if (o >= 1) return SOS_OK;
        if (ntb_try == 1) { sos_set_error("sos_conv2d_wgrad: patch does not fit LDS"); return SOS_ENOSPC; }
        ntb_try = ntb_try == 3 ? 2 : ntb_try >> 1;
    }
}

static int wg_choose_plan(const sos_wgrad_desc* d, const WgCtx& c, WgPlan* out) {
    const WgKey key = wg_key(d, c);
    const bool forced = getenv("SOS_WGRAD_TILE") || getenv("SOS_WGRAD_MT") || getenv("SOS_WGRAD_NTB") || getenv("SOS_WGRAD_OCC");
    if (!forced) {
        std::lock_guard<std::mutex> lk(wg_mu());
        auto f = wg_table().find(key);
        if (f != wg_table().end()) { *out = f->second; return SOS_OK; }
        auto g = wg_model_cache().find(key);
        if (g != wg_model_cache().end()) { *out = g->second; return SOS_OK; }
    }
    const int rc = wg_model_plan(d, c, out);
    if (rc) return rc;
    if (getenv("SOS_WGRAD_VERBOSE"))
        fprintf(stderr, "sos_conv2d_wgrad: %dx%d k%dx%d s%d d%dx%d M%d N%d %s-> MT=%d NTB=%d NC=%d TH=%d TW=%d order=%d dbuf=%d occ=%d lds=%d pitch=%d\n",
                d->Hg, d->Wg, d->kh, d->kw, d->stride, d->dil_h, d->dil_w, d->M, d->N, c.use16 ? "(16x16x32) " : "", out->mt, out->ntb,
                out->nc, 1 << out->lth, 1 << out->ltw, out->kord, out->db, out->occ, out->bufbytes, out->pw);
    if (!forced) {
        std::lock_guard<std::mutex> lk(wg_mu());
        wg_model_cache()[key] = *out;
    }
    return SOS_OK;
}

static int wg_launch(const sos_wgrad_desc* d, const WgCtx& c, const WgPlan& pl, hipStream_t s, const int what = 3) {
    WgParams p;
    p.g = (const bf16_t*)d->g; p.x = (const bf16_t*)d->x; p.partial = d->partial;
    p.B = d->B; p.Hg = d->Hg; p.Wg = d->Wg; p.g_cs = d->g_cs; p.g_off = d->g_off;
    p.Hx = d->Hx; p.Wx = d->Wx; p.x_cs = d->x_cs; p.x_off = d->x_off;
    p.M = d->M; p.N = d->N; p.Mp = (d->M + 31) / 32 * 32; p.Np = (d->N + 31) / 32 * 32;
    p.kw = d->kw; p.stride = d->stride; p.dh = d->dil_h; p.dw = d->dil_w;
    p.pad_t = d->pad_top; p.pad_l = d->pad_left; p.pad_mode = d->pad_mode;
    p.tT = c.temporal ? d->t_frames : 0; p.tcin = c.temporal ? d->t_cin : 0; p.tpad = c.temporal ? d->t_pad : 0;
    p.kh = c.khg; p.ntg = c.ntg; p.taps_all = c.taps_all;
    const int taps = c.taps, taps_all = c.taps_all, mt = pl.mt, ntb = pl.ntb;
    const bool use16 = c.use16;
    const int mgroups = (c.ntiles_m + mt - 1) / mt;
    p.NC = pl.nc; p.logTH = pl.lth; p.logTW = pl.ltw; p.dbuf = pl.db; p.kord = pl.kord; p.bufbytes = pl.bufbytes; p.npixp = pl.npixp;
    {
        const int TH = 1 << p.logTH, TW = 1 << p.logTW;
        p.tiles_h = (c.Hc + TH - 1) / TH; p.tiles_w = (c.Wc + TW - 1) / TW; p.ngw = (d->dil_w + p.NC - 1) / p.NC;
        p.PH = (TH - 1) * d->stride + c.khg; p.PW = pl.pw;        // patch pitch: (TW - 1) stride + kw, + 1 when that keeps it odd
        p.npix = p.NC * p.PH * p.PW;
    }
    const int occ = pl.occ;
    const size_t lds = (size_t)p.bufbytes * (p.dbuf ? 2 : 1) + (size_t)(256 + p.npixp) * 8;
    { const char* e = getenv("SOS_WGRAD_DBG"); p.dbg = e ? atoi(e) : 0; }
    p.nsteps = d->B * d->dil_h * p.ngw * p.tiles_h * p.tiles_w;
    int ksplit = d->ksplit;
    if (ksplit <= 0) {
        const int groups = use16 ? 1 : mgroups * ((c.ntiles_n + ntb - 1) / ntb) * p.ntg;
        ksplit = occ * WG_NCU / groups;
        if (ksplit < 1) ksplit = 1;
        const int cap = wg_max_split(d);
        if (ksplit > cap) ksplit = cap;
    }
    if (ksplit > p.nsteps) ksplit = p.nsteps;                            // never an empty split
    p.ksplit = ksplit;
    p.steps_per_split = (p.nsteps + ksplit - 1) / ksplit;
    p.ny = mgroups; p.nz = (c.ntiles_n + ntb - 1) / ntb * p.ntg;
    p.xcdmap = 0;
    if (!use16 && d->ksplit <= 0 && p.ny * p.nz > 1 && p.ny * p.nz <= 16 && !getenv("SOS_WGRAD_NOXCD")) {
        // one workgroup per CU: an XCD (32 CUs) takes floor(32 / groups) splits, all groups of a split on one XCD
        const int per_xcd = occ * 32 / (p.ny * p.nz);
        int ks8 = 8 * per_xcd;
        if (ks8 > p.nsteps) ks8 = p.nsteps / 8 * 8;
        const int cap = wg_max_split(d) / 8 * 8;
        if (ks8 > cap) ks8 = cap;
        if (ks8 >= 8 && ks8 * 100 >= ksplit * 93) {        // only if (almost) as many workgroups as one per CU remain
            ksplit = ks8;
            p.ksplit = ksplit;
            p.steps_per_split = (p.nsteps + ksplit - 1) / ksplit;
            p.xcdmap = 1;
        }
    }
    dim3 grid((unsigned)(ksplit * p.ny * p.nz), 1, 1);                       // see the id mapping in wgrad_kernel
    static sos_device_once attr_once;
    if (use16) grid = dim3((unsigned)ksplit, 1, 1);
#define SOS_WG_ATTR(MTV, NTBV) \
    (void)hipFuncSetAttribute((const void*)wgrad_kernel<MTV, NTBV>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
#define SOS_WG_CASE(MTV, NTBV) \
    if (mt == MTV && ntb == NTBV) hipLaunchKernelGGL((wgrad_kernel<MTV, NTBV>), grid, dim3(WG_THREADS), lds, s, p);
    (void)sos_per_device_once(attr_once, [] {       // every instantiation may use the full 160 KB of LDS
        SOS_WG_ATTR(1, 1) SOS_WG_ATTR(1, 2) SOS_WG_ATTR(1, 4) SOS_WG_ATTR(2, 1) SOS_WG_ATTR(2, 2) SOS_WG_ATTR(2, 4)
        SOS_WG_ATTR(3, 1) SOS_WG_ATTR(3, 2) SOS_WG_ATTR(3, 4) SOS_WG_ATTR(1, 3) SOS_WG_ATTR(2, 3) SOS_WG_ATTR(3, 3)
        (void)hipFuncSetAttribute((const void*)wgrad_kernel<3, 1, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute((const void*)wgrad_kernel<2, 1, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute((const void*)wgrad16_kernel<3, 3, 3>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute((const void*)wgrad16_kernel<3, 3, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        return (int)SOS_OK;
    });
    if (!(what & 1)) {
        // reduce only
    } else if (use16) {
        if (taps == 25) hipLaunchKernelGGL((wgrad16_kernel<3, 3, 3>), grid, dim3(WG_THREADS), lds, s, p);
        else hipLaunchKernelGGL((wgrad16_kernel<3, 3, 1>), grid, dim3(WG_THREADS), lds, s, p);
    } else {
        const bool bal = taps == 25 && ntb == 1 && mt >= 2 && !getenv("SOS_WGRAD_NOBAL");
        if (bal && mt == 3) hipLaunchKernelGGL((wgrad_kernel<3, 1, true>), grid, dim3(WG_THREADS), lds, s, p);
        else if (bal && mt == 2) hipLaunchKernelGGL((wgrad_kernel<2, 1, true>), grid, dim3(WG_THREADS), lds, s, p);
        else {
        SOS_WG_CASE(1, 1) SOS_WG_CASE(1, 2) SOS_WG_CASE(1, 4) SOS_WG_CASE(2, 1) SOS_WG_CASE(2, 2) SOS_WG_CASE(2, 4)
        SOS_WG_CASE(3, 1) SOS_WG_CASE(3, 2) SOS_WG_CASE(3, 4) SOS_WG_CASE(1, 3) SOS_WG_CASE(2, 3) SOS_WG_CASE(3, 3)
        }
    }
#undef SOS_WG_CASE
#undef SOS_WG_ATTR
    int rc = sos_check_launch("sos_conv2d_wgrad");
    if (rc) return rc;
    if (!(what & 2)) return SOS_OK;
    const long long total = (long long)d->M * d->N * taps_all;
    long long gb = (total + 63) / 64;
    if (gb > 8192) gb = 8192;
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)gb), dim3(256), 0, s, d->partial, ksplit, taps_all, d->M, d->N,
                       p.Mp, p.Np, d->dw, d->accumulate, d->scale, d->scale_dev);
    return sos_check_launch("sos_conv2d_wgrad(reduce)");
}

extern "C" int64_t sos_wgrad_workspace_bytes(const sos_wgrad_desc* d) {
    if (!d) return -1;
    const int64_t Mp = (d->M + 31) / 32 * 32, Np = (d->N + 31) / 32 * 32;
    return (int64_t)(d->ksplit > 0 ? d->ksplit : wg_max_split(d)) * d->kh * d->kw * Mp * Np * 4;
}

// what: 1 = the MFMA kernel (partial sums per pixel split), 2 = the deterministic reduce into dw, 3 = both (sos_conv2d_wgrad).
// The two halves recompute the same launch plan from the descriptor, so they may run on different streams (ABI 8).
static int wgrad_impl(const sos_wgrad_desc* d, sos_stream_t stream, const int what) {
    if (!d || !d->g || !d->x || !d->partial || !d->dw) { sos_set_error("sos_conv2d_wgrad: null pointer"); return SOS_EINVAL; }
    if (d->M < 1 || d->N < 1 || d->g_cs % 8 || d->x_cs % 8 || d->g_off % 8 || d->x_off % 8 || d->kh < 1 || d->kw < 1 ||
        d->stride < 1 || d->dil_h < 1 || d->dil_w < 1 || (d->stride > 1 && (d->dil_h > 1 || d->dil_w > 1)) ||
        d->B < 1) {
        sos_set_error("sos_conv2d_wgrad: bad descriptor");
        return SOS_EINVAL;
    }
    const bool temporal = d->t_taps > 1;
    if (temporal && (d->t_frames < 1 || d->B % d->t_frames || d->t_cin < 128 || d->t_cin % 128 || d->N != d->t_taps * d->t_cin ||
                     d->t_pad < 0 || d->t_pad >= d->t_taps || d->x_off + d->t_cin > d->x_cs)) {
        sos_set_error("sos_conv2d_wgrad: bad temporal taps (B=%d frames=%d taps=%d cin=%d N=%d)", d->B, d->t_frames, d->t_taps,
                      d->t_cin, d->N);
        return SOS_EINVAL;
    }
    // 1x1 kernels (Linear layers, LSTM projections): the pixel arrays of G and X are congruent, so the batch
    // is one long row -- full 256-pixel tiles instead of one ragged tile per (short) image.
    sos_wgrad_desc flat = *d;
    if (!temporal && d->kh == 1 && d->kw == 1 && d->stride == 1 && d->pad_top == 0 && d->pad_left == 0 && d->Hg == d->Hx && d->Wg == d->Wx) {
        const uint64_t npx = (uint64_t)d->B * d->Hg * d->Wg;
        if (npx * (uint64_t)d->g_cs * 2 < 0xfff00000ull && npx * (uint64_t)d->x_cs * 2 < 0xfff00000ull) {
            flat.B = 1; flat.Hg = flat.Hx = 1; flat.Wg = flat.Wx = (int)npx;
        }
    }
    const bool is_flat = !temporal && flat.B == 1 && flat.Hg == 1 && d->kh == 1 && d->kw == 1 && d->stride == 1 && d->pad_top == 0 &&
                         d->pad_left == 0 && d->Hg == d->Hx && d->Wg == d->Wx;
    d = &flat;
    if (is_flat && d->M >= 128 && d->N >= 128 && !getenv("SOS_WGRAD_NO_GEMM")) {
        WgGemmParams q;
        q.g = (const bf16_t*)d->g + d->g_off; q.x = (const bf16_t*)d->x + d->x_off; q.partial = d->partial;
        q.K = d->Wg; q.g_cs = d->g_cs; q.x_cs = d->x_cs;
        q.Mp = (d->M + 31) / 32 * 32; q.Np = (d->N + 31) / 32 * 32;
        q.tiles_n = (q.Np + 127) / 128;
        q.ntiles = ((q.Mp + 127) / 128) * q.tiles_n;
        const int cap = d->ksplit > 0 ? d->ksplit : wg_max_split(d);
        q.ksplit = wg_gemm_split(q.ntiles, q.K, (int64_t)q.Mp * q.Np * 4, cap);
        { const char* e = getenv("SOS_WGG_SPLIT"); if (e && atoi(e) >= 1 && atoi(e) <= cap) q.ksplit = atoi(e); }
        q.kper = ((q.K + q.ksplit - 1) / q.ksplit + WGG_KT - 1) / WGG_KT * WGG_KT;
        hipStream_t s = (hipStream_t)stream;
        static sos_device_once gemm_once;
        (void)sos_per_device_once(gemm_once, [] {
            (void)hipFuncSetAttribute((const void*)wgrad_gemm_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * WGG_BUF);
            return (int)SOS_OK;
        });
        if (what & 1) {
            hipLaunchKernelGGL(wgrad_gemm_kernel, dim3((unsigned)(q.ntiles * q.ksplit)), dim3(256), 2 * WGG_BUF, s, q);
            int rc = sos_check_launch("sos_conv2d_wgrad(gemm)");
            if (rc) return rc;
        }
        if (!(what & 2)) return SOS_OK;
        const long long total = (long long)d->M * d->N;
        long long gb = (total + 63) / 64;
        if (gb > 8192) gb = 8192;
        hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)gb), dim3(256), 0, s, d->partial, q.ksplit, 1, d->M, d->N,
                           q.Mp, q.Np, d->dw, d->accumulate, d->scale, d->scale_dev);
        return sos_check_launch("sos_conv2d_wgrad(reduce)");
    }
    int m16 = 0, n16 = 0;
    if (wg_thin_shape(d, is_flat, &m16, &n16)) {
        WgThinParams q;
        q.g = (const bf16_t*)d->g + d->g_off; q.x = (const bf16_t*)d->x + d->x_off; q.partial = d->partial;
        q.K = d->Wg; q.g_cs = d->g_cs; q.x_cs = d->x_cs;
        q.Mp = (d->M + 31) / 32 * 32; q.Np = (d->N + 31) / 32 * 32;
        const int ksplit = wg_thin_split(d, m16 + n16);
        q.kper = ((q.K + ksplit * 4 - 1) / (ksplit * 4) + 31) / 32 * 32;
        const size_t lds = (size_t)4 * WGT_NBUF * (m16 + n16) * 1024;
        hipStream_t s = (hipStream_t)stream;
        static sos_device_once thin_once;
#define SOS_WGT_ALL(F) F(1, 1) F(2, 1) F(3, 1) F(4, 1) F(6, 1) F(1, 2) F(1, 3) F(1, 4) F(1, 6)
#define SOS_WGT_ATTR(A, B) (void)hipFuncSetAttribute((const void*)wgrad_thin_kernel<A, B>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
#define SOS_WGT_CASE(A, B) if (m16 == A && n16 == B) hipLaunchKernelGGL((wgrad_thin_kernel<A, B>), dim3((unsigned)ksplit), dim3(256), lds, s, q);
        (void)sos_per_device_once(thin_once, [] { SOS_WGT_ALL(SOS_WGT_ATTR) return (int)SOS_OK; });
        if (what & 1) {
            SOS_WGT_ALL(SOS_WGT_CASE)
            int rc = sos_check_launch("sos_conv2d_wgrad(thin)");
            if (rc) return rc;
        }
#undef SOS_WGT_CASE
#undef SOS_WGT_ATTR
#undef SOS_WGT_ALL
        if (!(what & 2)) return SOS_OK;
        const long long total = (long long)d->M * d->N;
        long long gb = (total + 63) / 64;
        if (gb > 8192) gb = 8192;
        hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)gb), dim3(256), 0, s, d->partial, ksplit, 1, d->M, d->N,
                           q.Mp, q.Np, d->dw, d->accumulate, d->scale, d->scale_dev);
        return sos_check_launch("sos_conv2d_wgrad(reduce)");
    }
    if (wg_thin_taps_shape(d, temporal)) {
        WgThinTapParams q;
        q.g = (const bf16_t*)d->g + d->g_off; q.x = (const bf16_t*)d->x + d->x_off; q.partial = d->partial;
        q.K = d->B * d->Hg * d->Wg; q.H = d->Hg; q.W = d->Wg; q.g_cs = d->g_cs; q.x_cs = d->x_cs;
        q.Mp = (d->M + 31) / 32 * 32; q.Np = (d->N + 31) / 32 * 32;
        q.dil = d->dil_h; q.pad = d->pad_top; q.reflect = d->pad_mode == SOS_PAD_REFLECT;
        sos_wgrad_desc one = *d;
        one.Wg = q.K;                                   // (wg_thin_split looks at the pixel count only)
        const int ksplit = wg_thin_split(&one, 4 + 5);
        q.kper = ((q.K + ksplit * 4 - 1) / (ksplit * 4) + 31) / 32 * 32;
        const size_t lds = (size_t)4 * WGT_NBUF * (4 + 5) * 1024;
        hipStream_t s = (hipStream_t)stream;
        static sos_device_once taps_once;
        (void)sos_per_device_once(taps_once, [] {
            (void)hipFuncSetAttribute((const void*)wgrad_thin_taps_kernel<4, 5>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            return (int)SOS_OK;
        });
        if (what & 1) {
            hipLaunchKernelGGL((wgrad_thin_taps_kernel<4, 5>), dim3((unsigned)ksplit), dim3(256), lds, s, q);
            int rc = sos_check_launch("sos_conv2d_wgrad(thin taps)");
            if (rc) return rc;
        }
        if (!(what & 2)) return SOS_OK;
        const long long total = (long long)d->M * d->N * d->kh;
        long long gb = (total + 63) / 64;
        if (gb > 8192) gb = 8192;
        hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)gb), dim3(256), 0, s, d->partial, ksplit, d->kh, d->M, d->N,
                           q.Mp, q.Np, d->dw, d->accumulate, d->scale, d->scale_dev);
        return sos_check_launch("sos_conv2d_wgrad(reduce)");
    }
    const WgCtx cx = wg_ctx(d, temporal, is_flat);
    if (cx.taps > WG_WAVES * WG_PAIRS) { sos_set_error("sos_conv2d_wgrad: kernel with %d taps per row not supported", cx.taps); return SOS_ENOSPC; }
    if ((uint64_t)d->Hg * d->Wg * d->g_cs * 2 >= 0xffffff00ull || (uint64_t)d->Hx * d->Wx * d->x_cs * 2 >= 0xffffff00ull) {
        sos_set_error("sos_conv2d_wgrad: one image of an operand exceeds 4 GB");
        return SOS_ENOSPC;
    }
    WgPlan plan;
    int rc = wg_choose_plan(d, cx, &plan);
    if (rc) return rc;
    return wg_launch(d, cx, plan, (hipStream_t)stream, what);
}
extern "C" int sos_conv2d_wgrad(const sos_wgrad_desc* d, sos_stream_t stream) { return wgrad_impl(d, stream, 3); }
extern "C" int sos_conv2d_wgrad_partial(const sos_wgrad_desc* d, sos_stream_t stream) { return wgrad_impl(d, stream, 1); }
extern "C" int sos_conv2d_wgrad_reduce(const sos_wgrad_desc* d, sos_stream_t stream) { return wgrad_impl(d, stream, 2); }

// ---- measured plans (round 4).  sos_wgrad_tune times, for the SHAPE of `d`, every channel tile (MT x NTB, one or two workgroups
// per CU) with the cost model's six cheapest pixel tiles each, in two stages like sos_conv2d_tune (all candidates over `iters`
// launches; the five fastest and the cost model's pick again over 8x the launches, alternating twice; the model's pick is kept
// unless the winner beats it by more than 2 %), and remembers the winner for later sos_conv2d_wgrad calls of that shape.
// Synchronises; overwrites d->dw and d->partial (pass accumulate = 0 and scratch buffers).  Call outside timed regions / captures.
extern "C" int sos_wgrad_tune(const sos_wgrad_desc* d, int iters, float* best_ms, sos_stream_t stream) {
    if (!d || !d->g || !d->x || !d->partial || !d->dw) { sos_set_error("sos_wgrad_tune: null pointer"); return SOS_EINVAL; }
    if (d->accumulate) { sos_set_error("sos_wgrad_tune: needs accumulate = 0 (the launches would pile up in dw)"); return SOS_EINVAL; }
    const bool temporal = d->t_taps > 1;
    sos_wgrad_desc flat = *d;
    if (!temporal && d->kh == 1 && d->kw == 1 && d->stride == 1 && d->pad_top == 0 && d->pad_left == 0 && d->Hg == d->Hx && d->Wg == d->Wx) {
        const uint64_t npx = (uint64_t)d->B * d->Hg * d->Wg;
        if (npx * (uint64_t)d->g_cs * 2 < 0xfff00000ull && npx * (uint64_t)d->x_cs * 2 < 0xfff00000ull) {
            flat.B = 1; flat.Hg = flat.Hx = 1; flat.Wg = flat.Wx = (int)npx;
        }
    }
    const bool is_flat = !temporal && flat.B == 1 && flat.Hg == 1 && d->kh == 1 && d->kw == 1 && d->stride == 1 && d->pad_top == 0 &&
                         d->pad_left == 0 && d->Hg == d->Hx && d->Wg == d->Wx;
    d = &flat;
    if (best_ms) *best_ms = -1.f;
    if (is_flat && d->M >= 128 && d->N >= 128 && !getenv("SOS_WGRAD_NO_GEMM")) return SOS_OK;     // the GEMM path has no plan
    { int a, b; if (wg_thin_shape(d, is_flat, &a, &b) || wg_thin_taps_shape(d, temporal)) return SOS_OK; }   // nor have the streaming paths
    const WgCtx cx = wg_ctx(d, temporal, is_flat);
    if (cx.taps > WG_WAVES * WG_PAIRS) return SOS_OK;
    const WgKey key = wg_key(d, cx);
    {
        std::lock_guard<std::mutex> lk(wg_mu());
        if (wg_table().count(key)) return SOS_OK;
    }
    WgPlan model;
    int rc = wg_model_plan(d, cx, &model);
    if (rc) return rc;
    std::vector<WgPlan> cands;
    cands.push_back(model);
    const int mt_hi = cx.use16 ? 1 : std::min(3, cx.ntiles_m), ntb_hi = cx.use16 ? 1 : std::min(4, cx.ntiles_n);
    for (int mt = 1; mt <= mt_hi; ++mt)
        for (int ntb = 1; ntb <= ntb_hi && ntb * cx.taps <= WG_WAVES * WG_PAIRS; ++ntb)
            for (int occ = 1; occ <= (cx.use16 ? 1 : 2); ++occ) {
                std::vector<WgPlan> tiles;
                for (int lnc = 0; lnc <= 6; ++lnc)
                    for (int lth = 0; lth + lnc <= 8; ++lth)
                        for (int kord = 0; kord < 2; ++kord) {
                            WgPlan pl;
                            if (wg_make_plan(d, cx, cx.use16 ? model.mt : mt, cx.use16 ? model.ntb : ntb, lnc, lth, kord, occ, &pl)) tiles.push_back(pl);
                        }
                std::sort(tiles.begin(), tiles.end(), [](const WgPlan& a, const WgPlan& b) { return a.cost < b.cost; });
                for (size_t t = 0; t < tiles.size() && t < (cx.use16 ? 12u : 6u); ++t) cands.push_back(tiles[t]);
            }
    if (iters < 1) iters = 1;
    hipStream_t s = (hipStream_t)stream;
    hipEvent_t e0, e1;
    if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) { sos_set_error("sos_wgrad_tune: hipEventCreate failed"); return SOS_ELAUNCH; }
    auto time_plan = [&](const WgPlan& pl, const int reps, float* ms_out) {
        int r = wg_launch(d, cx, pl, s);                         // warm-up
        if (r) return r;
        (void)hipEventRecord(e0, s);
        for (int k = 0; k < reps && !r; ++k) r = wg_launch(d, cx, pl, s);
        (void)hipEventRecord(e1, s);
        if (r || hipEventSynchronize(e1) != hipSuccess) return r ? r : (int)SOS_ELAUNCH;
        float ms = 0.f;
        (void)hipEventElapsedTime(&ms, e0, e1);
        *ms_out = ms / reps;
        return (int)SOS_OK;
    };
    std::vector<std::pair<float, int>> timed;
    for (size_t i = 0; i < cands.size() && !rc; ++i) {
        float ms = 0.f;
        rc = time_plan(cands[i], iters, &ms);
        timed.push_back({ms, (int)i});
    }
    int besti = 0;
    float best = 0.f;
    if (!rc) {
        std::sort(timed.begin(), timed.end());
        std::vector<int> fin;
        for (size_t t = 0; t < timed.size() && t < 5; ++t) fin.push_back(timed[t].second);
        if (std::find(fin.begin(), fin.end(), 0) == fin.end()) fin.push_back(0);
        std::vector<float> acc(fin.size(), 0.f);
        for (int round = 0; round < 2 && !rc; ++round)
            for (size_t t = 0; t < fin.size() && !rc; ++t) {
                float ms = 0.f;
                rc = time_plan(cands[fin[t]], iters * 8, &ms);
                acc[t] += 0.5f * ms;
            }
        if (!rc) {
            size_t bt = 0, t0 = 0;
            for (size_t t = 0; t < fin.size(); ++t) { if (acc[t] < acc[bt]) bt = t; if (fin[t] == 0) t0 = t; }
            if (acc[bt] > 0.98f * acc[t0]) bt = t0;
            besti = fin[bt]; best = acc[bt];
            if (getenv("SOS_CONV_TUNE_VERBOSE")) {
                const WgPlan& w = cands[besti];
                fprintf(stderr, "wgrad tune %dx%d k%dx%d s%d d%dx%d M%d N%d B%d: %zu candidates, pick MT=%d NTB=%d NC=%d TH=%d TW=%d order=%d occ=%d dbuf=%d %.4f ms (model's MT=%d NTB=%d NC=%d TH=%d TW=%d order=%d occ=%d: %.4f ms)\n",
                        d->Hg, d->Wg, d->kh, d->kw, d->stride, d->dil_h, d->dil_w, d->M, d->N, d->B, cands.size(), w.mt, w.ntb, w.nc, 1 << w.lth,
                        1 << w.ltw, w.kord, w.occ, w.db, best, model.mt, model.ntb, model.nc, 1 << model.lth, 1 << model.ltw, model.kord, model.occ, acc[t0]);
            }
        }
    }
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    if (rc) return rc;
    {
        std::lock_guard<std::mutex> lk(wg_mu());
        wg_table().emplace(key, cands[besti]);           // first writer wins
    }
    if (best_ms) *best_ms = best;
    return SOS_OK;
}

// Text table: header `sos_wgrad_tune 1 nkey 12`, then 12 shape ints + (MT, NTB, log2 NC, log2 TH, order, workgroups per CU) per line.
// An entry is accepted on load only if wg_make_plan() offers that plan for that shape with this build.
#define SOS_WGTUNE_FORMAT 1
extern "C" int sos_wgrad_tune_save(const char* path) {
    if (!path) { sos_set_error("sos_wgrad_tune_save: null path"); return SOS_EINVAL; }
    char tmp[4096];
    snprintf(tmp, sizeof(tmp), "%s.tmp.%ld", path, (long)getpid());
    FILE* f = fopen(tmp, "w");
    if (!f) { sos_set_error("sos_wgrad_tune_save: cannot open %s", tmp); return SOS_EINVAL; }
    fprintf(f, "sos_wgrad_tune %d nkey 12\n", SOS_WGTUNE_FORMAT);
    {
        std::lock_guard<std::mutex> lk(wg_mu());
        for (const auto& kv : wg_table()) {
            for (int i = 0; i < 12; ++i) fprintf(f, "%d ", kv.first.v[i]);
            int lnc = 0;
            while ((1 << lnc) < kv.second.nc) ++lnc;
            fprintf(f, "%d %d %d %d %d %d\n", kv.second.mt, kv.second.ntb, lnc, kv.second.lth, kv.second.kord, kv.second.occ);
        }
    }
    if (fclose(f) != 0 || rename(tmp, path) != 0) {
        remove(tmp);
        sos_set_error("sos_wgrad_tune_save: cannot write %s", path);
        return SOS_EINVAL;
    }
    return SOS_OK;
}

extern "C" int sos_wgrad_tune_load(const char* path) {
    if (!path) { sos_set_error("sos_wgrad_tune_load: null path"); return SOS_EINVAL; }
    FILE* f = fopen(path, "r");
    if (!f) return 0;
    int fmt = -1, nkey = -1;
    if (fscanf(f, " sos_wgrad_tune %d nkey %d", &fmt, &nkey) != 2 || fmt != SOS_WGTUNE_FORMAT || nkey != 12) {
        fclose(f);
        sos_set_error("sos_wgrad_tune_load: %s was written by another build (format %d)", path, fmt);
        return SOS_EINVAL;
    }
    int n = 0;
    for (;;) {
        WgKey k;
        bool ok = true;
        for (int i = 0; i < 12 && ok; ++i) ok = fscanf(f, "%d", &k.v[i]) == 1;
        int mt, ntb, lnc, lth, kord, occ;
        if (!ok || fscanf(f, "%d %d %d %d %d %d", &mt, &ntb, &lnc, &lth, &kord, &occ) != 6) break;
        sos_wgrad_desc d;
        memset(&d, 0, sizeof(d));
        d.Hg = k.v[0]; d.Wg = k.v[1]; d.kh = k.v[2]; d.kw = k.v[3]; d.stride = k.v[4]; d.dil_h = k.v[5]; d.dil_w = k.v[6];
        d.M = k.v[7]; d.N = k.v[8]; d.t_taps = k.v[10];
        if (d.Hg < 1 || d.Wg < 1 || d.kh < 1 || d.kw < 1 || d.stride < 1 || d.dil_h < 1 || d.dil_w < 1 || d.M < 1 || d.N < 1) continue;
        const WgCtx cx = wg_ctx(&d, k.v[10] > 1, k.v[11] != 0);
        if ((cx.use16 ? 1 : 0) != k.v[9] || cx.taps > WG_WAVES * WG_PAIRS) continue;
        if (mt < 1 || mt > 3 || mt > cx.ntiles_m || ntb < 1 || ntb > 4 || ntb > cx.ntiles_n || ntb * cx.taps > WG_WAVES * WG_PAIRS ||
            lnc < 0 || lnc > 6 || lth < 0 || kord < 0 || kord > 1 || occ < 1 || occ > 2)
            continue;
        WgPlan pl;
        if (!wg_make_plan(&d, cx, mt, ntb, lnc, lth, kord, occ, &pl)) continue;
        std::lock_guard<std::mutex> lk(wg_mu());
        wg_table()[k] = pl;
        ++n;
    }
    fclose(f);
    return n;
}
