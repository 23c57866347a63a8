// conv.hip -- implicit-GEMM Conv2d / ConvTranspose2d-phase / Linear on bf16 MFMA (gfx950).
//
// Replaces, for the whole model family, the cuDNN conv + separate BatchNorm + activation kernels
// behind Conv2dBlock / ConvBlock (M1/networks.py:28-51, M2/networks.py:28-51), DownConvBlock
// (M2/networks.py:97-117: ReflectionPad2d + valid conv + BN + PReLU), the four output-parity
// phases of UpConvBlock's ConvTranspose2d(k3,s2,p1,output_padding=1) (M2/networks.py:120-149)
// and every nn.Linear / LSTM input projection (they are 1x1 convs).
//
// Design (MI355X): activations are NHWC bf16 so a pixel's channels are one contiguous run.
// A workgroup (4 waves, one per SIMD) owns 256 output pixels x (NT*32) output channels.  The 256
// pixels are a TH x TW grid in *dilation-strided* coordinates (optionally NC adjacent residue
// classes), so that all kh*kw taps of a dilated conv read from ONE (TH+kh-1) x (TW+kw-1) input
// patch: the patch is fetched from HBM/L2 once per channel chunk into LDS (halo amplification
// ~1.5x instead of kh*kw x), and the tap loop only streams the [cout][cin] weight slab of the tap
// (double buffered in LDS).  MFMA: v_mfma_f32_32x32x16_bf16 with the weight slab as the row
// operand and pixels as the column operand, so each lane ends up with 4 consecutive output
// channels of one pixel per accumulator group -> 8-byte bf16 stores in the fused epilogue
// (folded BN scale/shift or bias, ReLU / PReLU / Sigmoid).  LDS rows are padded by 16 B so the
// 16-lane ds_read_b128 groups hit distinct banks.
#include "sos_common.h"
#include <stdlib.h>
#include <string.h>
#include <unistd.h>
#include <algorithm>
#include <array>
#include <map>
#include <type_traits>
#include <vector>

typedef sos_half_t bf16x8 __attribute__((ext_vector_type(8)));   // 8 storage-type (bf16, or fp16 in the SOS_F16 build) MFMA operands
typedef float f32x16 __attribute__((ext_vector_type(16)));

// Ablation switches (SOS_CONV_DBG bit mask: 1 patch staging, 2 tap loop, 4 epilogue, 8 weight-slab
// loads, 16 tap barrier, 64 re-read one slab) exist only in `make ABLATE=1` builds: as run-time
// branches they cost the production tap loop a dozen scalar branches and ~40 register moves per tap.
#ifdef SOS_ABLATE
#define CDBG(bit) (p.dbg & (bit))
#else
#define CDBG(bit) 0
#endif

struct ConvParams {
    const bf16_t* in;
    const bf16_t* wgt;
    void* out;
    const float* scale;
    const float* shift;
    const float* slope;
    const int* wgather;
    int B, H, W, Wl, in_cs, cin_off, cin, ktot, cps, seg_stride;
    int tT, tk, tpad;       // temporal taps: frames per clip, taps, temporal padding (1, 1, 0: plain 2-D conv)
    int kh, kw, cout, cout_pad, cout_store;
    int stride, dh, dw, pad_t, pad_l, pad_mode;
    int Ho, Wo;
    int out_dtype, c_off, act, accum;
    float* stats;           // optional fused BatchNorm partial sums [tile][2][stats_c]
    int stats_c;
    // fused INPUT BatchNorm + ReLU (sos_conv_desc.in_scale / in_shift): every value read from the input image becomes
    // max(x * in_scale[c] + in_shift[c], 0) on its way into LDS (c counted from cin_off); padding stays zero
    const float* in_scale;
    const float* in_shift;
    // ragged batches (variable-length clips stored in buffers of the batch's maximum width): per-image logical input
    // width / valid output width, and the per-image stride of the column-gather table
    const int* wl_tab;
    const int* wo_tab;
    int wg_stride;
    long long sb, sh, sw, sc, third;
    // tiling
    int NC, TH, TW, PH, PW;  // pixel tile: NC residue classes x TH x TW strided pixels (<= 256 slots; any integers since round 3)
    int nchunks;            // cin / (16*KS)
    int npix;               // NC*PH*PW
    int tiles_h, tiles_w, ngw;
    int nblk;               // grid.x
    int lmap;               // lane -> pixel relabelling inside a 32-pixel column tile (0: natural order)
    int dbg;                // SOS_CONV_DBG ablation mask (0 in production)
    // reflection-pad fold of the output (sos_conv_desc.fold_*): output pixel (ho, wo) is cell (ho*fsy + foy, wo*fsx + fox) of
    // the padded domain; interior cells go to `out` as pixel (hp - fP, wp - fP) of a dense [B][fH][fW] tensor (pitch sw),
    // border cells to out2 ([B][fH + 2 fP][fW + 2 fP], pitch frow)
    void* out2;
    int fP, fH, fW, fsy, foy, fsx, fox, frow;
    long long fthird;
    // Divisions by tile / grid extents as multiplications (round 4): q = n / d == umulhi(n, mg) for mg = 2^32 / d + 1 whenever
    // n * d < 2^32 (mg == 0 stands for d == 1).  A 32-bit division by a run-time value is ~40 VALU (or SALU + VALU) instructions;
    // a tile's prologue / epilogue held ~15 of them per wave (slot -> (class, row, column) of the operand bases, the epilogue
    // rows and the store table, the patch pixel table, the block id decode) -- per-TILE work, but the few-tap layers' tiles
    // hold only 250-320 MFMAs per wave, and a wave's VALU issue competes with its SIMD partner's MFMA stream.
    unsigned mgTW, mgTH, mgPW, mgPH, mg_tiles_w, mg_ngw, mg_tiles_h, mg_dh, mg_kw;
};

__host__ __device__ __forceinline__ unsigned mg_of(int d) { return d <= 1 ? 0u : (unsigned)((1ull << 32) / (unsigned)d + 1ull); }
__host__ __device__ __forceinline__ int mg_div(int n, unsigned mg) {
    return mg ? (int)(((unsigned long long)(unsigned)n * mg) >> 32) : n;          // device: one v_mul_hi_u32 / s_mul_hi_u32
}
// slot m of a tile -> (class, row, column), as tile_decode() below, with the multiplications
__host__ __device__ __forceinline__ void tile_decode_mg(int m, const ConvParams& p, int& cls, int& i, int& j) {
    const int r = mg_div(m, p.mgTW);
    j = m - r * p.TW;
    cls = mg_div(r, p.mgTH);
    i = r - cls * p.TH;
}

// pixel slot m (0..255) of a tile -> (residue class, row, column); slots with cls >= NC are dead (tiles of NC x TH x TW < 256
// pixels: dilation 32 leaves 8 x 5.6 strided pixels per class -- 5 classes x 8 x 6 = 240 slots instead of 4 x 8 x 8 with a
// quarter of every MFMA column tile on padding).  Per-tile work only: the divisions never reach a tap loop.
__host__ __device__ __forceinline__ void tile_decode(int m, int TH, int TW, int& cls, int& i, int& j) {
    const int r = m / TW;
    j = m - r * TW;
    cls = r / TH;
    i = r - cls * TH;
}

__device__ __forceinline__ bf16x8 lds_frag(const char* p) {
    return __builtin_bit_cast(bf16x8, *(const uint4*)p);
}

typedef __attribute__((address_space(3))) void* lds_ptr_t;

// Byte offset (inside its image) of every pixel of the workgroup's input patch, or ~0 for a pixel that
// is zero padding.  Built once per tile; every channel chunk's staging pass reuses it.
__device__ __forceinline__ void build_pixel_table(const ConvParams& p, unsigned* tab, int tid, int hin0, int win0, int rw0,
                                                  const int Wl, const int* __restrict__ wgather) {
    const bool reflect = p.pad_mode == SOS_PAD_REFLECT;
    for (int pix = tid; pix < p.npix; pix += 256) {
        const int rr = mg_div(pix, p.mgPW), c = pix - rr * p.PW;
        int r = rr, cls = 0;
        if (p.NC > 1) { cls = mg_div(rr, p.mgPH); r = rr - cls * p.PH; }
        int h = hin0 + r * p.dh;
        int w = win0 + cls * p.stride + c * p.dw;
        bool ok = (rw0 + cls) < p.dw || cls == 0;
        if (reflect) { h = reflect_index(h, p.H); w = reflect_index(w, Wl); }
        ok = ok && h >= 0 && h < p.H && w >= 0 && w < Wl;
        unsigned off = 0xffffffffu;
        if (ok) {
            if (wgather) w = wgather[w];
            off = (unsigned)((h * p.W + w) * p.in_cs) * 2u;
        }
        tab[pix] = off;
    }
}

// Stage the input patch of one channel chunk with LDS-DMA (buffer_load_dwordx4 ... lds): the patch image
// is [npix][CPR + 1] 16-byte pieces (the last piece of a row is the bank-conflict pad), instruction i of
// a wave fills pieces [64 i, 64 i + 64).  A lane's source is table[pixel] + chunk offset; zero padding,
// the pad piece and lanes past the image get an out-of-range offset (the hardware writes zeros).  No data
// registers, no ds_write, and all of a wave's pieces are in flight at once.
// Which pieces a DMA instruction moves.  Piece-linear (rounds 1-3): instruction i fills pieces [64 i, 64 i + 64), so a lane's
// pixel is (64 i + lane) / RP -- a division and a remainder per (instruction, lane).  Pixel-aligned (round 4, when RP divides 64
// to within 8 lanes: every kernel variant with <= 4 k-steps per chunk and the 16-row kernel): instruction i fills the PPI = 64 / RP
// WHOLE pixels [PPI i, PPI i + PPI), so a lane's piece index and its pixel's offset inside the instruction are constants
// (lane % RP, lane / RP) and the pixel is PPI i + const; the 64 - PPI RP lanes at the end are switched off for the whole pass
// (their LDS slots belong to the next instruction's first pixel).  The patch image in LDS is the same [npix][RP] pieces.
template <int RP> struct StageMap {
    static constexpr int PPI = (64 / RP) * RP >= 56 ? 64 / RP : 0;          // 0: piece-linear
    static constexpr int LDS_STEP = PPI ? PPI * RP * 16 : 1024;              // LDS bytes from one instruction to the next
    int lp, lq;
    bool on;
    __device__ __forceinline__ StageMap(int lane) {
        lp = PPI ? lane / RP : 0;
        lq = PPI ? lane - lp * RP : 0;
        on = PPI ? lane < PPI * RP : true;
    }
    static __device__ __forceinline__ int ninstr(int npix) { return PPI ? (npix + PPI - 1) / PPI : (npix * RP + 63) >> 6; }
    __device__ __forceinline__ void at(int i, int lane, int& pix, int& q) const {
        if constexpr (PPI != 0) { pix = i * PPI + lp; q = lq; }
        else { const int L = i * 64 + lane; pix = L / RP; q = L - pix * RP; }
    }
    // may this lane write in instruction i?  (only the last instruction of a pass has lanes past the patch's last piece)
    __device__ __forceinline__ bool writes(int i, int lane, int npix) const {
        if constexpr (PPI != 0) return i * PPI + lp < npix;
        else return i * 64 + lane < npix * RP;
    }
};

// Stage the input patch of one channel chunk with LDS-DMA (buffer_load_dwordx4 ... lds): the patch image
// is [npix][CPR + 1] 16-byte pieces (the last piece of a row is the bank-conflict pad).  A lane's source is
// table[pixel] + chunk offset; zero padding,
// the pad piece and lanes past the image get an out-of-range offset (the hardware writes zeros).  No data
// registers, no ds_write, and all of a wave's pieces are in flight at once.
template <int CPR, bool PAD = true>
__device__ __forceinline__ void stage_patch_dma(const ConvParams& p, char* patch, unsigned tab_addr, int lane, int wave,
                                                __amdgpu_buffer_rsrc_t rsrc, unsigned cbytes, const int first = 0) {
    constexpr int RP = CPR + (PAD ? 1 : 0);
    typedef StageMap<RP> SM;
    const SM sm(lane);
    const int ninstr = SM::ninstr(p.npix);
    if (!sm.on) return;
    // Four DMA instructions per round: their four table entries are requested back to back and waited for ONCE (round 3: one
    // ds_read + s_waitcnt lgkmcnt(0) per instruction put ~11 serial LDS round trips per wave and chunk at the head of every
    // tile -- 20 % of a wave's life on the 3x3 / 7x1 layers, whose tiles hold 250-320 MFMAs per wave instead of 900).
    for (int i0 = first + wave; i0 < ninstr; i0 += 16) {
        unsigned ent[4], ta[4];
        int qq[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            int pix;
            sm.at(i0 + 4 * u, lane, pix, qq[u]);
            ta[u] = tab_addr + (unsigned)min(pix, p.npix - 1) * 4u;
        }
        // ONE asm statement: the four requests and their wait cannot be separated, and the early-clobber outputs cannot share a
        // register with an address that a later request still needs (as separate statements the compiler was free to copy or
        // spill an `ent` register between its ds_read and the wait -- ADVICE r3)
        asm volatile("ds_read_b32 %0, %4\n\tds_read_b32 %1, %5\n\tds_read_b32 %2, %6\n\tds_read_b32 %3, %7\n\ts_waitcnt lgkmcnt(0)"
                     : "=&v"(ent[0]), "=&v"(ent[1]), "=&v"(ent[2]), "=&v"(ent[3])
                     : "v"(ta[0]), "v"(ta[1]), "v"(ta[2]), "v"(ta[3])
                     : "memory");
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int i = i0 + 4 * u;
            const bool ok = qq[u] < CPR && ent[u] != 0xffffffffu;
            const unsigned voff = ok ? ent[u] + cbytes + (unsigned)qq[u] * 16u : 0xffffffffu;
            if (i < ninstr && sm.writes(i, lane, p.npix))
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_ptr_t)(patch + i * SM::LDS_STEP), 16, voff, 0, 0, 0);
        }
    }
}

// Round 4: the (pixel, piece) a lane moves in DMA instruction i -- hence its table entry, validity and source offset inside
// the image -- is the same for EVERY channel chunk of a tile; only the chunk's channel offset differs.  stage_prepare() resolves
// the first 16 * MAXR instructions of a workgroup's staging pass once per tile into per-lane offsets vb[r][u] (invalid lanes:
// an offset that stays beyond the buffer resource's range after any chunk offset < 64 KB is added), stage_issue() then costs
// one add per DMA instruction.  PMC per wave and tile before (ABLATE build, tools/probe/pmc_phases.sh): staging = 430 VALU +
// 332 SALU for 24 DMA instructions on the 96 -> 96 layer (2 chunks), 1 140 VALU + 940 SALU on 256 -> 256 3x3 (4 chunks) --
// a third of that layer's non-MFMA instructions.  Instructions past 16 * MAXR (patches > 64 KB) and temporal-tap layers
// (their chunk offset moves by whole frames) keep the table path.
#define SOS_STAGE_INVALID 0xfffe0000u
// vb[r][u]: instruction wave + 16 r + 4 u (all but the LAST instruction of the pass, whose lanes past the patch's last piece must
// not write -- the weight slab follows the patch: it gets its own offset register `vlast` and the only per-instruction lane mask)
template <int CPR, bool PAD, int MAXR>
__device__ __forceinline__ void stage_prepare(const ConvParams& p, unsigned tab_addr, int lane, int wave, unsigned (&vb)[MAXR][4],
                                              unsigned& vlast) {
    constexpr int RP = CPR + (PAD ? 1 : 0);
    typedef StageMap<RP> SM;
    const SM sm(lane);
    const int ninstr = SM::ninstr(p.npix);
    auto resolve = [&](const int i) -> unsigned {           // (one instruction; used for the last one)
        int pix, q;
        sm.at(i, lane, pix, q);
        unsigned ent;
        asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=&v"(ent) : "v"(tab_addr + (unsigned)min(pix, p.npix - 1) * 4u) : "memory");
        return (q < CPR && ent != 0xffffffffu && sm.writes(i, lane, p.npix)) ? ent + (unsigned)q * 16u : SOS_STAGE_INVALID;
    };
#pragma unroll
    for (int r = 0; r < MAXR; ++r) {
        const int i0 = wave + 16 * r;
        unsigned ent[4], ta[4];
        int qq[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            int pix;
            sm.at(i0 + 4 * u, lane, pix, qq[u]);
            ta[u] = tab_addr + (unsigned)min(pix, p.npix - 1) * 4u;
        }
        if (i0 < ninstr) {
            asm volatile("ds_read_b32 %0, %4\n\tds_read_b32 %1, %5\n\tds_read_b32 %2, %6\n\tds_read_b32 %3, %7\n\ts_waitcnt lgkmcnt(0)"
                         : "=&v"(ent[0]), "=&v"(ent[1]), "=&v"(ent[2]), "=&v"(ent[3])
                         : "v"(ta[0]), "v"(ta[1]), "v"(ta[2]), "v"(ta[3])
                         : "memory");
        } else {
            ent[0] = ent[1] = ent[2] = ent[3] = 0xffffffffu;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const bool ok = qq[u] < CPR && ent[u] != 0xffffffffu;
            vb[r][u] = ok ? ent[u] + (unsigned)qq[u] * 16u : SOS_STAGE_INVALID;
        }
    }
    vlast = SOS_STAGE_INVALID;
    if (ninstr <= 16 * MAXR && ((ninstr - 1) & 3) == wave) vlast = resolve(ninstr - 1);
}
template <int CPR, bool PAD, int MAXR>
__device__ __forceinline__ void stage_issue(const ConvParams& p, char* patch, unsigned tab_addr, int lane, int wave,
                                            __amdgpu_buffer_rsrc_t rsrc, unsigned cbytes, const unsigned (&vb)[MAXR][4], const unsigned vlast) {
    constexpr int RP = CPR + (PAD ? 1 : 0);
    typedef StageMap<RP> SM;
    const SM sm(lane);
    const int ninstr = SM::ninstr(p.npix);
    if (sm.on) {
#pragma unroll
        for (int r = 0; r < MAXR; ++r) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int i = wave + 16 * r + 4 * u;
                if (i < ninstr - 1) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_ptr_t)(patch + i * SM::LDS_STEP), 16, vb[r][u] + cbytes, 0, 0, 0);
            }
        }
    }
    if (ninstr > 16 * MAXR) stage_patch_dma<CPR, PAD>(p, patch, tab_addr, lane, wave, rsrc, cbytes, 16 * MAXR);
    else if (((ninstr - 1) & 3) == wave) {
        if (sm.on && sm.writes(ninstr - 1, lane, p.npix))
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_ptr_t)(patch + (ninstr - 1) * SM::LDS_STEP), 16, vlast + cbytes, 0, 0, 0);
    }
}

// Round 5 (VERDICT r4 #1c): patch staging THROUGH REGISTERS with the producer's training-mode BatchNorm + ReLU applied on the
// way -- the consumer reads the producer's RAW conv output and the separate bn_apply pass (read raw, write y: two tensor passes
// per block) disappears from the forward, the activated tensor from the tape.  Same (instruction, lane) -> (pixel, piece) map as
// stage_issue(): a lane's 16-byte piece always holds the same 8 channels of a chunk (StageMap: lq = lane % RP), so its 8 scale /
// shift values are loaded once per chunk; all of a wave's loads are issued before the first is used.  The arithmetic is
// bn_apply_kernel's (f32 fma, max, round to nearest even): the LDS image is bit-identical to staging the stored y.  Lanes whose
// source is padding / out of range (offset SOS_STAGE_INVALID) and the pad piece store zeros, like the DMA path's out-of-range lanes.
typedef unsigned sos_u32x4 __attribute__((ext_vector_type(4)));
template <int CPR, int MAXR>
__device__ __forceinline__ void stage_issue_bn(const ConvParams& p, char* patch, int lane, int wave, __amdgpu_buffer_rsrc_t rsrc,
                                               unsigned cbytes, const unsigned (&vb)[MAXR][4], const unsigned vlast, const int chan0) {
    constexpr int RP = CPR + 1;
    typedef StageMap<RP> SM;
    static_assert(SM::PPI != 0, "fused input BatchNorm needs the pixel-aligned staging map");
    const SM sm(lane);
    const int ninstr = SM::ninstr(p.npix);
    if (!sm.on) return;
    float sc[8], sh[8];
    {
        const int c0 = chan0 + min(sm.lq, CPR - 1) * 8;
        const float4 a0 = *(const float4*)(p.in_scale + c0), a1 = *(const float4*)(p.in_scale + c0 + 4);
        const float4 b0 = *(const float4*)(p.in_shift + c0), b1 = *(const float4*)(p.in_shift + c0 + 4);
        sc[0] = a0.x; sc[1] = a0.y; sc[2] = a0.z; sc[3] = a0.w; sc[4] = a1.x; sc[5] = a1.y; sc[6] = a1.z; sc[7] = a1.w;
        sh[0] = b0.x; sh[1] = b0.y; sh[2] = b0.z; sh[3] = b0.w; sh[4] = b1.x; sh[5] = b1.y; sh[6] = b1.z; sh[7] = b1.w;
    }
    auto xform = [&](const sos_u32x4 v, const bool valid) -> uint4 {
        const unsigned w[4] = {v.x, v.y, v.z, v.w};
        unsigned o[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float y0 = fmaxf(fmaf(sos_lo2f(w[e]), sc[2 * e], sh[2 * e]), 0.f);
            const float y1 = fmaxf(fmaf(sos_hi2f(w[e]), sc[2 * e + 1], sh[2 * e + 1]), 0.f);
            o[e] = valid ? pack2bf(y0, y1) : 0u;
        }
        return make_uint4(o[0], o[1], o[2], o[3]);
    };
    sos_u32x4 v[MAXR][4];
#pragma unroll
    for (int r = 0; r < MAXR; ++r)
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int i = wave + 16 * r + 4 * u;
            v[r][u] = sos_u32x4{0u, 0u, 0u, 0u};
            if (i < ninstr - 1) v[r][u] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)(vb[r][u] + cbytes), 0, 0);
        }
    sos_u32x4 vl = sos_u32x4{0u, 0u, 0u, 0u};
    const bool last_mine = ((ninstr - 1) & 3) == wave && sm.writes(ninstr - 1, lane, p.npix);
    if (last_mine) vl = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)(vlast + cbytes), 0, 0);
    char* dst = patch + lane * 16;
#pragma unroll
    for (int r = 0; r < MAXR; ++r)
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int i = wave + 16 * r + 4 * u;
            if (i < ninstr - 1) *(uint4*)(dst + i * SM::LDS_STEP) = xform(v[r][u], vb[r][u] != SOS_STAGE_INVALID && sm.lq < CPR);
        }
    if (last_mine) *(uint4*)(dst + (ninstr - 1) * SM::LDS_STEP) = xform(vl, vlast != SOS_STAGE_INVALID && sm.lq < CPR);
}

#if __HIP_DEVICE_COMPILE__
// ---- cooperative store of the staged bf16 output tile ([256 pixels][OROW bytes] in LDS at smem, the lo
// plane of the hi|hi|lo mode behind it): consecutive lanes write consecutive 16-byte pieces of a pixel's channel
// run (and consecutive pixels of a tile row are adjacent in memory when dil_w == 1).  PPX = 16-byte pieces per pixel.
// TIGHT (round 6): nothing but the staged tile in LDS -- a pixel's output offset lives in the 16-byte PAD of its own row (never read
// as data) instead of a table behind the tile, and the statistics' partial sums go over the tile once it has been walked: 256 x 208
// = 53 248 B for three n-tiles, which is what lets a THIRD workgroup into the CU (a third of the 160 KB at the 1 280-byte
// allocation granule is 53 760 B; the table made it 54 272).  Single-plane (not hi|hi|lo) outputs only.
template <int PPX, int OROW, int SLOTS = 256, bool TIGHT = false>
__device__ __forceinline__ void store_staged_tile(const ConvParams& p, char* smem, const int tid, const int b, const int n0,
                                                  const int ho_base, const int wo_base, const int rw0, const bool x3,
                                                  const int Wo) {
    char* ost_hi = smem;
    char* ost_lo = smem + SLOTS * OROW;
    // element offset of every tile pixel inside its output image (-1: outside), SLOTS / 256 entries per thread
    int* otab_ = (int*)(smem + SLOTS * OROW * (x3 ? 2 : 1));
    struct OTab {
        char* smem; int* tab;
        __device__ __forceinline__ int& operator[](const int m) const {
            if constexpr (TIGHT) return *(int*)(smem + m * OROW + (OROW - 16));
            else return tab[m];
        }
    };
    const OTab otab{smem, otab_};
    for (int m = tid; m < SLOTS; m += 256) {
        int cls, i, j;
        tile_decode_mg(m, p, cls, i, j);
        const int ho = ho_base + i * p.dh;
        const int wo = wo_base + cls + j * p.dw;
        const bool ok = cls < p.NC && ho < p.Ho && wo < Wo && (cls == 0 || rw0 + cls < p.dw);
        int off = ho * (int)p.sh + wo * (int)p.sw;
        if (p.fP) {          // reflection-pad fold: interior cells of the padded domain -> `out`; border cells -> out2, coded as -2 - offset
            const int hp = ho * p.fsy + p.foy, wp = wo * p.fsx + p.fox;
            const bool inner = hp >= p.fP && hp < p.fP + p.fH && wp >= p.fP && wp < p.fP + p.fW;
            off = inner ? ((hp - p.fP) * p.fW + (wp - p.fP)) * (int)p.sw : -2 - (hp * (p.fW + 2 * p.fP) + wp) * p.frow;
        }
        otab[m] = ok ? off : -1;
    }
    __syncthreads();
    bf16_t* opm = (bf16_t*)p.out + (long long)b * p.sb + p.c_off + n0;
    // fold mode: the padded scratch tensor (border cells), channels from 0
    bf16_t* op2 = p.fP ? (bf16_t*)p.out2 + (long long)b * (p.fH + 2 * p.fP) * (p.fW + 2 * p.fP) * p.frow + n0 : nullptr;
    if (!x3 && !p.accum && ((p.cout_store - n0) & 7) == 0) {
        // ---- common case (whole 8-channel pieces, plain store), round 4: ONE walk over the staged tile does the cooperative
        // store AND the fused BatchNorm statistics.  Thread = (16-byte piece cg of a pixel's channel run, pixel lane pl): its
        // piece index and every address term but the pixel's are loop invariants, a pixel costs one table read, one LDS read,
        // one add and the store (the former idx = m * PPX + q walk divided by PPX and rebuilt both addresses every iteration,
        // and the statistics read the tile a second time: 626 / 1 156 VALU per wave and tile without / with statistics on the
        // 96-channel layer, tools/probe/pmc_phases.sh).  Consecutive lanes still write consecutive 16-byte pieces.
        constexpr int PL = 256 / PPX;
        const int cg = tid % PPX, pl = tid / PPX;
        const int npiece = min(PPX, (p.cout_store - n0) >> 3);
        const bool st = p.stats != nullptr;
        float s[8], q[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) { s[e] = 0.f; q[e] = 0.f; }
        if (pl < PL) {
            const char* src = ost_hi + cg * 16;
            const bool wr = cg < npiece;
            typedef unsigned u32x4nt __attribute__((ext_vector_type(4)));
#pragma unroll 4
            for (int m = pl; m < SLOTS; m += PL) {
                const int off = otab[m];
                if (off == -1) continue;
                const uint4 hv = *(const uint4*)(src + m * OROW);
                if (st && off >= 0) {
                    const unsigned hw[4] = {hv.x, hv.y, hv.z, hv.w};
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const float v0 = sos_lo2f(hw[i]), v1 = sos_hi2f(hw[i]);
                        s[2 * i] += v0; q[2 * i] = fmaf(v0, v0, q[2 * i]);
                        s[2 * i + 1] += v1; q[2 * i + 1] = fmaf(v1, v1, q[2 * i + 1]);
                    }
                }
                if (wr) {
                    // streamed output (read back only after the whole tensor is written): non-temporal store, +0.9 % on inference
                    bf16_t* dst = (off >= 0 ? opm + off : op2 + (-2 - off)) + cg * 8;
                    __builtin_nontemporal_store(__builtin_bit_cast(u32x4nt, hv), (u32x4nt*)dst);
                }
            }
        }
        if (st) {
            // fixed-order sum over the pixel lanes (exactly the separate statistics pass's order of round 3: lanes, then tiles)
            float* red = (float*)(smem + SLOTS * OROW + SLOTS * 4);
            // (tiles of more than 256 slots, round 6: the 16 KB of partial sums go OVER the staged tile once every thread has
            // walked it -- one more barrier per tile instead of 16 KB that would cost the CU its second workgroup)
            if constexpr (SLOTS > 256 || TIGHT) { __syncthreads(); red = (float*)smem; }
#pragma unroll
            for (int e = 0; e < 8; ++e) { red[tid * 16 + e] = s[e]; red[tid * 16 + 8 + e] = q[e]; }
            __syncthreads();
            for (int o = tid; o < 2 * PPX * 8; o += 256) {
                const int which = o / (PPX * 8), cc = o - which * (PPX * 8);
                const int g8 = cc >> 3, e = cc & 7;
                float acc = 0.f;
                for (int l = 0; l < PL; ++l) acc += red[(l * PPX + g8) * 16 + which * 8 + e];
                if (n0 + cc < p.stats_c) p.stats[((size_t)which * p.stats_c + n0 + cc) * p.nblk + blockIdx.x] = acc;   // [2][C][tiles]
            }
        }
        return;
    }
    if (p.stats) {
        // ---- fused BatchNorm statistics of this tile (of the bf16-rounded values, exactly what a separate pass over
        // the stored tensor would see): thread = (8-channel group, pixel lane), then a fixed-order sum over the lanes
        constexpr int PL = 256 / PPX;
        const int cg = tid % PPX, pl = tid / PPX;
        float s[8], q[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) { s[e] = 0.f; q[e] = 0.f; }
        if (pl < PL) {
            for (int m = pl; m < SLOTS; m += PL) {
                if (otab[m] < 0) continue;
                const uint4 hv = *(const uint4*)(ost_hi + m * OROW + cg * 16);
                uint4 lv = make_uint4(0u, 0u, 0u, 0u);
                if (x3) lv = *(const uint4*)(ost_lo + m * OROW + cg * 16);
                const unsigned hw[4] = {hv.x, hv.y, hv.z, hv.w}, lw[4] = {lv.x, lv.y, lv.z, lv.w};
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float v0 = sos_lo2f(hw[i]) + sos_lo2f(lw[i]);
                    const float v1 = sos_hi2f(hw[i]) + sos_hi2f(lw[i]);
                    s[2 * i] += v0; q[2 * i] = fmaf(v0, v0, q[2 * i]);
                    s[2 * i + 1] += v1; q[2 * i + 1] = fmaf(v1, v1, q[2 * i + 1]);
                }
            }
        }
        float* red = (float*)(smem + SLOTS * OROW * (x3 ? 2 : 1) + SLOTS * 4);
#pragma unroll
        for (int e = 0; e < 8; ++e) { red[tid * 16 + e] = s[e]; red[tid * 16 + 8 + e] = q[e]; }
        __syncthreads();
        for (int o = tid; o < 2 * PPX * 8; o += 256) {
            const int which = o / (PPX * 8), cc = o - which * (PPX * 8);
            const int g8 = cc >> 3, e = cc & 7;
            float acc = 0.f;
            for (int l = 0; l < PL; ++l) acc += red[(l * PPX + g8) * 16 + which * 8 + e];
            if (n0 + cc < p.stats_c) p.stats[((size_t)which * p.stats_c + n0 + cc) * p.nblk + blockIdx.x] = acc;   // [2][C][tiles]
        }
    }
    // ---- general cooperative store (hi|hi|lo planes, gradient fan-in, ragged channel tails): consecutive lanes write consecutive
    // 16-byte pieces of a pixel's channel run (and consecutive pixels of a tile row are adjacent in memory when dil_w == 1).
    // (round 4: the same thread = (piece, pixel lane) walk as the common case: no division per piece, the piece's channel test
    // hoisted out of the loop)
    constexpr int PLG = 256 / PPX;
    const int q = tid % PPX, plg = tid / PPX;
    const int co = n0 + q * 8;
    if (plg >= PLG || co >= p.cout_store) return;
#pragma unroll 2
    for (int m = plg; m < SLOTS; m += PLG) {
        const int off = otab[m];
        if (off == -1) continue;
        const bool border = off < 0;                  // fold mode: a border cell of the padded scratch tensor (plain store)
        bf16_t* op = border ? op2 : opm;
        const int o = (border ? -2 - off : off) + q * 8;
        const int ith = border ? (int)p.fthird : (int)p.third;
        uint4 hv = *(const uint4*)(ost_hi + m * OROW + q * 16);
        if (co + 8 <= p.cout_store) {
            uint4 lv = make_uint4(0u, 0u, 0u, 0u);
            if (x3) lv = *(const uint4*)(ost_lo + m * OROW + q * 16);
            if (p.accum && !border) {
                // gradient fan-in: new = old + this, re-split into hi (+ lo)
                const uint4 oh = *(const uint4*)(op + o);
                uint4 ol = make_uint4(0u, 0u, 0u, 0u);
                if (x3) ol = *(const uint4*)(op + o + 2 * ith);
                const unsigned nh[4] = {hv.x, hv.y, hv.z, hv.w}, nl[4] = {lv.x, lv.y, lv.z, lv.w};
                const unsigned ph[4] = {oh.x, oh.y, oh.z, oh.w}, pl[4] = {ol.x, ol.y, ol.z, ol.w};
                unsigned rh[4], rl[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float a0 = sos_lo2f(nh[e]) + sos_lo2f(nl[e]) +
                                     sos_lo2f(ph[e]) + sos_lo2f(pl[e]);
                    const float a1 = sos_hi2f(nh[e]) + sos_hi2f(nl[e]) +
                                     sos_hi2f(ph[e]) + sos_hi2f(pl[e]);
                    const bf16_t h0 = f2bf(a0), h1 = f2bf(a1);
                    rh[e] = (unsigned)h0 | ((unsigned)h1 << 16);
                    rl[e] = (unsigned)f2bf(a0 - bf2f(h0)) | ((unsigned)f2bf(a1 - bf2f(h1)) << 16);
                }
                hv = make_uint4(rh[0], rh[1], rh[2], rh[3]);
                lv = make_uint4(rl[0], rl[1], rl[2], rl[3]);
            }
            *(uint4*)(op + o) = hv;
            if (x3) {
                *(uint4*)(op + o + ith) = hv;
                *(uint4*)(op + o + 2 * ith) = lv;
            }
        } else {
            const bf16_t* hs = (const bf16_t*)(ost_hi + m * OROW + q * 16);
            const bf16_t* ls = (const bf16_t*)(ost_lo + m * OROW + q * 16);
            for (int e = 0; co + e < p.cout_store; ++e) {
                op[o + e] = hs[e];
                if (x3) { op[o + e + ith] = hs[e]; op[o + e + 2 * ith] = ls[e]; }
            }
        }
    }
}

#endif

// SB: single weight-slab buffer (an extra barrier per tap, but a third workgroup fits a CU's LDS)
// INBN: the input is a producer's RAW conv output and its BatchNorm + ReLU is applied while the patch is staged (stage_issue_bn)
// PT (round 6): 32-pixel column tiles per wave.  2: a wave owns 64 pixels x NT * 32 output channels (a 256-slot workgroup tile).
// 3: 96 pixels -- a 384-slot tile.  The tap loop's per-tap barrier among four waves that share their SIMDs with another
// workgroup costs ~11 % (profiles/r05_wreg_probe.txt), and the fragment reads another 6 %: with NT = 3 a wave now issues 6
// ds_read_b128 per 9 MFMAs instead of 5 per 6, a tap holds 1.5x the MFMAs per barrier and per weight-slab byte, and the per-tile
// prologue / staging / epilogue overheads spread over 1.5x the work.  tools/probe/occ_probe.hip (profiles/r06_occ_probe.txt, edge
// waste of the tile included): 32 x 12 tile at two k-steps per chunk 0.501 of 2.5 PF against 0.471 for the production 32 x 8 /
// three k-steps tile; 512-slot tiles (PT = 4) reach 0.54 on whole tiles but lose it on the 178-column image's edges (0.49-0.50)
// and need 250 registers; a static issue priority per workgroup (s_setprio by HW_ID slot / TG_ID parity) and a THIRD resident
// workgroup (two k-steps per chunk, 48 KB) both measured no gain (0.494-0.496 vs 0.497).
// W3 (round 6): the double-slab kernel compiled for THREE workgroups per CU (<= 168 registers) with the TIGHT epilogue layout.  The
// phase ablation of the 96 -> 96 layer (profiles/r06_pt3_ablation.txt) prices the tap loop alone at 0.99 ms and the whole kernel at
// 1.19: with two workgroups per CU only a third of a workgroup's staging + epilogue (0.31 ms stand-alone) hides under the
// other's MFMAs -- a lone wave per SIMD that stops at a barrier every tap leaves the matrix pipe idle.
template <int NT, int KS, bool SB, bool INBN = false, int PT = 2, bool W3 = false>
__global__ __launch_bounds__(256, (SB || W3) ? 3 : (PT > 2 ? 2 : 1)) void conv_mfma_kernel(ConvParams p) {
#if __HIP_DEVICE_COMPILE__     // buffer-resource builtins exist in the device pass only; the host pass needs just the stub
    constexpr int KC = 16 * KS;                 // channels per chunk
    constexpr int PSTRIDE = KC * 2 + 16;        // bytes per patch pixel row (padded)
    constexpr int BSTRIDE = KC * 2 + 16;        // bytes per weight row (padded)
    constexpr int CPR = 2 * KS;                 // 16-byte pieces per row
    constexpr int BROWS = NT * 32;
    constexpr int BPIECES = BROWS * CPR;
    constexpr int NBREG = (BPIECES + 255) / 256;
    // double-buffered variants fill the slab by LDS-DMA in whole 1 KB instructions: a buffer is rounded up to that
    constexpr int BBYTES = SB ? BROWS * BSTRIDE : (BROWS * (CPR + 1) + 63) / 64 * 1024;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* patch = smem;
    const int boff0 = p.npix * PSTRIDE;          // weight slab buffers follow the patch
    unsigned* pixtab = (unsigned*)(smem + boff0 + (SB ? 1 : 2) * BBYTES);   // then the per-pixel source offsets

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, lhi = lane >> 5;

    // ---- block -> (b, rh, ti, gw, tj): XCD-aware bijective remap so that the blocks an XCD
    // runs back to back are neighbouring tiles (shared halos / rows stay in that XCD's L2).
    int bid = blockIdx.x;
    {
        const int nx = 8, q = p.nblk / nx, r = p.nblk % nx;
        const int xcd = bid % nx, loc = bid / nx;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
    }
    int t = bid, tq;
    tq = mg_div(t, p.mg_tiles_w); const int tj = t - tq * p.tiles_w; t = tq;
    tq = mg_div(t, p.mg_ngw); const int gw = t - tq * p.ngw; t = tq;
    tq = mg_div(t, p.mg_tiles_h); const int ti = t - tq * p.tiles_h; t = tq;
    tq = mg_div(t, p.mg_dh); const int rh = t - tq * p.dh; t = tq;
    const int b = t;
    const int n0 = blockIdx.y * BROWS;

    const int TH = p.TH, TW = p.TW;
    const int rw0 = gw * p.NC;
    // first output row/col (class 0 of the group) of this tile, and the input coordinate of patch (0,0)
    const int ho_base = rh + ti * TH * p.dh;
    const int wo_base = rw0 + tj * TW * p.dw;
    const int hin0 = ho_base * p.stride - p.pad_t;
    const int win0 = wo_base * p.stride - p.pad_l;
    // ragged batch: this image's own logical input width / valid output width; a tile past its end has nothing to do
    int Wl = p.Wl, Wo = p.Wo;
    if (p.wl_tab) {
        Wl = p.wl_tab[b]; Wo = p.wo_tab[b];
        if (wo_base >= Wo) return;
    }
    const int* wgather = p.wgather ? p.wgather + (long long)b * p.wg_stride : nullptr;

    // Which of the 32 pixels of an MFMA column tile a lane owns.  ds_read_b128 is serviced in four NON-contiguous 16-lane
    // groups ({0-3,12-15,20-27}, {4-11,16-19,28-31} and the same + 32: MI355X_MICROARCH.md, LDS): with the natural order
    // (lane = pixel) a group mixes pixels 0-3,12-15 of one tile row with pixels 4-11 of the NEXT row (TW = 16), whose
    // patch offset (PW pixels further) lands 4 of the 16 lanes on busy banks -- SQ_LDS_BANK_CONFLICT was 29 % of the
    // LDS-active cycles of the 96->96 layer.  Handing every hardware group 16 CONSECUTIVE pixels of one row makes the
    // fragment reads conflict free for every tile at least 16 pixels wide; the epilogue uses the same relabelling.
    int lpix = l31;
    if (p.lmap) {            // p.lmap is chosen on the host (pick_lane_map) by counting the conflicts of each relabelling
        const bool g0 = l31 < 4 || (l31 >= 12 && l31 < 16) || (l31 >= 20 && l31 < 28);
        const int rank = g0 ? (l31 < 4 ? l31 : (l31 < 16 ? l31 - 8 : l31 - 12)) : (l31 < 12 ? l31 - 4 : (l31 < 20 ? l31 - 8 : l31 - 16));
        if (p.lmap == 1) lpix = (g0 ? 0 : 16) + rank;                              // a hardware group = 16 consecutive pixels
        else lpix = ((g0 ? 0 : 1) + 2 * (rank >> 3)) * 8 + (rank & 7);              // TW = 8: a group = tile rows r and r + 2
    }
    // ---- per-lane pixel operand base addresses (tap (0,0), k-step 0)
    int abase[PT];
#pragma unroll
    for (int mt = 0; mt < PT; ++mt) {
        const int m = wave * (32 * PT) + mt * 32 + lpix;
        int cls, i, j;
        tile_decode_mg(m, p, cls, i, j);
        if (cls >= p.NC) cls = i = j = 0;          // dead slot: reads a valid patch pixel, its column is never stored
        abase[mt] = ((cls * p.PH + i * p.stride) * p.PW + j * p.stride) * PSTRIDE + lhi * 16;
    }
    const int bfrag_off = l31 * BSTRIDE + lhi * 16;

    // ---- weight-slab staging: the piece each thread moves is tap invariant
    int bsrc[NBREG], bdst[NBREG];
    {
        const int rmax = p.cout_pad - 1 - n0;   // rows past cout_pad are never stored: clamp, no branch
#pragma unroll
        for (int u = 0; u < NBREG; ++u) {
            const int idx = min(tid + u * 256, BPIECES - 1);
            const int row = idx / CPR, q = idx - row * CPR;
            bsrc[u] = min(row, rmax) * p.ktot + q * 8;
            bdst[u] = row * BSTRIDE + q * 16;
        }
    }
    const long long tap_stride = (long long)p.cout_pad * p.ktot;
    // Weight slab by LDS-DMA (double-buffered variants): the slab image in LDS is piece-linear (row * RPB + piece, the
    // last piece of a row being the 16-byte pad), so one DMA instruction fills 1 KB = 64 consecutive pieces; the
    // per-lane source offsets are tap invariant, the tap / chunk enters as the scalar offset.  No data registers, no
    // ds_write_b128 (13 LDS cycles each): 96->96 5x5 1.29 -> 1.23 ms.  (SB variants keep the register path: their
    // single buffer can only be refilled behind a barrier, where a DMA's latency would be exposed.)
    constexpr int RPB = CPR + 1;
    constexpr int WQ = BROWS * RPB;
    constexpr int WINSTR = (WQ + 63) / 64;
    constexpr int WPW = (WINSTR + 3) / 4;
    const int wv = __builtin_amdgcn_readfirstlane(wave);
    unsigned wvoff[WPW];
    {
        const int rmax = p.cout_pad - 1 - n0;
#pragma unroll
        for (int u = 0; u < WPW; ++u) {
            const int q = (wv + 4 * u) * 64 + lane;
            const int row = q / RPB, c = q - row * RPB;
            // pad slots and the tail of the last instruction: an offset beyond the buffer resource -- the hardware writes zeros
            // (their LDS bytes are never read) and fetches NOTHING (round 6; they used to re-read a valid piece: a fifth / a
            // seventh of the slab stream's L2 -> LDS bytes at two / three k-steps per chunk, and the slabs are 7 of the 8 GB a
            // launch of the 96 -> 96 layer moves by LDS-DMA)
            wvoff[u] = (c < CPR && row < BROWS) ? (unsigned)(((n0 + min(row, rmax)) * p.ktot + c * 8) * 2) : 0xffffffffu;
        }
    }
    const __amdgpu_buffer_rsrc_t w_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        (void*)p.wgt, 0, (unsigned)((long long)p.kh * p.kw * p.cout_pad * p.ktot * 2), 0x00020000);
    auto dma_slab = [&](const int buf, const unsigned soff) {
#pragma unroll
        for (int u = 0; u < WPW; ++u) {
            const int i = wv + 4 * u;
            if (i < WINSTR)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(w_rsrc, (lds_ptr_t)(smem + boff0 + buf * BBYTES + i * 1024), 16, wvoff[u],
                                                         soff, 0, 0);
        }
    };

    f32x16 acc[PT][NT];
#pragma unroll
    for (int mt = 0; mt < PT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[mt][nt][e] = 0.f;

    const int ntaps = p.kh * p.kw;
    // lane t: patch byte offset of tap t (<= 64 taps; validated on the host) -- one v_readlane per tap instead of
    // scalar row/column bookkeeping
    const int tl_ = min(lane, ntaps - 1), ta_ = mg_div(tl_, p.mg_kw);
    const int tapoff = (ta_ * p.PW + (tl_ - ta_ * p.kw)) * PSTRIDE;
    // temporal taps: the buffer resource spans the image's whole CLIP (tT frames), a chunk of temporal tap dt reads frame
    // tfr + dt - tpad; frames before / behind the clip fall outside the resource's range and are read as zeros
    const int tfr = b % p.tT;
    const long long frame_elems = (long long)p.H * p.W * p.in_cs;
    const long long in_b = (long long)(b - tfr) * p.H * p.W;
    const __amdgpu_buffer_rsrc_t in_rsrc =
        __builtin_amdgcn_make_buffer_rsrc((void*)(p.in + in_b * p.in_cs), 0, (unsigned)(p.tT * frame_elems) * 2u, 0x00020000);
    build_pixel_table(p, pixtab, tid, hin0, win0, rw0, Wl, wgather);
    constexpr int STG_R = 4;                     // staging rounds resolved once per tile (64 DMA instructions = 64 KB of patch)
    unsigned stg[STG_R][4], stg_last;
    const bool stg_fast = p.tk == 1;             // (temporal taps move the chunk offset by whole frames: table path)
    __syncthreads();                             // the pixel table is complete
    if (stg_fast && !CDBG(1)) stage_prepare<CPR, true, STG_R>(p, (unsigned)(uintptr_t)pixtab, lane, __builtin_amdgcn_readfirstlane(wave), stg, stg_last);

    int seg = 0, cin_seg = 0;                    // chunk cc = cin_seg-th chunk of channel segment seg (no division per chunk)
    for (int cc = 0; cc < p.nchunks; ++cc) {
        if (cc) __syncthreads();   // everyone is done reading the previous chunk's patch / weight buffers
        // ---- stage the input patch of this channel chunk
        if (!CDBG(1)) {
            if (stg_fast) {
                const unsigned cb = (unsigned)((p.cin_off + seg * p.seg_stride + cin_seg * KC) * 2);
                if constexpr (INBN) stage_issue_bn<CPR, STG_R>(p, patch, lane, __builtin_amdgcn_readfirstlane(wave), in_rsrc, cb, stg, stg_last, cin_seg * KC);
                else stage_issue<CPR, true, STG_R>(p, patch, (unsigned)(uintptr_t)pixtab, lane, __builtin_amdgcn_readfirstlane(wave), in_rsrc, cb, stg, stg_last);
            } else {
                const long long cbase = (long long)p.cin_off + (long long)(seg / p.tk) * p.seg_stride + (long long)cin_seg * KC +
                                        (long long)(tfr + seg % p.tk - p.tpad) * frame_elems;
                stage_patch_dma<CPR>(p, patch, (unsigned)(uintptr_t)pixtab, lane, __builtin_amdgcn_readfirstlane(wave), in_rsrc,
                                     (unsigned)(cbase * 2));
            }
        }
        if (++cin_seg == p.cps) { cin_seg = 0; ++seg; }
        // ---- weight slab of tap 0 straight into buffer 0
        if constexpr (!SB) dma_slab(0, (unsigned)(cc * KC * 2));
        else {
            const bf16_t* wsrc = p.wgt + ((long long)n0 * p.ktot + (long long)cc * KC);
#pragma unroll
            for (int u = 0; u < NBREG; ++u) {
                const uint4 v = *(const uint4*)(wsrc + bsrc[u]);
                if ((u + 1) * 256 <= BPIECES || tid + u * 256 < BPIECES) *(uint4*)(smem + boff0 + bdst[u]) = v;
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // this wave's patch pieces have landed
        __syncthreads();

        // Fragment pipeline: operands of k-step kk+1 are read from LDS while the MFMAs of kk run;
        // the pixel operand of the NEXT tap's first k-step is read before the barrier (the patch
        // does not change inside a chunk), only the weight operand has to wait for it.  The two
        // fragment buffers alternate along the flat (tap, k-step) sequence; F is the buffer a tap
        // starts in -- a compile-time constant, so the rotation costs no register moves (for odd KS
        // the tap loop is unrolled by two).
        bf16x8 af[2][PT], wfr[2][NT];
#pragma unroll
        for (int mt = 0; mt < PT; ++mt) af[0][mt] = lds_frag(patch + abase[mt]);
        const bf16_t* wtap = p.wgt + ((long long)n0 * p.ktot + (long long)cc * KC);
        auto tap_body = [&](auto ftag, const int tap) {
            constexpr int F = decltype(ftag)::value;
            const int cur = SB ? 0 : (tap & 1);
            // prefetch the next tap's weight slab into registers (lands while the MFMAs run).  The
            // loads are unconditional (last tap re-reads itself) so that the registers stay VGPRs: a
            // conditionally-defined array is demoted to scratch and the loads become synchronous.
            // (named scalars, not an array: with the scheduling barriers below an array is left in
            // scratch memory, which makes the "prefetch" synchronous)
            uint4 br0, br1, br2, br3, br4, br5, br6, br7;
            constexpr bool WD = !SB;               // next tap's slab by LDS-DMA (lands while the MFMAs run)
            if constexpr (WD) { if (tap + 1 < ntaps && !CDBG(8)) dma_slab(cur ^ 1, (unsigned)(((long long)(tap + 1) * tap_stride + cc * KC) * 2)); }
            if (tap + 1 < ntaps && !CDBG(64)) wtap += tap_stride;
#define SOS_BLOAD(i) if constexpr (NBREG > i && !WD) { if (!CDBG(8)) br##i = *(const uint4*)(wtap + bsrc[i]); else br##i = make_uint4(0,0,0,0); }
            SOS_BLOAD(0) SOS_BLOAD(1) SOS_BLOAD(2) SOS_BLOAD(3) SOS_BLOAD(4) SOS_BLOAD(5) SOS_BLOAD(6) SOS_BLOAD(7)
#undef SOS_BLOAD
            // hipcc otherwise sinks these loads to the end of the tap (right before their use) and
            // hoists every ds_read to just before its MFMA; pin the software pipeline explicitly.
            __builtin_amdgcn_sched_barrier(0);

            // patch byte offset of this tap and of the next one (lane t of tapoff; the last tap repeats itself)
            const int toff = __builtin_amdgcn_readlane(tapoff, tap);
            const int toffn = __builtin_amdgcn_readlane(tapoff, min(tap + 1, ntaps - 1));
            const char* ap[PT];
#pragma unroll
            for (int mt = 0; mt < PT; ++mt) ap[mt] = patch + abase[mt] + toff;
            const char* bp = smem + boff0 + cur * BBYTES + bfrag_off;
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) wfr[F][nt] = lds_frag(bp + nt * 32 * BSTRIDE);
#pragma unroll
            for (int kk = 0; kk < KS; ++kk) {
                constexpr int dummy = 0; (void)dummy;
                const int cb = (F + kk) & 1, nb = cb ^ 1;
                if (kk + 1 < KS) {
#pragma unroll
                    for (int mt = 0; mt < PT; ++mt) af[nb][mt] = lds_frag(ap[mt] + (kk + 1) * 32);
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) wfr[nb][nt] = lds_frag(bp + nt * 32 * BSTRIDE + (kk + 1) * 32);
                } else {
#pragma unroll
                    for (int mt = 0; mt < PT; ++mt) af[nb][mt] = lds_frag(patch + abase[mt] + toffn);
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
#pragma unroll
                    for (int mt = 0; mt < PT; ++mt) acc[mt][nt] = SOS_MFMA_32x32x16(wfr[cb][nt], af[cb][mt], acc[mt][nt], 0, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            if (SB) __syncthreads();                      // every wave is done with the (single) slab buffer
#define SOS_BSTORE(i)                                                                      \
    if constexpr (NBREG > i && !WD) {                                                               \
        if (!CDBG(8) && ((i + 1) * 256 <= BPIECES || tid + i * 256 < BPIECES))              \
            *(uint4*)(smem + boff0 + (SB ? 0 : (cur ^ 1)) * BBYTES + bdst[i]) = br##i;       \
    }
            SOS_BSTORE(0) SOS_BSTORE(1) SOS_BSTORE(2) SOS_BSTORE(3) SOS_BSTORE(4) SOS_BSTORE(5) SOS_BSTORE(6) SOS_BSTORE(7)
#undef SOS_BSTORE
            if constexpr (WD) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the next tap's slab has landed
            if (!CDBG(16)) __syncthreads();
        };
        const int ntaps_run = CDBG(2) ? 0 : ntaps;
        if constexpr (KS % 2 == 0) {
            for (int tap = 0; tap < ntaps_run; ++tap) tap_body(std::integral_constant<int, 0>{}, tap);
        } else {
            int tap = 0;
            for (; tap + 1 < ntaps_run; tap += 2) {
                tap_body(std::integral_constant<int, 0>{}, tap);
                tap_body(std::integral_constant<int, 1>{}, tap + 1);
            }
            if (tap < ntaps_run) tap_body(std::integral_constant<int, 0>{}, tap);
        }
    }

    // ---- fused epilogue.  Accumulator layout of v_mfma_f32_32x32x*: column = lane&31 (pixel),
    // row = (reg&3) + 8*(reg>>2) + 4*(lane>>5) (output channel within the 32-row tile).
    const float slope = (p.act == SOS_ACT_PRELU && p.slope) ? p.slope[0] : 0.f;
    if (CDBG(4)) return;
    const bool staged = p.out_dtype != SOS_DT_F32 && p.sc == 1;
    const bool x3 = p.out_dtype == SOS_DT_BF16X3;
    constexpr int OROW = NT * 64 + 16;           // bytes per staged output pixel row
    constexpr int SLOTS = 128 * PT;              // pixel slots of the workgroup's tile
    char* ost_hi = smem;
    char* ost_lo = smem + SLOTS * OROW;
    int mrow[PT];
    long long obase[PT];
    bool pix_ok[PT];
#pragma unroll
    for (int mt = 0; mt < PT; ++mt) {
        const int m = wave * (32 * PT) + mt * 32 + lpix;
        int cls, i, j;
        tile_decode_mg(m, p, cls, i, j);
        const int ho = ho_base + i * p.dh;
        const int wo = wo_base + cls + j * p.dw;
        mrow[mt] = m * OROW;
        pix_ok[mt] = cls < p.NC && ho < p.Ho && wo < Wo && (cls == 0 || rw0 + cls < p.dw);
        obase[mt] = (long long)b * p.sb + (long long)ho * p.sh + (long long)wo * p.sw;
    }
    // Common case (bf16 NHWC output, ReLU / PReLU / linear): the activation is the branch-free
    // max(y,0) + sn*min(y,0) and the conversion is the packed hardware one -- ~4 VALU per value and no
    // scalar branches; the general path below handles sigmoid, f32 / strided and hi|hi|lo outputs.
    if (staged && !x3 && p.act == SOS_ACT_NONE && !p.scale) {
        // raw output (training-mode forward convs, data gradients): accumulators straight to bf16
        const bool partial = n0 + NT * 32 > p.cout;           // channels past cout are stored as zeros
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int col = nt * 32 + g * 8 + lhi * 4;
                const int co = n0 + col;
                if (co >= p.cout_store) continue;
#pragma unroll
                for (int mt = 0; mt < PT; ++mt) {
                    float v[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = acc[mt][nt][g * 4 + e];
                    if (partial) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = (co + e < p.cout) ? v[e] : 0.f;
                    }
                    *(uint2*)(ost_hi + mrow[mt] + col * 2) = make_uint2(pack2bf(v[0], v[1]), pack2bf(v[2], v[3]));
                }
            }
        }
    } else if (staged && !x3 && p.act != SOS_ACT_SIGMOID && p.scale) {
        const float sn = p.act == SOS_ACT_RELU ? 0.f : (p.act == SOS_ACT_PRELU ? slope : 1.f);
        const bool relu = p.act == SOS_ACT_RELU;               // (wave-uniform) fma + max instead of fma + min + max + fma: the inference
                                                               // path's blocks are all BatchNorm + ReLU, two VALU per value less
        const bool partial = n0 + NT * 32 > p.cout;           // channels past cout are stored as zeros
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int col = nt * 32 + g * 8 + lhi * 4;
                const int co = n0 + col;                      // 4 consecutive channels co..co+3
                if (co >= p.cout_store) continue;
                const float4 sc4 = *(const float4*)(p.scale + co);
                const float4 sh4 = *(const float4*)(p.shift + co);
                const float scv[4] = {sc4.x, sc4.y, sc4.z, sc4.w};
                const float shv[4] = {sh4.x, sh4.y, sh4.z, sh4.w};
#pragma unroll
                for (int mt = 0; mt < PT; ++mt) {
                    float v[4];
                    if (relu) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = fmaxf(fmaf(acc[mt][nt][g * 4 + e], scv[e], shv[e]), 0.f);
                    } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float y = fmaf(acc[mt][nt][g * 4 + e], scv[e], shv[e]);
                        v[e] = fmaf(sn, fminf(y, 0.f), fmaxf(y, 0.f));
                    }
                    }
                    if (partial) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = (co + e < p.cout) ? v[e] : 0.f;
                    }
                    *(uint2*)(ost_hi + mrow[mt] + col * 2) = make_uint2(pack2bf(v[0], v[1]), pack2bf(v[2], v[3]));
                }
            }
        }
    } else {
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int col = nt * 32 + g * 8 + lhi * 4;
            const int co = n0 + col;                          // 4 consecutive channels co..co+3
            if (co >= p.cout_store) continue;
            float4 sc4 = make_float4(1.f, 1.f, 1.f, 1.f), sh4 = make_float4(0.f, 0.f, 0.f, 0.f);
            if (p.scale) { sc4 = *(const float4*)(p.scale + co); sh4 = *(const float4*)(p.shift + co); }
            const float scv[4] = {sc4.x, sc4.y, sc4.z, sc4.w};
            const float shv[4] = {sh4.x, sh4.y, sh4.z, sh4.w};
#pragma unroll
            for (int mt = 0; mt < PT; ++mt) {
                float v[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float y = fmaf(acc[mt][nt][g * 4 + e], scv[e], shv[e]);
                    if (p.act == SOS_ACT_RELU) y = fmaxf(y, 0.f);
                    else if (p.act == SOS_ACT_PRELU) y = y >= 0.f ? y : slope * y;
                    else if (p.act == SOS_ACT_SIGMOID) y = 1.0f / (1.0f + expf(-y));
                    v[e] = (co + e < p.cout) ? y : 0.f;
                }
                if (p.out_dtype == SOS_DT_F32) {
                    if (!pix_ok[mt]) continue;
                    float* op = (float*)p.out;
                    const long long o = obase[mt] + (long long)(p.c_off + co) * p.sc;
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (co + e < p.cout_store) op[o + e * p.sc] = v[e];
                    continue;
                }
                bf16_t hi[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) hi[e] = f2bf(v[e]);
                if (staged) {
                    // transpose through LDS so that global stores are whole 16-byte channel runs
                    *(uint2*)(ost_hi + mrow[mt] + col * 2) =
                        make_uint2((unsigned)hi[0] | ((unsigned)hi[1] << 16), (unsigned)hi[2] | ((unsigned)hi[3] << 16));
                    if (x3) {
                        bf16_t lo[4];
#pragma unroll
                        for (int e = 0; e < 4; ++e) lo[e] = f2bf(v[e] - bf2f(hi[e]));
                        *(uint2*)(ost_lo + mrow[mt] + col * 2) =
                            make_uint2((unsigned)lo[0] | ((unsigned)lo[1] << 16), (unsigned)lo[2] | ((unsigned)lo[3] << 16));
                    }
                } else {
                    if (!pix_ok[mt]) continue;
                    bf16_t* op = (bf16_t*)p.out;
                    const long long o = obase[mt] + (long long)(p.c_off + co) * p.sc;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        if (co + e >= p.cout_store) continue;
                        op[o + e * p.sc] = hi[e];
                        if (x3) {
                            op[o + e * p.sc + p.third] = hi[e];
                            op[o + e * p.sc + 2 * p.third] = f2bf(v[e] - bf2f(hi[e]));
                        }
                    }
                }
            }
        }
    }
    }
    if (!staged) return;
    store_staged_tile<NT * 4, OROW, SLOTS, W3>(p, smem, tid, b, n0, ho_base, wo_base, rw0, x3, Wo);
#endif
}

// ---- 16-row variant for small output-channel counts (cout <= 16 or 33..48; the 48-channel context layers):
// v_mfma_f32_16x16x32_bf16, NT16 tiles of 16 output channels instead of tiles of 32 (48 channels pay for 48, not 64).
// K = 32 per MFMA while a tap contributes cin = 16 KS channels, so the contraction runs over the FLAT (tap, channel)
// sequence in windows of two taps (2 * cin is a multiple of 32 for every cin % 16 == 0): lane group g of K-block kb
// reads 8-channel group q = 4 kb + g of the window, i.e. tap q / (cin/8) and channels 8 (q % (cin/8)) -- both the
// weight fragment (from the window's [2 taps][rows][cin] slab) and the pixel fragment (patch shifted by THAT tap).
// These per-lane offsets are window invariant.  One barrier per TWO taps; an odd last tap gets a zero slab.
typedef float f32x4 __attribute__((ext_vector_type(4)));
#ifndef SOS_C16_UPFRONT
#define SOS_C16_UPFRONT 0
#endif
#ifndef SOS_C16_XPF
#define SOS_C16_XPF 0       // 1: the next window's first pixel fragments requested before the barrier that closes a window (experiment: inside the noise)
#endif

#if __HIP_DEVICE_COMPILE__
// ---- epilogue of the 16-row kernels: D[m = cout][n = pixel]: lane = pixel + 16 * (cout / 4), register = cout % 4
template <int NT16, int PT>
__device__ __forceinline__ void conv16_epilogue(const ConvParams& p, f32x4 (&acc)[PT][NT16], char* smem, const int tid, const int wave,
                                                const int l15, const int g, const int b, const int ho_base, const int wo_base,
                                                const int rw0, const int Wo) {
    constexpr int ROWS = NT16 * 16, OROW = NT16 * 32 + 16, SLOTS = 64 * PT;
    const float slope = (p.act == SOS_ACT_PRELU && p.slope) ? p.slope[0] : 0.f;
    const bool partial = ROWS > p.cout, sig = p.act == SOS_ACT_SIGMOID;
    const float sn = p.act == SOS_ACT_RELU ? 0.f : (p.act == SOS_ACT_PRELU ? slope : 1.f);   // max(y,0) + sn*min(y,0)
    const bool raw = !p.scale && p.act == SOS_ACT_NONE;          // training forward convs, data gradients
    const bool relu16 = p.act == SOS_ACT_RELU;
    const bool x3out = p.out_dtype == SOS_DT_BF16X3;
#pragma unroll
    for (int nt = 0; nt < NT16; ++nt) {
        const int co = nt * 16 + 4 * g;                   // 4 consecutive channels co..co+3
        if (co >= p.cout_store) continue;
        float4 sc4 = make_float4(1.f, 1.f, 1.f, 1.f), sh4 = make_float4(0.f, 0.f, 0.f, 0.f);
        if (p.scale) { sc4 = *(const float4*)(p.scale + co); sh4 = *(const float4*)(p.shift + co); }
        const float scv[4] = {sc4.x, sc4.y, sc4.z, sc4.w}, shv[4] = {sh4.x, sh4.y, sh4.z, sh4.w};
#pragma unroll
        for (int pt = 0; pt < PT; ++pt) {
            float v[4];
            if (raw) {
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = acc[pt][nt][e];
            } else if (relu16) {                 // (wave-uniform branches: one code version per activation, no per-value selects)
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = fmaxf(fmaf(acc[pt][nt][e], scv[e], shv[e]), 0.f);
            } else if (sig) {
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = 1.0f / (1.0f + expf(-fmaf(acc[pt][nt][e], scv[e], shv[e])));
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float y = fmaf(acc[pt][nt][e], scv[e], shv[e]);
                    v[e] = fmaf(sn, fminf(y, 0.f), fmaxf(y, 0.f));
                }
            }
            if (partial) {
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = (co + e < p.cout) ? v[e] : 0.f;
            }
            const int m = wave * (16 * PT) + pt * 16 + l15;
            const unsigned h01 = pack2bf(v[0], v[1]), h23 = pack2bf(v[2], v[3]);
            *(uint2*)(smem + m * OROW + co * 2) = make_uint2(h01, h23);
            if (x3out) {              // hi|hi|lo output: the low parts go to the second staging plane
                *(uint2*)(smem + SLOTS * OROW + m * OROW + co * 2) =
                    make_uint2(pack2bf(v[0] - sos_lo2f(h01), v[1] - sos_hi2f(h01)), pack2bf(v[2] - sos_lo2f(h23), v[3] - sos_hi2f(h23)));
            }
        }
    }
    store_staged_tile<NT16 * 2, OROW, SLOTS>(p, smem, tid, b, 0, ho_base, wo_base, rw0, x3out, Wo);
}
#endif

// MODE 0: two window-slab buffers, the next window's DMA issued at the start of a window (lands in ~600 cycles of MFMAs or
// is waited for); 1: ONE buffer refilled behind a barrier (three workgroups per CU cover each other's refill latency);
// 2 (round 3, an A/B switch only: SOS_CONV16_MODE=2): a RING OF THREE buffers, the DMA of window w + 2 issued at the start of
// window w, counted vmcnt waits, one barrier per window, two workgroups per CU (66 KB).  Written on the hypothesis that the
// per-window refill wait is what holds mode 1 at 52 % MFMA-pipe busy; MEASURED SLOWER (48 -> 48 5x5 at B = 64: 0.410 ms
// against 0.348 ms in mode 1 and 0.395 ms in mode 0; the ISA shows no wait inside the window): what the kernel needs is the
// third resident workgroup, i.e. more waves to cover the fragment reads' LDS latency (7 ds_read_b128 per 12 MFMAs keep the
// LDS pipe ~55 % busy), not a hidden refill.
// PT: 16-pixel column tiles per wave.  4: a 256-pixel workgroup (three per CU in MODE 1).  8 (round 4): a wave owns 128 pixels x
// all NT16 * 16 output channels -- NT16 + 8 fragment reads per 8 NT16 MFMAs instead of NT16 + 4 per 4 NT16 (48 channels: 11 per 24
// instead of 7 per 12 ds_read_b128), and a 512-pixel workgroup streams every window slab once for twice the pixels (half the
// L2 -> LDS weight traffic and half the barriers per pixel); 96 accumulator registers, two workgroups per CU.
template <int NT16, int KS, int MODE, int PT = 4>
__global__ __launch_bounds__(256, (MODE == 1 && PT == 4) ? 3 : 2) void conv16_kernel(ConvParams p) {
#if __HIP_DEVICE_COMPILE__
    constexpr int KC = 16 * KS, G8 = 2 * KS;     // channels / 8-channel groups per tap
    // unpadded row pitches: lanes 16..31 of a fragment read address the SAME 16 rows as lanes 0..15, 16 bytes further
    // (next 8-channel group); with pitches of 32 / 96 bytes every 16-lane ds_read_b128 group covers all 64 banks once
    constexpr int PSTRIDE = KC * 2, BSTRIDE = KC * 2, CPR = 2 * KS;
    constexpr int BW = KS;                        // K-blocks of 32 channels per two-tap window
    constexpr int ROWS = NT16 * 16;
    constexpr int TAPBYTES = ROWS * BSTRIDE;
    constexpr int TPIECES = ROWS * CPR, WPIECES = 2 * TPIECES;
    // the double-buffered variant fills the window slab by LDS-DMA (whole 1 KB instructions; the slab image is piece-linear)
    constexpr int WINSTR = (WPIECES + 63) / 64, WPW = (WINSTR + 3) / 4;
    constexpr bool SB = MODE == 1;
    constexpr int NBUF = MODE == 1 ? 1 : (MODE == 2 ? 3 : 2);
    static_assert(MODE != 2 || WPW <= 3, "ring mode: the counted vmcnt waits cover at most three DMA instructions per wave and window");
    constexpr int WBYTES = SB ? 2 * TAPBYTES : WINSTR * 1024;
    constexpr int OROW = NT16 * 32 + 16;          // bytes per staged output pixel row
    constexpr int SLOTS = 64 * PT;                // pixel slots of the workgroup's tile
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* patch = smem;
    const int boff0 = p.npix * PSTRIDE;
    unsigned* pixtab = (unsigned*)(smem + boff0 + NBUF * WBYTES);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, g = lane >> 4;

    int bid = blockIdx.x;
    {
        const int nx = 8, q = p.nblk / nx, r = p.nblk % nx;
        const int xcd = bid % nx, loc = bid / nx;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
    }
    int t = bid, tq;
    tq = mg_div(t, p.mg_tiles_w); const int tj = t - tq * p.tiles_w; t = tq;
    tq = mg_div(t, p.mg_ngw); const int gw = t - tq * p.ngw; t = tq;
    tq = mg_div(t, p.mg_tiles_h); const int ti = t - tq * p.tiles_h; t = tq;
    tq = mg_div(t, p.mg_dh); const int rh = t - tq * p.dh; t = tq;
    const int b = t;
    const int TH = p.TH, TW = p.TW;
    const int rw0 = gw * p.NC;
    const int ho_base = rh + ti * TH * p.dh, wo_base = rw0 + tj * TW * p.dw;
    const int hin0 = ho_base * p.stride - p.pad_t, win0 = wo_base * p.stride - p.pad_l;
    int Wl = p.Wl, Wo = p.Wo;                     // ragged batch: per-image widths (see conv_mfma_kernel)
    if (p.wl_tab) {
        Wl = p.wl_tab[b]; Wo = p.wo_tab[b];
        if (wo_base >= Wo) return;
    }
    const int* wgather = p.wgather ? p.wgather + (long long)b * p.wg_stride : nullptr;

    // per-lane pixel operand base (tap (0,0)) of the wave's PT 16-pixel column tiles
    int pbase[PT];
#pragma unroll
    for (int pt = 0; pt < PT; ++pt) {
        const int m = wave * (16 * PT) + pt * 16 + l15;
        int cls, i, j;
        tile_decode_mg(m, p, cls, i, j);
        if (cls >= p.NC) cls = i = j = 0;          // dead slot (see conv_mfma_kernel)
        pbase[pt] = ((cls * p.PH + i * p.stride) * p.PW + j * p.stride) * PSTRIDE;
    }
    // per-lane, window-invariant fragment offsets of the BW K-blocks
    int aoff[BW], coff[BW];
    bool tap1[BW];
#pragma unroll
    for (int kb = 0; kb < BW; ++kb) {
        const int q = 4 * kb + g, tl = q / G8, c8 = q - tl * G8;
        aoff[kb] = tl * TAPBYTES + l15 * BSTRIDE + c8 * 16;
        coff[kb] = c8 * 16;
        tap1[kb] = tl != 0;
    }
    const long long tap_stride = (long long)p.cout_pad * p.ktot;
    // LDS-DMA source offsets (bytes from p.wgt for window 0; a window adds 2 taps): piece (tap tl, row, c).  A tap
    // past the last one lies beyond the buffer resource, where the hardware writes zeros: the odd window's zero half.
    const int wv = __builtin_amdgcn_readfirstlane(wave);
    unsigned wvoff[WPW];
#pragma unroll
    for (int u = 0; u < WPW; ++u) {
        const int idx = min((wv + 4 * u) * 64 + lane, WPIECES - 1);
        const int tl = idx / TPIECES, rem = idx - tl * TPIECES;
        const int row = rem / CPR, c = rem - row * CPR;
        wvoff[u] = (unsigned)((tl * tap_stride + (long long)row * p.ktot + c * 8) * 2);
    }
    const __amdgpu_buffer_rsrc_t w_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        (void*)p.wgt, 0, (unsigned)((long long)p.kh * p.kw * tap_stride * 2), 0x00020000);
    // hi|hi|lo three-pass mode: the contraction also runs over p.nchunks = 3 channel SEGMENTS (activation thirds x the
    // weight's [hi|lo|hi] K ranges); segment `sg` shifts the weight source by sg * cin channels
    unsigned segk = 0;
    auto dma_window = [&](const int w, const int buf) {
        const unsigned woff = (unsigned)((long long)w * 2 * tap_stride * 2) + segk;
#pragma unroll
        for (int u = 0; u < WPW; ++u) {
            const int i = wv + 4 * u;
            if (i < WINSTR)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(w_rsrc, (lds_ptr_t)(smem + boff0 + buf * WBYTES + i * 1024), 16,
                                                         wvoff[u] + woff, 0, 0, 0);
        }
    };

    f32x4 acc[PT][NT16];
#pragma unroll
    for (int pt = 0; pt < PT; ++pt)
#pragma unroll
        for (int nt = 0; nt < NT16; ++nt) acc[pt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int ntaps = p.kh * p.kw, nwin = (ntaps + 1) >> 1;
    const int tl_ = min(lane, ntaps - 1), ta_ = mg_div(tl_, p.mg_kw);
    const int tapoff16 = (ta_ * p.PW + (tl_ - ta_ * p.kw)) * PSTRIDE;   // lane t: tap t (<= 64 taps)
    const long long in_b = (long long)b * p.H * p.W;
    const __amdgpu_buffer_rsrc_t in_rsrc =
        __builtin_amdgcn_make_buffer_rsrc((void*)(p.in + in_b * p.in_cs), 0, (unsigned)(p.H * p.W * p.in_cs) * 2u, 0x00020000);
    build_pixel_table(p, pixtab, tid, hin0, win0, rw0, Wl, wgather);
    for (int sg = 0; sg < p.nchunks; ++sg) {
    // (first segment: publishes the pixel table; later ones: every wave has passed the last window's barrier, i.e. is done
    // reading the patch and the slabs)
    // (round 4: resolving the staging offsets once for the three segments of the hi|hi|lo mode, as conv_mfma_kernel does for its
    // chunks, costs this kernel 15 VGPRs and 21 spilled SGPRs: inference in `mixed` 1 623 -> 1 591 utt/s -- not adopted)
    __syncthreads();
    segk = (unsigned)(sg * p.cin * 2);
    if (!CDBG(1)) stage_patch_dma<CPR, false>(p, patch, (unsigned)(uintptr_t)pixtab, lane, __builtin_amdgcn_readfirstlane(wave), in_rsrc,
                                (unsigned)((p.cin_off + sg * p.seg_stride) * 2));
    if (!CDBG(8)) dma_window(0, 0);
    if constexpr (MODE == 2) { if (nwin > 1 && !CDBG(8)) dma_window(1, 1); }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // this wave's patch pieces have landed
    __syncthreads();

    int ring = 0;                                         // MODE 2: buffer of window w (w mod 3 without the division)
#if SOS_C16_XPF
    // pixel fragments of the NEXT window's first K-block: requested before the barrier(s) that close a window (the patch does
    // not change inside a segment), so that behind the barrier only the three weight fragments are still to be read
    bf16x8 fbn[PT];
    {
        const int po = (tap1[0] ? __builtin_amdgcn_readlane(tapoff16, min(1, ntaps - 1)) : __builtin_amdgcn_readlane(tapoff16, 0)) + coff[0];
#pragma unroll
        for (int pt = 0; pt < PT; ++pt) fbn[pt] = lds_frag(patch + pbase[pt] + po);
    }
#endif
    for (int w = 0; w < nwin; ++w) {
        const int cur = MODE == 2 ? ring : (SB ? 0 : (w & 1));
        if constexpr (MODE == 0) { if (w + 1 < nwin && !CDBG(8)) dma_window(w + 1, cur ^ 1); }       // next window's slab (lands while the MFMAs run)
        // ring: window w + 2 into the buffer window w - 1 was read from (every wave has passed the barrier that closed it)
        if constexpr (MODE == 2) { if (w + 2 < nwin && !CDBG(8)) dma_window(w + 2, ring == 0 ? 2 : ring - 1); }
        __builtin_amdgcn_sched_barrier(0);
        // patch byte offsets of the window's two taps: lane t of tapoff16 (two v_readlane instead of four scalar divisions by
        // kw per window: SQ counters showed 2.7 SALU per MFMA in this kernel against 1.5 in the 32-row one)
        const int toff0 = __builtin_amdgcn_readlane(tapoff16, 2 * w);
        const int toff1 = __builtin_amdgcn_readlane(tapoff16, min(2 * w + 1, ntaps - 1));
        const char* slab = smem + boff0 + cur * WBYTES;
        bf16x8 fa[2][NT16], fb[2][PT];
        auto read_block = [&](const int kb, const int buf) {
#pragma unroll
            for (int nt = 0; nt < NT16; ++nt) fa[buf][nt] = lds_frag(slab + aoff[kb] + nt * 16 * BSTRIDE);
            const int po = (tap1[kb] ? toff1 : toff0) + coff[kb];
#pragma unroll
            for (int pt = 0; pt < PT; ++pt) fb[buf][pt] = lds_frag(patch + pbase[pt] + po);
        };
#if SOS_C16_UPFRONT
        // experiment (round 3): every fragment of the window's BW K-blocks requested up front (7 BW ds_read_b128 in flight), the
        // MFMAs of block kb wait only for their own operands (LDS returns in order)
        bf16x8 ga[BW][NT16], gb[BW][PT];
#pragma unroll
        for (int kb = 0; kb < BW; ++kb) {
#pragma unroll
            for (int nt = 0; nt < NT16; ++nt) ga[kb][nt] = lds_frag(slab + aoff[kb] + nt * 16 * BSTRIDE);
            const int po = (tap1[kb] ? toff1 : toff0) + coff[kb];
#pragma unroll
            for (int pt = 0; pt < PT; ++pt) gb[kb][pt] = lds_frag(patch + pbase[pt] + po);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int kb = 0; kb < BW; ++kb) {
#pragma unroll
            for (int nt = 0; nt < NT16; ++nt)
#pragma unroll
                for (int pt = 0; pt < PT; ++pt)
                    acc[pt][nt] = SOS_MFMA_16x16x32(ga[kb][nt], gb[kb][pt], acc[pt][nt], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
        (void)fa; (void)fb; (void)read_block;
#else
#if SOS_C16_XPF
#pragma unroll
        for (int nt = 0; nt < NT16; ++nt) fa[0][nt] = lds_frag(slab + aoff[0] + nt * 16 * BSTRIDE);
#else
        read_block(0, 0);
#endif
#pragma unroll
        for (int kb = 0; kb < BW; ++kb) {
            const int cb = kb & 1;
            if (kb + 1 < BW) read_block(kb + 1, cb ^ 1);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int nt = 0; nt < NT16; ++nt)
#pragma unroll
                for (int pt = 0; pt < PT; ++pt) {
#if SOS_C16_XPF
                    acc[pt][nt] = SOS_MFMA_16x16x32(fa[cb][nt], kb == 0 ? fbn[pt] : fb[cb][pt], acc[pt][nt], 0, 0, 0);
#else
                    acc[pt][nt] = SOS_MFMA_16x16x32(fa[cb][nt], fb[cb][pt], acc[pt][nt], 0, 0, 0);
#endif
                }
            __builtin_amdgcn_sched_barrier(0);
        }
#if SOS_C16_XPF
        if (w + 1 < nwin) {
            const int t0n = __builtin_amdgcn_readlane(tapoff16, 2 * w + 2);
            const int t1n = __builtin_amdgcn_readlane(tapoff16, min(2 * w + 3, ntaps - 1));
            const int po = (tap1[0] ? t1n : t0n) + coff[0];
#pragma unroll
            for (int pt = 0; pt < PT; ++pt) fbn[pt] = lds_frag(patch + pbase[pt] + po);
        }
#endif
#endif
        if constexpr (SB) {
            // single buffer: refilled behind a barrier; the DMA's latency is covered by the two other workgroups of the
            // CU, and the LDS pipe is spared the ds_write_b128s of a register path (48->48 5x5: 0.356 -> 0.340 ms)
            __syncthreads();
            if (w + 1 < nwin && !CDBG(8)) dma_window(w + 1, 0);
        }
        if constexpr (MODE == 2) {
            // window w + 1 must have landed; this wave's pieces of window w + 2 (issued above: instructions wv, wv + 4, ...
            // of WINSTR) may stay in flight
            const int mine = (w + 2 < nwin && !CDBG(8)) ? (WINSTR - wv + 3) / 4 : 0;
            if (mine >= 3) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
            else if (mine == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
            else if (mine == 1) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            ring = ring == 2 ? 0 : ring + 1;
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __syncthreads();
    }
    }

    if (CDBG(4)) return;
    conv16_epilogue<NT16, PT>(p, acc, smem, tid, wave, l15, g, b, ho_base, wo_base, rw0, Wo);
#endif
}

// (Round 5, built and removed -- profiles/r05_conv16_chunk_resident.txt: a 16-row kernel with CHUNK-RESIDENT weights, i.e. per
// 16-channel chunk the patch and the weights of ALL taps in LDS at once: same LDS, bytes and MFMAs per tile as conv16_kernel, 6
// instead of 26 barriers per tile.  48 -> 48 5x5 at B = 64: 0.400 ms against 0.318 ms; the one 52 KB refill per chunk behind a
// barrier is exposed where thirteen 9 KB refills interleave with the other two workgroups' MFMAs.)

// ------------------------------------------------------------------------------ host side
static const size_t LDS_LIMIT = 160 * 1024;

typedef void (*conv_kernel_t)(ConvParams);

template <int NT, int KS, bool SB, bool INBN = false, int PT = 2, bool W3 = false>
static int launch_one(const ConvParams& p, dim3 grid, size_t lds, hipStream_t stream) {
    static sos_device_once attr_once;           // one per instantiation
    conv_kernel_t k = conv_mfma_kernel<NT, KS, SB, INBN, PT, W3>;
    const int arc = sos_per_device_once(attr_once, [k] {
        hipError_t e = hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_LIMIT);
        if (e != hipSuccess) {
            sos_set_error("sos_conv2d_fwd: hipFuncSetAttribute failed: %s", hipGetErrorString(e));
            return (int)SOS_ELAUNCH;
        }
        return (int)SOS_OK;
    });
    if (arc) return arc;
    hipLaunchKernelGGL(k, grid, dim3(256), lds, stream, p);
    return sos_check_launch("sos_conv2d_fwd");
}

// ks: k-steps per channel chunk; + 100: single weight-slab buffer (NT <= 3, ks <= 4 only)
template <int NT>
static int launch_ks(int ks, const ConvParams& p, dim3 grid, size_t lds, hipStream_t stream) {
    if (p.in_scale) {
        // fused input BatchNorm: the 96-channel context layers' tilings only (three n-tiles, 2 or 3 k-steps per chunk, double slab)
        // (stage_issue_bn loads in_scale / in_shift per KC-channel chunk: cin must be whole chunks, or it reads past the
        // f32[cin] arrays -- ADVICE r5)
        if constexpr (NT == 3) {
            if (ks == 3 && p.cin % 48 == 0) return launch_one<3, 3, false, true>(p, grid, lds, stream);
            if (ks == 2 && p.cin % 32 == 0) return launch_one<3, 2, false, true>(p, grid, lds, stream);
        }
        sos_set_error("sos_conv2d_fwd: fused input BatchNorm (in_scale) is not built for this tiling (nt=%d ks=%d)", NT, ks);
        return SOS_EINVAL;
    }
    if constexpr (NT == 3) {                    // 384-slot tiles (ks + 300): a wave owns 96 pixels x 96 output channels
        switch (ks) {
            case 302: return launch_one<3, 2, false, false, 3>(p, grid, lds, stream);
            case 303: return launch_one<3, 3, false, false, 3>(p, grid, lds, stream);
            // ks + 200: double slab, three workgroups per CU (<= 168 registers, TIGHT epilogue layout)
            case 201: return launch_one<3, 1, false, false, 2, true>(p, grid, lds, stream);
            case 202: return launch_one<3, 2, false, false, 2, true>(p, grid, lds, stream);
        }
    }
    if constexpr (NT <= 3) {
        switch (ks) {
            case 101: return launch_one<NT, 1, true>(p, grid, lds, stream);
            case 102: return launch_one<NT, 2, true>(p, grid, lds, stream);
            case 103: return launch_one<NT, 3, true>(p, grid, lds, stream);
            case 104: return launch_one<NT, 4, true>(p, grid, lds, stream);
        }
    }
    switch (ks) {
        case 1: return launch_one<NT, 1, false>(p, grid, lds, stream);
        case 2: return launch_one<NT, 2, false>(p, grid, lds, stream);
        case 3: return launch_one<NT, 3, false>(p, grid, lds, stream);
        case 4: return launch_one<NT, 4, false>(p, grid, lds, stream);
        case 5: return launch_one<NT, 5, false>(p, grid, lds, stream);
        case 6: return launch_one<NT, 6, false>(p, grid, lds, stream);
        case 8: return launch_one<NT, 8, false>(p, grid, lds, stream);
    }
    sos_set_error("sos_conv2d_fwd: unsupported k-steps %d", ks);
    return SOS_EINVAL;
}

static size_t lds_bytes(int npix, int nt, int ks) {           // ks 100..199: single slab buffer; 200..: three per CU; 300..: 384-slot tile (double slab)
    if (ks >= 300) ks -= 300;
    if (ks >= 200) ks -= 200;
    const bool single = ks >= 100;
    if (single) ks -= 100;
    const size_t row = (size_t)ks * 32 + 16;
    const size_t slab = single ? (size_t)nt * 32 * row : ((size_t)nt * 32 * (2 * ks + 1) + 63) / 64 * 1024;   // BBYTES of the kernel
    return (size_t)npix * row + (single ? 1 : 2) * slab + (size_t)npix * 4;
}

// the 16-row kernel (conv16_kernel) handles: one bf16 channel segment of 16 or 48 channels, bf16 NHWC output,
// cout <= 16 or 33..48 (i.e. shapes where tiles of 32 output channels waste MFMA rows)
// contracted channel ranges: hi|hi|lo thirds x temporal taps
static inline int nseg_eff(const sos_conv_desc* d) { return d->in_nseg * (d->t_taps > 1 ? d->t_taps : 1); }

static int nt16_for(const sos_conv_desc* d) {
    // plain 16-bit in -> out, or the three-segment hi|hi|lo mode in -> out (the parity-precision detector of 'mixed')
    const bool plain = d->in_nseg == 1 && d->out_dtype == SOS_DT_BF16, x3 = d->in_nseg == 3 && d->out_dtype == SOS_DT_BF16X3;
    if (!(plain || x3) || d->t_taps > 1 || d->out_sc != 1 || (d->cin != 16 && d->cin != 48)) return 0;
    if (d->cout <= 16) return 1;
    if (d->cout > 32 && d->cout <= 48) return 3;
    return 0;
}
static size_t lds_bytes16(int npix, int nt16, int ks, int mode) {     // mode 0: two slab buffers, 1: one, 2: ring of three
    const size_t row = (size_t)ks * 32;            // unpadded pitches
    const size_t slab = mode == 1 ? (size_t)2 * nt16 * 16 * row : ((size_t)2 * nt16 * 16 * 2 * ks + 63) / 64 * 1024;   // WBYTES
    return (size_t)npix * row + (mode == 1 ? 1 : (mode == 2 ? 3 : 2)) * slab + (size_t)npix * 4;
}

// One tiling choice: NC residue classes x (1<<lth) x (1<<ltw) pixels, 16*ks channels per chunk (ks == 0: the
// 16-row kernel, whole cin in one chunk).
struct ConvCfg {
    int NC, lth, ltw, ks;     // lth / ltw < 16: log2 of the tile height / width; >= 16: the (non-power-of-two) size + 16 (tdim())
    double cost;
};

static inline int tdim(int l) { return l < 16 ? 1 << l : l - 16; }      // tile size of an encoded ConvCfg.lth / ltw
static inline int tenc(int v) {                                           // and back (powers of two keep their log2)
    for (int l = 0; l < 16; ++l)
        if ((1 << l) == v) return l;
    return v + 16;
}

static int nt_for(const sos_conv_desc* d) {
    // output-channel tiles per block: as many as fit (<= 4), balanced over the n-blocks
    const int ntiles = d->cout_pad / 32;
    const int nby = (ntiles + 3) / 4;
    return (ntiles + nby - 1) / nby;
}

// Enumerate every legal tiling with a cost estimate: MFMA work wasted on out-of-range pixels,
// patch traffic (halo amplification), per-chunk overheads, and a bonus when two workgroups fit a
// CU's LDS (their load / epilogue phases then overlap each other's MFMA phase).
static std::vector<ConvCfg> enumerate_cfgs(const sos_conv_desc* d) {
    std::vector<ConvCfg> out;
    const int nt = nt_for(d);
    const int Hc = (d->Ho + d->dil_h - 1) / d->dil_h, Wc = (d->Wo + d->dil_w - 1) / d->dil_w;
    const int taps = d->kh * d->kw;
    static const int kscand[] = {8, 6, 5, 4, 3, 2, 1};
    const int k16 = d->cin / 16;
    // 16-row kernel with 512-pixel workgroups (conv16_kernel<.., PT = 8>; ks -4: double, -5: single slab): the wave owns 128
    // pixels x all output channels; candidates are the tiles of <= 512 slots whose patch still lets two workgroups into a CU
    // (or one, for the tuner to reject)
    auto add_tile16x = [&](const int NC, const int TH, const int TW) {
        const int nt16 = nt16_for(d);
        if (!nt16 || NC * TH * TW > 512 || NC * TH * TW <= 256) return;
        const int PH = (TH - 1) * d->stride + d->kh, PW = (TW - 1) * d->stride + d->kw;
        const int npix = NC * PH * PW;
        const long long th = (Hc + TH - 1) / TH, tw = (Wc + TW - 1) / TW, ngw = (d->dil_w + NC - 1) / NC;
        const double blocks = (double)th * tw * ngw * d->dil_h;
        const size_t stage = (size_t)512 * (nt16 * 32 + 16) * (d->out_dtype == SOS_DT_BF16X3 ? 2 : 1) + 2048 + 16384;
        for (int mode = 0; mode < 2; ++mode) {
            const size_t lds = std::max(lds_bytes16(npix, nt16, k16, mode), stage);
            if (lds > LDS_LIMIT) continue;
            double per_block = 0.75 * nseg_eff(d) * (0.9 * 512.0 * taps * k16 + 3.0 * npix * k16 + 40.0 * (6 + taps / 2));
            if (lds > LDS_LIMIT / 2) per_block *= 1.3;
            out.push_back({NC, tenc(TH), tenc(TW), -4 - mode, blocks * per_block});
        }
    };
    auto add_tile = [&](const int NC, const int TH, const int TW) {
        const int PH = (TH - 1) * d->stride + d->kh, PW = (TW - 1) * d->stride + d->kw;
        const int npix = NC * PH * PW;
        const long long th = (Hc + TH - 1) / TH, tw = (Wc + TW - 1) / TW, ngw = (d->dil_w + NC - 1) / NC;
        const double blocks = (double)th * tw * ngw * d->dil_h;
        if (const int nt16 = nt16_for(d)) {               // 16-row kernel on the same pixel tiling (ks 0: double, -1: single slab)
            // (the ring-of-three schedule, mode 2, is never a candidate: measured slower, see conv16_kernel; SOS_CONV16_MODE=2 forces it)
            for (int mode = 0; mode < 2; ++mode) {
                const size_t lds = lds_bytes16(npix, nt16, k16, mode);
                if (lds > LDS_LIMIT) continue;
                double per_block = 0.75 * nseg_eff(d) * (256.0 * taps * k16 + 3.0 * npix * k16 + 40.0 * (6 + taps / 2));
                if (lds > LDS_LIMIT / 2) per_block *= 1.3;
                else if (lds <= LDS_LIMIT / 3) per_block *= 0.95;
                out.push_back({NC, tenc(TH), tenc(TW), -mode, blocks * per_block});
            }
        }
        for (int ks : kscand) {
            if (k16 % ks) continue;
            const size_t lds = lds_bytes(npix, nt, ks);
            if (lds > LDS_LIMIT) continue;
            const int nchunks = nseg_eff(d) * k16 / ks;
            double per_block = 256.0 * taps * nchunks * ks + 3.0 * npix * nchunks * ks + 40.0 * nchunks * (6 + taps);
            if (lds > LDS_LIMIT / 2) per_block *= 1.3;     // a lone workgroup per CU hides nothing
            out.push_back({NC, tenc(TH), tenc(TW), ks, blocks * per_block});
            // single slab buffer: worth it only when it lets a third workgroup into the CU
            const size_t lds1 = lds_bytes(npix, nt, ks + 100);
            // (three n-tiles: the instance sits at the 168-register limit of three waves per SIMD -- ks >= 3 spills, 65-342 registers:
            // 11 ms instead of 1.5 on the 96 -> 96 layer; never a candidate)
            if ((nt <= 2 ? ks <= 4 : (nt == 3 && ks <= 2)) && lds1 <= LDS_LIMIT / 3 && lds > LDS_LIMIT / 3)
                out.push_back({NC, tenc(TH), tenc(TW), ks + 100, blocks * per_block * 0.93});
            // 128 output channels per workgroup need ~400 registers and > 80 KB of LDS (one workgroup per CU): two
            // n-blocks of 64 stage the patch twice but fit two or three workgroups (ks + 2000)
            if (nt == 4) {
                const size_t lds2 = lds_bytes(npix, 2, ks);
                if (lds2 <= LDS_LIMIT / 2)
                    out.push_back({NC, tenc(TH), tenc(TW), ks + 2000, blocks * 2.0 * (128.0 * taps * nchunks * ks + 3.0 * npix * nchunks * ks + 40.0 * nchunks * (6 + taps)) * 0.9});
            }
        }
    };
    for (int lnc = 0; lnc <= 8; ++lnc) {
        const int NC = 1 << lnc;
        if (NC > 1 && (d->stride > 1 || NC > d->dil_w)) break;
        for (int lth = 0; lth + lnc <= 8; ++lth) add_tile(NC, 1 << lth, 1 << (8 - lnc - lth));
    }
    // three workgroups per CU for 96-channel-wide workgroups (round 6, conv_mfma_kernel<3, ks, false, false, 2, true>; ks + 200): every
    // 256-slot tile whose tap-loop LDS (patch + two slabs + table) AND staged output tile fit a third of the CU.
    // SOS_CONV_NO_W3=1 removes them (A/B).
    static const size_t LDS_THIRD = (LDS_LIMIT / 3) / 1280 * 1280;      // 53 760: LDS is allocated in 1 280-byte granules on gfx950
    static const char* no_w3 = getenv("SOS_CONV_NO_W3");
    if (!(no_w3 && atoi(no_w3)) && nt == 3 && nseg_eff(d) == 1 && d->out_dtype == SOS_DT_BF16 && d->out_sc == 1 && d->cout_store % 8 == 0 &&
        !d->accumulate) {
        const size_t base = out.size();
        for (size_t i = 0; i < base; ++i) {
            const ConvCfg c = out[i];
            if (c.ks != 1 && c.ks != 2) continue;
            const int TH = tdim(c.lth), TW = tdim(c.ltw);
            const int PH = (TH - 1) * d->stride + d->kh, PW = (TW - 1) * d->stride + d->kw;
            if (lds_bytes(c.NC * PH * PW, 3, c.ks) > LDS_THIRD || (size_t)256 * (3 * 64 + 16) > LDS_THIRD) continue;
            out.push_back({c.NC, c.lth, c.ltw, c.ks + 200, c.cost * 0.93});
        }
    }
    // 384-slot tiles (round 6, conv_mfma_kernel<3, ks, false, false, 3>; ks + 300): 96-channel-wide workgroups of plain 16-bit
    // stride-1 layers.  Every (classes, height) with the widest width that fits, plus the widths / heights that cut the strided
    // image into equal parts (178 columns: 12 x 15 instead of 16 x 12).  SOS_CONV_NO_PT3=1 removes them (A/B).
    static const char* no_pt3 = getenv("SOS_CONV_NO_PT3");
    if (!(no_pt3 && atoi(no_pt3)) && nt == 3 && d->stride == 1 && nseg_eff(d) == 1 && d->out_dtype == SOS_DT_BF16 && d->out_sc == 1 &&
        !d->wl_tab && d->cout_store % 8 == 0) {
        static const int ncs[] = {1, 2, 3, 4, 6, 8, 12, 16, 24, 32}, ths[] = {1, 2, 3, 4, 6, 8, 12, 16, 24, 32, 48, 64};
        std::vector<std::array<int, 3>> seen;
        auto add384 = [&](const int NC, const int TH, const int TW) {
            if (NC < 1 || TH < 1 || TW < 2 || NC * TH * TW > 384 || NC * TH * TW < 300 || TH > 64 || TW > 64) return;
            const std::array<int, 3> t = {NC, TH, TW};
            if (std::find(seen.begin(), seen.end(), t) != seen.end()) return;
            seen.push_back(t);
            const int PH = TH - 1 + d->kh, PW = TW - 1 + d->kw;
            const int npix = NC * PH * PW;
            const long long th = (Hc + TH - 1) / TH, tw = (Wc + TW - 1) / TW, ngw = (d->dil_w + NC - 1) / NC;
            const double blocks = (double)th * tw * ngw * d->dil_h;
            for (int ks : {2, 3}) {
                if (k16 % ks) continue;
                const size_t stage = (size_t)384 * (3 * 64 + 16) + 384 * 4;
                const size_t lds = std::max(lds_bytes(npix, 3, ks + 300), stage);
                if (lds > LDS_LIMIT / 2) continue;           // (one workgroup per CU: never better than the 256-slot tiles)
                const int nchunks = k16 / ks;
                const double per_block = 0.94 * 384.0 * taps * nchunks * ks + 3.0 * npix * nchunks * ks + 40.0 * nchunks * (6 + taps);
                out.push_back({NC, tenc(TH), tenc(TW), ks + 300, blocks * per_block});
            }
        };
        for (int NC : ncs) {
            if (NC > d->dil_w) break;
            const int ngw = (d->dil_w + NC - 1) / NC;
            if ((d->dil_w + ngw - 1) / ngw != NC) continue;      // a smaller class count covers the same groups
            for (int TH : ths) {
                if (TH > Hc && TH != ths[0]) { if (TH / 2 >= Hc) break; }
                const int THe = std::min(TH, Hc);
                const int TWmax = std::min(384 / (NC * THe), std::min(Wc, 64));
                add384(NC, THe, TWmax);
                if (TWmax >= 2) { const int parts = (Wc + TWmax - 1) / TWmax; add384(NC, THe, (Wc + parts - 1) / parts); }
                { const int parts = (Hc + THe - 1) / THe, THb = (Hc + parts - 1) / parts;
                  const int TWb = std::min(384 / (NC * THb), std::min(Wc, 64));
                  add384(NC, THb, TWb);
                  if (TWb >= 2) { const int pw = (Wc + TWb - 1) / TWb; add384(NC, THb, (Wc + pw - 1) / pw); } }
            }
        }
    }
    // 512-pixel workgroups of the 16-row kernel: OPT-IN (SOS_CONV16_512=1, or SOS_CONV16_FORCE512=1 of the tests).  Round 4
    // measured them slower on every 5x5 shape (48 -> 48 at B = 64: 0.396 vs 0.369 ms forced against the table's 256-pixel tile on
    // the same box; a re-tune over 48 candidates per shape kept the 256-pixel tiles for all but two B = 8 1x1 heads): the
    // kernel wants its third resident workgroup more than fewer fragment reads per MFMA.
    static const char* en512 = getenv("SOS_CONV16_512");
    static const char* f512e = getenv("SOS_CONV16_FORCE512");
    if (nt16_for(d) && d->stride == 1 && ((en512 && atoi(en512)) || (f512e && atoi(f512e)))) {
        for (int lnc = 0; lnc <= 6; ++lnc) {
            const int NC = 1 << lnc;
            if (NC > 1 && NC > d->dil_w) break;
            for (int lth = 1; lth + lnc <= 7; ++lth) {
                const int TH = 1 << lth, TWf = 512 / (NC * TH);
                // the full 512 slots and the next narrower even widths (a 16 x 32 tile's 69 KB patch + slab + table is 1 KB over
                // half a CU's LDS; 16 x 30 fits two workgroups)
                for (int TW = TWf; TW >= TWf - 4 && TW >= 4; TW -= 2) add_tile16x(NC, TH, TW);
            }
        }
    }
    // Non-power-of-two tiles (round 3) for images whose strided extent per residue class is small: dilation 32 leaves 8 x 5.6
    // strided pixels per class, the U-Net's padded-domain gradients at dilation 16 6 x 4.8 -- 8-wide power-of-two tiles put a
    // quarter to a half of every MFMA column tile on padding.  Candidates: the strided height / width cut into 1..3 equal
    // parts, as many whole residue classes as fit 256 slots.  SOS_CONV_NPOT=0 removes them (A/B).
    static const char* npot_env = getenv("SOS_CONV_NPOT");
    if (!(npot_env && atoi(npot_env) == 0) && d->stride == 1 && (Hc <= 48 || Wc <= 48)) {
        for (int ph = 1; ph <= 3; ++ph)
            for (int pw = 1; pw <= 3; ++pw) {
                const int TH = (Hc + ph - 1) / ph, TW = (Wc + pw - 1) / pw;
                if (TH < 1 || TW < 2 || TH * TW > 256 || TH > 64 || TW > 64) continue;
                int NC = 256 / (TH * TW);
                if (NC > d->dil_w) NC = d->dil_w;
                if (NC < 1) continue;
                // fewer classes when the last group of a row of classes would be mostly empty
                const int ngw = (d->dil_w + NC - 1) / NC;
                NC = (d->dil_w + ngw - 1) / ngw;
                const bool pow2 = (TH & (TH - 1)) == 0 && (TW & (TW - 1)) == 0 && (NC & (NC - 1)) == 0 && NC * TH * TW == 256;
                if (pow2) continue;                       // already in the list above
                if (NC * TH * TW < 160) continue;         // no better than a power-of-two tile with padding
                add_tile(NC, TH, TW);
            }
    }
    std::sort(out.begin(), out.end(), [](const ConvCfg& a, const ConvCfg& b) { return a.cost < b.cost; });
    return out;
}

struct ShapeKey {
    int v[19];
    bool operator<(const ShapeKey& o) const { return memcmp(v, o.v, sizeof(v)) < 0; }
};

static ShapeKey shape_key(const sos_conv_desc* d) {
    ShapeKey k;
    const int vals[19] = {d->B, d->H, d->W, d->Wl, d->cin, nseg_eff(d), d->cout_pad, d->kh, d->kw, d->stride, d->dil_h,
                          d->dil_w, d->Ho, d->Wo, d->out_dtype, d->out_sc == 1 ? 1 : 0, d->pad_mode, d->w_gather ? 1 : 0,
                          d->cout};
    memcpy(k.v, vals, sizeof(vals));
    return k;
}

// The tiling table: shape -> tiling.  The only mutable global of the library; every access goes through the two
// helpers below under one mutex (nn.DataParallel calls the entry points from one host thread per replica).
static std::mutex& tuned_mutex() {
    static std::mutex mu;
    return mu;
}
static std::map<ShapeKey, ConvCfg>& tuned_cache() {
    static std::map<ShapeKey, ConvCfg> m;
    return m;
}
static bool tuned_lookup(const ShapeKey& k, ConvCfg* out) {
    std::lock_guard<std::mutex> g(tuned_mutex());
    auto it = tuned_cache().find(k);
    if (it == tuned_cache().end()) return false;
    *out = it->second;
    return true;
}
static std::vector<ConvCfg> enumerate_cfgs(const sos_conv_desc* d);
// A shape the table does not hold (another batch size, another clip length, a ragged batch's maxima) borrows the tiling
// of the SAME LAYER measured at another geometry: every key field equal except B, H, W, Wl, Ho, Wo; among those the
// entry with the same H / Ho and the most pixels wins (the BASELINE batch), provided the tiling is legal for the new
// shape.  Depends only on the (static) table: deterministic across processes.  The result is stored under the new key.
static bool tuned_borrow(const sos_conv_desc* d, const ShapeKey& k, ConvCfg* out) {
    static const int same[] = {4, 5, 6, 7, 8, 9, 10, 11, 14, 15, 16, 17, 18};
    ConvCfg best;
    long long best_score = -1;
    {
        std::lock_guard<std::mutex> g(tuned_mutex());
        for (const auto& kv : tuned_cache()) {
            bool ok = true;
            for (int i : same) ok = ok && kv.first.v[i] == k.v[i];
            if (!ok) continue;
            const long long score = ((kv.first.v[1] == k.v[1] && kv.first.v[12] == k.v[12]) ? (1LL << 50) : 0) +
                                    (long long)kv.first.v[0] * kv.first.v[12] * kv.first.v[13];
            if (score > best_score) { best_score = score; best = kv.second; }
        }
    }
    if (best_score < 0) return false;
    for (const ConvCfg& e : enumerate_cfgs(d))
        if (e.NC == best.NC && e.lth == best.lth && e.ltw == best.ltw && e.ks == best.ks) {
            *out = best;
            std::lock_guard<std::mutex> g(tuned_mutex());
            tuned_cache().emplace(k, best);
            return true;
        }
    return false;
}
static void tuned_store(const ShapeKey& k, const ConvCfg& c, bool overwrite) {
    std::lock_guard<std::mutex> g(tuned_mutex());
    if (overwrite) tuned_cache()[k] = c;
    else tuned_cache().emplace(k, c);
}

// ---------------------------------------------------------------------------------------------------------------------------
// Thin-input convolutions as per-wave streams (round 5): the layers whose input is the 16-channel packed module input (horizontal
// taps on the channel axis, §3.1c) -- 14 -> 96 1x1 and 10 -> 64 5x1 -- and the 1x1 data gradients of the 8-channel heads contract
// K = 16 channels per tap: one MFMA per tap and 32 pixels, and 6 (4) KB to WRITE per 1 (5) KB read.  On the tiled kernels a
// workgroup lives ~15 us for a 56 KB tile (pixel table, patch DMA, slab DMA, barrier, one tap, staged epilogue): 2.4-2.9 TB/s.
// Here a wave owns a contiguous run of pixels: the pixel operand of a 32-pixel stage is ONE 16-byte global load per lane (pixel =
// lane & 31, channels 8 (lane >> 5) ..: the MFMA fragment as it lies in memory), the weights of all taps stay in registers, the
// result goes through a wave-private LDS tile (no barrier) to 16-byte stores that cover the stage's contiguous output bytes.
// The pixel operands of the next two stages are in flight while a stage is multiplied.  Epilogue y = act(acc * scale + shift) as sos_conv_desc says.
struct ThinParams {
    const bf16_t* in; const bf16_t* wgt; bf16_t* out;
    const float* scale; const float* shift; const float* slope;
    long long K;                // pixels (B * H * W)
    int H, W, cout, cout_pad, dil, pad, reflect, act, per;
};

template <int NT, int KH>
__global__ __launch_bounds__(256) void conv_thin_kernel(ThinParams p) {
#if __HIP_DEVICE_COMPILE__
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int ROWB = NT * 64, PITCH = ROWB + 16;            // bytes of an output pixel row / of its LDS image
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    typedef __attribute__((address_space(3))) char lds_char;   // (explicit LDS pointers: through a generic char* the compiler fell
    lds_char* const lds = (lds_char*)smem;                      //  back to flat loads / stores and a scratch copy of the staged pieces)
    const int tile0 = wave * (2 * 32 * PITCH);              // two tiles per wave: one being stored, one being filled
    const int l31 = lane & 31, lhi = lane >> 5;
    const long long k0 = ((long long)blockIdx.x * 4 + wave) * p.per;
    const long long k1 = k0 + p.per < p.K ? k0 + p.per : p.K;
    const __amdgpu_buffer_rsrc_t rin = __builtin_amdgcn_make_buffer_rsrc((void*)p.in, 0, (unsigned)(p.K * 32), 0x00020000);
    // weights: fragment (tap, tile) = rows tile*32 + l31, channels 8 lhi ..
    bf16x8 wf[KH][NT];
#pragma unroll
    for (int t = 0; t < KH; ++t)
#pragma unroll
        for (int n = 0; n < NT; ++n)
            wf[t][n] = *(const bf16x8*)(p.wgt + ((size_t)(t * p.cout_pad + n * 32 + l31) * 16 + lhi * 8));
    // epilogue coefficients: [scale | shift] of all rows behind the waves' tiles in LDS (a lane's four consecutive rows of a
    // register quad are one broadcast 16-byte read each; 96 + 96 registers per lane otherwise)
    const bool affine = p.scale != nullptr;
    const float slope = (p.act == SOS_ACT_PRELU && p.slope) ? p.slope[0] : 0.f;
    const float sn = p.act == SOS_ACT_RELU ? 0.f : (p.act == SOS_ACT_PRELU ? slope : 1.f);
    constexpr int COEF = 4 * 2 * 32 * PITCH;                // byte offset of [scale | shift]
    if (affine) {
        for (int c = tid; c < NT * 32; c += 256) {
            *(__attribute__((address_space(3))) float*)(lds + COEF + c * 4) = p.scale[c];
            *(__attribute__((address_space(3))) float*)(lds + COEF + (NT * 32 + c) * 4) = p.shift[c];
        }
    }
    __syncthreads();                               // (the only barrier; before any wave can leave)
    if (k0 >= k1) return;
    // this lane's pixel of the next stage to fetch: flat index, row (for the vertical taps)
    long long lk = k0 + l31;
    int lw = (int)(lk % p.W), lh = (int)((lk / p.W) % p.H);
    auto fetch = [&](sos_u32x4 (&b)[KH]) {
#pragma unroll
        for (int t = 0; t < KH; ++t) {
            int hh = lh + t * p.dil - p.pad;
            if (p.reflect) hh = reflect_index(hh, p.H);
            const bool ok = (unsigned)hh < (unsigned)p.H && lk < k1;
            const unsigned off = ok ? (unsigned)((lk + (long long)(hh - lh) * p.W) * 32) + (unsigned)(lhi * 16) : 0xffffffffu;
            b[t] = __builtin_amdgcn_raw_buffer_load_b128(rin, (int)off, 0, 0);
        }
        lk += 32; lw += 32;
        while (lw >= p.W) { lw -= p.W; if (++lh == p.H) lh = 0; }
    };
    // A stage is split so that its STORES get time: on gfx950 loads and stores share vmcnt and may return out of order with each
    // other, so the compiler's wait for a stage's loads drains every store issued before it.  With [multiply, epilogue, store] per
    // stage a wave stood ~2 us per stage behind its own previous stores (4 us per stage, 3.6 TB/s; more loads in flight did not
    // help).  Order per step: multiply stage s + 1 (the wait for its loads drains the stores of stage s - 1, issued a whole epilogue
    // ago), refill its load slot, STORE stage s from its LDS tile, then the epilogue of stage s + 1 into the other tile.
    auto multiply = [&](const sos_u32x4 (&bx)[KH], f32x16 (&acc)[NT]) {
#pragma unroll
        for (int n = 0; n < NT; ++n)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[n][e] = 0.f;
#pragma unroll
        for (int t = 0; t < KH; ++t)
#pragma unroll
            for (int n = 0; n < NT; ++n) acc[n] = SOS_MFMA_32x32x16(wf[t][n], __builtin_bit_cast(bf16x8, bx[t]), acc[n], 0, 0, 0);
    };
    auto epilogue = [&](const f32x16 (&acc)[NT], const int tl) {
        // D[row][pixel l31]: four consecutive rows per register quad -> one 8-byte LDS write
#pragma unroll
        for (int n = 0; n < NT; ++n)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                float y[4];
                float4 s4 = make_float4(1.f, 1.f, 1.f, 1.f), h4 = make_float4(0.f, 0.f, 0.f, 0.f);
                if (affine) {
                    s4 = *(const __attribute__((address_space(3))) float4*)(lds + COEF + (n * 32 + 8 * q + 4 * lhi) * 4);
                    h4 = *(const __attribute__((address_space(3))) float4*)(lds + COEF + (NT * 32 + n * 32 + 8 * q + 4 * lhi) * 4);
                }
                const float sq[4] = {s4.x, s4.y, s4.z, s4.w}, hq[4] = {h4.x, h4.y, h4.z, h4.w};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int r = 4 * q + e;
                    float v = affine ? fmaf(acc[n][r], sq[e], hq[e]) : acc[n][r];
                    if (p.act == SOS_ACT_SIGMOID) v = 1.0f / (1.0f + expf(-v));
                    else if (p.act != SOS_ACT_NONE) v = fmaxf(v, 0.f) + sn * fminf(v, 0.f);
                    y[e] = (n * 32 + 8 * q + 4 * lhi + e < p.cout) ? v : 0.f;
                }
                *(__attribute__((address_space(3))) uint2*)(lds + tl + l31 * PITCH + (n * 32 + 8 * q + 4 * lhi) * 2) =
                    make_uint2(pack2bf(y[0], y[1]), pack2bf(y[2], y[3]));
            }
    };
    auto store = [&](const long long kb, const int tl) {
        // the stage's 32 x ROWB output bytes are contiguous in memory: 16-byte pieces, lane-contiguous
        const int npx = (int)(k1 - kb < 32 ? k1 - kb : 32);
        char* ob = (char*)p.out + kb * ROWB;
        // (all pieces are read into their own registers before the first store: a store whose data registers are reloaded for the
        // next piece makes the compiler wait for the PREVIOUS store to finish -- vmcnt(0) before each of the six)
        constexpr int NP = (32 * ROWB) / 1024;
        uint4 v[NP];
#pragma unroll
        for (int i = 0; i < NP; ++i) {
            const int j = i * 64 + lane, px = j / (ROWB / 16), pc = j - px * (ROWB / 16);
            v[i] = *(const __attribute__((address_space(3))) uint4*)(lds + tl + px * PITCH + pc * 16);
        }
#pragma unroll
        for (int i = 0; i < NP; ++i) {
            const int j = i * 64 + lane, px = j / (ROWB / 16);
            if (px < npx) *(uint4*)(ob + (size_t)j * 16) = v[i];
        }
    };
    sos_u32x4 b[2][KH];
    f32x16 acc[NT];
    fetch(b[0]);
    fetch(b[1]);
    multiply(b[0], acc);
    fetch(b[0]);
    epilogue(acc, tile0);
    int cur = 0;
    for (long long kb = k0; kb < k1; kb += 32) {
        const bool more = kb + 32 < k1;
        __builtin_amdgcn_sched_barrier(0);
        if (more) {
            if (cur == 0) { multiply(b[1], acc); fetch(b[1]); } else { multiply(b[0], acc); fetch(b[0]); }
        }
        __builtin_amdgcn_sched_barrier(0);
        store(kb, tile0 + cur * (32 * PITCH));
        __builtin_amdgcn_sched_barrier(0);
        if (more) epilogue(acc, tile0 + (cur ^ 1) * (32 * PITCH));
        cur ^= 1;
    }
#endif
}

// which descriptors take it (everything else of sos_conv_desc must be off), and with how many workgroups
static bool thin_conv_shape(const sos_conv_desc* d) {
    if (getenv("SOS_CONV_NO_THIN")) return false;            // (read per call: the equality test flips it)
    if (d->cin != 16 || d->in_cs != 16 || d->cin_off != 0 || d->in_nseg != 1 || d->w_gather || d->wl_tab || d->t_taps > 1 ||
        d->fold_pad || d->in_scale || d->accumulate || d->stats || d->out_dtype != SOS_DT_BF16)
        return false;
    if (d->kw != 1 || (d->kh != 1 && d->kh != 5) || d->stride != 1 || d->pad_left != 0 || d->Ho != d->H || d->Wo != d->W || d->W < 32)
        return false;
    if (d->kh == 5 ? d->cout_pad != 64 : d->cout_pad != 96) return false;             // (the two instances)
    if (d->pad_top != (d->kh - 1) / 2 * d->dil_h || (d->pad_mode == SOS_PAD_REFLECT && d->pad_top >= d->H)) return false;
    if (d->cout_store != d->cout_pad || d->out_c_off != 0 || d->out_sc != 1 || d->out_sw != d->cout_store ||
        d->out_sh != (int64_t)d->Wo * d->out_sw || d->out_sb != (int64_t)d->Ho * d->out_sh)
        return false;
    const long long K = (long long)d->B * d->H * d->W;
    return K >= 2048 && K * 32 < 0xfff00000ll;
}
static int thin_conv_launch(const sos_conv_desc* d, hipStream_t s) {
    ThinParams q;
    q.in = (const bf16_t*)d->in; q.wgt = (const bf16_t*)d->wgt; q.out = (bf16_t*)d->out;
    q.scale = d->scale; q.shift = d->shift; q.slope = d->act_param;
    q.K = (long long)d->B * d->H * d->W; q.H = d->H; q.W = d->W; q.cout = d->cout; q.cout_pad = d->cout_pad;
    q.dil = d->dil_h; q.pad = d->pad_top; q.reflect = d->pad_mode == SOS_PAD_REFLECT; q.act = d->act;
    static const int wgs = [] { const char* e = getenv("SOS_CONV_THIN_WGS"); return e && atoi(e) > 0 ? atoi(e) : 512; }();
    long long per = (q.K + wgs * 4 - 1) / (wgs * 4);
    per = (per + 31) / 32 * 32;
    q.per = (int)per;
    const unsigned grid = (unsigned)((q.K + per * 4 - 1) / (per * 4));
    if (d->kh == 5) hipLaunchKernelGGL((conv_thin_kernel<2, 5>), dim3(grid), dim3(256), 8 * 32 * (2 * 64 + 16) + 2 * 64 * 4, s, q);
    else hipLaunchKernelGGL((conv_thin_kernel<3, 1>), dim3(grid), dim3(256), 8 * 32 * (3 * 64 + 16) + 2 * 96 * 4, s, q);
    return sos_check_launch("sos_conv2d_fwd(thin)");
}

static int validate(const sos_conv_desc* d) {
    if (!d || !d->in || !d->wgt || !d->out || ((d->scale == nullptr) != (d->shift == nullptr))) {
        sos_set_error("sos_conv2d_fwd: null pointer");
        return SOS_EINVAL;
    }
    if (d->cin < 16 || d->cin % 16 || d->in_cs % 8 || d->cin_off % 8 || d->cout_pad % 32 || d->cout > d->cout_pad ||
        d->cout < 1 || d->kh < 1 || d->kw < 1 || d->kh * d->kw > 64 || d->stride < 1 || d->dil_h < 1 || d->dil_w < 1 ||
        (d->stride > 1 && (d->dil_h > 1 || d->dil_w > 1)) || d->B < 1 || d->Ho < 1 || d->Wo < 1 ||
        d->in_nseg < 1 || d->in_seg_stride % 8 || d->cin_off + (d->in_nseg - 1) * d->in_seg_stride + d->cin > d->in_cs ||
        d->cout_store < d->cout || d->out_dtype < 0 || d->out_dtype > 2 || (d->w_gather == nullptr && d->Wl != d->W)) {
        sos_set_error("sos_conv2d_fwd: bad descriptor (cin=%d in_cs=%d cin_off=%d cout=%d/%d k=%dx%d s=%d d=%dx%d)",
                      d->cin, d->in_cs, d->cin_off, d->cout, d->cout_pad, d->kh, d->kw, d->stride, d->dil_h, d->dil_w);
        return SOS_EINVAL;
    }
    if (d->t_taps > 1) {
        const long long frame_bytes = (long long)d->H * d->W * d->in_cs * 2;
        if (d->t_frames < 1 || d->B % d->t_frames || d->t_pad < 0 || d->t_pad >= d->t_taps || d->w_gather || d->wl_tab ||
            (d->t_frames + d->t_taps) * frame_bytes >= 0xffffff00ll) {
            sos_set_error("sos_conv2d_fwd: bad temporal taps (B=%d frames=%d taps=%d pad=%d; a clip plus the temporal halo "
                          "must stay below 4 GB)", d->B, d->t_frames, d->t_taps, d->t_pad);
            return SOS_EINVAL;
        }
    }
    if (d->stats && (d->out_dtype == SOS_DT_F32 || d->out_sc != 1 || d->stats_c < 1 || d->stats_c > d->cout_pad || d->accumulate)) {
        sos_set_error("sos_conv2d_fwd: fused statistics need a dense bf16 NHWC output without accumulation");
        return SOS_EINVAL;
    }
    if ((d->wl_tab == nullptr) != (d->wo_tab == nullptr) || (d->wl_tab && d->stats)) {
        sos_set_error("sos_conv2d_fwd: ragged batches need both wl_tab and wo_tab and no fused statistics");
        return SOS_EINVAL;
    }
    if ((d->in_scale == nullptr) != (d->in_shift == nullptr) ||
        (d->in_scale && (d->in_nseg != 1 || d->t_taps > 1 || d->cin % 8 || d->out_dtype == SOS_DT_BF16X3))) {
        sos_set_error("sos_conv2d_fwd: fused input BatchNorm needs in_scale AND in_shift, one 16-bit channel segment, no temporal taps");
        return SOS_EINVAL;
    }
    if ((long long)d->in_cs * 2 > 0xffff) {        // (a chunk's byte offset inside a pixel is added to pre-resolved 32-bit offsets: stage_issue)
        sos_set_error("sos_conv2d_fwd: pixel pitch of %d channels not supported (<= 32767)", d->in_cs);
        return SOS_EINVAL;
    }
    if ((uint64_t)d->H * d->W * d->in_cs * 2 >= 0xfff00000ull) {
        sos_set_error("sos_conv2d_fwd: one input image exceeds 4 GB");
        return SOS_ENOSPC;
    }
    if (d->fold_pad < 0 || (d->fold_pad > 0 && (!d->fold_pad_out || d->fold_H < 1 || d->fold_W < 1 || d->fold_sy < 1 || d->fold_sx < 1 ||
                                               d->fold_oy < 0 || d->fold_ox < 0 || d->fold_row < d->cout_store || d->fold_row % 8 ||
                                               d->out_dtype == SOS_DT_F32 || d->out_sc != 1 || d->stats || d->wl_tab ||
                                               (d->Ho - 1) * d->fold_sy + d->fold_oy >= d->fold_H + 2 * d->fold_pad ||
                                               (d->Wo - 1) * d->fold_sx + d->fold_ox >= d->fold_W + 2 * d->fold_pad ||
                                               (int64_t)(d->fold_H + 2 * d->fold_pad) * (d->fold_W + 2 * d->fold_pad) * d->fold_row >= 0x7ffffff0ll ||
                                               (int64_t)d->fold_H * d->fold_W * d->out_sw >= 0x7ffffff0ll))) {
        sos_set_error("sos_conv2d_fwd: bad reflection-pad fold (16-bit dense NHWC outputs inside the padded domain only)");
        return SOS_EINVAL;
    }
    // the staged (16-bit, channel-contiguous) store path keeps a pixel's offset inside its image in 32 bits
    if (d->out_dtype != SOS_DT_F32 && d->out_sc == 1 && !d->fold_pad &&
        (int64_t)(d->Ho - 1) * d->out_sh + (int64_t)(d->Wo - 1) * d->out_sw + d->out_c_off + d->cout_store >= 0x7ffffff0ll) {
        sos_set_error("sos_conv2d_fwd: one output image spans more than 2^31 elements (Ho=%d Wo=%d)", d->Ho, d->Wo);
        return SOS_ENOSPC;
    }
    if (d->pad_mode == SOS_PAD_REFLECT && (d->pad_top >= d->H || d->pad_left >= d->Wl)) {
        sos_set_error("sos_conv2d_fwd: reflect pad %d/%d needs a larger input (%dx%d)", d->pad_top, d->pad_left, d->H, d->Wl);
        return SOS_EINVAL;
    }
    return SOS_OK;
}

// ds_read_b128 lane groups of the 32 lanes that address distinct pixels (MI355X_MICROARCH.md, LDS): count, for each
// candidate lane -> pixel relabelling, the LDS cycles the pixel-fragment read of one column tile takes (a 16-byte slot is
// 4 of the 64 banks; lanes of a group on one slot serialise) and keep the cheapest.
static int pick_lane_map(const ConvParams& p, int pstride) {
    static const int grp[2][16] = {{0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27},
                                   {4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31}};
    const int TW = p.TW, TH = p.TH;
    int best = 0, best_cost = 1 << 30;
    for (int mode = 0; mode < 3; ++mode) {
        if (mode == 2 && TW != 8) continue;
        int cost = 0;
        for (int g = 0; g < 2; ++g) {
            int cnt[16] = {0}, worst = 0;
            for (int k = 0; k < 16; ++k) {
                const int l = grp[g][k];
                const int lp = mode == 0 ? l : (mode == 1 ? g * 16 + k : (g + 2 * (k >> 3)) * 8 + (k & 7));
                int cls, i, j;
                tile_decode(lp, TH, TW, cls, i, j);
                if (cls >= p.NC) cls = i = j = 0;
                const int off = ((cls * p.PH + i * p.stride) * p.PW + j * p.stride) * pstride;
                worst = std::max(worst, ++cnt[(off >> 4) & 15]);
            }
            cost += worst;
        }
        if (cost < best_cost) { best_cost = cost; best = mode; }
    }
    return best;
}

static int launch_cfg(const sos_conv_desc* d, const ConvCfg& c, hipStream_t s) {
    ConvParams p;
    p.in = (const bf16_t*)d->in; p.wgt = (const bf16_t*)d->wgt; p.out = d->out;
    p.scale = d->scale; p.shift = d->shift; p.slope = d->act_param; p.wgather = d->w_gather;
    p.B = d->B; p.H = d->H; p.W = d->W; p.Wl = d->Wl; p.in_cs = d->in_cs; p.cin_off = d->cin_off; p.cin = d->cin;
    p.kh = d->kh; p.kw = d->kw; p.cout = d->cout; p.cout_pad = d->cout_pad; p.cout_store = d->cout_store;
    p.stride = d->stride; p.dh = d->dil_h; p.dw = d->dil_w; p.pad_t = d->pad_top; p.pad_l = d->pad_left;
    p.pad_mode = d->pad_mode; p.Ho = d->Ho; p.Wo = d->Wo; p.out_dtype = d->out_dtype; p.c_off = d->out_c_off;
    p.stats = d->stats; p.stats_c = d->stats_c;
    p.in_scale = d->in_scale; p.in_shift = d->in_shift;
    p.wl_tab = d->wl_tab; p.wo_tab = d->wo_tab; p.wg_stride = d->w_gather_stride;
    p.act = d->act; p.accum = d->accumulate; p.sb = d->out_sb; p.sh = d->out_sh; p.sw = d->out_sw; p.sc = d->out_sc; p.third = d->out_third;
    p.out2 = d->fold_pad_out; p.fP = d->fold_pad; p.fH = d->fold_H; p.fW = d->fold_W; p.fsy = d->fold_sy; p.foy = d->fold_oy;
    p.fsx = d->fold_sx; p.fox = d->fold_ox; p.frow = d->fold_row; p.fthird = d->fold_third;
    // c.ks encodes: k-steps per chunk (% 100), + 100 single slab buffer, + 1000 * n-tiles per workgroup (0: default)
    const int nt = c.ks >= 1000 ? c.ks / 1000 : nt_for(d);
    const int ks_enc = c.ks >= 1000 ? c.ks % 1000 : c.ks;
    const int nby = (d->cout_pad / 32 + nt - 1) / nt;
    const int Hc = (d->Ho + d->dil_h - 1) / d->dil_h, Wc = (d->Wo + d->dil_w - 1) / d->dil_w;
    const int TH = tdim(c.lth), TW = tdim(c.ltw);
    p.NC = c.NC; p.TH = TH; p.TW = TW;
    const bool pt8 = c.ks <= -4;                          // 16-row kernel, 512-pixel workgroup
    const bool pt3 = c.ks >= 300 && c.ks < 400;           // 32-row kernel, 384-slot tile (three column tiles per wave)
    if (c.NC < 1 || TH < 1 || TW < 1 || c.NC * TH * TW > (pt8 ? 512 : (pt3 ? 384 : 256))) { sos_set_error("sos_conv2d_fwd: internal: tile %d x %d x %d", c.NC, TH, TW); return SOS_EINVAL; }
    p.PH = (TH - 1) * d->stride + d->kh; p.PW = (TW - 1) * d->stride + d->kw;
    p.npix = p.NC * p.PH * p.PW;
    p.cps = c.ks > 0 ? d->cin / (16 * (ks_enc % 100)) : 1;
    p.nchunks = p.cps * nseg_eff(d);
    p.ktot = d->cin * nseg_eff(d);
    p.tk = d->t_taps > 1 ? d->t_taps : 1; p.tT = d->t_taps > 1 ? d->t_frames : 1; p.tpad = d->t_taps > 1 ? d->t_pad : 0;
    p.seg_stride = d->in_seg_stride;
    p.tiles_h = (Hc + TH - 1) / TH; p.tiles_w = (Wc + TW - 1) / TW; p.ngw = (d->dil_w + p.NC - 1) / p.NC;
    const long long nblk = (long long)d->B * d->dil_h * p.tiles_h * p.ngw * p.tiles_w;
    if (nblk > 0x7fffffffLL) { sos_set_error("sos_conv2d_fwd: grid too large"); return SOS_EINVAL; }
    p.nblk = (int)nblk;
    // (n * d < 2^32 for every division the kernels make: block ids < 2^31 / (B dil_h) are divided by extents whose product is nblk / B)
    if (nblk * std::max(std::max(p.tiles_w, p.ngw), std::max(p.tiles_h, (int)d->dil_h)) >= (1ll << 32)) { sos_set_error("sos_conv2d_fwd: grid too large"); return SOS_EINVAL; }
    p.mgTW = mg_of(TW); p.mgTH = mg_of(TH); p.mgPW = mg_of(p.PW); p.mgPH = mg_of(p.PH);
    p.mg_tiles_w = mg_of(p.tiles_w); p.mg_ngw = mg_of(p.ngw); p.mg_tiles_h = mg_of(p.tiles_h); p.mg_dh = mg_of(d->dil_h); p.mg_kw = mg_of(d->kw);
    { const char* e = getenv("SOS_CONV_DBG"); p.dbg = e ? atoi(e) : 0; }
    p.lmap = 0;
    if (c.ks > 0) {
        static const char* nomap = getenv("SOS_CONV_NO_LANE_MAP");          // A/B switch
        if (!nomap) p.lmap = pick_lane_map(p, (ks_enc % 100) * 32 + 16);
    }
    if (c.ks <= 0) {                                     // 16-row kernel (ks 0: double-buffered slab, -1: single, -2: ring of three; -4 / -5: 512-pixel workgroups)
        int mode = pt8 ? -c.ks - 4 : -c.ks;
        if (!pt8) { static const char* e = getenv("SOS_CONV16_MODE"); if (e) mode = atoi(e); }      // A/B: force a slab mode
        const int nt16 = nt16_for(d), ks16 = d->cin / 16;
        if (!nt16) { sos_set_error("sos_conv2d_fwd: internal: 16-row tiling for an ineligible shape"); return SOS_EINVAL; }
        p.cps = 1; p.nchunks = nseg_eff(d);              // channel segments (3 in the hi|hi|lo mode), whole cin per segment
        if (mode < 0 || mode > (pt8 ? 1 : 2) || lds_bytes16(p.npix, nt16, ks16, mode) > LDS_LIMIT) mode = pt8 ? -c.ks - 4 : -c.ks;
        size_t lds16 = lds_bytes16(p.npix, nt16, ks16, mode);
        const size_t slots = pt8 ? 512 : 256;
        const size_t stage16 = slots * (nt16 * 32 + 16) * (d->out_dtype == SOS_DT_BF16X3 ? 2 : 1) + slots * 4 + (d->stats ? 16384 : 0);
        if (stage16 > lds16) lds16 = stage16;
        if (lds16 > LDS_LIMIT) { sos_set_error("sos_conv2d_fwd: internal: 16-row tile needs %zu bytes of LDS", lds16); return SOS_EINVAL; }
        conv_kernel_t k = nullptr;
#define SOS_C16(NTV, KSV)                                                                        \
        if (nt16 == NTV && ks16 == KSV) k = pt8 ? (mode == 1 ? conv16_kernel<NTV, KSV, 1, 8> : conv16_kernel<NTV, KSV, 0, 8>)     \
                                                : (mode == 1 ? conv16_kernel<NTV, KSV, 1> : (mode == 2 ? conv16_kernel<NTV, KSV, 2> : conv16_kernel<NTV, KSV, 0>));
        SOS_C16(1, 1) SOS_C16(1, 3) SOS_C16(3, 1) SOS_C16(3, 3)
#undef SOS_C16
        static sos_device_once attr16;
        (void)sos_per_device_once(attr16, [] {
#define SOS_C16A(NTV, KSV)                                                                                                        \
            (void)hipFuncSetAttribute((const void*)conv16_kernel<NTV, KSV, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_LIMIT);  \
            (void)hipFuncSetAttribute((const void*)conv16_kernel<NTV, KSV, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_LIMIT);  \
            (void)hipFuncSetAttribute((const void*)conv16_kernel<NTV, KSV, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_LIMIT);  \
            (void)hipFuncSetAttribute((const void*)conv16_kernel<NTV, KSV, 0, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_LIMIT);  \
            (void)hipFuncSetAttribute((const void*)conv16_kernel<NTV, KSV, 1, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_LIMIT);
            SOS_C16A(1, 1) SOS_C16A(1, 3) SOS_C16A(3, 1) SOS_C16A(3, 3)
#undef SOS_C16A
            return (int)SOS_OK;
        });
        hipLaunchKernelGGL(k, dim3((unsigned)nblk, 1), dim3(256), lds16, s, p);
        return sos_check_launch("sos_conv2d_fwd(16)");
    }
    dim3 grid((unsigned)nblk, (unsigned)nby);
    int ks_run = ks_enc;
    // fused input BatchNorm: built for the plain double-slab instances only -- a three-per-CU entry (ks + 200) runs as its plain
    // twin (same tile, same k-steps per chunk: the same summation order, bit-identical output)
    if (d->in_scale && ks_run >= 200 && ks_run < 300) ks_run -= 200;
    size_t lds = lds_bytes(p.npix, nt, ks_run);
    if (d->out_dtype != SOS_DT_F32 && d->out_sc == 1) {
        const size_t slots = pt3 ? 384 : 256;
        const bool w3 = ks_run >= 200 && ks_run < 300;               // three per CU: TIGHT epilogue layout (nothing but the staged tile)
        const size_t stage = w3 ? slots * (nt * 64 + 16)
                                : slots * (nt * 64 + 16) * (d->out_dtype == SOS_DT_BF16X3 ? 2 : 1) + slots * 4 +   // + pixel offsets
                                  ((d->stats && !pt3) ? 16384 : 0);  // + statistics scratch (384-slot tiles: over the staged tile)
        if (stage > lds) lds = stage;
        if (w3 && (d->out_dtype == SOS_DT_BF16X3 || nt != 3 || d->accumulate || d->cout_store % 8)) { sos_set_error("sos_conv2d_fwd: internal: three-per-CU tiling for an ineligible shape"); return SOS_EINVAL; }
        if (pt3 && (d->out_dtype == SOS_DT_BF16X3 || nt != 3)) { sos_set_error("sos_conv2d_fwd: internal: 384-slot tiling for an ineligible shape"); return SOS_EINVAL; }
    }
#ifdef SOS_ABLATE
    { static const char* e = getenv("SOS_CONV_LDS_PAD"); if (e) lds = std::min(lds + (size_t)atoi(e), LDS_LIMIT); }   // occupancy experiments
#endif
    switch (nt) {
        case 1: return launch_ks<1>(ks_run, p, grid, lds, s);
        case 2: return launch_ks<2>(ks_run, p, grid, lds, s);
        case 3: return launch_ks<3>(ks_run, p, grid, lds, s);
        case 4: return launch_ks<4>(ks_run, p, grid, lds, s);
    }
    sos_set_error("sos_conv2d_fwd: internal: nt=%d", nt);
    return SOS_EINVAL;
}

static long long tiles_of(const sos_conv_desc* d, const ConvCfg& c) {
    const int Hc = (d->Ho + d->dil_h - 1) / d->dil_h, Wc = (d->Wo + d->dil_w - 1) / d->dil_w;
    const int TH = tdim(c.lth), TW = tdim(c.ltw);
    return (long long)d->B * d->dil_h * ((Hc + TH - 1) / TH) * ((d->dil_w + c.NC - 1) / c.NC) * ((Wc + TW - 1) / TW);
}

// SOS_CONV16_FORCE512=1 (testing): every shape the 16-row kernel takes runs its cheapest 512-pixel-workgroup candidate (ks -5 / -4)
static bool forced_512(const sos_conv_desc* d, ConvCfg* out) {
    static const char* f512 = getenv("SOS_CONV16_FORCE512");
    if (!(f512 && atoi(f512)) || !nt16_for(d)) return false;
    for (const ConvCfg& e : enumerate_cfgs(d))
        if (e.ks <= -4) { *out = e; return true; }
    return false;
}

// SOS_CONV_FORCE_PT3=1 / SOS_CONV_FORCE_W3=1 (testing / A-B): every shape that has a 384-slot candidate (ks + 300) / a
// three-per-CU candidate (ks + 200) runs its cheapest one
static bool forced_pt3(const sos_conv_desc* d, ConvCfg* out) {
    static const char* f = getenv("SOS_CONV_FORCE_PT3");
    static const char* f3 = getenv("SOS_CONV_FORCE_W3");
    const bool pt3 = f && atoi(f), w3 = f3 && atoi(f3);
    if (!(pt3 || w3) || d->in_scale) return false;
    for (const ConvCfg& e : enumerate_cfgs(d))
        if ((pt3 && e.ks >= 300 && e.ks < 400) || (w3 && e.ks >= 200 && e.ks < 300)) { *out = e; return true; }
    return false;
}

extern "C" int64_t sos_conv2d_tile_count(const sos_conv_desc* d) {
    if (validate(d)) return -1;
    { ConvCfg c5; if (forced_512(d, &c5) || forced_pt3(d, &c5)) return tiles_of(d, c5); }
    static const char* force = getenv("SOS_CONV_FORCE_CFG");
    if (!force) {
        ConvCfg c;
        const ShapeKey k = shape_key(d);
        if (tuned_lookup(k, &c) || tuned_borrow(d, k, &c)) return tiles_of(d, c);
    }
    std::vector<ConvCfg> cfgs = enumerate_cfgs(d);
    if (cfgs.empty()) { sos_set_error("sos_conv2d_tile_count: no tile fits LDS"); return -1; }
    return tiles_of(d, cfgs[force ? (size_t)atol(force) % cfgs.size() : 0]);
}

extern "C" int sos_conv2d_fwd(const sos_conv_desc* d, sos_stream_t stream) {
    int rc = validate(d);
    if (rc) return rc;
    if (thin_conv_shape(d)) return thin_conv_launch(d, (hipStream_t)stream);
    { ConvCfg c5; if (forced_512(d, &c5) || forced_pt3(d, &c5)) return launch_cfg(d, c5, (hipStream_t)stream); }
    // SOS_CONV_FORCE_CFG=k (testing): use the k-th candidate tiling (mod count) instead of the tuned one
    static const char* force = getenv("SOS_CONV_FORCE_CFG");
    if (!force) {
        ConvCfg c;
        const ShapeKey k = shape_key(d);
        if (tuned_lookup(k, &c) || tuned_borrow(d, k, &c)) return launch_cfg(d, c, (hipStream_t)stream);
    }
    std::vector<ConvCfg> cfgs = enumerate_cfgs(d);
    if (cfgs.empty()) {
        sos_set_error("sos_conv2d_fwd: no tile fits LDS (cin=%d k=%dx%d)", d->cin, d->kh, d->kw);
        return SOS_ENOSPC;
    }
    const size_t pick = force ? (size_t)atol(force) % cfgs.size() : 0;
    if (getenv("SOS_CONV_LIST"))      // debugging aid: the chosen candidate of the cost-ordered list
        fprintf(stderr, "sos_conv2d_fwd: cfg %zu/%zu NC=%d TH=%d TW=%d ks=%d\n", pick, cfgs.size(), cfgs[pick].NC, tdim(cfgs[pick].lth),
                tdim(cfgs[pick].ltw), cfgs[pick].ks);
    return launch_cfg(d, cfgs[pick], (hipStream_t)stream);
}

// Measure the best few tilings of this descriptor's shape on the device (HIP events on `stream`,
// synchronises!) and remember the winner for later sos_conv2d_fwd calls of the same shape.
// Call outside of graph capture / timed regions.  Returns 0 and, if non-NULL, the winning time.
extern "C" int sos_conv2d_tune(const sos_conv_desc* d, int max_candidates, int iters, float* best_ms,
                               sos_stream_t stream) {
    int rc = validate(d);
    if (rc) return rc;
    const ShapeKey key = shape_key(d);
    { ConvCfg c; if (tuned_lookup(key, &c)) { if (best_ms) *best_ms = -1.f; return SOS_OK; } }
    std::vector<ConvCfg> cfgs = enumerate_cfgs(d);
    if (cfgs.empty()) { sos_set_error("sos_conv2d_tune: no tile fits LDS"); return SOS_ENOSPC; }
    if (max_candidates < 1) max_candidates = 1;
    if (iters < 1) iters = 1;
    hipStream_t s = (hipStream_t)stream;
    hipEvent_t e0, e1;
    if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) {
        sos_set_error("sos_conv2d_tune: hipEventCreate failed");
        return SOS_ELAUNCH;
    }
    float best = 1e30f;
    int besti = 0;
    const int n = std::min<int>((int)cfgs.size(), max_candidates);
    auto time_cfg = [&](const int i, const int reps, float* ms_out) {
        int r = launch_cfg(d, cfgs[i], s);                      // warm-up
        if (r) return r;
        (void)hipEventRecord(e0, s);
        for (int k = 0; k < reps && !r; ++k) r = launch_cfg(d, cfgs[i], s);
        (void)hipEventRecord(e1, s);
        if (r || hipEventSynchronize(e1) != hipSuccess) return r ? r : (int)SOS_ELAUNCH;
        float ms = 0.f;
        (void)hipEventElapsedTime(&ms, e0, e1);
        *ms_out = ms / reps;
        return (int)SOS_OK;
    };
    static const char* verbose = getenv("SOS_CONV_TUNE_VERBOSE");
    std::vector<std::pair<float, int>> timed;
    for (int i = 0; i < n; ++i) {
        float ms = 0.f;
        if ((rc = time_cfg(i, iters, &ms))) break;
        timed.push_back({ms, i});
        if (ms < best) { best = ms; besti = i; }
    }
    // Wide searches (round 4: more than 16 candidates, tools/make_tune_table.py --retune-wide) are decided in a second stage: a
    // 3-launch timing of ~100 candidates picks a lucky sample as often as a better tiling (a 48-candidate table of single-stage
    // timings measured no better than the 8-candidate one in round 3), so the five fastest are timed again over 8x the launches,
    // ALTERNATING twice, and the cost model's first candidate is kept unless the winner beats it by more than 2 %.
    if (!rc && n > 16) {
        std::sort(timed.begin(), timed.end());
        const int top = std::min<int>(5, (int)timed.size());
        std::vector<int> cand;
        for (int t = 0; t < top; ++t) cand.push_back(timed[t].second);
        if (std::find(cand.begin(), cand.end(), 0) == cand.end()) cand.push_back(0);
        std::vector<float> acc(cand.size(), 0.f);
        for (int round = 0; round < 2 && !rc; ++round)
            for (size_t t = 0; t < cand.size() && !rc; ++t) {
                float ms = 0.f;
                rc = time_cfg(cand[t], iters * 8, &ms);
                acc[t] += ms * 0.5f;
            }
        if (!rc) {
            size_t bt = 0, t0 = cand.size();
            for (size_t t = 0; t < cand.size(); ++t) { if (acc[t] < acc[bt]) bt = t; if (cand[t] == 0) t0 = t; }
            if (t0 < cand.size() && acc[bt] > 0.98f * acc[t0]) bt = t0;
            best = acc[bt]; besti = cand[bt];
            if (verbose)
                fprintf(stderr, "tune k%dx%d d%dx%d s%d cin%d cout%d %dx%d B%d: %d candidates, pick #%d NC=%d TH=%d TW=%d ks=%d %.4f ms (model's first %.4f ms)\n",
                        d->kh, d->kw, d->dil_h, d->dil_w, d->stride, d->cin, d->cout, d->Ho, d->Wo, d->B, n, besti, cfgs[besti].NC,
                        tdim(cfgs[besti].lth), tdim(cfgs[besti].ltw), cfgs[besti].ks, best, t0 < cand.size() ? acc[t0] : -1.f);
        }
    }
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    if (rc) return rc;
    tuned_store(key, cfgs[besti], false);       // first writer wins: concurrent tuners of one shape agree afterwards
    if (best_ms) *best_ms = best;
    return SOS_OK;
}

// Persist / restore the tiling table.  Text file: a header line `sos_conv_tune <format> abi <SOS_TUNE_KEY_GEN> nkey 19`
// (the generation of the shape-key / tiling-tuple MEANING: it changes when sos_conv_desc fields a key is built from change
// meaning, not with every sos_abi_version bump -- entries are re-validated against enumerate_cfgs() on load anyway),
// then 19 shape ints + 4 tiling ints per line (ks == 0 / -1: 16-row kernel).  The package ships a table for the
// BASELINE shapes (tune_table_gfx950.txt) so that every process -- and every rank of a data-parallel job -- runs the
// SAME tilings, hence the same summation order; timing-based autotuning is opt-in.
#define SOS_TUNE_FORMAT 2
#define SOS_TUNE_KEY_GEN 3

// the sos_conv_desc fields enumerate_cfgs() / nt16_for() read, rebuilt from a shape key
static sos_conv_desc desc_of_key(const ShapeKey& k) {
    sos_conv_desc d;
    memset(&d, 0, sizeof(d));
    d.B = k.v[0]; d.H = k.v[1]; d.W = k.v[2]; d.Wl = k.v[3]; d.cin = k.v[4]; d.in_nseg = k.v[5]; d.cout_pad = k.v[6];
    d.kh = k.v[7]; d.kw = k.v[8]; d.stride = k.v[9]; d.dil_h = k.v[10]; d.dil_w = k.v[11]; d.Ho = k.v[12]; d.Wo = k.v[13];
    d.out_dtype = k.v[14]; d.out_sc = k.v[15] ? 1 : 2; d.pad_mode = k.v[16]; d.cout = k.v[18];
    return d;
}

extern "C" int sos_conv2d_tune_save(const char* path) {
    if (!path) { sos_set_error("sos_conv2d_tune_save: null path"); return SOS_EINVAL; }
    // written to a temporary and renamed: a reader (another rank) never sees a half-written table
    char tmp[4096];
    snprintf(tmp, sizeof(tmp), "%s.tmp.%ld", path, (long)getpid());
    FILE* f = fopen(tmp, "w");
    if (!f) { sos_set_error("sos_conv2d_tune_save: cannot open %s", tmp); return SOS_EINVAL; }
    fprintf(f, "sos_conv_tune %d abi %d nkey 19\n", SOS_TUNE_FORMAT, SOS_TUNE_KEY_GEN);
    {
        std::lock_guard<std::mutex> g(tuned_mutex());
        for (const auto& kv : tuned_cache()) {
            for (int i = 0; i < 19; ++i) fprintf(f, "%d ", kv.first.v[i]);
            fprintf(f, "%d %d %d %d\n", kv.second.NC, kv.second.lth, kv.second.ltw, kv.second.ks);
        }
    }
    if (fclose(f) != 0 || rename(tmp, path) != 0) {
        remove(tmp);
        sos_set_error("sos_conv2d_tune_save: cannot write %s", path);
        return SOS_EINVAL;
    }
    return SOS_OK;
}

// Returns the number of entries accepted (0: no file), or a negative error for a file of another format / ABI.
// An entry is accepted only if its tiling is one enumerate_cfgs() would offer for that shape with this build (LDS
// fit, NC vs dilation, 16-row eligibility): a stale or foreign table can never select an invalid launch.
extern "C" int sos_conv2d_tune_load(const char* path) {
    if (!path) { sos_set_error("sos_conv2d_tune_load: null path"); return SOS_EINVAL; }
    FILE* f = fopen(path, "r");
    if (!f) return 0;                    // nothing cached yet
    int fmt = -1, abi = -1, nkey = -1;
    if (fscanf(f, " sos_conv_tune %d abi %d nkey %d", &fmt, &abi, &nkey) != 3 || fmt != SOS_TUNE_FORMAT ||
        abi != SOS_TUNE_KEY_GEN || nkey != 19) {
        fclose(f);
        sos_set_error("sos_conv2d_tune_load: %s was written by another build (format %d abi %d)", path, fmt, abi);
        return SOS_EINVAL;
    }
    int n = 0;
    for (;;) {
        ShapeKey k;
        ConvCfg c;
        bool ok = true;
        for (int i = 0; i < 19 && ok; ++i) ok = fscanf(f, "%d", &k.v[i]) == 1;
        if (!ok || fscanf(f, "%d %d %d %d", &c.NC, &c.lth, &c.ltw, &c.ks) != 4) break;
        c.cost = 0;
        const sos_conv_desc d = desc_of_key(k);
        if (d.cin < 16 || d.cin % 16 || d.cout_pad < 32 || d.cout_pad % 32 || d.kh < 1 || d.kw < 1 || d.kh * d.kw > 64 ||
            d.stride < 1 || d.dil_h < 1 || d.dil_w < 1 || d.Ho < 1 || d.Wo < 1 || d.in_nseg < 1 || d.cout < 1)
            continue;
        bool legal = false;
        for (const ConvCfg& e : enumerate_cfgs(&d))
            if (e.NC == c.NC && e.lth == c.lth && e.ltw == c.ltw && e.ks == c.ks) { legal = true; break; }
        if (!legal) continue;
        tuned_store(k, c, true);
        ++n;
    }
    fclose(f);
    return n;
}
