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

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

struct ConvParams {
    const bf16_t* in;
    const bf16_t* wgt;
    void* out;
    const float* scale;
    const float* shift;
    const float* slope;
    const int* wgather;
    int B, H, W, Wl, in_cs, cin_off, cin, ktot, cps, seg_stride;
    int kh, kw, cout, cout_pad, cout_store;
    int stride, dh, dw, pad_t, pad_l, pad_mode;
    int Ho, Wo;
    int out_dtype, c_off, act;
    long long sb, sh, sw, sc, third;
    // tiling
    int NC, logTH, logTW, PH, PW;
    int nchunks;            // cin / (16*KS)
    int npix;               // NC*PH*PW
    int tiles_h, tiles_w, ngw;
    int nblk;               // grid.x
};

__device__ __forceinline__ bf16x8 lds_frag(const char* p) {
    return __builtin_bit_cast(bf16x8, *(const uint4*)p);
}

template <int NT, int KS>
__global__ __launch_bounds__(256) void conv_mfma_kernel(ConvParams p) {
    constexpr int KC = 16 * KS;                 // channels per chunk
    constexpr int PSTRIDE = KC * 2 + 16;        // bytes per patch pixel row (padded)
    constexpr int BSTRIDE = KC * 2 + 16;        // bytes per weight row (padded)
    constexpr int CPR = 2 * KS;                 // 16-byte pieces per row
    constexpr int BROWS = NT * 32;
    constexpr int BPIECES = BROWS * CPR;
    constexpr int NBREG = (BPIECES + 255) / 256;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* patch = smem;
    char* bbuf[2];
    bbuf[0] = smem + (size_t)p.npix * PSTRIDE;
    bbuf[1] = bbuf[0] + BROWS * BSTRIDE;

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
    int t = bid;
    const int tj = t % p.tiles_w; t /= p.tiles_w;
    const int gw = t % p.ngw; t /= p.ngw;
    const int ti = t % p.tiles_h; t /= p.tiles_h;
    const int rh = t % p.dh; t /= p.dh;
    const int b = t;
    const int n0 = blockIdx.y * BROWS;

    const int TH = 1 << p.logTH, TW = 1 << p.logTW;
    const int rw0 = gw * p.NC;
    // first output row/col (class 0 of the group) of this tile, and the input coordinate of patch (0,0)
    const int ho_base = rh + ti * TH * p.dh;
    const int wo_base = rw0 + tj * TW * p.dw;
    const int hin0 = ho_base * p.stride - p.pad_t;
    const int win0 = wo_base * p.stride - p.pad_l;

    // ---- per-lane pixel operand base addresses (tap (0,0), k-step 0)
    int abase[2];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
        const int m = wave * 64 + mt * 32 + l31;
        const int j = m & (TW - 1);
        const int i = (m >> p.logTW) & (TH - 1);
        const int cls = m >> (p.logTW + p.logTH);
        abase[mt] = ((cls * p.PH + i * p.stride) * p.PW + j * p.stride) * PSTRIDE + lhi * 16;
    }
    const int bfrag_off = l31 * BSTRIDE + lhi * 16;

    f32x16 acc[2][NT];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[mt][nt][e] = 0.f;

    const int ntaps = p.kh * p.kw;
    const long long in_b = (long long)b * p.H * p.W;

    for (int cc = 0; cc < p.nchunks; ++cc) {
        __syncthreads();   // everyone is done reading the previous chunk's patch / weight buffers
        // ---- stage the input patch of this channel chunk: 16 lanes per pixel, 16 B per lane
        {
            const int cl = tid & 15;
            const long long cbase = (long long)p.cin_off + (long long)(cc / p.cps) * p.seg_stride + (long long)(cc % p.cps) * KC;
            for (int pix = tid >> 4; pix < p.npix; pix += 16) {
                const int c = pix % p.PW;
                const int rr = pix / p.PW;
                const int r = rr % p.PH;
                const int cls = rr / p.PH;
                int h = hin0 + r * p.dh;
                int w = win0 + cls * p.stride + c * p.dw;
                bool ok = (rw0 + cls) < p.dw || cls == 0;
                if (p.pad_mode == SOS_PAD_REFLECT) {
                    h = reflect_index(h, p.H);
                    w = reflect_index(w, p.Wl);
                } else {
                    ok = ok && h >= 0 && h < p.H && w >= 0 && w < p.Wl;
                }
                if (ok && p.wgather) w = p.wgather[w];
                const bf16_t* src = p.in + ((in_b + (long long)h * p.W + w) * p.in_cs + cbase);
                char* dst = patch + (size_t)pix * PSTRIDE;
                for (int q = cl; q < CPR; q += 16) {
                    uint4 v = make_uint4(0u, 0u, 0u, 0u);
                    if (ok) v = *(const uint4*)(src + q * 8);
                    *(uint4*)(dst + q * 16) = v;
                }
            }
        }
        // ---- weight slab of tap 0 straight into buffer 0
        {
            const bf16_t* wsrc = p.wgt + ((long long)n0 * p.ktot + (long long)cc * KC);
            for (int idx = tid; idx < BPIECES; idx += 256) {
                const int row = idx / CPR, q = idx - row * CPR;
                uint4 v = make_uint4(0u, 0u, 0u, 0u);
                if (n0 + row < p.cout_pad) v = *(const uint4*)(wsrc + (long long)row * p.ktot + q * 8);
                *(uint4*)(bbuf[0] + row * BSTRIDE + q * 16) = v;
            }
        }
        __syncthreads();

        int ta = 0, tb = 0;   // tap row / col
        for (int tap = 0; tap < ntaps; ++tap) {
            const int cur = tap & 1;
            // prefetch the next tap's weight slab into registers (lands while the MFMAs run)
            uint4 breg[NBREG];
            const bool more = tap + 1 < ntaps;
            if (more) {
                const bf16_t* wsrc = p.wgt + (((long long)(tap + 1) * p.cout_pad + n0) * p.ktot + (long long)cc * KC);
#pragma unroll
                for (int u = 0; u < NBREG; ++u) {
                    const int idx = tid + u * 256;
                    const int row = idx / CPR, q = idx - row * CPR;
                    breg[u] = make_uint4(0u, 0u, 0u, 0u);
                    if (idx < BPIECES && n0 + row < p.cout_pad)
                        breg[u] = *(const uint4*)(wsrc + (long long)row * p.ktot + q * 8);
                }
            }
            const int toff = (ta * p.PW + tb) * PSTRIDE;
            const char* a0p = patch + abase[0] + toff;
            const char* a1p = patch + abase[1] + toff;
            const char* bp = bbuf[cur] + bfrag_off;
#pragma unroll
            for (int kk = 0; kk < KS; ++kk) {
                const bf16x8 a0 = lds_frag(a0p + kk * 32);
                const bf16x8 a1 = lds_frag(a1p + kk * 32);
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    const bf16x8 wf = lds_frag(bp + nt * 32 * BSTRIDE + kk * 32);
                    acc[0][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf, a0, acc[0][nt], 0, 0, 0);
                    acc[1][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf, a1, acc[1][nt], 0, 0, 0);
                }
            }
            if (more) {
#pragma unroll
                for (int u = 0; u < NBREG; ++u) {
                    const int idx = tid + u * 256;
                    const int row = idx / CPR, q = idx - row * CPR;
                    if (idx < BPIECES) *(uint4*)(bbuf[cur ^ 1] + row * BSTRIDE + q * 16) = breg[u];
                }
            }
            __syncthreads();
            if (++tb == p.kw) { tb = 0; ++ta; }
        }
    }

    // ---- fused epilogue.  Accumulator layout of v_mfma_f32_32x32x*: column = lane&31 (pixel),
    // row = (reg&3) + 8*(reg>>2) + 4*(lane>>5) (output channel within the 32-row tile).
    const float slope = (p.act == SOS_ACT_PRELU && p.slope) ? p.slope[0] : 0.f;
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
        const int m = wave * 64 + mt * 32 + l31;
        const int j = m & (TW - 1);
        const int i = (m >> p.logTW) & (TH - 1);
        const int cls = m >> (p.logTW + p.logTH);
        const int ho = ho_base + i * p.dh;
        const int wo = wo_base + cls + j * p.dw;
        const bool pix_ok = ho < p.Ho && wo < p.Wo && (cls == 0 || rw0 + cls < p.dw);
        if (!pix_ok) continue;
        const long long obase = (long long)b * p.sb + (long long)ho * p.sh + (long long)wo * p.sw;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int co = n0 + nt * 32 + g * 8 + lhi * 4;   // 4 consecutive channels co..co+3
                if (co >= p.cout_store) continue;
                float v[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int c = co + e;
                    float y = 0.f;
                    if (c < p.cout) {
                        y = fmaf(acc[mt][nt][g * 4 + e], p.scale[c], p.shift[c]);
                        if (p.act == SOS_ACT_RELU) y = fmaxf(y, 0.f);
                        else if (p.act == SOS_ACT_PRELU) y = y >= 0.f ? y : slope * y;
                        else if (p.act == SOS_ACT_SIGMOID) y = 1.0f / (1.0f + expf(-y));
                    }
                    v[e] = y;
                }
                const long long o = obase + (long long)(p.c_off + co) * p.sc;
                if (p.out_dtype == SOS_DT_F32) {
                    float* op = (float*)p.out;
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (co + e < p.cout_store) op[o + e * p.sc] = v[e];
                } else {
                    bf16_t* op = (bf16_t*)p.out;
                    bf16_t hi[4], lo[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        hi[e] = f2bf(v[e]);
                        lo[e] = f2bf(v[e] - bf2f(hi[e]));
                    }
                    if (p.sc == 1 && co + 3 < p.cout_store) {
                        const uint2 hv = make_uint2((unsigned)hi[0] | ((unsigned)hi[1] << 16),
                                                    (unsigned)hi[2] | ((unsigned)hi[3] << 16));
                        *(uint2*)(op + o) = hv;
                        if (p.out_dtype == SOS_DT_BF16X3) {
                            *(uint2*)(op + o + p.third) = hv;
                            *(uint2*)(op + o + 2 * p.third) = make_uint2((unsigned)lo[0] | ((unsigned)lo[1] << 16),
                                                                          (unsigned)lo[2] | ((unsigned)lo[3] << 16));
                        }
                    } else {
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            if (co + e >= p.cout_store) continue;
                            op[o + e * p.sc] = hi[e];
                            if (p.out_dtype == SOS_DT_BF16X3) {
                                op[o + e * p.sc + p.third] = hi[e];
                                op[o + e * p.sc + 2 * p.third] = lo[e];
                            }
                        }
                    }
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------ host side
static const size_t LDS_LIMIT = 160 * 1024;

typedef void (*conv_kernel_t)(ConvParams);

template <int NT, int KS>
static int launch_one(const ConvParams& p, dim3 grid, size_t lds, hipStream_t stream) {
    static bool attr_set = false;
    conv_kernel_t k = conv_mfma_kernel<NT, KS>;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_LIMIT);
        if (e != hipSuccess) {
            sos_set_error("sos_conv2d_fwd: hipFuncSetAttribute failed: %s", hipGetErrorString(e));
            return SOS_ELAUNCH;
        }
        attr_set = true;
    }
    hipLaunchKernelGGL(k, grid, dim3(256), lds, stream, p);
    return sos_check_launch("sos_conv2d_fwd");
}

template <int NT>
static int launch_ks(int ks, const ConvParams& p, dim3 grid, size_t lds, hipStream_t stream) {
    switch (ks) {
        case 1: return launch_one<NT, 1>(p, grid, lds, stream);
        case 2: return launch_one<NT, 2>(p, grid, lds, stream);
        case 3: return launch_one<NT, 3>(p, grid, lds, stream);
        case 4: return launch_one<NT, 4>(p, grid, lds, stream);
        case 5: return launch_one<NT, 5>(p, grid, lds, stream);
        case 6: return launch_one<NT, 6>(p, grid, lds, stream);
        case 8: return launch_one<NT, 8>(p, grid, lds, stream);
    }
    sos_set_error("sos_conv2d_fwd: unsupported k-steps %d", ks);
    return SOS_EINVAL;
}

static size_t lds_bytes(int npix, int nt, int ks) {
    const size_t row = (size_t)ks * 32 + 16;
    return (size_t)npix * row + 2 * (size_t)nt * 32 * row;
}

static int pick_ks(int cin, int npix, int nt) {
    static const int cand[] = {8, 6, 5, 4, 3, 2, 1};
    const int k16 = cin / 16;
    for (int c : cand)
        if (k16 % c == 0 && lds_bytes(npix, nt, c) <= LDS_LIMIT) return c;
    return 0;
}

extern "C" int sos_conv2d_fwd(const sos_conv_desc* d, sos_stream_t stream) {
    if (!d || !d->in || !d->wgt || !d->out || !d->scale || !d->shift) {
        sos_set_error("sos_conv2d_fwd: null pointer");
        return SOS_EINVAL;
    }
    if (d->cin < 16 || d->cin % 16 || d->in_cs % 8 || d->cin_off % 8 || d->cout_pad % 32 || d->cout > d->cout_pad ||
        d->cout < 1 || d->kh < 1 || d->kw < 1 || d->stride < 1 || d->dil_h < 1 || d->dil_w < 1 ||
        (d->stride > 1 && (d->dil_h > 1 || d->dil_w > 1)) || d->B < 1 || d->Ho < 1 || d->Wo < 1 ||
        d->in_nseg < 1 || d->in_seg_stride % 8 || d->cin_off + (d->in_nseg - 1) * d->in_seg_stride + d->cin > d->in_cs || d->cout_store < d->cout || d->out_dtype < 0 || d->out_dtype > 2 ||
        (d->w_gather == nullptr && d->Wl != d->W)) {
        sos_set_error("sos_conv2d_fwd: bad descriptor (cin=%d in_cs=%d cin_off=%d cout=%d/%d k=%dx%d s=%d d=%dx%d)",
                      d->cin, d->in_cs, d->cin_off, d->cout, d->cout_pad, d->kh, d->kw, d->stride, d->dil_h, d->dil_w);
        return SOS_EINVAL;
    }
    if (d->pad_mode == SOS_PAD_REFLECT && (d->pad_top >= d->H || d->pad_left >= d->Wl)) {
        sos_set_error("sos_conv2d_fwd: reflect pad %d/%d needs a larger input (%dx%d)", d->pad_top, d->pad_left, d->H, d->Wl);
        return SOS_EINVAL;
    }
    ConvParams p;
    p.in = (const bf16_t*)d->in; p.wgt = (const bf16_t*)d->wgt; p.out = d->out;
    p.scale = d->scale; p.shift = d->shift; p.slope = d->act_param; p.wgather = d->w_gather;
    p.B = d->B; p.H = d->H; p.W = d->W; p.Wl = d->Wl; p.in_cs = d->in_cs; p.cin_off = d->cin_off; p.cin = d->cin;
    p.kh = d->kh; p.kw = d->kw; p.cout = d->cout; p.cout_pad = d->cout_pad; p.cout_store = d->cout_store;
    p.stride = d->stride; p.dh = d->dil_h; p.dw = d->dil_w; p.pad_t = d->pad_top; p.pad_l = d->pad_left;
    p.pad_mode = d->pad_mode; p.Ho = d->Ho; p.Wo = d->Wo; p.out_dtype = d->out_dtype; p.c_off = d->out_c_off;
    p.act = d->act; p.sb = d->out_sb; p.sh = d->out_sh; p.sw = d->out_sw; p.sc = d->out_sc; p.third = d->out_third;

    // output-channel tiles per block: as many as fit, balanced over the n-blocks
    const int ntiles = d->cout_pad / 32;
    const int nby = (ntiles + 3) / 4;
    const int nt = (ntiles + nby - 1) / nby;

    // ---- choose the pixel tile (NC classes x TH x TW = 256) by a cost model: MFMA work wasted on
    // out-of-range pixels versus patch traffic (halo amplification).
    const int Hc = (d->Ho + d->dil_h - 1) / d->dil_h, Wc = (d->Wo + d->dil_w - 1) / d->dil_w;
    const int taps = d->kh * d->kw;
    double best = 1e300;
    int bNC = 0, bTH = 0, bTW = 0, bKS = 0;
    for (int lnc = 0; lnc <= 8; ++lnc) {
        const int NC = 1 << lnc;
        if (NC > 1 && (d->stride > 1 || NC > d->dil_w)) break;
        for (int lth = 0; lth + lnc <= 8; ++lth) {
            const int ltw = 8 - lnc - lth;
            const int TH = 1 << lth, TW = 1 << ltw;
            const int PH = (TH - 1) * d->stride + d->kh, PW = (TW - 1) * d->stride + d->kw;
            const int npix = NC * PH * PW;
            const int ks = pick_ks(d->cin, npix, nt);
            if (!ks) continue;
            const long long th = (Hc + TH - 1) / TH, tw = (Wc + TW - 1) / TW, ngw = (d->dil_w + NC - 1) / NC;
            const double blocks = (double)th * tw * ngw * d->dil_h;
            // per block: MFMA time ~ 256*taps ; patch load ~ 3*npix ; per-chunk sync/latency overhead
            const double cost = blocks * (256.0 * taps + 3.0 * npix + 64.0 * (d->in_nseg * d->cin / (16 * ks)) * (1 + taps / 4.0));
            if (cost < best) { best = cost; bNC = NC; bTH = lth; bTW = ltw; bKS = ks; }
        }
    }
    if (!bKS) {
        sos_set_error("sos_conv2d_fwd: no tile fits LDS (cin=%d k=%dx%d)", d->cin, d->kh, d->kw);
        return SOS_ENOSPC;
    }
    p.NC = bNC; p.logTH = bTH; p.logTW = bTW;
    const int TH = 1 << bTH, TW = 1 << bTW;
    p.PH = (TH - 1) * d->stride + d->kh; p.PW = (TW - 1) * d->stride + d->kw;
    p.npix = p.NC * p.PH * p.PW;
    p.cps = d->cin / (16 * bKS);
    p.nchunks = p.cps * d->in_nseg;
    p.ktot = d->cin * d->in_nseg;
    p.seg_stride = d->in_seg_stride;
    p.tiles_h = (Hc + TH - 1) / TH; p.tiles_w = (Wc + TW - 1) / TW; p.ngw = (d->dil_w + p.NC - 1) / p.NC;
    const long long nblk = (long long)d->B * d->dil_h * p.tiles_h * p.ngw * p.tiles_w;
    if (nblk > 0x7fffffffLL) { sos_set_error("sos_conv2d_fwd: grid too large"); return SOS_EINVAL; }
    p.nblk = (int)nblk;
    dim3 grid((unsigned)nblk, (unsigned)nby);
    const size_t lds = lds_bytes(p.npix, nt, bKS);
    hipStream_t s = (hipStream_t)stream;
    switch (nt) {
        case 1: return launch_ks<1>(bKS, p, grid, lds, s);
        case 2: return launch_ks<2>(bKS, p, grid, lds, s);
        case 3: return launch_ks<3>(bKS, p, grid, lds, s);
        case 4: return launch_ks<4>(bKS, p, grid, lds, s);
    }
    sos_set_error("sos_conv2d_fwd: internal: nt=%d", nt);
    return SOS_EINVAL;
}
