// lstm.hip -- recurrent half of nn.LSTM(num_layers=1, bidirectional=True) on MFMA (gfx950).
//
// Reference: M1/networks.py:95,143-148 (input 2048, hidden 100) and M2/networks.py:64,88
// (input 3072, hidden 200); gate order i,f,g,o.  The input projection (x @ W_ih^T + b_ih + b_hh,
// > 99 % of the LSTM FLOPs) is done by sos_conv2d_fwd as a 1x1 conv; these kernels do the strictly
// sequential part, forward and BPTT.
//
// One persistent workgroup (8 waves) per (direction, 16 clips) walks the T steps.  The recurrent
// product of a step is v_mfma_f32_16x16x32_bf16 with the 16 clips as the N dimension:
//   forward   gates[4H x 16]  = W_hh [4H x H]  . h_{t-1} [H x 16]      (+ xproj)
//   backward  dh_rec[H x 16]  = W_hh^T [H x 4H] . dgates_t [4H x 16]
// W_hh is pre-packed once per weight version (sos_lstm_pack_whh) into the MFMA A-fragment order, so
// a fragment is one coalesced 1 KB load; it streams from L2 every step (320 KB per step for H = 200).
// The forward packs the rows of a 16-row tile as (unit u, gate q) -> row 4u + q: the accumulator
// layout (lane = clip + 16 * (row / 4), register = row % 4) then hands every lane the four gates
// of ONE (hidden unit, clip), so the cell update runs in registers with the cell state resident in
// a register; only h_t (bf16) goes through LDS, [clip][k], double buffered: one barrier per step.
// The backward gets dh_rec of (4 consecutive units, clip) per lane and does the gate-gradient
// arithmetic on it directly; dgates_t goes to global (f32, for the dW GEMMs) and to LDS (bf16,
// the next product's B operand).  With lo arrays given (bf16x3 mode) every product is the three
// passes hi*hi + hi*lo + lo*hi.
#include "sos_common.h"
#include <stdlib.h>

typedef sos_half_t bf16x8 __attribute__((ext_vector_type(8)));   // 8 storage-type (bf16, or fp16 in the SOS_F16 build) MFMA operands
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define LM_NB 16              // clips per workgroup (MFMA N)
#define LM_WAVES 8
#define LM_THREADS (LM_WAVES * 64)
#define LM_FT 8               // forward: 4-unit tiles per wave (H <= 256)
#define LM_BT 2               // backward: 16-unit tiles per wave
// ablation switches (SOS_LSTM_DBG: 1 no output stores, 2 no xproj loads, 4 one W tile re-read) only in `make ABLATE=1` builds
#ifdef SOS_ABLATE
#include <stdlib.h>
#define LDBG(bit) (dbg & (bit))
#else
#define LDBG(bit) 0
#endif

// hardware exp2 / reciprocal (1 ulp each): |err| ~2e-7, a dozen VALU cycles instead of ~100 for expf + IEEE division
__device__ __forceinline__ float sigmoidf_(float x) { return __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }
__device__ __forceinline__ float tanhf_(float x) { return 1.0f - 2.0f * __builtin_amdgcn_rcpf(1.0f + __expf(2.0f * x)); }
__device__ __forceinline__ bf16x8 frag(const uint4 v) { return __builtin_bit_cast(bf16x8, v); }

static inline int lm_kf(int H) { return (H + 31) / 32; }        // forward k-fragments (K = H)
static inline int lm_kb(int H) { return (4 * H + 31) / 32; }    // backward k-fragments (K = 4H)
static inline int lm_nj(int H) { return (H + 15) / 16; }        // backward 16-unit row tiles

// ---- packing: fwd [2][H/4 tiles][KF][64 lanes][8], bwd [2][NJ][KB][64][8] (bf16 hi, optional lo)
__global__ void lstm_pack_kernel(const float* __restrict__ whh, int H, bf16_t* __restrict__ fh, bf16_t* __restrict__ fl,
                                 bf16_t* __restrict__ bh, bf16_t* __restrict__ bl) {
    const int G = 4 * H, NT = H >> 2, KF = (H + 31) / 32, NJ = (H + 15) / 16, KB = (G + 31) / 32;
    const long long nf = 2LL * NT * KF * 512, nb = 2LL * NJ * KB * 512;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < nf + nb; i += (long long)gridDim.x * blockDim.x) {
        float v = 0.f;
        bf16_t *ph, *pl;
        if (i < nf) {
            const int e = (int)(i & 7), l = (int)((i >> 3) & 63);
            long long f = i >> 9;
            const int kk = (int)(f % KF); f /= KF;
            const int tile = (int)(f % NT), dir = (int)(f / NT);
            const int r = l & 15, row = (r & 3) * H + 4 * tile + (r >> 2), k = kk * 32 + 8 * (l >> 4) + e;
            if (k < H) v = whh[((size_t)dir * G + row) * H + k];
            ph = fh + i; pl = fl ? fl + i : nullptr;
        } else {
            const long long i2 = i - nf;
            const int e = (int)(i2 & 7), l = (int)((i2 >> 3) & 63);
            long long f = i2 >> 9;
            const int kk = (int)(f % KB); f /= KB;
            const int jt = (int)(f % NJ), dir = (int)(f / NJ);
            const int j = jt * 16 + (l & 15), r = kk * 32 + 8 * (l >> 4) + e;
            if (j < H && r < G) v = whh[((size_t)dir * G + r) * H + j];
            ph = bh + i2; pl = bl ? bl + i2 : nullptr;
        }
        const bf16_t hi = f2bf(v);
        *ph = hi;
        if (pl) *pl = f2bf(v - bf2f(hi));
    }
}

extern "C" int64_t sos_lstm_pack_bytes(int H, int backward) {
    if (H < 4 || H > 256 || (H & 3)) return -1;
    return backward ? 2LL * lm_nj(H) * lm_kb(H) * 1024 : 2LL * (H / 4) * lm_kf(H) * 1024;
}

extern "C" int sos_lstm_pack_whh(const float* whh, int H, void* fwd_hi, void* fwd_lo, void* bwd_hi, void* bwd_lo,
                                 sos_stream_t stream) {
    if (!whh || !fwd_hi || !bwd_hi || H < 4 || H > 256 || (H & 3) || ((fwd_lo == nullptr) != (bwd_lo == nullptr))) {
        sos_set_error("sos_lstm_pack_whh: bad args (H=%d)", H);
        return SOS_EINVAL;
    }
    hipLaunchKernelGGL(lstm_pack_kernel, dim3(512), dim3(256), 0, (hipStream_t)stream, whh, H, (bf16_t*)fwd_hi,
                       (bf16_t*)fwd_lo, (bf16_t*)bwd_hi, (bf16_t*)bwd_lo);
    return sos_check_launch("sos_lstm_pack_whh");
}

// ----------------------------------------------------------------------------------- forward
// xproj and dgates use the GATE-INTERLEAVED channel order [dir][unit j][i,f,g,o] (index dir*4H + 4j + q instead of
// torch's dir*4H + q*H + j): the four gate values a lane needs / produces are one 16-byte access.  The caller
// permutes the rows of W_ih and of the bias accordingly (and un-permutes the weight gradients).
// X3 = false (bf16): the W fragments of the NEXT tile (wrapping to the first tile of the next step: W does not
// depend on h) are in flight while the current tile is computed.  X3 = true (three-pass precision mode): hi and
// lo fragments of the current tile only (register budget).  In both, a tile's xproj values for the NEXT step
// are re-loaded right after this step consumed them.
// RES > 0 (plain 16-bit mode only): W_hh does not change over the T steps, so it is made RESIDENT instead of streamed from
// L2 every step: the first RES tiles of every wave live in registers (RES * KFT fragments of 4 VGPRs), the remaining
// tiles of the workgroup in LDS (one workgroup per CU: the whole 160 KB is its own) -- a step is then MFMAs, the h
// exchange through LDS and one barrier, no global weight traffic (H = 200: 320 KB per step before; 6.9 -> ~2 us / step).
// KFT = compile-time k-fragment count (register arrays need constant bounds).
template <bool X3, int RES = 0, int KFT = 8>
__global__ __launch_bounds__(LM_THREADS) void lstm_fwd_kernel(const float* __restrict__ xproj, const uint4* __restrict__ wh,
                                                               const uint4* __restrict__ wl, int B, int T, int H,
                                                               bf16_t* __restrict__ out, int out_cs, int out_x3,
                                                               long long third, float* __restrict__ save_gates,
                                                               float* __restrict__ save_c, int dbg,
                                                               const int* __restrict__ t_tab) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int KF = (H + 31) / 32, NT = H >> 2, G = 4 * H;
    const int HP = KF * 64 + 16;                         // bytes of one clip's h row (bf16, k padded to 32, +16: banks)
    constexpr int P = X3 ? 2 : 1;                        // hi (+ lo) planes
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int dir = blockIdx.y, b0 = blockIdx.x * LM_NB;
    const int n = lane & 15, g4 = lane >> 4;
    const int b = b0 + n;
    const bool live = b < B;
    // ragged batch: clip b has t_tab[b] <= T frames (rows past them are padding).  Every clip starts at step 0 from the
    // zero state -- the reverse direction at ITS last frame -- and simply goes idle (computes, stores nothing) after its
    // last step; the group runs to its longest clip.
    const int Tc = (t_tab && live) ? t_tab[b] : T;
    int Tg = T;
    if (t_tab) {
        Tg = 1;
        for (int c = 0; c < LM_NB; ++c)
            if (b0 + c < B) Tg = max(Tg, t_tab[b0 + c]);
    }
    for (int i = tid; i < 2 * P * LM_NB * HP / 4; i += LM_THREADS) ((unsigned*)smem)[i] = 0u;
    float creg[LM_FT];
#pragma unroll
    for (int ti = 0; ti < LM_FT; ++ti) creg[ti] = 0.f;
    const uint4* whd = wh + (size_t)dir * NT * KF * 64 + lane;
    const uint4* wld = X3 ? wl + (size_t)dir * NT * KF * 64 + lane : nullptr;
    const int cnt = NT > wave ? (NT - wave + LM_WAVES - 1) / LM_WAVES : 0;      // this wave's tiles (wave-uniform)
    // resident W_hh: register tiles, and the LDS image of the other tiles behind the two h buffers
    // (vector VALUES, one per register tile, not arrays: hipcc leaves a RES x KFT array of uint4 -- or four small ones
    // selected by the unrolled tile index -- in scratch memory, 17 us per step instead of 7)
    static_assert(RES <= 4, "register tiles are spelled out below");
    typedef unsigned wvec_t __attribute__((ext_vector_type(4 * KFT)));
    wvec_t wr0 = {}, wr1 = {}, wr2 = {}, wr3 = {};
    auto wput = [](wvec_t& w, const int kk, const uint4 v) { w[4 * kk] = v.x; w[4 * kk + 1] = v.y; w[4 * kk + 2] = v.z; w[4 * kk + 3] = v.w; };
    auto wget = [](const wvec_t& w, const int kk) { return make_uint4(w[4 * kk], w[4 * kk + 1], w[4 * kk + 2], w[4 * kk + 3]); };
    char* wlds = smem + (size_t)2 * P * LM_NB * HP;
    if constexpr (RES > 0) {
#pragma unroll
        for (int ti = 0; ti < LM_FT; ++ti) {
            if (ti >= cnt) continue;      // (not `break`: the loop must unroll completely for the register arrays)
            const int tile = wave + LM_WAVES * ti;
#pragma unroll
            for (int kk = 0; kk < KFT; ++kk) {
                const uint4 v = whd[((size_t)tile * KF + kk) * 64];
                if (ti == 0 && RES > 0) wput(wr0, kk, v);
                else if (ti == 1 && RES > 1) wput(wr1, kk, v);
                else if (ti == 2 && RES > 2) wput(wr2, kk, v);
                else if (ti == 3 && RES > 3) wput(wr3, kk, v);
                else *(uint4*)(wlds + ((size_t)(tile - RES * LM_WAVES) * KFT + kk) * 1024 + lane * 16) = v;
            }
        }
    }
    uint4 w0[8], w1[8];                                  // bf16: ping-pong of hi fragments; x3: hi and lo of the current tile
    f32x4 xc[LM_FT];
    auto load_frags = [&](uint4 (&w)[8], const uint4* base, const int tile) {
#pragma unroll
        for (int kk = 0; kk < 8; ++kk)
            if (kk < KF) w[kk] = base[((size_t)(LDBG(4) ? 0 : tile) * KF + kk) * 64];
    };
    auto load_x = [&](const int ti, const int step) {
        const int t = dir == 0 ? step : Tc - 1 - step;
        const float* xp = xproj + (((size_t)b * T + t) * 2 + dir) * G + ((wave + LM_WAVES * ti) * 4 + g4) * 4;
        if (live && step < Tc && !LDBG(2)) {
            const float4 x4 = *(const float4*)xp;          // gate-interleaved projection: one 16-byte load
            xc[ti] = f32x4{x4.x, x4.y, x4.z, x4.w};
        }
    };
#pragma unroll
    for (int ti = 0; ti < LM_FT; ++ti) {
        xc[ti] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (ti < cnt) load_x(ti, 0);
    }
    if (!X3 && RES == 0 && cnt > 0) load_frags(w0, whd, wave);
    __syncthreads();

    for (int step = 0; step < Tg; ++step) {
        const int t = dir == 0 ? step : Tc - 1 - step;
        const bool act = live && step < Tc;              // this lane's clip still has frames
        const char* hb = smem + (size_t)(step & 1) * P * LM_NB * HP;
        char* hn = smem + (size_t)((step + 1) & 1) * P * LM_NB * HP;
        // B fragments: h_{t-1}[k][clip], lane = clip + 16 * (k / 8)
        uint4 hf[8], hl[8];
#pragma unroll
        for (int kk = 0; kk < 8; ++kk) {
            if (kk < KF) {
                hf[kk] = *(const uint4*)(hb + n * HP + kk * 64 + g4 * 16);
                if (X3) hl[kk] = *(const uint4*)(hb + LM_NB * HP + n * HP + kk * 64 + g4 * 16);
            }
        }
        // h_{t-1} (this step's B operand) is also the previous step's OUTPUT: flush its rows to global here, as
        // whole 16-byte pieces of a clip's [dir*H, dir*H + H) run (H % 8 == 0; otherwise element stores below)
        if (step > 0 && (H & 7) == 0 && !LDBG(1)) {
            const int ppr = H >> 3;                      // pieces per clip row
            for (int i = tid; i < LM_NB * ppr; i += LM_THREADS) {
                const int c = i / ppr, q = i - c * ppr;
                const int Tcc = (t_tab && b0 + c < B) ? t_tab[b0 + c] : T;
                const int tp = dir == 0 ? step - 1 : Tcc - step;
                if (b0 + c < B && step <= Tcc) {
                    bf16_t* o = out + ((size_t)(b0 + c) * T + tp) * out_cs + dir * H + q * 8;
                    const uint4 hv = *(const uint4*)(hb + c * HP + q * 16);
                    *(uint4*)o = hv;
                    if (out_x3) { *(uint4*)(o + third) = hv; *(uint4*)(o + 2 * third) = *(const uint4*)(hb + LM_NB * HP + c * HP + q * 16); }
                }
            }
        }
        const size_t row = (size_t)b * T + t;
        // saved activations: kernel-native layout [clip group][t][dir][tile][clip][unit][i,f,g,o] -- one coalesced 1 KB store per tile
        const size_t nat = (((size_t)blockIdx.x * T + t) * 2 + dir) * NT;
#pragma unroll
        for (int ti = 0; ti < LM_FT; ++ti) {
            const int tile = wave + LM_WAVES * ti;       // wave-uniform
            if (ti >= cnt) continue;      // (not `break`: the loop must unroll completely for the register arrays)
            constexpr int dummy = 0; (void)dummy;
            if (X3) {
                load_frags(w0, whd, tile);
                load_frags(w1, wld, tile);
            } else if constexpr (RES == 0) {
                const int nxt = ti + 1 < cnt ? tile + LM_WAVES : wave;
                if (ti & 1) load_frags(w0, whd, nxt); else load_frags(w1, whd, nxt);
                __builtin_amdgcn_sched_barrier(0);       // keep the prefetch ahead of this tile's MFMAs
            }
            const int j = tile * 4 + g4;                 // this lane's hidden unit; registers = gates i,f,g,o
            f32x4 acc = xc[ti];
            load_x(ti, step + 1);
#pragma unroll
            for (int kk = 0; kk < 8; ++kk) {
                if (kk < KF) {
                    if (X3) {
                        acc = SOS_MFMA_16x16x32(frag(w0[kk]), frag(hf[kk]), acc, 0, 0, 0);
                        acc = SOS_MFMA_16x16x32(frag(w0[kk]), frag(hl[kk]), acc, 0, 0, 0);
                        acc = SOS_MFMA_16x16x32(frag(w1[kk]), frag(hf[kk]), acc, 0, 0, 0);
                    } else if constexpr (RES > 0) {
                        if (kk < KFT) {
                            constexpr int kq = 0;
                            (void)kq;
                            const int kc = kk < KFT ? kk : 0;
                            uint4 wv;
                            if (ti == 0 && RES > 0) wv = wget(wr0, kc);
                            else if (ti == 1 && RES > 1) wv = wget(wr1, kc);
                            else if (ti == 2 && RES > 2) wv = wget(wr2, kc);
                            else if (ti == 3 && RES > 3) wv = wget(wr3, kc);
                            else wv = *(const uint4*)(wlds + ((size_t)(tile - RES * LM_WAVES) * KFT + kk) * 1024 + lane * 16);
                            acc = SOS_MFMA_16x16x32(frag(wv), frag(hf[kk]), acc, 0, 0, 0);
                        }
                    } else {
                        acc = SOS_MFMA_16x16x32(frag((ti & 1) ? w1[kk] : w0[kk]), frag(hf[kk]), acc, 0, 0, 0);
                    }
                }
            }
            const float ig = sigmoidf_(acc[0]), fg = sigmoidf_(acc[1]), gt = tanhf_(acc[2]), og = sigmoidf_(acc[3]);
            const float c = fg * creg[ti] + ig * gt;
            const float h = og * tanhf_(c);
            creg[ti] = c;
            const bf16_t hi = f2bf(h);
            *(bf16_t*)(hn + n * HP + j * 2) = hi;
            bf16_t lo = 0;
            if (X3 || out_x3) lo = f2bf(h - bf2f(hi));
            if (X3) *(bf16_t*)(hn + LM_NB * HP + n * HP + j * 2) = lo;
            if (!LDBG(1)) {
                if (act && (H & 7)) {
                    bf16_t* o = out + row * out_cs + dir * H + j;
                    o[0] = hi;
                    if (out_x3) { o[third] = hi; o[2 * third] = lo; }
                }
                if (save_gates) {
                    *(float4*)(save_gates + ((nat + tile) * 64 + n * 4 + g4) * 4) = make_float4(ig, fg, gt, og);
                    save_c[(nat + tile) * 64 + n * 4 + g4] = c;
                }
            }
        }
        if (!X3 && RES == 0 && (cnt & 1)) {              // the wrap-around prefetch landed in w1: the next step starts in w0
#pragma unroll
            for (int kk = 0; kk < 8; ++kk) w0[kk] = w1[kk];
        }
        __syncthreads();
    }
    if ((H & 7) == 0 && !LDBG(1)) {                      // the last step's output (clips as long as the group)
        const char* hb = smem + (size_t)(Tg & 1) * P * LM_NB * HP;
        const int tp = dir == 0 ? Tg - 1 : 0;
        const int ppr = H >> 3;
        for (int i = tid; i < LM_NB * ppr; i += LM_THREADS) {
            const int c = i / ppr, q = i - c * ppr;
            const int Tcc = (t_tab && b0 + c < B) ? t_tab[b0 + c] : T;
            if (b0 + c < B && Tcc == Tg) {
                bf16_t* o = out + ((size_t)(b0 + c) * T + tp) * out_cs + dir * H + q * 8;
                const uint4 hv = *(const uint4*)(hb + c * HP + q * 16);
                *(uint4*)o = hv;
                if (out_x3) { *(uint4*)(o + third) = hv; *(uint4*)(o + 2 * third) = *(const uint4*)(hb + LM_NB * HP + c * HP + q * 16); }
            }
        }
    }
}

extern "C" int sos_lstm_bidir_fwd(const float* xproj, const void* wpk_hi, const void* wpk_lo, int64_t B, int64_t T, int H,
                                  void* out_bf16, int out_cs, int out_dtype, int64_t out_third, float* save_gates,
                                  float* save_c, const int32_t* lengths, sos_stream_t stream) {
    if (!xproj || !wpk_hi || !out_bf16 || B < 1 || T < 1 || H < 4 || H > 256 || (H & 3) || out_cs < 2 * H ||
        (out_dtype != SOS_DT_BF16 && out_dtype != SOS_DT_BF16X3) || ((save_gates == nullptr) != (save_c == nullptr)) ||
        ((out_dtype == SOS_DT_BF16X3) != (wpk_lo != nullptr)) || (out_cs & 7) || (out_third & 7)) {
        sos_set_error("sos_lstm_bidir_fwd: bad args (B=%lld T=%lld H=%d)", (long long)B, (long long)T, H);
        return SOS_EINVAL;
    }
    if (lengths && save_gates) {
        sos_set_error("sos_lstm_bidir_fwd: per-clip lengths are an inference feature (no saved activations)");
        return SOS_EINVAL;
    }
    dim3 grid((unsigned)((B + LM_NB - 1) / LM_NB), 2);
    int dbg = 0;
#ifdef SOS_ABLATE
    { const char* e = getenv("SOS_LSTM_DBG"); dbg = e ? atoi(e) : 0; }
#endif
    const size_t lds = (size_t)2 * (wpk_lo ? 2 : 1) * LM_NB * (lm_kf(H) * 64 + 16);
    // resident W_hh (plain 16-bit mode, the two hidden sizes of the reference: 100 and 200): 4 register tiles per wave, the
    // rest of the H/4 tiles in LDS
    constexpr int RES = 4;
    const int NT = H / 4, KF = lm_kf(H);
    const size_t wl_tiles = NT > RES * LM_WAVES ? (size_t)(NT - RES * LM_WAVES) : 0;
    const size_t lds_res = lds + wl_tiles * KF * 1024;
    static const char* no_res = getenv("SOS_LSTM_STREAM_W");          // A/B switch: stream W_hh from L2 as in round 1
    if (!wpk_lo && !no_res && lds_res <= 160 * 1024 && (KF == 7 || KF == 4)) {
        static sos_device_once once;
        (void)sos_per_device_once(once, [] {
            (void)hipFuncSetAttribute((const void*)lstm_fwd_kernel<false, RES, 7>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            (void)hipFuncSetAttribute((const void*)lstm_fwd_kernel<false, RES, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            return (int)SOS_OK;
        });
        if (KF == 7)
            hipLaunchKernelGGL((lstm_fwd_kernel<false, RES, 7>), grid, dim3(LM_THREADS), lds_res, (hipStream_t)stream, xproj,
                               (const uint4*)wpk_hi, (const uint4*)wpk_lo, (int)B, (int)T, H, (bf16_t*)out_bf16, out_cs, 0,
                               (long long)out_third, save_gates, save_c, dbg, lengths);
        else
            hipLaunchKernelGGL((lstm_fwd_kernel<false, RES, 4>), grid, dim3(LM_THREADS), lds_res, (hipStream_t)stream, xproj,
                               (const uint4*)wpk_hi, (const uint4*)wpk_lo, (int)B, (int)T, H, (bf16_t*)out_bf16, out_cs, 0,
                               (long long)out_third, save_gates, save_c, dbg, lengths);
        return sos_check_launch("sos_lstm_bidir_fwd");
    }
    if (wpk_lo)
        hipLaunchKernelGGL(lstm_fwd_kernel<true>, grid, dim3(LM_THREADS), lds, (hipStream_t)stream, xproj, (const uint4*)wpk_hi,
                           (const uint4*)wpk_lo, (int)B, (int)T, H, (bf16_t*)out_bf16, out_cs,
                           out_dtype == SOS_DT_BF16X3 ? 1 : 0, (long long)out_third, save_gates, save_c, dbg, lengths);
    else
        hipLaunchKernelGGL(lstm_fwd_kernel<false>, grid, dim3(LM_THREADS), lds, (hipStream_t)stream, xproj, (const uint4*)wpk_hi,
                           (const uint4*)wpk_lo, (int)B, (int)T, H, (bf16_t*)out_bf16, out_cs,
                           out_dtype == SOS_DT_BF16X3 ? 1 : 0, (long long)out_third, save_gates, save_c, dbg, lengths);
    return sos_check_launch("sos_lstm_bidir_fwd");
}

// ------------------------------------------------------------------------ backward through time
// Per step: A) per lane (4 consecutive hidden units, clip): dh = dh_out + dh_rec, dc = dc_rec +
// dh*o*(1-tanh(c)^2), gate pre-activation grads di,df,dg,do -> global dgates (f32 [B][T][2][4H]) and
// LDS [clip][4H] (bf16); dc_rec = dc*f.  B) dh_rec = W_hh^T . dgates on MFMA (accumulator layout = A's).
// dW_ih, dW_hh, the bias gradient and dx are GEMMs over dgates done by the wgrad / conv kernels.
__global__ __launch_bounds__(LM_THREADS) void lstm_bwd_kernel(const bf16_t* __restrict__ dh_out, int dh_cs, int dh_x3,
                                                               long long dh_third, const float* __restrict__ gates,
                                                               const float* __restrict__ csave, const uint4* __restrict__ wh,
                                                               const uint4* __restrict__ wl, int B, int T, int H,
                                                               float* __restrict__ dgates) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int G = 4 * H, NJ = (H + 15) / 16, KB = (G + 31) / 32;
    const int GP = KB * 64 + 16;                         // bytes of one clip's dgates row (bf16, padded)
    const int P = wl ? 2 : 1;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int dir = blockIdx.y, b0 = blockIdx.x * LM_NB;
    const int n = lane & 15, g4 = lane >> 4;
    const int b = b0 + n;
    const bool live = b < B;
    for (int i = tid; i < 2 * P * LM_NB * GP / 4; i += LM_THREADS) ((unsigned*)smem)[i] = 0u;
    f32x4 dhr[LM_BT], dcr[LM_BT];
#pragma unroll
    for (int ti = 0; ti < LM_BT; ++ti) { dhr[ti] = f32x4{0.f, 0.f, 0.f, 0.f}; dcr[ti] = dhr[ti]; }
    const uint4* whd = wh + (size_t)dir * NJ * KB * 64 + lane;
    const uint4* wld = wl ? wl + (size_t)dir * NJ * KB * 64 + lane : nullptr;
    __syncthreads();

    for (int step = 0; step < T; ++step) {
        const int t = dir == 0 ? T - 1 - step : step;     // reverse of the forward order
        const int tprev = dir == 0 ? t - 1 : t + 1;       // time index the forward pass came from
        char* dg = smem + (size_t)(step & 1) * P * LM_NB * GP;
        const size_t row = (size_t)b * T + t;
#pragma unroll
        for (int ti = 0; ti < LM_BT; ++ti) {
            const int jt = wave + LM_WAVES * ti;
            const int j0 = jt * 16 + 4 * g4;              // 4 consecutive hidden units (H % 4 == 0: all or none valid)
            if (jt >= NJ || j0 >= H || !live) continue;
            const bf16_t* dp = dh_out + row * dh_cs + dir * H + j0;
            const uint2 dhv = *(const uint2*)dp;
            float dh[4] = {bf2f((bf16_t)(dhv.x & 0xffffu)), bf2f((bf16_t)(dhv.x >> 16)), bf2f((bf16_t)(dhv.y & 0xffffu)),
                           bf2f((bf16_t)(dhv.y >> 16))};
            if (dh_x3) {
                const uint2 dl = *(const uint2*)(dp + 2 * dh_third);
                dh[0] += bf2f((bf16_t)(dl.x & 0xffffu)); dh[1] += bf2f((bf16_t)(dl.x >> 16));
                dh[2] += bf2f((bf16_t)(dl.y & 0xffffu)); dh[3] += bf2f((bf16_t)(dl.y >> 16));
            }
            // forward-native layout [tile = 4 units][clip][unit][i,f,g,o]: this lane's 4 units x 4 gates are 64 contiguous bytes
            const size_t nat = ((((size_t)blockIdx.x * T + t) * 2 + dir) * (H >> 2) + (j0 >> 2)) * 64 + n * 4;
            const size_t natp = ((((size_t)blockIdx.x * T + tprev) * 2 + dir) * (H >> 2) + (j0 >> 2)) * 64 + n * 4;
            float igv[4], fgv[4], gtv[4], ogv[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const float4 g4v = *(const float4*)(gates + (nat + u) * 4);
                igv[u] = g4v.x; fgv[u] = g4v.y; gtv[u] = g4v.z; ogv[u] = g4v.w;
            }
            const float4 c4 = *(const float4*)(csave + nat);
            float4 cp4 = make_float4(0.f, 0.f, 0.f, 0.f);
            if (tprev >= 0 && tprev < T) cp4 = *(const float4*)(csave + natp);
            const float cv[4] = {c4.x, c4.y, c4.z, c4.w}, cpv[4] = {cp4.x, cp4.y, cp4.z, cp4.w};
            float di[4], df[4], dgg[4], dov[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float dhe = dh[e] + dhr[ti][e];
                const float tc = tanhf_(cv[e]);
                const float dc = dcr[ti][e] + dhe * ogv[e] * (1.f - tc * tc);
                di[e] = dc * gtv[e] * igv[e] * (1.f - igv[e]);
                df[e] = dc * cpv[e] * fgv[e] * (1.f - fgv[e]);
                dgg[e] = dc * igv[e] * (1.f - gtv[e] * gtv[e]);
                dov[e] = dhe * tc * ogv[e] * (1.f - ogv[e]);
                dcr[ti][e] = dc * fgv[e];
            }
            float* dgp = dgates + (row * 2 + dir) * G + j0 * 4;     // gate-interleaved: [unit][i,f,g,o], 64 contiguous bytes
#pragma unroll
            for (int u = 0; u < 4; ++u) *(float4*)(dgp + 4 * u) = make_float4(di[u], df[u], dgg[u], dov[u]);
            const float* q4[4] = {di, df, dgg, dov};
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const unsigned h01 = pack2bf(q4[q][0], q4[q][1]), h23 = pack2bf(q4[q][2], q4[q][3]);
                *(uint2*)(dg + n * GP + (q * H + j0) * 2) = make_uint2(h01, h23);
                if (wl) {
                    const unsigned l01 = pack2bf(q4[q][0] - bf2f((bf16_t)(h01 & 0xffffu)), q4[q][1] - bf2f((bf16_t)(h01 >> 16)));
                    const unsigned l23 = pack2bf(q4[q][2] - bf2f((bf16_t)(h23 & 0xffffu)), q4[q][3] - bf2f((bf16_t)(h23 >> 16)));
                    *(uint2*)(dg + LM_NB * GP + n * GP + (q * H + j0) * 2) = make_uint2(l01, l23);
                }
            }
        }
        __syncthreads();
#pragma unroll
        for (int ti = 0; ti < LM_BT; ++ti) {
            const int jt = wave + LM_WAVES * ti;
            if (jt >= NJ) break;
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
            // chunks of 4 k-fragments, the next chunk's W loads in flight while this chunk is multiplied
            const uint4* wp = whd + (size_t)jt * KB * 64;
            const uint4* wp2 = wl ? wld + (size_t)jt * KB * 64 : nullptr;
            uint4 wa[4], wc2[4], la[4], lc2[4];
            auto load4 = [&](uint4 (&w)[4], uint4 (&l)[4], const int k0) {
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int kk = min(k0 + u, KB - 1);
                    w[u] = wp[(size_t)kk * 64];
                    if (wl) l[u] = wp2[(size_t)kk * 64];
                }
            };
            auto mul4 = [&](const uint4 (&w)[4], const uint4 (&l)[4], const int k0) {
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int kk = k0 + u;
                    if (kk < KB) {
                        const uint4 d0 = *(const uint4*)(dg + n * GP + kk * 64 + g4 * 16);
                        acc = SOS_MFMA_16x16x32(frag(w[u]), frag(d0), acc, 0, 0, 0);
                        if (wl) {
                            const uint4 d1 = *(const uint4*)(dg + LM_NB * GP + n * GP + kk * 64 + g4 * 16);
                            acc = SOS_MFMA_16x16x32(frag(w[u]), frag(d1), acc, 0, 0, 0);
                            acc = SOS_MFMA_16x16x32(frag(l[u]), frag(d0), acc, 0, 0, 0);
                        }
                    }
                }
            };
            load4(wa, la, 0);
            for (int k0 = 0; k0 < KB; k0 += 8) {
                load4(wc2, lc2, k0 + 4);
                __builtin_amdgcn_sched_barrier(0);
                mul4(wa, la, k0);
                load4(wa, la, k0 + 8);
                __builtin_amdgcn_sched_barrier(0);
                mul4(wc2, lc2, k0 + 4);
            }
            dhr[ti] = acc;
        }
    }
}

extern "C" int sos_lstm_bidir_bwd(const void* dh_out, int dh_cs, int dh_dtype, int64_t dh_third, const float* gates,
                                  const float* csave, const void* wtk_hi, const void* wtk_lo, int64_t B, int64_t T, int H,
                                  float* dgates, sos_stream_t stream) {
    if (!dh_out || !gates || !csave || !wtk_hi || !dgates || B < 1 || T < 1 || H < 4 || H > 256 || (H & 3) || dh_cs < 2 * H ||
        (dh_cs & 3) || (dh_dtype != SOS_DT_BF16 && dh_dtype != SOS_DT_BF16X3)) {
        sos_set_error("sos_lstm_bidir_bwd: bad args");
        return SOS_EINVAL;
    }
    const size_t lds = (size_t)2 * (wtk_lo ? 2 : 1) * LM_NB * (lm_kb(H) * 64 + 16);
    if (lds > 160 * 1024) { sos_set_error("sos_lstm_bidir_bwd: LDS"); return SOS_ENOSPC; }
    static sos_device_once attr_once;
    (void)sos_per_device_once(attr_once, [] {
        (void)hipFuncSetAttribute((const void*)lstm_bwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        return SOS_OK;
    });
    dim3 grid((unsigned)((B + LM_NB - 1) / LM_NB), 2);
    hipLaunchKernelGGL(lstm_bwd_kernel, grid, dim3(LM_THREADS), lds, (hipStream_t)stream, (const bf16_t*)dh_out, dh_cs,
                       dh_dtype == SOS_DT_BF16X3 ? 1 : 0, (long long)dh_third, gates, csave, (const uint4*)wtk_hi,
                       (const uint4*)wtk_lo, (int)B, (int)T, H, dgates);
    return sos_check_launch("sos_lstm_bidir_bwd");
}
