// lstm.hip -- recurrent half of nn.LSTM(num_layers=1, bidirectional=True), fp32 (gfx950).
//
// Reference: M1/networks.py:95,143-148 (input 2048, hidden 100) and M2/networks.py:64,88
// (input 3072, hidden 200); gate order i,f,g,o.  The input projection (x @ W_ih^T + b_ih + b_hh,
// > 99 % of the LSTM FLOPs) is done by sos_conv2d_fwd as a 1x1 conv on MFMA; this kernel does
// the strictly sequential part: one persistent workgroup per (direction, slice of NB clips) walks
// the T steps; thread j owns hidden unit j for all NB clips (cell state in registers), h_{t-1} is
// exchanged through a double-buffered LDS tile laid out [k][NB] so one 32-byte broadcast read
// feeds 8 clips, and W_hh^T ([k][4H], coalesced over j) streams from L2 every step.
#include "sos_common.h"

#define LSTM_NB 8

__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }

// Thread layout: thread = (gate q, group of 4 consecutive hidden units) -> one 16-byte W_hh^T load
// per k feeds 4 units x NB clips; the four gate pre-activations of a unit meet in LDS, then the
// first H threads do the cell update (cell state lives in LDS, [j][NB]).
__global__ __launch_bounds__(256) void lstm_kernel(const float* __restrict__ xproj, const float* __restrict__ whh_t,
                                                   int B, int T, int H, float* __restrict__ out_f32,
                                                   bf16_t* __restrict__ out_bf16, int out_cs, int x3,
                                                   long long third, float* __restrict__ save_gates,
                                                   float* __restrict__ save_c) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* hbuf = (float*)smem;                           // [2][H][NB]
    float* gbuf = hbuf + 2 * H * LSTM_NB;                 // [4H][NB] gate pre-activations
    float* cbuf = gbuf + 4 * H * LSTM_NB;                 // [H][NB] cell state
    const int tid = threadIdx.x;
    const int dir = blockIdx.y;
    const int b0 = blockIdx.x * LSTM_NB;
    const int G = 4 * H;
    const float* W = whh_t + (size_t)dir * H * G;         // [k][4H]
    const int nact = G >> 2;                               // active threads (4 gate columns each)
    const int col0 = tid * 4;                              // first gate column of this thread
    for (int idx = tid; idx < 2 * H * LSTM_NB + 4 * H * LSTM_NB + H * LSTM_NB; idx += blockDim.x) hbuf[idx] = 0.f;
    __syncthreads();

    for (int step = 0; step < T; ++step) {
        const int t = dir == 0 ? step : T - 1 - step;
        const float* hcur = hbuf + (size_t)(step & 1) * H * LSTM_NB;
        float* hnext = hbuf + (size_t)((step + 1) & 1) * H * LSTM_NB;
        if (tid < nact) {
            float g[4][LSTM_NB];
#pragma unroll
            for (int n = 0; n < LSTM_NB; ++n) {
                const int b = b0 + n;
                float4 x = make_float4(0.f, 0.f, 0.f, 0.f);
                if (b < B) x = *(const float4*)(xproj + (((size_t)b * T + t) * 2 + dir) * G + col0);
                g[0][n] = x.x; g[1][n] = x.y; g[2][n] = x.z; g[3][n] = x.w;
            }
            const float* wp = W + col0;
#pragma unroll 8
            for (int k = 0; k < H; ++k) {
                const float4 w = *(const float4*)(wp + (size_t)k * G);
                const float4 ha = *(const float4*)(hcur + k * LSTM_NB);
                const float4 hb = *(const float4*)(hcur + k * LSTM_NB + 4);
                const float hv[8] = {ha.x, ha.y, ha.z, ha.w, hb.x, hb.y, hb.z, hb.w};
#pragma unroll
                for (int n = 0; n < LSTM_NB; ++n) {
                    g[0][n] = fmaf(w.x, hv[n], g[0][n]);
                    g[1][n] = fmaf(w.y, hv[n], g[1][n]);
                    g[2][n] = fmaf(w.z, hv[n], g[2][n]);
                    g[3][n] = fmaf(w.w, hv[n], g[3][n]);
                }
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                *(float4*)(gbuf + (col0 + e) * LSTM_NB) = make_float4(g[e][0], g[e][1], g[e][2], g[e][3]);
                *(float4*)(gbuf + (col0 + e) * LSTM_NB + 4) = make_float4(g[e][4], g[e][5], g[e][6], g[e][7]);
            }
        }
        __syncthreads();
        // cell update: thread (j, half) handles 4 clips of hidden unit j
        for (int idx = tid; idx < 2 * H; idx += blockDim.x) {
            const int j = idx >> 1, n0 = (idx & 1) * 4;
            const float4 gi = *(const float4*)(gbuf + (0 * H + j) * LSTM_NB + n0);
            const float4 gf = *(const float4*)(gbuf + (1 * H + j) * LSTM_NB + n0);
            const float4 gg = *(const float4*)(gbuf + (2 * H + j) * LSTM_NB + n0);
            const float4 go = *(const float4*)(gbuf + (3 * H + j) * LSTM_NB + n0);
            float4 c4 = *(const float4*)(cbuf + j * LSTM_NB + n0);
            const float iv[4] = {gi.x, gi.y, gi.z, gi.w}, fv[4] = {gf.x, gf.y, gf.z, gf.w};
            const float gv[4] = {gg.x, gg.y, gg.z, gg.w}, ov[4] = {go.x, go.y, go.z, go.w};
            float cv[4] = {c4.x, c4.y, c4.z, c4.w}, hv[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float ig = sigmoidf_(iv[e]), fg = sigmoidf_(fv[e]), gt = tanhf(gv[e]), og = sigmoidf_(ov[e]);
                cv[e] = fg * cv[e] + ig * gt;
                hv[e] = og * tanhf(cv[e]);
                const int bb = b0 + n0 + e;
                if (save_gates && bb < B) {
                    float* sg = save_gates + (((size_t)bb * T + t) * 2 + dir) * G;
                    sg[j] = ig; sg[H + j] = fg; sg[2 * H + j] = gt; sg[3 * H + j] = og;
                    save_c[(((size_t)bb * T + t) * 2 + dir) * H + j] = cv[e];
                }
            }
            *(float4*)(cbuf + j * LSTM_NB + n0) = make_float4(cv[0], cv[1], cv[2], cv[3]);
            *(float4*)(hnext + j * LSTM_NB + n0) = make_float4(hv[0], hv[1], hv[2], hv[3]);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int b = b0 + n0 + e;
                if (b >= B) continue;
                const size_t row = (size_t)b * T + t;
                if (out_f32) out_f32[row * (2 * H) + dir * H + j] = hv[e];
                if (out_bf16) {
                    bf16_t* o = out_bf16 + row * out_cs + dir * H + j;
                    const bf16_t hi = f2bf(hv[e]);
                    o[0] = hi;
                    if (x3) {
                        o[third] = hi;
                        o[2 * third] = f2bf(hv[e] - bf2f(hi));
                    }
                }
            }
        }
        __syncthreads();
    }
}

extern "C" int sos_lstm_bidir_fwd(const float* xproj, const float* whh_t, int64_t B, int64_t T, int H,
                                  float* out_f32, void* out_bf16, int out_cs, int out_dtype, int64_t out_third,
                                  float* save_gates, float* save_c, sos_stream_t stream) {
    if (!xproj || !whh_t || (!out_f32 && !out_bf16) || B < 1 || T < 1 || H < 4 || H > 256 || (H & 3) ||
        (out_bf16 && out_cs < 2 * H) || (out_dtype != SOS_DT_BF16 && out_dtype != SOS_DT_BF16X3) ||
        ((save_gates == nullptr) != (save_c == nullptr))) {
        sos_set_error("sos_lstm_bidir_fwd: bad args (B=%lld T=%lld H=%d)", (long long)B, (long long)T, H);
        return SOS_EINVAL;
    }
    dim3 grid((unsigned)((B + LSTM_NB - 1) / LSTM_NB), 2);
    const size_t lds = (size_t)(2 + 4 + 1) * H * LSTM_NB * sizeof(float);
    hipLaunchKernelGGL(lstm_kernel, grid, dim3(256), lds, (hipStream_t)stream, xproj, whh_t, (int)B, (int)T, H,
                       out_f32, (bf16_t*)out_bf16, out_cs, out_dtype == SOS_DT_BF16X3 ? 1 : 0,
                       (long long)out_third, save_gates, save_c);
    return sos_check_launch("sos_lstm_bidir_fwd");
}

// ------------------------------------------------------------------------ backward through time
// One workgroup per (direction, NB clips) walks the steps in reverse.  Per step:
//   A) per (hidden unit, clip): dh = dh_out + dh_rec, dc = dc_rec + dh*o*(1-tanh(c)^2); gate
//      pre-activation grads di,df,dg,do -> LDS [4H][NB] and global dgates (fp32 [B][T][2][4H]);
//      dc_rec = dc*f.
//   B) dh_rec[k] = sum_r W_hh[r][k] * dgate[r]: thread = (4 consecutive k, slice of rows); the
//      slices meet in LDS.  W_hh ([4H][H], the layout torch stores) streams from L2.
// dW_ih, dW_hh, the bias gradient and dx are GEMMs over dgates done by the wgrad / conv kernels.
__global__ __launch_bounds__(256) void lstm_bwd_kernel(const bf16_t* __restrict__ dh_out, int dh_cs, int dh_x3,
                                                       long long dh_third, const float* __restrict__ gates,
                                                       const float* __restrict__ csave, const float* __restrict__ whh,
                                                       int B, int T, int H, float* __restrict__ dgates) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int G = 4 * H;
    float* dgl = (float*)smem;                 // [4H][NB]
    float* dhr = dgl + G * LSTM_NB;            // [H][NB]  recurrent dh
    float* dcr = dhr + H * LSTM_NB;            // [H][NB]  recurrent dc
    float* part = dcr + H * LSTM_NB;           // [RS][H][NB] partial dh_rec
    const int tid = threadIdx.x;
    const int dir = blockIdx.y;
    const int b0 = blockIdx.x * LSTM_NB;
    const float* W = whh + (size_t)dir * G * H;   // [4H][H]
    const int KG = H >> 2;                         // groups of 4 consecutive k
    const int RS = 256 / KG;                       // row slices
    const int rows_per = (G + RS - 1) / RS;
    for (int idx = tid; idx < 2 * H * LSTM_NB; idx += 256) dhr[idx] = 0.f;   // dhr and dcr are adjacent
    __syncthreads();
    for (int step = 0; step < T; ++step) {
        const int t = dir == 0 ? T - 1 - step : step;     // reverse of the forward order
        const int tprev = dir == 0 ? t - 1 : t + 1;       // time index the forward pass came from
        for (int idx = tid; idx < H * LSTM_NB; idx += 256) {
            const int j = idx / LSTM_NB, n = idx - j * LSTM_NB;
            const int b = b0 + n;
            float di = 0.f, df = 0.f, dg = 0.f, dov = 0.f, dcn = 0.f;
            if (b < B) {
                const size_t row = (size_t)b * T + t;
                const bf16_t* dp = dh_out + row * dh_cs + dir * H + j;
                float dh = bf2f(dp[0]);
                if (dh_x3) dh += bf2f(dp[2 * dh_third]);
                dh += dhr[idx];
                const float* gp = gates + (row * 2 + dir) * G;
                const float ig = gp[j], fg = gp[H + j], gt = gp[2 * H + j], og = gp[3 * H + j];
                const float c = csave[(row * 2 + dir) * H + j];
                const float cprev = (tprev >= 0 && tprev < T) ? csave[(((size_t)b * T + tprev) * 2 + dir) * H + j] : 0.f;
                const float tc = tanhf(c);
                const float dc = dcr[idx] + dh * og * (1.f - tc * tc);
                di = dc * gt * ig * (1.f - ig);
                df = dc * cprev * fg * (1.f - fg);
                dg = dc * ig * (1.f - gt * gt);
                dov = dh * tc * og * (1.f - og);
                dcn = dc * fg;
                float* dgp = dgates + (row * 2 + dir) * G;
                dgp[j] = di; dgp[H + j] = df; dgp[2 * H + j] = dg; dgp[3 * H + j] = dov;
            }
            dcr[idx] = dcn;
            dgl[(0 * H + j) * LSTM_NB + n] = di;
            dgl[(1 * H + j) * LSTM_NB + n] = df;
            dgl[(2 * H + j) * LSTM_NB + n] = dg;
            dgl[(3 * H + j) * LSTM_NB + n] = dov;
        }
        __syncthreads();
        {
            const int kg = tid % KG, rs = tid / KG;
            if (rs < RS) {
                float acc[4][LSTM_NB];
#pragma unroll
                for (int e = 0; e < 4; ++e)
#pragma unroll
                    for (int n = 0; n < LSTM_NB; ++n) acc[e][n] = 0.f;
                const int r0 = rs * rows_per, r1 = min(r0 + rows_per, G);
#pragma unroll 4
                for (int r = r0; r < r1; ++r) {
                    const float4 w = *(const float4*)(W + (size_t)r * H + kg * 4);
                    const float4 da = *(const float4*)(dgl + r * LSTM_NB);
                    const float4 db = *(const float4*)(dgl + r * LSTM_NB + 4);
                    const float dv[8] = {da.x, da.y, da.z, da.w, db.x, db.y, db.z, db.w};
#pragma unroll
                    for (int n = 0; n < LSTM_NB; ++n) {
                        acc[0][n] = fmaf(w.x, dv[n], acc[0][n]);
                        acc[1][n] = fmaf(w.y, dv[n], acc[1][n]);
                        acc[2][n] = fmaf(w.z, dv[n], acc[2][n]);
                        acc[3][n] = fmaf(w.w, dv[n], acc[3][n]);
                    }
                }
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float* pp = part + ((size_t)rs * H + kg * 4 + e) * LSTM_NB;
                    *(float4*)pp = make_float4(acc[e][0], acc[e][1], acc[e][2], acc[e][3]);
                    *(float4*)(pp + 4) = make_float4(acc[e][4], acc[e][5], acc[e][6], acc[e][7]);
                }
            }
        }
        __syncthreads();
        for (int idx = tid; idx < H * LSTM_NB; idx += 256) {
            float s = 0.f;
            for (int rs = 0; rs < RS; ++rs) s += part[(size_t)rs * H * LSTM_NB + idx];
            dhr[idx] = s;
        }
        __syncthreads();
    }
}

extern "C" int sos_lstm_bidir_bwd(const void* dh_out, int dh_cs, int dh_dtype, int64_t dh_third, const float* gates,
                                  const float* csave, const float* whh, int64_t B, int64_t T, int H, float* dgates,
                                  sos_stream_t stream) {
    if (!dh_out || !gates || !csave || !whh || !dgates || B < 1 || T < 1 || H < 4 || H > 256 || (H & 3) || dh_cs < 2 * H ||
        (dh_dtype != SOS_DT_BF16 && dh_dtype != SOS_DT_BF16X3)) {
        sos_set_error("sos_lstm_bidir_bwd: bad args");
        return SOS_EINVAL;
    }
    const int KG = H >> 2, RS = 256 / KG;
    const size_t lds = (size_t)(4 + 1 + 1 + RS) * H * LSTM_NB * sizeof(float);
    if (lds > 160 * 1024) { sos_set_error("sos_lstm_bidir_bwd: LDS"); return SOS_ENOSPC; }
    static bool attr = false;
    if (!attr) { (void)hipFuncSetAttribute((const void*)lstm_bwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); attr = true; }
    dim3 grid((unsigned)((B + LSTM_NB - 1) / LSTM_NB), 2);
    hipLaunchKernelGGL(lstm_bwd_kernel, grid, dim3(256), lds, (hipStream_t)stream, (const bf16_t*)dh_out, dh_cs,
                       dh_dtype == SOS_DT_BF16X3 ? 1 : 0, (long long)dh_third, gates, csave, whh, (int)B, (int)T, H, dgates);
    return sos_check_launch("sos_lstm_bidir_bwd");
}
