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
                                                   long long third) {
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
                cv[e] = sigmoidf_(fv[e]) * cv[e] + sigmoidf_(iv[e]) * tanhf(gv[e]);
                hv[e] = sigmoidf_(ov[e]) * tanhf(cv[e]);
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
                                  sos_stream_t stream) {
    if (!xproj || !whh_t || (!out_f32 && !out_bf16) || B < 1 || T < 1 || H < 4 || H > 256 || (H & 3) ||
        (out_bf16 && out_cs < 2 * H) || (out_dtype != SOS_DT_BF16 && out_dtype != SOS_DT_BF16X3)) {
        sos_set_error("sos_lstm_bidir_fwd: bad args (B=%lld T=%lld H=%d)", (long long)B, (long long)T, H);
        return SOS_EINVAL;
    }
    dim3 grid((unsigned)((B + LSTM_NB - 1) / LSTM_NB), 2);
    const size_t lds = (size_t)(2 + 4 + 1) * H * LSTM_NB * sizeof(float);
    hipLaunchKernelGGL(lstm_kernel, grid, dim3(256), lds, (hipStream_t)stream, xproj, whh_t, (int)B, (int)T, H,
                       out_f32, (bf16_t*)out_bf16, out_cs, out_dtype == SOS_DT_BF16X3 ? 1 : 0,
                       (long long)out_third);
    return sos_check_launch("sos_lstm_bidir_fwd");
}
