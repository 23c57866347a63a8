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

__global__ __launch_bounds__(256) void lstm_kernel(const float* __restrict__ xproj, const float* __restrict__ whh_t,
                                                   int B, int T, int H, float* __restrict__ out_f32,
                                                   bf16_t* __restrict__ out_bf16, int out_cs, int x3,
                                                   long long third) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* hbuf = (float*)smem;                      // [2][H][LSTM_NB]
    const int j = threadIdx.x;
    const int dir = blockIdx.y;
    const int b0 = blockIdx.x * LSTM_NB;
    const int G = 4 * H;
    const float* W = whh_t + (size_t)dir * H * G;    // [k][4H]
    float c[LSTM_NB], hreg[LSTM_NB];
#pragma unroll
    for (int n = 0; n < LSTM_NB; ++n) { c[n] = 0.f; hreg[n] = 0.f; }
    for (int idx = threadIdx.x; idx < 2 * H * LSTM_NB; idx += blockDim.x) hbuf[idx] = 0.f;
    __syncthreads();

    for (int step = 0; step < T; ++step) {
        const int t = dir == 0 ? step : T - 1 - step;
        const float* hcur = hbuf + (size_t)(step & 1) * H * LSTM_NB;
        float* hnext = hbuf + (size_t)((step + 1) & 1) * H * LSTM_NB;
        if (j < H) {
            float g[4][LSTM_NB];
#pragma unroll
            for (int n = 0; n < LSTM_NB; ++n) {
                const int b = b0 + n;
                if (b < B) {
                    const float* xp = xproj + (((size_t)b * T + t) * 2 + dir) * G;
#pragma unroll
                    for (int q = 0; q < 4; ++q) g[q][n] = xp[q * H + j];
                } else {
#pragma unroll
                    for (int q = 0; q < 4; ++q) g[q][n] = 0.f;
                }
            }
#pragma unroll 4
            for (int k = 0; k < H; ++k) {
                const float w0 = W[(size_t)k * G + j];
                const float w1 = W[(size_t)k * G + H + j];
                const float w2 = W[(size_t)k * G + 2 * H + j];
                const float w3 = W[(size_t)k * G + 3 * H + j];
                const float4 ha = *(const float4*)(hcur + k * LSTM_NB);
                const float4 hb = *(const float4*)(hcur + k * LSTM_NB + 4);
                const float hv[8] = {ha.x, ha.y, ha.z, ha.w, hb.x, hb.y, hb.z, hb.w};
#pragma unroll
                for (int n = 0; n < LSTM_NB; ++n) {
                    g[0][n] = fmaf(w0, hv[n], g[0][n]);
                    g[1][n] = fmaf(w1, hv[n], g[1][n]);
                    g[2][n] = fmaf(w2, hv[n], g[2][n]);
                    g[3][n] = fmaf(w3, hv[n], g[3][n]);
                }
            }
#pragma unroll
            for (int n = 0; n < LSTM_NB; ++n) {
                const float ig = sigmoidf_(g[0][n]), fg = sigmoidf_(g[1][n]);
                const float gg = tanhf(g[2][n]), og = sigmoidf_(g[3][n]);
                c[n] = fg * c[n] + ig * gg;
                hreg[n] = og * tanhf(c[n]);
            }
            float4 o0 = make_float4(hreg[0], hreg[1], hreg[2], hreg[3]);
            float4 o1 = make_float4(hreg[4], hreg[5], hreg[6], hreg[7]);
            *(float4*)(hnext + j * LSTM_NB) = o0;
            *(float4*)(hnext + j * LSTM_NB + 4) = o1;
#pragma unroll
            for (int n = 0; n < LSTM_NB; ++n) {
                const int b = b0 + n;
                if (b >= B) continue;
                const size_t row = (size_t)b * T + t;
                if (out_f32) out_f32[row * (2 * H) + dir * H + j] = hreg[n];
                if (out_bf16) {
                    bf16_t* o = out_bf16 + row * out_cs + dir * H + j;
                    const bf16_t hi = f2bf(hreg[n]);
                    o[0] = hi;
                    if (x3) {
                        o[third] = hi;
                        o[2 * third] = f2bf(hreg[n] - bf2f(hi));
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
    if (!xproj || !whh_t || (!out_f32 && !out_bf16) || B < 1 || T < 1 || H < 1 || H > 256 || (H & 3) ||
        (out_bf16 && out_cs < 2 * H) || (out_dtype != SOS_DT_BF16 && out_dtype != SOS_DT_BF16X3)) {
        sos_set_error("sos_lstm_bidir_fwd: bad args (B=%lld T=%lld H=%d)", (long long)B, (long long)T, H);
        return SOS_EINVAL;
    }
    dim3 grid((unsigned)((B + LSTM_NB - 1) / LSTM_NB), 2);
    const size_t lds = (size_t)2 * H * LSTM_NB * sizeof(float);
    hipLaunchKernelGGL(lstm_kernel, grid, dim3(256), lds, (hipStream_t)stream, xproj, whh_t, (int)B, (int)T, H,
                       out_f32, (bf16_t*)out_bf16, out_cs, out_dtype == SOS_DT_BF16X3 ? 1 : 0,
                       (long long)out_third);
    return sos_check_launch("sos_lstm_bidir_fwd");
}
