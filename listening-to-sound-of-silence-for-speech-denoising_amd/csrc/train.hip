// train.hip -- fused losses and optimizer step (gfx950, HBM-bound elementwise kernels).
//
// Reference: M2/agent.py:174,188-189 (nn.MSELoss on (n_pred, full_noise) and (rec, clean)),
// M1/agent.py:187,202 (nn.BCEWithLogitsLoss), M1/agent.py:177 / M2/agent.py:167 (optim.Adam,
// lr 1e-3, betas (0.9, 0.999), eps 1e-8).  Each loss kernel produces the scalar loss (two-stage,
// deterministic) and the gradient w.r.t. its first argument in one pass.
#include "sos_common.h"

#define LOSS_BLOCKS 1024

__global__ __launch_bounds__(256) void mse_kernel(const float* __restrict__ a, const float* __restrict__ b, long long n,
                                                  float gscale, float* __restrict__ grad, float* __restrict__ partial) {
    __shared__ float red[256];
    float acc = 0.f;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        const float d = a[i] - b[i];
        acc = fmaf(d, d, acc);
        if (grad) grad[i] = gscale * d;
    }
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) partial[blockIdx.x] = red[0];
}

__global__ __launch_bounds__(256) void bce_kernel(const float* __restrict__ x, const float* __restrict__ y, long long n,
                                                  float gscale, float* __restrict__ grad, float* __restrict__ partial) {
    __shared__ float red[256];
    float acc = 0.f;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        const float xv = x[i], yv = y[i];
        acc += fmaxf(xv, 0.f) - xv * yv + log1pf(expf(-fabsf(xv)));
        if (grad) grad[i] = gscale * (1.0f / (1.0f + expf(-xv)) - yv);
    }
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) partial[blockIdx.x] = red[0];
}

__global__ void loss_finalize_kernel(const float* __restrict__ partial, int nblk, double inv_n, float* __restrict__ loss) {
    double s = 0.0;
    for (int i = 0; i < nblk; ++i) s += (double)partial[i];
    loss[0] = (float)(s * inv_n);
}

static int loss_blocks(long long n) {
    long long b = (n + 255) / 256;
    if (b > LOSS_BLOCKS) b = LOSS_BLOCKS;
    return (int)(b < 1 ? 1 : b);
}

/* mean((a-b)^2) -> loss[0]; grad (optional) = upstream * 2(a-b)/n.  partial: f32 [1024]. */
extern "C" int sos_mse_loss(const float* a, const float* b, int64_t n, float upstream, float* loss, float* grad,
                            float* partial, sos_stream_t stream) {
    if (!a || !b || !loss || !partial || n < 1) { sos_set_error("sos_mse_loss: bad args"); return SOS_EINVAL; }
    const int nb = loss_blocks(n);
    hipLaunchKernelGGL(mse_kernel, dim3(nb), dim3(256), 0, (hipStream_t)stream, a, b, (long long)n,
                       upstream * 2.0f / (float)n, grad, partial);
    hipLaunchKernelGGL(loss_finalize_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, partial, nb, 1.0 / (double)n, loss);
    return sos_check_launch("sos_mse_loss");
}

/* mean(max(x,0) - x*y + log1p(exp(-|x|))) -> loss[0]; grad (optional) = upstream*(sigmoid(x)-y)/n. */
extern "C" int sos_bce_logits_loss(const float* x, const float* y, int64_t n, float upstream, float* loss, float* grad,
                                   float* partial, sos_stream_t stream) {
    if (!x || !y || !loss || !partial || n < 1) { sos_set_error("sos_bce_logits_loss: bad args"); return SOS_EINVAL; }
    const int nb = loss_blocks(n);
    hipLaunchKernelGGL(bce_kernel, dim3(nb), dim3(256), 0, (hipStream_t)stream, x, y, (long long)n, upstream / (float)n,
                       grad, partial);
    hipLaunchKernelGGL(loss_finalize_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, partial, nb, 1.0 / (double)n, loss);
    return sos_check_launch("sos_bce_logits_loss");
}

// ---- loss scale of the fp16 storage mode (sos_hip.h): max|g| folded with an integer atomic max on the bit patterns of
// the non-negative floats (order independent -> deterministic), then S = 2^floor(log2(target/amax)).
__global__ __launch_bounds__(256) void amax_kernel(const float* __restrict__ x, long long n, float* __restrict__ amax) {
    __shared__ float red[256];
    float m = 0.f;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) m = fmaxf(m, fabsf(x[i]));
    red[threadIdx.x] = m;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) red[threadIdx.x] = fmaxf(red[threadIdx.x], red[threadIdx.x + s]);
        __syncthreads();
    }
    if (threadIdx.x == 0) atomicMax((unsigned*)amax, __float_as_uint(red[0]));
}
__global__ void loss_scale_kernel(const float* __restrict__ amax, float target, float* __restrict__ scale2,
                                  const float* __restrict__ guard) {
    const float a = amax[0];
    if (guard && guard[1] > 0.f) target *= guard[1];          // back-off after overflows (sos_grad_guard)
    float S = 1.f;
    if (a > 0.f && a < 3.0e38f) {
        int e = (int)floorf(log2f(target / a));
        e = e < -40 ? -40 : (e > 40 ? 40 : e);
        S = exp2f((float)e);
    }
    scale2[0] = S;
    scale2[1] = 1.f / S;
}
__global__ void scale_kernel(float* __restrict__ x, long long n, const float* __restrict__ s) {
    const float k = s[0];
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) x[i] *= k;
}

extern "C" int sos_amax_f32(const float* x, int64_t n, float* amax, sos_stream_t stream) {
    if (!x || !amax || n < 1) { sos_set_error("sos_amax_f32: bad args"); return SOS_EINVAL; }
    hipLaunchKernelGGL(amax_kernel, dim3(loss_blocks(n)), dim3(256), 0, (hipStream_t)stream, x, (long long)n, amax);
    return sos_check_launch("sos_amax_f32");
}
extern "C" int sos_loss_scale(const float* amax, float target, float* scale2, const float* guard, sos_stream_t stream) {
    if (!amax || !scale2 || !(target > 0.f)) { sos_set_error("sos_loss_scale: bad args"); return SOS_EINVAL; }
    hipLaunchKernelGGL(loss_scale_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, amax, target, scale2, guard);
    return sos_check_launch("sos_loss_scale");
}
extern "C" int sos_scale_f32(float* x, int64_t n, const float* s, sos_stream_t stream) {
    if (!x || !s || n < 1) { sos_set_error("sos_scale_f32: bad args"); return SOS_EINVAL; }
    long long gb = (n + 255) / 256;
    if (gb > 2048) gb = 2048;
    hipLaunchKernelGGL(scale_kernel, dim3((unsigned)gb), dim3(256), 0, (hipStream_t)stream, x, (long long)n, s);
    return sos_check_launch("sos_scale_f32");
}

// ---- overflow guard (sos_hip.h): any Inf / NaN among a model's parameter gradients -> the step is skipped on the device
__global__ __launch_bounds__(256) void grad_guard_kernel(const float* __restrict__ g, long long n, unsigned* __restrict__ raw) {
    int bad = 0;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256)
        bad |= (__float_as_uint(g[i]) & 0x7f800000u) == 0x7f800000u;
    if (__syncthreads_or(bad) && threadIdx.x == 0) atomicOr(raw, 1u);
}
__global__ void grad_guard_finalize_kernel(float* __restrict__ st) {
    unsigned* raw = (unsigned*)(st + 4);
    const unsigned bad = *raw;
    *raw = 0u;
    float b = st[1] > 0.f ? st[1] : 1.f;
    if (bad) {
        st[0] = 1.f;
        st[3] += 1.f;
        st[2] = 0.f;
        b = fmaxf(0.5f * b, 9.5367431640625e-07f);          // 2^-20
    } else {
        st[0] = 0.f;
        st[2] += 1.f;
        if (st[2] >= (float)SOS_GUARD_GROWTH && b < 1.f) { b = fminf(1.f, 2.f * b); st[2] = 0.f; }
    }
    st[1] = b;
}
extern "C" int sos_grad_guard(const float* g, int64_t n, float* guard, int finalize, sos_stream_t stream) {
    if (!g || !guard || n < 1) { sos_set_error("sos_grad_guard: bad args"); return SOS_EINVAL; }
    hipLaunchKernelGGL(grad_guard_kernel, dim3(loss_blocks(n)), dim3(256), 0, (hipStream_t)stream, g, (long long)n,
                       (unsigned*)(guard + 4));
    if (finalize) hipLaunchKernelGGL(grad_guard_finalize_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, guard);
    return sos_check_launch("sos_grad_guard");
}

// Bias corrections count APPLIED updates (torch.cuda.amp.GradScaler does not advance `step` on a skipped update): the host's
// step counter counts attempts; the guard's [3] holds the steps skipped so far ON THE DEVICE, so the corrections are
// re-derived here from step - skipped whenever a step has ever been skipped (otherwise the host's values stand, bit for bit).
__device__ __forceinline__ void adam_applied_steps(const float* __restrict__ skip, float step, float b1, float b2, float& bc1,
                                                   float& bc2_sqrt) {
    if (skip && skip[3] != 0.f) {
        const float applied = fmaxf(step - skip[3], 1.f);
        bc1 = 1.0f - powf(b1, applied);
        bc2_sqrt = sqrtf(1.0f - powf(b2, applied));
    }
}

__global__ void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                            float* __restrict__ v, long long n, float lr, float b1, float b2, float eps, float wd,
                            float bc1, float bc2_sqrt, float gscale, const float* __restrict__ skip, float step) {
    if (skip && skip[0] != 0.f) return;
    adam_applied_steps(skip, step, b1, b2, bc1, bc2_sqrt);
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        float gi = g[i] * gscale;
        const float pi = p[i];
        if (wd != 0.f) gi = fmaf(wd, pi, gi);
        const float mi = b1 * m[i] + (1.f - b1) * gi;
        const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
        m[i] = mi;
        v[i] = vi;
        const float denom = sqrtf(vi) / bc2_sqrt + eps;
        p[i] = pi - (lr / bc1) * (mi / denom);
    }
}

/* torch.optim.Adam (amsgrad=False) single-tensor update; step = 1-based step count after increment. */
extern "C" int sos_adam_step(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2,
                             float eps, float weight_decay, int64_t step, float grad_scale, const float* skip,
                             sos_stream_t stream) {
    if (!p || !g || !m || !v || n < 1 || step < 1) { sos_set_error("sos_adam_step: bad args"); return SOS_EINVAL; }
    const float bc1 = 1.0f - powf(beta1, (float)step);
    const float bc2 = 1.0f - powf(beta2, (float)step);
    long long gb = (n + 255) / 256;
    if (gb > 2048) gb = 2048;
    hipLaunchKernelGGL(adam_kernel, dim3((unsigned)gb), dim3(256), 0, (hipStream_t)stream, p, g, m, v, (long long)n, lr,
                       beta1, beta2, eps, weight_decay, bc1, sqrtf(bc2), grad_scale, skip, (float)step);
    return sos_check_launch("sos_adam_step");
}


// ---- multi-tensor Adam: ONE launch updates every parameter tensor of a model (the per-tensor form costs 143 / 83 launches
// per step for the denoiser / detector).  tab: device array of tensors; chunks: device array of (tensor, chunk) pairs, one
// workgroup each, SOS_ADAM_CHUNK elements per chunk; all tensors share the hyper-parameters and the step count.
struct AdamTensor { float* p; const float* g; float* m; float* v; long long n; };
__global__ __launch_bounds__(256) void adam_multi_kernel(const AdamTensor* __restrict__ tab, const int2* __restrict__ chunks,
                                                         float lr, float b1, float b2, float eps, float wd, float bc1,
                                                         float bc2_sqrt, float gscale, const float* __restrict__ skip, float step) {
    if (skip && skip[0] != 0.f) return;         // overflow guard: the whole step is dropped (uniform over the grid)
    adam_applied_steps(skip, step, b1, b2, bc1, bc2_sqrt);
    const int2 c = chunks[blockIdx.x];
    const AdamTensor t = tab[c.x];
    const long long lo = (long long)c.y * SOS_ADAM_CHUNK;
    const long long hi = lo + SOS_ADAM_CHUNK < t.n ? lo + SOS_ADAM_CHUNK : t.n;
    for (long long i = lo + threadIdx.x; i < hi; i += 256) {
        float gi = t.g[i] * gscale;
        const float pi = t.p[i];
        if (wd != 0.f) gi = fmaf(wd, pi, gi);
        const float mi = b1 * t.m[i] + (1.f - b1) * gi;
        const float vi = b2 * t.v[i] + (1.f - b2) * gi * gi;
        t.m[i] = mi;
        t.v[i] = vi;
        const float denom = sqrtf(vi) / bc2_sqrt + eps;
        t.p[i] = pi - (lr / bc1) * (mi / denom);
    }
}

extern "C" int sos_adam_multi_step(const sos_adam_tensor* tensors, int n_tensors, const int32_t* chunks, int64_t n_chunks,
                                   float lr, float beta1, float beta2, float eps, float weight_decay, int64_t step,
                                   float grad_scale, const float* skip, sos_stream_t stream) {
    static_assert(sizeof(sos_adam_tensor) == sizeof(AdamTensor), "sos_adam_tensor layout");
    if (!tensors || !chunks || n_tensors < 1 || n_chunks < 1 || n_chunks > 0x7fffffff || step < 1) {
        sos_set_error("sos_adam_multi_step: bad args");
        return SOS_EINVAL;
    }
    const float bc1 = 1.0f - powf(beta1, (float)step);
    const float bc2 = 1.0f - powf(beta2, (float)step);
    hipLaunchKernelGGL(adam_multi_kernel, dim3((unsigned)n_chunks), dim3(256), 0, (hipStream_t)stream,
                       (const AdamTensor*)tensors, (const int2*)chunks, lr, beta1, beta2, eps, weight_decay, bc1, sqrtf(bc2),
                       grad_scale, skip, (float)step);
    return sos_check_launch("sos_adam_multi_step");
}


// ---- packed 16-bit weights refreshed after an optimizer step: ONE launch per model instead of ~10 torch kernels per
// packed tensor.  Every packed element is a pure relocation of one fp32 parameter element (tap-major order, channel
// padding, flipped / transposed / phase-sliced data-gradient layouts, the hi|lo|hi thirds of the three-pass mode), so the
// Python side records the relocation ONCE as a table of absolute source addresses (engine.PackRecorder) and this kernel
// replays it: idx > 0: the value at that address; idx < 0: its low part (v - float(half(v))) ; idx == 0: zero padding.
struct PackEntry { const long long* idx; bf16_t* out; long long n; };
__global__ __launch_bounds__(256) void gather_pack_kernel(const PackEntry* __restrict__ tab, const int2* __restrict__ chunks) {
    const int2 c = chunks[blockIdx.x];
    const PackEntry e = tab[c.x];
    const long long lo = (long long)c.y * SOS_ADAM_CHUNK;
    const long long hi = lo + SOS_ADAM_CHUNK < e.n ? lo + SOS_ADAM_CHUNK : e.n;
    // four elements per thread and step: the four index loads, then the four dependent value loads, are in flight together (one
    // element at a time ran at 0.76 TB/s: two dependent memory latencies per 14 bytes)
    for (long long i0 = lo + threadIdx.x; i0 < hi; i0 += 4 * 256) {
        long long a[4];
        float v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) a[u] = i0 + u * 256 < hi ? e.idx[i0 + u * 256] : 0;
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = a[u] ? *(const float*)(a[u] > 0 ? a[u] : -a[u]) : 0.f;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (i0 + u * 256 >= hi) continue;
            bf16_t r = 0;
            if (a[u] > 0) r = f2bf(v[u]);
            else if (a[u] < 0) r = f2bf(v[u] - bf2f(f2bf(v[u])));
            e.out[i0 + u * 256] = r;
        }
    }
}
extern "C" int sos_gather_pack_multi(const sos_pack_entry* entries, int n_entries, const int32_t* chunks, int64_t n_chunks,
                                     sos_stream_t stream) {
    static_assert(sizeof(sos_pack_entry) == sizeof(PackEntry), "sos_pack_entry layout");
    if (!entries || !chunks || n_entries < 1 || n_chunks < 1 || n_chunks > 0x7fffffff) {
        sos_set_error("sos_gather_pack_multi: bad args");
        return SOS_EINVAL;
    }
    hipLaunchKernelGGL(gather_pack_kernel, dim3((unsigned)n_chunks), dim3(256), 0, (hipStream_t)stream,
                       (const PackEntry*)entries, (const int2*)chunks);
    return sos_check_launch("sos_gather_pack_multi");
}
