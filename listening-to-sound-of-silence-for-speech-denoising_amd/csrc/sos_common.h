// sos_common.h -- shared device/host helpers for the gfx950 kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>
#include <mutex>

#include "../../include/sos_hip.h"

// ---- the 16-bit storage type of activations / packed weights.
// The library is built twice from the same sources: libsos_hip.so stores bfloat16 (precision modes "bf16" and the
// three-pass "bf16x3"), libsos_hip_f16.so (-DSOS_F16) stores IEEE half ("fp16": 11 significand bits instead of 8 at the
// same MFMA rate; gradients are kept in range by the power-of-two loss scale the Python side threads through the
// backward pass).  Kernels only touch the type through the helpers below; `bf16_t` is "the raw 16 storage bits".
typedef unsigned short bf16_t;

void sos_set_error(const char* fmt, ...);
int sos_check_launch(const char* what);

// One-time host-side setup per DEVICE (hipFuncSetAttribute is per device; nn.DataParallel drives several devices from
// one thread each): `f` runs once for the calling thread's current device, under a mutex.  No other mutable globals
// exist in the library besides the tiling cache (conv.hip, mutex-guarded) and the thread-local error string.
struct sos_device_once {
    std::mutex mu;
    uint64_t done = 0;
};
template <class F>
static inline int sos_per_device_once(sos_device_once& o, F&& f) {
    int dev = 0;
    (void)hipGetDevice(&dev);
    std::lock_guard<std::mutex> g(o.mu);
    if ((o.done >> (dev & 63)) & 1ull) return SOS_OK;
    const int rc = f();
    if (rc == SOS_OK) o.done |= 1ull << (dev & 63);
    return rc;
}

typedef float sos_f32x2 __attribute__((ext_vector_type(2)));
#ifdef SOS_F16
#define SOS_STORAGE_NAME "fp16"
typedef _Float16 sos_half_t;
typedef _Float16 sos_h16x2 __attribute__((ext_vector_type(2)));
#define SOS_MFMA_32x32x16 __builtin_amdgcn_mfma_f32_32x32x16_f16
#define SOS_MFMA_16x16x32 __builtin_amdgcn_mfma_f32_16x16x32_f16
// round-to-nearest-even float -> storage bits
__device__ __forceinline__ bf16_t f2bf(float f) { return __builtin_bit_cast(bf16_t, (_Float16)f); }
__device__ __forceinline__ float bf2f(bf16_t h) { return (float)__builtin_bit_cast(_Float16, h); }
// the two halves of a dword -> f32
__device__ __forceinline__ float sos_lo2f(unsigned u) { return (float)__builtin_bit_cast(sos_h16x2, u)[0]; }
__device__ __forceinline__ float sos_hi2f(unsigned u) { return (float)__builtin_bit_cast(sos_h16x2, u)[1]; }
#else
#define SOS_STORAGE_NAME "bf16"
typedef __bf16 sos_half_t;
typedef __bf16 sos_h16x2 __attribute__((ext_vector_type(2)));
#define SOS_MFMA_32x32x16 __builtin_amdgcn_mfma_f32_32x32x16_bf16
#define SOS_MFMA_16x16x32 __builtin_amdgcn_mfma_f32_16x16x32_bf16
// round-to-nearest-even float -> bf16 bits (inputs are finite activations/weights)
__device__ __forceinline__ bf16_t f2bf(float f) {
    unsigned u = __float_as_uint(f);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (bf16_t)(u >> 16);
}
__device__ __forceinline__ float bf2f(bf16_t h) { return __uint_as_float(((unsigned)h) << 16); }
__device__ __forceinline__ float sos_lo2f(unsigned u) { return __uint_as_float(u << 16); }
__device__ __forceinline__ float sos_hi2f(unsigned u) { return __uint_as_float(u & 0xffff0000u); }
#endif
// two floats -> packed storage pair, round to nearest even (gfx950: v_cvt_pk_bf16_f32 / v_cvt_pk_f16_f32)
__device__ __forceinline__ unsigned pack2bf(float a, float b) {
    const sos_f32x2 v = {a, b};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, sos_h16x2));
}

__device__ __forceinline__ int reflect_index(int i, int n) {
    // ReflectionPad2d / numpy 'reflect': -1 -> 1, n -> n-2; clamped so that coordinates that are
    // only needed by discarded (out-of-range) outputs stay inside the buffer.
    if (i < 0) i = -i;
    if (i >= n) i = 2 * (n - 1) - i;
    if (i < 0) i = 0;
    if (i >= n) i = n - 1;
    return i;
}
