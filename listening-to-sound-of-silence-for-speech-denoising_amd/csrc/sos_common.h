// sos_common.h -- shared device/host helpers for the gfx950 kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>

#include "../../include/sos_hip.h"

typedef unsigned short bf16_t;   // raw bfloat16 bits

void sos_set_error(const char* fmt, ...);
int sos_check_launch(const char* what);

// round-to-nearest-even float -> bf16 bits (inputs are finite activations/weights)
__device__ __forceinline__ bf16_t f2bf(float f) {
    unsigned u = __float_as_uint(f);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (bf16_t)(u >> 16);
}
// two floats -> packed bf16 pair (round to nearest even) with gfx950's v_cvt_pk_bf16_f32
typedef __bf16 sos_bf16x2 __attribute__((ext_vector_type(2)));
typedef float sos_f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned pack2bf(float a, float b) {
    const sos_f32x2 v = {a, b};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, sos_bf16x2));
}
__device__ __forceinline__ float bf2f(bf16_t h) { return __uint_as_float(((unsigned)h) << 16); }

__device__ __forceinline__ int reflect_index(int i, int n) {
    // ReflectionPad2d / numpy 'reflect': -1 -> 1, n -> n-2; clamped so that coordinates that are
    // only needed by discarded (out-of-range) outputs stay inside the buffer.
    if (i < 0) i = -i;
    if (i >= n) i = 2 * (n - 1) - i;
    if (i < 0) i = 0;
    if (i >= n) i = n - 1;
    return i;
}
