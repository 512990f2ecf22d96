// Shared helpers for the gfx950 kernels of libyolo3hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>
#include "../../include/yolo3hip.h"

typedef __attribute__((ext_vector_type(8))) short bf16x8;      // one 16-byte chunk of bf16
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;

void y3_set_error(const char *fmt, ...);
int y3_check_launch(const char *what);

#define Y3_REQUIRE(cond, code, ...)            \
    do {                                       \
        if (!(cond)) {                         \
            y3_set_error(__VA_ARGS__);         \
            return (code);                     \
        }                                      \
    } while (0)

// bf16 <-> f32, round-to-nearest-even; NaN stays NaN via the compiler's cvt (plain cast).
__device__ __forceinline__ float bf16_to_f32(unsigned short h) {
    return __uint_as_float(((unsigned int)h) << 16);
}
__device__ __forceinline__ unsigned short f32_to_bf16(float f) {
    __bf16 b = (__bf16)f;                       // hipcc -O3: v_cvt_pk_bf16_f32 (RNE, NaN preserved)
    return __builtin_bit_cast(unsigned short, b);
}
typedef __attribute__((ext_vector_type(2))) float f32x2_t;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
__device__ __forceinline__ unsigned int pack_bf16x2(float lo, float hi) {
    // one v_cvt_pk_bf16_f32 for the pair (RNE, NaN preserved), low half = lo
    const bf16x2_t b = __builtin_convertvector(f32x2_t{lo, hi}, bf16x2_t);
    return __builtin_bit_cast(unsigned int, b);
}

__device__ __forceinline__ float leaky01(float x) { return x > 0.0f ? x : 0.1f * x; }
// Same value bit for bit for every non-NaN x (signed zeros included): max(x, 0.1x) as a median with +inf,
// two instructions instead of three and no canonicalisation.
// max(x, slope * x) = leaky-ReLU (slope 0.1) or identity (slope 1), written as the machine instruction: through
// fmed3 / fmaxf the compiler first canonicalises both inputs (v_max_f32 x, x, x -- the signalling-NaN quieting
// that v_max_f32 in IEEE mode does by itself), three VALU operations per element instead of two.  Same value
// as x > 0 ? x : 0.1f * x for every non-NaN x.
__device__ __forceinline__ float leaky_max(float x, float slope) {
    const float t = slope * x;
    float r;
    asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(x), "v"(t));
    return r;
}

// fmaxf / fminf without the compiler's input canonicalisation (see leaky_max); identical for non-NaN inputs.
__device__ __forceinline__ float vmax_raw(float a, float b) {
    float r;
    asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ float vmin_raw(float a, float b) {
    float r;
    asm("v_min_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

__device__ __forceinline__ float leaky01_fast(float x) { return __builtin_amdgcn_fmed3f(x, 0.1f * x, __builtin_inff()); }

// DPP moves within a row of 16 lanes: CTRL 0x120 + n = row_ror:n (rotate), 0x150 + n = row_newbcast:n
// (lane n of the row to all its lanes).  No LDS traffic, one VALU instruction each.
template <int CTRL>
__device__ __forceinline__ unsigned dpp_u(unsigned v) {
    return (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, 0xf, 0xf, false);
}

static inline bool aligned16(const void *p) { return (((uintptr_t)p) & 15) == 0; }
