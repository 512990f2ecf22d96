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

// Compute units of the current device (MI355X: 256; fewer in a partitioned mode), queried once per device (api.hip).
// Sizes the persistent grids, the de-phasing condition and the wave-quantisation model of the convolution launchers.
int y3_device_cus();

// conv_c32.hip: the 3x3 32 -> 64 kernel y3_conv2d_fused dispatches to by shape (same packed weights, same results)
bool y3_conv_c32_ok(const y3_conv_desc *d, long out_bytes, long res_bytes);
int y3_conv_c32_launch(const y3_conv_desc *d, const void *in, const void *w, const float *bias, const void *residual,
                       void *out, const void *zeros, hipStream_t stream);
bool y3_conv_s2c64_ok(const y3_conv_desc *d, long out_bytes);
int y3_conv_s2c64_launch(const y3_conv_desc *d, const void *in, const void *w, const float *bias, void *out,
                         const void *zeros, hipStream_t stream);
bool y3_conv_1x1w_ok(const y3_conv_desc *d, long out_bytes);
int y3_conv_1x1w_launch(const y3_conv_desc *d, const void *in, const void *w, const float *bias, void *out,
                        const void *zeros, hipStream_t stream);

// conv_resblock128.hip: the 128-channel residual block (y3_resblock_fused dispatches to it)
long y3_resblock128_tiles(int batch, int h, int w);
int y3_resblock128_launch(const y3_resblock_desc *d, const void *x, const void *w1, const float *b1, const void *w2,
                          const float *b2, void *out, const void *zeros, hipStream_t stream);
long y3_resblock64_tiles(int batch, int h, int w);
int y3_resblock64_launch(const y3_resblock_desc *d, const void *x, const void *w1, const float *b1, const void *w2,
                         const float *b2, void *out, const void *zeros, hipStream_t stream);

// ---- shared by the convolution kernels (conv_igemm.hip, conv_resblock.hip) ------------------------------------
struct FastDiv { unsigned magic, shift; };      // q = (umulhi(n, magic) + n) >> shift, exact for n < 2^31

__device__ __forceinline__ int fast_div(int n, FastDiv d) {
    return (int)((__umulhi((unsigned)n, d.magic) + (unsigned)n) >> d.shift);
}

static inline FastDiv make_fastdiv(int d) {
    unsigned shift = 0;
    while ((1ull << shift) < (unsigned long long)d) ++shift;
    FastDiv f;
    f.magic = (unsigned)((((1ull << 32) * ((1ull << shift) - (unsigned long long)d)) / (unsigned long long)d) + 1);
    f.shift = shift;
    return f;
}

__device__ __forceinline__ int xcd_remap(int orig, int nblk, bool reverse = false) {
    // Blocks are dealt round-robin over the 8 XCDs; give each XCD a contiguous
    // range of logical tiles so that the n-tiles of one m-tile (which share the
    // gathered input) and neighbouring m-tiles (which share halo rows) meet in one L2.
    // reverse: the XCD walks its range from the end (Y3_CONV_REVERSE: consecutive layers alternate, so a layer
    // starts on what the previous one wrote last -- still in L2 / Infinity Cache).
    const int q = nblk >> 3, r = nblk & 7, x = orig & 7;
    const int start = x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q;
    const int len = q + (x < r ? 1 : 0), i = orig >> 3;
    return start + (reverse ? len - 1 - i : i);
}

// An 8-byte LDS store the compiler does not see as one.  hipcc orders every ds_write it emits after ALL outstanding
// LDS-DMA (s_waitcnt vmcnt(0): the DMA writes LDS and may alias), which would make a kernel wait for its prefetch of
// the NEXT tile before writing an intermediate image of this one.  The caller guarantees the regions are disjoint and
// publishes the store itself (s_waitcnt lgkmcnt(0) + barrier).
__device__ __forceinline__ void lds_store8_unordered(char *dst, unsigned lo, unsigned hi) {
    typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));
    const u32x2_t v = {lo, hi};
    const unsigned a = (unsigned)(size_t)(__attribute__((address_space(3))) char *)dst;
    asm volatile("ds_write_b64 %0, %1" ::"v"(a), "v"(v) : "memory");
}

// One LDS-DMA instruction: 16 bytes per lane from a per-lane global address to (wave-uniform LDS base + lane * 16).
__device__ __forceinline__ void glds16(const char *gsrc, char *lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)gsrc,
                                     (__attribute__((address_space(3))) void *)lds_wave_base, 16, 0, 0);
}
