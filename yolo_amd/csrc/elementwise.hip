// Bandwidth-bound NHWC kernels for gfx950: max-pool, and the un-fused fallbacks of
// upsample / route-concat / shortcut.  Every lane moves 16-byte channel chunks, so a
// wave touches whole 128-byte lines whenever c*elsize >= 128.
//   maxpool   : tf.nn.max_pool(padding='SAME')              reference build_model.py:207-210
//   upsample  : tf.image.resize_nearest_neighbor(2H, 2H)    reference build_model.py:236
//   copy      : tf.concat(axis=-1) slice                    reference build_model.py:191
//   add       : tf.math.add (shortcut)                      reference build_model.py:224
#include "common.h"

namespace {

constexpr int kThreads = 256;
constexpr int kMaxBlocks = 256 * 8;   // ~8 blocks per CU, grid-stride the rest

inline unsigned grid_for(long work) {
    long b = (work + kThreads - 1) / kThreads;
    if (b < 1) b = 1;
    return (unsigned)(b > kMaxBlocks ? kMaxBlocks : b);
}

template <bool BF16>
__device__ __forceinline__ u32x4 vmax(u32x4 a, u32x4 b) {
    u32x4 r;
    if constexpr (BF16) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float a0 = bf16_to_f32(a[j] & 0xffff), a1 = bf16_to_f32(a[j] >> 16);
            const float b0 = bf16_to_f32(b[j] & 0xffff), b1 = bf16_to_f32(b[j] >> 16);
            const unsigned lo = (a0 >= b0 ? a[j] : b[j]) & 0xffffu;
            const unsigned hi = (a1 >= b1 ? a[j] : b[j]) & 0xffff0000u;
            r[j] = lo | hi;
        }
    } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float x = __uint_as_float(a[j]), y = __uint_as_float(b[j]);
            r[j] = x >= y ? a[j] : b[j];
        }
    }
    return r;
}

template <bool BF16>
__device__ __forceinline__ u32x4 vadd(u32x4 a, u32x4 b) {
    u32x4 r;
    if constexpr (BF16) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
            r[j] = pack_bf16x2(bf16_to_f32(a[j] & 0xffff) + bf16_to_f32(b[j] & 0xffff),
                               bf16_to_f32(a[j] >> 16) + bf16_to_f32(b[j] >> 16));
    } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) r[j] = __float_as_uint(__uint_as_float(a[j]) + __uint_as_float(b[j]));
    }
    return r;
}

// in/out leading dimensions are in 16-byte chunks here.
template <bool BF16>
__global__ __launch_bounds__(kThreads) void maxpool_kernel(const u32x4 *in, u32x4 *out, int B, int H, int W,
                                                           int Ho, int Wo, int cch, int in_ldc, int out_ldc,
                                                           int size, int stride, int pad_t, int pad_l) {
    const long total = (long)B * Ho * Wo * cch;
    for (long i = (long)blockIdx.x * kThreads + threadIdx.x; i < total; i += (long)gridDim.x * kThreads) {
        const int ch = (int)(i % cch);
        long pix = i / cch;
        const int wo = (int)(pix % Wo);
        pix /= Wo;
        const int ho = (int)(pix % Ho);
        const int b = (int)(pix / Ho);
        const int h0 = ho * stride - pad_t, w0 = wo * stride - pad_l;
        bool have = false;
        u32x4 m = u32x4{0, 0, 0, 0};
        for (int dy = 0; dy < size; ++dy) {
            const int h = h0 + dy;
            if ((unsigned)h >= (unsigned)H) continue;        // padded cells never win
            for (int dx = 0; dx < size; ++dx) {
                const int w = w0 + dx;
                if ((unsigned)w >= (unsigned)W) continue;
                const u32x4 v = in[(((long)b * H + h) * W + w) * in_ldc + ch];
                m = have ? vmax<BF16>(m, v) : v;
                have = true;
            }
        }
        out[(((long)b * Ho + ho) * Wo + wo) * out_ldc + ch] = m;
    }
}

__global__ __launch_bounds__(kThreads) void upsample2x_kernel(const u32x4 *in, u32x4 *out, int B, int H, int W,
                                                              int cch, int in_ldc, int out_ldc) {
    const int Ho = 2 * H, Wo = 2 * W;
    const long total = (long)B * Ho * Wo * cch;
    for (long i = (long)blockIdx.x * kThreads + threadIdx.x; i < total; i += (long)gridDim.x * kThreads) {
        const int ch = (int)(i % cch);
        long pix = i / cch;
        const int wo = (int)(pix % Wo);
        pix /= Wo;
        const int ho = (int)(pix % Ho);
        const int b = (int)(pix / Ho);
        out[(((long)b * Ho + ho) * Wo + wo) * out_ldc + ch] =
            in[(((long)b * H + (ho >> 1)) * W + (wo >> 1)) * in_ldc + ch];
    }
}

__global__ __launch_bounds__(kThreads) void copy_channels_kernel(const u32x4 *in, u32x4 *out, long pixels,
                                                                 int cch, int in_ldc, int out_ldc) {
    const long total = pixels * cch;
    for (long i = (long)blockIdx.x * kThreads + threadIdx.x; i < total; i += (long)gridDim.x * kThreads) {
        const int ch = (int)(i % cch);
        const long pix = i / cch;
        out[pix * out_ldc + ch] = in[pix * in_ldc + ch];
    }
}

template <bool BF16>
__global__ __launch_bounds__(kThreads) void add_kernel(const u32x4 *a, const u32x4 *b, u32x4 *out, long pixels,
                                                       int cch, int a_ldc, int b_ldc, int out_ldc) {
    const long total = pixels * cch;
    for (long i = (long)blockIdx.x * kThreads + threadIdx.x; i < total; i += (long)gridDim.x * kThreads) {
        const int ch = (int)(i % cch);
        const long pix = i / cch;
        out[pix * out_ldc + ch] = vadd<BF16>(a[pix * a_ldc + ch], b[pix * b_ldc + ch]);
    }
}

// fp32 [pixels][cin] -> T [pixels][16/sizeof(T)] zero-padded: lets the first layer (cin = 3) run on the
// chunked implicit-GEMM loader.  One 16-byte store per pixel.
template <bool BF16>
__global__ __launch_bounds__(kThreads) void pad_input_kernel(const float *in, u32x4 *out, long pixels, int cin) {
    for (long i = (long)blockIdx.x * kThreads + threadIdx.x; i < pixels; i += (long)gridDim.x * kThreads) {
        float v[8];
#pragma unroll
        for (int c = 0; c < 8; ++c) v[c] = c < cin ? in[i * cin + c] : 0.0f;
        u32x4 o;
        if constexpr (BF16) {
            o[0] = pack_bf16x2(v[0], v[1]); o[1] = pack_bf16x2(v[2], v[3]);
            o[2] = pack_bf16x2(v[4], v[5]); o[3] = pack_bf16x2(v[6], v[7]);
        } else {
            o[0] = __float_as_uint(v[0]); o[1] = __float_as_uint(v[1]);
            o[2] = __float_as_uint(v[2]); o[3] = __float_as_uint(v[3]);
        }
        out[i] = o;
    }
}

__global__ __launch_bounds__(kThreads) void cast_f32_bf16_kernel(const float *in, unsigned short *out, long n) {
    for (long i = (long)blockIdx.x * kThreads + threadIdx.x; i < n; i += (long)gridDim.x * kThreads)
        out[i] = f32_to_bf16(in[i]);
}
__global__ __launch_bounds__(kThreads) void cast_bf16_f32_kernel(const unsigned short *in, float *out, long n) {
    for (long i = (long)blockIdx.x * kThreads + threadIdx.x; i < n; i += (long)gridDim.x * kThreads)
        out[i] = bf16_to_f32(in[i]);
}

int check_chunks(const char *what, int dtype, int c, int ld0, int ld1, const void *p0, const void *p1) {
    Y3_REQUIRE(dtype == Y3_F32 || dtype == Y3_BF16, Y3_E_INVALID, "%s: dtype %d", what, dtype);
    const int epc = dtype == Y3_BF16 ? 8 : 4;
    Y3_REQUIRE(c > 0 && c % epc == 0, Y3_E_INVALID, "%s: channels %d not a multiple of %d", what, c, epc);
    Y3_REQUIRE(ld0 % epc == 0 && ld1 % epc == 0 && ld0 >= c && ld1 >= c, Y3_E_ALIGN, "%s: leading dims %d/%d", what, ld0, ld1);
    Y3_REQUIRE(p0 && p1 && aligned16(p0) && aligned16(p1), Y3_E_ALIGN, "%s: pointers must be non-null and 16-byte aligned", what);
    return Y3_OK;
}

}  // namespace

extern "C" int y3_maxpool2d(int dtype, const void *in, void *out, int batch, int in_h, int in_w, int c,
                            int in_ld, int out_ld, int size, int stride, void *stream) {
    int rc = check_chunks("maxpool", dtype, c, in_ld, out_ld, in, out);
    if (rc) return rc;
    Y3_REQUIRE(batch > 0 && in_h > 0 && in_w > 0 && size > 0 && stride > 0, Y3_E_INVALID, "maxpool: bad shape");
    const int epc = dtype == Y3_BF16 ? 8 : 4;
    // TF SAME: out = ceil(in/stride); total pad = max((out-1)*stride + size - in, 0); before = total/2
    const int Ho = (in_h + stride - 1) / stride, Wo = (in_w + stride - 1) / stride;
    int ph = (Ho - 1) * stride + size - in_h; if (ph < 0) ph = 0;
    int pw = (Wo - 1) * stride + size - in_w; if (pw < 0) pw = 0;
    const long total = (long)batch * Ho * Wo * (c / epc);
    if (dtype == Y3_BF16)
        hipLaunchKernelGGL(maxpool_kernel<true>, dim3(grid_for(total)), dim3(kThreads), 0, (hipStream_t)stream,
                           (const u32x4 *)in, (u32x4 *)out, batch, in_h, in_w, Ho, Wo, c / epc, in_ld / epc, out_ld / epc,
                           size, stride, ph / 2, pw / 2);
    else
        hipLaunchKernelGGL(maxpool_kernel<false>, dim3(grid_for(total)), dim3(kThreads), 0, (hipStream_t)stream,
                           (const u32x4 *)in, (u32x4 *)out, batch, in_h, in_w, Ho, Wo, c / epc, in_ld / epc, out_ld / epc,
                           size, stride, ph / 2, pw / 2);
    return y3_check_launch("maxpool");
}

extern "C" int y3_upsample2x(int dtype, const void *in, void *out, int batch, int in_h, int in_w, int c,
                             int in_ld, int out_ld, void *stream) {
    int rc = check_chunks("upsample", dtype, c, in_ld, out_ld, in, out);
    if (rc) return rc;
    Y3_REQUIRE(batch > 0 && in_h > 0 && in_w > 0, Y3_E_INVALID, "upsample: bad shape");
    const int epc = dtype == Y3_BF16 ? 8 : 4;
    const long total = (long)batch * in_h * 2 * in_w * 2 * (c / epc);
    hipLaunchKernelGGL(upsample2x_kernel, dim3(grid_for(total)), dim3(kThreads), 0, (hipStream_t)stream,
                       (const u32x4 *)in, (u32x4 *)out, batch, in_h, in_w, c / epc, in_ld / epc, out_ld / epc);
    return y3_check_launch("upsample");
}

extern "C" int y3_copy_channels(int dtype, const void *in, void *out, int64_t pixels, int c, int in_ld,
                                int out_ld, void *stream) {
    int rc = check_chunks("copy_channels", dtype, c, in_ld, out_ld, in, out);
    if (rc) return rc;
    Y3_REQUIRE(pixels > 0, Y3_E_INVALID, "copy_channels: %ld pixels", (long)pixels);
    const int epc = dtype == Y3_BF16 ? 8 : 4;
    hipLaunchKernelGGL(copy_channels_kernel, dim3(grid_for(pixels * (c / epc))), dim3(kThreads), 0, (hipStream_t)stream,
                       (const u32x4 *)in, (u32x4 *)out, (long)pixels, c / epc, in_ld / epc, out_ld / epc);
    return y3_check_launch("copy_channels");
}

extern "C" int y3_shortcut_add(int dtype, const void *a, const void *b, void *out, int64_t pixels, int c,
                               int a_ld, int b_ld, int out_ld, void *stream) {
    int rc = check_chunks("shortcut_add", dtype, c, a_ld, b_ld, a, b);
    if (rc) return rc;
    rc = check_chunks("shortcut_add", dtype, c, out_ld, out_ld, out, out);
    if (rc) return rc;
    Y3_REQUIRE(pixels > 0, Y3_E_INVALID, "shortcut_add: %ld pixels", (long)pixels);
    const int epc = dtype == Y3_BF16 ? 8 : 4;
    const long total = pixels * (c / epc);
    if (dtype == Y3_BF16)
        hipLaunchKernelGGL(add_kernel<true>, dim3(grid_for(total)), dim3(kThreads), 0, (hipStream_t)stream,
                           (const u32x4 *)a, (const u32x4 *)b, (u32x4 *)out, (long)pixels, c / epc, a_ld / epc, b_ld / epc, out_ld / epc);
    else
        hipLaunchKernelGGL(add_kernel<false>, dim3(grid_for(total)), dim3(kThreads), 0, (hipStream_t)stream,
                           (const u32x4 *)a, (const u32x4 *)b, (u32x4 *)out, (long)pixels, c / epc, a_ld / epc, b_ld / epc, out_ld / epc);
    return y3_check_launch("shortcut_add");
}

extern "C" int y3_pad_input(int dtype, const float *in, void *out, int64_t pixels, int cin, void *stream) {
    Y3_REQUIRE(dtype == Y3_F32 || dtype == Y3_BF16, Y3_E_INVALID, "pad_input: dtype %d", dtype);
    Y3_REQUIRE(in && out && pixels > 0, Y3_E_INVALID, "pad_input: bad arguments");
    Y3_REQUIRE(cin >= 1 && cin <= (dtype == Y3_BF16 ? 8 : 4), Y3_E_INVALID, "pad_input: %d channels do not fit one 16-byte chunk", cin);
    Y3_REQUIRE(aligned16(out), Y3_E_ALIGN, "pad_input: output not 16-byte aligned");
    if (dtype == Y3_BF16)
        hipLaunchKernelGGL(pad_input_kernel<true>, dim3(grid_for(pixels)), dim3(kThreads), 0, (hipStream_t)stream, in, (u32x4 *)out, (long)pixels, cin);
    else
        hipLaunchKernelGGL(pad_input_kernel<false>, dim3(grid_for(pixels)), dim3(kThreads), 0, (hipStream_t)stream, in, (u32x4 *)out, (long)pixels, cin);
    return y3_check_launch("pad_input");
}

extern "C" int y3_cast_f32_to_bf16(const float *in, void *out, int64_t n, void *stream) {
    Y3_REQUIRE(in && out && n > 0, Y3_E_INVALID, "cast: bad arguments");
    hipLaunchKernelGGL(cast_f32_bf16_kernel, dim3(grid_for(n)), dim3(kThreads), 0, (hipStream_t)stream, in,
                       (unsigned short *)out, (long)n);
    return y3_check_launch("cast_f32_to_bf16");
}

extern "C" int y3_cast_bf16_to_f32(const void *in, float *out, int64_t n, void *stream) {
    Y3_REQUIRE(in && out && n > 0, Y3_E_INVALID, "cast: bad arguments");
    hipLaunchKernelGGL(cast_bf16_f32_kernel, dim3(grid_for(n)), dim3(kThreads), 0, (hipStream_t)stream,
                       (const unsigned short *)in, out, (long)n);
    return y3_check_launch("cast_bf16_to_f32");
}
