// Fused convolution for gfx950 (MI355X): implicit GEMM on MFMA.
//
// Replaces tf.pad + tf.layers.conv2d + batch_normalization + leaky_relu
// (reference build_model.py:151-178), with tf.math.add (:224, shortcut) and
// resize_nearest_neighbor (:236, upsample) fused into the epilogue and
// route/concat (:191) expressed through in_ld/out_ld channel slices.
//
// GEMM view:  D[n][m] = sum_k Wp[n][k] * X[m][k]
//   m = (b, ho, wo) output pixel, n = output channel, k = (kh, kw, cin).
// The weights are the MFMA "A" operand and the gathered activations the "B"
// operand, so every lane ends up holding 4 consecutive output CHANNELS of one
// pixel: the accumulators go to LDS with 16-byte writes and leave it as whole
// NHWC pixel rows (16-byte, fully coalesced stores; residual loads likewise).
//
// Data movement: both operands are staged global -> LDS with
// global_load_lds_dwordx4 (no VGPR round trip) into a double buffer, one
// barrier per k-step.  A k-step is one 128-byte LDS row per pixel / channel
// (64 bf16 or 32 fp32); 16-byte chunks are XOR-swizzled by ((row >> 1) & 7) on
// the SOURCE address (LDS-DMA writes lane-linear) and on the fragment read, which
// makes the ds_read_b128 fragment reads bank-conflict free.  Zero padding comes
// from pointing the lane's source address at a zero page.
#include "common.h"

namespace {

struct ConvArgs {
    const char *in, *w, *res, *zeros;
    const float *bias;
    char *out;
    int H, W, Cin, in_ld;
    int Ho, Wo, HoWo, Cout8, out_ld, res_ld;
    int ks, stride, pad, Kpad, M, n_tiles, flags, step_tap, step_cin;
};

__device__ __forceinline__ int xcd_remap(int orig, int nblk) {
    // Blocks are dealt round-robin over the 8 XCDs; give each XCD a contiguous
    // range of logical tiles so that the n-tiles of one m-tile (which share the
    // gathered input) and neighbouring m-tiles (which share halo rows) meet in one L2.
    const int q = nblk >> 3, r = nblk & 7, x = orig & 7;
    const int start = x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q;
    return start + (orig >> 3);
}

__device__ __forceinline__ void glds16(const char *gsrc, char *lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)gsrc,
                                     (__attribute__((address_space(3))) void *)lds_wave_base, 16, 0, 0);
}

template <bool BF16, int BM, int BN, int WAVES_M, int WAVES_N>
__global__ __launch_bounds__(256) void conv_igemm_kernel(const ConvArgs p) {
    constexpr int ES = BF16 ? 2 : 4;          // element bytes
    constexpr int EPC = 16 / ES;              // elements per 16-byte chunk
    constexpr int BKE = 128 / ES;             // elements per k-step
    constexpr int TM = BM / WAVES_M, TN = BN / WAVES_N;
    constexpr int MI = TM / 16, NI = TN / 16;
    constexpr int A_IT = BM * 8 / 256, B_IT = BN * 8 / 256;
    constexpr int STAGE = (BM + BN) * 128;    // bytes per pipeline buffer
    constexpr int CS_LD = BN + 4;             // floats per staged output row
    static_assert(WAVES_M * WAVES_N == 4, "256 threads");
    static_assert(A_IT >= 1 && B_IT >= 1, "tile too small for the loader");

    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;

    const int bid = xcd_remap(blockIdx.x, gridDim.x);
    const int mt = bid / p.n_tiles, nt = bid - mt * p.n_tiles;
    const int m0 = mt * BM, n0 = nt * BN;

    // ---- loader set-up: thread owns LDS chunk slot (tid&7) of rows (tid>>3)+32i --------
    const int lrow = tid >> 3;
    const int kc = ((tid & 7) ^ ((tid >> 4) & 7)) * EPC;   // source chunk of this LDS slot

    long a_base[A_IT];
    unsigned a_mask[A_IT];
#pragma unroll
    for (int i = 0; i < A_IT; ++i) {
        const int gm = m0 + lrow + 32 * i;
        a_base[i] = 0;
        a_mask[i] = 0;
        if (gm < p.M) {
            const int b = gm / p.HoWo;
            const int r = gm - b * p.HoWo;
            const int ho = r / p.Wo;
            const int wo = r - ho * p.Wo;
            const int hi0 = ho * p.stride - p.pad, wi0 = wo * p.stride - p.pad;
            a_base[i] = (((long)b * p.H + hi0) * p.W + wi0) * (long)p.in_ld * ES;
            unsigned mask = 0;
            for (int kh = 0; kh < p.ks; ++kh)
                for (int kw = 0; kw < p.ks; ++kw)
                    if ((unsigned)(hi0 + kh) < (unsigned)p.H && (unsigned)(wi0 + kw) < (unsigned)p.W)
                        mask |= 1u << (kh * p.ks + kw);
            a_mask[i] = mask;
        }
    }
    long w_off[B_IT];
#pragma unroll
    for (int i = 0; i < B_IT; ++i) w_off[i] = ((long)(n0 + lrow + 32 * i) * p.Kpad + kc) * ES;

    // k position of this thread's chunk: tap = k / Cin, cin = k % Cin, advanced by BKE per k-step
    // without a division (step_tap = BKE / Cin, step_cin = BKE % Cin come from the host).
    int tap = kc / p.Cin;
    int cin = kc - tap * p.Cin;

    auto stage = [&](int kt, int buf) {
        // a_mask has no bit at or above ks*ks, so padded k (tap >= ks*ks) reads the zero page too.
        const int th = p.ks == 3 ? (tap * 11) >> 5 : tap;          // tap / 3 for tap < 16
        const int tw = tap - th * p.ks;
        const long toff = ((long)(th * p.W + tw) * p.in_ld + cin) * ES;
        char *abase = smem + buf * STAGE + wave * 1024;
#pragma unroll
        for (int i = 0; i < A_IT; ++i) {
            const char *cand = p.in + (a_base[i] + toff);
            const char *src = ((a_mask[i] >> tap) & 1u) ? cand : p.zeros;
            glds16(src, abase + i * 4096);
        }
        char *bbase = abase + BM * 128;
#pragma unroll
        for (int i = 0; i < B_IT; ++i) glds16(p.w + w_off[i] + (long)kt * 128, bbase + i * 4096);
        tap += p.step_tap;
        cin += p.step_cin;
        const bool wrap = cin >= p.Cin;
        cin = wrap ? cin - p.Cin : cin;
        tap = wrap ? tap + 1 : tap;
    };

    // ---- main loop ------------------------------------------------------------------------
    f32x4 acc[NI][MI];
#pragma unroll
    for (int ni = 0; ni < NI; ++ni)
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) acc[ni][mi] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int frow = lane & 15;
    const int fsw = (lane >> 1) & 7;
    const int fq = lane >> 4;
    const int nk = p.Kpad / BKE;

    stage(0, 0);
    for (int kt = 0; kt < nk; ++kt) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's LDS-DMA of tile kt has landed
        __syncthreads();                                    // everyone's has; buffer (kt+1)&1 is free
        if (kt + 1 < nk) stage(kt + 1, (kt + 1) & 1);
        const char *As = smem + (kt & 1) * STAGE;
        const char *Bs = As + BM * 128;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const int so = ((ks * 4 + fq) ^ fsw) * 16;
            if constexpr (BF16) {
                bf16x8 xf[MI], wf[NI];
#pragma unroll
                for (int mi = 0; mi < MI; ++mi)
                    xf[mi] = *(const bf16x8 *)(As + (wm * TM + mi * 16 + frow) * 128 + so);
#pragma unroll
                for (int ni = 0; ni < NI; ++ni)
                    wf[ni] = *(const bf16x8 *)(Bs + (wn * TN + ni * 16 + frow) * 128 + so);
#pragma unroll
                for (int ni = 0; ni < NI; ++ni)
#pragma unroll
                    for (int mi = 0; mi < MI; ++mi)
                        acc[ni][mi] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[ni], xf[mi], acc[ni][mi], 0, 0, 0);
            } else {
                f32x4 xf[MI], wf[NI];
#pragma unroll
                for (int mi = 0; mi < MI; ++mi)
                    xf[mi] = *(const f32x4 *)(As + (wm * TM + mi * 16 + frow) * 128 + so);
#pragma unroll
                for (int ni = 0; ni < NI; ++ni)
                    wf[ni] = *(const f32x4 *)(Bs + (wn * TN + ni * 16 + frow) * 128 + so);
                // lane group g holds k = 4g..4g+3 of this 16-wide k block: MFMA j sums k = 4g+j over g.
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int ni = 0; ni < NI; ++ni)
#pragma unroll
                        for (int mi = 0; mi < MI; ++mi)
                            acc[ni][mi] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[ni][j], xf[mi][j], acc[ni][mi], 0, 0, 0);
            }
        }
    }

    // ---- epilogue: accumulators -> LDS (fp32) -> bias, leaky, residual -> NHWC rows ----------
    __syncthreads();
    float *Cs = (float *)smem;
#pragma unroll
    for (int ni = 0; ni < NI; ++ni)
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) {
            const int m = wm * TM + mi * 16 + frow;
            const int n = wn * TN + ni * 16 + fq * 4;
            *(f32x4 *)(Cs + m * CS_LD + n) = acc[ni][mi];
        }
    __syncthreads();

    constexpr int NCH = BN / 8;
    constexpr int O_IT = BM * NCH / 256;
    const bool leaky = p.flags & Y3_CONV_LEAKY;
    const bool has_res = p.flags & Y3_CONV_RESIDUAL;
    const bool ups = p.flags & Y3_CONV_UPS2X;
    const bool out_f32 = !BF16 || (p.flags & Y3_CONV_OUT_F32);
#pragma unroll
    for (int i = 0; i < O_IT; ++i) {
        const int q = tid + 256 * i;
        const int row = q / NCH, ch = q % NCH;
        const int gm = m0 + row, n = n0 + ch * 8;
        if (gm >= p.M || n >= p.Cout8) continue;
        f32x4 v0 = *(const f32x4 *)(Cs + row * CS_LD + ch * 8);
        f32x4 v1 = *(const f32x4 *)(Cs + row * CS_LD + ch * 8 + 4);
        v0 += *(const f32x4 *)(p.bias + n);
        v1 += *(const f32x4 *)(p.bias + n + 4);
        if (leaky) {
#pragma unroll
            for (int j = 0; j < 4; ++j) { v0[j] = leaky01(v0[j]); v1[j] = leaky01(v1[j]); }
        }
        if (has_res) {
            const char *rp = p.res + ((long)gm * p.res_ld + n) * ES;
            if constexpr (BF16) {
                const bf16x8 r = *(const bf16x8 *)rp;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    v0[j] += bf16_to_f32((unsigned short)r[j]);
                    v1[j] += bf16_to_f32((unsigned short)r[j + 4]);
                }
            } else {
                v0 += *(const f32x4 *)rp;
                v1 += *(const f32x4 *)(rp + 16);
            }
        }
        long opix = gm;
        long row2 = 0;
        if (ups) {
            const int b = gm / p.HoWo;
            const int r = gm - b * p.HoWo;
            const int ho = r / p.Wo;
            const int wo = r - ho * p.Wo;
            row2 = 2L * p.Wo;
            opix = ((long)b * 2 * p.Ho + 2 * ho) * row2 + 2 * wo;
        }
        const int reps = ups ? 4 : 1;
        for (int rpt = 0; rpt < reps; ++rpt) {
            const long pix = opix + (rpt & 1) + (rpt >> 1) * row2;
            if (out_f32) {
                float *op = (float *)p.out + pix * p.out_ld + n;
                *(f32x4 *)op = v0;
                *(f32x4 *)(op + 4) = v1;
            } else {
                u32x4 o;
                o[0] = pack_bf16x2(v0[0], v0[1]);
                o[1] = pack_bf16x2(v0[2], v0[3]);
                o[2] = pack_bf16x2(v1[0], v1[1]);
                o[3] = pack_bf16x2(v1[2], v1[3]);
                *(u32x4 *)(p.out + (pix * p.out_ld + n) * 2) = o;
            }
        }
    }
}

// ---- direct convolution: first layer (Cin = 3) and any Cin the chunked loader cannot take ----
// One thread per output pixel, CT output channels per thread, sequential fp32 fmaf over k.
// Weights are fp32 [k][n_pad] so that a k-row of CT weights is a wave-uniform (scalar) load.
struct DirectArgs {
    const char *in;
    const float *w, *bias;
    char *out;
    int H, W, Cin, in_ld, Ho, Wo, HoWo, Cout8, out_ld, ks, stride, pad, n_pad, M, flags;
};

template <bool IN_BF16, bool OUT_BF16, int CT>
__global__ __launch_bounds__(256) void conv_direct_kernel(const DirectArgs p) {
    const int gm = blockIdx.x * 256 + threadIdx.x;
    const int nb = blockIdx.y * CT;
    if (gm >= p.M) return;
    const int b = gm / p.HoWo;
    const int r = gm - b * p.HoWo;
    const int ho = r / p.Wo;
    const int wo = r - ho * p.Wo;
    const int hi0 = ho * p.stride - p.pad, wi0 = wo * p.stride - p.pad;
    float acc[CT];
#pragma unroll
    for (int j = 0; j < CT; ++j) acc[j] = 0.0f;
    for (int kh = 0; kh < p.ks; ++kh)
        for (int kw = 0; kw < p.ks; ++kw) {
            const int hi = hi0 + kh, wi = wi0 + kw;
            const bool ok = (unsigned)hi < (unsigned)p.H && (unsigned)wi < (unsigned)p.W;
            const long pix = ok ? ((long)b * p.H + hi) * p.W + wi : 0;
            const float *wr = p.w + (long)(kh * p.ks + kw) * p.Cin * p.n_pad + nb;
            for (int c = 0; c < p.Cin; ++c) {
                float x;
                if constexpr (IN_BF16) x = bf16_to_f32(((const unsigned short *)p.in)[pix * p.in_ld + c]);
                else x = ((const float *)p.in)[pix * p.in_ld + c];
                x = ok ? x : 0.0f;
#pragma unroll
                for (int j = 0; j < CT; ++j) acc[j] = fmaf(x, wr[j], acc[j]);
                wr += p.n_pad;
            }
        }
    const bool leaky = p.flags & Y3_CONV_LEAKY;
#pragma unroll
    for (int j = 0; j < CT; ++j) {
        acc[j] += p.bias[nb + j];
        if (leaky) acc[j] = leaky01(acc[j]);
    }
#pragma unroll
    for (int j = 0; j < CT; j += 8) {
        if (nb + j >= p.Cout8) break;
        if constexpr (OUT_BF16) {
            u32x4 o;
            o[0] = pack_bf16x2(acc[j], acc[j + 1]);
            o[1] = pack_bf16x2(acc[j + 2], acc[j + 3]);
            o[2] = pack_bf16x2(acc[j + 4], acc[j + 5]);
            o[3] = pack_bf16x2(acc[j + 6], acc[j + 7]);
            *(u32x4 *)(p.out + ((long)gm * p.out_ld + nb + j) * 2) = o;
        } else {
            float *op = (float *)p.out + (long)gm * p.out_ld + nb + j;
            *(f32x4 *)op = f32x4{acc[j], acc[j + 1], acc[j + 2], acc[j + 3]};
            *(f32x4 *)(op + 4) = f32x4{acc[j + 4], acc[j + 5], acc[j + 6], acc[j + 7]};
        }
    }
}

template <bool BF16, int BM, int BN, int WM, int WN>
int launch_igemm(const ConvArgs &a, hipStream_t stream) {
    constexpr int stage_bytes = 2 * (BM + BN) * 128;
    constexpr int cs_bytes = BM * (BN + 4) * 4;
    constexpr int lds = stage_bytes > cs_bytes ? stage_bytes : cs_bytes;
    auto kern = conv_igemm_kernel<BF16, BM, BN, WM, WN>;
    static thread_local int attr_dev = -1;
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (attr_dev != dev) {
        if (hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess) {
            y3_set_error("conv: cannot reserve %d bytes of LDS", lds);
            return Y3_E_LAUNCH;
        }
        attr_dev = dev;
    }
    const int m_tiles = (a.M + BM - 1) / BM;
    const long nblk = (long)m_tiles * a.n_tiles;
    if (nblk <= 0 || nblk > 0x7fffffffL) { y3_set_error("conv: grid of %ld blocks", nblk); return Y3_E_INVALID; }
    hipLaunchKernelGGL(kern, dim3((unsigned)nblk), dim3(256), lds, stream, a);
    return y3_check_launch("conv_igemm");
}

inline int bn_for(int out_c) { return out_c > 64 ? 128 : (out_c > 32 ? 64 : 32); }

bool use_direct(const y3_conv_desc *d) {
    const int epc = d->dtype == Y3_BF16 ? 8 : 4;
    if (d->flags & Y3_CONV_IN_F32 && d->dtype == Y3_BF16) return true;
    return (d->in_c % epc) != 0;
}

}  // namespace

extern "C" int y3_conv_weight_layout(y3_conv_desc *d) {
    Y3_REQUIRE(d != nullptr, Y3_E_INVALID, "conv: null descriptor");
    Y3_REQUIRE(d->dtype == Y3_F32 || d->dtype == Y3_BF16, Y3_E_INVALID, "conv: dtype %d", d->dtype);
    Y3_REQUIRE(d->ksize == 1 || d->ksize == 3, Y3_E_INVALID, "conv: ksize %d (1 or 3)", d->ksize);
    Y3_REQUIRE(d->in_c > 0 && d->out_c > 0, Y3_E_INVALID, "conv: channels %d -> %d", d->in_c, d->out_c);
    const int k = d->ksize * d->ksize * d->in_c;
    if (use_direct(d)) {
        d->k_pad = k;
        d->n_pad = (d->out_c + 31) / 32 * 32;
        return 1;
    }
    const int bke = d->dtype == Y3_BF16 ? 64 : 32;
    const int bn = bn_for(d->out_c);
    d->k_pad = (k + bke - 1) / bke * bke;
    d->n_pad = (d->out_c + bn - 1) / bn * bn;
    return 0;
}

extern "C" int y3_conv2d_fused(const y3_conv_desc *d, const void *in, const void *w, const float *bias,
                               const void *residual, void *out, const void *zeros, void *stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    Y3_REQUIRE(d && in && w && bias && out && zeros, Y3_E_INVALID, "conv: null pointer argument");
    y3_conv_desc chk = *d;
    const int family = y3_conv_weight_layout(&chk);
    if (family < 0) return family;
    Y3_REQUIRE(chk.k_pad == d->k_pad && chk.n_pad == d->n_pad, Y3_E_INVALID,
               "conv: k_pad/n_pad (%d,%d) do not match the layout (%d,%d)", d->k_pad, d->n_pad, chk.k_pad, chk.n_pad);
    Y3_REQUIRE(d->stride == 1 || d->stride == 2, Y3_E_INVALID, "conv: stride %d", d->stride);
    Y3_REQUIRE(d->batch > 0 && d->in_h > 0 && d->in_w > 0, Y3_E_INVALID, "conv: empty input");
    const int pad = d->ksize == 3 ? 1 : 0;
    const int exp_h = (d->in_h + pad - d->ksize) / d->stride + 1 + (d->stride == 1 ? pad : 0);
    const int exp_w = (d->in_w + pad - d->ksize) / d->stride + 1 + (d->stride == 1 ? pad : 0);
    Y3_REQUIRE(d->out_h == exp_h && d->out_w == exp_w, Y3_E_INVALID, "conv: output %dx%d, expected %dx%d",
               d->out_h, d->out_w, exp_h, exp_w);
    const bool bf16 = d->dtype == Y3_BF16;
    const bool out_f32 = !bf16 || (d->flags & Y3_CONV_OUT_F32);
    const int es = bf16 ? 2 : 4, oes = out_f32 ? 4 : 2;
    const int cout8 = (d->out_c + 7) / 8 * 8;
    Y3_REQUIRE(d->in_ld >= d->in_c && d->out_ld >= cout8, Y3_E_INVALID, "conv: in_ld %d / out_ld %d too small", d->in_ld, d->out_ld);
    Y3_REQUIRE(aligned16(out) && (d->out_ld * oes) % 16 == 0, Y3_E_ALIGN, "conv: output not 16-byte aligned");
    const long M = (long)d->batch * d->out_h * d->out_w;
    Y3_REQUIRE(M < 0x7fffffffL, Y3_E_INVALID, "conv: %ld output pixels", M);
    if (d->flags & Y3_CONV_RESIDUAL) {
        Y3_REQUIRE(residual != nullptr, Y3_E_INVALID, "conv: residual flag without pointer");
        Y3_REQUIRE(aligned16(residual) && (d->res_ld * es) % 16 == 0 && d->res_ld >= cout8, Y3_E_ALIGN, "conv: residual not aligned");
        Y3_REQUIRE(!(d->flags & Y3_CONV_UPS2X), Y3_E_INVALID, "conv: residual + upsample unsupported");
    }

    if (family == 1) {
        Y3_REQUIRE(!(d->flags & (Y3_CONV_RESIDUAL | Y3_CONV_UPS2X)), Y3_E_INVALID, "conv(direct): residual/upsample unsupported");
        DirectArgs a;
        a.in = (const char *)in; a.w = (const float *)w; a.bias = bias; a.out = (char *)out;
        a.H = d->in_h; a.W = d->in_w; a.Cin = d->in_c; a.in_ld = d->in_ld;
        a.Ho = d->out_h; a.Wo = d->out_w; a.HoWo = d->out_h * d->out_w; a.Cout8 = cout8; a.out_ld = d->out_ld;
        a.ks = d->ksize; a.stride = d->stride; a.pad = pad; a.n_pad = d->n_pad; a.M = (int)M; a.flags = d->flags;
        const bool in_bf16 = bf16 && !(d->flags & Y3_CONV_IN_F32);
        const int ct = (d->n_pad % 32 == 0 && d->out_c > 16) ? 32 : 16;
        dim3 grid((unsigned)((M + 255) / 256), (unsigned)((cout8 + ct - 1) / ct));
#define Y3_DIRECT(IB, OB, CT) hipLaunchKernelGGL((conv_direct_kernel<IB, OB, CT>), grid, dim3(256), 0, stream, a)
        if (ct == 32) {
            if (in_bf16) { if (out_f32) Y3_DIRECT(true, false, 32); else Y3_DIRECT(true, true, 32); }
            else         { if (out_f32) Y3_DIRECT(false, false, 32); else Y3_DIRECT(false, true, 32); }
        } else {
            if (in_bf16) { if (out_f32) Y3_DIRECT(true, false, 16); else Y3_DIRECT(true, true, 16); }
            else         { if (out_f32) Y3_DIRECT(false, false, 16); else Y3_DIRECT(false, true, 16); }
        }
#undef Y3_DIRECT
        return y3_check_launch("conv_direct");
    }

    Y3_REQUIRE(aligned16(in) && (d->in_ld * es) % 16 == 0, Y3_E_ALIGN, "conv: input not 16-byte aligned");
    Y3_REQUIRE(aligned16(w), Y3_E_ALIGN, "conv: weights not 16-byte aligned");
    Y3_REQUIRE(aligned16(bias), Y3_E_ALIGN, "conv: bias not 16-byte aligned");
    ConvArgs a;
    a.in = (const char *)in; a.w = (const char *)w; a.res = (const char *)residual; a.zeros = (const char *)zeros;
    a.bias = bias; a.out = (char *)out;
    a.H = d->in_h; a.W = d->in_w; a.Cin = d->in_c; a.in_ld = d->in_ld;
    a.Ho = d->out_h; a.Wo = d->out_w; a.HoWo = d->out_h * d->out_w; a.Cout8 = cout8;
    a.out_ld = d->out_ld; a.res_ld = d->res_ld;
    a.ks = d->ksize; a.stride = d->stride; a.pad = pad; a.Kpad = d->k_pad; a.M = (int)M; a.flags = d->flags;
    const int bn = bn_for(d->out_c);
    a.n_tiles = d->n_pad / bn;
    const int bke = bf16 ? 64 : 32;
    a.step_tap = bke / d->in_c;
    a.step_cin = bke % d->in_c;
    if (bf16) {
        if (bn == 128) return launch_igemm<true, 128, 128, 2, 2>(a, stream);
        if (bn == 64) return launch_igemm<true, 128, 64, 2, 2>(a, stream);
        return launch_igemm<true, 128, 32, 4, 1>(a, stream);
    }
    if (bn == 128) return launch_igemm<false, 128, 128, 2, 2>(a, stream);
    if (bn == 64) return launch_igemm<false, 128, 64, 2, 2>(a, stream);
    return launch_igemm<false, 128, 32, 4, 1>(a, stream);
}
