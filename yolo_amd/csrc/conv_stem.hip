// Fused network stem for gfx950 (MI355X), bf16 mode:
//   conv A  3x3 stride 1, 3 -> 32   on the fp32 image     (reference build_model.py:113-179, Darknet layer 0)
//   conv B  3x3 stride 2, 32 -> 64  (pad top/left, VALID; build_model.py:151-154, layer 1)
//   conv C  1x1, 64 -> 32           (layer 2; optional)
// each with folded BN, bias and leaky-ReLU.  Un-fused, A writes and B re-reads a 608x608x32 tensor
// (1.5 GB per 64 images each way) for 0.6 GFLOP/image; here that tensor lives only in LDS.
//
// One workgroup (4 waves) owns an 8 x 16 tile of B's output and loops over tiles (persistent):
//   1. the 19 x 35 x 3 image patch under the tile -> LDS as bf16 (next tile's patch is already in
//      flight in registers while this one is computed);
//   2. conv A on the 17 x 33 patch of A-outputs the tile needs: MFMA 16x16x32 with K = 27 (+5 zero),
//      B-operand fragments gathered from the image patch; results (bias, leaky, zero outside the
//      image = B's padding) go to LDS as bf16, stored once per horizontal tap phase kw so that
//      every fragment read of step 3 is a 16-pixel aligned run: P[kw][row][ox] = A(row, 2*ox + kw);
//   3. conv B: 9 taps x (32 channels = one MFMA k-step); the 64x288 weights stay in registers for
//      the whole kernel.  The epilogue pairs lanes with v_permlane16_swap so that every lane owns
//      8 consecutive channels: 16-byte stores, and exactly the B-operand layout conv C needs;
//   4. conv C straight from those registers.
// Weights are the MFMA A operand, pixels the B operand, as in conv_igemm.hip.
#include "common.h"

namespace {

constexpr int TH = 8, TW = 16;                 // tile of conv-B output pixels
constexpr int AP_H = 2 * TH + 1;               // 17 rows of conv-A output under the tile
constexpr int AP_W = 2 * TW + 1;               // 33 columns
constexpr int IM_H = AP_H + 2;                 // 19 image rows
constexpr int IM_W3 = (AP_W + 2) * 3;          // 105 floats per image-patch row
constexpr int IM_PITCH = 106;
constexpr int IM_ELEMS = IM_H * IM_W3;         // 1995
constexpr int IMG_ROWS_PER_WAVE = (IM_H + 3) / 4;   // 5: wave w stages image-patch rows w, w+4, ...; two 64-lane halves per row
constexpr int NPIX_A = AP_H * AP_W;            // 561
constexpr int NBLK_A = (NPIX_A + 15) / 16;     // 36
constexpr int P_BYTES = 3 * AP_H * TW * 64;    // 52 224
constexpr int IMG_BYTES = (IM_H * IM_PITCH * 2 + 15) / 16 * 16;
constexpr int BIAS_BYTES = (32 + 64 + 32) * 4;   // b0 | b1 | b2
constexpr int WC_BYTES = 32 * 64 * 2;          // conv C weights [32][64] bf16

struct StemArgs {
    const float *img;            // [B][H][W][3]
    const float *w0, *b0;        // conv A: fp32 [27][n0_pad] (direct layout), bias
    const char *w1;              // conv B: bf16 [64][k1_pad], k = tap*32 + c
    const float *b1;
    const char *w2;              // conv C: bf16 [32][k2_pad], k = c
    const float *b2;
    char *out1, *out2;
    int B, H, W, Ho, Wo, tiles_y, tiles_x, n0_pad, k1_pad, k2_pad, out1_ld, out2_ld, flags;
};

__device__ __forceinline__ f32x4 bias_act(f32x4 v, const float *bias, bool leaky) {     // bias: LDS
    v += *(const f32x4 *)bias;
    if (leaky) {
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = leaky01(v[j]);
    }
    return v;
}

// Lanes (row r, r^1 of 16) trade halves: on return the lane holds 8 consecutive channels of one
// pixel, starting at channel 16*nb + 4*fq (fq even) or 16*(nb+1) + 4*(fq-1) (fq odd), where x came
// from n-block nb and y from n-block nb + 1.
__device__ __forceinline__ u32x4 pair_lanes(f32x4 x, f32x4 y) {
    unsigned x0 = pack_bf16x2(x[0], x[1]), x1 = pack_bf16x2(x[2], x[3]);
    unsigned y0 = pack_bf16x2(y[0], y[1]), y1 = pack_bf16x2(y[2], y[3]);
    auto s0 = __builtin_amdgcn_permlane16_swap(x0, y0, false, false);
    auto s1 = __builtin_amdgcn_permlane16_swap(x1, y1, false, false);
    u32x4 o;
    o[0] = s0[0]; o[1] = s1[0]; o[2] = s0[1]; o[3] = s1[1];
    return o;
}

template <bool HAS_C>
__global__ __launch_bounds__(256, 2) void conv_stem_kernel(const StemArgs p) {
    __shared__ __attribute__((aligned(16))) char smem[P_BYTES + IMG_BYTES + BIAS_BYTES + WC_BYTES];
    char *P = smem;
    unsigned short *img = (unsigned short *)(smem + P_BYTES);
    float *bias_s = (float *)(smem + P_BYTES + IMG_BYTES);     // b0[32] | b1[64] | b2[32]
    char *wc_s = smem + P_BYTES + IMG_BYTES + BIAS_BYTES;       // conv C weights, row pitch 128 B, chunk c of row n at slot c ^ (n & 7)

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int frow = lane & 15, fq = lane >> 4;
    const bool leaky_a = p.flags & 1, leaky_b = p.flags & 2, leaky_c = p.flags & 4;

    // ---- weights that stay in registers ------------------------------------------------------
    // conv A k-slots: lane group fq < 3 holds image row kh = fq, (kw, c) = 0..7; group 3 holds the ninth
    // (kw, c) = (2, 2) of rows 0..2 and five zeros.  So a lane's 8 values are img[k_a + i * k_s], i < k_n.
    bf16x8 wa[2];
    const int k_a = fq < 3 ? fq * IM_PITCH : 8, k_s = fq < 3 ? 1 : IM_PITCH, k_n = fq < 3 ? 8 : 3;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int k = fq < 3 ? fq * 9 + i : i * 9 + 8;
#pragma unroll
        for (int nb = 0; nb < 2; ++nb) {
            const float v = i < k_n ? p.w0[k * p.n0_pad + nb * 16 + frow] : 0.0f;
            wa[nb][i] = (short)f32_to_bf16(v);
        }
    }
    if (tid < 32) bias_s[tid] = p.b0[tid];
    if (tid < 64) bias_s[32 + tid] = p.b1[tid];
    if constexpr (HAS_C) {
        if (tid < 32) bias_s[96 + tid] = p.b2[tid];
        const int n = tid >> 3, c = tid & 7;                       // 32 rows x 8 chunks = 256 threads
        *(u32x4 *)(wc_s + n * 128 + ((c ^ (n & 7)) * 16)) = *(const u32x4 *)(p.w2 + ((long)n * p.k2_pad + c * 8) * 2);
    }
    bf16x8 wb[9][4];
#pragma unroll
    for (int tap = 0; tap < 9; ++tap)
#pragma unroll
        for (int nb = 0; nb < 4; ++nb)
            wb[tap][nb] = *(const bf16x8 *)(p.w1 + ((long)(nb * 16 + frow) * p.k1_pad + tap * 32 + fq * 8) * 2);
    const int perm = ((fq & 1) << 1) | (fq >> 1);              // conv C: k-slot fq holds channels perm*8.. (see pair_lanes)

    const int tiles_img = p.tiles_y * p.tiles_x;
    const int total = p.B * tiles_img;
    const long row3 = (long)p.W * 3;

    // Patch rows are dealt to waves (row = wave + 4*q) so that the row's address and its inside-the-image
    // test are wave-uniform; lanes take columns lane and lane + 64 of the row's 105 floats.
    float pre[IMG_ROWS_PER_WAVE][2];
    unsigned pre_ok = 0;            // which of pre[] lie inside the image (applied when the patch is staged, so that
                                    // nothing consumes the loads, and waits for them, before the next tile)
    auto load_patch = [&](int t) {
        const int b = t / tiles_img;
        const int r = t - b * tiles_img;
        const int ty = r / p.tiles_x, tx = r - ty * p.tiles_x;
        const int y0 = 2 * ty * TH - 2, x0 = (2 * tx * TW - 2) * 3;
        const int ix0 = x0 + lane, ix1 = x0 + lane + 64;
        const bool okx0 = (unsigned)ix0 < (unsigned)row3, okx1 = lane + 64 < IM_W3 && (unsigned)ix1 < (unsigned)row3;
        const int cx0 = min(max(ix0, 0), (int)row3 - 1), cx1 = min(max(ix1, 0), (int)row3 - 1);
        pre_ok = 0;
#pragma unroll
        for (int q = 0; q < IMG_ROWS_PER_WAVE; ++q) {
            const int iy = y0 + wave + 4 * q;
            const bool oky = wave + 4 * q < IM_H && (unsigned)iy < (unsigned)p.H;
            const float *rowp = p.img + ((long)b * p.H + min(max(iy, 0), p.H - 1)) * row3;     // always load (clamped), select later
            pre[q][0] = rowp[cx0];
            pre[q][1] = rowp[cx1];
            pre_ok |= ((oky && okx0) ? 1u : 0u) << (2 * q) | ((oky && okx1) ? 2u : 0u) << (2 * q);
        }
    };

    // All weight loads retire here, and the compiler's wait-count model knows it (a builtin, not inline asm):
    // otherwise it assumes they may still be pending at the loop header and drains vmcnt, prefetch included,
    // before the first use of the weights in every iteration.
    __builtin_amdgcn_s_waitcnt(0x0F70);            // vmcnt(0)
    int t = blockIdx.x;
    if (t < total) load_patch(t);
    for (; t < total; t += gridDim.x) {
        const int b = t / tiles_img;
        const int r = t - b * tiles_img;
        const int ty = r / p.tiles_x, tx = r - ty * p.tiles_x;
        const int oy0 = ty * TH, ox0 = tx * TW;

        // ---- 1. image patch -> LDS (bf16) ---------------------------------------------------
#pragma unroll
        for (int q = 0; q < IMG_ROWS_PER_WAVE; ++q) {
            const int rr = wave + 4 * q;
            if (rr < IM_H) {
                img[rr * IM_PITCH + lane] = ((pre_ok >> (2 * q)) & 1u) ? f32_to_bf16(pre[q][0]) : (unsigned short)0;
                if (lane + 64 < IM_W3)
                    img[rr * IM_PITCH + lane + 64] = ((pre_ok >> (2 * q + 1)) & 1u) ? f32_to_bf16(pre[q][1]) : (unsigned short)0;
            }
        }
        __syncthreads();            // patch visible; every wave is past step 3 of the previous tile
        if (t + (int)gridDim.x < total) load_patch(t + gridDim.x);

        // ---- 2. conv A on the 17 x 33 patch -> P[kw][row][ox] --------------------------------
        for (int pb = wave; pb < NBLK_A; pb += 4) {
            const int pi = pb * 16 + frow;
            const bool valid = pi < NPIX_A;
            const int pp = valid ? pi : NPIX_A - 1;
            const int py = pp / AP_W, px = pp - py * AP_W;
            const int base = py * IM_PITCH + px * 3;
            bf16x8 xf;
#pragma unroll
            for (int i = 0; i < 8; ++i) xf[i] = i < k_n ? (short)img[base + k_a + i * k_s] : (short)0;
            const f32x4 z = {0.f, 0.f, 0.f, 0.f};
            f32x4 a[2];
            a[0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa[0], xf, z, 0, 0, 0);
            a[1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa[1], xf, z, 0, 0, 0);
            const int ay = 2 * oy0 - 1 + py, ax = 2 * ox0 - 1 + px;
            const bool inimg = (unsigned)ay < (unsigned)p.H && (unsigned)ax < (unsigned)p.W;
            const int oxe = px >> 1;
            const int sw0 = ((oxe >> 3) & 1) << 1, sw2 = (((oxe - 1) >> 3) & 1) << 1;
#pragma unroll
            for (int nb = 0; nb < 2; ++nb) {
                f32x4 v = bias_act(a[nb], bias_s + nb * 16 + fq * 4, leaky_a);
                uint2 o;
                o.x = inimg ? pack_bf16x2(v[0], v[1]) : 0u;
                o.y = inimg ? pack_bf16x2(v[2], v[3]) : 0u;
                const int chunk = nb * 2 + (fq >> 1), half = (fq & 1) * 8;
                if (valid) {
                    if (px & 1) {
                        *(uint2 *)(P + (((1 * AP_H + py) * TW + oxe) * 64) + ((chunk ^ sw0) * 16) + half) = o;
                    } else {
                        if (oxe < TW) *(uint2 *)(P + (((0 * AP_H + py) * TW + oxe) * 64) + ((chunk ^ sw0) * 16) + half) = o;
                        if (oxe >= 1) *(uint2 *)(P + (((2 * AP_H + py) * TW + oxe - 1) * 64) + ((chunk ^ sw2) * 16) + half) = o;
                    }
                }
            }
        }
        __syncthreads();

        // ---- 3. conv B: 9 taps, two output rows per wave ---------------------------------------
        // one output row at a time (four independent accumulator chains are enough to pace the MFMA pipe,
        // and the second row's accumulators would not fit next to the 144 weight registers)
        const int slot = (fq ^ (((frow >> 3) & 1) << 1)) * 16;
        const int cb = (fq & 1) ? 16 + (fq - 1) * 4 : fq * 4;       // first of this lane's 8 channels in a pair
#pragma unroll
        for (int mb = 0; mb < 2; ++mb) {
            f32x4 acc[4];
#pragma unroll
            for (int nb = 0; nb < 4; ++nb) acc[nb] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int kh = 0; kh < 3; ++kh)
#pragma unroll
                for (int kw = 0; kw < 3; ++kw) {
                    const int row = 2 * (2 * wave + mb) + kh;
                    const bf16x8 xf = *(const bf16x8 *)(P + ((kw * AP_H + row) * TW + frow) * 64 + slot);
#pragma unroll
                    for (int nb = 0; nb < 4; ++nb)
                        acc[nb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wb[kh * 3 + kw][nb], xf, acc[nb], 0, 0, 0);
                }

            // ---- epilogue B (+ conv C) --------------------------------------------------------
            const int oy = oy0 + 2 * wave + mb, ox = ox0 + frow;
            const bool ok = oy < p.Ho && ox < p.Wo;
            const long pix = ((long)b * p.Ho + oy) * p.Wo + ox;
            u32x4 y[2];
#pragma unroll
            for (int pr = 0; pr < 2; ++pr) {
                const f32x4 vx = bias_act(acc[2 * pr], bias_s + 32 + (2 * pr) * 16 + fq * 4, leaky_b);
                const f32x4 vy = bias_act(acc[2 * pr + 1], bias_s + 32 + (2 * pr + 1) * 16 + fq * 4, leaky_b);
                y[pr] = pair_lanes(vx, vy);
                if (ok) *(u32x4 *)(p.out1 + (pix * p.out1_ld + pr * 32 + cb) * 2) = y[pr];
            }
            if constexpr (HAS_C) {
                f32x4 c2[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
                for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                    for (int nb = 0; nb < 2; ++nb) {               // conv C's weights come from LDS, per use
                        const int n = nb * 16 + frow, c = kb * 4 + perm;
                        const bf16x8 wc = *(const bf16x8 *)(wc_s + n * 128 + ((c ^ (n & 7)) * 16));
                        c2[nb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wc, __builtin_bit_cast(bf16x8, y[kb]), c2[nb], 0, 0, 0);
                    }
                const f32x4 vx = bias_act(c2[0], bias_s + 96 + fq * 4, leaky_c);
                const f32x4 vy = bias_act(c2[1], bias_s + 96 + 16 + fq * 4, leaky_c);
                const u32x4 o = pair_lanes(vx, vy);
                if (ok) *(u32x4 *)(p.out2 + (pix * p.out2_ld + cb) * 2) = o;
            }
        }
    }
}

}  // namespace

extern "C" int y3_conv_stem(const y3_stem_desc *d, const float *image, const float *w0, const float *b0,
                            const void *w1, const float *b1, const void *w2, const float *b2,
                            void *out1, void *out2, void *stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    Y3_REQUIRE(d && image && w0 && b0 && w1 && b1 && out1, Y3_E_INVALID, "stem: null pointer argument");
    Y3_REQUIRE(d->batch > 0 && d->in_h > 1 && d->in_w > 1, Y3_E_INVALID, "stem: empty input");
    Y3_REQUIRE(d->c0 == 32 && d->c1 == 64 && (d->c2 == 0 || d->c2 == 32), Y3_E_INVALID,
               "stem: channels 3 -> %d -> %d -> %d (only 32, 64, 0|32)", d->c0, d->c1, d->c2);
    Y3_REQUIRE(d->n0_pad >= 32 && d->k1_pad >= 288 && d->k1_pad % 8 == 0, Y3_E_INVALID, "stem: weight layout (%d, %d)", d->n0_pad, d->k1_pad);
    Y3_REQUIRE(aligned16(w1) && aligned16(b0) && aligned16(b1) && aligned16(out1) && d->out1_ld >= 64 && d->out1_ld % 8 == 0,
               Y3_E_ALIGN, "stem: conv B operands not 16-byte aligned");
    const bool has_c = d->c2 != 0;
    if (has_c) {
        Y3_REQUIRE(w2 && b2 && out2, Y3_E_INVALID, "stem: conv C without weights/output");
        Y3_REQUIRE(d->k2_pad >= 64 && d->k2_pad % 8 == 0, Y3_E_INVALID, "stem: conv C weight layout %d", d->k2_pad);
        Y3_REQUIRE(aligned16(w2) && aligned16(b2) && aligned16(out2) && d->out2_ld >= 32 && d->out2_ld % 8 == 0, Y3_E_ALIGN,
                   "stem: conv C operands not 16-byte aligned");
    }
    StemArgs a;
    a.img = image; a.w0 = w0; a.b0 = b0; a.w1 = (const char *)w1; a.b1 = b1; a.w2 = (const char *)w2; a.b2 = b2;
    a.out1 = (char *)out1; a.out2 = (char *)out2;
    a.B = d->batch; a.H = d->in_h; a.W = d->in_w;
    a.Ho = (d->in_h - 2) / 2 + 1; a.Wo = (d->in_w - 2) / 2 + 1;        // pad top/left 1, VALID, stride 2
    a.tiles_y = (a.Ho + TH - 1) / TH; a.tiles_x = (a.Wo + TW - 1) / TW;
    a.n0_pad = d->n0_pad; a.k1_pad = d->k1_pad; a.k2_pad = d->k2_pad;
    a.out1_ld = d->out1_ld; a.out2_ld = d->out2_ld; a.flags = d->flags;
    const long total = (long)a.B * a.tiles_y * a.tiles_x;
    Y3_REQUIRE(total < 0x7fffffffL, Y3_E_INVALID, "stem: %ld tiles", total);
    const int grid = (int)(total < 512 ? total : 512);                 // two persistent workgroups per CU
    if (has_c) hipLaunchKernelGGL(conv_stem_kernel<true>, dim3(grid), dim3(256), 0, stream, a);
    else hipLaunchKernelGGL(conv_stem_kernel<false>, dim3(grid), dim3(256), 0, stream, a);
    return y3_check_launch("conv_stem");
}
