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
#include <type_traits>

namespace {

constexpr int TH = 8, TW = 16;                 // tile of conv-B output pixels
constexpr int AP_H = 2 * TH + 1;               // 17 rows of conv-A output under the tile
constexpr int AP_W = 2 * TW + 1;               // 33 columns
constexpr int IM_H = AP_H + 2;                 // 19 image rows
constexpr int IM_W = AP_W + 2;                 // 35 image columns
constexpr int IM_PITCH = 36;                   // LDS patch row pitch in pixels; a pixel is 4 bf16 (r, g, b, 0) = 8 bytes
constexpr int IM_ODD = (IM_PITCH / 2) * 8;     // a row holds its even columns first, then (144 bytes on) its odd ones:
                                               // lane frow reads columns 2*frow.. -> 8-byte lane stride, no bank conflict
constexpr int IM_PIX = IM_H * IM_W;            // 665
constexpr int IMG_LOADS = (IM_PIX + 255) / 256;   // 3 pixels per thread
constexpr int P_BYTES = 3 * AP_H * TW * 64;    // 52 224
constexpr int IMG_BYTES = IM_H * IM_PITCH * 8;  // 5 472
constexpr int BIAS_BYTES = (32 + 64 + 32) * 4;   // b0 | b1 | b2
constexpr int WC_BYTES = 32 * 64 * 2;          // conv C weights [32][64] bf16
constexpr int WA_BYTES = 4 * 64 * 16;          // conv A weight fragments [k-block][n-block][lane]
constexpr int WB8_BYTES = 4 * 64 * 16;         // conv B weight fragments of the ninth tap [n-block][lane]
constexpr int DUMP_BYTES = 64 * 8;             // target of stores that have no destination

struct StemArgs {
    const float *img;            // [B][H][W][3]
    const float *w0, *b0;        // conv A: fp32 [27][n0_pad] (direct layout), bias
    const char *w1;              // conv B: bf16 [64][k1_pad], k = tap*32 + c
    const float *b1;
    const char *w2;              // conv C: bf16 [32][k2_pad], k = c
    const float *b2;
    char *out1, *out2;
    int B, H, W, Ho, Wo, tiles_y, tiles_x, n0_pad, k1_pad, k2_pad, out1_ld, out2_ld, flags;
    unsigned out1_bytes, out2_bytes;
};

// leaky-ReLU as max(x, slope*x) (slope 0.1, or 1 for a linear layer; common.h leaky_max), branch-free.
// Bit-identical to x > 0 ? x : 0.1x for every non-NaN x.  The bias is not added here: it is the initial
// value of the MFMA accumulator (one LDS read instead of an LDS read and four adds).
__device__ __forceinline__ f32x4 act(f32x4 v, float slope) {
#pragma unroll
    for (int j = 0; j < 4; ++j) v[j] = leaky_max(v[j], slope);
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
    __shared__ __attribute__((aligned(16))) char smem[P_BYTES + IMG_BYTES + BIAS_BYTES + WC_BYTES + WA_BYTES + WB8_BYTES + DUMP_BYTES];
    char *P = smem;
    char *img = smem + P_BYTES;                                 // [IM_H][IM_PITCH] pixels of 4 bf16
    float *bias_s = (float *)(smem + P_BYTES + IMG_BYTES);     // b0[32] | b1[64] | b2[32]
    char *wc_s = smem + P_BYTES + IMG_BYTES + BIAS_BYTES;       // conv C weights, row pitch 128 B, chunk c of row n at slot c ^ (n & 7)

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int frow = lane & 15, fq = lane >> 4;
    char *wa_s = smem + P_BYTES + IMG_BYTES + BIAS_BYTES + WC_BYTES + lane * 16;   // this lane's conv A fragments, 1 KiB apart
    const float slope_a = (p.flags & 1) ? 0.1f : 1.0f, slope_b = (p.flags & 2) ? 0.1f : 1.0f, slope_c = (p.flags & 4) ? 0.1f : 1.0f;

    // ---- weights that stay in registers ------------------------------------------------------
    // conv A as two MFMA k-blocks over 8-byte (r,g,b,0) pixels: lane group fq < 3 holds image row kh = fq;
    // k-block 0 = pixels px, px+1 (kw 0, 1), k-block 1 = pixels px+2, px+3 (kw 2, and a pixel whose weights are
    // zero); the fourth channel and all of lane group 3 have zero weights too.  Every fragment read is then an
    // aligned ds_read_b64 at a compile-time offset from one per-lane address: no gather arithmetic.
    // The four weight fragments live in LDS (every wave writes the same values) and are re-read per block.
    if (wave == 0) {
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int nb = 0; nb < 2; ++nb) {
                bf16x8 w;
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const int kw = kb * 2 + (i >> 2), c = i & 3;
                    const bool used = fq < 3 && kw < 3 && c < 3;
                    const float v = used ? p.w0[(fq * 9 + kw * 3 + c) * p.n0_pad + nb * 16 + frow] : 0.0f;
                    w[i] = (short)f32_to_bf16(v);
                }
                *(bf16x8 *)(wa_s + (kb * 2 + nb) * 1024) = w;
            }
    }
    if (tid < IM_H) *(uint2 *)(img + tid * IM_PITCH * 8 + IM_ODD + (IM_W >> 1) * 8) = uint2{0u, 0u};      // the pad column (35, odd) stays zero
    if (tid < 32) bias_s[tid] = p.b0[tid];
    if (tid < 64) bias_s[32 + tid] = p.b1[tid];
    if constexpr (HAS_C) {
        if (tid < 32) bias_s[96 + tid] = p.b2[tid];
        const int n = tid >> 3, c = tid & 7;                       // 32 rows x 8 chunks = 256 threads
        *(u32x4 *)(wc_s + n * 128 + ((c ^ (n & 7)) * 16)) = *(const u32x4 *)(p.w2 + ((long)n * p.k2_pad + c * 8) * 2);
    }
    // conv B: taps 0..7 stay in registers (128 of them); the ninth tap's fragments would not fit beside the
    // rest of the kernel, they sit in LDS next to conv A's (wa_s + 4 KiB) and are re-read per output row.
    bf16x8 wb[8][4];
#pragma unroll
    for (int tap = 0; tap < 8; ++tap)
#pragma unroll
        for (int nb = 0; nb < 4; ++nb)
            wb[tap][nb] = *(const bf16x8 *)(p.w1 + ((long)(nb * 16 + frow) * p.k1_pad + tap * 32 + fq * 8) * 2);
    if (wave == 0) {
#pragma unroll
        for (int nb = 0; nb < 4; ++nb)
            *(bf16x8 *)(wa_s + WA_BYTES + nb * 1024) = *(const bf16x8 *)(p.w1 + ((long)(nb * 16 + frow) * p.k1_pad + 8 * 32 + fq * 8) * 2);
    }
    const int perm = ((fq & 1) << 1) | (fq >> 1);              // conv C: k-slot fq holds channels perm*8.. (see pair_lanes)

    const int tiles_img = p.tiles_y * p.tiles_x;
    const int total = p.B * tiles_img;
    // Outputs go through buffer descriptors: a lane with nothing to store gets an out-of-range offset and the
    // hardware drops its store.  No branch around the stores means their number per tile is fixed, which lets
    // the compiler wait for the prefetched image loads with an exact vmcnt instead of draining the stores too.
    const auto rsrc1 = __builtin_amdgcn_make_buffer_rsrc((void *)p.out1, 0, p.out1_bytes, 0x00020000);
    const auto rsrc2 = __builtin_amdgcn_make_buffer_rsrc((void *)(HAS_C ? p.out2 : p.out1), 0, HAS_C ? p.out2_bytes : 0u, 0x00020000);

    // Patch pixel i = tid + 256*q of the 19 x 35 patch: three floats, always loaded (clamped address); whether
    // it lies inside the image is applied when the patch is staged, so that nothing consumes the loads, and
    // waits for them, before the next tile.
    float pre[IMG_LOADS][3];
    unsigned pre_ok = 0;
    auto load_patch = [&](int t) {
        const int b = t / tiles_img;
        const int r = t - b * tiles_img;
        const int ty = r / p.tiles_x, tx = r - ty * p.tiles_x;
        const int y0 = 2 * ty * TH - 2, x0 = 2 * tx * TW - 2;
        pre_ok = 0;
        int tv = tid;                       // opaque per call: keeps the per-thread patch coordinates from being
        asm volatile("" : "+v"(tv));        // hoisted out of the tile loop into registers the weights need
#pragma unroll
        for (int q = 0; q < IMG_LOADS; ++q) {
            const int i = min(tv + 256 * q, IM_PIX - 1);
            const int rr = i / IM_W, cc = i - rr * IM_W;
            const int iy = y0 + rr, ix = x0 + cc;
            const bool ok = (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
            const float *px = p.img + (((long)b * p.H + min(max(iy, 0), p.H - 1)) * p.W + min(max(ix, 0), p.W - 1)) * 3;
            pre[q][0] = px[0]; pre[q][1] = px[1]; pre[q][2] = px[2];
            pre_ok |= (ok ? 1u : 0u) << q;
        }
    };

    // ---- per-lane constants of step 2 --------------------------------------------------------------
    // Waves 0, 2 take the even patch columns px = 2*frow, waves 1, 3 the odd ones px = 2*frow + 1; patch rows
    // py = (wave >> 1) + 2j.  Column 32 (17 pixels) is left over: wave 2 takes its rows 0..15, wave 3 row 16.
    auto p_addr = [&](int kw, int py, int ox, int nb) {     // byte offset in P of this lane's 8 bytes of n-block nb
        const int chunk = nb * 2 + (fq >> 1), sw = (ox >> 1) & 3;
        return ((kw * AP_H + py) * TW + ox) * 64 + ((chunk ^ sw) * 16) + (fq & 1) * 8;
    };
    const int par = wave & 1, pyw = wave >> 1;
    const int rowsel = fq < 3 ? fq : 0;
    const int px_l = 2 * frow + par;
    const int rd_l = (pyw + rowsel) * IM_PITCH * 8 + (frow + par) * 8;          // even-plane slot of column px_l (par 0) / px_l + 1 (par 1); + j * 2 * IM_PITCH * 8
    const int wA0 = p_addr(par, pyw, frow, 0), wA1 = p_addr(par, pyw, frow, 1);   // + j * 2 * TW * 64
    // even columns also feed P[2][.][ox - 1]; lane frow = 0 has no such slot and repeats its first store
    // (same bytes, same address) instead, so that the second store needs no per-lane mask
    const int wB0 = frow >= 1 ? p_addr(2, pyw, frow - 1, 0) : wA0, wB1 = frow >= 1 ? p_addr(2, pyw, frow - 1, 1) : wA1;
    const bool doB = par == 0;

    // All weight loads retire here, and the compiler's wait-count model knows it (a builtin, not inline asm):
    // otherwise it assumes they may still be pending at the loop header and drains vmcnt, prefetch included,
    // before the first use of the weights in every iteration.
    __builtin_amdgcn_s_waitcnt(0x0F70);            // vmcnt(0)
    // ---- 1. image patch -> LDS (bf16, 4 channels per pixel).  Runs at the END of the previous tile's
    // iteration (all waves are past step 2 by then), so that the wait for the prefetched loads sits behind a
    // known number of younger stores and the compiler can give it an exact vmcnt.
    auto stage_patch = [&]() {
    int tv = tid;
    asm volatile("" : "+v"(tv));        // see load_patch
#pragma unroll
    for (int q = 0; q < IMG_LOADS; ++q) {
        const int i = tv + 256 * q;
        const int rr = i / IM_W, cc = i - rr * IM_W;
        const bool ok = (pre_ok >> q) & 1u;
        uint2 o;
        o.x = ok ? pack_bf16x2(pre[q][0], pre[q][1]) : 0u;
        o.y = ok ? pack_bf16x2(pre[q][2], 0.0f) : 0u;
        if (i < IM_PIX) *(uint2 *)(img + rr * IM_PITCH * 8 + (cc & 1) * IM_ODD + (cc >> 1) * 8) = o;
    }
    };
    int t = blockIdx.x;
    if (t < total) {
        load_patch(t);
        stage_patch();
    }
    for (; t < total; t += gridDim.x) {
        const int b = t / tiles_img;
        const int r = t - b * tiles_img;
        const int ty = r / p.tiles_x, tx = r - ty * p.tiles_x;
        const int oy0 = ty * TH, ox0 = tx * TW;

        __syncthreads();            // patch visible; every wave is past step 3 of the previous tile
        bf16x8 wa[4];               // conv A's weight fragments: read once per tile, not once per block
#pragma unroll
        for (int i = 0; i < 4; ++i) wa[i] = *(const bf16x8 *)(wa_s + i * 1024);
        const f32x4 ba[2] = {*(const f32x4 *)(bias_s + fq * 4), *(const f32x4 *)(bias_s + 16 + fq * 4)};
        // unconditional (the last iteration re-loads a valid tile for nobody): a conditional prefetch makes the
        // compiler copy the loaded registers at the join, which waits for the loads right here
        load_patch(min(t + (int)gridDim.x, total - 1));

        // ---- 2. conv A on the 17 x 33 patch -> P[kw][row][ox] --------------------------------
        // odd = the block's first column is odd (waves 1, 3): columns c, c+1, c+2, c+3 are then the odd plane's slot
        // before rd, rd itself, the odd plane's slot at rd, and the even plane's next slot
        auto conv_a_block = [&](auto odd, const char *rd, bool in_img, char *a0, char *a1, bool do_a, char *b0, char *b1, bool do_b) {
            constexpr bool ODD = decltype(odd)::value;
            const uint2 r0 = *(const uint2 *)(rd + (ODD ? IM_ODD - 8 : 0)), r1 = *(const uint2 *)(rd + (ODD ? 0 : IM_ODD));
            const uint2 r2 = *(const uint2 *)(rd + (ODD ? IM_ODD : 8)), r3 = *(const uint2 *)(rd + (ODD ? 8 : IM_ODD + 8));
            const bf16x8 x0 = __builtin_bit_cast(bf16x8, u32x4{r0.x, r0.y, r1.x, r1.y});
            const bf16x8 x1 = __builtin_bit_cast(bf16x8, u32x4{r2.x, r2.y, r3.x, r3.y});
#pragma unroll
            for (int nb = 0; nb < 2; ++nb) {
                f32x4 a = ba[nb];
                a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa[nb], x0, a, 0, 0, 0);
                a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa[2 + nb], x1, a, 0, 0, 0);
                const f32x4 v = act(a, slope_a);
                uint2 o;                        // outside the image conv A's output is conv B's zero padding
                o.x = in_img ? pack_bf16x2(v[0], v[1]) : 0u;
                o.y = in_img ? pack_bf16x2(v[2], v[3]) : 0u;
                if (do_a) *(uint2 *)(nb ? a1 : a0) = o;
                if (do_b) *(uint2 *)(nb ? b1 : b0) = o;         // do_a, do_b are wave-uniform
            }
        };
        {
            const bool ax_ok = (unsigned)(2 * ox0 - 1 + px_l) < (unsigned)p.W;
            auto columns = [&](auto odd) {
#pragma unroll
                for (int j = 0; j < (AP_H + 1) / 2; ++j) {
                    const int py = pyw + 2 * j;                       // wave-uniform
                    if (py < AP_H) {
                        const bool ay_ok = (unsigned)(2 * oy0 - 1 + py) < (unsigned)p.H;
                        conv_a_block(odd, img + rd_l + j * (2 * IM_PITCH * 8), ay_ok && ax_ok, P + wA0 + j * (2 * TW * 64),
                                     P + wA1 + j * (2 * TW * 64), true, P + wB0 + j * (2 * TW * 64), P + wB1 + j * (2 * TW * 64), doB);
                    }
                }
            };
            if (par) columns(std::true_type{});
            else columns(std::false_type{});
            if (wave >= 2) {                                      // left-over column 32: lane frow <-> patch row py_x
                int fr = frow;
                asm volatile("" : "+v"(fr));
                const int py_x = wave == 2 ? fr : 16;
                const bool do_x = wave == 2 || frow == 0;
                const int rd_x = (py_x + rowsel) * IM_PITCH * 8 + 16 * 8;      // column 32: even plane, slot 16
                const int dump_x = P_BYTES + IMG_BYTES + BIAS_BYTES + WC_BYTES + WA_BYTES + WB8_BYTES + lane * 8;
                const int wX0 = do_x ? p_addr(2, py_x, 15, 0) : dump_x, wX1 = do_x ? p_addr(2, py_x, 15, 1) : dump_x;
                const bool in_x = (unsigned)(2 * oy0 - 1 + py_x) < (unsigned)p.H && (unsigned)(2 * ox0 + 31) < (unsigned)p.W;
                conv_a_block(std::false_type{}, img + rd_x, in_x, nullptr, nullptr, false, P + wX0, P + wX1, true);
            }
        }
        __syncthreads();

        // ---- 3. conv B: 9 taps, two output rows per wave ---------------------------------------
        // one output row at a time (four independent accumulator chains are enough to pace the MFMA pipe,
        // and the second row's accumulators would not fit next to the 144 weight registers)
        const int slot = (fq ^ ((frow >> 1) & 3)) * 16;       // same swizzle as p_addr: conflict-free ds_read_b128, 2-way ds_write_b64
        const int cb = (fq & 1) ? 16 + (fq - 1) * 4 : fq * 4;       // first of this lane's 8 channels in a pair
        bf16x8 wb8[4];              // ninth tap: read once per tile (the wave's two output rows share it)
#pragma unroll
        for (int nb = 0; nb < 4; ++nb) wb8[nb] = *(const bf16x8 *)(wa_s + WA_BYTES + nb * 1024);
#pragma unroll
        for (int mb = 0; mb < 2; ++mb) {
            f32x4 acc[4];
#pragma unroll
            for (int nb = 0; nb < 4; ++nb) acc[nb] = *(const f32x4 *)(bias_s + 32 + nb * 16 + fq * 4);
#pragma unroll
            for (int kh = 0; kh < 3; ++kh)
#pragma unroll
                for (int kw = 0; kw < 3; ++kw) {
                    const int row = 2 * (2 * wave + mb) + kh;
                    const bf16x8 xf = *(const bf16x8 *)(P + ((kw * AP_H + row) * TW + frow) * 64 + slot);
#pragma unroll
                    for (int nb = 0; nb < 4; ++nb) {
                        const bf16x8 w = kh * 3 + kw < 8 ? wb[(kh * 3 + kw) & 7][nb] : wb8[nb];
                        acc[nb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w, xf, acc[nb], 0, 0, 0);
                    }
                }

            // ---- epilogue B (+ conv C) --------------------------------------------------------
            const int oy = oy0 + 2 * wave + mb, ox = ox0 + frow;
            const bool ok = oy < p.Ho && ox < p.Wo;
            const unsigned pix = ok ? (unsigned)((b * p.Ho + oy) * p.Wo + ox) : 0x7fffffffu;
            const unsigned off1 = ok ? (pix * p.out1_ld + cb) * 2u : 0xffffff00u;
            u32x4 y[2];
#pragma unroll
            for (int pr = 0; pr < 2; ++pr) {
                const f32x4 vx = act(acc[2 * pr], slope_b), vy = act(acc[2 * pr + 1], slope_b);
                y[pr] = pair_lanes(vx, vy);
                __builtin_amdgcn_raw_buffer_store_b128(y[pr], rsrc1, off1, pr * 64, 0);
            }
            if constexpr (HAS_C) {
                f32x4 c2[2] = {*(const f32x4 *)(bias_s + 96 + fq * 4), *(const f32x4 *)(bias_s + 96 + 16 + fq * 4)};
#pragma unroll
                for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                    for (int nb = 0; nb < 2; ++nb) {               // conv C's weights come from LDS, per use
                        const int n = nb * 16 + frow, c = kb * 4 + perm;
                        const bf16x8 wc = *(const bf16x8 *)(wc_s + n * 128 + ((c ^ (n & 7)) * 16));
                        c2[nb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wc, __builtin_bit_cast(bf16x8, y[kb]), c2[nb], 0, 0, 0);
                    }
                const f32x4 vx = act(c2[0], slope_c), vy = act(c2[1], slope_c);
                const u32x4 o = pair_lanes(vx, vy);
                __builtin_amdgcn_raw_buffer_store_b128(o, rsrc2, ok ? (pix * p.out2_ld + cb) * 2u : 0xffffff00u, 0, 0);
            }
        }
        stage_patch();          // next tile's patch (a repeat of the last tile's at the very end)
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
    a.out1_ld = d->out1_ld; a.out2_ld = d->out2_ld; a.flags = d->flags & 7;
    const long bytes1 = (long)a.B * a.Ho * a.Wo * a.out1_ld * 2, bytes2 = (long)a.B * a.Ho * a.Wo * a.out2_ld * 2;
    Y3_REQUIRE(bytes1 < 0xffff0000L && bytes2 < 0xffff0000L, Y3_E_INVALID, "stem: output of %ld bytes exceeds a 4 GiB buffer descriptor", bytes1);
    a.out1_bytes = (unsigned)bytes1; a.out2_bytes = (unsigned)bytes2;
    const long total = (long)a.B * a.tiles_y * a.tiles_x;
    Y3_REQUIRE(total < 0x7fffffffL, Y3_E_INVALID, "stem: %ld tiles", total);
    const int grid = (int)(total < 512 ? total : 512);                 // two persistent workgroups per CU
    if (has_c) hipLaunchKernelGGL(conv_stem_kernel<true>, dim3(grid), dim3(256), 0, stream, a);
    else hipLaunchKernelGGL(conv_stem_kernel<false>, dim3(grid), dim3(256), 0, stream, a);
    return y3_check_launch("conv_stem");
}
