// Darknet residual block with 128 channels for gfx950 (MI355X), bf16, in one launch:
//
//     out = x + leaky(conv3x3(leaky(conv1x1(x; W1) + b1); W2) + b2)          x: [B, H, W, 128]
//
// (reference build_model.py:113-179 twice + the shortcut of :218-227; the two blocks at 152 x 152 of YOLOv3-608.)
// The recipe of conv_c32.hip, applied to a whole block -- what a CU can stage from L2 into LDS (~14 B/clk), not MFMA or
// HBM, is what bounds every LDS-DMA-fed kernel here, so nothing but the block's input goes through that path:
//   * the 3x3's weights live in registers for the whole launch: a wave owns 32 output channels of 4 tile rows (9 taps x
//     2 k-blocks x 2 channel blocks = 36 MFMA "A" fragments = 144 registers: every fragment read of y feeds two MFMAs --
//     with 16 channels per wave and one MFMA per read the LDS array, not the MFMA pipe, set the pace); the 1x1's 16 KB
//     sit in LDS;
//   * a persistent workgroup walks tiles of 8 rows x 16 pixels; the tile's input patch (10 x 18 pixels x 128 channels,
//     zero outside the picture) goes to LDS once by LDS-DMA, double-buffered (the next tile's patch travels under this
//     tile's MFMAs), 288 bytes per pixel so that fragment reads of 16 consecutive pixels are conflict-free;
//   * phase 1: the 1x1 on the 180 patch pixels (12 row blocks x 4 channel blocks, three (row block, block pair) units per
//     wave) -> leaky -> ONE bf16 rounding -> LDS image y[pixel][64] (160 bytes per pixel, zero outside the picture =
//     the 3x3's padding);
//   * phase 2: the 3x3 as 18 k-blocks per tile row: a tap is a constant displacement in the y image; per (tap, k-block)
//     one ds_read_b128 per tile row and two MFMAs on it;
//   * epilogue: leaky, permlane16_swap pairing (8 consecutive channels per lane), + x (the patch's own centre pixels,
//     from LDS) in fp32, one bf16 rounding, 16-byte stores.
// Two raw barriers per tile; the previous tile's four stores per lane stay in flight behind a counted vmcnt(4).
// Every sum runs over k in the order of the un-fused launches (bias first, k-blocks ascending), so the result is
// bit-identical to them and to resblock_kernel (tests/test_gpu_resblock.py).
#include "common.h"

typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;

namespace {

constexpr int R1_TW = 16, R1_TH = 8;               // tile: 8 rows x 16 pixels
constexpr int R1_PW = R1_TW + 2;                   // patch width (tile + halo)
constexpr int R1_NPOS = (R1_TH + 2) * R1_PW;       // 180 patch pixels
constexpr int R1_XP = 288;                         // bytes between patch pixels of x in LDS (256 + 32)
constexpr int R1_YP = 160;                         // ... of y (128 + 32)
constexpr int R1_NT = 512;
constexpr int R1_XSLOTS = R1_NPOS * (R1_XP / 16);  // 16-byte slots of an x patch (16 data + 2 pad per pixel)
constexpr int R1_XBUF = (R1_XSLOTS + R1_NT - 1) / R1_NT * R1_NT * 16;     // whole loader passes: 57 344 bytes
constexpr int R1_YROWS = (R1_NPOS + 15) / 16 * 16; // 192: phase 1 runs on 12 row blocks
constexpr int R1_YBUF = R1_NPOS * R1_YP;           // (row blocks past the patch are computed but not stored)
constexpr int R1_W1P = 288;                        // bytes between rows of W1 in LDS (256 + 32: conflict-free fragment reads)
constexpr int R1_W1BUF = 64 * R1_W1P;
constexpr int R1_BIASBUF = (64 + 128) * 4;         // b1 | b2 as fp32 (16 registers that phase 2's pipelined fragment reads need)
constexpr int R1_LDS = 2 * R1_XBUF + R1_YBUF + R1_W1BUF + R1_BIASBUF;     // 162 688 of 163 840 bytes
static_assert(R1_LDS <= 160 * 1024, "LDS budget");
static_assert(R1_YROWS * R1_XP <= R1_XBUF, "phase 1 reads whole row blocks of the x patch buffer");

struct R1Args {
    const char *x, *w1, *w2, *zeros;
    const float *b1, *b2;
    char *out;
    int H, W, x_ld, out_ld;
    int tiles_x, tiles_img, total_tiles, flags;    // flags: 1 leaky after the 1x1, 2 after the 3x3, 32 reverse tile order
    unsigned out_bytes;
    FastDiv fd_img, fd_tx;
};

__global__ __launch_bounds__(R1_NT) __attribute__((amdgpu_waves_per_eu(2, 2)))
void resblock128_kernel(const R1Args p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char *const ybuf = smem + 2 * R1_XBUF;
    char *const w1buf = ybuf + R1_YBUF;
    const char *const biasbuf = w1buf + R1_W1BUF;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int frow = lane & 15, fq = lane >> 4;
    const int nip = wave & 1;                      // phase 1: this wave's pair of 16-channel blocks
    const int wn = wave & 3, wm = wave >> 2;       // phase 2: output channels 32 wn .. 32 wn + 31 of tile rows 4 wm .. 4 wm + 3
    const float slope1 = (p.flags & 1) ? 0.1f : 1.0f, slope2 = (p.flags & 2) ? 0.1f : 1.0f;

    // ---- weights -> registers, once ------------------------------------------------------------------------------
    bf16x8 wf2[9][2][2];                           // 3x3: [tap][k-block of 32 mid channels][16-channel block]
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int ni = 0; ni < 2; ++ni)
                wf2[t][kb][ni] = *(const bf16x8 *)(p.w2 + ((32 * wn + 16 * ni + frow) * 576 + t * 64 + kb * 32 + fq * 8) * 2);
    const int cb = (fq & 1) ? 16 + (fq - 1) * 4 : fq * 4;      // after the swap pairing: this lane's first channel of the 32
    // biases -> LDS (read back per tile as the accumulators' initial values)
    if (tid < 64) ((float *)biasbuf)[tid] = p.b1[tid];
    else if (tid < 192) ((float *)biasbuf)[tid] = p.b2[tid - 64];
    auto bias1 = [&](int j) { return *(const f32x4 *)(biasbuf + ((nip * 2 + j) * 16 + fq * 4) * 4); };
    auto bias2 = [&](int ni) { return *(const f32x4 *)(biasbuf + 256 + (32 * wn + 16 * ni + fq * 4) * 4); };
    // W1 [64][128] bf16 -> LDS, rows 288 bytes apart (the first barrier of the tile loop publishes it)
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int q = tid + R1_NT * i;             // 16-byte chunk 0..1023: row q >> 4, chunk q & 15
        *(u32x4 *)(w1buf + (q >> 4) * R1_W1P + (q & 15) * 16) = *(const u32x4 *)(p.w1 + q * 16);
    }

    auto tile_coords = [&](int tile, int &b, int &h0, int &w0) {
        const int t = xcd_remap(tile, p.total_tiles, p.flags & 32);
        b = fast_div(t, p.fd_img);
        const int ti = t - b * p.tiles_img;
        const int ty = fast_div(ti, p.fd_tx);
        h0 = ty * R1_TH;
        w0 = (ti - ty * p.tiles_x) * R1_TW;
    };
    // the tile's 10 x 18 x 128 input patch -> LDS: slot s = 18 * pixel + chunk, chunks 16 and 17 are padding (not written)
    constexpr int NPASS = R1_XBUF / (R1_NT * 16);
    auto issue_x = [&](int b, int h0, int w0, char *buf, int i0 = 0, int i1 = NPASS) {
        // (the slot -> pixel arithmetic is recomputed per tile on purpose: hoisted out of the tile loop its 7 x 4 values
        // push the kernel past 256 registers and hipcc reloads them from scratch behind vmcnt(0) waits)
        int tidv = tid;
        asm volatile("" : "+v"(tidv));
#pragma unroll
        for (int i = i0; i < i1; ++i) {
            const int s = tidv + R1_NT * i;
            const int pos = (s * 3641) >> 16, c = s - pos * 18;                   // s / 18, exact below 4096
            const int pr = (pos * 3641) >> 16, pc = pos - pr * R1_PW;             // pos / 18
            const int h = h0 - 1 + pr, w = w0 - 1 + pc;
            const bool inside = (unsigned)h < (unsigned)p.H && (unsigned)w < (unsigned)p.W;
            const char *src = inside ? p.x + ((long)((b * p.H + h) * p.W + w) * p.x_ld * 2 + c * 16) : p.zeros;
            if (pos < R1_NPOS && c < 16) glds16(src, buf + i * (R1_NT * 16) + wave * 1024);
        }
    };
    const auto rs_out = __builtin_amdgcn_make_buffer_rsrc((void *)p.out, 0, p.out_bytes, 0x00020000);
    constexpr unsigned OOB = 0xffffff00u;

    int tile = blockIdx.x;
    if (tile >= p.total_tiles) return;
    int b, h0, w0;
    tile_coords(tile, b, h0, w0);
    issue_x(b, h0, w0, smem);
    int cur = 0;
    bool first = true;
    for (; tile < p.total_tiles; tile += gridDim.x) {
        // This wave's pieces of the tile's patch have landed: they are older than the previous tile's stores -- exactly
        // four per lane, the youngest operations -- which may stay in flight.
        if (first) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        first = false;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();              // ... in every wave; the other patch buffer and the y image are free
        asm volatile("" ::: "memory");
        const char *xb = smem + cur * R1_XBUF;
        const int next = tile + (int)gridDim.x;
        int nb = 0, nh0 = 0, nw0 = 0;
        const bool more = next < p.total_tiles;
        if (more) {                                // the next tile's patch travels under this tile's MFMAs
            tile_coords(next, nb, nh0, nw0);
        }
        // ---- phase 1: y = leaky(W1 x + b1) on the patch -----------------------------------------------------------
        int frv = frow;                            // (opaque per tile: keeps phase 1's per-lane addresses out of scratch,
        asm volatile("" : "+v"(frv));              // where hipcc parks them as loop invariants and reloads behind vmcnt(0))
        // W1's fragments of this wave's block pair once per tile (32 registers that phase 2's accumulators take over
        // afterwards), the patch fragments of row block j + 1 requested ahead of row block j's MFMAs, and the two
        // 16-channel chains interleaved so that no MFMA waits for the one before it
        {
            bf16x8 wa[2][4], xf[4];
#pragma unroll
            for (int q = 0; q < 2; ++q)
#pragma unroll
                for (int kb = 0; kb < 4; ++kb)
                    wa[q][kb] = *(const bf16x8 *)(w1buf + ((nip * 2 + q) * 16 + frv) * R1_W1P + (kb * 4 + fq) * 16);
#pragma unroll
            for (int kb = 0; kb < 4; ++kb) xf[kb] = *(const bf16x8 *)(xb + ((wave >> 1) * 16 + frv) * R1_XP + (kb * 4 + fq) * 16);
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                const int pos = ((wave + 8 * j) >> 1) * 16 + frv;
                if (j > 0) {
#pragma unroll
                    for (int kb = 0; kb < 4; ++kb) xf[kb] = *(const bf16x8 *)(xb + pos * R1_XP + (kb * 4 + fq) * 16);
                }
                __builtin_amdgcn_sched_barrier(0);
                f32x4 a1[2] = {bias1(0), bias1(1)};
#pragma unroll
                for (int kb = 0; kb < 4; ++kb)
#pragma unroll
                    for (int q = 0; q < 2; ++q) a1[q] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa[q][kb], xf[kb], a1[q], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                bool in = true;                    // outside the picture: the 3x3's zero padding
                {
                    const int pr = (pos * 3641) >> 16, pc = pos - pr * R1_PW;
                    const int h = h0 - 1 + pr, w = w0 - 1 + pc;
                    in = (unsigned)h < (unsigned)p.H && (unsigned)w < (unsigned)p.W;
                }
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    f32x4 v = a1[q];
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = leaky_max(v[e], slope1);
                    const unsigned o0 = in ? pack_bf16x2(v[0], v[1]) : 0u, o1 = in ? pack_bf16x2(v[2], v[3]) : 0u;
                    if (pos < R1_NPOS) lds_store8_unordered(ybuf + pos * R1_YP + ((nip * 2 + q) * 16 + fq * 4) * 2, o0, o1);
                }
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();              // the y image is complete
        asm volatile("" ::: "memory");
        // ---- phase 2: 3x3 over the y image: 32 output channels of 4 tile rows per wave ---------------------------------
        f32x4 acc[2][4];
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[ni][r] = bias2(ni);
        // Software pipeline at half-tap granularity, register-neutral: the four fragments of k-block kb are re-read for the
        // NEXT tap right after this tap's eight MFMAs of that k-block have been issued, so the LDS round trip runs under
        // the other k-block's MFMAs.  (All eight waves leave the barrier together: un-pipelined they all read, then all
        // multiply, and the LDS array and the MFMA pipes take turns idling.)
        {
            bf16x8 yf[2][4];
            int yoff = frow * R1_YP + fq * 16;     // (opaque per tile: one address register and immediate offsets, not 72 hoisted addresses)
            asm volatile("" : "+v"(yoff));
            const char *ylane = ybuf + yoff;
            auto rd = [&](int t, int kb) {
                const int dh = t / 3, dw = t - 3 * dh;
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    yf[kb][r] = *(const bf16x8 *)(ylane + ((4 * wm + r + dh) * R1_PW + dw) * R1_YP + kb * 64);
            };
            rd(0, 0);
            rd(0, 1);
            constexpr int NT9 = 9;
#pragma unroll
            for (int t = 0; t < NT9; ++t) {
#pragma unroll
                for (int kb = 0; kb < 2; ++kb) {
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int r = 0; r < 4; ++r)
#pragma unroll
                        for (int ni = 0; ni < 2; ++ni)
                            acc[ni][r] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf2[t][kb][ni], yf[kb][r], acc[ni][r], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                    if (t + 1 < NT9) rd(t + 1, kb);
                }
                __builtin_amdgcn_sched_barrier(0);
                if (t < NPASS) {
                    if (more) issue_x(nb, nh0, nw0, smem + (cur ^ 1) * R1_XBUF, t, t + 1);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
        // ---- epilogue: leaky, pairing, + x (the patch's centre, from LDS) in fp32, one bf16 rounding, 16-byte stores ----
        const int w = w0 + frow;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int h = h0 + 4 * wm + r;
            const u32x4 xr = *(const u32x4 *)(xb + ((4 * wm + r + 1) * R1_PW + 1 + frow) * R1_XP + (32 * wn + cb) * 2);
            f32x4 x = acc[0][r], y = acc[1][r];
            float lo[4], hi[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                x[e] = leaky_max(x[e], slope2);
                y[e] = leaky_max(y[e], slope2);
                const auto sw = __builtin_amdgcn_permlane16_swap(__float_as_uint(x[e]), __float_as_uint(y[e]), false, false);
                lo[e] = __uint_as_float(sw[0]);
                hi[e] = __uint_as_float(sw[1]);
            }
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                lo[2 * e] += bf16_to_f32(xr[e] & 0xffff);
                lo[2 * e + 1] += __uint_as_float(xr[e] & 0xffff0000u);
                hi[2 * e] += bf16_to_f32(xr[2 + e] & 0xffff);
                hi[2 * e + 1] += __uint_as_float(xr[2 + e] & 0xffff0000u);
            }
            u32x4 o;
            o[0] = pack_bf16x2(lo[0], lo[1]); o[1] = pack_bf16x2(lo[2], lo[3]);
            o[2] = pack_bf16x2(hi[0], hi[1]); o[3] = pack_bf16x2(hi[2], hi[3]);
            const bool ok = h < p.H && w < p.W;
            const unsigned off = ok ? ((unsigned)((b * p.H + h) * p.W + w) * (unsigned)p.out_ld + (unsigned)(32 * wn + cb)) * 2u : OOB;
            __builtin_amdgcn_raw_buffer_store_b128(o, rs_out, off, 0, 0);
        }
        b = nb; h0 = nh0; w0 = nw0;
        cur ^= 1;
    }
}

}  // namespace

long y3_resblock128_tiles(int batch, int h, int w) {
    return (long)batch * ((h + R1_TH - 1) / R1_TH) * ((w + R1_TW - 1) / R1_TW);
}

int y3_resblock128_launch(const y3_resblock_desc *d, const void *x, const void *w1, const float *b1, const void *w2,
                          const float *b2, void *out, const void *zeros, hipStream_t stream) {
    const long pixels = (long)d->batch * d->h * d->w;
    R1Args a;
    a.x = (const char *)x; a.w1 = (const char *)w1; a.w2 = (const char *)w2; a.zeros = (const char *)zeros;
    a.b1 = b1; a.b2 = b2; a.out = (char *)out;
    a.H = d->h; a.W = d->w; a.x_ld = d->x_ld; a.out_ld = d->out_ld;
    a.tiles_x = (d->w + R1_TW - 1) / R1_TW;
    a.tiles_img = a.tiles_x * ((d->h + R1_TH - 1) / R1_TH);
    a.total_tiles = (int)y3_resblock128_tiles(d->batch, d->h, d->w);
    a.flags = d->flags & (1 | 2 | 32);
    a.out_bytes = (unsigned)(pixels * d->out_ld * 2);
    a.fd_img = make_fastdiv(a.tiles_img);
    a.fd_tx = make_fastdiv(a.tiles_x);
    static thread_local int attr_dev = -1;
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (attr_dev != dev) {
        if (hipFuncSetAttribute((const void *)resblock128_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, R1_LDS) != hipSuccess) {
            y3_set_error("resblock128: cannot reserve %d bytes of LDS", R1_LDS);
            return Y3_E_LAUNCH;
        }
        attr_dev = dev;
    }
    const int cus = y3_device_cus();
    const int grid = a.total_tiles < cus ? a.total_tiles : cus;
    hipLaunchKernelGGL(resblock128_kernel, dim3((unsigned)grid), dim3(R1_NT), R1_LDS, stream, a);
    return y3_check_launch("resblock128");
}
