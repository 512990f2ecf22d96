// Darknet-53's first residual block for gfx950 (MI355X), bf16: [convolutional 1x1 64 -> 32], [convolutional 3x3 32 -> 64],
// [shortcut from=-3] (reference build_model.py:113-179 twice + :218-227; yolov3.cfg layers 2-4) as ONE launch on the
// 304 x 304 feature map of a 608 input: 5.9 M pixels per 64-image batch, the largest tensors of the network.
//
// In separate launches the block moves x three times (the stem's conv C reads it from LDS but writes y = W1 x; the
// 3x3 reads y back and x again as its shortcut operand) -- 2.3 GB per batch.  Here x is read once and the sum written
// once, 1.5 GB:
//   * both weight matrices live in registers for the whole launch (conv_c32.hip's layout for the 3x3: 72 registers per
//     wave; the 1x1's 32 x 64 matrix is 16 more);
//   * a persistent workgroup walks 16 x 16-pixel tiles; the tile's 18 x 18 x 64 patch of x goes to LDS once by LDS-DMA,
//     160 bytes per pixel (128 + 32 of pad: fragment reads of 16 consecutive pixels hit different bank groups),
//     double-buffered: the next tile's patch is requested before this tile's MFMAs;
//   * phase 1: y = leaky(W1 x + b1) on all 324 patch pixels (zero outside the picture: the 3x3's padding) -> an LDS
//     image with conv_c32.hip's 96-byte pixel pitch; phase 2: conv_c32.hip's tap loop over that image;
//   * the shortcut operand is the patch's centre, already in LDS.
// Bits: phase 1 is the stem's conv C (conv_stem.hip: bias as the initial accumulator, two k-blocks of 32 with the same
// channel-to-k-slot assignment, leaky, one bf16 rounding), phase 2 and the epilogue are conv3x3_c32_kernel's, so a
// plan with this kernel gives the same activations as the plan without it (tests/test_gpu_resblock.py).
#include "common.h"

namespace {

constexpr int R6_T = 16;                           // tile edge in pixels
constexpr int R6_P = R6_T + 2;                     // patch edge
constexpr int R6_NPOS = R6_P * R6_P;               // 324 patch pixels
constexpr int R6_XP = 160;                         // bytes between patch pixels of x in LDS (128 data + 32 pad)
constexpr int R6_YP = 96;                          // ... of the y image (64 data + 32 pad)
constexpr int R6_NT = 512;
constexpr int R6_XSLOTS = R6_NPOS * 10;            // 16-byte slots of a patch
constexpr int R6_XBUF = (R6_XSLOTS + R6_NT - 1) / R6_NT * R6_NT * 16;   // 57,344 bytes per patch buffer (whole loader passes)
constexpr int R6_NBLK = (R6_NPOS + 15) / 16;       // 21 row blocks of 16 patch pixels in phase 1
constexpr int R6_YBUF = R6_NPOS * R6_YP;           // 31,104
constexpr int R6_LDS = 2 * R6_XBUF + R6_YBUF;      // 145,792
constexpr int R6_K2PAD = 320;                      // packed 3x3 weight row: 9 * 32 = 288 padded to a multiple of 64
static_assert(R6_NBLK * 16 * R6_XP <= R6_XBUF, "phase 1 reads whole row blocks of the patch buffer");
static_assert(R6_LDS <= 160 * 1024, "LDS budget");

struct R6Args {
    const char *x, *w1, *w2, *zeros;
    const float *b1, *b2;
    char *out;
    int H, W, x_ld, out_ld;
    int tiles_x, tiles_img, total_tiles, flags;
    unsigned out_bytes;
    FastDiv fd_img, fd_tx;
};

__global__ __launch_bounds__(R6_NT) __attribute__((amdgpu_waves_per_eu(2, 2)))
void resblock64_kernel(const R6Args p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char *const ybuf = smem + 2 * R6_XBUF;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int frow = lane & 15, fq = lane >> 4;
    const int wm = wave >> 1, wn = wave & 1;       // phase 2: tile rows 4 wm .. 4 wm + 3, channels 32 wn .. 32 wn + 31
    const float slope1 = (p.flags & 1) ? 0.1f : 1.0f, slope2 = (p.flags & 2) ? 0.1f : 1.0f;

    // ---- weights -> registers, once ------------------------------------------------------------------------------
    bf16x8 wf[9][2];                               // 3x3: [tap][16-channel block]
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
            wf[t][ni] = *(const bf16x8 *)(p.w2 + ((wn * 32 + ni * 16 + frow) * R6_K2PAD + t * 32 + fq * 8) * 2);
    f32x4 bias2[2];
#pragma unroll
    for (int ni = 0; ni < 2; ++ni) bias2[ni] = *(const f32x4 *)(p.b2 + wn * 32 + ni * 16 + fq * 4);
    // 1x1: k-slot fq of k-block kb holds channels 32 kb + 8 perm .. + 7, the assignment of the stem's conv C
    const int perm = ((fq & 1) << 1) | (fq >> 1);
    bf16x8 w1f[2][2];                              // [k-block][16-channel block]
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
            w1f[kb][ni] = *(const bf16x8 *)(p.w1 + ((ni * 16 + frow) * 64 + (kb * 4 + perm) * 8) * 2);
    f32x4 bias1[2];
#pragma unroll
    for (int ni = 0; ni < 2; ++ni) bias1[ni] = *(const f32x4 *)(p.b1 + ni * 16 + fq * 4);
    const int cb = (fq & 1) ? 16 + (fq - 1) * 4 : fq * 4;      // after the swap pairing: this lane's first channel of the 32

    auto tile_coords = [&](int tile, int &b, int &h0, int &w0) {
        const int t = xcd_remap(tile, p.total_tiles, p.flags & 32);
        b = fast_div(t, p.fd_img);
        const int ti = t - b * p.tiles_img;
        const int ty = fast_div(ti, p.fd_tx);
        h0 = ty * R6_T;
        w0 = (ti - ty * p.tiles_x) * R6_T;
    };
    // the tile's 18 x 18 x 64 input patch -> LDS: slot s = 10 * pixel + chunk, chunks 8 and 9 are padding (not written).
    // A tile whose patch lies inside the picture (4 of 5 at 304 x 304) takes per-lane offsets computed once per launch:
    // two VALU instructions per loader pass instead of twenty (this kernel issues 1.4 VALU cycles per MFMA cycle).
    constexpr int NPASS = R6_XBUF / (R6_NT * 16);
    int rel[NPASS];                                // byte offset of this lane's slot from the patch's first pixel, -1 = none
#pragma unroll
    for (int i = 0; i < NPASS; ++i) {
        const int s = tid + R6_NT * i;
        const int pos = (s * 6554) >> 16, c = s - pos * 10;
        const int pr = (pos * 3641) >> 16, pc = pos - pr * R6_P;
        rel[i] = (pos < R6_NPOS && c < 8) ? (pr * p.W + pc) * p.x_ld * 2 + c * 16 : -1;
    }
    auto issue_x = [&](int b, int h0, int w0, char *buf, int i0 = 0, int i1 = NPASS) {
        if (h0 >= 1 && w0 >= 1 && h0 + R6_T + 1 <= p.H && w0 + R6_T + 1 <= p.W) {      // (wave-uniform)
            const char *base = p.x + (long)((b * p.H + h0 - 1) * p.W + w0 - 1) * p.x_ld * 2;
#pragma unroll
            for (int i = i0; i < i1; ++i)
                if (rel[i] >= 0) glds16(base + rel[i], buf + i * (R6_NT * 16) + wave * 1024);
            return;
        }
        int tidv = tid;
        asm volatile("" : "+v"(tidv));             // (per-tile arithmetic on purpose: see conv_resblock128.hip)
#pragma unroll
        for (int i = i0; i < i1; ++i) {
            const int s = tidv + R6_NT * i;
            const int pos = (s * 6554) >> 16, c = s - pos * 10;                   // s / 10, exact below 16384
            const int pr = (pos * 3641) >> 16, pc = pos - pr * R6_P;              // pos / 18, exact below 512
            const int h = h0 - 1 + pr, w = w0 - 1 + pc;
            const bool inside = (unsigned)h < (unsigned)p.H && (unsigned)w < (unsigned)p.W;
            const char *src = inside ? p.x + ((long)((b * p.H + h) * p.W + w) * p.x_ld * 2 + c * 16) : p.zeros;
            if (pos < R6_NPOS && c < 8) glds16(src, buf + i * (R6_NT * 16) + wave * 1024);
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
        const char *xb = smem + cur * R6_XBUF;
        const int next = tile + (int)gridDim.x;
        int nb = 0, nh0 = 0, nw0 = 0;
        const bool more = next < p.total_tiles;
        if (more) {                                // the next tile's patch travels under this tile's MFMAs
            tile_coords(next, nb, nh0, nw0);
        }
        // ---- phase 1: y = leaky(W1 x + b1) on the patch: row blocks wave, wave + 8, wave + 16 ----------------------------
        const bool interior = h0 >= 1 && w0 >= 1 && h0 + R6_T + 1 <= p.H && w0 + R6_T + 1 <= p.W;
        int frv = frow;                            // (opaque per tile: keeps phase 1's per-lane addresses out of scratch)
        asm volatile("" : "+v"(frv));
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const int blk = wave + 8 * j;
            if (blk < R6_NBLK) {                   // (wave-uniform)
                const int pos = blk * 16 + frv;
                bf16x8 xf[2];
#pragma unroll
                for (int kb = 0; kb < 2; ++kb) xf[kb] = *(const bf16x8 *)(xb + pos * R6_XP + (kb * 4 + perm) * 16);
                f32x4 a1[2] = {bias1[0], bias1[1]};
#pragma unroll
                for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                    for (int ni = 0; ni < 2; ++ni)
                        a1[ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w1f[kb][ni], xf[kb], a1[ni], 0, 0, 0);
                bool in = true;                    // outside the picture: the 3x3's zero padding
                if (!interior) {
                    const int pr = (pos * 3641) >> 16, pc = pos - pr * R6_P;
                    const int h = h0 - 1 + pr, w = w0 - 1 + pc;
                    in = (unsigned)h < (unsigned)p.H && (unsigned)w < (unsigned)p.W;
                }
#pragma unroll
                for (int ni = 0; ni < 2; ++ni) {
                    f32x4 v = a1[ni];
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = leaky_max(v[e], slope1);
                    const unsigned o0 = in ? pack_bf16x2(v[0], v[1]) : 0u, o1 = in ? pack_bf16x2(v[2], v[3]) : 0u;
                    if (pos < R6_NPOS) lds_store8_unordered(ybuf + pos * R6_YP + (ni * 16 + fq * 4) * 2, o0, o1);
                }
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();              // the y image is complete
        asm volatile("" ::: "memory");
        // ---- phase 2: 3x3 over the y image (conv_c32.hip's tap loop) ---------------------------------------------------
        f32x4 acc[2][4];
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
#pragma unroll
            for (int mi = 0; mi < 4; ++mi) acc[ni][mi] = bias2[ni];
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const int dh = t / 3, dw = t - 3 * dh;
#pragma unroll
            for (int mi = 0; mi < 4; ++mi) {
                const bf16x8 yf = *(const bf16x8 *)(ybuf + ((4 * wm + mi + dh) * R6_P + dw + frow) * R6_YP + fq * 16);
#pragma unroll
                for (int ni = 0; ni < 2; ++ni)
                    acc[ni][mi] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[t][ni], yf, acc[ni][mi], 0, 0, 0);
            }
            // one loader pass after each of the first taps (all of them at the top of the tile: +2 %): an LDS-DMA instruction
            // holds its wave for 60-180 cycles, which the SIMD's other wave fills with MFMAs here and cannot up there
            if (t < R6_XBUF / (R6_NT * 16)) {
                __builtin_amdgcn_sched_barrier(0);
                if (more) issue_x(nb, nh0, nw0, smem + (cur ^ 1) * R6_XBUF, t, t + 1);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        // ---- epilogue: leaky, pairing, + x (the patch's centre, from LDS) in fp32, one bf16 rounding, 16-byte stores ----
        const int w = w0 + frow;
#pragma unroll
        for (int mi = 0; mi < 4; ++mi) {
            const int h = h0 + 4 * wm + mi;
            const u32x4 xr = *(const u32x4 *)(xb + ((4 * wm + mi + 1) * R6_P + 1 + frow) * R6_XP + (32 * wn + cb) * 2);
            f32x4 x = acc[0][mi], y = acc[1][mi];
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

long y3_resblock64_tiles(int batch, int h, int w) {
    return (long)batch * ((h + R6_T - 1) / R6_T) * ((w + R6_T - 1) / R6_T);
}

int y3_resblock64_launch(const y3_resblock_desc *d, const void *x, const void *w1, const float *b1, const void *w2,
                         const float *b2, void *out, const void *zeros, hipStream_t stream) {
    const long pixels = (long)d->batch * d->h * d->w;
    Y3_REQUIRE((long)R6_P * d->w * d->x_ld * 2 < 0x7fffffffL, Y3_E_INVALID, "resblock64: rows of %d x %d elements are too long", d->w, d->x_ld);
    R6Args a;
    a.x = (const char *)x; a.w1 = (const char *)w1; a.w2 = (const char *)w2; a.zeros = (const char *)zeros;
    a.b1 = b1; a.b2 = b2; a.out = (char *)out;
    a.H = d->h; a.W = d->w; a.x_ld = d->x_ld; a.out_ld = d->out_ld;
    a.tiles_x = (d->w + R6_T - 1) / R6_T;
    a.tiles_img = a.tiles_x * ((d->h + R6_T - 1) / R6_T);
    a.total_tiles = (int)y3_resblock64_tiles(d->batch, d->h, d->w);
    a.flags = d->flags & (1 | 2 | 32);
    a.out_bytes = (unsigned)(pixels * d->out_ld * 2);
    a.fd_img = make_fastdiv(a.tiles_img);
    a.fd_tx = make_fastdiv(a.tiles_x);
    static thread_local int attr_dev = -1;
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (attr_dev != dev) {
        if (hipFuncSetAttribute((const void *)resblock64_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, R6_LDS) != hipSuccess) {
            y3_set_error("resblock64: cannot reserve %d bytes of LDS", R6_LDS);
            return Y3_E_LAUNCH;
        }
        attr_dev = dev;
    }
    const int cus = y3_device_cus();
    const int grid = a.total_tiles < cus ? a.total_tiles : cus;
    hipLaunchKernelGGL(resblock64_kernel, dim3((unsigned)grid), dim3(R6_NT), R6_LDS, stream, a);
    return y3_check_launch("resblock64");
}
