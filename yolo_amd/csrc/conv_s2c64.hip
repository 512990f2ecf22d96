// 3x3 stride-2 convolution with 64 input and 128 output channels for gfx950 (MI355X), bf16: Darknet-53's layer 5
// (reference build_model.py:113-179 with the top/left padding of :124,151-154), 304 x 304 -> 152 x 152 at a 608 input:
// 1.5 M output pixels per 64-image batch, 757 MB in and 378 MB out.
//
// conv_c32.hip's recipe at stride 2.  As an implicit GEMM the layer gathers every input pixel 2.25 times through LDS-DMA
// and re-stages the 147 KB weight matrix for every 256 pixels; here
//   * the WEIGHTS live in registers for the whole launch: a wave computes 32 output channels, 36 MFMA "A" fragments =
//     144 registers, loaded once (conv_resblock128.hip's layout for its 3x3);
//   * a persistent workgroup walks tiles of 4 x 16 output pixels; the tile's 9 x 33 x 64 input patch goes to LDS once by
//     LDS-DMA, 144 bytes per pixel (128 + 16 of pad: a fragment read takes 16 output pixels of a row = every second
//     patch pixel, 288 bytes apart, which the ds_read_b128 lane groups serve without a bank conflict), double-buffered:
//     the next tile's patch is requested among this tile's first taps;
//   * a tap is a constant displacement in the patch: per tap four fragment reads and eight MFMAs per wave.
// Every output is the same chain of MFMAs as in conv_igemm_ring_kernel (bias as the initial accumulator value, k-blocks
// of 32 = taps in order, channels in order, leaky, one bf16 rounding), so the results are bit-identical to it
// (tests/test_gpu_conv.py); y3_conv2d_fused picks this kernel by shape, nothing changes at the ABI.
#include "common.h"

namespace {

constexpr int S2_TH = 4, S2_TW = 16;               // output tile
constexpr int S2_PH = 2 * S2_TH + 1, S2_PW = 2 * S2_TW + 1;   // 9 x 33 input patch
constexpr int S2_NPOS = S2_PH * S2_PW;             // 297 patch pixels
constexpr int S2_PITCH = 144;                      // bytes between patch pixels in LDS
constexpr int S2_NT = 512;
constexpr int S2_SLOTS = S2_NPOS * 9;              // 16-byte slots of a patch (8 data + 1 pad per pixel)
constexpr int S2_NPASS = (S2_SLOTS + S2_NT - 1) / S2_NT;      // 6 loader passes
constexpr int S2_BUF = S2_NPASS * S2_NT * 16;      // 49,152 bytes per patch buffer
constexpr int S2_KPAD = 576;                       // packed weight row: 9 * 64
static_assert(S2_NPASS <= 9, "one loader pass per tap");

struct S2Args {
    const char *in, *w, *zeros;
    const float *bias;
    char *out;
    int H, W, Ho, Wo, in_ld, out_ld;               // input and output picture
    int tiles_x, tiles_img, total_tiles, flags;
    unsigned out_bytes;
    FastDiv fd_img, fd_tx;
};

__global__ __launch_bounds__(S2_NT) __attribute__((amdgpu_waves_per_eu(2, 2)))
void conv3x3s2_c64_kernel(const S2Args p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int frow = lane & 15, fq = lane >> 4;
    const int wn = wave & 3, wm = wave >> 2;       // this wave: channels 32 wn .. 32 wn + 31 of tile rows 2 wm, 2 wm + 1
    const float slope = (p.flags & Y3_CONV_LEAKY) ? 0.1f : 1.0f;

    // ---- the wave's 32 rows of the weight matrix as MFMA "A" fragments: [tap][k-block of 32 channels][16-channel block] ----
    bf16x8 wf[9][2][2];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int ni = 0; ni < 2; ++ni)
                wf[t][kb][ni] = *(const bf16x8 *)(p.w + ((32 * wn + 16 * ni + frow) * S2_KPAD + t * 64 + kb * 32 + fq * 8) * 2);
    f32x4 bias4[2];
#pragma unroll
    for (int ni = 0; ni < 2; ++ni) bias4[ni] = *(const f32x4 *)(p.bias + 32 * wn + 16 * ni + fq * 4);

    auto tile_coords = [&](int tile, int &b, int &h0, int &w0) {
        const int t = xcd_remap(tile, p.total_tiles, p.flags & Y3_CONV_REVERSE);
        b = fast_div(t, p.fd_img);
        const int ti = t - b * p.tiles_img;
        const int ty = fast_div(ti, p.fd_tx);
        h0 = ty * S2_TH;
        w0 = (ti - ty * p.tiles_x) * S2_TW;
    };
    // The tile's input patch -> LDS: slot s = 9 * pixel + chunk, chunk 8 is padding (not written); patch pixel (pr, pc) is
    // input pixel (2 h0 - 1 + pr, 2 w0 - 1 + pc), zero outside the picture.  A tile whose patch lies inside the picture
    // takes per-lane offsets computed once per launch (see conv_resblock64.hip).
    int rel[S2_NPASS];
#pragma unroll
    for (int i = 0; i < S2_NPASS; ++i) {
        const int s = tid + S2_NT * i;
        const int pos = (s * 7282) >> 16, c = s - pos * 9;                        // s / 9, exact below 4096
        const int pr = (pos * 1986) >> 16, pc = pos - pr * S2_PW;                 // pos / 33, exact below 512
        rel[i] = (pos < S2_NPOS && c < 8) ? (pr * p.W + pc) * p.in_ld * 2 + c * 16 : -1;
    }
    auto issue_x = [&](int b, int h0, int w0, char *buf, int i0, int i1) {
        const int hi0 = 2 * h0 - 1, wi0 = 2 * w0 - 1;
        if (hi0 >= 0 && wi0 >= 0 && hi0 + S2_PH <= p.H && wi0 + S2_PW <= p.W) {   // (wave-uniform)
            const char *base = p.in + (long)((b * p.H + hi0) * p.W + wi0) * p.in_ld * 2;
#pragma unroll
            for (int i = i0; i < i1; ++i)
                if (rel[i] >= 0) glds16(base + rel[i], buf + i * (S2_NT * 16) + wave * 1024);
            return;
        }
        int tidv = tid;
        asm volatile("" : "+v"(tidv));             // (per-tile arithmetic on purpose: see conv_resblock128.hip)
#pragma unroll
        for (int i = i0; i < i1; ++i) {
            const int s = tidv + S2_NT * i;
            const int pos = (s * 7282) >> 16, c = s - pos * 9;
            const int pr = (pos * 1986) >> 16, pc = pos - pr * S2_PW;
            const int h = hi0 + pr, w = wi0 + pc;
            const bool inside = (unsigned)h < (unsigned)p.H && (unsigned)w < (unsigned)p.W;
            const char *src = inside ? p.in + ((long)((b * p.H + h) * p.W + w) * p.in_ld * 2 + c * 16) : p.zeros;
            if (pos < S2_NPOS && c < 8) glds16(src, buf + i * (S2_NT * 16) + wave * 1024);
        }
    };
    const int cb = (fq & 1) ? 16 + (fq - 1) * 4 : fq * 4;      // after the swap pairing: this lane's first channel of the 32
    constexpr unsigned OOB = 0xffffff00u;
    const auto rs_out = __builtin_amdgcn_make_buffer_rsrc((void *)p.out, 0, p.out_bytes, 0x00020000);

    int tile = blockIdx.x;
    if (tile >= p.total_tiles) return;
    int b, h0, w0;
    tile_coords(tile, b, h0, w0);
    issue_x(b, h0, w0, smem, 0, S2_NPASS);
    int cur = 0;
    bool first = true;
    for (; tile < p.total_tiles; tile += gridDim.x) {
        // This wave's pieces of the tile's patch have landed: they are older than the previous tile's stores -- exactly
        // two per lane, the youngest operations -- which may stay in flight.
        if (first) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
        first = false;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();              // ... in every wave; nobody reads the other buffer any more
        asm volatile("" ::: "memory");             // (not __syncthreads: it would also wait for the stores)
        const char *xb = smem + cur * S2_BUF;
        const int next = tile + (int)gridDim.x;
        const bool more = next < p.total_tiles;
        int nb = 0, nh0 = 0, nw0 = 0;
        if (more) tile_coords(next, nb, nh0, nw0);
        int frv = frow;                            // (opaque per tile: keeps the fragment addresses out of scratch)
        asm volatile("" : "+v"(frv));
        f32x4 acc[2][2];
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
#pragma unroll
            for (int r = 0; r < 2; ++r) acc[ni][r] = bias4[ni];
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const int dh = t / 3, dw = t - 3 * dh;
            bf16x8 xf[2][2];                       // the tap's four fragment reads first, then its eight MFMAs
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int r = 0; r < 2; ++r)
                    xf[kb][r] = *(const bf16x8 *)(xb + ((2 * (2 * wm + r) + dh) * S2_PW + 2 * frv + dw) * S2_PITCH + (kb * 4 + fq) * 16);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int r = 0; r < 2; ++r)
#pragma unroll
                    for (int ni = 0; ni < 2; ++ni)
                        acc[ni][r] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[t][kb][ni], xf[kb][r], acc[ni][r], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            // one loader pass of the next tile's patch after each of the first taps: an LDS-DMA instruction holds its wave
            // for 60-180 cycles, which the SIMD's other wave fills with MFMAs here and cannot at the top of the tile
            if (t < S2_NPASS) {
                if (more) issue_x(nb, nh0, nw0, smem + (cur ^ 1) * S2_BUF, t, t + 1);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        // ---- epilogue: leaky, one bf16 rounding, pairing, 16-byte stores; lanes past the picture are dropped ----------------
        const int w = w0 + frow;
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const int h = h0 + 2 * wm + r;
            f32x4 x = acc[0][r], y = acc[1][r];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                x[e] = leaky_max(x[e], slope);
                y[e] = leaky_max(y[e], slope);
            }
            const unsigned x0 = pack_bf16x2(x[0], x[1]), x1 = pack_bf16x2(x[2], x[3]);
            const unsigned y0 = pack_bf16x2(y[0], y[1]), y1 = pack_bf16x2(y[2], y[3]);
            const auto s0 = __builtin_amdgcn_permlane16_swap(x0, y0, false, false);
            const auto s1 = __builtin_amdgcn_permlane16_swap(x1, y1, false, false);
            u32x4 o;
            o[0] = s0[0]; o[1] = s1[0]; o[2] = s0[1]; o[3] = s1[1];
            const bool ok = h < p.Ho && w < p.Wo;
            const unsigned off = ok ? ((unsigned)((b * p.Ho + h) * p.Wo + w) * (unsigned)p.out_ld + (unsigned)(32 * wn + cb)) * 2u : OOB;
            __builtin_amdgcn_raw_buffer_store_b128(o, rs_out, off, 0, 0);
        }
        b = nb; h0 = nh0; w0 = nw0;
        cur ^= 1;
    }
}

long s2_tiles(const y3_conv_desc *d) {
    return (long)d->batch * ((d->out_h + S2_TH - 1) / S2_TH) * ((d->out_w + S2_TW - 1) / S2_TW);
}

}  // namespace

// Does this descriptor run on conv3x3s2_c64_kernel?  (conv_igemm.hip asks; same packed weights, same results.)
bool y3_conv_s2c64_ok(const y3_conv_desc *d, long out_bytes) {
    if (d->dtype != Y3_BF16 || d->ksize != 3 || d->stride != 2 || d->in_c != 64 || d->out_c != 128) return false;
    if (d->k_pad != S2_KPAD || d->n_pad != 128) return false;
    if (d->flags & (Y3_CONV_UPS2X | Y3_CONV_OUT_F32 | Y3_CONV_IN_F32 | Y3_CONV_POOL2 | Y3_CONV_RESIDUAL)) return false;
    if (d->in_h != 2 * d->out_h || d->in_w != 2 * d->out_w) return false;         // (even input: out = in / 2, pad top/left)
    if (out_bytes >= 0xffff0000L || (long)d->batch * d->in_h * d->in_w >= 0x7fffffffL) return false;   // (pixel indices are ints;
    if ((long)S2_PH * d->in_w * d->in_ld * 2 >= 0x7fffffffL) return false;                                  //  so are offsets in a patch)
    // worth it only when the launch fills the chip several times over (persistent workgroups, weights loaded once each)
    const long tiles = s2_tiles(d);
    return tiles >= 4L * y3_device_cus() && tiles < 0x7fffffffL;
}

int y3_conv_s2c64_launch(const y3_conv_desc *d, const void *in, const void *w, const float *bias, void *out,
                         const void *zeros, hipStream_t stream) {
    const long pixels = (long)d->batch * d->out_h * d->out_w;
    S2Args a;
    a.in = (const char *)in; a.w = (const char *)w; a.zeros = (const char *)zeros;
    a.bias = bias; a.out = (char *)out;
    a.H = d->in_h; a.W = d->in_w; a.Ho = d->out_h; a.Wo = d->out_w; a.in_ld = d->in_ld; a.out_ld = d->out_ld;
    a.tiles_x = (d->out_w + S2_TW - 1) / S2_TW;
    a.tiles_img = a.tiles_x * ((d->out_h + S2_TH - 1) / S2_TH);
    a.total_tiles = (int)s2_tiles(d);
    a.flags = d->flags & (Y3_CONV_LEAKY | Y3_CONV_REVERSE);
    a.out_bytes = (unsigned)(pixels * d->out_ld * 2);
    a.fd_img = make_fastdiv(a.tiles_img);
    a.fd_tx = make_fastdiv(a.tiles_x);
    constexpr int lds = 2 * S2_BUF;
    static thread_local int attr_dev = -1;
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (attr_dev != dev) {
        if (hipFuncSetAttribute((const void *)conv3x3s2_c64_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess) {
            y3_set_error("conv_s2c64: cannot reserve %d bytes of LDS", lds);
            return Y3_E_LAUNCH;
        }
        attr_dev = dev;
    }
    const int cus = y3_device_cus();
    const int grid = a.total_tiles < cus ? a.total_tiles : cus;
    hipLaunchKernelGGL(conv3x3s2_c64_kernel, dim3((unsigned)grid), dim3(S2_NT), lds, stream, a);
    return y3_check_launch("conv3x3s2_c64");
}
