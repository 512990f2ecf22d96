// 3x3 stride-1 convolution with 32 input and 64 output channels for gfx950 (MI355X), bf16:
// Darknet-53's layer 3 (reference build_model.py:113-179 + the shortcut of :218-227 in the epilogue), 5.9 M output pixels
// per 64-image batch at 608 x 608 and 4 % of the network's FLOPs.
//
// As an implicit GEMM (conv_igemm.hip) this layer is bound by what a CU can stage from L2 into LDS: every input pixel
// (64 bytes) is gathered nine times, once per tap, 205 KB of LDS-DMA per 256-pixel tile for 16 KB of input.  Here
//   * the WEIGHTS live in registers for the whole launch: a wave computes 32 output channels of its 64 pixels, so its
//     MFMA "A" fragments are half the 64 x 288 matrix = 18 fragments = 72 registers, loaded once;
//   * a persistent workgroup walks 16 x 16-pixel tiles; a tile's input patch (18 x 18 pixels x 32 channels, zero outside
//     the picture = the padding) goes to LDS ONCE by LDS-DMA, 96 bytes per pixel (64 + 32 of pad: the fragment reads of
//     16 consecutive pixels then hit 16 different 16-byte bank groups), double-buffered: the next tile's patch and its
//     shortcut operand are requested before this tile's MFMAs;
//   * a tap is a constant displacement in that patch: per tap and 16-pixel row block one ds_read_b128 and four MFMAs.
// One barrier per tile, no weight traffic, 20 KB of input per tile instead of 205 KB.
// Every output is the same chain of MFMAs as in conv_igemm_kernel (bias as the initial accumulator value, k-blocks of
// 32 = taps in order, leaky, + shortcut in fp32, one bf16 rounding), so the results are bit-identical to it
// (tests/test_gpu_conv.py); y3_conv2d_fused picks this kernel by shape, nothing changes at the ABI.
#include "common.h"

namespace {

constexpr int C32_T = 16;                          // tile edge in pixels
constexpr int C32_P = C32_T + 2;                   // patch edge (tile + halo)
constexpr int C32_NPOS = C32_P * C32_P;            // 324 patch pixels
constexpr int C32_PITCH = 96;                      // bytes between patch pixels in LDS
constexpr int C32_SLOTS = C32_NPOS * 6;            // 16-byte slots of a patch (4 data + 2 pad per pixel)
constexpr int C32_BUF = (C32_SLOTS + 511) / 512 * 512 * 16;   // bytes per patch buffer (whole loader passes)
constexpr int C32_NT = 512;
constexpr int C32_KPAD = 320;                      // packed weight row: 9 * 32 = 288, padded to a multiple of 64

struct C32Args {
    const char *in, *w, *res, *zeros;
    const float *bias;
    char *out;
    int H, W, in_ld, out_ld, res_ld;
    int tiles_x, tiles_img, total_tiles, flags;
    unsigned out_bytes, res_bytes;
    FastDiv fd_img, fd_tx;
};

__global__ __launch_bounds__(C32_NT) __attribute__((amdgpu_waves_per_eu(2, 2)))
void conv3x3_c32_kernel(const C32Args p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int frow = lane & 15, fq = lane >> 4;
    const int wm = wave >> 1, wn = wave & 1;       // this wave: tile rows 4 wm .. 4 wm + 3, channels 32 wn .. 32 wn + 31
    const bool has_res = p.flags & Y3_CONV_RESIDUAL;
    const float slope = (p.flags & Y3_CONV_LEAKY) ? 0.1f : 1.0f;

    // ---- the whole weight matrix as MFMA "A" fragments: [tap][16-channel block], 16 bytes per lane each --------------
    bf16x8 wf[9][2];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
            wf[t][ni] = *(const bf16x8 *)(p.w + ((wn * 32 + ni * 16 + frow) * C32_KPAD + t * 32 + fq * 8) * 2);
    f32x4 bias4[2];
#pragma unroll
    for (int ni = 0; ni < 2; ++ni) bias4[ni] = *(const f32x4 *)(p.bias + wn * 32 + ni * 16 + fq * 4);

    auto tile_coords = [&](int tile, int &b, int &h0, int &w0) {
        const int t = xcd_remap(tile, p.total_tiles, p.flags & Y3_CONV_REVERSE);
        b = fast_div(t, p.fd_img);
        const int ti = t - b * p.tiles_img;
        const int ty = fast_div(ti, p.fd_tx);
        h0 = ty * C32_T;
        w0 = (ti - ty * p.tiles_x) * C32_T;
    };
    // a tile's 18 x 18 x 32 input patch -> LDS: slot s = 6 * pixel + chunk, chunks 4 and 5 are padding (not written)
    auto issue_x = [&](int b, int h0, int w0, char *buf, int i0 = 0, int i1 = C32_BUF / (C32_NT * 16)) {
#pragma unroll
        for (int i = i0; i < i1; ++i) {
            const int s = tid + C32_NT * i;
            const int pos = (s * 10923) >> 16, c = s - pos * 6;                   // s / 6, exact below 2048
            const int pr = (pos * 3641) >> 16, pc = pos - pr * C32_P;             // pos / 18, exact below 512
            const int h = h0 - 1 + pr, w = w0 - 1 + pc;
            const bool inside = (unsigned)h < (unsigned)p.H && (unsigned)w < (unsigned)p.W;
            const char *src = inside ? p.in + ((long)((b * p.H + h) * p.W + w) * p.in_ld * 2 + c * 16) : p.zeros;
            if (pos < C32_NPOS && c < 4) glds16(src, buf + i * (C32_NT * 16) + wave * 1024);
        }
    };
    // epilogue geometry: after the permlane16_swap pairing a lane owns 8 consecutive channels of one pixel
    const int cb = (fq & 1) ? 16 + (fq - 1) * 4 : fq * 4;
    constexpr unsigned OOB = 0xffffff00u;
    const auto rs_out = __builtin_amdgcn_make_buffer_rsrc((void *)p.out, 0, p.out_bytes, 0x00020000);
    const auto rs_res = __builtin_amdgcn_make_buffer_rsrc((void *)(has_res ? p.res : p.zeros), 0, has_res ? p.res_bytes : 0u, 0x00020000);
    // shortcut operand of this wave's four tile rows (its 32 channels = one block pair): one 16-byte load per row block
    auto res_load = [&](u32x4 (&rv)[4], int b, int h0, int w0) {
#pragma unroll
        for (int mi = 0; mi < 4; ++mi) {
            const int h = h0 + 4 * wm + mi;
            const unsigned pix = (unsigned)((b * p.H + h) * p.W + w0 + frow);
            rv[mi] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(
                rs_res, h < p.H ? (pix * (unsigned)p.res_ld + (unsigned)(wn * 32 + cb)) * 2u : OOB, 0, 0));
        }
    };

    int tile = blockIdx.x;
    if (tile >= p.total_tiles) return;
    int b, h0, w0;
    tile_coords(tile, b, h0, w0);
    u32x4 rva[4], rvb[4];
#pragma unroll
    for (int mi = 0; mi < 4; ++mi) rva[mi] = rvb[mi] = u32x4{0u, 0u, 0u, 0u};
    issue_x(b, h0, w0, smem);
    if (has_res) res_load(rva, b, h0, w0);
    int cur = 0;
    bool first = true;
    // one tile: rv = its shortcut operand (requested a tile ago), rvn receives the next tile's
    auto do_tile = [&](u32x4 (&rv)[4], u32x4 (&rvn)[4]) {
        // This wave's pieces of the tile's patch and the tile's shortcut operand have landed.  They are older than the
        // previous tile's stores -- exactly four per lane, the youngest operations -- which may stay in flight.
        if (first) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        first = false;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();              // ... in every wave; nobody reads the other buffer any more
        asm volatile("" ::: "memory");             // (not __syncthreads: it would also wait for the stores)
        const char *xb = smem + cur * C32_BUF;
        const int next = tile + (int)gridDim.x;
        int nb = 0, nh0 = 0, nw0 = 0;
        const bool more = next < p.total_tiles;
        if (more) {                                // the next tile's patch and shortcut operand travel under this tile's MFMAs
            tile_coords(next, nb, nh0, nw0);
            issue_x(nb, nh0, nw0, smem + (cur ^ 1) * C32_BUF);
            if (has_res) res_load(rvn, nb, nh0, nw0);
        }
        f32x4 acc[2][4];
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
#pragma unroll
            for (int mi = 0; mi < 4; ++mi) acc[ni][mi] = bias4[ni];
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const int dh = t / 3, dw = t - 3 * dh;
#pragma unroll
            for (int mi = 0; mi < 4; ++mi) {
                const bf16x8 xf = *(const bf16x8 *)(xb + ((4 * wm + mi + dh) * C32_P + dw + frow) * C32_PITCH + fq * 16);
#pragma unroll
                for (int ni = 0; ni < 2; ++ni)
                    acc[ni][mi] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[t][ni], xf, acc[ni][mi], 0, 0, 0);
            }
            // (measured and not kept: one loader pass after each of the first taps instead of all at the top of the tile --
            // what pays in resblock64_kernel made this kernel slower)
        }
        // ---- epilogue: leaky, + shortcut (fp32), one bf16 rounding, 16-byte stores; lanes past the picture are dropped ----
#pragma unroll
        for (int mi = 0; mi < 4; ++mi) {
            const int h = h0 + 4 * wm + mi;
            const unsigned pix = (unsigned)((b * p.H + h) * p.W + w0 + frow);
            {
                f32x4 x = acc[0][mi], y = acc[1][mi];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    x[j] = leaky_max(x[j], slope);
                    y[j] = leaky_max(y[j], slope);
                }
                u32x4 o;
                if (has_res) {
                    float lo[4], hi[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const auto sw = __builtin_amdgcn_permlane16_swap(__float_as_uint(x[j]), __float_as_uint(y[j]), false, false);
                        lo[j] = __uint_as_float(sw[0]);
                        hi[j] = __uint_as_float(sw[1]);
                    }
                    const u32x4 r = rv[mi];
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        lo[2 * j] += bf16_to_f32(r[j] & 0xffff);
                        lo[2 * j + 1] += __uint_as_float(r[j] & 0xffff0000u);
                        hi[2 * j] += bf16_to_f32(r[2 + j] & 0xffff);
                        hi[2 * j + 1] += __uint_as_float(r[2 + j] & 0xffff0000u);
                    }
                    o[0] = pack_bf16x2(lo[0], lo[1]); o[1] = pack_bf16x2(lo[2], lo[3]);
                    o[2] = pack_bf16x2(hi[0], hi[1]); o[3] = pack_bf16x2(hi[2], hi[3]);
                } else {
                    const unsigned x0 = pack_bf16x2(x[0], x[1]), x1 = pack_bf16x2(x[2], x[3]);
                    const unsigned y0 = pack_bf16x2(y[0], y[1]), y1 = pack_bf16x2(y[2], y[3]);
                    const auto s0 = __builtin_amdgcn_permlane16_swap(x0, y0, false, false);
                    const auto s1 = __builtin_amdgcn_permlane16_swap(x1, y1, false, false);
                    o[0] = s0[0]; o[1] = s1[0]; o[2] = s0[1]; o[3] = s1[1];
                }
                __builtin_amdgcn_raw_buffer_store_b128(o, rs_out, h < p.H ? (pix * (unsigned)p.out_ld + (unsigned)(wn * 32 + cb)) * 2u : OOB, 0, 0);
            }
        }
        b = nb; h0 = nh0; w0 = nw0;
        cur ^= 1;
        tile = next;
    };
    while (true) {                                 // (two copies of the tile body: the shortcut registers alternate roles)
        do_tile(rva, rvb);
        if (tile >= p.total_tiles) break;
        do_tile(rvb, rva);
        if (tile >= p.total_tiles) break;
    }
}

}  // namespace

// Does this descriptor run on conv3x3_c32_kernel?  (conv_igemm.hip asks; same packed weights, same results.)
bool y3_conv_c32_ok(const y3_conv_desc *d, long out_bytes, long res_bytes) {
    if (d->dtype != Y3_BF16 || d->ksize != 3 || d->stride != 1 || d->in_c != 32 || d->out_c != 64) return false;
    if (d->k_pad != C32_KPAD || d->n_pad != 64) return false;
    if (d->flags & (Y3_CONV_UPS2X | Y3_CONV_OUT_F32 | Y3_CONV_IN_F32 | Y3_CONV_POOL2)) return false;
    if (d->out_w % C32_T != 0 || d->out_h != d->in_h || d->out_w != d->in_w) return false;
    if (out_bytes >= 0xffff0000L || res_bytes >= 0xffff0000L) return false;
    // worth it only when the launch fills the chip several times over (persistent workgroups, weights loaded once each)
    const long tiles = (long)d->batch * ((d->out_h + C32_T - 1) / C32_T) * (d->out_w / C32_T);
    return tiles >= 4L * y3_device_cus() && tiles < 0x7fffffffL;
}

int y3_conv_c32_launch(const y3_conv_desc *d, const void *in, const void *w, const float *bias, const void *residual,
                       void *out, const void *zeros, hipStream_t stream) {
    const long pixels = (long)d->batch * d->out_h * d->out_w;
    Y3_REQUIRE((long)pixels * d->in_ld * 2 < 0x7fffffffffffL, Y3_E_INVALID, "conv_c32: input too large");
    C32Args a;
    a.in = (const char *)in; a.w = (const char *)w; a.res = (const char *)residual; a.zeros = (const char *)zeros;
    a.bias = bias; a.out = (char *)out;
    a.H = d->out_h; a.W = d->out_w; a.in_ld = d->in_ld; a.out_ld = d->out_ld; a.res_ld = d->res_ld;
    a.tiles_x = d->out_w / C32_T;
    a.tiles_img = a.tiles_x * ((d->out_h + C32_T - 1) / C32_T);
    a.total_tiles = (int)((long)d->batch * a.tiles_img);
    a.flags = d->flags & (Y3_CONV_LEAKY | Y3_CONV_RESIDUAL | Y3_CONV_REVERSE);
    a.out_bytes = (unsigned)(pixels * d->out_ld * 2);
    a.res_bytes = (d->flags & Y3_CONV_RESIDUAL) ? (unsigned)(pixels * d->res_ld * 2) : 0u;
    a.fd_img = make_fastdiv(a.tiles_img);
    a.fd_tx = make_fastdiv(a.tiles_x);
    constexpr int lds = 2 * C32_BUF;
    static thread_local int attr_dev = -1;
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (attr_dev != dev) {
        if (hipFuncSetAttribute((const void *)conv3x3_c32_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess) {
            y3_set_error("conv_c32: cannot reserve %d bytes of LDS", lds);
            return Y3_E_LAUNCH;
        }
        attr_dev = dev;
    }
    const int cus = y3_device_cus();
    const int grid = a.total_tiles < cus ? a.total_tiles : cus;
    hipLaunchKernelGGL(conv3x3_c32_kernel, dim3((unsigned)grid), dim3(C32_NT), lds, stream, a);
    return y3_check_launch("conv3x3_c32");
}
