// 1x1 convolutions whose whole weight matrix fits the workgroup's registers, for gfx950 (MI355X), bf16: the bottleneck
// convolutions of Darknet-53's residual stages (reference build_model.py:113-179; yolov3.cfg: 512 -> 256 on the 38 x 38 map,
// 256 -> 128 on 76 x 76, ten launches each at a 608 input; also 384 -> 128 after the 76 x 76 route).
//
// As an implicit GEMM (conv_igemm.hip) a 1x1 stages its weights through LDS again for every tile: 256 KB of weights for
// 192 pixels x 1 KB of activations at 512 -> 256, so 57 % of what a CU pulls from L2 is weights it already had, and a
// CU's L2 -> LDS path (~14 B/clk) is what bounds the launch.  Here
//   * the WEIGHTS live in registers for the whole launch: 8 waves x up to 128 registers = 256 KB; a wave owns 32 output
//     channels (two 16-channel blocks) and every k-block of them;
//   * a persistent workgroup walks tiles of 64 consecutive pixels (a 1x1 has no 2-D structure); the tile's activations
//     go to LDS once by LDS-DMA, rows Cin * 2 + 32 bytes apart (fragment reads of 16 consecutive pixels conflict-free),
//     double-buffered, the next tile's loader passes issued among this tile's k-blocks;
//   * per k-block a wave reads its pixel blocks' fragments and issues two MFMAs per fragment.
// Every output is the same chain of MFMAs as in the implicit-GEMM kernels (bias as the initial accumulator value, k-blocks
// of 32 channels in order, leaky, one bf16 rounding), so the results are bit-identical to them (tests/test_gpu_conv.py);
// y3_conv2d_fused picks this kernel by shape, nothing changes at the ABI.
#include "common.h"

namespace {

constexpr int W1_TP = 64;                          // pixels per tile
constexpr int W1_NT = 512;

struct W1Args {
    const char *in, *w, *zeros;
    const float *bias;
    char *out;
    int M, in_ld, out_ld, total_tiles, flags, k_pad;
    unsigned out_bytes;
};

// CIN input channels, COUT output channels; WN = COUT / 32 waves along the channels, WM = 8 / WN along the pixels
template <int CIN, int COUT>
__global__ __launch_bounds__(W1_NT) __attribute__((amdgpu_waves_per_eu(2, 2)))
void conv1x1_regw_kernel(const W1Args p) {
    constexpr int WN = COUT / 32, WM = 8 / WN;
    constexpr int NKB = CIN / 32;                  // k-blocks
    constexpr int MI = W1_TP / 16 / WM;            // 16-pixel blocks per wave
    constexpr int PITCH = CIN * 2 + 32;            // bytes between pixels in LDS
    constexpr int SPP = PITCH / 16;                // 16-byte slots per pixel (the last two are padding)
    constexpr int NPASS = (W1_TP * SPP + W1_NT - 1) / W1_NT;
    constexpr int BUF = NPASS * W1_NT * 16;
    static_assert(WN * WM == 8 && MI >= 1 && NKB * 2 * 4 <= 128, "wave grid / weight registers");
    static_assert(SPP % 4 == 2, "pixel pitch: 16 consecutive pixels on 16 different slots for ds_read_b128's lane groups");
    static_assert(2 * BUF <= 160 * 1024, "LDS budget");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int frow = lane & 15, fq = lane >> 4;
    const int wn = wave % WN, wm = wave / WN;      // this wave: channels 32 wn .. 32 wn + 31 of pixel blocks MI wm .. MI wm + MI - 1
    const float slope = (p.flags & Y3_CONV_LEAKY) ? 0.1f : 1.0f;

    // ---- the wave's 32 rows of the weight matrix as MFMA "A" fragments: [k-block][16-channel block] ---------------------
    bf16x8 wf[NKB][2];
#pragma unroll
    for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
            wf[kb][ni] = *(const bf16x8 *)(p.w + ((long)(32 * wn + 16 * ni + frow) * p.k_pad + kb * 32 + fq * 8) * 2);
    f32x4 bias4[2];
#pragma unroll
    for (int ni = 0; ni < 2; ++ni) bias4[ni] = *(const f32x4 *)(p.bias + 32 * wn + 16 * ni + fq * 4);

    // the tile's 64 x CIN activations -> LDS: slot s = SPP * pixel + chunk, the last two chunks of a pixel are padding.
    // Per-lane offsets once per launch; pixels past the end of the tensor (last tile) read the zero page.
    int rel[NPASS], rpx[NPASS];
#pragma unroll
    for (int i = 0; i < NPASS; ++i) {
        const int s = tid + W1_NT * i;
        const int px = s / SPP, c = s - px * SPP;
        rel[i] = (px < W1_TP && c < SPP - 2) ? px * p.in_ld * 2 + c * 16 : -1;
        rpx[i] = px;
    }
    auto first_pixel = [&](int tile) { return xcd_remap(tile, p.total_tiles, p.flags & Y3_CONV_REVERSE) * W1_TP; };
    auto issue_x = [&](int tile, char *buf, int i0, int i1) {
        const int pix0 = first_pixel(tile);
        const char *base = p.in + (long)pix0 * p.in_ld * 2;
        if (pix0 + W1_TP <= p.M) {                 // (wave-uniform)
#pragma unroll
            for (int i = i0; i < i1; ++i)
                if (rel[i] >= 0) glds16(base + rel[i], buf + i * (W1_NT * 16) + wave * 1024);
        } else {
#pragma unroll
            for (int i = i0; i < i1; ++i)
                if (rel[i] >= 0) glds16(pix0 + rpx[i] < p.M ? base + rel[i] : p.zeros, buf + i * (W1_NT * 16) + wave * 1024);
        }
    };
    const int cb = (fq & 1) ? 16 + (fq - 1) * 4 : fq * 4;      // after the swap pairing: this lane's first channel of the 32
    constexpr unsigned OOB = 0xffffff00u;
    const auto rs_out = __builtin_amdgcn_make_buffer_rsrc((void *)p.out, 0, p.out_bytes, 0x00020000);

    int tile = blockIdx.x;
    if (tile >= p.total_tiles) return;
    issue_x(tile, smem, 0, NPASS);
    int cur = 0;
    bool first = true;
    for (; tile < p.total_tiles; tile += gridDim.x) {
        // This wave's pieces of the tile have landed: they are older than the previous tile's MI stores per lane -- the
        // youngest operations -- which may stay in flight.
        if (first) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(MI) : "memory");
        first = false;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();              // ... in every wave; nobody reads the other buffer any more
        asm volatile("" ::: "memory");
        const int next = tile + (int)gridDim.x;
        const bool more = next < p.total_tiles;
        const int pix_t = first_pixel(tile);
        int xoff = (MI * wm * 16 + frow) * PITCH + fq * 16;    // (opaque per tile: one address register + immediates)
        asm volatile("" : "+v"(xoff));
        const char *xl = smem + cur * BUF + xoff;
        f32x4 acc[2][MI];
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
#pragma unroll
            for (int mi = 0; mi < MI; ++mi) acc[ni][mi] = bias4[ni];
        // k-blocks in pairs: the fragments of k-block kb + 2 are requested right after k-block kb's MFMAs have been issued
        bf16x8 xf[2][MI];
        auto rd = [&](int kb) {
#pragma unroll
            for (int mi = 0; mi < MI; ++mi) xf[kb & 1][mi] = *(const bf16x8 *)(xl + mi * 16 * PITCH + kb * 64);
        };
        rd(0);
        rd(1);
#pragma unroll
        for (int kb = 0; kb < NKB; ++kb) {
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                for (int ni = 0; ni < 2; ++ni)
                    acc[ni][mi] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[kb][ni], xf[kb & 1][mi], acc[ni][mi], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            if (kb + 2 < NKB) rd(kb + 2);
            // one loader pass of the next tile after each of the first k-blocks (an LDS-DMA instruction holds its wave for
            // 60-180 cycles, which the SIMD's other wave fills with MFMAs here and cannot at the top of the tile)
            if (kb < NPASS) {
                __builtin_amdgcn_sched_barrier(0);
                if (more) issue_x(next, smem + (cur ^ 1) * BUF, kb, kb + 1);
            }
        }
        static_assert(NPASS <= NKB, "one loader pass per k-block");
        // ---- epilogue: leaky, one bf16 rounding, pairing, 16-byte stores; pixels past the end are dropped -------------------
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) {
            const int pix = pix_t + (MI * wm + mi) * 16 + frow;
            f32x4 x = acc[0][mi], y = acc[1][mi];
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
            const unsigned off = pix < p.M ? ((unsigned)pix * (unsigned)p.out_ld + (unsigned)(32 * wn + cb)) * 2u : OOB;
            __builtin_amdgcn_raw_buffer_store_b128(o, rs_out, off, 0, 0);
        }
        cur ^= 1;
    }
}

template <int CIN, int COUT>
int launch_1x1w(const W1Args &a, hipStream_t stream) {
    constexpr int PITCH = CIN * 2 + 32, SPP = PITCH / 16;
    constexpr int NPASS = (W1_TP * SPP + W1_NT - 1) / W1_NT;
    constexpr int lds = 2 * NPASS * W1_NT * 16;
    static thread_local int attr_dev = -1;
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (attr_dev != dev) {
        if (hipFuncSetAttribute((const void *)conv1x1_regw_kernel<CIN, COUT>, hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess) {
            y3_set_error("conv_1x1w: cannot reserve %d bytes of LDS", lds);
            return Y3_E_LAUNCH;
        }
        attr_dev = dev;
    }
    const int cus = y3_device_cus();
    const int grid = a.total_tiles < cus ? a.total_tiles : cus;
    hipLaunchKernelGGL((conv1x1_regw_kernel<CIN, COUT>), dim3((unsigned)grid), dim3(W1_NT), lds, stream, a);
    return y3_check_launch("conv1x1_regw");
}

}  // namespace

// Does this descriptor run on conv1x1_regw_kernel?  (conv_igemm.hip asks; same packed weights, same results.)
bool y3_conv_1x1w_ok(const y3_conv_desc *d, long out_bytes) {
#ifdef Y3_NO_1X1W                                  // (A/B builds)
    return false;
#endif
    if (d->dtype != Y3_BF16 || d->ksize != 1 || d->stride != 1) return false;
    const bool s512 = d->in_c == 512 && d->out_c == 256, s256 = (d->in_c == 256 || d->in_c == 384) && d->out_c == 128;
    if (!s512 && !s256) return false;
    if (d->k_pad != d->in_c || d->n_pad != d->out_c) return false;
    if (d->flags & (Y3_CONV_UPS2X | Y3_CONV_OUT_F32 | Y3_CONV_IN_F32 | Y3_CONV_POOL2 | Y3_CONV_RESIDUAL)) return false;
    if (d->out_h != d->in_h || d->out_w != d->in_w) return false;
    const long M = (long)d->batch * d->out_h * d->out_w;
    if (out_bytes >= 0xffff0000L || M >= 0x7fffffffL - W1_TP || (long)W1_TP * d->in_ld * 2 >= 0x7fffffffL) return false;
    // worth it only when the launch fills the chip several times over (persistent workgroups, weights loaded once each)
    return (M + W1_TP - 1) / W1_TP >= 4L * y3_device_cus();
}

int y3_conv_1x1w_launch(const y3_conv_desc *d, const void *in, const void *w, const float *bias, void *out,
                        const void *zeros, hipStream_t stream) {
    const long M = (long)d->batch * d->out_h * d->out_w;
    W1Args a;
    a.in = (const char *)in; a.w = (const char *)w; a.zeros = (const char *)zeros;
    a.bias = bias; a.out = (char *)out;
    a.M = (int)M; a.in_ld = d->in_ld; a.out_ld = d->out_ld; a.k_pad = d->k_pad;
    a.total_tiles = (int)((M + W1_TP - 1) / W1_TP);
    a.flags = d->flags & (Y3_CONV_LEAKY | Y3_CONV_REVERSE);
    a.out_bytes = (unsigned)(M * d->out_ld * 2);
    if (d->in_c == 512) return launch_1x1w<512, 256>(a, stream);
    if (d->in_c == 384) return launch_1x1w<384, 128>(a, stream);       // (yolov3.cfg layer 99: the 76 x 76 route)
    return launch_1x1w<256, 128>(a, stream);
}
