// Fused Darknet residual block for gfx950 (MI355X), bf16:
//
//     out = x + leaky(conv3x3(leaky(conv1x1(x; W1) + b1); W2) + b2)
//
// i.e. reference build_model.py:113-179 applied to a [convolutional 1x1 C -> C/2] and the
// [convolutional 3x3 C/2 -> C] after it, plus the [shortcut from=-3] of :218-227, in ONE launch.
// Un-fused (conv_igemm.hip), the C/2-channel tensor between the two convolutions is written to HBM and read
// back through an LDS-DMA gather nine times over (once per tap), and the block's input is read twice (by the
// 1x1 and as the shortcut operand).  Here a workgroup owns a 2-D tile of 19 x 13 (or 19 x 12) output pixels:
//
//   phase 1  the 1x1 convolution on the tile plus its one-pixel halo ((13+2) x (19+2) = 315 positions, run as
//            320 GEMM rows): x and W1 staged by LDS-DMA exactly like conv_igemm_kernel (two buffers, one barrier per
//            64-deep k-step, XOR-swizzled 128-byte rows), MFMA 16x16x32, bias as the accumulators' initial value,
//            leaky-ReLU, ONE bf16 rounding -- the same operations in the same order as the un-fused 1x1 launch --
//            and the result goes to an LDS image y[position][C/2] (zero outside the picture = the 3x3's padding).
//   phase 2  the 3x3 convolution as an implicit GEMM whose activation operand never leaves LDS: the MFMA "B"
//            fragment of (pixel, tap, cin-block) is one ds_read_b128 of y at  base(pixel) + tap offset +
//            cin offset,  all taps being constant displacements in the pitched y image.  Only W2 is streamed
//            (LDS-DMA, two buffers): half the LDS-DMA instructions per MFMA of the un-fused kernel, no im2col
//            addressing, no per-tap masks.  y rows are C/2*2 + 32 bytes apart, which makes the fragment reads of
//            16 consecutive positions bank-conflict free (slot = (2 * position + chunk) mod 16).
//   epilogue as conv_igemm_kernel's register epilogue: leaky-ReLU, permlane16_swap pairing, the shortcut operand
//            (the tile's own x pixels, still in L2 / Infinity Cache from phase 1) added in fp32 before the single
//            bf16 rounding, 16-byte buffer stores.
//
// Every sum runs over k in the order of the two un-fused launches, so the result is BIT-IDENTICAL to them
// (tests/test_gpu_resblock.py); the planner may therefore fuse at large batch and not at batch 1 without breaking
// batch invariance.  GEMM rows of a tile: 13 "row blocks" (16 of a tile row's 19 pixels: consecutive positions, the
// conflict-free case) and 3 "leftover blocks" that collect columns 16..18 of all rows.
// 76 x 76 x 64 images = 1536 tiles = exactly 6 rounds of 256 workgroups (the un-fused 256-pixel tiling has 5.64).
#include "common.h"
#include <type_traits>

namespace {

constexpr int RB_TW = 19;                          // tile width in pixels (608/8 = 76 = 4 * 19, 152 = 8 * 19, 304 = 16 * 19)
constexpr int RB_TH = 13;                          // tile height, at most
constexpr int RB_PW = RB_TW + 2;                   // pitch of the y image in positions (tile row + halo)
constexpr int RB_NPOS = (RB_TH + 2) * RB_PW;       // 315 positions
constexpr int RB_M1 = 320;                         // ... run as 20 MFMA row blocks

// Workgroup barrier that orders LDS traffic only: __syncthreads() also waits for every outstanding global store
// (vmcnt(0)), which here would drain the previous tile's 16 result stores per lane at every barrier.
__device__ __forceinline__ void lds_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

struct ResArgs {
    const char *x, *w1, *w2, *zeros;
    const float *b1, *b2;
    char *out;
    int H, W, x_ld, out_ld;
    int tiles_x, tiles_img, th_base, th_rem;       // column tiles, tiles per image, rows per row tile (+1 for the first th_rem)
    int total_tiles, flags;                        // flags: 1 leaky after the 1x1, 2 leaky after the 3x3, 32 reverse tile order
    unsigned x_bytes, out_bytes;
    FastDiv fd_img, fd_tx;
};

template <int CM, int WM, int WN>
__global__ __launch_bounds__(64 * WM * WN) __attribute__((amdgpu_waves_per_eu(2, 2)))
void resblock_kernel(const ResArgs p) {
    constexpr int C = 2 * CM;
    constexpr int NT = 64 * WM * WN;
    constexpr int LROWS = NT / 8;                  // rows one loader pass covers
    constexpr int PIECE = NT * 16;                 // bytes one loader pass writes
    // phase 1: [320 positions] x [CM channels], k = C
    constexpr int TM1 = RB_M1 / WM, TN1 = CM / WN, MI1 = TM1 / 16, NI1 = TN1 / 16;
    constexpr int A1_IT = RB_M1 / LROWS;
    constexpr int NK1 = C / 64;                    // 64-deep k-steps of phase 1
    constexpr int XSTG = RB_M1 * 128;              // one k-step of x in LDS: 320 positions x 64 channels
    // phase 2: [256 pixel slots] x [C channels], k = 9 * CM
    constexpr int TM2 = 256 / WM, TN2 = C / WN, MI = TM2 / 16, NI = TN2 / 16, NP2 = NI / 2;
    constexpr int B2_IT = C / LROWS;
    constexpr int W2_STAGE = C * 128;
    constexpr int YS = CM * 2 + 32;                // bytes between positions of the y image
    constexpr int Y_BYTES = RB_M1 * YS;
    constexpr int W2_ODD = Y_BYTES;                // W2 buffer of odd k-steps
    constexpr int W2_EVEN = Y_BYTES + W2_STAGE;    // ... of even k-steps: inside (or beyond) the slot of x's FIRST k-step
    constexpr int KPT = CM / 64;                   // k-steps per tap
    constexpr int NK2 = 9 * KPT;
    constexpr int K2 = 9 * CM;
    static_assert(NT == 512 && LROWS == 64, "eight waves");
    static_assert(TM1 % 16 == 0 && TN1 % 16 == 0 && TM2 % 16 == 0 && TN2 % 32 == 0, "wave tiles");
    static_assert(RB_M1 % LROWS == 0 && C % LROWS == 0 && NK1 >= 2, "loader passes");
    static_assert(W2_EVEN >= (NK1 - 1) * XSTG, "W2's first buffer overlays x k-steps that are still unread");
    static_assert(NK1 * XSTG <= 160 * 1024 && W2_EVEN + W2_STAGE <= 160 * 1024, "LDS budget");

    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int frow = lane & 15, fq = lane >> 4, fsw = (lane >> 1) & 7;
    const int lrow = tid >> 3;
    const int kc16 = ((tid & 7) ^ ((tid >> 4) & 7)) * 16;        // source byte offset of this thread's LDS chunk slot
    const float slope1 = (p.flags & 1) ? 0.1f : 1.0f, slope2 = (p.flags & 2) ? 0.1f : 1.0f;
    using frag_t = bf16x8;

    // tile -> image, first row / column, rows
    auto tile_coords = [&](int tile, int &b, int &h0, int &w0, int &th) {
        const int t = xcd_remap(tile, p.total_tiles, p.flags & 32);
        b = fast_div(t, p.fd_img);
        const int ti = t - b * p.tiles_img;
        const int ty = fast_div(ti, p.fd_tx), tx = ti - ty * p.tiles_x;
        th = p.th_base + (ty < p.th_rem ? 1 : 0);
        h0 = ty < p.th_rem ? ty * (p.th_base + 1) : p.th_rem * (p.th_base + 1) + (ty - p.th_rem) * p.th_base;
        w0 = tx * RB_TW;
    };
    // x k-step kt lives in slot kt - 1, the FIRST k-step in the last slot: that slot is free again after one barrier and
    // takes W2's first k-step while phase 1 is still computing
    auto xslot = [](int kt) { return (kt == 0 ? NK1 - 1 : kt - 1) * XSTG; };
    // ALL of a tile's x (tile + halo, every channel) by LDS-DMA: NK1 * A1_IT pieces per thread, no register staging.
    // Positions outside the picture (or the tile's halo) read the zero page.
    auto issue_x = [&](int b, int h0, int w0, int th) {
#pragma unroll
        for (int i = 0; i < A1_IT; ++i) {
            const int pos = lrow + LROWS * i;
            const int pr = (pos * 3121) >> 16, pc = pos - pr * RB_PW;            // pos / 21, exact below 320
            const int h = h0 - 1 + pr, w = w0 - 1 + pc;
            const bool ok = pos < RB_NPOS && pr < th + 2 && (unsigned)h < (unsigned)p.H && (unsigned)w < (unsigned)p.W;
            const char *src = ok ? p.x + ((unsigned)((b * p.H + h) * p.W + w) * (unsigned)(p.x_ld * 2) + (unsigned)kc16) : p.zeros;
            const int step = ok ? 128 : 0;
#pragma unroll
            for (int kt = 0; kt < NK1; ++kt) glds16(src + kt * step, smem + xslot(kt) + i * PIECE + wave * 1024);
        }
    };
    auto piece2 = [&](int i, int kt) {             // W2 rows of k-step kt -> its buffer
        char *dst = smem + ((kt & 1) ? W2_ODD : W2_EVEN) + wave * 1024;
        glds16(p.w2 + ((long)(lrow + LROWS * i) * (K2 * 2) + kc16 + kt * 128), dst + i * PIECE);
    };

    // W1 as MFMA "A" fragments, 16 bytes per lane and (k-block, channel block), straight from L2 to registers (64 KB,
    // L2-resident): LDS holds all of x during phase 1.  (Requested at the top of a tile.  Requested ahead of the tile's
    // x, at the end of the previous tile, they would not queue behind its 20 LDS-DMA pieces per lane -- but carried
    // across the loop's back edge the 64 registers are spilled by hipcc, as inline-asm loads too: measured slower.)
    frag_t w1f[2 * NK1][NI1];
    auto w1_load = [&]() {
#pragma unroll
        for (int kb = 0; kb < 2 * NK1; ++kb)
#pragma unroll
            for (int ni = 0; ni < NI1; ++ni)
                w1f[kb][ni] = *(const frag_t *)(p.w1 + ((wn * TN1 + ni * 16 + frow) * C + kb * 32 + fq * 8) * 2);
    };
    f32x4 bias1[NI1];                              // phase 1's initial accumulator values
    auto bias1_load = [&]() {
#pragma unroll
        for (int ni = 0; ni < NI1; ++ni) bias1[ni] = *(const f32x4 *)(p.b1 + wn * TN1 + ni * 16 + fq * 4);
    };

    int tile = blockIdx.x;
    int b = 0, h0 = 0, w0 = 0, th = 0;
    if (tile < p.total_tiles) {
        tile_coords(tile, b, h0, w0, th);
        issue_x(b, h0, w0, th);
    }
    for (; tile < p.total_tiles; tile += gridDim.x) {
        // ---- phase 1: y = leaky(W1 x + b1) on the tile + halo -------------------------------------------------------
        constexpr int NMF1 = NI1 * MI1;
        bias1_load();
        w1_load();
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // this tile's x (requested before the last tile's stores) and W1 have landed
        lds_barrier();                             // ... in every wave
        f32x4 acc1[NI1][MI1];
#pragma unroll
        for (int ni = 0; ni < NI1; ++ni)
#pragma unroll
            for (int mi = 0; mi < MI1; ++mi) acc1[ni][mi] = bias1[ni];
#pragma unroll
        for (int kt = 0; kt < NK1; ++kt) {
            if (kt == 1) lds_barrier();            // every wave has read x's first k-step: its slot now takes W2's first k-step
            const char *Xs = smem + xslot(kt);
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                const int so = ((ks * 4 + fq) ^ fsw) * 16;
                frag_t xf[MI1];
#pragma unroll
                for (int mi = 0; mi < MI1; ++mi) xf[mi] = *(const frag_t *)(Xs + (wm * TM1 + mi * 16 + frow) * 128 + so);
#pragma unroll
                for (int j = 0; j < NMF1; ++j) {
                    if (kt == 1 && ks == 0) {
#pragma unroll
                        for (int q = 0; q < B2_IT; ++q) {
                            if (j == (q * NMF1) / B2_IT) {
                                __builtin_amdgcn_sched_barrier(0);
                                piece2(q, 0);
                                __builtin_amdgcn_sched_barrier(0);
                            }
                        }
                    }
                    const int ni = j / MI1, mi = j % MI1;
                    
                        acc1[ni][mi] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w1f[2 * kt + ks][ni], xf[mi], acc1[ni][mi], 0, 0, 0);
                }
            }
        }
        lds_barrier();                             // every wave is done with x: the y image overlays it
#pragma unroll
        for (int mi = 0; mi < MI1; ++mi) {
            const int pos = wm * TM1 + mi * 16 + frow;
            const int pr = (pos * 3121) >> 16, pc = pos - pr * RB_PW;
            const int h = h0 - 1 + pr, w = w0 - 1 + pc;
            const bool in = (unsigned)h < (unsigned)p.H && (unsigned)w < (unsigned)p.W;        // outside: the 3x3's zero padding
#pragma unroll
            for (int ni = 0; ni < NI1; ++ni) {
                f32x4 v = acc1[ni][mi];
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] = leaky_max(v[j], slope1);
                uint2 o;
                o.x = in ? pack_bf16x2(v[0], v[1]) : 0u;
                o.y = in ? pack_bf16x2(v[2], v[3]) : 0u;
                *(uint2 *)(smem + pos * YS + (wn * TN1 + ni * 16 + fq * 4) * 2) = o;
            }
        }

        // ---- phase 2: 3x3 over the y image ---------------------------------------------------------------------------
        // pixel of (row block, lane): blocks 0 .. th-1 are the first 16 columns of a tile row (consecutive positions: the
        // bank-conflict-free case), the blocks after them collect columns 16..18 of all rows
        auto block_pixel = [&](int mi, int &r, int &c) -> bool {
            const int bi = wm * MI + mi;
            r = bi;
            c = frow;
            if (bi < th) return true;
            const int q = (bi - th) * 16 + frow;
            r = (q * 43) >> 7;                     // q / 3 for q < 128
            c = 16 + q - 3 * r;
            return q < 3 * th;
        };
        int ybase[MI];                             // byte offset in y of the top-left tap of this lane's pixel, per row block
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) {
            int r, c;
            const bool v = block_pixel(mi, r, c);
            ybase[mi] = (v ? (r * RB_PW + c) * YS : 0) + fq * 16;
        }
        f32x4 acc[NI][MI];
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) {
            const f32x4 b0 = *(const f32x4 *)(p.b2 + wn * TN2 + ni * 16 + fq * 4);
#pragma unroll
            for (int mi = 0; mi < MI; ++mi) acc[ni][mi] = b0;
        }
        constexpr int NMF = NI * MI;
        // L2 prefetch.  A workgroup's memory pipe delivers HBM-sourced data at ~13 bytes per clock whatever the other
        // workgroups do (latency x outstanding lines), three times slower than L2 hits, and LDS has no room for the next
        // tile's x while phase 2 runs.  So during phase 2 every wave touches (one dword per 128-byte line, result
        // unused) the lines the tile boundary will need: the next tile's x and this tile's shortcut operand.
        int nb = 0, nh0 = 0, nw0 = 0, nth = 0;
        const bool have_next = tile + (int)gridDim.x < p.total_tiles;
        if (have_next) tile_coords(tile + (int)gridDim.x, nb, nh0, nw0, nth);
        // The loaded dword's register must stay allocated until the load has landed: the compiler does not know that an
        // asm load writes its output later, so `pfd` is kept alive up to the next s_waitcnt vmcnt(0) (pf_landed()).
        unsigned pfd = 0;
        auto touch = [&](const char *addr) {
            asm volatile("global_load_dword %0, %1, off" : "=v"(pfd) : "v"(addr) : "memory");
        };
        auto pf_landed = [&]() { asm volatile("" ::"v"(pfd) : "memory"); };
        auto prefetch_step = [&](int kt) {
            {
                constexpr int X0 = 2, XN = (RB_NPOS * (C * 2 / 128) + 511) / 512;        // k-steps that touch the next x
                constexpr int R0 = NK2 - 4, RN = (256 * (C * 2 / 128) + 511) / 512;      // ... this tile's shortcut lines
                constexpr int LPP = C * 2 / 128;                                         // lines per pixel
                if (kt >= X0 && kt < X0 + XN && have_next) {
                    const int L = ((kt - X0) * 8 + wave) * 64 + lane;
                    const int pos = L / LPP, q = L - pos * LPP;
                    const int pr = (pos * 3121) >> 16, pc = pos - pr * RB_PW;
                    const int h = nh0 - 1 + pr, w = nw0 - 1 + pc;
                    if (pos < RB_NPOS && pr < nth + 2 && (unsigned)h < (unsigned)p.H && (unsigned)w < (unsigned)p.W)
                        touch(p.x + ((unsigned)((nb * p.H + h) * p.W + w) * (unsigned)(p.x_ld * 2) + (unsigned)(q * 128)));
                }
                if (kt >= R0 && kt < R0 + RN) {
                    const int L = ((kt - R0) * 8 + wave) * 64 + lane;
                    const int m = L / LPP, q = L - m * LPP;                               // pixel slot (row-major in the tile)
                    const int r = (m * 27) >> 9, c = m - r * RB_TW;                       // m / 19 for m < 256
                    if (r < th)
                        touch(p.x + ((unsigned)((b * p.H + h0 + r) * p.W + w0 + c) * (unsigned)(p.x_ld * 2) + (unsigned)(q * 128)));
                }
            }
        };
        auto kstep2 = [&](int kt, auto prefetch) {
            constexpr bool PF = decltype(prefetch)::value;
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            lds_barrier();                         // W2 of k-step kt has landed everywhere (first time: the y image too)
            pf_landed();
            prefetch_step(kt);
            const int tap = kt / KPT;
            const int th3 = (tap * 11) >> 5;       // tap / 3
            const int toff = (th3 * RB_PW + (tap - 3 * th3)) * YS + (kt - tap * KPT) * 128;
            const char *Ws = smem + ((kt & 1) ? W2_ODD : W2_EVEN);
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                frag_t xf[MI], wf[NI];
#pragma unroll
                for (int mi = 0; mi < MI; ++mi) xf[mi] = *(const frag_t *)(smem + ybase[mi] + toff + ks * 64);
#pragma unroll
                for (int ni = 0; ni < NI; ++ni)
                    wf[ni] = *(const frag_t *)(Ws + (wn * TN2 + ni * 16 + frow) * 128 + ((ks * 4 + fq) ^ fsw) * 16);
#pragma unroll
                for (int j = 0; j < NMF; ++j) {
                    if constexpr (PF) {
                        if (ks == 0) {
#pragma unroll
                            for (int q = 0; q < B2_IT; ++q) {
                                if (j == (q * NMF) / B2_IT) {
                                    __builtin_amdgcn_sched_barrier(0);
                                    piece2(q, kt + 1);
                                    __builtin_amdgcn_sched_barrier(0);
                                }
                            }
                        }
                    }
                    const int ni = j / MI, mi = j % MI;
                    acc[ni][mi] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[ni], xf[mi], acc[ni][mi], 0, 0, 0);
                }
            }
        };
        {
            for (int kt = 0; kt + 1 < NK2; ++kt) kstep2(kt, std::true_type{});
        }
        kstep2(NK2 - 1, std::false_type{});
        lds_barrier();                             // the y image and W2's buffers are no longer read: LDS is free for the next x

        // ---- epilogue: leaky, + x (shortcut), one bf16 rounding; the next tile's x is requested before the stores ----
        const auto rs_out = __builtin_amdgcn_make_buffer_rsrc((void *)p.out, 0, p.out_bytes, 0x00020000);
        const auto rs_x = __builtin_amdgcn_make_buffer_rsrc((void *)p.x, 0, p.x_bytes, 0x00020000);
        const int cb = (fq & 1) ? 16 + (fq - 1) * 4 : fq * 4;
        const int nl = wn * TN2 + cb;              // this lane's first channel in pair 0 (after the swap)
        constexpr unsigned OOB = 0xffffff00u;
        unsigned pix[MI];
        u32x4 rv[MI][NP2];
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) {
            int r, c;
            const bool v = block_pixel(mi, r, c);
            pix[mi] = v ? (unsigned)((b * p.H + h0 + r) * p.W + w0 + c) : 0xffffffffu;
#pragma unroll
            for (int pr = 0; pr < NP2; ++pr)
                rv[mi][pr] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(
                    rs_x, v ? (pix[mi] * (unsigned)p.x_ld + (unsigned)nl) * 2u : OOB, pr * 64, 0));
        }
        u32x4 o[MI][NP2];
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) {
#pragma unroll
            for (int pr = 0; pr < NP2; ++pr) {
                f32x4 x = acc[2 * pr][mi], y = acc[2 * pr + 1][mi];
                float lo[4], hi[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    x[j] = leaky_max(x[j], slope2);
                    y[j] = leaky_max(y[j], slope2);
                    const auto sw = __builtin_amdgcn_permlane16_swap(__float_as_uint(x[j]), __float_as_uint(y[j]), false, false);
                    lo[j] = __uint_as_float(sw[0]);
                    hi[j] = __uint_as_float(sw[1]);
                }
                const u32x4 r = rv[mi][pr];
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    lo[2 * j] += bf16_to_f32(r[j] & 0xffff);
                    lo[2 * j + 1] += __uint_as_float(r[j] & 0xffff0000u);
                    hi[2 * j] += bf16_to_f32(r[2 + j] & 0xffff);
                    hi[2 * j + 1] += __uint_as_float(r[2 + j] & 0xffff0000u);
                }
                o[mi][pr][0] = pack_bf16x2(lo[0], lo[1]); o[mi][pr][1] = pack_bf16x2(lo[2], lo[3]);
                o[mi][pr][2] = pack_bf16x2(hi[0], hi[1]); o[mi][pr][3] = pack_bf16x2(hi[2], hi[3]);
            }
        }
        unsigned ooff[MI];
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
            ooff[mi] = pix[mi] != 0xffffffffu ? (pix[mi] * (unsigned)p.out_ld + (unsigned)nl) * 2u : OOB;
        pf_landed();                               // (the shortcut loads above are younger than any touch and have been waited for)
        __builtin_amdgcn_sched_barrier(0);
        const int next = tile + (int)gridDim.x;
        if (next < p.total_tiles) {                // the next tile's x is requested ahead of this tile's stores
            tile_coords(next, b, h0, w0, th);
            issue_x(b, h0, w0, th);
        }
        __builtin_amdgcn_sched_barrier(0);
        {
#pragma unroll
            for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                for (int pr = 0; pr < NP2; ++pr)
                    __builtin_amdgcn_raw_buffer_store_b128(o[mi][pr], rs_out, ooff[mi], pr * 64, 0);
        }
    }
}

template <int CM, int WM, int WN>
int launch_resblock(ResArgs a, hipStream_t stream) {
    constexpr int C = 2 * CM;
    constexpr int YB = RB_M1 * (CM * 2 + 32), XB = (C / 64) * RB_M1 * 128, W2S = C * 128;
    constexpr int lds = XB > YB + 2 * W2S ? XB : YB + 2 * W2S;
    auto kern = resblock_kernel<CM, WM, WN>;
    static thread_local int attr_dev = -1;
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (attr_dev != dev) {
        if (hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess) {
            y3_set_error("resblock: cannot reserve %d bytes of LDS", lds);
            return Y3_E_LAUNCH;
        }
        attr_dev = dev;
    }
    const int cus = y3_device_cus();
    const int grid = a.total_tiles < cus ? a.total_tiles : cus;
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(64 * WM * WN), lds, stream, a);
    return y3_check_launch("resblock");
}

// rows of a feature map -> row tiles of at most RB_TH rows, as even as possible
void row_tiles(int h, int *tiles_y, int *base, int *rem) {
    *tiles_y = (h + RB_TH - 1) / RB_TH;
    *base = h / *tiles_y;
    *rem = h % *tiles_y;
}

}  // namespace

extern "C" int y3_resblock_tiles(const y3_resblock_desc *d, int64_t *tiles) {
    Y3_REQUIRE(d != nullptr && tiles != nullptr, Y3_E_INVALID, "resblock: null pointer argument");
    Y3_REQUIRE(d->dtype == Y3_BF16, Y3_E_INVALID, "resblock: bf16 only (dtype %d)", d->dtype);
    Y3_REQUIRE(d->c == 256 || d->c == 128 || d->c == 64, Y3_E_INVALID, "resblock: %d channels (256, 128 or 64)", d->c);
    Y3_REQUIRE(d->batch > 0 && d->h > 0 && d->w > 0, Y3_E_INVALID, "resblock: %d x %d x %d", d->batch, d->h, d->w);
    if (d->c == 128) {                             // conv_resblock128.hip: 8 x 16-pixel tiles, any width
        *tiles = y3_resblock128_tiles(d->batch, d->h, d->w);
        return Y3_OK;
    }
    if (d->c == 64) {                              // conv_resblock64.hip: 16 x 16-pixel tiles, any width
        *tiles = y3_resblock64_tiles(d->batch, d->h, d->w);
        return Y3_OK;
    }
    Y3_REQUIRE(d->w % RB_TW == 0, Y3_E_INVALID, "resblock: %d x %d x %d (width must be a multiple of %d)", d->batch, d->h, d->w, RB_TW);
    int ty, base, rem;
    row_tiles(d->h, &ty, &base, &rem);
    *tiles = (int64_t)d->batch * ty * (d->w / RB_TW);
    return Y3_OK;
}

extern "C" int y3_resblock_fused(const y3_resblock_desc *d, const void *x, const void *w1, const float *b1,
                                 const void *w2, const float *b2, void *out, const void *zeros, void *stream) {
    int64_t tiles = 0;
    const int rc = y3_resblock_tiles(d, &tiles);
    if (rc) return rc;
    Y3_REQUIRE(x && w1 && b1 && w2 && b2 && out && zeros, Y3_E_INVALID, "resblock: null pointer argument");
    const int c = d->c, cm = c / 2;
    const int k2 = (9 * cm + 63) / 64 * 64;        // y3_conv_weight_layout pads packed rows to whole 64-element k-steps
    Y3_REQUIRE(d->k1_pad == c && d->n1_pad == cm && d->k2_pad == k2 && d->n2_pad == c, Y3_E_INVALID,
               "resblock: weight layout (%d,%d,%d,%d), expected (%d,%d,%d,%d)", d->k1_pad, d->n1_pad, d->k2_pad, d->n2_pad,
               c, cm, k2, c);
    Y3_REQUIRE(d->x_ld >= c && d->out_ld >= c, Y3_E_INVALID, "resblock: x_ld %d / out_ld %d too small", d->x_ld, d->out_ld);
    Y3_REQUIRE(aligned16(x) && aligned16(out) && aligned16(w1) && aligned16(w2) && aligned16(b1) && aligned16(b2) &&
               (d->x_ld * 2) % 16 == 0 && (d->out_ld * 2) % 16 == 0, Y3_E_ALIGN, "resblock: operands not 16-byte aligned");
    const long pixels = (long)d->batch * d->h * d->w;
    const long xb = pixels * d->x_ld * 2, ob = pixels * d->out_ld * 2;
    Y3_REQUIRE(xb < 0xffff0000L && ob < 0xffff0000L && tiles < 0x7fffffffL, Y3_E_INVALID,
               "resblock: tensors of %ld / %ld bytes exceed a 4 GiB buffer descriptor", xb, ob);
    // every tile reads its neighbours' halo pixels of x, so the sum cannot be written over x (unlike y3_conv2d_fused,
    // whose residual operand may be `out`: same lane, load before store)
    Y3_REQUIRE((const char *)x + xb <= (const char *)out || (const char *)out + ob <= (const char *)x, Y3_E_INVALID,
               "resblock: x and out overlap (tiles read their neighbours' pixels of x)");
    if (c == 128) return y3_resblock128_launch(d, x, w1, b1, w2, b2, out, zeros, (hipStream_t)stream);
    if (c == 64) return y3_resblock64_launch(d, x, w1, b1, w2, b2, out, zeros, (hipStream_t)stream);
    ResArgs a;
    a.x = (const char *)x; a.w1 = (const char *)w1; a.w2 = (const char *)w2; a.zeros = (const char *)zeros;
    a.b1 = b1; a.b2 = b2; a.out = (char *)out;
    a.H = d->h; a.W = d->w; a.x_ld = d->x_ld; a.out_ld = d->out_ld;
    int ty;
    row_tiles(d->h, &ty, &a.th_base, &a.th_rem);
    a.tiles_x = d->w / RB_TW;
    a.tiles_img = ty * a.tiles_x;
    a.total_tiles = (int)tiles;
    a.flags = d->flags & (1 | 2 | 32);
    a.x_bytes = (unsigned)xb; a.out_bytes = (unsigned)ob;
    a.fd_img = make_fastdiv(a.tiles_img);
    a.fd_tx = make_fastdiv(a.tiles_x);
    return launch_resblock<128, 2, 4>(a, (hipStream_t)stream);
}
