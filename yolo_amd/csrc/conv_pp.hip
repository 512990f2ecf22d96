// Implicit-GEMM convolution, 256 pixels x 256 channels per workgroup, bf16, register epilogue -- the "ping-pong"
// main loop for gfx950.  Same GEMM view, same packed weights, same LDS image (128-byte rows, 16-byte chunks
// XOR-swizzled by (row >> 1) & 7 on the source address and on the fragment read) and the same per-accumulator sum order
// (bias first, 32-deep k blocks ascending) as conv_igemm_kernel<true, 256, 256, 2, 4>, hence the same bits
// (tests/test_gpu_conv.py); conv_igemm.hip's launcher dispatches the layers that tile would run to this kernel.
// Replaces tf.layers.conv2d + batch_normalization + leaky_relu (+ tf.math.add) of build_model.py:151-178,224.
//
// Why another main loop.  conv_igemm_kernel has one barrier per 64-deep k-step: all eight waves read their twelve
// fragments at once, wait for all of them, and issue 64 MFMAs each; the two waves of a SIMD run in lockstep, so the
// matrix pipe idles through every LDS read burst and every LDS-DMA issue.  Measured with the loads and the epilogue
// removed (profiles/r03a): 1 380-1 510 TFLOP/s, i.e. the loop's own structure costs a third of the pipe.  Here a k-step
// is cut into TWO phases of 32 MFMAs (half of the wave's 128 x 64 tile, both 32-deep halves), and the two wave groups
// (waves 0-3 / 4-7 = the two waves of every SIMD) run half a phase apart: while one group issues its 32 MFMAs the other
// reads the next half's fragments and issues its LDS-DMA pieces, then they swap.  Two s_barriers per phase keep the
// alternation exact (the group that runs behind meets its barrier first, the one ahead last).
//
//   group 0:  LA | MA | LB | MB | LA' ...          L = fragment reads + LDS-DMA pieces + counted wait     M = 32 MFMAs
//   group 1:     | LA | MA | LB | MB ...           | = s_barrier
//
// A k-step's operands are four half-tiles of 16 KB: X0 / X1 = the pixel rows of the waves' first / second 64 pixels,
// W0 / W1 = the channel rows of their first / second 32 channels (W rows are stored [half][wave column][32]).
// Phase A reads W0, X0, W1 (16 ds_read_b128) and multiplies X0 by both; phase B reads X1 (8) and multiplies it by W1 and
// W0, which are still in registers: 24 fragment reads per k-step.  What was tried on the way (profiles/r03d): four
// phases of 16 MFMAs were 12-19 % SLOWER than conv_igemm_kernel -- the vector ALU work of a load segment (address
// selects of the im2col gather, LDS address arithmetic) is starved beside the partner's MFMAs, so a load segment must be
// almost free of it: the gather goes through a buffer descriptor (out-of-range offset = zero fill: 3 VALU instructions
// per piece), every LDS address is one of four per-lane bases + an immediate, waits are counted (s_waitcnt vmcnt(N)
// through the builtin; the compiler inserts none before a ds_read while LDS-DMA is in flight), and s_setprio made no
// difference (+-0.5 %).
//
// The loader runs ahead of the MFMAs across k-steps AND tiles: phase A of k-step u requests X1 of k-step u + 1, phase B
// requests X0, W0, W1 of k-step u + 2 -- every piece at least two phases after the last read of the bytes it overwrites
// -- and each phase retires what the next one reads with ONE counted wait (vmcnt(8): only the eight youngest requests
// may be outstanding); the barrier after it publishes that to the other waves.  The next tile's first k-steps are
// therefore in LDS before this tile's epilogue starts.  A segment (a tile, or a piece of one) holds an even number of
// k-steps, so the LDS buffer of a k-step is a compile-time constant.
//
// y3_conv2d_fused_ws adds the balanced split: conv_pp_split.hip, a second kernel with this one's loop -- measured slower
// inside the network than the partial round it removes (profiles/r03l), so the plan executor uses it only on request, and
// kept out of this kernel because its larger tile set-up, placed where it fits the registers (before a phase's fragment
// reads instead of under their latency), made EVERY launch 8-12 % slower while the two were one kernel (profiles/r03k).
#include "common.h"
#include "conv_common.h"
#include <type_traits>

namespace {

// LDS: X rows of k-step buffer 0 | X rows of buffer 1 | W rows of buffer 0 | W rows of buffer 1 | bias.  With the two
// buffers of an operand 32 KB apart every fragment read is one per-lane base register + a 16-bit immediate.
constexpr int PP_XBUF = 32 * 1024;                  // 256 rows of 128 bytes
constexpr int PP_W0 = 2 * PP_XBUF;
constexpr int PP_BIAS = 4 * PP_XBUF;                // fp32 bias of every output channel of the layer
constexpr int PP_MAX_COUT = 2048;
constexpr int PP_LDS = PP_BIAS + PP_MAX_COUT * 4;

constexpr int pp_wait_enc(int v) { return (v & 15) | (7 << 4) | (15 << 8) | ((v >> 4) << 14); }    // s_waitcnt vmcnt(v) only

template <int N> using ic = std::integral_constant<int, N>;

// One LDS-DMA instruction through a buffer descriptor: 16 bytes per lane from (descriptor base + per-lane byte offset)
// to (wave-uniform LDS address + lane * 16).  An offset past the descriptor's extent writes zeros: the zero padding of
// the convolution and the rows past the end of the tensor cost no address select and no zero page.
__device__ __forceinline__ void bufdma16(__amdgpu_buffer_rsrc_t rs, unsigned off, char *lds_wave_base) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void *)lds_wave_base, 16, off, 0, 0, 0);
}

// MH = 16-pixel row blocks per wave and phase: 4 = the 256-pixel tile (128 pixels per wave group, two halves of 64), 3 = the
// 192-pixel tile (96 per group, halves of 48) for launches whose tile count fits 192-pixel tiles better (19 x 19 at batch
// 64: 1.9 rounds instead of 1.4 -> 2).  MH = 3 keeps the pixel rows of an LDS buffer as [half][group][48], so that a
// half-tile (96 rows) is one run: a loader pass of all 512 threads (64 rows) + half a pass (32 rows) issued by the wave
// group of the same number as the half -- every wave issues 7 LDS-DMA instructions per k-step (8 at MH = 4), the two
// groups alternate in who issues the extra one.  Needs Cin % 64 == 0 (a k-step within one tap).
template <int MH>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2)))
void conv_pp_kernel(const ConvArgs p) {
    constexpr int BM = 64 * MH;                     // pixels per tile
    constexpr int VM_STEADY = MH == 4 ? 8 : 7;      // LDS-DMA instructions a wave may have in flight behind what a phase needs
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;
    const int nk = p.Kpad >> 6;
    const int total = p.total_tiles;
    const bool reverse = p.flags & Y3_CONV_REVERSE;

    // the layer's bias: LDS, so that a tile's accumulators start from a ds_read (a global load there would make the
    // compiler drain every LDS-DMA in flight)
    for (int i = tid; i < p.n_tiles * 64; i += 512) *(f32x4 *)(smem + PP_BIAS + i * 16) = *(const f32x4 *)(p.bias + i * 4);

    // de-phasing, as in conv_igemm_kernel: the workgroups that have a spare round start half a tile late
    if (p.delay_ticks != 0u && (int)blockIdx.x >= p.delay_from) {
        const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
        while (__builtin_amdgcn_s_memrealtime() - t0 < p.delay_ticks) __builtin_amdgcn_s_sleep(64);
    }

    // ---- loader: thread owns LDS chunk slot (tid & 7) of rows (tid >> 3) + 64 q, q = 0..3, of both operands ------
    const auto rs_in = __builtin_amdgcn_make_buffer_rsrc((void *)p.in, 0, p.in_bytes, 0x00020000);
    const auto rs_w = __builtin_amdgcn_make_buffer_rsrc((void *)p.w, 0, p.w_bytes, 0x00020000);
    const int lrow = tid >> 3;
    const int kc = ((tid & 7) ^ ((tid >> 4) & 7)) * 8;       // source chunk (elements) of this LDS slot
    const int tap0 = kc / p.Cin, cin0 = kc - tap0 * p.Cin;
    // this thread's pixel rows: MH = 4: loader passes q = 0..3 (rows 64 q + lrow); MH = 3: [0], [1] = the full pass of half
    // 0 / 1, [2] = the half pass this wave group issues (half wm)
    int a_base[MH == 4 ? 4 : 3];                    // byte offset of the row's window in the input (< 2 GiB: launcher)
    unsigned a_nmask[MH == 4 ? 4 : 3];              // bit t set = tap t of the row lies OUTSIDE the picture
    unsigned w_off[4];
    int tap = 0, cin = 0, tapc = 0, toff = 0;
    int l_tile = blockIdx.x, l_kt = 0, lb = 0, l_m0 = 0;     // wave-uniform: the k-step the loader is on, its LDS buffer
    bool l_live = l_tile < total;
#ifdef Y3_EXPERIMENTS                                        // timing experiments of libyolo3hip_exp.so (garbage results)
    const bool skip_x = p.flags & (1 << 29), skip_w = p.flags & (1 << 28);
#else
    constexpr bool skip_x = false, skip_w = false;
#endif

    auto l_row = [&](int slot, int pixel) {
        const int gm = l_m0 + pixel;
        long base = 0;
        unsigned mask = 0;
        if (gm < p.M) row_setup(p, gm, 2, base, mask);
        a_base[slot] = (int)base;
        a_nmask[slot] = ~mask;
    };
    auto l_rows = [&](int h) {                      // im2col set-up of the X rows of half h
        if constexpr (MH == 4) {                    // loader passes h, h + 2: LDS row = pixel
#pragma unroll
            for (int j = 0; j < 2; ++j) l_row(h + 2 * j, lrow + 64 * (h + 2 * j));
        } else {                                    // LDS row 96 h + r holds pixel 96 (r / 48) + 48 h + r % 48 (group r / 48)
            l_row(h, lrow < 48 ? 48 * h + lrow : 96 + 48 * h + (lrow - 48));
            if (wm == h) l_row(2, 96 + 48 * h + 16 + 8 * wn + (lane >> 3));     // rows 64 + 8 (wave & 3) + lane / 8 of the half
        }
    };
    auto l_tile_begin = [&]() {
        const int bid = xcd_remap(l_tile, total, reverse);
        const int mt = bid / p.n_tiles, nt = bid - mt * p.n_tiles;
        l_m0 = mt * BM;
#pragma unroll
        for (int q = 0; q < 4; ++q) {               // LDS row 64 q + lrow = [half][wave column][32 channels]
            const int idx = lrow + 64 * (q & 1);
            const int ch = nt * 256 + (idx >> 5) * 64 + (q >> 1) * 32 + (idx & 31);
            w_off[q] = ((unsigned)ch * (unsigned)p.Kpad + (unsigned)kc) * 2u;
        }
        tap = tap0;
        cin = cin0;
    };
    auto l_kstep_begin = [&]() {                    // this thread's (tap, channel) of the loader's k-step, then advance
        const int th = p.ks == 3 ? (tap * 11) >> 5 : tap;
        const int tw = tap - th * p.ks;
        toff = ((th * p.W + tw) * p.in_ld + cin) * 2;
        tapc = tap;
        tap += p.step_tap;
        cin += p.step_cin;
        const bool wrap = cin >= p.Cin;
        cin = wrap ? cin - p.Cin : cin;
        tap = wrap ? tap + 1 : tap;
    };
    auto l_advance = [&]() {                        // on to the next k-step (of this tile or of the workgroup's next)
        ++l_kt;
        lb ^= 1;
        if (l_kt == nk) {
            l_kt = 0;
            l_tile += (int)gridDim.x;
            l_live = l_tile < total;
            if (l_live) {
                l_tile_begin();
                l_rows(0);                          // (the rows of X1 are set up a phase later, before their pieces)
            }
        }
        if (l_live) l_kstep_begin();
    };
    auto stage_x = [&](int h) {                     // three vector instructions per piece besides the DMA itself
        if (!l_live || skip_x) return;
        if constexpr (MH == 4) {
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int q = h + 2 * j;
                const unsigned out_of_picture = (unsigned)__builtin_amdgcn_sbfe((int)a_nmask[q], tapc, 1);    // 0 or ~0
                bufdma16(rs_in, (unsigned)(a_base[q] + toff) | out_of_picture, smem + lb * PP_XBUF + q * 8192 + wave * 1024);
            }
        } else {
            const unsigned oop = (unsigned)__builtin_amdgcn_sbfe((int)a_nmask[h], tapc, 1);
            bufdma16(rs_in, (unsigned)(a_base[h] + toff) | oop, smem + lb * PP_XBUF + h * (96 * 128) + wave * 1024);
            if (wm == h) {                          // the half's last 32 rows: 8 per wave of the group with the half's number
                const unsigned oop2 = (unsigned)__builtin_amdgcn_sbfe((int)a_nmask[2], tapc, 1);
                bufdma16(rs_in, (unsigned)(a_base[2] + toff) | oop2, smem + lb * PP_XBUF + h * (96 * 128) + 64 * 128 + wn * 1024);
            }
        }
    };
    auto stage_w = [&](int h) {
        if (!l_live || skip_w) return;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int q = 2 * h + j;
            bufdma16(rs_w, w_off[q] + (unsigned)(l_kt * 128), smem + PP_W0 + lb * PP_XBUF + q * 8192 + wave * 1024);
        }
    };

    // ---- consumer addressing: four per-lane bases, everything else is an immediate ----------------------------------
    const int frow = lane & 15, fq = lane >> 4, fsw = (lane >> 1) & 7;
    const int so0 = ((0 + fq) ^ fsw) * 16, so1 = ((4 + fq) ^ fsw) * 16;
    constexpr int X_GROUP_ROWS = MH == 4 ? 128 : 48, X_HALF_ROWS = MH == 4 ? 64 : 96;    // LDS rows between the groups / the halves
    const int xb0 = (wm * X_GROUP_ROWS + frow) * 128 + so0, xb1 = (wm * X_GROUP_ROWS + frow) * 128 + so1;
    const int wb0 = PP_W0 + (wn * 32 + frow) * 128 + so0, wb1 = PP_W0 + (wn * 32 + frow) * 128 + so1;

    // ---- prologue: k-step 0 complete, k-step 1's X0, W0 and W1 in flight ---------------------------------------------
    if (l_live) {
        l_tile_begin();
        l_rows(0);
        l_rows(1);
        l_kstep_begin();
    }
    __syncthreads();                                // (the bias is in LDS; nothing else is outstanding yet)
    stage_x(0);
    stage_w(0);
    stage_w(1);
    stage_x(1);
    l_advance();
    stage_x(0);
    stage_w(0);
    stage_w(1);
    __builtin_amdgcn_s_waitcnt(pp_wait_enc(MH == 4 ? 6 : 5));      // (k-step 1's X0, W0, W1: at least 5 instructions)
    __builtin_amdgcn_s_barrier();

    for (int c_tile = blockIdx.x; c_tile < total; c_tile += (int)gridDim.x) {
    const int bid = xcd_remap(c_tile, total, reverse);
    const int mt = bid / p.n_tiles, nt = bid - mt * p.n_tiles;
    const int m0 = mt * BM, n0 = nt * 256;
    f32x4 acc[4][2 * MH];
#pragma unroll
    for (int ni = 0; ni < 4; ++ni) {
        const f32x4 b0 = *(const f32x4 *)(smem + PP_BIAS + (n0 + wn * 64 + ni * 16 + fq * 4) * 4);
#pragma unroll
        for (int mi = 0; mi < 2 * MH; ++mi) acc[ni][mi] = b0;
    }
    // One k-step on LDS buffer CB (compile-time, so that the fragment reads need no address arithmetic; the number of
    // k-steps is even: y3_conv_pp_ok): two phases of 32 MFMAs.
    auto kstep = [&](auto CB) {
        constexpr int cb = decltype(CB)::value;
        bf16x8 xf[MH][2], wf[4][2];
        auto read_x = [&](auto XH) {
            constexpr int xh = decltype(XH)::value;
#pragma unroll
            for (int mi = 0; mi < MH; ++mi) {
                xf[mi][0] = *(const bf16x8 *)(smem + xb0 + (cb * PP_XBUF + (xh * X_HALF_ROWS + mi * 16) * 128));
                xf[mi][1] = *(const bf16x8 *)(smem + xb1 + (cb * PP_XBUF + (xh * X_HALF_ROWS + mi * 16) * 128));
            }
        };
        auto read_w = [&](auto WH) {
            constexpr int wh = decltype(WH)::value;
#pragma unroll
            for (int ni = 0; ni < 2; ++ni) {
                wf[2 * wh + ni][0] = *(const bf16x8 *)(smem + wb0 + (cb * PP_XBUF + (wh * 128 + ni * 16) * 128));
                wf[2 * wh + ni][1] = *(const bf16x8 *)(smem + wb1 + (cb * PP_XBUF + (wh * 128 + ni * 16) * 128));
            }
        };
        auto mfmas = [&](auto XH, auto WH) {        // one quadrant, both 32-deep halves: 16 MFMAs
            constexpr int xh = decltype(XH)::value, wh = decltype(WH)::value;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int ni = 0; ni < 2; ++ni)
#pragma unroll
                    for (int mi = 0; mi < MH; ++mi)
                        acc[2 * wh + ni][MH * xh + mi] =
                            __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[2 * wh + ni][ks], xf[mi][ks], acc[2 * wh + ni][MH * xh + mi], 0, 0, 0);
        };
        // group 1 runs half a phase behind group 0: its barrier comes first, group 0's last.  A load segment ends
        // with its own fragment reads retired (lgkmcnt(0) BEFORE the barrier): what a phase has read may be
        // overwritten by requests issued in the next one.
        auto seg_load = [&]() {
            if (wm == 1) __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
        };
        auto seg_mfma = [&]() {
            __builtin_amdgcn_s_waitcnt(0xC07F);     // lgkmcnt(0) only (the builtin keeps the compiler's own count exact)
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
        };
        auto seg_end = [&]() {
            __builtin_amdgcn_sched_barrier(0);
            if (wm == 0) __builtin_amdgcn_s_barrier();
        };
        // phase A: X0 x (W0, W1); requests X1 of the loader's k-step (the next one)
        seg_load();
        read_w(ic<0>{});
        read_x(ic<0>{});
        read_w(ic<1>{});
        if (l_kt == 0 && l_live) l_rows(1);         // a new tile's X1 rows (its X0 rows were set up at the last phase B)
        stage_x(1);
        // X1 of this k-step has landed once only the eight youngest requests remain (the last phase B's six, these two)
        if (l_live) __builtin_amdgcn_s_waitcnt(pp_wait_enc(VM_STEADY));
        else __builtin_amdgcn_s_waitcnt(pp_wait_enc(0));
        seg_mfma();
        mfmas(ic<0>{}, ic<0>{});
        mfmas(ic<0>{}, ic<1>{});
        seg_end();
        // phase B: X1 x (W1, W0); requests X0, W0, W1 of the k-step after the next
        seg_load();
        read_x(ic<1>{});
        l_advance();
        stage_x(0);
        stage_w(0);
        stage_w(1);
        // X0, W0, W1 of the next k-step have landed once only the eight youngest remain (its X1, this phase's six)
        if (l_live) __builtin_amdgcn_s_waitcnt(pp_wait_enc(VM_STEADY));
        else __builtin_amdgcn_s_waitcnt(pp_wait_enc(0));
        seg_mfma();
        mfmas(ic<1>{}, ic<1>{});
        mfmas(ic<1>{}, ic<0>{});
        seg_end();
    };
    for (int kt = 0; kt < nk; kt += 2) {
        kstep(ic<0>{});
        kstep(ic<1>{});
    }
#ifdef Y3_EXPERIMENTS
    if (p.flags & (1 << 27)) {                      // timing experiment: no epilogue (one store keeps the accumulators live)
        float sacc = 0.f;
#pragma unroll
        for (int ni = 0; ni < 4; ++ni)
#pragma unroll
            for (int mi = 0; mi < 2 * MH; ++mi) sacc += acc[ni][mi][0] + acc[ni][mi][1] + acc[ni][mi][2] + acc[ni][mi][3];
        if (sacc == 123.456f) *(float *)p.out = sacc;
        continue;
    }
#endif
    u32x4 rv[2 * MH][2];
    if (p.flags & Y3_CONV_RESIDUAL) conv_epilogue_regs<32 * MH, 64, 2 * MH, 4, 0, 1>(p, acc, rv, m0, n0, lane, wm, wn);
    else conv_epilogue_regs<32 * MH, 64, 2 * MH, 4, 0, 0>(p, acc, rv, m0, n0, lane, wm, wn);
    }
}

}  // namespace

bool y3_conv_pp_ok(const ConvArgs &a, int n_pad, int rows) {
    if (!(a.flags & CONV_REG_EPILOGUE) || a.splits > 1) return false;
    if (n_pad % 256 != 0 || n_pad > PP_MAX_COUT || a.Kpad % 128 != 0 || a.Kpad / 64 < 4) return false;     // an even number of k-steps
    if (a.Cin % 8 != 0) return false;
    if (rows == 192 && a.Cin % 64 != 0) return false;                  // (the 192-pixel tile's loader: a k-step within one tap)
    const long in_bytes = ((long)(a.M / a.HoWo) * a.H * a.W + 2L * a.W + 2) * a.in_ld * 2;    // (+ the halo the base offsets may point at)
    if (in_bytes >= 0x7fff0000L || (long)n_pad * a.Kpad * 2 >= 0x7fff0000L) return false;
    return rows == 256 || rows == 192;
}

int y3_conv_pp_launch(ConvArgs a, int n_pad, int rows, long cus, int phase_pct, hipStream_t stream) {
    // with a caller's workspace, the launches that would leave a partial round take the balanced split (conv_pp_split.hip)
    if (rows == 256 && a.sk_slabs != nullptr && y3_conv_pp_slab_bytes(a, n_pad, cus) > 0) return y3_conv_pp_split_launch(a, n_pad, cus, stream);
    a.sk_slabs = nullptr;
    a.n_tiles = n_pad / 256;
    // extents of the LDS-DMA buffer descriptors: anything past them reads as zero (the loader relies on that)
    a.in_bytes = (unsigned)((((long)(a.M / a.HoWo) * a.H * a.W - 1) * a.in_ld + a.Cin) * 2);
    a.w_bytes = (unsigned)((long)n_pad * a.Kpad * 2);
    static thread_local int attr_dev = -1;
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (attr_dev != dev) {
        if (hipFuncSetAttribute((const void *)conv_pp_kernel<4>, hipFuncAttributeMaxDynamicSharedMemorySize, PP_LDS) != hipSuccess ||
            hipFuncSetAttribute((const void *)conv_pp_kernel<3>, hipFuncAttributeMaxDynamicSharedMemorySize, PP_LDS) != hipSuccess) {
            y3_set_error("conv_pp: cannot reserve %d bytes of LDS", PP_LDS);
            return Y3_E_LAUNCH;
        }
        attr_dev = dev;
    }
    const long m_tiles = ((long)a.M + rows - 1) / rows;
    const long nblk = m_tiles * a.n_tiles;
    if (nblk <= 0 || nblk > 0x7fffffffL) { y3_set_error("conv_pp: grid of %ld blocks", nblk); return Y3_E_INVALID; }
    a.total_tiles = (int)nblk;
    const long grid = nblk <= cus ? nblk : cus;          // one workgroup per CU (LDS), persistent over tiles
    // de-phasing as in conv_igemm.hip's launcher: with a spare round, the workgroups that have one tile fewer start
    // half a tile period late, so that their epilogues fall into the others' main loops
    a.delay_ticks = 0;
    a.delay_from = 0;
    const long rem = nblk % grid;
    if (grid == cus && nblk >= 2 * cus && rem != 0 && phase_pct > 0) {
        const double tile_us = 2.0 * rows * 256 * (double)a.Kpad / 5.4e6;         // at ~5.4 TFLOP/s per CU
        a.delay_ticks = (unsigned)(tile_us * phase_pct);                          // 100 ticks per us, percent
        a.delay_from = (int)rem;
    }
    if (rows == 256) hipLaunchKernelGGL(conv_pp_kernel<4>, dim3((unsigned)grid), dim3(512), PP_LDS, stream, a);
    else hipLaunchKernelGGL(conv_pp_kernel<3>, dim3((unsigned)grid), dim3(512), PP_LDS, stream, a);
    return y3_check_launch("conv_pp");
}
