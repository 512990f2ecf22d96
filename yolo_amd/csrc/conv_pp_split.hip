// conv_pp_kernel (conv_pp.hip: the 256 x 256 bf16 tile on the two-wave-group main loop) with the BALANCED SPLIT of
// y3_conv2d_fused_ws: same loop, same LDS image, same sum order, same bits -- and a work list (PpWork below) in which a
// launch that would leave a partial last round of tiles deals the k-steps of its last tiles evenly over the workgroups,
// cut tiles summed in order through fp32 slabs.  A kernel of its own because the larger segment set-up only fits the
// 256 registers when it runs BEFORE a phase's fragment reads instead of under their latency, and that order costs the
// plain launches 8-12 % (profiles/r03k).  Launched only with a caller's workspace, i.e. on request: inside the bench
// network the split measured slower than the partial round it removes (profiles/r03l).
// Replaces tf.layers.conv2d + batch_normalization + leaky_relu (+ tf.math.add) of build_model.py:151-178,224.
#include "common.h"
#include "conv_common.h"
#include <type_traits>

namespace {

// LDS: X rows of k-step buffer 0 | X rows of buffer 1 | W rows of buffer 0 | W rows of buffer 1 | bias | hand-off word.
// With the two buffers of an operand 32 KB apart every fragment read is one per-lane base register + a 16-bit immediate.
constexpr int PP_XBUF = 32 * 1024;                  // 256 rows of 128 bytes
constexpr int PP_W0 = 2 * PP_XBUF;
constexpr int PP_BIAS = 4 * PP_XBUF;                // fp32 bias of every output channel of the layer
constexpr int PP_MAX_COUT = 2048;
constexpr int PP_DEC = PP_BIAS + PP_MAX_COUT * 4;   // balanced split: 1 = the predecessor's partial sums are there, 2 = recompute
constexpr int PP_LDS = PP_DEC + 16;
constexpr int PP_SLAB = 256 * 256 * 4;              // bytes of one workgroup's fp32 partial sums (balanced split)
constexpr int PP_SLAB_AUX = 16;                     // cache policy of the slab accesses: sc1 (agent scope, written through; + nt: no gain)
constexpr int PP_SPIN_LIMIT = 1 << 17;              // polls of the predecessor's flag before giving up and recomputing

constexpr int pp_wait_enc(int v) { return (v & 15) | (7 << 4) | (15 << 8) | ((v >> 4) << 14); }    // s_waitcnt vmcnt(v) only

template <int N> using ic = std::integral_constant<int, N>;

// One LDS-DMA instruction through a buffer descriptor: 16 bytes per lane from (descriptor base + per-lane byte offset +
// wave-uniform offset) to (wave-uniform LDS address + lane * 16).  An offset past the descriptor's extent writes zeros:
// the zero padding of the convolution and the rows past the end of the tensor cost no address select and no zero page.
__device__ __forceinline__ void bufdma16(__amdgpu_buffer_rsrc_t rs, unsigned off, char *lds_wave_base, int uniform_off = 0) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void *)lds_wave_base, 16, off, uniform_off, 0, 0);
}

// The workgroup's work list.  Plain: whole tiles blockIdx.x, + gridDim.x, ...  Balanced split (p.sk_slabs != nullptr):
// an XCD group of G workgroups first shares p.lock_rounds whole rounds of its tiles as before (workgroup i: tiles i, i + G,
// ...: neighbouring tiles run at the same time on one L2, and the walk through the tensor stays the one the layers before
// and after expect), and the k-step pairs ("units") of the REMAINING tiles -- between half a tile and a tile and a half per
// workgroup -- are dealt evenly, each workgroup a contiguous range of them.  A tile cut between workgroups is summed IN
// ORDER: whoever has its first k-steps (PP_HEAD, from the bias) runs them as the first thing it does and leaves the fp32
// accumulators in its slab; a piece from the middle of a tile (PP_MID: slab in, slab out) runs after the workgroup's first
// whole tile, by when the head is long done; the last k-steps (PP_TAIL) are the last thing a workgroup does, starting from
// the predecessor's accumulators -- the same chain of MFMAs as one workgroup alone: same bits.  A range lies within two
// neighbouring tiles, so a workgroup leaves at most one slab and takes at most one, always its predecessor's.
struct PpWork {
    int nseg, has_head, has_mid, mid_at, rl, n_full, first_full, ja, jb, oa2, ob2;     // (oa2 / ob2: k-steps, even)
    int start, len, i, G;                           // the XCD group's range of logical tiles, this workgroup's place in it
};
enum { PP_FULL = 0, PP_HEAD = 1, PP_TAIL = 2, PP_MID = 3 };

__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2)))
void conv_pp_split_kernel(const ConvArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;
    const int nk = p.Kpad >> 6;
    const int total = p.total_tiles;
    const bool reverse = p.flags & Y3_CONV_REVERSE;
    const bool balanced = p.sk_slabs != nullptr;

    // ---- the work list (wave-uniform scalars) ------------------------------------------------------------------------
    PpWork wk;
    int cons_seg = -1;                              // the segment that starts from the predecessor's slab (PP_TAIL / PP_MID)
    if (balanced) {
        const int x = blockIdx.x & 7, q = total >> 3, r = total & 7;
        wk.i = blockIdx.x >> 3;
        wk.G = (int)gridDim.x >> 3;
        wk.start = x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q;
        wk.len = q + (x < r ? 1 : 0);
        wk.rl = p.lock_rounds;
        const int npt = nk >> 1;
        const int U = (wk.len - wk.rl * wk.G) * npt;        // units of the tiles behind the whole rounds
        auto cut = [&](int k) {                     // boundary k of the group's G shares, kept 2 units away from tile edges
            int u = (int)(((long)k * U) / wk.G);
            const int o = u % npt;
            if (o < 2) u -= o;
            else if (o > npt - 2) u += npt - o;
            return u;
        };
        const int u0 = cut(wk.i), u1 = cut(wk.i + 1);
        const bool any = u1 > u0;
        wk.ja = u0 / npt;
        wk.oa2 = any ? 2 * (u0 - wk.ja * npt) : 0;
        wk.jb = any ? (u1 - 1) / npt : wk.ja;
        wk.ob2 = any ? 2 * (u1 - wk.jb * npt) : nk;
        wk.has_mid = any && wk.ja == wk.jb && wk.oa2 > 0 && wk.ob2 < nk;
        const int has_tail = wk.oa2 > 0 && !wk.has_mid;
        wk.has_head = any && wk.ob2 < nk && !wk.has_mid;
        wk.mid_at = wk.rl > 0 ? 1 : 0;
        wk.first_full = wk.ja + has_tail;
        const int last_full = wk.jb - wk.has_head;
        wk.n_full = any && !wk.has_mid && last_full >= wk.first_full ? last_full - wk.first_full + 1 : 0;
        wk.nseg = wk.rl + wk.has_head + wk.has_mid + wk.n_full + has_tail;
        cons_seg = has_tail ? wk.nseg - 1 : (wk.has_mid ? wk.mid_at : -1);
    } else {
        wk.start = 0; wk.len = total; wk.ja = wk.jb = wk.oa2 = wk.ob2 = wk.has_head = wk.has_mid = wk.mid_at = wk.first_full = 0;
        wk.i = (int)blockIdx.x; wk.G = (int)gridDim.x;
        wk.rl = (int)blockIdx.x < total ? (total - 1 - (int)blockIdx.x) / (int)gridDim.x + 1 : 0;
        wk.n_full = 0;
        wk.nseg = wk.rl;
    }
    int cons_kt0 = wk.oa2;                          // 0 once the workgroup has decided to recompute the tile's first k-steps itself
    // segment s -> logical tile, first and one-past-last k-step, kind
    auto segment = [&](int sidx, int &tile, int &kt0, int &kt1, int &kind) {
        if (!balanced) {
            tile = xcd_remap((int)blockIdx.x + sidx * (int)gridDim.x, total, reverse);
            kt0 = 0; kt1 = nk; kind = PP_FULL;
            return;
        }
        int j, s = sidx;
        kt0 = 0; kt1 = nk; kind = PP_FULL;
        bool done = false;
        if (wk.has_head) {
            if (s == 0) { j = wk.rl * wk.G + wk.jb; kt1 = wk.ob2; kind = PP_HEAD; done = true; }
            --s;
        }
        if (!done && wk.has_mid) {
            if (s == wk.mid_at) { j = wk.rl * wk.G + wk.ja; kt0 = cons_kt0; kt1 = wk.ob2; kind = PP_MID; done = true; }
            else if (s > wk.mid_at) --s;
        }
        if (!done) {
            if (s < wk.rl) j = wk.i + s * wk.G;                                  // a whole round's tile
            else if (s - wk.rl < wk.n_full) j = wk.rl * wk.G + wk.first_full + (s - wk.rl);
            else { j = wk.rl * wk.G + wk.ja; kt0 = cons_kt0; kind = PP_TAIL; }
        }
        tile = wk.start + (reverse ? wk.len - 1 - j : j);
    };

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
    int a_base[4];                                  // byte offset of the row's window in the input (< 2 GiB: launcher)
    unsigned a_nmask[4];                            // bit t set = tap t of the row lies OUTSIDE the picture
    unsigned w_off = 0;                             // this thread's chunk of its weight row in loader pass 0
    int tap = 0, cin = 0, tapc = 0, toff = 0;
    int l_seg = 0, l_kt = 0, l_kt1 = 0, lb = 0, l_m0 = 0;    // wave-uniform: the k-step the loader is on, its LDS buffer
    bool l_live = wk.nseg > 0, l_newtile = false;

    auto l_rows = [&](int h) {                      // im2col set-up of the X rows of half h (loader passes h, h + 2)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int q = h + 2 * j;
            const int gm = l_m0 + lrow + 64 * q;
            long base = 0;
            unsigned mask = 0;
            if (gm < p.M) row_setup(p, gm, 2, base, mask);
            a_base[q] = (int)base;
            a_nmask[q] = ~mask;
        }
    };
    // The predecessor's flag, polled by ONE wave, a bounded number of times; the verdict goes to every wave through LDS
    // (read at least two k-steps = eight barriers later).  Not there in time -> this workgroup computes the head itself.
    auto decide_tail = [&]() {
        if (wave == 0) {
            const unsigned *flag = p.sk_flags + ((int)blockIdx.x - 8);
            int n = 0;
            unsigned f = __hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            while (f == 0u && ++n < PP_SPIN_LIMIT) {
                __builtin_amdgcn_s_sleep(16);
                f = __hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
#ifdef Y3_EXPERIMENTS
            if (p.flags & (1 << 30)) f = 0u;        // test hook: always take the recompute path
            // what the poll saw, for tools/pp_split_probe.py: polls << 8 | 3 (partial sums taken) or 5 (recomputed)
            if (lane == 0) __hip_atomic_store(p.sk_flags + ((int)blockIdx.x - 8), ((unsigned)n << 8) | (f != 0u ? 3u : 5u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#endif
            *(volatile int *)(smem + PP_DEC) = f != 0u ? 1 : 2;
        }
    };
    auto l_segment_begin = [&]() {                  // the loader enters segment l_seg
        int tile, kt0, kind;
        if (l_seg == cons_seg) cons_kt0 = __builtin_amdgcn_readfirstlane(*(volatile int *)(smem + PP_DEC)) == 1 ? wk.oa2 : 0;
        segment(l_seg, tile, kt0, l_kt1, kind);
        l_kt = kt0;
        const int mt = tile / p.n_tiles, nt = tile - mt * p.n_tiles;
        l_m0 = mt * 256;
        // LDS row 64 q + lrow of W = [half q >> 1][wave column][32 channels]: channel = (idx >> 5) * 64 + (q >> 1) * 32 +
        // (idx & 31), idx = lrow + 64 (q & 1) -- pass q is pass 0 moved by 128 (q & 1) + 32 (q >> 1) channels, a uniform offset
        w_off = ((unsigned)(nt * 256 + (lrow >> 5) * 64 + (lrow & 31)) * (unsigned)p.Kpad + (unsigned)kc) * 2u;
        const int k = kc + kt0 * 64;                // (kt0 != 0 only for the tail of a cut tile: once per launch)
        tap = k / p.Cin;
        cin = k - tap * p.Cin;
        l_rows(0);                                  // (the rows of X1 are set up a phase later, before their pieces)
        l_newtile = true;
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
    auto l_advance = [&]() {                        // on to the next k-step (of this segment or of the next)
        ++l_kt;
        lb ^= 1;
        if (l_seg == cons_seg - 1 && l_kt == l_kt1 - 2) decide_tail();      // (segments are at least four k-steps long)
        if (l_kt == l_kt1) {
            ++l_seg;
            l_live = l_seg < wk.nseg;
            if (l_live) l_segment_begin();
        }
        if (l_live) l_kstep_begin();
    };
    auto stage_x = [&](int h) {                     // three vector instructions per piece besides the DMA itself
        if (!l_live) return;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int q = h + 2 * j;
            const unsigned out_of_picture = (unsigned)__builtin_amdgcn_sbfe((int)a_nmask[q], tapc, 1);    // 0 or ~0
            bufdma16(rs_in, (unsigned)(a_base[q] + toff) | out_of_picture, smem + lb * PP_XBUF + q * 8192 + wave * 1024);
        }
    };
    auto stage_w = [&](int h) {
        if (!l_live) return;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int q = 2 * h + j;
            bufdma16(rs_w, w_off, smem + PP_W0 + lb * PP_XBUF + q * 8192 + wave * 1024,
                     l_kt * 128 + ((q & 1) * 128 + (q >> 1) * 32) * p.Kpad * 2);
        }
    };

    // ---- consumer addressing: four per-lane bases, everything else is an immediate ----------------------------------
    const int frow = lane & 15, fq = lane >> 4, fsw = (lane >> 1) & 7;
    const int so0 = ((0 + fq) ^ fsw) * 16, so1 = ((4 + fq) ^ fsw) * 16;
    const int xb0 = (wm * 128 + frow) * 128 + so0, xb1 = (wm * 128 + frow) * 128 + so1;
    const int wb0 = PP_W0 + (wn * 32 + frow) * 128 + so0, wb1 = PP_W0 + (wn * 32 + frow) * 128 + so1;

    // ---- prologue: the first k-step complete, the second one's X0, W0 and W1 in flight --------------------------------
    if (cons_seg == 0) decide_tail();               // (a workgroup that starts with the tail of a cut tile: no whole rounds)
    __syncthreads();                                // (the bias and the verdict are in LDS; nothing else is outstanding yet)
    if (l_live) {
        l_segment_begin();
        l_rows(1);
        l_newtile = false;
        l_kstep_begin();
    }
    stage_x(0);
    stage_w(0);
    stage_w(1);
    stage_x(1);
    l_advance();
    stage_x(0);
    stage_w(0);
    stage_w(1);
    __builtin_amdgcn_s_waitcnt(pp_wait_enc(6));
    __builtin_amdgcn_s_barrier();

    for (int c_seg = 0; c_seg < wk.nseg; ++c_seg) {
    int c_tile, c_kt0, c_kt1, c_kind;
    segment(c_seg, c_tile, c_kt0, c_kt1, c_kind);
    const int mt = c_tile / p.n_tiles, nt = c_tile - mt * p.n_tiles;
    const int m0 = mt * 256, n0 = nt * 256;
    f32x4 acc[4][8];
    const int slab_lane = (wave * 64 + lane) * 16;  // slab layout [accumulator 0..31][wave][lane] x 16 bytes
    if ((c_kind == PP_TAIL || c_kind == PP_MID) && c_kt0 != 0) {
        // the k-steps before these were summed by the workgroup before this one (same XCD group): continue from its sums
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        const auto rs_prev = __builtin_amdgcn_make_buffer_rsrc((void *)(p.sk_slabs + (long)((int)blockIdx.x - 8) * PP_SLAB), 0, PP_SLAB, 0x00020000);
#pragma unroll
        for (int ni = 0; ni < 4; ++ni)
#pragma unroll
            for (int mi = 0; mi < 8; ++mi)
                acc[ni][mi] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_prev, slab_lane, (ni * 8 + mi) * 8192, PP_SLAB_AUX));
    } else {
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) {
            const f32x4 b0 = *(const f32x4 *)(smem + PP_BIAS + (n0 + wn * 64 + ni * 16 + fq * 4) * 4);
#pragma unroll
            for (int mi = 0; mi < 8; ++mi) acc[ni][mi] = b0;
        }
    }
    // One k-step on LDS buffer CB (compile-time, so that the fragment reads need no address arithmetic; segments hold an
    // even number of k-steps): two phases of 32 MFMAs.
    auto kstep = [&](auto CB) {
        constexpr int cb = decltype(CB)::value;
        bf16x8 xf[4][2], wf[4][2];
        auto read_x = [&](auto XH) {
            constexpr int xh = decltype(XH)::value;
#pragma unroll
            for (int mi = 0; mi < 4; ++mi) {
                xf[mi][0] = *(const bf16x8 *)(smem + xb0 + (cb * PP_XBUF + (xh * 64 + mi * 16) * 128));
                xf[mi][1] = *(const bf16x8 *)(smem + xb1 + (cb * PP_XBUF + (xh * 64 + mi * 16) * 128));
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
                    for (int mi = 0; mi < 4; ++mi)
                        acc[2 * wh + ni][4 * xh + mi] =
                            __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[2 * wh + ni][ks], xf[mi][ks], acc[2 * wh + ni][4 * xh + mi], 0, 0, 0);
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
        if (l_newtile) {                            // a new tile's X1 rows (its X0 rows were set up at the last phase B);
            l_rows(1);                              // ahead of the fragment reads, while their 64 registers are free
            l_newtile = false;
        }
        __builtin_amdgcn_sched_barrier(0);
        read_w(ic<0>{});
        read_x(ic<0>{});
        read_w(ic<1>{});
        stage_x(1);
        // X1 of this k-step has landed once only the eight youngest requests remain (the last phase B's six, these two)
        if (l_live) __builtin_amdgcn_s_waitcnt(pp_wait_enc(8));
        else __builtin_amdgcn_s_waitcnt(pp_wait_enc(0));
        seg_mfma();
        mfmas(ic<0>{}, ic<0>{});
        mfmas(ic<0>{}, ic<1>{});
        seg_end();
        // phase B: X1 x (W1, W0); requests X0, W0, W1 of the k-step after the next
        seg_load();
        l_advance();                                // (ahead of the fragment reads: a new tile's set-up needs registers)
        __builtin_amdgcn_sched_barrier(0);
        read_x(ic<1>{});
        stage_x(0);
        stage_w(0);
        stage_w(1);
        // X0, W0, W1 of the next k-step have landed once only the eight youngest remain (its X1, this phase's six)
        if (l_live) __builtin_amdgcn_s_waitcnt(pp_wait_enc(8));
        else __builtin_amdgcn_s_waitcnt(pp_wait_enc(0));
        seg_mfma();
        mfmas(ic<1>{}, ic<1>{});
        mfmas(ic<1>{}, ic<0>{});
        seg_end();
    };
    for (int kt = c_kt0; kt < c_kt1; kt += 2) {
        kstep(ic<0>{});
        kstep(ic<1>{});
    }
    if (c_kind == PP_HEAD || c_kind == PP_MID) {
        // a cut tile with k-steps still to come: the fp32 sums go to this workgroup's slab, written through (sc1); every wave drains its
        // stores, the workgroup meets, ONE lane raises the flag (MI355X hand-off recipe: write-through payload + flag)
        const auto rs_own = __builtin_amdgcn_make_buffer_rsrc((void *)(p.sk_slabs + (long)blockIdx.x * PP_SLAB), 0, PP_SLAB, 0x00020000);
#pragma unroll
        for (int ni = 0; ni < 4; ++ni)
#pragma unroll
            for (int mi = 0; mi < 8; ++mi)
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, acc[ni][mi]), rs_own, slab_lane, (ni * 8 + mi) * 8192, PP_SLAB_AUX);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();               // (one more barrier for every wave: the two groups stay in step)
        if (tid == 0) __hip_atomic_store(p.sk_flags + blockIdx.x, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        continue;
    }
    u32x4 rv[8][2];
    if (p.flags & Y3_CONV_RESIDUAL) conv_epilogue_regs<128, 64, 8, 4, 0, 1>(p, acc, rv, m0, n0, lane, wm, wn);
    else conv_epilogue_regs<128, 64, 8, 4, 0, 0>(p, acc, rv, m0, n0, lane, wm, wn);
    }
}

}  // namespace

// Whether a launch of `nblk` tiles on `grid` persistent workgroups takes the balanced split: an XCD group of grid / 8
// workgroups shares its tiles' k-step pairs evenly, which needs at least one tile per workgroup (a cut tile then has
// exactly two owners) and pays only when whole tiles would leave a partial round.
static bool pp_balanced(long nblk, long grid, long cus) {
    return grid == cus && cus % 8 == 0 && nblk > grid && nblk % grid != 0 && (nblk >> 3) >= (grid >> 3);
}

long y3_conv_pp_slab_bytes(const ConvArgs &a, int n_pad, long cus) {
    const long nblk = (((long)a.M + 255) / 256) * (n_pad / 256);
    const long grid = nblk <= cus ? nblk : cus;
    return pp_balanced(nblk, grid, cus) ? grid * (long)PP_SLAB : 0;
}

int y3_conv_pp_split_launch(ConvArgs a, int n_pad, long cus, hipStream_t stream) {
    a.n_tiles = n_pad / 256;
    // extents of the LDS-DMA buffer descriptors: anything past them reads as zero (the loader relies on that)
    a.in_bytes = (unsigned)((((long)(a.M / a.HoWo) * a.H * a.W - 1) * a.in_ld + a.Cin) * 2);
    a.w_bytes = (unsigned)((long)n_pad * a.Kpad * 2);
    static thread_local int attr_dev = -1;
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (attr_dev != dev) {
        if (hipFuncSetAttribute((const void *)conv_pp_split_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, PP_LDS) != hipSuccess) {
            y3_set_error("conv_pp_split: cannot reserve %d bytes of LDS", PP_LDS);
            return Y3_E_LAUNCH;
        }
        attr_dev = dev;
    }
    const long m_tiles = ((long)a.M + 255) / 256;
    const long nblk = m_tiles * a.n_tiles;
    const long grid = nblk <= cus ? nblk : cus;          // one workgroup per CU (LDS), persistent over tiles
    if (nblk <= 0 || nblk > 0x7fffffffL || a.sk_slabs == nullptr || a.sk_flags == nullptr || !pp_balanced(nblk, grid, cus)) {
        y3_set_error("conv_pp_split: not a launch for the balanced split (%ld tiles)", nblk);
        return Y3_E_INVALID;
    }
    a.total_tiles = (int)nblk;
    // whole rounds first, as many as leave at least half a tile per workgroup to deal out: cut tiles then have at most
    // three owners, and a piece from the middle of a tile never has to wait for another one.  No de-phasing delay: the
    // workgroups de-phase by themselves, every one starts with a different share of a cut tile.
    const long q = nblk >> 3, G = grid >> 3;
    a.lock_rounds = (int)((2 * q - G) / (2 * G));
    a.delay_ticks = 0;
    a.delay_from = 0;
    hipLaunchKernelGGL(conv_pp_split_kernel, dim3((unsigned)grid), dim3(512), PP_LDS, stream, a);
    return y3_check_launch("conv_pp_split");
}
