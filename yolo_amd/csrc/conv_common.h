// Shared by the implicit-GEMM convolution kernels (conv_igemm.hip, conv_pp.hip): launch arguments, the im2col row
// set-up and the register epilogue.  Everything lives in an anonymous namespace: each translation unit has its own copy.
#pragma once
#include "common.h"

struct ConvArgs {
    const char *in, *w, *res, *zeros;
    const float *bias;
    char *out;
    int H, W, Cin, in_ld;
    int Ho, Wo, HoWo, Cout8, out_ld, res_ld;
    int ks, stride, pad, Kpad, M, n_tiles, flags, step_tap, step_cin, sub_tap, sub_cin;
    FastDiv fd_howo, fd_wo;
    unsigned out_bytes, res_bytes;     // extents for the buffer descriptors of the register epilogue
    unsigned in_bytes, w_bytes;        // ... and of conv_pp_kernel's LDS-DMA loader
    char *sk_slabs;                    // conv_pp_kernel's balanced split: one 256 KB fp32 slab per workgroup, or null (whole tiles)
    unsigned *sk_flags;                // ... one flag per workgroup, zero at launch
    int total_tiles;
    int delay_from;                    // persistent workgroups with blockIdx.x >= delay_from start delay_ticks late
    unsigned delay_ticks;              // in s_memrealtime ticks (100 MHz)
    int lock_rounds;                   // conv_pp_kernel's balanced split: whole rounds of tiles before the dealt remainder
    int splits, nk_split;              // split-K: a tile's k-steps are dealt to `splits` workgroups, nk_split each
    long slice_elems;                  // ... which write fp32 partial sums to out + split * slice_elems
};

namespace {

// Output row gm -> byte offset of its window's first input element and the 9-bit (1-bit for 1x1)
// mask of taps that fall inside the image.  Branch-free; the two divisions are multiplications.
__device__ __forceinline__ void row_setup(const ConvArgs &p, int gm, int es, long &base, unsigned &mask) {
    const int b = fast_div(gm, p.fd_howo);
    const int r = gm - b * p.HoWo;
    const int ho = fast_div(r, p.fd_wo);
    const int wo = r - ho * p.Wo;
    const int hi0 = ho * p.stride - p.pad, wi0 = wo * p.stride - p.pad;
    base = (((long)b * p.H + hi0) * p.W + wi0) * (long)p.in_ld * es;
    if (p.ks == 1) {
        mask = ((unsigned)hi0 < (unsigned)p.H && (unsigned)wi0 < (unsigned)p.W) ? 1u : 0u;
    } else {
        unsigned vw = 0, m = 0;
#pragma unroll
        for (int k = 0; k < 3; ++k) vw |= ((unsigned)(wi0 + k) < (unsigned)p.W ? 1u : 0u) << k;
#pragma unroll
        for (int k = 0; k < 3; ++k) m |= ((unsigned)(hi0 + k) < (unsigned)p.H ? vw : 0u) << (3 * k);
        mask = m;
    }
}

constexpr int CONV_NO_BIAS = 1 << 24;    // internal flag: split-K partial sums carry no bias

// Register epilogue (bf16 in, bf16 out, no upsampling): no LDS round trip, no barrier.  The accumulators were
// initialised with the bias.  A lane holds 4 consecutive channels of one pixel per 16-channel block; pairs of
// blocks are exchanged between lane rows r, r^1 with v_permlane16_swap so that each lane owns 8 consecutive
// channels: the residual arrives as one 16-byte load per lane, is added in fp32 before the single bf16
// rounding, and the result leaves as one 16-byte store (64 contiguous bytes per pixel and instruction).
// Loads and stores go through buffer descriptors: lanes past the M or Cout edge get an out-of-range offset,
// which the hardware ignores, so there is no branch around them.  All residual loads of the wave's tile are
// issued before the first is used.
constexpr int CONV_REG_EPILOGUE = 1 << 26;

// Residual pieces [FROM, TO) of the wave's tile (piece = mi * NI/2 + pair): one 16-byte buffer load each.
// The main loop's last k-step issues the first RES_EARLY pieces (its loader registers are free by then), so
// that they arrive under its MFMAs; the epilogue issues the rest.
template <int TM, int TN, int MI, int NI, int FROM, int TO>
__device__ __forceinline__ void conv_residual_load(const ConvArgs &p, u32x4 (&rv)[MI][NI / 2], int m0, int n0, int lane, int wm, int wn) {
    constexpr int NP2 = NI / 2;
    const int frow = lane & 15, fq = lane >> 4;
    const auto rs_res = __builtin_amdgcn_make_buffer_rsrc((void *)p.res, 0, p.res_bytes, 0x00020000);
    const int cb = (fq & 1) ? 16 + (fq - 1) * 4 : fq * 4;
    const int nl = n0 + wn * TN + cb;
    constexpr unsigned OOB = 0xffffff00u;
#pragma unroll
    for (int i = FROM; i < TO; ++i) {
        const int mi = i / NP2, pr = i % NP2;
        const int gm = m0 + wm * TM + mi * 16 + frow;
        const bool ok = gm < p.M && nl + pr * 32 < p.Cout8;
        rv[mi][pr] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(
            rs_res, ok ? ((unsigned)gm * (unsigned)p.res_ld + (unsigned)nl) * 2u : OOB, pr * 64, 0));
    }
}

// RESK: -1 = the shortcut flag is tested at run time, 0 / 1 = the caller has branched on it already (one branch around
// the whole epilogue instead of one per piece: the compiler's wait-count pass follows the control-flow graph, not the
// condition, and otherwise sees a path on which the residual loads are issued and never waited for).
template <int TM, int TN, int MI, int NI, int RES_EARLY, int RESK = -1>
__device__ __forceinline__ void conv_epilogue_regs(const ConvArgs &p, f32x4 (&acc)[NI][MI], u32x4 (&rv_early)[MI][NI / 2], int m0, int n0, int lane, int wm, int wn) {
    static_assert(NI % 2 == 0, "register epilogue pairs 16-channel blocks");
    constexpr int NP2 = NI / 2;
    const int frow = lane & 15, fq = lane >> 4;
    const bool has_res = RESK < 0 ? (p.flags & Y3_CONV_RESIDUAL) != 0 : RESK != 0;
    const float slope = (p.flags & Y3_CONV_LEAKY) ? 0.1f : 1.0f;
    const auto rs_out = __builtin_amdgcn_make_buffer_rsrc((void *)p.out, 0, p.out_bytes, 0x00020000);
    const int cb = (fq & 1) ? 16 + (fq - 1) * 4 : fq * 4;
    const int nl = n0 + wn * TN + cb;                       // this lane's first channel in pair 0
    constexpr unsigned OOB = 0xffffff00u;
    u32x4 rv_own[MI][NP2];
    u32x4 (&rv)[MI][NP2] = RES_EARLY > 0 ? rv_early : rv_own;   // a caller's array costs the 256x256 kernel 65 spilled registers
    if (has_res) conv_residual_load<TM, TN, MI, NI, RES_EARLY, MI * NP2>(p, rv, m0, n0, lane, wm, wn);
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
        const int gm = m0 + wm * TM + mi * 16 + frow;
#pragma unroll
        for (int pr = 0; pr < NP2; ++pr) {
            const bool ok = gm < p.M && nl + pr * 32 < p.Cout8;
            const unsigned off = ok ? ((unsigned)gm * (unsigned)p.out_ld + (unsigned)nl) * 2u : OOB;
            f32x4 x = acc[2 * pr][mi], y = acc[2 * pr + 1][mi];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                x[j] = leaky_max(x[j], slope);
                y[j] = leaky_max(y[j], slope);
            }
            u32x4 o;
            if (has_res) {
                float lo[4], hi[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {              // after the swap: lo = channels 0..3, hi = 4..7 of the lane's 8
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
                o[0] = pack_bf16x2(lo[0], lo[1]); o[1] = pack_bf16x2(lo[2], lo[3]);
                o[2] = pack_bf16x2(hi[0], hi[1]); o[3] = pack_bf16x2(hi[2], hi[3]);
            } else {
                const unsigned x0 = pack_bf16x2(x[0], x[1]), x1 = pack_bf16x2(x[2], x[3]);
                const unsigned y0 = pack_bf16x2(y[0], y[1]), y1 = pack_bf16x2(y[2], y[3]);
                const auto s0 = __builtin_amdgcn_permlane16_swap(x0, y0, false, false);
                const auto s1 = __builtin_amdgcn_permlane16_swap(x1, y1, false, false);
                o[0] = s0[0]; o[1] = s1[0]; o[2] = s0[1]; o[3] = s1[1];
            }
            __builtin_amdgcn_raw_buffer_store_b128(o, rs_out, off, pr * 64, 0);
        }
    }
}

}  // namespace

// conv_pp.hip: the two-wave-group main loop for the 256 x 256 bf16 tile with the register epilogue; conv_igemm.hip's
// launcher dispatches to it (same packed weights, same sums in the same order: same bits).  With a.sk_slabs / a.sk_flags
// set and a launch that would leave a partial round it hands over to conv_pp_split.hip (the balanced split).
bool y3_conv_pp_ok(const ConvArgs &a, int n_pad, int rows);             // rows: 256 or 192 pixels per tile
int y3_conv_pp_launch(ConvArgs a, int n_pad, int rows, long cus, int phase_pct, hipStream_t stream);
long y3_conv_pp_slab_bytes(const ConvArgs &a, int n_pad, long cus);      // workspace the balanced split wants (0 = none)
int y3_conv_pp_split_launch(ConvArgs a, int n_pad, long cus, hipStream_t stream);
