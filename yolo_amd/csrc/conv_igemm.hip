// Fused convolution for gfx950 (MI355X): implicit GEMM on MFMA.
//
// Replaces tf.pad + tf.layers.conv2d + batch_normalization + leaky_relu
// (reference build_model.py:151-178), with tf.math.add (:224, shortcut) and
// resize_nearest_neighbor (:236, upsample) fused into the epilogue and
// route/concat (:191) expressed through in_ld/out_ld channel slices.
//
// GEMM view:  D[n][m] = sum_k Wp[n][k] * X[m][k]
//   m = (b, ho, wo) output pixel, n = output channel, k = (kh, kw, cin).
// The weights are the MFMA "A" operand and the gathered activations the "B"
// operand, so every lane ends up holding 4 consecutive output CHANNELS of one
// pixel: the accumulators go to LDS with 16-byte writes and leave it as whole
// NHWC pixel rows (16-byte, fully coalesced stores; residual loads likewise).
//
// Data movement: both operands are staged global -> LDS with
// global_load_lds_dwordx4 (no VGPR round trip) into a double buffer, one
// barrier per k-step.  A k-step is one 128-byte LDS row per pixel / channel
// (64 bf16 or 32 fp32); 16-byte chunks are XOR-swizzled by ((row >> 1) & 7) on
// the SOURCE address (LDS-DMA writes lane-linear) and on the fragment read, which
// makes the ds_read_b128 fragment reads bank-conflict free.  Zero padding comes
// from pointing the lane's source address at a zero page.
//
// What runs by default (conv_variant picks per layer; everything else below is a measured alternative that no
// layer selects, reachable only in the -DY3_EXPERIMENTS flavour of the library (libyolo3hip_exp.so,
// include/yolo3hip_experimental.h: y3_set_conv_tile) for A/B runs and the tile-equality tests -- results in
// DESIGN.md section 3.1.  The default library has no knob, no process-global state and none of the timing paths):
//   256  conv_igemm_kernel<256,256, 8 waves>   Cout >= 256, few rounds         persistent, register epilogue
//   192  conv_igemm_kernel<192,256, 8 waves>   same, when it saves a round     persistent, register epilogue
//   257  conv_igemm_ring_kernel<256,128,RING 3> 65..128 output channels (bf16)  two workgroups per CU
//   260  conv_igemm_kernel<256,64, 8 waves>    33..64 output channels (bf16)   two workgroups per CU
//   128 / 64 / 32  conv_igemm_kernel<128,BN>   small layers, fp32 mode         2-3 workgroups per CU
//   direct  conv_direct_kernel                 Cin = 3 (first layer) when the fused stem is not used
// Alternatives: 129 (128x256), 258 (256x128, 8 waves), 130/131 (128x128 ring), debug bit 4 (ring for 256/128).
// Debug bits (high half of y3_set_conv_tile, Y3_EXPERIMENTS only): 1/8/16/32 timing experiments that skip loads or
// the epilogue (results are garbage), 2 no interleave, 64 force the LDS epilogue, 128 one workgroup per tile.
// Measured and removed: cross-tile prefetch, non-temporal epilogue stores / residual loads.
#include "common.h"
#include "conv_common.h"
#include "decode_box.h"
#include <type_traits>


namespace {

#ifdef Y3_EXPERIMENTS
// Experiment knobs of libyolo3hip_exp.so (include/yolo3hip_experimental.h): plain process-global state, set by
// one thread between launches -- a test and measurement facility, not part of the product library.
int g_conv_tile = 0;                // forced tile, see y3_set_conv_tile
int g_conv_debug = 0;               // its high half: debug bits
int g_conv_cus_cap = 0;             // CUs a launch may occupy (0 = all of the device's: y3_set_conv_cus)
int g_conv_phase_pct = 50;          // start offset of the de-phased workgroups, percent of a tile period (0 = off)
#else
constexpr int g_conv_tile = 0, g_conv_debug = 0, g_conv_cus_cap = 0, g_conv_phase_pct = 50;
#endif

int conv_cus() {
    const int cus = y3_device_cus();
    return g_conv_cus_cap > 0 && g_conv_cus_cap < cus ? g_conv_cus_cap : cus;
}

// Workgroup barrier that orders LDS traffic only.  __syncthreads() also drains every outstanding global access
// (s_waitcnt vmcnt(0)): in the decode epilogue that would put the record stores, the candidate stores and a global
// atomic's round trip on the critical path of every pass.
__device__ __forceinline__ void lds_only_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

// Head decode fused into a head convolution's epilogue (y3_conv2d_head_decode): what y3_decode needs for ONE scale.
struct HeadDecode {
    const float *letterbox;              // [B][4]: scale, off_h, off_w (utils/utils.py:100-116)
    float *records;                      // [B][N][8]
    int *cand_idx, *cand_count, *class_count;
    int N, C, scale_base, store_heads;   // boxes per image (all scales), classes, first box index of this scale
    float stride_x, stride_y, thr;
    float anchor[3][2];
};

// LDS of the decode epilogue behind the staged tile: s_cnt[2] | s_hist[2][96] | s_base[2] | s_list[2][EP_ROWS * 3] (ints)
constexpr int dec_cnt_bytes(int ep_rows) { return (2 + 2 * 96 + 2 + 2 * ep_rows * 3) * 4; }

// Epilogue shared by the convolution kernels: accumulators -> LDS (fp32) -> bias, leaky, residual ->
// NHWC rows with 16-byte stores, EP_ROWS tile rows per pass.
template <bool BF16, int BM, int BN, int WAVES_M, int WAVES_N, int EP_ROWS, int MI, int NI, bool DECODE = false>
__device__ __forceinline__ void conv_epilogue(const ConvArgs &p, f32x4 (&acc)[NI][MI], char *smem, int m0, int n0,
                                              int tid, int wm, int wn, long out_shift = 0, const HeadDecode *hd = nullptr) {
    constexpr int NT = 64 * WAVES_M * WAVES_N;
    constexpr int ES = BF16 ? 2 : 4;
    constexpr int TM = BM / WAVES_M, TN = BN / WAVES_N;
    constexpr int CS_LD = BN + 4;
    const int lane = tid & 63;
    const int frow = lane & 15, fq = lane >> 4;
    float *Cs = (float *)smem;
    constexpr int NCH = BN / 8;
    constexpr int O_IT = EP_ROWS * NCH / NT;
    static_assert(EP_ROWS * NCH % NT == 0, "epilogue chunks vs threads");
    const bool leaky = p.flags & Y3_CONV_LEAKY;
    const bool has_res = p.flags & Y3_CONV_RESIDUAL;
    const bool ups = p.flags & Y3_CONV_UPS2X;
    const bool out_f32 = !BF16 || (p.flags & Y3_CONV_OUT_F32);
    for (int pass = 0; pass < BM / EP_ROWS; ++pass) {
        // residual operand: all of this pass's loads go out before the LDS round trip, so their
        // latency hides under it instead of serialising the store loop.
        u32x4 rv[O_IT][BF16 ? 1 : 2];
        if (has_res) {
#pragma unroll
            for (int i = 0; i < O_IT; ++i) {
                const int q = tid + NT * i;
                const int row = q / NCH, ch = q % NCH;
                const int gm = m0 + pass * EP_ROWS + row, n = n0 + ch * 8;
                const bool ok = gm < p.M && n < p.Cout8;
                const char *rp = ok ? p.res + ((long)gm * p.res_ld + n) * ES : p.zeros;
                rv[i][0] = *(const u32x4 *)rp;
                if constexpr (!BF16) rv[i][1] = *(const u32x4 *)(rp + 16);
            }
        }
        // (decode epilogue: per-box constants requested here, ahead of the LDS round trip, not after its barrier)
        [[maybe_unused]] int d_b = 0, d_cell = 0;
        [[maybe_unused]] float d_scale = 1.0f, d_offh = 0.0f, d_offw = 0.0f;
        [[maybe_unused]] bool d_live = false;
        if constexpr (DECODE) {
            const int row = tid / 3, gm = m0 + pass * EP_ROWS + row;
            d_live = tid < EP_ROWS * 3 && gm < p.M;
            const int gmc = d_live ? gm : min(m0 + pass * EP_ROWS, p.M - 1);
            d_b = fast_div(gmc, p.fd_howo);
            d_cell = gmc - d_b * p.HoWo;
            d_scale = hd->letterbox[d_b * 4 + 0]; d_offh = hd->letterbox[d_b * 4 + 1]; d_offw = hd->letterbox[d_b * 4 + 2];
        }
        if (DECODE && pass > 0) lds_only_barrier();    // (the previous pass's global stores and atomics stay in flight)
        else __syncthreads();                  // LDS tiles / previous pass fully consumed
        if constexpr (DECODE) {                // candidate counters of this pass (read last before the barrier above)
            int *cz = (int *)(smem + EP_ROWS * CS_LD * 4);
            if (tid < 2 + 2 * 96) cz[tid] = 0;
        }
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) {
            const int rb = wm * TM + mi * 16 - pass * EP_ROWS;       // wave-uniform
            if (rb < 0 || rb >= EP_ROWS) continue;
#pragma unroll
            for (int ni = 0; ni < NI; ++ni) {
                const int n = wn * TN + ni * 16 + fq * 4;
                f32x4 v = acc[ni][mi];
                if constexpr (DECODE) v += *(const f32x4 *)(p.bias + n0 + n);      // the decode below reads finished head values
                *(f32x4 *)(Cs + (rb + frow) * CS_LD + n) = v;
            }
        }
        if constexpr (DECODE) lds_only_barrier();
        else __syncthreads();
#pragma unroll
        for (int i = 0; i < O_IT; ++i) {
            if constexpr (DECODE) {
                if (!hd->store_heads) break;         // the caller did not ask for the head tensor itself
            }
            const int q = tid + NT * i;
            const int row = q / NCH, ch = q % NCH;
            const int gm = m0 + pass * EP_ROWS + row, n = n0 + ch * 8;
            if (gm >= p.M || n >= p.Cout8) continue;
            f32x4 v0 = *(const f32x4 *)(Cs + row * CS_LD + ch * 8);
            f32x4 v1 = *(const f32x4 *)(Cs + row * CS_LD + ch * 8 + 4);
            if (!DECODE && !(p.flags & CONV_NO_BIAS)) {         // split-K partial sums: the bias is added once, by the finishing kernel
                v0 += *(const f32x4 *)(p.bias + n);
                v1 += *(const f32x4 *)(p.bias + n + 4);
            }
            if (leaky) {
#pragma unroll
                for (int j = 0; j < 4; ++j) { v0[j] = leaky01(v0[j]); v1[j] = leaky01(v1[j]); }
            }
            if (has_res) {
                if constexpr (BF16) {
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        v0[2 * j] += bf16_to_f32(rv[i][0][j] & 0xffff);
                        v0[2 * j + 1] += bf16_to_f32(rv[i][0][j] >> 16);
                        v1[2 * j] += bf16_to_f32(rv[i][0][2 + j] & 0xffff);
                        v1[2 * j + 1] += bf16_to_f32(rv[i][0][2 + j] >> 16);
                    }
                } else {
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        v0[j] += __uint_as_float(rv[i][0][j]);
                        v1[j] += __uint_as_float(rv[i][1][j]);
                    }
                }
            }
            long opix = gm;
            long row2 = 0;
            if (ups) {
                const int b = fast_div(gm, p.fd_howo);
                const int r = gm - b * p.HoWo;
                const int ho = fast_div(r, p.fd_wo);
                const int wo = r - ho * p.Wo;
                row2 = 2L * p.Wo;
                opix = ((long)b * 2 * p.Ho + 2 * ho) * row2 + 2 * wo;
            }
            const int reps = ups ? 4 : 1;
            for (int rpt = 0; rpt < reps; ++rpt) {
                const long pix = opix + (rpt & 1) + (rpt >> 1) * row2;
                if (out_f32) {
                    float *op = (float *)p.out + out_shift + pix * p.out_ld + n;
                    *(f32x4 *)op = v0;
                    *(f32x4 *)(op + 4) = v1;
                } else {
                    u32x4 o;
                    o[0] = pack_bf16x2(v0[0], v0[1]);
                    o[1] = pack_bf16x2(v0[2], v0[3]);
                    o[2] = pack_bf16x2(v1[0], v1[1]);
                    o[3] = pack_bf16x2(v1[2], v1[3]);
                    *(u32x4 *)(p.out + (pix * p.out_ld + n) * 2) = o;
                }
            }
        }
        if constexpr (DECODE) {
            // ---- head decode of this pass's EP_ROWS grid cells x 3 anchors, straight from the staged fp32 values --------
            // (what decode_lane_kernel does after copying the head back from HBM: same per-box code, decode_box.h)
            int *const s_cnt = (int *)(smem + EP_ROWS * CS_LD * 4);             // [2]         (zeroed at the top of the pass)
            int *const s_hist = s_cnt + 2;                                      // [2][96]
            int *const s_base = s_hist + 2 * 96;                                // [2]
            int *const s_list = s_base + 2;                                     // [2][EP_ROWS * 3]
            const int C = hd->C, E = 5 + C;
            const int row0 = m0 + pass * EP_ROWS;
            if (row0 >= p.M) break;                                  // block-uniform
            const int b_lo = fast_div(row0, p.fd_howo);              // a pass spans at most two images (HoWo >= EP_ROWS)
            if (hd->store_heads) lds_only_barrier();                 // the head stores above have read Cs (decode writes into it)
            if ((tid & ~63) < EP_ROWS * 3) {                         // wave-uniform
                const int row = tid / 3, an = tid - row * 3;
                const bool live = d_live;
                const int b = d_b, cell = d_cell;
                const int gy = fast_div(cell, p.fd_wo), gx = cell - gy * p.Wo;
                float *tl = Cs + (live ? row : 0) * CS_LD + an * E;
                const DecodedBox o = y3_decode_box(tl, live, C, gx, gy, hd->stride_x, hd->stride_y, hd->anchor[an][0],
                                                   hd->anchor[an][1], d_scale, d_offh, d_offw);
                const int pbox = hd->scale_base + cell * 3 + an;
                if (live) {
                    float *r = hd->records + ((long)b * hd->N + pbox) * 8;
                    *(f32x4 *)r = f32x4{o.ymin, o.xmin, o.ymax, o.xmax};
                    *(f32x4 *)(r + 4) = f32x4{o.score, __uint_as_float(o.m0), __uint_as_float(o.m1), __uint_as_float(o.m2)};
                    if (o.score >= hd->thr) {                  // inference.py:127
                        const int slot = b - b_lo;
                        s_list[slot * (EP_ROWS * 3) + atomicAdd(&s_cnt[slot], 1)] = pbox;
                        unsigned mm[3] = {o.m0, o.m1, o.m2};
                        for (int w = 0; w < 3; ++w)
                            while (mm[w]) { const int bit = __ffs(mm[w]) - 1; mm[w] &= mm[w] - 1; atomicAdd(&s_hist[slot * 96 + w * 32 + bit], 1); }
                    }
                }
            }
            lds_only_barrier();                                      // lists and counts complete (the record stores stay in flight)
            if (tid < 2 && s_cnt[tid]) s_base[tid] = atomicAdd(&hd->cand_count[b_lo + tid], s_cnt[tid]);
            if (tid < 2 * 96) {
                const int slot = tid / 96, c = tid - slot * 96;
                const int h = s_hist[slot * 96 + c];
                if (c < C && h) atomicAdd(&hd->class_count[(b_lo + slot) * C + c], h);
            }
            lds_only_barrier();                                      // (the two lanes that own s_base waited for their atomics' results)
#pragma unroll
            for (int slot = 0; slot < 2; ++slot) {
                const int n = s_cnt[slot];
                for (int i = tid; i < n; i += NT)
                    hd->cand_idx[(long)(b_lo + slot) * hd->N + s_base[slot] + i] = s_list[slot * (EP_ROWS * 3) + i];
            }
        }
    }
}


// Waves per SIMD the launcher counts on (workgroups per CU by LDS x waves per workgroup / 4 SIMDs).  Without it
// the compiler budgets 512 registers for a 4-wave workgroup, and a kernel that drifts past 256 (VGPR + AGPR)
// silently halves its occupancy: measured 1.5x slower on the 128x128 tile.
constexpr int conv_waves_per_simd(int lds_bytes, int waves) {
    const int wg = lds_bytes > 80 * 1024 ? 1 : (lds_bytes > 53 * 1024 ? 2 : 3);
    return (wg * waves + 3) / 4;
}

template <bool BF16, int BM, int BN, int WAVES_M, int WAVES_N, int EP_ROWS, bool INTERLEAVE, bool DECODE = false>
__global__ __launch_bounds__(64 * WAVES_M * WAVES_N)
__attribute__((amdgpu_waves_per_eu(conv_waves_per_simd(2 * (BM + BN) * 128, WAVES_M * WAVES_N), conv_waves_per_simd(2 * (BM + BN) * 128, WAVES_M * WAVES_N))))
void conv_igemm_kernel(const ConvArgs p, const HeadDecode hd) {
    constexpr int NT = 64 * WAVES_M * WAVES_N;   // threads
    constexpr int ES = BF16 ? 2 : 4;             // element bytes
    constexpr int EPC = 16 / ES;                 // elements per 16-byte chunk
    constexpr int BKE = 128 / ES;                // elements per k-step
    constexpr int TM = BM / WAVES_M, TN = BN / WAVES_N;
    constexpr int MI = TM / 16, NI = TN / 16;
    constexpr int LROWS = NT / 8;                // tile rows one loader pass covers
    constexpr int A_IT = BM / LROWS, B_IT = BN / LROWS;
    constexpr int PIECE = NT * 16;               // bytes one loader pass writes
    constexpr int STAGE = (BM + BN) * 128;       // bytes per pipeline buffer
    static_assert(BM % LROWS == 0 && BN % LROWS == 0, "tile vs loader");
    static_assert(BM % EP_ROWS == 0 && EP_ROWS % 16 == 0 && TM % 16 == 0, "epilogue passes");

    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;

    // Workgroups are persistent when the grid is smaller than the tile count: tile = blockIdx.x, + gridDim.x, ...
    // (gridDim.x is then a multiple of 8, so a workgroup's tiles keep its XCD).  The next tile's set-up and
    // first loads overlap the previous tile's stores, which nothing waits for.
    // De-phasing.  Equal tiles keep every workgroup in lockstep: all of them reach the HBM-bound epilogue
    // together, the chip waits for HBM, then HBM idles through the next main loop.  The workgroups that have a
    // spare round (see the launcher) therefore start half a tile period late, so that their epilogues fall
    // into the others' main loops and the bursts are smaller.
    if (p.delay_ticks != 0u && (int)blockIdx.x >= p.delay_from) {
        const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
        while (__builtin_amdgcn_s_memrealtime() - t0 < p.delay_ticks) __builtin_amdgcn_s_sleep(64);
    }
    // ---- loader state: thread owns LDS chunk slot (tid&7) of rows (tid>>3) + LROWS*i --------------
    const int lrow = tid >> 3;
    const int kc = ((tid & 7) ^ ((tid >> 4) & 7)) * EPC;   // source chunk of this LDS slot
    // k position of this thread's chunk: tap = k / Cin, cin = k % Cin, advanced by BKE per k-step
    // without a division (step_tap = BKE / Cin, step_cin = BKE % Cin come from the host).
    const int tap0 = kc / p.Cin, cin0 = kc - tap0 * p.Cin;
    long a_base[A_IT];
    unsigned a_mask[A_IT];
    long w_off[B_IT];
    int tap = 0, cin = 0;
    int kt0 = 0;                                 // first k-step of this workgroup's share of the tile (split-K)
    auto tile_setup = [&](int tile, int &tm0, int &tn0) {
        int bid = xcd_remap(tile, p.total_tiles, p.flags & Y3_CONV_REVERSE);
        if (p.splits > 1) {                      // uniform; the splits of one tile are neighbours
            const int sp = bid % p.splits;
            bid /= p.splits;
            kt0 = sp * p.nk_split;
        }
        const int mt = bid / p.n_tiles, nt = bid - mt * p.n_tiles;
        tm0 = mt * BM;
        tn0 = nt * BN;
#pragma unroll
        for (int i = 0; i < A_IT; ++i) {
            const int gm = tm0 + lrow + LROWS * i;
            a_base[i] = 0;
            a_mask[i] = 0;
            if (gm < p.M) row_setup(p, gm, ES, a_base[i], a_mask[i]);
        }
#pragma unroll
        for (int i = 0; i < B_IT; ++i) w_off[i] = ((long)(tn0 + lrow + LROWS * i) * p.Kpad + kc) * ES;
        tap = tap0;
        cin = cin0;
        if (kt0 != 0) {
            const int k = kc + kt0 * BKE;
            tap = k / p.Cin;
            cin = k - tap * p.Cin;
        }
    };

    // One k-step of loads is NP = A_IT + B_IT LDS-DMA pieces (1 KiB per wave each).  An LDS-DMA
    // instruction holds the issuing wave for ~60-180 cycles, so the pieces are not issued in a
    // burst after the barrier (the MFMA pipe would idle behind it): they are spread between
    // the MFMAs of the first half of the k-step, and land during the second half.
    constexpr int NP = A_IT + B_IT;
    constexpr int NMF = NI * MI;                 // MFMA groups per 32-deep half step
    long toff = 0;
    int tapc = 0;
    auto stage_begin = [&]() {
        // a_mask has no bit at or above ks*ks, so padded k (tap >= ks*ks) reads the zero page too.
        const int th = p.ks == 3 ? (tap * 11) >> 5 : tap;          // tap / 3 for tap < 16
        const int tw = tap - th * p.ks;
        toff = ((long)(th * p.W + tw) * p.in_ld + cin) * ES;
        tapc = tap;
        tap += p.step_tap;
        cin += p.step_cin;
        const bool wrap = cin >= p.Cin;
        cin = wrap ? cin - p.Cin : cin;
        tap = wrap ? tap + 1 : tap;
    };
#ifdef Y3_EXPERIMENTS
    const bool skip_a = p.flags & (1 << 29), skip_b = p.flags & (1 << 28);      // timing experiments: pieces not issued
#endif
    auto piece = [&](int q, int kt, int buf) {
        char *base = smem + buf * STAGE + wave * 1024;
#ifdef Y3_EXPERIMENTS
        if (q < A_IT ? skip_a : skip_b) return;
#endif
        if (q < A_IT) {
            const char *cand = p.in + (a_base[q] + toff);
            const char *src = ((a_mask[q] >> tapc) & 1u) ? cand : p.zeros;
            glds16(src, base + q * PIECE);
        } else {
            glds16(p.w + w_off[q - A_IT] + (long)kt * 128, base + BM * 128 + (q - A_IT) * PIECE);
        }
    };

    int pb = 0;                                  // pipeline buffer the next k-step reads
    for (int tile = blockIdx.x; tile < p.total_tiles; tile += gridDim.x) {
    int m0, n0;
    tile_setup(tile, m0, n0);
    stage_begin();
#pragma unroll
    for (int q = 0; q < NP; ++q) piece(q, kt0, pb);         // free: every wave is past the barrier of the last k-step

    // ---- main loop ------------------------------------------------------------------------
    const bool reg_epi = BF16 && (p.flags & CONV_REG_EPILOGUE);     // then the bias is the accumulators' initial value
    f32x4 acc[NI][MI];
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) {
        f32x4 b0 = f32x4{0.f, 0.f, 0.f, 0.f};
        if (reg_epi) b0 = *(const f32x4 *)(p.bias + n0 + wn * TN + ni * 16 + (lane >> 4) * 4);
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) acc[ni][mi] = b0;
    }

    const int frow = lane & 15;
    const int fsw = (lane >> 1) & 7;
    const int fq = lane >> 4;
    const int nk_all = p.Kpad / BKE;
    const int nk = p.splits > 1 ? min(nk_all, kt0 + p.nk_split) : nk_all;     // one past this workgroup's last k-step

    // residual pieces the last k-step fetches ahead of the epilogue (register epilogue only)
    constexpr int RES_EARLY = (BF16 && NI % 2 == 0 && MI * NI * 4 <= 96) ? MI * (NI / 2) : 0;
    using frag_t = typename std::conditional<BF16, bf16x8, f32x4>::type;
    auto mfma_group = [&](int j, const frag_t (&xq)[MI], const frag_t (&wq)[NI]) {
        const int ni = j / MI, mi = j % MI;
        if constexpr (BF16) {
            acc[ni][mi] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wq[ni], xq[mi], acc[ni][mi], 0, 0, 0);
        } else {
            // lane group g holds k = 4g..4g+3 of this 16-wide k block: MFMA i sums k = 4g+i over g.
#pragma unroll
            for (int i = 0; i < 4; ++i)
                acc[ni][mi] = __builtin_amdgcn_mfma_f32_16x16x4f32(wq[ni][i], xq[mi][i], acc[ni][mi], 0, 0, 0);
        }
    };
    // piece q of the next k-step goes before MFMA group floor(q * NMF / NP) of the first half-step
    auto pieces_before = [&](int j, int kt) {
#pragma unroll
        for (int q = (j * NP + NMF - 1) / NMF; q < ((j + 1) * NP + NMF - 1) / NMF; ++q) {
            if constexpr (INTERLEAVE) __builtin_amdgcn_sched_barrier(0);
            piece(q, kt + 1, pb ^ 1);
            if constexpr (INTERLEAVE) __builtin_amdgcn_sched_barrier(0);
        }
    };
    auto kstep = [&](int kt, auto prefetch, auto early) {
        constexpr bool PF = decltype(prefetch)::value;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's LDS-DMA of k-step kt has landed
        __syncthreads();                                    // everyone's has; the other buffer is free
        early();
        if constexpr (PF) stage_begin();
        const char *As = smem + pb * STAGE;
        const char *Bs = As + BM * 128;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const int so = ((ks * 4 + fq) ^ fsw) * 16;
            frag_t xf[MI], wf[NI];
#pragma unroll
            for (int mi = 0; mi < MI; ++mi)
                xf[mi] = *(const frag_t *)(As + (wm * TM + mi * 16 + frow) * 128 + so);
#pragma unroll
            for (int ni = 0; ni < NI; ++ni)
                wf[ni] = *(const frag_t *)(Bs + (wn * TN + ni * 16 + frow) * 128 + so);
#pragma unroll
            for (int j = 0; j < NMF; ++j) {
                if constexpr (PF) {
                    if (ks == 0) pieces_before(j, kt);
                }
                mfma_group(j, xf, wf);
            }
        }
        pb ^= 1;
    };
    u32x4 rv[MI][NI % 2 == 0 ? NI / 2 : 1];
    auto res_early = [&] {
        if constexpr (RES_EARLY > 0) {
            if (reg_epi && (p.flags & Y3_CONV_RESIDUAL)) {
                conv_residual_load<TM, TN, MI, NI, 0, RES_EARLY>(p, rv, m0, n0, lane, wm, wn);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    };
    for (int kt = kt0; kt + 1 < nk; ++kt) kstep(kt, std::true_type{}, [] {});
    kstep(nk - 1, std::false_type{}, res_early);

#ifdef Y3_EXPERIMENTS
    if (p.flags & (1 << 27)) {          // timing experiment: no epilogue (one store keeps the accumulators live)
        float sacc = 0.f;
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) {
#pragma unroll
            for (int mi = 0; mi < MI; ++mi) sacc += acc[ni][mi][0] + acc[ni][mi][1] + acc[ni][mi][2] + acc[ni][mi][3];
        }
        if (sacc == 123.456f) *(float *)p.out = sacc;
        continue;
    }
#endif
    if constexpr (BF16) {
        if (reg_epi) {
            conv_epilogue_regs<TM, TN, MI, NI, RES_EARLY>(p, acc, rv, m0, n0, lane, wm, wn);
            continue;
        }
    }
    conv_epilogue<BF16, BM, BN, WAVES_M, WAVES_N, EP_ROWS, MI, NI, DECODE>(p, acc, smem, m0, n0, tid, wm, wn,
                                                                           p.splits > 1 ? (long)(kt0 / p.nk_split) * p.slice_elems : 0L, &hd);
    __syncthreads();                             // the LDS epilogue's staging area overlaps the pipeline buffers
    }
}

// ---- ring-pipelined variant -------------------------------------------------------------------
// Same GEMM view and epilogue, different main loop: the LDS holds a ring of RING sub-tiles,
// each one 32 bf16 (16 fp32) deep = one 64-byte row per pixel / channel.  LDS-DMA loads run
// RING-1 sub-steps ahead and are retired with a COUNTED s_waitcnt vmcnt(N) + a raw s_barrier, so
// loads stay in flight across barriers instead of being drained once per k-step (the 2-stage
// kernel above spends ~40 % of its wave cycles in that drain: SQ_WAIT_ANY, profiles/).
// 64-byte rows: four rows share a 256-byte bank row, chunk c of row r sits at slot
// c ^ (3 * ((r >> 2) & 1)), which keeps the ds_read_b128 fragment reads conflict free.
template <bool BF16, int BM, int BN, int WAVES_M, int WAVES_N, int EP_ROWS, int RING>
__global__ __launch_bounds__(64 * WAVES_M * WAVES_N)
__attribute__((amdgpu_waves_per_eu(conv_waves_per_simd(RING * (BM + BN) * 64, WAVES_M * WAVES_N), conv_waves_per_simd(RING * (BM + BN) * 64, WAVES_M * WAVES_N))))
void conv_igemm_ring_kernel(const ConvArgs p) {
    constexpr int NT = 64 * WAVES_M * WAVES_N;
    constexpr int ES = BF16 ? 2 : 4;
    constexpr int EPC = 16 / ES;
    constexpr int BKS = 64 / ES;                 // elements per sub-step
    constexpr int TM = BM / WAVES_M, TN = BN / WAVES_N;
    constexpr int MI = TM / 16, NI = TN / 16;
    constexpr int LROWS = NT / 4;                // tile rows one loader pass covers (4 chunks a row)
    constexpr int A_IT = BM / LROWS, B_IT = BN / LROWS;
    constexpr int NP = A_IT + B_IT;              // LDS-DMA pieces per thread per sub-step
    constexpr int PIECE = NT * 16;
    constexpr int SLOT = (BM + BN) * 64;         // bytes per ring slot
    constexpr int NMF = NI * MI;
    static_assert(BM % LROWS == 0 && BN % LROWS == 0 && LROWS % 16 == 0, "tile vs loader");
    static_assert(RING >= 3 && NP * (RING - 2) < 64, "ring depth vs vmcnt range");

    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;

    const int bid = xcd_remap(blockIdx.x, gridDim.x, p.flags & Y3_CONV_REVERSE);
    const int mt = bid / p.n_tiles, nt = bid - mt * p.n_tiles;
    const int m0 = mt * BM, n0 = nt * BN;

    // loader: thread owns LDS slot (tid&3) of rows (tid>>2) + LROWS*i
    const int lrow = tid >> 2;
    const int kc = ((tid & 3) ^ (3 * ((tid >> 4) & 1))) * EPC;

    long a_base[A_IT];
    unsigned a_mask[A_IT];
#pragma unroll
    for (int i = 0; i < A_IT; ++i) {
        const int gm = m0 + lrow + LROWS * i;
        a_base[i] = 0;
        a_mask[i] = 0;
        if (gm < p.M) row_setup(p, gm, ES, a_base[i], a_mask[i]);
    }
    long w_off[B_IT];
#pragma unroll
    for (int i = 0; i < B_IT; ++i) w_off[i] = ((long)(n0 + lrow + LROWS * i) * p.Kpad + kc) * ES;

    int tap = kc / p.Cin;
    int cin = kc - tap * p.Cin;
    long toff = 0;
    int tapc = 0;
    auto stage_begin = [&]() {
        const int th = p.ks == 3 ? (tap * 11) >> 5 : tap;
        const int tw = tap - th * p.ks;
        toff = ((long)(th * p.W + tw) * p.in_ld + cin) * ES;
        tapc = tap;
        tap += p.sub_tap;
        cin += p.sub_cin;
        const bool wrap = cin >= p.Cin;
        cin = wrap ? cin - p.Cin : cin;
        tap = wrap ? tap + 1 : tap;
    };
    auto piece = [&](int q, int s, char *slot) {           // sub-tile s -> ring slot
        char *base = slot + wave * 1024;
        if (q < A_IT) {
            const char *cand = p.in + (a_base[q] + toff);
            const char *src = ((a_mask[q] >> tapc) & 1u) ? cand : p.zeros;
            glds16(src, base + q * PIECE);
        } else {
            glds16(p.w + w_off[q - A_IT] + (long)s * 64, base + BM * 64 + (q - A_IT) * PIECE);
        }
    };

    const bool reg_epi = BF16 && (p.flags & CONV_REG_EPILOGUE);     // then the bias is the accumulators' initial value
    f32x4 acc[NI][MI];
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) {
        f32x4 b0 = f32x4{0.f, 0.f, 0.f, 0.f};
        if (reg_epi) b0 = *(const f32x4 *)(p.bias + n0 + wn * TN + ni * 16 + (lane >> 4) * 4);
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) acc[ni][mi] = b0;
    }

    const int frow = lane & 15;
    const int fq = lane >> 4;
    const int so = (fq ^ (3 * ((lane >> 2) & 1))) * 16;     // this lane's chunk slot in a 64-byte row
    const int ns = p.Kpad / BKS;

    // prologue: sub-tiles 0 .. RING-2 in flight
#pragma unroll
    for (int s = 0; s < RING - 1; ++s) {
        if (s < ns) {
            stage_begin();
#pragma unroll
            for (int q = 0; q < NP; ++q) piece(q, s, smem + s * SLOT);
        }
    }

    int rd = 0;                 // ring slot of sub-tile s
    int wr = RING - 1;          // ring slot sub-tile s + RING - 1 goes to
    for (int s = 0; s < ns; ++s) {
        // this wave's pieces of sub-tile s have landed once at most the younger sub-tiles' remain in flight
        // (RING - 2 of them in the steady state, fewer at the tail)
        const int ahead = ns - 1 - s;
        constexpr auto wait_enc = [](int v) { return (v & 15) | (7 << 4) | (15 << 8) | ((v >> 4) << 14); };
        if (ahead >= RING - 2) __builtin_amdgcn_s_waitcnt(wait_enc(NP * (RING - 2)));
        else if (RING > 3 && ahead == 1) __builtin_amdgcn_s_waitcnt(wait_enc(NP));
        else __builtin_amdgcn_s_waitcnt(wait_enc(0));
        asm volatile("" ::: "memory");
        __builtin_amdgcn_s_barrier();      // everyone's pieces of s landed; slot wr (read in s-1) is free
        const bool pf = s + RING - 1 < ns;
        if (pf) stage_begin();
        const char *As = smem + rd * SLOT;
        const char *Bs = As + BM * 64;
        char *wslot = smem + wr * SLOT;
        using frag_t = typename std::conditional<BF16, bf16x8, f32x4>::type;
        frag_t xf[MI], wf[NI];
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) xf[mi] = *(const frag_t *)(As + (wm * TM + mi * 16 + frow) * 64 + so);
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) wf[ni] = *(const frag_t *)(Bs + (wn * TN + ni * 16 + frow) * 64 + so);
#pragma unroll
        for (int j = 0; j < NMF; ++j) {
#pragma unroll
            for (int q = 0; q < NP; ++q)
                if (j == (q * NMF) / NP) {
                    __builtin_amdgcn_sched_barrier(0);
                    if (pf) piece(q, s + RING - 1, wslot);
                    __builtin_amdgcn_sched_barrier(0);
                }
            const int ni = j / MI, mi = j % MI;
            if constexpr (BF16) {
                acc[ni][mi] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[ni], xf[mi], acc[ni][mi], 0, 0, 0);
            } else {
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    acc[ni][mi] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[ni][i], xf[mi][i], acc[ni][mi], 0, 0, 0);
            }
        }
        rd = rd + 1 == RING ? 0 : rd + 1;
        wr = wr + 1 == RING ? 0 : wr + 1;
    }
    if constexpr (BF16) {
        if (reg_epi) {
            u32x4 rv[MI][NI / 2];
            conv_epilogue_regs<TM, TN, MI, NI, 0>(p, acc, rv, m0, n0, lane, wm, wn);
            return;
        }
    }
    conv_epilogue<BF16, BM, BN, WAVES_M, WAVES_N, EP_ROWS, MI, NI>(p, acc, smem, m0, n0, tid, wm, wn);
}

// ---- direct convolution: first layer (Cin = 3) and any Cin the chunked loader cannot take ----
// One thread per output pixel, CT output channels per thread, sequential fp32 fmaf over k.
// Weights are fp32 [k][n_pad] so that a k-row of CT weights is a wave-uniform (scalar) load.
struct DirectArgs {
    const char *in;
    const float *w, *bias;
    char *out;
    int H, W, Cin, in_ld, Ho, Wo, HoWo, Cout8, out_ld, ks, stride, pad, n_pad, M, flags;
};

template <bool IN_BF16, bool OUT_BF16, int CT>
__global__ __launch_bounds__(256) void conv_direct_kernel(const DirectArgs p) {
    const int gm = blockIdx.x * 256 + threadIdx.x;
    const int nb = blockIdx.y * CT;
    if (gm >= p.M) return;
    const int b = gm / p.HoWo;
    const int r = gm - b * p.HoWo;
    const int ho = r / p.Wo;
    const int wo = r - ho * p.Wo;
    const int hi0 = ho * p.stride - p.pad, wi0 = wo * p.stride - p.pad;
    float acc[CT];
#pragma unroll
    for (int j = 0; j < CT; ++j) acc[j] = 0.0f;
    for (int kh = 0; kh < p.ks; ++kh)
        for (int kw = 0; kw < p.ks; ++kw) {
            const int hi = hi0 + kh, wi = wi0 + kw;
            const bool ok = (unsigned)hi < (unsigned)p.H && (unsigned)wi < (unsigned)p.W;
            const long pix = ok ? ((long)b * p.H + hi) * p.W + wi : 0;
            const float *wr = p.w + (long)(kh * p.ks + kw) * p.Cin * p.n_pad + nb;
            for (int c = 0; c < p.Cin; ++c) {
                float x;
                if constexpr (IN_BF16) x = bf16_to_f32(((const unsigned short *)p.in)[pix * p.in_ld + c]);
                else x = ((const float *)p.in)[pix * p.in_ld + c];
                x = ok ? x : 0.0f;
#pragma unroll
                for (int j = 0; j < CT; ++j) acc[j] = fmaf(x, wr[j], acc[j]);
                wr += p.n_pad;
            }
        }
    const bool leaky = p.flags & Y3_CONV_LEAKY;
#pragma unroll
    for (int j = 0; j < CT; ++j) {
        acc[j] += p.bias[nb + j];
        if (leaky) acc[j] = leaky01(acc[j]);
    }
#pragma unroll
    for (int j = 0; j < CT; j += 8) {
        if (nb + j >= p.Cout8) break;
        if constexpr (OUT_BF16) {
            u32x4 o;
            o[0] = pack_bf16x2(acc[j], acc[j + 1]);
            o[1] = pack_bf16x2(acc[j + 2], acc[j + 3]);
            o[2] = pack_bf16x2(acc[j + 4], acc[j + 5]);
            o[3] = pack_bf16x2(acc[j + 6], acc[j + 7]);
            *(u32x4 *)(p.out + ((long)gm * p.out_ld + nb + j) * 2) = o;
        } else {
            float *op = (float *)p.out + (long)gm * p.out_ld + nb + j;
            *(f32x4 *)op = f32x4{acc[j], acc[j + 1], acc[j + 2], acc[j + 3]};
            *(f32x4 *)(op + 4) = f32x4{acc[j + 4], acc[j + 5], acc[j + 6], acc[j + 7]};
        }
    }
}

// First layer + the 2x2 stride-2 max pool that follows it (YOLOv3-tiny: conv 3->16 @416, maxpool), Y3_CONV_POOL2:
// one thread per POOLED pixel and 16 output channels.  The 4x4xCIN input patch is loaded once; each of the four
// convolution outputs under the pool window is the same sequential fmaf chain over k as in conv_direct_kernel
// (+ bias, leaky), so max-then-round equals the unfused round-then-max bit for bit (rounding is monotonic).
// The full-resolution activation (the largest tensor of the network) never reaches HBM.
template <bool OUT_BF16, int CIN>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2)))      // 258 registers would halve the occupancy
void conv_direct_pool_kernel(const DirectArgs p) {
    constexpr int CT = 16;
    const int Wp = p.Wo >> 1, HpWp = (p.Ho >> 1) * Wp;
    const int nb = blockIdx.y * CT;
    // this block's 27 x CT weights in LDS: as wave-uniform scalar loads the unrolled loop below keeps all 432 live
    // and the compiler spills them lane by lane (1 800 v_readlane per thread); broadcast ds_reads cost nothing like that
    __shared__ __attribute__((aligned(16))) float wsm[9 * CIN * CT];
    for (int i = threadIdx.x; i < 9 * CIN * CT; i += 256) wsm[i] = p.w[(long)(i / CT) * p.n_pad + nb + (i % CT)];
    __syncthreads();
    const int gm = blockIdx.x * 256 + threadIdx.x;           // pooled pixel
    if (gm >= p.M) return;
    const int b = gm / HpWp;
    const int r = gm - b * HpWp;
    const int hp = r / Wp;
    const int wp = r - hp * Wp;
    const int hi0 = 2 * hp - 1, wi0 = 2 * wp - 1;
    float patch[4][4][CIN];
#pragma unroll
    for (int dy = 0; dy < 4; ++dy)
#pragma unroll
        for (int dx = 0; dx < 4; ++dx) {
            const int hi = hi0 + dy, wi = wi0 + dx;
            const bool ok = (unsigned)hi < (unsigned)p.H && (unsigned)wi < (unsigned)p.W;
            const long pix = ok ? ((long)b * p.H + hi) * p.W + wi : 0;
#pragma unroll
            for (int c = 0; c < CIN; ++c) {
                const float x = ((const float *)p.in)[pix * p.in_ld + c];
                patch[dy][dx][c] = ok ? x : 0.0f;
            }
        }
    float acc[4][CT];
#pragma unroll
    for (int o = 0; o < 4; ++o)
#pragma unroll
        for (int j = 0; j < CT; ++j) acc[o][j] = 0.0f;
#pragma unroll
    for (int kh = 0; kh < 3; ++kh)
#pragma unroll
        for (int kw = 0; kw < 3; ++kw)
#pragma unroll
            for (int c = 0; c < CIN; ++c) {
                const float *wr = wsm + ((kh * 3 + kw) * CIN + c) * CT;
#pragma unroll
                for (int o = 0; o < 4; ++o) {
                    const float x = patch[(o >> 1) + kh][(o & 1) + kw][c];
#pragma unroll
                    for (int j = 0; j < CT; ++j) acc[o][j] = fmaf(x, wr[j], acc[o][j]);
                }
            }
    const bool leaky = p.flags & Y3_CONV_LEAKY;
    float best[CT];
#pragma unroll
    for (int j = 0; j < CT; ++j) {
        const float bj = p.bias[nb + j];
#pragma unroll
        for (int o = 0; o < 4; ++o) {
            float v = acc[o][j] + bj;
            if (leaky) v = leaky01(v);
            best[j] = o == 0 ? v : fmaxf(best[j], v);
        }
    }
#pragma unroll
    for (int j = 0; j < CT; j += 8) {
        if (nb + j >= p.Cout8) break;
        if constexpr (OUT_BF16) {
            u32x4 o;
            o[0] = pack_bf16x2(best[j], best[j + 1]);
            o[1] = pack_bf16x2(best[j + 2], best[j + 3]);
            o[2] = pack_bf16x2(best[j + 4], best[j + 5]);
            o[3] = pack_bf16x2(best[j + 6], best[j + 7]);
            *(u32x4 *)(p.out + ((long)gm * p.out_ld + nb + j) * 2) = o;
        } else {
            float *op = (float *)p.out + (long)gm * p.out_ld + nb + j;
            *(f32x4 *)op = f32x4{best[j], best[j + 1], best[j + 2], best[j + 3]};
            *(f32x4 *)(op + 4) = f32x4{best[j + 4], best[j + 5], best[j + 6], best[j + 7]};
        }
    }
}

template <bool BF16, int BM, int BN, int WM, int WN, int EP_ROWS, bool IL, bool DEC = false>
int launch_igemm_(ConvArgs a, int n_pad, hipStream_t stream, const HeadDecode *hd = nullptr) {
    constexpr int stage_bytes = 2 * (BM + BN) * 128;
    constexpr int cs_bytes = EP_ROWS * (BN + 4) * 4 + (DEC ? dec_cnt_bytes(EP_ROWS) : 0);     // (+ the decode epilogue's candidate lists)
    constexpr int lds = stage_bytes > cs_bytes ? stage_bytes : cs_bytes;
    static_assert(lds <= 160 * 1024, "LDS budget");
    auto kern = conv_igemm_kernel<BF16, BM, BN, WM, WN, EP_ROWS, IL, DEC>;
    a.n_tiles = n_pad / BN;
    static thread_local int attr_dev = -1;
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (attr_dev != dev) {
        if (hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess) {
            y3_set_error("conv: cannot reserve %d bytes of LDS", lds);
            return Y3_E_LAUNCH;
        }
        attr_dev = dev;
    }
    const int m_tiles = (a.M + BM - 1) / BM;
    const long nblk = (long)m_tiles * a.n_tiles * (a.splits > 1 ? a.splits : 1);
    if (nblk <= 0 || nblk > 0x7fffffffL) { y3_set_error("conv: grid of %ld blocks", nblk); return Y3_E_INVALID; }
    a.total_tiles = (int)nblk;
    // persistent grid: as many workgroups as stay resident (LDS-limited), each walking several tiles
    const long cus = conv_cus();
    const long resident = cus * (lds > 80 * 1024 ? 1 : (lds > 53 * 1024 ? 2 : 3));
    const long grid = (g_conv_debug & 128) || nblk <= resident ? nblk : resident;
    a.delay_ticks = 0;
    a.delay_from = 0;
    // De-phase only where it is free and measured to pay: one workgroup per CU (the 256- and 192-row tiles),
    // several rounds, and a last round that is not full -- the workgroups with one tile fewer to do start half
    // a tile period late (76x76 3x3 layers: -6 %).  Delaying workgroups that have no spare round, or one of a
    // CU's two co-resident workgroups of the 128-row tiles, measured neutral to slightly slower.
    const long rem = nblk % grid;
    if (grid == resident && resident == cus && nblk >= 4 * resident && rem != 0 && g_conv_phase_pct > 0) {
        const double tile_us = 2.0 * BM * BN * (double)a.Kpad / (BF16 ? 4.8e6 : 0.3e6);      // at ~4.8 TFLOP/s per CU
        a.delay_ticks = (unsigned)(tile_us * g_conv_phase_pct);                              // 100 ticks per us, percent
        a.delay_from = (int)rem;
    }
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(64 * WM * WN), lds, stream, a, hd ? *hd : HeadDecode{});
    return y3_check_launch("conv_igemm");
}


template <bool BF16, int BM, int BN, int WM, int WN, int EP_ROWS, int RING = 4>
int launch_ring(ConvArgs a, int n_pad, hipStream_t stream) {
    constexpr int ring_bytes = RING * (BM + BN) * 64;
    constexpr int cs_bytes = EP_ROWS * (BN + 4) * 4;
    constexpr int lds = ring_bytes > cs_bytes ? ring_bytes : cs_bytes;
    static_assert(lds <= 160 * 1024, "LDS budget");
    auto kern = conv_igemm_ring_kernel<BF16, BM, BN, WM, WN, EP_ROWS, RING>;
    a.n_tiles = n_pad / BN;
    static thread_local int attr_dev = -1;
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (attr_dev != dev) {
        if (hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess) {
            y3_set_error("conv: cannot reserve %d bytes of LDS", lds);
            return Y3_E_LAUNCH;
        }
        attr_dev = dev;
    }
    const int m_tiles = (a.M + BM - 1) / BM;
    const long nblk = (long)m_tiles * a.n_tiles;
    if (nblk <= 0 || nblk > 0x7fffffffL) { y3_set_error("conv: grid of %ld blocks", nblk); return Y3_E_INVALID; }
    a.total_tiles = (int)nblk;
    a.delay_ticks = 0; a.delay_from = 0;
    // with the register epilogue the LDS staging area of the other epilogue is not needed: more workgroups per CU
    const int lds_used = (a.flags & CONV_REG_EPILOGUE) ? ring_bytes : lds;
    hipLaunchKernelGGL(kern, dim3((unsigned)nblk), dim3(64 * WM * WN), lds_used, stream, a);
    return y3_check_launch("conv_igemm_ring");
}

template <bool BF16, int BM, int BN, int WM, int WN, int EP_ROWS>
int launch_igemm(const ConvArgs &a, int n_pad, hipStream_t stream, const HeadDecode *hd = nullptr) {
    if constexpr (BF16 && BN == 256 && BM >= 192) {      // the tiles that hold all 255 head channels of their pixels
        if (hd) return launch_igemm_<BF16, BM, BN, WM, WN, EP_ROWS, true, true>(a, n_pad, stream, hd);
    }
    if (hd) { y3_set_error("conv: this tile has no fused head decode"); return Y3_E_INVALID; }
    if (g_conv_debug & 2) return launch_igemm_<BF16, BM, BN, WM, WN, EP_ROWS, false>(a, n_pad, stream);
    return launch_igemm_<BF16, BM, BN, WM, WN, EP_ROWS, true>(a, n_pad, stream);
}

// Tile choice -> 256 (256x256, 8 waves), 192 (192x256), 128 (128x128, 2 workgroups per CU), 64 (128x64),
// 32 (128x32).  L2 -> LDS bytes per FLOP fall with the tile area, so the big tiles win whenever they fill
// the chip; what decides between them is wave quantisation: rounds = ceil(tiles / resident workgroups),
// priced by the measured relative cost of one tile (profiles/r01_layer_report_v3_*: a 192x256 tile costs
// 0.80 of a 256x256 one, a 128x128 one 0.55 with two of them resident per CU).
// (A workspace for conv_pp_kernel's balanced split does not change the choice: measured inside the network the split
// costs more than the partial round it removes -- profiles/r03l -- so it never tips the scale towards the 256 x 256 tile.)
inline int conv_variant(const y3_conv_desc *d, long M) {
    const int bn = d->out_c > 64 ? 128 : (d->out_c > 32 ? 64 : 32);
    if (bn == 64 && d->dtype == Y3_BF16) {
        // 33..64 output channels: 256 pixels x 64 channels on 8 waves, two workgroups per CU (half the tiles, hence
        // half the per-tile set-up and weight re-reads of the 128x64 tile: -7 % on 3x3 32->64 @304); 261 = ring form
        if (g_conv_tile == 64 || g_conv_tile == 261) return g_conv_tile;
        return M >= 256L * 512 ? 260 : 64;
    }
    if (bn != 128) return bn;
    int tile = g_conv_tile;
    if (tile == 0) {
        tile = 128;
        // 65..128 output channels, bf16, enough tiles for two workgroups per CU: the 256-pixel x 128-channel ring
        // kernel (half the weight re-reads of the 128x128 tile; measured -3..-6 % on the 1x1 256->128 and
        // 3x3 64->128 layers once its occupancy was pinned to two workgroups per CU)
        if (d->n_pad == 128 && d->dtype == Y3_BF16 && M >= 256L * 256) tile = 257;
        if (d->n_pad % 256 == 0) {
            const long cus = conv_cus();
            auto rounds = [&](long bm, long bnn, long per_cu) {
                const long tiles = ((M + bm - 1) / bm) * (d->n_pad / bnn);
                return (tiles + cus * per_cu - 1) / (cus * per_cu);
            };
            const long c256 = rounds(256, 256, 1) * 256, c192 = rounds(192, 256, 1) * 205, c128 = rounds(128, 128, 2) * 142;
            tile = (c256 <= c192 && c256 <= c128) ? 256 : (c192 <= c128 ? 192 : 128);
        }
    }
    if ((tile >= 192 || tile == 129) && tile != 257 && tile != 258 && d->n_pad % 256 != 0) tile = 128;
    return tile;
}

// which launches of the 256 x 256 tile take conv_pp.hip's main loop (y3_conv_pp_ok has the per-launch conditions)
// (returns the pixel rows of conv_pp_kernel's tile, 0 = not a launch for it; 512 / 704 = y3_set_conv_tile's codes that force it)
inline int conv_pp_rows(int v) {
    if ((v == 256 && g_conv_tile == 0 && g_conv_debug == 0) || (v == 512 && g_conv_tile == 512)) return 256;
    if ((v == 192 && g_conv_tile == 0 && g_conv_debug == 0) || (v == 704 && g_conv_tile == 704)) return 192;
    return 0;
}

inline int bn_for(int out_c) { return out_c > 64 ? 128 : (out_c > 32 ? 64 : 32); }

bool use_direct(const y3_conv_desc *d) {
    const int epc = d->dtype == Y3_BF16 ? 8 : 4;
    if (d->flags & Y3_CONV_IN_F32 && d->dtype == Y3_BF16) return true;
    return (d->in_c % epc) != 0;
}

}  // namespace

extern "C" int y3_conv_weight_layout(y3_conv_desc *d) {
    Y3_REQUIRE(d != nullptr, Y3_E_INVALID, "conv: null descriptor");
    Y3_REQUIRE(d->dtype == Y3_F32 || d->dtype == Y3_BF16, Y3_E_INVALID, "conv: dtype %d", d->dtype);
    Y3_REQUIRE(d->ksize == 1 || d->ksize == 3, Y3_E_INVALID, "conv: ksize %d (1 or 3)", d->ksize);
    Y3_REQUIRE(d->in_c > 0 && d->out_c > 0, Y3_E_INVALID, "conv: channels %d -> %d", d->in_c, d->out_c);
    const int k = d->ksize * d->ksize * d->in_c;
    if (use_direct(d)) {
        d->k_pad = k;
        d->n_pad = (d->out_c + 31) / 32 * 32;
        return 1;
    }
    const int bke = d->dtype == Y3_BF16 ? 64 : 32;
    const int bn = bn_for(d->out_c);
    d->k_pad = (k + bke - 1) / bke * bke;
    d->n_pad = (d->out_c + bn - 1) / bn * bn;
    return 0;
}

// splits > 1: the launch writes `splits` fp32 partial sums per output to `workspace` (no bias, activation,
// shortcut or upsampling: conv_splitk_finish_kernel applies those once).
static int conv_launch(const y3_conv_desc *d, const void *in, const void *w, const float *bias,
                       const void *residual, void *out, const void *zeros, hipStream_t stream, float *workspace, int splits,
                       const HeadDecode *hd = nullptr, void *pp_slabs = nullptr, long pp_slab_bytes = 0, unsigned *pp_flags = nullptr) {
    Y3_REQUIRE(d && in && w && bias && out && zeros, Y3_E_INVALID, "conv: null pointer argument");
    y3_conv_desc chk = *d;
    const int family = y3_conv_weight_layout(&chk);
    if (family < 0) return family;
    Y3_REQUIRE(chk.k_pad == d->k_pad && chk.n_pad == d->n_pad, Y3_E_INVALID,
               "conv: k_pad/n_pad (%d,%d) do not match the layout (%d,%d)", d->k_pad, d->n_pad, chk.k_pad, chk.n_pad);
    Y3_REQUIRE(d->stride == 1 || d->stride == 2, Y3_E_INVALID, "conv: stride %d", d->stride);
    // the reference pads top/left for any stride > 1 (build_model.py:151-154): a 1x1 stride-2 conv would sample rows
    // -1, 1, 3, ... -- not what this kernel's pad = 0 does, and no shipped cfg has one
    Y3_REQUIRE(!(d->ksize == 1 && d->stride != 1), Y3_E_INVALID, "conv: 1x1 with stride %d unsupported", d->stride);
    Y3_REQUIRE(d->batch > 0 && d->in_h > 0 && d->in_w > 0, Y3_E_INVALID, "conv: empty input");
    const int pad = d->ksize == 3 ? 1 : 0;
    const int exp_h = (d->in_h + pad - d->ksize) / d->stride + 1 + (d->stride == 1 ? pad : 0);
    const int exp_w = (d->in_w + pad - d->ksize) / d->stride + 1 + (d->stride == 1 ? pad : 0);
    Y3_REQUIRE(d->out_h == exp_h && d->out_w == exp_w, Y3_E_INVALID, "conv: output %dx%d, expected %dx%d",
               d->out_h, d->out_w, exp_h, exp_w);
    const bool bf16 = d->dtype == Y3_BF16;
    const bool out_f32 = !bf16 || (d->flags & Y3_CONV_OUT_F32);
    const int es = bf16 ? 2 : 4, oes = out_f32 ? 4 : 2;
    const int cout8 = (d->out_c + 7) / 8 * 8;
    Y3_REQUIRE(d->in_ld >= d->in_c && d->out_ld >= cout8, Y3_E_INVALID, "conv: in_ld %d / out_ld %d too small", d->in_ld, d->out_ld);
    Y3_REQUIRE(aligned16(out) && (d->out_ld * oes) % 16 == 0, Y3_E_ALIGN, "conv: output not 16-byte aligned");
    const long M = (long)d->batch * d->out_h * d->out_w;
    Y3_REQUIRE(M < 0x7fffffffL, Y3_E_INVALID, "conv: %ld output pixels", M);
    if (d->flags & Y3_CONV_RESIDUAL) {
        Y3_REQUIRE(residual != nullptr, Y3_E_INVALID, "conv: residual flag without pointer");
        Y3_REQUIRE(aligned16(residual) && (d->res_ld * es) % 16 == 0 && d->res_ld >= cout8, Y3_E_ALIGN, "conv: residual not aligned");
        Y3_REQUIRE(!(d->flags & Y3_CONV_UPS2X), Y3_E_INVALID, "conv: residual + upsample unsupported");
    }

    if (family == 1) {
        Y3_REQUIRE(!(d->flags & (Y3_CONV_RESIDUAL | Y3_CONV_UPS2X)), Y3_E_INVALID, "conv(direct): residual/upsample unsupported");
        DirectArgs a;
        if (d->flags & Y3_CONV_POOL2) {        // conv 3x3 s1 (Cin = 3, fp32 input) + maxpool 2x2 s2 in one launch; `out` is the pooled tensor
            const bool in_f32 = !bf16 || (d->flags & Y3_CONV_IN_F32);
            Y3_REQUIRE(d->ksize == 3 && d->stride == 1 && d->in_c == 3 && in_f32 && d->out_h % 2 == 0 && d->out_w % 2 == 0, Y3_E_INVALID,
                       "conv(pool2): needs 3x3 stride 1, 3 fp32 input channels and even output size (got k%d s%d c%d %dx%d)",
                       d->ksize, d->stride, d->in_c, d->out_h, d->out_w);
            a.in = (const char *)in; a.w = (const float *)w; a.bias = bias; a.out = (char *)out;
            a.H = d->in_h; a.W = d->in_w; a.Cin = d->in_c; a.in_ld = d->in_ld;
            a.Ho = d->out_h; a.Wo = d->out_w; a.HoWo = d->out_h * d->out_w; a.Cout8 = cout8; a.out_ld = d->out_ld;
            a.ks = 3; a.stride = 1; a.pad = 1; a.n_pad = d->n_pad; a.M = (int)(M / 4); a.flags = d->flags;
            dim3 grid((unsigned)((a.M + 255) / 256), (unsigned)((cout8 + 15) / 16));
            if (out_f32) hipLaunchKernelGGL((conv_direct_pool_kernel<false, 3>), grid, dim3(256), 0, stream, a);
            else hipLaunchKernelGGL((conv_direct_pool_kernel<true, 3>), grid, dim3(256), 0, stream, a);
            return y3_check_launch("conv_direct_pool");
        }
        a.in = (const char *)in; a.w = (const float *)w; a.bias = bias; a.out = (char *)out;
        a.H = d->in_h; a.W = d->in_w; a.Cin = d->in_c; a.in_ld = d->in_ld;
        a.Ho = d->out_h; a.Wo = d->out_w; a.HoWo = d->out_h * d->out_w; a.Cout8 = cout8; a.out_ld = d->out_ld;
        a.ks = d->ksize; a.stride = d->stride; a.pad = pad; a.n_pad = d->n_pad; a.M = (int)M; a.flags = d->flags;
        const bool in_bf16 = bf16 && !(d->flags & Y3_CONV_IN_F32);
        const int ct = (d->n_pad % 32 == 0 && d->out_c > 16) ? 32 : 16;
        dim3 grid((unsigned)((M + 255) / 256), (unsigned)((cout8 + ct - 1) / ct));
#define Y3_DIRECT(IB, OB, CT) hipLaunchKernelGGL((conv_direct_kernel<IB, OB, CT>), grid, dim3(256), 0, stream, a)
        if (ct == 32) {
            if (in_bf16) { if (out_f32) Y3_DIRECT(true, false, 32); else Y3_DIRECT(true, true, 32); }
            else         { if (out_f32) Y3_DIRECT(false, false, 32); else Y3_DIRECT(false, true, 32); }
        } else {
            if (in_bf16) { if (out_f32) Y3_DIRECT(true, false, 16); else Y3_DIRECT(true, true, 16); }
            else         { if (out_f32) Y3_DIRECT(false, false, 16); else Y3_DIRECT(false, true, 16); }
        }
#undef Y3_DIRECT
        return y3_check_launch("conv_direct");
    }

    Y3_REQUIRE(aligned16(in) && (d->in_ld * es) % 16 == 0, Y3_E_ALIGN, "conv: input not 16-byte aligned");
    Y3_REQUIRE(aligned16(w), Y3_E_ALIGN, "conv: weights not 16-byte aligned");
    Y3_REQUIRE(aligned16(bias), Y3_E_ALIGN, "conv: bias not 16-byte aligned");
    ConvArgs a;
    a.in = (const char *)in; a.w = (const char *)w; a.res = (const char *)residual; a.zeros = (const char *)zeros;
    a.bias = bias; a.out = (char *)out;
    a.H = d->in_h; a.W = d->in_w; a.Cin = d->in_c; a.in_ld = d->in_ld;
    a.Ho = d->out_h; a.Wo = d->out_w; a.HoWo = d->out_h * d->out_w; a.Cout8 = cout8;
    a.out_ld = d->out_ld; a.res_ld = d->res_ld;
    a.ks = d->ksize; a.stride = d->stride; a.pad = pad; a.Kpad = d->k_pad; a.M = (int)M; a.flags = d->flags;
#ifdef Y3_EXPERIMENTS
    if (g_conv_debug & 1) a.H = 0;             // timing experiment: every A tap reads the zero page
    if (g_conv_debug & 8) a.flags |= 1 << 29;  // timing experiment: no A loads at all (garbage results)
    if (g_conv_debug & 16) a.flags |= 1 << 28; // timing experiment: no B loads at all (garbage results)
    if (g_conv_debug & 32) a.flags |= 1 << 27; // timing experiment: no epilogue (no results)
#endif
    const int bke = bf16 ? 64 : 32;
    a.n_tiles = 0;
    a.step_tap = bke / d->in_c;
    a.step_cin = bke % d->in_c;
    a.fd_howo = make_fastdiv(a.HoWo);
    a.fd_wo = make_fastdiv(a.Wo);
    a.sub_tap = (bke / 2) / d->in_c;
    a.sub_cin = (bke / 2) % d->in_c;
    int v = conv_variant(d, M);
    a.splits = 1; a.nk_split = 0; a.slice_elems = 0;
    if (splits > 1) {
        const int nk = d->k_pad / bke;
        a.splits = splits;
        a.nk_split = (nk + splits - 1) / splits;
        a.slice_elems = M * (long)d->n_pad;
        a.out = (char *)workspace; a.out_ld = d->n_pad; a.Cout8 = d->n_pad; a.res = nullptr;
        a.flags = (d->flags & (Y3_CONV_IN_F32 | Y3_CONV_REVERSE)) | Y3_CONV_OUT_F32 | CONV_NO_BIAS;
        v = bn_for(d->out_c);                   // the 128-row tiles (conv_splitk_splits only splits those)
    }
    {   // register epilogue: bf16 -> bf16 without upsampling, tensors within a 4 GiB buffer descriptor
        const long ob = M * (long)d->out_ld * 2, rb = M * (long)d->res_ld * 2;
        a.out_bytes = a.res_bytes = 0;
        if (splits <= 1 && bf16 && !out_f32 && !(d->flags & Y3_CONV_UPS2X) && ob < 0xffff0000L && rb < 0xffff0000L && !(g_conv_debug & 64)) {
            a.flags |= CONV_REG_EPILOGUE;
            a.out_bytes = (unsigned)ob;
            a.res_bytes = (unsigned)rb;
        }
    }
    // 3x3 32 -> 64 on a large map (Darknet-53 layer 3): weights in registers, input patch resident in LDS (conv_c32.hip)
    if (!hd && splits <= 1 && g_conv_tile == 0 && g_conv_debug == 0 && (a.flags & CONV_REG_EPILOGUE) &&
        y3_conv_c32_ok(d, M * (long)d->out_ld * 2, (d->flags & Y3_CONV_RESIDUAL) ? M * (long)d->res_ld * 2 : 0L))
        return y3_conv_c32_launch(d, in, w, bias, residual, out, zeros, stream);
    // 3x3 stride 2, 64 -> 128 on a large map (Darknet-53 layer 5): the same recipe (conv_s2c64.hip)
    if (!hd && splits <= 1 && g_conv_tile == 0 && g_conv_debug == 0 && (a.flags & CONV_REG_EPILOGUE) &&
        y3_conv_s2c64_ok(d, M * (long)d->out_ld * 2))
        return y3_conv_s2c64_launch(d, in, w, bias, out, zeros, stream);
    // 1x1 512 -> 256 / 256 -> 128 on large launches: the whole weight matrix in registers (conv_1x1w.hip)
    if (!hd && splits <= 1 && g_conv_tile == 0 && g_conv_debug == 0 && (a.flags & CONV_REG_EPILOGUE) &&
        y3_conv_1x1w_ok(d, M * (long)d->out_ld * 2))
        return y3_conv_1x1w_launch(d, in, w, bias, out, zeros, stream);
    if (hd) {                                      // fused head decode: only the two tiles that carry it
        Y3_REQUIRE(bf16 && !(a.flags & CONV_REG_EPILOGUE) && d->n_pad == 256, Y3_E_INVALID, "conv_head_decode: not a bf16 -> fp32 head launch");
        if (v == 256) return launch_igemm<true, 256, 256, 2, 4, 128>(a, d->n_pad, stream, hd);
        return launch_igemm<true, 192, 256, 2, 4, 96>(a, d->n_pad, stream, hd);
    }
    const bool ring = g_conv_debug & 4;            // ring-pipelined kernels (experiment switch)
    if (ring && v == 192) v = 256;
    if (ring && (v == 256 || v == 128)) {
        if (bf16) return v == 256 ? launch_ring<true, 256, 256, 2, 4, 128>(a, d->n_pad, stream)
                                  : launch_ring<true, 128, 128, 2, 2, 128>(a, d->n_pad, stream);
        return v == 256 ? launch_ring<false, 256, 256, 2, 4, 128>(a, d->n_pad, stream)
                        : launch_ring<false, 128, 128, 2, 2, 128>(a, d->n_pad, stream);
    }
    // the 256 x 256 bf16 tile with the register epilogue runs on the ping-pong main loop (conv_pp.hip): same bits
    // (y3_set_conv_tile(256) keeps the one-barrier-per-k-step kernel for A/B runs and the tile-equality tests, 512 forces this one)
    if (bf16 && splits <= 1 && !hd && conv_pp_rows(v) && y3_conv_pp_ok(a, d->n_pad, conv_pp_rows(v))) {
        a.sk_slabs = nullptr;
        a.sk_flags = nullptr;
        if (pp_slabs != nullptr && conv_pp_rows(v) == 256) {        // the caller's workspace for the balanced split (y3_conv2d_fused_ws)
            const long need = y3_conv_pp_slab_bytes(a, d->n_pad, conv_cus());
            Y3_REQUIRE(pp_slab_bytes >= need, Y3_E_INVALID, "conv_ws: workspace of %ld bytes, %ld needed", pp_slab_bytes, need);
            Y3_REQUIRE(pp_flags != nullptr && aligned16(pp_slabs) && aligned16(pp_flags), Y3_E_ALIGN, "conv_ws: workspace / flags null or not 16-byte aligned");
            if (need > 0) { a.sk_slabs = (char *)pp_slabs; a.sk_flags = pp_flags; }
        }
        return y3_conv_pp_launch(a, d->n_pad, conv_pp_rows(v), conv_cus(), g_conv_phase_pct, stream);
    }
    if (v == 512) v = 256;
    if (v == 704) v = 192;
    if (bf16) {
        if (v == 257) return launch_ring<true, 256, 128, 2, 2, 128, 3>(a, d->n_pad, stream);
        if (v == 256) return launch_igemm<true, 256, 256, 2, 4, 128>(a, d->n_pad, stream);
        if (v == 129) return launch_igemm<true, 128, 256, 2, 4, 64>(a, d->n_pad, stream);     // experiment
        if (v == 258) return launch_igemm<true, 256, 128, 4, 2, 128>(a, d->n_pad, stream);    // 256 pixels x 128 channels, 8 waves
        if (v == 130) return launch_ring<true, 128, 128, 2, 2, 128, 3>(a, d->n_pad, stream);   // 48 KB: 3 per CU
        if (v == 131) return launch_ring<true, 128, 128, 2, 2, 128, 4>(a, d->n_pad, stream);   // 64 KB: 2 per CU, 3 sub-steps ahead
        if (v == 192) return launch_igemm<true, 192, 256, 2, 4, 96>(a, d->n_pad, stream);
        if (v == 128) return launch_igemm<true, 128, 128, 2, 2, 128>(a, d->n_pad, stream);
        if (v == 260) return launch_igemm<true, 256, 64, 4, 2, 128>(a, d->n_pad, stream);     // 256 pixels x 64 channels, 8 waves
        if (v == 261) return launch_ring<true, 256, 64, 2, 2, 128, 3>(a, d->n_pad, stream);    // same tile, ring, 4 waves
        if (v == 64) return launch_igemm<true, 128, 64, 2, 2, 128>(a, d->n_pad, stream);
        return launch_igemm<true, 128, 32, 4, 1, 128>(a, d->n_pad, stream);
    }
    if (v == 257) return launch_ring<false, 256, 128, 2, 2, 128, 3>(a, d->n_pad, stream);
    if (v == 256) return launch_igemm<false, 256, 256, 2, 4, 128>(a, d->n_pad, stream);
    if (v == 192) return launch_igemm<false, 192, 256, 2, 4, 96>(a, d->n_pad, stream);
    if (v == 128) return launch_igemm<false, 128, 128, 2, 2, 128>(a, d->n_pad, stream);
    if (v == 64) return launch_igemm<false, 128, 64, 2, 2, 128>(a, d->n_pad, stream);
    return launch_igemm<false, 128, 32, 4, 1, 128>(a, d->n_pad, stream);
}

extern "C" int y3_conv2d_fused(const y3_conv_desc *d, const void *in, const void *w, const float *bias,
                               const void *residual, void *out, const void *zeros, void *stream) {
    return conv_launch(d, in, w, bias, residual, out, zeros, (hipStream_t)stream, nullptr, 1);
}

// ---- the same convolution with a caller-owned workspace for conv_pp_kernel's balanced split ------------------------
// Whole 256 x 256 tiles on 256 persistent workgroups leave a partial last round (5.64 -> 6 at 76 x 76, batch 64; 1.42 -> 2
// at 19 x 19).  With a workspace the k-steps of an XCD group's tiles are dealt evenly instead, a cut tile's head and tail
// summed in order by two workgroups through an fp32 slab (conv_pp.hip): same bits, no partial round.
extern "C" int y3_conv_workspace_bytes(const y3_conv_desc *d, int64_t *slab_bytes) {
    Y3_REQUIRE(d && slab_bytes, Y3_E_INVALID, "conv_workspace_bytes: null pointer argument");
    *slab_bytes = 0;
    y3_conv_desc chk = *d;
    if (y3_conv_weight_layout(&chk) != 0 || chk.n_pad != d->n_pad || chk.k_pad != d->k_pad) return Y3_OK;
    if (d->dtype != Y3_BF16 || (d->flags & (Y3_CONV_OUT_F32 | Y3_CONV_UPS2X)) || d->out_h <= 0 || d->out_w <= 0) return Y3_OK;
    const long M = (long)d->batch * d->out_h * d->out_w;
    if (M >= 0x7fffffffL || conv_pp_rows(conv_variant(d, M)) != 256) return Y3_OK;
    ConvArgs a{};
    a.flags = CONV_REG_EPILOGUE; a.splits = 1; a.Kpad = d->k_pad; a.Cin = d->in_c; a.M = (int)M;
    a.HoWo = d->out_h * d->out_w; a.H = d->in_h; a.W = d->in_w; a.in_ld = d->in_ld;
    const long ob = M * (long)d->out_ld * 2, rb = M * (long)d->res_ld * 2;
    if (ob < 0xffff0000L && rb < 0xffff0000L && y3_conv_pp_ok(a, d->n_pad, 256)) *slab_bytes = y3_conv_pp_slab_bytes(a, d->n_pad, conv_cus());
    return Y3_OK;
}

extern "C" int y3_conv2d_fused_ws(const y3_conv_desc *d, const void *in, const void *w, const float *bias, const void *residual,
                                  void *out, const void *zeros, void *slabs, int64_t slab_bytes, void *flags, void *stream) {
    return conv_launch(d, in, w, bias, residual, out, zeros, (hipStream_t)stream, nullptr, 1, nullptr, slabs, (long)slab_bytes,
                       (unsigned *)flags);
}

// ---- a head convolution with the YOLO decode of its scale in the epilogue -----------------------------------------
// The three head convolutions (1x1 -> 3 * (5 + C) channels, linear, fp32 out) are the producers of everything
// y3_decode reads; un-fused, the fp32 heads (497 MB at 608, batch 64) are written and read straight back.  Here the
// tile's finished fp32 values are decoded where they are staged for the store (LDS), with decode_lane_kernel's own
// per-box code (decode_box.h: identical records), and the head tensor itself is written only if the caller wants it.
namespace {
bool head_decode_ok(const y3_conv_desc *d, const y3_decode_desc *dd, int scale, const char **why) {
    *why = nullptr;
    if (d->dtype != Y3_BF16 || !(d->flags & Y3_CONV_OUT_F32)) *why = "bf16 convolution with fp32 output required";
    else if (d->flags & (Y3_CONV_LEAKY | Y3_CONV_RESIDUAL | Y3_CONV_UPS2X | Y3_CONV_POOL2 | Y3_CONV_IN_F32)) *why = "head convolutions are linear and plain";
    else if (scale < 0 || scale >= dd->num_scales) *why = "scale index out of range";
    else if (dd->num_anchors != 3 || dd->num_classes < 1 || dd->num_classes > 80 || d->out_c != 3 * (5 + dd->num_classes)) *why = "3 anchors x (5 + classes <= 80) output channels required";
    else if (d->n_pad != 256) *why = "one 256-channel tile must hold all head channels";
    else if (dd->grid_h[scale] != d->out_h || dd->grid_w[scale] != d->out_w || dd->batch != d->batch) *why = "decode grid and convolution output differ";
    else if (d->out_h * d->out_w < 128) *why = "grid smaller than an epilogue pass";
    if (*why) return false;
    // the decode lives in the 256- and 192-pixel tiles; a launch the heuristic would give 128-pixel tiles runs on the
    // 192-pixel one instead (same sums, same bits), which only pays when it still fills a good part of the chip
    if ((long)d->batch * d->out_h * d->out_w < 8192) { *why = "launch too small for the 256 / 192-pixel tiles"; return false; }
    return true;
}
}  // namespace

extern "C" int y3_conv_head_decode_ok(const y3_conv_desc *d, const y3_decode_desc *dd, int scale) {
    if (!d || !dd) return 0;
    y3_conv_desc chk = *d;
    if (y3_conv_weight_layout(&chk) != 0 || chk.n_pad != d->n_pad || chk.k_pad != d->k_pad) return 0;
    const char *why;
    return head_decode_ok(d, dd, scale, &why) ? 1 : 0;
}

extern "C" int y3_conv2d_head_decode(const y3_conv_desc *d, const void *in, const void *w, const float *bias, void *out,
                                     const void *zeros, const y3_decode_desc *dd, int scale, const float *letterbox,
                                     float *records, int32_t *cand_idx, int32_t *cand_count, int32_t *class_count, void *stream) {
    Y3_REQUIRE(d && dd && letterbox && records && cand_idx && cand_count && class_count && zeros, Y3_E_INVALID, "conv_head_decode: null pointer");
    const char *why;
    Y3_REQUIRE(head_decode_ok(d, dd, scale, &why), Y3_E_INVALID, "conv_head_decode: %s", why);
    Y3_REQUIRE(aligned16(records), Y3_E_ALIGN, "conv_head_decode: records not 16-byte aligned");
    HeadDecode h;
    h.letterbox = letterbox; h.records = records; h.cand_idx = cand_idx; h.cand_count = cand_count; h.class_count = class_count;
    h.C = dd->num_classes;
    int base = 0, total = 0;
    for (int l = 0; l < dd->num_scales; ++l) {
        if (l == scale) base = total;
        total += dd->grid_h[l] * dd->grid_w[l] * dd->num_anchors;
    }
    Y3_REQUIRE(total == dd->boxes_per_image, Y3_E_INVALID, "conv_head_decode: boxes_per_image %d, grids give %d", dd->boxes_per_image, total);
    h.N = total; h.scale_base = base; h.store_heads = out != nullptr;
    h.stride_x = dd->stride_x[scale]; h.stride_y = dd->stride_y[scale]; h.thr = dd->score_threshold;
    for (int a = 0; a < 3; ++a) { h.anchor[a][0] = dd->anchors[scale][a][0]; h.anchor[a][1] = dd->anchors[scale][a][1]; }
    // (without a head tensor the kernel stores nothing through `out`; the zero page only satisfies the launcher's checks)
    return conv_launch(d, in, w, bias, nullptr, out ? out : (void *)zeros, zeros, (hipStream_t)stream, nullptr, 1, &h);
}

// ---- split-K: small launches (few output tiles, long k) --------------------------------------------------------
// At batch 1 a 19x19 layer is 24 tiles of 72 k-steps: 24 busy CUs, 50 us.  Split over k, `splits` workgroups per
// tile write fp32 partial sums to a caller-owned workspace [splits][M][n_pad]; conv_splitk_finish_kernel adds
// them in a fixed order (deterministic), then bias, leaky-ReLU, shortcut, store (2x2 when upsampling).
namespace {

int conv_splitk_splits(const y3_conv_desc *d) {
    y3_conv_desc chk = *d;
    if (y3_conv_weight_layout(&chk) != 0) return 1;                  // direct-convolution family
    const long M = (long)d->batch * d->out_h * d->out_w;
    const int bn = bn_for(d->out_c);
    const long tiles = ((M + 127) / 128) * (d->n_pad / bn);
    const int nk = d->k_pad / (d->dtype == Y3_BF16 ? 64 : 32);
    if (tiles >= 160 || nk < 8) return 1;                            // enough workgroups already, or nothing to split
    // Measured model (batch 1, 608): an un-split tile takes nk * t_step; split s ways it takes (nk / s) * t_step
    // + the partial sums' round trip through the workspace (s * M * n_pad * 8 bytes at ~3 TB/s) + ~6 us for the
    // second launch.  Take the best s; split only if that wins by 20 %.
    const double t_step = d->dtype == Y3_BF16 ? 0.76 : 1.8;          // us per k-step of a 128-row tile
    const double a = nk * t_step, b = (double)M * d->n_pad * 8.0 / 3.0e6;
    int best = 1;
    double t_best = a * 0.8;
    for (int s = 2; s <= 16 && s <= nk / 4 && tiles * s <= 1024; ++s) {
        const int per = (nk + s - 1) / s;
        if ((nk + per - 1) / per != s) continue;                     // would leave an empty split
        const double t = per * t_step + b * s + 6.0;
        if (t < t_best) { t_best = t; best = s; }
    }
    return best;
}

struct FinishArgs {
    const float *ws, *bias;
    const char *res;
    char *out;
    long M, slice_elems;
    int splits, n_pad, cout8, out_ld, res_ld, Ho, Wo, flags, bf16;
};

__global__ __launch_bounds__(256) void conv_splitk_finish_kernel(const FinishArgs p) {
    const int nch = p.cout8 / 8;
    const long q = (long)blockIdx.x * 256 + threadIdx.x;
    if (q >= p.M * nch) return;
    const long gm = q / nch;
    const int n = (int)(q - gm * nch) * 8;
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = 0.0f;
    for (int s = 0; s < p.splits; ++s) {                             // fixed order: deterministic
        const float *w = p.ws + s * p.slice_elems + gm * p.n_pad + n;
        const f32x4 a = *(const f32x4 *)w, b = *(const f32x4 *)(w + 4);
#pragma unroll
        for (int j = 0; j < 4; ++j) { v[j] += a[j]; v[4 + j] += b[j]; }
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        v[j] += p.bias[n + j];
        if (p.flags & Y3_CONV_LEAKY) v[j] = leaky01(v[j]);
    }
    if (p.flags & Y3_CONV_RESIDUAL) {
        if (p.bf16) {
            const u32x4 r = *(const u32x4 *)(p.res + (gm * p.res_ld + n) * 2);
#pragma unroll
            for (int j = 0; j < 4; ++j) { v[2 * j] += bf16_to_f32(r[j] & 0xffff); v[2 * j + 1] += bf16_to_f32(r[j] >> 16); }
        } else {
            const float *r = (const float *)p.res + gm * p.res_ld + n;
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] += r[j];
        }
    }
    const bool ups = p.flags & Y3_CONV_UPS2X;
    long opix = gm, row2 = 0;
    if (ups) {
        const long hw = (long)p.Ho * p.Wo;
        const long b = gm / hw, r = gm - b * hw;
        const long ho = r / p.Wo, wo = r - ho * p.Wo;
        row2 = 2L * p.Wo;
        opix = (b * 2 * p.Ho + 2 * ho) * row2 + 2 * wo;
    }
    const bool out_f32 = !p.bf16 || (p.flags & Y3_CONV_OUT_F32);
    for (int rpt = 0; rpt < (ups ? 4 : 1); ++rpt) {
        const long pix = opix + (rpt & 1) + (rpt >> 1) * row2;
        if (out_f32) {
            float *op = (float *)p.out + pix * p.out_ld + n;
            *(f32x4 *)op = f32x4{v[0], v[1], v[2], v[3]};
            *(f32x4 *)(op + 4) = f32x4{v[4], v[5], v[6], v[7]};
        } else {
            u32x4 o;
#pragma unroll
            for (int j = 0; j < 4; ++j) o[j] = pack_bf16x2(v[2 * j], v[2 * j + 1]);
            *(u32x4 *)(p.out + (pix * p.out_ld + n) * 2) = o;
        }
    }
}

}  // namespace

extern "C" int y3_conv_splitk_plan(const y3_conv_desc *d, int32_t *splits, int64_t *workspace_bytes) {
    Y3_REQUIRE(d && splits && workspace_bytes, Y3_E_INVALID, "conv_splitk_plan: null pointer argument");
    const int s = conv_splitk_splits(d);
    *splits = s;
    *workspace_bytes = s > 1 ? (int64_t)s * d->batch * d->out_h * d->out_w * d->n_pad * 4 : 0;
    return Y3_OK;
}

extern "C" int y3_conv2d_splitk(const y3_conv_desc *d, const void *in, const void *w, const float *bias,
                                const void *residual, void *out, const void *zeros, void *workspace,
                                int64_t workspace_bytes, void *stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    Y3_REQUIRE(d != nullptr, Y3_E_INVALID, "conv: null descriptor");
    const int s = conv_splitk_splits(d);
    if (s <= 1) return conv_launch(d, in, w, bias, residual, out, zeros, stream, nullptr, 1);
    const long M = (long)d->batch * d->out_h * d->out_w;
    Y3_REQUIRE(workspace && workspace_bytes >= (int64_t)s * M * d->n_pad * 4, Y3_E_INVALID,
               "conv_splitk: workspace of %ld bytes, %ld needed", (long)workspace_bytes, (long)s * M * d->n_pad * 4);
    Y3_REQUIRE(aligned16(workspace), Y3_E_ALIGN, "conv_splitk: workspace not 16-byte aligned");
    const int rc = conv_launch(d, in, w, bias, residual, out, zeros, stream, (float *)workspace, s);
    if (rc) return rc;
    const bool bf16 = d->dtype == Y3_BF16;
    FinishArgs f;
    f.ws = (const float *)workspace; f.bias = bias; f.res = (const char *)residual; f.out = (char *)out;
    f.M = M; f.slice_elems = M * (long)d->n_pad; f.splits = s; f.n_pad = d->n_pad;
    f.cout8 = (d->out_c + 7) / 8 * 8; f.out_ld = d->out_ld; f.res_ld = d->res_ld;
    f.Ho = d->out_h; f.Wo = d->out_w; f.flags = d->flags; f.bf16 = bf16 ? 1 : 0;
    const long work = M * (f.cout8 / 8);
    hipLaunchKernelGGL(conv_splitk_finish_kernel, dim3((unsigned)((work + 255) / 256)), dim3(256), 0, stream, f);
    return y3_check_launch("conv_splitk_finish");
}

// Which kernel a descriptor runs on, as rocprof will name it (for benches and profiles).
extern "C" int y3_conv_kernel_name(const y3_conv_desc *d, char *buf, int len) {
    Y3_REQUIRE(d && buf && len > 0, Y3_E_INVALID, "conv_kernel_name: bad arguments");
    y3_conv_desc chk = *d;
    const int family = y3_conv_weight_layout(&chk);
    if (family < 0) return family;
    const char *t = d->dtype == Y3_BF16 ? "true" : "false";
    if (family == 1) {
        snprintf(buf, (size_t)len, (d->flags & Y3_CONV_POOL2) ? "conv_direct_pool_kernel" : "conv_direct_kernel");
        return Y3_OK;
    }
    const long M = (long)d->batch * d->out_h * d->out_w;
    if (d->dtype == Y3_BF16 && g_conv_tile == 0 && g_conv_debug == 0 && d->out_ld > 0 &&
        y3_conv_c32_ok(d, M * (long)d->out_ld * 2, (d->flags & Y3_CONV_RESIDUAL) ? M * (long)d->res_ld * 2 : 0L)) {
        snprintf(buf, (size_t)len, "conv3x3_c32_kernel");
        return Y3_OK;
    }
    if (d->dtype == Y3_BF16 && g_conv_tile == 0 && g_conv_debug == 0 && d->out_ld > 0 && y3_conv_s2c64_ok(d, M * (long)d->out_ld * 2)) {
        snprintf(buf, (size_t)len, "conv3x3s2_c64_kernel");
        return Y3_OK;
    }
    if (d->dtype == Y3_BF16 && g_conv_tile == 0 && g_conv_debug == 0 && d->out_ld > 0 && y3_conv_1x1w_ok(d, M * (long)d->out_ld * 2)) {
        snprintf(buf, (size_t)len, "conv1x1_regw_kernel<%d, %d>", d->in_c, d->out_c);
        return Y3_OK;
    }
    const int v = conv_variant(d, M);
    if (d->dtype == Y3_BF16 && conv_pp_rows(v) && !(d->flags & (Y3_CONV_OUT_F32 | Y3_CONV_UPS2X)) && d->out_ld > 0) {
        ConvArgs a{};                                   // what conv_launch would hand to y3_conv_pp_ok
        a.flags = CONV_REG_EPILOGUE; a.splits = 1; a.Kpad = d->k_pad; a.Cin = d->in_c; a.M = (int)M;
        a.HoWo = d->out_h * d->out_w; a.H = d->in_h; a.W = d->in_w; a.in_ld = d->in_ld;
        const long ob = M * (long)d->out_ld * 2, rb = M * (long)d->res_ld * 2;
        if (ob < 0xffff0000L && rb < 0xffff0000L && y3_conv_pp_ok(a, d->n_pad, conv_pp_rows(v))) {
            snprintf(buf, (size_t)len, "conv_pp_kernel<%d>", conv_pp_rows(v) / 64);
            return Y3_OK;
        }
    }
    const char *k = (g_conv_debug & 4) && v >= 128 ? "conv_igemm_ring_kernel" : "conv_igemm_kernel";
    if (v == 257) snprintf(buf, (size_t)len, "conv_igemm_ring_kernel<%s, 256, 128, 2, 2", t);
    else if (v == 129) snprintf(buf, (size_t)len, "%s<%s, 128, 256, 2, 4", k, t);
    else if (v == 258) snprintf(buf, (size_t)len, "conv_igemm_kernel<%s, 256, 128, 4, 2", t);
    else if (v == 260) snprintf(buf, (size_t)len, "conv_igemm_kernel<%s, 256, 64, 4, 2", t);
    else if (v == 261) snprintf(buf, (size_t)len, "conv_igemm_ring_kernel<%s, 256, 64, 2, 2", t);
    else if (v == 130 || v == 131) snprintf(buf, (size_t)len, "conv_igemm_ring_kernel<%s, 128, 128, 2, 2", t);
    else if (v == 256 || v == 512) snprintf(buf, (size_t)len, "%s<%s, 256, 256, 2, 4", k, t);
    else if (v == 192 || v == 704) snprintf(buf, (size_t)len, "%s<%s, 192, 256, 2, 4", k, t);
    else if (v == 128) snprintf(buf, (size_t)len, "%s<%s, 128, 128, 2, 2", k, t);
    else if (v == 64) snprintf(buf, (size_t)len, "%s<%s, 128, 64, 2, 2", k, t);
    else snprintf(buf, (size_t)len, "%s<%s, 128, 32, 4, 1", k, t);
    return Y3_OK;
}

#ifdef Y3_EXPERIMENTS
// ---- libyolo3hip_exp.so only (include/yolo3hip_experimental.h) ----
// Experiment knob: force the conv tile (0 = heuristic).
// Experiment knob: how many CUs one convolution launch may occupy (persistent grid size); two streams with
// 128 each run two layers side by side.
extern "C" int y3_set_conv_cus(int cus) {
    Y3_REQUIRE(cus == 0 || (cus >= 8 && cus <= 256 && cus % 8 == 0), Y3_E_INVALID, "conv CUs %d (0 = all, or 8..256 in multiples of 8)", cus);
    g_conv_cus_cap = cus;
    return Y3_OK;
}

// Experiment knob: start offset of the de-phased persistent workgroups, percent of a tile period (0 = off).
extern "C" int y3_set_conv_phase(int percent) {
    Y3_REQUIRE(percent >= 0 && percent <= 400, Y3_E_INVALID, "conv phase %d%% (0..400)", percent);
    g_conv_phase_pct = percent;
    return Y3_OK;
}

extern "C" int y3_set_conv_tile(int tile) {
    g_conv_debug = tile >> 16;                 // high half: debug bits for timing experiments
    tile &= 0xffff;
    Y3_REQUIRE(tile == 0 || tile == 512 || tile == 704 || tile == 128 || tile == 129 || tile == 130 || tile == 131 || tile == 192 || tile == 256 || tile == 64 || tile == 257 || tile == 258 || tile == 260 || tile == 261, Y3_E_INVALID, "conv tile %d (0, 128, 192, 256 or 257)", tile);
    g_conv_tile = tile;
    return Y3_OK;
}
#endif  // Y3_EXPERIMENTS
