// Per-class greedy NMS + output packing for gfx950.
//   reference inference.py:136-167 (80 x tf.image.non_max_suppression on the host CPU,
//   then batch_gather / concat).
// One persistent workgroup walks (image, class) pairs; empty pairs cost one scalar load.
// For a non-empty pair the workgroup compacts the class's candidates into an LDS index
// list with an LDS-resident alive map, then repeats at most max_boxes times:
//   pick the alive candidate with the largest (score, -index) key   (block max-reduce)
//   every wave tests its alive candidates' IoU against it in parallel and clears the
//   alive flag where IoU > threshold.
// This visits candidates in exactly the order of the sequential greedy algorithm
// (descending score, lower index first on ties) and applies the same suppression
// predicate, so the kept indices are identical to it by construction; cost is
// O(max_boxes * n) instead of O(n^2), with no sort.  IoU is fp32 in TF's operation
// order; the file is built with -ffp-contract=off and IEEE division.
#include "common.h"

namespace {

constexpr int kThreads = 256;
constexpr int kWsHeader = 128;      // ints before a bucketed image's candidate list: start[0..C]
constexpr int kSlots = 2;          // candidates per lane in nms_wave_tier: up to 128 per (image, class) pair
constexpr int kBSlots = 4;         // candidates per lane in nms_block_tier (256 threads): up to 1024 per pair
constexpr int kBlockMax = kThreads * kBSlots;

__device__ __forceinline__ bool iou_gt(const f32x4 bi, const f32x4 bj, float thr) {
    const float ymin_i = fminf(bi[0], bi[2]), xmin_i = fminf(bi[1], bi[3]);
    const float ymax_i = fmaxf(bi[0], bi[2]), xmax_i = fmaxf(bi[1], bi[3]);
    const float ymin_j = fminf(bj[0], bj[2]), xmin_j = fminf(bj[1], bj[3]);
    const float ymax_j = fmaxf(bj[0], bj[2]), xmax_j = fmaxf(bj[1], bj[3]);
    const float area_i = (ymax_i - ymin_i) * (xmax_i - xmin_i);
    const float area_j = (ymax_j - ymin_j) * (xmax_j - xmin_j);
    if (area_i <= 0.0f || area_j <= 0.0f) return false;
    const float iymin = fmaxf(ymin_i, ymin_j), ixmin = fmaxf(xmin_i, xmin_j);
    const float iymax = fminf(ymax_i, ymax_j), ixmax = fminf(xmax_i, xmax_j);
    const float inter = fmaxf(iymax - iymin, 0.0f) * fmaxf(ixmax - ixmin, 0.0f);
    const float iou = inter / (area_i + area_j - inter);
    return iou > thr;
}

// Maximum of a 64-bit key over the wavefront, the same value in every lane: four rotate-and-compare DPP
// steps inside each row of 16 lanes, then the four row results through v_readlane.  (A ds_bpermute
// butterfly is twelve dependent LDS round trips, which dominated a greedy round.)
template <int CTRL>
__device__ __forceinline__ unsigned long long max_dpp(unsigned long long v) {
    const unsigned long long w = ((unsigned long long)dpp_u<CTRL>((unsigned)(v >> 32)) << 32) | dpp_u<CTRL>((unsigned)v);
    return w > v ? w : v;
}
// The per-box half of iou_gt, hoisted: corners ordered, area computed once.  iou_gt_norm(norm(a), norm(b))
// performs exactly iou_gt(a, b)'s operations on the same values, so the two agree bit for bit.
struct NBox { float ymin, xmin, ymax, xmax, area; };
__device__ __forceinline__ NBox norm_box(const f32x4 b) {
    NBox n;
    n.ymin = fminf(b[0], b[2]); n.xmin = fminf(b[1], b[3]);
    n.ymax = fmaxf(b[0], b[2]); n.xmax = fmaxf(b[1], b[3]);
    n.area = (n.ymax - n.ymin) * (n.xmax - n.xmin);
    return n;
}
__device__ __forceinline__ bool iou_gt_norm(const NBox &i, const NBox &j, float thr) {
    // branch-free (the early return cost a divergent branch per slot and round); same arithmetic for valid pairs,
    // and an invalid pair's quotient (possibly inf / NaN) is masked
    const bool valid = (i.area > 0.0f) & (j.area > 0.0f);
    const float iymin = vmax_raw(i.ymin, j.ymin), ixmin = vmax_raw(i.xmin, j.xmin);      // boxes are finite: raw min/max
    const float iymax = vmin_raw(i.ymax, j.ymax), ixmax = vmin_raw(i.xmax, j.xmax);
    const float inter = fmaxf(iymax - iymin, 0.0f) * fmaxf(ixmax - ixmin, 0.0f);
    const float iou = inter / (i.area + j.area - inter);
    return valid & (iou > thr);
}

__device__ __forceinline__ unsigned long long wave_max_u64(unsigned long long v) {
    v = max_dpp<0x128>(v);
    v = max_dpp<0x124>(v);
    v = max_dpp<0x122>(v);
    v = max_dpp<0x121>(v);
    unsigned long long best = 0ull;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const unsigned long long w = ((unsigned long long)__builtin_amdgcn_readlane((unsigned)(v >> 32), 16 * r) << 32) |
                                     (unsigned)__builtin_amdgcn_readlane((unsigned)v, 16 * r);
        best = w > best ? w : best;
    }
    return best;
}

// key: high word = score bits (scores are >= +0, so the bit pattern is monotonic),
// low word = ~pooled index  ->  max key = highest score, lowest index on ties.
__device__ __forceinline__ unsigned long long make_key(float score, int idx) {
    return ((unsigned long long)__float_as_uint(score) << 32) | (unsigned)(~(unsigned)idx);
}

// ws (nullable): per image, class-bucketed candidate lists from nms_bucket_kernel.  With it, pairs of at most
// small_limit candidates belong to nms_tiers_kernel (nms_wave_tier / nms_block_tier) and are skipped here, and the others copy their
// bucket instead of scanning the image's whole candidate list.
__global__ __launch_bounds__(kThreads) void nms_kernel(const float *records, const int *cand_idx, const int *cand_count,
                                                       const int *class_count, int B, int N, int C, int max_boxes,
                                                       float thr, int *kept_idx, int *kept_rank, int *kept_count,
                                                       const int *ws, int ws_stride, int small_limit, int lds_bytes) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    int *s_list = (int *)smem;                                   // [N] pooled indices of this class
    unsigned char *s_alive = (unsigned char *)(s_list + N);      // [N] LDS-resident suppression state
    __shared__ unsigned long long s_part[2][4];
    __shared__ f32x4 s_wbox[2][4];
    __shared__ int s_n;
    __shared__ int s_rank[kThreads];
    __shared__ int s_kept[kThreads];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int slot = blockIdx.x; slot < B * C; slot += gridDim.x) {
        // class-major walk: a crowded class is crowded in every image, and b * C + c strided by the grid would hand
        // all of its images to the same few workgroups (C = 80, grid 256: 16 of them)
        const int c = slot / B, b = slot - c * B;
        const int pair = b * C + c;
        int *kidx = kept_idx + (long)pair * max_boxes;
        const int ncls = class_count[pair];                      // wave-uniform
        if (ncls == 0) {
            if (tid == 0 && ws == nullptr) kept_count[pair] = 0;
            continue;
        }
        const int *wsb = ws ? ws + (long)b * ws_stride : nullptr;
        const bool bucketed = wsb && wsb[0] >= 0;                // block-uniform
        if (bucketed && ncls <= small_limit) continue;           // nms_tiers_kernel's
        __syncthreads();                                         // previous pair's LDS use is over
        if (tid == 0) s_n = 0;
        __syncthreads();
        // ---- compact this class's candidates (any order; selection is by key) ----
        int nimg = cand_count[b];
        if (nimg > N) nimg = N;
        const float *rec = records + (long)b * N * 8;
        const int *cl = cand_idx + (long)b * N;
        const int cw = 5 + (c >> 5);
        const unsigned cbit = 1u << (c & 31);
        // Bucketed pairs whose boxes and keys (24 bytes a candidate) fit the LDS keep them there: a sweep then reads
        // LDS instead of 20 bytes per candidate from L2, and the winner's box comes from the winning wave, not from
        // memory (dense scenes: thousands of candidates per class, 51 sweeps each).
        const int nseg = min(ncls, N);
        const bool cached = bucketed && (long)nseg * 24 <= lds_bytes;       // block-uniform
        f32x4 *c_box = (f32x4 *)smem;                                       // [nseg]
        unsigned long long *c_key = (unsigned long long *)(smem + (long)nseg * 16);   // [nseg], 0 = suppressed
        const int *seg = bucketed ? wsb + kWsHeader + wsb[c] : nullptr;
        if (cached) {
            for (int i = tid; i < nseg; i += kThreads) {
                const int idx = seg[i];
                c_box[i] = *(const f32x4 *)(rec + (long)idx * 8);
                c_key[i] = make_key(rec[(long)idx * 8 + 4], idx);
            }
            if (tid == 0) s_n = nseg;
        } else if (bucketed) {
            for (int i = tid; i < nseg; i += kThreads) { s_list[i] = seg[i]; s_alive[i] = 1; }
            if (tid == 0) s_n = nseg;
        }
        for (int i0 = 0; i0 < (bucketed ? 0 : nimg); i0 += kThreads) {
            const int i = i0 + tid;
            bool mine = false;
            int idx = 0;
            if (i < nimg) {
                idx = cl[i];
                mine = (__float_as_uint(rec[(long)idx * 8 + cw]) & cbit) != 0;
            }
            const unsigned long long bal = __ballot(mine);
            int wbase = 0;
            if (lane == 0 && bal) wbase = atomicAdd(&s_n, __popcll(bal));
            wbase = __shfl(wbase, 0);
            if (mine) {
                const int pos = wbase + __popcll(bal & ((1ull << lane) - 1ull));
                if (pos < N) { s_list[pos] = idx; s_alive[pos] = 1; }
            }
        }
        __syncthreads();
        const int n = min(s_n, N);

        // ---- greedy selection --------------------------------------------------------------
        int kept = 0;
        f32x4 win = f32x4{0.f, 0.f, 0.f, 0.f};
        int win_idx = -1;
        while (kept < max_boxes) {
            // sweep: suppress against the last winner, find the best survivor
            unsigned long long best = 0ull;
            f32x4 bbox = f32x4{0.f, 0.f, 0.f, 0.f};
            if (cached) {
                for (int i = tid; i < n; i += kThreads) {
                    const unsigned long long key = c_key[i];
                    if (key == 0ull) continue;
                    const f32x4 bx = c_box[i];
                    if (win_idx >= 0 && ((int)(~(unsigned)key) == win_idx || iou_gt(bx, win, thr))) { c_key[i] = 0ull; continue; }
                    if (key > best) { best = key; bbox = bx; }
                }
            } else {
                for (int i = tid; i < n; i += kThreads) {
                    if (!s_alive[i]) continue;
                    const int idx = s_list[i];
                    const f32x4 bx = *(const f32x4 *)(rec + (long)idx * 8);
                    if (win_idx >= 0 && (idx == win_idx || iou_gt(bx, win, thr))) { s_alive[i] = 0; continue; }
                    const unsigned long long key = make_key(rec[(long)idx * 8 + 4], idx);
                    best = key > best ? key : best;
                }
            }
            const unsigned long long lbest = best;
            best = wave_max_u64(best);
            if (cached && best != 0ull && lbest == best) s_wbox[kept & 1][wave] = bbox;      // keys are unique: one lane
            if (lane == 0) s_part[kept & 1][wave] = best;
            __syncthreads();
            const unsigned long long p0 = s_part[kept & 1][0], p1 = s_part[kept & 1][1];
            const unsigned long long p2 = s_part[kept & 1][2], p3 = s_part[kept & 1][3];
            unsigned long long bb = p0 > p1 ? p0 : p1;
            const unsigned long long cc = p2 > p3 ? p2 : p3;
            bb = bb > cc ? bb : cc;
            if (bb == 0ull) break;                               // nothing alive (block-uniform)
            win_idx = (int)(~(unsigned)bb);
            if (cached) {
                const int ww = bb == p0 ? 0 : (bb == p1 ? 1 : (bb == p2 ? 2 : 3));
                win = s_wbox[kept & 1][ww];
            } else {
                win = *(const f32x4 *)(rec + (long)win_idx * 8);
            }
            if (tid == 0) s_kept[kept] = win_idx;                // written out once below (see nms_block_tier)
            ++kept;
        }
        if (tid == 0) kept_count[pair] = kept;
        __syncthreads();                                         // s_kept complete
        if (tid < kept) kidx[tid] = s_kept[tid];

        // ---- optional: tf nms_index = position in the order-preserving class-masked array ----
        if (kept_rank != nullptr) {
            s_rank[tid] = 0;
            __syncthreads();
            for (int k = 0; k < kept; ++k) {
                const int kid = s_kept[k];
                int cnt = 0;
                for (int i = tid; i < n; i += kThreads) cnt += (cached ? seg[i] : s_list[i]) < kid ? 1 : 0;
#pragma unroll
                for (int o = 32; o >= 1; o >>= 1) cnt += __shfl_xor(cnt, o);
                if (lane == 0 && cnt) atomicAdd(&s_rank[k], cnt);
            }
            __syncthreads();
            if (tid < kept) kept_rank[(long)pair * max_boxes + tid] = s_rank[tid];
        }
    }
}


// ---- class bucketing: one workgroup per image --------------------------------------------------------
// ws[b] = { start[0..C] (exclusive prefix sums of class_count), pad to kWsHeader, list[...] } where
// list[start[c] .. start[c+1]) are the pooled indices of the image's candidates of class c (any order).
// A candidate tied between classes is listed under each.  If that makes the lists longer than N
// (degenerate all-tie inputs), ws[b][0] = -1 and the image is left to nms_kernel's scan.
__global__ __launch_bounds__(1024) void nms_bucket_kernel(const float *records, const int *cand_idx, const int *cand_count,
                                                          const int *class_count, int N, int C, int *ws, int ws_stride) {
    __shared__ int s_start[128], s_fill[128];
    const int b = blockIdx.x, tid = threadIdx.x;
    if (tid < 128) s_fill[tid] = 0;
    if (tid == 0) {
        int acc = 0;
        for (int c = 0; c < C; ++c) { s_start[c] = acc; acc += class_count[b * C + c]; }
        s_start[C] = acc;
    }
    __syncthreads();
    int *wsb = ws + (long)b * ws_stride;
    if (s_start[C] > N) {
        if (tid == 0) wsb[0] = -1;
        return;
    }
    if (tid <= C) wsb[tid] = s_start[tid];
    const int nimg = min(cand_count[b], N);
    const float *rec = records + (long)b * N * 8;
    const int *cl = cand_idx + (long)b * N;
    for (int i = tid; i < nimg; i += 1024) {
        const int idx = cl[i];
#pragma unroll
        for (int w = 0; w < 3; ++w) {
            unsigned m = __float_as_uint(rec[(long)idx * 8 + 5 + w]);
            while (m) {
                const int c = w * 32 + __ffs(m) - 1;
                m &= m - 1;
                if (c < C) {
                    const int pos = s_start[c] + atomicAdd(&s_fill[c], 1);
                    if (pos < N) wsb[kWsHeader + pos] = idx;
                }
            }
        }
    }
}

// ---- one wavefront per (image, class) pair: candidates live in registers ----------------------------------
// Same greedy order and suppression predicate as nms_kernel (largest (score, -index) key first, IoU > thr
// strict), for pairs of at most 64 * kSlots candidates: no LDS, no workgroup barrier, the winner's box
// travels by v_readlane.  Typical pairs have a handful of candidates, so thousands of pairs run at once.
__device__ __forceinline__ void nms_wave_tier(int block, int blocks, const float *records, const int *class_count, const int *ws,
                                              int ws_stride, int B, int N, int C, int max_boxes, float thr,
                                              int *kept_idx, int *kept_rank, int *kept_count) {
    const int lane = threadIdx.x & 63;
    const int wave0 = (block * kThreads + (int)threadIdx.x) >> 6, nwaves = (blocks * kThreads) >> 6;
    for (int pair = wave0; pair < B * C; pair += nwaves) {
        const int b = pair / C, c = pair - b * C;
        const int n = __builtin_amdgcn_readfirstlane(class_count[pair]);
        if (n == 0) {
            if (lane == 0) kept_count[pair] = 0;
            continue;
        }
        const int *wsb = ws + (long)b * ws_stride;
        if (n > 64 * kSlots || __builtin_amdgcn_readfirstlane(wsb[0]) < 0) continue;        // a workgroup kernel's
        const int *list = wsb + kWsHeader + __builtin_amdgcn_readfirstlane(wsb[c]);
        const float *rec = records + (long)b * N * 8;
        const int ns = (n + 63) >> 6;             // slots in use (wave-uniform): the loops below skip the rest
        f32x4 box[kSlots];
        NBox nb[kSlots];
        unsigned long long key[kSlots];
        int idxs[kSlots];
#pragma unroll
        for (int k = 0; k < kSlots; ++k) {
            key[k] = 0ull;
            idxs[k] = 0x7fffffff;
            box[k] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (k < ns) {
                const int i = lane + 64 * k;
                const int idx = list[i < n ? i : 0];
                idxs[k] = i < n ? idx : 0x7fffffff;
                box[k] = *(const f32x4 *)(rec + (long)idx * 8);
                key[k] = i < n ? make_key(rec[(long)idx * 8 + 4], idx) : 0ull;
            }
            nb[k] = norm_box(box[k]);
        }
        int kept = 0;
        int mine[4] = {0, 0, 0, 0};               // lane l holds kept index l + 64q in mine[q] (max_boxes <= 256)
        int *kidx = kept_idx + (long)pair * max_boxes;
        while (kept < max_boxes) {
            unsigned long long best = key[0];
#pragma unroll
            for (int k = 1; k < kSlots; ++k)
                if (k < ns) best = key[k] > best ? key[k] : best;
            best = wave_max_u64(best);
            if (best == 0ull) break;                                 // nothing alive (wave-uniform)
            f32x4 cand = f32x4{0.f, 0.f, 0.f, 0.f};
            bool own = false;
#pragma unroll
            for (int k = 0; k < kSlots; ++k)
                if (k < ns && key[k] == best) { cand = box[k]; own = true; key[k] = 0ull; }
            const int src = __ffsll((unsigned long long)__ballot(own)) - 1;      // indices are unique: one owner
            f32x4 win;
#pragma unroll
            for (int q = 0; q < 4; ++q)
                win[q] = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(cand[q]), src));
            const NBox nwin = norm_box(win);
            const int win_idx = (int)(~(unsigned)best);
            if (lane == 0) kidx[kept] = win_idx;
#pragma unroll
            for (int q = 0; q < 4; ++q)
                if ((kept >> 6) == q && (kept & 63) == lane) mine[q] = win_idx;
            ++kept;
#pragma unroll
            for (int k = 0; k < kSlots; ++k)
                if (k < ns) key[k] = iou_gt_norm(nb[k], nwin, thr) ? 0ull : key[k];
        }
        if (lane == 0) kept_count[pair] = kept;
        if (kept_rank != nullptr) {               // tf nms_index: position in the order-preserving class-masked array
            for (int kk = 0; kk < kept; ++kk) {
                int v = mine[0];
#pragma unroll
                for (int q = 1; q < 4; ++q) v = (kk >> 6) == q ? mine[q] : v;
                const int kid = __builtin_amdgcn_readlane(v, kk & 63);
                int cnt = 0;
#pragma unroll
                for (int k = 0; k < kSlots; ++k)
                    if (k < ns) cnt += __popcll((unsigned long long)__ballot(idxs[k] < kid));
                if (lane == 0) kept_rank[(long)pair * max_boxes + kk] = cnt;
            }
        }
    }
}

// ---- one 256-thread workgroup per (image, class) pair of 64*kSlots < n <= 1024 candidates ----------------
// Up to four candidates per thread, in registers.  One barrier per round: every wave publishes its best key
// AND that candidate's box; after the barrier each thread picks the winning wave's pair from LDS (buffers
// alternate between rounds).  Same order and predicate as the other two kernels.
__device__ __forceinline__ void nms_block_tier(int block, int blocks, const float *records, const int *class_count, const int *ws,
                                               int ws_stride, int B, int N, int C, int max_boxes, float thr,
                                               int *kept_idx, int *kept_rank, int *kept_count) {
    __shared__ unsigned long long s_key[2][4];
    __shared__ f32x4 s_box[2][4];
    __shared__ int s_kept[kThreads], s_rank[kThreads];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int pair = block; pair < B * C; pair += blocks) {
        const int b = pair / C, c = pair - b * C;
        const int n = class_count[pair];
        const int *wsb = ws + (long)b * ws_stride;
        if (n <= 64 * kSlots || n > kBlockMax || wsb[0] < 0) continue;         // block-uniform
        const int *list = wsb + kWsHeader + wsb[c];
        const float *rec = records + (long)b * N * 8;
        const int ns = (n + kThreads - 1) / kThreads;             // slots in use (block-uniform)
        f32x4 box[kBSlots];
        NBox nb[kBSlots];
        unsigned long long key[kBSlots];
        int idxs[kBSlots];
#pragma unroll
        for (int k = 0; k < kBSlots; ++k) {
            const int i = tid + kThreads * k;
            const int idx = list[i < n ? i : 0];
            idxs[k] = i < n ? idx : 0x7fffffff;
            box[k] = *(const f32x4 *)(rec + (long)idx * 8);
            nb[k] = norm_box(box[k]);
            key[k] = i < n ? make_key(rec[(long)idx * 8 + 4], idx) : 0ull;
        }
        int *kidx = kept_idx + (long)pair * max_boxes;
        int kept = 0;
        __syncthreads();                                         // previous pair's LDS use is over
        while (kept < max_boxes) {
            unsigned long long lbest = key[0];
#pragma unroll
            for (int k = 1; k < kBSlots; ++k) lbest = key[k] > lbest ? key[k] : lbest;
            const unsigned long long wbest = wave_max_u64(lbest);
            if (wbest != 0ull && lbest == wbest) {               // indices are unique: one lane, one slot
                f32x4 cand = box[0];
#pragma unroll
                for (int k = 1; k < kBSlots; ++k) cand = key[k] == wbest ? box[k] : cand;
                s_box[kept & 1][wave] = cand;
            }
            if (lane == 0) s_key[kept & 1][wave] = wbest;
            __syncthreads();
            unsigned long long best = s_key[kept & 1][0];
            int ww = 0;
#pragma unroll
            for (int w = 1; w < 4; ++w) { const unsigned long long v = s_key[kept & 1][w]; if (v > best) { best = v; ww = w; } }
            if (best == 0ull) break;                             // nothing alive (block-uniform)
            const NBox win = norm_box(s_box[kept & 1][ww]);
            // (kept indices go to LDS and are written out once at the end: a global store here would be drained by the
            // fence in every round's __syncthreads -- ~2 us of store latency on the critical path of each round)
            if (tid == 0) s_kept[kept] = (int)(~(unsigned)best);
            ++kept;
#pragma unroll
            for (int k = 0; k < kBSlots; ++k) {
                // branch-free on purpose: as `if (alive && (key == best || iou)) key = 0` this update was
                // dropped by hipcc 7.2 for some lanes (caught by the parity test against y3_nms)
                if (k < ns) {                       // block-uniform
                    const bool kill = key[k] == best || iou_gt_norm(nb[k], win, thr);
                    key[k] = kill ? 0ull : key[k];
                }
            }
        }
        if (tid == 0) kept_count[pair] = kept;
        __syncthreads();                                         // s_kept complete
        if (tid < kept) kidx[tid] = s_kept[tid];
        if (kept_rank != nullptr) {
            s_rank[tid] = 0;
            __syncthreads();
            for (int kk = 0; kk < kept; ++kk) {
                int cnt = 0;
#pragma unroll
                for (int k = 0; k < kBSlots; ++k) cnt += __popcll((unsigned long long)__ballot(idxs[k] < s_kept[kk]));
                if (lane == 0 && cnt) atomicAdd(&s_rank[kk], cnt);
            }
            __syncthreads();
            if (tid < kept) kept_rank[(long)pair * max_boxes + tid] = s_rank[tid];
        }
    }
}

// The two register-resident tiers in ONE launch: workgroups [0, block_blocks) run the workgroup-per-pair tier -- first,
// because a crowded pair's up to max_boxes sequential rounds are the launch's critical path -- and the others the
// wavefront-per-pair tier, which fills the chip around them (as two launches their times added: 22 + 120 us).
__global__ __launch_bounds__(kThreads) void nms_tiers_kernel(const float *records, const int *class_count, const int *ws,
                                                             int ws_stride, int B, int N, int C, int max_boxes, float thr,
                                                             int *kept_idx, int *kept_rank, int *kept_count, int block_blocks) {
    if ((int)blockIdx.x < block_blocks)
        nms_block_tier((int)blockIdx.x, block_blocks, records, class_count, ws, ws_stride, B, N, C, max_boxes, thr, kept_idx, kept_rank, kept_count);
    else
        nms_wave_tier((int)blockIdx.x - block_blocks, (int)gridDim.x - block_blocks, records, class_count, ws, ws_stride, B, N, C, max_boxes, thr,
                      kept_idx, kept_rank, kept_count);
}

__global__ __launch_bounds__(kThreads) void pack_kernel(const float *records, const int *kept_idx, const int *kept_count,
                                                        int N, int C, int max_boxes, float *out_boxes,
                                                        float *out_scores, int *out_classes, int *out_count) {
    __shared__ int s_n[128], s_off[128];
    const int b = blockIdx.x, tid = threadIdx.x;
    if (tid < C) s_n[tid] = min(kept_count[b * C + tid], max_boxes);       // (C <= 127: checked by the caller)
    __syncthreads();
    if (tid <= C) {                                                    // exclusive prefix, every thread its own (C is small)
        int acc = 0;
        for (int c = 0; c < tid; ++c) acc += s_n[c];
        s_off[tid] = acc;
        if (tid == C) out_count[b] = acc;
    }
    __syncthreads();
    const long cap = (long)C * max_boxes;
    const float *rec = records + (long)b * N * 8;
    // class-major, inference.py:163-165: output slot i belongs to the class whose offset range holds it (binary search in
    // the LDS prefix table; one pass over the image's <= C * max_boxes detections instead of one short pass per class)
    const int total = s_off[C];
    for (int i = tid; i < total; i += kThreads) {
        int lo = 0, hi = C;                                        // s_off[lo] <= i < s_off[hi]
        while (hi - lo > 1) {
            const int mid = (lo + hi) >> 1;
            if (s_off[mid] <= i) lo = mid; else hi = mid;
        }
        const int c = lo, k = i - s_off[c];
        const int idx = kept_idx[((long)b * C + c) * max_boxes + k];
        const long o = b * cap + i;
        *(f32x4 *)(out_boxes + o * 4) = *(const f32x4 *)(rec + (long)idx * 8);
        out_scores[o] = rec[(long)idx * 8 + 4];
        out_classes[o] = c;
    }
}

}  // namespace

extern "C" int y3_nms_lds_bytes(int boxes_per_image) { return boxes_per_image * 5 + 16; }

static int launch_nms(const float *records, const int32_t *cand_idx, const int32_t *cand_count, const int32_t *class_count,
                      int batch, int boxes_per_image, int num_classes, int max_boxes, float iou_threshold, int32_t *workspace,
                      int32_t *kept_idx, int32_t *kept_rank, int32_t *kept_count, hipStream_t stream) {
    Y3_REQUIRE(records && cand_idx && cand_count && class_count && kept_idx && kept_count, Y3_E_INVALID, "nms: null pointer");
    Y3_REQUIRE(batch > 0 && boxes_per_image > 0 && num_classes > 0 && num_classes <= 96, Y3_E_INVALID, "nms: bad shape");
    Y3_REQUIRE(max_boxes > 0 && max_boxes <= kThreads, Y3_E_INVALID, "nms: max_boxes %d (1..%d)", max_boxes, kThreads);
    Y3_REQUIRE(aligned16(records), Y3_E_ALIGN, "nms: records not 16-byte aligned");
    const int lds_scan = (boxes_per_image * 5 + 15) / 16 * 16;
    Y3_REQUIRE(lds_scan <= 150 * 1024, Y3_E_INVALID, "nms: %d boxes per image exceed the LDS-resident limit (30720)", boxes_per_image);
    // with a workspace: room for 24-byte cached candidates as well (up to 6144 of them), one workgroup per CU then
    const int lds_cache = workspace ? (boxes_per_image < 6144 ? boxes_per_image : 6144) * 24 : 0;
    const int lds = lds_scan > lds_cache ? lds_scan : lds_cache;
    static thread_local int attr_dev = -1, attr_lds = 0;
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (attr_dev != dev || attr_lds < lds) {
        if (hipFuncSetAttribute((const void *)nms_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024) != hipSuccess) {
            y3_set_error("nms: cannot reserve LDS");
            return Y3_E_LAUNCH;
        }
        attr_dev = dev; attr_lds = 150 * 1024;
    }
    const int ws_stride = boxes_per_image + kWsHeader;
    const int pairs = batch * num_classes;
    if (workspace) {
        hipLaunchKernelGGL(nms_bucket_kernel, dim3(batch), dim3(1024), 0, stream, records, cand_idx, cand_count, class_count,
                           boxes_per_image, num_classes, workspace, ws_stride);
        int wblocks = (pairs + 3) / 4;
        if (wblocks > 256 * 8) wblocks = 256 * 8;
        hipLaunchKernelGGL(nms_tiers_kernel, dim3(pairs + wblocks), dim3(kThreads), 0, stream, records, class_count, workspace, ws_stride,
                           batch, boxes_per_image, num_classes, max_boxes, iou_threshold, kept_idx, kept_rank, kept_count, pairs);
    }
    const int per_cu = lds > 0 ? (155 * 1024) / (lds + 1024) : 8;
    int blocks = 256 * (per_cu < 1 ? 1 : (per_cu > 8 ? 8 : per_cu));
    if (blocks > pairs) blocks = pairs;
    hipLaunchKernelGGL(nms_kernel, dim3(blocks), dim3(kThreads), lds, stream, records, cand_idx, cand_count,
                       class_count, batch, boxes_per_image, num_classes, max_boxes, iou_threshold, kept_idx, kept_rank, kept_count,
                       workspace, ws_stride, kBlockMax, lds);
    return y3_check_launch("nms");
}

extern "C" int y3_nms(const float *records, const int32_t *cand_idx, const int32_t *cand_count,
                      const int32_t *class_count, int batch, int boxes_per_image, int num_classes, int max_boxes,
                      float iou_threshold, int32_t *kept_idx, int32_t *kept_rank, int32_t *kept_count, void *stream) {
    return launch_nms(records, cand_idx, cand_count, class_count, batch, boxes_per_image, num_classes, max_boxes, iou_threshold,
                      nullptr, kept_idx, kept_rank, kept_count, (hipStream_t)stream);
}

extern "C" int64_t y3_nms_workspace_ints(int batch, int boxes_per_image) {
    return (int64_t)batch * (boxes_per_image + kWsHeader);
}

extern "C" int y3_nms_bucketed(const float *records, const int32_t *cand_idx, const int32_t *cand_count,
                               const int32_t *class_count, int batch, int boxes_per_image, int num_classes, int max_boxes,
                               float iou_threshold, int32_t *workspace, int32_t *kept_idx, int32_t *kept_rank,
                               int32_t *kept_count, void *stream) {
    Y3_REQUIRE(workspace != nullptr, Y3_E_INVALID, "nms_bucketed: null workspace");
    return launch_nms(records, cand_idx, cand_count, class_count, batch, boxes_per_image, num_classes, max_boxes, iou_threshold,
                      workspace, kept_idx, kept_rank, kept_count, (hipStream_t)stream);
}

// ---- the one entry point that allocates nothing and takes no workspace -----------------------------------------
// y3_nms_bucketed's class-bucketed candidate lists (its `workspace` argument) live in a file-scope device array
// instead.  Each GPU the library runs on holds its own copy of the array until the process ends; the array's name
// is only a host-side handle, so the device address is looked up once per device (hipGetSymbolAddress, with that
// device current) and that pointer is what the kernels, like y3_nms_bucketed's, receive as an argument.
// Nothing in the array has to start at zero: nms_bucket_kernel writes every header word and every list entry the
// later kernels read, whatever an earlier call left behind.  Calls that overlap on one GPU share the array and race.
namespace {
constexpr long kStaticWsInts = Y3_NMS_STATIC_MAX_INTS;       // 8 MiB per GPU: batch * (boxes_per_image + 128) ints
static __device__ __align__(256) int scratch[kStaticWsInts];
int *g_scratch_ptr[64] = {nullptr};                          // per device; written once, under g_scratch_lock
int g_scratch_lock = 0;
}  // namespace

extern "C" int y3_nms_bucketed_static(const float *records, const int32_t *cand_idx, const int32_t *cand_count,
                                      const int32_t *class_count, int batch, int boxes_per_image, int num_classes,
                                      int max_boxes, float iou_threshold, int32_t *kept_idx, int32_t *kept_rank,
                                      int32_t *kept_count, void *stream) {
    Y3_REQUIRE(batch > 0 && boxes_per_image > 0, Y3_E_INVALID, "nms_bucketed_static: bad shape");
    const long need = (long)batch * (boxes_per_image + kWsHeader);
    Y3_REQUIRE(need <= kStaticWsInts, Y3_E_INVALID,
               "nms_bucketed_static: batch %d x %d boxes needs %ld ints of scratch, the built-in array holds %ld "
               "(use y3_nms_bucketed with a workspace of your own)", batch, boxes_per_image, need, kStaticWsInts);
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) {
        y3_set_error("nms_bucketed_static: device %d (0..63 supported)", dev);
        return Y3_E_LAUNCH;
    }
    int *ws = __atomic_load_n(&g_scratch_ptr[dev], __ATOMIC_ACQUIRE);
    if (ws == nullptr) {                                     // first call on this device: resolve ITS copy of the array
        while (__atomic_exchange_n(&g_scratch_lock, 1, __ATOMIC_ACQUIRE)) {}
        ws = g_scratch_ptr[dev];
        if (ws == nullptr) {
            void *p = nullptr;
            const hipError_t e = hipGetSymbolAddress(&p, HIP_SYMBOL(scratch));
            if (e != hipSuccess || p == nullptr) {
                __atomic_store_n(&g_scratch_lock, 0, __ATOMIC_RELEASE);
                y3_set_error("nms_bucketed_static: hipGetSymbolAddress: %s", hipGetErrorString(e));
                return Y3_E_LAUNCH;
            }
            ws = (int *)p;
            __atomic_store_n(&g_scratch_ptr[dev], ws, __ATOMIC_RELEASE);
        }
        __atomic_store_n(&g_scratch_lock, 0, __ATOMIC_RELEASE);
    }
    return launch_nms(records, cand_idx, cand_count, class_count, batch, boxes_per_image, num_classes, max_boxes, iou_threshold,
                      ws, kept_idx, kept_rank, kept_count, (hipStream_t)stream);
}

#ifdef Y3_EXPERIMENTS
// ---- libyolo3hip_exp.so only: the all-pairs IoU bitmask form of per-class NMS, for the A/B in DESIGN.md 3.5 -------
// BASELINE's north_star sketches NMS as "a wavefront-level IoU bitmask kernel with an LDS-resident suppression scan".
// This is that algorithm, written to be fast, so that the choice of the greedy-round kernels above rests on a
// measurement (tools/nms_ab.py) and not only on the operation count:
//   (1) bm_sort_kernel   a workgroup sorts a pair's candidates by (score desc, index asc) -- bitonic sort of the 64-bit
//                        keys in LDS -- and writes the order back over the pair's bucket list;
//   (2) bm_mask_kernel   a workgroup takes 256 rows of a pair's n x n matrix: lane = row i, and for every 64-candidate
//                        column block at or right of its own (staged in LDS, read as broadcasts) 64 IoU tests ->
//                        one 64-bit word mask[i][cb], bit jj = "j = 64 cb + jj > i and IoU(i, j) > thr";
//   (3) bm_scan_kernel   one wavefront per pair: the suppressed-set bitmap lives in registers (two words per lane);
//                        at most max_boxes times {first candidate not suppressed (ffs + wave minimum), keep it, OR its
//                        mask row into the bitmap}.
// Same order and the same predicate (iou_gt_norm) as the greedy kernels, hence the same kept indices.  n <= 8192 per
// pair; the masks of an image must fit mask_words_per_image 64-bit words (status[b] = 1 otherwise, nothing kept).
namespace {

constexpr int kBmMaxN = 8192;
constexpr int kBmRows = 256;            // rows of a pair's matrix per bm_mask_kernel work item

__device__ __forceinline__ long bm_pair_offset(const int *class_count, int b, int C, int c, long *total) {
    long off = 0, acc = 0;
    for (int q = 0; q < C; ++q) {
        const long n = class_count[b * C + q];
        if (q == c) off = acc;
        acc += n * ((n + 63) >> 6);
    }
    *total = acc;
    return off;
}

__global__ __launch_bounds__(kThreads) void bm_sort_kernel(const float *records, const int *class_count, int *ws, int ws_stride,
                                                           int B, int N, int C, long mask_words, int *status) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    unsigned long long *keys = (unsigned long long *)smem;
    const int tid = threadIdx.x;
    for (int slot = blockIdx.x; slot < B * C; slot += gridDim.x) {
        const int c = slot / B, b = slot - c * B;                // class-major, as nms_kernel
        const int n = class_count[b * C + c];
        int *wsb = ws + (long)b * ws_stride;
        if (c == 0 && tid == 0) {
            long total;
            bm_pair_offset(class_count, b, C, 0, &total);
            int bad = total > mask_words || wsb[0] < 0;
            for (int q = 0; q < C; ++q) bad |= class_count[b * C + q] > kBmMaxN;
            status[b] = bad;
        }
        if (n < 2 || n > kBmMaxN || wsb[0] < 0) continue;        // block-uniform
        int *list = wsb + kWsHeader + wsb[c];
        const float *rec = records + (long)b * N * 8;
        int P = 2;
        while (P < n) P <<= 1;
        __syncthreads();                                         // previous pair's LDS use is over
        for (int i = tid; i < P; i += kThreads) {
            unsigned long long k = 0ull;
            if (i < n) { const int idx = list[i]; k = make_key(rec[(long)idx * 8 + 4], idx); }
            keys[i] = k;
        }
        __syncthreads();
        for (int k = 2; k <= P; k <<= 1) {
            for (int j = k >> 1; j > 0; j >>= 1) {
                for (int t = tid; t < (P >> 1); t += kThreads) {
                    const int i = 2 * t - (t & (j - 1)), l = i + j;
                    const bool desc = (i & k) == 0;
                    const unsigned long long a = keys[i], bb = keys[l];
                    if ((a < bb) == desc) { keys[i] = bb; keys[l] = a; }
                }
                __syncthreads();
            }
        }
        for (int i = tid; i < n; i += kThreads) list[i] = (int)(~(unsigned)keys[i]);
    }
}

__global__ __launch_bounds__(kThreads) void bm_mask_kernel(const float *records, const int *class_count, const int *ws,
                                                           int ws_stride, int B, int N, int C, float thr,
                                                           unsigned long long *mask, long mask_words, const int *status) {
    __shared__ f32x4 s_col[4][64];                               // a column block's boxes, corners ordered
    __shared__ float s_area[4][64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    constexpr int kSplit = 8;            // workgroups that share a pair: work item (pair, s) takes row chunks s, s + 8, ...
    for (long item = blockIdx.x; item < (long)B * C * kSplit; item += gridDim.x) {
        const int s0 = (int)(item % kSplit);
        const int slot = (int)(item / kSplit);
        const int c = slot / B, b = slot - c * B;
        const int n = class_count[b * C + c];
        if (s0 * kBmRows >= n || n < 2 || status[b]) continue;               // block-uniform
        const int *wsb = ws + (long)b * ws_stride;
        const int *list = wsb + kWsHeader + wsb[c];
        const float *rec = records + (long)b * N * 8;
        const int W = (n + 63) >> 6;
        long total;
        unsigned long long *m = mask + (long)b * mask_words + bm_pair_offset(class_count, b, C, c, &total);
        for (int chunk = s0; chunk * kBmRows < n; chunk += kSplit) {
            const int rb = chunk * (kBmRows / 64) + wave;                    // this wave's row block
            if (rb >= W) continue;                                           // wave-uniform; no workgroup barrier below
            const int i = rb * 64 + lane;
            const NBox bi = norm_box(*(const f32x4 *)(rec + (long)list[i < n ? i : 0] * 8));
            for (int cb = rb; cb < W; ++cb) {
                const int jl = cb * 64 + lane;
                const NBox bl = norm_box(*(const f32x4 *)(rec + (long)list[jl < n ? jl : 0] * 8));
                s_col[wave][lane] = f32x4{bl.ymin, bl.xmin, bl.ymax, bl.xmax};
                s_area[wave][lane] = bl.area;
                __builtin_amdgcn_wave_barrier();
                unsigned long long word = 0ull;
                const int jn = min(64, n - cb * 64);
                for (int jj = 0; jj < jn; ++jj) {
                    const f32x4 q = s_col[wave][jj];
                    NBox bj;
                    bj.ymin = q[0]; bj.xmin = q[1]; bj.ymax = q[2]; bj.xmax = q[3]; bj.area = s_area[wave][jj];
                    const bool hit = (cb * 64 + jj > i) & iou_gt_norm(bj, bi, thr);
                    word |= (unsigned long long)hit << jj;
                }
                if (i < n) m[(long)i * W + cb] = word;
                __builtin_amdgcn_wave_barrier();
            }
        }
    }
}

__global__ __launch_bounds__(kThreads) void bm_scan_kernel(const int *class_count, const int *ws, int ws_stride, int B, int C,
                                                           int max_boxes, const unsigned long long *mask, long mask_words,
                                                           const int *status, int *kept_idx, int *kept_count) {
    const int lane = threadIdx.x & 63;
    const int wave0 = (blockIdx.x * kThreads + threadIdx.x) >> 6, nwaves = (gridDim.x * kThreads) >> 6;
    for (int pair = wave0; pair < B * C; pair += nwaves) {
        const int b = pair / C, c = pair - b * C;
        const int n = __builtin_amdgcn_readfirstlane(class_count[pair]);
        int *kidx = kept_idx + (long)pair * max_boxes;
        if (n == 0 || __builtin_amdgcn_readfirstlane(status[b])) {
            if (lane == 0) kept_count[pair] = 0;
            continue;
        }
        const int *wsb = ws + (long)b * ws_stride;
        const int *list = wsb + kWsHeader + __builtin_amdgcn_readfirstlane(wsb[c]);
        const int W = (n + 63) >> 6;
        long total;
        const unsigned long long *m = mask + (long)b * mask_words + bm_pair_offset(class_count, b, C, c, &total);
        // suppressed-set bitmap: lane l holds words l and l + 64; bits at or beyond n start out set
        unsigned long long rem[2];
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int w = lane + 64 * q;
            const int valid = n - w * 64;                        // candidates in word w
            rem[q] = valid >= 64 ? 0ull : (valid <= 0 ? ~0ull : (~0ull << valid));
        }
        int kept = 0;
        while (kept < max_boxes) {
            int first = 0x7fffffff;                              // first candidate of this lane's words that is still alive
#pragma unroll
            for (int q = 1; q >= 0; --q)
                if (~rem[q] != 0ull) first = (lane + 64 * q) * 64 + __ffsll((unsigned long long)~rem[q]) - 1;
            // wave minimum: the keys' DPP maximum on the complemented value
            const unsigned long long best = wave_max_u64((unsigned long long)(unsigned)(0x7fffffff - first));
            const int i = 0x7fffffff - (int)(unsigned)best;
            if (i >= n) break;                                   // nothing alive (wave-uniform)
            if (lane == 0) kidx[kept] = list[i];
            ++kept;
            const int rb = i >> 6;
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const int w = lane + 64 * q;
                unsigned long long row = 0ull;
                if (w >= rb && w < W) row = m[(long)i * W + w];
                if (w == rb) row |= 1ull << (i & 63);            // the kept candidate itself leaves the alive set
                rem[q] |= row;
            }
        }
        if (lane == 0) kept_count[pair] = kept;
    }
}

}  // namespace

extern "C" int y3_nms_bitmask(const float *records, const int32_t *cand_idx, const int32_t *cand_count, const int32_t *class_count,
                              int batch, int boxes_per_image, int num_classes, int max_boxes, float iou_threshold,
                              int32_t *workspace, uint64_t *mask, int64_t mask_words_per_image, int32_t *status,
                              int32_t *kept_idx, int32_t *kept_count, void *stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    Y3_REQUIRE(records && cand_idx && cand_count && class_count && workspace && mask && status && kept_idx && kept_count,
               Y3_E_INVALID, "nms_bitmask: null pointer");
    Y3_REQUIRE(batch > 0 && boxes_per_image > 0 && num_classes > 0 && num_classes <= 96 && max_boxes > 0 && mask_words_per_image > 0,
               Y3_E_INVALID, "nms_bitmask: bad shape");
    static thread_local int attr_dev = -1;
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (attr_dev != dev) {
        if (hipFuncSetAttribute((const void *)bm_sort_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, kBmMaxN * 8) != hipSuccess) {
            y3_set_error("nms_bitmask: cannot reserve LDS");
            return Y3_E_LAUNCH;
        }
        attr_dev = dev;
    }
    const int ws_stride = boxes_per_image + kWsHeader;
    const int pairs = batch * num_classes;
    hipLaunchKernelGGL(nms_bucket_kernel, dim3(batch), dim3(1024), 0, stream, records, cand_idx, cand_count, class_count,
                       boxes_per_image, num_classes, workspace, ws_stride);
    hipLaunchKernelGGL(bm_sort_kernel, dim3(pairs < 512 ? pairs : 512), dim3(kThreads), kBmMaxN * 8, stream, records, class_count,
                       workspace, ws_stride, batch, boxes_per_image, num_classes, (long)mask_words_per_image, status);
    hipLaunchKernelGGL(bm_mask_kernel, dim3(2048), dim3(kThreads), 0, stream, records, class_count, workspace, ws_stride, batch,
                       boxes_per_image, num_classes, iou_threshold, (unsigned long long *)mask, (long)mask_words_per_image, status);
    int sblocks = (pairs + 3) / 4;
    if (sblocks > 2048) sblocks = 2048;
    hipLaunchKernelGGL(bm_scan_kernel, dim3(sblocks), dim3(kThreads), 0, stream, class_count, workspace, ws_stride, batch,
                       num_classes, max_boxes, (const unsigned long long *)mask, (long)mask_words_per_image, status, kept_idx, kept_count);
    return y3_check_launch("nms_bitmask");
}
#endif  // Y3_EXPERIMENTS

extern "C" int y3_pack_outputs(const float *records, const int32_t *kept_idx, const int32_t *kept_count, int batch,
                               int boxes_per_image, int num_classes, int max_boxes, float *out_boxes,
                               float *out_scores, int32_t *out_classes, int32_t *out_count, void *stream) {
    Y3_REQUIRE(records && kept_idx && kept_count && out_boxes && out_scores && out_classes && out_count, Y3_E_INVALID, "pack: null pointer");
    Y3_REQUIRE(batch > 0 && num_classes > 0 && num_classes <= 127 && max_boxes > 0, Y3_E_INVALID, "pack: bad shape");
    Y3_REQUIRE(aligned16(records) && aligned16(out_boxes), Y3_E_ALIGN, "pack: records/out_boxes not 16-byte aligned");
    hipLaunchKernelGGL(pack_kernel, dim3(batch), dim3(kThreads), 0, (hipStream_t)stream, records, kept_idx, kept_count,
                       boxes_per_image, num_classes, max_boxes, out_boxes, out_scores, out_classes, out_count);
    return y3_check_launch("pack");
}
