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

__device__ __forceinline__ unsigned long long wave_max_u64(unsigned long long v) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) {
        const unsigned lo = __shfl_xor((unsigned)v, o), hi = __shfl_xor((unsigned)(v >> 32), o);
        const unsigned long long w = ((unsigned long long)hi << 32) | lo;
        v = w > v ? w : v;
    }
    return v;
}

// key: high word = score bits (scores are >= +0, so the bit pattern is monotonic),
// low word = ~pooled index  ->  max key = highest score, lowest index on ties.
__device__ __forceinline__ unsigned long long make_key(float score, int idx) {
    return ((unsigned long long)__float_as_uint(score) << 32) | (unsigned)(~(unsigned)idx);
}

__global__ __launch_bounds__(kThreads) void nms_kernel(const float *records, const int *cand_idx, const int *cand_count,
                                                       const int *class_count, int B, int N, int C, int max_boxes,
                                                       float thr, int *kept_idx, int *kept_rank, int *kept_count) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    int *s_list = (int *)smem;                                   // [N] pooled indices of this class
    unsigned char *s_alive = (unsigned char *)(s_list + N);      // [N] LDS-resident suppression state
    __shared__ unsigned long long s_part[2][4];
    __shared__ int s_n;
    __shared__ int s_rank[kThreads];
    __shared__ int s_kept[kThreads];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int pair = blockIdx.x; pair < B * C; pair += gridDim.x) {
        const int b = pair / C, c = pair - b * C;
        int *kidx = kept_idx + (long)pair * max_boxes;
        const int ncls = class_count[pair];                      // wave-uniform
        if (ncls == 0) {
            if (tid == 0) kept_count[pair] = 0;
            continue;
        }
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
        for (int i0 = 0; i0 < nimg; i0 += kThreads) {
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
            for (int i = tid; i < n; i += kThreads) {
                if (!s_alive[i]) continue;
                const int idx = s_list[i];
                const f32x4 bx = *(const f32x4 *)(rec + (long)idx * 8);
                if (win_idx >= 0 && (idx == win_idx || iou_gt(bx, win, thr))) { s_alive[i] = 0; continue; }
                const unsigned long long key = make_key(rec[(long)idx * 8 + 4], idx);
                best = key > best ? key : best;
            }
            best = wave_max_u64(best);
            if (lane == 0) s_part[kept & 1][wave] = best;
            __syncthreads();
            const unsigned long long p0 = s_part[kept & 1][0], p1 = s_part[kept & 1][1];
            const unsigned long long p2 = s_part[kept & 1][2], p3 = s_part[kept & 1][3];
            unsigned long long bb = p0 > p1 ? p0 : p1;
            const unsigned long long cc = p2 > p3 ? p2 : p3;
            bb = bb > cc ? bb : cc;
            if (bb == 0ull) break;                               // nothing alive (block-uniform)
            win_idx = (int)(~(unsigned)bb);
            win = *(const f32x4 *)(rec + (long)win_idx * 8);
            if (tid == 0) { kidx[kept] = win_idx; s_kept[kept] = win_idx; }
            ++kept;
        }
        if (tid == 0) kept_count[pair] = kept;

        // ---- optional: tf nms_index = position in the order-preserving class-masked array ----
        if (kept_rank != nullptr) {
            s_rank[tid] = 0;
            __syncthreads();
            for (int k = 0; k < kept; ++k) {
                const int kid = s_kept[k];
                int cnt = 0;
                for (int i = tid; i < n; i += kThreads) cnt += s_list[i] < kid ? 1 : 0;
#pragma unroll
                for (int o = 32; o >= 1; o >>= 1) cnt += __shfl_xor(cnt, o);
                if (lane == 0 && cnt) atomicAdd(&s_rank[k], cnt);
            }
            __syncthreads();
            if (tid < kept) kept_rank[(long)pair * max_boxes + tid] = s_rank[tid];
        }
    }
}

__global__ __launch_bounds__(kThreads) void pack_kernel(const float *records, const int *kept_idx, const int *kept_count,
                                                        int N, int C, int max_boxes, float *out_boxes,
                                                        float *out_scores, int *out_classes, int *out_count) {
    __shared__ int s_off[128];
    const int b = blockIdx.x, tid = threadIdx.x;
    if (tid == 0) {
        int acc = 0;
        for (int c = 0; c < C; ++c) { s_off[c] = acc; acc += min(kept_count[b * C + c], max_boxes); }
        s_off[C] = acc;
        out_count[b] = acc;
    }
    __syncthreads();
    const long cap = (long)C * max_boxes;
    const float *rec = records + (long)b * N * 8;
    for (int c = 0; c < C; ++c) {                                  // class-major, inference.py:163-165
        const int cnt = s_off[c + 1] - s_off[c];
        for (int k = tid; k < cnt; k += kThreads) {
            const int idx = kept_idx[((long)b * C + c) * max_boxes + k];
            const long o = b * cap + s_off[c] + k;
            *(f32x4 *)(out_boxes + o * 4) = *(const f32x4 *)(rec + (long)idx * 8);
            out_scores[o] = rec[(long)idx * 8 + 4];
            out_classes[o] = c;
        }
    }
}

}  // namespace

extern "C" int y3_nms_lds_bytes(int boxes_per_image) { return boxes_per_image * 5 + 16; }

extern "C" int y3_nms(const float *records, const int32_t *cand_idx, const int32_t *cand_count,
                      const int32_t *class_count, int batch, int boxes_per_image, int num_classes, int max_boxes,
                      float iou_threshold, int32_t *kept_idx, int32_t *kept_rank, int32_t *kept_count, void *stream) {
    Y3_REQUIRE(records && cand_idx && cand_count && class_count && kept_idx && kept_count, Y3_E_INVALID, "nms: null pointer");
    Y3_REQUIRE(batch > 0 && boxes_per_image > 0 && num_classes > 0 && num_classes <= 96, Y3_E_INVALID, "nms: bad shape");
    Y3_REQUIRE(max_boxes > 0 && max_boxes <= kThreads, Y3_E_INVALID, "nms: max_boxes %d (1..%d)", max_boxes, kThreads);
    Y3_REQUIRE(aligned16(records), Y3_E_ALIGN, "nms: records not 16-byte aligned");
    const int lds = (boxes_per_image * 5 + 15) / 16 * 16;
    Y3_REQUIRE(lds <= 150 * 1024, Y3_E_INVALID, "nms: %d boxes per image exceed the LDS-resident limit (30720)", boxes_per_image);
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
    const int per_cu = lds > 0 ? (155 * 1024) / (lds + 1024) : 8;
    int blocks = 256 * (per_cu < 1 ? 1 : (per_cu > 8 ? 8 : per_cu));
    if (blocks > batch * num_classes) blocks = batch * num_classes;
    hipLaunchKernelGGL(nms_kernel, dim3(blocks), dim3(kThreads), lds, (hipStream_t)stream, records, cand_idx, cand_count,
                       class_count, batch, boxes_per_image, num_classes, max_boxes, iou_threshold, kept_idx, kept_rank, kept_count);
    return y3_check_launch("nms");
}

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
