// YOLO head decode for gfx950: get_head + correct_boxes + get_boxes_and_scores + the
// score threshold of predict(), fused over all scales in one launch.
//   reference utils/utils.py:43-96, 100-129, 132-160; inference.py:110-127.
// Bandwidth-bound: the head tensors are read exactly once, 16 lanes per box so that a
// box's 85 consecutive floats are read as coalesced 64-byte segments; each box leaves
// one 32-byte record {ymin,xmin,ymax,xmax,score,classmask[3]}.  Boxes whose score
// passes the threshold are appended to a per-image candidate list (block-aggregated:
// one atomic per block) and counted per class (LDS histogram) for the NMS kernel.
// fp32 throughout, no multiply-add contraction (the file is built with
// -ffp-contract=off), IEEE division, operation order as in the reference.
#include "common.h"
#include "decode_box.h"

namespace {

#ifdef Y3_EXPERIMENTS
bool g_decode_force_v1 = false;      // test knob of libyolo3hip_exp.so: y3_decode_variant (yolo3hip_experimental.h)
#else
constexpr bool g_decode_force_v1 = false;
#endif

constexpr int kThreads = 256;
constexpr int kChunk = 1024;      // boxes per block
constexpr int kMaxClasses = 96;   // 3 x 32-bit class mask

struct DecodeArgs {
    y3_decode_desc d;
    const float *head[3];
    const float *letterbox;
    float *records;
    int *cand_idx, *cand_count, *class_count;
    int scale_base[4];
};

__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }

// 16-lane (one DPP row = one box group) all-reduce and broadcast without LDS traffic: row_ror:n rotates
// within the row, so four rotate-and-combine steps leave the full result in every lane; row_newbcast:n
// copies lane n of the row to all of its lanes.
__device__ __forceinline__ float row_max(float v) {
    v = fmaxf(v, __uint_as_float(dpp_u<0x128>(__float_as_uint(v))));
    v = fmaxf(v, __uint_as_float(dpp_u<0x124>(__float_as_uint(v))));
    v = fmaxf(v, __uint_as_float(dpp_u<0x122>(__float_as_uint(v))));
    v = fmaxf(v, __uint_as_float(dpp_u<0x121>(__float_as_uint(v))));
    return v;
}
__device__ __forceinline__ unsigned row_or(unsigned v) {
    v |= dpp_u<0x128>(v);
    v |= dpp_u<0x124>(v);
    v |= dpp_u<0x122>(v);
    v |= dpp_u<0x121>(v);
    return v;
}
template <int LANE>
__device__ __forceinline__ float row_bcast(float v) { return __uint_as_float(dpp_u<0x150 + LANE>(__float_as_uint(v))); }

__global__ __launch_bounds__(kThreads) void decode_kernel(const DecodeArgs a) {
    __shared__ int s_list[kChunk];
    __shared__ int s_hist[kMaxClasses];
    __shared__ int s_n, s_base;

    const int b = blockIdx.y;
    const int N = a.d.boxes_per_image;
    const int C = a.d.num_classes, A = a.d.num_anchors;
    const int tid = threadIdx.x;
    const int j = tid & 15;            // lane within the 16-lane box group
    const int grp = tid >> 4;          // 16 groups per block
    if (tid < kMaxClasses) s_hist[tid] = 0;
    if (tid == 0) s_n = 0;
    __syncthreads();

    const float scale = a.letterbox[b * 4 + 0];
    const float off_h = a.letterbox[b * 4 + 1];
    const float off_w = a.letterbox[b * 4 + 2];
    const float thr = a.d.score_threshold;
    const int p_begin = blockIdx.x * kChunk;
    const int p_end = min(p_begin + kChunk, N);

    for (int pb = p_begin; pb < p_end; pb += 16) {
        const int p = pb + grp;
        const bool live = p < p_end;
        const int pp = live ? p : p_end - 1;                  // keep all lanes in the shuffles
        const int l = pp >= a.scale_base[2] ? 2 : (pp >= a.scale_base[1] ? 1 : 0);
        const int q = pp - a.scale_base[l];
        const int cell = q / A, an = q - cell * A;
        const int gw = a.d.grid_w[l];
        const int gy = cell / gw, gx = cell - gy * gw;
        const float *t = a.head[l] + ((long)b * a.d.grid_h[l] * gw + cell) * a.d.head_ld[l] + an * (5 + C);

        // all seven loads unconditional (clamped index, select afterwards): a load under a lane condition
        // sits in its own branch with its own wait, which serialises seven memory round trips per box
        const float tbr = t[min(j, 4)];                       // lanes 0..4: tx,ty,tw,th,tobj
        float tc[6];
#pragma unroll
        for (int i = 0; i < 6; ++i) tc[i] = t[5 + min(j + 16 * i, C - 1)];
        const float tb = j < 5 ? tbr : 0.0f;
        float tmax = -__builtin_inff();
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            tc[i] = j + 16 * i < C ? tc[i] : -__builtin_inff();
            tmax = fmaxf(tmax, tc[i]);
        }
        tmax = row_max(tmax);
        // The class probabilities are sigmoid(t_c) (utils.py:87) and only their maximum and the classes that
        // attain it are used (utils.py:150-154).  sigmoid is increasing, so only logits close enough to the
        // largest one that their fp32 sigmoid could still reach the maximum need the exact evaluation: a
        // logit more than `win` below tmax has a true sigmoid at least 8 ulp below sigmoid(tmax)
        // (win * sigmoid'(tmax) = 1e-6 * sigmoid(tmax)), the fp32 evaluation errs by 2-3.  Where sigmoid
        // underflows (tmax < -80) every class is evaluated.
        const float win = tmax < -80.0f ? __builtin_inff() : 1e-6f * (1.0f + __expf(tmax)) + 1e-6f * fabsf(tmax);
        const float tcut = tmax - win;
        unsigned need = 0;
#pragma unroll
        for (int i = 0; i < 6; ++i) need |= (j + 16 * i < C && tc[i] >= tcut) ? 1u << i : 0u;
        float pc[6] = {-1.0f, -1.0f, -1.0f, -1.0f, -1.0f, -1.0f};
        float pmax = -1.0f;
        while (__builtin_amdgcn_ballot_w64(need != 0)) {      // usually one round: the arg-max class of each box
            const int i = need ? __ffs(need) - 1 : -1;
            need &= need - 1;
            float tv = 0.0f;
#pragma unroll
            for (int k = 0; k < 6; ++k) tv = i == k ? tc[k] : tv;
            const float pv = sigmoidf_(tv);                                            // utils.py:87
#pragma unroll
            for (int k = 0; k < 6; ++k) pc[k] = i == k ? pv : pc[k];
            pmax = i >= 0 ? fmaxf(pmax, pv) : pmax;
        }
        pmax = row_max(pmax);                                                          // utils.py:150
        unsigned m0 = 0, m1 = 0, m2 = 0;
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            const int c = j + 16 * i;
            if (c < C && pc[i] >= pmax) {                                              // utils.py:151
                if (i < 2) m0 |= 1u << c; else if (i < 4) m1 |= 1u << (c - 32); else m2 |= 1u << (c - 64);
            }
        }
        m0 = row_or(m0); m1 = row_or(m1); m2 = row_or(m2);

        // utils.py:84-92: one transcendental per lane (lanes 0,1,4: sigmoid; lanes 2,3: exp), then shared
        const bool is_wh = j == 2 || j == 3;
        const float e = expf(is_wh ? tb : -tb);
        const float f = is_wh ? e : 1.0f / (1.0f + e);
        const float sx = row_bcast<0>(f), sy = row_bcast<1>(f), ew = row_bcast<2>(f);
        const float eh = row_bcast<3>(f), conf = row_bcast<4>(f);
        const float bx = (sx + (float)gx) * a.d.stride_x[l];
        const float by = (sy + (float)gy) * a.d.stride_y[l];
        const float bw = ew * a.d.anchors[l][an][0];
        const float bh = eh * a.d.anchors[l][an][1];
        // utils.py:118-126: lane j < 4 computes corner j = ymin, xmin, ymax, xmax
        const bool isx = j & 1;
        const float ctr = isx ? bx : by, half = (isx ? bw : bh) / 2.0f, off = (isx ? off_w : off_h) / 2.0f;
        const float corner = (((j & 2) ? ctr + half : ctr - half) - off) / scale;
        const float score = conf * (pmax * 1.0f);                                      // utils.py:153-154

        if (live) {
            float v;
            switch (j) {
                case 0: case 1: case 2: case 3: v = corner; break;
                case 4: v = score; break;
                case 5: v = __uint_as_float(m0); break;
                case 6: v = __uint_as_float(m1); break;
                default: v = __uint_as_float(m2); break;
            }
            if (j < 8) a.records[((long)b * N + p) * 8 + j] = v;
            if (j == 4 && score >= thr) {                                              // inference.py:127
                s_list[atomicAdd(&s_n, 1)] = p;
                unsigned mm[3] = {m0, m1, m2};
                for (int w = 0; w < 3; ++w)
                    while (mm[w]) { const int bit = __ffs(mm[w]) - 1; mm[w] &= mm[w] - 1; atomicAdd(&s_hist[w * 32 + bit], 1); }
            }
        }
    }
    __syncthreads();
    const int n = s_n;
    if (tid == 0) s_base = n ? atomicAdd(&a.cand_count[b], n) : 0;
    if (tid < C && s_hist[tid]) atomicAdd(&a.class_count[b * C + tid], s_hist[tid]);
    __syncthreads();
    const int base = s_base;
    for (int i = tid; i < n; i += kThreads) a.cand_idx[(long)b * N + base + i] = s_list[i];
}


// ---- decode, one lane per box ----------------------------------------------------------------------------
// A wavefront takes 16 consecutive grid cells of one scale (48 boxes): their head values, one contiguous run
// of 16 * head_ld floats, are copied to LDS by LDS-DMA (16-byte coalesced, no registers), one cell per
// instruction with the cells 16 bytes apart in LDS so that lanes reading the same element of different
// cells spread over the banks.  Lane (cell i, anchor a) then walks its own 5 + C values: the class maximum
// costs one ds_read + one v_max per class per 48 boxes instead of a cross-lane reduction per box, and the
// exact sigmoid is evaluated only for logits that can reach the maximum (see `win` below).
// Same arithmetic, in the same order, as decode_kernel: the two produce identical records.
constexpr int kCellsPerWave = 16;
constexpr int kCellPitch = 1024 + 16;                       // bytes between cells in LDS (head_ld <= 256)
constexpr int kWaveLds = kCellsPerWave * kCellPitch;

struct DecodeArgs2 {
    DecodeArgs a;
    int groups[3];          // 16-cell groups per scale
    int groups_per_image, slices;
};

__device__ __forceinline__ void lds_dma16(const void *gsrc, void *lds_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)gsrc,
                                     (__attribute__((address_space(3))) void *)lds_base, 16, 0, 0);
}

__global__ __launch_bounds__(kThreads) void decode_lane_kernel(const DecodeArgs2 q) {
    const DecodeArgs &a = q.a;
    __shared__ __attribute__((aligned(16))) char s_head[4 * kWaveLds];
    __shared__ int s_list[4 * 48 * 8];       // candidates of this block (flushed when nearly full)
    __shared__ int s_hist[kMaxClasses];
    __shared__ int s_n, s_base;

    const int b = blockIdx.y;
    const int N = a.d.boxes_per_image;
    const int C = a.d.num_classes, A = a.d.num_anchors, E = 5 + C;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    if (tid < kMaxClasses) s_hist[tid] = 0;
    if (tid == 0) s_n = 0;
    __syncthreads();

    const float scale = a.letterbox[b * 4 + 0];
    const float off_h = a.letterbox[b * 4 + 1];
    const float off_w = a.letterbox[b * 4 + 2];
    const float thr = a.d.score_threshold;
    char *my = s_head + wave * kWaveLds;
    // This lane's cell within the group and its anchor: cells 0..7 on lanes 0..23, cells 8..15 on lanes 32..55.  Cells
    // c and c + 8 are 8 * 260 dwords apart = the same LDS bank; ds_read_b32 serves a wave in two 32-lane halves, so
    // keeping them in different halves removes the 2-way conflict (35 % of this kernel's LDS cycles before).
    const int l5 = lane & 31;
    const int c8 = l5 / A, an = l5 - c8 * A;
    const int ci = c8 + 8 * (lane >> 5);
    const bool has_box = c8 < 8 && ci < kCellsPerWave;

    for (int t = (blockIdx.x * 4 + wave); t < q.groups_per_image; t += q.slices * 4) {
        const int l = t >= q.groups[0] + q.groups[1] ? 2 : (t >= q.groups[0] ? 1 : 0);
        const int g = t - (l == 2 ? q.groups[0] + q.groups[1] : (l == 1 ? q.groups[0] : 0));
        const int gw = a.d.grid_w[l], ncell = a.d.grid_h[l] * gw, ld = a.d.head_ld[l];
        const int cell0 = g * kCellsPerWave;
        const float *src = a.head[l] + ((long)b * ncell + cell0) * ld;
        const int ncopy = min(kCellsPerWave, ncell - cell0);
        // the previous group's LDS reads of this wave are complete: they were consumed by VALU
        for (int i = 0; i < ncopy; ++i) {                      // wave-uniform trip count
            const int off = lane * 16 < ld * 4 ? lane * 16 : 0;     // lanes past the cell repeat its first chunk
            lds_dma16((const char *)(src + (long)i * ld) + off, my + i * kCellPitch);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

        const int cell = cell0 + ci;
        const bool live = has_box && cell < ncell;
        float *tl = (float *)(my + (live ? ci : 0) * kCellPitch) + an * E;
        const int gy = cell / gw, gx = cell - gy * gw;
        const DecodedBox o = y3_decode_box(tl, live, C, gx, gy, a.d.stride_x[l], a.d.stride_y[l], a.d.anchors[l][an < 3 ? an : 0][0],
                                           a.d.anchors[l][an < 3 ? an : 0][1], scale, off_h, off_w);
        const float ymin = o.ymin, xmin = o.xmin, ymax = o.ymax, xmax = o.xmax, score = o.score;
        const unsigned m0 = o.m0, m1 = o.m1, m2 = o.m2;

        const int p = a.scale_base[l] + cell * A + an;
        if (live) {
            float *r = a.records + ((long)b * N + p) * 8;
            *(f32x4 *)r = f32x4{ymin, xmin, ymax, xmax};
            *(f32x4 *)(r + 4) = f32x4{score, __uint_as_float(m0), __uint_as_float(m1), __uint_as_float(m2)};
        }
        const bool hit = live && score >= thr;                                         // inference.py:127
        const unsigned long long hb = __builtin_amdgcn_ballot_w64(hit);
        if (hb) {
            int wbase = 0;
            if (lane == 0) wbase = atomicAdd(&s_n, __popcll(hb));
            wbase = __builtin_amdgcn_readfirstlane(wbase);
            if (hit) {
                s_list[wbase + __popcll(hb & ((1ull << lane) - 1ull))] = p;
                unsigned mm[3] = {m0, m1, m2};
                for (int w = 0; w < 3; ++w)
                    while (mm[w]) { const int bit = __ffs(mm[w]) - 1; mm[w] &= mm[w] - 1; atomicAdd(&s_hist[w * 32 + bit], 1); }
            }
        }
        // flush the block's candidate list before it could overflow (block-uniform decision needs a barrier, so
        // the list is sized for every group this block will ever take: see the launch)
    }
    __syncthreads();
    const int n = s_n;
    if (tid == 0) s_base = n ? atomicAdd(&a.cand_count[b], n) : 0;
    if (tid < C && s_hist[tid]) atomicAdd(&a.class_count[b * C + tid], s_hist[tid]);
    __syncthreads();
    const int base = s_base;
    for (int i = tid; i < n; i += kThreads) a.cand_idx[(long)b * N + base + i] = s_list[i];
}

// Dense forms (API parity with the reference helpers; not on the fused hot path).
__global__ __launch_bounds__(kThreads) void get_head_kernel(const float *head, long total, int gh, int gw, int ld, int A,
                                                            int C, const float *anchors, float sx, float sy,
                                                            float *xy, float *wh, float *conf, float *cls) {
    const int E = 5 + C;
    for (long i = (long)blockIdx.x * kThreads + threadIdx.x; i < total; i += (long)gridDim.x * kThreads) {
        const int k = (int)(i % E);
        const long box = i / E;
        const int an = (int)(box % A);
        const long cellb = box / A;                  // b*gh*gw + cell
        const int cell = (int)(cellb % ((long)gh * gw));
        const int gy = cell / gw, gx = cell - gy * gw;
        const float t = head[cellb * ld + an * E + k];
        if (k == 0) xy[box * 2] = (sigmoidf_(t) + (float)gx) * sx;
        else if (k == 1) xy[box * 2 + 1] = (sigmoidf_(t) + (float)gy) * sy;
        else if (k < 4) wh[box * 2 + (k - 2)] = expf(t) * anchors[an * 2 + (k - 2)];
        else if (k == 4) conf[box] = sigmoidf_(t);
        else cls[box * C + (k - 5)] = sigmoidf_(t);
    }
}

__global__ __launch_bounds__(kThreads) void correct_boxes_kernel(const float *xy, const float *wh, long n, float scale,
                                                                 float off_h, float off_w, float *boxes) {
    for (long i = (long)blockIdx.x * kThreads + threadIdx.x; i < n; i += (long)gridDim.x * kThreads) {
        const float bx = xy[i * 2], by = xy[i * 2 + 1];
        const float hw = wh[i * 2] / 2.0f, hh = wh[i * 2 + 1] / 2.0f;
        boxes[i * 4 + 0] = ((by - hh) - off_h / 2.0f) / scale;
        boxes[i * 4 + 1] = ((bx - hw) - off_w / 2.0f) / scale;
        boxes[i * 4 + 2] = ((by + hh) - off_h / 2.0f) / scale;
        boxes[i * 4 + 3] = ((bx + hw) - off_w / 2.0f) / scale;
    }
}

__global__ __launch_bounds__(kThreads) void expand_scores_kernel(const float *rec, long n, int C, float *boxes, float *scores) {
    const long total = n * C;
    for (long i = (long)blockIdx.x * kThreads + threadIdx.x; i < total; i += (long)gridDim.x * kThreads) {
        const int c = (int)(i % C);
        const long box = i / C;
        const float *r = rec + box * 8;
        const unsigned m = __float_as_uint(r[5 + (c >> 5)]);
        scores[i] = ((m >> (c & 31)) & 1u) ? r[4] : 0.0f;
        if (c < 4) boxes[box * 4 + c] = r[c];
    }
}

inline unsigned grid_for(long work) {
    long b = (work + kThreads - 1) / kThreads;
    if (b < 1) b = 1;
    return (unsigned)(b > 2048 ? 2048 : b);
}

}  // namespace

extern "C" int y3_decode(const y3_decode_desc *d, const float *head0, const float *head1, const float *head2,
                         const float *letterbox, float *records, int32_t *cand_idx, int32_t *cand_count,
                         int32_t *class_count, void *stream) {
    Y3_REQUIRE(d && head0 && letterbox && records && cand_idx && cand_count && class_count, Y3_E_INVALID, "decode: null pointer");
    Y3_REQUIRE(d->num_scales >= 1 && d->num_scales <= 3, Y3_E_INVALID, "decode: %d scales", d->num_scales);
    Y3_REQUIRE(d->num_classes >= 1 && d->num_classes <= kMaxClasses, Y3_E_INVALID, "decode: %d classes (max %d)", d->num_classes, kMaxClasses);
    Y3_REQUIRE(d->num_anchors >= 1 && d->num_anchors <= 3, Y3_E_INVALID, "decode: %d anchors per scale", d->num_anchors);
    Y3_REQUIRE(d->batch >= 1 && d->batch <= 65535, Y3_E_INVALID, "decode: batch %d", d->batch);
    DecodeArgs a;
    a.d = *d;
    const float *heads[3] = {head0, head1, head2};
    int base = 0;
    for (int l = 0; l < 3; ++l) {
        a.scale_base[l] = base;
        a.head[l] = heads[l < d->num_scales ? l : 0];
        if (l < d->num_scales) {
            Y3_REQUIRE(heads[l] != nullptr, Y3_E_INVALID, "decode: head %d is null", l);
            Y3_REQUIRE(d->grid_h[l] > 0 && d->grid_w[l] > 0, Y3_E_INVALID, "decode: grid %d empty", l);
            Y3_REQUIRE(d->head_ld[l] >= d->num_anchors * (5 + d->num_classes), Y3_E_INVALID, "decode: head_ld[%d]=%d too small", l, d->head_ld[l]);
            base += d->grid_h[l] * d->grid_w[l] * d->num_anchors;
        } else {
            a.scale_base[l] = 0x7fffffff;
            a.d.grid_h[l] = a.d.grid_w[l] = 1;
        }
    }
    a.scale_base[3] = base;
    Y3_REQUIRE(base == d->boxes_per_image, Y3_E_INVALID, "decode: boxes_per_image %d, grids give %d", d->boxes_per_image, base);
    a.letterbox = letterbox; a.records = records; a.cand_idx = cand_idx; a.cand_count = cand_count; a.class_count = class_count;
    // lane-per-box kernel: 3 anchors x 16 cells fit a wavefront, a cell fits one LDS-DMA instruction, 16-byte
    // aligned cells; anything else takes the 16-lanes-per-box kernel
    bool lane_ok = d->num_anchors == 3 && !g_decode_force_v1;
    for (int l = 0; l < d->num_scales; ++l)
        lane_ok = lane_ok && d->head_ld[l] <= 256 && d->head_ld[l] % 4 == 0 && aligned16(heads[l]);
    if (lane_ok) {
        DecodeArgs2 q;
        q.a = a;
        q.groups_per_image = 0;
        for (int l = 0; l < 3; ++l) {
            q.groups[l] = l < d->num_scales ? (d->grid_h[l] * d->grid_w[l] + kCellsPerWave - 1) / kCellsPerWave : 0;
            q.groups_per_image += q.groups[l];
        }
        // every block takes at most 8 groups per wave: its LDS candidate list holds 4 * 48 * 8 entries
        q.slices = (q.groups_per_image + 31) / 32;
        dim3 grid((unsigned)q.slices, (unsigned)d->batch);
        hipLaunchKernelGGL(decode_lane_kernel, grid, dim3(kThreads), 0, (hipStream_t)stream, q);
        return y3_check_launch("decode");
    }
    dim3 grid((unsigned)((base + kChunk - 1) / kChunk), (unsigned)d->batch);
    hipLaunchKernelGGL(decode_kernel, grid, dim3(kThreads), 0, (hipStream_t)stream, a);
    return y3_check_launch("decode");
}

#ifdef Y3_EXPERIMENTS
// Test knob (libyolo3hip_exp.so only): 1 = always the 16-lanes-per-box kernel, 0 = automatic.
extern "C" int y3_decode_variant(int v) {
    g_decode_force_v1 = v == 1;
    return Y3_OK;
}
#endif

extern "C" int y3_get_head(const float *head, int batch, int grid_h, int grid_w, int head_ld, int num_anchors,
                           int num_classes, const float *anchors_wh, float stride_x, float stride_y,
                           float *box_xy, float *box_wh, float *conf, float *cls, void *stream) {
    Y3_REQUIRE(head && anchors_wh && box_xy && box_wh && conf && cls, Y3_E_INVALID, "get_head: null pointer");
    Y3_REQUIRE(batch > 0 && grid_h > 0 && grid_w > 0 && num_anchors > 0 && num_classes > 0, Y3_E_INVALID, "get_head: bad shape");
    Y3_REQUIRE(head_ld >= num_anchors * (5 + num_classes), Y3_E_INVALID, "get_head: head_ld too small");
    const long total = (long)batch * grid_h * grid_w * num_anchors * (5 + num_classes);
    hipLaunchKernelGGL(get_head_kernel, dim3(grid_for(total)), dim3(kThreads), 0, (hipStream_t)stream, head, total,
                       grid_h, grid_w, head_ld, num_anchors, num_classes, anchors_wh, stride_x, stride_y,
                       box_xy, box_wh, conf, cls);
    return y3_check_launch("get_head");
}

extern "C" int y3_correct_boxes(const float *box_xy, const float *box_wh, int64_t n, float scale, float off_h,
                                float off_w, float *boxes, void *stream) {
    Y3_REQUIRE(box_xy && box_wh && boxes && n > 0, Y3_E_INVALID, "correct_boxes: bad arguments");
    hipLaunchKernelGGL(correct_boxes_kernel, dim3(grid_for(n)), dim3(kThreads), 0, (hipStream_t)stream, box_xy,
                       box_wh, (long)n, scale, off_h, off_w, boxes);
    return y3_check_launch("correct_boxes");
}

extern "C" int y3_expand_scores(const float *records, int64_t n, int num_classes, float *boxes, float *box_scores,
                                void *stream) {
    Y3_REQUIRE(records && boxes && box_scores && n > 0, Y3_E_INVALID, "expand_scores: bad arguments");
    Y3_REQUIRE(num_classes >= 4 && num_classes <= kMaxClasses, Y3_E_INVALID, "expand_scores: %d classes", num_classes);
    hipLaunchKernelGGL(expand_scores_kernel, dim3(grid_for(n * num_classes)), dim3(kThreads), 0, (hipStream_t)stream,
                       records, (long)n, num_classes, boxes, box_scores);
    return y3_check_launch("expand_scores");
}
