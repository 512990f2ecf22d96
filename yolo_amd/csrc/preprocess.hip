// Letterbox pre-process on the GPU: the device counterpart of the reference's resize_image
// (utils/utils.py:7-40: aspect-preserving cv2.INTER_CUBIC resize, centre pad with 128) followed by the
// `/255.` of inference.py:242, straight into the fp32 NHWC tensor the network reads.
// One thread per output pixel (three channels).  The bicubic is Keys' kernel with a = -0.75, source
// coordinate (x + 0.5) * scale - 0.5, replicated border, evaluated in float64 in exactly the order of
// oracle/letterbox.py (columns first, taps in order, no contraction: -ffp-contract=off), so the uint8 result
// and the fp32 output equal the oracle's bit for bit.  (cv2 itself uses 11-bit fixed point: parity with
// the real library is unpinned, see the oracle's header.)
// HBM-bound: 3 source bytes in (through L2, 16 taps) and 12 bytes out per pixel.
#include "common.h"

namespace {

struct LetterboxArgs {
    const unsigned char *img;
    float *out;
    int h, w, stride, in_h, in_w, new_h, new_w, top, left;
    double sx, sy;
};

__device__ __forceinline__ void cubic(double t, double (&w)[4]) {
    const double a = -0.75;
    w[0] = ((a * (t + 1) - 5 * a) * (t + 1) + 8 * a) * (t + 1) - 4 * a;
    w[1] = ((a + 2) * t - (a + 3)) * t * t + 1;
    w[2] = ((a + 2) * (1 - t) - (a + 3)) * (1 - t) * (1 - t) + 1;
    w[3] = 1.0 - w[0] - w[1] - w[2];
}

__global__ __launch_bounds__(256) void letterbox_kernel(const LetterboxArgs a) {
    const long total = (long)a.in_h * a.in_w;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int oy = (int)(i / a.in_w), ox = (int)(i - (long)oy * a.in_w);
        const int y = oy - a.top, x = ox - a.left;
        double v[3] = {128.0, 128.0, 128.0};
        if ((unsigned)y < (unsigned)a.new_h && (unsigned)x < (unsigned)a.new_w) {
            if (a.new_h == a.h && a.new_w == a.w) {
                const unsigned char *p = a.img + (long)y * a.stride + x * 3;
                v[0] = p[0]; v[1] = p[1]; v[2] = p[2];
            } else {
                const double fx = (x + 0.5) * a.sx - 0.5, fy = (y + 0.5) * a.sy - 0.5;
                const double bx = floor(fx), by = floor(fy);
                double wx[4], wy[4];
                cubic(fx - bx, wx);
                cubic(fy - by, wy);
                int ix[4], iy[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    ix[k] = min(max((int)bx + k - 1, 0), a.w - 1) * 3;
                    iy[k] = min(max((int)by + k - 1, 0), a.h - 1);
                }
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    double acc = 0.0;
#pragma unroll
                    for (int ky = 0; ky < 4; ++ky) {
                        const unsigned char *row = a.img + (long)iy[ky] * a.stride + c;
                        double r = 0.0;
#pragma unroll
                        for (int kx = 0; kx < 4; ++kx) r = r + wx[kx] * (double)row[ix[kx]];
                        acc = acc + wy[ky] * r;
                    }
                    v[c] = fmin(fmax(floor(acc + 0.5), 0.0), 255.0);
                }
            }
        }
        float *o = a.out + i * 3;
        o[0] = (float)(v[0] / 255.0); o[1] = (float)(v[1] / 255.0); o[2] = (float)(v[2] / 255.0);
    }
}

}  // namespace

extern "C" int y3_letterbox_u8(const uint8_t *image, int h, int w, int row_stride_bytes, float *out, int in_h,
                               int in_w, void *stream) {
    Y3_REQUIRE(image && out, Y3_E_INVALID, "letterbox: null pointer");
    Y3_REQUIRE(h > 0 && w > 0 && in_h > 0 && in_w > 0 && row_stride_bytes >= 3 * w, Y3_E_INVALID,
               "letterbox: bad shape %dx%d (stride %d) -> %dx%d", h, w, row_stride_bytes, in_h, in_w);
    LetterboxArgs a;
    a.img = image; a.out = out; a.h = h; a.w = w; a.stride = row_stride_bytes; a.in_h = in_h; a.in_w = in_w;
    const double sw = (double)in_w / (double)w, sh = (double)in_h / (double)h;
    const double scale = sw < sh ? sw : sh;                                  // utils/utils.py:20
    a.new_w = (int)((double)w * scale); a.new_h = (int)((double)h * scale);    // int() truncation, :22-23
    Y3_REQUIRE(a.new_w > 0 && a.new_h > 0, Y3_E_INVALID, "letterbox: image %dx%d collapses at %dx%d", h, w, in_h, in_w);
    a.top = (in_h - a.new_h) / 2; a.left = (in_w - a.new_w) / 2;             // :34-37
    a.sx = (double)w / (double)a.new_w; a.sy = (double)h / (double)a.new_h;
    const long total = (long)in_h * in_w;
    long blocks = (total + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(letterbox_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, a);
    return y3_check_launch("letterbox");
}
