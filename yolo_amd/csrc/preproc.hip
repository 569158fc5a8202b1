// Image pre-processing on the device: uint8 HWC camera/decoder images -> the network's input tensor, one launch per batch.
//
// Stands for the host chain of the reference: cv2.resize (bilinear) + letterbox border + BGR->RGB + HWC->CHW + /255
// (yolov5_official/utils.py:40-75 letterbox, yolov5_official/val.py:68-69) and cv2.resize + /255 - mean / std + channel
// swap (data/transformers.py:456-471 base_transform, data/dataset.py:224-229).
//
// Resampling restates OpenCV's generic 8-bit INTER_LINEAR path (imgproc resize.cpp): sample position
// (d + 0.5) * scale - 0.5 with scale = 1 / (dst / src) in double, coefficients rounded to 11-bit fixed point, a
// horizontal pass into 32-bit integers and the vertical pass ((b0 * (S0 >> 4)) >> 16) + ((b1 * (S1 >> 4)) >> 16) + 2) >> 2;
// an exact 2x reduction is the 2x2 box mean (a + b + c + d + 2) >> 2, as OpenCV substitutes there.
// Bound: HBM (source bytes read once through L2, output written once).
#include "ydet_internal.h"

namespace {

struct PreArgs {
    const ydet_u8_image* images;
    void* out;
    int out_h, out_w, out_dtype, layout, swap_rb, scale_mode;
    int pad[3];
    float mean[3], stdv[3];
};

struct Tap {
    int i0, i1;     // the two source indices (already clamped)
    int c0, c1;     // 11-bit coefficients
};

// Source taps of destination index d along one axis of length `src` resized to `dst`.
__device__ __forceinline__ Tap axis_tap(int d, int src, int dst, bool zero_at_edge) {
    const double scale = 1.0 / ((double)dst / (double)src);
    float f = (float)(((double)d + 0.5) * scale - 0.5);
    int s = (int)floorf(f);
    f -= (float)s;
    Tap t;
    if (zero_at_edge) {              // horizontal: the coefficient is dropped where a neighbour is missing
        if (s < 0) { f = 0.f; s = 0; }
        if (s >= src - 1) { f = 0.f; s = src - 1; }
        t.i0 = s;
        t.i1 = min(s + 1, src - 1);
    } else {                         // vertical: rows are clamped, coefficients kept
        t.i0 = min(max(s, 0), src - 1);
        t.i1 = min(max(s + 1, 0), src - 1);
    }
    t.c0 = __float2int_rn((1.f - f) * 2048.f);
    t.c1 = __float2int_rn(f * 2048.f);
    return t;
}

template <typename TOUT>
__device__ __forceinline__ TOUT finish(float v);
template <> __device__ __forceinline__ float finish<float>(float v) { return v; }
template <> __device__ __forceinline__ __bf16 finish<__bf16>(float v) { return (__bf16)v; }
template <> __device__ __forceinline__ uint8_t finish<uint8_t>(float v) { return (uint8_t)v; }

template <typename TOUT>
__global__ __launch_bounds__(256) void preprocess_u8_kernel(PreArgs a) {
    const int ox = blockIdx.x * 64 + (threadIdx.x & 63);
    const int oy = blockIdx.y * 4 + (threadIdx.x >> 6);
    const int n = blockIdx.z;
    if (ox >= a.out_w || oy >= a.out_h) return;
    const ydet_u8_image im = a.images[n];
    int v[3] = {a.pad[0], a.pad[1], a.pad[2]};
    const int rx = ox - im.left, ry = oy - im.top;
    if ((unsigned)rx < (unsigned)im.rw && (unsigned)ry < (unsigned)im.rh) {
        const uint8_t* src = im.data;
        if (im.rw == im.w && im.rh == im.h) {
            const uint8_t* p = src + (size_t)ry * im.pitch + rx * 3;
            v[0] = p[0]; v[1] = p[1]; v[2] = p[2];
        } else if (im.w == 2 * im.rw && im.h == 2 * im.rh) {
            const uint8_t* p = src + (size_t)(2 * ry) * im.pitch + 2 * rx * 3;
            const uint8_t* q = p + im.pitch;
#pragma unroll
            for (int c = 0; c < 3; ++c) v[c] = (p[c] + p[3 + c] + q[c] + q[3 + c] + 2) >> 2;
        } else {
            const Tap tx = axis_tap(rx, im.w, im.rw, true);
            const Tap ty = axis_tap(ry, im.h, im.rh, false);
            const uint8_t* r0 = src + (size_t)ty.i0 * im.pitch;
            const uint8_t* r1 = src + (size_t)ty.i1 * im.pitch;
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const int s0 = r0[tx.i0 * 3 + c] * tx.c0 + r0[tx.i1 * 3 + c] * tx.c1;
                const int s1 = r1[tx.i0 * 3 + c] * tx.c0 + r1[tx.i1 * 3 + c] * tx.c1;
                v[c] = (((ty.c0 * (s0 >> 4)) >> 16) + ((ty.c1 * (s1 >> 4)) >> 16) + 2) >> 2;
            }
        }
    }
    TOUT* out = reinterpret_cast<TOUT*>(a.out);
    const size_t plane = (size_t)a.out_h * a.out_w;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const int sc = a.swap_rb ? 2 - c : c;        // output channel c shows source channel sc
        float f = (float)v[sc];
        if (a.scale_mode >= 1) f = f / 255.f;
        if (a.scale_mode == 2) f = (f - a.mean[sc]) / a.stdv[sc];
        const size_t at = a.layout == 0 ? ((size_t)n * 3 + c) * plane + (size_t)oy * a.out_w + ox
                                        : ((size_t)n * plane + (size_t)oy * a.out_w + ox) * 3 + c;
        out[at] = finish<TOUT>(f);
    }
}

}  // namespace

extern "C" int ydet_preprocess_u8(const ydet_u8_image* images, const ydet_preproc_desc* d, void* out, void* stream) {
    ydet::clear_error();
    YDET_CHECK_ARG(d != nullptr, "preprocess: null descriptor");
    YDET_CHECK_ARG(d->N >= 0 && d->out_h > 0 && d->out_w > 0, "preprocess: bad output shape %d x %d x %d", d->N, d->out_h, d->out_w);
    YDET_CHECK_ARG(d->N <= 65535, "preprocess: at most 65535 images per call (got %d)", d->N);
    YDET_CHECK_ARG(d->out_dtype == YDET_F32 || d->out_dtype == YDET_BF16 || d->out_dtype == YDET_U8,
                   "preprocess: bad output dtype %d", d->out_dtype);
    YDET_CHECK_ARG(d->layout == 0 || d->layout == 1, "preprocess: layout must be 0 (NCHW) or 1 (NHWC)");
    YDET_CHECK_ARG(d->scale_mode >= 0 && d->scale_mode <= 2, "preprocess: bad scale mode %d", d->scale_mode);
    YDET_CHECK_ARG(d->out_dtype != YDET_U8 || d->scale_mode == 0, "preprocess: uint8 output carries raw values only");
    for (int c = 0; c < 3; ++c) {
        YDET_CHECK_ARG(d->pad_value[c] >= 0 && d->pad_value[c] <= 255, "preprocess: border value %d", d->pad_value[c]);
        YDET_CHECK_ARG(d->scale_mode != 2 || d->std[c] != 0.f, "preprocess: std[%d] is zero", c);
    }
    if (d->N == 0) return YDET_OK;
    YDET_CHECK_ARG(images != nullptr && out != nullptr, "preprocess: null pointer");
    PreArgs a;
    a.images = images; a.out = out;
    a.out_h = d->out_h; a.out_w = d->out_w; a.out_dtype = d->out_dtype; a.layout = d->layout;
    a.swap_rb = d->swap_rb != 0; a.scale_mode = d->scale_mode;
    for (int c = 0; c < 3; ++c) { a.pad[c] = d->pad_value[c]; a.mean[c] = d->mean[c]; a.stdv[c] = d->std[c]; }
    const dim3 grid((d->out_w + 63) / 64, (d->out_h + 3) / 4, d->N);
    YDET_CHECK_ARG(grid.y <= 65535, "preprocess: output height %d too large", d->out_h);
    hipStream_t s = ydet::as_stream(stream);
    if (d->out_dtype == YDET_F32) preprocess_u8_kernel<float><<<grid, 256, 0, s>>>(a);
    else if (d->out_dtype == YDET_BF16) preprocess_u8_kernel<__bf16><<<grid, 256, 0, s>>>(a);
    else preprocess_u8_kernel<uint8_t><<<grid, 256, 0, s>>>(a);
    YDET_CHECK_LAUNCH("preprocess_u8");
    return YDET_OK;
}
