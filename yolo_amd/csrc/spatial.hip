// Pooling, resampling and layout kernels on NHWC channel slices (gfx950).  All of them are HBM-bound
// element movers: one thread handles one pixel x one 16-byte channel vector (8 bf16 / 4 fp32), so a
// wave reads and writes whole contiguous channel runs.  Slices whose channel count or offsets are not
// multiples of the vector width fall back to one element per thread.
#include "ydet_internal.h"

using namespace ydet;

namespace {

template <typename T, int V>
struct Vec {
    T v[V];
};

template <typename T, int V>
__device__ __forceinline__ Vec<T, V> vload(const T* p) {
    Vec<T, V> r;
    if constexpr (V * sizeof(T) == 16) {
        *reinterpret_cast<uint4*>(r.v) = *reinterpret_cast<const uint4*>(p);
    } else {
#pragma unroll
        for (int i = 0; i < V; ++i) r.v[i] = p[i];
    }
    return r;
}
template <typename T, int V>
__device__ __forceinline__ void vstore(T* p, const Vec<T, V>& r) {
    if constexpr (V * sizeof(T) == 16) {
        *reinterpret_cast<uint4*>(p) = *reinterpret_cast<const uint4*>(r.v);
    } else {
#pragma unroll
        for (int i = 0; i < V; ++i) p[i] = r.v[i];
    }
}
template <typename T, int V>
__device__ __forceinline__ void vmax(Vec<T, V>& a, const Vec<T, V>& b) {
#pragma unroll
    for (int i = 0; i < V; ++i) a.v[i] = to_f32(b.v[i]) > to_f32(a.v[i]) ? b.v[i] : a.v[i];
}

struct SliceArgs {
    const char* x;
    char* y;
    int N, H, W, C, x_ld, x_coff, y_ld, y_coff;
    int Ho, Wo;
    long long total;  // output pixels * channel vectors
    int cv;           // channel vectors per pixel
    int mode;
};

#define SLICE_INDEX()                                                                  \
    const long long gid = (long long)blockIdx.x * blockDim.x + threadIdx.x;             \
    if (gid >= a.total) return;                                                         \
    const int cvi = (int)(gid % a.cv);                                                  \
    long long p = gid / a.cv;                                                           \
    const int wo = (int)(p % a.Wo);                                                     \
    p /= a.Wo;                                                                          \
    const int ho = (int)(p % a.Ho);                                                     \
    const int n = (int)(p / a.Ho);                                                      \
    const int c = cvi * V;                                                              \
    const T* xin = reinterpret_cast<const T*>(a.x);                                     \
    T* yout = reinterpret_cast<T*>(a.y);                                                \
    (void)xin; (void)yout; (void)c; (void)n; (void)ho; (void)wo;

#define XAT(nn, hh, ww) (xin + ((size_t)((size_t)(nn) * a.H + (hh)) * a.W + (ww)) * a.x_ld + a.x_coff + c)
#define YAT(nn, hh, ww) (yout + ((size_t)((size_t)(nn) * a.Ho + (hh)) * a.Wo + (ww)) * a.y_ld + a.y_coff + c)

// MaxPool2d(2,2)
template <typename T, int V>
__global__ void maxpool2_kernel(const SliceArgs a) {
    SLICE_INDEX();
    Vec<T, V> m = vload<T, V>(XAT(n, 2 * ho, 2 * wo));
    vmax(m, vload<T, V>(XAT(n, 2 * ho, 2 * wo + 1)));
    vmax(m, vload<T, V>(XAT(n, 2 * ho + 1, 2 * wo)));
    vmax(m, vload<T, V>(XAT(n, 2 * ho + 1, 2 * wo + 1)));
    vstore<T, V>(YAT(n, ho, wo), m);
}

// ZeroPad2d((0,1,0,1)) + MaxPool2d(2, stride 1): the zero padding competes in the max.
template <typename T, int V>
__global__ void maxpool2s1_kernel(const SliceArgs a) {
    SLICE_INDEX();
    Vec<T, V> m = vload<T, V>(XAT(n, ho, wo));
    Vec<T, V> z;
#pragma unroll
    for (int i = 0; i < V; ++i) z.v[i] = from_f32<T>(0.f);
    const bool r = wo + 1 < a.W, b = ho + 1 < a.H;
    vmax(m, r ? vload<T, V>(XAT(n, ho, wo + 1)) : z);
    vmax(m, b ? vload<T, V>(XAT(n, ho + 1, wo)) : z);
    vmax(m, (r && b) ? vload<T, V>(XAT(n, ho + 1, wo + 1)) : z);
    vstore<T, V>(YAT(n, ho, wo), m);
}

// SPP: 5x5, 9x9, 13x13 stride-1 max pools (implicit -inf padding) in one sweep of the 13x13 window.
template <typename T, int V>
__global__ void spp_kernel(const SliceArgs a) {
    SLICE_INDEX();
    float m5[V], m9[V], m13[V];
#pragma unroll
    for (int i = 0; i < V; ++i) m5[i] = m9[i] = m13[i] = -INFINITY;
    for (int dy = -6; dy <= 6; ++dy) {
        const int hh = ho + dy;
        if ((unsigned)hh >= (unsigned)a.H) continue;
        const int ady = dy < 0 ? -dy : dy;
        for (int dx = -6; dx <= 6; ++dx) {
            const int ww = wo + dx;
            if ((unsigned)ww >= (unsigned)a.W) continue;
            const int adx = dx < 0 ? -dx : dx;
            const int r = ady > adx ? ady : adx;
            const Vec<T, V> v = vload<T, V>(XAT(n, hh, ww));
#pragma unroll
            for (int i = 0; i < V; ++i) {
                const float f = to_f32(v.v[i]);
                m13[i] = fmaxf(m13[i], f);
                if (r <= 4) m9[i] = fmaxf(m9[i], f);
                if (r <= 2) m5[i] = fmaxf(m5[i], f);
            }
        }
    }
    Vec<T, V> o;
    T* y0 = YAT(n, ho, wo);
#pragma unroll
    for (int i = 0; i < V; ++i) o.v[i] = from_f32<T>(m5[i]);
    vstore<T, V>(y0, o);
#pragma unroll
    for (int i = 0; i < V; ++i) o.v[i] = from_f32<T>(m9[i]);
    vstore<T, V>(y0 + a.C, o);
#pragma unroll
    for (int i = 0; i < V; ++i) o.v[i] = from_f32<T>(m13[i]);
    vstore<T, V>(y0 + 2 * a.C, o);
}

// SPP for small maps (H*W <= kSppMaxHW, every SPP in the reference models runs on 19x19 .. 20x20): one
// workgroup owns one image x 4 channel vectors, keeps the plane in LDS and builds the three pools as a
// cascade of separable 5-wide max filters (5x5, then 5x5 of that = 9x9, then again = 13x13: exact for max,
// and implicit -inf padding is preserved because out-of-range taps are simply skipped at every stage).
constexpr int kSppMaxHW = 512;
constexpr int kSppGV = 4;   // channel vectors (16 bytes each) per workgroup

// bf16 planes are kept in LDS as order-preserving 16-bit integers (the low 15 bits of a negative value inverted: the
// transform is its own inverse), so that a tap is four packed 16-bit integer maxima per 16-byte vector instead of
// eight unpack / compare / repack sequences: the passes were bound by that arithmetic, not by LDS.  (For finite values
// the result is fmaxf's; +0 wins over -0.)
typedef short short8v __attribute__((ext_vector_type(8)));
__device__ __forceinline__ short8v bf16_order_keys(short8v v) { return v ^ ((v >> 15) & (short)0x7fff); }

template <typename T, int V>
__global__ __launch_bounds__(256) void spp_lds_kernel(const SliceArgs a) {
    __shared__ __attribute__((aligned(16))) T buf[2][kSppMaxHW * kSppGV * V];
    const int hw = a.H * a.W;
    const int groups = a.cv / kSppGV;           // host guarantees cv % kSppGV == 0
    const int n = blockIdx.x / groups;
    const int c = (blockIdx.x - n * groups) * kSppGV * V;
    const T* xin = reinterpret_cast<const T*>(a.x) + (size_t)n * hw * a.x_ld + a.x_coff + c;
    T* yout = reinterpret_cast<T*>(a.y) + (size_t)n * hw * a.y_ld + a.y_coff + c;
    const int items = hw * kSppGV;
    if constexpr (sizeof(T) == 2 && V == 8) {
        short8v* k0 = reinterpret_cast<short8v*>(&buf[0][0]);
        short8v* k1 = reinterpret_cast<short8v*>(&buf[1][0]);
        for (int it = threadIdx.x; it < items; it += 256) {
            const int p = it / kSppGV, v = it - p * kSppGV;
            k0[it] = bf16_order_keys(*reinterpret_cast<const short8v*>(xin + (size_t)p * a.x_ld + v * V));
        }
        __syncthreads();
        for (int stage = 0; stage < 3; ++stage) {
            for (int it = threadIdx.x; it < items; it += 256) {          // row pass: k0 -> k1
                const int p = it / kSppGV, v = it - p * kSppGV;
                const int h = p / a.W, w = p - h * a.W;
                short8v m = k0[it];
#pragma unroll
                for (int d = -2; d <= 2; ++d) {
                    if (d == 0) continue;
                    if ((unsigned)(w + d) < (unsigned)a.W) m = __builtin_elementwise_max(m, k0[it + d * kSppGV]);
                }
                k1[it] = m;
            }
            __syncthreads();
            for (int it = threadIdx.x; it < items; it += 256) {          // column pass: k1 -> k0 and this stage's output slice
                const int p = it / kSppGV, v = it - p * kSppGV;
                const int h = p / a.W;
                short8v m = k1[it];
#pragma unroll
                for (int d = -2; d <= 2; ++d) {
                    if (d == 0) continue;
                    if ((unsigned)(h + d) < (unsigned)a.H) m = __builtin_elementwise_max(m, k1[it + d * a.W * kSppGV]);
                }
                k0[it] = m;
                *reinterpret_cast<short8v*>(yout + (size_t)p * a.y_ld + stage * a.C + v * V) = bf16_order_keys(m);
            }
            __syncthreads();
        }
        return;
    }
    for (int it = threadIdx.x; it < items; it += 256) {
        const int p = it / kSppGV, v = it - p * kSppGV;
        vstore<T, V>(&buf[0][it * V], vload<T, V>(xin + (size_t)p * a.x_ld + v * V));
    }
    __syncthreads();
    for (int stage = 0; stage < 3; ++stage) {
        // row pass: buf[0] -> buf[1]
        for (int it = threadIdx.x; it < items; it += 256) {
            const int p = it / kSppGV, v = it - p * kSppGV;
            const int h = p / a.W, w = p - h * a.W;
            Vec<T, V> m = vload<T, V>(&buf[0][it * V]);
#pragma unroll
            for (int d = -2; d <= 2; ++d) {
                if (d == 0) continue;
                const int ww = w + d;
                if ((unsigned)ww < (unsigned)a.W) vmax(m, vload<T, V>(&buf[0][((h * a.W + ww) * kSppGV + v) * V]));
            }
            vstore<T, V>(&buf[1][it * V], m);
        }
        __syncthreads();
        // column pass: buf[1] -> buf[0] and the output slice of this stage
        for (int it = threadIdx.x; it < items; it += 256) {
            const int p = it / kSppGV, v = it - p * kSppGV;
            const int h = p / a.W, w = p - h * a.W;
            Vec<T, V> m = vload<T, V>(&buf[1][it * V]);
#pragma unroll
            for (int d = -2; d <= 2; ++d) {
                if (d == 0) continue;
                const int hh = h + d;
                if ((unsigned)hh < (unsigned)a.H) vmax(m, vload<T, V>(&buf[1][((hh * a.W + w) * kSppGV + v) * V]));
            }
            vstore<T, V>(&buf[0][it * V], m);
            vstore<T, V>(yout + (size_t)p * a.y_ld + stage * a.C + v * V, m);
        }
        __syncthreads();
    }
}

// Upsample x2: mode 0 nearest, mode 1 bilinear with align_corners=True (fp32 interpolation, the
// operation order of ATen's upsample_bilinear2d: w0h*(w0w*v00 + w1w*v01) + w1h*(w0w*v10 + w1w*v11)).
template <typename T, int V>
__global__ void upsample2x_kernel(const SliceArgs a) {
    SLICE_INDEX();
    if (a.mode == 0) {
        vstore<T, V>(YAT(n, ho, wo), vload<T, V>(XAT(n, ho >> 1, wo >> 1)));
        return;
    }
    const float sh = a.Ho > 1 ? (float)(a.H - 1) / (float)(a.Ho - 1) : 0.f;
    const float sw = a.Wo > 1 ? (float)(a.W - 1) / (float)(a.Wo - 1) : 0.f;
    const float fh = sh * (float)ho, fw = sw * (float)wo;
    const int h0 = (int)fh, w0 = (int)fw;
    const int h1 = h0 + (h0 < a.H - 1 ? 1 : 0), w1 = w0 + (w0 < a.W - 1 ? 1 : 0);
    const float lh1 = fh - (float)h0, lw1 = fw - (float)w0;
    const float lh0 = 1.f - lh1, lw0 = 1.f - lw1;
    const Vec<T, V> v00 = vload<T, V>(XAT(n, h0, w0)), v01 = vload<T, V>(XAT(n, h0, w1));
    const Vec<T, V> v10 = vload<T, V>(XAT(n, h1, w0)), v11 = vload<T, V>(XAT(n, h1, w1));
    Vec<T, V> o;
#pragma unroll
    for (int i = 0; i < V; ++i) {
        const float top = lw0 * to_f32(v00.v[i]) + lw1 * to_f32(v01.v[i]);
        const float bot = lw0 * to_f32(v10.v[i]) + lw1 * to_f32(v11.v[i]);
        o.v[i] = from_f32<T>(lh0 * top + lh1 * bot);
    }
    vstore<T, V>(YAT(n, ho, wo), o);
}

// Space to depth.  mode 0 reorg: out channel (sy*2+sx)*C + c; mode 1 focus: q = sx*2+sy.
template <typename T, int V>
__global__ void space2depth_kernel(const SliceArgs a) {
    SLICE_INDEX();
    T* y0 = YAT(n, ho, wo);
#pragma unroll
    for (int sy = 0; sy < 2; ++sy)
#pragma unroll
        for (int sx = 0; sx < 2; ++sx) {
            const int qi = a.mode == 0 ? sy * 2 + sx : sx * 2 + sy;
            vstore<T, V>(y0 + qi * a.C, vload<T, V>(XAT(n, 2 * ho + sy, 2 * wo + sx)));
        }
}

template <typename T, int V>
__global__ void copy_slice_kernel(const SliceArgs a) {
    SLICE_INDEX();
    vstore<T, V>(YAT(n, ho, wo), vload<T, V>(XAT(n, ho, wo)));
}

enum SliceKind { K_POOL2, K_POOL2S1, K_SPP, K_UPS, K_S2D, K_COPY };

template <typename T, int V>
void launch_kind(int kind, const SliceArgs& a, unsigned grid, hipStream_t s) {
    switch (kind) {
        case K_POOL2: hipLaunchKernelGGL((maxpool2_kernel<T, V>), dim3(grid), dim3(256), 0, s, a); break;
        case K_POOL2S1: hipLaunchKernelGGL((maxpool2s1_kernel<T, V>), dim3(grid), dim3(256), 0, s, a); break;
        case K_SPP: hipLaunchKernelGGL((spp_kernel<T, V>), dim3(grid), dim3(256), 0, s, a); break;
        case K_UPS: hipLaunchKernelGGL((upsample2x_kernel<T, V>), dim3(grid), dim3(256), 0, s, a); break;
        case K_S2D: hipLaunchKernelGGL((space2depth_kernel<T, V>), dim3(grid), dim3(256), 0, s, a); break;
        default: hipLaunchKernelGGL((copy_slice_kernel<T, V>), dim3(grid), dim3(256), 0, s, a); break;
    }
}

int run_slice(int kind, const ydet_slice_desc* d, int mode, hipStream_t stream) {
    YDET_CHECK_ARG(d && d->x && d->y, "slice op: null pointer");
    YDET_CHECK_ARG(dtype_ok(d->dtype), "slice op: bad dtype %d", d->dtype);
    YDET_CHECK_ARG(d->N > 0 && d->H > 0 && d->W > 0 && d->C > 0, "slice op: empty shape");
    int Ho = d->H, Wo = d->W, out_c = d->C;
    if (kind == K_POOL2 || kind == K_S2D) {
        YDET_CHECK_ARG(d->H % 2 == 0 && d->W % 2 == 0, "slice op: H, W must be even (%d,%d)", d->H, d->W);
        Ho = d->H / 2; Wo = d->W / 2;
    }
    if (kind == K_UPS) { Ho = 2 * d->H; Wo = 2 * d->W; YDET_CHECK_ARG(mode == 0 || mode == 1, "upsample: mode"); }
    if (kind == K_S2D) { out_c = 4 * d->C; YDET_CHECK_ARG(mode == 0 || mode == 1, "space_to_depth: mode"); }
    if (kind == K_SPP) out_c = 3 * d->C;
    YDET_CHECK_ARG(d->x_coff >= 0 && d->x_coff + d->C <= d->x_ld, "slice op: input slice exceeds x_ld");
    YDET_CHECK_ARG(d->y_coff >= 0 && d->y_coff + out_c <= d->y_ld, "slice op: output slice exceeds y_ld");

    const int esz = (int)dtype_size(d->dtype);
    const int vw = 16 / esz;
    const bool vec = d->C % vw == 0 && d->x_ld % vw == 0 && d->x_coff % vw == 0 && d->y_ld % vw == 0 &&
                     d->y_coff % vw == 0 && ((uintptr_t)d->x & 15) == 0 && ((uintptr_t)d->y & 15) == 0;
    SliceArgs a;
    a.x = (const char*)d->x; a.y = (char*)d->y;
    a.N = d->N; a.H = d->H; a.W = d->W; a.C = d->C;
    a.x_ld = d->x_ld; a.x_coff = d->x_coff; a.y_ld = d->y_ld; a.y_coff = d->y_coff;
    a.Ho = Ho; a.Wo = Wo; a.mode = mode;
    a.cv = vec ? d->C / vw : d->C;
    a.total = (long long)d->N * Ho * Wo * a.cv;
    const long long grid = (a.total + 255) / 256;
    YDET_CHECK_ARG(grid > 0 && grid < 0x7fffffffLL, "slice op: grid");
    if (kind == K_SPP && vec && d->H * d->W <= kSppMaxHW && a.cv % kSppGV == 0) {
        const unsigned g = (unsigned)(d->N * (a.cv / kSppGV));
        if (d->dtype == YDET_BF16)
            hipLaunchKernelGGL((spp_lds_kernel<__bf16, 8>), dim3(g), dim3(256), 0, stream, a);
        else
            hipLaunchKernelGGL((spp_lds_kernel<float, 4>), dim3(g), dim3(256), 0, stream, a);
        YDET_CHECK_LAUNCH("spp_lds_kernel");
        return YDET_OK;
    }
    if (d->dtype == YDET_BF16) {
        if (vec) launch_kind<__bf16, 8>(kind, a, (unsigned)grid, stream);
        else launch_kind<__bf16, 1>(kind, a, (unsigned)grid, stream);
    } else {
        if (vec) launch_kind<float, 4>(kind, a, (unsigned)grid, stream);
        else launch_kind<float, 1>(kind, a, (unsigned)grid, stream);
    }
    YDET_CHECK_LAUNCH("slice kernel");
    return YDET_OK;
}

// ---- API-edge layout conversion ------------------------------------------------------------------
struct LayoutArgs {
    const char* x;
    char* y;
    int N, C, H, W, ld, coff, c_pad, focus;
    long long total;
};

// NCHW -> NHWC (optionally through the Focus slicing).  One thread per output pixel: the source
// reads are coalesced along W per channel plane, the destination write is one short contiguous run.
template <typename TX, typename TY>
__global__ void nchw_to_nhwc_kernel(const LayoutArgs a) {
    const long long gid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= a.total) return;
    const int Ho = a.focus ? a.H / 2 : a.H, Wo = a.focus ? a.W / 2 : a.W;
    const int wo = (int)(gid % Wo);
    const int ho = (int)((gid / Wo) % Ho);
    const int n = (int)(gid / ((long long)Wo * Ho));
    const TX* x = reinterpret_cast<const TX*>(a.x);
    TY* y = reinterpret_cast<TY*>(a.y) + (size_t)gid * a.ld + a.coff;
    const size_t plane = (size_t)a.H * a.W;
    const int cout = a.focus ? 4 * a.C : a.C;
    if (a.focus) {
        for (int qi = 0; qi < 4; ++qi) {
            const int dy = qi & 1, dx = qi >> 1;  // (0,0),(1,0),(0,1),(1,1)
            for (int c = 0; c < a.C; ++c)
                y[qi * a.C + c] = from_f32<TY>(
                    to_f32(x[((size_t)n * a.C + c) * plane + (size_t)(2 * ho + dy) * a.W + (2 * wo + dx)]));
        }
    } else {
        for (int c = 0; c < a.C; ++c)
            y[c] = from_f32<TY>(to_f32(x[((size_t)n * a.C + c) * plane + (size_t)ho * a.W + wo]));
    }
    for (int c = cout; c < a.c_pad; ++c) y[c] = from_f32<TY>(0.f);
}

// NHWC slice -> NCHW.  Tiled through LDS (32 pixels x 32 channels) so both sides are coalesced.
template <typename TX, typename TY>
__global__ void nhwc_to_nchw_kernel(const LayoutArgs a) {
    __shared__ float tile[32][33];
    const long long hw = (long long)a.H * a.W;
    const long long p0 = (long long)blockIdx.x * 32;
    const int c0 = blockIdx.y * 32, n = blockIdx.z;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 256 threads: 32 x 8
    const TX* x = reinterpret_cast<const TX*>(a.x);
    TY* y = reinterpret_cast<TY*>(a.y);
    for (int r = ty; r < 32; r += 8) {
        const long long p = p0 + r;
        const int c = c0 + tx;
        if (p < hw && c < a.C) tile[r][tx] = to_f32(x[((size_t)n * hw + p) * a.ld + a.coff + c]);
    }
    __syncthreads();
    for (int r = ty; r < 32; r += 8) {
        const int c = c0 + r;
        const long long p = p0 + tx;
        if (p < hw && c < a.C) y[((size_t)n * a.C + c) * hw + p] = from_f32<TY>(tile[tx][r]);
    }
}

}  // namespace

extern "C" int ydet_maxpool2x2_nhwc(const ydet_slice_desc* d, void* stream) {
    clear_error();
    return run_slice(K_POOL2, d, 0, as_stream(stream));
}
extern "C" int ydet_maxpool2x2s1_zeropad_nhwc(const ydet_slice_desc* d, void* stream) {
    clear_error();
    return run_slice(K_POOL2S1, d, 0, as_stream(stream));
}
extern "C" int ydet_spp_nhwc(const ydet_slice_desc* d, void* stream) {
    clear_error();
    return run_slice(K_SPP, d, 0, as_stream(stream));
}
extern "C" int ydet_upsample2x_nhwc(const ydet_slice_desc* d, int mode, void* stream) {
    clear_error();
    return run_slice(K_UPS, d, mode, as_stream(stream));
}
extern "C" int ydet_space_to_depth_nhwc(const ydet_slice_desc* d, int mode, void* stream) {
    clear_error();
    return run_slice(K_S2D, d, mode, as_stream(stream));
}
extern "C" int ydet_copy_slice_nhwc(const ydet_slice_desc* d, void* stream) {
    clear_error();
    return run_slice(K_COPY, d, 0, as_stream(stream));
}

int ydet_launch_slice(int op_kind, const ydet_slice_desc* d, int mode, hipStream_t stream) {
    switch (op_kind) {
        case YDET_OP_MAXPOOL2: return run_slice(K_POOL2, d, 0, stream);
        case YDET_OP_MAXPOOL2S1: return run_slice(K_POOL2S1, d, 0, stream);
        case YDET_OP_SPP: return run_slice(K_SPP, d, 0, stream);
        case YDET_OP_UPSAMPLE: return run_slice(K_UPS, d, mode, stream);
        case YDET_OP_SPACE2DEPTH: return run_slice(K_S2D, d, mode, stream);
        case YDET_OP_COPY: return run_slice(K_COPY, d, 0, stream);
        default: return fail(YDET_ERR_ARG, "slice op: unknown kind %d", op_kind);
    }
}

int ydet_launch_nchw_to_nhwc(const void* x, void* y, int N, int C, int H, int W, int y_ld, int y_coff,
                             int c_pad, int x_dtype, int y_dtype, int focus, hipStream_t stream) {
    YDET_CHECK_ARG(x && y, "nchw_to_nhwc: null pointer");
    YDET_CHECK_ARG(dtype_ok(x_dtype) && dtype_ok(y_dtype), "nchw_to_nhwc: bad dtype");
    YDET_CHECK_ARG(N > 0 && C > 0 && H > 0 && W > 0, "nchw_to_nhwc: empty shape");
    YDET_CHECK_ARG(!focus || (H % 2 == 0 && W % 2 == 0), "nchw_to_nhwc: focus needs even H, W");
    const int cout = focus ? 4 * C : C;
    YDET_CHECK_ARG(c_pad >= cout && y_coff >= 0 && y_coff + c_pad <= y_ld, "nchw_to_nhwc: c_pad/y_ld");
    LayoutArgs a;
    a.x = (const char*)x; a.y = (char*)y; a.N = N; a.C = C; a.H = H; a.W = W;
    a.ld = y_ld; a.coff = y_coff; a.c_pad = c_pad; a.focus = focus;
    a.total = (long long)N * (focus ? (H / 2) * (W / 2) : H * W);
    const unsigned grid = (unsigned)((a.total + 255) / 256);
    if (x_dtype == YDET_F32 && y_dtype == YDET_F32)
        hipLaunchKernelGGL((nchw_to_nhwc_kernel<float, float>), dim3(grid), dim3(256), 0, stream, a);
    else if (x_dtype == YDET_F32)
        hipLaunchKernelGGL((nchw_to_nhwc_kernel<float, __bf16>), dim3(grid), dim3(256), 0, stream, a);
    else if (y_dtype == YDET_F32)
        hipLaunchKernelGGL((nchw_to_nhwc_kernel<__bf16, float>), dim3(grid), dim3(256), 0, stream, a);
    else
        hipLaunchKernelGGL((nchw_to_nhwc_kernel<__bf16, __bf16>), dim3(grid), dim3(256), 0, stream, a);
    YDET_CHECK_LAUNCH("nchw_to_nhwc_kernel");
    return YDET_OK;
}

int ydet_launch_nhwc_to_nchw(const void* x, void* y, int N, int C, int H, int W, int x_ld, int x_coff,
                             int x_dtype, int y_dtype, hipStream_t stream) {
    YDET_CHECK_ARG(x && y, "nhwc_to_nchw: null pointer");
    YDET_CHECK_ARG(dtype_ok(x_dtype) && dtype_ok(y_dtype), "nhwc_to_nchw: bad dtype");
    YDET_CHECK_ARG(N > 0 && C > 0 && H > 0 && W > 0 && N < 65536, "nhwc_to_nchw: shape");
    YDET_CHECK_ARG(x_coff >= 0 && x_coff + C <= x_ld, "nhwc_to_nchw: slice exceeds x_ld");
    LayoutArgs a;
    a.x = (const char*)x; a.y = (char*)y; a.N = N; a.C = C; a.H = H; a.W = W;
    a.ld = x_ld; a.coff = x_coff; a.c_pad = C; a.focus = 0; a.total = 0;
    const dim3 grid((unsigned)(((long long)H * W + 31) / 32), (unsigned)((C + 31) / 32), (unsigned)N);
    if (x_dtype == YDET_F32 && y_dtype == YDET_F32)
        hipLaunchKernelGGL((nhwc_to_nchw_kernel<float, float>), grid, dim3(256), 0, stream, a);
    else if (x_dtype == YDET_F32)
        hipLaunchKernelGGL((nhwc_to_nchw_kernel<float, __bf16>), grid, dim3(256), 0, stream, a);
    else if (y_dtype == YDET_F32)
        hipLaunchKernelGGL((nhwc_to_nchw_kernel<__bf16, float>), grid, dim3(256), 0, stream, a);
    else
        hipLaunchKernelGGL((nhwc_to_nchw_kernel<__bf16, __bf16>), grid, dim3(256), 0, stream, a);
    YDET_CHECK_LAUNCH("nhwc_to_nchw_kernel");
    return YDET_OK;
}

extern "C" int ydet_nchw_to_nhwc(const void* x, void* y, int N, int C, int H, int W, int y_ld, int y_coff,
                                 int c_pad, int x_dtype, int y_dtype, int focus, void* stream) {
    clear_error();
    return ydet_launch_nchw_to_nhwc(x, y, N, C, H, W, y_ld, y_coff, c_pad, x_dtype, y_dtype, focus,
                                    as_stream(stream));
}
extern "C" int ydet_nhwc_to_nchw(const void* x, void* y, int N, int C, int H, int W, int x_ld, int x_coff,
                                 int x_dtype, int y_dtype, void* stream) {
    clear_error();
    return ydet_launch_nhwc_to_nchw(x, y, N, C, H, W, x_ld, x_coff, x_dtype, y_dtype, as_stream(stream));
}
