// First layer of the YOLOv5 networks in one pass over the NCHW image (gfx950): Focus slicing (space to depth by 2,
// channel order (dy,dx) = (0,0),(1,0),(0,1),(1,1)) + 3x3 convolution + folded BatchNorm + activation, bf16 MFMA with
// fp32 accumulation, NHWC bf16 output.  The image is read once in its native layout and the 12-channel focus tensor
// never exists in memory.
//
// A workgroup produces an 8 x 32 patch of output pixels.  It gathers the 20 x 68 x 3 image patch underneath into an
// LDS tile laid out like the focus tensor, F[10][34][16] bf16 (12 channels + 4 zeros per pixel, 32 bytes), so the
// im2col operand of an MFMA K-step (2 filter taps x 16 channels) is one 16-byte LDS read per lane.  The packed
// weights (k = tap*16 + channel, 5 K-steps) stay in registers for the whole kernel.
#include <type_traits>

#include "ydet_internal.h"

using namespace ydet;

namespace {

constexpr int TH = 8, TW = 32;            // output pixels per workgroup
constexpr int FH = TH + 2, FW = TW + 2;   // focus pixels underneath (3x3, pad 1)
constexpr int KSTEPS = 5;                 // 9 taps x 16 channels = 144 -> 160 = 5 x 32

struct StemArgs {
    const char* x;      // [N][3][H][W]
    const char* w;      // packed [cout_pad][160] bf16
    const float* bias;
    char* y;            // NHWC bf16
    int N, H, W, Ho, Wo, Cout, y_ld, y_coff;
    int tiles_x, tiles_y;
    int act;
    float slope;
    int Hp, Wp;         // rgb_conv3x3_kernel<.., POOL>: size of the pooled map that is written
};

template <typename TX>
struct Pair;   // two horizontally adjacent pixels of the image
template <>
struct Pair<float> {
    typedef float2 type;
    static __device__ __forceinline__ void split(const float2& v, float& a, float& b) { a = v.x; b = v.y; }
};
template <>
struct Pair<__bf16> {
    typedef unsigned int type;
    static __device__ __forceinline__ void split(const unsigned int& v, float& a, float& b) {
        a = __uint_as_float(v << 16);
        b = __uint_as_float(v & 0xffff0000u);
    }
};

// uint8 images carry raw pixel values: a pair is two bytes, and a value becomes bf16(float(v) / 255) -- the
// `img.float() / 255` of yolov5_official/val.py:68-69 -- through a 256-entry table (IEEE division, done once per entry)
template <>
struct Pair<uint8_t> {
    typedef unsigned short type;
};

template <typename TX, int FN>
// the 32-channel form fits five workgroups per CU when held to 96 registers (90 used, no scratch); left to itself the
// compiler takes 118 and four
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(FN == 2 ? 5 : 2, FN == 2 ? 5 : 2))) void focus_conv3x3_kernel(const StemArgs a) {
    __shared__ __attribute__((aligned(16))) __bf16 F[FH * FW * 16];
    __shared__ __bf16 lut[std::is_same<TX, uint8_t>::value ? 256 : 1];
    if constexpr (std::is_same<TX, uint8_t>::value) lut[threadIdx.x] = (__bf16)((float)threadIdx.x / 255.0f);
    typedef typename Pair<TX>::type pair_t;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // consecutive patches of an image share halo pixels: keep them on one XCD (one L2)
    const unsigned bid = xcd_remap(blockIdx.x, gridDim.x);
    const int tx = bid % a.tiles_x, ty = bid / a.tiles_x, n = blockIdx.y;
    const int i0 = ty * TH, j0 = tx * TW;      // first output pixel of the patch
    const int y0 = 2 * (i0 - 1), x0 = 2 * (j0 - 1);   // first image pixel underneath

    // ---- weights -> registers: lane (row = lane&15, g = lane>>4) holds k = 32 s + 8 g .. + 7 of one output channel.
    // MFMA i of a pair (i = 2h, 2h+1) takes channel 32 h + 8 (row>>2) + 4 (i&1) + (row&3) on its row, so a lane ends
    // up with the 8 CONSECUTIVE channels 32 h + 8 g .. + 7 of its pixel and stores them as one 16-byte piece.
    bf16x8 wf[FN][KSTEPS];
    const int wrow = 8 * ((lane & 15) >> 2) + (lane & 3);
#pragma unroll
    for (int i = 0; i < FN; ++i)
#pragma unroll
        for (int s = 0; s < KSTEPS; ++s)
            wf[i][s] = *reinterpret_cast<const bf16x8*>(a.w + ((size_t)(32 * (i >> 1) + 4 * (i & 1) + wrow) * (KSTEPS * 32) + s * 32 + (lane >> 4) * 8) * 2);

    // ---- image patch -> LDS focus tile: a thread builds whole focus pixels (2 x 2 image pixels x 3 channels = six
    // 2-pixel loads) and writes each as two 16-byte pieces, zero channels included ------------------------------------
    constexpr int NFP = FH * FW;                        // 340 focus pixels
    constexpr int ITER = (NFP + 255) / 256;
    pair_t px[ITER][6];
    const TX* img = reinterpret_cast<const TX*>(a.x) + (size_t)n * 3 * a.H * a.W;
#pragma unroll
    for (int it = 0; it < ITER; ++it) {
        const int fp = tid + 256 * it;
        const int fy = fp / FW, fx = fp - fy * FW;
        const int Y = y0 + 2 * fy, X = x0 + 2 * fx;     // X and W are even: a pair is inside or outside as a whole
        const bool colok = fp < NFP && (unsigned)X < (unsigned)a.W;
#pragma unroll
        for (int dy = 0; dy < 2; ++dy) {
            const bool ok = colok && (unsigned)(Y + dy) < (unsigned)a.H;
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                px[it][dy * 3 + c] = pair_t();
                if (ok) px[it][dy * 3 + c] = *reinterpret_cast<const pair_t*>(img + ((size_t)c * a.H + Y + dy) * a.W + X);
            }
        }
    }
    if constexpr (std::is_same<TX, uint8_t>::value) __syncthreads();   // the table is complete
#pragma unroll
    for (int it = 0; it < ITER; ++it) {
        const int fp = tid + 256 * it;
        if (fp >= NFP) continue;
        bf16x8 lo, hi;                                   // focus channels 0..7 and 8..15 (12..15 zero)
        __bf16 ch[16];
#pragma unroll
        for (int dy = 0; dy < 2; ++dy)
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                if constexpr (std::is_same<TX, uint8_t>::value) {
                    ch[dy * 3 + c] = lut[px[it][dy * 3 + c] & 0xff];
                    ch[(2 + dy) * 3 + c] = lut[px[it][dy * 3 + c] >> 8];
                } else {
                    float v0, v1;
                    Pair<TX>::split(px[it][dy * 3 + c], v0, v1);
                    ch[dy * 3 + c] = (__bf16)v0;            // (dy, dx = 0)
                    ch[(2 + dy) * 3 + c] = (__bf16)v1;      // (dy, dx = 1)
                }
            }
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            lo[e] = ch[e];
            hi[e] = e < 4 ? ch[8 + e] : (__bf16)0.f;
        }
        *reinterpret_cast<bf16x8*>(F + fp * 16) = lo;
        *reinterpret_cast<bf16x8*>(F + fp * 16 + 8) = hi;
    }
    __syncthreads();

    // ---- MFMA: wave w owns output rows 2w, 2w+1; pixel group j = row (j>>1), columns 16 (j&1) .. + 15 ----------
    f32x4 acc[FN][4];
#pragma unroll
    for (int i = 0; i < FN; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int p = lane & 15, g = lane >> 4;
#pragma unroll
    for (int s = 0; s < KSTEPS; ++s) {
        const int tap = 2 * s + (g >> 1);           // K-step s covers taps 2s and 2s+1; tap 9 is padding
        const int kh = tap / 3, kw = tap - kh * 3;
        bf16x8 xf[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int r = 2 * wave + (j >> 1), c = 16 * (j & 1) + p;
            const bf16x8 v = *reinterpret_cast<const bf16x8*>(F + (((r + (tap < 9 ? kh : 0)) * FW + c + (tap < 9 ? kw : 0)) * 16 + (g & 1) * 8));
            xf[j] = tap < 9 ? v : bf16x8{};
        }
#pragma unroll
        for (int i = 0; i < FN; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[i][s], xf[j], acc[i][j], 0, 0, 0);
    }

    // ---- epilogue: lane holds channels 32 h + 8 g .. + 7 (MFMAs 2h and 2h+1) of pixel p of group j --------------------
    auto finish = [&](auto silu) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int oi = i0 + 2 * wave + (j >> 1), oj = j0 + 16 * (j & 1) + p;
            if (oi >= a.Ho || oj >= a.Wo) continue;
            char* yrow = a.y + ((((size_t)n * a.Ho + oi) * a.Wo + oj) * a.y_ld + a.y_coff) * 2;
#pragma unroll
            for (int h = 0; h < FN / 2; ++h) {
                const int co = 32 * h + 8 * g;
                if (co >= a.Cout) continue;
                float v[8];
#pragma unroll
                for (int e = 0; e < 4; ++e) { v[e] = acc[2 * h][j][e]; v[4 + e] = acc[2 * h + 1][j][e]; }
                if (a.bias) {
                    const float4 b0 = *reinterpret_cast<const float4*>(a.bias + co);
                    const float4 b1 = *reinterpret_cast<const float4*>(a.bias + co + 4);
                    v[0] += b0.x; v[1] += b0.y; v[2] += b0.z; v[3] += b0.w;
                    v[4] += b1.x; v[5] += b1.y; v[6] += b1.z; v[7] += b1.w;
                }
                bf16x8 o;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    float x = v[e];
                    if (decltype(silu)::value)
                        x = x * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(x * -1.44269504088896340736f));
                    else
                        x = x > 0.f ? x : x * a.slope;
                    o[e] = (__bf16)x;
                }
                if (co + 8 <= a.Cout) {
                    *reinterpret_cast<bf16x8*>(yrow + co * 2) = o;
                } else {   // Cout is a multiple of 4: the last piece of e.g. 44 channels is half a vector
                    bf16x4 o4 = {o[0], o[1], o[2], o[3]};
                    *reinterpret_cast<bf16x4*>(yrow + co * 2) = o4;
                }
            }
        }
    };
    if (a.act == YDET_ACT_SILU) finish(std::true_type{});
    else finish(std::false_type{});
}

// ---- plain first layer: 3x3 / stride 1 / pad 1 over the 3 image channels ---------------------------------------------
// (Darknet-19 / -53 / CSPDarknet / the tiny trunks: models/basemodel/d19.py:22,88,133,175,221.)  Same scheme as the Focus
// kernel with image pixels in place of focus pixels: the 10 x 34 x 3 patch under an 8 x 32 output patch sits in LDS as
// P[340][4] bf16 (3 channels + a zero, 8 bytes), K is tap-major with 4 channels per tap (k = tap*4 + c: 36 -> 64 = two
// K steps), so a lane's 8 k-values are two taps = two 8-byte LDS reads.  As the generic convolution this layer ran on
// channel-padded NHWC input behind a layout kernel and was the slowest launch of every one of those networks.
// POOL: the MaxPool2d(2, 2) that follows the layer in the Darknet-19 and tiny trunks (models/basemodel/d19.py:29, :175-182)
// is taken in the epilogue -- a lane holds both rows of its column, the neighbouring column is the neighbouring lane --
// and only the pooled map (a quarter of the bytes) is written.  max commutes with the rounding to bf16, so the result
// is that of the two launches.
template <typename TX, int FN, bool POOL = false>
__global__ __launch_bounds__(256) void rgb_conv3x3_kernel(const StemArgs a) {
    __shared__ __attribute__((aligned(16))) __bf16 P[FH * FW * 4];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const unsigned bid = xcd_remap(blockIdx.x, gridDim.x);
    const int tx = bid % a.tiles_x, ty = bid / a.tiles_x, n = blockIdx.y;
    const int i0 = ty * TH, j0 = tx * TW;
    const int y0 = i0 - 1, x0 = j0 - 1;

    bf16x8 wf[FN][2];
    const int wrow = 8 * ((lane & 15) >> 2) + (lane & 3);
#pragma unroll
    for (int i = 0; i < FN; ++i)
#pragma unroll
        for (int s = 0; s < 2; ++s)
            wf[i][s] = *reinterpret_cast<const bf16x8*>(a.w + ((size_t)(32 * (i >> 1) + 4 * (i & 1) + wrow) * 64 + s * 32 + (lane >> 4) * 8) * 2);

    constexpr int NP = FH * FW;
    const TX* img = reinterpret_cast<const TX*>(a.x) + (size_t)n * 3 * a.H * a.W;
    for (int p = tid; p < NP; p += 256) {
        const int py = p / FW, pxx = p - py * FW;
        const int Y = y0 + py, X = x0 + pxx;
        const bool ok = (unsigned)Y < (unsigned)a.H && (unsigned)X < (unsigned)a.W;
        bf16x4 v = {(__bf16)0.f, (__bf16)0.f, (__bf16)0.f, (__bf16)0.f};
        if (ok) {
#pragma unroll
            for (int c = 0; c < 3; ++c) v[c] = (__bf16)to_f32(img[((size_t)c * a.H + Y) * a.W + X]);
        }
        *reinterpret_cast<bf16x4*>(P + p * 4) = v;
    }
    __syncthreads();

    f32x4 acc[FN][4];
#pragma unroll
    for (int i = 0; i < FN; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int p = lane & 15, g = lane >> 4;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        const int t0 = 8 * s + 2 * g, t1 = t0 + 1;            // the lane's two taps of this K step (9.. are padding)
        bf16x8 xf[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int r = 2 * wave + (j >> 1), c = 16 * (j & 1) + p;
            bf16x4 lo = {(__bf16)0.f, (__bf16)0.f, (__bf16)0.f, (__bf16)0.f}, hi = lo;
            if (t0 < 9) lo = *reinterpret_cast<const bf16x4*>(P + ((r + t0 / 3) * FW + c + t0 % 3) * 4);
            if (t1 < 9) hi = *reinterpret_cast<const bf16x4*>(P + ((r + t1 / 3) * FW + c + t1 % 3) * 4);
            xf[j] = bf16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        }
#pragma unroll
        for (int i = 0; i < FN; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[i][s], xf[j], acc[i][j], 0, 0, 0);
    }

    const bool silu = a.act == YDET_ACT_SILU;
    if constexpr (POOL) {
        const int pi = (i0 >> 1) + wave;                       // rows 2 wave and 2 wave + 1 of the tile -> one pooled row
#pragma unroll
        for (int jj = 0; jj < 2; ++jj) {
            const int pj = (j0 >> 1) + 8 * jj + (p >> 1);      // columns p and p ^ 1 -> one pooled column (even lanes store)
#pragma unroll
            for (int h = 0; h < FN / 2; ++h) {
                const int co = 32 * h + 8 * g;
                float m[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    float top = e < 4 ? acc[2 * h][jj][e] : acc[2 * h + 1][jj][e - 4];
                    float bot = e < 4 ? acc[2 * h][jj + 2][e] : acc[2 * h + 1][jj + 2][e - 4];
                    if (a.bias && co < a.Cout) { const float b = a.bias[co + e]; top += b; bot += b; }
                    if (silu) {
                        top = top * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(top * -1.44269504088896340736f));
                        bot = bot * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(bot * -1.44269504088896340736f));
                    } else {
                        top = top > 0.f ? top : top * a.slope;
                        bot = bot > 0.f ? bot : bot * a.slope;
                    }
                    const float v = fmaxf(top, bot);
                    m[e] = fmaxf(v, __shfl_xor(v, 1));
                }
                if ((p & 1) || pi >= a.Hp || pj >= a.Wp || co >= a.Cout) continue;
                bf16x8 o;
#pragma unroll
                for (int e = 0; e < 8; ++e) o[e] = (__bf16)m[e];
                *reinterpret_cast<bf16x8*>(a.y + ((((size_t)n * a.Hp + pi) * a.Wp + pj) * a.y_ld + a.y_coff + co) * 2) = o;
            }
        }
        return;
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int oi = i0 + 2 * wave + (j >> 1), oj = j0 + 16 * (j & 1) + p;
        if (oi >= a.Ho || oj >= a.Wo) continue;
        char* yrow = a.y + ((((size_t)n * a.Ho + oi) * a.Wo + oj) * a.y_ld + a.y_coff) * 2;
#pragma unroll
        for (int h = 0; h < FN / 2; ++h) {
            const int co = 32 * h + 8 * g;
            if (co >= a.Cout) continue;
            float v[8];
#pragma unroll
            for (int e = 0; e < 4; ++e) { v[e] = acc[2 * h][j][e]; v[4 + e] = acc[2 * h + 1][j][e]; }
            if (a.bias) {
                const float4 b0 = *reinterpret_cast<const float4*>(a.bias + co);
                const float4 b1 = *reinterpret_cast<const float4*>(a.bias + co + 4);
                v[0] += b0.x; v[1] += b0.y; v[2] += b0.z; v[3] += b0.w;
                v[4] += b1.x; v[5] += b1.y; v[6] += b1.z; v[7] += b1.w;
            }
            bf16x8 o;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                float x = v[e];
                if (silu) x = x * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(x * -1.44269504088896340736f));
                else x = x > 0.f ? x : x * a.slope;
                o[e] = (__bf16)x;
            }
            *reinterpret_cast<bf16x8*>(yrow + co * 2) = o;
        }
    }
}

}  // namespace

int ydet_launch_rgb_conv(const ydet_conv_desc* d, int x_dtype, hipStream_t stream) {
    YDET_CHECK_ARG(d != nullptr && d->x && d->w && d->y, "rgb_conv3x3: null pointer");
    YDET_CHECK_ARG(d->groups <= 1 && d->dilation <= 1, "rgb_conv3x3: a plain convolution (one group, no dilation)");
    YDET_CHECK_ARG(dtype_ok(x_dtype), "rgb_conv3x3: bad image dtype");
    YDET_CHECK_ARG(d->in_dtype == YDET_BF16 && d->out_dtype == YDET_BF16, "rgb_conv3x3: bf16 weights and output only");
    YDET_CHECK_ARG(d->N > 0 && d->N <= 65535 && d->H > 0 && d->W > 0, "rgb_conv3x3: image %d x %d x %d", d->N, d->H, d->W);
    const bool pool = d->H % 2 == 0 && d->W % 2 == 0 && d->Ho == d->H / 2 && d->Wo == d->W / 2;   // the 2x2 / stride 2 max pool folded in
    YDET_CHECK_ARG(d->Cin == 3 && d->KH == 3 && d->KW == 3 && d->stride == 1 && d->pad == 1 && (pool || (d->Ho == d->H && d->Wo == d->W)),
                   "rgb_conv3x3: 3 image channels and a 3x3 / stride 1 / pad 1 convolution only (Ho x Wo = H x W, or H/2 x W/2 for the pooled form)");
    YDET_CHECK_ARG(d->Cout >= 8 && d->Cout <= 64 && d->Cout % 8 == 0 && d->cout_pad >= 64, "rgb_conv3x3: Cout=%d (multiple of 8, <= 64)", d->Cout);
    YDET_CHECK_ARG(d->kp == 64, "rgb_conv3x3: weights must be packed for 4 input channels per tap (kp = 64)");
    YDET_CHECK_ARG(d->y_ld % 8 == 0 && d->y_coff % 8 == 0 && d->y_coff >= 0 && d->y_coff + d->Cout <= d->y_ld,
                   "rgb_conv3x3: output slice (y_ld, y_coff multiples of 8)");
    YDET_CHECK_ARG(d->res == nullptr && d->post_scale == nullptr && d->y2 == nullptr, "rgb_conv3x3: no residual / second stage / destination");
    YDET_CHECK_ARG(((uintptr_t)d->x & 3) == 0 && ((uintptr_t)d->w & 15) == 0 && ((uintptr_t)d->y & 15) == 0 &&
                       ((uintptr_t)d->bias & 15) == 0, "rgb_conv3x3: pointer alignment");
    StemArgs a;
    a.x = (const char*)d->x; a.w = (const char*)d->w; a.bias = d->bias; a.y = (char*)d->y;
    a.N = d->N; a.H = d->H; a.W = d->W; a.Ho = d->H; a.Wo = d->W; a.Cout = d->Cout; a.y_ld = d->y_ld; a.y_coff = d->y_coff;
    a.Hp = d->Ho; a.Wp = d->Wo;
    a.tiles_x = (a.Wo + TW - 1) / TW; a.tiles_y = (a.Ho + TH - 1) / TH;
    a.act = d->act;
    a.slope = d->act == YDET_ACT_NONE ? 1.0f : d->slope;
    const dim3 grid((unsigned)(a.tiles_x * a.tiles_y), (unsigned)a.N);
    const bool wide = d->Cout > 32;
    if (pool) {
        if (x_dtype == YDET_F32) {
            if (wide) hipLaunchKernelGGL((rgb_conv3x3_kernel<float, 4, true>), grid, dim3(256), 0, stream, a);
            else hipLaunchKernelGGL((rgb_conv3x3_kernel<float, 2, true>), grid, dim3(256), 0, stream, a);
        } else {
            if (wide) hipLaunchKernelGGL((rgb_conv3x3_kernel<__bf16, 4, true>), grid, dim3(256), 0, stream, a);
            else hipLaunchKernelGGL((rgb_conv3x3_kernel<__bf16, 2, true>), grid, dim3(256), 0, stream, a);
        }
    } else if (x_dtype == YDET_F32) {
        if (wide) hipLaunchKernelGGL((rgb_conv3x3_kernel<float, 4>), grid, dim3(256), 0, stream, a);
        else hipLaunchKernelGGL((rgb_conv3x3_kernel<float, 2>), grid, dim3(256), 0, stream, a);
    } else {
        if (wide) hipLaunchKernelGGL((rgb_conv3x3_kernel<__bf16, 4>), grid, dim3(256), 0, stream, a);
        else hipLaunchKernelGGL((rgb_conv3x3_kernel<__bf16, 2>), grid, dim3(256), 0, stream, a);
    }
    YDET_CHECK_LAUNCH("rgb_conv3x3_kernel");
    return YDET_OK;
}

extern "C" int ydet_rgb_conv3x3_nchw(const ydet_conv_desc* d, int x_dtype, void* stream) {
    clear_error();
    return ydet_launch_rgb_conv(d, x_dtype, as_stream(stream));
}

namespace {
}  // namespace

int ydet_launch_focus_conv(const ydet_conv_desc* d, int x_dtype, hipStream_t stream) {
    YDET_CHECK_ARG(d != nullptr && d->x && d->w && d->y, "focus_conv: null pointer");
    YDET_CHECK_ARG(d->groups <= 1 && d->dilation <= 1, "focus_conv: a plain convolution (one group, no dilation)");
    YDET_CHECK_ARG(dtype_ok(x_dtype) || x_dtype == YDET_U8, "focus_conv: bad image dtype");
    YDET_CHECK_ARG(d->in_dtype == YDET_BF16 && d->out_dtype == YDET_BF16, "focus_conv: bf16 weights and output only");
    YDET_CHECK_ARG(d->N > 0 && d->N <= 65535 && d->H > 0 && d->W > 0 && d->H % 2 == 0 && d->W % 2 == 0,
                   "focus_conv: image %d x %d x %d", d->N, d->H, d->W);
    YDET_CHECK_ARG(d->Cin == 3 && d->KH == 3 && d->KW == 3 && d->stride == 1 && d->pad == 1,
                   "focus_conv: 3 image channels and a 3x3 / stride 1 / pad 1 convolution only");
    YDET_CHECK_ARG(d->Ho == d->H / 2 && d->Wo == d->W / 2, "focus_conv: output must be H/2 x W/2");
    YDET_CHECK_ARG(d->Cout >= 1 && d->Cout <= 64 && d->Cout % 4 == 0 && d->cout_pad >= 64, "focus_conv: Cout=%d (multiple of 4, <= 64)", d->Cout);
    YDET_CHECK_ARG(d->kp == KSTEPS * 32, "focus_conv: weights must be packed for 16 input channels (kp = %d)", KSTEPS * 32);
    YDET_CHECK_ARG(d->y_ld % 8 == 0 && d->y_coff % 8 == 0 && d->y_coff >= 0 && d->y_coff + d->Cout <= d->y_ld,
                   "focus_conv: output slice (y_ld, y_coff multiples of 8)");
    YDET_CHECK_ARG(d->res == nullptr && d->post_scale == nullptr, "focus_conv: no residual / second stage");
    YDET_CHECK_ARG(((uintptr_t)d->x & (x_dtype == YDET_U8 ? 1 : 7)) == 0 && ((uintptr_t)d->w & 15) == 0 && ((uintptr_t)d->y & 15) == 0 &&
                       ((uintptr_t)d->bias & 15) == 0, "focus_conv: pointer alignment");
    StemArgs a;
    a.x = (const char*)d->x; a.w = (const char*)d->w; a.bias = d->bias; a.y = (char*)d->y;
    a.N = d->N; a.H = d->H; a.W = d->W; a.Ho = d->Ho; a.Wo = d->Wo; a.Cout = d->Cout; a.y_ld = d->y_ld; a.y_coff = d->y_coff;
    a.tiles_x = (a.Wo + TW - 1) / TW; a.tiles_y = (a.Ho + TH - 1) / TH;
    a.act = d->act;
    a.slope = d->act == YDET_ACT_NONE ? 1.0f : d->slope;
    const dim3 grid((unsigned)(a.tiles_x * a.tiles_y), (unsigned)a.N);
    const bool wide = d->Cout > 32;
    if (x_dtype == YDET_F32) {
        if (wide) hipLaunchKernelGGL((focus_conv3x3_kernel<float, 4>), grid, dim3(256), 0, stream, a);
        else hipLaunchKernelGGL((focus_conv3x3_kernel<float, 2>), grid, dim3(256), 0, stream, a);
    } else if (x_dtype == YDET_U8) {
        if (wide) hipLaunchKernelGGL((focus_conv3x3_kernel<uint8_t, 4>), grid, dim3(256), 0, stream, a);
        else hipLaunchKernelGGL((focus_conv3x3_kernel<uint8_t, 2>), grid, dim3(256), 0, stream, a);
    } else {
        if (wide) hipLaunchKernelGGL((focus_conv3x3_kernel<__bf16, 4>), grid, dim3(256), 0, stream, a);
        else hipLaunchKernelGGL((focus_conv3x3_kernel<__bf16, 2>), grid, dim3(256), 0, stream, a);
    }
    YDET_CHECK_LAUNCH("focus_conv3x3_kernel");
    return YDET_OK;
}

extern "C" int ydet_focus_conv3x3_nchw(const ydet_conv_desc* d, int x_dtype, void* stream) {
    clear_error();
    return ydet_launch_focus_conv(d, x_dtype, as_stream(stream));
}
