// The first TWO layers of the YOLOv5 networks (plus the 1x1 pair that opens the first C3) in one pass over the NCHW
// image (gfx950): Focus slicing + Conv(12, 32, 3, 1) + Conv(32, 64, 3, 2) [+ 1x1 follower on the 64 channels], each with
// its folded BatchNorm and activation (yolov5s.yaml backbone rows 0-2; yolov5_official/modules.py Focus, Conv, C3).
// Run one after the other these layers are bound by HBM: the 32-channel map at half the image size is the largest
// tensor of the whole network (written once, read once).  Here it exists only as a 17 x 33 pixel patch in LDS.
//
// A workgroup (4 waves) produces an 8 x 16 patch of the second convolution's output pixels:
//   P0  the 38 x 70 x 3 image patch underneath -> LDS focus tile F[19][35] (12 channels + 4 zeros, bf16, 32-byte pitch)
//   P1  first convolution on the 17 x 33 patch A of ITS output pixels the second one needs (36 MFMA pixel groups of 16,
//       K = 9 taps x 16 channels), bias + activation, bf16; pixels outside the map are the second convolution's zero padding
//       -> LDS A (over F), even and odd columns apart so that a stride-2 tap reads 16 consecutive pixels (64-byte slots)
//   P2  second convolution (K = 9 taps x 32 channels) from A; waves split 2 (channels) x 2 (rows)
//   P3  optional 1x1 follower: the finished 128 x 64 tile goes through LDS (128-byte slots) once more
// LDS banking: a ds_read_b128 is served in the lane groups {0-3, 12-15, 20-27}, {4-11, 16-19, 28-31} (+32), i.e. half of
// a group reads 16-byte chunk c of eight pixels and the other half chunk c + 1 of the eight pixels in between (pixel =
// lane & 15, chunk = lane >> 4).  An odd pitch in chunks -- 3, 5, 9: what sixteen CONSECUTIVE lanes would want -- makes
// every such read a two-way conflict (the counters showed half of this kernel's LDS cycles as conflicts).  Conflict-free
// for any start pixel: a pitch of 2 chunks as it is (F); unpadded 4- and 8-chunk slots with the chunk index XORed with
// (slot >> 1) & 3 (A) or slot & 7 (T).  Row strides of A are multiples of 8 slots, so the XOR term of a tap depends only
// on the lane and on whether the tap is the third of its row: two per-lane bases, every other tap offset an immediate.
//
// RGB variant (the Darknet-style trunks of the v3 / v4 detectors, models/basemodel/d19.py:83-123 and :215-265, whose first
// two layers are Conv(3, 32, 3, p=1) and Conv(32, 64, 3, p=1, s=2)): no slicing, F holds the 19 x 35 image pixels
// themselves in 8-byte slots (3 channels + a zero), K is tap-major with 4 channels per tap (k = tap * 4 + channel:
// 36 -> 64, the packing of the stand-alone first-layer kernel of stem.hip), so P1 is TWO K steps per pixel group -- a
// lane's eight k-values are two taps, two 8-byte LDS reads -- where 16-channel slots would make it five; everything
// behind P1 is shared.  The 32-channel map at FULL image size -- 1.5 GB for 64 images of 608 x 608 -- is never written.
#include <type_traits>

#include "ydet_internal.h"

using namespace ydet;

namespace {

constexpr int T_H = 8, T_W = 16;                  // output pixels of the second convolution per workgroup
constexpr int A_H = 2 * T_H + 1, A_W = 2 * T_W + 1;   // first-convolution pixels underneath (3x3, stride 2, pad 1)
constexpr int A_N = A_H * A_W;                    // 561
constexpr int A_FRAGS = (A_N + 15) / 16;          // 36 pixel groups
constexpr int A_HALF = 24;                        // 17 even columns, then (from slot 24) 16 odd ones per row
constexpr int A_ROW = 40;                         // pixel slots per row; both multiples of 8 (see the XOR term above)
constexpr int A_PITCH = 64;                       // bytes per pixel slot: 32 channels, chunk index XOR (slot >> 1) & 3
constexpr int F_H = A_H + 2, F_W = A_W + 2;       // focus pixels underneath (3x3, stride 1, pad 1)
constexpr int F_N = F_H * F_W;                    // 665
constexpr int F_PITCH = 32;                       // 16 channels, no padding
constexpr int T_PITCH = 128;                      // 64 channels, chunk index XOR slot & 7
constexpr int KSTEPS0 = 5;                        // first convolution: 9 taps x 16 channels = 144 -> 160
constexpr int LDS_BYTES = A_H * A_ROW * A_PITCH;  // 43520 (three workgroups per CU) >= F (21280), >= T (16384)
static_assert(A_HALF % 8 == 0 && A_ROW % 8 == 0 && A_HALF >= (A_W + 1) / 2 && A_ROW >= A_HALF + A_W / 2, "A row layout");
static_assert(LDS_BYTES >= F_N * F_PITCH && LDS_BYTES >= T_H * T_W * T_PITCH, "the three tiles share one LDS region");
static_assert(A_FRAGS % 4 == 0, "pixel groups split evenly over four waves");

struct PairArgs {
    const char* x;        // [N][3][H][W]
    const char* w0;       // first convolution, packed [>=32][160] bf16 (k = tap*16 + focus channel)
    const float* b0;
    const char* w1;       // second convolution, packed [>=64][288] bf16 (k = tap*32 + channel)
    const float* b1;
    const char* pw;       // follower, packed [>=64][64] bf16, or null
    const float* pwb;
    char* y;
    char* y2;
    int N, H, W, H1, W1, Ho, Wo;
    int cout;             // channels written per pixel (64, or the follower's)
    int y_ld, y_coff, y2_ld, y2_coff, y_split;
    int tiles_x, tiles_y;
    int act0, act1;
    float slope0, slope1;
    int pw_rows;          // rows of pw that exist
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
template <>
struct Pair<uint8_t> {   // raw pixel values: bf16(float(v) / 255) through a 256-entry table (yolov5_official/val.py:68-69)
    typedef unsigned short type;
};

template <bool SILU>
__device__ __forceinline__ float act_fast(float x, float slope) {
    if (SILU) return x * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(x * -1.44269504088896340736f));
    return x > 0.f ? x : x * slope;
}

// eight finished values (accumulators of MFMAs 2h and 2h+1, bias added, activated) as one 16-byte piece.  (Starting
// the accumulators from the bias instead costs a register move each; zero accumulators are an inline constant.)
template <bool SILU>
__device__ __forceinline__ bf16x8 finish8(const f32x4& lo, const f32x4& hi, const float4& b0, const float4& b1, float slope) {
    const float v[8] = {lo[0] + b0.x, lo[1] + b0.y, lo[2] + b0.z, lo[3] + b0.w, hi[0] + b1.x, hi[1] + b1.y, hi[2] + b1.z, hi[3] + b1.w};
    bf16x8 o;
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = (__bf16)act_fast<SILU>(v[e], slope);
    return o;
}

template <typename TX, bool FOLLOW, bool SILU0, bool SILU1, bool RGB = false>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3, 3))) void focus_pair_kernel(const PairArgs a) {
    static_assert(!RGB || !std::is_same<TX, uint8_t>::value, "raw pixel values: Focus variant only");
    __shared__ __attribute__((aligned(16))) char lds[LDS_BYTES];
    __shared__ __bf16 lut[std::is_same<TX, uint8_t>::value ? 256 : 1];
    if constexpr (std::is_same<TX, uint8_t>::value) lut[threadIdx.x] = (__bf16)((float)threadIdx.x / 255.0f);
    typedef typename Pair<TX>::type pair_t;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int p16 = lane & 15, g = lane >> 4;
    const unsigned bid = xcd_remap(blockIdx.x, gridDim.x);   // neighbouring patches of an image share halo: one XCD
    const int tx = bid % a.tiles_x, ty = bid / a.tiles_x, n = blockIdx.y;
    const int oi0 = ty * T_H, oj0 = tx * T_W;                // first output pixel
    const int ar0 = 2 * oi0 - 1, ac0 = 2 * oj0 - 1;          // first pixel of A (in the first convolution's output map)
    const int Y0 = RGB ? ar0 - 1 : 2 * (ar0 - 1), X0 = RGB ? ac0 - 1 : 2 * (ac0 - 1);        // first image pixel

    // ---- P0: image patch -> F.  A thread builds whole focus pixels (2 x 2 image pixels x 3 channels = six 2-pixel loads)
    constexpr int ITER = (F_N + 255) / 256;
    pair_t px[RGB ? 1 : ITER][6];
    TX px1[RGB ? ITER : 1][3];          // RGB variant: a thread fetches single pixels, three channels each
    const TX* img = reinterpret_cast<const TX*>(a.x) + (size_t)n * 3 * a.H * a.W;
    // (offsets inside one image fit 32 bits -- checked on the host -- and the image base is uniform)
    const bool interior0 = RGB ? Y0 >= 0 && Y0 + F_H <= a.H && X0 >= 0 && X0 + F_W <= a.W
                               : Y0 >= 0 && Y0 + 2 * F_H <= a.H && X0 >= 0 && X0 + 2 * F_W <= a.W;   // workgroup-uniform
    if constexpr (RGB) {
#pragma unroll
        for (int it = 0; it < ITER; ++it) {
            const int fp = tid + 256 * it;
            const int fq = min(fp, F_N - 1);
            const int fy = fq / F_W, fx = fq - fy * F_W;
            const int Y = Y0 + fy, X = X0 + fx;
            const bool ok = interior0 || (fp < F_N && (unsigned)Y < (unsigned)a.H && (unsigned)X < (unsigned)a.W);
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                px1[it][c] = TX();
                if (ok) px1[it][c] = img[(unsigned)(c * a.H + Y) * (unsigned)a.W + (unsigned)X];
            }
        }
    } else if (interior0) {
#pragma unroll
        for (int it = 0; it < ITER; ++it) {
            const int fp = min(tid + 256 * it, F_N - 1);
            const int fy = fp / F_W, fx = fp - fy * F_W;
            const unsigned o0 = (unsigned)((Y0 + 2 * fy) * a.W + X0 + 2 * fx), plane = (unsigned)(a.H * a.W);
#pragma unroll
            for (int dy = 0; dy < 2; ++dy)
#pragma unroll
                for (int c = 0; c < 3; ++c)
                    px[it][dy * 3 + c] = *reinterpret_cast<const pair_t*>(img + (o0 + (unsigned)c * plane + (unsigned)dy * (unsigned)a.W));
        }
    } else {
#pragma unroll
        for (int it = 0; it < ITER; ++it) {
            const int fp = tid + 256 * it;
            const int fy = fp / F_W, fx = fp - fy * F_W;
            const int Y = Y0 + 2 * fy, X = X0 + 2 * fx;          // X and W are even: a pair is inside or outside as a whole
            const bool colok = fp < F_N && (unsigned)X < (unsigned)a.W;
#pragma unroll
            for (int dy = 0; dy < 2; ++dy) {
                const bool ok = colok && (unsigned)(Y + dy) < (unsigned)a.H;
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    px[it][dy * 3 + c] = pair_t();
                    if (ok) px[it][dy * 3 + c] = *reinterpret_cast<const pair_t*>(img + ((unsigned)(c * a.H + Y + dy) * (unsigned)a.W + (unsigned)X));
                }
            }
        }
    }
    // weights of the first convolution: lane (row = lane&15, g) holds k = 32 s + 8 g .. + 7 of one output channel; MFMA i
    // takes channel 8 (row>>2) + 4 i + (row&3) on its row, so a lane ends up with the 8 consecutive channels 8 g .. + 7
    constexpr int KS0 = RGB ? 2 : KSTEPS0;          // K steps of the first convolution
    bf16x8 wf[2][KS0];
    const int wrow = 8 * (p16 >> 2) + (p16 & 3);
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int s = 0; s < KS0; ++s)
            wf[i][s] = *reinterpret_cast<const bf16x8*>(a.w0 + ((size_t)(4 * i + wrow) * (KS0 * 32) + s * 32 + g * 8) * 2);
    // K beyond the ninth tap is padding: zero weights there, whatever the packed rows hold, so the pixel operand of those
    // lanes may be any finite values (Focus: the second half of the last K step; RGB: lane group g of K step s holds
    // taps 8 s + 2 g and 8 s + 2 g + 1, i.e. everything of step 1 but the first tap of group 0)
    if constexpr (RGB) {
        if (g >= 1) wf[0][1] = wf[1][1] = bf16x8{};
        else
#pragma unroll
            for (int e = 4; e < 8; ++e) wf[0][1][e] = wf[1][1][e] = (__bf16)0.f;
    } else {
        if (g >= 2) wf[0][KSTEPS0 - 1] = wf[1][KSTEPS0 - 1] = bf16x8{};
    }
    if constexpr (std::is_same<TX, uint8_t>::value) __syncthreads();   // the table is complete
#pragma unroll
    for (int it = 0; it < ITER; ++it) {
        const int fp = tid + 256 * it;
        if (fp >= F_N) continue;
        if constexpr (RGB) {
            bf16x4 lo = bf16x4{};
#pragma unroll
            for (int c = 0; c < 3; ++c) lo[c] = (__bf16)(float)px1[it][c];
            *reinterpret_cast<bf16x4*>(lds + fp * 8) = lo;
            continue;
        }
        __bf16 ch[12];
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
                    ch[dy * 3 + c] = (__bf16)v0;            // focus channel (dy, dx = 0)
                    ch[(2 + dy) * 3 + c] = (__bf16)v1;      // (dy, dx = 1)
                }
            }
        bf16x8 lo, hi;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            lo[e] = ch[e];
            hi[e] = e < 4 ? ch[8 + e] : (__bf16)0.f;
        }
        *reinterpret_cast<bf16x8*>(lds + fp * F_PITCH) = lo;
        *reinterpret_cast<bf16x8*>(lds + fp * F_PITCH + 16) = hi;
    }
    __syncthreads();

    // ---- P1: first convolution on the 36 pixel groups of A (wave w: groups w, w + 4, ...) ---------------------------
    constexpr int NG = A_FRAGS / 4;
    f32x4 acc0[NG][2];
    int fbase[NG];
#pragma unroll
    for (int t = 0; t < NG; ++t) {
        acc0[t][0] = acc0[t][1] = f32x4{0.f, 0.f, 0.f, 0.f};
        int p = (wave + 4 * t) * 16 + p16;
        p = p < A_N ? p : A_N - 1;
        const int r = p / A_W, c = p - r * A_W;
        fbase[t] = RGB ? (r * F_W + c) * 8 : (r * F_W + c) * F_PITCH + (g & 1) * 16;
    }
#pragma unroll
    for (int s = 0; s < KS0; ++s) {
        bf16x8 xf[NG];
        if constexpr (RGB) {
            // K step s: lane group g holds taps 8 s + 2 g and 8 s + 2 g + 1 (taps beyond the ninth: finite values times zero weights)
            const int ta = 8 * s + 2 * g, tb = ta + 1;
            const int pa = ta < 9 ? ta : 0, pb = tb < 9 ? tb : 0;
            const int offa = ((pa / 3) * F_W + pa % 3) * 8, offb = ((pb / 3) * F_W + pb % 3) * 8;
#pragma unroll
            for (int t = 0; t < NG; ++t) {
                const bf16x4 lo = *reinterpret_cast<const bf16x4*>(lds + fbase[t] + offa);
                const bf16x4 hi = *reinterpret_cast<const bf16x4*>(lds + fbase[t] + offb);
                xf[t] = bf16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
            }
        } else {
            const int tap = 2 * s + (g >> 1);           // K-step s covers taps 2 s and 2 s + 1; tap 9 is padding
            const int tp = tap < 9 ? tap : 0;
            const int kh = tp / 3, kw = tp - kh * 3;
            const int toff = (kh * F_W + kw) * F_PITCH;
#pragma unroll
            for (int t = 0; t < NG; ++t) {
                xf[t] = *reinterpret_cast<const bf16x8*>(lds + fbase[t] + toff);   // (tap 9: finite values times zero weights)
            }
        }
#pragma unroll
        for (int t = 0; t < NG; ++t)
#pragma unroll
            for (int i = 0; i < 2; ++i) acc0[t][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[i][s], xf[t], acc0[t][i], 0, 0, 0);
    }
    // weights of the second convolution (requested here: their latency runs under the epilogue below).  Waves split
    // 2 (channels: wn) x 2 (rows: wm); MFMA i of a wave takes channel 32 wn + 8 (row>>2) + 4 i + (row&3).
    const int wn = wave & 1, wm = wave >> 1;
    bf16x8 w1f[2][9];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int t = 0; t < 9; ++t)
            w1f[i][t] = *reinterpret_cast<const bf16x8*>(a.w1 + ((size_t)(32 * wn + 4 * i + wrow) * 288 + t * 32 + g * 8) * 2);
    __syncthreads();     // every wave is done reading F: A may take its place
    // pixels of A outside the map are the second convolution's zero padding (border workgroups only)
    const float4 b0lo = *reinterpret_cast<const float4*>(a.b0 + 8 * g), b0hi = *reinterpret_cast<const float4*>(a.b0 + 8 * g + 4);
    const bool interior1 = ar0 >= 0 && ar0 + A_H <= a.H1 && ac0 >= 0 && ac0 + A_W <= a.W1;     // workgroup-uniform
#pragma unroll
    for (int t = 0; t < NG; ++t) {
        const int p = (wave + 4 * t) * 16 + p16;
        const int r = p / A_W, c = p - r * A_W;
        bf16x8 o = finish8<SILU0>(acc0[t][0], acc0[t][1], b0lo, b0hi, a.slope0);
        if (!interior1 && !((unsigned)(ar0 + r) < (unsigned)a.H1 && (unsigned)(ac0 + c) < (unsigned)a.W1)) o = bf16x8{};
        const int slot = r * A_ROW + (c & 1) * A_HALF + (c >> 1);
        if (p < A_N) *reinterpret_cast<bf16x8*>(lds + slot * A_PITCH + ((g ^ ((slot >> 1) & 3)) << 4)) = o;
    }
    __syncthreads();

    // ---- P2: second convolution: output row 4 wm + j (j < 4), columns p16; tap (kh, kw) of pixel (oi, oj) is A pixel
    // (2 oi + kh, 2 oj + kw) = slot (kw & 1) * 17 + oj + (kw >> 1) of that row
    f32x4 acc1[2][4];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc1[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    // (per-lane bases for taps kw = 0, 1 and for kw = 2, which sits one slot further on: the XOR term differs)
    const char* abase0 = lds + ((2 * 4 * wm) * A_ROW + p16) * A_PITCH + ((g ^ ((p16 >> 1) & 3)) << 4);
    const char* abase1 = lds + ((2 * 4 * wm) * A_ROW + p16 + 1) * A_PITCH + ((g ^ (((p16 + 1) >> 1) & 3)) << 4);
#pragma unroll
    for (int t = 0; t < 9; ++t) {
        const int kh = t / 3, kw = t - kh * 3;
        bf16x8 xf[4];
#pragma unroll
        for (int j = 0; j < 4; ++j)
            xf[j] = *reinterpret_cast<const bf16x8*>((kw >> 1 ? abase1 : abase0) + ((2 * j + kh) * A_ROW + (kw & 1) * A_HALF) * A_PITCH);
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc1[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w1f[i][t], xf[j], acc1[i][j], 0, 0, 0);
    }

    // where a lane's 16-byte piece (channels co .. co + 7 of output pixel row 4 wm + j, column p16) goes
    auto store_piece = [&](int j, int co, const bf16x8& o) {
        const int oi = oi0 + 4 * wm + j, oj = oj0 + p16;
        if (oi >= a.Ho || oj >= a.Wo || co >= a.cout) return;
        const size_t m = ((size_t)n * a.Ho + oi) * a.Wo + oj;
        char* dst = (a.y2 && co >= a.y_split) ? a.y2 + (m * a.y2_ld + a.y2_coff + (co - a.y_split)) * 2
                                              : a.y + (m * a.y_ld + a.y_coff + co) * 2;
        *reinterpret_cast<bf16x8*>(dst) = o;
    };
    const int co = 32 * wn + 8 * g;
    const float4 b1lo = *reinterpret_cast<const float4*>(a.b1 + co), b1hi = *reinterpret_cast<const float4*>(a.b1 + co + 4);
    if constexpr (!FOLLOW) {
#pragma unroll
        for (int j = 0; j < 4; ++j) store_piece(j, co, finish8<SILU1>(acc1[0][j], acc1[1][j], b1lo, b1hi, a.slope1));
        return;
    } else {
        // ---- P3: the finished tile through LDS, then the 1x1 convolution over its 64 channels (two K-steps) ------------
        bf16x8 pwf[2][2];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int row = 32 * wn + 4 * i + wrow;
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                pwf[i][s] = bf16x8{};
                if (row < a.pw_rows) pwf[i][s] = *reinterpret_cast<const bf16x8*>(a.pw + ((size_t)row * 64 + s * 32 + g * 8) * 2);
            }
        }
        __syncthreads();     // every wave is done reading A
#pragma unroll
        for (int j = 0; j < 4; ++j)
            *reinterpret_cast<bf16x8*>(lds + ((4 * wm + j) * 16 + p16) * T_PITCH + (((co >> 3) ^ (p16 & 7)) << 4)) =
                finish8<SILU1>(acc1[0][j], acc1[1][j], b1lo, b1hi, a.slope1);
        __syncthreads();
        f32x4 acc2[2][4];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc2[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            bf16x8 xf[4];
#pragma unroll
            for (int j = 0; j < 4; ++j)
                xf[j] = *reinterpret_cast<const bf16x8*>(lds + ((4 * wm + j) * 16 + p16) * T_PITCH + (((4 * s + g) ^ (p16 & 7)) << 4));
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc2[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(pwf[i][s], xf[j], acc2[i][j], 0, 0, 0);
        }
        if (co < a.cout) {
            const float4 plo = *reinterpret_cast<const float4*>(a.pwb + co), phi = *reinterpret_cast<const float4*>(a.pwb + co + 4);
#pragma unroll
            for (int j = 0; j < 4; ++j) store_piece(j, co, finish8<SILU1>(acc2[0][j], acc2[1][j], plo, phi, a.slope1));
        }
    }
}

template <typename TX, bool FOLLOW, bool RGB = false>
void launch_acts(const PairArgs& a, dim3 grid, hipStream_t s) {
    const bool s0 = a.act0 == YDET_ACT_SILU, s1 = a.act1 == YDET_ACT_SILU;
    if (s0 && s1) hipLaunchKernelGGL((focus_pair_kernel<TX, FOLLOW, true, true, RGB>), grid, dim3(256), 0, s, a);
    else if (!s0 && !s1) hipLaunchKernelGGL((focus_pair_kernel<TX, FOLLOW, false, false, RGB>), grid, dim3(256), 0, s, a);
    else if (s0) hipLaunchKernelGGL((focus_pair_kernel<TX, FOLLOW, true, false, RGB>), grid, dim3(256), 0, s, a);
    else hipLaunchKernelGGL((focus_pair_kernel<TX, FOLLOW, false, true, RGB>), grid, dim3(256), 0, s, a);
}

}  // namespace

int ydet_launch_focus_pair(const ydet_conv_desc* f, const ydet_conv_desc* d, int x_dtype, hipStream_t stream, bool rgb) {
    YDET_CHECK_ARG(f != nullptr && d != nullptr && f->x && f->w && f->bias && d->w && d->bias && d->y, "focus_pair: null pointer");
    YDET_CHECK_ARG(dtype_ok(x_dtype) || (x_dtype == YDET_U8 && !rgb), "focus_pair: bad image dtype");
    YDET_CHECK_ARG(f->groups <= 1 && f->dilation <= 1 && d->groups <= 1 && d->dilation <= 1, "focus_pair: plain convolutions only");
    YDET_CHECK_ARG(f->in_dtype == YDET_BF16 && d->in_dtype == YDET_BF16 && d->out_dtype == YDET_BF16, "focus_pair: bf16 weights and output only");
    YDET_CHECK_ARG(f->N > 0 && f->N <= 65535 && f->H > 0 && f->W > 0 && (rgb || (f->H % 2 == 0 && f->W % 2 == 0)),
                   "focus_pair: image %d x %d x %d", f->N, f->H, f->W);
    YDET_CHECK_ARG(f->Cin == 3 && f->KH == 3 && f->KW == 3 && f->stride == 1 && f->pad == 1 && f->Cout == 32 &&
                       f->kp == (rgb ? 64 : KSTEPS0 * 32),
                   "focus_pair: first layer must be [Focus +] 3x3 / stride 1 / pad 1 with 32 output channels (kp = %d)", rgb ? 64 : KSTEPS0 * 32);
    YDET_CHECK_ARG(f->res == nullptr && f->post_scale == nullptr && f->y2 == nullptr && f->pw_w == nullptr,
                   "focus_pair: first layer: no residual / second stage / destination / follower");
    YDET_CHECK_ARG(3LL * f->H * f->W < (1LL << 31), "focus_pair: image of %d x %d pixels", f->H, f->W);
    const int H1 = rgb ? f->H : f->H / 2, W1 = rgb ? f->W : f->W / 2;
    YDET_CHECK_ARG(!rgb || f->act == d->act, "rgb_pair: both layers share one activation");
    YDET_CHECK_ARG(d->N == f->N && d->H == H1 && d->W == W1 && d->Cin == 32 && d->KH == 3 && d->KW == 3 && d->stride == 2 && d->pad == 1 &&
                       d->Cout == 64 && d->kp == 288 && d->cout_pad >= 64,
                   "focus_pair: second layer must be 3x3 / stride 2 / pad 1, 32 -> 64 channels on the %d x %d map", H1, W1);
    YDET_CHECK_ARG(d->Ho == (H1 - 1) / 2 + 1 && d->Wo == (W1 - 1) / 2 + 1, "focus_pair: second layer output %d x %d", d->Ho, d->Wo);
    YDET_CHECK_ARG(d->res == nullptr && d->post_scale == nullptr && d->x_up == nullptr && d->cout_store == 0,
                   "focus_pair: second layer: no residual / second stage / upsampled source / padded store");
    const int cout = d->pw_w ? d->pw_cout : 64;
    if (d->pw_w) YDET_CHECK_ARG(d->pw_bias && d->pw_kp == 64 && d->pw_cout >= 8 && d->pw_cout <= 64 && d->pw_cout % 8 == 0,
                                "focus_pair: follower with %d outputs (multiple of 8, <= 64; pw_kp = 64)", d->pw_cout);
    if (d->y2) {
        YDET_CHECK_ARG(d->y_split > 0 && d->y_split < cout && d->y_split % 8 == 0 && d->y2_ld % 8 == 0 && d->y2_coff % 8 == 0 &&
                           d->y2_coff >= 0 && d->y2_coff + (cout - d->y_split) <= d->y2_ld && ((uintptr_t)d->y2 & 15) == 0,
                       "focus_pair: second destination (split, ld, coff multiples of 8)");
    }
    const int ny = d->y2 ? d->y_split : cout;
    YDET_CHECK_ARG(d->y_ld % 8 == 0 && d->y_coff % 8 == 0 && d->y_coff >= 0 && d->y_coff + ny <= d->y_ld,
                   "focus_pair: output slice (y_ld, y_coff multiples of 8)");
    YDET_CHECK_ARG(((uintptr_t)f->x & (x_dtype == YDET_U8 ? 1 : 7)) == 0 && ((uintptr_t)f->w & 15) == 0 && ((uintptr_t)d->w & 15) == 0 &&
                       ((uintptr_t)d->y & 15) == 0 && ((uintptr_t)f->bias & 15) == 0 && ((uintptr_t)d->bias & 15) == 0 &&
                       ((uintptr_t)d->pw_w & 15) == 0 && ((uintptr_t)d->pw_bias & 15) == 0,
                   "focus_pair: pointer alignment");
    PairArgs a;
    a.x = (const char*)f->x; a.w0 = (const char*)f->w; a.b0 = f->bias;
    a.w1 = (const char*)d->w; a.b1 = d->bias; a.pw = (const char*)d->pw_w; a.pwb = d->pw_bias;
    a.y = (char*)d->y; a.y2 = (char*)d->y2;
    a.N = f->N; a.H = f->H; a.W = f->W; a.H1 = H1; a.W1 = W1; a.Ho = d->Ho; a.Wo = d->Wo;
    a.cout = cout;
    a.y_ld = d->y_ld; a.y_coff = d->y_coff; a.y2_ld = d->y2_ld; a.y2_coff = d->y2_coff; a.y_split = d->y_split;
    a.tiles_x = (a.Wo + T_W - 1) / T_W; a.tiles_y = (a.Ho + T_H - 1) / T_H;
    a.act0 = f->act; a.slope0 = f->act == YDET_ACT_NONE ? 1.0f : f->slope;
    a.act1 = d->act; a.slope1 = d->act == YDET_ACT_NONE ? 1.0f : d->slope;
    a.pw_rows = d->pw_w ? d->pw_cout : 0;
    const dim3 grid((unsigned)(a.tiles_x * a.tiles_y), (unsigned)a.N);
    if (rgb) {      // (instantiated for the activation pairs the Darknet trunks use: both SiLU or both leaky / none)
        const bool silu = a.act0 == YDET_ACT_SILU, follow = d->pw_w != nullptr, f32 = x_dtype == YDET_F32;
        auto go = [&](auto kernel) { hipLaunchKernelGGL(kernel, grid, dim3(256), 0, stream, a); };
        if (f32 && follow) silu ? go(focus_pair_kernel<float, true, true, true, true>) : go(focus_pair_kernel<float, true, false, false, true>);
        else if (f32) silu ? go(focus_pair_kernel<float, false, true, true, true>) : go(focus_pair_kernel<float, false, false, false, true>);
        else if (follow) silu ? go(focus_pair_kernel<__bf16, true, true, true, true>) : go(focus_pair_kernel<__bf16, true, false, false, true>);
        else silu ? go(focus_pair_kernel<__bf16, false, true, true, true>) : go(focus_pair_kernel<__bf16, false, false, false, true>);
    } else if (d->pw_w) {
        if (x_dtype == YDET_F32) launch_acts<float, true>(a, grid, stream);
        else if (x_dtype == YDET_U8) launch_acts<uint8_t, true>(a, grid, stream);
        else launch_acts<__bf16, true>(a, grid, stream);
    } else {
        if (x_dtype == YDET_F32) launch_acts<float, false>(a, grid, stream);
        else if (x_dtype == YDET_U8) launch_acts<uint8_t, false>(a, grid, stream);
        else launch_acts<__bf16, false>(a, grid, stream);
    }
    YDET_CHECK_LAUNCH("focus_pair_kernel");
    return YDET_OK;
}

extern "C" int ydet_focus_conv_pair_nchw(const ydet_conv_desc* focus, const ydet_conv_desc* next, int x_dtype, void* stream) {
    clear_error();
    return ydet_launch_focus_pair(focus, next, x_dtype, as_stream(stream), false);
}

extern "C" int ydet_rgb_conv_pair_nchw(const ydet_conv_desc* first, const ydet_conv_desc* next, int x_dtype, void* stream) {
    clear_error();
    return ydet_launch_focus_pair(first, next, x_dtype, as_stream(stream), true);
}
