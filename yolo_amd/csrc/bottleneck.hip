// A 32- or 64-channel residual bottleneck as one kernel (the 32-channel form first): y = x + act(conv3x3(act(conv1x1(x)))), or
// y = act(x + conv3x3(act(conv1x1(x)))) -- the CSP residual block of utils/moduleUtils.py:79-91 -- (yolov5_official/modules.py:208-218
// with c1 = c2 = c_ = 32, the first C3 stage of YOLOv5s).  Run as two convolutions this pair is bound by L2 traffic
// (the 3x3 tap walk re-reads every input pixel nine times) and by the round trip of the intermediate map through HBM;
// here a workgroup keeps the input patch and the intermediate patch of an 8 x 32 output tile in LDS.
#include "ydet_internal.h"

namespace {
using namespace ydet;

constexpr int TH = 8, TW = 32;                 // output tile (rows x columns)
constexpr int PH = TH + 2, PW = TW + 2;        // patch with the 3x3 halo
constexpr int NPIX = PH * PW;                  // 340 patch pixels
constexpr int NGRP = (NPIX + 15) / 16;         // 22 groups of 16 patch pixels for the 1x1 stage
constexpr int PITCH = 96;                      // bytes per patch pixel in LDS: 64 of data + 32.  A ds_read_b128 is served in
                                               // lane groups {0-3, 12-15, 20-27}, {4-11, 16-19, 28-31} (+32): with pixel =
                                               // lane & 15 and 16-byte chunk = lane >> 4 a group's sixteen 16-byte slots
                                               // (6 * pixel + chunk) mod 16 are all different at this pitch (at 80 bytes
                                               // seven of every eight collided)
constexpr int PATCH_BYTES = NPIX * PITCH;      // 32640: two patches stay just inside the 64 KB of static LDS
constexpr int NLOAD = (NPIX * 4 + 255) / 256;  // 16-byte pieces of the input patch per thread

struct BneckArgs {
    const char* x;
    char* y;
    const char* w1;
    const float* b1;
    const char* w2;
    const float* b2;
    int N, H, W;
    int x_ld, x_coff, y_ld, y_coff;
    int kp1, kp2;
    int act;
    float slope;
    int add;
    int res_before;       // the shortcut goes in before the second activation: act(x + conv3x3(...)) (utils/moduleUtils.py CSPResBlock)
    int tiles_x, tiles_per_img, n_tiles;
    // C3 tail (32-channel kernel only): cv3 = 1x1 over [this bottleneck's output | z], 64 -> 64, written to y3
    const char* w3;
    const float* b3;
    const char* z;
    char* y3;
    int z_ld, z_coff, y3_ld, y3_coff, kp3;
};

// The activation is a compile-time choice: read from the arguments it became a chain of branches around every single
// value (the kernels finish 8 to 64 values per lane and tile).
template <bool SILU>
__device__ __forceinline__ float act_fn(float v, float slope) {
    if constexpr (SILU) return v * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(v * -1.44269504088896340736f));
    return v > 0.f ? v : v * slope;
}

// TAIL: the block is the last bottleneck of a C3 whose concat [bottleneck output | cv2(x)] feeds cv3 (a 1x1, 64 -> 64,
// yolov5_official/modules.py:252-253).  The eight channels a lane has just finished for its pixel ARE the B fragment of
// cv3's first K step, the other half's eight channels (one 16-byte load) the second: cv3 runs on the tile in registers
// and neither the bottleneck's output nor the concat is written.
template <bool TAIL, bool SILU>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void bottleneck_c32_kernel(const BneckArgs a) {
    // xs: input patch, bf16 [pixel][32] at PITCH; ts: act(conv1x1) of the patch, zero outside the image.  One array: the
    // 1x1 stage's last pixel group reads 12 rows past xs, i.e. into ts (those results are dropped)
    __shared__ __attribute__((aligned(16))) char smem[2 * PATCH_BYTES];
    char* xs = smem;
    char* ts = smem + PATCH_BYTES;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int px = lane & 15, q = lane >> 4;
    const float slope = a.act == YDET_ACT_LEAKY ? a.slope : 1.0f;   // ACT_NONE: leaky with slope 1

    // Weights as MFMA A operands, kept in registers for every tile of this workgroup.  A row r of fragment i is output
    // channel (r>>2)*8 + i*4 + (r&3): the D rows a lane holds (4q..4q+3 of both fragments) are then the eight
    // consecutive channels 8q..8q+7, one 16-byte piece.
    const int arow = lane & 15;
    bf16x8 w1f[2], w2f[9][2];
    float b1v[2][4], b2v[2][4];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int co = (arow >> 2) * 8 + i * 4 + (arow & 3);
        w1f[i] = *reinterpret_cast<const bf16x8*>(a.w1 + ((size_t)co * a.kp1 + q * 8) * 2);
#pragma unroll
        for (int t = 0; t < 9; ++t)
            w2f[t][i] = *reinterpret_cast<const bf16x8*>(a.w2 + ((size_t)co * a.kp2 + t * 32 + q * 8) * 2);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            b1v[i][e] = a.b1[q * 8 + i * 4 + e];
            b2v[i][e] = a.b2[q * 8 + i * 4 + e];
        }
    }

    // cv3: fragments (2h, 2h+1) cover output channels 32h .. 32h+31 with the same row permutation, K step s = half s
    bf16x8 w3f[TAIL ? 4 : 1][2];
    float b3v[TAIL ? 4 : 1][4];
    if constexpr (TAIL) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int co = (i >> 1) * 32 + (arow >> 2) * 8 + (i & 1) * 4 + (arow & 3);
#pragma unroll
            for (int h = 0; h < 2; ++h)
                w3f[i][h] = *reinterpret_cast<const bf16x8*>(a.w3 + ((size_t)co * a.kp3 + h * 32 + q * 8) * 2);
#pragma unroll
            for (int e = 0; e < 4; ++e) b3v[i][e] = a.b3[(i >> 1) * 32 + q * 8 + (i & 1) * 4 + e];
        }
    }

    // input patch pieces of a tile: thread-private registers, fetched one tile ahead
    uint4 pre[NLOAD];
    auto fetch = [&](int tile) {
        const int n = tile / a.tiles_per_img, r = tile - n * a.tiles_per_img;
        const int ty = r / a.tiles_x, tx = r - ty * a.tiles_x;
        const int iy0 = ty * TH - 1, ix0 = tx * TW - 1;
#pragma unroll
        for (int k = 0; k < NLOAD; ++k) {
            const int idx = tid + k * 256, p = idx >> 2, ch = idx & 3;
            const int pr = p / PW, pc = p - pr * PW;
            const int iy = iy0 + pr, ix = ix0 + pc;
            const bool ok = p < NPIX && (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W;
            pre[k] = uint4{0, 0, 0, 0};
            if (ok)
                pre[k] = *reinterpret_cast<const uint4*>(a.x + ((((size_t)n * a.H + iy) * a.W + ix) * a.x_ld + a.x_coff + ch * 8) * 2);
        }
    };

    int tile = blockIdx.x;
    if (tile < a.n_tiles) fetch(tile);
    for (; tile < a.n_tiles; tile += gridDim.x) {
        const int n = tile / a.tiles_per_img, r = tile - n * a.tiles_per_img;
        const int ty = r / a.tiles_x, tx = r - ty * a.tiles_x;
        const int oy0 = ty * TH, ox0 = tx * TW;
        // ---- patch -> LDS ---------------------------------------------------------------------------------------
#pragma unroll
        for (int k = 0; k < NLOAD; ++k) {
            const int idx = tid + k * 256, p = idx >> 2, ch = idx & 3;
            if (p < NPIX) *reinterpret_cast<uint4*>(xs + p * PITCH + ch * 16) = pre[k];
        }
        __syncthreads();
        if (tile + (int)gridDim.x < a.n_tiles) fetch(tile + gridDim.x);   // lands while this tile is computed
        // ---- 1x1 over the whole patch -> ts (zero where the patch leaves the image: the 3x3's padding) -----------
        for (int g = wave; g < NGRP; g += 4) {
            const int p = g * 16 + px;
            const bf16x8 xb = *reinterpret_cast<const bf16x8*>(xs + p * PITCH + q * 16);
            f32x4 acc0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w1f[0], xb, f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
            f32x4 acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w1f[1], xb, f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
            const int pr = p / PW, pc = p - pr * PW;
            const bool in = p < NPIX && (unsigned)(oy0 - 1 + pr) < (unsigned)a.H && (unsigned)(ox0 - 1 + pc) < (unsigned)a.W;
            bf16x8 t;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float v0 = act_fn<SILU>(acc0[e] + b1v[0][e], slope), v1 = act_fn<SILU>(acc1[e] + b1v[1][e], slope);
                t[e] = (__bf16)(in ? v0 : 0.f);
                t[4 + e] = (__bf16)(in ? v1 : 0.f);
            }
            if (p < NPIX) *reinterpret_cast<bf16x8*>(ts + p * PITCH + q * 16) = t;
        }
        __syncthreads();
        // ---- 3x3 over ts, + residual, -> y: a wave owns two output rows, i.e. four groups of 16 pixels ----------
        // (with the cv3 tail the groups run one after the other: unrolled they need more than the 256 registers that
        // two workgroups per CU leave a lane)
#pragma unroll(TAIL ? 1 : 4)
        for (int gi = 0; gi < 4; ++gi) {
            const int row = wave * 2 + (gi >> 1), col0 = (gi & 1) * 16;
            const int oy = oy0 + row, ox = ox0 + col0 + px;
            const bool inside = oy < a.H && ox < a.W;
            bf16x8 zb = {};
            if constexpr (TAIL) {   // the other half of cv3's input, requested before the tap walk
                if (inside) zb = *reinterpret_cast<const bf16x8*>(a.z + ((((size_t)n * a.H + oy) * a.W + ox) * a.z_ld + a.z_coff + q * 8) * 2);
            }
            f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                const int p = (row + t / 3) * PW + col0 + t % 3 + px;
                const bf16x8 tb = *reinterpret_cast<const bf16x8*>(ts + p * PITCH + q * 16);
                acc0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w2f[t][0], tb, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w2f[t][1], tb, acc1, 0, 0, 0);
            }
            // (the shortcut is a select, not a branch: the spot in xs is readable either way)
            const bf16x8 res = *reinterpret_cast<const bf16x8*>(xs + ((row + 1) * PW + col0 + 1 + px) * PITCH + q * 16);
            const bool before = a.res_before != 0, after = a.add != 0 && !before;
            bf16x8 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float r0 = (float)res[e], r1 = (float)res[4 + e];
                float u0 = acc0[e] + b2v[0][e], u1 = acc1[e] + b2v[1][e];
                u0 = before ? u0 + r0 : u0;
                u1 = before ? u1 + r1 : u1;
                float v0 = act_fn<SILU>(u0, slope), v1 = act_fn<SILU>(u1, slope);
                v0 = after ? v0 + r0 : v0;
                v1 = after ? v1 + r1 : v1;
                o[e] = (__bf16)v0;
                o[4 + e] = (__bf16)v1;
            }
            if constexpr (!TAIL) {
                if (inside)
                    *reinterpret_cast<bf16x8*>(a.y + ((((size_t)n * a.H + oy) * a.W + ox) * a.y_ld + a.y_coff + q * 8) * 2) = o;
            } else {
                f32x4 acc3[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    acc3[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w3f[i][0], o, f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
                    acc3[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w3f[i][1], zb, acc3[i], 0, 0, 0);
                }
                char* yrow = a.y3 + ((((size_t)n * a.H + oy) * a.W + ox) * a.y3_ld + a.y3_coff) * 2;
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    bf16x8 o3;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        o3[e] = (__bf16)act_fn<SILU>(acc3[2 * h][e] + b3v[2 * h][e], slope);
                        o3[4 + e] = (__bf16)act_fn<SILU>(acc3[2 * h + 1][e] + b3v[2 * h + 1][e], slope);
                    }
                    if (inside) *reinterpret_cast<bf16x8*>(yrow + (h * 32 + q * 8) * 2) = o3;
                }
            }
        }
        __syncthreads();   // xs / ts are rewritten by the next tile
    }
}


// ---- 64 channels throughout -----------------------------------------------------------------------------------------
// Same scheme for the 64-channel bottlenecks (the second C3 stage of YOLOv5s at 80 x 80), where the weights no longer
// fit one wave's registers: the four waves split the OUTPUT CHANNELS (16 each: 18 A fragments of the 3x3 stage and 2
// of the 1x1 stage per lane) and every wave walks all pixel groups of an 8 x 16 output tile.  Patches at a 160-byte
// pixel pitch (128 of data + 32: the sixteen 16-byte slots (10 * pixel + chunk) mod 16 of a ds_read_b128 lane group are
// all different; at 144 bytes seven of every eight collided).  The finished tile
// goes through LDS once more so that global stores are 16-byte pieces of whole pixels.
namespace c64 {
constexpr int TH = 8, TW = 16;
constexpr int PH = TH + 2, PW = TW + 2;
constexpr int NPIX = PH * PW;                   // 180 patch pixels
constexpr int NGRP = (NPIX + 15) / 16;          // 12 groups for the 1x1 stage (the last one reads 12 rows of slack)
constexpr int PITCH = 160;
constexpr int PATCH_BYTES = NPIX * PITCH;       // 25920
constexpr int SLACK = (NGRP * 16 - NPIX) * PITCH;
constexpr int NLOAD = (NPIX * 8 + 255) / 256;   // 16-byte pieces of the input patch per thread
}  // namespace c64

template <bool SILU>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void bottleneck_c64_kernel(const BneckArgs a) {
    constexpr int TH = c64::TH, TW = c64::TW, PW = c64::PW, NPIX = c64::NPIX, NGRP = c64::NGRP, PITCH = c64::PITCH;
    constexpr int PATCH_BYTES = c64::PATCH_BYTES, SLACK = c64::SLACK, NLOAD = c64::NLOAD;
    // xs | ts | slack: the 1x1 stage's last group reads past xs into ts (its results are dropped), ts likewise into slack
    __shared__ __attribute__((aligned(16))) char smem[2 * PATCH_BYTES + SLACK];
    char* xs = smem;
    char* ts = smem + PATCH_BYTES;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int px = lane & 15, q = lane >> 4;
    const float slope = a.act == YDET_ACT_LEAKY ? a.slope : 1.0f;

    // A fragments of this wave's 16 output channels: row r = lane & 15 is channel 16 * wave + r, k chunk q
    const int co_a = wave * 16 + (lane & 15);
    bf16x8 w1f[2], w2f[9][2];
#pragma unroll
    for (int kh2 = 0; kh2 < 2; ++kh2) {
        w1f[kh2] = *reinterpret_cast<const bf16x8*>(a.w1 + ((size_t)co_a * a.kp1 + kh2 * 32 + q * 8) * 2);
#pragma unroll
        for (int t = 0; t < 9; ++t)
            w2f[t][kh2] = *reinterpret_cast<const bf16x8*>(a.w2 + ((size_t)co_a * a.kp2 + t * 64 + kh2 * 32 + q * 8) * 2);
    }
    // D rows of a lane: channels 16 * wave + 4 q .. + 3
    const int co_d = wave * 16 + q * 4;
    float b1v[4], b2v[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) { b1v[e] = a.b1[co_d + e]; b2v[e] = a.b2[co_d + e]; }

    uint4 pre[NLOAD];
    auto fetch = [&](int tile) {
        const int n = tile / a.tiles_per_img, r = tile - n * a.tiles_per_img;
        const int ty = r / a.tiles_x, tx = r - ty * a.tiles_x;
        const int iy0 = ty * TH - 1, ix0 = tx * TW - 1;
#pragma unroll
        for (int k = 0; k < NLOAD; ++k) {
            const int idx = tid + k * 256, p = idx >> 3, ch = idx & 7;
            const int pr = p / PW, pc = p - pr * PW;
            const int iy = iy0 + pr, ix = ix0 + pc;
            const bool ok = p < NPIX && (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W;
            pre[k] = uint4{0, 0, 0, 0};
            if (ok)
                pre[k] = *reinterpret_cast<const uint4*>(a.x + ((((size_t)n * a.H + iy) * a.W + ix) * a.x_ld + a.x_coff + ch * 8) * 2);
        }
    };

    // two waves per SIMD (the A fragments alone are 80 registers): the next tile's patch is fetched into registers
    // while this one is computed
    int tile = blockIdx.x;
    if (tile < a.n_tiles) fetch(tile);
    for (; tile < a.n_tiles; tile += gridDim.x) {
        const int n = tile / a.tiles_per_img, r = tile - n * a.tiles_per_img;
        const int ty = r / a.tiles_x, tx = r - ty * a.tiles_x;
        const int oy0 = ty * TH, ox0 = tx * TW;
        // ---- patch -> LDS ---------------------------------------------------------------------------------------
#pragma unroll
        for (int k = 0; k < NLOAD; ++k) {
            const int idx = tid + k * 256, p = idx >> 3, ch = idx & 7;
            if (p < NPIX) *reinterpret_cast<uint4*>(xs + p * PITCH + ch * 16) = pre[k];
        }
        __syncthreads();
        if (tile + (int)gridDim.x < a.n_tiles) fetch(tile + gridDim.x);
        // ---- 1x1 over the whole patch: this wave's 16 channels of every patch pixel -> ts -------------------------
        static_assert(NGRP % 4 == 0, "four pixel groups at a time");
#pragma unroll
        for (int g0 = 0; g0 < NGRP; g0 += 4) {     // four groups in flight: their LDS reads, then their MFMAs
            bf16x8 x0[4], x1[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int p = (g0 + u) * 16 + px;
                x0[u] = *reinterpret_cast<const bf16x8*>(xs + p * PITCH + q * 16);
                x1[u] = *reinterpret_cast<const bf16x8*>(xs + p * PITCH + 64 + q * 16);
            }
            f32x4 acc[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) acc[u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w1f[0], x0[u], f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
#pragma unroll
            for (int u = 0; u < 4; ++u) acc[u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w1f[1], x1[u], acc[u], 0, 0, 0);
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int p = (g0 + u) * 16 + px;
                const int pr = p / PW, pc = p - pr * PW;
                const bool in = p < NPIX && (unsigned)(oy0 - 1 + pr) < (unsigned)a.H && (unsigned)(ox0 - 1 + pc) < (unsigned)a.W;
                bf16x4 t;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float v = act_fn<SILU>(acc[u][e] + b1v[e], slope);
                    t[e] = (__bf16)(in ? v : 0.f);
                }
                if (p < NPIX) *reinterpret_cast<bf16x4*>(ts + p * PITCH + co_d * 2) = t;
            }
        }
        __syncthreads();
        // ---- 3x3 over ts: this wave's 16 channels of all 8 rows (groups of 16 pixels), four groups at a time --------
        // (a finished value replaces the residual it was added to: only this lane reads that spot of xs)
#pragma unroll
        for (int gs = 0; gs < TH; gs += 4) {
            f32x4 acc[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) acc[u] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int t = 0; t < 9; ++t) {
#pragma unroll
                for (int kh2 = 0; kh2 < 2; ++kh2) {
                    bf16x8 tb[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u)
                        tb[u] = *reinterpret_cast<const bf16x8*>(ts + ((gs + u + t / 3) * PW + t % 3 + px) * PITCH + kh2 * 64 + q * 16);
#pragma unroll
                    for (int u = 0; u < 4; ++u) acc[u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w2f[t][kh2], tb[u], acc[u], 0, 0, 0);
                }
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int row = gs + u;
                char* spot = xs + ((row + 1) * PW + 1 + px) * PITCH + co_d * 2;
                const bf16x4 res = *reinterpret_cast<const bf16x4*>(spot);
                const bool before = a.res_before != 0, after = a.add != 0 && !before;
                bf16x4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float r = (float)res[e];
                    float v = acc[u][e] + b2v[e];
                    v = act_fn<SILU>(before ? v + r : v, slope);
                    v = after ? v + r : v;
                    o[e] = (__bf16)v;
                }
                *reinterpret_cast<bf16x4*>(spot) = o;
            }
        }
        __syncthreads();   // the finished tile sits at the centre pixels of xs
#pragma unroll
        for (int k = 0; k < TH * TW * 8 / 256; ++k) {
            const int idx = tid + k * 256, p = idx >> 3, ch = idx & 7;
            const int oy = oy0 + p / TW, ox = ox0 + p % TW;
            const uint4 v = *reinterpret_cast<const uint4*>(xs + ((p / TW + 1) * PW + 1 + p % TW) * PITCH + ch * 16);
            if (oy < a.H && ox < a.W)
                *reinterpret_cast<uint4*>(a.y + ((((size_t)n * a.H + oy) * a.W + ox) * a.y_ld + a.y_coff + ch * 8) * 2) = v;
        }
        __syncthreads();   // xs / ts are rewritten by the next tile
    }
}

// ---- 64 -> 32 -> 64 channels: the residual block of the Darknet-53 trunk at half the image size ---------------------------
// x + act(conv3x3(act(conv1x1(x)))) with a 1x1 that halves the channels (utils/moduleUtils.py:45-59, resblock(64)): run as
// two convolutions the pair writes and re-reads the 32-channel map -- 378 MB for 64 images of 608 x 608 -- and the 3x3, with
// K = 288 and 64 output channels, leaves the matrix pipe mostly idle.  Same scheme as above on an 8 x 16 output tile:
// xs = input patch (64 channels, 160-byte pitch), ts = act(conv1x1) of the patch (32 channels, 96-byte pitch, zero outside
// the image).  1x1 stage: every wave takes all 32 channels (two A fragments x two K steps) of a quarter of the pixel
// groups.  3x3 stage: the waves split 2 (channel halves) x 2 (row halves), 18 A fragments each in registers, a pixel
// fragment feeds two MFMAs.
#ifndef RB64_WPE
#define RB64_WPE 2
#endif
namespace rb64 {
constexpr int TH = 8, TW = 16;
constexpr int PH = TH + 2, PW = TW + 2;
constexpr int NPIX = PH * PW;                   // 180 patch pixels
constexpr int NGRP = (NPIX + 15) / 16;          // 12 groups (the last one reads 12 pixels past xs, into ts: dropped)
constexpr int XP = 160, TP = 96;
constexpr int XS_BYTES = NPIX * XP, TS_BYTES = NPIX * TP;
constexpr int NLOAD = (NPIX * 8 + 255) / 256;
static_assert(NGRP * 16 * XP <= XS_BYTES + TS_BYTES, "the last pixel group of the 1x1 stage stays inside the array");
static_assert(NGRP % 4 == 0, "pixel groups split evenly over four waves");
}  // namespace rb64

template <bool SILU>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(RB64_WPE, RB64_WPE))) void resblock_c64_kernel(const BneckArgs a) {
    constexpr int TH = rb64::TH, TW = rb64::TW, PW = rb64::PW, NPIX = rb64::NPIX, NGRP = rb64::NGRP, XP = rb64::XP, TP = rb64::TP;
    constexpr int NLOAD = rb64::NLOAD;
    __shared__ __attribute__((aligned(16))) char smem[rb64::XS_BYTES + rb64::TS_BYTES];
    char* xs = smem;
    char* ts = smem + rb64::XS_BYTES;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int px = lane & 15, q = lane >> 4;
    const int hc = wave & 1, hr = wave >> 1;      // 3x3 stage: channel half, row half
    const float slope = a.act == YDET_ACT_LEAKY ? a.slope : 1.0f;   // ACT_NONE: leaky with slope 1

    // A row r of fragment i is channel (r>>2)*8 + i*4 + (r&3) of its 32: the D rows a lane holds (4q..4q+3 of both
    // fragments) are then eight consecutive channels, one 16-byte piece.
    const int arow = lane & 15;
    bf16x8 w1f[2][2], w2f[9][2];
    float b1v[2][4], b2v[2][4];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int co = (arow >> 2) * 8 + i * 4 + (arow & 3);
#pragma unroll
        for (int s = 0; s < 2; ++s)
            w1f[i][s] = *reinterpret_cast<const bf16x8*>(a.w1 + ((size_t)co * a.kp1 + s * 32 + q * 8) * 2);
#pragma unroll
        for (int t = 0; t < 9; ++t)
            w2f[t][i] = *reinterpret_cast<const bf16x8*>(a.w2 + ((size_t)(32 * hc + co) * a.kp2 + t * 32 + q * 8) * 2);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            b1v[i][e] = a.b1[q * 8 + i * 4 + e];
            b2v[i][e] = a.b2[32 * hc + q * 8 + i * 4 + e];
        }
    }

    uint4 pre[NLOAD];
    auto fetch = [&](int tile) {
        const int n = tile / a.tiles_per_img, r = tile - n * a.tiles_per_img;
        const int ty = r / a.tiles_x, tx = r - ty * a.tiles_x;
        const int iy0 = ty * TH - 1, ix0 = tx * TW - 1;
#pragma unroll
        for (int k = 0; k < NLOAD; ++k) {
            const int idx = tid + k * 256, p = idx >> 3, ch = idx & 7;
            const int pr = p / PW, pc = p - pr * PW;
            const int iy = iy0 + pr, ix = ix0 + pc;
            const bool ok = p < NPIX && (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W;
            pre[k] = uint4{0, 0, 0, 0};
            if (ok)
                pre[k] = *reinterpret_cast<const uint4*>(a.x + ((((size_t)n * a.H + iy) * a.W + ix) * a.x_ld + a.x_coff + ch * 8) * 2);
        }
    };

    int tile = blockIdx.x;
    if (tile < a.n_tiles) fetch(tile);
    for (; tile < a.n_tiles; tile += gridDim.x) {
        const int n = tile / a.tiles_per_img, r = tile - n * a.tiles_per_img;
        const int ty = r / a.tiles_x, tx = r - ty * a.tiles_x;
        const int oy0 = ty * TH, ox0 = tx * TW;
        // ---- patch -> LDS ---------------------------------------------------------------------------------------
#pragma unroll
        for (int k = 0; k < NLOAD; ++k) {
            const int idx = tid + k * 256, p = idx >> 3, ch = idx & 7;
            if (p < NPIX) *reinterpret_cast<uint4*>(xs + p * XP + ch * 16) = pre[k];
        }
        __syncthreads();
        if (tile + (int)gridDim.x < a.n_tiles) fetch(tile + gridDim.x);   // lands while this tile is computed
        // ---- 1x1 over the whole patch -> ts (zero where the patch leaves the image: the 3x3's padding) -----------
#pragma unroll
        for (int gi = 0; gi < NGRP / 4; ++gi) {
            const int p = (wave + 4 * gi) * 16 + px;
            const bf16x8 x0 = *reinterpret_cast<const bf16x8*>(xs + p * XP + q * 16);
            const bf16x8 x1 = *reinterpret_cast<const bf16x8*>(xs + p * XP + 64 + q * 16);
            f32x4 acc0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w1f[0][0], x0, f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
            f32x4 acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w1f[1][0], x0, f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w1f[0][1], x1, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w1f[1][1], x1, acc1, 0, 0, 0);
            const int pr = p / PW, pc = p - pr * PW;
            const bool in = p < NPIX && (unsigned)(oy0 - 1 + pr) < (unsigned)a.H && (unsigned)(ox0 - 1 + pc) < (unsigned)a.W;
            bf16x8 t;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float v0 = act_fn<SILU>(acc0[e] + b1v[0][e], slope), v1 = act_fn<SILU>(acc1[e] + b1v[1][e], slope);
                t[e] = (__bf16)(in ? v0 : 0.f);
                t[4 + e] = (__bf16)(in ? v1 : 0.f);
            }
            if (p < NPIX) *reinterpret_cast<bf16x8*>(ts + p * TP + q * 16) = t;
        }
        __syncthreads();
        // ---- 3x3 over ts, + residual, -> y: this wave's 32 channels of its four output rows ------------------------
#pragma unroll
        for (int gi = 0; gi < 4; ++gi) {
            const int row = 4 * hr + gi;
            const int oy = oy0 + row, ox = ox0 + px;
            f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                const int p = (row + t / 3) * PW + t % 3 + px;
                const bf16x8 tb = *reinterpret_cast<const bf16x8*>(ts + p * TP + q * 16);
                acc0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w2f[t][0], tb, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w2f[t][1], tb, acc1, 0, 0, 0);
            }
            // (the shortcut is a select, not a branch: the spot in xs is readable either way)
            const bf16x8 res = *reinterpret_cast<const bf16x8*>(xs + ((row + 1) * PW + 1 + px) * XP + hc * 64 + q * 16);
            const bool before = a.res_before != 0, after = a.add != 0 && !before;
            bf16x8 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float r0 = (float)res[e], r1 = (float)res[4 + e];
                float u0 = acc0[e] + b2v[0][e], u1 = acc1[e] + b2v[1][e];
                u0 = before ? u0 + r0 : u0;
                u1 = before ? u1 + r1 : u1;
                float v0 = act_fn<SILU>(u0, slope), v1 = act_fn<SILU>(u1, slope);
                v0 = after ? v0 + r0 : v0;
                v1 = after ? v1 + r1 : v1;
                o[e] = (__bf16)v0;
                o[4 + e] = (__bf16)v1;
            }
            if (oy < a.H && ox < a.W)
                *reinterpret_cast<bf16x8*>(a.y + ((((size_t)n * a.H + oy) * a.W + ox) * a.y_ld + a.y_coff + 32 * hc + q * 8) * 2) = o;
        }
        __syncthreads();   // xs / ts are rewritten by the next tile
    }
}

}  // namespace

int ydet_launch_bottleneck(const ydet_conv_desc* c1, const ydet_conv_desc* c2, const ydet_conv_desc* c3, hipStream_t stream) {
    using namespace ydet;
    YDET_CHECK_ARG(c1 && c2, "bottleneck_fused: null descriptor");
    YDET_CHECK_ARG(c1->groups <= 1 && c1->dilation <= 1 && c2->groups <= 1 && c2->dilation <= 1 &&
                       (!c3 || (c3->groups <= 1 && c3->dilation <= 1)), "bottleneck_fused: plain convolutions (one group, no dilation)");
    YDET_CHECK_ARG(c1->x && c1->w && c1->bias && c2->w && c2->bias && (c2->y || c3), "bottleneck_fused: null x/w/bias/y pointer");
    YDET_CHECK_ARG(c1->in_dtype == YDET_BF16 && c1->out_dtype == YDET_BF16 && c2->in_dtype == YDET_BF16 &&
                       c2->out_dtype == YDET_BF16, "bottleneck_fused: bf16 only");
    const int C = c1->Cin, Cm = c1->Cout;         // block channels, channels between the two stages
    YDET_CHECK_ARG(((C == 32 || C == 64) && Cm == C && c2->Cin == C && c2->Cout == C) ||
                       (C == 64 && Cm == 32 && c2->Cin == 32 && c2->Cout == 64 && !c3),
                   "bottleneck_fused: 32 or 64 channels throughout, or 64 -> 32 -> 64");
    YDET_CHECK_ARG(c1->KH == 1 && c1->KW == 1 && c1->stride == 1 && c1->pad == 0 && c2->KH == 3 && c2->KW == 3 &&
                       c2->stride == 1 && c2->pad == 1, "bottleneck_fused: a 1x1 followed by a 3x3 / stride 1 / pad 1");
    YDET_CHECK_ARG(c1->N > 0 && c1->H > 0 && c1->W > 0 && c2->N == c1->N && c2->H == c1->H && c2->W == c1->W &&
                       c2->Ho == c1->H && c2->Wo == c1->W, "bottleneck_fused: shapes of the two stages differ");
    YDET_CHECK_ARG(c1->kp == C && c2->kp == 9 * Cm && c1->cout_pad >= Cm && c2->cout_pad >= C, "bottleneck_fused: kp %d / %d",
                   c1->kp, c2->kp);
    YDET_CHECK_ARG(c1->act == c2->act && c1->slope == c2->slope && c1->act >= 0 && c1->act <= 2,
                   "bottleneck_fused: both stages share one activation");
    YDET_CHECK_ARG(c1->res == nullptr && c1->post_scale == nullptr && c2->post_scale == nullptr && c1->y2 == nullptr &&
                       c2->y2 == nullptr, "bottleneck_fused: no residual on the first stage, no second stage or destination");
    YDET_CHECK_ARG(c1->x_ld % 8 == 0 && c1->x_coff % 8 == 0 && c1->x_coff >= 0 && c1->x_coff + C <= c1->x_ld &&
                       (c3 || (c2->y_ld % 8 == 0 && c2->y_coff % 8 == 0 && c2->y_coff >= 0 && c2->y_coff + C <= c2->y_ld)),
                   "bottleneck_fused: channel slices must be 8-aligned and inside their rows");
    YDET_CHECK_ARG(((uintptr_t)c1->x & 15) == 0 && (c3 || ((uintptr_t)c2->y & 15) == 0) && ((uintptr_t)c1->w & 15) == 0 &&
                       ((uintptr_t)c2->w & 15) == 0, "bottleneck_fused: pointers must be 16-byte aligned");
    if (c3) {
        YDET_CHECK_ARG(C == 32 && c3->x && c3->w && c3->bias && c3->y, "c3_tail_fused: 32-channel bottleneck, non-null x/w/bias/y of cv3");
        YDET_CHECK_ARG(c3->KH == 1 && c3->KW == 1 && c3->stride == 1 && c3->pad == 0 && c3->Cin == 64 && c3->Cout == 64 &&
                           c3->kp == 64 && c3->cout_pad >= 64 && c3->N == c1->N && c3->H == c1->H && c3->W == c1->W,
                       "c3_tail_fused: cv3 must be a 1x1 convolution 64 -> 64 over the bottleneck's map");
        YDET_CHECK_ARG(c3->in_dtype == YDET_BF16 && c3->out_dtype == YDET_BF16 && c3->act == c1->act && c3->slope == c1->slope &&
                           c3->res == nullptr && c3->post_scale == nullptr && c3->y2 == nullptr && c3->x_up == nullptr,
                       "c3_tail_fused: cv3 shares the activation, bf16, no residual / second stage / destination / source");
        YDET_CHECK_ARG(c3->x_ld % 8 == 0 && c3->x_coff % 8 == 0 && c3->x_coff >= 0 && c3->x_coff + 32 <= c3->x_ld &&
                           c3->y_ld % 8 == 0 && c3->y_coff % 8 == 0 && c3->y_coff >= 0 && c3->y_coff + 64 <= c3->y_ld &&
                           ((uintptr_t)c3->x & 15) == 0 && ((uintptr_t)c3->y & 15) == 0 && ((uintptr_t)c3->w & 15) == 0,
                       "c3_tail_fused: cv3 slices (x: the 32 channels of the other half) must be 8-aligned, pointers 16-byte aligned");
        YDET_CHECK_ARG(c3->y != c1->x, "c3_tail_fused: the output must not be the bottleneck's input buffer");
    }
    const bool add = c2->res != nullptr;
    if (add)
        YDET_CHECK_ARG(c2->res == c1->x && c2->res_ld == c1->x_ld && c2->res_coff == c1->x_coff &&
                           (c2->res_mode == YDET_RES_AFTER_ACT || (c2->res_mode == YDET_RES_BEFORE_ACT && !c3)),
                       "bottleneck_fused: the residual is the block's own input, added last (or, without a cv3 tail, before the second activation)");
    // the halo of a tile is read while other workgroups write their outputs: in-place operation would race
    const bool same_buf = !c3 && c2->y == c1->x && c2->y_ld == c1->x_ld;
    YDET_CHECK_ARG(!same_buf || c2->y_coff + C <= c1->x_coff || c1->x_coff + C <= c2->y_coff,
                   "bottleneck_fused: output slice overlaps the input slice");
    BneckArgs a;
    a.x = (const char*)c1->x; a.y = (char*)c2->y;
    a.w1 = (const char*)c1->w; a.b1 = c1->bias; a.w2 = (const char*)c2->w; a.b2 = c2->bias;
    a.N = c1->N; a.H = c1->H; a.W = c1->W;
    a.x_ld = c1->x_ld; a.x_coff = c1->x_coff; a.y_ld = c2->y_ld; a.y_coff = c2->y_coff;
    a.kp1 = c1->kp; a.kp2 = c2->kp;
    a.act = c1->act; a.slope = c1->slope; a.add = add ? 1 : 0;
    a.res_before = add && c2->res_mode == YDET_RES_BEFORE_ACT ? 1 : 0;
    a.w3 = nullptr; a.b3 = nullptr; a.z = nullptr; a.y3 = nullptr; a.z_ld = a.z_coff = a.y3_ld = a.y3_coff = a.kp3 = 0;
    if (c3) {
        a.w3 = (const char*)c3->w; a.b3 = c3->bias; a.z = (const char*)c3->x; a.y3 = (char*)c3->y;
        a.z_ld = c3->x_ld; a.z_coff = c3->x_coff; a.y3_ld = c3->y_ld; a.y3_coff = c3->y_coff; a.kp3 = c3->kp;
    }
    const int tw = C == 32 ? TW : c64::TW, th = C == 32 ? TH : c64::TH;
    a.tiles_x = (a.W + tw - 1) / tw;
    const long long tiles_y = (a.H + th - 1) / th;
    const long long per_img = tiles_y * a.tiles_x, n_tiles = per_img * a.N;
    YDET_CHECK_ARG(n_tiles < (1ll << 30), "bottleneck_fused: %lld tiles", n_tiles);
    a.tiles_per_img = (int)per_img; a.n_tiles = (int)n_tiles;
    // Both kernels hold two workgroups per CU (registers): exactly that many workgroups walk over the tiles, so the
    // weights are fetched once per resident workgroup (measured best among 1x, 2x, 3x, 4x the CU count and one
    // workgroup per one to four tiles).
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0)
        cus = 256;
    const long long grid = n_tiles < 2LL * cus ? n_tiles : 2LL * cus;
    const bool silu = a.act == YDET_ACT_SILU;
    const dim3 g((unsigned)grid), b(256);
    if (C == 32 && c3) {
        if (silu) hipLaunchKernelGGL((bottleneck_c32_kernel<true, true>), g, b, 0, stream, a);
        else hipLaunchKernelGGL((bottleneck_c32_kernel<true, false>), g, b, 0, stream, a);
    } else if (C == 32) {
        if (silu) hipLaunchKernelGGL((bottleneck_c32_kernel<false, true>), g, b, 0, stream, a);
        else hipLaunchKernelGGL((bottleneck_c32_kernel<false, false>), g, b, 0, stream, a);
    } else if (Cm == 32) {
        const dim3 g3((unsigned)(n_tiles < (long long)RB64_WPE * cus ? n_tiles : (long long)RB64_WPE * cus));
        if (silu) hipLaunchKernelGGL(resblock_c64_kernel<true>, g3, b, 0, stream, a);
        else hipLaunchKernelGGL(resblock_c64_kernel<false>, g3, b, 0, stream, a);
    } else {
        if (silu) hipLaunchKernelGGL(bottleneck_c64_kernel<true>, g, b, 0, stream, a);
        else hipLaunchKernelGGL(bottleneck_c64_kernel<false>, g, b, 0, stream, a);
    }
    YDET_CHECK_LAUNCH("bottleneck kernel");
    return YDET_OK;
}

extern "C" int ydet_bottleneck_fused(const ydet_conv_desc* cv1, const ydet_conv_desc* cv2, void* stream) {
    ydet::clear_error();
    return ydet_launch_bottleneck(cv1, cv2, nullptr, ydet::as_stream(stream));
}

extern "C" int ydet_c3_tail_fused(const ydet_conv_desc* cv1, const ydet_conv_desc* cv2, const ydet_conv_desc* cv3, void* stream) {
    ydet::clear_error();
    if (!cv3) return ydet::fail(YDET_ERR_ARG, "c3_tail_fused: null cv3 descriptor");
    return ydet_launch_bottleneck(cv1, cv2, cv3, ydet::as_stream(stream));
}
