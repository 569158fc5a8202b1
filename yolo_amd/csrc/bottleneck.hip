// A 32-channel residual bottleneck as one kernel: y = x + act(conv3x3(act(conv1x1(x)))) (yolov5_official/modules.py:208-218
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
constexpr int PITCH = 80;                      // bytes per patch pixel in LDS: 64 of data + 16, so that the 16-byte reads
                                               // of 8 consecutive pixels cover all 32 banks exactly once
constexpr int PATCH_BYTES = NGRP * 16 * PITCH; // 28160
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
    int tiles_x, tiles_per_img, n_tiles;
};

__device__ __forceinline__ float act_fn(float v, bool silu, float slope) {
    if (silu) return v * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(v * -1.44269504088896340736f));
    return v > 0.f ? v : v * slope;
}

__global__ __launch_bounds__(256) void bottleneck_c32_kernel(const BneckArgs a) {
    __shared__ __attribute__((aligned(16))) char xs[PATCH_BYTES];   // input patch, bf16 [pixel][32] at PITCH
    __shared__ __attribute__((aligned(16))) char ts[PATCH_BYTES];   // act(conv1x1) of the patch, zero outside the image

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int px = lane & 15, q = lane >> 4;
    const bool silu = a.act == YDET_ACT_SILU;
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

    // the rows behind the last patch pixel (read by the last 1x1 group) stay zero
    for (int i = NPIX * PITCH + tid * 16; i < PATCH_BYTES; i += 256 * 16) *reinterpret_cast<uint4*>(xs + i) = uint4{0, 0, 0, 0};

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
                t[e] = (__bf16)(in ? act_fn(acc0[e] + b1v[0][e], silu, slope) : 0.f);
                t[4 + e] = (__bf16)(in ? act_fn(acc1[e] + b1v[1][e], silu, slope) : 0.f);
            }
            *reinterpret_cast<bf16x8*>(ts + p * PITCH + q * 16) = t;
        }
        __syncthreads();
        // ---- 3x3 over ts, + residual, -> y: a wave owns two output rows, i.e. four groups of 16 pixels ----------
#pragma unroll
        for (int gi = 0; gi < 4; ++gi) {
            const int row = wave * 2 + (gi >> 1), col0 = (gi & 1) * 16;
            f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                const int p = (row + t / 3) * PW + col0 + t % 3 + px;
                const bf16x8 tb = *reinterpret_cast<const bf16x8*>(ts + p * PITCH + q * 16);
                acc0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w2f[t][0], tb, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w2f[t][1], tb, acc1, 0, 0, 0);
            }
            const int oy = oy0 + row, ox = ox0 + col0 + px;
            bf16x8 res;
            if (a.add) res = *reinterpret_cast<const bf16x8*>(xs + ((row + 1) * PW + col0 + 1 + px) * PITCH + q * 16);
            bf16x8 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float v0 = act_fn(acc0[e] + b2v[0][e], silu, slope), v1 = act_fn(acc1[e] + b2v[1][e], silu, slope);
                if (a.add) { v0 += (float)res[e]; v1 += (float)res[4 + e]; }
                o[e] = (__bf16)v0;
                o[4 + e] = (__bf16)v1;
            }
            if (oy < a.H && ox < a.W)
                *reinterpret_cast<bf16x8*>(a.y + ((((size_t)n * a.H + oy) * a.W + ox) * a.y_ld + a.y_coff + q * 8) * 2) = o;
        }
        __syncthreads();   // xs / ts are rewritten by the next tile
    }
}

}  // namespace

int ydet_launch_bottleneck_c32(const ydet_conv_desc* c1, const ydet_conv_desc* c2, hipStream_t stream) {
    using namespace ydet;
    YDET_CHECK_ARG(c1 && c2, "bottleneck_c32: null descriptor");
    YDET_CHECK_ARG(c1->x && c1->w && c1->bias && c2->w && c2->bias && c2->y, "bottleneck_c32: null x/w/bias/y pointer");
    YDET_CHECK_ARG(c1->in_dtype == YDET_BF16 && c1->out_dtype == YDET_BF16 && c2->in_dtype == YDET_BF16 &&
                       c2->out_dtype == YDET_BF16, "bottleneck_c32: bf16 only");
    YDET_CHECK_ARG(c1->Cin == 32 && c1->Cout == 32 && c2->Cin == 32 && c2->Cout == 32, "bottleneck_c32: 32 channels throughout");
    YDET_CHECK_ARG(c1->KH == 1 && c1->KW == 1 && c1->stride == 1 && c1->pad == 0 && c2->KH == 3 && c2->KW == 3 &&
                       c2->stride == 1 && c2->pad == 1, "bottleneck_c32: a 1x1 followed by a 3x3 / stride 1 / pad 1");
    YDET_CHECK_ARG(c1->N > 0 && c1->H > 0 && c1->W > 0 && c2->N == c1->N && c2->H == c1->H && c2->W == c1->W &&
                       c2->Ho == c1->H && c2->Wo == c1->W, "bottleneck_c32: shapes of the two stages differ");
    YDET_CHECK_ARG(c1->kp == 32 && c2->kp == 288 && c1->cout_pad >= 32 && c2->cout_pad >= 32, "bottleneck_c32: kp %d / %d",
                   c1->kp, c2->kp);
    YDET_CHECK_ARG(c1->act == c2->act && c1->slope == c2->slope && c1->act >= 0 && c1->act <= 2,
                   "bottleneck_c32: both stages share one activation");
    YDET_CHECK_ARG(c1->res == nullptr && c1->post_scale == nullptr && c2->post_scale == nullptr,
                   "bottleneck_c32: no residual on the first stage, no second epilogue stage");
    YDET_CHECK_ARG(c1->x_ld % 8 == 0 && c1->x_coff % 8 == 0 && c1->x_coff >= 0 && c1->x_coff + 32 <= c1->x_ld &&
                       c2->y_ld % 8 == 0 && c2->y_coff % 8 == 0 && c2->y_coff >= 0 && c2->y_coff + 32 <= c2->y_ld,
                   "bottleneck_c32: channel slices must be 8-aligned and inside their rows");
    const bool add = c2->res != nullptr;
    if (add)
        YDET_CHECK_ARG(c2->res == c1->x && c2->res_ld == c1->x_ld && c2->res_coff == c1->x_coff &&
                           c2->res_mode == YDET_RES_AFTER_ACT, "bottleneck_c32: the residual is the block's own input, added last");
    // the halo of a tile is read while other workgroups write their outputs: in-place operation would race
    const bool same_buf = c2->y == c1->x && c2->y_ld == c1->x_ld;
    YDET_CHECK_ARG(!same_buf || c2->y_coff + 32 <= c1->x_coff || c1->x_coff + 32 <= c2->y_coff,
                   "bottleneck_c32: output slice overlaps the input slice");
    BneckArgs a;
    a.x = (const char*)c1->x; a.y = (char*)c2->y;
    a.w1 = (const char*)c1->w; a.b1 = c1->bias; a.w2 = (const char*)c2->w; a.b2 = c2->bias;
    a.N = c1->N; a.H = c1->H; a.W = c1->W;
    a.x_ld = c1->x_ld; a.x_coff = c1->x_coff; a.y_ld = c2->y_ld; a.y_coff = c2->y_coff;
    a.kp1 = c1->kp; a.kp2 = c2->kp;
    a.act = c1->act; a.slope = c1->slope; a.add = add ? 1 : 0;
    a.tiles_x = (a.W + TW - 1) / TW;
    const long long tiles_y = (a.H + TH - 1) / TH;
    const long long per_img = tiles_y * a.tiles_x, n_tiles = per_img * a.N;
    YDET_CHECK_ARG(n_tiles < (1ll << 30), "bottleneck_c32: %lld tiles", n_tiles);
    a.tiles_per_img = (int)per_img; a.n_tiles = (int)n_tiles;
    // a few tiles per workgroup amortise the weight fetch; at least two rounds of workgroups keep every CU busy
    long long grid = n_tiles;
    if (grid > 2048) grid = (n_tiles + 3) / 4 > 2048 ? (n_tiles + 3) / 4 : 2048;
    hipLaunchKernelGGL(bottleneck_c32_kernel, dim3((unsigned)grid), dim3(256), 0, stream, a);
    YDET_CHECK_LAUNCH("bottleneck_c32_kernel");
    return YDET_OK;
}

extern "C" int ydet_bottleneck_c32(const ydet_conv_desc* cv1, const ydet_conv_desc* cv2, void* stream) {
    ydet::clear_error();
    return ydet_launch_bottleneck_c32(cv1, cv2, ydet::as_stream(stream));
}
