// Streaming 1x1 convolution for gfx950: bf16 in, bf16 out, stride 1, 128 or 256 input channels, many pixels.
//
// Such a layer moves a pixel's input row in and its output row out and does a few hundred FLOP per byte of neither:
// it is bound by HBM, and what it needs is many independent loads in flight on every CU at all times -- not the
// load / compute / store phases, one barrier per K step, of the tiled kernel of conv.hip, whose workgroups of a
// launch all run the same phase at the same time.  Here:
//
//  * The whole weight tile (128 output channels x K) is laid down in LDS ONCE per workgroup, as NK images of one K
//    step each in the layout of conv.hip (64-byte rows, chunk index XOR-swizzled, rows permuted so that a lane ends
//    up with 16-byte pieces of its pixels' output rows).  After that the workgroup never synchronises again.
//  * Pixels never pass through LDS: the 16 bytes a lane needs of a pixel's row for a K step -- 8 consecutive
//    channels -- ARE its MFMA B fragment (lane = pixel l & 15, channels 8 (l >> 4) .. + 7 of the step), so a wave
//    loads the fragments of its 16 FM pixels straight from the NHWC map into registers.
//  * Every wave walks over its own tiles of 16 FM pixels (persistent, strided over all waves of the launch) with two
//    register buffers: the loads of tile t + 2 are issued as soon as the MFMAs of tile t have read its buffer, before
//    the epilogue of tile t, so a wave always has one to two tiles of loads in flight and its stores never stand in
//    front of the loads it waits for next.
//  * Same MFMA, same K order, same epilogue arithmetic as conv_dma_kernel: results are bit-identical to it.
//
// A nearest-neighbour 2x upsample folded into the input (x_up of ConvArgs) only changes which row a lane reads for the
// first up_C channels.
#include <stdlib.h>

#include "conv_common.h"

using namespace ydet;

namespace ydet {
namespace conv {
namespace {

constexpr int PW_BN = 128;                 // output channels per workgroup
constexpr int PW_WSTEP = PW_BN * 64;       // bytes of the weight image of one K step

struct PwArgs {
    int tiles_n;        // channel tiles
    int gm;             // workgroups per channel tile
    long long wtiles;   // wave tiles (16 FM pixels each)
};

__device__ __forceinline__ void glds16(const char* src, char* dst) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                     (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
}

// NK: K steps of 32 channels; FM: pixel fragments per wave tile; PW_NW: waves per workgroup; MINW: waves per SIMD the
// register allocation leaves room for.
template <int NK, int FM, int PW_NW, int MINW, bool UP>
__global__ __launch_bounds__(64 * PW_NW, MINW) void conv_pw_kernel(const ConvArgs a, const PwArgs g) {
    constexpr int TP = 16 * FM;            // pixels per wave tile
    constexpr int FN = PW_BN / 16;         // channel fragments: every wave holds all 128 channels of its pixels
    constexpr int NP = FN / 2;             // 16-byte output pieces per lane and pixel
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* cst = reinterpret_cast<float*>(smem + NK * PW_WSTEP);

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const unsigned bid = xcd_remap(blockIdx.x, gridDim.x);
    const int tile_n = (int)(bid % (unsigned)g.tiles_n);
    const int mb = (int)(bid / (unsigned)g.tiles_n);
    const int n0 = tile_n * PW_BN;

    // ---- weights and per-channel constants -> LDS, once ----------------------------------------------------------------
    {
        const int srow = lane >> 2;
        const int lchunk = (lane & 3) ^ chunk_swz(srow >> 2);
        for (int i = wave; i < PW_BN / 16; i += PW_NW) {     // groups of sixteen image rows
            // image row 16 i + r holds channel 32 (i >> 1) + 8 (r >> 2) + 4 (i & 1) + (r & 3)
            const int r = srow;
            const int ch = 32 * (i >> 1) + 8 * (r >> 2) + 4 * (i & 1) + (r & 3);
            const char* src = a.w + ((size_t)(n0 + ch) * a.kp + lchunk * 8) * 2;
#pragma unroll
            for (int s = 0; s < NK; ++s) glds16(src + s * 64, smem + s * PW_WSTEP + i * 1024);
        }
        if (tid < PW_BN) {
            cst[tid] = a.bias ? a.bias[n0 + tid] : 0.f;
            cst[PW_BN + tid] = a.post_scale ? a.post_scale[n0 + tid] : 0.f;
            cst[2 * PW_BN + tid] = a.post_scale ? a.post_shift[n0 + tid] : 0.f;
        }
        wait_vmcnt<0>();
        __syncthreads();
    }

    const int px = lane & 15, q = lane >> 4;
    const int fr_off = px * 64 + ((q ^ chunk_swz(px >> 2)) << 4);
    const long long wstride = (long long)g.gm * PW_NW;
    const int nk_up = UP ? a.up_C >> 5 : 0;
    const int HoWo = a.Ho * a.Wo;
    const float r_wo = 1.0f / (float)a.Wo, r_ho = 1.0f / (float)a.Ho;

    // the B fragments of one wave tile, straight from the map
    auto load_tile = [&](long long wt, bf16x8 (&b)[FM][NK]) {
        const long long p0 = wt * TP;
        int n_0 = 0, ho_0 = 0, wo_0 = 0;
        if constexpr (UP) {
            const unsigned mu = (unsigned)p0;
            n_0 = (int)(mu / (unsigned)HoWo);
            const unsigned rem = mu - (unsigned)n_0 * (unsigned)HoWo;
            ho_0 = (int)(rem / (unsigned)a.Wo);
            wo_0 = (int)(rem - (unsigned)ho_0 * (unsigned)a.Wo);
        }
#pragma unroll
        for (int j = 0; j < FM; ++j) {
            long long m = p0 + j * 16 + px;
            const bool okm = m < a.M;
            if (!okm) m = a.M - 1;         // (never stored)
            const char* p = a.x + ((size_t)m * a.x_ld + a.x_coff + q * 8) * 2;
            const char* pu = p;
            if constexpr (UP) {            // pixel (n, ho, wo) reads (n, ho / 2, wo / 2) of the half-size map
                const int cw = wo_0 + j * 16 + px, qw = div_small(cw, a.Wo, r_wo), wo = cw - qw * a.Wo;
                const int chh = ho_0 + qw, qh = div_small(chh, a.Ho, r_ho), ho = chh - qh * a.Ho;
                const int n = okm ? n_0 + qh : 0;
                const long long pix = ((long long)n * (a.Ho >> 1) + (ho >> 1)) * (a.Wo >> 1) + (wo >> 1);
                pu = a.x_up + (pix * a.up_ld + a.up_coff + q * 8) * 2;
            }
#pragma unroll
            for (int s = 0; s < NK; ++s) {
                const bf16x8* src = reinterpret_cast<const bf16x8*>((UP && s < nk_up ? pu : p) + s * 64);
#if defined(YDET_PW_NT) && YDET_PW_NT
                b[j][s] = __builtin_nontemporal_load(src);
#else
                b[j][s] = *src;
#endif
            }
        }
    };

    // one wave tile: K loop out of the resident weights and the buffer, refill of the buffer, epilogue
    auto process = [&](long long wt, bf16x8 (&b)[FM][NK]) {
        f32x4 acc[FN][FM];
#pragma unroll
        for (int i = 0; i < FN; ++i)
#pragma unroll
            for (int j = 0; j < FM; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < NK; ++s) {
            bf16x8 fa[FN];
#pragma unroll
            for (int i = 0; i < FN; ++i) fa[i] = *reinterpret_cast<const bf16x8*>(smem + s * PW_WSTEP + i * 1024 + fr_off);
#pragma unroll
            for (int i = 0; i < FN; ++i)
#pragma unroll
                for (int j = 0; j < FM; ++j) acc[i][j] = mma(fa[i], b[j][s], acc[i][j]);
            __builtin_amdgcn_sched_barrier(0);     // (the weight fragments of later steps are not fetched ahead: registers)
        }
        // the buffer is free: the tile after next goes into it, ahead of this tile's stores
        if (wt + 2 * wstride < g.wtiles) load_tile(wt + 2 * wstride, b);
        __builtin_amdgcn_sched_barrier(0);

        with_epilogue_mode(a, [&](auto mode) {
            constexpr int MODE = decltype(mode)::value;      // (never a residual here)
#pragma unroll
            for (int j = 0; j < FM; ++j) {
                const long long m = wt * TP + j * 16 + px;
                const bool okm = m < a.M;
#pragma unroll
                for (int p = 0; p < NP; ++p) {
                    const int cl = 32 * p + 8 * q;
                    bf16x8 o;
#pragma unroll
                    for (int hf = 0; hf < 2; ++hf) {
                        const int i = 2 * p + hf;
                        float v[4] = {acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]};
                        const float r4[4] = {0.f, 0.f, 0.f, 0.f};
                        finish4<MODE>(v, r4, cst + cl + 4 * hf, cst + PW_BN + cl + 4 * hf, cst + 2 * PW_BN + cl + 4 * hf, a);
#pragma unroll
                        for (int e = 0; e < 4; ++e) o[4 * hf + e] = (__bf16)v[e];
                    }
                    const int co = n0 + cl;
                    if (!okm || co >= a.cout_store) continue;
                    char* dst = co >= a.y_split ? a.y2 + ((size_t)m * a.y2_ld + a.y2_coff + (co - a.y_split)) * 2
                                                : a.y + ((size_t)m * a.y_ld + a.y_coff + co) * 2;
                    *reinterpret_cast<bf16x8*>(dst) = o;
                }
            }
        });
    };

    bf16x8 buf0[FM][NK], buf1[FM][NK];
    long long wt = (long long)mb * PW_NW + wave;
    if (wt < g.wtiles) load_tile(wt, buf0);
    if (wt + wstride < g.wtiles) load_tile(wt + wstride, buf1);
    while (wt < g.wtiles) {
        process(wt, buf0);
        wt += wstride;
        if (wt >= g.wtiles) break;
        process(wt, buf1);
        wt += wstride;
    }
}

bool shape_ok(const ConvArgs& a) {
    return a.KH == 1 && a.KW == 1 && a.stride == 1 && a.pad == 0 && a.groups <= 1 && a.dilation <= 1 && a.kp == a.Cin &&
           (a.Cin == 128 || a.Cin == 256) && a.Cout > 64 && a.res == nullptr && a.pw_w == nullptr && a.splits <= 1 &&
           a.y_ld % 8 == 0 && a.y_coff % 8 == 0 && a.cout_store % 8 == 0 && (!a.y2 || (a.y2_ld % 8 == 0 && a.y2_coff % 8 == 0)) &&
           a.up_C % 32 == 0 && a.M < (1LL << 31);
}

template <int NK, int FM, int PW_NW, int MINW>
int launch_pw(const ConvArgs& a, hipStream_t stream) {
    constexpr int LDS = NK * PW_WSTEP + 3 * PW_BN * (int)sizeof(float);
    PwArgs g;
    g.tiles_n = (a.Cout + PW_BN - 1) / PW_BN;
    g.wtiles = (a.M + 16 * FM - 1) / (16 * FM);
    // every wave of a channel tile gets the same number of wave tiles (+- 1)
    const DeviceInfo dev = current_device_info();
    long long slots = dev.lds_per_cu / LDS;
    if (slots > (MINW * 4) / PW_NW) slots = (MINW * 4) / PW_NW;
    if (slots < 1) slots = 1;
    long long gm = slots * dev.cus / g.tiles_n;
    const long long need = (g.wtiles + PW_NW - 1) / PW_NW;
    if (gm > need) gm = need;
    if (gm < 1) gm = 1;
    const long long rounds = (g.wtiles + gm * PW_NW - 1) / (gm * PW_NW);
    gm = (g.wtiles + rounds * PW_NW - 1) / (rounds * PW_NW);
    g.gm = (int)gm;
    const unsigned grid = (unsigned)(gm * g.tiles_n);
    if (a.up_C) {
        static unsigned char optin[kMaxDevices];
        const int rc = raise_dynamic_lds_limit(reinterpret_cast<const void*>(&conv_pw_kernel<NK, FM, PW_NW, MINW, true>), LDS, optin, "conv_pw");
        if (rc != YDET_OK) return rc;
        hipLaunchKernelGGL((conv_pw_kernel<NK, FM, PW_NW, MINW, true>), dim3(grid), dim3(64 * PW_NW), LDS, stream, a, g);
    } else {
        static unsigned char optin[kMaxDevices];
        const int rc = raise_dynamic_lds_limit(reinterpret_cast<const void*>(&conv_pw_kernel<NK, FM, PW_NW, MINW, false>), LDS, optin, "conv_pw");
        if (rc != YDET_OK) return rc;
        hipLaunchKernelGGL((conv_pw_kernel<NK, FM, PW_NW, MINW, false>), dim3(grid), dim3(64 * PW_NW), LDS, stream, a, g);
    }
    YDET_CHECK_LAUNCH("conv_pw_kernel");
    return YDET_OK;
}

}  // namespace

bool conv_pw_takes(const ConvArgs& a) {
    static const char* sw = getenv("YDET_PW");          // YDET_PW=0: every 1x1 convolution stays with the tiled kernel
    if (sw && sw[0] == '0') return false;
    // (a launch of few pixels is bound by latencies, not by HBM: the tiled kernel splits it over more workgroups)
    return shape_ok(a) && a.M >= 65536;
}

int launch_conv_pw(const ConvArgs& a, hipStream_t stream) {
    if (kernel_probe().on) {
        snprintf(kernel_probe().name, sizeof(kernel_probe().name), "conv_pw<k%d>", a.Cin);
        return YDET_OK;
    }
    // Measured on MI355X (YOLOv5s 640 at batch 64, 80x80 and 40x40 maps): with 128 input channels eight waves per
    // workgroup on 16-pixel tiles at four waves per SIMD (125 registers) beat four waves on 32-pixel tiles by 7 %;
    // with 256 the buffers of a 16-pixel tile alone are 64 registers and two waves per SIMD it is.
    if (a.Cin == 128) return launch_pw<4, 1, 8, 4>(a, stream);
    return launch_pw<8, 1, 4, 2>(a, stream);
}

}  // namespace conv
}  // namespace ydet
