// Implicit-GEMM convolution for gfx950 (MI355X): Conv2d + folded BatchNorm + LeakyReLU/SiLU
// (+ residual) on NHWC activations, bf16 or fp32 MFMA with fp32 accumulation.
//
// GEMM view:  D[co, m] = sum_k Wp[co, k] * X[m, k]
//   m  = flattened output pixel (n, ho, wo)             -> the MFMA column (lane & 15)
//   co = output channel                                  -> the MFMA row
//   k  = (kh*KW + kw)*Cin + c, gathered on the fly from the NHWC input (never materialised).
// Channels sit on the MFMA row so that every lane ends up with 4 consecutive output channels of
// one pixel per fragment: the epilogue stores 16-byte pieces straight into the NHWC output.
//
// Both operands are K-contiguous (packed weights [Cout][K], NHWC input), so one K-step of a tile is
// a set of 64-byte rows = 4 chunks of 16 B: 32 bf16 or 16 fp32 values.  The same staging code, LDS
// image and fragment reads serve both dtypes; only the MFMA differs
// (1 x v_mfma_f32_16x16x32_bf16 or 4 x v_mfma_f32_16x16x4_f32 per fragment pair and K-step).
//
// LDS image: row r of a tile at byte r*64, its logical chunk c stored at physical chunk
// c ^ F[(r>>2)&3], F = {0,2,3,1}: with that permutation the 16 lanes of every ds_read_b128 lane
// group (rows l&15, chunk l>>4) touch 16 distinct 16-byte bank slots, so fragment reads are
// conflict-free on 64-byte rows.
//
// Two kernels share that image:
//   conv_dma_kernel     Cin a multiple of the K-step (every layer but the first): a K-step never straddles a
//                       filter tap, so the tap is workgroup-uniform (scalar) and each staged row is one
//                       precomputed pointer + a per-tap validity bit.  Tiles go global -> LDS directly
//                       (LDS-DMA, global_load_lds_dwordx4; padded taps read a zero page) through a 3-stage
//                       ring with two tiles in flight behind a counted vmcnt and one raw barrier per K-step.
//                       Weight rows are staged in an order that leaves every lane with whole 16-byte pieces of
//                       its pixels' output rows, which it stores straight from registers.
//   conv_generic_kernel any Cin that is a multiple of the chunk: per-thread tap tracking, global ->
//                       registers -> LDS, two stages.
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include <type_traits>

#include "conv_common.h"

using namespace ydet;
using namespace ydet::conv;

namespace {

// Detect decode of one pyramid level, done in the epilogue of its 1x1 prediction convolution (the tile holds all
// na*no channels of its pixels).
struct HeadArgs {
    float* z;          // [N][rows_per_img][no]
    float* row_max;    // [N][rows_per_img] or null
    int na, no, hw, nx, ny, transposed, level_off;
    long long rows_per_img;
    float stride, rd;  // rd = 1/nx (1/ny when transposed)
    float anchor[16];
};

// BN x BM output tile (channels x pixels) per 256-thread workgroup, WARPS_N x WARPS_M waves.
template <typename TIN, typename TOUT, int BN, int BM, int WARPS_N, int WARPS_M, bool IS1X1>
__global__ __launch_bounds__(256) void conv_generic_kernel(const ConvArgs a) {
    static_assert(WARPS_N * WARPS_M == 4, "four waves per workgroup");
    constexpr int EPC = 16 / (int)sizeof(TIN);  // elements per 16-byte chunk
    constexpr int BK = 4 * EPC;                 // K elements per step (64 bytes)
    constexpr int WN = BN / WARPS_N, WM = BM / WARPS_M;
    constexpr int FN = WN / 16, FM = WM / 16;
    constexpr int WL = (BN + 63) / 64, XL = BM / 64;
    constexpr int STAGE = (BN + BM) * 64;
    using frag_t = typename Frag<TIN>::type;

    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wn = wave % WARPS_N, wm = wave / WARPS_N;

    const unsigned bid = xcd_remap(blockIdx.x, gridDim.x);
    const int tile_n = bid % a.tiles_n;
    const long long tile_m = bid / a.tiles_n;
    const int n0 = tile_n * BN;
    const long long m0 = tile_m * BM;

    // ---- staging roles: thread -> (row tid>>2 (+64 i), chunk tid&3) -------------------------------
    const int ld_row = tid >> 2, ld_ch = tid & 3;
    const int st_off = ld_row * 64 + ((ld_ch ^ chunk_swz(tid >> 4)) << 4);

    // weight rows: always in bounds (packed weights are padded to cout_pad x kp)
    const char* wptr[WL];
#pragma unroll
    for (int i = 0; i < WL; ++i)
        wptr[i] = a.w + ((size_t)(n0 + ld_row + 64 * i) * a.kp + ld_ch * EPC) * sizeof(TIN);

    // pixel rows
    bool xok[XL];
    int hi0[XL], wi0[XL];
    const char* xbase[XL];
    const int HoWo = a.Ho * a.Wo;
#pragma unroll
    for (int i = 0; i < XL; ++i) {
        const long long m = m0 + ld_row + 64 * i;
        xok[i] = m < a.M;
        const long long mm = xok[i] ? m : 0;
        const int n = (int)(mm / HoWo);
        const int rem = (int)(mm - (long long)n * HoWo);
        const int ho = rem / a.Wo, wo = rem - ho * a.Wo;
        hi0[i] = ho * a.stride - a.pad;
        wi0[i] = wo * a.stride - a.pad;
        xbase[i] = a.x + ((size_t)n * a.H * a.W * a.x_ld + a.x_coff) * sizeof(TIN);
    }

    // running position of this thread's chunk inside the K axis: tap (kh,kw) and channel c
    int kc = ld_ch * EPC, kh = 0, kw = 0;
    if (!IS1X1) {
        while (kc >= a.Cin) {
            kc -= a.Cin;
            if (++kw == a.KW) { kw = 0; ++kh; }
        }
    }

    u32x4 wreg[WL], xreg[XL];
    const u32x4 zero4 = {0u, 0u, 0u, 0u};

    auto load_tile = [&]() {
#pragma unroll
        for (int i = 0; i < WL; ++i) {
            if (BN >= 64 || ld_row < BN) wreg[i] = *reinterpret_cast<const u32x4*>(wptr[i]);
            wptr[i] += BK * sizeof(TIN);
        }
#pragma unroll
        for (int i = 0; i < XL; ++i) {
            bool ok;
            size_t off;
            if (IS1X1) {
                ok = xok[i] && kc < a.Cin;
                off = ((size_t)(hi0[i] * a.W + wi0[i]) * a.x_ld + kc) * sizeof(TIN);
            } else {
                const int hi = hi0[i] + kh, wi = wi0[i] + kw;
                ok = xok[i] && kh < a.KH && (unsigned)hi < (unsigned)a.H && (unsigned)wi < (unsigned)a.W;
                off = ((size_t)(hi * a.W + wi) * a.x_ld + kc) * sizeof(TIN);
            }
            xreg[i] = ok ? *reinterpret_cast<const u32x4*>(xbase[i] + off) : zero4;
        }
        // advance to the next K-step
        kc += BK;
        if (!IS1X1) {
            while (kc >= a.Cin) {
                kc -= a.Cin;
                if (++kw == a.KW) { kw = 0; ++kh; }
            }
        }
    };
    auto store_tile = [&](int stage) {
        char* ws = smem + stage * STAGE;
        char* xs = ws + BN * 64;
#pragma unroll
        for (int i = 0; i < WL; ++i)
            if (BN >= 64 || ld_row < BN) *reinterpret_cast<u32x4*>(ws + st_off + i * 64 * 64) = wreg[i];
#pragma unroll
        for (int i = 0; i < XL; ++i) *reinterpret_cast<u32x4*>(xs + st_off + i * 64 * 64) = xreg[i];
    };

    f32x4 acc[FN][FM];
#pragma unroll
    for (int i = 0; i < FN; ++i)
#pragma unroll
        for (int j = 0; j < FM; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // fragment read offset of this lane inside a 16-row group
    const int fr_off = (lane & 15) * 64 + (((lane >> 4) ^ chunk_swz(lane >> 2)) << 4);

    auto compute = [&](int stage) {
        const char* ws = smem + stage * STAGE + (wn * WN) * 64 + fr_off;
        const char* xs = smem + stage * STAGE + BN * 64 + (wm * WM) * 64 + fr_off;
        frag_t fa[FN], fb[FM];
#pragma unroll
        for (int i = 0; i < FN; ++i) fa[i] = *reinterpret_cast<const frag_t*>(ws + i * 16 * 64);
#pragma unroll
        for (int j = 0; j < FM; ++j) fb[j] = *reinterpret_cast<const frag_t*>(xs + j * 16 * 64);
#pragma unroll
        for (int i = 0; i < FN; ++i)
#pragma unroll
            for (int j = 0; j < FM; ++j) acc[i][j] = mma(fa[i], fb[j], acc[i][j]);
    };

    const int nk = a.kp / BK;
    load_tile();
    store_tile(0);
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        const bool more = kt + 1 < nk;
        if (more) load_tile();
        compute(kt & 1);
        if (more) store_tile((kt + 1) & 1);
        __syncthreads();
    }

    // ---- epilogue: bias, activation, residual, store -----------------------------------------------
    const int q = lane >> 4, px = lane & 15;
    with_epilogue_mode(a, [&](auto mode) {
        constexpr int MODE = decltype(mode)::value;
#pragma unroll
        for (int j = 0; j < FM; ++j) {
            const long long m = m0 + wm * WM + j * 16 + px;
            if (m >= a.M) continue;
            char* yrow = a.y + ((size_t)m * a.y_ld + a.y_coff) * sizeof(TOUT);
            const char* rrow = a.res ? a.res + ((size_t)m * a.res_ld + a.res_coff) * sizeof(TIN) : nullptr;
#pragma unroll
            for (int i = 0; i < FN; ++i) {
                const int co = n0 + wn * WN + i * 16 + q * 4;
                if (co >= a.Cout) continue;
                float v[4] = {acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]};
                float r[4] = {0.f, 0.f, 0.f, 0.f};
                const float* b4 = a.bias ? a.bias + co : nullptr;
                if (a.vec_store) {
                    if (rrow) load4<TIN>(rrow + (size_t)co * sizeof(TIN), r);
                    finish4<MODE>(v, r, b4, a.post_scale + co, a.post_shift + co, a);
                    store4<TOUT>(yrow + (size_t)co * sizeof(TOUT), v);
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (rrow && co + e < a.Cout) r[e] = to_f32(reinterpret_cast<const TIN*>(rrow)[co + e]);
                    finish4<-1>(v, r, b4, a.post_scale + co, a.post_shift + co, a);
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (co + e < a.Cout) reinterpret_cast<TOUT*>(yrow)[co + e] = from_f32<TOUT>(v[e]);
                }
            }
        }
    });
}


// Debugging aid (tools/conv_trace.hip builds this file with -DYDET_TRACE): cycle stamps of wave 0 of the first
// 64 workgroups at up to 32 points of conv_dma_kernel.
#ifdef YDET_TRACE
__device__ unsigned long long g_trace[64 * 32];
#define YDET_STAMP(slot)                                                                      \
    do {                                                                                      \
        if (blockIdx.x < 64 && threadIdx.x == 0 && (slot) < 32)                               \
            g_trace[blockIdx.x * 32 + (slot)] = __builtin_readcyclecounter();                 \
    } while (0)
#else
#define YDET_STAMP(slot) do { } while (0)
#endif

// Waves per SIMD the register allocation must leave room for: the eight-wave tile runs two workgroups per CU (its
// LDS ring allows exactly two), the 128 x 128 four-wave tile three.
constexpr int conv_dma_min_waves(int bn, int bm, int nw, bool head, bool pipe = false) {
    if (pipe) return 2;      // (a wave of the pipelined form holds a 128 x 64 accumulator and two sets of operand fragments)
    return head ? (nw == 8 ? 4 : 1) : nw == 8 ? 4 : (nw == 4 && bn == 128 && bm == 128) ? 3 : 1;
}

// KCH: 16-byte chunks per LDS row = K extent of a ring stage (4: 64-byte rows, 32 bf16 of K per step; 8: 128-byte rows,
// two MFMA K-substeps per step, every LDS-DMA instruction fetches whole 128-byte lines).  NST: stages of the ring.
// PIPE: the K loop is software-pipelined over MFMA sub-steps (see below).
template <typename TIN, typename TOUT, int BN, int BM, int WARPS_N, int WARPS_M, bool HEAD = false, int KCH = 4, int NST = 3, bool PIPE = false>
__global__ __launch_bounds__(64 * WARPS_N * WARPS_M, conv_dma_min_waves(BN, BM, WARPS_N * WARPS_M, HEAD, PIPE))
void conv_dma_kernel(const ConvArgs a, const HeadArgs h) {
    constexpr int NW = WARPS_N * WARPS_M;         // waves per workgroup (4 or 8)
    constexpr int NT = 64 * NW;
    static_assert(NW == 4 || NW == 8 || NW == 16, "four, eight or sixteen waves per workgroup");
    constexpr int EPC = 16 / (int)sizeof(TIN);
    static_assert(KCH == 4 || (KCH == 8 && sizeof(TIN) == 2 && !HEAD), "128-byte rows: bf16 inputs, no decoding head");
    static_assert(NST >= 2 && NST <= 4, "two to four ring stages");
    constexpr int BK = KCH * EPC;
    constexpr int RB = KCH * 16;                  // bytes per LDS row
    constexpr int RPI = 64 / KCH;                 // rows one LDS-DMA instruction (64 lanes x 16 bytes) fills
    constexpr int KSUB = KCH / 4;                 // MFMA K-substeps per ring stage
    constexpr int WN = BN / WARPS_N, WM = BM / WARPS_M;
    constexpr int FN = WN / 16, FM = WM / 16;
    constexpr int WROWS = BN < RPI * NW ? RPI * NW : BN;   // weight rows staged (so every wave stages whole row groups)
    constexpr int WJ = WROWS / (RPI * NW), XJ = BM / (RPI * NW);  // row groups per wave: weights, pixels
    static_assert(WJ >= 1 && XJ >= 1 && WROWS % (RPI * NW) == 0 && BM % (RPI * NW) == 0, "tile too small for the wave count");
    constexpr int LPT = WJ + XJ;                  // LDS-DMA instructions per wave and tile
    constexpr int STAGE = (WROWS + BM) * RB;
    constexpr int NSTAGE = NST;
    using frag_t = typename Frag<TIN>::type;

    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wn = wave % WARPS_N, wm = wave / WARPS_N;
    YDET_STAMP(0);

    unsigned bid = xcd_remap(blockIdx.x, gridDim.x);
    int split = 0;
    if (!HEAD && a.splits > 1) {          // (consecutive ids = the tiles of one split: they share its slab of the weights)
        split = (int)(bid / (unsigned)a.tiles_per_split);
        bid -= (unsigned)split * (unsigned)a.tiles_per_split;
    }
    const int tile_n = bid % a.tiles_n;
    const long long tile_m = bid / a.tiles_n;
    const int n0 = tile_n * BN;
    const long long m0 = tile_m * BM;

    // per-channel epilogue constants of this tile: fetched into registers now (nothing waits for them here -- a
    // dependent LDS store would put a cold global load's latency in front of the K loop) and parked behind the ring
    // right after the K loop, where the epilogue reads them
    YDET_STAMP(22);
    constexpr int HEAD_RP = BN + 4;   // floats per pixel row of the fp32 tile a decoding head stages in LDS
    // (a decoding head lays down the pixels of one row of waves at a time: WM of them)
    constexpr int CST_OFF = HEAD && WM * HEAD_RP * 4 > NSTAGE * STAGE ? WM * HEAD_RP * 4 : NSTAGE * STAGE;
    float* cst = reinterpret_cast<float*>(smem + CST_OFF);
    float cst_b = 0.f, cst_ps = 0.f, cst_pb = 0.f;
    if (tid < BN) {
        if (a.bias) cst_b = a.bias[n0 + tid];
        if (a.post_scale) {
            cst_ps = a.post_scale[n0 + tid];
            cst_pb = a.post_shift[n0 + tid];
        }
    }

    // ---- staging roles ------------------------------------------------------------------------------
    // wave-instruction j of a wave fills LDS rows (j*NW + wave)*RPI .. +RPI-1 of the weight (j < WJ) or pixel
    // image; lane l lands at physical chunk l % KCH of row l / KCH and therefore fetches logical chunk
    // (l % KCH) ^ swizzle(row & 15) (the swizzle lives on the source address, the LDS image stays lane-linear).
    // 64-byte rows: swizzle F[(row >> 2) & 3]; 128-byte rows: (row >> 1) & 7 -- either way the sixteen rows a
    // ds_read_b128 lane group reads at one logical chunk fall on sixteen different 16-byte bank groups.
    auto row_swz = [](int row15) { return KCH == 4 ? chunk_swz(row15 >> 2) : (row15 >> 1) & 7; };
    const int srow = lane / KCH;
    const int lchunk = (lane % KCH) ^ row_swz((wave * RPI + srow) & 15);
    // Which output channel sits on which row of the weight image is free.  With bf16 outputs row (i, r) of a wave's WN
    // rows (fragment i, MFMA row r = 4 q + e) takes channel 32 (i>>1) + 8 q + 4 (i&1) + e: the accumulator rows a lane
    // ends up with in fragments 2 s and 2 s + 1 are then EIGHT CONSECUTIVE channels (one 16-byte piece), and the four
    // lane groups q of a pixel hold 64 consecutive bytes of it, so the epilogue stores straight from registers.  fp32
    // outputs keep the natural order (a fragment's four rows are a 16-byte piece already).
    constexpr bool PERM = sizeof(TOUT) == 2;
    const char* wsrc[WJ];
#pragma unroll
    for (int j = 0; j < WJ; ++j) {
        const int R = (j * NW + wave) * RPI + srow;
        int ch = R;
        if (PERM && R < BN) {
            const int i = (R % WN) >> 4, r = R & 15;
            ch = (R / WN) * WN + 32 * (i >> 1) + 8 * (r >> 2) + 4 * (i & 1) + (r & 3);
        }
        wsrc[j] = a.w + ((size_t)(n0 + ch) * a.kp + lchunk * EPC) * sizeof(TIN);
    }

    YDET_STAMP(23);
    const bool pointwise = a.KH == 1 && a.KW == 1 && a.stride == 1 && a.pad == 0;
    const char* xsrc[XJ];
    const char* xup[XJ];
    unsigned xmask[XJ];
    const int HoWo = a.Ho * a.Wo;
    // (image, row, column) of the tile's first pixel: one wave-uniform pair of divisions (M < 2^31, checked on the
    // host); a lane's pixels are at most BM further on, which float reciprocals resolve exactly
    int n_0 = 0, ho_0 = 0, wo_0 = 0;
    float r_wo = 0.f, r_ho = 0.f;
    if (!pointwise || a.up_C) {
        const unsigned mu = (unsigned)m0;
        n_0 = (int)(mu / (unsigned)HoWo);
        const unsigned rem = mu - (unsigned)n_0 * (unsigned)HoWo;
        ho_0 = (int)(rem / (unsigned)a.Wo);
        wo_0 = (int)(rem - (unsigned)ho_0 * (unsigned)a.Wo);
        r_wo = 1.0f / (float)a.Wo;
        r_ho = 1.0f / (float)a.Ho;
    }
#pragma unroll
    for (int j = 0; j < XJ; ++j) {
        const int row = (j * NW + wave) * RPI + srow;
        const long long m = m0 + row;
        const bool okm = m < a.M;
        if (pointwise) {
            xsrc[j] = a.x + ((size_t)(okm ? m : 0) * a.x_ld + a.x_coff + lchunk * EPC) * sizeof(TIN);
            xmask[j] = okm ? 1u : 0u;
            xup[j] = xsrc[j];
            if (a.up_C) {      // pixel (n, ho, wo) reads (n, ho/2, wo/2) of the half-size map for its first up_C channels
                const int cw = wo_0 + row, qw = div_small(cw, a.Wo, r_wo), wo = cw - qw * a.Wo;
                const int ch = ho_0 + qw, qh = div_small(ch, a.Ho, r_ho), ho = ch - qh * a.Ho;
                const int n = okm ? n_0 + qh : 0;
                const long long pix = ((long long)n * (a.Ho >> 1) + (ho >> 1)) * (a.Wo >> 1) + (wo >> 1);
                xup[j] = a.x_up + (pix * a.up_ld + a.up_coff + lchunk * EPC) * (long long)sizeof(TIN);
            }
        } else {
            // no division in the tap walk either: validity of each tap is one bit of a mask
            const int cw = wo_0 + row, qw = div_small(cw, a.Wo, r_wo), wo = cw - qw * a.Wo;
            const int ch = ho_0 + qw, qh = div_small(ch, a.Ho, r_ho), ho = ch - qh * a.Ho;
            const int n = n_0 + qh;
            const int hi0 = ho * a.stride - a.pad, wi0 = wo * a.stride - a.pad;
            unsigned mk = 0;
            if (a.KH == 3 && a.KW == 3) {      // the common case, unrolled
                unsigned colm = 0;
#pragma unroll
                for (int kw = 0; kw < 3; ++kw) colm |= (unsigned)(wi0 + kw) < (unsigned)a.W ? 1u << kw : 0u;
#pragma unroll
                for (int kh = 0; kh < 3; ++kh) mk |= (unsigned)(hi0 + kh) < (unsigned)a.H ? colm << (3 * kh) : 0u;
            } else {
                unsigned bit = 1u;
                for (int kh = 0; kh < a.KH; ++kh) {
                    const bool rowok = (unsigned)(hi0 + kh) < (unsigned)a.H;
                    for (int kw = 0; kw < a.KW; ++kw, bit <<= 1)
                        if (rowok && (unsigned)(wi0 + kw) < (unsigned)a.W) mk |= bit;
                }
            }
            xmask[j] = okm ? mk : 0u;
            const long long pix = ((long long)n * a.H + hi0) * a.W + wi0;  // may be "before" the image: masked
            xsrc[j] = a.x + (pix * a.x_ld + a.x_coff + lchunk * EPC) * (long long)sizeof(TIN);
            xup[j] = xsrc[j];
        }
    }
    const char* zsrc = reinterpret_cast<const char*>(g_zero_chunk);
    YDET_STAMP(24);

    // workgroup-uniform walk over K: tap index and channel offset of the current K-step
    int tap = 0, c0 = 0, kh = 0, kw = 0;
    long long tapoff = 0;  // byte offset of (kh, kw, c0) relative to a row pointer
    int ktile = 0;         // next tile to issue
    int nk = a.kp / BK;    // K-steps of this workgroup
    long long wbase = 0;   // byte offset of its first K value in a weight row
    if (!HEAD && a.splits > 1) {
        if (a.split_taps > 0) {
            tap = split * a.split_taps;
            const int ntap = min(a.split_taps, a.KH * a.KW - tap);
            kh = tap / a.KW;
            kw = tap - kh * a.KW;
            nk = ntap * a.Cin / BK;
            wbase = (long long)tap * a.Cin * (long long)sizeof(TIN);
        } else {
            c0 = split * a.split_cin;
            nk = a.split_cin / BK;
            wbase = (long long)c0 * (long long)sizeof(TIN);
        }
        tapoff = ((long long)(kh * a.W + kw) * a.x_ld + c0) * (long long)sizeof(TIN);
    }

    // Pipelined form: both operands are addressed through buffer descriptors with 32-bit offsets (a third of the address
    // registers of the pointer form), and a padding tap is an offset beyond the descriptor's extent, which the hardware
    // answers with zeros.  A weight row group j of a wave differs from group 0 by a wave-uniform number of rows (see the
    // permutation above), so one lane offset serves all groups and the rest rides in the scalar offset.
    unsigned xoff[PIPE ? XJ : 1] = {};
    unsigned woff_lane = 0, wuni[PIPE ? WJ : 1] = {};
    const char* xb = a.x + (size_t)a.x_coff * sizeof(TIN);
    const __amdgpu_buffer_rsrc_t rsrc_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(xb), 0, (int)(unsigned)a.x_span, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrc_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(a.w), 0, (int)(unsigned)a.w_span, 0x00020000);
    if constexpr (PIPE) {
#pragma unroll
        for (int j = 0; j < XJ; ++j) xoff[j] = (unsigned)(unsigned long long)(xsrc[j] - xb);
        const int r = RPI == 16 ? srow : (wave & 1) * 8 + srow;                  // row inside its group of sixteen
        const int lrow = PERM ? 8 * (r >> 2) + (r & 3) : r;
        woff_lane = (unsigned)(((n0 + lrow) * a.kp + lchunk * EPC) * (int)sizeof(TIN));
#pragma unroll
        for (int j = 0; j < WJ; ++j) {
            const int g = RPI == 16 ? j * NW + wave : (j * NW + wave) >> 1;      // group of sixteen rows
            const int urow = PERM ? (g / FN) * WN + 32 * ((g % FN) >> 1) + 4 * ((g % FN) & 1) : 16 * g;
            wuni[j] = (unsigned)(urow * a.kp * (int)sizeof(TIN));
        }
    }
    auto issue = [&](int stage) {
        char* dst = smem + stage * STAGE + wave * 1024;
        if constexpr (PIPE) {
            const unsigned ws = (unsigned)wbase + (unsigned)ktile * (unsigned)(BK * sizeof(TIN));
#pragma unroll
            for (int j = 0; j < WJ; ++j)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_w, (__attribute__((address_space(3))) void*)(dst + j * (NW * 1024)), 16,
                                                         (int)woff_lane, (int)(ws + wuni[j]), 0, 0);
            const unsigned toff = (unsigned)tapoff;
#pragma unroll
            for (int j = 0; j < XJ; ++j) {
                const bool ok = (xmask[j] >> tap) & 1u;
                const unsigned voff = ok ? xoff[j] + toff : 0xFFFFFFF0u;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_x, (__attribute__((address_space(3))) void*)(dst + WROWS * RB + j * (NW * 1024)),
                                                         16, (int)voff, 0, 0, 0);
            }
            // (the walk over taps without branches: the issue sits in the middle of a block of MFMAs)
            ++ktile;
            c0 += BK;
            const int wrap = c0 == a.Cin ? 1 : 0;
            c0 = wrap ? 0 : c0;
            tap += wrap;
            kw += wrap;
            const int wrap2 = kw == a.KW ? 1 : 0;
            kw = wrap2 ? 0 : kw;
            kh += wrap2;
            tapoff = (long long)(unsigned)(((kh * a.W + kw) * a.x_ld + c0) * (int)sizeof(TIN));
            return;
        }
        const long long woff = wbase + (long long)ktile * BK * (long long)sizeof(TIN);
#pragma unroll
        for (int j = 0; j < WJ; ++j)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(wsrc[j] + woff),
                                             (__attribute__((address_space(3))) void*)(dst + j * (NW * 1024)), 16, 0, 0);
        const bool from_up = c0 < a.up_C;      // workgroup-uniform (pointwise convs only: tap stays 0)
#pragma unroll
        for (int j = 0; j < XJ; ++j) {
            const bool ok = (xmask[j] >> tap) & 1u;
            const char* src = ok ? (from_up ? xup[j] : xsrc[j]) + tapoff : zsrc;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                             (__attribute__((address_space(3))) void*)(dst + WROWS * RB + j * (NW * 1024)), 16,
                                             0, 0);
        }
        ++ktile;
        c0 += BK;
        if (c0 == a.Cin) {
            c0 = 0;
            ++tap;
            if (++kw == a.KW) { kw = 0; ++kh; }
        }
        tapoff = ((long long)(kh * a.W + kw) * a.x_ld + c0) * (long long)sizeof(TIN);
    };

    f32x4 acc[FN][FM];
#pragma unroll
    for (int i = 0; i < FN; ++i)
#pragma unroll
        for (int j = 0; j < FM; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    int fr_off[KSUB];
#pragma unroll
    for (int kk = 0; kk < KSUB; ++kk) fr_off[kk] = (lane & 15) * RB + (((4 * kk + (lane >> 4)) ^ row_swz(lane & 15)) << 4);

    // weights of a 1x1 follower (see below): requested now, so that their latency runs under the K loop
    constexpr bool PW_OK = WARPS_N == 1 && BN == 64 && !HEAD && sizeof(TIN) == 2;
    // (rows in the same permuted order as the first convolution's, so the finished tile again holds 16-byte pieces per
    // lane; the channels a lane holds for K step s are 32 s + 8 q .. + 7, the natural k order of a fragment)
    bf16x8 pw_a[PW_OK ? 4 : 1][2];
    if constexpr (PW_OK) {
        if (a.pw_w) {
#pragma unroll
            for (int i2 = 0; i2 < 4; ++i2) {
                const int r = lane & 15;
                const char* wrow = a.pw_w + (size_t)(32 * (i2 >> 1) + 8 * (r >> 2) + 4 * (i2 & 1) + (r & 3)) * a.pw_kp * 2;
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2)
                    pw_a[i2][s2] = *reinterpret_cast<const bf16x8*>(wrow + (32 * s2 + 8 * (lane >> 4)) * 2);
            }
        }
    }

    YDET_STAMP(1);
    if constexpr (PIPE) {
        // Software-pipelined K loop.  The operand fragments of MFMA sub-step u + 1 are fetched from LDS while the MFMAs of
        // sub-step u run out of registers (two fragment sets, used alternately), so the matrix pipe never waits for an LDS
        // round trip behind a barrier.  The registers act as one more ring stage: once every wave holds the last fragments
        // of tile kt, that tile's LDS stage is refilled with tile kt + NST.  One barrier per tile, in front of its last
        // sub-step: by then every wave has (1) waited for its own LDS-DMA pieces of tile kt + 1 and (2) received the last
        // fragments of tile kt.  Sums are accumulated in the same order as in the plain loop: results are bit-identical.
        // (Built for 128-byte rows -- two sub-steps per tile -- and a two-stage ring.  The 64-byte-row form with four
        // waves per 128 x 256 tile was measured level with the plain eight-wave tile on layers of 128 output channels:
        // those are bound by the operand traffic of their narrow tile, not by the loop.)
        static_assert(!HEAD && NST == 2 && KSUB == 2, "pipelined loop: 128-byte rows, two stages");
        frag_t fa[2][FN], fb[2][FM];
        auto read_frags = [&](int stage, int kk, frag_t (&A)[FN], frag_t (&B)[FM]) {
            const char* ws = smem + stage * STAGE + (wn * WN) * RB + fr_off[kk];
            const char* xs = smem + stage * STAGE + WROWS * RB + (wm * WM) * RB + fr_off[kk];
#pragma unroll
            for (int i = 0; i < FN; ++i) A[i] = *reinterpret_cast<const frag_t*>(ws + i * 16 * RB);
#pragma unroll
            for (int j = 0; j < FM; ++j) B[j] = *reinterpret_cast<const frag_t*>(xs + j * 16 * RB);
        };
#pragma unroll
        for (int p = 0; p < NST; ++p)
            if (p < nk) issue(p);
        if (nk >= NST) wait_vmcnt<(NST - 1) * LPT>();
        else wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        read_frags(0, 0, fa[0], fb[0]);
        int st_cur = 0;
        // one sub-step: P = which fragment set holds it; MORE = a sub-step follows (known at compile time, so that the
        // fragment reads and the MFMAs sit in ONE basic block and can be interleaved: the last tile is peeled off the loop)
        // ISSUE: the block starts with the refill of the stage just vacated (tile kt + NST); the tiles that no longer
        // refill are peeled off the loop as well, so the refill needs no branch.
        auto substep = [&](auto par, auto more, auto refill, int kt, int kk) {
            constexpr int P = decltype(par)::value;
            constexpr bool MORE = decltype(more)::value;
            constexpr bool ISSUE = decltype(refill)::value;
            const bool last_sub = kk == KSUB - 1;
            const int st_next = st_cur + 1 == NST ? 0 : st_cur + 1;
            if (last_sub) {
                wait_vmcnt<0>();
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
                asm volatile("" ::: "memory");
                if constexpr (ISSUE) issue(st_cur);
            }
            if constexpr (MORE) read_frags(last_sub ? st_next : st_cur, last_sub ? 0 : kk + 1, fa[P ^ 1], fb[P ^ 1]);
#pragma unroll
            for (int i = 0; i < FN; ++i)
#pragma unroll
                for (int j = 0; j < FM; ++j) acc[i][j] = mma(fa[P][i], fb[P][j], acc[i][j]);
            // order inside the block: the first MFMA (its operands arrived a sub-step ago) leads, then one fragment read per
            // two MFMAs -- no read is waited for before the next sub-step.  (The LDS-DMA instructions of a refill stay in
            // front of the block: spread between the MFMAs they measured 1 % slower.)
            if constexpr (MORE) {
                if constexpr (ISSUE) __builtin_amdgcn_sched_group_barrier(0x020, LPT, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
#pragma unroll
                for (int r = 0; r < FN + FM; ++r) {
                    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
                }
            }
            if (last_sub) st_cur = st_next;
        };
        constexpr std::integral_constant<int, 0> P0{};
        constexpr std::integral_constant<int, 1> P1{};
        constexpr std::true_type Y{};
        constexpr std::false_type N{};
        {
            int kt = 0;
            for (; kt + NST < nk; ++kt) {       // tiles whose stage is refilled
                substep(P0, Y, N, kt, 0);
                substep(P1, Y, Y, kt, 1);
            }
            for (; kt + 1 < nk; ++kt) {
                substep(P0, Y, N, kt, 0);
                substep(P1, Y, N, kt, 1);
            }
            substep(P0, Y, N, nk - 1, 0);
            substep(P1, N, N, nk - 1, 1);
        }
    } else {
#pragma unroll
    for (int p = 0; p < NSTAGE - 1; ++p)
        if (p < nk) issue(p);
    YDET_STAMP(2);
    for (int kt = 0; kt < nk; ++kt) {
        // tile kt has landed once at most the younger tiles issued so far (NSTAGE - 2 of them, fewer at the end;
        // LPT instructions each) are still in flight
        if (NSTAGE >= 4 && kt + 2 < nk) wait_vmcnt<2 * LPT>();
        else if (NSTAGE >= 3 && kt + 1 < nk) wait_vmcnt<LPT>();
        else wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        YDET_STAMP(3 + kt);
        // this step's operand fragments are requested first, so that their LDS latency runs under the address
        // arithmetic and issue of the refill below instead of in front of the MFMAs
        const char* ws = smem + (kt % NSTAGE) * STAGE + (wn * WN) * RB;
        const char* xs = smem + (kt % NSTAGE) * STAGE + WROWS * RB + (wm * WM) * RB;
        frag_t fa[KSUB][FN], fb[KSUB][FM];
#pragma unroll
        for (int kk = 0; kk < KSUB; ++kk) {
#pragma unroll
            for (int i = 0; i < FN; ++i) fa[kk][i] = *reinterpret_cast<const frag_t*>(ws + fr_off[kk] + i * 16 * RB);
#pragma unroll
            for (int j = 0; j < FM; ++j) fb[kk][j] = *reinterpret_cast<const frag_t*>(xs + fr_off[kk] + j * 16 * RB);
        }
        asm volatile("" ::: "memory");   // keep the reads above the refill
        // every wave is past its reads of stage (kt - 1) % NSTAGE: refill it with tile kt + NSTAGE - 1
        if (kt + NSTAGE - 1 < nk) issue((kt + NSTAGE - 1) % NSTAGE);
#pragma unroll
        for (int kk = 0; kk < KSUB; ++kk)
#pragma unroll
            for (int i = 0; i < FN; ++i)
#pragma unroll
                for (int j = 0; j < FM; ++j) acc[i][j] = mma(fa[kk][i], fb[kk][j], acc[i][j]);
    }

    }

    // ---- optional 1x1 follower on the tile in registers ------------------------------------------------------
    // A wave of the 64-channel tile holds every channel of its pixels.  Finished (bias, activation, bf16) the values a
    // lane holds for one pixel -- channels 32 s + 8 q .. + 7, from fragments i = 2 s and 2 s + 1 --
    // are taken as the eight k values of B fragment s; the A fragments of the second convolution are loaded with their
    // columns in that same order (one 16-byte piece per lane), so the second GEMM needs no LDS pass and the first map
    // never leaves the chip.
    if constexpr (WARPS_N == 1 && BN == 64 && !HEAD && sizeof(TIN) == 2) {
        if (a.pw_w) {
            const int q_ = lane >> 4;
            const bool silu1 = a.act == YDET_ACT_SILU;
            bf16x8 bfr[2][FM];
            // (the activation is chosen once, outside the 32 values a lane finishes: chosen per value it compiled to a
            // branch around each of them)
            auto first_stage = [&](auto silu) {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float4 b1 = *reinterpret_cast<const float4*>(a.pw_b1 + 32 * (i >> 1) + 8 * q_ + 4 * (i & 1));
                    const float bb[4] = {b1.x, b1.y, b1.z, b1.w};
#pragma unroll
                    for (int j = 0; j < FM; ++j)
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            float x = acc[i][j][e] + bb[e];
                            if constexpr (decltype(silu)::value)
                                x = x * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(x * -1.44269504088896340736f));
                            else
                                x = x > 0.f ? x : x * a.slope;
                            bfr[i >> 1][j][(i & 1) * 4 + e] = (__bf16)x;
                        }
                }
            };
            if (silu1) first_stage(std::true_type{});
            else first_stage(std::false_type{});
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < FM; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int i2 = 0; i2 < 4; ++i2)
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
                    for (int j = 0; j < FM; ++j) acc[i2][j] = mma(pw_a[i2][s2], bfr[s2][j], acc[i2][j]);
        }
    }

    // ---- epilogue -------------------------------------------------------------------------------------
    YDET_STAMP(28);
    if (tid < BN) {
        cst[tid] = cst_b;
        cst[BN + tid] = cst_ps;
        cst[2 * BN + tid] = cst_pb;
    }
    __syncthreads();   // constants visible; also: every wave is done with the ring
    const int q = lane >> 4, px = lane & 15;
    if constexpr (HEAD) {
        // LDS: the sigmoids of WM pixels (one row of waves) in decoded-row order at 0, per-pixel records of all BM pixels
        // behind the per-channel constants
        float* stage = reinterpret_cast<float*>(smem);
        int* pix_row = reinterpret_cast<int*>(cst + 3 * BN);        // first z row of the pixel (anchor 0), -1: no pixel
        float* pix_gx = reinterpret_cast<float*>(pix_row + BM);
        float* pix_gy = pix_gx + BM;
        float* anc = pix_gy + BM;                                     // anchor (w, h) pairs of the level
        if (tid < BM) {
            const long long m = m0 + tid;
            int row = -1;
            float gx = 0.f, gy = 0.f;
            if (m < a.M) {
                const int n = (int)((unsigned)m / (unsigned)h.hw);
                const int cell = (int)((unsigned)m - (unsigned)n * (unsigned)h.hw);
                const int d = h.transposed ? h.ny : h.nx;
                const int qd = div_small(cell, d, h.rd), rem = cell - qd * d;
                // grid offset of the cell: (col, row), or with the transposed table flat cell f holds (f / ny, f % ny)
                gx = h.transposed ? (float)qd : (float)rem;
                gy = h.transposed ? (float)rem : (float)qd;
                row = (int)((long long)n * h.rows_per_img + h.level_off + cell);
            }
            pix_row[tid] = row; pix_gx[tid] = gx; pix_gy[tid] = gy;
        }
        // (1) The raw prediction map straight from registers, and the sigmoid of every value laid down in LDS in the
        // order of the decoded rows: zt[anchor][pixel][column], so each anchor's part of the tile is one run of z.
        // A lane holds channels chan(i) .. + 3 of its pixels in fragment i (with bf16 raw maps the weight rows are permuted
        // so that fragments 2 s and 2 s + 1 together are the EIGHT consecutive channels of one 16-byte piece); where
        // each of those sixteen channels lands inside an anchor's rows is worked out once per lane, not per value.
        // Pad channels (na * no .. 255) land in a spare float behind the last anchor.
        float* zt = stage;
        const int per_an = WM * h.no;                                // floats per anchor and part
        static_assert(FN % 2 == 0, "fragments pair up");
        auto chan = [&](int i) { return wn * WN + (PERM ? 32 * (i >> 1) + 8 * q + 4 * (i & 1) : 16 * i + 4 * q); };
        YDET_STAMP(20);
        // raw map out, sigmoids in place (every wave at once: this is the VALU-heavy part)
#pragma unroll
        for (int j = 0; j < FM; ++j) {
            const int pl = wm * WM + j * 16 + px;
            const long long m = m0 + pl;
            const bool okm = m < a.M;
            char* yrow = a.y + ((size_t)(okm ? m : 0) * a.y_ld + a.y_coff + n0) * sizeof(TOUT);
#pragma unroll
            for (int i = 0; i < FN; ++i) {
                const float4 b = *reinterpret_cast<const float4*>(cst + chan(i));
                acc[i][j][0] += b.x; acc[i][j][1] += b.y; acc[i][j][2] += b.z; acc[i][j][3] += b.w;
            }
            if constexpr (PERM) {
#pragma unroll
                for (int sp = 0; sp < FN / 2; ++sp) {
                    const int c0 = chan(2 * sp);
                    const bf16x8 t = {(__bf16)acc[2 * sp][j][0], (__bf16)acc[2 * sp][j][1], (__bf16)acc[2 * sp][j][2], (__bf16)acc[2 * sp][j][3],
                                      (__bf16)acc[2 * sp + 1][j][0], (__bf16)acc[2 * sp + 1][j][1], (__bf16)acc[2 * sp + 1][j][2],
                                      (__bf16)acc[2 * sp + 1][j][3]};
                    if (okm && n0 + c0 < a.cout_store) __builtin_nontemporal_store(t, reinterpret_cast<bf16x8*>(yrow + c0 * 2));
                }
            } else {
#pragma unroll
                for (int i = 0; i < FN; ++i) {
                    const int c0 = chan(i);
                    if (okm && n0 + c0 < a.cout_store)
                        __builtin_nontemporal_store(f32x4{acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]},
                                                    reinterpret_cast<f32x4*>(yrow + c0 * sizeof(TOUT)));
                }
            }
#pragma unroll
            for (int i = 0; i < FN; ++i)
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[i][j][e] = sigmoid_fast(acc[i][j][e]);
        }
        if (tid < 16) anc[tid] = h.anchor[tid];
        // value (i, e) of pixel pl (of its part) goes to zt[min(zoff[i][e] + pl * no, dump)]: a pad channel's offset is
        // `dump` itself, a spare float behind the last anchor's rows (the region holds WM * 260 floats, the rows at most
        // WM * 256)
        const int dump = h.na * per_an;
        int zoff[FN][4];
        {
            const float rno = 1.0f / (float)h.no;
#pragma unroll
            for (int i = 0; i < FN; ++i) {
                const int c0 = chan(i);
                int an = div_small(c0, h.no, rno), o = c0 - an * h.no;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    zoff[i][e] = an < h.na ? an * per_an + o : dump;
                    const bool wrap = o + 1 == h.no;
                    o = wrap ? 0 : o + 1;
                    an += wrap ? 1 : 0;
                }
            }
        }
        const int an_step = h.hw * h.no;                                   // floats of z between anchors (< 2^31 checked)
        for (int part = 0; part < WARPS_M; ++part) {
            // (2) the waves that hold this part's pixels lay their sigmoids down
            if (wm == part) {
#pragma unroll
                for (int j = 0; j < FM; ++j) {
                    const int pofs = (j * 16 + px) * h.no;
#pragma unroll
                    for (int i = 0; i < FN; ++i)
#pragma unroll
                        for (int e = 0; e < 4; ++e) zt[min(zoff[i][e] + pofs, dump)] = acc[i][j][e];
                }
            }
            __syncthreads();
            YDET_STAMP(23);
            // (3) decode (yolov5_official/model.py:56-81).  Row maxima: one thread per (pixel, anchor) walks the classes.
            const int* prow = pix_row + part * WM;
            for (int item = tid; item < WM * h.na; item += NT) {
                const int an = item / WM, pl = item - an * WM;
                const int row = prow[pl];
                float* src = zt + an * per_an + pl * h.no;
                const float4 s4 = make_float4(src[0], src[1], src[2], src[3]);
                if (h.row_max && row >= 0) {
                    const float obj = src[4];
                    float mx[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};   // independent chains: reads overlap
                    int c = 5;
                    for (; c + 8 <= h.no; c += 8) {
                        float t[8];
#pragma unroll
                        for (int e = 0; e < 8; ++e) t[e] = src[c + e];
#pragma unroll
                        for (int e = 0; e < 8; ++e) mx[e & 3] = fmaxf(mx[e & 3], t[e] * obj);
                    }
                    for (; c < h.no; ++c) mx[0] = fmaxf(mx[0], src[c] * obj);
                    h.row_max[(long long)row + (long long)an * h.hw] = fmaxf(fmaxf(mx[0], mx[1]), fmaxf(mx[2], mx[3]));
                }
                // the box: centre from the cell offset and the stride, size from the anchor
                const float tx = s4.x * 2.0f, ty = s4.y * 2.0f, tw = s4.z * 2.0f, th = s4.w * 2.0f;
                src[0] = (tx - 0.5f + pix_gx[part * WM + pl]) * h.stride;
                src[1] = (ty - 0.5f + pix_gy[part * WM + pl]) * h.stride;
                src[2] = tw * tw * anc[2 * an];
                src[3] = th * th * anc[2 * an + 1];
            }
            __syncthreads();
            YDET_STAMP(24);
            // (4) z: when the part's pixels are consecutive rows of one image, each anchor's share is one run of WM*no
            // floats in z as well, and it goes out as whole 16-byte pieces
            {
                const int row0 = prow[0];
                const bool run = row0 >= 0 && prow[WM - 1] == row0 + WM - 1;
                const long long z0 = (long long)row0 * h.no;                       // first float of anchor 0's run
                const bool vec = run && (z0 & 3) == 0 && (an_step & 3) == 0 && (per_an & 3) == 0 && ((size_t)h.z & 15) == 0;
                if (vec) {
                    const int per4 = per_an / 4, total4 = per4 * h.na;
                    const float rper4 = 1.0f / (float)per4;
                    float* zb = h.z + z0;
                    constexpr int ZU = NT >= 512 ? 4 : 8;
                    for (int k0 = tid; k0 < total4; k0 += NT * ZU) {
                        float4 v[ZU];
#pragma unroll
                        for (int u = 0; u < ZU; ++u) {
                            const int k = k0 + u * NT;
                            if (k < total4) v[u] = *reinterpret_cast<const float4*>(zt + 4 * k);
                        }
#pragma unroll
                        for (int u = 0; u < ZU; ++u) {
                            const int k = k0 + u * NT;
                            if (k < total4) {
                                const int an = div_small(k, per4, rper4);
                                __builtin_nontemporal_store(f32x4{v[u].x, v[u].y, v[u].z, v[u].w},
                                                            reinterpret_cast<f32x4*>(zb + (size_t)(unsigned)(an * an_step + (k - an * per4) * 4)));
                            }
                        }
                    }
                } else {
                    const float rno = 1.0f / (float)h.no;
                    for (int an = 0; an < h.na; ++an)
                        for (int f = tid; f < per_an; f += NT) {
                            const int pl = div_small(f, h.no, rno), o = f - pl * h.no;
                            const int row = prow[pl];
                            if (row >= 0) h.z[((long long)row + (long long)an * h.hw) * h.no + o] = zt[an * per_an + f];
                        }
                }
            }
            if (part + 1 < WARPS_M) __syncthreads();     // the next part overwrites the rows just read
        }
        YDET_STAMP(25);
        return;
    }
    // The values a lane holds for a pixel are whole 16-byte pieces of the output row (see PERM above): they leave
    // straight from registers, no pass through LDS.  Piece p of a lane: CP channels from chan(p) on, out of fragments
    // 2 p and 2 p + 1 (bf16 outputs) or fragment p (fp32 outputs).
    char* ybase = a.y + (long long)split * a.split_stride;      // (a K split writes partial sums: one slab per split)
    constexpr int CP = PERM ? 8 : 4;                  // channels per piece
    constexpr int NP = PERM ? FN / 2 : FN;            // pieces per lane and pixel
    static_assert(!PERM || FN % 2 == 0, "bf16 pieces pair up fragments");
    auto chan = [&](int p) { return wn * WN + (PERM ? 32 * p + 8 * q : 16 * p + 4 * q); };
    if (a.lds_epilogue) {
      with_epilogue_mode(a, [&](auto mode) {
        constexpr int MODE = decltype(mode)::value;
        constexpr bool NORES = MODE >= 0 && (MODE & 8) != 0;
        YDET_STAMP(20);
#pragma unroll
        for (int j = 0; j < FM; ++j) {
            const long long m = m0 + wm * WM + j * 16 + px;
            const bool okm = m < a.M;
            const long long mm = okm ? m : 0;
            // residual pieces of this pixel, requested before the first value is finished
            float r[NP][CP];
#pragma unroll
            for (int p = 0; p < NP; ++p)
#pragma unroll
                for (int c = 0; c < CP; ++c) r[p][c] = 0.f;
            if constexpr (!NORES) {
                if (a.res) {
                    const char* rrow = a.res + ((size_t)mm * a.res_ld + a.res_coff + n0) * sizeof(TIN);
#pragma unroll
                    for (int p = 0; p < NP; ++p) {
                        if (okm && n0 + chan(p) < a.Cout) {
                            if constexpr (sizeof(TIN) == 2 && CP == 8) {
                                const bf16x8 t = *reinterpret_cast<const bf16x8*>(rrow + chan(p) * 2);
#pragma unroll
                                for (int e = 0; e < 8; ++e) r[p][e] = (float)t[e];
                            } else {
                                load4<TIN>(rrow + (size_t)chan(p) * sizeof(TIN), r[p]);
                            }
                        }
                    }
                }
            }
#pragma unroll
            for (int p = 0; p < NP; ++p) {
                const int cl = chan(p);
                float o[CP];
#pragma unroll
                for (int hf = 0; hf < CP / 4; ++hf) {
                    const int i = PERM ? 2 * p + hf : p;
                    float v[4] = {acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]};
                    const float r4[4] = {r[p][4 * hf], r[p][4 * hf + 1], r[p][4 * hf + 2], r[p][4 * hf + 3]};
                    finish4<MODE>(v, r4, cst + cl + 4 * hf, cst + BN + cl + 4 * hf, cst + 2 * BN + cl + 4 * hf, a);
                    o[4 * hf] = v[0]; o[4 * hf + 1] = v[1]; o[4 * hf + 2] = v[2]; o[4 * hf + 3] = v[3];
                }
                const int co = n0 + cl;
                if (!okm || co >= a.cout_store) continue;
                // channels from y_split on belong to the second destination
                char* dst = co >= a.y_split ? a.y2 + ((size_t)m * a.y2_ld + a.y2_coff + (co - a.y_split)) * sizeof(TOUT)
                                            : ybase + ((size_t)m * a.y_ld + a.y_coff + co) * sizeof(TOUT);
                if constexpr (PERM) {
                    bf16x8 t;
#pragma unroll
                    for (int e = 0; e < 8; ++e) t[e] = (__bf16)o[e];
                    *reinterpret_cast<bf16x8*>(dst) = t;
                } else {
                    store4<TOUT>(dst, o);
                }
            }
        }
        YDET_STAMP(30);
      });
      return;
    }
    // output slices that rule out 16-byte pieces: element stores, the variant read at run time
    {
#pragma unroll
        for (int j = 0; j < FM; ++j) {
            const long long m = m0 + wm * WM + j * 16 + px;
            if (m >= a.M) continue;
            char* yrow = ybase + ((size_t)m * a.y_ld + a.y_coff) * sizeof(TOUT);
            const char* rrow = a.res ? a.res + ((size_t)m * a.res_ld + a.res_coff) * sizeof(TIN) : nullptr;
#pragma unroll
            for (int i = 0; i < FN; ++i) {
                const int cl = wn * WN + (PERM ? 32 * (i >> 1) + 8 * q + 4 * (i & 1) : 16 * i + 4 * q);
                const int co = n0 + cl;
                if (co >= a.Cout) continue;
                float v[4] = {acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]};
                float r[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (rrow && co + e < a.Cout) r[e] = to_f32(reinterpret_cast<const TIN*>(rrow)[co + e]);
                finish4<-1>(v, r, cst + cl, cst + BN + cl, cst + 2 * BN + cl, a);
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (co + e < a.Cout) reinterpret_cast<TOUT*>(yrow)[co + e] = from_f32<TOUT>(v[e]);
            }
        }
    }
}

// Grouped and / or dilated convolutions (nn.Conv2d(groups=, dilation=); DWConv and Bottleneck(g=) of
// yolov5_official/modules.py:19-21, :208-218; `d`, `g` of utils/moduleUtils.py:9-13): one thread per (pixel, output
// channel), channels fastest so that stores and the depthwise reads are contiguous; fp32 accumulation in tap-major,
// channel-minor order.  Such layers carry a small share of a network's work (a depthwise 3x3 is 9 MACs per output);
// the MFMA kernels stay reserved for dense convolutions.
template <typename TIN, typename TOUT>
__global__ __launch_bounds__(256) void conv_direct_kernel(const ConvArgs a) {
    const long long total = a.M * (long long)a.cout_store;
    const int HoWo = a.Ho * a.Wo, cpg = a.Cout / a.groups;
    const TIN* X = reinterpret_cast<const TIN*>(a.x);
    const TIN* Wt = reinterpret_cast<const TIN*>(a.w);
    for (long long gid = (long long)blockIdx.x * 256 + threadIdx.x; gid < total; gid += (long long)gridDim.x * 256) {
        const long long m = gid / a.cout_store;
        const int co = (int)(gid - m * a.cout_store);
        float v = 0.f;
        if (co < a.Cout) {
            const int n = (int)(m / HoWo), rem = (int)(m - (long long)n * HoWo);
            const int ho = rem / a.Wo, wo = rem - ho * a.Wo;
            const int c_first = (co / cpg) * a.cin_group;
            const TIN* wrow = Wt + (size_t)co * a.kp;
            for (int kh = 0; kh < a.KH; ++kh) {
                const int hi = ho * a.stride - a.pad + kh * a.dilation;
                if ((unsigned)hi >= (unsigned)a.H) continue;
                for (int kw = 0; kw < a.KW; ++kw) {
                    const int wi = wo * a.stride - a.pad + kw * a.dilation;
                    if ((unsigned)wi >= (unsigned)a.W) continue;
                    const TIN* xp = X + ((size_t)((long long)n * a.H + hi) * a.W + wi) * a.x_ld + a.x_coff + c_first;
                    const TIN* wp = wrow + (kh * a.KW + kw) * a.cin_group;
                    for (int c = 0; c < a.cin_group; ++c) v += to_f32(xp[c]) * to_f32(wp[c]);
                }
            }
            if (a.bias) v += a.bias[co];
            float r = 0.f;
            if (a.res) r = to_f32(reinterpret_cast<const TIN*>(a.res)[(size_t)m * a.res_ld + a.res_coff + co]);
            if (a.res_mode == YDET_RES_BEFORE_ACT) v += r;
            v = a.act == YDET_ACT_SILU ? activate(v, YDET_ACT_SILU, 0.f) : (v > 0.f ? v : v * a.slope);
            if (a.res_mode == YDET_RES_AFTER_ACT) v += r;
            if (a.post_scale) v = activate(v * a.post_scale[co] + a.post_shift[co], a.post_act, a.post_slope);
        }
        reinterpret_cast<TOUT*>(a.y)[(size_t)m * a.y_ld + a.y_coff + co] = from_f32<TOUT>(v);
    }
}

template <typename TIN, typename TOUT>
int launch_direct(const ConvArgs& a, hipStream_t stream) {
    if (kernel_probe().on) {
        snprintf(kernel_probe().name, sizeof(kernel_probe().name), "conv_direct");
        return YDET_OK;
    }
    const long long total = a.M * (long long)a.cout_store;
    long long blocks = (total + 255) / 256;
    if (blocks > 256 * 64) blocks = 256 * 64;       // grid-stride beyond 64 workgroups per CU
    hipLaunchKernelGGL((conv_direct_kernel<TIN, TOUT>), dim3((unsigned)blocks), dim3(256), 0, stream, a);
    YDET_CHECK_LAUNCH("conv_direct_kernel");
    return YDET_OK;
}

template <typename TIN, typename TOUT, int BN, int BM, int WARPS_N, int WARPS_M, int KCH = 4, int NST = 3, bool PIPE = false>
int launch_tile(const ConvArgs& a0, hipStream_t stream) {
    ConvArgs a = a0;
    a.tiles_n = (a.Cout + BN - 1) / BN;
    const long long tiles_m = (a.M + BM - 1) / BM;
    a.tiles_per_split = (int)(tiles_m * a.tiles_n);
    const long long nblk = tiles_m * a.tiles_n * (a.splits > 1 ? a.splits : 1);
    if (nblk <= 0 || nblk > 0x7fffffffLL) return fail(YDET_ERR_ARG, "conv: grid of %lld blocks", nblk);
    constexpr int BK = KCH * 16 / (int)sizeof(TIN);
    if constexpr (KCH != 4) {
        if (a.Cin % BK != 0 || a.KH * a.KW > 32)
            return fail(YDET_ERR_ARG, "conv: the %d-byte-row tile needs Cin to be a multiple of %d", KCH * 16, BK);
    }
    // uniform-tap LDS-DMA kernel; the 32-channel tile keeps the register-staged kernel (measured faster there:
    // its weight image is half padding under the DMA group layout)
    if (BN >= 64 && a.Cin % BK == 0 && a.KH * a.KW <= 32) {
        constexpr int NW = WARPS_N * WARPS_M;
        constexpr int RPI = 64 / KCH;
        constexpr int WROWS = BN < RPI * NW ? RPI * NW : BN;
        const size_t lds_dma = NST * (size_t)(WROWS + BM) * (KCH * 16) + 3 * BN * sizeof(float);
        // 16-byte pieces straight from registers: output (and residual) slices must be 16-byte aligned
        a.lds_epilogue = (a.y_ld % (16 / (int)sizeof(TOUT)) == 0 && a.y_coff % (16 / (int)sizeof(TOUT)) == 0 &&
                          a.cout_store % (16 / (int)sizeof(TOUT)) == 0 &&
                          (!a.res || (a.res_ld % (16 / (int)sizeof(TIN)) == 0 && a.res_coff % (16 / (int)sizeof(TIN)) == 0)))
                             ? 1
                             : 0;
        if (a.y2 && !a.lds_epilogue) return fail(YDET_ERR_ARG, "conv: a second destination needs 16-byte aligned output slices");
        if (kernel_probe().on) {
            snprintf(kernel_probe().name, sizeof(kernel_probe().name), "conv_dma<%dx%d,%s>", BN, BM, dtype_pair_name<TIN, TOUT>());
            return YDET_OK;
        }
        if (lds_dma > 64 * 1024) {   // beyond the default dynamic-LDS limit: opt in once per instantiation and device
            static unsigned char optin[kMaxDevices];
            const int rc = raise_dynamic_lds_limit(
                reinterpret_cast<const void*>(&conv_dma_kernel<TIN, TOUT, BN, BM, WARPS_N, WARPS_M, false, KCH, NST, PIPE>), (int)lds_dma, optin, "conv");
            if (rc != YDET_OK) return rc;
        }
        hipLaunchKernelGGL((conv_dma_kernel<TIN, TOUT, BN, BM, WARPS_N, WARPS_M, false, KCH, NST, PIPE>), dim3((unsigned)nblk),
                           dim3(64 * NW), lds_dma, stream, a, HeadArgs{});
        YDET_CHECK_LAUNCH("conv_dma_kernel");
        return YDET_OK;
    }
    if (a.y2) return fail(YDET_ERR_ARG, "conv: a second destination needs Cin to be a multiple of %d and at least 64 output channels", BK);
    if (a.pw_w) return fail(YDET_ERR_ARG, "conv: a 1x1 follower needs the LDS-DMA kernel");
    if (a.x_up) return fail(YDET_ERR_ARG, "conv: an upsampled source needs the LDS-DMA kernel (Cin a multiple of %d, more than 32 output channels)", BK);
    if constexpr (WARPS_N * WARPS_M != 4 || KCH != 4 || PIPE) {
        return fail(YDET_ERR_ARG, "conv: the %dx%d tile needs Cin to be a multiple of %d", BN, BM, BK);
    } else {
        const size_t lds = 2 * (size_t)(BN + BM) * 64;
        const bool is1x1 = a.KH == 1 && a.KW == 1 && a.pad == 0;
        if (kernel_probe().on) {
            snprintf(kernel_probe().name, sizeof(kernel_probe().name), "conv_generic<%dx%d,%s,%s>", BN, BM, is1x1 ? "1x1" : "kxk",
                     dtype_pair_name<TIN, TOUT>());
            return YDET_OK;
        }
        if (is1x1)
            hipLaunchKernelGGL((conv_generic_kernel<TIN, TOUT, BN, BM, WARPS_N, WARPS_M, true>), dim3((unsigned)nblk),
                               dim3(256), lds, stream, a);
        else
            hipLaunchKernelGGL((conv_generic_kernel<TIN, TOUT, BN, BM, WARPS_N, WARPS_M, false>), dim3((unsigned)nblk),
                               dim3(256), lds, stream, a);
        YDET_CHECK_LAUNCH("conv_generic_kernel");
        return YDET_OK;
    }
}

// 128 x 256 tile (8 waves, 2 workgroups per CU) or 128 x 128 (4 waves, 3 per CU)?  The wide tile moves 25 % fewer
// bytes through the LDS-DMA path per FLOP, which is what bounds a K step, and halves the number of workgroups.  It
// (bf16 outputs only: an fp32 output tile no longer fits the LDS epilogue in one pass).  It
// pays when that turns a grid slightly larger than the resident slots (a second, nearly empty round) into one that
// fits, and never when it leaves CUs without work.  Measured on MI355X (256 CUs); mirrored by
// yolo_amd/engine.py:conv_kernel_label.
static bool wide_tile_pays(long long M, int Cout, int K, bool dma_ok) {
    if (!dma_ok) return false;                       // the wide tile exists for the LDS-DMA kernel only
    const long long tn = (Cout + 127) / 128;
    const double b128 = (double)((M + 127) / 128 * tn), b256 = (double)((M + 255) / 256 * tn);
    if (b256 < 256.0) return false;
    const double w128 = b128 / 768.0, w256 = b256 / 512.0;
    const double r128 = ceil(w128) / w128, r256 = ceil(w256) / w256;   // >= 1: slots paid for / slots used
    const double gain = K >= 256 ? 0.85 : 1.0;
    return r256 * gain < r128 - 0.02;
}

// The pipelined tiles address their operands with 32-bit offsets and have no path for an upsampled source or a K split.
static bool pipe_ok(const ConvArgs& a) {
    return a.x_span > 0 && a.x_span < 0xFFFFFFF0LL && a.w_span < 0x7FFFFFFFLL && a.up_C == 0 && a.splits <= 1 && a.pw_w == nullptr;
}

template <typename TIN, typename TOUT>
int launch_dtype(const ConvArgs& a, hipStream_t stream) {
    if (a.groups > 1 || a.dilation > 1) return launch_direct<TIN, TOUT>(a, stream);
    // YDET_CONV_TILE=<channels>x<pixels> forces a tile shape (tuning aid)
    static const char* forced = getenv("YDET_CONV_TILE");
    if (forced) {
        int bn = 0, bm = 0, kch = 4, nst = 3;
        const int got = sscanf(forced, "%dx%dk%ds%d", &bn, &bm, &kch, &nst);
        if constexpr (sizeof(TIN) == 2 && sizeof(TOUT) == 2) {
            // 256x256p: the software-pipelined form (eight waves of 128 x 64, 128-byte rows, two stages, Cin a multiple of 64)
            if (got == 2 && strchr(forced, 'p') && a.KH * a.KW <= 32 && pipe_ok(a)) {
                if (bn == 256 && bm == 256 && a.Cin % 64 == 0 && a.Cout > 128) return launch_tile<TIN, TOUT, 256, 256, 2, 4, 8, 2, true>(a, stream);
                if (bn == 256 && bm == 192 && a.Cin % 64 == 0 && a.Cout > 128) return launch_tile<TIN, TOUT, 256, 192, 2, 4, 8, 2, true>(a, stream);
            }
            // <channels>x<pixels>k8s<stages>: the 128-byte-row variants (where Cin is a multiple of 64; other layers
            // keep the default choice)
            if (got == 4 && kch == 8 && a.Cin % 64 == 0 && a.KH * a.KW <= 32 && a.Cout > 64) {
                if (bn == 128 && bm == 128 && nst == 2) return launch_tile<TIN, TOUT, 128, 128, 2, 2, 8, 2>(a, stream);
                if (bn == 128 && bm == 128 && nst == 4) return launch_tile<TIN, TOUT, 128, 128, 2, 2, 8, 4>(a, stream);
                if (bn == 128 && bm == 64 && nst == 3) return launch_tile<TIN, TOUT, 128, 64, 2, 2, 8, 3>(a, stream);
            }
        }
        if (got == 2) {
            if (bn == 32 && bm == 256) return launch_tile<TIN, TOUT, 32, 256, 1, 4>(a, stream);
            if (bn == 64 && bm == 128) return launch_tile<TIN, TOUT, 64, 128, 1, 4>(a, stream);
            if (bn == 64 && bm == 256) return launch_tile<TIN, TOUT, 64, 256, 1, 4>(a, stream);
            if (bn == 128 && bm == 128) return launch_tile<TIN, TOUT, 128, 128, 2, 2>(a, stream);
            if (bn == 128 && bm == 64) return launch_tile<TIN, TOUT, 128, 64, 2, 2>(a, stream);
            if (bn == 128 && bm == 256) return launch_tile<TIN, TOUT, 128, 256, 2, 4>(a, stream);
            if constexpr (sizeof(TIN) == 2 && sizeof(TOUT) == 2) {
                if (bn == 256 && bm == 256 && a.Cout > 128 && a.Cin % 32 == 0 && a.KH * a.KW <= 32)
                    return launch_tile<TIN, TOUT, 256, 256, 4, 4>(a, stream);
            }
        }
    }
    if constexpr (sizeof(TIN) == 2 && sizeof(TOUT) == 2) {
        if (conv_pw_takes(a)) return launch_conv_pw(a, stream);
        if (conv_patch_takes(a)) return launch_conv_patch(a, stream);
    }
    if (a.Cout <= 32) return launch_tile<TIN, TOUT, 32, 256, 1, 4>(a, stream);
    if (a.Cout <= 64) return launch_tile<TIN, TOUT, 64, 128, 1, 4>(a, stream);
    if constexpr (sizeof(TIN) == 2 && sizeof(TOUT) == 2) {
        // Wide layers (256 output channels and more) with K >= 512 and enough pixels for about one tile per CU: the
        // 256 x 256 tile with sixteen waves (one workgroup per CU) moves a third fewer bytes through the LDS-DMA path per
        // FLOP than two 128 x 256 workgroups.  Measured on MI355X, isolated and bit-identical: 3x3 256 -> 512 at 38 x 38
        // 222 -> 204 us, 3x3 512 -> 1024 at 19 x 19 251 -> 235 us, 3x3 / 2 128 -> 256 at 80 x 80 78.5 -> 75.0 us, level
        // on the 20 x 20 1x1 layers; with one channel tile and 100 pixel tiles (a 3x3 256 -> 256 at 20 x 20) it leaves
        // most of the chip idle (37 -> 63 us).  YDET_CONV_256=0 turns the rule off.
        static const bool t256_on = !(getenv("YDET_CONV_256") && atoi(getenv("YDET_CONV_256")) == 0);
        const int rem256 = a.Cout % 256;
        const long long tiles256 = (a.M + 255) / 256 * ((a.Cout + 255) / 256), tiles192 = (a.M + 191) / 192 * ((a.Cout + 255) / 256);
        static const bool pipe_on = !(getenv("YDET_CONV_PIPE") && atoi(getenv("YDET_CONV_PIPE")) == 0);
        const bool piped = pipe_on && a.Cin % 64 == 0 && pipe_ok(a);
        // (the pipelined form also comes with 192-pixel tiles: 182 tiles of 256 pixels leave a quarter of the CUs idle where
        // 242 of 192 pixels fill them -- YOLOv4's 19 x 19 layers with 512 output channels)
        if (t256_on && a.Cout >= 256 && (rem256 == 0 || rem256 > 128) && a.Cin % 32 == 0 && a.KH * a.KW <= 32 &&
            a.KH * a.KW * a.Cin >= 512 && (piped ? tiles192 : tiles256) >= 192 && a.splits <= 1) {
            // ... as eight waves of 128 x 64 with the software-pipelined K loop where the shape allows it (128-byte rows:
            // Cin a multiple of 64; operands within 4 GB): bit-identical, 6-15 % faster on MI355X (3x3 256 -> 512 at 38 x 38
            // 215 -> 191 us = 1.14 PFLOP/s, 3x3 512 -> 1024 at 19 x 19 238 -> 207 us, 3x3 128 -> 256 at 76 x 76 272 -> 256 us)
            if (piped) {
                // One workgroup per CU: a launch costs its rounds of 256 tiles x the tile's pixels.  Where 192-pixel tiles
                // fill the last round and 256-pixel tiles leave it mostly empty, the narrower tile wins although it moves more
                // bytes per FLOP (measured, bit-identical: 1x1 512 -> 256 at 38 x 38, 361 tiles vs 482: 41.0 -> 37.0 us; 3x3
                // 512 -> 1024 at 19 x 19, 364 vs 484: 213 -> 175 us = 1.25 PFLOP/s; 1x1 1024 -> 512 at 19 x 19 38.0 -> 28.9 us;
                // at equal cost -- 3x3 256 -> 512 at 38 x 38 -- the wide tile is 4 % faster).
                const long long tn = (a.Cout + 255) / 256;
                const long long r256 = ((a.M + 255) / 256 * tn + 255) / 256 * 256, r192 = ((a.M + 191) / 192 * tn + 255) / 256 * 192;
                if (r192 * 100 < r256 * 92) return launch_tile<TIN, TOUT, 256, 192, 2, 4, 8, 2, true>(a, stream);
                return launch_tile<TIN, TOUT, 256, 256, 2, 4, 8, 2, true>(a, stream);
            }
            return launch_tile<TIN, TOUT, 256, 256, 4, 4>(a, stream);
        }
        // Few tiles and a long K: the launch is one or two rounds of workgroups whose time is the K loop's, and the
        // 128-byte-row variant walks K in half as many steps (half the barriers and waits; two stages of twice the
        // size).  With many tiles per CU the three-stage ring at three workgroups per CU hides more (measured).
        const long long tiles = (a.M + 127) / 128 * ((a.Cout + 127) / 128);
        const int K = a.KH * a.KW * a.Cin;
        static const bool k8_on = !(getenv("YDET_CONV_K8") && atoi(getenv("YDET_CONV_K8")) == 0);
        if (k8_on && a.Cin % 64 == 0 && a.KH * a.KW <= 32 && K >= 256 && (tiles <= 512 || (tiles <= 1024 && K >= 2048))) {
            // At most one tile per CU and a long K (single images, the deepest layers): the launch is as long as one
            // workgroup's walk over K, which waits for HBM at every step -- a four-stage ring (128 KB of LDS, one
            // workgroup per CU is all there is anyway) keeps three steps in flight instead of one.  The order of the
            // sums does not change.
            if (tiles <= 256 && K >= 1024) return launch_tile<TIN, TOUT, 128, 128, 2, 2, 8, 4>(a, stream);
            return launch_tile<TIN, TOUT, 128, 128, 2, 2, 8, 2>(a, stream);
        }
    }
    if (sizeof(TOUT) == 2 && wide_tile_pays(a.M, a.Cout, a.KH * a.KW * a.Cin, a.Cin % (64 / (int)sizeof(TIN)) == 0 && a.KH * a.KW <= 32))
        return launch_tile<TIN, TOUT, 128, 256, 2, 4>(a, stream);
    return launch_tile<TIN, TOUT, 128, 128, 2, 2>(a, stream);
}

}  // namespace

static int fill_conv_args(const ydet_conv_desc* d, ConvArgs& a) {
    YDET_CHECK_ARG(d != nullptr, "conv: null descriptor");
    YDET_CHECK_ARG(d->x && d->w && d->y, "conv: null x/w/y pointer");
    YDET_CHECK_ARG(dtype_ok(d->in_dtype) && dtype_ok(d->out_dtype), "conv: bad dtype %d/%d", d->in_dtype,
                   d->out_dtype);
    YDET_CHECK_ARG(!(d->in_dtype == YDET_F32 && d->out_dtype == YDET_BF16), "conv: f32 -> bf16 is not built");
    const int epc = d->in_dtype == YDET_BF16 ? 8 : 4;
    const int bk = 4 * epc;
    YDET_CHECK_ARG(d->N > 0 && d->H > 0 && d->W > 0 && d->Cin > 0 && d->Cout > 0, "conv: empty shape");
    YDET_CHECK_ARG(d->KH > 0 && d->KW > 0 && d->stride > 0 && d->pad >= 0, "conv: bad kernel geometry");
    const int groups = d->groups > 0 ? d->groups : 1, dil = d->dilation > 0 ? d->dilation : 1;
    const bool direct = groups > 1 || dil > 1;
    YDET_CHECK_ARG(d->Ho == (d->H + 2 * d->pad - dil * (d->KH - 1) - 1) / d->stride + 1 &&
                       d->Wo == (d->W + 2 * d->pad - dil * (d->KW - 1) - 1) / d->stride + 1 && d->Ho > 0 && d->Wo > 0,
                   "conv: Ho/Wo (%d,%d) do not match H,W,k,s,p,d", d->Ho, d->Wo);
    const int cin_group = direct ? (d->cin_group > 0 ? d->cin_group : d->Cin / groups) : d->Cin;
    if (direct) {
        YDET_CHECK_ARG(d->Cout % groups == 0 && cin_group >= 1 && (long long)cin_group * groups <= d->Cin,
                       "conv: %d groups of %d input channels over Cin=%d, Cout=%d", groups, cin_group, d->Cin, d->Cout);
        YDET_CHECK_ARG(d->y2 == nullptr && d->x_up == nullptr && d->pw_w == nullptr,
                       "conv: a grouped / dilated convolution excludes y2, x_up and a 1x1 follower");
    }
    YDET_CHECK_ARG(d->Cin % epc == 0 && d->x_ld % epc == 0 && d->x_coff % epc == 0,
                   "conv: Cin/x_ld/x_coff (%d,%d,%d) must be multiples of %d", d->Cin, d->x_ld, d->x_coff, epc);
    YDET_CHECK_ARG(d->x_coff >= 0 && d->x_coff + d->Cin <= d->x_ld, "conv: input slice exceeds x_ld");
    const int cout_eff = d->pw_w ? d->pw_cout : d->Cout;   // channels of the finished tile
    if (d->pw_w) {
        YDET_CHECK_ARG(d->in_dtype == YDET_BF16 && d->out_dtype == YDET_BF16 && d->Cout == 64 && d->pw_cout > 32 && d->pw_cout <= 64 &&
                           d->pw_kp == 64 && d->Cin % bk == 0 && d->KH * d->KW <= 32 && d->bias && d->pw_bias &&
                           ((uintptr_t)d->pw_w & 15) == 0 && ((uintptr_t)d->pw_bias & 15) == 0,
                       "conv: a 1x1 follower needs bf16, 64 channels out of the first convolution, 33..64 out of the second "
                       "(weights packed for 64 inputs), Cin a multiple of %d and both biases", bk);
        YDET_CHECK_ARG(d->res == nullptr && d->post_scale == nullptr && d->cout_store == 0 && d->x_up == nullptr,
                       "conv: a 1x1 follower excludes a residual, a second epilogue stage, cout_store and an upsampled source");
    }
    const int y_ch = d->y2 ? d->y_split : cout_eff;   // channels that land in y
    YDET_CHECK_ARG(d->y_coff >= 0 && d->y_coff + y_ch <= d->y_ld, "conv: output slice exceeds y_ld");
    if (d->y2) {
        const int vec = d->out_dtype == YDET_BF16 ? 8 : 4;
        YDET_CHECK_ARG(d->y_split > 0 && d->y_split < cout_eff && d->y_split % vec == 0 && cout_eff % vec == 0 &&
                           d->y2_ld % vec == 0 && d->y2_coff % vec == 0 && d->y2_coff >= 0 &&
                           d->y2_coff + cout_eff - d->y_split <= d->y2_ld && ((uintptr_t)d->y2 & 15) == 0,
                       "conv: second destination (split %d of %d channels, ld %d, coff %d)", d->y_split, d->Cout, d->y2_ld, d->y2_coff);
        YDET_CHECK_ARG(d->cout_store == 0 && d->res == nullptr, "conv: a second destination excludes cout_store and a residual");
    }
    YDET_CHECK_ARG(d->cout_store == 0 || (d->cout_store >= d->Cout && d->cout_store <= d->cout_pad &&
                                          d->y_coff + d->cout_store <= d->y_ld),
                   "conv: cout_store %d (need Cout <= cout_store <= cout_pad and inside y_ld)", d->cout_store);
    YDET_CHECK_ARG(d->kp % bk == 0 && d->kp >= d->KH * d->KW * cin_group, "conv: kp %d (need multiple of %d, >= %d)",
                   d->kp, bk, d->KH * d->KW * cin_group);
    YDET_CHECK_ARG(d->cout_pad % 128 == 0 && d->cout_pad >= d->Cout, "conv: cout_pad %d", d->cout_pad);
    YDET_CHECK_ARG(d->act >= 0 && d->act <= 2 && d->res_mode >= 0 && d->res_mode <= 2, "conv: bad act/res_mode");
    YDET_CHECK_ARG((d->res != nullptr) == (d->res_mode != YDET_RES_NONE), "conv: res pointer vs res_mode");
    YDET_CHECK_ARG((d->post_scale != nullptr) == (d->post_shift != nullptr) && d->post_act >= 0 && d->post_act <= 2 &&
                       ((uintptr_t)d->post_scale & 15) == 0 && ((uintptr_t)d->post_shift & 15) == 0,
                   "conv: post_scale/post_shift must both be given (16-byte aligned) or both be null");
    YDET_CHECK_ARG(((uintptr_t)d->x & 15) == 0 && ((uintptr_t)d->w & 15) == 0 && ((uintptr_t)d->y & 15) == 0 &&
                       ((uintptr_t)d->bias & 15) == 0 && ((uintptr_t)d->res & 15) == 0,
                   "conv: pointers must be 16-byte aligned");
    if (d->res) YDET_CHECK_ARG(d->res_coff >= 0 && d->res_coff + d->Cout <= d->res_ld, "conv: residual slice");
    YDET_CHECK_ARG((long long)d->H * d->W * d->x_ld < (1LL << 31), "conv: one image exceeds 2^31 elements");
    YDET_CHECK_ARG((long long)d->N * d->Ho * d->Wo < (1LL << 31), "conv: more than 2^31 output pixels");
    YDET_CHECK_ARG(d->Ho < (1 << 21) && d->Wo < (1 << 21), "conv: output map side beyond 2^21");

    a.x = (const char*)d->x; a.w = (const char*)d->w; a.bias = d->bias; a.res = (const char*)d->res;
    a.y = (char*)d->y;
    a.N = d->N; a.H = d->H; a.W = d->W; a.Cin = d->Cin; a.Cout = d->Cout; a.KH = d->KH; a.KW = d->KW;
    a.stride = d->stride; a.pad = d->pad; a.Ho = d->Ho; a.Wo = d->Wo;
    a.x_ld = d->x_ld; a.x_coff = d->x_coff; a.y_ld = d->y_ld; a.y_coff = d->y_coff;
    a.res_ld = d->res_ld; a.res_coff = d->res_coff;
    a.kp = d->kp; a.act = d->act; a.res_mode = d->res_mode;
    a.slope = d->act == YDET_ACT_NONE ? 1.0f : d->slope;   // the leaky form with slope 1 is the identity
    a.M = (long long)d->N * d->Ho * d->Wo;
    a.tiles_n = 0;
    a.lds_epilogue = 0;
    a.post_scale = d->post_scale; a.post_shift = d->post_shift; a.post_act = d->post_act; a.post_slope = d->post_slope;
    a.cout_store = d->cout_store > 0 ? d->cout_store : d->Cout;
    a.y2 = (char*)d->y2; a.y2_ld = d->y2_ld; a.y2_coff = d->y2_coff;
    a.y_split = d->y2 ? d->y_split : 0x7fffffff;
    a.pw_w = nullptr; a.pw_b1 = nullptr; a.pw_kp = 0;
    if (d->pw_w) {      // from here on the tile is the second convolution's: its channel count, its bias
        a.pw_w = (const char*)d->pw_w; a.pw_b1 = d->bias; a.pw_kp = d->pw_kp;
        a.bias = d->pw_bias; a.Cout = d->pw_cout; a.cout_store = d->pw_cout;
        a.vec_store = (d->pw_cout % 4 == 0 && d->y_ld % 4 == 0 && d->y_coff % 4 == 0) ? 1 : 0;
    }
    a.groups = groups; a.dilation = dil; a.cin_group = cin_group;
    a.splits = 0; a.tiles_per_split = 0; a.split_taps = 0; a.split_cin = 0; a.split_stride = 0;
    a.x_span = ((long long)d->N * d->H * d->W * d->x_ld - d->x_coff) * (long long)dtype_size(d->in_dtype);
    a.w_span = (long long)d->cout_pad * d->kp * (long long)dtype_size(d->in_dtype);
    a.x_up = nullptr; a.up_ld = 0; a.up_coff = 0; a.up_C = 0;
    if (d->x_up) {
        YDET_CHECK_ARG(d->KH == 1 && d->KW == 1 && d->stride == 1 && d->pad == 0 && d->H % 2 == 0 && d->W % 2 == 0,
                       "conv: an upsampled source needs a 1x1 convolution over an even-sized map");
        YDET_CHECK_ARG(d->x_up_C > 0 && d->x_up_C <= d->Cin && d->x_up_C % bk == 0 && d->Cin % bk == 0 && d->Cout > 32,
                       "conv: upsampled channels %d of %d (multiples of %d, more than 32 output channels)", d->x_up_C, d->Cin, bk);
        YDET_CHECK_ARG(d->x_up_ld % epc == 0 && d->x_up_coff % epc == 0 && d->x_up_coff >= 0 &&
                           d->x_up_coff + d->x_up_C <= d->x_up_ld && ((uintptr_t)d->x_up & 15) == 0,
                       "conv: upsampled source slice (ld %d, coff %d)", d->x_up_ld, d->x_up_coff);
        a.x_up = (const char*)d->x_up; a.up_ld = d->x_up_ld; a.up_coff = d->x_up_coff; a.up_C = d->x_up_C;
    }
    a.vec_store = (d->Cout % 4 == 0 && d->y_ld % 4 == 0 && d->y_coff % 4 == 0 &&
                   (!d->res || (d->res_ld % 4 == 0 && d->res_coff % 4 == 0)))
                      ? 1
                      : 0;
    return YDET_OK;
}

// ---- K split over workgroups ------------------------------------------------------------------------------------------
namespace {
struct SplitPlan {
    int splits, split_taps, split_cin, ldw;
    long long tiles;
    size_t bytes;
};

// Whether (and how) a launch is split: at most 64 tiles of 128 x 128 (a quarter of the CUs), K >= 1024, bf16 inputs with
// Cin a multiple of 64, nothing fused in that needs the whole K in one workgroup.  Taps of a k x k convolution in equal
// groups, channels of a 1x1 one in equal ranges of a multiple of 64, towards one workgroup per CU.
bool plan_splitk(const ydet_conv_desc* d, const ConvArgs& a, SplitPlan& p) {
    if (d->in_dtype != YDET_BF16 || a.groups > 1 || a.dilation > 1 || a.pw_w || a.x_up || a.Cin % 64 != 0 || a.KH * a.KW > 32 ||
        a.Cout <= 64)
        return false;
    p.tiles = (a.M + 127) / 128 * ((a.Cout + 127) / 128);
    const long long K = (long long)a.KH * a.KW * a.Cin;
    if (p.tiles > 64 || K < 1024) return false;
    const int want = (int)(256 / p.tiles) < 16 ? (int)(256 / p.tiles) : 16;
    const int taps = a.KH * a.KW;
    p.split_taps = p.split_cin = 0;
    if (taps > 1) {
        p.split_taps = (taps + want - 1) / want;
        p.splits = (taps + p.split_taps - 1) / p.split_taps;
    } else {
        int s = want;
        while (s > 1 && a.Cin % (s * 64) != 0) --s;
        p.splits = s;
        p.split_cin = a.Cin / s;
    }
    if (p.splits < 2) return false;
    p.ldw = (a.Cout + 3) / 4 * 4;
    if (p.ldw > d->cout_pad) return false;
    p.bytes = (size_t)p.splits * (size_t)a.M * (size_t)p.ldw * sizeof(float);
    return true;
}

struct ReduceArgs {
    const float* ws;
    int splits, ldw;
    long long slab;     // floats per split
};

// Sum of the splits' partial sums, then what the convolution's own epilogue does: bias, activation, residual (either
// order), the optional second stage; four channels per thread, stored as one piece where the slice allows.
template <typename TOUT>
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const ConvArgs a, const ReduceArgs r) {
    const int per = r.ldw / 4;
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    const long long m = idx / per;
    const int c = (int)(idx - m * per) * 4;
    if (m >= a.M) return;
    const float* src = r.ws + m * r.ldw + c;
    float v[4] = {0.f, 0.f, 0.f, 0.f};
    for (int s = 0; s < r.splits; ++s) {
        const float4 t = *reinterpret_cast<const float4*>(src + (long long)s * r.slab);
        v[0] += t.x; v[1] += t.y; v[2] += t.z; v[3] += t.w;
    }
    float res[4] = {0.f, 0.f, 0.f, 0.f};
    if (a.res) {
        const __bf16* rr = reinterpret_cast<const __bf16*>(a.res) + (size_t)m * a.res_ld + a.res_coff + c;
#pragma unroll
        for (int e = 0; e < 4; ++e)
            if (c + e < a.Cout) res[e] = (float)rr[e];
    }
    finish4<-1>(v, res, a.bias ? a.bias + c : nullptr, a.post_scale ? a.post_scale + c : nullptr,
                a.post_scale ? a.post_shift + c : nullptr, a);
    const bool second = c >= a.y_split;
    char* row = second ? a.y2 + ((size_t)m * a.y2_ld + a.y2_coff + (c - a.y_split)) * sizeof(TOUT)
                       : a.y + ((size_t)m * a.y_ld + a.y_coff + c) * sizeof(TOUT);
    const bool whole = c + 4 <= a.cout_store && (second || c + 4 <= a.y_split) && a.vec_store && ((uintptr_t)row & (4 * sizeof(TOUT) - 1)) == 0;
    if (whole) {
        store4<TOUT>(row, v);
    } else {
#pragma unroll
        for (int e = 0; e < 4; ++e)
            if (c + e < a.cout_store && !(c < a.y_split && c + e >= a.y_split)) reinterpret_cast<TOUT*>(row)[e] = from_f32<TOUT>(v[e]);
    }
}

int launch_splitk(const ydet_conv_desc* d, const ConvArgs& a, const SplitPlan& p, hipStream_t stream) {
    YDET_CHECK_ARG(((uintptr_t)d->splitk_ws & 255) == 0, "conv: splitk_ws must be 256-byte aligned");
    ConvArgs part = a;
    part.bias = nullptr; part.res = nullptr; part.res_mode = YDET_RES_NONE; part.act = YDET_ACT_NONE; part.slope = 1.0f;
    part.post_scale = nullptr; part.post_shift = nullptr; part.y2 = nullptr; part.y_split = 0x7fffffff;
    part.y = (char*)d->splitk_ws; part.y_ld = p.ldw; part.y_coff = 0; part.cout_store = p.ldw; part.vec_store = 1;
    part.splits = p.splits; part.split_taps = p.split_taps; part.split_cin = p.split_cin;
    part.split_stride = (long long)a.M * p.ldw * (long long)sizeof(float);
    const int rc = launch_tile<__bf16, float, 128, 128, 2, 2, 8, 4>(part, stream);
    if (rc != YDET_OK) return rc;
    ReduceArgs r;
    r.ws = (const float*)d->splitk_ws; r.splits = p.splits; r.ldw = p.ldw; r.slab = a.M * p.ldw;
    const long long items = a.M * (p.ldw / 4);
    const unsigned blocks = (unsigned)((items + 255) / 256);
    if (d->out_dtype == YDET_BF16) hipLaunchKernelGGL(splitk_reduce_kernel<__bf16>, dim3(blocks), dim3(256), 0, stream, a, r);
    else hipLaunchKernelGGL(splitk_reduce_kernel<float>, dim3(blocks), dim3(256), 0, stream, a, r);
    YDET_CHECK_LAUNCH("splitk_reduce_kernel");
    return YDET_OK;
}
}  // namespace

extern "C" size_t ydet_conv_splitk_workspace_bytes(const ydet_conv_desc* d) {
    clear_error();
    ConvArgs a;
    SplitPlan p;
    if (!d || fill_conv_args(d, a) != YDET_OK) return 0;
    return plan_splitk(d, a, p) ? (p.bytes + 255) / 256 * 256 : 0;
}

int ydet_launch_conv(const ydet_conv_desc* d, hipStream_t stream) {
    ConvArgs a;
    const int rc = fill_conv_args(d, a);
    if (rc != YDET_OK) return rc;
    if (d->splitk_ws) {
        SplitPlan p;
        if (plan_splitk(d, a, p) && (size_t)d->splitk_ws_bytes >= p.bytes) return launch_splitk(d, a, p, stream);
    }
    if (d->in_dtype == YDET_BF16)
        return d->out_dtype == YDET_BF16 ? launch_dtype<__bf16, __bf16>(a, stream)
                                         : launch_dtype<__bf16, float>(a, stream);
    return launch_dtype<float, float>(a, stream);
}

// 1x1 prediction convolution of one pyramid level + Detect decode in its epilogue: 256-channel tiles, so a workgroup
// owns every channel of its pixels.  Two shapes: 64 pixels and four waves, or 128 pixels and eight waves (two rows of
// waves whose pixels go through the decode one row after the other, so the LDS the decoded rows take is the same):
// the wide one fetches the weights half as often per pixel and keeps four waves on a SIMD where the narrow one keeps
// two; it needs enough pixels to fill the chip.
template <typename TIN, typename TRAW, int BM, int WARPS_M, int NST>
static int launch_head_tile(const ConvArgs& a0, const HeadArgs& h, hipStream_t stream) {
    constexpr int BN = 256, WM = BM / WARPS_M;
    ConvArgs a = a0;
    a.tiles_n = 1;
    const long long nblk = (a.M + BM - 1) / BM;
    if (nblk <= 0 || nblk > 0x7fffffffLL) return fail(YDET_ERR_ARG, "head_conv_decode: grid of %lld blocks", nblk);
    const int ring = NST * (BN + BM) * 64, tile = WM * (BN + 4) * 4;
    const int lds = (tile > ring ? tile : ring) + 3 * BN * (int)sizeof(float) + 3 * BM * 4 + 16 * 4;
    static unsigned char optin[kMaxDevices];
    const int rc = raise_dynamic_lds_limit(reinterpret_cast<const void*>(&conv_dma_kernel<TIN, TRAW, BN, BM, 4, WARPS_M, true, 4, NST>), lds,
                                           optin, "head_conv_decode");
    if (rc != YDET_OK) return rc;
    hipLaunchKernelGGL((conv_dma_kernel<TIN, TRAW, BN, BM, 4, WARPS_M, true, 4, NST>), dim3((unsigned)nblk), dim3(256 * WARPS_M), lds,
                       stream, a, h);
    YDET_CHECK_LAUNCH("conv_dma_kernel<head>");
    return YDET_OK;
}

template <typename TIN, typename TRAW>
static int launch_head(const ConvArgs& a, const HeadArgs& h, hipStream_t stream) {
    // YDET_HEAD_TILE=64|128 forces a shape (tuning aid)
    static const int forced = getenv("YDET_HEAD_TILE") ? atoi(getenv("YDET_HEAD_TILE")) : 0;
    // (measured at 64 images of 640 x 640: 196 / 67 / 38 us for the three levels with the narrow shape, 173 / 63 / 30 us
    // with the wide one -- the 20 x 20 level, 200 wide workgroups with K = 512, gains most; below that many workgroups the
    // narrow shape keeps more CUs busy)
    const bool wide = forced ? forced == 128 : a.M >= 16 * 1024;
    if (wide) return launch_head_tile<TIN, TRAW, 128, 2, 2>(a, h, stream);
    return launch_head_tile<TIN, TRAW, 64, 1, 3>(a, h, stream);
}

int ydet_launch_head_decode(const ydet_conv_desc* d, const ydet_head_desc* hd, hipStream_t stream) {
    ConvArgs a;
    const int rc = fill_conv_args(d, a);
    if (rc != YDET_OK) return rc;
    YDET_CHECK_ARG(hd != nullptr && hd->z != nullptr, "head_conv_decode: null head descriptor / z");
    YDET_CHECK_ARG(a.groups == 1 && a.dilation == 1, "head_conv_decode: a plain convolution (one group, no dilation)");
    const int bk = d->in_dtype == YDET_BF16 ? 32 : 16;
    YDET_CHECK_ARG(d->KH == 1 && d->KW == 1 && d->stride == 1 && d->pad == 0 && d->Cin % bk == 0,
                   "head_conv_decode: a 1x1 convolution with Cin a multiple of %d", bk);
    YDET_CHECK_ARG((d->out_dtype == YDET_F32 || d->in_dtype == YDET_BF16) && d->act == YDET_ACT_NONE && d->res == nullptr &&
                       d->post_scale == nullptr,
                   "head_conv_decode: plain prediction map expected (fp32, or bf16 with bf16 inputs)");
    YDET_CHECK_ARG(hd->na >= 1 && hd->na <= 8 && hd->no >= 5 && hd->no <= 128 && hd->na * hd->no == d->Cout && d->Cout <= 256,
                   "head_conv_decode: na=%d no=%d Cout=%d (na*no = Cout <= 256, no <= 128)", hd->na, hd->no, d->Cout);
    const int pc = d->out_dtype == YDET_BF16 ? 8 : 4;     // channels per 16-byte piece of the raw map
    YDET_CHECK_ARG(d->cout_pad >= 256 && d->y_ld % pc == 0 && d->y_coff % pc == 0 && a.cout_store % pc == 0,
                   "head_conv_decode: weights padded to 256 rows and a %d-aligned output slice", pc);
    YDET_CHECK_ARG(hd->ny == d->Ho && hd->nx == d->Wo && (long long)hd->ny * hd->nx < (1 << 22) &&
                       (long long)hd->ny * hd->nx * d->Cout < (1ll << 31), "head_conv_decode: map size");
    YDET_CHECK_ARG(hd->rows_per_image >= (long long)hd->level_row_offset + (long long)hd->na * hd->ny * hd->nx &&
                       hd->level_row_offset >= 0, "head_conv_decode: level rows exceed rows_per_image");
    HeadArgs h;
    h.z = hd->z; h.row_max = hd->row_max;
    h.na = hd->na; h.no = hd->no; h.ny = hd->ny; h.nx = hd->nx; h.hw = hd->ny * hd->nx;
    h.transposed = hd->grid_transposed ? 1 : 0;
    h.level_off = hd->level_row_offset; h.rows_per_img = hd->rows_per_image;
    h.stride = hd->stride;
    h.rd = 1.0f / (float)(h.transposed ? hd->ny : hd->nx);
    for (int i = 0; i < 16; ++i) h.anchor[i] = i < 2 * hd->na ? hd->anchor_wh[i] : 0.f;
    if (d->in_dtype != YDET_BF16) return launch_head<float, float>(a, h, stream);
    return d->out_dtype == YDET_BF16 ? launch_head<__bf16, __bf16>(a, h, stream) : launch_head<__bf16, float>(a, h, stream);
}

extern "C" int ydet_head_conv_decode(const ydet_conv_desc* d, const ydet_head_desc* head, void* stream) {
    clear_error();
    return ydet_launch_head_decode(d, head, as_stream(stream));
}

namespace ydet {
namespace conv {
KernelProbe& kernel_probe() {
    static thread_local KernelProbe p = {false, {0}};
    return p;
}
}  // namespace conv
}  // namespace ydet

extern "C" int ydet_conv_kernel_name(const ydet_conv_desc* d, char* out, int n) {
    clear_error();
    if (!d || !out || n <= 0) return fail(YDET_ERR_ARG, "conv_kernel_name: null argument");
    ydet_conv_desc q = *d;
    q.splitk_ws = nullptr;                  // (a K split launches the same kernel family on partial sums)
    KernelProbe& p = kernel_probe();
    p.on = true;
    p.name[0] = 0;
    const int rc = ydet_launch_conv(&q, nullptr);
    p.on = false;
    if (rc != YDET_OK) return rc;
    snprintf(out, (size_t)n, "%s", p.name);
    return YDET_OK;
}

extern "C" int ydet_conv2d_nhwc(const ydet_conv_desc* d, void* stream) {
    clear_error();
    return ydet_launch_conv(d, as_stream(stream));
}
