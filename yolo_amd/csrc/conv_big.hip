// Big-tile implicit-GEMM convolution for gfx950: bf16 in, bf16 out, 128 output channels x (32 * FM) pixels per tile,
// ONE 512-thread workgroup per CU that walks over its tiles (persistent), for layers whose grid fills the chip.
//
// Same GEMM view, packed weights, LDS image (64-byte rows, chunk index XOR-swizzled on the DMA source address) and
// epilogue arithmetic as conv_dma_kernel of conv.hip.  What differs:
//
//  * Tile width chosen per layer.  The pixel extent of a tile is 2 FM fragments of 16 pixels, 6 <= FM <= 14 (a
//    template parameter), picked on the host so that the tile count is a multiple of the CU count or just below
//    one: 102400 pixels (40 x 40 x 64 images) are 247 tiles of 416 pixels, one per CU, where 256-pixel tiles leave a
//    quarter of the MFMA slots of a round empty.
//  * Waves 4 (channels) x 2 (pixels): a wave owns 32 channels x FM pixel fragments.  The two
//    halves of the workgroup (waves 0-3, waves 4-7: the two waves of each SIMD) run ONE PHASE APART: while one half
//    issues the MFMAs of a K step, the other fetches its fragments of the next step from LDS and issues its share of
//    the LDS-DMA refill; two raw barriers per K step keep the distance.  A SIMD's matrix pipe therefore always has
//    exactly one wave feeding it and no wave holds fragments of two steps.  The K loop has no branch but its own.
//  * The LDS-DMA stream runs three K steps ahead of the MFMAs through a four-stage ring and does not stop at tile
//    boundaries: the first steps of the next tile are in flight while the current tile's epilogue runs (the epilogue
//    stores from registers and does not touch the ring).  Per-channel constants of a tile travel the same way into a
//    small double-buffered area behind the ring.
//  * vmcnt immediates assume that every K step is ng LDS-DMA instructions per wave; any other vector-memory
//    operation in between (the epilogue's stores, the constants) only makes such a wait stricter, never weaker.
#include <stdlib.h>

#include "conv_common.h"

using namespace ydet;

namespace ydet {
namespace conv {
namespace {

constexpr int BG_BN = 128;                    // output channels per tile
constexpr int BG_NW = 8;                      // waves per workgroup
constexpr int BG_S = 4;                       // ring stages
constexpr int BG_AHEAD = 3;                   // K steps the DMA stream runs ahead
constexpr int BG_FM_MIN = 6, BG_FM_MAX = 14;   // pixel fragments per wave (a tile is 2 FM fragments = 32 FM pixels)
constexpr int BG_STAGE = (BG_BN + 32 * BG_FM_MAX) * 64;
constexpr int BG_CST = 4 * BG_BN * 4;         // bias, post-scale, post-shift of one tile (+ the unused half of the second DMA)
constexpr int BG_LDS = BG_S * BG_STAGE + 2 * BG_CST;

struct BigArgs {
    int tiles_n;   // channel tiles
    int tiles;     // all tiles
};

// Diagnostic build (-DYDET_BIG_TRACE): cycles per phase segment, summed over all K steps, of every wave of workgroup 0.
#ifdef YDET_BIG_TRACE
__device__ unsigned long long g_big_trace[8 * 8 + 8 * 8];
#define BG_STAMP(var)                                                                   \
    unsigned long long var;                                                             \
    __builtin_amdgcn_sched_barrier(0);                                                  \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(var)::"memory");          \
    __builtin_amdgcn_sched_barrier(0)
#define BG_ACC(slot, from, to) tr[slot] += (to) - (from)
#else
#define BG_STAMP(var) do { } while (0)
#define BG_ACC(slot, from, to) do { } while (0)
#endif

// Zeros every padded tap is loaded from (LDS-DMA cannot predicate a lane to zero): long enough for the channel walk
// of one tap (Cin <= 2048 bf16).
static __device__ __attribute__((aligned(16))) unsigned int g_zero_page[1024];

__device__ __forceinline__ void glds16(const char* src, char* dst) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                     (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
}

template <int FM, bool UP>
__global__ __launch_bounds__(64 * BG_NW, 2) void conv_big_kernel(const ConvArgs a, const BigArgs g) {
    constexpr int PF = 2 * FM;                    // pixel fragments (row groups) per tile
    constexpr int BMT = 16 * PF;                  // pixels per tile
    constexpr int XJ = (PF + 7) / 8;              // pixel row groups a wave stages, at most
    constexpr int NGW = 1 + XJ;                   // LDS-DMA instructions per K step of the waves that stage XJ groups
    constexpr int NFULL = PF - 8 * (XJ - 1);      // waves 0 .. NFULL-1 stage XJ groups, the others XJ - 1
    extern __shared__ __attribute__((aligned(16))) char smem[];

#ifdef YDET_BIG_TRACE
    BG_STAMP(ts0);
    const unsigned long long rt0 = __builtin_amdgcn_s_memrealtime();
#endif
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wn = wave & 3;                  // 32-channel slice of the tile
    const int grp = wave >> 2;                // 0: leading half, 1: trailing half (one phase behind)
    const bool full = wave < NFULL;           // this wave stages XJ pixel row groups
    const int G = gridDim.x;
    const int bid = (int)xcd_remap(blockIdx.x, gridDim.x);
    const int nk = a.kp >> 5;                 // K steps per tile (32 bf16 each)
    const int my_tiles = bid < g.tiles ? (g.tiles - bid + G - 1) / G : 0;
    const int total_steps = my_tiles * nk;

    const int srow = lane >> 2;
    const int lchunk = (lane & 3) ^ chunk_swz(lane >> 4);
    const bool pointwise = a.KH == 1 && a.KW == 1 && a.stride == 1 && a.pad == 0;
    const char* zsrc = reinterpret_cast<const char*>(g_zero_page);

    // ---- the DMA stream: tile it is in, K step inside that tile, per-lane sources of that tile ----------------------
    int dtile = bid, dk = 0, dseq = 0;
    int dc0 = 0, dkw = 0;
    unsigned dtapbit = 1u;
    int dtapbase = 0;                         // byte offset of the current tap from a pixel's first tap
    bool dnew = true;                         // the next step is the first of a tap: its per-lane sources are due
    const char* wsrc = nullptr;
    const char* xsrc[XJ];                     // per row group: the lane's pixel at tap (0, 0), channel chunk of the lane
    const char* xup[UP ? XJ : 1];
    const char* cur[XJ];                      // ... at the current tap, or the zero page where that tap is padding
    unsigned xmask[XJ];
    const int tap_w = a.x_ld * 2, tap_h = (a.W - a.KW) * a.x_ld * 2;   // byte steps to the next tap / the next tap row

    auto setup = [&](int tile) {
        const int HoWo = a.Ho * a.Wo;
        const float r_wo = 1.0f / (float)a.Wo, r_ho = 1.0f / (float)a.Ho;
        const int tile_n = tile % g.tiles_n, tile_m = tile / g.tiles_n;
        const int n0 = tile_n * BG_BN;
        const unsigned m0 = (unsigned)tile_m * (unsigned)BMT;
        // weight row of this lane: rows are staged in the order that leaves every lane with whole 16-byte pieces
        // (fragment i, MFMA row r = 4 q + e of a wave's 32 rows holds channel 8 q + 4 i + e)
        {
            const int R = wave * 16 + srow, i = (R >> 4) & 1, r = R & 15;
            const int ch = (R & ~31) + 8 * (r >> 2) + 4 * i + (r & 3);
            wsrc = a.w + ((size_t)(n0 + ch) * a.kp + lchunk * 8) * 2;
        }
        // per-channel constants of the tile -> LDS (the last wave stages the fewest row groups)
        if (wave == BG_NW - 1) {
            char* cdst = smem + BG_S * BG_STAGE + ((tile / G) & 1) * BG_CST;   // by the parity of this workgroup's tile count
            const int l32 = lane & 31;
            const char* s0 = a.bias ? (const char*)(a.bias + n0 + 4 * l32) : zsrc;
            const char* s1 = a.post_scale ? (const char*)(a.post_scale + n0 + 4 * l32) : zsrc;
            glds16(lane < 32 ? s0 : s1, cdst);
            if (a.post_scale) glds16(lane < 32 ? (const char*)(a.post_shift + n0 + 4 * l32) : zsrc, cdst + 1024);
        }
        int n_0 = 0, ho_0 = 0, wo_0 = 0;
        if (!pointwise || UP) {
            n_0 = (int)(m0 / (unsigned)HoWo);
            const unsigned rem = m0 - (unsigned)n_0 * (unsigned)HoWo;
            ho_0 = (int)(rem / (unsigned)a.Wo);
            wo_0 = (int)(rem - (unsigned)ho_0 * (unsigned)a.Wo);
        }
#pragma unroll
        for (int t = 0; t < XJ; ++t) {
            // (the last group of a wave that stages one fewer is computed like the others and never issued)
            const int row = (wave + 8 * t) * 16 + srow;
            const long long m = (long long)m0 + row;
            const bool okm = m < a.M && (t < XJ - 1 || full);
            if (pointwise) {
                xsrc[t] = a.x + ((size_t)(okm ? m : 0) * a.x_ld + a.x_coff + lchunk * 8) * 2;
                xmask[t] = okm ? 1u : 0u;
                if constexpr (UP) {
                    const int cw = wo_0 + row, qw = div_small(cw, a.Wo, r_wo), wo = cw - qw * a.Wo;
                    const int ch = ho_0 + qw, qh = div_small(ch, a.Ho, r_ho), ho = ch - qh * a.Ho;
                    const int n = okm ? n_0 + qh : 0;
                    const long long pix = ((long long)n * (a.Ho >> 1) + (ho >> 1)) * (a.Wo >> 1) + (wo >> 1);
                    xup[t] = a.x_up + (pix * a.up_ld + a.up_coff + lchunk * 8) * 2ll;
                }
            } else {
                const int cw = wo_0 + row, qw = div_small(cw, a.Wo, r_wo), wo = cw - qw * a.Wo;
                const int ch = ho_0 + qw, qh = div_small(ch, a.Ho, r_ho), ho = ch - qh * a.Ho;
                const int n = n_0 + qh;
                const int hi0 = ho * a.stride - a.pad, wi0 = wo * a.stride - a.pad;
                unsigned mk = 0;
                if (a.KH == 3 && a.KW == 3) {
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
                xmask[t] = okm ? mk : 0u;
                const long long pix = ((long long)n * a.H + hi0) * a.W + wi0;   // may lie before the image: masked
                xsrc[t] = a.x + (pix * a.x_ld + a.x_coff + lchunk * 8) * 2ll;
                if constexpr (UP) xup[t] = xsrc[t];
            }
        }
    };

    // one K step of the stream into its ring stage; moves on to the next tile behind a tile's last step
    auto issue_step = [&]() {
        if (dnew) {      // once per tap, not per step: which lanes read padding
            const bool from_up = UP && dc0 < a.up_C;
#pragma unroll
            for (int t = 0; t < XJ; ++t) {
                const char* base = xsrc[t];
                if constexpr (UP) base = from_up ? xup[t] : xsrc[t];
                cur[t] = (xmask[t] & dtapbit) != 0u ? base + dtapbase : zsrc;
            }
            dnew = false;
        }
        char* dst = smem + (dseq & (BG_S - 1)) * BG_STAGE;
        glds16(wsrc + (long long)dk * 64, dst + wave * 1024);
        const long long coff = (long long)dc0 * 2;
#pragma unroll
        for (int t = 0; t < XJ; ++t) {
            if (t == XJ - 1 && !full) break;
            glds16(cur[t] + coff, dst + BG_BN * 64 + (wave + 8 * t) * 1024);
        }
        ++dseq;
        ++dk;
        dc0 += 32;
        if (UP && dc0 == a.up_C) dnew = true;
        if (dc0 == a.Cin) {
            dc0 = 0;
            dtapbit <<= 1;
            dtapbase += tap_w;
            if (++dkw == a.KW) { dkw = 0; dtapbase += tap_h; }
            dnew = true;
        }
        if (dk == nk) {
            dk = 0; dtapbit = 1u; dc0 = 0; dkw = 0; dtapbase = 0; dnew = true;
            dtile += G;
            if (dtile < g.tiles) setup(dtile);
        }
    };
    // step s has landed (this wave's share) once at most the `behind` steps issued after it are in flight
    auto wait_landed = [&](int behind) {
        if (behind >= 2) {
            if (full) wait_vmcnt<2 * NGW>(); else wait_vmcnt<2 * NGW - 2>();
        } else if (behind == 1) {
            if (full) wait_vmcnt<NGW>(); else wait_vmcnt<NGW - 1>();
        } else {
            wait_vmcnt<0>();
        }
    };

    if (total_steps == 0) return;        // (never with grid <= tiles)
    setup(dtile);
    BG_STAMP(tp1);
#pragma nounroll
    for (int p = 0; p < BG_AHEAD; ++p)
        if (dseq < total_steps) issue_step();
    BG_STAMP(tp2);

    f32x4 acc[2][FM];
    const int fr_off = (lane & 15) * 64 + (((lane >> 4) ^ chunk_swz(lane >> 2)) << 4);
    const char* a_rd = smem + (wn * 32) * 64 + fr_off;
    const char* b_rd = smem + BG_BN * 64 + grp * (FM * 1024) + fr_off;
    const int q = lane >> 4, px = lane & 15;

    wait_landed(dseq - 1);               // step 0 landed (every wave's share) before anybody reads it
    __builtin_amdgcn_s_barrier();
    if (grp == 1) __builtin_amdgcn_s_barrier();      // the trailing half starts one phase later

    int cseq = 0;
#ifdef YDET_BIG_TRACE
    unsigned long long tr[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long tm[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    { BG_STAMP(ts1); tm[0] = ts1 - ts0; tm[7] = ts0; tr[7] = ((tp1 - ts0) << 32) | (tp2 - ts0); }
#endif
#pragma nounroll
    for (int ti = 0; ti < my_tiles; ++ti) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < FM; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

#pragma nounroll
        for (int k = 0; k < nk; ++k, ++cseq) {
            // ---- phase A: this step's fragments out of LDS, this wave's share of the refill ---------------------------
            BG_STAMP(t0);
            const int soff = (cseq & (BG_S - 1)) * BG_STAGE;
            bf16x8 fa[2], fb[FM];
#ifdef YDET_EXP_NO_READS      // timing experiments only (wrong results)
            if (k == 0) {
#endif
            fa[0] = *reinterpret_cast<const bf16x8*>(a_rd + soff);
            fa[1] = *reinterpret_cast<const bf16x8*>(a_rd + soff + 1024);
#pragma unroll
            for (int j = 0; j < FM; ++j) fb[j] = *reinterpret_cast<const bf16x8*>(b_rd + soff + j * 1024);
#ifdef YDET_EXP_NO_READS
            }
#endif
            asm volatile("" ::: "memory");
#ifdef YDET_BIG_TRACE      // two stamps that wait for nothing (read behind the lgkmcnt(0) below)
            unsigned long long tq0, tq1;
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("s_memtime %0" : "=s"(tq0)::"memory");
            __builtin_amdgcn_sched_barrier(0);
#endif
#ifdef YDET_EXP_NO_DMA
            if (dseq < total_steps) { if (dseq < 4) issue_step(); else { ++dseq; } }
#else
            if (dseq < total_steps) issue_step();
#endif
#ifdef YDET_BIG_TRACE
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("s_memtime %0" : "=s"(tq1)::"memory");
            __builtin_amdgcn_sched_barrier(0);
#endif
            BG_STAMP(t1);
#ifdef YDET_BIG_TRACE
            tm[5] += tq0 - t0; tm[6] += tq1 - tq0;      // issue of the LDS reads; issue of the refill
#endif
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            BG_STAMP(t2);
            // the step after this one: landed before the barrier that lets the leading half read it
            const int behind = dseq - (cseq + 2);
            if (grp == 1) wait_landed(behind);
            BG_STAMP(t3);
            __builtin_amdgcn_s_barrier();
            BG_STAMP(t4);
            // ---- phase B: the MFMAs -----------------------------------------------------------------------------------
            __builtin_amdgcn_s_setprio(1);
#ifdef YDET_EXP_NO_MFMA
            if (k == 0)
#endif
#pragma unroll
            for (int j = 0; j < FM; ++j) {
                acc[0][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[0], fb[j], acc[0][j], 0, 0, 0);
                acc[1][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[1], fb[j], acc[1][j], 0, 0, 0);
            }
            __builtin_amdgcn_s_setprio(0);
            BG_STAMP(t5);
            BG_ACC(0, t0, t1); BG_ACC(1, t1, t2); BG_ACC(2, t2, t3); BG_ACC(3, t3, t4); BG_ACC(4, t4, t5);
            if (k + 1 == nk && grp == 1) break;      // the trailing half finishes its tile inside this phase (below)
            if (grp == 0) wait_landed(behind);
            BG_STAMP(t6);
            __builtin_amdgcn_s_barrier();
            BG_STAMP(t7);
            BG_ACC(5, t5, t6); BG_ACC(6, t6, t7);
        }

#ifdef YDET_BIG_TRACE
        { BG_STAMP(tl); tm[1] = tl - tm[7]; }      // end of the (last) K loop since kernel start
#endif
        // ---- epilogue: bias, activation, residual, 16-byte pieces straight from registers -----------------------------
        {
            const int tile = bid + ti * G;
            const int tile_n = tile % g.tiles_n, tile_m = tile / g.tiles_n;
            const int n0 = tile_n * BG_BN;
            const long long m0 = (long long)tile_m * BMT + grp * (FM * 16) + px;
            const float* cst = reinterpret_cast<const float*>(smem + BG_S * BG_STAGE + (ti & 1) * BG_CST);
            const int cl = wn * 32 + 8 * q;          // first of this lane's eight channels inside the tile
            const int co = n0 + cl;
            with_epilogue_mode(a, [&](auto mode) {
                constexpr int MODE = decltype(mode)::value;
                constexpr bool NORES = MODE >= 0 && (MODE & 8) != 0;
#pragma unroll
                for (int j = 0; j < FM; ++j) {
                    const long long m = m0 + j * 16;
                    const bool okm = m < a.M;
                    float r[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                    if constexpr (!NORES) {
                        if (a.res && okm && co < a.Cout) {
                            const bf16x8 t = *reinterpret_cast<const bf16x8*>(a.res + ((size_t)m * a.res_ld + a.res_coff + co) * 2);
#pragma unroll
                            for (int e = 0; e < 8; ++e) r[e] = (float)t[e];
                        }
                    }
                    bf16x8 o;
#pragma unroll
                    for (int hf = 0; hf < 2; ++hf) {
                        float v[4] = {acc[hf][j][0], acc[hf][j][1], acc[hf][j][2], acc[hf][j][3]};
                        const float r4[4] = {r[4 * hf], r[4 * hf + 1], r[4 * hf + 2], r[4 * hf + 3]};
                        finish4<MODE>(v, r4, cst + cl + 4 * hf, cst + BG_BN + cl + 4 * hf, cst + 2 * BG_BN + cl + 4 * hf, a);
#pragma unroll
                        for (int e = 0; e < 4; ++e) o[4 * hf + e] = (__bf16)v[e];
                    }
                    if (okm && co < a.cout_store) {
                        char* dst = co >= a.y_split ? a.y2 + ((size_t)m * a.y2_ld + a.y2_coff + (co - a.y_split)) * 2
                                                    : a.y + ((size_t)m * a.y_ld + a.y_coff + co) * 2;
                        *reinterpret_cast<bf16x8*>(dst) = o;
                    }
                }
            });
        }
#ifdef YDET_BIG_TRACE
        { BG_STAMP(te); tm[2] = te - tm[7]; }      // end of the (last) epilogue since kernel start
#endif
        // the trailing half owes the barrier that ends its last phase B (not behind the workgroup's very last tile)
        if (grp == 1 && ti + 1 < my_tiles) {
            __builtin_amdgcn_s_barrier();
            ++cseq;
        }
    }
#ifdef YDET_BIG_TRACE
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    { BG_STAMP(tz); tm[3] = tz - tm[7]; tm[4] = (__builtin_amdgcn_s_memrealtime() - rt0); }   // stores acknowledged; 100 MHz ticks
    if (blockIdx.x == 0 && lane == 0)
        for (int i = 0; i < 8; ++i) { g_big_trace[wave * 8 + i] = tr[i]; g_big_trace[64 + wave * 8 + i] = tm[i]; }
#endif
}

struct Pick {
    int fm;
    int tiles;
};

// Pixel fragments per wave: the count whose tile grid wastes the fewest MFMA slots over whole rounds of `cus`
// workgroups (the two halves of a workgroup take turns on the matrix pipe, so a K step costs 2 FM fragments; a tile
// pays a fixed prologue / epilogue share on top).
Pick pick_fm(long long M, int tiles_n, int nk, int cus) {
    static const char* forced = getenv("YDET_BIG_FM");
    if (forced) {
        const int fm = atoi(forced);
        if (fm >= BG_FM_MIN && fm <= BG_FM_MAX) return Pick{fm, (int)((M + 32 * fm - 1) / (32 * fm)) * tiles_n};
    }
    Pick best{0, 0};
    double best_cost = 0.0;
    for (int fm = BG_FM_MIN; fm <= BG_FM_MAX; ++fm) {
        const long long tiles = (M + 32 * fm - 1) / (32 * fm) * tiles_n;
        const long long rounds = (tiles + cus - 1) / cus;
        const double cost = (double)rounds * fm * (nk + 6.0);
        if (best.fm == 0 || cost < best_cost - 1e-9 || (cost < best_cost + 1e-9 && fm > best.fm)) {
            best = Pick{fm, (int)tiles};
            best_cost = cost;
        }
    }
    return best;
}

int device_cus() {
    static int cus = 0;
    if (cus == 0) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) cus = prop.multiProcessorCount;
        if (cus <= 0) cus = 256;
    }
    return cus;
}

bool shape_ok(const ConvArgs& a) {
    return a.Cout >= 128 && a.Cin % 32 == 0 && a.Cin <= 2048 && a.KH * a.KW <= 32 && a.kp / 32 >= 4 && a.pw_w == nullptr &&
           a.y_ld % 8 == 0 && a.y_coff % 8 == 0 && a.cout_store % 8 == 0 &&
           (!a.res || (a.res_ld % 8 == 0 && a.res_coff % 8 == 0)) && (!a.y2 || (a.y2_ld % 8 == 0 && a.y2_coff % 8 == 0));
}

template <int FM>
int launch_fm(const ConvArgs& a, const BigArgs& g, int grid, hipStream_t stream) {
    static const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_big_kernel<FM, false>),
                                                       hipFuncAttributeMaxDynamicSharedMemorySize, BG_LDS);
    if (attr != hipSuccess) return fail(YDET_ERR_LAUNCH, "conv: cannot raise the dynamic LDS limit: %s", hipGetErrorString(attr));
    hipLaunchKernelGGL((conv_big_kernel<FM, false>), dim3((unsigned)grid), dim3(64 * BG_NW), BG_LDS, stream, a, g);
    YDET_CHECK_LAUNCH("conv_big_kernel");
    return YDET_OK;
}

}  // namespace

bool conv_big_takes(const ConvArgs& a) {
    // Opt-in (YDET_BIG=1).  Measured on MI355X inside YOLOv5s 640 at batch 64: level with the 128 x 256 tile of
    // conv.hip on the 3x3 layers with K >= 1152 (43-47 us both), 5-15 % behind it on K = 512 .. 576 and on layers with
    // several tiles per CU -- one workgroup per CU means nothing runs beside its prologue (5-6 us until the first
    // operands have landed) and its store-bound epilogue (106 KB per CU at ~10 B/clk), and a tile's stores hold up the
    // counted waits of the next tile's first K steps.
    static const char* on = getenv("YDET_BIG");
    if (!on || on[0] == '0' || !shape_ok(a) || a.up_C) return false;   // (the upsampled-source form is not built)
    const int tiles_n = (a.Cout + BG_BN - 1) / BG_BN;
    const int cus = device_cus();
    const Pick p = pick_fm(a.M, tiles_n, a.kp / 32, cus);
    // only when the K loop dominates the launch and the grid fills the chip (three quarters of the CUs busy in the
    // last round)
    static const char* force = getenv("YDET_BIG_ALWAYS");
    if (a.kp / 32 < 16 && !force) return false;
    const int rounds = (p.tiles + cus - 1) / cus;
    return p.tiles >= (rounds - 1) * cus + (3 * cus) / 4;
}

int launch_conv_big(const ConvArgs& a0, hipStream_t stream) {
    ConvArgs a = a0;
    BigArgs g;
    g.tiles_n = (a.Cout + BG_BN - 1) / BG_BN;
    a.tiles_n = g.tiles_n;
    const int cus = device_cus();
    const Pick p = pick_fm(a.M, g.tiles_n, a.kp / 32, cus);
    g.tiles = p.tiles;
    if (g.tiles <= 0) return fail(YDET_ERR_ARG, "conv: empty tile grid");
    const int grid = g.tiles < cus ? g.tiles : cus;
    switch (p.fm) {
        case 6: return launch_fm<6>(a, g, grid, stream);
        case 7: return launch_fm<7>(a, g, grid, stream);
        case 8: return launch_fm<8>(a, g, grid, stream);
        case 9: return launch_fm<9>(a, g, grid, stream);
        case 10: return launch_fm<10>(a, g, grid, stream);
        case 11: return launch_fm<11>(a, g, grid, stream);
        case 12: return launch_fm<12>(a, g, grid, stream);
        case 13: return launch_fm<13>(a, g, grid, stream);
        default: return launch_fm<14>(a, g, grid, stream);
    }
}

}  // namespace conv
}  // namespace ydet

#ifdef YDET_BIG_TRACE
extern "C" int ydet_debug_big_trace(unsigned long long* host64) {
    return (int)hipMemcpyFromSymbol(host64, HIP_SYMBOL(ydet::conv::g_big_trace), sizeof(unsigned long long) * 128);
}
#endif
