// 3x3 / stride 1 / pad 1 convolution over a map with 64 or 128 input channels, for gfx950: bf16 in, bf16 out.
//
// The tiled kernel of conv.hip gathers every tap of every K step from global memory: a pixel's row crosses the
// L2 -> LDS path nine times, and with 128 output channels the narrow tile makes that path, not the matrix pipe, the
// bound.  Here a workgroup owns TH whole rows of one image and 128 output channels.  The (TH + 2) x (W + 2) patch of input
// pixels under its outputs is laid down in LDS ONCE (out-of-image pixels as zeros: the padding), and the nine taps of
// all K steps read their pixel operand from there at a per-tap offset; only the weights stream through an LDS-DMA ring.
//
//  * Patch layout: planes of 8 channels, patch[plane][pixel] of 16-byte elements, each plane padded to a multiple of 256
//    bytes.  A lane's MFMA B fragment (pixel l & 15, channels 8 (l >> 4) .. + 7 of the 32-channel K step) is ONE element:
//    the sixteen lanes a ds_read_b128 serves per cycle read sixteen consecutive pixels of one or two planes whose bases
//    differ by whole bank rows, so fragment reads are conflict-free for every tap offset (a pitch-per-pixel layout
//    cannot be: the lane groups of ds_read_b128 mix two channel groups).
//  * Eight waves: four across the channels (32 each: two A fragments) x two across the pixels (FMW fragments each).
//  * K loop over (tap, 64-channel half): weights through a three-stage ring of 128 rows x 128 bytes (the LDS image of
//    conv.hip: swizzled 16-byte chunks, rows permuted so that a lane ends with 16-byte pieces of its pixels' output
//    rows), software-pipelined like conv.hip's pipelined tile: the fragments of sub-step u + 1 are fetched while the MFMAs
//    of sub-step u run, one barrier per 64 channels of K.
//    The A fragments ping-pong between two register sets; the B fragments have one set: fragment j of the next sub-step
//    is fetched right behind the two MFMAs that consumed fragment j of this one.
//  * Persistent workgroups, one per CU (the patch and the ring fill its LDS).  Every CU reaches its fill and its epilogue
//    at about the same time, so whatever a tile needs from memory is requested under the K loop of the tile before: the
//    next patch and the residual pieces wait in registers, the ring runs on across tile boundaries.
//  * K order: tap-major, channel-minor -- the order of the tiled kernel, so results equal its results bit for bit.
#include <stdlib.h>

#include "conv_common.h"

using namespace ydet;

namespace ydet {
namespace conv {
namespace {

constexpr int PT_BN = 128;              // output channels per workgroup
constexpr int PT_NW = 8;                // waves per workgroup
constexpr int PT_STAGE = PT_BN * 128;   // bytes of one ring stage: 128 weight rows x 64 channels of K
constexpr int PT_NST = 3;
constexpr int PT_NPASS = 12;            // most patch passes (64 pixels x 8 planes each) a workgroup makes per tile
constexpr int PT_PATCH_OFF = PT_NST * PT_STAGE + 2 * 3 * PT_BN * (int)sizeof(float);   // ring, two sets of constants, patch

struct PatchArgs {
    int TH;            // output rows per workgroup
    int tiles_h;       // row tiles per image
    int tiles_n;       // channel tiles
    int PW;            // patch width in pixels (W + 2)
    int NPP;           // patch pixels (TH + 2) * PW
    int NPPpad;        // ... rounded up to 16 (a plane is a whole number of 256-byte bank rows)
    float r_pw, r_w;   // 1 / PW, 1 / W
    long long tiles;   // tiles of the launch
    long long res_span;   // bytes from the residual's first element to the end of its last pixel's row
};

// Debugging aid (tools/build_variant.sh <name> conv_patch - -DYDET_PATCH_TRACE): cycle counts of the eight waves of
// workgroup 0 over its first tile.  A stamp costs a scalar-memory round trip, so they sit where the wave has drained its
// LDS reads anyway.
#ifdef YDET_PATCH_TRACE
__device__ unsigned long long g_ptrace[8 * 16];
#define PT_NOW() __builtin_readcyclecounter()
#else
#define PT_NOW() 0ull
#endif

template <int I, int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, N>(f);
    }
}

// A persistent workgroup per CU walks tiles id, id + grid, ...  While the K loop of one tile runs, the patch of the next
// is fetched into registers (one or two 16-byte loads per K step, behind the weight stages of the ring) and the ring runs
// on into the next tile's first stages; between two K loops a workgroup only writes those registers to LDS and stores its
// outputs.
template <int CIN, int FMW>
__global__ __launch_bounds__(64 * PT_NW, 2) void conv_patch_kernel(const ConvArgs a, const PatchArgs g) {
    constexpr int KH64 = CIN / 64;        // 64-channel halves per tap = ring stages per tap
    constexpr int NKS = 9 * KH64;         // ring stages of the whole K
    constexpr int FN = 2;                 // channel fragments per wave
    constexpr int LPT = 2;                // LDS-DMA instructions per wave and ring stage
    constexpr int PPS = (PT_NPASS + NKS - 1) / NKS;       // patch loads per wave and K step
    static_assert(NKS % PT_NST == 0, "a tile starts at ring stage 0");
    auto patch_loads_at = [](int ks) constexpr { return ks * PPS + PPS <= PT_NPASS ? PPS : ks * PPS < PT_NPASS ? PT_NPASS - ks * PPS : 0; };
    extern __shared__ __attribute__((aligned(256))) char smem[];
    char* ring = smem;
    float* cst2 = reinterpret_cast<float*>(smem + PT_NST * PT_STAGE);               // [2][3][PT_BN]: by tile parity
    char* patch = smem + PT_PATCH_OFF;
    static_assert(PT_PATCH_OFF % 256 == 0, "patch planes start on a bank row");

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wn = wave & 3, wm = wave >> 2;
    const int px = lane & 15, q = lane >> 4;

    // tile id -> (channel tile, image, first row): consecutive ids share the patch (channel tiles) and neighbouring rows
    auto decode = [&](unsigned id, int& n0, int& img, int& r0) {
        const unsigned tile_n = id % (unsigned)g.tiles_n;
        id /= (unsigned)g.tiles_n;
        img = (int)(id / (unsigned)g.tiles_h);
        r0 = (int)(id - (unsigned)img * (unsigned)g.tiles_h) * g.TH;
        n0 = (int)tile_n * PT_BN;
    };
    const unsigned first = xcd_remap(blockIdx.x, gridDim.x);
    int n0, img, r0;
    decode(first, n0, img, r0);

    const char* xb = a.x + (size_t)a.x_coff * 2;
    const __amdgpu_buffer_rsrc_t rsrc_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(xb), 0, (int)(unsigned)a.x_span, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrc_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(a.w), 0, (int)(unsigned)a.w_span, 0x00020000);

    // ---- weight ring: staging roles (the 128-byte-row image of conv.hip) -------------------------------------------------
    // LDS-DMA instruction j of a wave fills rows (j * 8 + wave) * 8 .. + 7 of a stage; lane l lands at physical chunk l % 8
    // of row l / 8 and fetches logical chunk (l % 8) ^ ((row & 15) >> 1).  Row 16 g + r of the stage holds channel
    // 32 (g >> 1) + 8 (r >> 2) + 4 (g & 1) + (r & 3): a wave's accumulator rows are then eight consecutive channels per lane.
    const int srow = lane >> 3;
    const int r16 = (wave & 1) * 8 + srow;
    const int lchunk = (lane & 7) ^ ((r16 >> 1) & 7);
    const unsigned woff_lane = (unsigned)(((8 * (r16 >> 2) + (r16 & 3)) * a.kp + lchunk * 8) * 2);
    unsigned wuni[LPT];
#pragma unroll
    for (int j = 0; j < LPT; ++j) {
        const int grp = (j * PT_NW + wave) >> 1;          // group of sixteen rows
        wuni[j] = (unsigned)((32 * (grp >> 1) + 4 * (grp & 1)) * a.kp * 2);
    }
    auto issue = [&](int stage, int ks, int n0_) {        // ring stage <- K values [64 ks, 64 ks + 64) of rows n0_ .. + 127
        char* dst = ring + stage * PT_STAGE + wave * 1024;
        const unsigned base = (unsigned)n0_ * (unsigned)a.kp * 2u + (unsigned)ks * 128u;
#pragma unroll
        for (int j = 0; j < LPT; ++j)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_w, (__attribute__((address_space(3))) void*)(dst + j * (PT_NW * 1024)), 16,
                                                     (int)woff_lane, (int)(wuni[j] + base), 0, 0);
    };

    // ---- patch: 16-byte pieces global -> registers -> LDS ----------------------------------------------------------------
    // One pass of the workgroup covers 64 pixels x 8 planes: a wave takes eight pixels, lane l pixel l & 7, plane l >> 3
    // (each pixel's eight pieces are one whole 128-byte line; measured: fetching sixteen pixels x four planes per wave, which
    // spares the LDS writes their bank conflicts, is slower on balance).  A pass that has nothing to fetch still issues its load (out of range: zeros, no traffic):
    // the waits of the K loop count instructions.
    const int npass = ((g.NPPpad + 63) >> 6) * KH64;
    int lane_t = lane;                                    // (re-derived per tile: keeps the sixteen addresses out of registers)
    auto patch_load = [&](int it, int img_, int r0_, bool live) -> u32x4 {
        const int blk = KH64 == 1 ? it : it >> 1, half = KH64 == 1 ? 0 : it & 1;
        const int pp = blk * 64 + wave * 8 + (lane_t & 7);
        const int pl = half * 8 + (lane_t >> 3);
        const int pr = div_small(pp, g.PW, g.r_pw), pc = pp - pr * g.PW;
        const int row = r0_ - 1 + pr, col = pc - 1;
        const bool ok = live && it < npass && pp < g.NPP && (unsigned)row < (unsigned)a.H && (unsigned)col < (unsigned)a.W;
        const unsigned off = (unsigned)((((long long)img_ * a.H * a.W + (long long)row * a.W + col) * a.x_ld + pl * 8) * 2);
        return __builtin_amdgcn_raw_buffer_load_b128(rsrc_x, ok ? off : 0xFFFFFFF0u, 0, 0);
    };
    auto patch_store = [&](int it, const u32x4& v) {
        const int blk = KH64 == 1 ? it : it >> 1, half = KH64 == 1 ? 0 : it & 1;
        const int pp = blk * 64 + wave * 8 + (lane_t & 7);
        const int pl = half * 8 + (lane_t >> 3);
        if (it < npass && pp < g.NPPpad) *reinterpret_cast<u32x4*>(patch + ((size_t)pl * g.NPPpad + pp) * 16) = v;
    };
    auto cst_load = [&](int n0_, float (&c)[3]) {
        c[0] = a.bias ? a.bias[n0_ + tid] : 0.f;
        c[1] = a.post_scale ? a.post_scale[n0_ + tid] : 0.f;
        c[2] = a.post_scale ? a.post_shift[n0_ + tid] : 0.f;
    };
    auto cst_store = [&](int par, const float (&c)[3]) {
        float* d = cst2 + par * 3 * PT_BN;
        d[tid] = c[0];
        d[PT_BN + tid] = c[1];
        d[2 * PT_BN + tid] = c[2];
    };

    // ---- per-lane operand addresses (the same for every tile) ---------------------------------------------------------------
    // B: fragment j of this wave = output pixels (wm * FMW + j) * 16 + px of the tile, in row-major order; its element for
    // tap (0, 0) and plane q sits at (q * NPPpad + orow * PW + ocol) * 16 (the patch starts one pixel up and left).
    unsigned pb[FMW];
#pragma unroll
    for (int j = 0; j < FMW; ++j) {
        int m = (wm * FMW + j) * 16 + px;
        if (m >= g.TH * a.W) m = 0;                         // (never stored)
        const int orow = div_small(m, a.W, g.r_w), ocol = m - orow * a.W;
        pb[j] = (unsigned)((q * g.NPPpad + orow * g.PW + ocol) * 16);
    }
    // A: rows wn * 32 + 16 i + px of the stage, logical chunk 4 kk + q
    int fa_off[2];
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) fa_off[kk] = (wn * 32 + px) * 128 + (((4 * kk + q) ^ ((px >> 1) & 7)) << 4);

    // ---- first tile: ring stages 0 .. NST - 1, patch and constants ----------------------------------------------------------
    u32x4 pv[PT_NPASS];
    float c3[3] = {0.f, 0.f, 0.f};
#pragma unroll
    for (int p = 0; p < PT_NST; ++p) issue(p, p, n0);
#pragma unroll
    for (int it = 0; it < PT_NPASS; ++it) pv[it] = patch_load(it, img, r0, true);
    if (tid < PT_BN) cst_load(n0, c3);
    wait_vmcnt<0>();
#pragma unroll
    for (int it = 0; it < PT_NPASS; ++it) patch_store(it, pv[it]);
    if (tid < PT_BN) cst_store(0, c3);
    __syncthreads();

    // Fragments of sub-step (ks, kk): A from ring stage ks % NST (two sets, ping-ponged), B from the patch at the tap's
    // offset, planes 8 (ks % KH64) + 4 kk + q.  B has ONE set: fragment j of the next sub-step is fetched right behind the two
    // MFMAs that consumed fragment j of this one, fourteen or ten MFMAs ahead of its own use.
    bf16x8 fa[2][FN], fb[FMW];
    auto read_a = [&](int ks, int kk, bf16x8 (&A)[FN]) {
        const char* ws = ring + (ks % PT_NST) * PT_STAGE + fa_off[kk];
#pragma unroll
        for (int i = 0; i < FN; ++i) A[i] = *reinterpret_cast<const bf16x8*>(ws + i * 16 * 128);
    };
    auto read_b = [&](int ks, int kk, int j) -> bf16x8 {
        const int tap = KH64 == 1 ? ks : ks >> 1, half = KH64 == 1 ? 0 : ks & 1;
        const int kh = tap >= 6 ? 2 : tap >= 3 ? 1 : 0, kw = tap - 3 * kh;
        const unsigned u = (unsigned)(((half * 8 + kk * 4) * g.NPPpad + kh * g.PW + kw) * 16);
        return *reinterpret_cast<const bf16x8*>(patch + pb[j] + u);
    };

    // residual pieces: fetched through a buffer whose extent is zero when the layer has none (the loads then return zeros
    // without traffic: the K loop's waits count instructions, so the instruction stream is the same either way)
    const __amdgpu_buffer_rsrc_t rsrc_r = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<char*>(a.res ? a.res + (size_t)a.res_coff * 2 : a.x), 0, a.res ? (int)(unsigned)g.res_span : 0, 0x00020000);
    // one residual piece per K step, over the steps NKS - 1 - FMW .. NKS - 2
    auto res_loads_at = [](int ks) constexpr { return ks >= NKS - 1 - FMW && ks <= NKS - 2 ? 1 : 0; };
    static_assert(NKS - 1 - FMW >= 0, "a K step per residual piece");

    const unsigned tiles = (unsigned)g.tiles;
    int par = 0;
    for (unsigned id = first; id < tiles; id += gridDim.x, par ^= 1) {
        const bool has_next = id + gridDim.x < tiles;
        int n0n = n0, imgn = img, r0n = r0;
        if (has_next) decode(id + gridDim.x, n0n, imgn, r0n);
        // (ahead of this tile's ring refills: older than everything the K loop waits for)
        if (has_next && tid < PT_BN) cst_load(n0n, c3);

        f32x4 acc[FN][FMW];
#pragma unroll
        for (int i = 0; i < FN; ++i)
#pragma unroll
            for (int j = 0; j < FMW; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

        asm volatile("" : "+v"(lane_t));
        [[maybe_unused]] const unsigned long long tr_t0 = PT_NOW();
        [[maybe_unused]] unsigned long long tr_last = tr_t0, tr_work = 0, tr_vm = 0, tr_bar = 0;
        const int rows = min(g.TH, a.H - r0);                 // output rows of this tile
        const int npix = rows * a.W;                          // output pixels of this tile
        const int cl = wn * 32 + 8 * q;                       // first of the lane's eight channels inside the tile
        const int co = n0 + cl;
        const unsigned gm0 = (unsigned)(((long long)img * a.H + r0) * a.W);              // rows of a tile are consecutive pixels
        u32x4 rr[FMW];
        read_a(0, 0, fa[0]);
#pragma unroll
        for (int j = 0; j < FMW; ++j) fb[j] = read_b(0, 0, j);
        // sub-step (ks, kk): the MFMAs of its fragments with the reads of the next sub-step's between them
        auto substep = [&](auto par_, auto more, auto ks_, auto kk_) {
            constexpr int P = decltype(par_)::value;
            constexpr bool MORE = decltype(more)::value;
            constexpr int ks = decltype(ks_)::value;
            constexpr int kk = decltype(kk_)::value;
            if constexpr (kk == 1) {
                // Every wave holds the last A fragments of stage ks (the B reads behind them may still be out; before the last
                // barrier of a tile a wave drains those too: the patch is rewritten after it).  Stage ks + 1 has landed: it left
                // three steps ago, and a wave has since issued the patch and residual loads of two steps and one more stage (at
                // steps 0 and 1 everything older was awaited between the tiles).
                constexpr int VM = LPT + patch_loads_at(ks - 2) + res_loads_at(ks - 2) + patch_loads_at(ks - 1) + res_loads_at(ks - 1);
#ifdef YDET_PATCH_TRACE
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                const unsigned long long ta = PT_NOW();
                if constexpr (ks >= 2) wait_vmcnt<VM>();
                const unsigned long long tb = PT_NOW();
                __builtin_amdgcn_s_barrier();
                const unsigned long long tc = PT_NOW();
                tr_work += ta - tr_last; tr_vm += tb - ta; tr_bar += tc - tb; tr_last = tc;
#else
                if constexpr (ks >= 2) wait_vmcnt<VM>();
                if constexpr (ks == NKS - 1) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                else wait_lgkmcnt<FMW>();
                __builtin_amdgcn_s_barrier();
#endif
                asm volatile("" ::: "memory");
                if constexpr (ks + PT_NST < NKS) issue(ks % PT_NST, ks + PT_NST, n0);
                else issue(ks % PT_NST, ks + PT_NST - NKS, n0n);
                static_for<0, PPS>([&](auto p_) {
                    constexpr int it = ks * PPS + decltype(p_)::value;
                    if constexpr (it < PT_NPASS) pv[it] = patch_load(it, imgn, r0n, has_next);
                });
                if constexpr (res_loads_at(ks) != 0) {
                    constexpr int j = ks - (NKS - 1 - FMW);
                    const int m = (wm * FMW + j) * 16 + px;
                    const unsigned off = ((gm0 + (unsigned)m) * (unsigned)a.res_ld + (unsigned)co) * 2u;
                    rr[j] = __builtin_amdgcn_raw_buffer_load_b128(rsrc_r, m < npix && co < a.Cout ? off : 0xFFFFFFF0u, 0, 0);
                }
            }
            if constexpr (MORE) {
                if constexpr (kk == 0) read_a(ks, 1, fa[P ^ 1]);
                else read_a(ks + 1, 0, fa[P ^ 1]);
            }
#pragma unroll
            for (int j = 0; j < FMW; ++j) {
#pragma unroll
                for (int i = 0; i < FN; ++i) acc[i][j] = mma(fa[P][i], fb[j], acc[i][j]);
                if constexpr (MORE) fb[j] = kk == 0 ? read_b(ks, 1, j) : read_b(ks + 1, 0, j);
            }
            if constexpr (kk == 1) __builtin_amdgcn_sched_group_barrier(0x020, LPT + patch_loads_at(ks) + res_loads_at(ks), 0);
            if constexpr (MORE) __builtin_amdgcn_sched_group_barrier(0x100, FN, 0);
#pragma unroll
            for (int j = 0; j < FMW; ++j) {
                __builtin_amdgcn_sched_group_barrier(0x008, FN, 0);
                if constexpr (MORE) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            }
        };
        constexpr std::integral_constant<int, 0> I0{};
        constexpr std::integral_constant<int, 1> I1{};
        static_for<0, NKS - 1>([&](auto ks_) {
            substep(I0, std::true_type{}, ks_, I0);
            substep(I1, std::true_type{}, ks_, I1);
        });
        substep(I0, std::true_type{}, std::integral_constant<int, NKS - 1>{}, I0);
        substep(I1, std::false_type{}, std::integral_constant<int, NKS - 1>{}, I1);

        // ---- between two K loops: the next tile's patch and constants into LDS (no wave reads this patch any more: the
        //      last fragments were in registers before the last barrier) ------------------------------------------------------
        [[maybe_unused]] const unsigned long long tr_k = PT_NOW();
        wait_vmcnt<0>();
        [[maybe_unused]] const unsigned long long tr_v = PT_NOW();
        asm volatile("" : "+v"(lane_t));
        if (has_next) {
#pragma unroll
            for (int it = 0; it < PT_NPASS; ++it) patch_store(it, pv[it]);
            if (tid < PT_BN) cst_store(par ^ 1, c3);
        }

        // ---- epilogue: bias, activation, residual; 16-byte pieces straight from registers -------------------------------------
        [[maybe_unused]] const unsigned long long tr_p = PT_NOW();
        const float* cst = cst2 + par * 3 * PT_BN;
        with_epilogue_mode(a, [&](auto mode) {
            constexpr int MODE = decltype(mode)::value;
            constexpr bool NORES = MODE >= 0 && (MODE & 8) != 0;
            // (the residual pieces arrived under the K loop)
#pragma unroll
            for (int j = 0; j < FMW; ++j) {
                float r[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                if constexpr (!NORES) {
                    const bf16x8 t = __builtin_bit_cast(bf16x8, rr[j]);
#pragma unroll
                    for (int e = 0; e < 8; ++e) r[e] = (float)t[e];
                }
                bf16x8 o;
#pragma unroll
                for (int hf = 0; hf < 2; ++hf) {
                    float v[4] = {acc[hf][j][0], acc[hf][j][1], acc[hf][j][2], acc[hf][j][3]};
                    const float r4[4] = {r[4 * hf], r[4 * hf + 1], r[4 * hf + 2], r[4 * hf + 3]};
                    finish4<MODE>(v, r4, cst + cl + 4 * hf, cst + PT_BN + cl + 4 * hf, cst + 2 * PT_BN + cl + 4 * hf, a);
#pragma unroll
                    for (int e = 0; e < 4; ++e) o[4 * hf + e] = (__bf16)v[e];
                }
                // (measured: holding the pieces back and storing them under the next tile's K loop costs more than it hides)
                const int m = (wm * FMW + j) * 16 + px;
                if (m < npix && co < a.cout_store)
                    *reinterpret_cast<bf16x8*>(a.y + ((size_t)(gm0 + (unsigned)m) * a.y_ld + a.y_coff + co) * 2) = o;
            }
        });
#ifdef YDET_PATCH_TRACE
        const unsigned long long tr_e = PT_NOW();
        __syncthreads();
        const unsigned long long tr_s = PT_NOW();
        if (blockIdx.x == 0 && id == first && lane == 0) {
            unsigned long long* o = g_ptrace + wave * 16;
            o[0] = tr_work; o[1] = tr_vm; o[2] = tr_bar; o[3] = tr_k - tr_t0; o[4] = tr_v - tr_k; o[5] = tr_p - tr_v;
            o[6] = tr_e - tr_p; o[7] = tr_s - tr_e;
        }
#else
        __syncthreads();                 // the next patch and constants are in LDS
#endif
        n0 = n0n; img = imgn; r0 = r0n;
    }
    wait_vmcnt<0>();                     // (the ring ran on past the last tile)
}

long long res_span_bytes(const ConvArgs& a) { return a.res ? ((long long)a.N * a.H * a.W * a.res_ld - a.res_coff) * 2 : 0; }

struct PatchPlan {
    int fmw, TH;
    size_t lds;
    long long tiles;
};

// Tile height and pixel fragments per wave (7 or 5) for a map of width W: as many whole rows as fit the 2 x FMW sixteen-pixel
// fragments of the two rows of waves and the LDS of a CU; of the two forms the one with the lower cost rounds x FMW
// (a tile costs its fragment slots whether they are used or not; rounds of 256 workgroups, one per CU).
bool plan_patch(const ConvArgs& a, PatchPlan& p) {
    const int PW = a.W + 2;
    long long best = -1;
    for (int fmw : {7, 5}) {
        int TH = (32 * fmw) / a.W;
        if (TH > a.H) TH = a.H;
        for (; TH >= 1; --TH) {
            const int npad = ((TH + 2) * PW + 15) / 16 * 16;
            const size_t lds = PT_PATCH_OFF + (size_t)(a.Cin / 8) * npad * 16;
            if (lds > 160 * 1024 || (npad + 63) / 64 * (a.Cin / 64) > PT_NPASS) continue;
            const long long tiles = (long long)a.N * ((a.H + TH - 1) / TH) * ((a.Cout + PT_BN - 1) / PT_BN);
            const long long cost = (tiles + 255) / 256 * fmw;
            if (best < 0 || cost < best) {
                best = cost;
                p.fmw = fmw; p.TH = TH; p.lds = lds; p.tiles = tiles;
            }
            break;
        }
    }
    return best >= 0;
}

template <int CIN, int FMW>
int launch_patch_t(const ConvArgs& a, const PatchPlan& p, hipStream_t stream) {
    PatchArgs g;
    g.TH = p.TH;
    g.tiles_h = (a.H + p.TH - 1) / p.TH;
    g.tiles_n = (a.Cout + PT_BN - 1) / PT_BN;
    g.PW = a.W + 2;
    g.NPP = (p.TH + 2) * g.PW;
    g.NPPpad = (g.NPP + 15) / 16 * 16;
    g.r_pw = 1.0f / (float)g.PW;
    g.r_w = 1.0f / (float)a.W;
    g.tiles = p.tiles;
    g.res_span = res_span_bytes(a);
    static unsigned char optin[kMaxDevices];
    const int rc = raise_dynamic_lds_limit(reinterpret_cast<const void*>(&conv_patch_kernel<CIN, FMW>), 160 * 1024, optin, "conv_patch");
    if (rc != YDET_OK) return rc;
    const long long cus = current_device_info().cus;
    hipLaunchKernelGGL((conv_patch_kernel<CIN, FMW>), dim3((unsigned)std::min(p.tiles, cus)), dim3(64 * PT_NW), p.lds, stream, a, g);
    YDET_CHECK_LAUNCH("conv_patch_kernel");
    return YDET_OK;
}

}  // namespace

bool conv_patch_takes(const ConvArgs& a) {
    static const char* sw = getenv("YDET_PATCH");        // YDET_PATCH=0: these layers stay with the tiled kernel
    if (sw && sw[0] == '0') return false;
    if (!(a.KH == 3 && a.KW == 3 && a.stride == 1 && a.pad == 1 && a.groups <= 1 && a.dilation <= 1 && (a.Cin == 64 || a.Cin == 128) &&
          a.kp == 9 * a.Cin && a.Cout >= 96 && a.pw_w == nullptr && a.up_C == 0 && a.splits <= 1 && a.y2 == nullptr &&
          a.x_ld % 8 == 0 && a.x_coff % 8 == 0 && a.y_ld % 8 == 0 && a.y_coff % 8 == 0 && a.cout_store % 8 == 0 &&
          (!a.res || (a.res_ld % 8 == 0 && a.res_coff % 8 == 0)) && a.x_span > 0 && a.x_span < 0xFFFFFFF0LL &&
          a.w_span < 0x7FFFFFFFLL && res_span_bytes(a) < 0xFFFFFFF0LL && a.W <= 224 && (long long)a.N * a.H * a.W < (1LL << 31)))
        return false;
    PatchPlan p;
    // (a launch needs enough tiles to fill the chip: below that the tiled kernel's finer tiles serve better)
    return plan_patch(a, p) && p.tiles >= 256;
}

int launch_conv_patch(const ConvArgs& a, hipStream_t stream) {
    PatchPlan p;
    if (!plan_patch(a, p)) return fail(YDET_ERR_ARG, "conv_patch: no tile fits this shape");
    if (kernel_probe().on) {
        snprintf(kernel_probe().name, sizeof(kernel_probe().name), "conv_patch<c%d>", a.Cin);
        return YDET_OK;
    }
    if (a.Cin == 64) return p.fmw == 7 ? launch_patch_t<64, 7>(a, p, stream) : launch_patch_t<64, 5>(a, p, stream);
    return p.fmw == 7 ? launch_patch_t<128, 7>(a, p, stream) : launch_patch_t<128, 5>(a, p, stream);
}

}  // namespace conv
}  // namespace ydet

#ifdef YDET_PATCH_TRACE
extern "C" int ydet_debug_patch_trace(unsigned long long* out) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(ydet::conv::g_ptrace), sizeof(unsigned long long) * 8 * 16);
}
#endif
