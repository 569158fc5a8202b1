// Box decode kernels (gfx950).  Both are HBM-bound: every head value is read once and every decoded
// value written once.  Compiled with -ffp-contract=off so that the arithmetic keeps the operation
// order of the Python expressions it replaces.
#include "ydet_internal.h"

using namespace ydet;

namespace {

constexpr int kMaxLevels = 4;
constexpr int kMaxAnchors = 8;

__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }
// --------------------------------------------------------------------------------------------------
// YOLOv5 Detect decode.
// --------------------------------------------------------------------------------------------------
struct V5LevelArgs {   // one launch per pyramid level
    const char* head;   // NHWC [N][ny*nx][ld]
    float* out;         // first output row of this level in image 0
    float* row_max;     // same for the optional row maxima (or null)
    int ny, nx, ld, hw; // hw = ny*nx
    int na, no, transposed;
    long long rows_per_img;
    float stride, rd;   // rd = 1/nx (1/ny when transposed)
    float anchor[2 * kMaxAnchors];
};

// One wave decodes whole pixels, anchor by anchor: the `no` channels of one (pixel, anchor) are contiguous in the
// NHWC head map and form one contiguous output row, lane l holds columns l, l+64, ... (OR registers cover
// no <= 64*OR), so loads and stores are contiguous runs, the column index of a lane never changes, and the per-box
// maximum of class score x objectness is one masked wave reduction.  PPW pixels x NA anchors are loaded before the
// first is processed (NA = 0: any anchor count, one anchor's loads in flight at a time).
template <typename T, int OR, int PPW, int NA>
__global__ __launch_bounds__(256) void decode_v5_kernel(const V5LevelArgs a) {
    typedef const T __attribute__((address_space(1)))* in_ptr;      // global address space: no flat instructions
    typedef float __attribute__((address_space(1)))* out_ptr;
    constexpr int NAR = NA > 0 ? NA : 1;
    const int lane = threadIdx.x & 63;
    const int wave = blockIdx.x * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int n = blockIdx.y;
    const int na = NA > 0 ? NA : a.na;
    const int no = a.no;
    const bool col_ok[OR > 1 ? 2 : 1] = {lane < no, OR > 1 ? 64 + lane < no : false};   // registers 0 and 1

    in_ptr hp[PPW];
    out_ptr op[PPW], mp[PPW];
    bool live[PPW];
    float g_l[PPW];   // grid offset of this lane's column (lane 0: x, lane 1: y)
#pragma unroll
    for (int i = 0; i < PPW; ++i) {
        const int cell = wave * PPW + i;
        live[i] = cell < a.hw;
        const int c = live[i] ? cell : 0;
        hp[i] = (in_ptr)(a.head) + ((long long)n * a.hw + c) * a.ld;
        const long long orow = (long long)n * a.rows_per_img + c;
        op[i] = (out_ptr)(a.out) + orow * no;
        mp[i] = (out_ptr)(a.row_max) + orow;
        // grid offsets: (col, row) of the cell, or -- "transposed" -- the np.meshgrid(arange(ny), arange(nx)) table of
        // shape (nx, ny) re-read as (ny, nx): flat cell f holds (f / ny, f % ny), which is (row, col) on square maps.
        const int d = a.transposed ? a.ny : a.nx;
        const int q = div_small(c, d, a.rd);
        const int rem = c - q * d;
        const float gx = a.transposed ? (float)q : (float)rem;
        const float gy = a.transposed ? (float)rem : (float)q;
        g_l[i] = lane == 0 ? gx : gy;
    }

    auto load = [&](int i, int an, float (&raw)[OR]) {
#pragma unroll
        for (int r = 0; r < OR; ++r) {
            const int o = r * 64 + lane;
            const bool ok = r < 2 ? col_ok[r < 2 ? r : 0] : o < no;
            raw[r] = ok ? to_f32(hp[i][an * no + o]) : 0.f;
        }
    };
    auto finish = [&](int i, int an, const float (&raw)[OR]) {
        out_ptr ob = op[i] + (long long)an * a.hw * no;
        float v[OR];
#pragma unroll
        for (int r = 0; r < OR; ++r) v[r] = sigmoid_fast(raw[r]);
        {   // columns 0..3 (lanes 0..3 of register 0): box centre and size
            const float s = v[0];
            const float t = s * 2.0f;
            const float centre = (t - 0.5f + g_l[i]) * a.stride;
            const float size = t * t * a.anchor[2 * an + (lane & 1)];
            v[0] = lane < 2 ? centre : lane < 4 ? size : s;
        }
#pragma unroll
        for (int r = 0; r < OR; ++r) {
            const int o = r * 64 + lane;
            const bool ok = r < 2 ? col_ok[r < 2 ? r : 0] : o < no;
            if (ok) ob[o] = v[r];
        }
        if (a.row_max) {
            const float obj = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v[0]), 4));
            float m = lane >= 5 && col_ok[0] ? v[0] * obj : -INFINITY;
#pragma unroll
            for (int r = 1; r < OR; ++r) {
                const bool ok = r < 2 ? col_ok[r < 2 ? r : 0] : r * 64 + lane < no;
                if (ok) m = fmaxf(m, v[r] * obj);
            }
            m = wave_max(m);
            if (lane == 0) mp[i][(long long)an * a.hw] = m;
        }
    };

    if (NA > 0) {
        float raw[PPW][NAR][OR];
#pragma unroll
        for (int i = 0; i < PPW; ++i)
#pragma unroll
            for (int an = 0; an < NAR; ++an) load(i, an, raw[i][an]);
#pragma unroll
        for (int i = 0; i < PPW; ++i) {
            if (!live[i]) continue;   // wave-uniform
#pragma unroll
            for (int an = 0; an < NAR; ++an) finish(i, an, raw[i][an]);
        }
    } else {
        for (int an = 0; an < na; ++an) {
            float raw[PPW][OR];
#pragma unroll
            for (int i = 0; i < PPW; ++i) load(i, an, raw[i]);
#pragma unroll
            for (int i = 0; i < PPW; ++i)
                if (live[i]) finish(i, an, raw[i]);
        }
    }
}

// --------------------------------------------------------------------------------------------------
// v2/v3/v4 head split + decode.
// --------------------------------------------------------------------------------------------------
struct FAArgs {
    const char* head[kMaxLevels];
    int H[kMaxLevels], W[kMaxLevels], ld[kMaxLevels];
    int box_off[kMaxLevels + 1];
    float anchor[kMaxLevels][2 * kMaxAnchors];
    int nl, N, A, nc, mode, transposed;
    float mult, input_size;
    float* boxes;
    float* scores;
    long long total;  // N * boxes per image
};

template <typename T>
__global__ __launch_bounds__(256) void decode_fa_kernel(const FAArgs a) {
    const long long gid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= a.total) return;
    const int per_img = a.box_off[a.nl];
    const int n = (int)(gid / per_img);
    const int b = (int)(gid - (long long)n * per_img);
    int l = 0;
    while (l + 1 < a.nl && b >= a.box_off[l + 1]) ++l;
    const int lb = b - a.box_off[l];
    const int cell = lb / a.A, an = lb - cell * a.A;
    const int row = cell / a.W[l], col = cell - row * a.W[l];
    const T* px = reinterpret_cast<const T*>(a.head[l]) + ((size_t)n * a.H[l] * a.W[l] + cell) * a.ld[l];

    const float conf = sigmoidf_(to_f32(px[an]));
    // softmax over the nc class logits of this anchor, times objectness
    const T* cl = px + a.A + an * a.nc;
    float mx = -INFINITY;
    for (int c = 0; c < a.nc; ++c) mx = fmaxf(mx, to_f32(cl[c]));
    float sum = 0.f;
    for (int c = 0; c < a.nc; ++c) sum += expf(to_f32(cl[c]) - mx);
    float* sc = a.scores + (size_t)gid * a.nc;
    for (int c = 0; c < a.nc; ++c) sc[c] = expf(to_f32(cl[c]) - mx) / sum * conf;

    const T* t = px + a.A * (1 + a.nc) + an * 4;
    // reference grid: np.meshgrid(arange(H), arange(W)) stacked and flattened: cell k holds (k / H, k % H)
    const float gx = a.transposed ? (float)(cell / a.H[l]) : (float)col;
    const float gy = a.transposed ? (float)(cell % a.H[l]) : (float)row;
    const float aw = a.anchor[l][2 * an], ah = a.anchor[l][2 * an + 1];
    float cx, cy, w, h;
    if (a.mode == 0) {  // xywh_to_xxyy: everything scaled by mult
        cx = (sigmoidf_(to_f32(t[0])) + gx) * a.mult;
        cy = (sigmoidf_(to_f32(t[1])) + gy) * a.mult;
        w = expf(to_f32(t[2])) * aw * a.mult;
        h = expf(to_f32(t[3])) * ah * a.mult;
    } else {  // xywh_to_xxyyv3: only the centre is scaled
        cx = (sigmoidf_(to_f32(t[0])) + gx) * a.mult;
        cy = (sigmoidf_(to_f32(t[1])) + gy) * a.mult;
        w = expf(to_f32(t[2])) * aw;
        h = expf(to_f32(t[3])) * ah;
    }
    float4 bb;
    bb.x = fminf(fmaxf((cx - w / 2.0f) / a.input_size, 0.f), 1.f);
    bb.y = fminf(fmaxf((cy - h / 2.0f) / a.input_size, 0.f), 1.f);
    bb.z = fminf(fmaxf((cx + w / 2.0f) / a.input_size, 0.f), 1.f);
    bb.w = fminf(fmaxf((cy + h / 2.0f) / a.input_size, 0.f), 1.f);
    reinterpret_cast<float4*>(a.boxes)[gid] = bb;
}

// The same decode with coalesced traffic: a workgroup owns `cells` consecutive cells of one level of one image.  Their
// head rows are one contiguous run of the NHWC map and their A boxes each one contiguous run of both outputs, so the
// rows are staged through LDS in 16-byte pieces, every thread decodes boxes out of LDS (the arithmetic above, value for
// value), and the scores leave through LDS as contiguous pieces.  One thread per box read 25 scattered values and wrote
// 20 strided ones: a wave's load touched some fifty cache lines per instruction.
struct FATileArgs {
    FAArgs f;
    int level, cells, tiles;     // tiles per image of this level
};

template <typename T>
__global__ __launch_bounds__(256) void decode_fa_tiled_kernel(const FATileArgs ta) {
    extern __shared__ __attribute__((aligned(16))) float fa_lds[];
    const FAArgs& a = ta.f;
    const int l = ta.level, n = blockIdx.y;
    const int hw = a.H[l] * a.W[l], ld = a.ld[l];
    const int cell0 = blockIdx.x * ta.cells;
    const int nc_ = min(ta.cells, hw - cell0);           // cells of this tile
    float* rows = fa_lds;                                 // [cells][ld]
    float* outs = fa_lds + (size_t)ta.cells * ld;         // [cells * A][nc]
    const T* src = reinterpret_cast<const T*>(a.head[l]) + ((size_t)n * hw + cell0) * ld;
    for (int i = threadIdx.x; i < nc_ * ld; i += 256) rows[i] = to_f32(src[i]);
    __syncthreads();
    const int nb = nc_ * a.A;
    const size_t box0 = (size_t)n * a.box_off[a.nl] + a.box_off[l] + (size_t)cell0 * a.A;
    for (int b = threadIdx.x; b < nb; b += 256) {
        const int lc = b / a.A, an = b - lc * a.A;
        const int cell = cell0 + lc;
        const int row = cell / a.W[l], col = cell - row * a.W[l];
        const float* px = rows + (size_t)lc * ld;
        const float conf = sigmoidf_(px[an]);
        const float* cl = px + a.A + an * a.nc;
        float mx = -INFINITY;
        for (int c = 0; c < a.nc; ++c) mx = fmaxf(mx, cl[c]);
        float sum = 0.f;
        for (int c = 0; c < a.nc; ++c) sum += expf(cl[c] - mx);
        float* sc = outs + (size_t)b * a.nc;
        for (int c = 0; c < a.nc; ++c) sc[c] = expf(cl[c] - mx) / sum * conf;
        const float* t = px + a.A * (1 + a.nc) + an * 4;
        const float gx = a.transposed ? (float)(cell / a.H[l]) : (float)col;
        const float gy = a.transposed ? (float)(cell % a.H[l]) : (float)row;
        const float aw = a.anchor[l][2 * an], ah = a.anchor[l][2 * an + 1];
        float cx, cy, w, h;
        if (a.mode == 0) {
            cx = (sigmoidf_(t[0]) + gx) * a.mult;
            cy = (sigmoidf_(t[1]) + gy) * a.mult;
            w = expf(t[2]) * aw * a.mult;
            h = expf(t[3]) * ah * a.mult;
        } else {
            cx = (sigmoidf_(t[0]) + gx) * a.mult;
            cy = (sigmoidf_(t[1]) + gy) * a.mult;
            w = expf(t[2]) * aw;
            h = expf(t[3]) * ah;
        }
        float4 bb;
        bb.x = fminf(fmaxf((cx - w / 2.0f) / a.input_size, 0.f), 1.f);
        bb.y = fminf(fmaxf((cy - h / 2.0f) / a.input_size, 0.f), 1.f);
        bb.z = fminf(fmaxf((cx + w / 2.0f) / a.input_size, 0.f), 1.f);
        bb.w = fminf(fmaxf((cy + h / 2.0f) / a.input_size, 0.f), 1.f);
        reinterpret_cast<float4*>(a.boxes)[box0 + b] = bb;
    }
    __syncthreads();
    float* dst = a.scores + box0 * a.nc;
    for (int i = threadIdx.x; i < nb * a.nc; i += 256) dst[i] = outs[i];
}

}  // namespace

extern "C" int ydet_decode_v5(const ydet_v5_level* levels, int nl, int N, int na, int no, int grid_transposed,
                              int head_dtype, float* out, float* row_max, void* stream) {
    clear_error();
    YDET_CHECK_ARG(levels && out, "decode_v5: null pointer");
    YDET_CHECK_ARG(nl >= 1 && nl <= kMaxLevels, "decode_v5: nl=%d (1..%d)", nl, kMaxLevels);
    YDET_CHECK_ARG(na >= 1 && na <= kMaxAnchors, "decode_v5: na=%d (1..%d)", na, kMaxAnchors);
    YDET_CHECK_ARG(N > 0 && N <= 65535 && no >= 5, "decode_v5: N=%d no=%d", N, no);
    YDET_CHECK_ARG(no <= 1024, "decode_v5: no=%d exceeds 1024", no);
    YDET_CHECK_ARG(dtype_ok(head_dtype), "decode_v5: bad dtype");
    long long rows = 0;
    for (int l = 0; l < nl; ++l) {
        const ydet_v5_level& L = levels[l];
        YDET_CHECK_ARG(L.head && L.ny > 0 && L.nx > 0 && L.ld >= na * no, "decode_v5: level %d shape", l);
        YDET_CHECK_ARG((long long)L.ny * L.nx < (1 << 22), "decode_v5: level %d has too many cells", l);
        rows += (long long)na * L.ny * L.nx;
        YDET_CHECK_ARG(rows < (1LL << 30), "decode_v5: too many boxes per image");
    }
    hipStream_t s = as_stream(stream);
    long long off = 0;
    for (int l = 0; l < nl; ++l) {
        const ydet_v5_level& L = levels[l];
        V5LevelArgs a;
        a.head = (const char*)L.head;
        a.out = out + off * no;
        a.row_max = row_max ? row_max + off : nullptr;
        a.ny = L.ny; a.nx = L.nx; a.ld = L.ld; a.hw = L.ny * L.nx;
        a.na = na; a.no = no; a.transposed = grid_transposed ? 1 : 0;
        a.rows_per_img = rows;
        a.stride = L.stride;
        a.rd = 1.0f / (float)(a.transposed ? L.ny : L.nx);
        for (int i = 0; i < 2 * kMaxAnchors; ++i) a.anchor[i] = i < 2 * na ? L.anchor_wh[i] : 0.f;
        off += (long long)na * a.hw;
        const bool small = no <= 128;
        const int ppw = small ? 4 : 1;
        const dim3 grid((unsigned)((a.hw + 4 * ppw - 1) / (4 * ppw)), (unsigned)N);
        if (head_dtype == YDET_BF16) {
            if (small && na == 3) hipLaunchKernelGGL((decode_v5_kernel<__bf16, 2, 4, 3>), grid, dim3(256), 0, s, a);
            else if (small) hipLaunchKernelGGL((decode_v5_kernel<__bf16, 2, 4, 0>), grid, dim3(256), 0, s, a);
            else hipLaunchKernelGGL((decode_v5_kernel<__bf16, 16, 1, 0>), grid, dim3(256), 0, s, a);
        } else {
            if (small && na == 3) hipLaunchKernelGGL((decode_v5_kernel<float, 2, 4, 3>), grid, dim3(256), 0, s, a);
            else if (small) hipLaunchKernelGGL((decode_v5_kernel<float, 2, 4, 0>), grid, dim3(256), 0, s, a);
            else hipLaunchKernelGGL((decode_v5_kernel<float, 16, 1, 0>), grid, dim3(256), 0, s, a);
        }
        YDET_CHECK_LAUNCH("decode_v5_kernel");
    }
    return YDET_OK;
}

extern "C" int ydet_decode_v2v3(const ydet_fa_level* levels, int nl, int N, int A, int nc, int mode, float mult,
                                float input_size, int grid_transposed, int head_dtype, float* boxes,
                                float* scores, void* stream) {
    clear_error();
    YDET_CHECK_ARG(levels && boxes && scores, "decode_v2v3: null pointer");
    YDET_CHECK_ARG(nl >= 1 && nl <= kMaxLevels, "decode_v2v3: nl=%d", nl);
    YDET_CHECK_ARG(A >= 1 && A <= kMaxAnchors && nc >= 1 && N > 0, "decode_v2v3: A=%d nc=%d N=%d", A, nc, N);
    YDET_CHECK_ARG(mode == 0 || mode == 1, "decode_v2v3: mode");
    YDET_CHECK_ARG(dtype_ok(head_dtype), "decode_v2v3: bad dtype");
    YDET_CHECK_ARG(((uintptr_t)boxes & 15) == 0, "decode_v2v3: boxes must be 16-byte aligned");
    FAArgs a;
    a.nl = nl; a.N = N; a.A = A; a.nc = nc; a.mode = mode; a.transposed = grid_transposed ? 1 : 0;
    a.mult = mult; a.input_size = input_size; a.boxes = boxes; a.scores = scores;
    long long off = 0;
    for (int l = 0; l < nl; ++l) {
        const ydet_fa_level& L = levels[l];
        YDET_CHECK_ARG(L.head && L.H > 0 && L.W > 0 && L.ld >= A * (5 + nc), "decode_v2v3: level %d shape", l);
        a.head[l] = (const char*)L.head; a.H[l] = L.H; a.W[l] = L.W; a.ld[l] = L.ld;
        for (int i = 0; i < 2 * kMaxAnchors; ++i) a.anchor[l][i] = i < 2 * A ? L.anchor_wh[i] : 0.f;
        a.box_off[l] = (int)off;
        off += (long long)A * L.H * L.W;
        YDET_CHECK_ARG(off < (1LL << 30), "decode_v2v3: too many boxes per image");
    }
    for (int l = nl; l <= kMaxLevels; ++l) a.box_off[l] = (int)off;
    a.total = (long long)N * off;
    // tiled form: one launch per level, as many cells per workgroup as 48 KB of LDS hold (rows + scores), at most 64
    bool tiled = N <= 65535;
    int cells[kMaxLevels];
    for (int l = 0; l < nl && tiled; ++l) {
        const size_t per_cell = ((size_t)a.ld[l] + (size_t)A * nc) * sizeof(float);
        cells[l] = (int)((48 * 1024) / per_cell);
        if (cells[l] > 64) cells[l] = 64;
        if (cells[l] < 8) tiled = false;
    }
    if (tiled) {
        for (int l = 0; l < nl; ++l) {
            FATileArgs ta;
            ta.f = a; ta.level = l; ta.cells = cells[l];
            ta.tiles = (a.H[l] * a.W[l] + cells[l] - 1) / cells[l];
            const size_t lds = (size_t)cells[l] * ((size_t)a.ld[l] + (size_t)A * nc) * sizeof(float);
            const dim3 grid((unsigned)ta.tiles, (unsigned)N);
            if (head_dtype == YDET_BF16)
                hipLaunchKernelGGL(decode_fa_tiled_kernel<__bf16>, grid, dim3(256), lds, as_stream(stream), ta);
            else
                hipLaunchKernelGGL(decode_fa_tiled_kernel<float>, grid, dim3(256), lds, as_stream(stream), ta);
            YDET_CHECK_LAUNCH("decode_fa_tiled_kernel");
        }
        return YDET_OK;
    }
    const long long grid = (a.total + 255) / 256;
    YDET_CHECK_ARG(grid > 0 && grid < 0x7fffffffLL, "decode_v2v3: grid");
    if (head_dtype == YDET_BF16)
        hipLaunchKernelGGL(decode_fa_kernel<__bf16>, dim3((unsigned)grid), dim3(256), 0, as_stream(stream), a);
    else
        hipLaunchKernelGGL(decode_fa_kernel<float>, dim3((unsigned)grid), dim3(256), 0, as_stream(stream), a);
    YDET_CHECK_LAUNCH("decode_fa_kernel");
    return YDET_OK;
}
