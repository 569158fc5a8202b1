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
struct V5Args {
    const char* head[kMaxLevels];
    int ny[kMaxLevels], nx[kMaxLevels], ld[kMaxLevels];
    int box_off[kMaxLevels + 1];  // first output row of each level; [nl] = rows per image
    int pix_off[kMaxLevels + 1];  // first pixel of each level; [nl] = pixels per image
    float stride[kMaxLevels];
    float anchor[kMaxLevels][2 * kMaxAnchors];
    int nl, N, na, no, transposed;
    float* out;
    float* row_max;       // optional [N][rows per image]
    long long total_pix;  // N * pixels per image
};

// One wave decodes whole pixels: the na*no head channels of a pixel are contiguous in the NHWC map, lane l holds
// channels l, l+64, ... in registers (KR registers cover na*no <= 64*KR), so reads are full contiguous runs and the
// na output rows of the pixel leave as contiguous runs of `no` floats.  With the pixel in registers, the per-box
// maximum of class score x objectness is a masked wave reduction.
template <typename T, int KR, int PPW>
__global__ __launch_bounds__(256) void decode_v5_kernel(const V5Args a) {
    const int lane = threadIdx.x & 63;
    const long long wave = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int CH = a.na * a.no;
    const int pix_per_img = a.pix_off[a.nl], rows_per_img = a.box_off[a.nl];

    int an_k[KR], o_k[KR];   // anchor and output column of channel k*64 + lane (pixel independent)
#pragma unroll
    for (int k = 0; k < KR; ++k) {
        const int ch = k * 64 + lane;
        int an = 0;
        while (an + 1 < a.na && ch >= (an + 1) * a.no) ++an;
        an_k[k] = an;
        o_k[k] = ch < CH ? ch - an * a.no : -1;
    }

    const T* hp[PPW];
    int lv[PPW], cellv[PPW], nv[PPW];
    float raw[PPW][KR];
#pragma unroll
    for (int i = 0; i < PPW; ++i) {
        const long long g = wave * PPW + i;
        lv[i] = -1;
        if (g >= a.total_pix) continue;
        const int n = (int)(g / pix_per_img);
        const int pi = (int)(g - (long long)n * pix_per_img);
        int l = 0;
        while (l + 1 < a.nl && pi >= a.pix_off[l + 1]) ++l;
        const int cell = pi - a.pix_off[l];
        lv[i] = l; cellv[i] = cell; nv[i] = n;
        hp[i] = reinterpret_cast<const T*>(a.head[l]) + ((long long)n * a.ny[l] * a.nx[l] + cell) * a.ld[l];
#pragma unroll
        for (int k = 0; k < KR; ++k) raw[i][k] = o_k[k] >= 0 ? to_f32(hp[i][k * 64 + lane]) : 0.f;
    }

#pragma unroll
    for (int i = 0; i < PPW; ++i) {
        const int l = lv[i];
        if (l < 0) continue;   // wave-uniform
        const int cell = cellv[i];
        const int nx = a.nx[l], ny = a.ny[l];
        const int row = cell / nx, col = cell - row * nx;
        // "transposed": the np.meshgrid(arange(ny), arange(nx)) table of shape (nx, ny) re-read as (ny, nx):
        // flat cell f holds (f / ny, f % ny) -- (row, col) on square maps.
        const float gx = a.transposed ? (float)(cell / ny) : (float)col;
        const float gy = a.transposed ? (float)(cell % ny) : (float)row;
        const float stride = a.stride[l];
        const long long orow0 = (long long)nv[i] * rows_per_img + a.box_off[l] + cell;   // + an*ny*nx
        float v[KR];
#pragma unroll
        for (int k = 0; k < KR; ++k) {
            const int o = o_k[k];
            if (o < 0) { v[k] = 0.f; continue; }
            const int an = an_k[k];
            const float s = sigmoidf_(raw[i][k]);
            float val = s;
            if (o < 2) {
                val = (s * 2.0f - 0.5f + (o == 0 ? gx : gy)) * stride;
            } else if (o < 4) {
                const float t = s * 2.0f;
                val = t * t * a.anchor[l][2 * an + (o - 2)];
            }
            v[k] = val;
            a.out[(orow0 + (long long)an * ny * nx) * a.no + o] = val;
        }
        if (a.row_max) {
            for (int an = 0; an < a.na; ++an) {
                const int ca = an * a.no + 4;
                float t = 0.f;
#pragma unroll
                for (int k = 0; k < KR; ++k) t = (ca >> 6) == k ? v[k] : t;
                const float obj = __shfl(t, ca & 63);
                float m = -INFINITY;
#pragma unroll
                for (int k = 0; k < KR; ++k)
                    if (an_k[k] == an && o_k[k] >= 5) m = fmaxf(m, v[k] * obj);
#pragma unroll
                for (int sft = 32; sft >= 1; sft >>= 1) m = fmaxf(m, __shfl_xor(m, sft));
                if (lane == 0) a.row_max[orow0 + (long long)an * ny * nx] = m;
            }
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

}  // namespace

extern "C" int ydet_decode_v5(const ydet_v5_level* levels, int nl, int N, int na, int no, int grid_transposed,
                              int head_dtype, float* out, float* row_max, void* stream) {
    clear_error();
    YDET_CHECK_ARG(levels && out, "decode_v5: null pointer");
    YDET_CHECK_ARG(nl >= 1 && nl <= kMaxLevels, "decode_v5: nl=%d (1..%d)", nl, kMaxLevels);
    YDET_CHECK_ARG(na >= 1 && na <= kMaxAnchors, "decode_v5: na=%d (1..%d)", na, kMaxAnchors);
    YDET_CHECK_ARG(N > 0 && no >= 5, "decode_v5: N=%d no=%d", N, no);
    YDET_CHECK_ARG((long long)na * no <= 1024, "decode_v5: na*no=%lld exceeds 1024", (long long)na * no);
    YDET_CHECK_ARG(dtype_ok(head_dtype), "decode_v5: bad dtype");
    V5Args a;
    a.nl = nl; a.N = N; a.na = na; a.no = no; a.transposed = grid_transposed ? 1 : 0; a.out = out; a.row_max = row_max;
    long long off = 0, pix = 0;
    for (int l = 0; l < nl; ++l) {
        const ydet_v5_level& L = levels[l];
        YDET_CHECK_ARG(L.head && L.ny > 0 && L.nx > 0 && L.ld >= na * no, "decode_v5: level %d shape", l);
        a.head[l] = (const char*)L.head; a.ny[l] = L.ny; a.nx[l] = L.nx; a.ld[l] = L.ld; a.stride[l] = L.stride;
        for (int i = 0; i < 2 * kMaxAnchors; ++i) a.anchor[l][i] = i < 2 * na ? L.anchor_wh[i] : 0.f;
        a.box_off[l] = (int)off;
        a.pix_off[l] = (int)pix;
        off += (long long)na * L.ny * L.nx;
        pix += (long long)L.ny * L.nx;
        YDET_CHECK_ARG(off < (1LL << 30), "decode_v5: too many boxes per image");
    }
    for (int l = nl; l <= kMaxLevels; ++l) { a.box_off[l] = (int)off; a.pix_off[l] = (int)pix; }
    a.total_pix = (long long)N * pix;
    const bool small = na * no <= 256;
    const int ppw = small ? 4 : 1;
    const long long waves = (a.total_pix + ppw - 1) / ppw;
    const long long grid = (waves + 3) / 4;
    YDET_CHECK_ARG(grid > 0 && grid < 0x7fffffffLL, "decode_v5: grid");
    hipStream_t s = as_stream(stream);
    if (head_dtype == YDET_BF16) {
        if (small) hipLaunchKernelGGL((decode_v5_kernel<__bf16, 4, 4>), dim3((unsigned)grid), dim3(256), 0, s, a);
        else hipLaunchKernelGGL((decode_v5_kernel<__bf16, 16, 1>), dim3((unsigned)grid), dim3(256), 0, s, a);
    } else {
        if (small) hipLaunchKernelGGL((decode_v5_kernel<float, 4, 4>), dim3((unsigned)grid), dim3(256), 0, s, a);
        else hipLaunchKernelGGL((decode_v5_kernel<float, 16, 1>), dim3((unsigned)grid), dim3(256), 0, s, a);
    }
    YDET_CHECK_LAUNCH("decode_v5_kernel");
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
    const long long grid = (a.total + 255) / 256;
    YDET_CHECK_ARG(grid > 0 && grid < 0x7fffffffLL, "decode_v2v3: grid");
    if (head_dtype == YDET_BF16)
        hipLaunchKernelGGL(decode_fa_kernel<__bf16>, dim3((unsigned)grid), dim3(256), 0, as_stream(stream), a);
    else
        hipLaunchKernelGGL(decode_fa_kernel<float>, dim3((unsigned)grid), dim3(256), 0, as_stream(stream), a);
    YDET_CHECK_LAUNCH("decode_fa_kernel");
    return YDET_OK;
}
