// The alternative NMS flavours of yolov5_official/utils.py for gfx950:
//   non_max_suppression2      (:318-460)  per-class suppression by one of seven methods
//   diou_non_max_suppression  (:463-548)  Cluster-Weighted DIoU-NMS over the max_box best candidates of an image
// Compiled with -ffp-contract=off: every IoU / DIoU expression is written in the operation order of the reference's
// tensor expressions and must round like them (no fused multiply-add).
//
// non_max_suppression2 as five launches:
//   nms2_hist     every (box, class) candidate -- obj > conf, 2 < w, h < 4096, cls*obj > conf, and in best-class mode
//                 the box's first-maximum class only -- counted into a 2048-bin score histogram per (image, class)
//   nms2_thresh   per (image, class): the lowest bin that still belongs to the `cap` best candidates
//   nms2_collect  candidates at or above that bin appended to the (image, class) list as 64-bit keys
//   nms2_class    one workgroup per (image, class): rank sort, first `cap` rows, the method's sequential loop with the
//                 IoUs of the current head against all remaining rows taken in parallel
//   nms2_merge    per image: the classes' rows, ascending class, written out in descending score order
//   (+ nms2_order, the rank of every output row, spread over the grid).
#include "ydet_internal.h"

using namespace ydet;

namespace {

typedef unsigned long long u64;

__device__ __forceinline__ unsigned f2key(float f) {
    const unsigned u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float key2f(unsigned k) {
    return __uint_as_float((k & 0x80000000u) ? (k & 0x7FFFFFFFu) : ~k);
}
static inline unsigned f2key_host(float f) {
    union { float f; unsigned u; } cv;
    cv.f = f;
    return (cv.u & 0x80000000u) ? ~cv.u : (cv.u | 0x80000000u);
}

constexpr int kBins = 2048;
constexpr int kListCap = 2048;     // candidates of one (image, class) that reach the sort
constexpr int kClassCap = 500;     // yolov5_official/utils.py:390-391

enum { M_HARD_BATCH = 0, M_HARD = 1, M_HARD_MYSELF = 2, M_AND = 3, M_MERGE = 4, M_SOFT = 5, M_DIOU = 6 };

struct Nms2Args {
    const float* pred;       // [N][nboxes][no]
    int N, nboxes, no, nc;
    float conf, thr;
    int multi, method, cap;  // cap: rows of a class that enter the method (500, or the list capacity for hard_nms_batch)
    unsigned base16;
    // workspace
    unsigned* hist;          // [N][nc][kBins]
    unsigned* thr_bin;       // [N][nc]
    unsigned* cursor;        // [N][nc]
    unsigned* img_max;       // [N] orderable key of the largest candidate coordinate (hard_nms_batch)
    unsigned* flags;         // [0]: some list overflowed
    u64* list;               // [N][nc][kListCap]
    float* cls_rows;         // [N][nc][cap][6]
    unsigned* cls_count;     // [N][nc]
    float* tmp_rows;         // [N][nc*cap][6] ascending class
    // outputs
    float* out;              // [N][out_cap][6]
    int32_t* out_count;      // [N]
    int out_cap;
};

__device__ __forceinline__ unsigned score_bin(float s, unsigned base16) {
    const unsigned b = (f2key(s) >> 16) - base16;
    return b > (unsigned)(kBins - 1) ? (unsigned)(kBins - 1) : b;
}

// calls visit(class, score) for every candidate of box `p` (a row of the prediction tensor)
template <typename F>
__device__ __forceinline__ void nms2_visit_box(const Nms2Args& a, const float* p, F visit) {
    const float obj = p[4], w = p[2], h = p[3];
    if (!(obj > a.conf)) return;                                     // :336
    if (!(w > 2.f && h > 2.f && w < 4096.f && h < 4096.f)) return;  // :339
    if (a.multi) {                                                   // :352-359
        for (int c = 0; c < a.nc; ++c) {
            const float s = p[5 + c] * obj;
            if (s > a.conf) visit(c, s);
        }
    } else {                                                         // :360-362 (first maximum)
        float best = p[5] * obj;
        int bc = 0;
        for (int c = 1; c < a.nc; ++c) {
            const float s = p[5 + c] * obj;
            if (s > best) { best = s; bc = c; }
        }
        if (best > a.conf) visit(bc, best);
    }
}

__global__ __launch_bounds__(256) void nms2_hist_kernel(const Nms2Args a) {
    const long long gid = (long long)blockIdx.x * 256 + threadIdx.x;
    if (gid >= (long long)a.N * a.nboxes) return;
    const int img = (int)(gid / a.nboxes);
    const float* p = a.pred + (size_t)gid * a.no;
    bool any = false;
    nms2_visit_box(a, p, [&](int c, float s) {
        atomicAdd(&a.hist[((size_t)img * a.nc + c) * kBins + score_bin(s, a.base16)], 1u);
        any = true;
    });
    if (any && a.method == M_HARD_BATCH) {
        const float cx = p[0], cy = p[1], w = p[2], h = p[3];
        const float m = fmaxf(fmaxf(cx - w / 2.f, cy - h / 2.f), fmaxf(cx + w / 2.f, cy + h / 2.f));
        atomicMax(&a.img_max[img], f2key(m));
    }
}

// one wave per (image, class): lowest bin b with count(bins >= b) >= cap (0 when there are fewer candidates)
__global__ __launch_bounds__(64) void nms2_thresh_kernel(const Nms2Args a) {
    const int ic = blockIdx.x;
    const unsigned* h = a.hist + (size_t)ic * kBins;
    const int lane = threadIdx.x;
    // lane l owns bins [32 l, 32 l + 32); suffix sums over lanes, then inside the lane that crosses `cap`
    unsigned mine = 0;
    for (int b = 0; b < 32; ++b) mine += h[lane * 32 + b];
    unsigned above = 0;   // candidates in higher lanes
    for (int l = 63; l > 0; --l) {
        const unsigned v = __shfl(mine, l);
        if (lane < l) above += v;
    }
    const bool crosses = above < (unsigned)a.cap && above + mine >= (unsigned)a.cap;
    const unsigned long long vote = __ballot(crosses);
    unsigned result = 0;
    if (vote) {
        const int src = __ffsll((long long)vote) - 1;
        unsigned b_found = 0;
        if (lane == src) {
            unsigned acc = above;
            int b = 31;
            for (; b > 0; --b) {
                acc += h[lane * 32 + b];
                if (acc >= (unsigned)a.cap) break;
            }
            b_found = (unsigned)(lane * 32 + b);
        }
        result = __shfl(b_found, src);
    }
    if (lane == 0) {
        a.thr_bin[ic] = result;
        // hard_nms_batch has no per-class limit in the reference: more candidates than the list holds cannot be served
        if (a.method == M_HARD_BATCH && above + mine > (unsigned)kListCap) a.flags[0] = 1u;
    }
}

__global__ __launch_bounds__(256) void nms2_collect_kernel(const Nms2Args a) {
    const long long gid = (long long)blockIdx.x * 256 + threadIdx.x;
    if (gid >= (long long)a.N * a.nboxes) return;
    const int img = (int)(gid / a.nboxes);
    const unsigned box = (unsigned)(gid - (long long)img * a.nboxes);
    const float* p = a.pred + (size_t)gid * a.no;
    nms2_visit_box(a, p, [&](int c, float s) {
        const size_t ic = (size_t)img * a.nc + c;
        if (score_bin(s, a.base16) < a.thr_bin[ic]) return;
        const unsigned pos = atomicAdd(&a.cursor[ic], 1u);
        if (pos < (unsigned)kListCap) a.list[ic * kListCap + pos] = ((u64)f2key(s) << 32) | (u64)(0xFFFFFFFFu - box);
        else a.flags[0] = 1u;
    });
}

// IoU of yolov5_official/utils.py:288-296 (bbox_iou2, x1y1x2y2) and its DIoU form (:297-309)
__device__ __forceinline__ float iou2(const float4& b1, const float4& b2, bool diou) {
    const float iw = fmaxf(fminf(b1.z, b2.z) - fmaxf(b1.x, b2.x), 0.f);
    const float ih = fmaxf(fminf(b1.w, b2.w) - fmaxf(b1.y, b2.y), 0.f);
    const float inter = iw * ih;
    const float w1 = b1.z - b1.x, h1 = b1.w - b1.y, w2 = b2.z - b2.x, h2 = b2.w - b2.y;
    const float uni = (w1 * h1 + 1e-16f) + w2 * h2 - inter;
    const float iou = inter / uni;
    if (!diou) return iou;
    const float cw = fmaxf(b1.z, b2.z) - fminf(b1.x, b2.x), ch = fmaxf(b1.w, b2.w) - fminf(b1.y, b2.y);
    const float c2 = cw * cw + ch * ch + 1e-16f;
    const float dx = (b2.x + b2.z) - (b1.x + b1.z), dy = (b2.y + b2.w) - (b1.y + b1.w);
    const float rho2 = dx * dx / 4.f + dy * dy / 4.f;
    return iou - rho2 / c2;
}
// the overlap torchvision's nms thresholds (boxes may carry a per-class offset)
__device__ __forceinline__ float iou_tv(const float4& a, const float4& b) {
    const float iw = fmaxf(fminf(a.z, b.z) - fmaxf(a.x, b.x), 0.f), ih = fmaxf(fminf(a.w, b.w) - fmaxf(a.y, b.y), 0.f);
    const float inter = iw * ih;
    return inter / ((a.z - a.x) * (a.w - a.y) + (b.z - b.x) * (b.w - b.y) - inter);
}

constexpr int kClsThreads = 256;

// sum over the workgroup in a fixed order (tree over threads); every thread gets the result
__device__ __forceinline__ float block_sum(float v, float* red) {
    const int t = threadIdx.x;
    red[t] = v;
    __syncthreads();
    for (int s = kClsThreads / 2; s > 0; s >>= 1) {
        if (t < s) red[t] = red[t] + red[t + s];
        __syncthreads();
    }
    const float r = red[0];
    __syncthreads();
    return r;
}

__global__ __launch_bounds__(kClsThreads) void nms2_class_kernel(const Nms2Args a) {
    __shared__ u64 keys[kListCap];            // the list; once ranked, its memory holds the two `alive` index arrays
    __shared__ float4 box[kListCap];
    __shared__ float score[kListCap];
    __shared__ float red[kClsThreads];
    unsigned short (*alive)[kListCap] = reinterpret_cast<unsigned short (*)[kListCap]>(keys);
    __shared__ unsigned s_cnt[kClsThreads / 64 + 1];
    const int ic = blockIdx.x, img = ic / a.nc, c = ic - img * a.nc;
    const int t = threadIdx.x;
    const int n = (int)min(a.cursor[ic], (unsigned)kListCap);
    float* out = a.cls_rows + (size_t)ic * a.cap * 6;
    if (n == 0) {
        if (t == 0) a.cls_count[ic] = 0;
        return;
    }
    const u64* list = a.list + (size_t)ic * kListCap;
    for (int i = t; i < n; i += kClsThreads) keys[i] = list[i];
    __syncthreads();
    // rank sort, descending key: score first, lower box index first among equal scores (:371)
    const int m = min(n, a.cap);
    const float* pimg = a.pred + (size_t)img * a.nboxes * a.no;
    float off = 0.f;
    if (a.method == M_HARD_BATCH) off = (float)c * (key2f(a.img_max[img]) + 1.f);   // torchvision's coordinate trick
    for (int i = t; i < n; i += kClsThreads) {
        const u64 k = keys[i];
        int rank = 0;
        for (int j = 0; j < n; ++j) rank += keys[j] > k ? 1 : 0;
        if (rank < m) {
            const unsigned b = 0xFFFFFFFFu - (unsigned)(k & 0xFFFFFFFFull);
            const float* p = pimg + (size_t)b * a.no;
            const float cx = p[0], cy = p[1], w = p[2], h = p[3];
            box[rank] = make_float4(cx - w / 2.f, cy - h / 2.f, cx + w / 2.f, cy + h / 2.f);   // :591-598
            score[rank] = p[5 + c] * p[4];
        }
    }
    __syncthreads();
    for (int i = t; i < m; i += kClsThreads) alive[0][i] = (unsigned short)i;
    __syncthreads();
    int nout = 0;
    auto emit = [&](int r) {      // called by every thread with the same r; thread 0 writes
        if (t == 0) {
            float* o = out + (size_t)nout * 6;
            const float4 b = box[r];
            o[0] = b.x; o[1] = b.y; o[2] = b.z; o[3] = b.w; o[4] = score[r]; o[5] = (float)c;
        }
        ++nout;
    };
    if (m == 1) {                 // :387-389
        emit(0);
        if (t == 0) a.cls_count[ic] = 1;
        return;
    }
    // the sequential loops of :394-454: `cur` rows remain, listed in alive[ph]; the head is alive[ph][0]
    int cur = m, ph = 0;
    const int method = a.method;
    while (cur > 0) {
        const int head = alive[ph][0];
        if (method == M_AND) {
            if (cur <= 1) break;                                  // :409
        } else if (method != M_MERGE) {
            emit(head);                                           // :400 / :435 / :449
            if (cur == 1) break;
        } else if (cur == 1) {                                    // :418-420
            emit(head);
            break;
        }
        const float4 hb = method == M_HARD_BATCH
                              ? make_float4(box[head].x + off, box[head].y + off, box[head].z + off, box[head].w + off)
                              : box[head];
        // every remaining row against the head: does it stay?
        const int first = method == M_MERGE ? 0 : 1;             // merge compares the head with itself too (:421)
        float sx = 0.f, sy = 0.f, sz = 0.f, sw = 0.f, ss = 0.f;  // merge: weighted sums of the flagged rows
        float mx = -INFINITY;                                     // and: largest IoU
        unsigned keep_bits[kListCap / kClsThreads];
#pragma unroll
        for (int u = 0; u < kListCap / kClsThreads; ++u) {
            keep_bits[u] = 0;
            const int pos = first + t + u * kClsThreads;
            if (pos >= cur) continue;
            const int r = alive[ph][pos];
            bool stays;
            if (method == M_HARD_BATCH || method == M_HARD) {
                float4 rb = box[r];
                if (method == M_HARD_BATCH) rb = make_float4(rb.x + off, rb.y + off, rb.z + off, rb.w + off);
                stays = !(iou_tv(hb, rb) > a.thr);
            } else if (method == M_MERGE) {
                const bool flagged = iou2(hb, box[r], false) > a.thr;
                if (flagged) {
                    const float wgt = score[r];
                    sx += wgt * box[r].x; sy += wgt * box[r].y; sz += wgt * box[r].z; sw += wgt * box[r].w; ss += wgt;
                }
                stays = !flagged && pos != 0;                      // :425 (the head always goes: it is its own duplicate)
            } else if (method == M_SOFT) {
                const float v = iou2(hb, box[r], false);
                const float s2 = score[r] * expf(-(v * v) / 0.5f);  // :443
                score[r] = s2;
                stays = s2 > a.conf;                               // :444
            } else {
                const float v = iou2(hb, box[r], method == M_DIOU);
                mx = fmaxf(mx, v);
                stays = v < a.thr;                                 // :405 / :413 / :454
            }
            keep_bits[u] = stays ? 1u : 0u;
        }
        if (method == M_MERGE) {                                  // :422-424
            const float tx = block_sum(sx, red), ty = block_sum(sy, red), tz = block_sum(sz, red), tw = block_sum(sw, red);
            const float ts = block_sum(ss, red);
            if (t == 0) box[head] = make_float4(tx / ts, ty / ts, tz / ts, tw / ts);
            __syncthreads();
            emit(head);
        } else if (method == M_AND) {                             // :411-412
            red[t] = mx;
            __syncthreads();
            for (int s = kClsThreads / 2; s > 0; s >>= 1) {
                if (t < s) red[t] = fmaxf(red[t], red[t + s]);
                __syncthreads();
            }
            const bool overlapped = red[0] > 0.5f;
            __syncthreads();
            if (overlapped) emit(head);
        }
        // compaction in order: positions are (t + u*256), so order by u then t: per-u prefix over threads
        int base = 0;
        const int lane = t & 63, wv = t >> 6;
#pragma unroll
        for (int u = 0; u < kListCap / kClsThreads; ++u) {
            const unsigned long long bal = __ballot(keep_bits[u] != 0);
            if (lane == 0) s_cnt[wv] = (unsigned)__popcll(bal);
            __syncthreads();
            int before = 0, total = 0;
            for (int w2 = 0; w2 < kClsThreads / 64; ++w2) {
                const int cw = (int)s_cnt[w2];
                if (w2 < wv) before += cw;
                total += cw;
            }
            if (keep_bits[u]) {
                const int pos = first + t + u * kClsThreads;
                alive[ph ^ 1][base + before + __popcll(bal & ((1ull << lane) - 1ull))] = alive[ph][pos];
            }
            base += total;
            __syncthreads();
        }
        cur = base;
        ph ^= 1;
    }
    if (t == 0) a.cls_count[ic] = (unsigned)nout;
}

// per image: rows of all classes, ascending class (the order of det_max, :384-457), into tmp_rows; total into out_count
__global__ __launch_bounds__(256) void nms2_merge_kernel(const Nms2Args a) {
    const int img = blockIdx.x;
    float* tmp = a.tmp_rows + (size_t)img * a.nc * a.cap * 6;
    int base = 0;
    for (int c = 0; c < a.nc; ++c) {
        const int cnt = (int)a.cls_count[(size_t)img * a.nc + c];
        const float* src = a.cls_rows + ((size_t)img * a.nc + c) * a.cap * 6;
        for (int e = threadIdx.x; e < cnt * 6; e += 256) tmp[(size_t)base * 6 + e] = src[e];
        base += cnt;
    }
    if (threadIdx.x == 0) a.out_count[img] = base < a.out_cap ? base : a.out_cap;
}

// rank of every row among its image's rows by descending score (equal scores: earlier row first), scatter to out
__global__ __launch_bounds__(256) void nms2_order_kernel(const Nms2Args a) {
    __shared__ float tile[256];
    const int img = blockIdx.y;
    int total = 0;
    for (int c = 0; c < a.nc; ++c) total += (int)a.cls_count[(size_t)img * a.nc + c];
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (blockIdx.x * 256 >= total) return;
    const float* tmp = a.tmp_rows + (size_t)img * a.nc * a.cap * 6;
    const float mine = i < total ? tmp[(size_t)i * 6 + 4] : 0.f;
    int rank = 0;
    for (int j0 = 0; j0 < total; j0 += 256) {
        const int j = j0 + threadIdx.x;
        tile[threadIdx.x] = j < total ? tmp[(size_t)j * 6 + 4] : -INFINITY;
        __syncthreads();
        const int lim = min(256, total - j0);
        if (i < total)
            for (int k = 0; k < lim; ++k) {
                const float s = tile[k];
                rank += (s > mine || (s == mine && j0 + k < i)) ? 1 : 0;
            }
        __syncthreads();
    }
    if (i < total && rank < a.out_cap) {
        float* o = a.out + ((size_t)img * a.out_cap + rank) * 6;
        for (int e = 0; e < 6; ++e) o[e] = tmp[(size_t)i * 6 + e];
    }
}

size_t nms2_layout(int N, int nc, int cap, size_t* off) {
    size_t o = 0;
    auto take = [&](size_t bytes) { size_t r = o; o += (bytes + 255) & ~(size_t)255; return r; };
    const size_t ncl = (size_t)N * nc;
    const size_t v[8] = {take(ncl * kBins * 4), take(ncl * 4), take(ncl * 4), take((size_t)N * 4), take(256),
                         take(ncl * kListCap * 8), take(ncl * cap * 6 * 4), take(ncl * 4)};
    const size_t tmp = take(ncl * cap * 6 * 4);
    if (off) { for (int i = 0; i < 8; ++i) off[i] = v[i]; off[8] = tmp; }
    return o;
}

int method_cap(int method) { return method == M_HARD_BATCH ? kListCap : kClassCap; }

// ==================================================================================================
// Cluster-Weighted DIoU-NMS (:463-548) on the candidates the YOLOv5 selection kernels left sorted
// ==================================================================================================
// :210-244 on two xyxy boxes
__device__ __forceinline__ float jaccard_diou_pair(const float4& A, const float4& B) {
    const float iw = fmaxf(fminf(A.z, B.z) - fmaxf(A.x, B.x), 0.f), ih = fmaxf(fminf(A.w, B.w) - fmaxf(A.y, B.y), 0.f);
    const float inter = iw * ih;
    const float area_a = (A.z - A.x) * (A.w - A.y), area_b = (B.z - B.x) * (B.w - B.y);
    const float uni = area_a + area_b - inter;
    const float x1 = (A.z + A.x) / 2.f, y1 = (A.w + A.y) / 2.f, x2 = (B.z + B.x) / 2.f, y2 = (B.w + B.y) / 2.f;
    const float cr = fmaxf(A.z, B.z), cl = fminf(A.x, B.x), ct = fminf(A.y, B.y), cb = fmaxf(A.w, B.w);
    const float dx = x2 - x1, dy = y2 - y1, ew = cr - cl, eh = cb - ct;
    const float D = (dx * dx + dy * dy) / (ew * ew + eh * eh + 1e-7f);
    return inter / (uni + 1e-7f) - powf(D, 0.7f);
}

}  // namespace

namespace ydet {

// Bit rows of one image's suppression relations, built and iterated by one workgroup (declared in nms.hip's runner).
struct ClusterArgs {
    const float4* sbox;      // [N][stride] offset boxes, descending score
    const u64* sorted;       // [N][stride] keys (score, ~candidate index)
    const unsigned* ns;      // per image: candidates (stride 16 words: the selection kernels' per-image record, word 4)
    const float* pred;
    int N, nboxes, no, nc, stride, max_box;
    float thr;
    u64* ge;                 // [N][max_box][words]  bit i of row j: X[i][j] >= thr  (i < j)
    u64* gt;                 // ... X[i][j] > thr
    float* out;              // [N][max_box][6]
    int32_t* out_count;
};

}  // namespace ydet

namespace {

constexpr int kCluThreads = 1024;
constexpr int kCluMaxBox = 2048;

__global__ __launch_bounds__(kCluThreads) void cluster_diou_kernel(const ydet::ClusterArgs a) {
    __shared__ float4 box[kCluMaxBox];
    __shared__ u64 act[2][kCluMaxBox / 64];     // rows still multiplying the matrix (E), previous / new
    __shared__ u64 nz[kCluMaxBox / 64];         // rows with a non-zero entry right of the diagonal
    __shared__ u64 keepw[kCluMaxBox / 64];
    __shared__ int s_same, s_kept[kCluMaxBox / 64 + 1];
    const int img = blockIdx.x, t = threadIdx.x;
    const int n = min((int)a.ns[(size_t)img * 16 + 4], a.max_box);
    const int mb = a.max_box, words = (mb + 63) >> 6;
    const float4* sb = a.sbox + (size_t)img * a.stride;
    for (int i = t; i < mb; i += kCluThreads) box[i] = i < n ? sb[i] : make_float4(0.f, 0.f, 0.f, 0.f);   // :522 pads with zeros
    for (int w = t; w < words; w += kCluThreads) { nz[w] = 0; act[0][w] = ~0ull; }
    __syncthreads();
    u64* ge = a.ge + (size_t)img * mb * words;
    u64* gt = a.gt + (size_t)img * mb * words;
    // the matrix of :527, thresholded: one (column j, word of rows i) per step of a thread
    for (int item = t; item < mb * words; item += kCluThreads) {
        const int j = item / words, w = item - j * words;
        u64 bge = 0, bgt = 0, bnz = 0;
        const float4 bj = box[j];
        const int i1 = min(64 * w + 64, j);       // rows i < j only (triu, diagonal = 1)
        for (int i = 64 * w; i < i1; ++i) {
            const float x = jaccard_diou_pair(box[i], bj);
            if (x >= a.thr) bge |= 1ull << (i & 63);
            if (x > a.thr) bgt |= 1ull << (i & 63);
            if (x != 0.f) bnz |= 1ull << (i & 63);
        }
        ge[item] = bge;
        gt[item] = bgt;
        if (bnz) atomicOr(&nz[w], bnz);
    }
    __syncthreads();
    // :528-535.  A = X * E_prev, B = X * E_new with E_new[j] = (max_i A[i][j] < thr); A.equal(B) when the rows whose
    // flag changed hold nothing but zeros.  (Every column of A holds a zero, so a column maximum is never below 0.)
    int ph = 0;
    for (int it = 0; it < 200; ++it) {
        for (int w = t; w < words; w += kCluThreads) act[ph ^ 1][w] = 0;
        if (t == 0) s_same = 1;
        __syncthreads();
        for (int j = t; j < mb; j += kCluThreads) {
            bool hit = false;
            const u64* row = ge + (size_t)j * words;
            for (int w = 0; w <= (j >> 6) && !hit; ++w) hit = (row[w] & act[ph][w]) != 0;
            if (!hit && 0.f < a.thr) atomicOr(&act[ph ^ 1][j >> 6], 1ull << (j & 63));
        }
        __syncthreads();
        for (int w = t; w < words; w += kCluThreads)
            if ((act[ph][w] ^ act[ph ^ 1][w]) & nz[w]) s_same = 0;
        __syncthreads();
        const bool same = s_same != 0;
        __syncthreads();
        if (same || it == 199) break;
        ph ^= 1;
    }
    // here act[ph] = E of the last A, act[ph ^ 1] = E of the last B
    for (int w = t; w < words; w += kCluThreads) keepw[w] = 0;
    __syncthreads();
    for (int j = t; j < mb; j += kCluThreads) {                   // :536 keep = maxA <= thr
        bool hit = false;
        const u64* row = gt + (size_t)j * words;
        for (int w = 0; w <= (j >> 6) && !hit; ++w) hit = (row[w] & act[ph][w]) != 0;
        if (!hit && 0.f <= a.thr) atomicOr(&keepw[j >> 6], 1ull << (j & 63));
    }
    __syncthreads();
    if (t == 0) {
        int acc = 0;
        for (int w = 0; w < words; ++w) { s_kept[w] = acc; acc += __popcll(keepw[w]); }
        s_kept[words] = acc;
        a.out_count[img] = acc;
    }
    __syncthreads();
    // :538-546 weighted coordinates of the kept rows: row i of B, entries above 0.8, times the scores; the diagonal
    // carries the row's own score
    const u64* sorted = a.sorted + (size_t)img * a.stride;
    const float* pimg = a.pred + (size_t)img * a.nboxes * a.no;
    float* out = a.out + (size_t)img * mb * 6;
    auto row_of = [&](int s, float4& b, float& sc, float& cl) {   // pred1 row s: box without the class offset
        if (s >= n) { b = make_float4(0.f, 0.f, 0.f, 0.f); sc = 0.f; cl = 0.f; return; }
        const u64 k = sorted[s];
        const unsigned idx = 0xFFFFFFFFu - (unsigned)(k & 0xFFFFFFFFull);
        const unsigned bx = idx / (unsigned)a.nc, c = idx - bx * (unsigned)a.nc;
        const float* p = pimg + (size_t)bx * a.no;
        const float cx = p[0], cy = p[1], w = p[2], h = p[3];
        b = make_float4(cx - w / 2.f, cy - h / 2.f, cx + w / 2.f, cy + h / 2.f);
        sc = p[5 + c] * p[4];
        cl = (float)c;
    };
    for (int i = t; i < mb; i += kCluThreads) {
        if (!((keepw[i >> 6] >> (i & 63)) & 1ull)) continue;
        const int pos = s_kept[i >> 6] + __popcll(keepw[i >> 6] & ((1ull << (i & 63)) - 1ull));
        float4 bi; float si, ci;
        row_of(i, bi, si, ci);
        float sx = si * bi.x, sy = si * bi.y, sz = si * bi.z, sw = si * bi.w, ss = si;
        if ((act[ph ^ 1][i >> 6] >> (i & 63)) & 1ull) {
            const float4 oi = box[i];
            for (int j = i + 1; j < n; ++j) {
                const float x = jaccard_diou_pair(oi, box[j]);
                if (x > 0.8f) {
                    float4 bj; float sj, cj;
                    row_of(j, bj, sj, cj);
                    const float wgt = x * sj;
                    sx += wgt * bj.x; sy += wgt * bj.y; sz += wgt * bj.z; sw += wgt * bj.w; ss += wgt;
                }
            }
        }
        float* o = out + (size_t)pos * 6;
        o[0] = sx / ss; o[1] = sy / ss; o[2] = sz / ss; o[3] = sw / ss; o[4] = si; o[5] = ci;
    }
}

}  // namespace

namespace ydet {
size_t cluster_diou_ws_bytes(int N, int max_box) {
    const size_t words = (size_t)((max_box + 63) >> 6);
    return 2 * (((size_t)N * max_box * words * 8 + 255) & ~(size_t)255);
}
int launch_cluster_diou(ClusterArgs a, void* ws, hipStream_t stream) {
    YDET_CHECK_ARG(a.max_box >= 1 && a.max_box <= kCluMaxBox, "nms_cluster_diou: max_box=%d (1..%d)", a.max_box, kCluMaxBox);
    const size_t words = (size_t)((a.max_box + 63) >> 6);
    const size_t one = ((size_t)a.N * a.max_box * words * 8 + 255) & ~(size_t)255;
    a.ge = (u64*)ws;
    a.gt = (u64*)((char*)ws + one);
    hipLaunchKernelGGL(cluster_diou_kernel, dim3((unsigned)a.N), dim3(kCluThreads), 0, stream, a);
    YDET_CHECK_LAUNCH("cluster_diou_kernel");
    return YDET_OK;
}
}  // namespace ydet

extern "C" size_t ydet_nms2_workspace_bytes(int N, int nboxes, int no, int method) {
    (void)nboxes;
    if (N <= 0 || no < 6 || method < 0 || method > M_DIOU) return 0;
    return nms2_layout(N, no - 5, method_cap(method), nullptr);
}

extern "C" int ydet_nms2_out_cap(int no, int method) {
    if (no < 6 || method < 0 || method > M_DIOU) return 0;
    return (no - 5) * method_cap(method);
}

extern "C" int ydet_nms2(const float* pred, int N, int nboxes, int no, float conf_thres, float nms_thres, int multi_cls,
                         int method, float* out_rows, int out_cap, int32_t* out_count, int32_t* overflow, void* workspace,
                         size_t workspace_bytes, void* stream) {
    clear_error();
    YDET_CHECK_ARG(pred && out_rows && out_count && overflow && workspace, "nms2: null pointer");
    YDET_CHECK_ARG(N > 0 && N <= 65535 && nboxes > 0 && no >= 6, "nms2: N=%d nboxes=%d no=%d", N, nboxes, no);
    YDET_CHECK_ARG(method >= 0 && method <= M_DIOU, "nms2: method %d", method);
    YDET_CHECK_ARG(out_cap >= 1, "nms2: out_cap=%d", out_cap);
    YDET_CHECK_ARG(((uintptr_t)workspace & 255) == 0 && ((uintptr_t)pred & 3) == 0, "nms2: alignment");
    const int nc = no - 5, cap = method_cap(method);
    YDET_CHECK_ARG((long long)N * nc < (1ll << 31) && (long long)N * nboxes < (1ll << 40), "nms2: problem too large");
    size_t off[9];
    const size_t need = nms2_layout(N, nc, cap, off);
    if (workspace_bytes < need) return fail(YDET_ERR_WORKSPACE, "nms2: workspace %zu < %zu bytes", workspace_bytes, need);
    hipStream_t s = as_stream(stream);
    char* ws = (char*)workspace;
    Nms2Args a;
    a.pred = pred; a.N = N; a.nboxes = nboxes; a.no = no; a.nc = nc;
    a.conf = conf_thres; a.thr = nms_thres;
    a.multi = (multi_cls || conf_thres < 0.01f) ? 1 : 0;          // :352
    a.method = method; a.cap = cap;
    a.base16 = f2key_host(conf_thres) >> 16;
    a.hist = (unsigned*)(ws + off[0]); a.thr_bin = (unsigned*)(ws + off[1]); a.cursor = (unsigned*)(ws + off[2]);
    a.img_max = (unsigned*)(ws + off[3]); a.flags = (unsigned*)(ws + off[4]); a.list = (u64*)(ws + off[5]);
    a.cls_rows = (float*)(ws + off[6]); a.cls_count = (unsigned*)(ws + off[7]); a.tmp_rows = (float*)(ws + off[8]);
    a.out = out_rows; a.out_count = out_count; a.out_cap = out_cap;
    // histograms, thresholds, cursors, maxima and the flag word are one contiguous zero fill
    if (hipMemsetAsync(ws + off[0], 0, off[5] - off[0], s) != hipSuccess) return fail(YDET_ERR_LAUNCH, "nms2: memset failed");
    const long long boxes = (long long)N * nboxes;
    const unsigned gb = (unsigned)((boxes + 255) / 256);
    hipLaunchKernelGGL(nms2_hist_kernel, dim3(gb), dim3(256), 0, s, a);
    hipLaunchKernelGGL(nms2_thresh_kernel, dim3((unsigned)(N * nc)), dim3(64), 0, s, a);
    hipLaunchKernelGGL(nms2_collect_kernel, dim3(gb), dim3(256), 0, s, a);
    hipLaunchKernelGGL(nms2_class_kernel, dim3((unsigned)(N * nc)), dim3(kClsThreads), 0, s, a);
    hipLaunchKernelGGL(nms2_merge_kernel, dim3((unsigned)N), dim3(256), 0, s, a);
    const unsigned chunks = (unsigned)(((long long)nc * cap + 255) / 256);
    hipLaunchKernelGGL(nms2_order_kernel, dim3(chunks, (unsigned)N), dim3(256), 0, s, a);
    if (hipMemcpyAsync(overflow, a.flags, sizeof(int32_t), hipMemcpyDeviceToDevice, s) != hipSuccess)
        return fail(YDET_ERR_LAUNCH, "nms2: copy of the overflow flag failed");
    YDET_CHECK_LAUNCH("nms2 kernels");
    return YDET_OK;
}
