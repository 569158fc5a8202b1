// Non-maximum suppression for gfx950: candidate filtering, exact top-k pre-filter, rank sort and a
// chunked greedy NMS whose 64x64 intra-chunk suppression matrix is built with wave-wide ORs.
// Compiled with -ffp-contract=off: the IoU expressions must round exactly like the scalar
// CPU code they replace (inter / (area_i + area_j - inter), no fused multiply-add).
//
// Shared machinery
//   order key  : 64-bit, unique per candidate: (orderable(score) << 32) | tie, so "greater key" is
//                a strict total order; sorting is a rank count (#keys greater than mine).
//   greedy NMS : sorted boxes are processed in chunks of 64.  Phase A builds the chunk's 64x64
//                "i suppresses j" matrix in parallel; phase B walks the 64 rows serially (one
//                lane) against the bitset of already-removed boxes; phase C applies the chunk's
//                kept boxes to every later box in parallel.
#include <mutex>

#include "ydet_internal.h"

using namespace ydet;

namespace {

typedef unsigned long long u64;

__device__ __forceinline__ unsigned f2key(float f) {
    const unsigned u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
static inline unsigned f2key_host(float f) {
    union { float f; unsigned u; } cv;
    cv.f = f;
    const unsigned u = cv.u;
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

// torchvision-style overlap test: w,h = max(0, .), suppress when ovr > thr.
// numpy-style (utils/utils.py nms): w,h = max(1e-28, .), survive when ovr <= thr (NaN suppresses).
template <bool EPS_LE>
__device__ __forceinline__ bool suppresses(const float4& bi, float ai, const float4& bj, float aj, float thr) {
    const float xx1 = fmaxf(bi.x, bj.x), yy1 = fmaxf(bi.y, bj.y);
    const float xx2 = fminf(bi.z, bj.z), yy2 = fminf(bi.w, bj.w);
    if (EPS_LE) {
        const float w = fmaxf(1e-28f, xx2 - xx1), h = fmaxf(1e-28f, yy2 - yy1);
        const float inter = w * h;
        const float ovr = inter / (ai + aj - inter);
        return !(ovr <= thr);
    } else {
        const float w = fmaxf(0.f, xx2 - xx1), h = fmaxf(0.f, yy2 - yy1);
        const float inter = w * h;
        const float ovr = inter / (ai + aj - inter);
        return ovr > thr;
    }
}
__device__ __forceinline__ float box_area(const float4& b) { return (b.z - b.x) * (b.w - b.y); }

struct NmsShared {
    float4 cbox[64];
    float carea[64];
    u64 rowmask[64];
    u64 kept_word;
};

// Greedy NMS over `ns` boxes already sorted by descending priority.  load(s) returns box s.
// removed / kept: LDS bitsets of (ns+63)/64 words, zeroed here.  NT threads (256 or 1024).
template <bool EPS_LE, int NT = 256, typename LoadBox>
__device__ void greedy_nms_block(int ns, float thr, LoadBox load, NmsShared& sh, u64* removed, u64* kept) {
    constexpr int PARTS = NT / 64, PER = 64 / PARTS;   // threads per box of a chunk, boxes each of them tests
    const int tid = threadIdx.x;
    const int nwords = (ns + 63) >> 6;
    for (int w = tid; w < nwords; w += NT) { removed[w] = 0; kept[w] = 0; }
    __syncthreads();
    for (int c = 0; c < nwords; ++c) {
        const int base = c << 6;
        const int cn = min(64, ns - base);
        if (tid < 64) {
            float4 b = make_float4(0.f, 0.f, 0.f, 0.f);
            if (tid < cn) b = load(base + tid);
            sh.cbox[tid] = b;
            sh.carea[tid] = box_area(b);
        }
        __syncthreads();
        // phase A: thread (i = tid / PARTS, part = tid % PARTS) tests j = part*PER .. part*PER+PER-1, j > i
        {
            const int i = tid / PARTS, part = tid % PARTS;
            u64 bits = 0;
            if (i < cn) {
                const float4 bi = sh.cbox[i];
                const float ai = sh.carea[i];
#pragma unroll
                for (int jj = 0; jj < PER; ++jj) {
                    const int j = part * PER + jj;
                    if (j > i && j < cn && suppresses<EPS_LE>(bi, ai, sh.cbox[j], sh.carea[j], thr))
                        bits |= 1ull << j;
                }
            }
#pragma unroll
            for (int d = 1; d < PARTS; d <<= 1) bits |= __shfl_xor(bits, d);
            if (part == 0) sh.rowmask[i] = bits;
        }
        __syncthreads();
        // phase B: serial walk over the chunk.  One wave, lane i holds row i's mask; the walk visits only boxes still
        // alive and fetches their masks with readlane (wave-uniform, no LDS round trip per step)
        if (tid < 64) {
            const u64 mine = tid < cn ? sh.rowmask[tid] : 0ull;
            const u64 valid = cn == 64 ? ~0ull : ((1ull << cn) - 1ull);
            u64 rem = removed[c], kp = 0;
            u64 cand = ~rem & valid;
            while (cand) {
                const int i = __ffsll((long long)cand) - 1;
                kp |= 1ull << i;
                const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)mine, i);
                const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(mine >> 32), i);
                rem |= ((u64)hi << 32) | (u64)lo;
                cand = ~rem & valid & ~(((1ull << i) << 1) - 1ull);   // alive and behind i
            }
            if (tid == 0) {
                kept[c] = kp;
                sh.kept_word = kp;
            }
        }
        __syncthreads();
        // phase C: kept boxes of this chunk against every later box
        const u64 kp = sh.kept_word;
        if (kp) {
            constexpr int CP = NT >= 1024 ? 4 : 1;     // threads per later box, each takes 64 / CP of the chunk's boxes
            for (int t = tid; t < (ns - base - 64) * CP; t += NT) {
                const int j = base + 64 + t / CP;
                if ((removed[j >> 6] >> (j & 63)) & 1ull) continue;
                const float4 bj = load(j);
                const float aj = box_area(bj);
                u64 m = CP == 1 ? kp : kp & (((1ull << (64 / CP)) - 1ull) << ((t % CP) * (64 / CP)));
                while (m) {
                    const int i = __ffsll((long long)m) - 1;
                    m &= m - 1;
                    if (suppresses<EPS_LE>(sh.cbox[i], sh.carea[i], bj, aj, thr)) {
                        atomicOr(&removed[j >> 6], 1ull << (j & 63));
                        break;
                    }
                }
            }
        }
        __syncthreads();
    }
}

// ==================================================================================================
// YOLOv5 non_max_suppression
// ==================================================================================================
constexpr int kHistBins = 2048;
constexpr int kListCap = 4096;   // candidates kept for the final sort (>= max_nms)
constexpr int kMaxNms = 4096;
constexpr int kTileFloats = 8192;  // LDS staging tile of prediction rows (exact-select path)
constexpr int kScanRows = 128;     // rows per workgroup of the row-maximum scan
constexpr int kScanTile = 11008;   // floats of its LDS tile (128 rows of 85 + slack)
constexpr int kPickThreads = 1024;
constexpr int kNmsLdsBoxes = 1024; // sorted boxes the greedy NMS keeps in LDS (16 KB)

struct V5Meta {  // per image, 16 words
    unsigned total;       // rows (boxes) with at least one candidate above conf
    unsigned thr_bin;     // lowest score bin that enters the list
    unsigned list_count;  // entries appended to the list
    unsigned overflow;    // list would exceed kListCap: exact select path
    unsigned ns;          // sorted candidates entering the NMS
    unsigned pad[11];
};

struct V5Args {
    const float* pred;
    int N, nboxes, no, nc, rows_per_block, scan_rows;
    float conf, iou;
    int multi, agnostic, max_nms, max_det;
    float max_wh;
    const uint8_t* class_mask;
    unsigned base16;  // (orderable(conf) >> 16)
    const float* row_max;  // [N][nboxes]: max over classes of cls*obj (-inf: no candidate); caller's or workspace
    // workspace
    float* row_max_ws;     // [N][nboxes]
    V5Meta* meta;     // [N]
    u64* list;        // [N][kListCap]
    u64* sorted;      // [N][kMaxNms]
    float4* sbox;     // [N][kMaxNms]  offset boxes in sorted order
    float* out_det;
    int32_t* out_count;
};

__device__ __forceinline__ unsigned score_bin(float s, unsigned base16) {
    const unsigned b = (f2key(s) >> 16) - base16;
    return b > (unsigned)(kHistBins - 1) ? (unsigned)(kHistBins - 1) : b;
}

// Stages rows [row0, row0+rows) of one image into LDS and calls visit(candidate index, score) for
// every candidate: yolov5_official/utils.py:122 (obj > conf), :151 (cls *= obj), :157-162.
template <typename F>
__device__ __forceinline__ void v5_visit(const V5Args& a, const float* img, int row0, int rows, float* tile, F visit) {
    const int nelem = rows * a.no;
    const float* src = img + (size_t)row0 * a.no;
    for (int e = threadIdx.x; e < nelem; e += blockDim.x) tile[e] = src[e];
    __syncthreads();
    if (a.multi) {
        for (int e = threadIdx.x; e < nelem; e += blockDim.x) {
            const int r = e / a.no, o = e - r * a.no;
            if (o < 5) continue;
            const float obj = tile[r * a.no + 4];
            if (!(obj > a.conf)) continue;
            const float s = tile[e] * obj;
            if (!(s > a.conf)) continue;
            if (a.class_mask && !a.class_mask[o - 5]) continue;
            visit((unsigned)((row0 + r) * a.nc + (o - 5)), s);
        }
    } else {
        for (int r = threadIdx.x; r < rows; r += blockDim.x) {
            const float* p = tile + r * a.no;
            const float obj = p[4];
            if (!(obj > a.conf)) continue;
            float best = p[5] * obj;
            int bj = 0;
            for (int j = 1; j < a.nc; ++j) {
                const float v = p[5 + j] * obj;
                if (v > best) { best = v; bj = j; }
            }
            if (!(best > a.conf)) continue;
            if (a.class_mask && !a.class_mask[bj]) continue;
            visit((unsigned)((row0 + r) * a.nc + bj), best);
        }
    }
    __syncthreads();
}

// Row maxima: for every box the largest cls*obj over the (unmasked) classes, -inf when obj <= conf.  One pass over
// the predictions; 128 rows per workgroup staged through LDS (row pitch `no` floats: an odd pitch such as 85 makes
// the per-row walks below bank-conflict free), two threads per row.
__global__ __launch_bounds__(256) void v5_rowmax_kernel(const V5Args a) {
    __shared__ __attribute__((aligned(16))) float tile[kScanTile];
    const int img = blockIdx.y;
    const int row0 = blockIdx.x * a.scan_rows;
    const int rows = min(a.scan_rows, a.nboxes - row0);
    const int nelem = rows * a.no;
    const float* src = a.pred + ((size_t)img * a.nboxes + row0) * a.no;
    if (((uintptr_t)src & 15) == 0) {
        const int nv = nelem >> 2;
        for (int i = threadIdx.x; i < nv; i += 256)
            reinterpret_cast<float4*>(tile)[i] = reinterpret_cast<const float4*>(src)[i];
        for (int e = (nv << 2) + threadIdx.x; e < nelem; e += 256) tile[e] = src[e];
    } else {
        for (int e = threadIdx.x; e < nelem; e += 256) tile[e] = src[e];
    }
    __syncthreads();
    const int r = threadIdx.x >> 1, half = threadIdx.x & 1;
    float m = -INFINITY, obj = 0.f;
    if (r < rows) {
        const float* p = tile + r * a.no;
        obj = p[4];
        const int c0 = half ? a.nc >> 1 : 0, c1 = half ? a.nc : a.nc >> 1;
        if (a.class_mask && a.multi) {
            for (int c = c0; c < c1; ++c)
                if (a.class_mask[c]) m = fmaxf(m, p[5 + c] * obj);
        } else if (a.class_mask) {
            // best-class mode picks the best class over ALL classes and filters afterwards
            // (yolov5_official/utils.py:160-166): a row whose best class is masked out has no candidate at all
            int bj = -1;
            for (int c = c0; c < c1; ++c) {
                const float v = p[5 + c] * obj;
                if (bj < 0 || v > m) { m = v; bj = c; }
            }
            const float mo = __shfl_xor(m, 1);
            const int bo = __shfl_xor(bj, 1);
            // first maximum: the lower half wins ties (an empty half has bj < 0)
            const bool other = bo >= 0 && (bj < 0 || (half ? !(m > mo) : mo > m));
            if (other) { m = mo; bj = bo; }
            if (bj < 0 || !a.class_mask[bj]) m = -INFINITY;
        } else {
            for (int c = c0; c < c1; ++c) m = fmaxf(m, p[5 + c] * obj);
        }
    }
    m = fmaxf(m, __shfl_xor(m, 1));
    if (r < rows && half == 0) a.row_max_ws[(size_t)img * a.nboxes + row0 + r] = obj > a.conf ? m : -INFINITY;
}

// Lowest bin b with sum_{i>=b} hist[i] >= want (0 when the total is smaller); called by every thread of the
// workgroup, which must be at least 256 wide.  part: 256 words of scratch.  *total_out: sum of all bins.
__device__ unsigned select_bin(const unsigned* hist, unsigned* part, unsigned* s_out, unsigned want, unsigned* total_out) {
    const int tid = threadIdx.x;
    if (tid < 256) {
        unsigned s = 0;
        for (int i = 0; i < 8; ++i) s += hist[tid * 8 + i];
        part[tid] = s;
    }
    __syncthreads();
    if (tid == 0) {
        unsigned total = 0;
        for (int i = 0; i < 256; ++i) total += part[i];
        unsigned thr_bin = 0;
        if (total > want) {
            unsigned acc = 0;
            int t = 255;
            for (; t >= 0; --t) {
                if (acc + part[t] >= want) break;
                acc += part[t];
            }
            int b = t * 8 + 7;
            for (;; --b) {
                acc += hist[b];
                if (acc >= want || b == t * 8) break;
            }
            thr_bin = (unsigned)b;
        }
        s_out[0] = thr_bin;
        s_out[1] = total;
    }
    __syncthreads();
    if (total_out) *total_out = s_out[1];
    return s_out[0];
}

// One workgroup per image.  (1) histogram of the row maxima above conf and the lowest bin b such that at least
// max_nms rows lie in bins >= b: each such row owns a candidate that good, so the max_nms best candidates all lie
// in bins >= b.  (2) rows in bins >= b are queued (a queue longer than the candidate list means the list would
// overflow too).  (3) the queued rows' candidates -- yolov5_official/utils.py:122 (obj > conf), :151 (cls *= obj),
// :157-159 (every class above conf) or :160-162 (best class) -- are histogrammed the same way, which gives the
// final bin b2 >= b, and those in bins >= b2 go to the list.
__global__ __launch_bounds__(kPickThreads) void v5_pick_kernel(const V5Args a) {
    __shared__ unsigned hist[kHistBins];
    __shared__ unsigned part[256];
    __shared__ unsigned queue[kListCap];
    __shared__ unsigned s_sel[2], s_nq, s_count;
    const int img = blockIdx.x, tid = threadIdx.x;
    const float* rm = a.row_max + (size_t)img * a.nboxes;
    const unsigned base16 = a.base16;
    for (int i = tid; i < kHistBins; i += kPickThreads) hist[i] = 0;
    if (tid == 0) { s_nq = 0; s_count = 0; }
    __syncthreads();
    // the row maxima of this thread stay in registers between the two passes (up to kRmRegs * 1024 boxes; the rest is
    // read again), and all loads of a pass are issued before the first use
    constexpr int kRmRegs = 32;
    float rv[kRmRegs];
#pragma unroll
    for (int k = 0; k < kRmRegs; ++k) {
        const int r = tid + k * kPickThreads;
        rv[k] = r < a.nboxes ? rm[r] : -INFINITY;
    }
#pragma unroll
    for (int k = 0; k < kRmRegs; ++k)
        if (rv[k] > a.conf) atomicAdd(&hist[score_bin(rv[k], base16)], 1u);
    for (int r = tid + kRmRegs * kPickThreads; r < a.nboxes; r += kPickThreads) {
        const float v = rm[r];
        if (v > a.conf) atomicAdd(&hist[score_bin(v, base16)], 1u);
    }
    __syncthreads();
    unsigned total_rows;
    const unsigned row_bin = select_bin(hist, part, s_sel, (unsigned)a.max_nms, &total_rows);
    auto enqueue = [&](int r, float v) {
        if (v > a.conf && score_bin(v, base16) >= row_bin) {
            const unsigned pos = atomicAdd(&s_nq, 1u);
            if (pos < (unsigned)kListCap) queue[pos] = (unsigned)r;
        }
    };
#pragma unroll
    for (int k = 0; k < kRmRegs; ++k) enqueue(tid + k * kPickThreads, rv[k]);
    for (int r = tid + kRmRegs * kPickThreads; r < a.nboxes; r += kPickThreads) enqueue(r, rm[r]);
    for (int i = tid; i < kHistBins; i += kPickThreads) hist[i] = 0;
    __syncthreads();
    V5Meta& m = a.meta[img];
    const unsigned nq = s_nq;
    if (nq > (unsigned)kListCap) {   // workgroup-uniform
        if (tid == 0) { m.total = total_rows; m.thr_bin = row_bin; m.ns = 0; m.list_count = 0; m.overflow = 1; }
        return;
    }
    const float* pimg = a.pred + (size_t)img * a.nboxes * a.no;
    u64* list = a.list + (size_t)img * kListCap;
    // visit(row, cls, score) for every candidate of the queued rows
    auto for_candidates = [&](auto visit) {
        if (a.multi) {
            // (row, class) pairs, consecutive threads on consecutive classes of a row; eight pairs per thread in
            // flight, their loads issued before the first test
            constexpr int U = 8;
            const unsigned npair = nq * (unsigned)a.nc;
            const float rnc = 1.0f / (float)a.nc;
            for (unsigned p0 = tid; p0 < npair; p0 += kPickThreads * U) {
                unsigned row[U];
                int cls[U];
                float obj[U], val[U];
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const unsigned p = p0 + u * kPickThreads;
                    const bool in = p < npair;
                    // the float-reciprocal quotient is exact below 2^22: nq <= 4096 rows, so up to 1000 classes
                    const unsigned q = !in ? 0u : a.nc <= 1000 ? (unsigned)div_small((int)p, a.nc, rnc) : p / (unsigned)a.nc;
                    cls[u] = in ? (int)(p - q * (unsigned)a.nc) : 0;
                    row[u] = queue[q];
                    const float* pr = pimg + (size_t)row[u] * a.no;
                    obj[u] = in ? pr[4] : 0.f;
                    val[u] = in ? pr[5 + cls[u]] : 0.f;
                }
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    if (p0 + u * kPickThreads >= npair || !(obj[u] > a.conf)) continue;
                    const float sc = val[u] * obj[u];
                    if (!(sc > a.conf)) continue;
                    if (a.class_mask && !a.class_mask[cls[u]]) continue;
                    visit(row[u], cls[u], sc);
                }
            }
        } else {
            for (unsigned q = tid; q < nq; q += kPickThreads) {
                const unsigned row = queue[q];
                const float* pr = pimg + (size_t)row * a.no;
                const float obj = pr[4];
                if (!(obj > a.conf)) continue;
                float best = pr[5] * obj;
                int bj = 0;
                for (int j = 1; j < a.nc; ++j) {
                    const float v = pr[5 + j] * obj;
                    if (v > best) { best = v; bj = j; }
                }
                if (!(best > a.conf)) continue;
                if (a.class_mask && !a.class_mask[bj]) continue;
                visit(row, bj, best);
            }
        }
    };
    for_candidates([&](unsigned, int, float sc) {
        const unsigned b = score_bin(sc, base16);
        if (b >= row_bin) atomicAdd(&hist[b], 1u);
    });
    __syncthreads();
    const unsigned thr_bin = select_bin(hist, part, s_sel, (unsigned)a.max_nms, nullptr);
    for_candidates([&](unsigned row, int cls, float sc) {
        if (score_bin(sc, base16) >= thr_bin && score_bin(sc, base16) >= row_bin) {
            const unsigned pos = atomicAdd(&s_count, 1u);
            if (pos < (unsigned)kListCap)
                list[pos] = ((u64)f2key(sc) << 32) | (u64)(0xFFFFFFFFu - (row * (unsigned)a.nc + (unsigned)cls));
        }
    });
    __syncthreads();
    if (tid == 0) {
        m.total = total_rows;
        m.thr_bin = thr_bin;
        m.ns = 0;
        m.list_count = s_count;
        m.overflow = s_count > (unsigned)kListCap ? 1u : 0u;
    }
}

// Exact path for images whose threshold bin holds more than kListCap candidates (heavy score ties):
// byte-wise radix select of the max_nms-th largest 64-bit key, then a collect of all keys >= it.
// One workgroup per image re-reads the image once per digit; it runs only when flagged.
__global__ __launch_bounds__(256) void v5_exact_select_kernel(const V5Args a) {
    __shared__ float tile[kTileFloats];
    __shared__ unsigned dh[256];
    __shared__ u64 s_prefix, s_mask;
    __shared__ unsigned s_remaining, s_count;
    const int img = blockIdx.x;
    V5Meta* m = a.meta + img;
    if (!m->overflow) return;
    const float* pimg = a.pred + (size_t)img * a.nboxes * a.no;
    if (threadIdx.x == 0) { s_prefix = 0; s_mask = 0; s_remaining = (unsigned)a.max_nms; s_count = 0; }
    __syncthreads();
    for (int shift = 56; shift >= 0; shift -= 8) {
        for (int i = threadIdx.x; i < 256; i += 256) dh[i] = 0;
        __syncthreads();
        const u64 prefix = s_prefix, mask = s_mask;
        for (int row0 = 0; row0 < a.nboxes; row0 += a.rows_per_block) {
            const int rows = min(a.rows_per_block, a.nboxes - row0);
            v5_visit(a, pimg, row0, rows, tile, [&](unsigned idx, float s) {
                const u64 key = ((u64)f2key(s) << 32) | (u64)(0xFFFFFFFFu - idx);
                if ((key & mask) == prefix) atomicAdd(&dh[(unsigned)(key >> shift) & 255u], 1u);
            });
        }
        if (threadIdx.x == 0) {
            unsigned acc = 0, rem = s_remaining;
            int d = 255;
            for (; d > 0; --d) {
                if (acc + dh[d] >= rem) break;
                acc += dh[d];
            }
            s_remaining = rem - acc;
            s_prefix = prefix | ((u64)d << shift);
            s_mask = mask | (0xFFull << shift);
        }
        __syncthreads();
    }
    const u64 kth = s_prefix;
    u64* list = a.list + (size_t)img * kListCap;
    for (int row0 = 0; row0 < a.nboxes; row0 += a.rows_per_block) {
        const int rows = min(a.rows_per_block, a.nboxes - row0);
        v5_visit(a, pimg, row0, rows, tile, [&](unsigned idx, float s) {
            const u64 key = ((u64)f2key(s) << 32) | (u64)(0xFFFFFFFFu - idx);
            if (key >= kth) {
                const unsigned pos = atomicAdd(&s_count, 1u);
                if (pos < (unsigned)kListCap) list[pos] = key;
            }
        });
    }
    __syncthreads();
    if (threadIdx.x == 0) m->list_count = s_count;
}

// Rank sort of the candidate list (descending key), keep the first max_nms, and gather their boxes:
// xywh -> xyxy (yolov5_official/utils.py:591-598) plus the class offset cls*max_wh in fp32 (:180-181).
__global__ __launch_bounds__(256) void v5_rank_kernel(const V5Args a) {
    __shared__ u64 keys[kListCap];
    const int img = blockIdx.x;
    V5Meta* m = a.meta + img;
    const int n = (int)min(m->list_count, (unsigned)kListCap);
    const u64* list = a.list + (size_t)img * kListCap;
    for (int i = threadIdx.x; i < n; i += 256) keys[i] = list[i];
    __syncthreads();
    const int ns = min(n, a.max_nms);
    u64* sorted = a.sorted + (size_t)img * kMaxNms;
    float4* sbox = a.sbox + (size_t)img * kMaxNms;
    const float* pimg = a.pred + (size_t)img * a.nboxes * a.no;
    for (int i = threadIdx.x; i < n; i += 256) {
        const u64 k = keys[i];
        int rank = 0;
        for (int j = 0; j < n; ++j) rank += keys[j] > k ? 1 : 0;
        if (rank < ns) {
            sorted[rank] = k;
            const unsigned idx = 0xFFFFFFFFu - (unsigned)(k & 0xFFFFFFFFull);
            const unsigned box = idx / (unsigned)a.nc, cls = idx - box * (unsigned)a.nc;
            const float* p = pimg + (size_t)box * a.no;
            const float cx = p[0], cy = p[1], w = p[2], h = p[3];
            const float c = a.agnostic ? 0.f : (float)cls * a.max_wh;
            float4 b;
            b.x = (cx - w / 2.f) + c;
            b.y = (cy - h / 2.f) + c;
            b.z = (cx + w / 2.f) + c;
            b.w = (cy + h / 2.f) + c;
            sbox[rank] = b;
        }
    }
    if (threadIdx.x == 0) m->ns = (unsigned)ns;
}

constexpr int kNmsThreads = 1024;
__global__ __launch_bounds__(kNmsThreads) void v5_nms_kernel(const V5Args a) {
    __shared__ NmsShared sh;
    __shared__ u64 removed[kMaxNms / 64], kept[kMaxNms / 64];
    __shared__ int wbase[kMaxNms / 64 + 1];
    __shared__ float4 lbox[kNmsLdsBoxes];   // the sorted boxes, when they fit: every chunk re-reads the later ones
    const int img = blockIdx.x;
    const V5Meta* m = a.meta + img;
    const int ns = (int)m->ns;
    const float4* sbox = a.sbox + (size_t)img * kMaxNms;
    float* out = a.out_det + (size_t)img * a.max_det * 6;
    int count = 0;
    if (ns > 0) {
        if (ns <= kNmsLdsBoxes) {    // workgroup-uniform
            for (int s = threadIdx.x; s < ns; s += kNmsThreads) lbox[s] = sbox[s];
            __syncthreads();
            greedy_nms_block<false, kNmsThreads>(ns, a.iou, [&](int s) { return lbox[s]; }, sh, removed, kept);
        } else {
            greedy_nms_block<false, kNmsThreads>(ns, a.iou, [&](int s) { return sbox[s]; }, sh, removed, kept);
        }
        const int nwords = (ns + 63) >> 6;
        if (threadIdx.x == 0) {
            int acc = 0;
            for (int w = 0; w < nwords; ++w) { wbase[w] = acc; acc += __popcll(kept[w]); }
            wbase[nwords] = acc;
        }
        __syncthreads();
        count = min(wbase[nwords], a.max_det);
        const u64* sorted = a.sorted + (size_t)img * kMaxNms;
        const float* pimg = a.pred + (size_t)img * a.nboxes * a.no;
        for (int s = threadIdx.x; s < ns; s += kNmsThreads) {
            const u64 kw = kept[s >> 6];
            if (!((kw >> (s & 63)) & 1ull)) continue;
            const int pos = wbase[s >> 6] + __popcll(kw & ((1ull << (s & 63)) - 1ull));
            if (pos >= a.max_det) continue;
            const u64 k = sorted[s];
            const unsigned idx = 0xFFFFFFFFu - (unsigned)(k & 0xFFFFFFFFull);
            const unsigned box = idx / (unsigned)a.nc, cls = idx - box * (unsigned)a.nc;
            const float* p = pimg + (size_t)box * a.no;
            const float cx = p[0], cy = p[1], w = p[2], h = p[3];
            float* o = out + (size_t)pos * 6;
            o[0] = cx - w / 2.f;
            o[1] = cy - h / 2.f;
            o[2] = cx + w / 2.f;
            o[3] = cy + h / 2.f;
            o[4] = p[5 + cls] * p[4];
            o[5] = (float)cls;
        }
    }
    for (int e = count * 6 + threadIdx.x; e < a.max_det * 6; e += kNmsThreads) out[e] = 0.f;
    if (threadIdx.x == 0) a.out_count[img] = count;
}

size_t v5_ws_layout(int N, int nboxes, size_t* o_hist, size_t* o_meta, size_t* o_list, size_t* o_sorted,
                    size_t* o_sbox) {
    size_t off = 0;
    auto take = [&](size_t bytes) { size_t r = off; off += (bytes + 255) & ~(size_t)255; return r; };
    const size_t h = take((size_t)N * nboxes * sizeof(float));   // row maxima
    const size_t m = take((size_t)N * sizeof(V5Meta));
    const size_t l = take((size_t)N * kListCap * sizeof(u64));
    const size_t s = take((size_t)N * kMaxNms * sizeof(u64));
    const size_t b = take((size_t)N * kMaxNms * sizeof(float4));
    if (o_hist) { *o_hist = h; *o_meta = m; *o_list = l; *o_sorted = s; *o_sbox = b; }
    return off;
}

// ==================================================================================================
// Per-class NMS of the v2/v3/v4 detectors (utils/utils.py postprocess + nms)
// ==================================================================================================
struct FAArgs {
    const float* boxes;   // [N][n][4]
    const float* scores;  // [N][n][nc]
    int N, n, nc;
    float conf, thr;
    uint8_t* keep;        // [N][n]
    int32_t* cls;         // [N][n]
    float* best;          // [N][n]
    // workspace
    unsigned* count;      // [N][nc]
    unsigned* cursor;     // [N][nc]
    unsigned* seg;        // [N][n]  box indices grouped by class
    unsigned* sorted;     // [N][n]  same, sorted by descending score inside each class
    float4* sbox;         // [N][n]  the boxes in that order (one gather, then contiguous reads)
    unsigned* alive;      // [N][n]  second index list of the greedy kernel (the first one reuses seg)
    // long class segments: suppression bit matrix built by the whole chip, then walked by one workgroup
    unsigned* unit_prefix;   // [N * nc + 1]  running count of 64-row strips over the segments that got a matrix
    u64* mat_off;            // [N * nc]      word offset of the segment's matrix inside its image's region, ~0: none
    unsigned* queue;         // next strip to hand out
    u64* mat;                // [N][mat_words]
    u64 mat_words;           // per image; 0: no matrix region in the workspace
};

// argmax class (first maximum), threshold, per-class population count.  The 256 boxes of a workgroup are one
// contiguous run of the score matrix: it is staged through LDS in coalesced order (a thread reading its own nc values
// straight from memory touches a different cache line than each of its neighbours).
constexpr int kFaScoreMaxNc = 40;     // 256 * nc floats of LDS; wider score rows take the direct path

// atomicAdd(&counter[key], 1) for the lanes with `on`, one atomic per distinct key of the wave instead of one per
// lane (thousands of boxes of an image share a handful of argmax classes: per-lane atomics on those few words run
// one after the other).  Returns the value the lane's own add would have returned, in lane order within the wave.
__device__ __forceinline__ unsigned wave_count_add(unsigned* counter, bool on, unsigned key) {
    const int lane = threadIdx.x & 63;
    unsigned result = 0;
    u64 todo = __ballot(on);
    while (todo) {
        const int leader = __ffsll((long long)todo) - 1;
        const unsigned k0 = (unsigned)__builtin_amdgcn_readlane((int)key, leader);
        const u64 same = __ballot(on && key == k0);
        unsigned base = 0;
        if (lane == leader) base = atomicAdd(&counter[k0], (unsigned)__popcll(same));
        base = (unsigned)__builtin_amdgcn_readlane((int)base, leader);
        if (on && key == k0) result = base + (unsigned)__popcll(same & ((1ull << lane) - 1ull));
        todo &= ~same;
    }
    return result;
}

__global__ __launch_bounds__(256) void fa_score_kernel(const FAArgs a) {
    __shared__ float tile[256 * kFaScoreMaxNc];
    const long long total = (long long)a.N * a.n;
    const long long g0 = (long long)blockIdx.x * 256;
    const long long gid = g0 + threadIdx.x;
    const bool staged = a.nc <= kFaScoreMaxNc;
    if (staged) {
        const int rows = (int)min((long long)256, total - g0);
        const float* src = a.scores + (size_t)g0 * a.nc;
        for (int i = threadIdx.x; i < rows * a.nc; i += 256) tile[i] = src[i];
        __syncthreads();
    }
    if (gid >= total) return;
    const int img = (int)(gid / a.n);
    const float* s = staged ? tile + threadIdx.x * a.nc : a.scores + (size_t)gid * a.nc;
    float best = s[0];
    int bj = 0;
    for (int j = 1; j < a.nc; ++j) {
        const float v = s[j];
        if (v > best) { best = v; bj = j; }
    }
    a.cls[gid] = bj;
    a.best[gid] = best;
    a.keep[gid] = 0;
    wave_count_add(a.count, best >= a.conf, (unsigned)img * (unsigned)a.nc + (unsigned)bj);
}

__device__ __forceinline__ unsigned fa_class_offset(const unsigned* count, int cls) {
    unsigned off = 0;
    for (int c = 0; c < cls; ++c) off += count[c];
    return off;
}

__global__ __launch_bounds__(256) void fa_scatter_kernel(const FAArgs a) {
    const long long gid = (long long)blockIdx.x * 256 + threadIdx.x;
    const bool in = gid < (long long)a.N * a.n;
    const bool on = in && a.best[in ? gid : 0] >= a.conf;
    const int img = in ? (int)(gid / a.n) : 0;
    const int c = on ? a.cls[gid] : 0;
    const unsigned slot = wave_count_add(a.cursor, on, (unsigned)img * (unsigned)a.nc + (unsigned)c);   // (every lane takes part)
    if (!on) return;
    const unsigned box = (unsigned)(gid - (long long)img * a.n);
    a.seg[(size_t)img * a.n + fa_class_offset(a.count + (size_t)img * a.nc, c) + slot] = box;
}

// Segments of kFaSortMin .. kFaSortMax boxes are sorted in LDS by one workgroup (bitonic network on the 64-bit keys:
// n log^2 n compare-exchanges where counting makes n^2 comparisons); shorter ones, and the rare longer ones, are ranked
// by counting.
constexpr int kFaSortMin = 512, kFaSortMax = 8192;

// grid (chunk, class, image): rank of each member inside its class segment.
// Order: descending score; equal scores: higher box index first (ascending stable sort, reversed).
__global__ __launch_bounds__(256) void fa_rank_kernel(const FAArgs a) {
    __shared__ u64 tk[256];
    const int img = blockIdx.z, c = blockIdx.y;
    const unsigned* count = a.count + (size_t)img * a.nc;
    const unsigned cnt = count[c];
    const unsigned m = blockIdx.x * 256 + threadIdx.x;
    if (blockIdx.x * 256 >= cnt) return;
    if (cnt >= (unsigned)kFaSortMin && cnt <= (unsigned)kFaSortMax) return;     // fa_sort_kernel's
    const unsigned off = fa_class_offset(count, c);
    const unsigned* seg = a.seg + (size_t)img * a.n + off;
    const float* best = a.best + (size_t)img * a.n;
    u64 mykey = 0;
    unsigned mybox = 0;
    if (m < cnt) {
        mybox = seg[m];
        mykey = ((u64)f2key(best[mybox]) << 32) | mybox;
    }
    unsigned rank = 0;
    for (unsigned t0 = 0; t0 < cnt; t0 += 256) {
        const unsigned j = t0 + threadIdx.x;
        __syncthreads();
        if (j < cnt) {
            const unsigned b = seg[j];
            tk[threadIdx.x] = ((u64)f2key(best[b]) << 32) | b;
        }
        __syncthreads();
        const unsigned lim = min(256u, cnt - t0);
        for (unsigned q = 0; q < lim; ++q) rank += tk[q] > mykey ? 1u : 0u;
    }
    if (m < cnt) {
        a.sorted[(size_t)img * a.n + off + rank] = mybox;
        a.sbox[(size_t)img * a.n + off + rank] = reinterpret_cast<const float4*>(a.boxes)[(size_t)img * a.n + mybox];
    }
}

// grid (class, image), 64 KB of dynamic LDS.  Same order as the counting kernel: descending 64-bit key
// (orderable(score) << 32 | box), keys are unique; the padding up to a power of two carries key 0 and sinks to the end.
__global__ __launch_bounds__(1024) void fa_sort_kernel(const FAArgs a) {
    extern __shared__ __attribute__((aligned(16))) u64 s_keys[];
    const int img = blockIdx.y, c = blockIdx.x, tid = threadIdx.x;
    const unsigned* count = a.count + (size_t)img * a.nc;
    const int cnt = (int)count[c];
    if (cnt < kFaSortMin || cnt > kFaSortMax) return;
    const unsigned off = fa_class_offset(count, c);
    const unsigned* seg = a.seg + (size_t)img * a.n + off;
    const float* best = a.best + (size_t)img * a.n;
    const int P = 1 << (32 - __clz(cnt - 1));          // next power of two >= cnt
    for (int i = tid; i < P; i += 1024) {
        u64 key = 0ull;
        if (i < cnt) {
            const unsigned b = seg[i];
            key = ((u64)f2key(best[b]) << 32) | b;
        }
        s_keys[i] = key;
    }
    __syncthreads();
    for (int k = 2; k <= P; k <<= 1)
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int t = tid; t < (P >> 1); t += 1024) {
                const int lo = ((t & ~(j - 1)) << 1) | (t & (j - 1)), hi = lo + j;
                const bool desc = (lo & k) == 0;
                const u64 x = s_keys[lo], y = s_keys[hi];
                if ((x < y) == desc) { s_keys[lo] = y; s_keys[hi] = x; }
            }
            __syncthreads();
        }
    for (int i = tid; i < cnt; i += 1024) {
        const unsigned b = (unsigned)s_keys[i];
        a.sorted[(size_t)img * a.n + off + i] = b;
        a.sbox[(size_t)img * a.n + off + i] = reinterpret_cast<const float4*>(a.boxes)[(size_t)img * a.n + b];
    }
}

constexpr int kFaMaxWords = 2048;  // up to 131072 boxes per class segment

// Class segments of at least this many boxes take the matrix path when their matrix fits the image's region (shorter
// ones are a few rounds of the walk; from about a thousand boxes on, mostly kept, the walk takes hundreds of
// microseconds: every round's survivors test up to 64 kept boxes one after the other).
constexpr int kFaMatrixMin = 256;
constexpr u64 kFaNoMatrix = ~0ull;

// The matrix of a segment of cnt boxes (W = ceil(cnt / 64) blocks of 64): strips of 64 rows, strip rb holding the
// tiles of column blocks rb .. W - 1 one after the other, a tile = 64 words, one per COLUMN: word j has bit i set when
// box 64 rb + i suppresses box 64 cb + j (i before j on the diagonal tile).  Only the upper triangle exists.
__device__ __forceinline__ u64 fa_strip_words(int W, int rb) { return 64ull * ((u64)rb * (u64)W - (u64)rb * (u64)(rb - 1) / 2ull); }
static inline u64 fa_matrix_words(int cnt) {
    const u64 W = ((u64)cnt + 63) >> 6;
    return 64ull * (W * (W + 1) / 2ull);
}

__device__ __forceinline__ float vmax_raw(float x, float y) { float r; asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(x), "v"(y)); return r; }
__device__ __forceinline__ float vmin_raw(float x, float y) { float r; asm("v_min_f32 %0, %1, %2" : "=v"(r) : "v"(x), "v"(y)); return r; }
__device__ __forceinline__ float vfloor28_raw(float x) { float r; asm("v_max_f32 %0, 0x10fd87b6, %1" : "=v"(r) : "v"(x)); return r; }   // max(1e-28f, x)

// v_cmp_class_f32 as a lane mask (every lane of the wave must be active)
__device__ __forceinline__ u64 class_mask(float x, int classes) {
    u64 m;
    asm("v_cmp_class_f32 %0, %1, %2" : "=s"(m) : "v"(x), "s"(classes));
    return m;
}

// One 64 x 64 tile: lane j holds column box bj and returns its word, bit i = suppresses<true>(row box i, bj).
// The verdict is !(fl(inter / uni) <= thr).  Rounding is monotone and thr is a float, so inter <= thr uni (in the reals)
// means survive and inter >= nextup(thr) uni means suppressed; both are decided without the division wherever inter is
// outside thr (1 -+ 2^-21) uni, products rounded (2^-24 each, with the bound on the product keeping them out of the
// subnormals).  If any column of a group of eight rows is inside that band, holds a NaN, or thr uni is not a positive
// normal number (uni = +0 and uni < 0 aside), the wave evaluates the reference expression for the group: the result is the
// division's, bit for bit.
// inter and uni are computed exactly as suppresses<true> does (the bare v_max / v_min differ from fmaxf / fminf only for
// signalling NaNs, which end up undecided).  Rows are read from LDS at wave-uniform addresses.  GENERIC: the diagonal
// tile (row i tests the columns beyond i) or a strip of fewer than 64 rows.
// A lane collects its column's verdicts one bit per row: bits = 2 bits + verdict is one add-with-carry with the verdict
// mask as the carry-in (rows 0..31 and 32..63 in a register each, first row in the top bit).  All mask logic is scalar
// and kept short: the scalar unit is shared by the four SIMDs of a CU.
template <bool GENERIC>
__device__ __forceinline__ u64 tile_column_word(const float4* rbox, const float* rarea, const float4& bj, float aj, float thr,
                                                float thr_lo, float thr_hi, u64 cols, bool diag, int nrow) {
    unsigned half[2];
    const bool thr_nonneg = thr >= 0.f;
#pragma unroll 1
    for (int hf = 0; hf < 2; ++hf) {
        unsigned bits = 0u;
#pragma unroll 1
        for (int g = 0; g < 4; ++g) {
            const int i0 = hf * 32 + g * 8;
            u64 words[8];
            u64 undecided = thr_nonneg ? 0ull : ~0ull;      // (a negative or NaN threshold: the reference expression throughout)
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const float4 bi = rbox[i0 + k];
                const float ai = rarea[i0 + k];
                const float xx1 = vmax_raw(bi.x, bj.x), yy1 = vmax_raw(bi.y, bj.y);
                const float xx2 = vmin_raw(bi.z, bj.z), yy2 = vmin_raw(bi.w, bj.w);
                const float w = vfloor28_raw(xx2 - xx1), h = vfloor28_raw(yy2 - yy1);
                const float inter = w * h;
                const float uni = ai + aj - inter;
                const float p_lo = thr_lo * uni, p_hi = thr_hi * uni;
                const u64 m_hi = __builtin_amdgcn_ballot_w64(inter >= p_hi);
                const u64 m_lo = __builtin_amdgcn_ballot_w64(inter <= p_lo);
                const u64 m_ok = __builtin_amdgcn_ballot_w64(p_lo >= 1e-30f);
                // Boxes without area (clipped away at the image border) are common and leave uni at or below zero.
                // uni = +0: inter / uni is +inf or NaN, suppressed either way -- and inter >= p_hi = 0 says so already.
                // uni negative and finite (two such boxes overlapping along a line: inter is a tiny positive number; a
                // finite uni means a finite inter): the quotient is negative or -0, which is <= thr for any thr >= 0 --
                // survives.
                const u64 m_zero = class_mask(uni, 0x40);         // +0
                const u64 m_neg = class_mask(uni, 0x18);          // -normal | -subnormal
                words[k] = m_hi & ~m_neg;
                undecided |= ~(((m_hi | m_lo) & m_ok) | m_zero | m_neg);
            }
            if (__builtin_expect((undecided & cols) != 0ull, 0)) {
#pragma unroll
                for (int k = 0; k < 8; ++k)
                    words[k] = __builtin_amdgcn_ballot_w64(suppresses<true>(rbox[i0 + k], rarea[i0 + k], bj, aj, thr));
            }
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const int i = i0 + k;
                const u64 live = !GENERIC ? cols : i >= nrow ? 0ull : diag ? cols & ~((2ull << i) - 1ull) : cols;
                const u64 word = words[k] & live;
                asm("v_addc_co_u32 %0, vcc, %0, %0, %1" : "+v"(bits) : "s"(word) : "vcc");
            }
        }
        half[hf] = __builtin_bitreverse32(bits);
    }
    return ((u64)half[1] << 32) | (u64)half[0];
}

// One workgroup: which segments get a matrix (in class order while the image's region has room), where, and the
// running strip count the matrix kernel's queue is searched with.
__global__ __launch_bounds__(1024) void fa_plan_kernel(const FAArgs a) {
    __shared__ unsigned s_scan[1024];
    __shared__ unsigned s_carry;
    const int tid = threadIdx.x;
    if (tid == 0) { s_carry = 0; *a.queue = 0; }
    __syncthreads();
    for (int base = 0; base < a.N; base += 1024) {
        const int img = base + tid;
        unsigned units = 0;
        if (img < a.N) {
            u64 used = 0;
            for (int c = 0; c < a.nc; ++c) {
                const unsigned cnt = a.count[(size_t)img * a.nc + c];
                const u64 W = ((u64)cnt + 63) >> 6, words = 64ull * (W * (W + 1) / 2ull);
                const bool sel = cnt >= (unsigned)kFaMatrixMin && used + words <= a.mat_words;
                a.mat_off[(size_t)img * a.nc + c] = sel ? used : kFaNoMatrix;
                if (sel) { used += words; units += (unsigned)W; }
            }
        }
        s_scan[tid] = units;
        __syncthreads();
        for (int d = 1; d < 1024; d <<= 1) {
            const unsigned v = tid >= d ? s_scan[tid - d] : 0u;
            __syncthreads();
            s_scan[tid] += v;
            __syncthreads();
        }
        unsigned run = s_scan[tid] - units + s_carry;
        if (img < a.N)
            for (int c = 0; c < a.nc; ++c) {
                const size_t s = (size_t)img * a.nc + c;
                a.unit_prefix[s] = run;
                if (a.mat_off[s] != kFaNoMatrix) run += (a.count[s] + 63u) >> 6;
            }
        __syncthreads();
        if (tid == 1023) s_carry += s_scan[1023];
        __syncthreads();
    }
    if (tid == 0) a.unit_prefix[(size_t)a.N * a.nc] = s_carry;
}

// Strips are handed out through a counter; a workgroup's four waves share a strip's 64 row boxes (LDS, read at
// wave-uniform addresses) and take its tiles in turn: lane j holds column box j and collects the verdicts of the 64 rows
// against it.  Every pair gets the verdict of the expression the round-by-round walk below evaluates, operands in the
// same order.
__global__ __launch_bounds__(256) void fa_matrix_kernel(const FAArgs a) {
    __shared__ float4 s_rbox[64];
    __shared__ float s_rarea[64];
    __shared__ unsigned s_unit;
    const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int S = a.N * a.nc;
    const unsigned total = a.unit_prefix[S];
    const float thr_lo = a.thr * 0.99999952316284f, thr_hi = a.thr * 1.00000047683716f;   // thr (1 -+ 2^-21)
    while (true) {
        if (tid == 0) s_unit = atomicAdd(a.queue, 1u);
        __syncthreads();
        const unsigned u = (unsigned)__builtin_amdgcn_readfirstlane((int)s_unit);
        if (u >= total) break;           // (the same u for the whole workgroup)
        int lo = 0, hi = S;              // unit_prefix[lo] <= u < unit_prefix[hi]
        while (hi - lo > 1) {
            const int mid = (lo + hi) >> 1;
            if (a.unit_prefix[mid] <= u) lo = mid; else hi = mid;
        }
        const int seg = lo, rb = (int)(u - a.unit_prefix[seg]);
        const int img = seg / a.nc, c = seg - img * a.nc;
        const unsigned* count = a.count + (size_t)img * a.nc;
        const int cnt = (int)count[c];
        const float4* sbox = a.sbox + (size_t)img * a.n + fa_class_offset(count, c);
        const int W = (cnt + 63) >> 6, r0 = rb * 64;
        u64* strip = a.mat + (size_t)img * a.mat_words + a.mat_off[seg] + fa_strip_words(W, rb);
        if (tid < 64) {
            const float4 b = r0 + tid < cnt ? sbox[r0 + tid] : make_float4(0.f, 0.f, 0.f, 0.f);
            s_rbox[tid] = b;
            s_rarea[tid] = box_area(b);
        }
        __syncthreads();
        const int nrow = min(64, cnt - r0);
        for (int cb = rb + wv; cb < W; cb += 4) {
            const int cj = cb * 64 + lane;
            const bool okj = cj < cnt;
            const float4 bj = okj ? sbox[cj] : make_float4(0.f, 0.f, 0.f, 0.f);
            const float aj = box_area(bj);
            const u64 cols = __ballot(okj);
            const bool diag = cb == rb;
            const u64 word = diag || nrow < 64
                                 ? tile_column_word<true>(s_rbox, s_rarea, bj, aj, a.thr, thr_lo, thr_hi, cols, diag, nrow)
                                 : tile_column_word<false>(s_rbox, s_rarea, bj, aj, a.thr, thr_lo, thr_hi, cols, false, 64);
            strip[(size_t)(cb - rb) * 64 + lane] = word;
        }
        __syncthreads();                 // the row boxes and s_unit are rewritten next
    }
}

constexpr int kFaNmsThreads = 1024;   // four threads per later box when a chunk's survivors are applied (long class segments)

// One workgroup per (class, image) segment, boxes in descending score order.  The greedy rule of utils/utils.py:15-42
// in rounds: the first 64 boxes still alive form a chunk, its 64 x 64 "i suppresses j" matrix is built in parallel
// and resolved serially by one wave (readlane'd row masks), then every other box still alive is tested against the
// chunk's kept boxes and the survivors are written, in order, to the second index list.  A segment of thousands of
// boxes of which a few hundred survive shrinks geometrically: the work is the sum of the alive counts, not
// (segment length) x (chunks).
__global__ __launch_bounds__(kFaNmsThreads) void fa_nms_kernel(const FAArgs a) {
    __shared__ NmsShared sh;
    __shared__ unsigned s_entry[64];
    __shared__ unsigned s_wcnt[kFaNmsThreads / 64];
    __shared__ unsigned s_out;
    const int img = blockIdx.y, c = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const unsigned* count = a.count + (size_t)img * a.nc;
    const int cnt = (int)count[c];
    if (cnt == 0) return;
    const unsigned off = fa_class_offset(count, c);
    const unsigned* sorted = a.sorted + (size_t)img * a.n + off;
    const float4* sbox = a.sbox + (size_t)img * a.n + off;
    unsigned* cur = a.seg + (size_t)img * a.n + off;      // (the class lists are no longer needed once ranked)
    unsigned* nxt = a.alive + (size_t)img * a.n + off;
    uint8_t* keep = a.keep + (size_t)img * a.n;
    if (a.mat_words && a.mat_off[(size_t)img * a.nc + c] != kFaNoMatrix) {
        // The segment's suppression matrix exists: walk its strips.  removed[w] collects the columns of block w that a kept
        // box of an earlier strip suppresses.
        __shared__ u64 s_removed[kFaMaxWords];
        const int W = (cnt + 63) >> 6;
        for (int w = tid; w < W; w += kFaNmsThreads) s_removed[w] = 0ull;
        __syncthreads();
        const u64* mat = a.mat + (size_t)img * a.mat_words + a.mat_off[(size_t)img * a.nc + c];
        // One barrier per strip: while waves 1..15 apply what strip rb keeps to the blocks beyond rb + 1, the first wave
        // applies it to block rb + 1 itself and goes on to resolve that block's diagonal tile -- the only serial chain.
        // A strip's words are requested one strip ahead (they do not depend on what is kept).
        __shared__ u64 s_kept[2];
        constexpr int kOthers = kFaNmsThreads - 64;       // threads of waves 1..15: one column each per pass
        constexpr int kPre = 6;                           // passes held in registers (segments of up to 92 blocks entirely)
        // The diagonal tile of a strip, by the first wave.  A position depends on the positions before it only, so the
        // greedy set K is the one fixed point of K = open & ~{j : column j has a row of K}, and iterating from any start
        // settles at least one more leading position per pass (64 passes at most).
        auto resolve_diag = [&](int rb, u64 colw, u64 removed) {
            const int nrow = min(64, cnt - rb * 64);
            const u64 valid = nrow == 64 ? ~0ull : ((1ull << nrow) - 1ull);
            const u64 open = valid & ~removed;
            u64 kp = open;
            for (int pass = 0; pass < 64; ++pass) {
                const u64 nk = open & ~__ballot((colw & kp) != 0ull);
                if (nk == kp) break;
                kp = nk;
            }
            if ((kp >> lane) & 1ull) keep[sorted[rb * 64 + lane]] = 1;
            if (lane == 0) s_kept[rb & 1] = kp;
        };
        // first wave: diagonal word of strip rb + 1 and the word of block rb + 1 in strip rb; the others: columns of the
        // blocks beyond rb + 1 in strip rb
        u64 w0_diag = 0ull, w0_next = 0ull, pre[kPre];
        auto request = [&](int rb) {          // what iteration rb consumes
            const u64* strip = mat + fa_strip_words(W, rb);
            if (wv == 0) {
                if (rb + 1 < W) {
                    w0_next = strip[64 + lane];
                    w0_diag = mat[fa_strip_words(W, rb + 1) + lane];
                }
            } else {
                const int later = (W - rb - 2) * 64;      // columns beyond block rb + 1
#pragma unroll
                for (int q = 0; q < kPre; ++q) {
                    const int col = tid - 64 + q * kOthers;
                    pre[q] = col < later ? strip[128 + col] : 0ull;
                }
            }
        };
        if (wv == 0) resolve_diag(0, mat[lane], 0ull);
        request(0);
        __syncthreads();
        for (int rb = 0; rb + 1 < W; ++rb) {
            const u64* strip = mat + fa_strip_words(W, rb);
            const u64 kp = s_kept[rb & 1];
            const u64 c_diag = w0_diag, c_next = w0_next;
            u64 cur[kPre];
#pragma unroll
            for (int q = 0; q < kPre; ++q) cur[q] = pre[q];
            request(rb + 1);
            if (wv == 0) {
                const u64 removed = s_removed[rb + 1] | __ballot((c_next & kp) != 0ull);
                resolve_diag(rb + 1, c_diag, removed);
            } else {
                const int later = (W - rb - 2) * 64;
#pragma unroll
                for (int q = 0; q < kPre; ++q) {
                    const int col = tid - 64 + q * kOthers;
                    if (col - lane < later) {              // (the same for the whole wave)
                        const u64 hit = __ballot((cur[q] & kp) != 0ull);
                        if (lane == 0 && hit) s_removed[rb + 2 + (col >> 6)] |= hit;
                    }
                }
                for (int col = tid - 64 + kPre * kOthers; col - lane < later; col += kOthers) {
                    const u64 hit = __ballot((strip[128 + col] & kp) != 0ull);
                    if (lane == 0 && hit) s_removed[rb + 2 + (col >> 6)] |= hit;
                }
            }
            __syncthreads();
        }
        return;
    }
    for (int s = tid; s < cnt; s += kFaNmsThreads) cur[s] = (unsigned)s;
    __syncthreads();
    int L = cnt;
    while (L > 0) {
        const int cn = min(64, L);
        if (tid < 64) {
            const unsigned e = tid < cn ? cur[tid] : 0u;
            const float4 b = tid < cn ? sbox[e] : make_float4(0.f, 0.f, 0.f, 0.f);
            s_entry[tid] = e;
            sh.cbox[tid] = b;
            sh.carea[tid] = box_area(b);
        }
        if (tid == 0) s_out = 0;
        __syncthreads();
        {   // the chunk's matrix: thread (i = tid / 16, part = tid % 16) tests j = 4 part .. 4 part + 3, j > i
            constexpr int PARTS = kFaNmsThreads / 64, PER = 64 / PARTS;
            const int i = tid / PARTS, part = tid % PARTS;
            u64 bits = 0;
            if (i < cn) {
                const float4 bi = sh.cbox[i];
                const float ai = sh.carea[i];
#pragma unroll
                for (int jj = 0; jj < PER; ++jj) {
                    const int j = part * PER + jj;
                    if (j > i && j < cn && suppresses<true>(bi, ai, sh.cbox[j], sh.carea[j], a.thr)) bits |= 1ull << j;
                }
            }
#pragma unroll
            for (int d = 1; d < PARTS; d <<= 1) bits |= __shfl_xor(bits, d);
            if (part == 0) sh.rowmask[i] = bits;
        }
        __syncthreads();
        if (tid < 64) {          // serial resolve: every box of the chunk was alive when the round began
            const u64 mine = tid < cn ? sh.rowmask[tid] : 0ull;
            const u64 valid = cn == 64 ? ~0ull : ((1ull << cn) - 1ull);
            u64 rem = 0, kp = 0, cand = valid;
            while (cand) {
                const int i = __ffsll((long long)cand) - 1;
                kp |= 1ull << i;
                const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)mine, i);
                const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(mine >> 32), i);
                rem |= ((u64)hi << 32) | (u64)lo;
                cand = ~rem & valid & ~(((1ull << i) << 1) - 1ull);
            }
            if ((kp >> tid) & 1ull) keep[sorted[s_entry[tid]]] = 1;
            if (tid == 0) sh.kept_word = kp;
        }
        __syncthreads();
        const u64 kp = sh.kept_word;
        // every other box still alive against the chunk's kept boxes; survivors to the other list, order preserved
        for (int t0 = 64; t0 < L; t0 += kFaNmsThreads) {
            const int pos = t0 + tid;
            bool survives = false;
            unsigned e = 0;
            if (pos < L) {
                e = cur[pos];
                const float4 bj = sbox[e];
                const float aj = box_area(bj);
                survives = true;
                u64 m = kp;
                while (m) {
                    const int i = __ffsll((long long)m) - 1;
                    m &= m - 1;
                    if (suppresses<true>(sh.cbox[i], sh.carea[i], bj, aj, a.thr)) { survives = false; break; }
                }
            }
            const u64 bal = __ballot(survives);
            if (lane == 0) s_wcnt[wv] = (unsigned)__popcll(bal);
            __syncthreads();
            unsigned before = 0, total = 0;
#pragma unroll
            for (int w = 0; w < kFaNmsThreads / 64; ++w) {
                const unsigned cw = s_wcnt[w];
                if (w < wv) before += cw;
                total += cw;
            }
            const unsigned base = s_out;
            if (survives) nxt[base + before + (unsigned)__popcll(bal & ((1ull << lane) - 1ull))] = e;
            __syncthreads();
            if (tid == 0) s_out = base + total;
        }
        __syncthreads();
        L = (int)s_out;
        unsigned* t = cur; cur = nxt; nxt = t;
        __syncthreads();
    }
}

struct FaLayout {
    size_t count, cursor, seg, sorted, sbox, alive, prefix, matoff, queue;   // byte offsets
    size_t base_bytes;   // everything the walk needs
    size_t mat;          // offset of the matrix region
    u64 mat_words;       // words per image of a full-size region
    size_t bytes;        // with the full-size region
};

FaLayout fa_ws_layout(int N, int n, int nc) {
    FaLayout L;
    size_t off = 0;
    auto take = [&](size_t bytes) { size_t r = off; off += (bytes + 255) & ~(size_t)255; return r; };
    L.count = take((size_t)N * nc * sizeof(unsigned));
    L.cursor = take((size_t)N * nc * sizeof(unsigned));
    L.seg = take((size_t)N * n * sizeof(unsigned));
    L.sorted = take((size_t)N * n * sizeof(unsigned));
    L.sbox = take((size_t)N * n * sizeof(float4));      // sorted boxes
    L.alive = take((size_t)N * n * sizeof(unsigned));   // second index list
    L.prefix = take(((size_t)N * nc + 1) * sizeof(unsigned));
    L.matoff = take((size_t)N * nc * sizeof(u64));
    L.queue = take(sizeof(unsigned));
    L.base_bytes = off;
    // matrix region per image: what ONE class holding every box needs (the upper triangle of n x n bits; any split of the
    // boxes over classes needs less), so with a full-size workspace no segment ever goes without
    L.mat_words = n >= kFaMatrixMin ? fa_matrix_words(n) : 0;
    L.mat = take((size_t)N * L.mat_words * sizeof(u64));
    L.bytes = off;
    return L;
}

// box_iou: yolov5_official/utils.py:601-611 (clamp(0), no epsilon)
__global__ void box_iou_kernel(const float4* a, int na, const float4* b, int nb, float* out) {
    const long long gid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= (long long)na * nb) return;
    const int i = (int)(gid / nb), j = (int)(gid - (long long)i * nb);
    const float4 p = a[i], q = b[j];
    const float a1 = (p.z - p.x) * (p.w - p.y), a2 = (q.z - q.x) * (q.w - q.y);
    const float w = fmaxf(fminf(p.z, q.z) - fmaxf(p.x, q.x), 0.f);
    const float h = fmaxf(fminf(p.w, q.w) - fmaxf(p.y, q.y), 0.f);
    const float inter = w * h;
    out[gid] = inter / (a1 + a2 - inter);
}


// process_batch (yolov5_official/utils.py:1103-1123) for a batch of images, one workgroup per image.  Per prediction:
// the label of its own class with the largest IoU (first one on ties, NaN if any IoU is NaN -- both as torch.max);
// a prediction whose best IoU exceeds iouv[0] claims that label unless an earlier prediction of the class already
// did (the reference walks predictions in index order), and only a claiming prediction gets its row best > iouv.
// The reference's early exit once every label is claimed changes nothing: no later prediction can claim by then.
__global__ __launch_bounds__(256) void process_batch_kernel(const float* det, const int32_t* count, int max_det,
                                                             const float* labels, const int32_t* label_start,
                                                             const float* iouv, int niou, uint8_t* correct) {
    __shared__ float s_best[kMaxNms];
    __shared__ int s_lab[kMaxNms];
    __shared__ float s_cls[kMaxNms];
    const int img = blockIdx.x;
    const int n = min(count[img], max_det);
    const int l0 = label_start[img], l1 = label_start[img + 1];
    const float* d = det + (size_t)img * max_det * 6;
    uint8_t* out = correct + (size_t)img * max_det * niou;
    for (int j = threadIdx.x; j < n; j += 256) {
        const float4 p = make_float4(d[j * 6 + 0], d[j * 6 + 1], d[j * 6 + 2], d[j * 6 + 3]);
        const float pc = d[j * 6 + 5];
        const float a1 = (p.z - p.x) * (p.w - p.y);
        float best = -INFINITY;
        int lab = -1;
        bool nan = false;
        for (int l = l0; l < l1; ++l) {
            const float* q = labels + (size_t)l * 5;
            if (q[0] != pc) continue;
            const float a2 = (q[3] - q[1]) * (q[4] - q[2]);
            const float w = fmaxf(fminf(p.z, q[3]) - fmaxf(p.x, q[1]), 0.f);
            const float h = fmaxf(fminf(p.w, q[4]) - fmaxf(p.y, q[2]), 0.f);
            const float inter = w * h;
            const float iou = inter / (a1 + a2 - inter);
            if (iou != iou) nan = true;
            if (iou > best) { best = iou; lab = l; }
        }
        s_best[j] = nan ? NAN : best;
        s_lab[j] = lab;
        s_cls[j] = pc;
    }
    __syncthreads();
    const float thr0 = iouv[0];
    for (int j = threadIdx.x; j < max_det; j += 256) {
        bool claims = false;
        float best = 0.f;
        if (j < n) {
            best = s_best[j];
            const int lab = s_lab[j];
            claims = lab >= 0 && best > thr0;
            if (claims) {
                const float pc = s_cls[j];
                for (int k = 0; k < j; ++k)
                    if (s_lab[k] == lab && s_cls[k] == pc && s_best[k] > thr0) { claims = false; break; }
            }
        }
        for (int t = 0; t < niou; ++t) out[(size_t)j * niou + t] = claims && best > iouv[t] ? 1 : 0;
    }
}

// VOC detection-to-ground-truth matching (data/vocapi_evaluator.py:284-318) for every (class, image) group of a whole
// evaluation in one launch.  The reference walks one class's detections in descending confidence and, per detection,
// takes the ground-truth box of its image with the largest overlap (float64, first maximum, NaN wins as in numpy):
// overlap > thr and the box is not "difficult": true positive if no earlier detection claimed the box, else false
// positive; overlap not above thr: false positive; a difficult box: neither.  Claims never cross images, and whether a
// detection claims depends only on the earlier detections whose best box is the same one, so each detection finds its
// best box independently (phase 1) and then looks back over its group (phase 2).  One workgroup per group.
__global__ __launch_bounds__(256) void voc_match_kernel(const double* bb, const int32_t* members, const int32_t* grp_start,
                                                         const double* gt, const uint8_t* difficult,
                                                         const int32_t* gt_start, double thr, double* ws_ov,
                                                         int32_t* ws_j, uint8_t* flags) {
    const int g = blockIdx.x;
    const int m0 = grp_start[g], m1 = grp_start[g + 1];
    const int g0 = gt_start[g], g1 = gt_start[g + 1];
    for (int i = m0 + threadIdx.x; i < m1; i += 256) {
        const int k = members[i];
        const double x1 = bb[4 * k], y1 = bb[4 * k + 1], x2 = bb[4 * k + 2], y2 = bb[4 * k + 3];
        double best = -INFINITY;
        int jbest = -1;
        bool nan = false;
        for (int j = g0; j < g1; ++j) {
            const double* q = gt + 4 * (size_t)j;
            const double iw = fmax(fmin(q[2], x2) - fmax(q[0], x1), 0.0);
            const double ih = fmax(fmin(q[3], y2) - fmax(q[1], y1), 0.0);
            const double inter = iw * ih;
            const double uni = (x2 - x1) * (y2 - y1) + (q[2] - q[0]) * (q[3] - q[1]) - inter;
            const double ov = inter / uni;
            if (ov != ov && !nan) { nan = true; jbest = j; }      // numpy's max / argmax return the first NaN
            if (!nan && ov > best) { best = ov; jbest = j; }
        }
        ws_ov[i] = nan ? (double)NAN : best;
        ws_j[i] = jbest;
    }
    __syncthreads();
    for (int i = m0 + threadIdx.x; i < m1; i += 256) {
        const double ov = ws_ov[i];
        const int j = ws_j[i];
        uint8_t f = 2;                                   // false positive
        if (ov > thr) {
            if (difficult[j]) f = 0;
            else {
                f = 1;
                for (int e = m0; e < i; ++e)
                    if (ws_j[e] == j && ws_ov[e] > thr) { f = 2; break; }
            }
        }
        flags[members[i]] = f;
    }
}

}  // namespace

extern "C" size_t ydet_nms_v5_workspace_bytes(int N, int nboxes, int no) {
    (void)no;
    if (N <= 0 || nboxes <= 0) return 0;
    return v5_ws_layout(N, nboxes, nullptr, nullptr, nullptr, nullptr, nullptr);
}

static int nms_v5_run(const float* pred, int N, int nboxes, int no, float conf_thres, float iou_thres,
                      int multi_label, int agnostic, int max_nms, int max_det, float max_wh,
                      const uint8_t* class_mask, const float* row_max, float* out_det, int32_t* out_count,
                      void* workspace, size_t workspace_bytes, void* stream) {
    YDET_CHECK_ARG(pred && out_det && out_count && workspace, "nms_v5: null pointer");
    YDET_CHECK_ARG(N > 0 && nboxes > 0 && no >= 6, "nms_v5: N=%d nboxes=%d no=%d", N, nboxes, no);
    YDET_CHECK_ARG(no <= kTileFloats, "nms_v5: no=%d exceeds %d", no, kTileFloats);
    YDET_CHECK_ARG(((uintptr_t)pred & 3) == 0, "nms_v5: pred must be 4-byte aligned");
    YDET_CHECK_ARG(max_nms >= 1 && max_nms <= kMaxNms, "nms_v5: max_nms=%d (1..%d)", max_nms, kMaxNms);
    YDET_CHECK_ARG(max_det >= 1 && max_det <= max_nms, "nms_v5: max_det=%d (1..max_nms)", max_det);
    const int nc = no - 5;
    YDET_CHECK_ARG((long long)nboxes * nc < 0xFFFFFFFFLL, "nms_v5: nboxes*nc does not fit 32 bits");
    YDET_CHECK_ARG(((uintptr_t)workspace & 255) == 0, "nms_v5: workspace must be 256-byte aligned");
    size_t oh, om, ol, os, ob;
    const size_t need = v5_ws_layout(N, nboxes, &oh, &om, &ol, &os, &ob);
    if (workspace_bytes < need) return fail(YDET_ERR_WORKSPACE, "nms_v5: workspace %zu < %zu bytes", workspace_bytes, need);
    hipStream_t s = as_stream(stream);
    char* ws = (char*)workspace;
    V5Args a;
    a.pred = pred; a.N = N; a.nboxes = nboxes; a.no = no; a.nc = nc;
    a.rows_per_block = kTileFloats / no;
    if (a.rows_per_block > 128) a.rows_per_block = 128;
    a.scan_rows = kScanTile / no < kScanRows ? kScanTile / no : kScanRows;
    a.conf = conf_thres; a.iou = iou_thres;
    a.multi = (multi_label && nc > 1) ? 1 : 0;
    a.agnostic = agnostic ? 1 : 0;
    a.max_nms = max_nms; a.max_det = max_det; a.max_wh = max_wh; a.class_mask = class_mask;
    a.base16 = f2key_host(conf_thres) >> 16;
    a.row_max_ws = (float*)(ws + oh); a.meta = (V5Meta*)(ws + om); a.list = (u64*)(ws + ol);
    // a caller-supplied row maximum is over all classes: with a class filter the scan has to run
    const bool scan = row_max == nullptr || class_mask != nullptr;
    a.row_max = scan ? a.row_max_ws : row_max;
    a.sorted = (u64*)(ws + os); a.sbox = (float4*)(ws + ob);
    a.out_det = out_det; a.out_count = out_count;
    YDET_CHECK_ARG(N <= 65535, "nms_v5: N=%d exceeds 65535", N);
    if (scan) {
        const dim3 grid2((unsigned)((nboxes + a.scan_rows - 1) / a.scan_rows), (unsigned)N);
        hipLaunchKernelGGL(v5_rowmax_kernel, grid2, dim3(256), 0, s, a);
    }
    hipLaunchKernelGGL(v5_pick_kernel, dim3((unsigned)N), dim3(kPickThreads), 0, s, a);
    hipLaunchKernelGGL(v5_exact_select_kernel, dim3((unsigned)N), dim3(256), 0, s, a);
    hipLaunchKernelGGL(v5_rank_kernel, dim3((unsigned)N), dim3(256), 0, s, a);
    hipLaunchKernelGGL(v5_nms_kernel, dim3((unsigned)N), dim3(kNmsThreads), 0, s, a);
    YDET_CHECK_LAUNCH("nms_v5 kernels");
    return YDET_OK;
}

extern "C" int ydet_nms_v5(const float* pred, int N, int nboxes, int no, float conf_thres, float iou_thres,
                           int multi_label, int agnostic, int max_nms, int max_det, float max_wh,
                           const uint8_t* class_mask, const float* row_max, float* out_det, int32_t* out_count,
                           void* workspace, size_t workspace_bytes, void* stream) {
    clear_error();
    return nms_v5_run(pred, N, nboxes, no, conf_thres, iou_thres, multi_label, agnostic, max_nms, max_det, max_wh, class_mask,
                      row_max, out_det, out_count, workspace, workspace_bytes, stream);
}

// ---- Cluster-Weighted DIoU-NMS (yolov5_official/utils.py:463-548): the same candidate selection, another ending ------
namespace ydet {
struct ClusterArgs {
    const float4* sbox; const u64* sorted; const unsigned* ns; const float* pred;
    int N, nboxes, no, nc, stride, max_box;
    float thr;
    u64* ge; u64* gt; float* out; int32_t* out_count;
};
size_t cluster_diou_ws_bytes(int N, int max_box);
int launch_cluster_diou(ClusterArgs a, void* ws, hipStream_t stream);
}  // namespace ydet

extern "C" size_t ydet_nms_cluster_diou_workspace_bytes(int N, int nboxes, int no, int max_box) {
    (void)no;
    if (N <= 0 || nboxes <= 0 || max_box <= 0) return 0;
    return v5_ws_layout(N, nboxes, nullptr, nullptr, nullptr, nullptr, nullptr) + cluster_diou_ws_bytes(N, max_box);
}

extern "C" int ydet_nms_cluster_diou(const float* pred, int N, int nboxes, int no, float conf_thres, float iou_thres,
                                     int max_box, int agnostic, const uint8_t* class_mask, float* out_rows,
                                     int32_t* out_count, void* workspace, size_t workspace_bytes, void* stream) {
    clear_error();
    YDET_CHECK_ARG(pred && out_rows && out_count && workspace, "nms_cluster_diou: null pointer");
    YDET_CHECK_ARG(N > 0 && N <= 65535 && nboxes > 0 && no >= 6 && no <= kTileFloats, "nms_cluster_diou: N=%d nboxes=%d no=%d", N, nboxes, no);
    YDET_CHECK_ARG(max_box >= 1 && max_box <= 2048, "nms_cluster_diou: max_box=%d (1..2048)", max_box);
    YDET_CHECK_ARG(iou_thres > 0.f, "nms_cluster_diou: iou_thres must be positive");
    YDET_CHECK_ARG(((uintptr_t)workspace & 255) == 0 && ((uintptr_t)pred & 3) == 0, "nms_cluster_diou: alignment");
    const int nc = no - 5;
    YDET_CHECK_ARG((long long)nboxes * nc < 0xFFFFFFFFLL, "nms_cluster_diou: nboxes*nc does not fit 32 bits");
    size_t oh, om, ol, os, ob;
    const size_t base = v5_ws_layout(N, nboxes, &oh, &om, &ol, &os, &ob);
    const size_t need = base + cluster_diou_ws_bytes(N, max_box);
    if (workspace_bytes < need) return fail(YDET_ERR_WORKSPACE, "nms_cluster_diou: workspace %zu < %zu bytes", workspace_bytes, need);
    hipStream_t s = as_stream(stream);
    char* ws = (char*)workspace;
    V5Args a;
    a.pred = pred; a.N = N; a.nboxes = nboxes; a.no = no; a.nc = nc;
    a.rows_per_block = kTileFloats / no;
    if (a.rows_per_block > 128) a.rows_per_block = 128;
    a.scan_rows = kScanTile / no < kScanRows ? kScanTile / no : kScanRows;
    a.conf = conf_thres; a.iou = iou_thres;
    a.multi = nc > 1 ? 1 : 0;                                     // :475 overrides the argument
    a.agnostic = agnostic ? 1 : 0;
    a.max_nms = max_box; a.max_det = max_box; a.max_wh = 4096.f; a.class_mask = class_mask;
    a.base16 = f2key_host(conf_thres) >> 16;
    a.row_max_ws = (float*)(ws + oh); a.meta = (V5Meta*)(ws + om); a.list = (u64*)(ws + ol);
    a.row_max = a.row_max_ws;
    a.sorted = (u64*)(ws + os); a.sbox = (float4*)(ws + ob);
    a.out_det = nullptr; a.out_count = nullptr;
    const dim3 grid2((unsigned)((nboxes + a.scan_rows - 1) / a.scan_rows), (unsigned)N);
    hipLaunchKernelGGL(v5_rowmax_kernel, grid2, dim3(256), 0, s, a);
    hipLaunchKernelGGL(v5_pick_kernel, dim3((unsigned)N), dim3(kPickThreads), 0, s, a);
    hipLaunchKernelGGL(v5_exact_select_kernel, dim3((unsigned)N), dim3(256), 0, s, a);
    hipLaunchKernelGGL(v5_rank_kernel, dim3((unsigned)N), dim3(256), 0, s, a);
    YDET_CHECK_LAUNCH("nms_cluster_diou selection kernels");
    ydet::ClusterArgs c;
    c.sbox = a.sbox; c.sorted = a.sorted; c.ns = reinterpret_cast<const unsigned*>(a.meta); c.pred = pred;
    c.N = N; c.nboxes = nboxes; c.no = no; c.nc = nc; c.stride = kMaxNms; c.max_box = max_box;
    c.thr = iou_thres; c.ge = nullptr; c.gt = nullptr; c.out = out_rows; c.out_count = out_count;
    return ydet::launch_cluster_diou(c, ws + base, s);
}

// ---- the same operator for callers that must not allocate ---------------------------------------------------------
// The workspace of ydet_nms_v5 (row maxima, per-image records, candidate list, sorted keys and boxes) lives in a
// device array that is part of the library's code object: every GPU the library runs on holds its own copy until the
// process ends.  In host code the array's NAME is only a handle: its device address is asked for once per device
// (hipGetSymbolAddress, with that device current) and kept; kernels are given that pointer like any other workspace.
constexpr size_t kStaticNmsBytes = (size_t)YDET_NMS_V5_STATIC_BYTES;
static __device__ __align__(256) unsigned char g_nms_scratch[kStaticNmsBytes];

namespace {
std::mutex g_scratch_mutex;
void* g_scratch_ptr[kMaxDevices];      // per device: address of that device's copy of g_nms_scratch (null: not asked yet)

int nms_scratch_for_current_device(void** out) {
    int dev = -1;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDevices)
        return fail(YDET_ERR_LAUNCH, "nms_v5_static: cannot tell the current device (%d)", dev);
    std::lock_guard<std::mutex> lock(g_scratch_mutex);
    if (!g_scratch_ptr[dev]) {
        void* p = nullptr;
        const hipError_t e = hipGetSymbolAddress(&p, HIP_SYMBOL(g_nms_scratch));
        if (e != hipSuccess || !p)
            return fail(YDET_ERR_LAUNCH, "nms_v5_static: hipGetSymbolAddress failed on device %d: %s", dev, hipGetErrorString(e));
        g_scratch_ptr[dev] = p;
    }
    *out = g_scratch_ptr[dev];
    return YDET_OK;
}
}  // namespace

extern "C" int ydet_nms_v5_static(const float* pred, int N, int nboxes, int no, float conf_thres, float iou_thres,
                                  int multi_label, int agnostic, int max_nms, int max_det, float max_wh,
                                  const uint8_t* class_mask, const float* row_max, float* out_det, int32_t* out_count,
                                  void* stream) {
    clear_error();
    YDET_CHECK_ARG(N > 0 && nboxes > 0, "nms_v5_static: N=%d nboxes=%d", N, nboxes);
    const size_t need = v5_ws_layout(N, nboxes, nullptr, nullptr, nullptr, nullptr, nullptr);
    if (need > kStaticNmsBytes)
        return fail(YDET_ERR_WORKSPACE,
                    "nms_v5_static: %d images x %d boxes need %zu bytes of scratch, the built-in buffer has %zu "
                    "(use ydet_nms_v5 with a caller-owned workspace)", N, nboxes, need, kStaticNmsBytes);
    void* ws = nullptr;
    const int rc = nms_scratch_for_current_device(&ws);
    if (rc != YDET_OK) return rc;
    // nothing in the workspace has to start at zero: every word a kernel reads was written earlier in the same call
    return nms_v5_run(pred, N, nboxes, no, conf_thres, iou_thres, multi_label, agnostic, max_nms, max_det, max_wh, class_mask,
                      row_max, out_det, out_count, ws, kStaticNmsBytes, stream);
}

extern "C" size_t ydet_nms_perclass_workspace_bytes(int N, int n, int nc) {
    if (N <= 0 || n <= 0 || nc <= 0) return 0;
    return fa_ws_layout(N, n, nc).bytes;
}

extern "C" size_t ydet_nms_perclass_workspace_bytes_min(int N, int n, int nc) {
    if (N <= 0 || n <= 0 || nc <= 0) return 0;
    return fa_ws_layout(N, n, nc).base_bytes;
}

extern "C" int ydet_nms_perclass(const float* boxes, const float* scores, int N, int n, int nc, float conf_thresh,
                                 float nms_thresh, uint8_t* out_keep, int32_t* out_cls, float* out_score,
                                 void* workspace, size_t workspace_bytes, void* stream) {
    clear_error();
    YDET_CHECK_ARG(boxes && scores && out_keep && out_cls && out_score && workspace, "nms_perclass: null pointer");
    YDET_CHECK_ARG(N > 0 && n > 0 && nc > 0, "nms_perclass: N=%d n=%d nc=%d", N, n, nc);
    YDET_CHECK_ARG(n <= kFaMaxWords * 64, "nms_perclass: n=%d exceeds %d", n, kFaMaxWords * 64);
    YDET_CHECK_ARG(N <= 65535 && nc <= 65535, "nms_perclass: N or nc exceeds 65535");
    YDET_CHECK_ARG(((uintptr_t)boxes & 15) == 0, "nms_perclass: boxes must be 16-byte aligned");
    YDET_CHECK_ARG(((uintptr_t)workspace & 255) == 0, "nms_perclass: workspace must be 256-byte aligned");
    const FaLayout L = fa_ws_layout(N, n, nc);
    if (workspace_bytes < L.base_bytes)
        return fail(YDET_ERR_WORKSPACE, "nms_perclass: workspace %zu < %zu bytes", workspace_bytes, L.base_bytes);
    hipStream_t s = as_stream(stream);
    char* ws = (char*)workspace;
    FAArgs a;
    a.boxes = boxes; a.scores = scores; a.N = N; a.n = n; a.nc = nc; a.conf = conf_thresh; a.thr = nms_thresh;
    a.keep = out_keep; a.cls = out_cls; a.best = out_score;
    a.count = (unsigned*)(ws + L.count); a.cursor = (unsigned*)(ws + L.cursor);
    a.seg = (unsigned*)(ws + L.seg); a.sorted = (unsigned*)(ws + L.sorted);
    a.sbox = (float4*)(ws + L.sbox); a.alive = (unsigned*)(ws + L.alive);
    a.unit_prefix = (unsigned*)(ws + L.prefix); a.mat_off = (u64*)(ws + L.matoff); a.queue = (unsigned*)(ws + L.queue);
    a.mat = (u64*)(ws + L.mat);
    // a workspace of ydet_nms_perclass_workspace_bytes holds the full matrix region; a smaller one (at least the base
    // part) gets what is left per image, down to none: segments without a matrix are walked round by round
    a.mat_words = 0;
    if (L.mat_words && workspace_bytes > L.mat)
        a.mat_words = std::min<u64>(L.mat_words, (u64)((workspace_bytes - L.mat) / sizeof(u64) / (size_t)N));
    if (a.mat_words < fa_matrix_words(kFaMatrixMin)) a.mat_words = 0;
    if (hipMemsetAsync(ws + L.count, 0, L.seg - L.count, s) != hipSuccess)
        return fail(YDET_ERR_LAUNCH, "nms_perclass: hipMemsetAsync failed");
    const long long total = (long long)N * n;
    const unsigned g1 = (unsigned)((total + 255) / 256);
    hipLaunchKernelGGL(fa_score_kernel, dim3(g1), dim3(256), 0, s, a);
    hipLaunchKernelGGL(fa_scatter_kernel, dim3(g1), dim3(256), 0, s, a);
    hipLaunchKernelGGL(fa_rank_kernel, dim3((unsigned)((n + 255) / 256), (unsigned)nc, (unsigned)N), dim3(256), 0, s, a);
    if (n >= kFaSortMin)
        hipLaunchKernelGGL(fa_sort_kernel, dim3((unsigned)nc, (unsigned)N), dim3(1024), kFaSortMax * sizeof(u64), s, a);
    if (a.mat_words) {
        hipLaunchKernelGGL(fa_plan_kernel, dim3(1), dim3(1024), 0, s, a);
        hipLaunchKernelGGL(fa_matrix_kernel, dim3(2048), dim3(256), 0, s, a);
    }
    hipLaunchKernelGGL(fa_nms_kernel, dim3((unsigned)nc, (unsigned)N), dim3(kFaNmsThreads), 0, s, a);
    YDET_CHECK_LAUNCH("nms_perclass kernels");
    return YDET_OK;
}

extern "C" int ydet_box_iou(const float* a, int na, const float* b, int nb, float* iou, void* stream) {
    clear_error();
    YDET_CHECK_ARG(na >= 0 && nb >= 0, "box_iou: negative size");
    if (na == 0 || nb == 0) return YDET_OK;
    YDET_CHECK_ARG(a && b && iou, "box_iou: null pointer");
    YDET_CHECK_ARG(((uintptr_t)a & 15) == 0 && ((uintptr_t)b & 15) == 0, "box_iou: boxes must be 16-byte aligned");
    const long long total = (long long)na * nb;
    hipLaunchKernelGGL(box_iou_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, as_stream(stream),
                       (const float4*)a, na, (const float4*)b, nb, iou);
    YDET_CHECK_LAUNCH("box_iou_kernel");
    return YDET_OK;
}

extern "C" int ydet_process_batch(const float* det, const int32_t* count, int B, int max_det, const float* labels,
                                  const int32_t* label_start, const float* iouv, int niou, uint8_t* correct,
                                  void* stream) {
    clear_error();
    YDET_CHECK_ARG(B >= 0 && max_det >= 1 && max_det <= kMaxNms && niou >= 1, "process_batch: B=%d max_det=%d niou=%d (max_det <= %d)",
                   B, max_det, niou, kMaxNms);
    if (B == 0) return YDET_OK;
    YDET_CHECK_ARG(det && count && label_start && iouv && correct, "process_batch: null pointer");
    hipLaunchKernelGGL(process_batch_kernel, dim3((unsigned)B), dim3(256), 0, as_stream(stream), det, count, max_det, labels,
                       label_start, iouv, niou, correct);
    YDET_CHECK_LAUNCH("process_batch_kernel");
    return YDET_OK;
}

extern "C" size_t ydet_voc_match_workspace_bytes(int n_det) {
    return n_det <= 0 ? 0 : (size_t)n_det * (sizeof(double) + sizeof(int32_t));
}

extern "C" int ydet_voc_match(const double* boxes, int n_det, const int32_t* members, const int32_t* group_start,
                              int n_groups, const double* gt_boxes, const uint8_t* gt_difficult,
                              const int32_t* gt_start, double ovthresh, uint8_t* flags, void* workspace,
                              size_t workspace_bytes, void* stream) {
    clear_error();
    YDET_CHECK_ARG(n_det >= 0 && n_groups >= 0, "voc_match: n_det=%d n_groups=%d", n_det, n_groups);
    if (n_det == 0 || n_groups == 0) return YDET_OK;
    YDET_CHECK_ARG(boxes && members && group_start && gt_start && flags, "voc_match: null pointer");
    if (workspace_bytes < ydet_voc_match_workspace_bytes(n_det) || workspace == nullptr)
        return fail(YDET_ERR_WORKSPACE, "voc_match: workspace of %zu bytes, need %zu", workspace_bytes,
                    ydet_voc_match_workspace_bytes(n_det));
    YDET_CHECK_ARG(((uintptr_t)workspace & 7) == 0, "voc_match: workspace must be 8-byte aligned");
    double* ws_ov = reinterpret_cast<double*>(workspace);
    int32_t* ws_j = reinterpret_cast<int32_t*>(ws_ov + n_det);
    hipLaunchKernelGGL(voc_match_kernel, dim3((unsigned)n_groups), dim3(256), 0, as_stream(stream), boxes, members,
                       group_start, gt_boxes, gt_difficult, gt_start, ovthresh, ws_ov, ws_j, flags);
    YDET_CHECK_LAUNCH("voc_match_kernel");
    return YDET_OK;
}
