// What a hand-written streaming kernel gets out of HBM on this GPU, on the working sets of the HBM-bound layers
// (stand-alone tool, not part of libydet.so):
//   hipcc --offload-arch=gfx950 -O3 tools/bench_membw.hip -o build/bench_membw && build/bench_membw [quick]
//
// Kernels: 16 bytes per lane, 256-thread workgroups, a persistent grid-stride loop; per wave-instruction 1 KiB contiguous.
//   copy   c[i] = a[i]                 (1 read, 1 write: a layer that moves its map in and out)
//   add2   c[i] = a[i] ^ b[i]          (2 reads, 1 write: a 1x1 layer over a concatenation / with a residual)
//   read   sum of a[i] (one store per thread at the end)      write   c[i] = const
// Swept: workgroups per CU (4 waves each), loads in flight per lane (unroll), nontemporal loads / stores.
// Working sets (bf16 maps of YOLOv5s at batch 64): 105 MB (80x80x128 channels), chained over four buffers so that a
// launch reads what the launch before it wrote and nothing is re-read before 420 MB have gone by; the same ping-ponged
// between two buffers (210 MB: inside the 256 MB memory-side cache); 550 MB -> 550 MB (the decoded rows of a batch).
// Times are hipEvent intervals around bursts of back-to-back launches; GB/s counts bytes read + bytes written.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <vector>

#define CK(x)                                                                                  \
    do {                                                                                       \
        hipError_t e_ = (x);                                                                   \
        if (e_ != hipSuccess) {                                                                \
            fprintf(stderr, "%s:%d %s: %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_)); \
            exit(1);                                                                           \
        }                                                                                      \
    } while (0)

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
enum { COPY = 0, ADD2 = 1, READ = 2, WRITE = 3 };

template <bool NT>
__device__ __forceinline__ u32x4 ld(const u32x4* p) {
    if constexpr (NT) return __builtin_nontemporal_load(p);
    else return *p;
}
template <bool NT>
__device__ __forceinline__ void st(u32x4* p, u32x4 v) {
    if constexpr (NT) __builtin_nontemporal_store(v, p);
    else *p = v;
}

// U: 16-byte loads in flight per lane and operand before the first use
template <int MODE, int U, bool NTL, bool NTS>
__global__ __launch_bounds__(256) void stream_kernel(const u32x4* __restrict__ a, const u32x4* __restrict__ b, u32x4* __restrict__ c,
                                                     long long n16) {
    const long long stride = (long long)gridDim.x * 256;
    u32x4 acc = {0u, 0u, 0u, 0u};
    for (long long i0 = (long long)blockIdx.x * 256 + threadIdx.x; i0 < n16; i0 += stride * U) {
        u32x4 va[U], vb[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const long long i = i0 + u * stride;
            if (MODE != WRITE && i < n16) va[u] = ld<NTL>(a + i);
            if (MODE == ADD2 && i < n16) vb[u] = ld<NTL>(b + i);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const long long i = i0 + u * stride;
            if (i >= n16) continue;
            if (MODE == COPY) st<NTS>(c + i, va[u]);
            else if (MODE == ADD2) st<NTS>(c + i, va[u] ^ vb[u]);
            else if (MODE == READ) acc ^= va[u];
            else st<NTS>(c + i, u32x4{1u, 2u, 3u, (unsigned)i});
        }
    }
    if (MODE == READ && (acc[0] ^ acc[1] ^ acc[2] ^ acc[3]) == 0x12345678u) c[threadIdx.x] = acc;   // (keeps the loads alive)
}

struct Variant {
    const char* name;
    int mode, unroll, ntl, nts;
    void (*fn)(const u32x4*, const u32x4*, u32x4*, long long);
};

template <int MODE, int U, bool NTL, bool NTS>
Variant make(const char* name) {
    return Variant{name, MODE, U, NTL, NTS, stream_kernel<MODE, U, NTL, NTS>};
}

static double time_burst(const Variant& v, int grid, const std::vector<u32x4*>& bufs, size_t bytes, int reps, int chain) {
    // chain: number of buffers walked (launch k reads buffer k % chain, writes (k + 1) % chain; ADD2 also reads (k + 2) % chain)
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    const long long n16 = (long long)(bytes / 16);
    auto launch = [&](int k) {
        const u32x4* a = bufs[k % chain];
        u32x4* c = bufs[(k + 1) % chain];
        const u32x4* b = bufs[(k + 2) % chain];
        hipLaunchKernelGGL(v.fn, dim3(grid), dim3(256), 0, 0, a, b, c, n16);
    };
    for (int k = 0; k < 2 * chain; ++k) launch(k);
    std::vector<float> ms;
    for (int t = 0; t < 5; ++t) {
        CK(hipEventRecord(e0, 0));
        for (int k = 0; k < reps; ++k) launch(k);
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
        float m = 0.f;
        CK(hipEventElapsedTime(&m, e0, e1));
        ms.push_back(m / reps);
    }
    std::sort(ms.begin(), ms.end());
    CK(hipEventDestroy(e0));
    CK(hipEventDestroy(e1));
    return ms[2];
}

int main(int argc, char** argv) {
    const bool quick = argc > 1 && !strcmp(argv[1], "quick");
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    printf("# %s, %d CUs\n", prop.name, cus);
    const size_t MAP = 64ull * 80 * 80 * 128 * 2;        // 104.9 MB
    const size_t ROWS = 64ull * 25200 * 85 * 4;          // 548.4 MB
    std::vector<u32x4*> maps(4), big(2);
    for (auto& p : maps) { CK(hipMalloc((void**)&p, MAP)); CK(hipMemset(p, 0x3c, MAP)); }
    for (auto& p : big) { CK(hipMalloc((void**)&p, ROWS)); CK(hipMemset(p, 0x3c, ROWS)); }

    std::vector<Variant> vars = {
        make<COPY, 1, false, false>("copy u1"),        make<COPY, 2, false, false>("copy u2"),
        make<COPY, 4, false, false>("copy u4"),        make<COPY, 8, false, false>("copy u8"),
        make<COPY, 4, true, false>("copy u4 ntL"),     make<COPY, 4, false, true>("copy u4 ntS"),
        make<COPY, 4, true, true>("copy u4 ntLS"),     make<COPY, 8, true, true>("copy u8 ntLS"),
        make<ADD2, 2, false, false>("add2 u2"),        make<ADD2, 4, false, false>("add2 u4"),
        make<ADD2, 4, true, true>("add2 u4 ntLS"),     make<ADD2, 4, false, true>("add2 u4 ntS"),
        make<READ, 4, false, false>("read u4"),        make<READ, 8, false, false>("read u8"),
        make<READ, 8, true, false>("read u8 ntL"),     make<WRITE, 4, false, false>("write u4"),
        make<WRITE, 4, false, true>("write u4 ntS"),
    };
    const int wgs_per_cu[] = {1, 2, 4, 8};
    struct Set { const char* name; std::vector<u32x4*>* bufs; size_t bytes; int chain; };
    Set sets[] = {{"105MB chain4 (cold: 420 MB between re-reads)", &maps, MAP, 4},
                  {"105MB pingpong (210 MB: memory-side cache)", &maps, MAP, 2},
                  {"550MB -> 550MB", &big, ROWS, 2}};
    for (const Set& s : sets) {
        printf("\n## %s\n%-16s", s.name, "variant");
        for (int w : wgs_per_cu) printf("  %2d wg/CU", w);
        printf("   (GB/s, bytes read + written)\n");
        for (const Variant& v : vars) {
            if (quick && (v.unroll == 1 || v.unroll == 2)) continue;
            if (v.mode == ADD2 && s.chain < 3) continue;
            printf("%-16s", v.name);
            for (int w : wgs_per_cu) {
                const double ms = time_burst(v, cus * w, *s.bufs, s.bytes, quick ? 8 : 16, s.chain);
                const double moved = (v.mode == COPY ? 2.0 : v.mode == ADD2 ? 3.0 : 1.0) * (double)s.bytes;
                printf("  %8.0f", moved / (ms * 1e-3) / 1e9);
            }
            printf("\n");
            fflush(stdout);
        }
    }
    return 0;
}
