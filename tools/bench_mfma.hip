// Sustained bf16 MFMA rate of the whole chip with operands in registers (debugging aid, not part of libydet.so):
//   hipcc --offload-arch=gfx950 -O2 tools/bench_mfma.hip -o tools/bin/bench_mfma && tools/bin/bench_mfma
// Every SIMD runs WAVES waves that issue independent MFMAs back to back for about a tenth of a second; printed per
// instruction shape: TFLOP/s over the wall time of the launch (the last column is the first wave's own span over that
// time: below the clock when the waves of a SIMD take turns).  What the matrix pipes sustain under power limits is the
// ceiling the convolution kernels are measured against.  Measured on MI355X: 16x16x32 2.10 / 2.31 / 2.39 PFLOP/s at
// 1 / 2 / 4 waves per SIMD, 32x32x16 2.04 PFLOP/s at any (it draws more power: 1.99 GHz).  The MFMAs accumulate IN PLACE
// through inline asm: written with the builtin, this loop came out with accumulators rotated through overlapping
// register ranges (a[24:27] = mfma(.., a[22:25])) and reached 1.3 PFLOP/s -- a kernel's ISA is worth a look for that.
#include <hip/hip_runtime.h>
#include <cstdio>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;

template <int SHAPE>
__global__ __launch_bounds__(256) void mfma_kernel(float* out, int reps, unsigned long long* cyc) {
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(0.001f * (threadIdx.x + i)); b[i] = (__bf16)(0.002f * (threadIdx.x % 7 + i)); }
    const unsigned long long t0 = __builtin_readcyclecounter();
    float s = 0.f;
    if constexpr (SHAPE == 16) {
        f32x4 c[8];
        for (int i = 0; i < 8; ++i) c[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int r = 0; r < reps; ++r) {
#pragma unroll
            for (int i = 0; i < 8; ++i)      // (inline asm: accumulate in place -- the builtin's loop got rotated, overlapping registers)
                asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(c[i]) : "v"(a), "v"(b));
        }
        for (int i = 0; i < 8; ++i) s += c[i][0];
    } else {
        f32x16 c[4];
        for (int i = 0; i < 4; ++i)
            for (int e = 0; e < 16; ++e) c[i][e] = 0.f;
        for (int r = 0; r < reps; ++r) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
                asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(c[i]) : "v"(a), "v"(b));
        }
        for (int i = 0; i < 4; ++i) s += c[i][0];
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
    if (s == 12345.678f) out[0] = s;
}

template <int SHAPE>
void run(const char* name, int waves_per_simd, int reps) {
    float* out;
    unsigned long long* cyc;
    (void)hipMalloc(&out, 4);
    (void)hipMalloc(&cyc, 8);
    const int blocks = 256 * waves_per_simd;          // 256 CUs, four waves (one per SIMD) per workgroup
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    for (int warm = 0; warm < 2; ++warm) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL(mfma_kernel<SHAPE>, dim3(blocks), dim3(256), 0, 0, out, reps, cyc);
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
    }
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, e0, e1);
    unsigned long long c = 0;
    (void)hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    // 16x16x32: 8 MFMAs of 16384 FLOP per repetition; 32x32x16: 4 of 32768
    const double flop = (double)blocks * 4.0 * reps * 131072.0;
    printf("%-12s %d waves/SIMD: %7.1f TFLOP/s  (%.1f ms, cycle counter %.2f GHz)\n", name, waves_per_simd, flop / ms / 1e9, ms,
           (double)c / ms / 1e6);
    (void)hipFree(out);
    (void)hipFree(cyc);
}

int main() {
    for (int w : {1, 2, 4}) {
        run<16>("16x16x32", w, 400000 / w);
        run<32>("32x32x16", w, 400000 / w);
    }
    return 0;
}
