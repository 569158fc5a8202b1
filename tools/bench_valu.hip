// Issue cost of single VALU instructions on gfx950 (debugging aid, not part of libydet.so):
//   hipcc --offload-arch=gfx950 -O2 tools/bench_valu.hip -o tools/bin/bench_valu && tools/bin/bench_valu
// One wave per SIMD runs 8 independent chains x 4096 repetitions of one instruction between two s_memtime stamps;
// printed: shader cycles per wave-instruction.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define CHAINS 8
#define REPS 4096

#define KERNEL(NAME, ASM, CONSTRAINT)                                                              \
    __global__ void NAME(unsigned long long* out, float seed) {                                  \
        float v[CHAINS];                                                                          \
        for (int i = 0; i < CHAINS; ++i) v[i] = seed + 0.001f * (threadIdx.x + i);                \
        const unsigned long long t0 = __builtin_readcyclecounter();                               \
        for (int r = 0; r < REPS; ++r) {                                                          \
            _Pragma("unroll") for (int i = 0; i < CHAINS; ++i) asm volatile(ASM : CONSTRAINT(v[i])); \
        }                                                                                         \
        const unsigned long long t1 = __builtin_readcyclecounter();                               \
        float s = 0.f;                                                                            \
        for (int i = 0; i < CHAINS; ++i) s += v[i];                                               \
        /* the issue arbiter favours the oldest wave: the figure that counts is first start to last end of a workgroup */ \
        if (blockIdx.x == 0 && (threadIdx.x & 63) == 0) {                                         \
            atomicMin(&out[2], t0);                                                               \
            atomicMax(&out[3], t1);                                                               \
        }                                                                                         \
        if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = t1 - t0;                                \
        if (s == 12345.678f) out[1] = 1;                                                          \
    }

KERNEL(k_mul_f32, "v_mul_f32 %0, 0x3f800100, %0", "+v")
KERNEL(k_exp_f32, "v_exp_f32 %0, %0", "+v")
KERNEL(k_rcp_f32, "v_rcp_f32 %0, %0", "+v")
KERNEL(k_exp_f16, "v_exp_f16 %0, %0", "+v")
KERNEL(k_rcp_f16, "v_rcp_f16 %0, %0", "+v")
KERNEL(k_cvt_f16, "v_cvt_f16_f32 %0, %0", "+v")
KERNEL(k_pk_mul_f16, "v_pk_mul_f16 %0, %0, %0", "+v")
KERNEL(k_fma_f32, "v_fma_f32 %0, %0, %0, %0", "+v")
KERNEL(k_sqrt_f32, "v_sqrt_f32 %0, %0", "+v")
KERNEL(k_log_f32, "v_log_f32 %0, %0", "+v")

// one transcendental followed by three multiplies on other registers: do the two kinds share an issue slot?
KERNEL(k_mix_exp_3mul, "v_exp_f32 %0, %0\n v_mul_f32 v100, 0x3f800100, v100\n v_mul_f32 v101, 0x3f800100, v101\n v_mul_f32 v102, 0x3f800100, v102", "+v")

typedef void (*kern_t)(unsigned long long*, float);

int main() {
    unsigned long long* d;
    if (hipMalloc(&d, 32) != hipSuccess) return 1;
    struct { const char* name; kern_t k; } ks[] = {
        {"v_mul_f32", k_mul_f32}, {"v_fma_f32", k_fma_f32}, {"v_exp_f32", k_exp_f32}, {"v_rcp_f32", k_rcp_f32},
        {"v_sqrt_f32", k_sqrt_f32}, {"v_log_f32", k_log_f32}, {"v_exp_f16", k_exp_f16}, {"v_rcp_f16", k_rcp_f16},
        {"v_cvt_f16_f32", k_cvt_f16}, {"v_pk_mul_f16", k_pk_mul_f16}, {"exp + 3 mul", k_mix_exp_3mul}};
    for (auto& e : ks) {
        for (int waves = 1; waves <= 4; waves *= 2) {   // waves per SIMD
            unsigned long long h[4] = {0, 0, ~0ull, 0};
            hipMemcpy(d, h, 32, hipMemcpyHostToDevice);
            hipLaunchKernelGGL(e.k, dim3(waves > 4 ? 512 : 256), dim3(256 * (waves > 4 ? 4 : waves)), 0, 0, d, 1.0f);
            if (hipDeviceSynchronize() != hipSuccess) return 2;
            hipMemcpy(h, d, 32, hipMemcpyDeviceToHost);
            printf("%-14s %d wave(s)/SIMD: oldest wave %.2f ticks per own instruction; SIMD throughput %.2f ticks per wave-instruction\n",
                   e.name, waves, (double)h[0] / (double)(CHAINS * REPS),
                   (double)(h[3] - h[2]) / (double)(CHAINS * REPS) / (double)waves);
        }
    }
    return 0;
}
