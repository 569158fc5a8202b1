// Shared helpers of libydet.so (host + device).  gfx950 only.
#pragma once
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdint.h>
#include <stdio.h>

#include <type_traits>

#include "ydet.h"

namespace ydet {

// Records the text returned by ydet_last_error() on this thread and returns `code`.
int fail(int code, const char* fmt, ...);
void clear_error();

#define YDET_CHECK_ARG(cond, ...)                                   \
    do {                                                            \
        if (!(cond)) return ::ydet::fail(YDET_ERR_ARG, __VA_ARGS__); \
    } while (0)

#define YDET_CHECK_LAUNCH(what)                                                                  \
    do {                                                                                         \
        hipError_t e__ = hipGetLastError();                                                      \
        if (e__ != hipSuccess)                                                                   \
            return ::ydet::fail(YDET_ERR_LAUNCH, "%s: %s", what, hipGetErrorString(e__));        \
    } while (0)

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;

static inline size_t dtype_size(int dt) { return dt == YDET_BF16 ? 2 : 4; }
static inline bool dtype_ok(int dt) { return dt == YDET_F32 || dt == YDET_BF16; }

__device__ __forceinline__ float to_f32(float v) { return v; }
__device__ __forceinline__ float to_f32(__bf16 v) { return (float)v; }
template <typename T>
__device__ __forceinline__ T from_f32(float v);
template <>
__device__ __forceinline__ float from_f32<float>(float v) { return v; }
template <>
__device__ __forceinline__ __bf16 from_f32<__bf16>(float v) { return (__bf16)v; }

// Workgroup id remap that hands each XCD (blocks are dealt round-robin over 8 XCDs) a contiguous
// range of logical ids, so tiles that share operands share an L2.  Bijective for every nwg.
__device__ __forceinline__ unsigned xcd_remap(unsigned bid, unsigned nwg) {
    const unsigned q = nwg >> 3, r = nwg & 7u, xcd = bid & 7u, local = bid >> 3;
    const unsigned base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + local;
}

// ---- pieces shared by the Detect decode kernel and the head convolution that decodes in its epilogue --------------
// hardware exp2 / reciprocal (v_exp_f32, v_rcp_f32): relative error below 1e-6 for |x| < 16, absolute error below
// 1e-8 beyond; about 5 instructions instead of 25
__device__ __forceinline__ float sigmoid_fast(float x) {
    return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(x * -1.44269504088896340736f));
}

// max over the 64 lanes of a wave, returned in every lane: four DPP steps inside each row of 16 lanes, then the
// four row results are combined as scalars.
__device__ __forceinline__ float wave_max(float v) {
    auto step = [](float x, auto ctrl) {
        const int y = __builtin_amdgcn_update_dpp(__float_as_int(x), __float_as_int(x), decltype(ctrl)::value, 0xf, 0xf, false);
        return fmaxf(x, __int_as_float(y));
    };
    v = step(v, std::integral_constant<int, 0xB1>{});    // quad_perm [1,0,3,2]
    v = step(v, std::integral_constant<int, 0x4E>{});    // quad_perm [2,3,0,1]
    v = step(v, std::integral_constant<int, 0x141>{});   // row_half_mirror
    v = step(v, std::integral_constant<int, 0x140>{});   // row_mirror
    const float r0 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 0));
    const float r1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 16));
    const float r2 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 32));
    const float r3 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 48));
    return fmaxf(fmaxf(r0, r1), fmaxf(r2, r3));
}

// x / d for 0 <= x < 2^22, 0 < d, given rd = 1.0f / d: the float product is off by less than one.
__device__ __forceinline__ int div_small(int x, int d, float rd) {
    int q = (int)((float)x * rd);
    const int r = x - q * d;
    if (r < 0) --q;
    else if (r >= d) ++q;
    return q;
}

static inline hipStream_t as_stream(void* s) { return reinterpret_cast<hipStream_t>(s); }

// ---- per-device bookkeeping (runtime.hip) ---------------------------------------------------------------------------
// One process may drive several GPUs; whatever the library remembers about "the device" is kept per device ordinal.
constexpr int kMaxDevices = 64;
struct DeviceInfo {
    int cus;          // compute units
    int lds_per_cu;   // bytes of LDS a CU offers its resident workgroups
};
// Properties of the calling thread's current device (queried once per device); 256 CUs / 160 KB when the query fails.
DeviceInfo current_device_info();
// A kernel that takes more than 64 KB of dynamic LDS must be opted in (hipFuncAttributeMaxDynamicSharedMemorySize), and
// the opt-in holds for the device that was current when it was made: `done` is the caller's per-kernel flag array
// (one static array of kMaxDevices per kernel instantiation), set once the current device has been opted in.
int raise_dynamic_lds_limit(const void* kernel, int bytes, unsigned char (&done)[kMaxDevices], const char* what);

}  // namespace ydet

// Internal launchers shared by the public entry points and ydet_run_ops.
int ydet_launch_conv(const ydet_conv_desc* d, hipStream_t stream);
int ydet_launch_head_decode(const ydet_conv_desc* d, const ydet_head_desc* hd, hipStream_t stream);
int ydet_launch_focus_conv(const ydet_conv_desc* d, int x_dtype, hipStream_t stream);
int ydet_launch_focus_pair(const ydet_conv_desc* focus, const ydet_conv_desc* next, int x_dtype, hipStream_t stream, bool rgb);
int ydet_launch_rgb_conv(const ydet_conv_desc* d, int x_dtype, hipStream_t stream);
int ydet_launch_bottleneck(const ydet_conv_desc* c1, const ydet_conv_desc* c2, const ydet_conv_desc* c3 /* or null */, hipStream_t stream);
