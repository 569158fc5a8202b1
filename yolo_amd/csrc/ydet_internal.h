// Shared helpers of libydet.so (host + device).  gfx950 only.
#pragma once
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdint.h>
#include <stdio.h>

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

static inline hipStream_t as_stream(void* s) { return reinterpret_cast<hipStream_t>(s); }

}  // namespace ydet

// Internal launchers shared by the public entry points and ydet_run_ops.
int ydet_launch_conv(const ydet_conv_desc* d, hipStream_t stream);
int ydet_launch_focus_conv(const ydet_conv_desc* d, int x_dtype, hipStream_t stream);
