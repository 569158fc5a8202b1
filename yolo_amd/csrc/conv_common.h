// Pieces shared by the convolution kernels of libydet.so (conv.hip, conv_pw.hip): the launch arguments, the LDS
// image helpers, the MFMA wrappers and the epilogue arithmetic.  gfx950 only.
#pragma once
#include <math.h>
#include <stdio.h>

#include <type_traits>

#include "ydet_internal.h"

namespace ydet {
namespace conv {

struct ConvArgs {
    const char* x;
    const char* w;
    const float* bias;
    const char* res;
    char* y;
    int N, H, W, Cin, Cout, KH, KW, stride, pad, Ho, Wo;
    int x_ld, x_coff, y_ld, y_coff, res_ld, res_coff;
    int kp, act, res_mode;
    float slope;
    long long M;      // N*Ho*Wo
    int tiles_n;      // channel tiles
    int vec_store;    // 4-channel vector stores are legal
    int cout_store;   // channels written per pixel (>= Cout; the extra ones are zero-weight rows)
    int lds_epilogue; // 16-byte coalesced stores through LDS are legal
    const float* post_scale;  // second per-channel affine + activation applied to the finished value, or null
    const float* post_shift;
    int post_act;
    float post_slope;
    char* y2;         // second destination: output channels y_split.. go to y2 (slice y2_ld / y2_coff), or null
    int y2_ld, y2_coff;
    int y_split;      // INT_MAX without a second destination
    const char* pw_w; // a 1x1 convolution run on the finished tile in registers (64-channel tile only): its weights
    const float* pw_b1;   // [cout_pad][pw_kp]; pw_b1 = bias of the FIRST convolution (a.bias is then the second's), or null
    int pw_kp;
    int groups, dilation, cin_group;   // direct kernel only (groups > 1 or dilation > 1)
    const char* x_up; // pointwise convs only: input channels [0, up_C) are read from this map of half the size at
    int up_ld, up_coff, up_C;   // (h/2, w/2) -- a nearest-neighbour 2x upsample that is never materialised; 0: none
    // K split over workgroups (launches of a few tiles and a long K): workgroup id = split * tiles_per_split + tile;
    // split s walks filter taps [s * split_taps, ...) (split_taps > 0) or input channels [s * split_cin, ...) of a 1x1
    // convolution, and writes its fp32 partial sums at y + s * split_stride.  splits <= 1: the whole K, as ever.
    int splits, tiles_per_split, split_taps, split_cin;
    long long split_stride;
    // bytes from the first byte of the input slice (x + x_coff) to the end of the tensor, and of the packed weights: the
    // extents of the buffer descriptors the pipelined kernel addresses them through (32-bit offsets; an offset beyond the
    // extent reads zeros, which is how padding taps are fetched)
    long long x_span, w_span;
};

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ int chunk_swz(int g) { return (0x78 >> (2 * (g & 3))) & 3; }

template <typename T>
struct Frag;
template <>
struct Frag<__bf16> {
    using type = bf16x8;
};
template <>
struct Frag<float> {
    using type = f32x4;
};

__device__ __forceinline__ f32x4 mma(const bf16x8& a, const bf16x8& b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}
// 16 fp32 k-values per step: lane group g = lane>>4 holds k = 4g..4g+3 of its row in both operands,
// MFMA j consumes element j of every lane, so the four MFMAs together cover all 16 k exactly once.
__device__ __forceinline__ f32x4 mma(const f32x4& a, const f32x4& b, f32x4 c) {
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(a[0], b[0], c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(a[1], b[1], c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(a[2], b[2], c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(a[3], b[3], c, 0, 0, 0);
    return c;
}

__device__ __forceinline__ float activate(float v, int act, float slope) {
    if (act == YDET_ACT_LEAKY) return v > 0.f ? v : v * slope;
    if (act == YDET_ACT_SILU) return v * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(v * -1.44269504088896340736f));
    return v;
}

// Four consecutive output channels of one pixel: bias, residual (either order), activation, and the optional second
// affine + activation (a BatchNorm + activation that follows a concatenation).  MODE >= 0 fixes the variant at compile
// time (bit 0: SiLU, else leaky with a.slope -- slope 1 is "no activation"; bit 1: second stage present; bit 2: the
// residual goes in before the activation; bit 3: there is no residual, r is not even looked at), so the 64 values a
// lane finishes run through straight-line code; MODE < 0
// reads the variant from the arguments (the rarely used store paths, to keep the number of instantiations down).
// b4 / ps4 / pb4 point at the four channels' bias, post-scale and post-shift (global memory or the tile's LDS copy).
template <int MODE>
__device__ __forceinline__ void finish4(float (&v)[4], const float (&r)[4], const float* b4, const float* ps4,
                                        const float* pb4, const ConvArgs& a) {
    if (b4) {
        const float4 b = *reinterpret_cast<const float4*>(b4);
        v[0] += b.x; v[1] += b.y; v[2] += b.z; v[3] += b.w;
    }
    const bool silu = MODE >= 0 ? (MODE & 1) != 0 : a.act == YDET_ACT_SILU;
    const bool post = MODE >= 0 ? (MODE & 2) != 0 : a.post_scale != nullptr;
    const bool before = MODE >= 0 ? (MODE & 4) != 0 : a.res_mode == YDET_RES_BEFORE_ACT;
    constexpr bool nores = MODE >= 0 && (MODE & 8) != 0;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        float x = before ? v[e] + r[e] : v[e];     // run-time form: "after" and "none" share one path (r is zero)
        if (silu) {
            // x * sigmoid(x) on the hardware exp2 / reciprocal units (relative error ~1e-6, far inside the bf16 / 1e-3
            // bars): 5 instructions instead of ~25 with an IEEE division
            x = x * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(x * -1.44269504088896340736f));
        } else {
            x = x > 0.f ? x : x * a.slope;
        }
        v[e] = (before || nores) ? x : x + r[e];
    }
    if (post) {
        const float4 ps = *reinterpret_cast<const float4*>(ps4);
        const float4 pb = *reinterpret_cast<const float4*>(pb4);
        v[0] = activate(v[0] * ps.x + pb.x, a.post_act, a.post_slope);
        v[1] = activate(v[1] * ps.y + pb.y, a.post_act, a.post_slope);
        v[2] = activate(v[2] * ps.z + pb.z, a.post_act, a.post_slope);
        v[3] = activate(v[3] * ps.w + pb.w, a.post_act, a.post_slope);
    }
}

// Calls f(std::integral_constant<int, MODE>) for the launch's epilogue variant: the two common ones (leaky / SiLU,
// residual absent or added last, no second stage) are compiled in, everything else takes the run-time form.
template <typename F>
__device__ __forceinline__ void with_epilogue_mode(const ConvArgs& a, F f) {
    const bool plain = a.post_scale == nullptr && a.res_mode != YDET_RES_BEFORE_ACT;
    if (plain && a.act == YDET_ACT_SILU) {
        if (a.res) f(std::integral_constant<int, 1>{});
        else f(std::integral_constant<int, 9>{});      // most layers: no residual unpacking or masking at all
    } else if (plain) {
        if (a.res) f(std::integral_constant<int, 0>{});
        else f(std::integral_constant<int, 8>{});
    } else {
        f(std::integral_constant<int, -1>{});
    }
}

template <typename T>
__device__ __forceinline__ void load4(const char* p, float (&v)[4]);
template <>
__device__ __forceinline__ void load4<float>(const char* p, float (&v)[4]) {
    const float4 t = *reinterpret_cast<const float4*>(p);
    v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
}
template <>
__device__ __forceinline__ void load4<__bf16>(const char* p, float (&v)[4]) {
    const bf16x4 t = *reinterpret_cast<const bf16x4*>(p);
    v[0] = (float)t[0]; v[1] = (float)t[1]; v[2] = (float)t[2]; v[3] = (float)t[3];
}
template <typename T>
__device__ __forceinline__ void store4(char* p, const float (&v)[4]);
template <>
__device__ __forceinline__ void store4<float>(char* p, const float (&v)[4]) {
    *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
}
template <>
__device__ __forceinline__ void store4<__bf16>(char* p, const float (&v)[4]) {
    bf16x4 t;
    t[0] = (__bf16)v[0]; t[1] = (__bf16)v[1]; t[2] = (__bf16)v[2]; t[3] = (__bf16)v[3];
    *reinterpret_cast<bf16x4*>(p) = t;
}

// 16 zero bytes every padded tap is loaded from (LDS-DMA cannot predicate a lane to zero).
static __device__ __attribute__((aligned(16))) unsigned int g_zero_chunk[4] = {0u, 0u, 0u, 0u};

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

template <int N>
__device__ __forceinline__ void wait_lgkmcnt() {
    asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(N) : "memory");
}

// Dry run.  While a thread asks which kernel a descriptor would get (ydet_conv_kernel_name), the launchers write the
// kernel's label here -- the labels of yolo_amd/engine.py:conv_kernel_label and tools/rocprof_summary.py -- and launch
// nothing (no HIP call is made: the question can be asked on a machine without a GPU).
struct KernelProbe {
    bool on;
    char name[96];
};
KernelProbe& kernel_probe();   // (thread-local; defined in conv.hip)
template <typename TIN, typename TOUT>
inline const char* dtype_pair_name() {
    return sizeof(TIN) == 2 ? (sizeof(TOUT) == 2 ? "bf16->bf16" : "bf16->f32") : "f32->f32";
}

// The streaming 1x1 kernel of conv_pw.hip (bf16 -> bf16, 128 or 256 input channels, many pixels; bit-identical to the
// tiled kernel): whether it takes the launch, and the launch itself.
bool conv_pw_takes(const ConvArgs& a);
int launch_conv_pw(const ConvArgs& a, hipStream_t stream);

// The patch-in-LDS 3x3 kernel of conv_patch.hip (bf16 -> bf16, stride 1, pad 1, 64 or 128 input channels, at least 96
// output channels; bit-identical to the tiled kernel): whether it takes the launch, and the launch itself.
bool conv_patch_takes(const ConvArgs& a);
int launch_conv_patch(const ConvArgs& a, hipStream_t stream);

}  // namespace conv
}  // namespace ydet
