// Stand-alone timing trace of conv_dma_kernel (debugging aid, not part of libydet.so):
//   hipcc --offload-arch=gfx950 -O3 -Iinclude -Iyolo_amd/csrc tools/conv_trace.hip -o build/conv_trace
//   build/conv_trace N H W Cin Cout k stride
// Prints, for a few workgroups, the cycle stamps relative to the earliest workgroup start.
#define YDET_TRACE 1
#include "../yolo_amd/csrc/conv.hip"

#include <cstdlib>
#include <cstring>
#include <vector>

namespace ydet {
int fail(int code, const char* fmt, ...) {
    va_list ap; va_start(ap, fmt); vfprintf(stderr, fmt, ap); va_end(ap); fprintf(stderr, "\n");
    return code;
}
void clear_error() {}
DeviceInfo current_device_info() { return DeviceInfo{256, 160 * 1024}; }
int raise_dynamic_lds_limit(const void* kernel, int bytes, unsigned char (&)[kMaxDevices], const char* what) {
    if (hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes) != hipSuccess)
        return fail(YDET_ERR_ARG, "%s: dynamic LDS limit", what);
    return YDET_OK;
}
namespace conv {   // the streaming 1x1 and patch kernels live in other translation units of the library; not traced here
bool conv_pw_takes(const ConvArgs&) { return false; }
int launch_conv_pw(const ConvArgs&, hipStream_t) { return YDET_ERR_ARG; }
bool conv_patch_takes(const ConvArgs&) { return false; }
int launch_conv_patch(const ConvArgs&, hipStream_t) { return YDET_ERR_ARG; }
}  // namespace conv
}  // namespace ydet

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

int main(int argc, char** argv) {
    if (argc < 8) { fprintf(stderr, "usage: conv_trace N H W Cin Cout k stride\n"); return 2; }
    const int N = atoi(argv[1]), H = atoi(argv[2]), W = atoi(argv[3]), Cin = atoi(argv[4]), Cout = atoi(argv[5]);
    const int k = atoi(argv[6]), st = atoi(argv[7]);
    const int pad = k / 2, Ho = (H + 2 * pad - k) / st + 1, Wo = (W + 2 * pad - k) / st + 1;
    const int kp = (k * k * Cin + 31) / 32 * 32, cout_pad = (Cout + 127) / 128 * 128;
    void *x, *w, *y; float* b;
    const size_t xb = (size_t)N * H * W * Cin * 2, wb = (size_t)cout_pad * kp * 2, yb = (size_t)N * Ho * Wo * Cout * 2;
    CK(hipMalloc(&x, xb)); CK(hipMalloc(&w, wb)); CK(hipMalloc(&y, yb)); CK(hipMalloc((void**)&b, cout_pad * 4));
    CK(hipMemset(x, 0x11, xb)); CK(hipMemset(w, 0x11, wb)); CK(hipMemset(b, 0, cout_pad * 4));
    ydet_conv_desc d = {};
    d.x = x; d.w = w; d.bias = b; d.y = y;
    d.N = N; d.H = H; d.W = W; d.Cin = Cin; d.Cout = Cout; d.KH = d.KW = k; d.stride = st; d.pad = pad; d.Ho = Ho; d.Wo = Wo;
    d.x_ld = Cin; d.y_ld = Cout; d.kp = kp; d.cout_pad = cout_pad; d.act = YDET_ACT_SILU; d.in_dtype = d.out_dtype = YDET_BF16;
    // "head" as 8th argument: the prediction conv with the Detect decode in its epilogue (na = 3, no = Cout / 3)
    const bool head = argc > 8 && !strcmp(argv[8], "head");
    ydet_head_desc hd = {};
    if (head) {
        d.act = YDET_ACT_NONE; d.out_dtype = YDET_F32; d.cout_pad = 256; d.y_ld = 256; d.cout_store = 256;
        CK(hipFree(w)); CK(hipMalloc(&w, (size_t)256 * kp * 2)); CK(hipMemset(w, 0x11, (size_t)256 * kp * 2)); d.w = w;
        CK(hipFree(y)); CK(hipMalloc(&y, (size_t)N * Ho * Wo * 256 * 4)); d.y = y;
        CK(hipFree(b)); CK(hipMalloc((void**)&b, 256 * 4)); CK(hipMemset(b, 0, 256 * 4)); d.bias = b;
        hd.na = 3; hd.no = Cout / 3; hd.ny = Ho; hd.nx = Wo; hd.stride = 8.f; hd.rows_per_image = 3LL * Ho * Wo;
        for (int i = 0; i < 6; ++i) hd.anchor_wh[i] = 10.f + i;
        CK(hipMalloc((void**)&hd.z, (size_t)N * hd.rows_per_image * hd.no * 4));
        CK(hipMalloc((void**)&hd.row_max, (size_t)N * hd.rows_per_image * 4));
    }
    auto launch = [&]() { return head ? ydet_launch_head_decode(&d, &hd, 0) : ydet_launch_conv(&d, 0); };
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i) if (launch()) return 1;
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0, 0));
    const int reps = 20;
    for (int i = 0; i < reps; ++i) launch();
    CK(hipEventRecord(e1, 0)); CK(hipDeviceSynchronize());
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    printf("avg %.2f us per launch (%d back to back)\n", ms * 1e3 / reps, reps);
    std::vector<unsigned long long> t(64 * 32);
    CK(hipMemcpyFromSymbol(t.data(), HIP_SYMBOL(g_trace), t.size() * 8));
    unsigned long long t0 = ~0ull;
    for (int blk = 0; blk < 64; ++blk) if (t[blk * 32] && t[blk * 32] < t0) t0 = t[blk * 32];
    for (int blk : {0, 1, 8, 63}) {
        printf("wg %2d:", blk);
        for (int s = 0; s < 32; ++s) if (t[blk * 32 + s]) printf(" [%d]%llu", s, t[blk * 32 + s] - t0);
        printf("\n");
    }
    return 0;
}
