/*
 * ydet.h -- C ABI of libydet.so: the MI355X (gfx950) detection-inference hot path.
 *
 * The reference project (he13689/yolo) is pure Python and has no FFI of its own; the
 * operators below are the ones its hot path invokes through torch / numpy / torchvision.
 * Each entry point names the reference code it stands in for (paths relative to the
 * reference root).  A binding (ctypes / cffi / pybind) only needs this header.
 *
 * Conventions
 *   - every pointer is a raw DEVICE pointer unless the comment says "host";
 *   - activations are NHWC: element (n,h,w,c) of a tensor with leading dimension `ld`
 *     and channel offset `coff` lives at base[((n*H + h)*W + w)*ld + coff + c]
 *     (ld/coff let several producers write disjoint channel slices of one buffer, which is
 *     how torch.cat(dim=1) disappears);
 *   - every function is asynchronous on `stream` (a hipStream_t passed as void*), returns 0
 *     on success and a negative ydet_status otherwise; ydet_last_error() gives the text;
 *   - the library owns no device memory: buffers and workspaces belong to the caller;
 *   - nothing throws, nothing calls exit().
 */
#ifndef YDET_H
#define YDET_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum {
    YDET_OK = 0,
    YDET_ERR_ARG = -1,      /* bad shape / alignment / enum */
    YDET_ERR_LAUNCH = -2,   /* hipLaunchKernel or another HIP call failed */
    YDET_ERR_WORKSPACE = -3 /* workspace too small */
} ydet_status;

typedef enum { YDET_F32 = 0, YDET_BF16 = 1 } ydet_dtype;
typedef enum { YDET_ACT_NONE = 0, YDET_ACT_LEAKY = 1, YDET_ACT_SILU = 2 } ydet_act;
typedef enum { YDET_RES_NONE = 0, YDET_RES_AFTER_ACT = 1, YDET_RES_BEFORE_ACT = 2 } ydet_res_mode;

/* Text of the last error raised on the calling thread ("" if none). */
const char* ydet_last_error(void);
/* ABI version of the library (bumped when a struct below changes); equals YDET_ABI_VERSION of the header it
 * was built from. */
#define YDET_ABI_VERSION 16
int ydet_abi_version(void);

/* ------------------------------------------------------------------------------------------
 * Convolution + folded BatchNorm + activation (+ residual).
 * Replaces: utils/moduleUtils.py:8-24 (Conv = Conv2d(bias) + BN + LeakyReLU(0.1)),
 *           yolov5_official/modules.py:24-41 (Conv = Conv2d + BN + SiLU), models/common.py:22-37
 *           (same with LeakyReLU(0.01)), the bias-only 1x1 prediction convs
 *           (models/v3tiny.py:54,59; yolov5_official/model.py:42,58), and the residual adds of
 *           utils/moduleUtils.py:58,88-91 and yolov5_official/modules.py:218.
 *
 * y = act(conv(x, w) + bias) [+ res]            res_mode = AFTER_ACT   (x + act(bn(conv)))
 * y = act(conv(x, w) + bias + res)              res_mode = BEFORE_ACT  (act(x + bn(conv)))
 * then, if post_scale is given,  y = post_act(post_scale * y + post_shift).
 *
 * Weights are pre-packed by the caller as w[cout_pad][kp], row-major, in `in_dtype`, with
 * k = (kh*KW + kw)*Cin + c, rows >= Cout and columns >= KH*KW*Cin zero.  BatchNorm is folded into
 * w and bias by the caller.  `bias` has cout_pad fp32 entries (or is NULL).
 * Requirements: Cin, x_ld, x_coff multiples of 8 (bf16) or 4 (f32); kp a multiple of 32 (bf16) or
 * 16 (f32); cout_pad a multiple of 128 and >= Cout.  The residual has the output's shape.
 * ------------------------------------------------------------------------------------------ */
typedef struct {
    const void* x;
    const void* w;
    const float* bias;
    const void* res;
    void* y;
    int32_t N, H, W, Cin;
    int32_t Cout, KH, KW, stride, pad;
    int32_t Ho, Wo;
    int32_t x_ld, x_coff;
    int32_t y_ld, y_coff;
    int32_t res_ld, res_coff;
    int32_t kp, cout_pad;
    int32_t act;      /* ydet_act */
    float slope;      /* LeakyReLU negative slope */
    int32_t res_mode; /* ydet_res_mode */
    int32_t in_dtype; /* dtype of x, w and res */
    int32_t out_dtype;/* dtype of y */
    int32_t cout_store; /* 0, or the number of channels written per pixel, Cout <= cout_store <= cout_pad:
                           the extra channels are the zero rows of w (value act(0) = 0); lets a head with
                           255 channels be stored as 256 with whole 16-byte pieces */
    /* Optional second stage, y = post_act(post_scale[c] * y + post_shift[c]) applied to the finished value:
     * the BatchNorm + LeakyReLU that follows a torch.cat whose producer already ended in BN + activation
     * (utils/moduleUtils.py:209-214 `tie`, models/v5v2.py:86-97 `last`).  Both arrays have cout_pad fp32
     * entries; both NULL = no second stage. */
    const float* post_scale;
    const float* post_shift;
    int32_t post_act;  /* ydet_act */
    float post_slope;
    /* Optional second destination: output channels [y_split, Cout) are written to the slice (y2, y2_ld, y2_coff)
     * instead of y, which then holds channels [0, y_split) only.  Lets one convolution stand for two that read the
     * same input but feed different buffers (C3: cv1 -> the bottleneck chain, cv2 -> the concat,
     * yolov5_official/modules.py:252).  y_split and Cout multiples of 8 (bf16) / 4 (f32); needs Cin a multiple of
     * the K step and Cout > 32; excludes res and cout_store.  NULL = one destination. */
    void* y2;
    int32_t y2_ld, y2_coff, y_split;
    /* Optional second source of a 1x1 / stride 1 convolution: input channels [0, x_up_C) are read from the slice
     * (x_up, x_up_ld, x_up_coff) of a map of HALF the size, pixel (h, w) taking (h/2, w/2); channels [x_up_C, Cin) come
     * from x as usual (x keeps its layout: its first x_up_C channels are simply never read).  This is
     * nn.Upsample(scale_factor=2, mode="nearest") followed by torch.cat and a 1x1 convolution (yolov5s.yaml head rows
     * 11-13 and 15-17; yolov5_official/modules.py:303-310 Concat) without the upsampled map ever being written.
     * H, W even; x_up_C and Cin multiples of the K step (32 bf16 / 16 f32); Cout > 32.  NULL = one source. */
    const void* x_up;
    int32_t x_up_ld, x_up_coff, x_up_C;
    /* Optional 1x1 follower: a second convolution (1x1 / stride 1, pw_cout output channels, weights pw_w packed like w
     * for 64 input channels, pw_kp = 64, bias pw_bias) applied to the finished values of this one while they are still
     * in registers:  y = act(pw(act(conv(x) + bias)) + pw_bias), one activation for both.  The output slices (y, y2,
     * y_split) then hold the follower's channels; the 64-channel map in between is never written.  This is
     * `Conv(32, 64, 3, 2)` followed by the C3's cv1 | cv2 pair (yolov5s.yaml backbone rows 1-2,
     * yolov5_official/modules.py:252).  Needs bf16, Cout == 64, 32 < pw_cout <= 64, Cin a multiple of 32; excludes res,
     * post_*, cout_store and x_up.  NULL = no follower. */
    const void* pw_w;
    const float* pw_bias;
    int32_t pw_cout, pw_kp;
    /* Grouped and dilated convolutions (nn.Conv2d(groups=g, dilation=d): utils/moduleUtils.py:9-13 `g`, `d`;
     * yolov5_official/modules.py:19-21 DWConv, :208-218 Bottleneck(g=)).  groups > 1: output channel co belongs to
     * group co / (Cout / groups) and reads the cin_group = (real input channels) / groups input channels from
     * group * cin_group on; weights are packed [cout_pad][kp] with k = (kh*KW + kw)*cin_group + c.  dilation > 1:
     * tap (kh, kw) reads input pixel (ho*stride - pad + kh*dilation, ...), Ho = (H + 2 pad - dilation (KH-1) - 1) /
     * stride + 1.  Either one sends the launch to a direct (non-MFMA) kernel; y2, x_up, pw_w and cout_store are then
     * not available.  0 means 1 (cin_group: Cin / groups). */
    int32_t groups, dilation, cin_group;
    /* Optional K split, for launches of a few tiles and a long K (single images, the deepest layers: one workgroup per
     * tile walks all of K alone while most of the chip idles): with a workspace of at least
     * ydet_conv_splitk_workspace_bytes(d) bytes (256-byte aligned) the filter taps of a k x k convolution, or channel
     * ranges of a 1x1 one, are spread over several workgroups whose fp32 partial sums a second kernel adds up before
     * bias / activation / residual.  The result differs from the unsplit launch by the order of the fp32 additions
     * only -- and therefore, unlike everything else here, may differ between batch sizes.  NULL = never split.  bf16
     * inputs, Cin a multiple of 64; excludes pw_w and x_up. */
    void* splitk_ws;
    int64_t splitk_ws_bytes;
} ydet_conv_desc;
int ydet_conv2d_nhwc(const ydet_conv_desc* d, void* stream);
/* Bytes of splitk_ws with which ydet_conv2d_nhwc would split this convolution's K; 0: it would not. */
size_t ydet_conv_splitk_workspace_bytes(const ydet_conv_desc* d);
/* Which kernel ydet_conv2d_nhwc would launch for this descriptor, as the label the per-kernel timings and the profile
 * summaries are keyed by ("conv_dma<256x192,bf16->bf16>", "conv_patch<c128>", "conv_pw<k256>", ...), written to out
 * (n bytes).  Host arithmetic only: nothing is launched and no HIP call is made, so it also answers on a machine without
 * a GPU (for 256 CUs).  No counterpart in the reference: a diagnostic of this library's dispatch. */
int ydet_conv_kernel_name(const ydet_conv_desc* d, char* out, int n);

/* Focus + the first convolution, straight from the NCHW image: yolov5_official/modules.py:285-293 (the slicing
 * torch.cat([x[..., ::2, ::2], x[..., 1::2, ::2], x[..., ::2, 1::2], x[..., 1::2, 1::2]], 1) followed by Conv) when it is
 * layer 0 of the network (yolov5_official/yamls/yolov5s.yaml backbone[0]); also the NCHW -> NHWC edge conversion.
 * d->x is the image [N, 3, H, W] (x_dtype, contiguous, H and W even), d->H / d->W its size, d->Cin = 3; the
 * convolution is 3x3 / stride 1 / pad 1 over the 12 focus channels, with w / bias packed as for
 * ydet_conv2d_nhwc for 16 input channels (focus channel q*3 + c, then 4 zeros; kp = 160); d->y is the NHWC bf16
 * slice [N, H/2, W/2, Cout], Cout <= 64.  bf16 weights / output only; x_ld, x_coff, res and post_* are unused.
 * x_dtype YDET_U8: the image holds raw 0..255 pixel values and every value enters as bf16(float(v) / 255), the
 * `img.float() / 255` of yolov5_official/val.py:68-69 folded into the load (the fp32 image is never written). */
int ydet_focus_conv3x3_nchw(const ydet_conv_desc* d, int x_dtype, void* stream);

/* Focus + the first convolution + the SECOND convolution (+ its 1x1 follower) as one kernel, straight from the NCHW
 * image: rows 0-1 of yolov5_official/yamls/yolov5s.yaml's backbone, `Focus(3, 32, 3)` and `Conv(32, 64, 3, 2)`
 * (yolov5_official/modules.py:285-293, :34-46), and through next->pw_w the cv1 | cv2 pair of the C3 in row 2
 * (modules.py:242-253).  The 32-channel map between the two layers -- the largest tensor of the network -- exists only
 * as patches in LDS.  `focus` as for ydet_focus_conv3x3_nchw with Cout = 32 (its y is not used); `next` as for
 * ydet_conv2d_nhwc: 3x3 / stride 2 / pad 1, Cin = 32, Cout = 64, H x W the half-size map, kp = 288 (its x is not used),
 * optionally with a follower (pw_w, pw_bias, pw_cout, pw_kp = 64) and a second destination (y2, y_split); no res, post_*,
 * x_up or cout_store.  bf16 weights and output.  Same bf16 intermediate values and accumulation order as the layers run
 * one by one. */
int ydet_focus_conv_pair_nchw(const ydet_conv_desc* focus, const ydet_conv_desc* next, int x_dtype, void* stream);

/* The first TWO layers of the Darknet-53 / CSPDarknet-53 trunks as one kernel, straight from the NCHW image:
 * `Conv(3, 32, 3, p=1)` and `Conv(32, 64, 3, p=1, s=2)` (models/basemodel/d19.py:83-123 layer_1, :215-265 layer1; the
 * Conv of utils/moduleUtils.py:8-24).  The 32-channel map at full image size between them exists only as patches in
 * LDS.  `first`: x = the image [N, 3, H, W] (x_dtype fp32 or bf16), Cin = 3, 3x3 / stride 1 / pad 1, Cout = 32, w / bias
 * packed as for ydet_conv2d_nhwc for 16 input channels (the 3 image channels, then zeros; kp = 160), its y is not used;
 * `next` as for ydet_focus_conv_pair_nchw with H x W the image size (follower and second destination included); both
 * layers share one activation.  Same bf16 intermediate values and accumulation order as ydet_rgb_conv3x3_nchw followed
 * by ydet_conv2d_nhwc. */
int ydet_rgb_conv_pair_nchw(const ydet_conv_desc* first, const ydet_conv_desc* next, int x_dtype, void* stream);

/* The first layer of the Darknet-style trunks straight from the NCHW image: `Conv(3, C, k=3, p=1)` (utils/moduleUtils.py:8-24
 * as used at models/basemodel/d19.py:22,88,133,175,221) including the NCHW -> NHWC edge conversion.  d->x is the image
 * [N, 3, H, W] (x_dtype fp32 or bf16, contiguous), d->Cin = 3; 3x3 / stride 1 / pad 1; w / bias packed as for
 * ydet_conv2d_nhwc for 4 input channels (the 3 image channels + a zero; kp = 64); d->y is the NHWC bf16 slice
 * [N, H, W, Cout], Cout a multiple of 8, <= 64.  bf16 weights / output only; x_ld, x_coff, res, post_* unused.
 * Pooled form: with H and W even and Ho x Wo = H/2 x W/2 the MaxPool2d(2, 2) that follows the layer in the Darknet-19
 * and tiny trunks (models/basemodel/d19.py:29 conv1, :175-182 conv_1 / maxpool_1) is taken in the epilogue and d->y is
 * the pooled slice [N, H/2, W/2, Cout] (the result of the two launches, bit for bit). */
int ydet_rgb_conv3x3_nchw(const ydet_conv_desc* d, int x_dtype, void* stream);

/* A residual bottleneck with C = 32 or 64 channels throughout as ONE kernel: yolov5_official/modules.py:208-218,
 * `x + self.cv2(self.cv1(x))` with cv1 = Conv(C, C, 1, 1) and cv2 = Conv(C, C, 3, 1) (the bottlenecks of the first two
 * C3 stages of yolov5s.yaml).  cv1 describes the 1x1 stage (x, w, bias, shape, activation; its y is not used: the
 * intermediate map never leaves the chip), cv2 the 3x3 / stride 1 / pad 1 stage (w, bias, y; its x is not used).
 * cv2->res is NULL (no shortcut) or equal to cv1's input slice with res_mode = AFTER_ACT.  Both stages share one
 * activation; bf16 only.  The output slice must not overlap the input slice (a tile's halo is read while other
 * workgroups write their outputs).  Same bf16 intermediate and accumulation order as the two stages run through
 * ydet_conv2d_nhwc. */
int ydet_bottleneck_fused(const ydet_conv_desc* cv1, const ydet_conv_desc* cv2, void* stream);

/* The tail of a C3 block as ONE kernel: its last 32-channel bottleneck (cv1, cv2 as for ydet_bottleneck_fused; cv2->y is
 * not used) and `self.cv3(torch.cat((self.m(self.cv1(x)), self.cv2(x)), dim=1))` (yolov5_official/modules.py:252-253)
 * with cv3 a 1x1 convolution 64 -> 64 sharing the activation.  cv3->x is the slice holding the OTHER half of the concat
 * (the 32 channels of C3.cv2(x)); its weights are packed for 64 input channels, bottleneck output first; cv3->y is the
 * block's output slice.  Neither the bottleneck's output nor the concat is written.  bf16 only. */
int ydet_c3_tail_fused(const ydet_conv_desc* cv1, const ydet_conv_desc* cv2, const ydet_conv_desc* cv3, void* stream);

/* ------------------------------------------------------------------------------------------
 * Pooling / resampling / layout operators on NHWC slices.
 * ------------------------------------------------------------------------------------------ */
typedef struct {
    const void* x;
    void* y;
    int32_t N, H, W, C;     /* input spatial size and channels moved */
    int32_t x_ld, x_coff;
    int32_t y_ld, y_coff;
    int32_t dtype;
} ydet_slice_desc;

/* MaxPool2d(2, 2): models/basemodel/d19.py:22,27,34,43,53,175-187.  Output H/2 x W/2. */
int ydet_maxpool2x2_nhwc(const ydet_slice_desc* d, void* stream);
/* ZeroPad2d((0,1,0,1)) + MaxPool2d(2, stride 1): models/basemodel/d19.py:190-193.  The padded
 * zeros take part in the max.  Output H x W. */
int ydet_maxpool2x2s1_zeropad_nhwc(const ydet_slice_desc* d, void* stream);
/* SPP: MaxPool2d(k, 1, k//2) for k = 5, 9, 13 (-inf padding): utils/moduleUtils.py:63-75,
 * yolov5_official/modules.py:271-282.  Writes the three pooled maps at channel offsets
 * y_coff, y_coff + C, y_coff + 2C of y (the un-pooled input is expected to sit in front of
 * them already, so the concatenation costs nothing). */
int ydet_spp_nhwc(const ydet_slice_desc* d, void* stream);
/* nn.Upsample(scale_factor=2): mode 0 nearest (models/v4.py:128,133; yolov5s.yaml head),
 * mode 1 bilinear with align_corners=True (models/v3tiny.py:55, models/v3spp.py:80-82). */
int ydet_upsample2x_nhwc(const ydet_slice_desc* d, int mode, void* stream);
/* Space to depth on NHWC, H x W x C -> H/2 x W/2 x 4C.
 *   mode 0 (reorg_layer, utils/moduleUtils.py:28-41): out channel (sy*2+sx)*C + c
 *   mode 1 (Focus slicing, yolov5_official/modules.py:293): out channel q*C + c with
 *          q <-> (dy,dx) = (0,0),(1,0),(0,1),(1,1) */
int ydet_space_to_depth_nhwc(const ydet_slice_desc* d, int mode, void* stream);
/* Channel-slice copy (the torch.cat fallback when a producer could not write in place). */
int ydet_copy_slice_nhwc(const ydet_slice_desc* d, void* stream);

/* API-edge layout conversion.  x is NCHW [N,C,H,W] of x_dtype (contiguous); y is NHWC with y_ld
 * channels per pixel in y_dtype; channels C..c_pad-1 of the slice are written as zero.
 * focus != 0 applies the Focus slicing on the fly (y is then H/2 x W/2 x 4C). */
int ydet_nchw_to_nhwc(const void* x, void* y, int N, int C, int H, int W, int y_ld, int y_coff,
                      int c_pad, int x_dtype, int y_dtype, int focus, void* stream);
/* NHWC slice -> contiguous NCHW (family-A raw head maps, models/v5.py:65-87). */
int ydet_nhwc_to_nchw(const void* x, void* y, int N, int C, int H, int W, int x_ld, int x_coff,
                      int x_dtype, int y_dtype, void* stream);

/* ------------------------------------------------------------------------------------------
 * Box decode.
 * ------------------------------------------------------------------------------------------ */
typedef struct {
    const void* head; /* NHWC head map [N, ny, nx, ld], channel a*no + o */
    int32_t ny, nx, ld;
    float stride;
    float anchor_wh[2 * 8]; /* pixel anchors (w,h) per anchor, na <= 8 */
} ydet_v5_level;

/* Detect.forward in eval mode: yolov5_official/model.py:56-81 (grid_transposed = 0: cell (row, col)
 * holds the offset (col, row)) and models/yolo.py:42-71 (grid_transposed = 1: its np.meshgrid table makes
 * flat cell f = row*nx + col hold (f / ny, f % ny), which is (row, col) on square maps).
 *   s = sigmoid(raw); xy = (2 s - 0.5 + grid) * stride; wh = (2 s)^2 * anchor; rest = s
 * out is [N, sum_l na*ny*nx, no] fp32, row = offset(level) + (a*ny + row)*nx + col.  no <= 1024.
 * row_max (optional, [N, sum_l na*ny*nx] fp32): for every output row the maximum over classes of
 * out[.., 5+c] * out[.., 4] (the exact fp32 products); ydet_nms_v5 accepts it in place of its own scan. */
int ydet_decode_v5(const ydet_v5_level* levels /* host */, int nl, int N, int na, int no,
                   int grid_transposed, int head_dtype, float* out, float* row_max, void* stream);

/* One pyramid level of Detect as ONE kernel: the bias-only 1x1 prediction convolution (yolov5_official/model.py:42,58
 * `self.m[i]`) with the eval-mode decode of lines 56-81 in its epilogue.  `d` describes the convolution
 * (1x1, no activation, Cout = na*no <= 256, weights padded to 256 rows); its output d->y is the raw map
 * the reference returns as x[i] (NHWC, channel a*no + o): fp32, or -- with bf16 inputs -- bf16 (out_dtype), the
 * counterpart of the half-precision x[i] a `model.half()` reference returns; z and row_max are computed from the fp32
 * accumulators either way.  The decoded rows go to z [N, rows_per_image, no] (fp32) at rows
 * level_row_offset + (a*ny + row)*nx + col, and max_c(z[5+c] * z[4]) per row to row_max (may be NULL) -- exactly what
 * ydet_decode_v5 would produce from d->y.  no <= 128. */
typedef struct {
    float* z;
    float* row_max;
    int64_t rows_per_image;
    int32_t level_row_offset;
    int32_t na, no, ny, nx, grid_transposed;
    float stride;
    float anchor_wh[2 * 8];
} ydet_head_desc;
int ydet_head_conv_decode(const ydet_conv_desc* d, const ydet_head_desc* head, void* stream);

typedef struct {
    const void* head; /* NHWC head map [N, H, W, ld], channels [A conf | A*nc cls | A*4 box] */
    int32_t H, W, ld;
    float anchor_wh[2 * 8]; /* per-anchor (w,h) as the reference stores them for this scale */
} ydet_fa_level;

/* Head split + decode of the v2/v3/v4 detectors: models/v3tiny.py:88-111,137-147 (and the
 * identical tails of v2d19/v3spp/v4) + utils/utils.py:46-98.
 *   conf = sigmoid(ch a); cls = softmax over ch A + a*nc + c; score = cls * conf
 *   mode 0 ("v2", xywh_to_xxyy):   c = (sigmoid(t_xy) + g) * mult;  wh = exp(t_wh) * anchor * mult
 *   mode 1 ("v3", xywh_to_xxyyv3): c = (sigmoid(t_xy) + g) * mult;  wh = exp(t_wh) * anchor
 *   box = clamp((c -/+ wh/2) / input_size, 0, 1); g = (k / H, k % H) for flat cell k = row*W + col -- the
 *   reference's grid (utils/utils.py:167-169), i.e. (row, col) on its square maps -- or (col, row) when
 *   grid_transposed = 0.
 * boxes [N, n, 4], scores [N, n, nc], n = sum_l H*W*A, box index = offset(level) + (row*W+col)*A + a. */
int ydet_decode_v2v3(const ydet_fa_level* levels /* host */, int nl, int N, int A, int nc, int mode,
                     float mult, float input_size, int grid_transposed, int head_dtype,
                     float* boxes, float* scores, void* stream);

/* ------------------------------------------------------------------------------------------
 * Non-maximum suppression.
 * ------------------------------------------------------------------------------------------ */
/* non_max_suppression: yolov5_official/utils.py:113-195 (with torchvision.ops.nms, line 182,
 * restated as: greedy over descending score, suppress when inter/(a_i + a_j - inter) > iou_thres,
 * w,h = max(0, .), ties broken by lower candidate index).
 * pred [N, nboxes, no] fp32 (xywh, obj, nc class scores).  Per image: candidates are boxes with
 * obj > conf and (multi_label) every class with obj*cls > conf, else the best class; the top
 * max_nms by score enter the NMS with boxes offset by cls*max_wh (0 if agnostic), all in fp32;
 * at most max_det survive.  class_mask (device, nc bytes, may be NULL) keeps only classes whose
 * byte is non-zero (the `classes=` filter, line 165-166).
 * row_max (device, [N, nboxes] fp32, may be NULL): per box max_c(pred[5+c] * pred[4]) as written by
 * ydet_decode_v5 for this very `pred`; it replaces the full scan of pred (the candidates themselves are still
 * read from pred).  Only valid when every class score is <= 1, as sigmoid outputs are.  Ignored with a class_mask.
 * out_det [N, max_det, 6] = (x1,y1,x2,y2,score,cls) in descending score order, rows >= count zero;
 * out_count [N] int32.  workspace: ydet_nms_v5_workspace_bytes(N, nboxes, no) bytes.
 * Limits: max_nms <= 4096, max_det <= max_nms. */
size_t ydet_nms_v5_workspace_bytes(int N, int nboxes, int no);
int ydet_nms_v5(const float* pred, int N, int nboxes, int no, float conf_thres, float iou_thres,
                int multi_label, int agnostic, int max_nms, int max_det, float max_wh,
                const uint8_t* class_mask, const float* row_max, float* out_det, int32_t* out_count,
                void* workspace, size_t workspace_bytes, void* stream);

/* ydet_nms_v5 for callers that cannot allocate device memory (code under HIP graph capture, tight serving loops,
 * harnesses that forbid hipMalloc): no counterpart in the reference.  Same arguments, same kernels and bit-identical
 * results, but the workspace is a YDET_NMS_V5_STATIC_BYTES device array built into the library (each GPU the library
 * runs on holds its copy until the process ends), and the call itself allocates nothing.  Shapes whose workspace
 * (ydet_nms_v5_workspace_bytes) exceeds that size are rejected with YDET_ERR_WORKSPACE: 64 images x 25 200 boxes fit.
 * The first call on a device resolves the array's address (one hipGetSymbolAddress; make that call outside a graph
 * capture).  All calls on one GPU share the one array: calls that overlap on a GPU (two streams, two threads) race;
 * serialise them or use ydet_nms_v5 with separate workspaces. */
#define YDET_NMS_V5_STATIC_BYTES (16u << 20)
int ydet_nms_v5_static(const float* pred, int N, int nboxes, int no, float conf_thres, float iou_thres,
                       int multi_label, int agnostic, int max_nms, int max_det, float max_wh,
                       const uint8_t* class_mask, const float* row_max, float* out_det, int32_t* out_count,
                       void* stream);

/* non_max_suppression2: yolov5_official/utils.py:318-460.  pred [N, nboxes, 5+nc] fp32 as for ydet_nms_v5.
 * Candidates: obj > conf_thres (:336), 2 < w, h < 4096 (:339), cls*obj > conf_thres for every class (multi_cls, or
 * conf_thres < 0.01, :352-359) or for the first-maximum class only (:360-362); descending score (:371; equal scores:
 * lower box first).  Per class the first 500 rows (:390-391) go through `method`:
 *   YDET_NMS2_HARD_BATCH / YDET_NMS2_HARD  torchvision's rule (suppress IoU > nms_thres), :377 / :395; the batch form
 *                          offsets boxes by class * (largest coordinate + 1) IN FP32 for any number of candidates
 *                          (torchvision's coordinate-offset formulation of batched_nms; releases that switch to a
 *                          per-class loop without offsets above 4000 boxes can differ from it at the IoU threshold
 *                          by the rounding of the offset coordinates, ~0.004 px at offsets near 5e4) and has no
 *                          500-row limit: a class with more than 2048 candidates sets *overflow
 *   YDET_NMS2_HARD_MYSELF  :398-405 (keep bbox_iou2 < nms_thres)      YDET_NMS2_DIOU  :447-454 (DIoU instead)
 *   YDET_NMS2_AND          :408-413 (a head needs a neighbour with IoU > 0.5)
 *   YDET_NMS2_MERGE        :416-425 (score-weighted mean of the head's cluster; nms_thres must be below 1)
 *   YDET_NMS2_SOFT         :428-444 (scores decay by exp(-iou^2 / 0.5), rows at or below conf_thres leave)
 * Output per image: out_rows [N, out_cap, 6] = (x1, y1, x2, y2, score, cls) in descending score order (:456-458),
 * out_count [N] rows valid (an image without detections has 0 rows where the reference leaves None); out_cap >=
 * ydet_nms2_out_cap(no, method) holds every case.  *overflow (device int32) is set non-zero when one (image, class)
 * list could not hold its candidates (more than 2048 scores inside one histogram bin): the result is then incomplete. */
enum { YDET_NMS2_HARD_BATCH = 0, YDET_NMS2_HARD = 1, YDET_NMS2_HARD_MYSELF = 2, YDET_NMS2_AND = 3, YDET_NMS2_MERGE = 4,
       YDET_NMS2_SOFT = 5, YDET_NMS2_DIOU = 6 };
size_t ydet_nms2_workspace_bytes(int N, int nboxes, int no, int method);
int ydet_nms2_out_cap(int no, int method);
int ydet_nms2(const float* pred, int N, int nboxes, int no, float conf_thres, float nms_thres, int multi_cls,
              int method, float* out_rows, int out_cap, int32_t* out_count, int32_t* overflow, void* workspace,
              size_t workspace_bytes, void* stream);

/* diou_non_max_suppression: yolov5_official/utils.py:463-548 (Cluster-Weighted DIoU-NMS, the alternative the author
 * toggles at yolov5_official/val.py:89-90).  Candidates as in non_max_suppression with multi_label = (nc > 1) (:475),
 * optional class filter (:505-506), descending score, the first max_box (<= 2048) of an image (:521-523); their
 * pairwise jaccard_diou (:210-244) on boxes offset by class * 4096 (agnostic: no offset), upper triangle; the
 * row-masking iteration of :528-535 (at most 200 rounds); keep = column maximum <= iou_thres (:536); coordinates
 * replaced by the score-weighted mean over the entries above 0.8 of each row (:538-543).
 * out_rows [N, max_box, 6]: the kept rows of every image in slot order, out_count [N]; like the reference the result
 * includes one row (NaN box, score 0, class 0) per unused slot.  iou_thres must be positive. */
size_t ydet_nms_cluster_diou_workspace_bytes(int N, int nboxes, int no, int max_box);
int ydet_nms_cluster_diou(const float* pred, int N, int nboxes, int no, float conf_thres, float iou_thres,
                          int max_box, int agnostic, const uint8_t* class_mask, float* out_rows, int32_t* out_count,
                          void* workspace, size_t workspace_bytes, void* stream);

/* postprocess + nms: utils/utils.py:101-133 and :15-42.
 * boxes [N, n, 4], scores [N, n, nc] fp32.  Per image: cls = argmax (first maximum),
 * score = max, candidates have score >= conf_thresh; per class greedy NMS in descending score
 * (equal scores: higher box index first, i.e. a stable ascending sort reversed) with
 * w,h = max(1e-28, .) and survivors ovr <= nms_thresh (NaN suppresses).
 * out_keep [N, n] uint8 (1 = kept, original box order), out_cls [N, n] int32 (argmax class),
 * out_score [N, n] fp32 (best score).
 * Workspace: ..._workspace_bytes holds, beside the index lists, room for the suppression bit matrices of long class
 * segments (256 boxes and more: every pair tested by the whole chip at once, then one workgroup per segment walks the
 * rows; the upper triangle of n x n bits per image); anything from ..._workspace_bytes_min up is accepted, and segments
 * whose matrix finds no room are walked round by round instead (same result, slower when thousands of boxes of one
 * class survive). */
size_t ydet_nms_perclass_workspace_bytes(int N, int n, int nc);
size_t ydet_nms_perclass_workspace_bytes_min(int N, int n, int nc);
int ydet_nms_perclass(const float* boxes, const float* scores, int N, int n, int nc,
                      float conf_thresh, float nms_thresh, uint8_t* out_keep, int32_t* out_cls,
                      float* out_score, void* workspace, size_t workspace_bytes, void* stream);

/* box_iou: yolov5_official/utils.py:601-611.  a [na,4], b [nb,4] xyxy fp32 -> iou [na, nb]. */
int ydet_box_iou(const float* a, int na, const float* b, int nb, float* iou, void* stream);

/* process_batch: yolov5_official/utils.py:1103-1123 (the TP matrix of yolov5_official/val.py:27-186), for a batch of
 * images in one launch.  det [B, max_det, 6] (xyxy, conf, cls; rows >= count[b] ignored) and count [B] as written by
 * ydet_nms_v5 (boxes already in the labels' coordinate space); labels [L, 5] (cls, xyxy) grouped by image,
 * label_start [B + 1] their offsets; iouv [niou] ascending thresholds.  correct [B, max_det, niou] uint8: row j is
 * `best_iou > iouv` when prediction j is the first of its class to reach IoU > iouv[0] with its best same-class
 * label, else zeros.  All pointers are device pointers.  max_det <= 4096. */
int ydet_process_batch(const float* det, const int32_t* count, int B, int max_det, const float* labels,
                       const int32_t* label_start, const float* iouv, int niou, uint8_t* correct, void* stream);

/* The detection-to-ground-truth matching of the VOC metric: data/vocapi_evaluator.py:284-318 (the loop over one
 * class's detections in descending confidence inside voc_eval), for every class and image of an evaluation in one
 * launch.  A group is one (class, image) pair.  boxes [n_det, 4] float64 xyxy as the reference reads them back from
 * its result files; members [n_det] lists detection indices group by group, inside a group in the order the reference
 * visits them (descending confidence; the caller's argsort fixes ties); group_start [n_groups + 1] offsets into
 * members; gt_boxes [m, 4] float64 / gt_difficult [m] grouped the same way with offsets gt_start [n_groups + 1].
 * flags [n_det] (indexed by detection): 1 true positive, 2 false positive, 0 neither (best box is "difficult").
 * Overlaps are float64, the best box is the first maximum, a NaN overlap wins the maximum and is a false positive,
 * exactly as the numpy expressions of the reference behave.  workspace: ydet_voc_match_workspace_bytes(n_det). */
size_t ydet_voc_match_workspace_bytes(int n_det);
int ydet_voc_match(const double* boxes, int n_det, const int32_t* members, const int32_t* group_start, int n_groups,
                   const double* gt_boxes, const uint8_t* gt_difficult, const int32_t* gt_start, double ovthresh,
                   uint8_t* flags, void* workspace, size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------------------------
 * Image pre-processing: a batch of uint8 HWC images (3 interleaved channels, any sizes) -> the network input.
 * Replaces the host chain  letterbox (yolov5_official/utils.py:40-75: cv2.resize INTER_LINEAR + cv2.copyMakeBorder
 * with 114) -> img[:, :, ::-1].transpose(2, 0, 1) -> img / 255 (yolov5_official/val.py:68-69), and
 * base_transform (data/transformers.py:456-471: cv2.resize -> /255 -> -mean -> /std) -> img[:, :, (2, 1, 0)] ->
 * permute(2, 0, 1) (data/dataset.py:224-229).
 *
 * Per image: the source is resampled to rw x rh (== w x h: copied; exactly half in both directions: 2x2 box mean;
 * otherwise OpenCV's 8-bit bilinear arithmetic, see csrc/preproc.hip) and placed with its top-left corner at
 * (left, top) of the out_w x out_h canvas; canvas pixels outside it take pad_value.  The geometry (rw, rh, left,
 * top) is the caller's: letterbox and base_transform differ only there.  Then per channel, in the source's channel
 * order c:  scale_mode 0: v;  1: v / 255;  2: (v / 255 - mean[c]) / std[c]  (IEEE fp32, in this order).  Output
 * channel k shows source channel (swap_rb ? 2 - k : k).  out: [N, 3, out_h, out_w] (layout 0) or
 * [N, out_h, out_w, 3] (layout 1) of out_dtype (YDET_U8 only with scale_mode 0).
 * ------------------------------------------------------------------------------------------ */
#define YDET_U8 2 /* output dtype of ydet_preprocess_u8, image dtype of ydet_focus_conv3x3_nchw */
typedef struct {
    const uint8_t* data;  /* device pointer; pixel (y, x) channel c at data[y*pitch + 3*x + c] */
    int32_t h, w, pitch;
    int32_t rh, rw;       /* resampled size */
    int32_t top, left;    /* position of the resampled image in the canvas */
    int32_t reserved;
} ydet_u8_image;
typedef struct {
    int32_t N, out_h, out_w;
    int32_t out_dtype, layout, swap_rb, scale_mode;
    int32_t pad_value[3]; /* per source channel, 0..255 */
    float mean[3], std[3];/* per source channel (scale_mode 2) */
} ydet_preproc_desc;
int ydet_preprocess_u8(const ydet_u8_image* images /* device, N entries */, const ydet_preproc_desc* d /* host */,
                       void* out, void* stream);

/* ------------------------------------------------------------------------------------------
 * Launch-list executor: one call runs a whole pre-planned forward pass
 * (the Python loop of Model.forward_once, yolov5_official/model.py:220-234, and the forward()
 * bodies of models/v2d19.py, v3tiny.py, v3spp.py, v4.py).
 * ------------------------------------------------------------------------------------------ */
typedef enum {
    YDET_OP_CONV = 0,
    YDET_OP_MAXPOOL2 = 1,
    YDET_OP_MAXPOOL2S1 = 2,
    YDET_OP_SPP = 3,
    YDET_OP_UPSAMPLE = 4,      /* iarg = mode */
    YDET_OP_SPACE2DEPTH = 5,   /* iarg = mode */
    YDET_OP_COPY = 6,
    YDET_OP_NCHW2NHWC = 7,     /* slice: x=src NCHW, C = real channels; iarg = focus, iarg2 = c_pad,
                                  iarg3 = source dtype */
    YDET_OP_NHWC2NCHW = 8,     /* iarg3 = destination dtype */
    YDET_OP_FOCUS_CONV = 9,    /* conv: as ydet_focus_conv3x3_nchw; iarg3 = image dtype */
    YDET_OP_HEAD_DECODE = 10,  /* conv + head: as ydet_head_conv_decode */
    YDET_OP_BOTTLENECK = 11,/* conv + conv2: as ydet_bottleneck_fused */
    YDET_OP_C3_TAIL = 12,   /* conv + conv2 + conv3: as ydet_c3_tail_fused */
    YDET_OP_RGB_CONV = 13,  /* conv: as ydet_rgb_conv3x3_nchw; iarg3 = image dtype */
    YDET_OP_FOCUS_PAIR = 14,/* conv + conv2: as ydet_focus_conv_pair_nchw; iarg3 = image dtype */
    YDET_OP_RGB_PAIR = 15   /* conv + conv2: as ydet_rgb_conv_pair_nchw; iarg3 = image dtype */
} ydet_op_kind;

typedef struct {
    int32_t kind;
    int32_t iarg, iarg2, iarg3;
    ydet_conv_desc conv;   /* YDET_OP_CONV, YDET_OP_FOCUS_CONV, YDET_OP_HEAD_DECODE */
    ydet_slice_desc slice; /* pooling / resampling / layout ops */
    ydet_head_desc head;   /* YDET_OP_HEAD_DECODE */
    ydet_conv_desc conv2;  /* YDET_OP_BOTTLENECK, YDET_OP_C3_TAIL: the 3x3 stage */
    ydet_conv_desc conv3;  /* YDET_OP_C3_TAIL: cv3 */
} ydet_op;

int ydet_run_ops(const ydet_op* ops /* host */, int n_ops, void* stream);

/* Several independent launch lists (e.g. the same network planned for two halves of the batch) run side by side:
 * list 0 on `stream`, the others on side streams the library keeps per device, forked after the work already queued
 * on `stream` and joined back into it before the call returns -- to the caller it behaves like one launch on `stream`.
 * Kernels of different lists overlap on the GPU, which fills the tails of grids that do not cover all 256 CUs.
 * n_lists <= 4.  No counterpart in the reference (its forward pass is one Python loop on one stream). */
int ydet_run_op_lists(const ydet_op* const* lists /* host */, const int* counts /* host */, int n_lists, void* stream);

/* Measurement aid (no counterpart in the reference, which has no profiler): the same executor, but a
 * hipEvent is recorded on `stream` before the first op and after every op, into marks
 * first_mark .. first_mark + n_ops of `timer`.  Nothing synchronises; after the caller has synchronised
 * the stream, ydet_timer_elapsed returns the milliseconds between consecutive marks. */
typedef struct ydet_timer ydet_timer;
ydet_timer* ydet_timer_create(int n_marks);
void ydet_timer_destroy(ydet_timer* t);
int ydet_run_ops_timed(const ydet_op* ops /* host */, int n_ops, void* stream, ydet_timer* t, int first_mark);
int ydet_timer_elapsed(ydet_timer* t, int first_mark, int n_intervals, float* ms /* host */);

#ifdef __cplusplus
}
#endif
#endif /* YDET_H */
