// K4: NHWC fp16 convolution as implicit GEMM on MFMA (gfx950), fused bias + SiLU + residual +
// concat-slice addressing.  Internal C++ interface used by the YOLOv8 runner and by the raw
// eioku_conv2d_f16 test entry point.
#pragma once

#include <hip/hip_fp16.h>

#include <cstdint>
#include <vector>

#include "common.h"

namespace eioku {

// A tensor is an NHWC fp16 buffer; a *slice* is `C` channels starting at channel `coff` of a buffer
// whose pixels are `cstride` channels apart (this is how concat / chunk are expressed: no copies).
struct Slice {
  __half* ptr = nullptr;
  int cstride = 0;
  int coff = 0;
};

struct ConvWeights {
  int cout = 0, cin = 0, ks = 1, stride = 1;
  int nf = 1;       // 16-wide cout fragments per workgroup tile
  int ntiles = 0;   // cout tiles = ceil(cout / (16*nf))
  int nchunks = 0;  // 32-channel input chunks = ceil(cin / 32)
  __half* d_w = nullptr;  // [ntiles][nchunks][taps][16*nf][32] fp16, zero padded
  float* d_b = nullptr;   // [ntiles*16*nf] fp32, zero padded
  double flops_per_pixel() const { return 2.0 * cout * cin * ks * ks; }
};

// Packs torch-layout fp32 weights [cout][cin][ks][ks] (+ bias[cout], may be null) for the kernel
// and uploads them.  Values are rounded to fp16 (RNE) exactly as `tensor.half()` does.
int conv_weights_create(ConvWeights* cw, int cout, int cin, int ks, int stride, const float* w,
                        const float* b);
void conv_weights_destroy(ConvWeights* cw);

// kActReLU: relu(conv + bias) [then + res]; kActResReLU: relu(fp16(conv + bias) + res) - the torchvision BasicBlock's
// order (the activation follows the residual sum); the two ReLU modes serve the ResNet18 of classify_places
enum ConvAct { kActNone = 0, kActSiLU = 1, kActReLU = 2, kActResReLU = 3 };

// out = act(conv(in) + bias) [+ res]; output fp16 into `out`, or fp32 into `out_f32` (dense
// [N,Ho,Wo,cout]) when out_f32 != nullptr.  H, W: input size; output is ceil-free standard
// (H + 2*pad - ks)/stride + 1 with pad = ks/2.
// `fused` (stem only: k=3, s=2, Cin=8): the input is not a tensor but K3's letterbox of a BGR u8 batch, evaluated
// inside the kernel's staging loads (mode 0 = copy, 2 = exact 1/2 area; H, W = letterboxed size); `in` is ignored.
struct FusedInput {
  const uint8_t* bgr;   // device, [N][src_h][src_w][3]
  int src_h, src_w;     // original frame
  int new_h, new_w;     // resized (unpadded) size
  int top, left;        // padding offsets
  int mode;             // 0: copy, 2: exact-1/2 area average
  // mode 0 only: letterboxed pixel (y, x) of the image region is source pixel (step y + off, step x + off).  1 / 0 = the
  // plain copy; step 3, off 1 = cv2.INTER_LINEAR at an exact 1/3 scale (1080p -> 360 x 640), whose tap weights are
  // (1, 0) everywhere: a decimation
  int step = 1, off = 0;
};
bool fused_input_ok(const ConvWeights& cw, const FusedInput& f, Slice res, const float* out_f32);
// `up` (1x1 only): the first c_split input channels are the nearest-2x upsample of `src` (half resolution), read in
// place at (y/2, x/2); the remaining channels come from `in` as usual (the neck's [up(x) | skip] concats).
struct UpSource {
  Slice src;
  int c_split;
};
// `post` (3x3 persistent kernel only, see conv_post_ok): a following 1x1 convolution applied to the tile while it
// is still on chip; `out` then receives post's output and cw's own output tensor is never written.
bool conv_post_ok(const ConvWeights& cw, const ConvWeights& post);
int conv_forward(const ConvWeights& cw, Slice in, int N, int H, int W, Slice out, float* out_f32,
                 Slice res, int act, hipStream_t stream, const FusedInput* fused = nullptr,
                 const ConvWeights* post = nullptr, int post_act = kActSiLU, unsigned long long* clsmax = nullptr,
                 const struct UpSource* up = nullptr);
// Two chained 3x3 stride-1 layers with 16 / 32 channels (a C2f Bottleneck) as one launch: out = act_b(B(act_a(A(in))))
// [+ in when `residual`]; the intermediate tensor stays in LDS.  Bit-identical to the two separate launches.
bool conv_chain_ok(const ConvWeights& a, const ConvWeights& b);
// `cat_w` (conv_chain_cat_ok: the 16-channel C2f with one Bottleneck): the C2f's closing 1x1 over the concat runs in
// the same launch; `cat_in` = the concat buffer (its leading chunks are read from memory, the Bottleneck's output
// stays on chip and `out` may be empty), `cat_out` = the 1x1's output slice.
bool conv_chain_cat_ok(const ConvWeights& a, const ConvWeights& b, const ConvWeights& c2);
int conv_chain_forward(const ConvWeights& a, const ConvWeights& b, Slice in, int N, int H, int W, Slice out,
                       bool residual, int act_a, int act_b, hipStream_t stream, const ConvWeights* cat_w = nullptr,
                       Slice cat_in = Slice{}, Slice cat_out = Slice{}, int cat_act = kActSiLU);
// YOLOv8n front end as one launch: fused letterbox (copy mode, 4-pixel aligned) -> stem (3 -> 16, s2) -> 3x3 s2
// (16 -> 32) -> 1x1 (32 -> 32); H, W = letterboxed input size, `out` = the 1x1's output slice.  Bit-identical to the
// separate launches (k_conv3x3_c8 + the 3x3+1x1 pair).
bool conv_stem_chain_ok(const ConvWeights& stem, const ConvWeights& c1, const ConvWeights& post, const FusedInput& f, int W);
int conv_stem_chain_forward(const ConvWeights& stem, const ConvWeights& c1, const ConvWeights& post, const FusedInput& f,
                            int N, int H, int W, Slice out, int act0, int act1, int act2, hipStream_t stream);
// `clsmax` (1x1, no activation, one cout tile): instead of the output tensor, per pixel one 64-bit word
// (argmax channel << 32 | float bits of max_c(conv + bias)); ties go to the lower channel.
bool conv_clsmax_ok(const ConvWeights& cw, int act);

inline int conv_out_dim(int x, int ks, int stride) { return (x + 2 * (ks / 2) - ks) / stride + 1; }

}  // namespace eioku
