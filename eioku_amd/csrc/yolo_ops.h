// K3 / K5 / K6 / K7: the non-GEMM kernels of the detection stage.
#pragma once

#include "conv.h"

namespace eioku {

// One detection in ORIGINAL image pixels (after scale_boxes + clip), plus provenance for parity.
struct Det {
  float x1, y1, x2, y2;
  float conf;
  int32_t cls;
  int32_t anchor;  // index into the concatenated anchor list (P3 | P4 | P5), row-major per level
  int32_t pad;
};
static_assert(sizeof(Det) == 32, "Det layout is part of the C ABI");

// Pre-NMS candidate in letterboxed-image pixels.
struct Cand {
  float x1, y1, x2, y2;
  float conf;
  int32_t cls;
  int32_t anchor;
  int32_t pad;
};

// K3. uint8 BGR (n,h,w,3) -> fp16 NHWC8 (n,H2,W2,8): cv2.resize INTER_LINEAR (fixed point 11 bit) or
// the 2x2 area fast path, 114 padding, BGR->RGB, /255, channels 3..7 zero.
//   xofs/yofs : source column/row of the first tap per destination column/row   (int32, new_w / new_h)
//   xa/ya     : first-tap fixed-point weight (second = 2048 - first handled as separate table)
struct LetterboxPlan {
  int src_h, src_w;        // original frame
  int new_h, new_w;        // resized (unpadded) size
  int top, left;           // padding offsets
  int out_h, out_w;        // letterboxed size (multiple of 32 for rect mode)
  int mode;                // 0 = copy (no resize), 1 = bilinear fixed point, 2 = 2x2 area
  const int32_t* xofs;     // [new_w]  (device)
  const int32_t* yofs;     // [new_h]
  const int16_t* xalpha;   // [new_w][2]
  const int16_t* ybeta;    // [new_h][2]
};
int letterbox_forward(const uint8_t* bgr, int n, const LetterboxPlan& p, __half* out_nhwc8, hipStream_t stream);

// K5. 5x5 stride-1 max pool (pad 2, -inf) on a channel slice -> another slice (SPPF chain).
int maxpool5_forward(Slice in, Slice out, int N, int H, int W, int C, hipStream_t stream);
// K5. three chained pools (SPPF): out, out + out_step, out + 2*out_step channels receive pool, pool^2, pool^3 of `in`.
int sppf_pools_forward(Slice in, Slice out, int out_step, int N, int H, int W, int C, hipStream_t stream);
// K5. nearest 2x upsample of a slice into a slice of a (2H,2W) buffer (fused concat write).
int upsample2x_forward(Slice in, Slice out, int N, int H, int W, int C, hipStream_t stream);

// K6. DFL decode + class max + confidence filter -> compacted per-image candidate lists.
//   box[l]: fp32 [N,Hl,Wl,64], cls[l]: fp32 [N,Hl,Wl,nc]; strides 8/16/32.
//   cands: [N][max_cand], counts: [N] (must be zeroed by the caller).
int decode_forward(const float* const box[3], const float* const cls[3], int N, const int Hl[3],
                   const int Wl[3], int nc, float conf_thres, Cand* cands, int32_t* counts,
                   int max_cand, hipStream_t stream, const unsigned long long* const clsmax[3] = nullptr);
//   clsmax[l] (optional, then cls may be null): [N,Hl,Wl] words (argmax << 32 | max-logit bits) from the class conv.

// K6 with a lazy box branch (detect()): `clsmax` words decide which anchors pass; only for those the box branch's
// last 1x1 conv (lb: its fp16 input per level, packed weights with ONE 64-row tile, bias) is evaluated -- into the
// same rows of box[l] the dense conv would have written -- and decoded.  lvl_list [N][A] / lvl_counts [N][3]
// (zeroed by the caller) are workspace.
struct LazyConv3 {  // one 3x3 s1 layer with 64 output channels, bias + SiLU, evaluated at listed pixels
  const __half* in;
  int in_cs, nchunks;
  const uint4* wgt;  // packed [tile][chunk][tap][rows_tile][32]
  int rows_tile;     // 16 * nf of the packing
  const float* bias;
  __half* out;
  int out_cs;
};
struct LazyBox {
  const __half* in[3];
  const uint4* wgt[3];
  const float* bias[3];
  int in_cs, nchunks;
  // deep = true: the branch's two 3x3 layers are lazy as well (c0 at the 3x3 neighbourhoods of the passing anchors,
  // c1 at the anchors); flat1 / flat0 [level] are pixel-list workspaces (N*A_l and 9*N*A_l ints), fcnt 6 zeroed ints
  bool deep = false;
  LazyConv3 c0[3], c1[3];
  int32_t* flat1[3];
  int32_t* flat0[3];
  int32_t* fcnt;
};
int decode_lazy_forward(float* const box[3], const unsigned long long* const clsmax[3], const LazyBox& lb, int N,
                        const int Hl[3], const int Wl[3], int nc, float conf_thres, Cand* cands, int32_t* counts,
                        int32_t* lvl_list, int32_t* lvl_counts, int max_cand, hipStream_t stream);

// K7. per-image: sort by (conf desc, anchor asc), class-aware greedy NMS (IoU > thr suppresses),
// keep <= max_det, then scale_boxes to the original frame and clip.
struct ScaleParams {
  float gain;        // (float)min(out_h/src_h, out_w/src_w)
  float pad_x, pad_y;
  float src_w, src_h;
};
int nms_forward(const Cand* cands, const int32_t* counts, int N, int max_cand, float iou_thres,
                int max_det, float max_wh, ScaleParams sp, Det* dets, int32_t* det_counts,
                hipStream_t stream);

}  // namespace eioku
