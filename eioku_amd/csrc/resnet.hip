// K11: Places365 scene classification - ResNet18 + softmax top-k.
//
// Replaces what ModelManager.classify_places computes per sampled frame
// (/root/reference/ml-service/src/services/model_manager.py:560-713): PIL antialiased bilinear resize to 224 x 224
// (transforms.Resize on a PIL image), ToTensor + Normalize, torchvision resnet18 with a 365-way fc, softmax, descending
// sort, top_k.  On the device:
//   k_pil_resize_h / k_pil_resize_v   Pillow's two-pass 8-bit resample (22-bit fixed-point taps from the host, uint8
//                                     intermediate, bit-exact), the second pass fused with ToTensor / Normalize -> fp16 NHWC4
//   k_stem7x7                         7x7 / stride 2 / pad 3 convolution 3 -> 64 + bias + ReLU as implicit GEMM on
//                                     v_mfma_f32_16x16x32_f16: the K axis of a k-step is 8 taps x 4 channels, a lane's B
//                                     fragment two 8-byte pixels of the LDS patch (HBM-bound: 0.4 MB in, 1.6 MB out per frame)
//   k_maxpool3s2                      3x3 / stride 2 / pad 1 max pool, 8 channels per thread
//   the BasicBlocks                   K4's 3x3 kernels (conv.hip) with the two ReLU epilogues (kActReLU, kActResReLU); the
//                                     stride-2 1x1 of a downsample branch runs as a 3x3 / stride 2 whose only non-zero tap
//                                     is the centre (pad 1 puts that tap on input pixel (2y, 2x)): three small layers, no
//                                     new kernel
//   k_places_head                     global 7x7 average pool + fc (fp16 weights, fp32 accumulate) + softmax + full
//                                     descending sort of the 365 probabilities (LDS bitonic) -> top_k (prob, class)
// Numerics: fp16 storage of weights and activations, fp32 accumulation - the detector's arithmetic; the reference runs
// fp32 (tests bound the drift against the fp32 oracle and assert exact top-k classes where the logits separate them).
#include <array>
#include <string>
#include <vector>

#include "conv.h"

using namespace eioku;

namespace {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef float float4v __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr int kIn = 224;
constexpr int kPrec = 22;  // Pillow Resample.c PRECISION_BITS for 8-bit pixels

// ---- Pillow's resample ------------------------------------------------------------------------------------------
// bounds [out][2] = {first input index, taps}; kk [out][ksize] int32 taps scaled by 2^22 (host: places.py / oracle)
// horizontal: src [N][h][w][3] u8 -> tmp [N][h][224][3] u8; one thread per (row, output column)
__global__ __launch_bounds__(256) void k_pil_resize_h(const uint8_t* __restrict__ src, int N, int h, int w,
                                                      const int* __restrict__ bounds, const int* __restrict__ kk, int ksize,
                                                      uint8_t* __restrict__ tmp) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= (long long)N * h * kIn) return;
  const int xx = (int)(i % kIn);
  const long long row = i / kIn;
  const int lo = bounds[2 * xx], n = bounds[2 * xx + 1];
  const uint8_t* p = src + ((size_t)row * w + lo) * 3;
  int s0 = 1 << (kPrec - 1), s1 = s0, s2 = s0;
  for (int t = 0; t < n; ++t) {
    const int k = kk[xx * ksize + t];
    s0 += p[3 * t] * k;
    s1 += p[3 * t + 1] * k;
    s2 += p[3 * t + 2] * k;
  }
  auto clip8 = [](int v) { return (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v)); };
  uint8_t* o = tmp + (size_t)i * 3;
  o[0] = clip8(s0 >> kPrec);
  o[1] = clip8(s1 >> kPrec);
  o[2] = clip8(s2 >> kPrec);
}

// vertical + ToTensor + Normalize: tmp [N][h][224][3] u8 (B, G, R) -> out [N][224][224][4] fp16 (R, G, B, 0) with
// fp32 (v / 255 - mean) / std in torch's op order, one RNE rounding to fp16
__global__ __launch_bounds__(256) void k_pil_resize_v(const uint8_t* __restrict__ tmp, int N, int h,
                                                      const int* __restrict__ bounds, const int* __restrict__ kk, int ksize,
                                                      __half* __restrict__ out, uint8_t* __restrict__ out_u8) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= (long long)N * kIn * kIn) return;
  const int x = (int)(i % kIn);
  const int yy = (int)((i / kIn) % kIn);
  const int n = (int)(i / (kIn * kIn));
  const int lo = bounds[2 * yy], cnt = bounds[2 * yy + 1];
  const uint8_t* p = tmp + (((size_t)n * h + lo) * kIn + x) * 3;
  int s0 = 1 << (kPrec - 1), s1 = s0, s2 = s0;
  for (int t = 0; t < cnt; ++t) {
    const int k = kk[yy * ksize + t];
    s0 += p[(size_t)t * kIn * 3] * k;
    s1 += p[(size_t)t * kIn * 3 + 1] * k;
    s2 += p[(size_t)t * kIn * 3 + 2] * k;
  }
  auto clip8 = [](int v) { return v < 0 ? 0 : (v > 255 ? 255 : v); };
  const int b = clip8(s0 >> kPrec), g = clip8(s1 >> kPrec), r = clip8(s2 >> kPrec);
  if (out_u8) {  // parity helper: the resized RGB image itself
    out_u8[(size_t)i * 3] = (uint8_t)r;
    out_u8[(size_t)i * 3 + 1] = (uint8_t)g;
    out_u8[(size_t)i * 3 + 2] = (uint8_t)b;
  }
  if (out) {
    const float fr = ((float)r / 255.0f - 0.485f) / 0.229f;
    const float fg = ((float)g / 255.0f - 0.456f) / 0.224f;
    const float fb = ((float)b / 255.0f - 0.406f) / 0.225f;
    const f16x4 v = {(_Float16)fr, (_Float16)fg, (_Float16)fb, (_Float16)0.f};
    *reinterpret_cast<u32x2*>(out + (size_t)i * 4) = __builtin_bit_cast(u32x2, v);
  }
}

// ---- stem: 7x7 / s2 / p3, 4 -> 64 channels (channel 3 is zero), + bias + ReLU -----------------------------------
// Workgroup = 4 waves -> one 8 x 16 tile of output pixels of one image, all 64 couts (4 fragments).  The 21 x 37 input
// patch (8 B per pixel) sits in LDS; k-step s covers taps 8s .. 8s + 7 (49 taps padded to 56 with zero weights; the
// padded taps re-read tap 48's pixel so that no lane multiplies garbage), lane (r, u) taking taps 8s + 2u, 8s + 2u + 1.
// Weights: [7 k-steps][4 fragments][64 lanes] 16-byte A operands, packed on the host.
constexpr int kSTH = 8, kSTW = 16, kSPH = (kSTH - 1) * 2 + 7, kSPW = (kSTW - 1) * 2 + 7, kSteps = 7;
__global__ __launch_bounds__(256) void k_stem7x7(const __half* __restrict__ in, int N, const uint4* __restrict__ wgt,
                                                 const float* __restrict__ bias, __half* __restrict__ out) {
  constexpr int Ho = kIn / 2, Wo = kIn / 2;
  __shared__ __attribute__((aligned(16))) unsigned long long patch[kSPH * kSPW + 3];
  __shared__ __attribute__((aligned(16))) uint4 wt[kSteps * 4 * 64];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int tw = blockIdx.x, th = blockIdx.y, n = blockIdx.z;
  const int ih0 = th * kSTH * 2 - 3, iw0 = tw * kSTW * 2 - 3;
  for (int i = tid; i < kSPH * kSPW; i += 256) {
    const int py = i / kSPW, px = i - py * kSPW;
    const int ih = ih0 + py, iw = iw0 + px;
    unsigned long long v = 0;
    if ((unsigned)ih < (unsigned)kIn && (unsigned)iw < (unsigned)kIn)
      v = *reinterpret_cast<const unsigned long long*>(in + (((size_t)n * kIn + ih) * kIn + iw) * 4);
    patch[i] = v;
  }
  for (int i = tid; i < kSteps * 4 * 64; i += 256) wt[i] = wgt[i];
  __syncthreads();
  const int r = lane & 15, u = lane >> 4;
  float4v acc[2][4];
#pragma unroll
  for (int m = 0; m < 2; ++m)
#pragma unroll
    for (int f = 0; f < 4; ++f) acc[m][f] = float4v{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int s = 0; s < kSteps; ++s) {
    int t0 = 8 * s + 2 * u, t1 = t0 + 1;
    t0 = t0 > 48 ? 48 : t0;
    t1 = t1 > 48 ? 48 : t1;
    const int ty0 = t0 / 7, tx0 = t0 - 7 * ty0, ty1 = t1 / 7, tx1 = t1 - 7 * ty1;
    half8 b[2];
#pragma unroll
    for (int m = 0; m < 2; ++m) {
      const int py = (wave * 2 + m) * 2, px = r * 2;
      const unsigned long long lo = patch[(py + ty0) * kSPW + px + tx0], hi = patch[(py + ty1) * kSPW + px + tx1];
      const u32x4 v = {(unsigned)lo, (unsigned)(lo >> 32), (unsigned)hi, (unsigned)(hi >> 32)};
      b[m] = __builtin_bit_cast(half8, v);
    }
#pragma unroll
    for (int f = 0; f < 4; ++f) {
      const uint4 w = wt[(s * 4 + f) * 64 + lane];
      const half8 a = *reinterpret_cast<const half8*>(&w);
#pragma unroll
      for (int m = 0; m < 2; ++m) acc[m][f] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b[m], acc[m][f], 0, 0, 0);
    }
  }
  const int ow = tw * kSTW + r;
#pragma unroll
  for (int m = 0; m < 2; ++m) {
    const int oh = th * kSTH + wave * 2 + m;
    if (oh >= Ho || ow >= Wo) continue;
    __half* o = out + (((size_t)n * Ho + oh) * Wo + ow) * 64;
#pragma unroll
    for (int f = 0; f < 4; ++f) {
      const int c0 = f * 16 + u * 4;
      const float4 bb = *reinterpret_cast<const float4*>(bias + c0);
      float4v v = acc[m][f] + float4v{bb.x, bb.y, bb.z, bb.w};
      v = __builtin_elementwise_max(v, float4v{0.f, 0.f, 0.f, 0.f});
      const f16x4 hv = __builtin_convertvector(v, f16x4);
      *reinterpret_cast<u32x2*>(o + c0) = __builtin_bit_cast(u32x2, hv);
    }
  }
}

// 3x3 / stride 2 / pad 1 max pool over NHWC fp16, C % 8 == 0: one thread per (output pixel, 8 channels)
__global__ __launch_bounds__(256) void k_maxpool3s2(const __half* __restrict__ in, int N, int H, int W, int C,
                                                    __half* __restrict__ out) {
  const int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1, C8 = C / 8;
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= (long long)N * Ho * Wo * C8) return;
  const int c8 = (int)(i % C8);
  long long p = i / C8;
  const int ox = (int)(p % Wo);
  p /= Wo;
  const int oy = (int)(p % Ho), n = (int)(p / Ho);
  half8 best;
#pragma unroll
  for (int j = 0; j < 8; ++j) best[j] = (_Float16)(-65504.f);
  for (int dy = 0; dy < 3; ++dy) {
    const int y = oy * 2 - 1 + dy;
    if ((unsigned)y >= (unsigned)H) continue;
    for (int dx = 0; dx < 3; ++dx) {
      const int x = ox * 2 - 1 + dx;
      if ((unsigned)x >= (unsigned)W) continue;
      const uint4 v = *reinterpret_cast<const uint4*>(in + (((size_t)n * H + y) * W + x) * C + c8 * 8);
      const half8 hv = *reinterpret_cast<const half8*>(&v);
#pragma unroll
      for (int j = 0; j < 8; ++j) best[j] = hv[j] > best[j] ? hv[j] : best[j];
    }
  }
  *reinterpret_cast<uint4*>(out + (((size_t)n * Ho + oy) * Wo + ox) * C + c8 * 8) = *reinterpret_cast<const uint4*>(&best);
}

// global average pool (HW pixels x 512 channels) -> fc (nc x 512 fp16 weights, fp32 accumulate) -> softmax -> descending
// sort by (probability desc, class asc) -> the first top_k.  One workgroup per image; nc <= 512.
__global__ __launch_bounds__(256) void k_places_head(const __half* __restrict__ x, int HW, const __half* __restrict__ fcw,
                                                     const float* __restrict__ fcb, int nc, int top_k,
                                                     float* __restrict__ logits_out, float* __restrict__ prob_out,
                                                     int* __restrict__ idx_out) {
  __shared__ float s_pool[512];
  __shared__ float s_val[512];
  __shared__ int s_idx[512];
  __shared__ float s_red[4];
  const int n = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const __half* xi = x + (size_t)n * HW * 512;
  for (int c = tid; c < 512; c += 256) {
    float s = 0.f;
    for (int p = 0; p < HW; ++p) s += __half2float(xi[(size_t)p * 512 + c]);
    s_pool[c] = s / (float)HW;
  }
  __syncthreads();
  for (int j = wave; j < nc; j += 4) {  // a wave per output: lane l owns channels 8 l .. 8 l + 7
    const uint4 wv = *reinterpret_cast<const uint4*>(fcw + (size_t)j * 512 + lane * 8);
    const half8 w = *reinterpret_cast<const half8*>(&wv);
    float s = 0.f;
#pragma unroll
    for (int t = 0; t < 8; ++t) s += s_pool[lane * 8 + t] * (float)w[t];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    if (lane == 0) {
      s += fcb[j];
      s_val[j] = s;
      if (logits_out) logits_out[(size_t)n * nc + j] = s;
    }
  }
  __syncthreads();
  // softmax in fp32: exp(v - max) / sum
  float m = -INFINITY;
  for (int j = tid; j < nc; j += 256) m = fmaxf(m, s_val[j]);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
  if (lane == 0) s_red[wave] = m;
  __syncthreads();
  m = fmaxf(fmaxf(s_red[0], s_red[1]), fmaxf(s_red[2], s_red[3]));
  __syncthreads();
  float e[2], sum = 0.f;
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    const int j = tid + 256 * t;
    e[t] = j < nc ? expf(s_val[j] - m) : 0.f;
    sum += e[t];
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o, 64);
  if (lane == 0) s_red[wave] = sum;
  __syncthreads();
  sum = (s_red[0] + s_red[1]) + (s_red[2] + s_red[3]);
  __syncthreads();
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    const int j = tid + 256 * t;
    s_val[j] = j < nc ? e[t] / sum : -1.f;  // padding sorts last
    s_idx[j] = j;
  }
  __syncthreads();
  // bitonic sort of 512 (value desc, index asc)
  for (int k = 2; k <= 512; k <<= 1)
    for (int j = k >> 1; j > 0; j >>= 1) {
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const int i = tid + 256 * t, l = i ^ j;
        if (l > i) {
          const float vi = s_val[i], vl = s_val[l];
          const int ii = s_idx[i], il = s_idx[l];
          const bool i_first = vi > vl || (vi == vl && ii < il);  // i belongs before l in descending order
          const bool up = (i & k) == 0;
          if (up ? !i_first : i_first) {
            s_val[i] = vl; s_val[l] = vi;
            s_idx[i] = il; s_idx[l] = ii;
          }
        }
      }
      __syncthreads();
    }
  for (int r = tid; r < top_k; r += 256) {
    prob_out[(size_t)n * top_k + r] = s_val[r];
    idx_out[(size_t)n * top_k + r] = s_idx[r];
  }
}

struct Layer {
  std::string name;
  int cout, cin, k, stride;
};

std::vector<Layer> resnet18_layers() {
  std::vector<Layer> L;
  L.push_back({"conv1", 64, 3, 7, 2});
  const int widths[4] = {64, 128, 256, 512};
  for (int li = 0; li < 4; ++li) {
    const int c = widths[li], s = li == 0 ? 1 : 2, cin0 = li == 0 ? 64 : c / 2;
    for (int b = 0; b < 2; ++b) {
      const int st = b == 0 ? s : 1, ci = b == 0 ? cin0 : c;
      const std::string p = "layer" + std::to_string(li + 1) + "." + std::to_string(b);
      L.push_back({p + ".conv1", c, ci, 3, st});
      L.push_back({p + ".conv2", c, c, 3, 1});
      if (b == 0 && (st != 1 || ci != c)) L.push_back({p + ".downsample.0", c, ci, 1, st});
    }
  }
  return L;
}

}  // namespace

struct eioku_resnet {
  int nc = 365;
  std::vector<Layer> layers;
  std::vector<ConvWeights> w;   // index = layer (0 unused: the stem has its own packing)
  std::vector<bool> set;
  uint4* stem_w = nullptr;
  float* stem_b = nullptr;
  __half* fc_w = nullptr;
  float* fc_b = nullptr;
  bool fc_set = false;
  // activations for the current batch size
  int cap_n = 0;
  __half* in224 = nullptr;   // [N][224][224][4]
  __half* a112 = nullptr;    // [N][112][112][64]
  __half* buf[4] = {};       // [N][56][56][64] each (the largest block tensor)
  uint8_t* tmp = nullptr;    // resize intermediate [N][h][224][3]
  size_t tmp_cap = 0;
  uint8_t* src = nullptr;    // staged host frames
  size_t src_cap = 0;
  int* tables = nullptr;     // xbounds | xk | ybounds | yk
  size_t tables_cap = 0;
  float* logits = nullptr;   // [N][nc]
  float* probs = nullptr;    // [N][nc]
  int* idx = nullptr;
  double flops_last = 0;
};

namespace {

int ensure_batch(eioku_resnet* r, int N) {
  if (N <= r->cap_n) return EIOKU_OK;
  for (void* p : {(void*)r->in224, (void*)r->a112, (void*)r->buf[0], (void*)r->buf[1], (void*)r->buf[2], (void*)r->buf[3],
                  (void*)r->logits, (void*)r->probs, (void*)r->idx})
    if (p) (void)hipFree(p);
  r->cap_n = 0;
  EIOKU_HIP_CHECK(hipMalloc((void**)&r->in224, (size_t)N * kIn * kIn * 4 * 2));
  EIOKU_HIP_CHECK(hipMalloc((void**)&r->a112, (size_t)N * 112 * 112 * 64 * 2));
  for (int i = 0; i < 4; ++i) EIOKU_HIP_CHECK(hipMalloc((void**)&r->buf[i], (size_t)N * 56 * 56 * 64 * 2));
  EIOKU_HIP_CHECK(hipMalloc((void**)&r->logits, (size_t)N * r->nc * 4));
  EIOKU_HIP_CHECK(hipMalloc((void**)&r->probs, (size_t)N * 512 * 4));
  EIOKU_HIP_CHECK(hipMalloc((void**)&r->idx, (size_t)N * 512 * 4));
  r->cap_n = N;
  return EIOKU_OK;
}

// the network on r->in224 -> (logits, sorted probabilities, classes) in the handle's buffers
int run_network(eioku_resnet* r, int N, int top_k, hipStream_t stream) {
  for (size_t i = 0; i < r->layers.size(); ++i) EIOKU_REQUIRE(r->set[i], "convolution %s has no weights", r->layers[i].name.c_str());
  EIOKU_REQUIRE(r->fc_set, "fc has no weights");
  double flops = 0;
  hipLaunchKernelGGL(k_stem7x7, dim3(112 / kSTW, 112 / kSTH, (unsigned)N), dim3(256), 0, stream, r->in224, N, r->stem_w, r->stem_b,
                     r->a112);
  EIOKU_LAUNCH_CHECK();
  flops += 2.0 * 64 * 3 * 49 * 112 * 112 * N;
  {
    const long long work = (long long)N * 56 * 56 * 8;
    hipLaunchKernelGGL(k_maxpool3s2, dim3((unsigned)((work + 255) / 256)), dim3(256), 0, stream, r->a112, N, 112, 112, 64, r->buf[0]);
    EIOKU_LAUNCH_CHECK();
  }
  int x = 0, H = 56;  // buf[x] holds the block input at H x H
  size_t li = 1;
  for (int stage = 0; stage < 4; ++stage)
    for (int b = 0; b < 2; ++b) {
      const Layer& c1 = r->layers[li];
      const Layer& c2 = r->layers[li + 1];
      const bool down = li + 2 < r->layers.size() && r->layers[li + 2].name.find("downsample") != std::string::npos &&
                        r->layers[li + 2].name.compare(0, 8, c1.name, 0, 8) == 0;
      const int t = (x + 1) & 3, idn = (x + 2) & 3, y = (x + 3) & 3;
      const int Ho = conv_out_dim(H, 3, c1.stride);
      Slice in{r->buf[x], c1.cin, 0}, mid{r->buf[t], c1.cout, 0}, out{r->buf[y], c2.cout, 0}, res = in;
      int rc = conv_forward(r->w[li], in, N, H, H, mid, nullptr, Slice{}, kActReLU, stream);
      if (rc) return rc;
      flops += r->w[li].flops_per_pixel() * N * Ho * Ho;
      if (down) {
        res = Slice{r->buf[idn], c2.cout, 0};
        rc = conv_forward(r->w[li + 2], in, N, H, H, res, nullptr, Slice{}, kActNone, stream);
        if (rc) return rc;
        flops += 2.0 * c1.cout * c1.cin * N * Ho * Ho;  // algorithmic: a 1x1
      }
      rc = conv_forward(r->w[li + 1], mid, N, Ho, Ho, out, nullptr, res, kActResReLU, stream);
      if (rc) return rc;
      flops += r->w[li + 1].flops_per_pixel() * N * Ho * Ho;
      x = y;
      H = Ho;
      li += down ? 3 : 2;
    }
  hipLaunchKernelGGL(k_places_head, dim3((unsigned)N), dim3(256), 0, stream, r->buf[x], H * H, r->fc_w, r->fc_b, r->nc, top_k,
                     r->logits, r->probs, r->idx);
  EIOKU_LAUNCH_CHECK();
  flops += 2.0 * 512 * r->nc * N;
  r->flops_last = flops;
  return EIOKU_OK;
}

}  // namespace

extern "C" {

int eioku_resnet18_create(int num_classes, eioku_resnet_t** out) {
  EIOKU_REQUIRE_INIT();
  EIOKU_REQUIRE(out && num_classes >= 1 && num_classes <= 512, "bad argument");
  auto* r = new eioku_resnet();
  r->nc = num_classes;
  r->layers = resnet18_layers();
  r->w.resize(r->layers.size());
  r->set.assign(r->layers.size(), false);
  *out = r;
  return EIOKU_OK;
}

void eioku_resnet18_destroy(eioku_resnet_t* r) {
  if (!r) return;
  (void)hipDeviceSynchronize();
  for (auto& w : r->w) conv_weights_destroy(&w);
  for (void* p : {(void*)r->stem_w, (void*)r->stem_b, (void*)r->fc_w, (void*)r->fc_b, (void*)r->in224, (void*)r->a112,
                  (void*)r->buf[0], (void*)r->buf[1], (void*)r->buf[2], (void*)r->buf[3], (void*)r->tmp, (void*)r->src,
                  (void*)r->tables, (void*)r->logits, (void*)r->probs, (void*)r->idx})
    if (p) (void)hipFree(p);
  delete r;
}

int eioku_resnet18_num_convs(const eioku_resnet_t* r) { return r ? (int)r->layers.size() : 0; }

int eioku_resnet18_conv_info(const eioku_resnet_t* r, int idx, char* name, size_t cap, int* cout, int* cin, int* ksize,
                             int* stride) {
  EIOKU_REQUIRE(r && idx >= 0 && idx < (int)r->layers.size(), "bad convolution index %d", idx);
  const Layer& l = r->layers[idx];
  if (name && cap) snprintf(name, cap, "%s", l.name.c_str());
  if (cout) *cout = l.cout;
  if (cin) *cin = l.cin;
  if (ksize) *ksize = l.k;
  if (stride) *stride = l.stride;
  return EIOKU_OK;
}

// weight: HOST fp32 [cout][cin][k][k] with BatchNorm already folded in, bias HOST fp32 [cout]
int eioku_resnet18_set_conv(eioku_resnet_t* r, int idx, const float* w, const float* b) {
  EIOKU_REQUIRE_INIT();
  EIOKU_REQUIRE(r && idx >= 0 && idx < (int)r->layers.size() && w && b, "bad argument");
  const Layer& l = r->layers[idx];
  if (idx == 0) {
    // stem: A operand of (k-step s, fragment f, lane (row, u)) = cout f*16 + row, k = 8u .. 8u + 7 = taps 8s + 2u, + 1 x
    // 4 channels (channel 3 and taps >= 49 are zero)
    std::vector<_Float16> pw((size_t)kSteps * 4 * 64 * 8, (_Float16)0.f);
    for (int s = 0; s < kSteps; ++s)
      for (int f = 0; f < 4; ++f)
        for (int lane = 0; lane < 64; ++lane) {
          const int row = lane & 15, u = lane >> 4, co = f * 16 + row;
          for (int e = 0; e < 8; ++e) {
            const int tap = 8 * s + 2 * u + (e >> 2), ch = e & 3;
            if (tap < 49 && ch < 3) pw[(((size_t)s * 4 + f) * 64 + lane) * 8 + e] = (_Float16)w[((size_t)co * 3 + ch) * 49 + tap];
          }
        }
    if (!r->stem_w) EIOKU_HIP_CHECK(hipMalloc((void**)&r->stem_w, pw.size() * 2));
    if (!r->stem_b) EIOKU_HIP_CHECK(hipMalloc((void**)&r->stem_b, 64 * 4));
    EIOKU_HIP_CHECK(hipMemcpy(r->stem_w, pw.data(), pw.size() * 2, hipMemcpyHostToDevice));
    EIOKU_HIP_CHECK(hipMemcpy(r->stem_b, b, 64 * 4, hipMemcpyHostToDevice));
  } else {
    conv_weights_destroy(&r->w[idx]);
    int rc;
    if (l.k == 1) {  // stride-2 1x1 -> 3x3 / s2 with the weights on the centre tap
      std::vector<float> w3((size_t)l.cout * l.cin * 9, 0.f);
      for (size_t i = 0; i < (size_t)l.cout * l.cin; ++i) w3[i * 9 + 4] = w[i];
      rc = conv_weights_create(&r->w[idx], l.cout, l.cin, 3, l.stride, w3.data(), b);
    } else {
      rc = conv_weights_create(&r->w[idx], l.cout, l.cin, l.k, l.stride, w, b);
    }
    if (rc) return rc;
  }
  r->set[idx] = true;
  return EIOKU_OK;
}

int eioku_resnet18_set_fc(eioku_resnet_t* r, const float* w, const float* b) {
  EIOKU_REQUIRE_INIT();
  EIOKU_REQUIRE(r && w && b, "bad argument");
  std::vector<_Float16> hw((size_t)r->nc * 512);
  for (size_t i = 0; i < hw.size(); ++i) hw[i] = (_Float16)w[i];
  if (!r->fc_w) EIOKU_HIP_CHECK(hipMalloc((void**)&r->fc_w, hw.size() * 2));
  if (!r->fc_b) EIOKU_HIP_CHECK(hipMalloc((void**)&r->fc_b, (size_t)r->nc * 4));
  EIOKU_HIP_CHECK(hipMemcpy(r->fc_w, hw.data(), hw.size() * 2, hipMemcpyHostToDevice));
  EIOKU_HIP_CHECK(hipMemcpy(r->fc_b, b, (size_t)r->nc * 4, hipMemcpyHostToDevice));
  r->fc_set = true;
  return EIOKU_OK;
}

// Resize + normalise only (parity helper and first stage of eioku_resnet18_classify): n BGR u8 frames (h x w, host or
// device) -> out_f16 [n][224][224][4] fp16 (R, G, B, 0; device, optional) and / or out_rgb_u8 [n][224][224][3] (device,
// optional: Pillow's resized image).  Tables from the host (Pillow precompute_coeffs, eioku_amd/places.py):
// xbounds [224][2], xk [224][kx], ybounds [224][2], yk [224][ky], all int32 HOST.
int eioku_places_preprocess(eioku_resnet_t* r, const uint8_t* bgr, int n, int h, int w, const int32_t* xbounds,
                            const int32_t* xk, int kx, const int32_t* ybounds, const int32_t* yk, int ky, void* out_f16,
                            uint8_t* out_rgb_u8, int mem, void* stream_) {
  EIOKU_REQUIRE_INIT();
  EIOKU_REQUIRE(r && n >= 0 && h > 0 && w > 0 && kx > 0 && ky > 0 && xbounds && xk && ybounds && yk, "bad argument");
  EIOKU_REQUIRE(mem == EIOKU_MEM_HOST || mem == EIOKU_MEM_DEVICE, "bad mem flag %d", mem);
  if (n == 0) return EIOKU_OK;
  EIOKU_REQUIRE(bgr, "NULL frames");
  hipStream_t stream = (hipStream_t)stream_;
  const size_t tb = (size_t)kIn * (2 + kx + 2 + ky) * 4;
  if (r->tables_cap < tb) {
    if (r->tables) (void)hipFree(r->tables);
    EIOKU_HIP_CHECK(hipMalloc((void**)&r->tables, tb));
    r->tables_cap = tb;
  }
  int* d_xb = r->tables;
  int* d_xk = d_xb + kIn * 2;
  int* d_yb = d_xk + kIn * kx;
  int* d_yk = d_yb + kIn * 2;
  EIOKU_HIP_CHECK(hipMemcpyAsync(d_xb, xbounds, kIn * 2 * 4, hipMemcpyHostToDevice, stream));
  EIOKU_HIP_CHECK(hipMemcpyAsync(d_xk, xk, (size_t)kIn * kx * 4, hipMemcpyHostToDevice, stream));
  EIOKU_HIP_CHECK(hipMemcpyAsync(d_yb, ybounds, kIn * 2 * 4, hipMemcpyHostToDevice, stream));
  EIOKU_HIP_CHECK(hipMemcpyAsync(d_yk, yk, (size_t)kIn * ky * 4, hipMemcpyHostToDevice, stream));
  const uint8_t* d_src = bgr;
  const size_t sb = (size_t)n * h * w * 3;
  if (mem == EIOKU_MEM_HOST) {
    if (r->src_cap < sb) {
      if (r->src) (void)hipFree(r->src);
      EIOKU_HIP_CHECK(hipMalloc((void**)&r->src, sb));
      r->src_cap = sb;
    }
    EIOKU_HIP_CHECK(hipMemcpyAsync(r->src, bgr, sb, hipMemcpyHostToDevice, stream));
    d_src = r->src;
  }
  const size_t tmpb = (size_t)n * h * kIn * 3;
  if (r->tmp_cap < tmpb) {
    if (r->tmp) (void)hipFree(r->tmp);
    EIOKU_HIP_CHECK(hipMalloc((void**)&r->tmp, tmpb));
    r->tmp_cap = tmpb;
  }
  const long long w1 = (long long)n * h * kIn, w2 = (long long)n * kIn * kIn;
  hipLaunchKernelGGL(k_pil_resize_h, dim3((unsigned)((w1 + 255) / 256)), dim3(256), 0, stream, d_src, n, h, w, d_xb, d_xk, kx, r->tmp);
  hipLaunchKernelGGL(k_pil_resize_v, dim3((unsigned)((w2 + 255) / 256)), dim3(256), 0, stream, r->tmp, n, h, d_yb, d_yk, ky,
                     (__half*)out_f16, out_rgb_u8);
  EIOKU_LAUNCH_CHECK();
  if (mem == EIOKU_MEM_HOST) EIOKU_HIP_CHECK(hipStreamSynchronize(stream));  // the staged host buffer may be reused by the caller
  return EIOKU_OK;
}

// The raw network: in [n][224][224][4] fp16 (device) -> logits_out [n][nc] fp32 (device).  Asynchronous.
int eioku_resnet18_forward(eioku_resnet_t* r, const void* in_nhwc4_f16, int n, float* logits_out, void* stream_) {
  EIOKU_REQUIRE_INIT();
  EIOKU_REQUIRE(r && n >= 0, "bad argument");
  if (n == 0) return EIOKU_OK;
  EIOKU_REQUIRE(in_nhwc4_f16 && logits_out, "NULL buffer");
  hipStream_t stream = (hipStream_t)stream_;
  int rc = ensure_batch(r, n);
  if (rc) return rc;
  EIOKU_HIP_CHECK(hipMemcpyAsync(r->in224, in_nhwc4_f16, (size_t)n * kIn * kIn * 4 * 2, hipMemcpyDeviceToDevice, stream));
  rc = run_network(r, n, 1, stream);
  if (rc) return rc;
  EIOKU_HIP_CHECK(hipMemcpyAsync(logits_out, r->logits, (size_t)n * r->nc * 4, hipMemcpyDeviceToDevice, stream));
  return EIOKU_OK;
}

// classify_places' per-frame arithmetic on n BGR frames (host or device): resize / normalise -> network -> softmax ->
// descending sort -> prob_out [n][top_k] fp32, class_out [n][top_k] int32 (HOST when mem == EIOKU_MEM_HOST, else
// device); logits_out optional ([n][nc], same side).  Host outputs synchronise the stream.
int eioku_resnet18_classify(eioku_resnet_t* r, const uint8_t* bgr, int n, int h, int w, const int32_t* xbounds,
                            const int32_t* xk, int kx, const int32_t* ybounds, const int32_t* yk, int ky, int top_k,
                            float* prob_out, int32_t* class_out, float* logits_out, int mem, void* stream_) {
  EIOKU_REQUIRE_INIT();
  EIOKU_REQUIRE(r && n >= 0 && top_k >= 1 && top_k <= r->nc, "bad argument (top_k %d of %d classes)", top_k, r ? r->nc : 0);
  if (n == 0) return EIOKU_OK;
  EIOKU_REQUIRE(prob_out && class_out, "NULL output");
  hipStream_t stream = (hipStream_t)stream_;
  int rc = ensure_batch(r, n);
  if (rc) return rc;
  rc = eioku_places_preprocess(r, bgr, n, h, w, xbounds, xk, kx, ybounds, yk, ky, r->in224, nullptr, mem, stream_);
  if (rc) return rc;
  rc = run_network(r, n, top_k, stream);
  if (rc) return rc;
  const hipMemcpyKind kind = mem == EIOKU_MEM_HOST ? hipMemcpyDeviceToHost : hipMemcpyDeviceToDevice;
  EIOKU_HIP_CHECK(hipMemcpyAsync(prob_out, r->probs, (size_t)n * top_k * 4, kind, stream));
  EIOKU_HIP_CHECK(hipMemcpyAsync(class_out, r->idx, (size_t)n * top_k * 4, kind, stream));
  if (logits_out) EIOKU_HIP_CHECK(hipMemcpyAsync(logits_out, r->logits, (size_t)n * r->nc * 4, kind, stream));
  if (mem == EIOKU_MEM_HOST) EIOKU_HIP_CHECK(hipStreamSynchronize(stream));
  return EIOKU_OK;
}

int eioku_resnet18_last_flops(const eioku_resnet_t* r, double* flops) {
  EIOKU_REQUIRE(r && flops, "NULL argument");
  *flops = r->flops_last;
  return EIOKU_OK;
}

}  // extern "C"
