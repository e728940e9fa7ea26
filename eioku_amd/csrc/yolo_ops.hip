// K3 letterbox, K5 maxpool/upsample glue, K6 DFL decode, K7 wavefront NMS.
//
// These are the exactness-critical, non-GEMM pieces of the detection stage: integer fixed-point
// resize (OpenCV INTER_LINEAR semantics), fp32 box decode in the Ultralytics operation order, and
// class-aware greedy NMS with the torchvision IoU expression, compiled with -ffp-contract=off so no
// multiply-add is fused that the CPU reference does not fuse.
#include "yolo_ops.h"

namespace eioku {

namespace {

// ---------------------------------------------------------------------------------------------
// K3 letterbox + normalise
// ---------------------------------------------------------------------------------------------
struct LbArgs {
  const uint8_t* bgr;
  __half* out;
  int n, src_h, src_w, new_h, new_w, top, left, out_h, out_w, mode;
  const int32_t* xofs;
  const int32_t* yofs;
  const int16_t* xalpha;
  const int16_t* ybeta;
};

__global__ __launch_bounds__(256) void k_letterbox(LbArgs a) {
  const long long total = (long long)a.n * a.out_h * a.out_w;
  long long i = blockIdx.x * 256ll + threadIdx.x;
  if (i >= total) return;
  const int ox = (int)(i % a.out_w);
  const long long t = i / a.out_w;
  const int oy = (int)(t % a.out_h);
  const int n = (int)(t / a.out_h);
  const int y = oy - a.top, x = ox - a.left;
  int v[3] = {114, 114, 114};
  if (y >= 0 && y < a.new_h && x >= 0 && x < a.new_w) {
    const uint8_t* img = a.bgr + (size_t)n * a.src_h * a.src_w * 3;
    if (a.mode == 0) {
      const uint8_t* p = img + ((size_t)y * a.src_w + x) * 3;
      v[0] = p[0]; v[1] = p[1]; v[2] = p[2];
    } else if (a.mode == 2) {  // INTER_LINEAR at exactly 1/2 scale takes OpenCV's 2x2 area path
      const uint8_t* p0 = img + ((size_t)(2 * y) * a.src_w + 2 * x) * 3;
      const uint8_t* p1 = p0 + (size_t)a.src_w * 3;
#pragma unroll
      for (int c = 0; c < 3; ++c) v[c] = (p0[c] + p0[c + 3] + p1[c] + p1[c + 3] + 2) >> 2;
    } else {
      const int sy = a.yofs[y];
      const int r0 = min(max(sy, 0), a.src_h - 1), r1 = min(max(sy + 1, 0), a.src_h - 1);
      const int b0 = a.ybeta[2 * y], b1 = a.ybeta[2 * y + 1];
      const int sx = a.xofs[x];
      const int sx1 = min(sx + 1, a.src_w - 1);
      const int a0 = a.xalpha[2 * x], a1 = a.xalpha[2 * x + 1];
      const uint8_t* q0 = img + (size_t)r0 * a.src_w * 3;
      const uint8_t* q1 = img + (size_t)r1 * a.src_w * 3;
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        const int h0 = q0[sx * 3 + c] * a0 + q0[sx1 * 3 + c] * a1;  // HResizeLinear, 11-bit weights
        const int h1 = q1[sx * 3 + c] * a0 + q1[sx1 * 3 + c] * a1;
        v[c] = (((b0 * (h0 >> 4)) >> 16) + ((b1 * (h1 >> 4)) >> 16) + 2) >> 2;  // VResizeLinear 8u
      }
    }
  }
  __half o[8];
  o[0] = __float2half_rn((float)v[2] / 255.0f);  // BGR -> RGB
  o[1] = __float2half_rn((float)v[1] / 255.0f);
  o[2] = __float2half_rn((float)v[0] / 255.0f);
#pragma unroll
  for (int c = 3; c < 8; ++c) o[c] = __float2half_rn(0.f);
  *reinterpret_cast<uint4*>(a.out + (size_t)i * 8) = *reinterpret_cast<uint4*>(o);
}

// ---------------------------------------------------------------------------------------------
// K5 glue
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_sppf_pools(const __half* in, int in_cs, __half* out, int out_cs, int out_step,
                                                    int H, int W) {
  typedef _Float16 half8 __attribute__((ext_vector_type(8)));
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_pool[];
  half8* a = reinterpret_cast<half8*>(smem_pool);
  half8* b = a + H * W;
  half8* t = b + H * W;  // row maxima of the current stage
  const int u = blockIdx.x, n = blockIdx.y, tid = threadIdx.x, npx = H * W;
  const size_t base = (size_t)n * npx;
  for (int p = tid; p < npx; p += 256) a[p] = *reinterpret_cast<const half8*>(in + (base + p) * in_cs + u * 8);
  __syncthreads();
  // a 5 x 5 maximum is the column maximum of the row maxima: 5 + 5 LDS reads per pixel instead of 25, and with the
  // window clamped into the map (a repeated element does not change a maximum) both passes are branch-free and unrolled
  // (r3: the nested run-time loops of the 2-D window made this 30 us launch the longest non-conv step of the backbone)
  for (int stage = 0; stage < 3; ++stage) {
    const half8* src = (stage & 1) ? b : a;
    half8* dst = (stage & 1) ? a : b;
    for (int p = tid; p < npx; p += 256) {
      const int y = p / W, x = p - y * W;
      half8 m = src[p];
#pragma unroll
      for (int d = 1; d <= 2; ++d) {
        m = __builtin_elementwise_max(m, src[y * W + (x - d < 0 ? 0 : x - d)]);
        m = __builtin_elementwise_max(m, src[y * W + (x + d >= W ? W - 1 : x + d)]);
      }
      t[p] = m;
    }
    __syncthreads();
    for (int p = tid; p < npx; p += 256) {
      const int y = p / W, x = p - y * W;
      half8 m = t[p];
#pragma unroll
      for (int d = 1; d <= 2; ++d) {
        m = __builtin_elementwise_max(m, t[(y - d < 0 ? 0 : y - d) * W + x]);
        m = __builtin_elementwise_max(m, t[(y + d >= H ? H - 1 : y + d) * W + x]);
      }
      if (stage < 2) dst[p] = m;
      *reinterpret_cast<half8*>(out + (base + p) * out_cs + stage * out_step + u * 8) = m;
    }
    __syncthreads();
  }
}

__global__ __launch_bounds__(256) void k_maxpool5(const __half* in, int in_cs, __half* out, int out_cs,
                                                  int N, int H, int W, int C8) {
  const long long total = (long long)N * H * W * C8;
  long long i = blockIdx.x * 256ll + threadIdx.x;
  if (i >= total) return;
  const int u = (int)(i % C8);
  long long t = i / C8;
  const int x = (int)(t % W);
  t /= W;
  const int y = (int)(t % H);
  const int n = (int)(t / H);
  typedef _Float16 half8 __attribute__((ext_vector_type(8)));
  const _Float16 ninf = -__builtin_inff16();
  half8 m = {ninf, ninf, ninf, ninf, ninf, ninf, ninf, ninf};
  for (int dy = -2; dy <= 2; ++dy) {
    const int yy = y + dy;
    if (yy < 0 || yy >= H) continue;
    for (int dx = -2; dx <= 2; ++dx) {
      const int xx = x + dx;
      if (xx < 0 || xx >= W) continue;
      const uint4 v = *reinterpret_cast<const uint4*>(in + (((size_t)n * H + yy) * W + xx) * in_cs + u * 8);
      m = __builtin_elementwise_max(m, *reinterpret_cast<const half8*>(&v));
    }
  }
  *reinterpret_cast<uint4*>(out + (((size_t)n * H + y) * W + x) * out_cs + u * 8) = *reinterpret_cast<uint4*>(&m);
}

__global__ __launch_bounds__(256) void k_upsample2x(const __half* in, int in_cs, __half* out, int out_cs,
                                                    int N, int H, int W, int C8) {
  const long long total = (long long)N * H * W * C8;
  long long i = blockIdx.x * 256ll + threadIdx.x;
  if (i >= total) return;
  const int u = (int)(i % C8);
  long long t = i / C8;
  const int x = (int)(t % W);
  t /= W;
  const int y = (int)(t % H);
  const int n = (int)(t / H);
  const uint4 v = *reinterpret_cast<const uint4*>(in + (((size_t)n * H + y) * W + x) * in_cs + u * 8);
  const int W2 = 2 * W, H2 = 2 * H;
#pragma unroll
  for (int dy = 0; dy < 2; ++dy)
#pragma unroll
    for (int dx = 0; dx < 2; ++dx)
      *reinterpret_cast<uint4*>(out + (((size_t)n * H2 + 2 * y + dy) * W2 + 2 * x + dx) * out_cs + u * 8) = v;
}

// ---------------------------------------------------------------------------------------------
// K6 decode: DFL(softmax 16 . arange) -> ltrb -> xywh*stride -> xyxy ; class max + conf filter
// ---------------------------------------------------------------------------------------------
struct DecArgs {
  const float* box[3];
  const float* cls[3];
  int H[3], W[3], A0[3];  // A0 = anchor index base of the level
  int N, nc, A;
  float conf;
  Cand* cands;      // dense [N][A]
  unsigned long long* keys;  // compact [N][A]
  int32_t* counts;
  const unsigned long long* cm[3];  // class-max words per level (k_decode_cm) or null
  // lazy box branch (detect()): the box branch's last 1x1 conv runs only for anchors that passed the threshold
  int32_t* lvl_list;    // [N][A]: level l's passing anchors (level-local index) at [n][A0[l] + k]
  int32_t* lvl_counts;  // [N][3]
  float* box_w[3];      // writable alias of box[] (sparse rows filled by k_box_gather)
  const __half* bin[3]; // input of the box conv per level, fp16 [N][H][W][bin_cs]
  const uint4* bwgt[3]; // packed 1x1 weights [nchunks][64][4 units]
  const float* bbias[3];
  int bin_cs, bnchunks;
  // deep lazy box branch: flat per-level pixel lists (global pixel index n*H*W + y*W + x) of the passing anchors
  // (flat1) and of their in-map 3x3 neighbourhoods (flat0, duplicates allowed), with device-side counts
  int32_t* flat1[3];
  int32_t* flat0[3];
  int32_t* fcnt;  // [6]: cnt1[3], cnt0[3]; nullptr = not requested
};

__device__ __forceinline__ float dfl_side(const float* __restrict__ l) {
  float m = l[0];
#pragma unroll
  for (int i = 1; i < 16; ++i) m = fmaxf(m, l[i]);
  float e[16];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    e[i] = expf(l[i] - m);
    s = s + e[i];
  }
  float d = 0.f;
#pragma unroll
  for (int i = 0; i < 16; ++i) d = d + (e[i] / s) * (float)i;
  return d;
}

// 16 lanes per anchor: the class row is read with coalesced float4 loads and reduced with shuffles
// (first maximum wins, as torch.max does); lane 0 of the group decodes the box of the few anchors
// that pass the confidence filter, in the reference's sequential fp32 order.
__global__ __launch_bounds__(256) void k_decode(DecArgs a) {
  const long long total = (long long)a.N * a.A;
  const long long gid = blockIdx.x * 256ll + threadIdx.x;
  const long long i = gid >> 4;
  const int sub = (int)(gid & 15);
  if (i >= total) return;  // uniform per 16-lane group; shuffles below stay inside the group
  const int n = (int)(i / a.A);
  const int an = (int)(i % a.A);
  const int lvl = an >= a.A0[2] ? 2 : (an >= a.A0[1] ? 1 : 0);
  const int loc = an - a.A0[lvl];
  const int W = a.W[lvl], H = a.H[lvl];
  const size_t pix = (size_t)n * H * W + loc;
  const float* cl = a.cls[lvl] + pix * a.nc;
  float best = -INFINITY;
  int bj = 0x7FFFFFFF;
  if ((a.nc & 3) == 0) {
    for (int c = sub * 4; c < a.nc; c += 64) {
      const float4 v = *reinterpret_cast<const float4*>(cl + c);
      if (v.x > best) { best = v.x; bj = c; }
      if (v.y > best) { best = v.y; bj = c + 1; }
      if (v.z > best) { best = v.z; bj = c + 2; }
      if (v.w > best) { best = v.w; bj = c + 3; }
    }
  } else {
    for (int c = sub; c < a.nc; c += 16) {
      const float v = cl[c];
      if (v > best) { best = v; bj = c; }
    }
  }
#pragma unroll
  for (int off = 8; off > 0; off >>= 1) {
    const float ob = __shfl_xor(best, off, 16);
    const int oj = __shfl_xor(bj, off, 16);
    if (ob > best || (ob == best && oj < bj)) { best = ob; bj = oj; }
  }
  if (sub != 0) return;
  const float conf = 1.0f / (1.0f + expf(-best));
  if (!(conf > a.conf)) return;
  const float* bl = a.box[lvl] + pix * 64;
  float side[4];
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    float l[16];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const float4 v = *reinterpret_cast<const float4*>(bl + s * 16 + q * 4);
      l[q * 4] = v.x; l[q * 4 + 1] = v.y; l[q * 4 + 2] = v.z; l[q * 4 + 3] = v.w;
    }
    side[s] = dfl_side(l);
  }
  const float stride = (float)(8 << lvl);
  const float ax = (float)(loc % W) + 0.5f, ay = (float)(loc / W) + 0.5f;
  // dist2bbox(xywh=True): x1y1 = anchor - lt ; x2y2 = anchor + rb ; c = (x1y1+x2y2)/2 ; wh = x2y2-x1y1
  const float bx1 = ax - side[0], by1 = ay - side[1], bx2 = ax + side[2], by2 = ay + side[3];
  const float cx = ((bx1 + bx2) / 2.0f) * stride, cy = ((by1 + by2) / 2.0f) * stride;
  const float w = (bx2 - bx1) * stride, h = (by2 - by1) * stride;
  // xywh2xyxy inside non_max_suppression
  const float hw = w / 2.0f, hh = h / 2.0f;
  Cand c;
  c.x1 = cx - hw; c.y1 = cy - hh; c.x2 = cx + hw; c.y2 = cy + hh;
  c.conf = conf; c.cls = bj; c.anchor = an; c.pad = 0;
  a.cands[(size_t)n * a.A + an] = c;
  const int pos = atomicAdd(&a.counts[n], 1);
  a.keys[(size_t)n * a.A + pos] =
      ((unsigned long long)__float_as_uint(conf) << 32) | (unsigned long long)(0xFFFFFFFFu - (unsigned)an);
}

// Same decode, but the class branch arrives as per-anchor {max logit, argmax} words written by the class conv's
// own epilogue (conv.hip, clsmax): one thread per anchor, 8 bytes instead of the 4*nc-byte logit row.
__global__ __launch_bounds__(256) void k_decode_cm(DecArgs a) {
  const long long total = (long long)a.N * a.A;
  const long long i = blockIdx.x * 256ll + threadIdx.x;
  if (i >= total) return;
  const int n = (int)(i / a.A);
  const int an = (int)(i % a.A);
  const int lvl = an >= a.A0[2] ? 2 : (an >= a.A0[1] ? 1 : 0);
  const int loc = an - a.A0[lvl];
  const int W = a.W[lvl], H = a.H[lvl];
  const size_t pix = (size_t)n * H * W + loc;
  const unsigned long long word = a.cm[lvl][pix];
  const float best = __uint_as_float((unsigned)(word & 0xFFFFFFFFull));
  const int bj = (int)(word >> 32);
  const float conf = 1.0f / (1.0f + expf(-best));
  if (!(conf > a.conf)) return;
  const float* bl = a.box[lvl] + pix * 64;
  float side[4];
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    float l[16];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const float4 v = *reinterpret_cast<const float4*>(bl + s * 16 + q * 4);
      l[q * 4] = v.x; l[q * 4 + 1] = v.y; l[q * 4 + 2] = v.z; l[q * 4 + 3] = v.w;
    }
    side[s] = dfl_side(l);
  }
  const float stride = (float)(8 << lvl);
  const float ax = (float)(loc % W) + 0.5f, ay = (float)(loc / W) + 0.5f;
  // dist2bbox(xywh=True): x1y1 = anchor - lt ; x2y2 = anchor + rb ; c = (x1y1+x2y2)/2 ; wh = x2y2-x1y1
  const float bx1 = ax - side[0], by1 = ay - side[1], bx2 = ax + side[2], by2 = ay + side[3];
  const float cx = ((bx1 + bx2) / 2.0f) * stride, cy = ((by1 + by2) / 2.0f) * stride;
  const float w = (bx2 - bx1) * stride, h = (by2 - by1) * stride;
  // xywh2xyxy inside non_max_suppression
  const float hw = w / 2.0f, hh = h / 2.0f;
  Cand c;
  c.x1 = cx - hw; c.y1 = cy - hh; c.x2 = cx + hw; c.y2 = cy + hh;
  c.conf = conf; c.cls = bj; c.anchor = an; c.pad = 0;
  a.cands[(size_t)n * a.A + an] = c;
  const int pos = atomicAdd(&a.counts[n], 1);
  a.keys[(size_t)n * a.A + pos] =
      ((unsigned long long)__float_as_uint(conf) << 32) | (unsigned long long)(0xFFFFFFFFu - (unsigned)an);
}

// ---- lazy box branch -----------------------------------------------------------------------------------------
// phase 1: threshold on the class-max word, compact key per image (as before) + per-(image, level) anchor list
__global__ __launch_bounds__(256) void k_decode_pass(DecArgs a) {
  const int n = blockIdx.y;  // (image, anchor) from the grid: the 64-bit div/mod of a flat index cost more than the rest
  const int an = blockIdx.x * 256 + threadIdx.x;
  const bool valid = an < a.A;
  const int lvl = an >= a.A0[2] ? 2 : (an >= a.A0[1] ? 1 : 0);
  const int loc = an - a.A0[lvl];
  const int W = a.W[lvl], H = a.H[lvl];
  const size_t pix = (size_t)n * H * W + (valid ? loc : 0);
  float conf = 0.f;
  bool pass = false;
  if (valid) {
    const unsigned long long word = a.cm[lvl][pix];
    const float best = __uint_as_float((unsigned)(word & 0xFFFFFFFFull));
    conf = 1.0f / (1.0f + expf(-best));
    pass = conf > a.conf;
  }
  if (pass) {
    const int pos = atomicAdd(&a.counts[n], 1);
    a.keys[(size_t)n * a.A + pos] =
        ((unsigned long long)__float_as_uint(conf) << 32) | (unsigned long long)(0xFFFFFFFFu - (unsigned)an);
    const int k = atomicAdd(&a.lvl_counts[n * 3 + lvl], 1);
    a.lvl_list[(size_t)n * a.A + a.A0[lvl] + k] = loc;
  }
  if (!a.fcnt) return;
  // flat lists for the deep lazy path: ONE atomic per (wave, level, list) reserves the wave's range -- three global
  // counters taking one atomic per passing anchor serialised the whole kernel (13 -> 93 us)
  const int lane = threadIdx.x & 63;
  const int y = loc / W, x = loc - y * W;
  int nb[9], c = 0;
  if (pass) {
#pragma unroll
    for (int dy = -1; dy <= 1; ++dy)
#pragma unroll
      for (int dx = -1; dx <= 1; ++dx)
        if ((unsigned)(y + dy) < (unsigned)H && (unsigned)(x + dx) < (unsigned)W) nb[c++] = (int)pix + dy * W + dx;
  }
  int incl = c;  // inclusive prefix sum of c over the wave
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const int v = __shfl_up(incl, off, 64);
    if (lane >= off) incl += v;
  }
  for (int l = 0; l < 3; ++l) {
    const unsigned long long m = __ballot(pass && lvl == l);
    if (m == 0) continue;
    const int leader = __ffsll((long long)m) - 1;
    const int last = 63 - __clzll((long long)m);
    // the level's lanes are contiguous in a wave (anchors are ordered by level): their neighbour counts are a
    // contiguous slice of the prefix sums
    const int before = leader > 0 ? __shfl(incl, leader - 1, 64) : 0;
    const int total0 = __shfl(incl, last, 64) - before;
    int base1 = 0, base0 = 0;
    if (lane == leader) {
      base1 = atomicAdd(&a.fcnt[l], __popcll(m));
      base0 = atomicAdd(&a.fcnt[3 + l], total0);
    }
    base1 = __shfl(base1, leader, 64);
    base0 = __shfl(base0, leader, 64);
    if (pass && lvl == l) {
      a.flat1[l][base1 + __popcll(m & ((1ull << lane) - 1ull))] = (int)pix;
      const int o = base0 + (incl - c) - before;
      for (int i = 0; i < c; ++i) a.flat0[l][o + i] = nb[i];
    }
  }
}

// phase 2: the box branch's last 1x1 conv (cin -> 64, fp32, no activation) for the listed anchors only: a wave
// gathers 32 anchors' input rows as its B fragments; same MFMA, same k order as the dense k_conv1x1, so the 64
// logits written (sparsely) into the box map are bit-identical to what the dense conv would have stored there.
__global__ __launch_bounds__(256) void k_box_gather(DecArgs a) {
  typedef _Float16 half8 __attribute__((ext_vector_type(8)));
  typedef float float4v __attribute__((ext_vector_type(4)));
  const int n = blockIdx.y / 3, lvl = blockIdx.y % 3;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int cnt = a.lvl_counts[n * 3 + lvl];
  const int g0 = (blockIdx.x * 4 + wave) * 32;
  if (g0 >= cnt) return;
  const int r = lane & 15, u = lane >> 4;
  const int32_t* list = a.lvl_list + (size_t)n * a.A + a.A0[lvl];
  const size_t img = (size_t)n * a.H[lvl] * a.W[lvl];
  int loc[2];
  const __half* row[2];
#pragma unroll
  for (int m = 0; m < 2; ++m) {
    const int k = g0 + m * 16 + r;
    loc[m] = list[k < cnt ? k : cnt - 1];
    row[m] = a.bin[lvl] + (img + loc[m]) * a.bin_cs + u * 8;
  }
  float4v acc[2][4];
#pragma unroll
  for (int m = 0; m < 2; ++m)
#pragma unroll
    for (int f = 0; f < 4; ++f) acc[m][f] = float4v{0.f, 0.f, 0.f, 0.f};
  const uint4* wl = a.bwgt[lvl] + r * 4 + u;  // packed [kc][64 rows][4 units], unswizzled in global memory
  // bias and every chunk's input fragments are requested up front (r3): per chunk and per store they were dependent round
  // trips - and with a load pending, each epilogue store waited for the one before it (vmcnt completes in order)
  float4 biasr[4];
#pragma unroll
  for (int f = 0; f < 4; ++f) biasr[f] = *reinterpret_cast<const float4*>(a.bbias[lvl] + f * 16 + u * 4);
  constexpr int kMaxKc = 4;
  uint4 bfv[kMaxKc][2];
#pragma unroll
  for (int kc = 0; kc < kMaxKc; ++kc)
#pragma unroll
    for (int m = 0; m < 2; ++m) bfv[kc][m] = *reinterpret_cast<const uint4*>(row[m] + (kc < a.bnchunks ? kc : 0) * 32);
#pragma unroll
  for (int kc = 0; kc < kMaxKc; ++kc) {
    if (kc >= a.bnchunks) break;
    half8 bf[2];
#pragma unroll
    for (int m = 0; m < 2; ++m) bf[m] = *reinterpret_cast<const half8*>(&bfv[kc][m]);
#pragma unroll
    for (int f = 0; f < 4; ++f) {
      const uint4 w = wl[(kc * 64 + f * 16) * 4];
      const half8 af = *reinterpret_cast<const half8*>(&w);
#pragma unroll
      for (int m = 0; m < 2; ++m) acc[m][f] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af, bf[m], acc[m][f], 0, 0, 0);
    }
  }
#pragma unroll
  for (int m = 0; m < 2; ++m) {
    if (g0 + m * 16 + r >= cnt) continue;
    float* o = a.box_w[lvl] + (img + loc[m]) * 64;
#pragma unroll
    for (int f = 0; f < 4; ++f) {
      const float4 b = biasr[f];
      *reinterpret_cast<float4v*>(o + f * 16 + u * 4) = acc[m][f] + float4v{b.x, b.y, b.z, b.w};
    }
  }
}

// phase 3: DFL decode of the passing anchors (one thread per compact key)
__global__ __launch_bounds__(256) void k_decode_boxes(DecArgs a) {
  const int n = blockIdx.y;
  const int pos = blockIdx.x * 256 + threadIdx.x;
  if (pos >= a.counts[n]) return;
  const unsigned long long key = a.keys[(size_t)n * a.A + pos];
  const int an = (int)(0xFFFFFFFFu - (unsigned)(key & 0xFFFFFFFFull));
  const float conf = __uint_as_float((unsigned)(key >> 32));
  const int lvl = an >= a.A0[2] ? 2 : (an >= a.A0[1] ? 1 : 0);
  const int loc = an - a.A0[lvl];
  const int W = a.W[lvl], H = a.H[lvl];
  const size_t pix = (size_t)n * H * W + loc;
  const int bj = (int)(a.cm[lvl][pix] >> 32);
  const float* bl = a.box[lvl] + pix * 64;
  float side[4];
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    float l[16];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const float4 v = *reinterpret_cast<const float4*>(bl + s * 16 + q * 4);
      l[q * 4] = v.x; l[q * 4 + 1] = v.y; l[q * 4 + 2] = v.z; l[q * 4 + 3] = v.w;
    }
    side[s] = dfl_side(l);
  }
  const float stride = (float)(8 << lvl);
  const float ax = (float)(loc % W) + 0.5f, ay = (float)(loc / W) + 0.5f;
  // dist2bbox(xywh=True): x1y1 = anchor - lt ; x2y2 = anchor + rb ; c = (x1y1+x2y2)/2 ; wh = x2y2-x1y1
  const float bx1 = ax - side[0], by1 = ay - side[1], bx2 = ax + side[2], by2 = ay + side[3];
  const float cx = ((bx1 + bx2) / 2.0f) * stride, cy = ((by1 + by2) / 2.0f) * stride;
  const float w = (bx2 - bx1) * stride, h = (by2 - by1) * stride;
  // xywh2xyxy inside non_max_suppression
  const float hw = w / 2.0f, hh = h / 2.0f;
  Cand c;
  c.x1 = cx - hw; c.y1 = cy - hh; c.x2 = cx + hw; c.y2 = cy + hh;
  c.conf = conf; c.cls = bj; c.anchor = an; c.pad = 0;
  a.cands[(size_t)n * a.A + an] = c;
}

// ---------------------------------------------------------------------------------------------
// K7 NMS: one workgroup per image.  LDS bitonic sort of (conf, anchor) keys, then wave 0 runs the
// greedy pass 64 candidates at a time: every lane tests its box against the kept list, the chunk is
// resolved in score order with ballots.
// ---------------------------------------------------------------------------------------------
struct NmsArgs {
  const Cand* cands;
  const unsigned long long* keys;
  const int32_t* counts;
  int A, max_det;
  float iou, max_wh;
  ScaleParams sp;
  Det* dets;
  int32_t* det_counts;
};

__device__ __forceinline__ bool iou_gt(float ax1, float ay1, float ax2, float ay2, float aarea, float bx1,
                                       float by1, float bx2, float by2, float barea, float thr) {
  const float xx1 = fmaxf(ax1, bx1), yy1 = fmaxf(ay1, by1);
  const float xx2 = fminf(ax2, bx2), yy2 = fminf(ay2, by2);
  const float w = fmaxf(0.f, xx2 - xx1), h = fmaxf(0.f, yy2 - yy1);
  const float inter = w * h;
  const float ovr = inter / (aarea + barea - inter);
  return ovr > thr;
}

__global__ __launch_bounds__(256) void k_nms(NmsArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned long long* key = reinterpret_cast<unsigned long long*>(smem);
  const int n = blockIdx.x, tid = threadIdx.x;
  const int C = min(a.counts[n], a.A);
  int P = 1;
  while (P < C) P <<= 1;
  for (int i = tid; i < P; i += 256) key[i] = i < C ? a.keys[(size_t)n * a.A + i] : 0ull;
  __syncthreads();
  // descending bitonic sort
  for (int k = 2; k <= P; k <<= 1) {
    for (int j = k >> 1; j > 0; j >>= 1) {
      for (int i = tid; i < P; i += 256) {
        const int l = i ^ j;
        if (l > i) {
          const unsigned long long x = key[i], y = key[l];
          const bool desc = (i & k) == 0;
          if (desc ? (x < y) : (x > y)) {
            key[i] = y;
            key[l] = x;
          }
        }
      }
      __syncthreads();
    }
  }
  if (tid >= 64) return;
  // kept list lives after the keys (P*8 bytes, 16-aligned): 5 floats per box
  float* kb = reinterpret_cast<float*>(smem + (size_t)(P > 0 ? P : 1) * 8);
  int* kanchor = reinterpret_cast<int*>(kb + 5 * a.max_det);
  const int lane = tid;
  const Cand* cd = a.cands + (size_t)n * a.A;
  int kept = 0;
  for (int base = 0; base < C && kept < a.max_det; base += 64) {
    const int i = base + lane;
    const bool valid = i < C;
    float x1 = 0, y1 = 0, x2 = 0, y2 = 0, area = 0;
    int an = 0;
    if (valid) {
      an = (int)(0xFFFFFFFFu - (unsigned)(key[i] & 0xFFFFFFFFull));
      const Cand c = cd[an];
      const float off = (float)c.cls * a.max_wh;  // class-aware: boxes + cls * max_wh
      x1 = c.x1 + off; y1 = c.y1 + off; x2 = c.x2 + off; y2 = c.y2 + off;
      area = (x2 - x1) * (y2 - y1);
    }
    bool alive = valid;
    for (int k = 0; k < kept; ++k) {
      const float* q = kb + 5 * k;
      if (alive && iou_gt(q[0], q[1], q[2], q[3], q[4], x1, y1, x2, y2, area, a.iou)) alive = false;
    }
    unsigned long long mask = __ballot(alive);
    while (mask != 0 && kept < a.max_det) {
      const int l = __ffsll((long long)mask) - 1;  // best surviving score in the chunk
      const float lx1 = __shfl(x1, l, 64), ly1 = __shfl(y1, l, 64), lx2 = __shfl(x2, l, 64),
                  ly2 = __shfl(y2, l, 64), la = __shfl(area, l, 64);
      const int lan = __shfl(an, l, 64);
      if (lane == 0) {
        float* q = kb + 5 * kept;
        q[0] = lx1; q[1] = ly1; q[2] = lx2; q[3] = ly2; q[4] = la;
        kanchor[kept] = lan;
      }
      kept++;
      if (lane == l) alive = false;
      if (alive && lane > l && iou_gt(lx1, ly1, lx2, ly2, la, x1, y1, x2, y2, area, a.iou)) alive = false;
      mask = __ballot(alive) & ~((2ull << l) - 1ull);
    }
  }
  __builtin_amdgcn_s_waitcnt(0);  // lane 0's LDS writes are visible to the wave below
  // emit: original (un-offset) boxes -> scale_boxes -> clip
  for (int k = lane; k < kept; k += 64) {
    const Cand c = cd[kanchor[k]];
    Det d;
    d.x1 = fminf(fmaxf((c.x1 - a.sp.pad_x) / a.sp.gain, 0.f), a.sp.src_w);
    d.y1 = fminf(fmaxf((c.y1 - a.sp.pad_y) / a.sp.gain, 0.f), a.sp.src_h);
    d.x2 = fminf(fmaxf((c.x2 - a.sp.pad_x) / a.sp.gain, 0.f), a.sp.src_w);
    d.y2 = fminf(fmaxf((c.y2 - a.sp.pad_y) / a.sp.gain, 0.f), a.sp.src_h);
    d.conf = c.conf;
    d.cls = c.cls;
    d.anchor = c.anchor;
    d.pad = 0;
    a.dets[(size_t)n * a.max_det + k] = d;
  }
  if (lane == 0) a.det_counts[n] = kept;
}

inline unsigned blocks_for(long long total) { return (unsigned)((total + 255) / 256); }

}  // namespace

int letterbox_forward(const uint8_t* bgr, int n, const LetterboxPlan& p, __half* out, hipStream_t stream) {
  if (n == 0) return EIOKU_OK;
  LbArgs a{bgr, out, n, p.src_h, p.src_w, p.new_h, p.new_w, p.top, p.left, p.out_h, p.out_w, p.mode,
           p.xofs, p.yofs, p.xalpha, p.ybeta};
  hipLaunchKernelGGL(k_letterbox, dim3(blocks_for((long long)n * p.out_h * p.out_w)), dim3(256), 0, stream, a);
  EIOKU_LAUNCH_CHECK();
  return EIOKU_OK;
}

// SPPF's three chained 5x5 pools in one launch: a workgroup keeps one (frame, 8-channel unit) plane in LDS, so
// the chain y1 = pool(x), y2 = pool(y1), y3 = pool(y2) never re-reads HBM (three ~14 us launches at 20x20 -> one).
int sppf_pools_forward(Slice in, Slice out, int out_step, int N, int H, int W, int C, hipStream_t stream) {
  EIOKU_REQUIRE(C % 8 == 0 && in.cstride % 8 == 0 && in.coff % 8 == 0 && out.cstride % 8 == 0 && out.coff % 8 == 0 &&
                    out_step % 8 == 0, "maxpool slices must be 8-channel aligned");
  EIOKU_REQUIRE((size_t)H * W * 48 <= 144 * 1024, "plane %dx%d too large for the fused SPPF pools", H, W);
  if (N == 0) return EIOKU_OK;
  const size_t lds = (size_t)H * W * 48;  // two ping-pong planes + the row maxima, 16 B per pixel each
  static size_t attr = 0;
  if (lds > 64 * 1024 && lds > attr) {
    EIOKU_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_sppf_pools), hipFuncAttributeMaxDynamicSharedMemorySize,
                                        (int)lds));
    attr = lds;
  }
  hipLaunchKernelGGL(k_sppf_pools, dim3((unsigned)(C / 8), (unsigned)N), dim3(256), lds, stream, in.ptr + in.coff, in.cstride,
                     out.ptr + out.coff, out.cstride, out_step, H, W);
  EIOKU_LAUNCH_CHECK();
  return EIOKU_OK;
}

int maxpool5_forward(Slice in, Slice out, int N, int H, int W, int C, hipStream_t stream) {
  EIOKU_REQUIRE(C % 8 == 0 && in.cstride % 8 == 0 && in.coff % 8 == 0 && out.cstride % 8 == 0 && out.coff % 8 == 0,
                "maxpool slices must be 8-channel aligned");
  if (N == 0) return EIOKU_OK;
  hipLaunchKernelGGL(k_maxpool5, dim3(blocks_for((long long)N * H * W * (C / 8))), dim3(256), 0, stream,
                     in.ptr + in.coff, in.cstride, out.ptr + out.coff, out.cstride, N, H, W, C / 8);
  EIOKU_LAUNCH_CHECK();
  return EIOKU_OK;
}

int upsample2x_forward(Slice in, Slice out, int N, int H, int W, int C, hipStream_t stream) {
  EIOKU_REQUIRE(C % 8 == 0 && in.cstride % 8 == 0 && in.coff % 8 == 0 && out.cstride % 8 == 0 && out.coff % 8 == 0,
                "upsample slices must be 8-channel aligned");
  if (N == 0) return EIOKU_OK;
  hipLaunchKernelGGL(k_upsample2x, dim3(blocks_for((long long)N * H * W * (C / 8))), dim3(256), 0, stream,
                     in.ptr + in.coff, in.cstride, out.ptr + out.coff, out.cstride, N, H, W, C / 8);
  EIOKU_LAUNCH_CHECK();
  return EIOKU_OK;
}

// A 3x3 stride-1 convolution (Cin = 32*nchunks -> 64, bias + SiLU, fp16 out) evaluated only at the listed pixels:
// the box branch's two 3x3 layers are needed at the anchors that pass the threshold (layer 1) and at their 3x3
// neighbourhoods (layer 0), ~1-10 % of the map.  A wave takes 32 listed pixels as its two B fragments and gathers
// their taps from the dense input; weights come straight from the packed global tile (L2 resident).  Same MFMA,
// same (chunk outer, tap inner) accumulation order, same epilogue arithmetic as the dense kernels: the fp16 values
// written at those pixels are the ones the dense launch would have written.
struct GConv {
  const __half* in;
  const uint4* wgt;
  const float* bias;
  __half* out;
  const int32_t* list;
  const int32_t* count;
  int H, W, in_cs, out_cs, nchunks, rows_tile;
};
struct GArgs {
  GConv l[3];
};

__global__ __launch_bounds__(256) void k_conv3x3_gather(GArgs ga) {
  typedef _Float16 half8 __attribute__((ext_vector_type(8)));
  typedef _Float16 half4 __attribute__((ext_vector_type(4)));
  typedef float float4v __attribute__((ext_vector_type(4)));
  typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
  typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
  const GConv& g = ga.l[blockIdx.y];
  const int cnt = *g.count;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r = lane & 15, q = lane >> 4;
  const int HW = g.H * g.W;
  // the four bias vectors once, before the loop (r3): loaded inside the epilogue each store waited for its own bias load
  // and, vmcnt being in order, for the store before it - eight round trips per pixel group
  float4 biasr[4];
#pragma unroll
  for (int f = 0; f < 4; ++f) biasr[f] = *reinterpret_cast<const float4*>(g.bias + f * 16 + q * 4);
  for (int g0 = (blockIdx.x * 4 + wave) * 32; g0 < cnt; g0 += gridDim.x * 4 * 32) {
    int gp[2], py[2], px[2];
    const __half* base[2];
#pragma unroll
    for (int m = 0; m < 2; ++m) {
      const int k = g0 + m * 16 + r;
      gp[m] = g.list[k < cnt ? k : cnt - 1];
      const int n = gp[m] / HW, rem = gp[m] - n * HW;
      py[m] = rem / g.W;
      px[m] = rem - py[m] * g.W;
      base[m] = g.in + (size_t)gp[m] * g.in_cs + q * 8;
    }
    float4v acc[2][4];
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
      for (int f = 0; f < 4; ++f) acc[m][f] = float4v{0.f, 0.f, 0.f, 0.f};
    // per pixel: which taps fall inside the map (bit t), and their element offsets
    unsigned okm[2] = {0, 0};
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
      for (int tap = 0; tap < 9; ++tap) {
        const int dy = tap / 3 - 1, dx = tap % 3 - 1;
        if ((unsigned)(py[m] + dy) < (unsigned)g.H && (unsigned)(px[m] + dx) < (unsigned)g.W) okm[m] |= 1u << tap;
      }
    auto wptr = [&](int cc, int tap, int f) {
      const int tile = (f * 16) / g.rows_tile, row = f * 16 - tile * g.rows_tile + r;
      return g.wgt + ((size_t)(tile * g.nchunks + cc) * 9 + tap) * g.rows_tile * 4 + row * 4 + q;
    };
    for (int cc = 0; cc < g.nchunks; ++cc) {
      // the chunk's 18 activation fragments in flight together (one load -> wait -> MFMA per step was a chain of
      // ~70 dependent L2 round trips per wave), weights one tap ahead
      u32x4 bv[9][2];
#pragma unroll
      for (int tap = 0; tap < 9; ++tap) {
        const int dy = tap / 3 - 1, dx = tap % 3 - 1;
#pragma unroll
        for (int m = 0; m < 2; ++m) {
          const bool ok = (okm[m] >> tap) & 1u;
          bv[tap][m] = *reinterpret_cast<const u32x4*>(base[m] + (ok ? (dy * g.W + dx) * g.in_cs : 0) + cc * 32);
        }
      }
      u32x4 aw[2][4];
#pragma unroll
      for (int f = 0; f < 4; ++f) aw[0][f] = *reinterpret_cast<const u32x4*>(wptr(cc, 0, f));
#pragma unroll
      for (int tap = 0; tap < 9; ++tap) {
        if (tap + 1 < 9) {
#pragma unroll
          for (int f = 0; f < 4; ++f) aw[(tap + 1) & 1][f] = *reinterpret_cast<const u32x4*>(wptr(cc, tap + 1, f));
        }
#pragma unroll
        for (int f = 0; f < 4; ++f) {
          const half8 af = __builtin_bit_cast(half8, aw[tap & 1][f]);
#pragma unroll
          for (int m = 0; m < 2; ++m) {
            const bool ok = (okm[m] >> tap) & 1u;
            const half8 bf = __builtin_bit_cast(half8, ok ? bv[tap][m] : u32x4{0, 0, 0, 0});
            acc[m][f] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af, bf, acc[m][f], 0, 0, 0);
          }
        }
      }
    }
#pragma unroll
    for (int m = 0; m < 2; ++m) {
      if (g0 + m * 16 + r >= cnt) continue;
#pragma unroll
      for (int f = 0; f < 4; ++f) {
        const float4 b = biasr[f];
        float4v v = acc[m][f] + float4v{b.x, b.y, b.z, b.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = v[j] * __builtin_amdgcn_rcpf(1.0f + __expf(-v[j]));  // conv.hip's silu_f32
        const half4 h = __builtin_convertvector(v, half4);
        *reinterpret_cast<u32x2*>(g.out + (size_t)gp[m] * g.out_cs + f * 16 + q * 4) = __builtin_bit_cast(u32x2, h);
      }
    }
  }
}

// Lazy box branch: threshold on the class-max words, box conv for the passing anchors only, DFL decode of those.
int decode_lazy_forward(float* const box[3], const unsigned long long* const clsmax[3], const LazyBox& lb, int N,
                        const int Hl[3], const int Wl[3], int nc, float conf_thres, Cand* cands, int32_t* counts,
                        int32_t* lvl_list, int32_t* lvl_counts, int max_cand, hipStream_t stream) {
  if (N == 0) return EIOKU_OK;
  DecArgs a{};
  int A = 0, amax = 0;
  for (int l = 0; l < 3; ++l) {
    a.box[l] = box[l];
    a.box_w[l] = box[l];
    a.cls[l] = nullptr;
    a.cm[l] = clsmax[l];
    a.H[l] = Hl[l];
    a.W[l] = Wl[l];
    a.A0[l] = A;
    A += Hl[l] * Wl[l];
    if (Hl[l] * Wl[l] > amax) amax = Hl[l] * Wl[l];
    a.bin[l] = lb.in[l];
    a.bwgt[l] = lb.wgt[l];
    a.bbias[l] = lb.bias[l];
  }
  EIOKU_REQUIRE(A == max_cand, "candidate capacity %d != anchors %d", max_cand, A);
  a.bin_cs = lb.in_cs;
  EIOKU_REQUIRE(lb.nchunks >= 1 && lb.nchunks <= 4, "box branch input of %d channels: k_box_gather stages at most 128", lb.nchunks * 32);
  a.bnchunks = lb.nchunks;
  a.N = N;
  a.nc = nc;
  a.A = A;
  a.conf = conf_thres;
  a.cands = cands;
  a.keys = reinterpret_cast<unsigned long long*>(cands + (size_t)N * A);
  a.counts = counts;
  a.lvl_list = lvl_list;
  a.lvl_counts = lvl_counts;
  a.fcnt = nullptr;
  if (lb.deep) {
    a.fcnt = lb.fcnt;
    for (int l = 0; l < 3; ++l) {
      a.flat1[l] = lb.flat1[l];
      a.flat0[l] = lb.flat0[l];
    }
  }
  hipLaunchKernelGGL(k_decode_pass, dim3((unsigned)((A + 255) / 256), (unsigned)N), dim3(256), 0, stream, a);
  if (lb.deep) {
    // layer 0 at the neighbourhoods, then layer 1 at the anchors; both read / write the dense buffers sparsely
    GArgs g0, g1;
    for (int l = 0; l < 3; ++l) {
      g0.l[l] = GConv{lb.c0[l].in, lb.c0[l].wgt, lb.c0[l].bias, lb.c0[l].out, lb.flat0[l], lb.fcnt + 3 + l, Hl[l], Wl[l],
                      lb.c0[l].in_cs, lb.c0[l].out_cs, lb.c0[l].nchunks, lb.c0[l].rows_tile};
      g1.l[l] = GConv{lb.c1[l].in, lb.c1[l].wgt, lb.c1[l].bias, lb.c1[l].out, lb.flat1[l], lb.fcnt + l, Hl[l], Wl[l],
                      lb.c1[l].in_cs, lb.c1[l].out_cs, lb.c1[l].nchunks, lb.c1[l].rows_tile};
    }
    hipLaunchKernelGGL(k_conv3x3_gather, dim3(512, 3), dim3(256), 0, stream, g0);
    hipLaunchKernelGGL(k_conv3x3_gather, dim3(128, 3), dim3(256), 0, stream, g1);
  }
  hipLaunchKernelGGL(k_box_gather, dim3((unsigned)((amax + 127) / 128), (unsigned)(N * 3)), dim3(256), 0, stream, a);
  hipLaunchKernelGGL(k_decode_boxes, dim3((unsigned)((A + 255) / 256), (unsigned)N), dim3(256), 0, stream, a);
  EIOKU_LAUNCH_CHECK();
  return EIOKU_OK;
}

int decode_forward(const float* const box[3], const float* const cls[3], int N, const int Hl[3],
                   const int Wl[3], int nc, float conf_thres, Cand* cands, int32_t* counts, int max_cand,
                   hipStream_t stream, const unsigned long long* const clsmax[3]) {
  if (N == 0) return EIOKU_OK;
  DecArgs a;
  int A = 0;
  for (int l = 0; l < 3; ++l) {
    a.box[l] = box[l];
    a.cls[l] = cls ? cls[l] : nullptr;
    a.cm[l] = clsmax ? clsmax[l] : nullptr;
    a.H[l] = Hl[l];
    a.W[l] = Wl[l];
    a.A0[l] = A;
    A += Hl[l] * Wl[l];
  }
  EIOKU_REQUIRE(A == max_cand, "candidate capacity %d != anchors %d", max_cand, A);
  a.N = N;
  a.nc = nc;
  a.A = A;
  a.conf = conf_thres;
  a.cands = cands;
  // keys live right after the dense candidate array (see yolo.hip workspace layout)
  a.keys = reinterpret_cast<unsigned long long*>(cands + (size_t)N * A);
  a.counts = counts;
  if (clsmax) hipLaunchKernelGGL(k_decode_cm, dim3(blocks_for((long long)N * A)), dim3(256), 0, stream, a);
  else hipLaunchKernelGGL(k_decode, dim3(blocks_for((long long)N * A * 16)), dim3(256), 0, stream, a);
  EIOKU_LAUNCH_CHECK();
  return EIOKU_OK;
}

int nms_forward(const Cand* cands, const int32_t* counts, int N, int max_cand, float iou_thres, int max_det,
                float max_wh, ScaleParams sp, Det* dets, int32_t* det_counts, hipStream_t stream) {
  if (N == 0) return EIOKU_OK;
  EIOKU_REQUIRE(max_cand <= 16384, "more than 16384 anchors per image (%d) is not supported by the LDS sort", max_cand);
  EIOKU_REQUIRE(max_det > 0 && max_det <= 1024, "max_det %d out of range", max_det);
  int P = 1;
  while (P < max_cand) P <<= 1;
  const size_t lds = (size_t)P * 8 + (size_t)max_det * 6 * 4;
  static size_t lds_attr = 0;
  if (lds > 64 * 1024 && lds > lds_attr) {
    EIOKU_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_nms),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    lds_attr = lds;
  }
  NmsArgs a;
  a.cands = cands;
  a.keys = reinterpret_cast<const unsigned long long*>(cands + (size_t)N * max_cand);
  a.counts = counts;
  a.A = max_cand;
  a.max_det = max_det;
  a.iou = iou_thres;
  a.max_wh = max_wh;
  a.sp = sp;
  a.dets = dets;
  a.det_counts = det_counts;
  hipLaunchKernelGGL(k_nms, dim3(N), dim3(256), lds, stream, a);
  EIOKU_LAUNCH_CHECK();
  return EIOKU_OK;
}

}  // namespace eioku
