// Scene scoring kernels - HBM-bound byte/integer work.
//
// Both kernels keep the previous frame's tile in registers while a workgroup walks a run of
// consecutive frames, so every frame byte is read from HBM once (plus one extra frame per run
// of `seg` frames).  Per-frame sums are exact integers: wave shuffle-reduce -> LDS -> one
// 64-bit integer atomic per (workgroup, frame, channel); integer addition is associative, so
// the result does not depend on scheduling.
//
//   K1 k_sad_luma  : sum |Y_t - Y_{t-1}|            algorithmic bytes = W*H   per frame
//   K2 k_hsv_sums  : OpenCV 8-bit BGR->HSV, then sum |c_t - c_{t-1}| for c in {H,S,V}
//                                                   algorithmic bytes = 3*W*H per frame
#include "common.h"

#include <cmath>
#include <cstdlib>

using namespace eioku;

namespace {

constexpr int kBlock = 256;
constexpr int kG = 8;  // frames whose partial sums live in registers between reductions

// OpenCV RGB2HSV_b tables (hsv_shift = 12, hrange = 180), filled once on the host with
// cvRound == lrint semantics, then copied to constant memory.
__constant__ int c_sdiv[256];
__constant__ int c_hdiv[256];
bool g_tables_ready = false;

int ensure_tables() {
  if (g_tables_ready) return EIOKU_OK;
  int sdiv[256], hdiv[256];
  sdiv[0] = hdiv[0] = 0;
  for (int i = 1; i < 256; ++i) {
    sdiv[i] = (int)lrint((255 << 12) / (1.0 * i));
    hdiv[i] = (int)lrint((180 << 12) / (6.0 * i));
  }
  EIOKU_HIP_CHECK(hipMemcpyToSymbol(HIP_SYMBOL(c_sdiv), sdiv, sizeof(sdiv)));
  EIOKU_HIP_CHECK(hipMemcpyToSymbol(HIP_SYMBOL(c_hdiv), hdiv, sizeof(hdiv)));
  g_tables_ready = true;
  return EIOKU_OK;
}

// ---------------------------------------------------------------------------------------
// shared reduction tail: acc[G][C] per thread -> atomics on out[(t0+k)*C + c]
// ---------------------------------------------------------------------------------------
template <int G, int C>
__device__ __forceinline__ void flush_sums(unsigned (&acc)[G][C], unsigned (*s_red)[G * C], int t0,
                                           int t_end, unsigned long long* out) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
#pragma unroll
  for (int k = 0; k < G; ++k)
#pragma unroll
    for (int c = 0; c < C; ++c) {
      unsigned v = wave_reduce_add(acc[k][c]);
      if (lane == 0) s_red[wave][k * C + c] = v;
    }
  __syncthreads();
  if (tid < G * C) {
    int k = tid / C;
    if (t0 + k < t_end) {
      unsigned long long v = 0;
#pragma unroll
      for (int wv = 0; wv < kBlock / 64; ++wv) v += s_red[wv][tid];
      if (v) atomicAdd(&out[(size_t)(t0 + k) * C + (tid % C)], v);
    }
  }
  __syncthreads();
}

// ---------------------------------------------------------------------------------------
// K1: luma SAD.  Contiguous planes: thread owns U x 16 B of the plane.
// ---------------------------------------------------------------------------------------
// Per-frame sums of one run of frames: every wave reduces its lanes after each frame and adds the result to the
// workgroup's LDS accumulators (one LDS atomic per wave, frame and channel); the workgroup flushes them to global
// 64-bit atomics once, at the end of its run.  The frame loop is a plain rolled loop with NO branch in its body (frames
// past the run re-read the run's last frame and add nothing): behind a branch the compiler cannot count the prefetch
// in vmcnt and waits for ALL loads before it touches the current frame; unrolled, it hoists eight frames of loads and
// spills.
constexpr int kMaxSeg = 64;

template <int C>
__device__ __forceinline__ void flush_run(const unsigned* s_acc, int t_begin, int t_end, unsigned long long* out) {
  __syncthreads();
  for (int i = threadIdx.x; i < (t_end - t_begin) * C; i += kBlock) {
    const unsigned v = s_acc[i];
    if (v) atomicAdd(&out[(size_t)t_begin * C + i], (unsigned long long)v);
  }
}

template <int U>
__global__ __launch_bounds__(kBlock) void k_sad_luma(const uint8_t* __restrict__ frames,
                                                     size_t frame_stride, int n,
                                                     unsigned long long plane_bytes,
                                                     const uint8_t* __restrict__ prev, int seg,
                                                     unsigned long long* __restrict__ sad) {
  __shared__ unsigned s_acc[kMaxSeg];
  const int tid = threadIdx.x, lane = tid & 63;
  const unsigned long long nvec = plane_bytes >> 4;  // the <16-byte tail goes to k_sad_luma_strided
  const int t_begin = blockIdx.y * seg;
  const int t_end = min(n, t_begin + seg);
  if (tid < kMaxSeg) s_acc[tid] = 0;
  __syncthreads();

  // Lanes past the end of the plane read vector 0 (a valid address) and are masked to zero
  unsigned long long vidx[U];
  unsigned mask[U];
#pragma unroll
  for (int u = 0; u < U; ++u) {
    vidx[u] = ((unsigned long long)blockIdx.x * U + u) * kBlock + tid;
    mask[u] = vidx[u] < nvec ? 0xFFFFFFFFu : 0u;
    if (!mask[u]) vidx[u] = 0;
  }

  uint4 p[U], cur[U];
  const uint8_t* pf = t_begin > 0 ? frames + (size_t)(t_begin - 1) * frame_stride : prev;
  unsigned use = pf != nullptr ? 0xFFFFFFFFu : 0u;
#pragma unroll
  for (int u = 0; u < U; ++u) {
    p[u] = make_uint4(0, 0, 0, 0);
    if (pf) p[u] = reinterpret_cast<const uint4*>(pf)[vidx[u]];
    cur[u] = reinterpret_cast<const uint4*>(frames + (size_t)t_begin * frame_stride)[vidx[u]];
  }
#pragma unroll 1
  for (int t = t_begin; t < t_end; ++t) {
    uint4 nxt[U];
    const int tn = t + 1 < t_end ? t + 1 : t;
#pragma unroll
    for (int u = 0; u < U; ++u) nxt[u] = reinterpret_cast<const uint4*>(frames + (size_t)tn * frame_stride)[vidx[u]];
    unsigned a = 0;
#pragma unroll
    for (int u = 0; u < U; ++u) {
      unsigned d = 0;
      d = __builtin_amdgcn_sad_u8(cur[u].x, p[u].x, d);
      d = __builtin_amdgcn_sad_u8(cur[u].y, p[u].y, d);
      d = __builtin_amdgcn_sad_u8(cur[u].z, p[u].z, d);
      d = __builtin_amdgcn_sad_u8(cur[u].w, p[u].w, d);
      a += d & mask[u];
      p[u] = cur[u];
      cur[u] = nxt[u];
    }
    a = wave_sum_u32(a & use);
    if (lane == 0 && a) atomicAdd(&s_acc[t - t_begin], a);
    use = 0xFFFFFFFFu;
  }
  flush_run<1>(s_acc, t_begin, t_end, sad);
}

// Generic (row-strided or unaligned) planes: byte loads, 16 pixels per thread.
__global__ __launch_bounds__(kBlock) void k_sad_luma_strided(const uint8_t* __restrict__ frames,
                                                             size_t frame_stride,
                                                             size_t row_stride, int n, int h, int w,
                                                             unsigned long long pix_begin,
                                                             const uint8_t* __restrict__ prev,
                                                             int seg,
                                                             unsigned long long* __restrict__ sad) {
  __shared__ unsigned s_red[kBlock / 64][kG];
  const int tid = threadIdx.x;
  const unsigned long long npix = (unsigned long long)h * w;
  const unsigned long long i0 = pix_begin + ((unsigned long long)blockIdx.x * kBlock + tid) * 16;
  const int t_begin = blockIdx.y * seg;
  const int t_end = min(n, t_begin + seg);
  size_t off[16];
  bool live[16];
#pragma unroll
  for (int j = 0; j < 16; ++j) {
    unsigned long long i = i0 + j;
    live[j] = i < npix;
    unsigned long long y = live[j] ? i / w : 0;
    off[j] = live[j] ? y * row_stride + (i - y * w) : 0;
  }
  uint8_t p[16];
  const uint8_t* pf = t_begin > 0 ? frames + (size_t)(t_begin - 1) * frame_stride : prev;
  bool have_prev = pf != nullptr;
#pragma unroll
  for (int j = 0; j < 16; ++j) p[j] = (have_prev && live[j]) ? pf[off[j]] : 0;
  for (int t0 = t_begin; t0 < t_end; t0 += kG) {
    unsigned acc[kG][1];
#pragma unroll
    for (int k = 0; k < kG; ++k) acc[k][0] = 0;
#pragma unroll
    for (int k = 0; k < kG; ++k) {
      const int t = t0 + k;
      if (t < t_end) {
        const uint8_t* f = frames + (size_t)t * frame_stride;
#pragma unroll
        for (int j = 0; j < 16; ++j) {
          uint8_t c = live[j] ? f[off[j]] : 0;
          if (have_prev) acc[k][0] += (unsigned)abs((int)c - (int)p[j]);
          p[j] = c;
        }
        have_prev = true;
      }
    }
    flush_sums<kG, 1>(acc, s_red, t0, t_end, sad);
  }
}

// ---------------------------------------------------------------------------------------
// K2: BGR -> HSV (OpenCV RGB2HSV_b, hrange 180) + per-channel SAD against the previous frame
// ---------------------------------------------------------------------------------------
// 24-bit multiply-adds spelled out: left to the compiler, __mul24 of operands whose range it can see became a
// quarter-rate v_mul_lo_u32 (20 of the 32 multiplies per 16 pixels)
__device__ __forceinline__ int mad_u24(int a, int b, int c) {
  int d;
  asm("v_mad_u32_u24 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
  return d;
}
__device__ __forceinline__ int mad_i24(int a, int b, int c) {
  int d;
  asm("v_mad_i32_i24 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
  return d;
}
__device__ __forceinline__ int max3_u(int a, int b, int c) {
  int d;
  asm("v_max3_u32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
  return d;
}
__device__ __forceinline__ int min3_u(int a, int b, int c) {
  int d;
  asm("v_min3_u32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
  return d;
}

__device__ __forceinline__ void hsv_px(int b, int g, int r, const int* __restrict__ sdiv,
                                       const int* __restrict__ hdiv, int& H, int& S, int& V) {
  const int v = max3_u(b, g, r);
  const int vmin = min3_u(b, g, r);
  const int diff = v - vmin;
  const int s = mad_u24(diff, sdiv[v], 2048) >> 12;
  // OpenCV's mask arithmetic, (vr & (g - b)) + (~vr & ((vg & (b - r + 2 diff)) + (~vg & (r - g + 4 diff)))), as two
  // selects over values that are all computed first (nothing for the compiler to branch around)
  const int hr = g - b, hg = b - r + 2 * diff, hb = r - g + 4 * diff;
  const int hsel = v == g ? hg : hb;
  int h = v == r ? hr : hsel;
  h = mad_i24(h, hdiv[diff], 2048) >> 12;  // arithmetic shift: floor, as in the C source
  // h in [-90, 180]: "h += 180 if h < 0" as an unsigned minimum (a negative h is a huge unsigned number)
  H = (int)min((unsigned)h, (unsigned)(h + 180));
  S = s;
  V = v;
}

// 4 pixels = 12 bytes = 3 dwords -> packed H, S, V quads (one byte per pixel)
__device__ __forceinline__ void hsv_quad(unsigned d0, unsigned d1, unsigned d2,
                                         const int* __restrict__ sdiv,
                                         const int* __restrict__ hdiv, unsigned& H, unsigned& S,
                                         unsigned& V) {
  int h0, s0, v0, h1, s1, v1, h2, s2, v2, h3, s3, v3;
  hsv_px(d0 & 0xFF, (d0 >> 8) & 0xFF, (d0 >> 16) & 0xFF, sdiv, hdiv, h0, s0, v0);
  hsv_px(d0 >> 24, d1 & 0xFF, (d1 >> 8) & 0xFF, sdiv, hdiv, h1, s1, v1);
  hsv_px((d1 >> 16) & 0xFF, d1 >> 24, d2 & 0xFF, sdiv, hdiv, h2, s2, v2);
  hsv_px((d2 >> 8) & 0xFF, (d2 >> 16) & 0xFF, d2 >> 24, sdiv, hdiv, h3, s3, v3);
  H = (unsigned)h0 | ((unsigned)h1 << 8) | ((unsigned)h2 << 16) | ((unsigned)h3 << 24);
  S = (unsigned)s0 | ((unsigned)s1 << 8) | ((unsigned)s2 << 16) | ((unsigned)s3 << 24);
  V = (unsigned)v0 | ((unsigned)v1 << 8) | ((unsigned)v2 << 16) | ((unsigned)v3 << 24);
}

// The same conversion on 16-bit pairs (VOP3P): two pixels per instruction for everything up to the hue numerator,
// one 24-bit multiply-add per pixel and channel, byte permutes instead of shifts.  k_hsv_sums is VALU-issue bound
// (PMC: SQ_ACTIVE_INST_VALU = 99 % of the kernel's cycles at 35 instructions per pixel); this form needs ~21.
//   * sdiv4[v] = 4 sdiv[v], hdiv16[d] = 16 hdiv[d], the multiplicand of S is 4 diff: both products come out scaled
//     by 16, so that (x + 2048) >> 12 is the byte / half-word at bit 16 of (16 x + 32768): no shift instructions;
//   * v == r / v == g as sign masks of r - v, g - v (both <= 0), the two selects as v_bfi_b32 on the pair;
//   * "h += 180 if h < 0" as an unsigned 16-bit minimum of h and h + 180.
typedef short s16x2 __attribute__((ext_vector_type(2)));
typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));

template <int W>
__device__ __forceinline__ unsigned word_shl2(unsigned a) {  // 4 * (16-bit half W of a)
  unsigned d;
  if (W == 0) asm("v_lshlrev_b32_sdwa %0, 2, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_0" : "=v"(d) : "v"(a));
  else asm("v_lshlrev_b32_sdwa %0, 2, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1" : "=v"(d) : "v"(a));
  return d;
}
template <int K>
__device__ __forceinline__ unsigned pk_mad_i16(unsigned a, unsigned c) {  // a * K + c on both halves
  unsigned d;
  asm("v_pk_mad_i16 %0, %1, %2, %3 op_sel_hi:[1,0,1]" : "=v"(d) : "v"(a), "n"(K), "v"(c));
  return d;
}
__device__ __forceinline__ unsigned pk_sign_i16(unsigned a) {  // 0xffff in every half that is negative
  unsigned d;
  asm("v_pk_ashrrev_i16 %0, 15, %1 op_sel_hi:[0,1]" : "=v"(d) : "v"(a));
  return d;
}
__device__ __forceinline__ unsigned bfi(unsigned mask, unsigned x, unsigned y) {  // (mask & x) | (~mask & y)
  unsigned d;
  asm("v_bfi_b32 %0, %1, %2, %3" : "=v"(d) : "v"(mask), "v"(x), "v"(y));
  return d;
}

__device__ __forceinline__ void hsv_pair(unsigned B, unsigned G, unsigned R, const int* __restrict__ sdiv4,
                                         const int* __restrict__ hdiv16, unsigned& Hp, int& s0, int& s1, unsigned& Vp) {
  const u16x2 b = __builtin_bit_cast(u16x2, B), g = __builtin_bit_cast(u16x2, G), r = __builtin_bit_cast(u16x2, R);
  const u16x2 v = __builtin_elementwise_max(__builtin_elementwise_max(b, g), r);
  const u16x2 vmin = __builtin_elementwise_min(__builtin_elementwise_min(b, g), r);
  const u16x2 diff = v - vmin;
  const s16x2 bs = __builtin_bit_cast(s16x2, b), gs = __builtin_bit_cast(s16x2, g), rs = __builtin_bit_cast(s16x2, r);
  const s16x2 vs = __builtin_bit_cast(s16x2, v), ds = __builtin_bit_cast(s16x2, diff);
  // (spelled out: the compiler turned the sign masks into per-half compares + selects + a re-pack)
  const unsigned hr = __builtin_bit_cast(unsigned, gs - bs);
  const unsigned hg = pk_mad_i16<2>(__builtin_bit_cast(unsigned, ds), __builtin_bit_cast(unsigned, bs - rs));
  const unsigned hb = pk_mad_i16<4>(__builtin_bit_cast(unsigned, ds), __builtin_bit_cast(unsigned, rs - gs));
  const unsigned nr = pk_sign_i16(__builtin_bit_cast(unsigned, rs - vs));  // all ones where v != r
  const unsigned ng = pk_sign_i16(__builtin_bit_cast(unsigned, gs - vs));
  const int hn = (int)bfi(nr, bfi(ng, hb, hg), hr);
  // table byte offsets 4 v, 4 diff of the two pixels: shift and half-word select in one SDWA instruction each
  const unsigned v4a = word_shl2<0>(__builtin_bit_cast(unsigned, v)), v4b = word_shl2<1>(__builtin_bit_cast(unsigned, v));
  const unsigned d4a = word_shl2<0>(__builtin_bit_cast(unsigned, diff)), d4b = word_shl2<1>(__builtin_bit_cast(unsigned, diff));
  const char* st = reinterpret_cast<const char*>(sdiv4);
  const char* ht = reinterpret_cast<const char*>(hdiv16);
  s0 = mad_u24((int)d4a, *reinterpret_cast<const int*>(st + v4a), 32768);  // S of the pixel = byte 2
  s1 = mad_u24((int)d4b, *reinterpret_cast<const int*>(st + v4b), 32768);
  const int h0 = mad_i24((int)(short)(hn & 0xFFFF), *reinterpret_cast<const int*>(ht + d4a), 32768);  // h = high half
  const int h1 = mad_i24(hn >> 16, *reinterpret_cast<const int*>(ht + d4b), 32768);
  const u16x2 h = __builtin_bit_cast(u16x2, __builtin_amdgcn_perm((unsigned)h1, (unsigned)h0, 0x07060302u));
  Hp = __builtin_bit_cast(unsigned, __builtin_elementwise_min(h, (u16x2)(h + (unsigned short)180)));
  Vp = __builtin_bit_cast(unsigned, v);
}

__device__ __forceinline__ void hsv_quad_pk(unsigned d0, unsigned d1, unsigned d2, const int* __restrict__ sdiv4,
                                            const int* __restrict__ hdiv16, unsigned& H, unsigned& S, unsigned& V) {
  // 12 bytes b0 g0 r0 b1 | g1 r1 b2 g2 | r2 b3 g3 r3 -> zero-extended 16-bit pairs (v_perm_b32: selector 0x0c = 0x00;
  // indices 0-3 = bytes of the second operand, 4-7 = bytes of the first)
  const unsigned B01 = __builtin_amdgcn_perm(d0, d0, 0x0c030c00u), G01 = __builtin_amdgcn_perm(d1, d0, 0x0c040c01u);
  const unsigned R01 = __builtin_amdgcn_perm(d1, d0, 0x0c050c02u), B23 = __builtin_amdgcn_perm(d2, d1, 0x0c050c02u);
  const unsigned G23 = __builtin_amdgcn_perm(d2, d1, 0x0c060c03u), R23 = __builtin_amdgcn_perm(d2, d2, 0x0c030c00u);
  unsigned H01, H23, V01, V23;
  int s0, s1, s2, s3;
  hsv_pair(B01, G01, R01, sdiv4, hdiv16, H01, s0, s1, V01);
  hsv_pair(B23, G23, R23, sdiv4, hdiv16, H23, s2, s3, V23);
  H = __builtin_amdgcn_perm(H23, H01, 0x06040200u);
  V = __builtin_amdgcn_perm(V23, V01, 0x06040200u);
  S = __builtin_amdgcn_perm((unsigned)s1, (unsigned)s0, 0x0c0c0602u) | __builtin_amdgcn_perm((unsigned)s3, (unsigned)s2, 0x06020c0cu);
}

struct Quad3 {
  unsigned d0, d1, d2;
};

template <bool ALIGNED>
__device__ __forceinline__ Quad3 load_quad(const uint8_t* __restrict__ f, unsigned long long q) {
  Quad3 r;
  if (ALIGNED) {
    const unsigned* p = reinterpret_cast<const unsigned*>(f) + q * 3;
    r.d0 = p[0];
    r.d1 = p[1];
    r.d2 = p[2];
  } else {
    const uint8_t* p = f + q * 12;
    r.d0 = p[0] | (p[1] << 8) | (p[2] << 16) | ((unsigned)p[3] << 24);
    r.d1 = p[4] | (p[5] << 8) | (p[6] << 16) | ((unsigned)p[7] << 24);
    r.d2 = p[8] | (p[9] << 8) | (p[10] << 16) | ((unsigned)p[11] << 24);
  }
  return r;
}

// thread owns Q pixel quads per frame; a workgroup covers Q*256 quads = Q*1024 pixels; the next frame is in flight
// (registers) while one is converted.  Loop structure and per-frame reduction: see k_sad_luma.
template <int Q, bool ALIGNED>
__global__ __launch_bounds__(kBlock, 4) void k_hsv_sums(const uint8_t* __restrict__ frames,
                                                     size_t frame_stride, int n,
                                                     unsigned long long npix,
                                                     const uint8_t* __restrict__ prev, int seg,
                                                     unsigned long long* __restrict__ sums) {
  __shared__ int s_sdiv[256];
  __shared__ int s_hdiv[256];
  __shared__ unsigned s_acc[kMaxSeg * 3];
  const int tid = threadIdx.x, lane = tid & 63;
  s_sdiv[tid] = 4 * c_sdiv[tid];   // pre-scaled for hsv_quad_pk
  s_hdiv[tid] = 16 * c_hdiv[tid];
  if (tid < kMaxSeg * 3) s_acc[tid] = 0;
  __syncthreads();

  const unsigned long long nquads = npix >> 2;  // the <4-pixel tail goes to k_hsv_sums_tail
  const int t_begin = blockIdx.y * seg;
  const int t_end = min(n, t_begin + seg);

  // Lanes past the end of the frame read quad 0 (a valid address) and are masked to zero, so
  // the hot loop has no per-lane branches.
  unsigned long long q[Q];
  unsigned mask[Q];
#pragma unroll
  for (int u = 0; u < Q; ++u) {
    q[u] = ((unsigned long long)blockIdx.x * Q + u) * kBlock + tid;
    mask[u] = q[u] < nquads ? 0xFFFFFFFFu : 0u;
    if (!mask[u]) q[u] = 0;
  }

  unsigned pH[Q], pS[Q], pV[Q];
  const uint8_t* pf = t_begin > 0 ? frames + (size_t)(t_begin - 1) * frame_stride : prev;
  unsigned use = pf != nullptr ? 0xFFFFFFFFu : 0u;
  Quad3 cur[Q];
#pragma unroll
  for (int u = 0; u < Q; ++u) {
    pH[u] = pS[u] = pV[u] = 0;
    if (pf) {
      Quad3 d = load_quad<ALIGNED>(pf, q[u]);
      hsv_quad_pk(d.d0, d.d1, d.d2, s_sdiv, s_hdiv, pH[u], pS[u], pV[u]);
      pH[u] &= mask[u];
      pS[u] &= mask[u];
      pV[u] &= mask[u];
    }
    cur[u] = load_quad<ALIGNED>(frames + (size_t)t_begin * frame_stride, q[u]);
  }
  // One frame: the next one is requested first (registers `nx`), then `c` is converted and compared with the previous
  // frame's H / S / V.  Two steps per trip with the register sets swapped, so that no frame is ever copied; only the
  // workgroup that holds the end of the frame pays for the lane masks (MASKED is workgroup-uniform).
  auto step = [&](int t, Quad3 (&c)[Q], Quad3 (&nx)[Q], auto masked_tag) {
    constexpr bool MASKED = decltype(masked_tag)::value;
    {
      const int tn = t + 1 < t_end ? t + 1 : t;
      const uint8_t* f = frames + (size_t)tn * frame_stride;
#pragma unroll
      for (int u = 0; u < Q; ++u) nx[u] = load_quad<ALIGNED>(f, q[u]);
    }
    unsigned aH = 0, aS = 0, aV = 0;
#pragma unroll
    for (int u = 0; u < Q; ++u) {
      unsigned H, S, V;
      hsv_quad_pk(c[u].d0, c[u].d1, c[u].d2, s_sdiv, s_hdiv, H, S, V);
      if (MASKED) {
        H &= mask[u];
        S &= mask[u];
        V &= mask[u];
      }
      aH = __builtin_amdgcn_sad_u8(H, pH[u], aH);
      aS = __builtin_amdgcn_sad_u8(S, pS[u], aS);
      aV = __builtin_amdgcn_sad_u8(V, pV[u], aV);
      pH[u] = H;
      pS[u] = S;
      pV[u] = V;
    }
    aH = wave_sum_u32(aH & use);
    aS = wave_sum_u32(aS & use);
    aV = wave_sum_u32(aV & use);
    if (lane == 0) {
      unsigned* a = s_acc + (t - t_begin) * 3;
      if (aH) atomicAdd(a, aH);
      if (aS) atomicAdd(a + 1, aS);
      if (aV) atomicAdd(a + 2, aV);
    }
    use = 0xFFFFFFFFu;
  };
  auto run = [&](auto masked_tag) {
    Quad3 alt[Q];
    int t = t_begin;
#pragma unroll 1
    for (; t + 1 < t_end; t += 2) {
      step(t, cur, alt, masked_tag);
      step(t + 1, alt, cur, masked_tag);
    }
    if (t < t_end) step(t, cur, alt, masked_tag);
  };
  if (((unsigned long long)blockIdx.x + 1) * Q * kBlock > nquads) run(std::true_type{});
  else run(std::false_type{});
  flush_run<3>(s_acc, t_begin, t_end, sums);
}

// K1 on decoded BGR frames (single-pass ingest: the frames the detectors read are the only copy in HBM): the luma
// is OpenCV's 8-bit COLOR_BGR2YUV_I420 luma, Y = (269484 R + 528482 G + 102760 B + (16 << 20) + (1 << 19)) >> 20
// (BT.601 studio range, 20-bit fixed point: eioku_amd/frames.py::bgr_to_luma_bt601 on the host), then |Y_t - Y_{t-1}|
// summed per frame.  Same structure as k_hsv_sums; 3 bytes per pixel, ~8 integer operations per pixel.
__device__ __forceinline__ unsigned luma_quad(unsigned d0, unsigned d1, unsigned d2) {
  auto y = [](unsigned b, unsigned g, unsigned r) {
    return (__umul24(r, 269484u) + __umul24(g, 528482u) + __umul24(b, 102760u) + ((16u << 20) + (1u << 19))) >> 20;
  };
  const unsigned y0 = y(d0 & 0xFF, (d0 >> 8) & 0xFF, (d0 >> 16) & 0xFF);
  const unsigned y1 = y(d0 >> 24, d1 & 0xFF, (d1 >> 8) & 0xFF);
  const unsigned y2 = y((d1 >> 16) & 0xFF, d1 >> 24, d2 & 0xFF);
  const unsigned y3 = y((d2 >> 8) & 0xFF, (d2 >> 16) & 0xFF, d2 >> 24);
  return y0 | (y1 << 8) | (y2 << 16) | (y3 << 24);
}

template <int Q, bool ALIGNED>
__global__ __launch_bounds__(kBlock) void k_sad_luma_bgr(const uint8_t* __restrict__ frames, size_t frame_stride, int n,
                                                         unsigned long long npix, const uint8_t* __restrict__ prev, int seg,
                                                         unsigned long long* __restrict__ sad) {
  __shared__ unsigned s_acc[kMaxSeg];
  const int tid = threadIdx.x, lane = tid & 63;
  if (tid < kMaxSeg) s_acc[tid] = 0;
  __syncthreads();
  const unsigned long long nquads = npix >> 2;  // a < 4-pixel tail is added by the host wrapper's caller (width % 4 == 0 required)
  const int t_begin = blockIdx.y * seg;
  const int t_end = min(n, t_begin + seg);
  unsigned long long q[Q];
  unsigned mask[Q];
#pragma unroll
  for (int u = 0; u < Q; ++u) {
    q[u] = ((unsigned long long)blockIdx.x * Q + u) * kBlock + tid;
    mask[u] = q[u] < nquads ? 0xFFFFFFFFu : 0u;
    if (!mask[u]) q[u] = 0;
  }
  unsigned pY[Q];
  const uint8_t* pf = t_begin > 0 ? frames + (size_t)(t_begin - 1) * frame_stride : prev;
  unsigned use = pf != nullptr ? 0xFFFFFFFFu : 0u;
  Quad3 cur[Q];
#pragma unroll
  for (int u = 0; u < Q; ++u) {
    pY[u] = 0;
    if (pf) {
      const Quad3 d = load_quad<ALIGNED>(pf, q[u]);
      pY[u] = luma_quad(d.d0, d.d1, d.d2) & mask[u];
    }
    cur[u] = load_quad<ALIGNED>(frames + (size_t)t_begin * frame_stride, q[u]);
  }
#pragma unroll 1
  for (int t = t_begin; t < t_end; ++t) {
    Quad3 nxt[Q];
    {
      const int tn = t + 1 < t_end ? t + 1 : t;
      const uint8_t* f = frames + (size_t)tn * frame_stride;
#pragma unroll
      for (int u = 0; u < Q; ++u) nxt[u] = load_quad<ALIGNED>(f, q[u]);
    }
    unsigned a = 0;
#pragma unroll
    for (int u = 0; u < Q; ++u) {
      const unsigned Y = luma_quad(cur[u].d0, cur[u].d1, cur[u].d2) & mask[u];
      a = __builtin_amdgcn_sad_u8(Y, pY[u], a);
      pY[u] = Y;
      cur[u] = nxt[u];
    }
    a = wave_sum_u32(a & use);
    if (lane == 0 && a) atomicAdd(&s_acc[t - t_begin], a);
    use = 0xFFFFFFFFu;
  }
  flush_run<1>(s_acc, t_begin, t_end, sad);
}

// pixels [pix_begin, npix) (< 4 of them): one thread per pixel walks all frames.
__global__ void k_hsv_sums_tail(const uint8_t* __restrict__ frames, size_t frame_stride, int n,
                                unsigned long long pix_begin, unsigned long long npix,
                                const uint8_t* __restrict__ prev,
                                unsigned long long* __restrict__ sums) {
  __shared__ int s_sdiv[256];
  __shared__ int s_hdiv[256];
  for (int i = threadIdx.x; i < 256; i += blockDim.x) {
    s_sdiv[i] = c_sdiv[i];
    s_hdiv[i] = c_hdiv[i];
  }
  __syncthreads();
  unsigned long long i = pix_begin + threadIdx.x;
  if (i >= npix) return;
  int pH = 0, pS = 0, pV = 0;
  bool have_prev = prev != nullptr;
  if (have_prev) hsv_px(prev[i * 3], prev[i * 3 + 1], prev[i * 3 + 2], s_sdiv, s_hdiv, pH, pS, pV);
  for (int t = 0; t < n; ++t) {
    const uint8_t* px = frames + (size_t)t * frame_stride + i * 3;
    int H, S, V;
    hsv_px(px[0], px[1], px[2], s_sdiv, s_hdiv, H, S, V);
    if (have_prev) {
      atomicAdd(&sums[(size_t)t * 3 + 0], (unsigned long long)abs(H - pH));
      atomicAdd(&sums[(size_t)t * 3 + 1], (unsigned long long)abs(S - pS));
      atomicAdd(&sums[(size_t)t * 3 + 2], (unsigned long long)abs(V - pV));
    }
    pH = H;
    pS = S;
    pV = V;
    have_prev = true;
  }
}

__global__ __launch_bounds__(kBlock) void k_bgr2hsv(const uint8_t* __restrict__ bgr,
                                                    unsigned long long npix,
                                                    uint8_t* __restrict__ out) {
  __shared__ int s_sdiv[256];
  __shared__ int s_hdiv[256];
  s_sdiv[threadIdx.x] = c_sdiv[threadIdx.x];
  s_hdiv[threadIdx.x] = c_hdiv[threadIdx.x];
  __syncthreads();
  unsigned long long i = blockIdx.x * (unsigned long long)kBlock + threadIdx.x;
  unsigned long long stride = (unsigned long long)gridDim.x * kBlock;
  for (; i < npix; i += stride) {
    int H, S, V;
    hsv_px(bgr[i * 3], bgr[i * 3 + 1], bgr[i * 3 + 2], s_sdiv, s_hdiv, H, S, V);
    out[i * 3] = (uint8_t)H;
    out[i * 3 + 1] = (uint8_t)S;
    out[i * 3 + 2] = (uint8_t)V;
  }
}

// Split n frames into runs of `seg` (multiple of kG) so the grid has >= ~8 workgroups per CU
// when the frame is small, while keeping the one-extra-frame-per-run overhead low.
// per_cu: workgroups the launch should offer every CU.  Every run converts (K2) or reads (K1) the frame before its
// first one again, so shorter runs cost 1 / seg extra work: K2, VALU-bound with 5 resident workgroups per CU, is 3 %
// faster at 1080p with runs of 16 (per_cu = 5) than with runs of 8 (per_cu = 8); measured 100.9 vs 104.1 us.
int pick_seg(int n, unsigned long long blocks_x, int per_cu = 8) {
  const unsigned long long want = (unsigned long long)num_cus() * per_cu;
  int seg = ((n + kG - 1) / kG) * kG;  // one run
  if (seg > kMaxSeg) seg = kMaxSeg;     // the run's sums live in LDS
  while (seg > kG && blocks_x * (unsigned long long)((n + seg - 1) / seg) < want) {
    seg = ((seg / 2 + kG - 1) / kG) * kG;
  }
  return seg < kG ? kG : seg;
}

}  // namespace

extern "C" {

int eioku_scene_sad_luma(const uint8_t* y_frames, int n, int h, int w, size_t row_stride,
                         size_t frame_stride, const uint8_t* prev, uint64_t* sad_out, int mem,
                         void* stream_) {
  EIOKU_REQUIRE_INIT();
  EIOKU_REQUIRE(n >= 0 && h > 0 && w > 0, "bad shape n=%d h=%d w=%d", n, h, w);
  EIOKU_REQUIRE(mem == EIOKU_MEM_HOST || mem == EIOKU_MEM_DEVICE, "bad mem flag %d", mem);
  EIOKU_REQUIRE(row_stride >= (size_t)w, "row_stride %zu < w %d", row_stride, w);
  EIOKU_REQUIRE(frame_stride >= row_stride * (size_t)(h - 1) + (size_t)w || n <= 1,
                "frame_stride %zu too small", frame_stride);
  if (n == 0) return EIOKU_OK;
  EIOKU_REQUIRE(y_frames && sad_out, "NULL pointer");
  hipStream_t stream = (hipStream_t)stream_;

  const uint8_t* d_frames = y_frames;
  const uint8_t* d_prev = prev;
  unsigned long long* d_out = (unsigned long long*)sad_out;
  const size_t plane_span = row_stride * (size_t)(h - 1) + (size_t)w;
  if (mem == EIOKU_MEM_HOST) {
    size_t total = frame_stride * (size_t)(n - 1) + plane_span;
    uint8_t* din = (uint8_t*)scratch(kSlotIn, total);
    d_out = (unsigned long long*)scratch(kSlotOut, sizeof(uint64_t) * n);
    if (!din || !d_out) return EIOKU_ENOMEM;
    EIOKU_HIP_CHECK(hipMemcpyAsync(din, y_frames, total, hipMemcpyHostToDevice, stream));
    d_frames = din;
    if (prev) {
      uint8_t* dp = (uint8_t*)scratch(kSlotPrev, plane_span);
      if (!dp) return EIOKU_ENOMEM;
      EIOKU_HIP_CHECK(hipMemcpyAsync(dp, prev, plane_span, hipMemcpyHostToDevice, stream));
      d_prev = dp;
    }
  }
  EIOKU_HIP_CHECK(hipMemsetAsync(d_out, 0, sizeof(uint64_t) * n, stream));

  const bool contiguous = row_stride == (size_t)w;
  const bool aligned = (((uintptr_t)d_frames | frame_stride | (d_prev ? (uintptr_t)d_prev : 0)) & 15) == 0;
  if (contiguous && aligned) {
    // 4 x 16 B per thread, loads issued frame by frame (r02 sweep: 4.6-4.7 TB/s at 64 x 1080p = 28 us for 133 MB, the
    // copy-kernel rate of this chip minus launch ramp; deeper explicit prefetch was SLOWER: 3.8-4.5 TB/s)
    constexpr int U = 4;
    unsigned long long plane = (unsigned long long)h * w;
    unsigned long long nvec = plane >> 4;
    unsigned long long bx = (nvec + (unsigned long long)kBlock * U - 1) / ((unsigned long long)kBlock * U);
    if (bx == 0) bx = 1;
    // HBM-bound: a run re-reads the frame before its first one, so long runs win as soon as every CU has a workgroup
    // (64 x 1080p, runs of 8 / 16 / 32 frames: 26.7 / 26.5 / 25.6 us)
    int seg = pick_seg(n, bx, 1);
    dim3 grid((unsigned)bx, (unsigned)((n + seg - 1) / seg));
    if (nvec) {
      prof_start(EIOKU_PROF_SCENE_SAD, stream);
      hipLaunchKernelGGL((k_sad_luma<U>), grid, dim3(kBlock), 0, stream, d_frames, frame_stride, n, plane, d_prev, seg, d_out);
      prof_stop(EIOKU_PROF_SCENE_SAD, stream);
    }
    if (plane & 15)  // ragged tail (< 16 bytes)
      hipLaunchKernelGGL(k_sad_luma_strided, dim3(1, 1), dim3(kBlock), 0, stream, d_frames,
                         frame_stride, row_stride, n, h, w, nvec << 4, d_prev,
                         ((n + kG - 1) / kG) * kG, d_out);
  } else {
    unsigned long long npix = (unsigned long long)h * w;
    unsigned long long bx = (npix + kBlock * 16ull - 1) / (kBlock * 16ull);
    int seg = pick_seg(n, bx);
    dim3 grid((unsigned)bx, (unsigned)((n + seg - 1) / seg));
    hipLaunchKernelGGL(k_sad_luma_strided, grid, dim3(kBlock), 0, stream, d_frames, frame_stride,
                       row_stride, n, h, w, 0ull, d_prev, seg, d_out);
  }
  EIOKU_LAUNCH_CHECK();
  if (mem == EIOKU_MEM_HOST) {
    EIOKU_HIP_CHECK(hipMemcpyAsync(sad_out, d_out, sizeof(uint64_t) * n, hipMemcpyDeviceToHost, stream));
    EIOKU_HIP_CHECK(hipStreamSynchronize(stream));
  }
  return EIOKU_OK;
}

int eioku_scene_hsv_sums(const uint8_t* bgr_frames, int n, int h, int w, size_t frame_stride,
                         const uint8_t* prev, uint64_t* sums_out, int mem, void* stream_) {
  EIOKU_REQUIRE_INIT();
  EIOKU_REQUIRE(n >= 0 && h > 0 && w > 0, "bad shape n=%d h=%d w=%d", n, h, w);
  EIOKU_REQUIRE(mem == EIOKU_MEM_HOST || mem == EIOKU_MEM_DEVICE, "bad mem flag %d", mem);
  const size_t frame_bytes = (size_t)h * w * 3;
  EIOKU_REQUIRE(frame_stride >= frame_bytes || n <= 1, "frame_stride %zu < frame bytes %zu",
                frame_stride, frame_bytes);
  if (n == 0) return EIOKU_OK;
  EIOKU_REQUIRE(bgr_frames && sums_out, "NULL pointer");
  int rc = ensure_tables();
  if (rc) return rc;
  hipStream_t stream = (hipStream_t)stream_;

  const uint8_t* d_frames = bgr_frames;
  const uint8_t* d_prev = prev;
  unsigned long long* d_out = (unsigned long long*)sums_out;
  if (mem == EIOKU_MEM_HOST) {
    size_t total = frame_stride * (size_t)(n - 1) + frame_bytes;
    uint8_t* din = (uint8_t*)scratch(kSlotIn, total);
    d_out = (unsigned long long*)scratch(kSlotOut, sizeof(uint64_t) * 3 * n);
    if (!din || !d_out) return EIOKU_ENOMEM;
    EIOKU_HIP_CHECK(hipMemcpyAsync(din, bgr_frames, total, hipMemcpyHostToDevice, stream));
    d_frames = din;
    if (prev) {
      uint8_t* dp = (uint8_t*)scratch(kSlotPrev, frame_bytes);
      if (!dp) return EIOKU_ENOMEM;
      EIOKU_HIP_CHECK(hipMemcpyAsync(dp, prev, frame_bytes, hipMemcpyHostToDevice, stream));
      d_prev = dp;
    }
  }
  EIOKU_HIP_CHECK(hipMemsetAsync(d_out, 0, sizeof(uint64_t) * 3 * n, stream));

  // Q = 4 quads per thread, one frame ahead, five waves per SIMD: the r02 sweep (profiles/r02_scene_sweep.txt) moved K2 by
  // < 4 % over quads 1..4 x frames in flight 1..4 x 4..16 workgroups per CU - it is bound by its ~22 integer VALU
  // operations per pixel (PMC: the vector pipes are 87 % busy at 1080p), not by bytes in flight.  (Q = 2 for small
  // frames - 64 x 640^2 offers only 800 workgroups to 1280 slots - measured the same: 31.4 us.)
  constexpr int Q = 4;
  const unsigned long long npix = (unsigned long long)h * w;
  unsigned long long nquads = npix >> 2;
  unsigned long long bx = (nquads + (unsigned long long)kBlock * Q - 1) / ((unsigned long long)kBlock * Q);
  if (bx == 0) bx = 1;
  int seg = pick_seg(n, bx, 5);
  dim3 grid((unsigned)bx, (unsigned)((n + seg - 1) / seg));
  const bool aligned = (((uintptr_t)d_frames | frame_stride | (d_prev ? (uintptr_t)d_prev : 0)) & 3) == 0;
  if (nquads) {
    prof_start(EIOKU_PROF_SCENE_HSV, stream);
    if (aligned)
      hipLaunchKernelGGL((k_hsv_sums<Q, true>), grid, dim3(kBlock), 0, stream, d_frames, frame_stride, n, npix, d_prev,
                         seg, d_out);
    else
      hipLaunchKernelGGL((k_hsv_sums<Q, false>), grid, dim3(kBlock), 0, stream, d_frames, frame_stride, n, npix, d_prev,
                         seg, d_out);
    prof_stop(EIOKU_PROF_SCENE_HSV, stream);
  }
  if (npix & 3)
    hipLaunchKernelGGL(k_hsv_sums_tail, dim3(1), dim3(64), 0, stream, d_frames, frame_stride, n,
                       nquads << 2, npix, d_prev, d_out);
  EIOKU_LAUNCH_CHECK();
  if (mem == EIOKU_MEM_HOST) {
    EIOKU_HIP_CHECK(hipMemcpyAsync(sums_out, d_out, sizeof(uint64_t) * 3 * n, hipMemcpyDeviceToHost, stream));
    EIOKU_HIP_CHECK(hipStreamSynchronize(stream));
  }
  return EIOKU_OK;
}

int eioku_scene_sad_luma_bgr(const uint8_t* bgr_frames, int n, int h, int w, size_t frame_stride, const uint8_t* prev,
                             uint64_t* sad_out, int mem, void* stream_) {
  EIOKU_REQUIRE_INIT();
  EIOKU_REQUIRE(n >= 0 && h > 0 && w > 0, "bad shape n=%d h=%d w=%d", n, h, w);
  EIOKU_REQUIRE(mem == EIOKU_MEM_HOST || mem == EIOKU_MEM_DEVICE, "bad mem flag %d", mem);
  EIOKU_REQUIRE(((unsigned long long)h * w) % 4 == 0, "h*w = %llu pixels: must be a multiple of 4", (unsigned long long)h * w);
  const size_t frame_bytes = (size_t)h * w * 3;
  EIOKU_REQUIRE(frame_stride >= frame_bytes || n <= 1, "frame_stride %zu too small", frame_stride);
  if (n == 0) return EIOKU_OK;
  EIOKU_REQUIRE(bgr_frames && sad_out, "NULL buffer");
  hipStream_t stream = (hipStream_t)stream_;
  const uint8_t* d_frames = bgr_frames;
  const uint8_t* d_prev = prev;
  unsigned long long* d_out = (unsigned long long*)sad_out;
  if (mem == EIOKU_MEM_HOST) {
    const size_t total = frame_stride * (size_t)(n - 1) + frame_bytes;
    uint8_t* din = (uint8_t*)scratch(kSlotIn, total);
    d_out = (unsigned long long*)scratch(kSlotOut, sizeof(uint64_t) * n);
    if (!din || !d_out) return EIOKU_ENOMEM;
    EIOKU_HIP_CHECK(hipMemcpyAsync(din, bgr_frames, total, hipMemcpyHostToDevice, stream));
    d_frames = din;
    if (prev) {
      uint8_t* dp = (uint8_t*)scratch(kSlotPrev, frame_bytes);
      if (!dp) return EIOKU_ENOMEM;
      EIOKU_HIP_CHECK(hipMemcpyAsync(dp, prev, frame_bytes, hipMemcpyHostToDevice, stream));
      d_prev = dp;
    }
  }
  EIOKU_HIP_CHECK(hipMemsetAsync(d_out, 0, sizeof(uint64_t) * n, stream));
  constexpr int Q = 4;
  const unsigned long long npix = (unsigned long long)h * w;
  unsigned long long bx = ((npix >> 2) + (unsigned long long)kBlock * Q - 1) / ((unsigned long long)kBlock * Q);
  if (bx == 0) bx = 1;
  const int seg = pick_seg(n, bx);
  dim3 grid((unsigned)bx, (unsigned)((n + seg - 1) / seg));
  const bool aligned = (((uintptr_t)d_frames | frame_stride | (d_prev ? (uintptr_t)d_prev : 0)) & 3) == 0;
  prof_start(EIOKU_PROF_SCENE_SAD, stream);
  if (aligned)
    hipLaunchKernelGGL((k_sad_luma_bgr<Q, true>), grid, dim3(kBlock), 0, stream, d_frames, frame_stride, n, npix, d_prev, seg, d_out);
  else
    hipLaunchKernelGGL((k_sad_luma_bgr<Q, false>), grid, dim3(kBlock), 0, stream, d_frames, frame_stride, n, npix, d_prev, seg, d_out);
  prof_stop(EIOKU_PROF_SCENE_SAD, stream);
  EIOKU_LAUNCH_CHECK();
  if (mem == EIOKU_MEM_HOST) {
    EIOKU_HIP_CHECK(hipMemcpyAsync(sad_out, d_out, sizeof(uint64_t) * n, hipMemcpyDeviceToHost, stream));
    EIOKU_HIP_CHECK(hipStreamSynchronize(stream));
  }
  return EIOKU_OK;
}

int eioku_bgr2hsv(const uint8_t* bgr, size_t n_pixels, uint8_t* hsv_out, int mem, void* stream_) {
  EIOKU_REQUIRE_INIT();
  EIOKU_REQUIRE(mem == EIOKU_MEM_HOST || mem == EIOKU_MEM_DEVICE, "bad mem flag %d", mem);
  if (n_pixels == 0) return EIOKU_OK;
  EIOKU_REQUIRE(bgr && hsv_out, "NULL pointer");
  int rc = ensure_tables();
  if (rc) return rc;
  hipStream_t stream = (hipStream_t)stream_;
  const uint8_t* din = bgr;
  uint8_t* dout = hsv_out;
  if (mem == EIOKU_MEM_HOST) {
    uint8_t* a = (uint8_t*)scratch(kSlotIn, n_pixels * 3);
    uint8_t* b = (uint8_t*)scratch(kSlotOut, n_pixels * 3);
    if (!a || !b) return EIOKU_ENOMEM;
    EIOKU_HIP_CHECK(hipMemcpyAsync(a, bgr, n_pixels * 3, hipMemcpyHostToDevice, stream));
    din = a;
    dout = b;
  }
  unsigned long long g = (n_pixels + kBlock - 1) / kBlock;
  unsigned long long cap = (unsigned long long)num_cus() * 16;
  if (g > cap) g = cap;
  hipLaunchKernelGGL(k_bgr2hsv, dim3((unsigned)g), dim3(kBlock), 0, stream, din,
                     (unsigned long long)n_pixels, dout);
  EIOKU_LAUNCH_CHECK();
  if (mem == EIOKU_MEM_HOST) {
    EIOKU_HIP_CHECK(hipMemcpyAsync(hsv_out, dout, n_pixels * 3, hipMemcpyDeviceToHost, stream));
    EIOKU_HIP_CHECK(hipStreamSynchronize(stream));
  }
  return EIOKU_OK;
}

}  // extern "C"
