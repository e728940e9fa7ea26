// K8: all-MiniLM-L6-v2 style BERT encoder + masked mean pooling + L2 normalisation (gfx950).
//
// The reference holds intent only for this stage (.kiro/specs/semantic-video-search/design.md:54-57,
// 1096-1103); BASELINE.json asks for embeddings within 1e-4 relative of the fp32 CPU path.  gfx950 has no TF32-like
// mode and its exact-fp32 matrix pipe is 157 TFLOP/s, so the default route splits every operand into two bf16 terms
// (v = hi + lo to 2^-18, round to nearest) and runs a.w ~ a_hi.w_hi + a_lo.w_hi + a_hi.w_lo on the bf16 matrix cores with
// fp32 accumulation; LayerNorm / softmax / GELU(erf) are fp32.  Embeddings land at 7-10 % of the 1e-4 bar against the
// float64 oracle (tools/embed_margin.py).  The exact-fp32 kernels stay reachable for the cross-check test
// (EIOKU_GEMM_BF16=0, EIOKU_GEMM_S=0, EIOKU_ATTN_MFMA=0, EIOKU_ATTN_BF16=0).
//
//   k_embed_ln        word + position + token_type(0) embedding gather, LayerNorm -> fp32 x + bf16 hi / lo planes
//   k_gemm_bf_s       C = A . W^T + bias [, GELU] on split bf16: A and W arrive as planes, 128x128 (M >= 8192) or
//                     64x64 tiles, 64-deep register-staged stages, XCD-aware tile order, straight-line epilogue that
//                     writes fp32 or planes
//   k_attention_bf    per (segment, head), S <= 128: S^T = K Q^T and O^T = V^T P^T on split bf16, softmax in registers
//   k_attention_mfma / k_attention8 / k_attention   the exact-fp32 and longer-sequence variants
//   k_add_ln_fixed    LayerNorm(sum of split-K planes + bias + residual) -> fp32 x + planes
//   k_pool_norm       attention-mask weighted mean over tokens, then x / max(|x|, 1e-12)
//   k_gemm_f32*       exact-fp32 MFMA GEMMs (v_mfma_f32_32x32x2_f32): the fallback route
#include "common.h"

#include <cstdlib>
#include <type_traits>

#include <cmath>
#include <string>
#include <vector>

using namespace eioku;

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4b __attribute__((ext_vector_type(4)));
typedef unsigned u32x2b __attribute__((ext_vector_type(2)));

// round-to-nearest-even to bf16, result in the upper 16 bits (finite inputs)
__device__ __forceinline__ unsigned bf16_rne_bits(float v) {
  const unsigned u = __float_as_uint(v);
  return (u + 0x7FFFu + ((u >> 16) & 1u)) & 0xFFFF0000u;
}

// hi = RNE(v) (so |v - hi| <= 2^-9 |v| and the difference is exact in fp32), lo = RNE(v - hi): v = hi + lo up to
// 2^-18 |v|, and the dropped lo.lo products are 2^-18 relative as well -- four times tighter than truncation,
// which left the encoder's output 2.06e-5 off against a 2.0e-5 bar
__device__ __forceinline__ void split_bf16_pair(float a, float b, unsigned& hi, unsigned& lo) {
  const unsigned ha = bf16_rne_bits(a), hb = bf16_rne_bits(b);
  hi = (ha >> 16) | hb;
  lo = (bf16_rne_bits(a - __uint_as_float(ha)) >> 16) | bf16_rne_bits(b - __uint_as_float(hb));
}

// one value -> its two terms (the same split as split_bf16_pair), for the producers that hand a GEMM its A operand as planes
__device__ __forceinline__ void split_bf16_one(float a, unsigned short& hi, unsigned short& lo) {
  const unsigned ha = bf16_rne_bits(a);
  hi = (unsigned short)(ha >> 16);
  lo = (unsigned short)(bf16_rne_bits(a - __uint_as_float(ha)) >> 16);
}

// ---------------------------------------------------------------------------------------------
// LayerNorm helpers: one wave per token row, H <= 1024, H % 64 == 0
// ---------------------------------------------------------------------------------------------
template <int MAXV>
__device__ __forceinline__ void wave_layernorm(float (&x)[MAXV], int nv, int H, const float* __restrict__ g,
                                               const float* __restrict__ b, float eps, float* __restrict__ out,
                                               int lane, unsigned short* __restrict__ ohi = nullptr,
                                               unsigned short* __restrict__ olo = nullptr) {
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < MAXV; ++i)
    if (i < nv) s += x[i];
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
  const float mean = s / (float)H;
  float v = 0.f;
#pragma unroll
  for (int i = 0; i < MAXV; ++i)
    if (i < nv) {
      const float d = x[i] - mean;
      v += d * d;
    }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  const float rstd = 1.0f / sqrtf(v / (float)H + eps);
#pragma unroll
  for (int i = 0; i < MAXV; ++i)
    if (i < nv) {
      const int c = lane + 64 * i;
      const float y = (x[i] - mean) * rstd * g[c] + b[c];
      out[c] = y;
      if (ohi) split_bf16_one(y, ohi[c], olo[c]);  // the next GEMM's A operand, already in its two bf16 terms
    }
}

__global__ __launch_bounds__(256) void k_embed_ln(const int32_t* __restrict__ ids, int T, int S, int H, int vocab,
                                                  const float* __restrict__ wemb, const float* __restrict__ pemb,
                                                  const float* __restrict__ temb, const float* __restrict__ g,
                                                  const float* __restrict__ b, float eps, float* __restrict__ out,
                                                  unsigned short* __restrict__ ohi, unsigned short* __restrict__ olo) {
  const int lane = threadIdx.x & 63;
  const int t = (blockIdx.x * 256 + threadIdx.x) >> 6;
  if (t >= T) return;
  int id = ids[t];
  id = id < 0 ? 0 : (id >= vocab ? vocab - 1 : id);  // checked on the host as well
  const int pos = t % S;
  float x[16];
  const int nv = H / 64;
#pragma unroll
  for (int i = 0; i < 16; ++i)
    if (i < nv) {
      const int c = lane + 64 * i;
      x[i] = (wemb[(size_t)id * H + c] + temb[c]) + pemb[(size_t)pos * H + c];
    }
  wave_layernorm<16>(x, nv, H, g, b, eps, out + (size_t)t * H, lane, ohi ? ohi + (size_t)t * H : nullptr,
                     ohi ? olo + (size_t)t * H : nullptr);
}

// Fixed-width variant (H = 64*NV): every load of the row (up to 4 split-K planes, bias, residual) is issued
// before the first add, so one token costs one memory round trip instead of NV x (nsplit + 2) dependent ones.
template <int NV>
__global__ __launch_bounds__(256) void k_add_ln_fixed(const float* __restrict__ a, int nsplit, const float* __restrict__ abias,
                                                      const float* __restrict__ r, int T, const float* __restrict__ g,
                                                      const float* __restrict__ b, float eps, float* __restrict__ out,
                                                      unsigned short* __restrict__ ohi, unsigned short* __restrict__ olo) {
  constexpr int H = 64 * NV;
  const int lane = threadIdx.x & 63;
  const int t = (blockIdx.x * 256 + threadIdx.x) >> 6;
  if (t >= T) return;
  float pl[4][NV], rv[NV], bv[NV];
#pragma unroll
  for (int sp = 0; sp < 4; ++sp) {
    const int spc = sp < nsplit ? sp : nsplit - 1;  // planes past nsplit: re-read the last one, never added
#pragma unroll
    for (int i = 0; i < NV; ++i) pl[sp][i] = a[((size_t)spc * T + t) * H + lane + 64 * i];
  }
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    rv[i] = r[(size_t)t * H + lane + 64 * i];
    bv[i] = abias ? abias[lane + 64 * i] : 0.f;
  }
  float x[NV];
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    float v = pl[0][i];
#pragma unroll
    for (int sp = 1; sp < 4; ++sp)
      if (sp < nsplit) v += pl[sp][i];
    if (abias) v += bv[i];
    x[i] = v + rv[i];
  }
  wave_layernorm<NV>(x, NV, H, g, b, eps, out + (size_t)t * H, lane, ohi ? ohi + (size_t)t * H : nullptr,
                     ohi ? olo + (size_t)t * H : nullptr);
}

__global__ __launch_bounds__(256) void k_add_ln(const float* __restrict__ a, int nsplit, const float* __restrict__ abias,
                                                const float* __restrict__ r, int T, int H, const float* __restrict__ g,
                                                const float* __restrict__ b, float eps, float* __restrict__ out,
                                                unsigned short* __restrict__ ohi, unsigned short* __restrict__ olo) {
  const int lane = threadIdx.x & 63;
  const int t = (blockIdx.x * 256 + threadIdx.x) >> 6;
  if (t >= T) return;
  float x[16];
  const int nv = H / 64;
#pragma unroll
  for (int i = 0; i < 16; ++i)
    if (i < nv) {
      const int c = lane + 64 * i;
      // a = nsplit split-K partial planes of the producing GEMM (summed in plane order: deterministic) [+ its bias]
      float v = a[(size_t)t * H + c];
      for (int sp = 1; sp < nsplit; ++sp) v += a[((size_t)sp * T + t) * H + c];
      if (abias) v += abias[c];
      x[i] = v + r[(size_t)t * H + c];
    }
  wave_layernorm<16>(x, nv, H, g, b, eps, out + (size_t)t * H, lane, ohi ? ohi + (size_t)t * H : nullptr,
                     ohi ? olo + (size_t)t * H : nullptr);
}

// ---------------------------------------------------------------------------------------------
// GEMM: C[M][N] = A[M][K] . W[N][K]^T + bias[N]  (+ GELU).  N % 128 == 0, K % 32 == 0.
// ---------------------------------------------------------------------------------------------
constexpr int kBK = 32;

// erf for the GELU epilogue: 1 - exp(t q(t) - t) with t = min(|a|, 4) (erf = 1 in fp32 from 3.92) and q a degree-8
// minimax polynomial of (log erfc(t) + t) / t on [0, 4] (tools/fit_erf.py fits it and checks it: GELU(x) within 4.4e-7
// ABSOLUTE of 40-digit values on [-8, 8], i.e. within one fp32 ulp of |x| >= 4 and far inside the path's 1e-4 bar;
// near 0 the RELATIVE error of erf is not ulp-level - 1 - exp(small) - but GELU multiplies it by x / 2).  One
// branch-free form in 14 instructions; the library erff costs ~40, a two-branch 1.7-ulp polynomial 22, and the GELU
// epilogue of the FFN1 GEMM runs 100 M of them per 512-segment batch.
__device__ __forceinline__ float erf_poly(float a) {
  const float t = fminf(fabsf(a), 4.0f);
  float q = 2.050339617e-06f;
  q = fmaf(q, t, -3.688787547e-05f);
  q = fmaf(q, t, 2.615261183e-04f);
  q = fmaf(q, t, -7.679130649e-04f);
  q = fmaf(q, t, -1.059674076e-03f);
  q = fmaf(q, t, 2.006150223e-02f);
  q = fmaf(q, t, -1.031126305e-01f);
  q = fmaf(q, t, -6.365721822e-01f);
  q = fmaf(q, t, -1.283802688e-01f);
  return copysignf(1.0f - __expf(fmaf(q, t, -t)), a);
}

__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.0f + erf_poly(x * 0.70710678118654752440f)); }

// BM x BN output tile per 4-wave workgroup (waves 2 x 2, each (BM/2) x (BN/2) = MI x NI MFMA tiles).
// 128x128 for large M (MFMA-bound); 64x64 when M is small so that a batch of a few segments still
// spreads over the chip instead of serialising K inside 24 workgroups.
template <int EPI, int BM, int BN>  // EPI 0: bias, 1: bias + GELU
__global__ __launch_bounds__(256, 2) void k_gemm_f32(const float* __restrict__ A, int lda, const float* __restrict__ W,
                                                     const float* __restrict__ bias, float* __restrict__ C, int ldc,
                                                     int M, int N, int K, int kchunks) {
  constexpr int WM = BM / 2, WN = BN / 2, MI = WM / 32, NI = WN / 32;
  constexpr int RA = BM / 32, RW = BN / 32;  // staged rows per thread
  __shared__ __attribute__((aligned(16))) float4 sA[2][BM * 8];
  __shared__ __attribute__((aligned(16))) float4 sW[2][BN * 8];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int m0 = blockIdx.x * BM, n0 = blockIdx.y * BN;
  const int half = lane >> 5, l31 = lane & 31;

  f32x16 acc[MI][NI];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NI; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // staging map: unit = tid & 7, rows (tid>>3) + 32*it
  const int sunit = tid & 7, srow = tid >> 3;
  // native vector registers (arrays of the HIP float4 struct can land in scratch) and unconditional loads
  // (rows clamped to M-1: the compiler then counts vmcnt instead of draining the queue at every use)
  f32x4 ra[RA], rw[RW];
  const int kc0 = blockIdx.z * kchunks;  // split-K: this workgroup owns chunks [kc0, kc0 + kchunks)
  auto issue = [&](int kc) {
#pragma unroll
    for (int it = 0; it < RA; ++it) {
      int m = m0 + srow + 32 * it;
      if (m >= M) m = M - 1;
      ra[it] = *reinterpret_cast<const f32x4*>(A + (size_t)m * lda + (kc0 + kc) * kBK + sunit * 4);
    }
#pragma unroll
    for (int it = 0; it < RW; ++it)
      rw[it] = *reinterpret_cast<const f32x4*>(W + (size_t)(n0 + srow + 32 * it) * K + (kc0 + kc) * kBK + sunit * 4);
  };
  auto commit = [&](int buf) {
#pragma unroll
    for (int it = 0; it < RA; ++it) {
      const int row = srow + 32 * it;
      *reinterpret_cast<f32x4*>(&sA[buf][row * 8 + (sunit ^ ((row >> 1) & 7))]) = ra[it];
    }
#pragma unroll
    for (int it = 0; it < RW; ++it) {
      const int row = srow + 32 * it;
      *reinterpret_cast<f32x4*>(&sW[buf][row * 8 + (sunit ^ ((row >> 1) & 7))]) = rw[it];
    }
  };

  const int nk = kchunks;
  C += (size_t)blockIdx.z * M * ldc;  // split-K partial planes; the consumer (k_add_ln) sums them and adds the bias
  issue(0);
  commit(0);
  __syncthreads();
  for (int kc = 0; kc < nk; ++kc) {
    const int buf = kc & 1;
    if (kc + 1 < nk) issue(kc + 1);
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      float4 a[MI], w[NI];
#pragma unroll
      for (int i = 0; i < MI; ++i) {
        const int row = wm * WM + i * 32 + l31;
        a[i] = sA[buf][row * 8 + ((2 * t + half) ^ ((row >> 1) & 7))];
      }
#pragma unroll
      for (int j = 0; j < NI; ++j) {
        const int row = wn * WN + j * 32 + l31;
        w[j] = sW[buf][row * 8 + ((2 * t + half) ^ ((row >> 1) & 7))];
      }
#pragma unroll
      for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j) {
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].x, w[j].x, acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].y, w[j].y, acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].z, w[j].z, acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].w, w[j].w, acc[i][j], 0, 0, 0);
        }
    }
    if (kc + 1 < nk) commit(buf ^ 1);
    __syncthreads();
  }
  // epilogue: lane = column n, registers = rows m
#pragma unroll
  for (int j = 0; j < NI; ++j) {
    const int n = n0 + wn * WN + j * 32 + l31;
    const float bv = bias ? bias[n] : 0.f;
#pragma unroll
    for (int i = 0; i < MI; ++i) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = m0 + wm * WM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
        if (m < M) {
          float v = acc[i][j][r] + bv;
          if (EPI == 1) v = gelu_erf(v);
          C[(size_t)m * ldc + n] = v;
        }
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------
// attention: grid (B, heads), PARTS lanes per query (each owns the keys j = part mod PARTS), block =
// PARTS*S threads rounded to 64, head_dim == 32.  qkv: [B*S][3H] = Q | K | V; ctx: [B*S][H]
// ---------------------------------------------------------------------------------------------
// S <= 128: the whole (segment, head) on the exact-fp32 matrix core, one workgroup of 4 waves x 32 queries.
// The scores are computed TRANSPOSED, S^T = K . Q^T (v_mfma_f32_32x32x2_f32: A = 32 keys x 2 dims, B = 2 dims x 32
// queries), so that a lane ends up with one query (its column) and 64 of that query's 128 keys in its accumulator
// registers, the other 64 in lane ^ 32: the softmax reductions are in-register plus ONE cross-lane exchange.  And
// the accumulator registers are then already the B operand of O^T = V^T . P^T -- the k axis of an MFMA may be
// summed in any order, so step r pairs exactly the two keys that register r holds in the two half-waves: no
// transposition, no LDS round trip for P.  128 MFMAs per wave instead of ~12 k scalar FMAs per lane: 21 -> ~8 us.
// Q / K columns are XOR-swizzled by the row in LDS (a column read is then conflict-free); rows >= S are zero, mask 0.
__global__ __launch_bounds__(256) void k_attention_mfma(const float* __restrict__ qkv, const uint8_t* __restrict__ mask, int S,
                                                        int H, float* __restrict__ ctx, unsigned short* __restrict__ chi,
                                                        unsigned short* __restrict__ clo) {
  constexpr int SP = 128, QS = 32;
  __shared__ __attribute__((aligned(16))) float sQ[SP * QS];
  __shared__ __attribute__((aligned(16))) float sK[SP * QS];
  __shared__ float sV[SP * 32];
  __shared__ float sM[SP];
  const int b = blockIdx.x, h = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const size_t row0 = (size_t)b * S;
  for (int i0 = 0; i0 < SP * 8; i0 += 4 * 256) {
    f32x4 qq[4], kk[4], vv[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int i = i0 + t * 256 + tid;
      const int j = (i >> 3) < S ? (i >> 3) : S - 1;
      const float* base = qkv + (row0 + j) * (size_t)(3 * H) + h * 32 + (i & 7) * 4;
      qq[t] = *reinterpret_cast<const f32x4*>(base);
      kk[t] = *reinterpret_cast<const f32x4*>(base + H);
      vv[t] = *reinterpret_cast<const f32x4*>(base + 2 * H);
    }
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int i = i0 + t * 256 + tid;
      const int j = i >> 3, u = i & 7;
      const bool in = j < S;
      // element (j, d) of Q / K lives at column d ^ (j & 31): a column read over 32 consecutive rows then touches 32
      // different banks, and an aligned group of 4 dims stays one (permuted) 16-byte store
      const int pj = j & 3, gu = (u ^ ((j >> 2) & 7)) * 4;
      f32x4 qp, kp;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        qp[e] = in ? qq[t][e ^ pj] : 0.f;
        kp[e] = in ? kk[t][e ^ pj] : 0.f;
      }
      *reinterpret_cast<f32x4*>(&sQ[j * QS + gu]) = qp;
      *reinterpret_cast<f32x4*>(&sK[j * QS + gu]) = kp;
      *reinterpret_cast<f32x4*>(&sV[j * 32 + u * 4]) = in ? vv[t] : f32x4{0.f, 0.f, 0.f, 0.f};
    }
  }
  for (int j = tid; j < SP; j += 256) sM[j] = (j < S && mask[row0 + j]) ? 1.f : 0.f;
  __syncthreads();
  const int c = lane & 31, half = lane >> 5, q0 = wave * 32;
  if (q0 >= S) return;  // a whole wave of padding queries (no barrier below)
  float bq[16];
#pragma unroll
  for (int t = 0; t < 16; ++t) bq[t] = sQ[(q0 + c) * QS + ((2 * t + half) ^ c)];  // (q0 + c) & 31 == c
  f32x16 sc[4];
#pragma unroll
  for (int kt = 0; kt < 4; ++kt) {
#pragma unroll
    for (int r = 0; r < 16; ++r) sc[kt][r] = 0.f;
#pragma unroll
    for (int t = 0; t < 16; ++t)
      sc[kt] = __builtin_amdgcn_mfma_f32_32x32x2f32(sK[(kt * 32 + c) * QS + ((2 * t + half) ^ c)], bq[t], sc[kt], 0, 0, 0);
  }
  // sc[kt][r] = <k_key, q_query>, key = kt*32 + (r & 3) + 8*(r >> 2) + 4*half, query = q0 + c
  const float rinv = 0.17677669529663687f;  // 1 / sqrt(32)
  float mx = -INFINITY;
#pragma unroll
  for (int kt = 0; kt < 4; ++kt)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int key = kt * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
      const float s = sM[key] != 0.f ? sc[kt][r] * rinv : -INFINITY;
      sc[kt][r] = s;
      mx = fmaxf(mx, s);
    }
  mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
  float l = 0.f;
#pragma unroll
  for (int kt = 0; kt < 4; ++kt)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float e = sc[kt][r] != -INFINITY ? expf(sc[kt][r] - mx) : 0.f;
      sc[kt][r] = e;
      l += e;
    }
  l += __shfl_xor(l, 32, 64);
  f32x16 o;
#pragma unroll
  for (int r = 0; r < 16; ++r) o[r] = 0.f;
#pragma unroll
  for (int kt = 0; kt < 4; ++kt)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int key = kt * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
      o = __builtin_amdgcn_mfma_f32_32x32x2f32(sV[key * 32 + c], sc[kt][r], o, 0, 0, 0);
    }
  // o[r] = O[query][dim], dim = (r & 3) + 8*(r >> 2) + 4*half
  const int qi = q0 + c;
  if (qi >= S) return;
  const float rl = l > 0.f ? 1.0f / l : 0.f;  // fully masked segment -> zeros (its pooled vector is 0 anyway)
  const size_t o0 = (row0 + qi) * (size_t)H + h * 32;
  if (chi) {  // the out-projection GEMM is the only reader: hand it the two bf16 terms instead of fp32 (same bytes)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      unsigned h0, l0, h1, l1;
      split_bf16_pair(o[4 * g] * rl, o[4 * g + 1] * rl, h0, l0);
      split_bf16_pair(o[4 * g + 2] * rl, o[4 * g + 3] * rl, h1, l1);
      *reinterpret_cast<u32x2b*>(chi + o0 + 8 * g + 4 * half) = u32x2b{h0, h1};
      *reinterpret_cast<u32x2b*>(clo + o0 + 8 * g + 4 * half) = u32x2b{l0, l1};
    }
    return;
  }
  float* op = ctx + o0;
#pragma unroll
  for (int g = 0; g < 4; ++g)
    *reinterpret_cast<f32x4*>(op + 8 * g + 4 * half) = f32x4{o[4 * g] * rl, o[4 * g + 1] * rl, o[4 * g + 2] * rl, o[4 * g + 3] * rl};
}

// r3: the same (segment, head) attention on split bf16 (v = hi + lo, three product terms, fp32 accumulate - the GEMMs'
// scheme).  The fp32 matrix pipe (64 cycles per 32x32x2 step) made k_attention_mfma matrix-bound: 8192 MFMA cycles per
// wave; here QK^T and PV are 24 + 24 v_mfma_f32_32x32x16_bf16 = 1536 cycles.  Same structure: scores transposed
// (S^T = K Q^T), so a lane owns one query and the softmax is in-register plus one cross-lane exchange, and the
// accumulator registers of a key tile are, eight at a time, the B operand of O^T = V^T P^T: registers 8g..8g+7 of the two
// half-waves hold the 16 keys 16g..16g+15 of the tile, and V^T sits in LDS with its keys in exactly that order, so that a
// lane's A operand is one 16-byte read.  LDS: Q / K rows of 64 B (4 slots, slot ^= (row >> 2) & 3), V^T rows of 256 B
// (16 slots, slot ^= dim & 15): every ds_read_b128 lane group touches 16 different bank quads.
__global__ __launch_bounds__(256) void k_attention_bf(const float* __restrict__ qkv, const uint8_t* __restrict__ mask, int S,
                                                      int H, float* __restrict__ ctx, unsigned short* __restrict__ chi,
                                                      unsigned short* __restrict__ clo) {
  constexpr int SP = 128;
  __shared__ __attribute__((aligned(16))) u32x4b sQ[2][SP * 4];   // [plane][row][4 slots of 8 dims]
  __shared__ __attribute__((aligned(16))) u32x4b sK[2][SP * 4];
  __shared__ __attribute__((aligned(16))) unsigned short sVt[2][32 * SP];  // [plane][dim][128 permuted keys]
  __shared__ float sM[SP];
  const int b = blockIdx.x, h = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const size_t row0 = (size_t)b * S;
  // staging: thread -> (row i >> 3, 4-dim group i & 7), 4 passes of 256 threads cover 128 rows x 8 groups
  for (int i0 = 0; i0 < SP * 8; i0 += 4 * 256) {
    f32x4 qq[4], kk[4], vv[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int i = i0 + t * 256 + tid;
      const int j = (i >> 3) < S ? (i >> 3) : S - 1;
      const float* base = qkv + (row0 + j) * (size_t)(3 * H) + h * 32 + (i & 7) * 4;
      qq[t] = *reinterpret_cast<const f32x4*>(base);
      kk[t] = *reinterpret_cast<const f32x4*>(base + H);
      vv[t] = *reinterpret_cast<const f32x4*>(base + 2 * H);
    }
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int i = i0 + t * 256 + tid;
      const int j = i >> 3, u = i & 7;
      const bool in = j < S;
      unsigned qh0, ql0, qh1, ql1, kh0, kl0, kh1, kl1;
      split_bf16_pair(in ? qq[t][0] : 0.f, in ? qq[t][1] : 0.f, qh0, ql0);
      split_bf16_pair(in ? qq[t][2] : 0.f, in ? qq[t][3] : 0.f, qh1, ql1);
      split_bf16_pair(in ? kk[t][0] : 0.f, in ? kk[t][1] : 0.f, kh0, kl0);
      split_bf16_pair(in ? kk[t][2] : 0.f, in ? kk[t][3] : 0.f, kh1, kl1);
      // dims 4u..4u+3 = half (u & 1) of slot u >> 1 of row j
      const int slot = (u >> 1) ^ ((j >> 2) & 3), off = (j * 4 + slot) * 16 + (u & 1) * 8;
      *reinterpret_cast<u32x2b*>(reinterpret_cast<unsigned char*>(sQ[0]) + off) = u32x2b{qh0, qh1};
      *reinterpret_cast<u32x2b*>(reinterpret_cast<unsigned char*>(sQ[1]) + off) = u32x2b{ql0, ql1};
      *reinterpret_cast<u32x2b*>(reinterpret_cast<unsigned char*>(sK[0]) + off) = u32x2b{kh0, kh1};
      *reinterpret_cast<u32x2b*>(reinterpret_cast<unsigned char*>(sK[1]) + off) = u32x2b{kl0, kl1};
      // V^T: key j -> position kt*32 + g*16 + half*8 + i8 with x = j & 15 = (i8 & 3) + 8 (i8 >> 2) + 4 half
      const int x = j & 15, pos = (j & ~15) + ((x >> 2) & 1) * 8 + (x & 3) + 4 * (x >> 3);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int d = u * 4 + e;
        unsigned short vh, vl;
        split_bf16_one(in ? vv[t][e] : 0.f, vh, vl);
        const int p = d * SP + (((pos >> 3) ^ (d & 15)) << 3) + (pos & 7);
        sVt[0][p] = vh;
        sVt[1][p] = vl;
      }
    }
  }
  for (int j = tid; j < SP; j += 256) sM[j] = (j < S && mask[row0 + j]) ? 1.f : 0.f;
  __syncthreads();
  const int c = lane & 31, half = lane >> 5, q0 = wave * 32;
  if (q0 >= S) return;  // a whole wave of padding queries (no barrier below)
  // this wave's queries as B operands: 2 k-steps x (hi, lo)
  bf16x8 bqh[2], bql[2];
  {
    const int qr = q0 + c;
#pragma unroll
    for (int st = 0; st < 2; ++st) {
      const int slot = (st * 2 + half) ^ ((qr >> 2) & 3);
      bqh[st] = __builtin_bit_cast(bf16x8, sQ[0][qr * 4 + slot]);
      bql[st] = __builtin_bit_cast(bf16x8, sQ[1][qr * 4 + slot]);
    }
  }
  f32x16 sc[4];
#pragma unroll
  for (int kt = 0; kt < 4; ++kt) {
#pragma unroll
    for (int r = 0; r < 16; ++r) sc[kt][r] = 0.f;
    const int kr = kt * 32 + c;
#pragma unroll
    for (int st = 0; st < 2; ++st) {
      const int slot = (st * 2 + half) ^ ((kr >> 2) & 3);
      const bf16x8 akh = __builtin_bit_cast(bf16x8, sK[0][kr * 4 + slot]);
      const bf16x8 akl = __builtin_bit_cast(bf16x8, sK[1][kr * 4 + slot]);
      sc[kt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(akh, bqh[st], sc[kt], 0, 0, 0);
      sc[kt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(akl, bqh[st], sc[kt], 0, 0, 0);
      sc[kt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(akh, bql[st], sc[kt], 0, 0, 0);
    }
  }
  // sc[kt][r] = <k_key, q_query>, key = kt*32 + (r & 3) + 8*(r >> 2) + 4*half, query = q0 + c
  const float rinv = 0.17677669529663687f;  // 1 / sqrt(32)
  float mx = -INFINITY;
#pragma unroll
  for (int kt = 0; kt < 4; ++kt)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int key = kt * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
      const float s = sM[key] != 0.f ? sc[kt][r] * rinv : -INFINITY;
      sc[kt][r] = s;
      mx = fmaxf(mx, s);
    }
  mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
  float l = 0.f;
#pragma unroll
  for (int kt = 0; kt < 4; ++kt)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float e = sc[kt][r] != -INFINITY ? expf(sc[kt][r] - mx) : 0.f;
      sc[kt][r] = e;
      l += e;
    }
  l += __shfl_xor(l, 32, 64);
  f32x16 o;
#pragma unroll
  for (int r = 0; r < 16; ++r) o[r] = 0.f;
#pragma unroll
  for (int kt = 0; kt < 4; ++kt)
#pragma unroll
    for (int g = 0; g < 2; ++g) {
      // P^T operand: this lane's 8 probabilities of key block (kt, g), split into their two bf16 terms
      unsigned ph[4], pl[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) split_bf16_pair(sc[kt][8 * g + 2 * e], sc[kt][8 * g + 2 * e + 1], ph[e], pl[e]);
      const bf16x8 bph = __builtin_bit_cast(bf16x8, u32x4b{ph[0], ph[1], ph[2], ph[3]});
      const bf16x8 bpl = __builtin_bit_cast(bf16x8, u32x4b{pl[0], pl[1], pl[2], pl[3]});
      // V^T operand: dim c, keys of the same block and half, one 16-byte unit
      const int unit = ((kt * 4 + g * 2 + half) ^ (c & 15));
      const bf16x8 avh = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4b*>(&sVt[0][c * SP + unit * 8]));
      const bf16x8 avl = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4b*>(&sVt[1][c * SP + unit * 8]));
      o = __builtin_amdgcn_mfma_f32_32x32x16_bf16(avh, bph, o, 0, 0, 0);
      o = __builtin_amdgcn_mfma_f32_32x32x16_bf16(avl, bph, o, 0, 0, 0);
      o = __builtin_amdgcn_mfma_f32_32x32x16_bf16(avh, bpl, o, 0, 0, 0);
    }
  // o[r] = O[query][dim], dim = (r & 3) + 8*(r >> 2) + 4*half
  const int qi = q0 + c;
  if (qi >= S) return;
  const float rl = l > 0.f ? 1.0f / l : 0.f;  // fully masked segment -> zeros (its pooled vector is 0 anyway)
  const size_t o0 = (row0 + qi) * (size_t)H + h * 32;
  if (chi) {
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      unsigned h0, l0, h1, l1;
      split_bf16_pair(o[4 * g] * rl, o[4 * g + 1] * rl, h0, l0);
      split_bf16_pair(o[4 * g + 2] * rl, o[4 * g + 3] * rl, h1, l1);
      *reinterpret_cast<u32x2b*>(chi + o0 + 8 * g + 4 * half) = u32x2b{h0, h1};
      *reinterpret_cast<u32x2b*>(clo + o0 + 8 * g + 4 * half) = u32x2b{l0, l1};
    }
    return;
  }
  float* op = ctx + o0;
#pragma unroll
  for (int g = 0; g < 4; ++g)
    *reinterpret_cast<f32x4*>(op + 8 * g + 4 * half) = f32x4{o[4 * g] * rl, o[4 * g + 1] * rl, o[4 * g + 2] * rl, o[4 * g + 3] * rl};
}

// S <= 256: 8 lanes per query (keys j = lane, lane + 8, ...), 32 queries per workgroup.  Each lane keeps its <= 32
// scores in registers (the old kernel computed every score twice: once for the max, once for the exponent) and
// the K / V rows are swizzled by row so that the 8 rows a query group reads at once hit 8 different bank groups.
__global__ __launch_bounds__(256) void k_attention8(const float* __restrict__ qkv, const uint8_t* __restrict__ mask, int S,
                                                    int H, float* __restrict__ ctx) {
  extern __shared__ __attribute__((aligned(16))) float sm8[];
  float4* sK = reinterpret_cast<float4*>(sm8);               // [S][8], unit u of row j at j*8 + (u ^ (j & 7))
  float4* sV = sK + (size_t)S * 8;
  float* sM = reinterpret_cast<float*>(sV + (size_t)S * 8);  // [S] 1/0
  const int b = blockIdx.x, h = blockIdx.y, tid = threadIdx.x;
  const size_t row0 = (size_t)b * S;
  for (int i0 = 0; i0 < S * 8; i0 += 4 * 256) {
    f32x4 kk[4], vv[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      int i = i0 + t * 256 + tid;
      if (i >= S * 8) i = S * 8 - 1;
      const float* base = qkv + (row0 + (i >> 3)) * (size_t)(3 * H) + h * 32 + (i & 7) * 4;
      kk[t] = *reinterpret_cast<const f32x4*>(base + H);
      vv[t] = *reinterpret_cast<const f32x4*>(base + 2 * H);
    }
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int i = i0 + t * 256 + tid;
      if (i < S * 8) {
        const int j = i >> 3, u = (i & 7) ^ (j & 7);
        *reinterpret_cast<f32x4*>(&sK[j * 8 + u]) = kk[t];
        *reinterpret_cast<f32x4*>(&sV[j * 8 + u]) = vv[t];
      }
    }
  }
  for (int j = tid; j < S; j += 256) sM[j] = mask[row0 + j] ? 1.f : 0.f;
  __syncthreads();
  const int qi = blockIdx.z * 32 + (tid >> 3), part = tid & 7;
  const bool live = qi < S;
  float q[32];
  {
    const float* qp = qkv + (row0 + (live ? qi : 0)) * (size_t)(3 * H) + h * 32;
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const f32x4 v = *reinterpret_cast<const f32x4*>(qp + u * 4);
      q[u * 4] = v[0]; q[u * 4 + 1] = v[1]; q[u * 4 + 2] = v[2]; q[u * 4 + 3] = v[3];
    }
  }
  const float rinv = 0.17677669529663687f;  // 1 / sqrt(32)
  constexpr int MAXJ = 32;
  float sc[MAXJ];
  float mx = -INFINITY;
#pragma unroll
  for (int jj = 0; jj < MAXJ; ++jj) {
    const int j = part + 8 * jj;
    sc[jj] = -INFINITY;
    if (j < S) {  // uniform over the wave for all but the last partial group
      float s = 0.f;
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const float4 k = sK[j * 8 + (u ^ (j & 7))];
        s += q[u * 4] * k.x;
        s += q[u * 4 + 1] * k.y;
        s += q[u * 4 + 2] * k.z;
        s += q[u * 4 + 3] * k.w;
      }
      if (sM[j] != 0.f) sc[jj] = s * rinv;
      mx = fmaxf(mx, sc[jj]);
    }
  }
#pragma unroll
  for (int off = 1; off < 8; off <<= 1) mx = fmaxf(mx, __shfl_xor(mx, off, 64));
  float acc[32];
#pragma unroll
  for (int c = 0; c < 32; ++c) acc[c] = 0.f;
  float l = 0.f;
#pragma unroll
  for (int jj = 0; jj < MAXJ; ++jj) {
    const int j = part + 8 * jj;
    if (j < S && sc[jj] != -INFINITY) {
      const float e = expf(sc[jj] - mx);
      l += e;
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const float4 v = sV[j * 8 + (u ^ (j & 7))];
        acc[u * 4] += e * v.x;
        acc[u * 4 + 1] += e * v.y;
        acc[u * 4 + 2] += e * v.z;
        acc[u * 4 + 3] += e * v.w;
      }
    }
  }
#pragma unroll
  for (int off = 1; off < 8; off <<= 1) {
    l += __shfl_xor(l, off, 64);
#pragma unroll
    for (int c = 0; c < 32; ++c) acc[c] += __shfl_xor(acc[c], off, 64);
  }
  if (!live || part != 0) return;
  float* op = ctx + (row0 + qi) * (size_t)H + h * 32;
  const float rl = l > 0.f ? 1.0f / l : 0.f;  // fully masked segment -> zeros (its pooled vector is 0 anyway)
#pragma unroll
  for (int u = 0; u < 8; ++u)
    *reinterpret_cast<f32x4*>(op + u * 4) = f32x4{acc[u * 4] * rl, acc[u * 4 + 1] * rl, acc[u * 4 + 2] * rl, acc[u * 4 + 3] * rl};
}

template <int PARTS>
__global__ __launch_bounds__(1024) void k_attention(const float* __restrict__ qkv, const uint8_t* __restrict__ mask,
                                                    int S, int H, float* __restrict__ ctx) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  float4* sK = reinterpret_cast<float4*>(sm);       // [S][8]
  float4* sV = sK + (size_t)S * 8;                  // [S][8]
  float* sM = reinterpret_cast<float*>(sV + (size_t)S * 8);  // [S] 1/0
  const int b = blockIdx.x, h = blockIdx.y, tid = threadIdx.x;
  const size_t row0 = (size_t)b * S;
  // K and V rows of this (segment, head) -> LDS, 4 + 4 loads in flight per thread
  for (int i0 = 0; i0 < S * 8; i0 += 4 * blockDim.x) {
    f32x4 kk[4], vv[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      int i = i0 + t * blockDim.x + tid;
      if (i >= S * 8) i = S * 8 - 1;
      const float* base = qkv + (row0 + (i >> 3)) * (size_t)(3 * H) + h * 32 + (i & 7) * 4;
      kk[t] = *reinterpret_cast<const f32x4*>(base + H);
      vv[t] = *reinterpret_cast<const f32x4*>(base + 2 * H);
    }
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int i = i0 + t * blockDim.x + tid;
      if (i < S * 8) {
        *reinterpret_cast<f32x4*>(&sK[i]) = kk[t];
        *reinterpret_cast<f32x4*>(&sV[i]) = vv[t];
      }
    }
  }
  for (int j = tid; j < S; j += blockDim.x) sM[j] = mask[row0 + j] ? 1.f : 0.f;
  __syncthreads();
  // queries are split over blockIdx.z as well: (segment, head) alone is B x heads ~ 100 workgroups for a batch
  // of 8 segments, a third of the chip
  const int qi = blockIdx.z * (blockDim.x / PARTS) + tid / PARTS, part = tid % PARTS;
  const bool live = qi < S;  // lanes of a query group stay together for the shuffles below
  float q[32];
  {
    const float* qp = qkv + (row0 + (live ? qi : 0)) * (size_t)(3 * H) + h * 32;
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const float4 v = *reinterpret_cast<const float4*>(qp + u * 4);
      q[u * 4] = v.x; q[u * 4 + 1] = v.y; q[u * 4 + 2] = v.z; q[u * 4 + 3] = v.w;
    }
  }
  const float rinv = 0.17677669529663687f;  // 1 / sqrt(32): scores / sqrt(head_size)
  auto score = [&](int j) {
    float s = 0.f;
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const float4 k = sK[j * 8 + u];
      s += q[u * 4] * k.x;
      s += q[u * 4 + 1] * k.y;
      s += q[u * 4 + 2] * k.z;
      s += q[u * 4 + 3] * k.w;
    }
    return s * rinv;
  };
  float mx = -INFINITY;
  for (int j = part; j < S; j += PARTS)
    if (sM[j] != 0.f) mx = fmaxf(mx, score(j));
#pragma unroll
  for (int off = 1; off < PARTS; off <<= 1) mx = fmaxf(mx, __shfl_xor(mx, off, 64));
  float acc[32];
#pragma unroll
  for (int c = 0; c < 32; ++c) acc[c] = 0.f;
  float l = 0.f;
  for (int j = part; j < S; j += PARTS) {
    if (sM[j] == 0.f) continue;
    const float e = expf(score(j) - mx);
    l += e;
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const float4 v = sV[j * 8 + u];
      acc[u * 4] += e * v.x;
      acc[u * 4 + 1] += e * v.y;
      acc[u * 4 + 2] += e * v.z;
      acc[u * 4 + 3] += e * v.w;
    }
  }
#pragma unroll
  for (int off = 1; off < PARTS; off <<= 1) {
    l += __shfl_xor(l, off, 64);
#pragma unroll
    for (int c = 0; c < 32; ++c) acc[c] += __shfl_xor(acc[c], off, 64);
  }
  if (!live || part != 0) return;
  float* op = ctx + (row0 + qi) * (size_t)H + h * 32;
  const float rl = l > 0.f ? 1.0f / l : 0.f;  // fully masked segment -> zeros (its pooled vector is 0 anyway)
#pragma unroll
  for (int u = 0; u < 8; ++u)
    *reinterpret_cast<float4*>(op + u * 4) =
        make_float4(acc[u * 4] * rl, acc[u * 4 + 1] * rl, acc[u * 4 + 2] * rl, acc[u * 4 + 3] * rl);
}

// mean pooling (mask weighted) + L2 normalise; one block (H threads) per segment
__global__ __launch_bounds__(1024) void k_pool_norm(const float* __restrict__ x, const uint8_t* __restrict__ mask,
                                                    int S, int H, float* __restrict__ out) {
  // blockDim.x = G * Hp (Hp = H rounded up to 64, G = 1 or 2 token groups): group g sums tokens g, g + G, ...
  // with 32 loads in flight per thread; the mask row sits in LDS (it used to be re-loaded by every thread)
  __shared__ float red[16];
  __shared__ float smask[1024];
  __shared__ float psum[1024];
  const int b = blockIdx.x, tid = threadIdx.x;
  const int Hp = (H + 63) & ~63, G = blockDim.x / Hp;
  const int c = tid % Hp, g = tid / Hp;
  for (int t = tid; t < S; t += blockDim.x) smask[t] = mask[(size_t)b * S + t] ? 1.f : 0.f;
  __syncthreads();
  float s = 0.f;
  const int cc = c < H ? c : H - 1;
  constexpr int PB = 32;
  for (int t0 = g; t0 < S; t0 += PB * G) {
    float xv[PB];
#pragma unroll
    for (int i = 0; i < PB; ++i) {
      const int t = t0 + i * G < S ? t0 + i * G : S - 1;
      xv[i] = x[((size_t)b * S + t) * H + cc];
    }
#pragma unroll
    for (int i = 0; i < PB; ++i)
      if (t0 + i * G < S) s += xv[i] * smask[t0 + i * G];
  }
  if (g > 0) psum[(g - 1) * Hp + c] = s;
  __syncthreads();
  float v = 0.f;
  if (g == 0) {
    for (int k = 1; k < G; ++k) s += psum[(k - 1) * Hp + c];  // group order: deterministic
    float cnt = 0.f;
    for (int t = 0; t < S; ++t) cnt += smask[t];
    v = c < H ? s / fmaxf(cnt, 1e-9f) : 0.f;
  }
  float sq = v * v;  // zero in the helper groups
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) sq += __shfl_xor(sq, off, 64);
  if ((tid & 63) == 0) red[tid >> 6] = sq;
  __syncthreads();
  float tot = 0.f;
  for (int w = 0; w < (int)(blockDim.x >> 6); ++w) tot += red[w];
  const float nrm = fmaxf(sqrtf(tot), 1e-12f);
  if (g == 0 && c < H) out[(size_t)b * H + c] = v / nrm;
}

}  // namespace

// ---------------------------------------------------------------------------------------------
// handle
// ---------------------------------------------------------------------------------------------
struct eioku_bert {
  int vocab, H, L, heads, ffn, max_pos, type_vocab;
  float eps;
  struct Tensor {
    std::string name;
    std::vector<int> shape;
    float* dev = nullptr;
    bool set = false;
    size_t numel() const {
      size_t n = 1;
      for (int s : shape) n *= s;
      return n;
    }
  };
  std::vector<Tensor> tensors;
  // workspace
  int32_t* d_ids = nullptr; size_t ids_cap = 0;
  uint8_t* d_mask = nullptr; size_t mask_cap = 0;
  float* x = nullptr; float* y = nullptr; float* qkv = nullptr; float* ctx = nullptr; float* mid = nullptr;
  size_t x_cap = 0, y_cap = 0, qkv_cap = 0, ctx_cap = 0, mid_cap = 0;
  float* d_out = nullptr; size_t out_cap = 0;
  // activations as bf16 hi | lo planes for the GEMMs that read them (x, ctx, mid): [2][T][width] each
  unsigned short* xp = nullptr; unsigned short* cp = nullptr; unsigned short* mp = nullptr;
  size_t xp_cap = 0, cp_cap = 0, mp_cap = 0;
  double flops_last = 0;
  // GEMM weights pre-split into bf16 hi / lo planes (same RNE split the kernel applies on the fly), per layer
  // [qkv | out | ffn1 | ffn2]; rebuilt lazily after set_tensor
  struct SplitW {
    unsigned short* hi = nullptr;
    unsigned short* lo = nullptr;
  };
  std::vector<SplitW> wsplit;  // 4 per layer
  bool wsplit_dirty = true;
};

namespace {

template <typename T>
int grow(T** p, size_t* cap, size_t bytes) {
  if (*cap >= bytes) return EIOKU_OK;
  if (*p) {
    (void)hipDeviceSynchronize();
    (void)hipFree(*p);
    *p = nullptr;
    *cap = 0;
  }
  EIOKU_HIP_CHECK(hipMalloc((void**)p, bytes));
  *cap = bytes;
  return EIOKU_OK;
}

int find(const eioku_bert* m, const std::string& name) {
  for (size_t i = 0; i < m->tensors.size(); ++i)
    if (m->tensors[i].name == name) return (int)i;
  return -1;
}

const float* tp(const eioku_bert* m, const std::string& name) { return m->tensors[find(m, name)].dev; }

// Small-M GEMM (M < 8192: the ingest path's 8 segments = 1024 tokens).  With 32-deep k-chunks a 64x64 tile is a
// chain of K/32 dependent load -> LDS -> MFMA rounds (12 for K = 384) and a launch is ~100-400 workgroups, so the
// chain length IS the kernel's duration.  Here a stage is 128 deep: 16 loads in flight per thread, 64 MFMAs per
// wave between barriers, 3 stages for K = 384; one LDS buffer (64 KB) + register prefetch keeps 2 workgroups/CU.
template <int EPI>
__global__ __launch_bounds__(256, 2) void k_gemm_f32_s(const float* __restrict__ A, int lda, const float* __restrict__ W,
                                                       const float* __restrict__ bias, float* __restrict__ C, int ldc,
                                                       int M, int N, int K, int kstages) {
  constexpr int BM = 64, BN = 64, BKS = 128, UPR = BKS / 4;  // units (float4) per staged row
  constexpr int RA = BM * UPR / 256, RW = BN * UPR / 256;    // 8 + 8 float4 per thread and stage
  extern __shared__ __attribute__((aligned(16))) float4 sg[];
  float4* sA = sg;               // [BM][UPR], unit index swizzled with the row
  float4* sW = sg + BM * UPR;    // [BN][UPR]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int m0 = blockIdx.x * BM, n0 = blockIdx.y * BN;
  const int half = lane >> 5, l31 = lane & 31;
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  // staging map: unit = tid & 31, rows (tid >> 5) + 8 * it
  const int sunit = tid & 31, srow = tid >> 5;
  const int k0 = blockIdx.z * kstages * BKS;
  f32x4 ra[RA], rw[RW];
  auto issue = [&](int st) {
#pragma unroll
    for (int it = 0; it < RA; ++it) {
      int m = m0 + srow + 8 * it;
      if (m >= M) m = M - 1;
      ra[it] = *reinterpret_cast<const f32x4*>(A + (size_t)m * lda + k0 + st * BKS + sunit * 4);
    }
#pragma unroll
    for (int it = 0; it < RW; ++it)
      rw[it] = *reinterpret_cast<const f32x4*>(W + (size_t)(n0 + srow + 8 * it) * K + k0 + st * BKS + sunit * 4);
  };
  auto commit = [&]() {
#pragma unroll
    for (int it = 0; it < RA; ++it) {
      const int row = srow + 8 * it;
      *reinterpret_cast<f32x4*>(&sA[row * UPR + ((sunit & 16) | ((sunit ^ row) & 15))]) = ra[it];
    }
#pragma unroll
    for (int it = 0; it < RW; ++it) {
      const int row = srow + 8 * it;
      *reinterpret_cast<f32x4*>(&sW[row * UPR + ((sunit & 16) | ((sunit ^ row) & 15))]) = rw[it];
    }
  };
  C += (size_t)blockIdx.z * M * ldc;
  issue(0);
  for (int st = 0; st < kstages; ++st) {
    commit();
    __syncthreads();
    if (st + 1 < kstages) issue(st + 1);
    const int arow = wm * 32 + l31, wrow = wn * 32 + l31;
#pragma unroll
    for (int t = 0; t < UPR / 2; ++t) {
      const int u = 2 * t + half;
      const float4 a = sA[arow * UPR + ((u & 16) | ((u ^ arow) & 15))];
      const float4 w = sW[wrow * UPR + ((u & 16) | ((u ^ wrow) & 15))];
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, w.x, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, w.y, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, w.z, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, w.w, acc, 0, 0, 0);
    }
    __syncthreads();
  }
  const int n = n0 + wn * 32 + l31;
  const float bv = bias ? bias[n] : 0.f;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int m = m0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
    if (m < M) {
      float v = acc[r] + bv;
      if (EPI == 1) v = gelu_erf(v);
      C[(size_t)m * ldc + n] = v;
    }
  }
}

// Split-bf16 twin of k_gemm_f32_s.  The exact-fp32 matrix pipe (157 TFLOP/s, 64 cycles per 32x32x2 step) is what
// bounds these GEMMs (the 1024-token QKV GEMM is two rounds of 64x64 tiles at full fp32-MFMA rate).  Each operand
// value is split into two bf16 terms (v = hi + lo + O(2^-17 |v|)) by the staging threads on the way into LDS and
// a . w ~ a_hi.w_hi + a_lo.w_hi + a_hi.w_lo runs as 3 v_mfma_f32_32x32x16_bf16 per 16 k (96 cycles instead of 512),
// accumulated in fp32.  Per-product error ~2^-16 relative; the encoder's outputs stay within the path's 1e-4 bar
// (tests/test_bert_gpu.py, float64 oracle).
// weights -> bf16 hi / lo planes, once per load: the GEMM then copies 8 + 8 bytes instead of splitting 16
__global__ __launch_bounds__(256) void k_split_planes(const float* __restrict__ w, size_t npairs, unsigned* __restrict__ hi,
                                                      unsigned* __restrict__ lo) {
  const size_t i = blockIdx.x * 256ull + threadIdx.x;
  if (i >= npairs) return;
  unsigned h, l;
  split_bf16_pair(w[2 * i], w[2 * i + 1], h, l);
  hi[i] = h;
  lo[i] = l;
}

// Epilogue of the split-bf16 GEMMs.  The bias values are loaded first and waited for ONCE; the store loops are
// branch-free on full tiles (rows past M exist only in the last row block) and the fp32 / planes choice is made outside
// them.  With `if (m < M)` and `if (Chi)` around every store the compiler could not count the stores in flight behind the
// bias load and put an s_waitcnt vmcnt(0) in front of every value: 64 store round trips per lane, one after the other -
// the longest phase of the FFN1 GEMM (r3 ISA reading; profiles/r03_gemm_ablation.txt "neither": 255 us).
template <int EPI, int WMT, int WNT, int BM>
__device__ __forceinline__ void gemm_epilogue(f32x16 (&acc)[WMT][WNT], const float* __restrict__ bias, float* __restrict__ C,
                                              unsigned short* __restrict__ Chi, unsigned short* __restrict__ Clo, int ldc,
                                              int M, int m0, int n0, int wm, int wn, int lane) {
  const int half = lane >> 5, l31 = lane & 31;
  float bvs[WNT];
#pragma unroll
  for (int j = 0; j < WNT; ++j) bvs[j] = bias ? bias[n0 + (wn * WNT + j) * 32 + l31] : 0.f;
  __builtin_amdgcn_s_waitcnt(0x0f70);  // vmcnt(0)
  __builtin_amdgcn_sched_barrier(0);
  auto store_tile = [&](auto guard_tag, auto planes_tag) {
    constexpr bool GUARD = decltype(guard_tag)::value, PLANES = decltype(planes_tag)::value;
#pragma unroll
    for (int j = 0; j < WNT; ++j) {
      const int n = n0 + (wn * WNT + j) * 32 + l31;
#pragma unroll
      for (int i = 0; i < WMT; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int m = m0 + (wm * WMT + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
          if (GUARD && m >= M) continue;
          float v = acc[i][j][r] + bvs[j];
          if (EPI == 1) v = gelu_erf(v);
          if (PLANES) {
            // lanes n, n + 1 trade one term (quad_perm [1,0,3,2]): the even lane stores the pair of hi terms, the odd
            // lane the pair of lo terms - one 4-byte store per lane, 128 contiguous bytes per 32 lanes, as on the fp32 side
            unsigned short h, l;
            split_bf16_one(v, h, l);
            const bool odd = lane & 1;
            const unsigned got = (unsigned)__builtin_amdgcn_mov_dpp((int)(odd ? h : l), 0xB1, 0xF, 0xF, true);
            const unsigned word = odd ? (got | ((unsigned)l << 16)) : ((unsigned)h | (got << 16));
            *reinterpret_cast<unsigned*>((odd ? Clo : Chi) + (size_t)m * ldc + (n & ~1)) = word;
          } else {
            C[(size_t)m * ldc + n] = v;
          }
        }
    }
  };
  const bool full = m0 + BM <= M;  // workgroup-uniform
  if (Chi) {
    if (full) store_tile(std::false_type{}, std::true_type{});
    else store_tile(std::true_type{}, std::true_type{});
  } else {
    if (full) store_tile(std::false_type{}, std::false_type{});
    else store_tile(std::true_type{}, std::false_type{});
  }
}

// WMT x WNT: 32x32 MFMA tiles per wave (the workgroup's tile is 64 WMT x 64 WNT).  1 x 1 for the small-M ingest path
// (more workgroups); 2 x 2 for M >= 8192: 8 LDS fragment reads per 12 MFMAs instead of 16.  Stages are 64 deep.
//
// Operands: W always as the bf16 hi / lo planes k_split_planes wrote at load time.  AS: the A operand arrives as planes
// too, written by its producer (LayerNorm, attention, the GELU epilogue below: `Chi` / `Clo` non-null = write planes
// instead of fp32 C).  The split is the same function of the same fp32 value wherever it runs, so the results are
// bit-identical to splitting in the staging threads (AS = false, kept for a caller without planes).
//
// LDS image: a row is 16 slots of 16 B = [8 k-units of the hi plane | 8 of the lo plane]; slot(row, unit, plane) =
// (unit ^ (row & 7)) | ((plane ^ ((row >> 3) & 1)) << 3): the 16 rows a ds_read_b128 lane group touches fall on 16
// different slots (the r2 layout used 8 of them: 2-way conflicts on every fragment read, a third of the LDS cycles in
// profiles/r03_pmc_gemm_ffn1_before.txt).
//
// Tile order: workgroups i, i + 8, i + 16 ... share an XCD (and its L2); they walk the N tiles of ONE row block before
// the next, so a row block's A panel is fetched from HBM once and hit in L2 by the other N / BN - 1 workgroups.  With
// blockIdx.x = row block (r2) every N tile re-streamed the whole A operand: 1.2 GB fetched for a 100 MB operand in FFN1,
// L2 hit rate 39 %.
template <int EPI, int WMT = 1, int WNT = 1, bool AS = false, int WGM = 2>
__global__ __launch_bounds__(128 * WGM, WGM == 2 ? 2 : 1) void k_gemm_bf_s(const float* __restrict__ A, int lda,
                                                      const unsigned short* __restrict__ Whi,
                                                      const unsigned short* __restrict__ Wlo,
                                                      const float* __restrict__ bias, float* __restrict__ C, int ldc,
                                                      int M, int N, int K, int kstages,
                                                      const unsigned short* __restrict__ Ahi,
                                                      const unsigned short* __restrict__ Alo,
                                                      unsigned short* __restrict__ Chi, unsigned short* __restrict__ Clo) {
  // (r3, measured and removed: a double-buffered variant with 32-deep stages, two register sets two stages ahead and one
  // barrier per stage - LDS stores of stage s + 1 beside the MFMAs of stage s, all waits counted - was bit-identical and
  // SLOWER, 7.71 vs 7.35 ms per 512 x 128 batch: its 64-byte row segments double the L2 requests of loads that are
  // L2-bandwidth bound already.)
  // WGM x 2 waves: the row block is 32 WMT WGM rows.  WGM = 4 (8 waves, 256 x 128 tiles, a quarter less L2 -> LDS traffic per
  // MFMA, one workgroup per CU) measured level with 2 (223 vs 227 us per GEMM: profiles/r03_gemm_ablation.txt) and is not
  // dispatched
  constexpr int BKS = 64, BM = 32 * WGM * WMT, BN = 64 * WNT, UPR = 16, NT = 128 * WGM, RPP = NT / 16;  // rows per staging pass
  constexpr int RA = BM / RPP, RW = BN / RPP;  // 16-byte loads per thread and stage (plane route), float4s on the fp32 route
  extern __shared__ __attribute__((aligned(16))) u32x4b sgb[];
  u32x4b* sA = sgb;              // [BM][UPR]
  u32x4b* sW = sgb + BM * UPR;   // [BN][UPR]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int ntn = N / BN;
  const int bj = blockIdx.x >> 3;
  const int mt = (bj / ntn) * 8 + (blockIdx.x & 7), nt = bj - (bj / ntn) * ntn;
  const int m0 = mt * BM, n0 = nt * BN;
  if (m0 >= M) return;  // the row-block count is padded to a multiple of 8 (whole workgroup, before any barrier)
  const int half = lane >> 5, l31 = lane & 31;
  auto slot = [](int row, int unit, int plane) { return (unit ^ (row & 7)) | ((plane ^ ((row >> 3) & 1)) << 3); };
  f32x16 acc[WMT][WNT];
#pragma unroll
  for (int i = 0; i < WMT; ++i)
#pragma unroll
    for (int j = 0; j < WNT; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  // plane staging: thread -> (row tid >> 4 (+ 16 it), plane (tid >> 3) & 1, unit tid & 7): 8 lanes cover 128 contiguous bytes
  const int pu = tid & 7, ppl = (tid >> 3) & 1, prow = tid >> 4;
  // fp32 staging (AS = false): float4 index tid & 15 of row tid >> 4 (+ 16 it)
  const int sfu = tid & 15;
  const int k0 = blockIdx.z * kstages * BKS;
  f32x4 ra[AS ? 1 : RA];
  u32x4b rap[AS ? RA : 1], rwp[RW];
  const unsigned short* Wp = ppl ? Wlo : Whi;
  const unsigned short* Ap = ppl ? Alo : Ahi;
  auto issue = [&](int st) {
#pragma unroll
    for (int it = 0; it < RA; ++it) {
      int m = m0 + prow + RPP * it;
      if (m >= M) m = M - 1;
      if (AS) rap[it] = *reinterpret_cast<const u32x4b*>(Ap + (size_t)m * lda + k0 + st * BKS + pu * 8);
      else ra[it] = *reinterpret_cast<const f32x4*>(A + (size_t)m * lda + k0 + st * BKS + sfu * 4);
    }
#pragma unroll
    for (int it = 0; it < RW; ++it)
      rwp[it] = *reinterpret_cast<const u32x4b*>(Wp + (size_t)(n0 + prow + RPP * it) * K + k0 + st * BKS + pu * 8);
  };
  auto commit = [&]() {
#pragma unroll
    for (int it = 0; it < RA; ++it) {
      const int row = prow + RPP * it;
      if (AS) {
        sA[row * UPR + slot(row, pu, ppl)] = rap[it];
      } else {
        unsigned h0, l0, h1, l1;
        split_bf16_pair(ra[it][0], ra[it][1], h0, l0);
        split_bf16_pair(ra[it][2], ra[it][3], h1, l1);
        unsigned char* b = reinterpret_cast<unsigned char*>(sA + row * UPR);
        *reinterpret_cast<u32x2b*>(b + slot(row, sfu >> 1, 0) * 16 + (sfu & 1) * 8) = u32x2b{h0, h1};
        *reinterpret_cast<u32x2b*>(b + slot(row, sfu >> 1, 1) * 16 + (sfu & 1) * 8) = u32x2b{l0, l1};
      }
    }
#pragma unroll
    for (int it = 0; it < RW; ++it) {
      const int row = prow + RPP * it;
      sW[row * UPR + slot(row, pu, ppl)] = rwp[it];
    }
  };
  C += (size_t)blockIdx.z * M * ldc;
  issue(0);
  for (int st = 0; st < kstages; ++st) {
    commit();
    __syncthreads();
    if (st + 1 < kstages) issue(st + 1);
#pragma unroll
    for (int s8 = 0; s8 < BKS / 16; ++s8) {
      // (r3: reading group s8 + 1's fragments ahead of group s8's MFMAs, pinned with sched_group_barrier, measured
      // slower - 7.44 vs 7.21 ms per 512 x 128 batch: the allocator keeps one register set and waits behind each read)
      bf16x8 ah[WMT], al[WMT], wh[WNT], wl[WNT];
#pragma unroll
      for (int i = 0; i < WMT; ++i) {
        const int arow = (wm * WMT + i) * 32 + l31;
        const u32x4b* ap = sA + arow * UPR;
        ah[i] = __builtin_bit_cast(bf16x8, ap[slot(arow, 2 * s8 + half, 0)]);
        al[i] = __builtin_bit_cast(bf16x8, ap[slot(arow, 2 * s8 + half, 1)]);
      }
#pragma unroll
      for (int j = 0; j < WNT; ++j) {
        const int wrow = (wn * WNT + j) * 32 + l31;
        const u32x4b* wp = sW + wrow * UPR;
        wh[j] = __builtin_bit_cast(bf16x8, wp[slot(wrow, 2 * s8 + half, 0)]);
        wl[j] = __builtin_bit_cast(bf16x8, wp[slot(wrow, 2 * s8 + half, 1)]);
      }
#pragma unroll
      for (int i = 0; i < WMT; ++i)
#pragma unroll
        for (int j = 0; j < WNT; ++j) {
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], wh[j], acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[i], wh[j], acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], wl[j], acc[i][j], 0, 0, 0);
        }
    }
    __syncthreads();
  }
  gemm_epilogue<EPI, WMT, WNT, BM>(acc, bias, C, Chi, Clo, ldc, M, m0, n0, wm, wn, lane);
}

constexpr int kMaxSplit = 4;

void launch_add_ln(const float* a, int nsplit, const float* abias, const float* r, int T, int H, const float* g,
                   const float* b, float eps, float* out, hipStream_t stream, unsigned short* ohi = nullptr,
                   unsigned short* olo = nullptr) {
  const dim3 grid((unsigned)(((size_t)T * 64 + 255) / 256)), block(256);
  if (H == 384) hipLaunchKernelGGL(k_add_ln_fixed<6>, grid, block, 0, stream, a, nsplit, abias, r, T, g, b, eps, out, ohi, olo);
  else if (H == 768) hipLaunchKernelGGL(k_add_ln_fixed<12>, grid, block, 0, stream, a, nsplit, abias, r, T, g, b, eps, out, ohi, olo);
  else hipLaunchKernelGGL(k_add_ln, grid, block, 0, stream, a, nsplit, abias, r, T, H, g, b, eps, out, ohi, olo);
}

// Split-K factor for the two GEMMs that feed k_add_ln (N = hidden): with M = a few hundred tokens a 64x64 tiling
// of an [M x 384] output is ~100 workgroups on 256 CUs, each walking all of K serially; `splits` planes of
// partial sums (summed, in plane order, by k_add_ln) put 2-4x as many workgroups on the chip.
int pick_splits(int M, int N, int K) {
  if (M >= 8192) return 1;
  const int blocks = ((M + 63) / 64) * (N / 64);
  if (K % 128 == 0) {  // k_gemm_f32_s: stages of 128; the largest split <= kMaxSplit that divides them evenly
    const int stages = K / 128;
    int best = 1;
    for (int s = 2; s <= kMaxSplit; ++s)
      if (stages % s == 0 && blocks * best < 2 * num_cus()) best = s;
    return best;
  }
  int s = 1;
  while (s < kMaxSplit && blocks * s < 2 * num_cus() && (K / kBK) % (2 * s) == 0) s *= 2;
  return s;
}

struct Planes {  // an activation tensor as its two bf16 terms, [rows][ld] each
  unsigned short* hi = nullptr;
  unsigned short* lo = nullptr;
};

bool env_on(const char* name) { return !(getenv(name) && atoi(getenv(name)) == 0); }

// true when gemm() runs this shape on k_gemm_bf_s with pre-split weights: the kernels that take / write planes
bool gemm_takes_planes(int K, int splits) {
  static const bool on = env_on("EIOKU_GEMM_BF16") && env_on("EIOKU_GEMM_S");
  return on && K % (128 * splits) == 0;
}

template <int EPI, int WMT, int WNT, bool AS, int WGM = 2>
void launch_bf64(dim3 grid, size_t lds, hipStream_t stream, const float* A, int lda, const float* W, const unsigned short* whi,
                 const unsigned short* wlo, const float* bias, float* C, int ldc, int M, int N, int K, int kst, const Planes* ap,
                 const Planes* cp) {
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_gemm_bf_s<EPI, WMT, WNT, AS, WGM>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
    attr = true;
  }
  hipLaunchKernelGGL((k_gemm_bf_s<EPI, WMT, WNT, AS, WGM>), grid, dim3(128 * WGM), lds, stream, A, lda, whi, wlo, bias, C, ldc, M, N, K, kst,
                     ap ? ap->hi : nullptr, ap ? ap->lo : nullptr, cp ? cp->hi : nullptr, cp ? cp->lo : nullptr);
}

// C = A . W^T (+ bias, + GELU when epi == 1).  splits > 1: C receives `splits` partial planes [splits][M][ldc]
// WITHOUT bias (epi must be 0); the consumer adds them up.  `ap` / `cp` (only where gemm_takes_planes()): A is read
// from / C is written as bf16 hi + lo planes (A / C themselves are then not touched).
int gemm(const float* A, int lda, const float* W, const float* bias, float* C, int ldc, int M, int N, int K, int epi,
         int splits, hipStream_t stream, const eioku_bert::SplitW* ws = nullptr, const Planes* ap = nullptr,
         const Planes* cp = nullptr) {
  EIOKU_REQUIRE(N % 128 == 0 && K % kBK == 0, "gemm shape N=%d K=%d must be multiples of 128 / 32", N, K);
  EIOKU_REQUIRE(splits >= 1 && (K / kBK) % splits == 0 && (splits == 1 || epi == 0), "bad split-K %d", splits);
  const int kchunks = K / kBK / splits;
  if (splits > 1) bias = nullptr;
  const bool s_route = K % (128 * splits) == 0 && env_on("EIOKU_GEMM_S");
  const bool bf = env_on("EIOKU_GEMM_BF16");
  const unsigned short* whi = ws ? ws->hi : nullptr;
  const unsigned short* wlo = ws ? ws->lo : nullptr;
  EIOKU_REQUIRE(!(ap || cp) || (s_route && bf && whi && (!cp || splits == 1)), "planes on a GEMM route that has none");
  prof_start(EIOKU_PROF_GEMM, stream);
  // EIOKU_GEMM_BF16=0 / EIOKU_GEMM_S=0 / EIOKU_ATTN_MFMA=0 route the encoder through the fp32-FMA kernels (the
  // numerics cross-check of tests/test_bert_gpu.py::test_fp32_fma_route_matches_the_mfma_route)
  if (s_route && bf && whi) {
    const int kst = K / 64 / splits;  // 64-deep stages
    if (M >= 8192 && splits == 1) {
      // 128 x 128 tiles (64 KB of LDS, two workgroups per CU)
      const dim3 grid((unsigned)(((M + 127) / 128 + 7) / 8 * 8 * (N / 128)), 1u, 1u);  // row blocks padded to the 8 XCDs
      const size_t lds = (size_t)(128 + 128) * 64 * 4;
#define EIOKU_L(E, AS_) launch_bf64<E, 2, 2, AS_>(grid, lds, stream, A, lda, W, whi, wlo, bias, C, ldc, M, N, K, kst, ap, cp)
      if (epi == 1) { if (ap) EIOKU_L(1, true); else EIOKU_L(1, false); }
      else { if (ap) EIOKU_L(0, true); else EIOKU_L(0, false); }
#undef EIOKU_L
    } else {
      // small M (the ingest path: a few segments per call, running beside the detector): 64 x 64 tiles, 32 KB of LDS
      // per workgroup.  64-deep stages are as fast alone as 128-deep ones (0.457 vs 0.460 ms per 8 x 128 tokens) but
      // two workgroups no longer take 128 of a CU's 160 KB away from the conv kernels on the other streams: +1.5 % on
      // the overlapped step.  Same k order, so the results are bit-identical.
      const dim3 grid((unsigned)(((M + 63) / 64 + 7) / 8 * 8 * (N / 64)), 1u, (unsigned)splits);
      const size_t lds = (size_t)(64 + 64) * 64 * 4;
#define EIOKU_L(E, AS_) launch_bf64<E, 1, 1, AS_>(grid, lds, stream, A, lda, W, whi, wlo, bias, C, ldc, M, N, K, kst, ap, cp)
      if (epi == 1) { if (ap) EIOKU_L(1, true); else EIOKU_L(1, false); }
      else { if (ap) EIOKU_L(0, true); else EIOKU_L(0, false); }
#undef EIOKU_L
    }
  } else if (M >= 8192 && !(bf && s_route)) {
    dim3 grid((unsigned)((M + 127) / 128), (unsigned)(N / 128), (unsigned)splits);
    if (epi == 1) hipLaunchKernelGGL((k_gemm_f32<1, 128, 128>), grid, dim3(256), 0, stream, A, lda, W, bias, C, ldc, M, N, K, kchunks);
    else hipLaunchKernelGGL((k_gemm_f32<0, 128, 128>), grid, dim3(256), 0, stream, A, lda, W, bias, C, ldc, M, N, K, kchunks);
  } else if (s_route) {
    dim3 grid((unsigned)((M + 63) / 64), (unsigned)(N / 64), (unsigned)splits);
    const size_t lds = (size_t)(64 + 64) * 128 * 4;
    static bool attr = false;
    if (!attr) {
      EIOKU_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_gemm_f32_s<0>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
      EIOKU_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_gemm_f32_s<1>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
      attr = true;
    }
    const int kstages = K / 128 / splits;
    // EIOKU_GEMM_BF16=0, or a caller without the handle's weight planes: the exact-fp32 matrix pipe
    if (epi == 1) hipLaunchKernelGGL((k_gemm_f32_s<1>), grid, dim3(256), lds, stream, A, lda, W, bias, C, ldc, M, N, K, kstages);
    else hipLaunchKernelGGL((k_gemm_f32_s<0>), grid, dim3(256), lds, stream, A, lda, W, bias, C, ldc, M, N, K, kstages);
  } else {
    dim3 grid((unsigned)((M + 63) / 64), (unsigned)(N / 64), (unsigned)splits);
    if (epi == 1) hipLaunchKernelGGL((k_gemm_f32<1, 64, 64>), grid, dim3(256), 0, stream, A, lda, W, bias, C, ldc, M, N, K, kchunks);
    else hipLaunchKernelGGL((k_gemm_f32<0, 64, 64>), grid, dim3(256), 0, stream, A, lda, W, bias, C, ldc, M, N, K, kchunks);
  }
  prof_stop(EIOKU_PROF_GEMM, stream);
  EIOKU_LAUNCH_CHECK();
  return EIOKU_OK;
}

}  // namespace

extern "C" {

int eioku_bert_create(int vocab, int hidden, int layers, int heads, int ffn, int max_pos, int type_vocab,
                      float ln_eps, eioku_bert** out) {
  EIOKU_REQUIRE_INIT();
  EIOKU_REQUIRE(out, "NULL out");
  EIOKU_REQUIRE(hidden % 128 == 0 && hidden <= 1024, "hidden %d must be a multiple of 128 and <= 1024", hidden);
  EIOKU_REQUIRE(heads > 0 && hidden / heads == 32 && hidden % heads == 0, "head_dim must be 32 (hidden %d / heads %d)", hidden, heads);
  EIOKU_REQUIRE(ffn % 128 == 0, "ffn %d must be a multiple of 128", ffn);
  EIOKU_REQUIRE(vocab > 0 && layers > 0 && max_pos > 0 && type_vocab > 0, "bad config");
  auto* m = new eioku_bert();
  m->vocab = vocab; m->H = hidden; m->L = layers; m->heads = heads; m->ffn = ffn; m->max_pos = max_pos;
  m->type_vocab = type_vocab; m->eps = ln_eps;
  auto add = [&](const std::string& n, std::vector<int> shape) {
    eioku_bert::Tensor t;
    t.name = n;
    t.shape = std::move(shape);
    m->tensors.push_back(t);
  };
  const int H = hidden;
  add("embeddings.word_embeddings.weight", {vocab, H});
  add("embeddings.position_embeddings.weight", {max_pos, H});
  add("embeddings.token_type_embeddings.weight", {type_vocab, H});
  add("embeddings.LayerNorm.weight", {H});
  add("embeddings.LayerNorm.bias", {H});
  for (int l = 0; l < layers; ++l) {
    const std::string p = "encoder.layer." + std::to_string(l) + ".";
    // query | key | value stacked into one [3H][H] operand (one GEMM), filled through three sub-tensors
    add(p + "attention.self.query.weight", {H, H});
    add(p + "attention.self.key.weight", {H, H});
    add(p + "attention.self.value.weight", {H, H});
    add(p + "attention.self.query.bias", {H});
    add(p + "attention.self.key.bias", {H});
    add(p + "attention.self.value.bias", {H});
    add(p + "attention.output.dense.weight", {H, H});
    add(p + "attention.output.dense.bias", {H});
    add(p + "attention.output.LayerNorm.weight", {H});
    add(p + "attention.output.LayerNorm.bias", {H});
    add(p + "intermediate.dense.weight", {ffn, H});
    add(p + "intermediate.dense.bias", {ffn});
    add(p + "output.dense.weight", {H, ffn});
    add(p + "output.dense.bias", {H});
    add(p + "output.LayerNorm.weight", {H});
    add(p + "output.LayerNorm.bias", {H});
  }
  // allocate; q/k/v weights and biases of a layer are contiguous so the fused GEMM can use them in place
  for (size_t i = 0; i < m->tensors.size(); ++i) {
    auto& t = m->tensors[i];
    const bool qw = t.name.find("attention.self.query.weight") != std::string::npos;
    const bool qb = t.name.find("attention.self.query.bias") != std::string::npos;
    if (qw || qb) {
      const size_t n = t.numel();
      float* base = nullptr;
      if (hipMalloc((void**)&base, 3 * n * sizeof(float)) != hipSuccess) {
        set_error("hipMalloc failed for %s", t.name.c_str());
        return EIOKU_ENOMEM;
      }
      m->tensors[i].dev = base;
      m->tensors[i + 1].dev = base + n;
      m->tensors[i + 2].dev = base + 2 * n;
    } else if (!t.dev) {
      if (hipMalloc((void**)&t.dev, t.numel() * sizeof(float)) != hipSuccess) {
        set_error("hipMalloc failed for %s", t.name.c_str());
        return EIOKU_ENOMEM;
      }
    }
  }
  *out = m;
  return EIOKU_OK;
}

void eioku_bert_destroy(eioku_bert* m) {
  if (!m) return;
  (void)hipDeviceSynchronize();
  for (size_t i = 0; i < m->tensors.size(); ++i) {
    const auto& n = m->tensors[i].name;
    const bool sub = n.find("self.key.") != std::string::npos || n.find("self.value.") != std::string::npos;
    if (!sub && m->tensors[i].dev) (void)hipFree(m->tensors[i].dev);
  }
  for (auto& w : m->wsplit) {
    if (w.hi) (void)hipFree(w.hi);
    if (w.lo) (void)hipFree(w.lo);
  }
  void* bufs[] = {m->d_ids, m->d_mask, m->x, m->y, m->qkv, m->ctx, m->mid, m->d_out, m->xp, m->cp, m->mp};
  for (void* b : bufs)
    if (b) (void)hipFree(b);
  delete m;
}

int eioku_bert_num_tensors(const eioku_bert* m) { return m ? (int)m->tensors.size() : 0; }

int eioku_bert_tensor_info(const eioku_bert* m, int idx, char* name, size_t cap, int* rows, int* cols) {
  EIOKU_REQUIRE(m && idx >= 0 && idx < (int)m->tensors.size(), "bad tensor index %d", idx);
  const auto& t = m->tensors[idx];
  if (name && cap) snprintf(name, cap, "%s", t.name.c_str());
  if (rows) *rows = t.shape[0];
  if (cols) *cols = t.shape.size() > 1 ? t.shape[1] : 1;
  return EIOKU_OK;
}

int eioku_bert_set_tensor(eioku_bert* m, int idx, const float* host, size_t numel) {
  EIOKU_REQUIRE_INIT();
  EIOKU_REQUIRE(m && idx >= 0 && idx < (int)m->tensors.size() && host, "bad argument");
  auto& t = m->tensors[idx];
  EIOKU_REQUIRE(numel == t.numel(), "%s: expected %zu elements, got %zu", t.name.c_str(), t.numel(), numel);
  EIOKU_HIP_CHECK(hipMemcpy(t.dev, host, numel * sizeof(float), hipMemcpyHostToDevice));
  t.set = true;
  m->wsplit_dirty = true;
  return EIOKU_OK;
}

// ids: int32 [B][S] (0-padded), mask: uint8 [B][S]; out: float32 [B][H] unit vectors.
int eioku_bert_embed(eioku_bert* m, const int32_t* ids, const uint8_t* mask, int B, int S, float* out, int mem,
                     void* stream_) {
  EIOKU_REQUIRE_INIT();
  EIOKU_REQUIRE(m && B >= 0 && S > 0, "bad argument");
  EIOKU_REQUIRE(S <= m->max_pos && S <= 512, "sequence length %d exceeds max positions %d / 512", S, m->max_pos);
  EIOKU_REQUIRE(mem == EIOKU_MEM_HOST || mem == EIOKU_MEM_DEVICE, "bad mem flag %d", mem);
  if (B == 0) return EIOKU_OK;
  EIOKU_REQUIRE(ids && mask && out, "NULL buffer");
  for (const auto& t : m->tensors) EIOKU_REQUIRE(t.set, "tensor %s has no weights", t.name.c_str());
  hipStream_t stream = (hipStream_t)stream_;
  const int H = m->H, T = B * S;
  int rc;
  if (m->wsplit_dirty) {  // (re)build the bf16 hi / lo planes of the four GEMM operands of every layer
    if (m->wsplit.empty()) m->wsplit.resize((size_t)4 * m->L);
    for (int l = 0; l < m->L; ++l) {
      const std::string p = "encoder.layer." + std::to_string(l) + ".";
      const char* names[4] = {"attention.self.query.weight", "attention.output.dense.weight", "intermediate.dense.weight",
                              "output.dense.weight"};
      const size_t numel[4] = {(size_t)3 * H * H, (size_t)H * H, (size_t)m->ffn * H, (size_t)H * m->ffn};
      for (int k = 0; k < 4; ++k) {
        auto& w = m->wsplit[(size_t)4 * l + k];
        if (!w.hi) {
          EIOKU_HIP_CHECK(hipMalloc((void**)&w.hi, numel[k] * 2));
          EIOKU_HIP_CHECK(hipMalloc((void**)&w.lo, numel[k] * 2));
        }
        const size_t npairs = numel[k] / 2;
        hipLaunchKernelGGL(k_split_planes, dim3((unsigned)((npairs + 255) / 256)), dim3(256), 0, stream, tp(m, p + names[k]),
                           npairs, (unsigned*)w.hi, (unsigned*)w.lo);
      }
    }
    EIOKU_LAUNCH_CHECK();
    m->wsplit_dirty = false;
  }
  const int32_t* d_ids = ids;
  const uint8_t* d_mask = mask;
  float* d_out = out;
  if (mem == EIOKU_MEM_HOST) {
    for (int i = 0; i < T; ++i) EIOKU_REQUIRE(ids[i] >= 0 && ids[i] < m->vocab, "token id %d at %d outside the vocabulary", ids[i], i);
    if ((rc = grow(&m->d_ids, &m->ids_cap, (size_t)T * 4))) return rc;
    if ((rc = grow(&m->d_mask, &m->mask_cap, (size_t)T))) return rc;
    if ((rc = grow(&m->d_out, &m->out_cap, (size_t)B * H * 4))) return rc;
    EIOKU_HIP_CHECK(hipMemcpyAsync(m->d_ids, ids, (size_t)T * 4, hipMemcpyHostToDevice, stream));
    EIOKU_HIP_CHECK(hipMemcpyAsync(m->d_mask, mask, (size_t)T, hipMemcpyHostToDevice, stream));
    d_ids = m->d_ids;
    d_mask = m->d_mask;
    d_out = m->d_out;
  }
  // every GEMM of a layer on the split-bf16 kernels (the default): each activation a GEMM reads is written by its
  // producer as bf16 hi | lo planes; ctx and the FFN's 4H-wide intermediate then exist only in that form
  const int sp_o = pick_splits(T, H, H), sp_f = pick_splits(T, H, m->ffn);
  // EIOKU_GEMM_PLANES=0: activations stay fp32 and the GEMMs split them in their staging threads (the r2 route; same
  // function of the same values, so the embeddings must be the same bytes: tests/test_bert_gpu.py)
  static const bool planes_on = env_on("EIOKU_GEMM_PLANES");
  const bool planes = planes_on && gemm_takes_planes(H, 1) && gemm_takes_planes(H, sp_o) && gemm_takes_planes(m->ffn, sp_f);
  if ((rc = grow(&m->x, &m->x_cap, (size_t)T * H * 4))) return rc;
  if ((rc = grow(&m->y, &m->y_cap, (size_t)kMaxSplit * T * H * 4))) return rc;
  if ((rc = grow(&m->qkv, &m->qkv_cap, (size_t)T * 3 * H * 4))) return rc;
  Planes xP, cP, mP;
  if (planes) {
    if ((rc = grow(&m->xp, &m->xp_cap, (size_t)2 * T * H * 2))) return rc;
    if ((rc = grow(&m->cp, &m->cp_cap, (size_t)2 * T * H * 2))) return rc;
    if ((rc = grow(&m->mp, &m->mp_cap, (size_t)2 * T * m->ffn * 2))) return rc;
    xP.hi = m->xp; xP.lo = m->xp + (size_t)T * H;
    cP.hi = m->cp; cP.lo = m->cp + (size_t)T * H;
    mP.hi = m->mp; mP.lo = m->mp + (size_t)T * m->ffn;
  } else {
    if ((rc = grow(&m->mid, &m->mid_cap, (size_t)T * m->ffn * 4))) return rc;
  }
  const bool amfma = env_on("EIOKU_ATTN_MFMA") && S <= 128 && H == m->heads * 32;
  // the other attention kernels write fp32 ctx (split afterwards by k_split_planes when the GEMMs take planes)
  if (!planes || !amfma)
    if ((rc = grow(&m->ctx, &m->ctx_cap, (size_t)T * H * 4))) return rc;

  const unsigned tok_blocks = (unsigned)(((size_t)T * 64 + 255) / 256);
  hipLaunchKernelGGL(k_embed_ln, dim3(tok_blocks), dim3(256), 0, stream, d_ids, T, S, H, m->vocab,
                     tp(m, "embeddings.word_embeddings.weight"), tp(m, "embeddings.position_embeddings.weight"),
                     tp(m, "embeddings.token_type_embeddings.weight"), tp(m, "embeddings.LayerNorm.weight"),
                     tp(m, "embeddings.LayerNorm.bias"), m->eps, m->x, xP.hi, xP.lo);
  EIOKU_LAUNCH_CHECK();
  const int parts = S <= 256 ? 4 : 1;
  const int qsplit = parts == 4 ? (S + 31) / 32 : 1;  // 32 queries x 4 lanes = 2 waves per workgroup
  const int athreads = parts == 4 ? 128 : ((S + 63) / 64) * 64;
  const size_t alds = (size_t)S * 8 * 16 * 2 + (size_t)S * 4;
  if (alds > 64 * 1024) {
    static bool attr = false;
    if (!attr) {
      EIOKU_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_attention<1>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, 140 * 1024));
      attr = true;
    }
  }
  const Planes* xa = planes ? &xP : nullptr;
  const Planes* ca = planes ? &cP : nullptr;
  const Planes* ma = planes ? &mP : nullptr;
  for (int l = 0; l < m->L; ++l) {
    const std::string p = "encoder.layer." + std::to_string(l) + ".";
    if ((rc = gemm(m->x, H, tp(m, p + "attention.self.query.weight"), tp(m, p + "attention.self.query.bias"), m->qkv,
                   3 * H, T, 3 * H, H, 0, 1, stream, &m->wsplit[(size_t)4 * l + 0], xa))) return rc;
    static const bool abf = env_on("EIOKU_ATTN_BF16");  // 0: the exact-fp32 matrix pipe (k_attention_mfma)
    if (amfma && abf) {
      hipLaunchKernelGGL(k_attention_bf, dim3(B, m->heads), dim3(256), 0, stream, m->qkv, d_mask, S, H, m->ctx, cP.hi, cP.lo);
    } else if (amfma) {
      hipLaunchKernelGGL(k_attention_mfma, dim3(B, m->heads), dim3(256), 0, stream, m->qkv, d_mask, S, H, m->ctx, cP.hi, cP.lo);
    } else {
      if (parts == 4)
        hipLaunchKernelGGL(k_attention8, dim3(B, m->heads, qsplit), dim3(256), alds, stream, m->qkv, d_mask, S, H, m->ctx);
      else
        hipLaunchKernelGGL(k_attention<1>, dim3(B, m->heads), dim3(athreads), alds, stream, m->qkv, d_mask, S, H, m->ctx);
      if (planes) {
        const size_t npairs = (size_t)T * H / 2;
        hipLaunchKernelGGL(k_split_planes, dim3((unsigned)((npairs + 255) / 256)), dim3(256), 0, stream, m->ctx, npairs,
                           (unsigned*)cP.hi, (unsigned*)cP.lo);
      }
    }
    EIOKU_LAUNCH_CHECK();
    if ((rc = gemm(m->ctx, H, tp(m, p + "attention.output.dense.weight"), tp(m, p + "attention.output.dense.bias"), m->y,
                   H, T, H, H, 0, sp_o, stream, &m->wsplit[(size_t)4 * l + 1], ca))) return rc;
    launch_add_ln(m->y, sp_o, sp_o > 1 ? tp(m, p + "attention.output.dense.bias") : nullptr, m->x, T, H,
                       tp(m, p + "attention.output.LayerNorm.weight"), tp(m, p + "attention.output.LayerNorm.bias"),
                       m->eps, m->x, stream, xP.hi, xP.lo);
    EIOKU_LAUNCH_CHECK();
    if ((rc = gemm(m->x, H, tp(m, p + "intermediate.dense.weight"), tp(m, p + "intermediate.dense.bias"), m->mid, m->ffn,
                   T, m->ffn, H, 1, 1, stream, &m->wsplit[(size_t)4 * l + 2], xa, ma))) return rc;
    if ((rc = gemm(m->mid, m->ffn, tp(m, p + "output.dense.weight"), tp(m, p + "output.dense.bias"), m->y, H, T, H,
                   m->ffn, 0, sp_f, stream, &m->wsplit[(size_t)4 * l + 3], ma))) return rc;
    // the last layer's output feeds the pooling only: no planes
    const bool last = l + 1 == m->L;
    launch_add_ln(m->y, sp_f, sp_f > 1 ? tp(m, p + "output.dense.bias") : nullptr, m->x, T, H,
                       tp(m, p + "output.LayerNorm.weight"), tp(m, p + "output.LayerNorm.bias"), m->eps, m->x, stream,
                       last ? nullptr : xP.hi, last ? nullptr : xP.lo);
    EIOKU_LAUNCH_CHECK();
  }
  const int pthreads = ((H + 63) / 64) * 64;
  // two token groups when they fit in a workgroup and in the 1024-float staging arrays
  const int pgroups = (2 * pthreads <= 1024 && S <= 1024) ? 2 : 1;
  EIOKU_REQUIRE(S <= 1024, "sequence length %d exceeds the pooling kernel's mask buffer", S);
  hipLaunchKernelGGL(k_pool_norm, dim3(B), dim3(pgroups * pthreads), 0, stream, m->x, d_mask, S, H, d_out);
  EIOKU_LAUNCH_CHECK();
  m->flops_last = (double)T * m->L * (2.0 * H * 3 * H + 2.0 * H * H + 4.0 * H * m->ffn) +
                  (double)B * m->heads * m->L * 4.0 * S * S * 32;
  if (mem == EIOKU_MEM_HOST) {
    EIOKU_HIP_CHECK(hipMemcpyAsync(out, d_out, (size_t)B * H * 4, hipMemcpyDeviceToHost, stream));
    EIOKU_HIP_CHECK(hipStreamSynchronize(stream));
  }
  return EIOKU_OK;
}

int eioku_bert_last_flops(const eioku_bert* m, double* flops) {
  EIOKU_REQUIRE(m && flops, "NULL argument");
  *flops = m->flops_last;
  return EIOKU_OK;
}

}  // extern "C"
