// K4 conv_igemm_bias_silu: NHWC fp16 convolution as implicit GEMM on v_mfma_f32_16x16x32_f16.
//
// Workgroup (256 threads = 4 waves) -> one 8x16 tile of output pixels of one image x one tile of
// 16*NF output channels.  Per 32-channel input chunk:
//   1. the input halo patch ((8-1)*S+KS) x ((16-1)*S+KS) pixels x 32 channels is loaded ONCE from
//      HBM/L2 with 16-byte coalesced loads (zero filled outside the image / past Cin) into LDS,
//      64 B per pixel, XOR-swizzled so that every ds_read_b128 fragment read is conflict free;
//   2. the weight tile [taps][16*NF][32] (pre-packed contiguously on the host) is copied to LDS;
//   3. for each of the KS*KS taps the MFMA B fragments (activations: 16 pixels x 32 ch) are read
//      straight out of the patch at the tap's shifted position - the 9x im2col re-read never
//      leaves the CU - and multiplied by the A fragments (weights: 16 cout x 32 ch).
// D = W . X^T orientation: each lane ends with 4 consecutive output channels of one pixel, so the
// epilogue (bias, SiLU, fp16 round, residual add, concat-slice offset) stores 8 contiguous bytes.
//
// Numerics: fp16 operands, fp32 MFMA accumulation, bias/SiLU in fp32, one RNE rounding to fp16;
// a residual is added as fp16(float(y16) + float(x16)), i.e. the fp16 network `x + cv2(cv1(x))`.
#include "conv.h"

#include <algorithm>
#include <cstdlib>

namespace eioku {

namespace {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float float4v __attribute__((ext_vector_type(4)));

constexpr int kTH = 8, kTW = 16;

struct ConvArgs {
  const __half* in;
  const uint4* wgt;
  const float* bias;
  __half* out;
  float* out_f32;
  const __half* res;
  int N, H, W, Cin, in_cs;
  int Ho, Wo, Cout, out_cs;
  int res_cs;
  int tiles_w, tiles_h, nchunks, act;
};

__device__ __forceinline__ float silu_f32(float v) {
  return v * __builtin_amdgcn_rcpf(1.0f + __expf(-v));
}

template <int NF, int KS, int S>
__global__ __launch_bounds__(256) void k_conv_igemm(ConvArgs a) {
  constexpr int PH = (kTH - 1) * S + KS;
  constexpr int PW = (kTW - 1) * S + KS;
  constexpr int PWH = (PW + 1) / 2;            // S == 2: columns stored de-interleaved (even | odd)
  constexpr int PWS = (S == 2) ? 2 * PWH : PW;  // LDS row pitch in pixels
  constexpr int TAPS = KS * KS;
  constexpr int PAD = KS / 2;
  constexpr int PATCH_U = PH * PWS * 4;  // uint4 units
  constexpr int WT_U = TAPS * 16 * NF * 4;

  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  uint4* patch = reinterpret_cast<uint4*>(smem);
  uint4* wt = patch + PATCH_U;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  int bx = blockIdx.x;
  const int tw = bx % a.tiles_w;
  bx /= a.tiles_w;
  const int th = bx % a.tiles_h;
  const int n = bx / a.tiles_h;
  const int oh0 = th * kTH, ow0 = tw * kTW;
  const int ih0 = oh0 * S - PAD, iw0 = ow0 * S - PAD;
  const int co_tile = blockIdx.y;

  float4v acc[2][NF];
#pragma unroll
  for (int m = 0; m < 2; ++m)
#pragma unroll
    for (int f = 0; f < NF; ++f) acc[m][f] = float4v{0.f, 0.f, 0.f, 0.f};

  const __half* in_n = a.in + (size_t)n * a.H * a.W * a.in_cs;
  const uint4* wsrc = a.wgt + (size_t)co_tile * a.nchunks * WT_U;

  for (int cc = 0; cc < a.nchunks; ++cc) {
    // ---- 1. halo patch: HBM/L2 -> LDS, each input byte of the tile read once per chunk ----
    for (int idx = tid; idx < PH * PW * 4; idx += 256) {
      const int pix = idx >> 2, unit = idx & 3;
      const int py = pix / PW, px = pix - py * PW;
      const int ih = ih0 + py, iw = iw0 + px;
      const int c = cc * 32 + unit * 8;
      uint4 v = make_uint4(0, 0, 0, 0);
      if ((unsigned)ih < (unsigned)a.H && (unsigned)iw < (unsigned)a.W && c < a.Cin)
        v = *reinterpret_cast<const uint4*>(in_n + ((size_t)ih * a.W + iw) * a.in_cs + c);
      const int col = (S == 2) ? ((px & 1) * PWH + (px >> 1)) : px;
      const int p = py * PWS + col;
      patch[p * 4 + (unit ^ ((p >> 1) & 3))] = v;
    }
    // ---- 2. weight tile (contiguous on the host side) ----
    for (int idx = tid; idx < WT_U; idx += 256) {
      const int row = idx >> 2, unit = idx & 3;
      wt[row * 4 + (unit ^ ((row >> 1) & 3))] = wsrc[(size_t)cc * WT_U + idx];
    }
    __syncthreads();
    // ---- 3. taps: B fragments from the shifted patch, A fragments from the weight tile ----
#pragma unroll
    for (int tap = 0; tap < TAPS; ++tap) {
      const int kh = tap / KS, kw = tap % KS;
      half8 bfrag[2];
#pragma unroll
      for (int m = 0; m < 2; ++m) {
        const int py = (wave * 2 + m) * S + kh;
        const int px = (lane & 15) * S + kw;
        const int col = (S == 2) ? ((px & 1) * PWH + (px >> 1)) : px;
        const int p = py * PWS + col;
        uint4 u = patch[p * 4 + ((lane >> 4) ^ ((p >> 1) & 3))];
        bfrag[m] = *reinterpret_cast<half8*>(&u);
      }
#pragma unroll
      for (int f = 0; f < NF; ++f) {
        const int row = tap * 16 * NF + f * 16 + (lane & 15);
        uint4 u = wt[row * 4 + ((lane >> 4) ^ ((row >> 1) & 3))];
        half8 afrag = *reinterpret_cast<half8*>(&u);
#pragma unroll
        for (int m = 0; m < 2; ++m)
          acc[m][f] = __builtin_amdgcn_mfma_f32_16x16x32_f16(afrag, bfrag[m], acc[m][f], 0, 0, 0);
      }
    }
    __syncthreads();
  }

  // ---- epilogue: lane = pixel (lane&15), 4 consecutive couts ((lane>>4)*4 + j) per fragment ----
  const int ow = ow0 + (lane & 15);
#pragma unroll
  for (int m = 0; m < 2; ++m) {
    const int oh = oh0 + wave * 2 + m;
    if (oh >= a.Ho || ow >= a.Wo) continue;
    const size_t opix = ((size_t)n * a.Ho + oh) * a.Wo + ow;
#pragma unroll
    for (int f = 0; f < NF; ++f) {
      const int c0 = co_tile * 16 * NF + f * 16 + (lane >> 4) * 4;
      if (c0 >= a.Cout) continue;
      const float4 b = *reinterpret_cast<const float4*>(a.bias + c0);
      float v[4] = {acc[m][f][0] + b.x, acc[m][f][1] + b.y, acc[m][f][2] + b.z, acc[m][f][3] + b.w};
      if (a.act == kActSiLU) {
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = silu_f32(v[j]);
      }
      if (a.out_f32) {
        float* o = a.out_f32 + opix * a.Cout + c0;
#pragma unroll
        for (int j = 0; j < 4; ++j)
          if (c0 + j < a.Cout) o[j] = v[j];
        continue;
      }
      __half h[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) h[j] = __float2half_rn(v[j]);
      if (c0 + 3 < a.Cout) {
        if (a.res) {
          const uint2 r = *reinterpret_cast<const uint2*>(a.res + opix * a.res_cs + c0);
          const __half* rh = reinterpret_cast<const __half*>(&r);
#pragma unroll
          for (int j = 0; j < 4; ++j) h[j] = __float2half_rn(__half2float(h[j]) + __half2float(rh[j]));
        }
        *reinterpret_cast<uint2*>(a.out + opix * a.out_cs + c0) = *reinterpret_cast<uint2*>(h);
      } else {
#pragma unroll
        for (int j = 0; j < 4; ++j)
          if (c0 + j < a.Cout) {
            __half o = h[j];
            if (a.res) o = __float2half_rn(__half2float(o) + __half2float(a.res[opix * a.res_cs + c0 + j]));
            a.out[opix * a.out_cs + c0 + j] = o;
          }
      }
    }
  }
}

template <int NF, int KS, int S>
int launch(const ConvArgs& a, int ntiles, hipStream_t stream) {
  constexpr int PH = (kTH - 1) * S + KS;
  constexpr int PW = (kTW - 1) * S + KS;
  constexpr int PWS = (S == 2) ? 2 * ((PW + 1) / 2) : PW;
  constexpr size_t lds = (size_t)(PH * PWS * 4 + KS * KS * 16 * NF * 4) * 16;
  static bool attr_set = false;
  if (!attr_set && lds > 64 * 1024) {
    EIOKU_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_conv_igemm<NF, KS, S>),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    attr_set = true;
  }
  dim3 grid((unsigned)(a.tiles_w * a.tiles_h * a.N), (unsigned)ntiles);
  hipLaunchKernelGGL((k_conv_igemm<NF, KS, S>), grid, dim3(256), lds, stream, a);
  EIOKU_LAUNCH_CHECK();
  return EIOKU_OK;
}

template <int KS, int S>
int launch_nf(int nf, const ConvArgs& a, int ntiles, hipStream_t stream) {
  switch (nf) {
    case 1: return launch<1, KS, S>(a, ntiles, stream);
    case 2: return launch<2, KS, S>(a, ntiles, stream);
    case 3: return launch<3, KS, S>(a, ntiles, stream);
    case 4: return launch<4, KS, S>(a, ntiles, stream);
    case 5: return launch<5, KS, S>(a, ntiles, stream);
    case 6: return launch<6, KS, S>(a, ntiles, stream);
    case 8: return launch<8, KS, S>(a, ntiles, stream);
  }
  set_error("unsupported nf %d", nf);
  return EIOKU_EINVAL;
}

// Fewest padded channels first, then the widest tile (fewer re-reads of the input patch).
int pick_nf(int cout, int ks) {
  const int frags = (cout + 15) / 16;
  const int cands[] = {8, 6, 5, 4, 3, 2, 1};
  int best = 1, best_waste = 1 << 30;
  for (int nf : cands) {
    if (ks == 3 && nf > 6) continue;  // LDS: 9 taps x 16*NF x 64 B
    int waste = ((frags + nf - 1) / nf) * nf - frags;
    if (waste < best_waste) {
      best_waste = waste;
      best = nf;
    }
  }
  return best;
}

}  // namespace

int conv_weights_create(ConvWeights* cw, int cout, int cin, int ks, int stride, const float* w,
                        const float* b) {
  EIOKU_REQUIRE(cout > 0 && cin > 0, "bad conv shape cout=%d cin=%d", cout, cin);
  EIOKU_REQUIRE(ks == 1 || ks == 3, "kernel size %d not supported (1 or 3)", ks);
  EIOKU_REQUIRE((stride == 1) || (stride == 2 && ks == 3), "stride %d with k=%d not supported", stride, ks);
  EIOKU_REQUIRE(cin % 8 == 0, "cin %d must be a multiple of 8 (NHWC 16-byte units)", cin);
  cw->cout = cout;
  cw->cin = cin;
  cw->ks = ks;
  cw->stride = stride;
  cw->nf = pick_nf(cout, ks);
  const int tile = 16 * cw->nf;
  cw->ntiles = (cout + tile - 1) / tile;
  cw->nchunks = (cin + 31) / 32;
  const int taps = ks * ks;
  const size_t wn = (size_t)cw->ntiles * cw->nchunks * taps * tile * 32;
  std::vector<_Float16> pw(wn, (_Float16)0.f);
  for (int co = 0; co < cout; ++co) {
    const int t = co / tile, cit = co % tile;
    for (int ci = 0; ci < cin; ++ci) {
      const int cc = ci / 32, cic = ci % 32;
      for (int tap = 0; tap < taps; ++tap) {
        const float v = w[((size_t)co * cin + ci) * taps + tap];  // [cout][cin][kh][kw]
        pw[((((size_t)t * cw->nchunks + cc) * taps + tap) * tile + cit) * 32 + cic] = (_Float16)v;
      }
    }
  }
  std::vector<float> pb((size_t)cw->ntiles * tile, 0.f);
  if (b)
    for (int co = 0; co < cout; ++co) pb[co] = b[co];
  EIOKU_HIP_CHECK(hipMalloc((void**)&cw->d_w, wn * sizeof(_Float16)));
  EIOKU_HIP_CHECK(hipMalloc((void**)&cw->d_b, pb.size() * sizeof(float)));
  EIOKU_HIP_CHECK(hipMemcpy(cw->d_w, pw.data(), wn * sizeof(_Float16), hipMemcpyHostToDevice));
  EIOKU_HIP_CHECK(hipMemcpy(cw->d_b, pb.data(), pb.size() * sizeof(float), hipMemcpyHostToDevice));
  return EIOKU_OK;
}

void conv_weights_destroy(ConvWeights* cw) {
  if (cw->d_w) (void)hipFree(cw->d_w);
  if (cw->d_b) (void)hipFree(cw->d_b);
  cw->d_w = nullptr;
  cw->d_b = nullptr;
}

int conv_forward(const ConvWeights& cw, Slice in, int N, int H, int W, Slice out, float* out_f32,
                 Slice res, int act, hipStream_t stream) {
  EIOKU_REQUIRE(cw.d_w, "conv weights not created");
  EIOKU_REQUIRE(in.ptr && (out.ptr || out_f32), "NULL tensor");
  EIOKU_REQUIRE(in.cstride % 8 == 0 && in.coff % 8 == 0, "input slice must be 8-channel aligned");
  // cout < 4 (the 1-class face head) only ever takes the scalar store path
  EIOKU_REQUIRE(out_f32 || cw.cout < 4 || (out.cstride % 4 == 0 && out.coff % 4 == 0),
                "output slice must be 4-channel aligned");
  EIOKU_REQUIRE(!res.ptr || cw.cout < 4 || (res.cstride % 4 == 0 && res.coff % 4 == 0),
                "residual slice must be 4-channel aligned");
  if (N == 0) return EIOKU_OK;
  ConvArgs a;
  a.in = in.ptr + in.coff;
  a.wgt = reinterpret_cast<const uint4*>(cw.d_w);
  a.bias = cw.d_b;
  a.out = out.ptr ? out.ptr + out.coff : nullptr;
  a.out_f32 = out_f32;
  a.res = res.ptr ? res.ptr + res.coff : nullptr;
  a.N = N;
  a.H = H;
  a.W = W;
  a.Cin = cw.cin;
  a.in_cs = in.cstride;
  a.Ho = conv_out_dim(H, cw.ks, cw.stride);
  a.Wo = conv_out_dim(W, cw.ks, cw.stride);
  a.Cout = cw.cout;
  a.out_cs = out.cstride;
  a.res_cs = res.cstride;
  a.tiles_w = (a.Wo + kTW - 1) / kTW;
  a.tiles_h = (a.Ho + kTH - 1) / kTH;
  a.nchunks = cw.nchunks;
  a.act = act;
  prof_start(EIOKU_PROF_CONV, stream);
  int rc;
  if (cw.ks == 3 && cw.stride == 1) rc = launch_nf<3, 1>(cw.nf, a, cw.ntiles, stream);
  else if (cw.ks == 3 && cw.stride == 2) rc = launch_nf<3, 2>(cw.nf, a, cw.ntiles, stream);
  else rc = launch_nf<1, 1>(cw.nf, a, cw.ntiles, stream);
  prof_stop(EIOKU_PROF_CONV, stream);
  return rc;
}

}  // namespace eioku

// ---------------------------------------------------------------------------------------------
// C ABI: a single convolution (parity tests / building block for callers that own their graph)
// ---------------------------------------------------------------------------------------------
using namespace eioku;

extern "C" {

int eioku_conv2d_f16(const void* in_nhwc, int n, int h, int w, int in_cstride, int in_coff, int cin,
                     const float* weight_oihw, const float* bias, int cout, int ksize, int stride,
                     int act_silu, const void* residual, int res_cstride, int res_coff, void* out_nhwc,
                     int out_cstride, int out_coff, float* out_f32, void* stream_) {
  EIOKU_REQUIRE_INIT();
  hipStream_t stream = (hipStream_t)stream_;
  ConvWeights cw;
  int rc = conv_weights_create(&cw, cout, cin, ksize, stride, weight_oihw, bias);
  if (rc) return rc;
  Slice in{(__half*)in_nhwc, in_cstride, in_coff};
  Slice out{(__half*)out_nhwc, out_cstride, out_coff};
  Slice res{(__half*)residual, res_cstride, res_coff};
  rc = conv_forward(cw, in, n, h, w, out, out_f32, res, act_silu ? kActSiLU : kActNone, stream);
  if (rc == EIOKU_OK) {
    hipError_t e = hipStreamSynchronize(stream);
    if (e != hipSuccess) {
      set_error("conv kernel failed: %s", hipGetErrorString(e));
      rc = EIOKU_EHIP;
    }
  }
  conv_weights_destroy(&cw);
  return rc;
}

}  // extern "C"
