// Synthetic inputs generated in HBM: counter-based splitmix64, bit-identical to
// oracle/prng.py, so bench-size inputs (400 MB of frames, 15 GB of vectors) never cross PCIe.
#include "common.h"

using namespace eioku;

namespace {

__global__ void k_synth_u64(unsigned long long seed, unsigned long long offset,
                            unsigned long long n, unsigned long long* out) {
  unsigned long long i = blockIdx.x * (unsigned long long)blockDim.x + threadIdx.x;
  unsigned long long stride = (unsigned long long)gridDim.x * blockDim.x;
  for (; i < n; i += stride) out[i] = splitmix64_at(seed, offset + i);
}

__global__ void k_synth_bytes(unsigned long long seed, unsigned long long n, uint8_t* out) {
  unsigned long long nw = (n + 7) / 8;
  unsigned long long i = blockIdx.x * (unsigned long long)blockDim.x + threadIdx.x;
  unsigned long long stride = (unsigned long long)gridDim.x * blockDim.x;
  for (; i < nw; i += stride) {
    unsigned long long v = splitmix64_at(seed, i);
    if (i * 8 + 8 <= n) {
      *reinterpret_cast<unsigned long long*>(out + i * 8) = v;  // out is hipMalloc-aligned
    } else {
      for (unsigned long long b = i * 8; b < n; ++b) out[b] = (uint8_t)(v >> (8 * (b - i * 8)));
    }
  }
}

__device__ __forceinline__ float irwin_hall4(unsigned long long x) {
  int s = (int)(x & 0xFFFF) + (int)((x >> 16) & 0xFFFF) + (int)((x >> 32) & 0xFFFF) +
          (int)(x >> 48);
  return (float)(s - 131070) * (float)(1.0 / 37837.22);  // same constant as np.float32(1.0/37837.22)
}

// one wave per row; optional L2 normalisation (sum order: lane-strided then xor-tree)
__global__ void k_synth_normal(unsigned long long seed, unsigned long long rows, int dim,
                               int normalise, float* out) {
  const int lane = threadIdx.x & 63;
  unsigned long long row = (blockIdx.x * (unsigned long long)blockDim.x + threadIdx.x) >> 6;
  unsigned long long nwaves = ((unsigned long long)gridDim.x * blockDim.x) >> 6;
  for (; row < rows; row += nwaves) {
    float ss = 0.f;
    for (int c = lane; c < dim; c += 64) {
      float v = irwin_hall4(splitmix64_at(seed, row * dim + c));
      ss += v * v;
    }
    float scale = 1.f;
    if (normalise) {
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) ss += __shfl_xor(ss, off, 64);
      scale = 1.0f / sqrtf(ss);
    }
    for (int c = lane; c < dim; c += 64) {
      float v = irwin_hall4(splitmix64_at(seed, row * dim + c));
      out[row * dim + c] = v * scale;
    }
  }
}

__global__ void k_synth_frames(unsigned long long seed, unsigned long long first_frame, int n,
                               int h, int w, const int32_t* __restrict__ params,
                               uint8_t* __restrict__ out) {
  const unsigned long long per = (unsigned long long)h * w * 3;
  const unsigned long long npix = (unsigned long long)n * h * w;
  unsigned long long p = blockIdx.x * (unsigned long long)blockDim.x + threadIdx.x;
  unsigned long long stride = (unsigned long long)gridDim.x * blockDim.x;
  for (; p < npix; p += stride) {
    unsigned long long i = p / ((unsigned long long)h * w);
    unsigned long long rem = p - i * (unsigned long long)h * w;
    int y = (int)(rem / w), x = (int)(rem - (unsigned long long)y * w);
    const int32_t* pr = params + i * 5;
    int grad = (x * pr[3] + y * pr[4]) >> 10;
    unsigned long long j = (first_frame + i) * per + rem * 3;  // byte index in the stream
    unsigned long long w0 = splitmix64_at(seed, j >> 3);
    unsigned long long w1 = splitmix64_at(seed, (j >> 3) + 1);
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      unsigned long long jj = j + c;
      unsigned long long wv = ((jj >> 3) == (j >> 3)) ? w0 : w1;
      int b = (int)((wv >> (8 * (jj & 7))) & 0xFF);
      int v = pr[c] + grad + (b % 9) - 4;
      v = v < 0 ? 0 : (v > 255 ? 255 : v);
      out[i * per + rem * 3 + c] = (uint8_t)v;
    }
  }
}

inline int grid_for(unsigned long long n, int block) {
  unsigned long long g = (n + block - 1) / block;
  unsigned long long cap = (unsigned long long)num_cus() * 16;
  return (int)(g < 1 ? 1 : (g > cap ? cap : g));
}

}  // namespace

extern "C" {

int eioku_synth_u64(uint64_t seed, uint64_t offset, uint64_t n, uint64_t* out_dev, void* stream) {
  EIOKU_REQUIRE_INIT();
  EIOKU_REQUIRE(out_dev || n == 0, "out_dev is NULL");
  if (n == 0) return EIOKU_OK;
  hipLaunchKernelGGL(k_synth_u64, dim3(grid_for(n, 256)), dim3(256), 0, (hipStream_t)stream, seed,
                     offset, n, (unsigned long long*)out_dev);
  EIOKU_LAUNCH_CHECK();
  return EIOKU_OK;
}

int eioku_synth_bytes(uint64_t seed, uint64_t n, uint8_t* out_dev, void* stream) {
  EIOKU_REQUIRE_INIT();
  EIOKU_REQUIRE(out_dev || n == 0, "out_dev is NULL");
  EIOKU_REQUIRE(((uintptr_t)out_dev & 7) == 0, "out_dev must be 8-byte aligned");
  if (n == 0) return EIOKU_OK;
  hipLaunchKernelGGL(k_synth_bytes, dim3(grid_for((n + 7) / 8, 256)), dim3(256), 0,
                     (hipStream_t)stream, seed, n, out_dev);
  EIOKU_LAUNCH_CHECK();
  return EIOKU_OK;
}

int eioku_synth_normal_f32(uint64_t seed, uint64_t rows, int dim, int l2_normalise, float* out_dev,
                           void* stream) {
  EIOKU_REQUIRE_INIT();
  EIOKU_REQUIRE(dim > 0, "dim must be positive");
  EIOKU_REQUIRE(out_dev || rows == 0, "out_dev is NULL");
  if (rows == 0) return EIOKU_OK;
  hipLaunchKernelGGL(k_synth_normal, dim3(grid_for(rows * 64, 256)), dim3(256), 0,
                     (hipStream_t)stream, seed, rows, dim, l2_normalise, out_dev);
  EIOKU_LAUNCH_CHECK();
  return EIOKU_OK;
}

int eioku_synth_frames_bgr(uint64_t seed, uint64_t first_frame, int n, int h, int w,
                           const int32_t* params_dev, uint8_t* out_dev, void* stream) {
  EIOKU_REQUIRE_INIT();
  EIOKU_REQUIRE(n >= 0 && h > 0 && w > 0, "bad frame shape n=%d h=%d w=%d", n, h, w);
  EIOKU_REQUIRE(params_dev && out_dev, "NULL pointer");
  if (n == 0) return EIOKU_OK;
  hipLaunchKernelGGL(k_synth_frames, dim3(grid_for((uint64_t)n * h * w, 256)), dim3(256), 0,
                     (hipStream_t)stream, seed, first_frame, n, h, w, params_dev, out_dev);
  EIOKU_LAUNCH_CHECK();
  return EIOKU_OK;
}

}  // extern "C"
