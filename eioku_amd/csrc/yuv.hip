// YUV 4:2:0 -> BGR on the device: the colour conversion that cv2's cap.read()
// performs on the CPU in the reference's frame loops (/root/reference/ml-service/src/services/model_manager.py:237-297,
// 331-398).  The single-pass ingest uploads the decoder's planes - 1.5 bytes per pixel over PCIe instead of 3 - scores
// the Y plane directly with K1 (the plane ffmpeg's select filter scores: bit-exact scene scores) and converts to the
// BGR frames K2 and the detectors read with OpenCV's own integer arithmetic (cv2.COLOR_YUV2BGR_I420 / _NV12: BT.601
// studio range, 20-bit fixed point [PUBLIC-LIB]; oracle/yuv.py), so detections are the ones the BGR route gives.
// HBM-bound: 1.5 B read + 3 B written per pixel.  A thread converts a 4 x 2 pixel block (two Y dwords, two U, two V).
#include "common.h"

using namespace eioku;

namespace {

constexpr int kCY = 1220542, kCUB = 2116026, kCUG = -409993, kCVG = -852492, kCVR = 1673527, kShift = 20;

__device__ __forceinline__ unsigned sat8(int v) { return (unsigned)(v < 0 ? 0 : (v > 255 ? 255 : v)); }

// frames: [n][3h/2][w] (OpenCV Mat layout: Y plane, then I420: U plane | V plane, NV12: interleaved UV rows)
template <bool NV12>
__global__ __launch_bounds__(256) void k_yuv420_to_bgr(const uint8_t* __restrict__ yuv, int n, int h, int w,
                                                       uint8_t* __restrict__ bgr) {
  const int w4 = w >> 2, h2 = h >> 1;
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= (long long)n * h2 * w4) return;
  const int bx = (int)(i % w4);
  const int by = (int)((i / w4) % h2);
  const int f = (int)(i / ((long long)w4 * h2));
  const uint8_t* fr = yuv + (size_t)f * (h * 3 / 2) * w;
  const unsigned y0 = *reinterpret_cast<const unsigned*>(fr + (size_t)(2 * by) * w + 4 * bx);
  const unsigned y1 = *reinterpret_cast<const unsigned*>(fr + (size_t)(2 * by + 1) * w + 4 * bx);
  unsigned u2, v2;  // two chroma samples each: low byte = left pair
  if (NV12) {
    const unsigned uv = *reinterpret_cast<const unsigned*>(fr + (size_t)h * w + (size_t)by * w + 4 * bx);  // U0 V0 U1 V1
    u2 = (uv & 0xFFu) | ((uv >> 8) & 0xFF00u);
    v2 = ((uv >> 8) & 0xFFu) | ((uv >> 16) & 0xFF00u);
  } else {
    const uint8_t* up = fr + (size_t)h * w + (size_t)by * (w >> 1) + 2 * bx;
    const uint8_t* vp = up + (size_t)h2 * (w >> 1);
    u2 = *reinterpret_cast<const unsigned short*>(up);
    v2 = *reinterpret_cast<const unsigned short*>(vp);
  }
  uint8_t* o0 = bgr + (((size_t)f * h + 2 * by) * w + 4 * bx) * 3;
  uint8_t* o1 = o0 + (size_t)w * 3;
  unsigned cb[2][4], cg[2][4], cr[2][4];  // the block's 4 x 2 pixels, one channel each
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    const int u = (int)((u2 >> (8 * (p >> 1))) & 0xFFu) - 128, v = (int)((v2 >> (8 * (p >> 1))) & 0xFFu) - 128;
    const int buv = (1 << (kShift - 1)) + kCUB * u;
    const int guv = (1 << (kShift - 1)) + kCUG * u + kCVG * v;
    const int ruv = (1 << (kShift - 1)) + kCVR * v;
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      const int yy = (int)(((r ? y1 : y0) >> (8 * p)) & 0xFFu) - 16;
      const int ys = (yy < 0 ? 0 : yy) * kCY;
      cb[r][p] = sat8((ys + buv) >> kShift);
      cg[r][p] = sat8((ys + guv) >> kShift);
      cr[r][p] = sat8((ys + ruv) >> kShift);
      // opaque to the optimiser from here: without it hipcc (ROCm 7.2, -O3) folded clamp + shift + or into byte-select
      // (SDWA) forms whose untouched destination bytes kept the unclamped sum - pixel 0's R and pixel 1's B came out as
      // bytes 2 / 3 of the green sum (found by the colour-lattice test)
      asm volatile("" : "+v"(cb[r][p]), "+v"(cg[r][p]), "+v"(cr[r][p]));
    }
  }
  // 12 bytes per row = B0 G0 R0 B1 | G1 R1 B2 G2 | R2 B3 G3 R3; rows start 12-byte aligned in a 4-byte aligned buffer
#pragma unroll
  for (int r = 0; r < 2; ++r) {
    unsigned* o = reinterpret_cast<unsigned*>(r ? o1 : o0);
    o[0] = cb[r][0] | (cg[r][0] << 8) | (cr[r][0] << 16) | (cb[r][1] << 24);
    o[1] = cg[r][1] | (cr[r][1] << 8) | (cb[r][2] << 16) | (cg[r][2] << 24);
    o[2] = cr[r][2] | (cb[r][3] << 8) | (cg[r][3] << 16) | (cr[r][3] << 24);
  }
}

}  // namespace

extern "C" {

// yuv: n frames of (3h/2) x w bytes (OpenCV's I420 / NV12 Mat layout), host or device; bgr_out: [n][h][w][3] on the same
// side.  layout 0 = I420 (Y, U, V planes), 1 = NV12 (Y plane, interleaved UV).  h % 2 == 0, w % 4 == 0.
int eioku_yuv420_to_bgr(const uint8_t* yuv, int n, int h, int w, int layout, uint8_t* bgr_out, int mem, void* stream_) {
  EIOKU_REQUIRE_INIT();
  EIOKU_REQUIRE(n >= 0 && h > 0 && w > 0 && h % 2 == 0 && w % 4 == 0, "bad shape n=%d h=%d w=%d (h even, w a multiple of 4)", n, h, w);
  EIOKU_REQUIRE(layout == 0 || layout == 1, "layout %d (0 = I420, 1 = NV12)", layout);
  EIOKU_REQUIRE(mem == EIOKU_MEM_HOST || mem == EIOKU_MEM_DEVICE, "bad mem flag %d", mem);
  if (n == 0) return EIOKU_OK;
  EIOKU_REQUIRE(yuv && bgr_out, "NULL pointer");
  hipStream_t stream = (hipStream_t)stream_;
  const size_t in_b = (size_t)n * (h * 3 / 2) * w, out_b = (size_t)n * h * w * 3;
  const uint8_t* d_in = yuv;
  uint8_t* d_out = bgr_out;
  if (mem == EIOKU_MEM_HOST) {
    uint8_t* a = (uint8_t*)scratch(kSlotIn, in_b);
    uint8_t* b = (uint8_t*)scratch(kSlotOut, out_b);
    if (!a || !b) return EIOKU_ENOMEM;
    EIOKU_HIP_CHECK(hipMemcpyAsync(a, yuv, in_b, hipMemcpyHostToDevice, stream));
    d_in = a;
    d_out = b;
  }
  EIOKU_REQUIRE((((uintptr_t)d_in | (uintptr_t)d_out) & 3) == 0, "buffers must be 4-byte aligned");
  const long long work = (long long)n * (h / 2) * (w / 4);
  if (layout == 1)
    hipLaunchKernelGGL(k_yuv420_to_bgr<true>, dim3((unsigned)((work + 255) / 256)), dim3(256), 0, stream, d_in, n, h, w, d_out);
  else
    hipLaunchKernelGGL(k_yuv420_to_bgr<false>, dim3((unsigned)((work + 255) / 256)), dim3(256), 0, stream, d_in, n, h, w, d_out);
  EIOKU_LAUNCH_CHECK();
  if (mem == EIOKU_MEM_HOST) {
    EIOKU_HIP_CHECK(hipMemcpyAsync(bgr_out, d_out, out_b, hipMemcpyDeviceToHost, stream));
    EIOKU_HIP_CHECK(hipStreamSynchronize(stream));
  }
  return EIOKU_OK;
}

}  // extern "C"
