// Shared host-side plumbing for libeioku_hip.so (gfx950 only).
#pragma once

#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>

#include "../../include/eioku_hip.h"

namespace eioku {

void set_error(const char* fmt, ...);
bool initialised();
int num_cus();

// Scratch device buffer owned by the library, grown on demand (EIOKU_MEM_HOST staging and
// per-call workspaces).  Slot ids keep independent users from aliasing each other.
enum ScratchSlot { kSlotIn = 0, kSlotPrev, kSlotOut, kSlotWork0, kSlotWork1, kSlotWork2, kNumSlots };
void* scratch(ScratchSlot slot, size_t bytes);  // nullptr + error set on failure

// hipEvent brackets around tagged kernel launches (no-ops unless eioku_prof_enable(1)).
void prof_start(int tag, hipStream_t stream);
void prof_stop(int tag, hipStream_t stream);
bool prof_enabled();

}  // namespace eioku

#define EIOKU_HIP_CHECK(expr)                                                                  \
  do {                                                                                         \
    hipError_t _e = (expr);                                                                    \
    if (_e != hipSuccess) {                                                                    \
      ::eioku::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__,      \
                         __LINE__);                                                            \
      return EIOKU_EHIP;                                                                       \
    }                                                                                          \
  } while (0)

#define EIOKU_REQUIRE(cond, ...)          \
  do {                                    \
    if (!(cond)) {                        \
      ::eioku::set_error(__VA_ARGS__);    \
      return EIOKU_EINVAL;                \
    }                                     \
  } while (0)

#define EIOKU_REQUIRE_INIT()                                                \
  do {                                                                      \
    if (!::eioku::initialised()) {                                          \
      ::eioku::set_error("eioku_init() has not been called");               \
      return EIOKU_ENODEV;                                                  \
    }                                                                       \
  } while (0)

// Launch-error check that does not synchronise.
#define EIOKU_LAUNCH_CHECK() EIOKU_HIP_CHECK(hipGetLastError())

namespace eioku {

__device__ __forceinline__ unsigned long long splitmix64_at(unsigned long long seed,
                                                            unsigned long long i) {
  unsigned long long z = seed + (i + 1ull) * 0x9E3779B97F4A7C15ull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}

template <typename T>
__device__ __forceinline__ T wave_reduce_add(T v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
  return v;
}

// Sum of a 32-bit value over the wave on the DPP crossbar: six v_add_u32_dpp and one v_readlane_b32, against ~25
// instructions (ds_bpermute + address + bounds select per step) for the __shfl_down form -- the per-frame reductions
// of the VALU-bound scene kernels.  The result is wave-uniform (an SGPR).
__device__ __forceinline__ unsigned wave_sum_u32(unsigned v) {
  v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xF, 0xF, true);   // quad_perm [1,0,3,2]
  v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E, 0xF, 0xF, true);   // quad_perm [2,3,0,1]
  v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x141, 0xF, 0xF, true);  // row_half_mirror
  v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x140, 0xF, 0xF, true);  // row_mirror: every lane = its row's sum
  v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xA, 0xF, true);  // row_bcast:15 into rows 1 and 3
  v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xC, 0xF, true);  // row_bcast:31 into rows 2 and 3
  return (unsigned)__builtin_amdgcn_readlane((int)v, 63);
}

}  // namespace eioku
