// Lifecycle, error reporting and scratch memory of libeioku_hip.so.
#include "common.h"

#include <string>
#include <utility>
#include <vector>

namespace eioku {

namespace {
thread_local char g_err[512] = "";
bool g_init = false;
int g_device = -1;
int g_cus = 0;
void* g_scratch[kNumSlots] = {};
size_t g_scratch_bytes[kNumSlots] = {};
bool g_prof = false;
unsigned g_prof_mask = ~0u;
std::vector<std::pair<hipEvent_t, hipEvent_t>> g_prof_events[EIOKU_PROF_NUM_TAGS];
size_t g_prof_used[EIOKU_PROF_NUM_TAGS] = {};
}  // namespace

void prof_start(int tag, hipStream_t stream) {
  if (!g_prof || !((g_prof_mask >> tag) & 1u)) return;
  auto& pool = g_prof_events[tag];
  if (g_prof_used[tag] == pool.size()) {
    hipEvent_t a, b;
    if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) return;
    pool.emplace_back(a, b);
  }
  (void)hipEventRecord(pool[g_prof_used[tag]].first, stream);
}

void prof_stop(int tag, hipStream_t stream) {
  if (!g_prof || !((g_prof_mask >> tag) & 1u)) return;
  auto& pool = g_prof_events[tag];
  if (g_prof_used[tag] >= pool.size()) return;
  (void)hipEventRecord(pool[g_prof_used[tag]].second, stream);
  g_prof_used[tag]++;
}

bool prof_enabled() { return g_prof; }

void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

bool initialised() { return g_init; }
int num_cus() { return g_cus; }

void* scratch(ScratchSlot slot, size_t bytes) {
  if (bytes == 0) bytes = 16;
  if (g_scratch_bytes[slot] >= bytes) return g_scratch[slot];
  if (g_scratch[slot]) {
    // the previous buffer may still be in use by queued kernels
    (void)hipDeviceSynchronize();
    (void)hipFree(g_scratch[slot]);
    g_scratch[slot] = nullptr;
    g_scratch_bytes[slot] = 0;
  }
  size_t want = (bytes + (1u << 20) - 1) & ~size_t((1u << 20) - 1);
  void* p = nullptr;
  hipError_t e = hipMalloc(&p, want);
  if (e != hipSuccess) {
    set_error("hipMalloc(%zu) failed: %s", want, hipGetErrorString(e));
    return nullptr;
  }
  g_scratch[slot] = p;
  g_scratch_bytes[slot] = want;
  return p;
}

}  // namespace eioku

using namespace eioku;

extern "C" {

int eioku_abi_version(void) { return EIOKU_ABI_VERSION; }

const char* eioku_last_error(void) { return g_err; }

int eioku_init(int device_id) {
  int count = 0;
  hipError_t e = hipGetDeviceCount(&count);
  if (e != hipSuccess || count == 0) {
    set_error("no HIP device visible (%s)", e == hipSuccess ? "count=0" : hipGetErrorString(e));
    return EIOKU_ENODEV;
  }
  if (device_id < 0 || device_id >= count) {
    set_error("device_id %d out of range [0,%d)", device_id, count);
    return EIOKU_EINVAL;
  }
  EIOKU_HIP_CHECK(hipSetDevice(device_id));
  hipDeviceProp_t prop;
  EIOKU_HIP_CHECK(hipGetDeviceProperties(&prop, device_id));
  if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
    set_error("device %d is %s; this library is built for gfx950 only", device_id,
              prop.gcnArchName);
    return EIOKU_ENODEV;
  }
  g_cus = prop.multiProcessorCount;
  g_device = device_id;
  g_init = true;
  return EIOKU_OK;
}

void eioku_shutdown(void) {
  if (!g_init) return;
  (void)hipDeviceSynchronize();
  for (int i = 0; i < kNumSlots; ++i) {
    if (g_scratch[i]) (void)hipFree(g_scratch[i]);
    g_scratch[i] = nullptr;
    g_scratch_bytes[i] = 0;
  }
  g_init = false;
}

int eioku_prof_enable(int on) {
  // 0: off; 1: every tag; otherwise a mask: bit (tag + 1) enables that tag's brackets only
  g_prof = on != 0;
  g_prof_mask = on == 1 ? ~0u : ((unsigned)on >> 1);
  return EIOKU_OK;
}

int eioku_prof_reset(void) {
  for (int t = 0; t < EIOKU_PROF_NUM_TAGS; ++t) g_prof_used[t] = 0;
  return EIOKU_OK;
}

int eioku_prof_read(int tag, double* total_ms, uint64_t* launches) {
  EIOKU_REQUIRE(tag >= 0 && tag < EIOKU_PROF_NUM_TAGS, "bad prof tag %d", tag);
  double total = 0;
  for (size_t i = 0; i < g_prof_used[tag]; ++i) {
    auto& ev = g_prof_events[tag][i];
    EIOKU_HIP_CHECK(hipEventSynchronize(ev.second));
    float ms = 0;
    EIOKU_HIP_CHECK(hipEventElapsedTime(&ms, ev.first, ev.second));
    total += ms;
  }
  if (total_ms) *total_ms = total;
  if (launches) *launches = g_prof_used[tag];
  return EIOKU_OK;
}

int eioku_device_info(char* name, size_t name_cap, int* compute_units, uint64_t* hbm_bytes) {
  EIOKU_REQUIRE_INIT();
  hipDeviceProp_t prop;
  EIOKU_HIP_CHECK(hipGetDeviceProperties(&prop, g_device));
  if (name && name_cap) {
    snprintf(name, name_cap, "%s (%s)", prop.name, prop.gcnArchName);
  }
  if (compute_units) *compute_units = prop.multiProcessorCount;
  if (hbm_bytes) *hbm_bytes = prop.totalGlobalMem;
  return EIOKU_OK;
}

}  // extern "C"
