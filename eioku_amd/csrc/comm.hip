// C1 through the C ABI: a row-sharded exact index answers a search with the local shard search, ONE RCCL all-gather
// of the packed per-rank (D, I) answers over xGMI, and a local merge (k_topk_merge) -- SURVEY.md 8(e) row 2,
// BASELINE.json north_star "thin C-ABI ... single RCCL all-gather".  eioku_amd/search.py::ShardedFlatL2 does the same
// through torch.distributed; this file is for callers that own no torch process group (a C++ / Go service binding
// the library directly).
//
// RCCL is bound at run time (dlopen, preferring a copy already loaded in the process: torch ships its own
// librccl.so and two RCCL instances in one process are a waste of a bootstrap thread each), so libeioku_hip.so
// itself has no link-time dependency on it and single-GPU users never touch it.
#include <dlfcn.h>
#include <rccl/rccl.h>

#include <mutex>

#include "common.h"

namespace {

using namespace eioku;

struct Rccl {
  void* handle = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*CommAbort)(ncclComm_t) = nullptr;
  ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
  bool ok = false;
};

Rccl* rccl() {
  static Rccl r;
  static std::once_flag once;
  std::call_once(once, [] {
    const char* names[] = {"librccl.so", "librccl.so.1"};
    for (const char* n : names)
      if (!r.handle) r.handle = dlopen(n, RTLD_NOW | RTLD_NOLOAD | RTLD_GLOBAL);  // the process' own copy first
    for (const char* n : names)
      if (!r.handle) r.handle = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
    if (!r.handle) return;
    r.GetUniqueId = reinterpret_cast<decltype(r.GetUniqueId)>(dlsym(r.handle, "ncclGetUniqueId"));
    r.CommInitRank = reinterpret_cast<decltype(r.CommInitRank)>(dlsym(r.handle, "ncclCommInitRank"));
    r.CommDestroy = reinterpret_cast<decltype(r.CommDestroy)>(dlsym(r.handle, "ncclCommDestroy"));
    r.CommAbort = reinterpret_cast<decltype(r.CommAbort)>(dlsym(r.handle, "ncclCommAbort"));  // optional
    r.AllGather = reinterpret_cast<decltype(r.AllGather)>(dlsym(r.handle, "ncclAllGather"));
    r.GetErrorString = reinterpret_cast<decltype(r.GetErrorString)>(dlsym(r.handle, "ncclGetErrorString"));
    r.ok = r.GetUniqueId && r.CommInitRank && r.CommDestroy && r.AllGather && r.GetErrorString;
  });
  return &r;
}

#define EIOKU_RCCL_REQUIRE()                                                                      \
  Rccl* R = rccl();                                                                                \
  do {                                                                                             \
    if (!R->ok) {                                                                                  \
      set_error("RCCL is not available: %s", R->handle ? "missing symbols" : "librccl.so not found"); \
      return EIOKU_ENODEV;                                                                         \
    }                                                                                              \
  } while (0)

#define EIOKU_RCCL_CHECK(expr)                                                             \
  do {                                                                                     \
    ncclResult_t _r = (expr);                                                              \
    if (_r != ncclSuccess) {                                                               \
      set_error("%s failed: %s (%s:%d)", #expr, R->GetErrorString(_r), __FILE__, __LINE__); \
      return EIOKU_EHIP;                                                                   \
    }                                                                                      \
  } while (0)

// One rank's answer as a single message: [nq*k] float distances, padded to 8 bytes, then [nq*k] int64 global ids.
__host__ __device__ inline size_t payload_words(long long n) { return (size_t)((n + 1) / 2 + n); }  // int64 words

// a rank whose own search failed still owes its peers a message: all padding (FLT_MAX / -1), as an empty shard answers
__global__ void k_pad_answer(long long n, long long* __restrict__ out) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  reinterpret_cast<float*>(out)[i] = 3.402823466e+38f;
  if (i == n - 1 && (n & 1)) reinterpret_cast<float*>(out)[n] = 0.f;
  out[(n + 1) / 2 + i] = -1;
}

__global__ void k_pack_answer(const float* __restrict__ D, const long long* __restrict__ I, long long n,
                              long long id_base, long long* __restrict__ out) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  reinterpret_cast<float*>(out)[i] = D[i];
  if (i == n - 1 && (n & 1)) reinterpret_cast<float*>(out)[n] = 0.f;  // the pad word travels too: keep it defined
  const long long id = I[i];
  out[(n + 1) / 2 + i] = id < 0 ? id : id + id_base;
}

// [world] messages -> the merge kernel's contiguous [world][nq*k] distance and id lists
__global__ void k_unpack_answers(const long long* __restrict__ in, int world, long long n, float* __restrict__ dl,
                                 long long* __restrict__ il) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n * world) return;
  const long long r = i / n, j = i - r * n;
  const long long* msg = in + (size_t)r * payload_words(n);
  dl[i] = reinterpret_cast<const float*>(msg)[j];
  il[i] = msg[(n + 1) / 2 + j];
}

}  // namespace

struct eioku_comm {
  ncclComm_t comm = nullptr;
  int rank = 0, world = 1;
  // per-communicator workspace, grown on demand
  float* Dloc = nullptr; long long* Iloc = nullptr; size_t loc_cap = 0;
  long long* send = nullptr; long long* recv = nullptr; size_t msg_cap = 0;
  float* dl = nullptr; long long* il = nullptr; size_t list_cap = 0;
};

extern "C" {

int eioku_comm_unique_id(unsigned char* id128) {
  EIOKU_REQUIRE(id128, "NULL id buffer");
  EIOKU_RCCL_REQUIRE();
  ncclUniqueId id;
  EIOKU_RCCL_CHECK(R->GetUniqueId(&id));
  static_assert(sizeof(id) == EIOKU_COMM_ID_BYTES, "ncclUniqueId size");
  memcpy(id128, &id, sizeof(id));
  return EIOKU_OK;
}

int eioku_comm_create(const unsigned char* id128, int rank, int world, eioku_comm_t** out) {
  EIOKU_REQUIRE_INIT();
  EIOKU_REQUIRE(id128 && out, "NULL argument");
  EIOKU_REQUIRE(world >= 1 && rank >= 0 && rank < world, "rank %d / world %d out of range", rank, world);
  EIOKU_RCCL_REQUIRE();
  ncclUniqueId id;
  memcpy(&id, id128, sizeof(id));
  auto* c = new eioku_comm();
  c->rank = rank;
  c->world = world;
  const ncclResult_t r = R->CommInitRank(&c->comm, world, id, rank);  // collective: every rank calls it
  if (r != ncclSuccess) {
    set_error("ncclCommInitRank(rank %d of %d) failed: %s", rank, world, R->GetErrorString(r));
    delete c;
    return EIOKU_EHIP;
  }
  *out = c;
  return EIOKU_OK;
}

void eioku_comm_destroy(eioku_comm_t* c) {
  if (!c) return;
  (void)hipDeviceSynchronize();
  if (c->comm && rccl()->ok) (void)rccl()->CommDestroy(c->comm);
  for (void* p : {(void*)c->Dloc, (void*)c->Iloc, (void*)c->send, (void*)c->recv, (void*)c->dl, (void*)c->il})
    if (p) (void)hipFree(p);
  delete c;
}

int eioku_comm_rank(const eioku_comm_t* c, int* rank, int* world) {
  EIOKU_REQUIRE(c, "NULL communicator");
  if (rank) *rank = c->rank;
  if (world) *world = c->world;
  return EIOKU_OK;
}

int eioku_index_search_sharded(eioku_index_t* ix, eioku_comm_t* c, long long id_base, const float* q, int nq, int k,
                               float* D, int64_t* I, void* stream_) {
  EIOKU_REQUIRE_INIT();
  EIOKU_REQUIRE(ix && c, "NULL index / communicator");
  EIOKU_REQUIRE(nq >= 0 && k >= 1 && k <= 32, "nq %d / k %d out of range (1 <= k <= 32)", nq, k);
  EIOKU_REQUIRE(id_base >= 0, "id_base %lld is negative", id_base);
  if (nq == 0) return EIOKU_OK;  // every rank passes the same nq: nobody enters the collective
  EIOKU_REQUIRE(q && D && I, "NULL buffer");
  EIOKU_RCCL_REQUIRE();
  EIOKU_REQUIRE(c->comm, "the communicator was aborted by an earlier failure");
  hipStream_t stream = (hipStream_t)stream_;
  const long long n = (long long)nq * k;
  const size_t words = payload_words(n);
  // Workspace first.  A rank that cannot even allocate its message cannot take part in the collective its peers are
  // about to enter: it aborts the communicator (ncclCommAbort fails the peers' pending all-gather instead of leaving
  // them blocked in it) and reports the error.  [The world > 1 path of this entry has not run on hardware yet: one-GPU
  // leases only; tests/test_comm_gpu.py forms a world of one.]
  auto grow_all = [&]() -> int {
    if (c->loc_cap < (size_t)n) {
      (void)hipStreamSynchronize(stream);
      if (c->Dloc) (void)hipFree(c->Dloc);
      if (c->Iloc) (void)hipFree(c->Iloc);
      c->Dloc = nullptr;
      c->Iloc = nullptr;
      c->loc_cap = 0;
      EIOKU_HIP_CHECK(hipMalloc((void**)&c->Dloc, (size_t)n * sizeof(float)));
      EIOKU_HIP_CHECK(hipMalloc((void**)&c->Iloc, (size_t)n * sizeof(long long)));
      c->loc_cap = (size_t)n;
    }
    if (c->msg_cap < words) {
      (void)hipStreamSynchronize(stream);
      if (c->send) (void)hipFree(c->send);
      if (c->recv) (void)hipFree(c->recv);
      c->send = c->recv = nullptr;
      c->msg_cap = 0;
      EIOKU_HIP_CHECK(hipMalloc((void**)&c->send, words * sizeof(long long)));
      EIOKU_HIP_CHECK(hipMalloc((void**)&c->recv, words * sizeof(long long) * (size_t)c->world));
      c->msg_cap = words;
    }
    if (c->list_cap < (size_t)n * c->world) {
      (void)hipStreamSynchronize(stream);
      if (c->dl) (void)hipFree(c->dl);
      if (c->il) (void)hipFree(c->il);
      c->dl = nullptr;
      c->il = nullptr;
      c->list_cap = 0;
      EIOKU_HIP_CHECK(hipMalloc((void**)&c->dl, (size_t)n * c->world * sizeof(float)));
      EIOKU_HIP_CHECK(hipMalloc((void**)&c->il, (size_t)n * c->world * sizeof(long long)));
      c->list_cap = (size_t)n * c->world;
    }
    return EIOKU_OK;
  };
  int rc = grow_all();
  if (rc != EIOKU_OK) {
    if (c->world > 1 && c->comm && R->CommAbort) {
      (void)R->CommAbort(c->comm);
      c->comm = nullptr;
    }
    return rc;
  }
  // 1. this rank's shard (an empty shard answers with padding: FLT_MAX / -1, as FAISS pads).  A search that fails
  // locally must not strand the peers: this rank still sends a message - all padding - and returns its error after
  // the collective, so the others complete (with this shard missing from their answer) and the caller sees the failure.
  const int search_rc = eioku_index_search(ix, q, nq, k, c->Dloc, (int64_t*)c->Iloc, EIOKU_MEM_DEVICE, stream_);
  char search_err[512] = "";
  if (search_rc != EIOKU_OK) snprintf(search_err, sizeof search_err, "%s", eioku_last_error());
  // 2. one message per rank, one all-gather
  const unsigned blocks = (unsigned)((n + 255) / 256);
  if (search_rc == EIOKU_OK)
    hipLaunchKernelGGL(k_pack_answer, dim3(blocks), dim3(256), 0, stream, c->Dloc, c->Iloc, n, id_base, c->send);
  else
    hipLaunchKernelGGL(k_pad_answer, dim3(blocks), dim3(256), 0, stream, n, c->send);
  EIOKU_LAUNCH_CHECK();
  EIOKU_RCCL_CHECK(R->AllGather(c->send, c->recv, words, ncclInt64, c->comm, stream));
  if (search_rc != EIOKU_OK) {
    set_error("local shard search failed (the collective was completed with a padding message): %s", search_err);
    return search_rc;
  }
  // 3. local merge of world x k candidates per query: every rank ends with the same answer
  const unsigned ublocks = (unsigned)((n * c->world + 255) / 256);
  hipLaunchKernelGGL(k_unpack_answers, dim3(ublocks), dim3(256), 0, stream, c->recv, c->world, n, c->dl, c->il);
  EIOKU_LAUNCH_CHECK();
  return eioku_topk_merge(c->dl, (const int64_t*)c->il, c->world, nq, k, D, I, stream_);
}

}  // extern "C"
