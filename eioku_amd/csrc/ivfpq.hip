// K10: IVF-PQ building blocks (FAISS IndexIVFPQ semantics, L2, by_residual, nbits = 8).
//
// The reference holds intent only for vector search (.kiro/specs/semantic-video-search/design.md:
// 35-40); BASELINE.json cfg5 names IndexIVF-PQ (100M x 384, nlist 4096).  Host orchestration (training
// loops, list bookkeeping) is Python (eioku_amd/ivfpq.py); the data-parallel steps are here:
//
//   k_kmeans_accumulate  per-cluster sums in 64-bit FIXED POINT (2^-32): integer atomics are
//                        associative, so centroids are bit-reproducible run to run (float atomics are not)
//   k_kmeans_finalize    sum / count -> centroid (empty cluster keeps its previous centroid)
//   k_pq_assign          nearest of 256 sub-centroids per (vector, sub-quantiser) on x - coarse[list]
//                        (training assignment AND encoding); sub-codebook in LDS
//   k_ivf_histogram / k_ivf_scatter   counting sort of vectors into inverted lists
//   k_ivfpq_scan         per (query, probe): residual LUT (m x 256) in LDS, ADC sum over the list's
//                        codes (HBM-bound: m bytes per code), per-thread top-k, block merge
//
// The coarse assignment / probe selection reuses K9 (k_flat_l2 with k = 1 / nprobe).
#include "common.h"

#include <cfloat>

using namespace eioku;

namespace {

constexpr double kFix = 4294967296.0;  // 2^32

__global__ __launch_bounds__(256) void k_kmeans_accumulate(const float* __restrict__ x, long long n, int d,
                                                           const long long* __restrict__ assign,
                                                           long long* __restrict__ sums, int* __restrict__ counts) {
  // one wave per row: lanes stride the dims
  const int lane = threadIdx.x & 63;
  long long row = (blockIdx.x * 256ll + threadIdx.x) >> 6;
  const long long nw = ((long long)gridDim.x * 256) >> 6;
  for (; row < n; row += nw) {
    const long long c = assign[row];
    if (c < 0) continue;
    for (int j = lane; j < d; j += 64) {
      const long long q = __double2ll_rn((double)x[(size_t)row * d + j] * kFix);
      atomicAdd(reinterpret_cast<unsigned long long*>(&sums[(size_t)c * d + j]), (unsigned long long)q);
    }
    if (lane == 0) atomicAdd(&counts[c], 1);
  }
}

__global__ __launch_bounds__(256) void k_kmeans_finalize(const long long* __restrict__ sums, const int* __restrict__ counts,
                                                         int k, int d, float* __restrict__ centroids) {
  const long long i = blockIdx.x * 256ll + threadIdx.x;
  if (i >= (long long)k * d) return;
  const int c = (int)(i / d);
  if (counts[c] > 0) centroids[i] = (float)(((double)sums[i] / kFix) / (double)counts[c]);
}

// x: [n][d] ; coarse (optional): [nlist][d] with list[n] -> residual = x - coarse[list]
// pq: [m][256][dsub] ; codes: [n][m] uint8.  One thread per (vector, sub-quantiser).
template <int DSUB>
__global__ __launch_bounds__(256) void k_pq_assign(const float* __restrict__ x, long long n, int d, int m,
                                                   const float* __restrict__ coarse, const long long* __restrict__ list,
                                                   const float* __restrict__ pq, uint8_t* __restrict__ codes,
                                                   float* __restrict__ resid_out) {
  __shared__ float cb[256 * DSUB];
  const int j = blockIdx.y;  // sub-quantiser
  for (int i = threadIdx.x; i < 256 * DSUB; i += 256) cb[i] = pq[(size_t)j * 256 * DSUB + i];
  __syncthreads();
  long long row = blockIdx.x * 256ll + threadIdx.x;
  const long long stride = (long long)gridDim.x * 256;
  for (; row < n; row += stride) {
    float v[DSUB];
#pragma unroll
    for (int t = 0; t < DSUB; ++t) v[t] = x[(size_t)row * d + j * DSUB + t];
    if (coarse) {
      const long long l = list[row];
#pragma unroll
      for (int t = 0; t < DSUB; ++t) v[t] = v[t] - coarse[(size_t)l * d + j * DSUB + t];
    }
    if (resid_out) {
#pragma unroll
      for (int t = 0; t < DSUB; ++t) resid_out[(size_t)row * d + j * DSUB + t] = v[t];
    }
    float best = FLT_MAX;
    int bi = 0;
    for (int c = 0; c < 256; ++c) {
      float s = 0.f;
#pragma unroll
      for (int t = 0; t < DSUB; ++t) {
        const float df = v[t] - cb[c * DSUB + t];
        s += df * df;
      }
      if (s < best) {  // first minimum wins
        best = s;
        bi = c;
      }
    }
    if (codes) codes[(size_t)row * m + j] = (uint8_t)bi;
  }
}

__global__ __launch_bounds__(256) void k_ivf_histogram(const long long* __restrict__ list, long long n, int* __restrict__ counts) {
  long long i = blockIdx.x * 256ll + threadIdx.x;
  const long long stride = (long long)gridDim.x * 256;
  for (; i < n; i += stride) atomicAdd(&counts[list[i]], 1);
}

// offsets: exclusive prefix of counts (host-computed: nlist is small); cursor starts as a copy of offsets
__global__ __launch_bounds__(256) void k_ivf_scatter(const long long* __restrict__ list, long long n, int m,
                                                     const uint8_t* __restrict__ codes, long long id_base,
                                                     int* __restrict__ cursor, uint8_t* __restrict__ list_codes,
                                                     long long* __restrict__ list_ids) {
  long long i = blockIdx.x * 256ll + threadIdx.x;
  const long long stride = (long long)gridDim.x * 256;
  for (; i < n; i += stride) {
    const int pos = atomicAdd(&cursor[list[i]], 1);
    for (int j = 0; j < m; ++j) list_codes[(size_t)pos * m + j] = codes[(size_t)i * m + j];
    list_ids[pos] = id_base + i;
  }
}

// ---- scan -------------------------------------------------------------------------------------------
template <int K>
struct SmallTop {
  float v[K];
  long long id[K];
  __device__ __forceinline__ void init() {
#pragma unroll
    for (int p = 0; p < K; ++p) {
      v[p] = FLT_MAX;
      id[p] = -1;
    }
  }
  __device__ __forceinline__ bool better(float x, long long i, float y, long long j) const {
    return x < y || (x == y && (j < 0 || i < j));
  }
  __device__ __forceinline__ void insert(float x, long long i) {
    if (!better(x, i, v[K - 1], id[K - 1])) return;
#pragma unroll
    for (int p = K - 1; p > 0; --p) {
      const bool shift = better(x, i, v[p - 1], id[p - 1]);
      const bool here = !shift && better(x, i, v[p], id[p]);
      const float nv = shift ? v[p - 1] : (here ? x : v[p]);
      const long long ni = shift ? id[p - 1] : (here ? i : id[p]);
      v[p] = nv;
      id[p] = ni;
    }
    if (better(x, i, v[0], id[0])) {
      v[0] = x;
      id[0] = i;
    }
  }
};

// FAISS' precomputed-table decomposition of the ADC look-up table (IndexIVFPQ::use_precomputed_table = 1):
//   || (q - c_l)_j - p_jc ||^2 = || (q - c_l)_j ||^2  +  ( ||p_jc||^2 + 2 c_l,j . p_jc )  +  ( -2 q_j . p_jc )
// the middle term depends on (list, j, c) only - one table per index, built once - the last on (query, j, c) only -
// one table per query and search - and the first sums over j to ||q - c_l||^2, a scalar per (query, list).  The
// per-(query, list) table is then m x 256 additions of two streamed rows instead of m x 256 x dsub multiply-adds over
// the whole PQ codebook (393 KB of L2 reads per workgroup at m = 48, dsub = 8: it cost as much as scanning the list).
// out[v][j][c] = alpha ||p_jc||^2 + beta (vecs[v]_j . p_jc);  grid (nvec), block 256
template <int DSUB>
__global__ __launch_bounds__(256) void k_ivfpq_tables(const float* __restrict__ vecs, int d, int m,
                                                      const float* __restrict__ pq, float alpha, float beta,
                                                      float* __restrict__ out) {
  extern __shared__ __attribute__((aligned(16))) float sv[];  // the vector
  const int v = blockIdx.x, tid = threadIdx.x;
  for (int t = tid; t < d; t += 256) sv[t] = vecs[(size_t)v * d + t];
  __syncthreads();
  typedef float f32x4 __attribute__((ext_vector_type(4)));
  constexpr int V4 = DSUB / 4;
  const int nent = m * 256;
  for (int e = tid; e < nent; e += 256) {
    const int j = e >> 8;
    float nn = 0.f, dp = 0.f;
#pragma unroll
    for (int t4 = 0; t4 < V4; ++t4) {
      const f32x4 p = *reinterpret_cast<const f32x4*>(pq + (size_t)e * DSUB + t4 * 4);
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        nn += p[t] * p[t];
        dp += sv[j * DSUB + t4 * 4 + t] * p[t];
      }
    }
    out[(size_t)v * nent + e] = alpha * nn + beta * dp;
  }
}

// grid (nq, nprobe); block 256.  probes: [nq][nprobe] list ids (-1 = none).
// out: pd/pi [nprobe][nq][K]  (the [list][nq][k] layout k_topk_merge takes)
template <int K, int DSUB, bool PRE = false>
__global__ __launch_bounds__(256) void k_ivfpq_scan(const float* __restrict__ q, int nq, int d, int m,
                                                    const long long* __restrict__ probes, int nprobe,
                                                    const float* __restrict__ coarse, const float* __restrict__ pq,
                                                    const int* __restrict__ offsets, const int* __restrict__ sizes,
                                                    const uint8_t* __restrict__ list_codes,
                                                    const long long* __restrict__ list_ids, float* __restrict__ pd,
                                                    long long* __restrict__ pi, const float* __restrict__ t2 = nullptr,
                                                    const float* __restrict__ t3 = nullptr) {
  extern __shared__ __attribute__((aligned(16))) float lut[];  // [m][256], then merge area
  const int qi = blockIdx.x, pr = blockIdx.y, tid = threadIdx.x;
  const long long l = probes[(size_t)qi * nprobe + pr];
  SmallTop<K> top;
  top.init();
  if (l >= 0) {
    // residual query r = q - coarse[l] once, in LDS (behind the LUT; the merge area reuses both later)
    float* rq = lut + (size_t)m * 256;
    for (int t = tid; t < d; t += 256) rq[t] = q[(size_t)qi * d + t] - coarse[(size_t)l * d + t];
    __syncthreads();
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    const int nent = m * 256;
    float term1 = 0.f;  // PRE: || q - c_l ||^2, added to every distance of the list
    if (PRE) {
      // LUT[e] = T2[l][e] + T3[q][e]: two coalesced rows, 16 bytes per lane and load
      __shared__ float s_part[4];
      const f32x4* a = reinterpret_cast<const f32x4*>(t2 + (size_t)l * nent);
      const f32x4* b = reinterpret_cast<const f32x4*>(t3 + (size_t)qi * nent);
      for (int e4 = tid; e4 < nent / 4; e4 += 256) reinterpret_cast<f32x4*>(lut)[e4] = a[e4] + b[e4];
      float part = 0.f;
      for (int t = tid; t < d; t += 256) part += rq[t] * rq[t];
      part = wave_reduce_add(part);
      if ((tid & 63) == 0) s_part[tid >> 6] = part;
      __syncthreads();
      term1 = (s_part[0] + s_part[1]) + (s_part[2] + s_part[3]);
    } else {
    // LUT[j][c] = || r_j - pq[j][c] ||^2 : 16-byte loads, LB entries in flight per thread (one entry per loop
    // iteration was 48 dependent L2 round trips per workgroup), same summation order as before
    constexpr int V4 = DSUB / 4, LB = 8;
    for (int e0 = 0; e0 < nent; e0 += 256 * LB) {
      f32x4 rows[LB][V4];
#pragma unroll
      for (int b = 0; b < LB; ++b) {
        int e = e0 + b * 256 + tid;
        if (e >= nent) e = nent - 1;
#pragma unroll
        for (int t4 = 0; t4 < V4; ++t4) rows[b][t4] = *reinterpret_cast<const f32x4*>(pq + (size_t)e * DSUB + t4 * 4);
      }
#pragma unroll
      for (int b = 0; b < LB; ++b) {
        const int e = e0 + b * 256 + tid;
        const int j = (e < nent ? e : nent - 1) >> 8;
        float s = 0.f;
#pragma unroll
        for (int t4 = 0; t4 < V4; ++t4) {
          const float* r = rq + j * DSUB + t4 * 4;
#pragma unroll
          for (int t = 0; t < 4; ++t) {
            const float df = r[t] - rows[b][t4][t];
            s += df * df;
          }
        }
        if (e < nent) lut[e] = s;
      }
    }
    }
    __syncthreads();
    const int off = offsets[l], sz = sizes[l];
    const bool vec_codes = (m & 15) == 0 && m <= 64 && ((size_t)off * m & 15) == 0;  // rows stay 16-byte aligned
    if (vec_codes) {
      // codes and id of the NEXT vector are in flight while the current one is looked up
      typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
      constexpr int MAXW = 4;  // m <= 64
      const int nw = m >> 4;
      u32x4 cn[MAXW];
      long long idn = -1;
      auto fetch = [&](int i) {
        const int ii = i < sz ? i : sz - 1;
        const uint8_t* code = list_codes + (size_t)(off + ii) * m;
#pragma unroll
        for (int w = 0; w < MAXW; ++w)
          cn[w] = *reinterpret_cast<const u32x4*>(code + (w < nw ? w : 0) * 16);
        idn = list_ids[off + ii];
      };
      if (sz > 0) fetch(tid);
      for (int i = tid; i < sz; i += 256) {
        u32x4 c[MAXW];
#pragma unroll
        for (int w = 0; w < MAXW; ++w) c[w] = cn[w];
        const long long id = idn;
        fetch(i + 256);
        float s = 0.f;
#pragma unroll
        for (int w = 0; w < MAXW; ++w) {
          if (w < nw) {
#pragma unroll
            for (int b = 0; b < 16; ++b) s += lut[(w * 16 + b) * 256 + ((c[w][b >> 2] >> (8 * (b & 3))) & 0xFFu)];
          }
        }
        top.insert(PRE ? s + term1 : s, id);
      }
    } else {
      for (int i = tid; i < sz; i += 256) {
        const uint8_t* code = list_codes + (size_t)(off + i) * m;
        float s = 0.f;
        for (int j = 0; j < m; ++j) s += lut[j * 256 + code[j]];
        top.insert(PRE ? s + term1 : s, list_ids[off + i]);
      }
    }
  }
  __syncthreads();
  // block merge: every thread publishes its list, thread 0..K-1 rounds of block argmin are overkill for
  // K <= 32: serial K-way pick by one wave over 256 heads
  float* mv = lut;                                                   // [256][K]
  long long* mi = reinterpret_cast<long long*>(lut + 256 * K);       // [256][K]
#pragma unroll
  for (int p = 0; p < K; ++p) {
    mv[tid * K + p] = top.v[p];
    mi[tid * K + p] = top.id[p];
  }
  __syncthreads();
  if (tid < 64) {
    int head[4] = {0, 0, 0, 0};  // this lane owns lists tid, tid+64, tid+128, tid+192
    for (int r = 0; r < K; ++r) {
      float bv = FLT_MAX;
      long long bi = -1;
      int bs = 0;
#pragma unroll
      for (int s4 = 0; s4 < 4; ++s4) {
        if (head[s4] < K) {
          const float v = mv[(tid + 64 * s4) * K + head[s4]];
          const long long i = mi[(tid + 64 * s4) * K + head[s4]];
          if (i >= 0 && (v < bv || (v == bv && (bi < 0 || i < bi)))) {
            bv = v;
            bi = i;
            bs = s4;
          }
        }
      }
      // wave argmin
      float wv = bv;
      long long wi = bi < 0 ? 0x7FFFFFFFFFFFFFFFll : bi;
      int wl = tid;
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) {
        const float ov = __shfl_xor(wv, o, 64);
        const long long oi = __shfl_xor(wi, o, 64);
        const int ol = __shfl_xor(wl, o, 64);
        if (ov < wv || (ov == wv && oi < wi) || (ov == wv && oi == wi && ol < wl)) {
          wv = ov;
          wi = oi;
          wl = ol;
        }
      }
      const bool found = wi != 0x7FFFFFFFFFFFFFFFll;
      if (tid == 0) {
        pd[((size_t)pr * nq + qi) * K + r] = found ? wv : FLT_MAX;
        pi[((size_t)pr * nq + qi) * K + r] = found ? wi : -1;
      }
      if (found && tid == wl) {
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4)
          if (s4 == bs) head[s4]++;
      }
    }
  }
}

inline unsigned grid_cap(long long work, int per_block) {
  long long g = (work + per_block - 1) / per_block;
  const long long cap = (long long)num_cus() * 16;
  if (g > cap) g = cap;
  if (g < 1) g = 1;
  return (unsigned)g;
}

}  // namespace

extern "C" {

// centroids[k][d] <- mean of the rows assigned to each cluster (assign[i] in [0,k) or -1); clusters
// without members keep their current centroid.  counts_out[k] (optional, device) receives the sizes.
int eioku_kmeans_update(const float* x_dev, long long n, int d, const long long* assign_dev, int k,
                        float* centroids_dev, int* counts_out_dev, void* stream_) {
  EIOKU_REQUIRE_INIT();
  EIOKU_REQUIRE(x_dev && assign_dev && centroids_dev && n >= 0 && d > 0 && k > 0, "bad argument");
  hipStream_t stream = (hipStream_t)stream_;
  long long* sums = (long long*)scratch(kSlotWork0, (size_t)k * d * 8);
  int* counts = (int*)scratch(kSlotWork1, (size_t)k * 4);
  if (!sums || !counts) return EIOKU_ENOMEM;
  EIOKU_HIP_CHECK(hipMemsetAsync(sums, 0, (size_t)k * d * 8, stream));
  EIOKU_HIP_CHECK(hipMemsetAsync(counts, 0, (size_t)k * 4, stream));
  if (n > 0) {
    hipLaunchKernelGGL(k_kmeans_accumulate, dim3(grid_cap(n * 64, 256)), dim3(256), 0, stream, x_dev, n, d, assign_dev,
                       sums, counts);
    EIOKU_LAUNCH_CHECK();
  }
  hipLaunchKernelGGL(k_kmeans_finalize, dim3((unsigned)(((long long)k * d + 255) / 256)), dim3(256), 0, stream, sums,
                     counts, k, d, centroids_dev);
  EIOKU_LAUNCH_CHECK();
  if (counts_out_dev)
    EIOKU_HIP_CHECK(hipMemcpyAsync(counts_out_dev, counts, (size_t)k * 4, hipMemcpyDeviceToDevice, stream));
  return EIOKU_OK;
}

// The two halves of eioku_kmeans_update, for a build sharded over GPUs: every rank ADDS its rows into sums / counts
// (caller-zeroed int64 [k][d] in 2^-32 fixed point / int32 [k]), the ranks all-reduce the integers (exact and order
// independent: the sharded centroids are the single-GPU ones bit for bit) and every rank finalises.
int eioku_kmeans_accumulate(const float* x_dev, long long n, int d, const long long* assign_dev, int k,
                            long long* sums_dev, int* counts_dev, void* stream_) {
  EIOKU_REQUIRE_INIT();
  EIOKU_REQUIRE(x_dev && assign_dev && sums_dev && counts_dev && n >= 0 && d > 0 && k > 0, "bad argument");
  if (n == 0) return EIOKU_OK;
  hipLaunchKernelGGL(k_kmeans_accumulate, dim3(grid_cap(n * 64, 256)), dim3(256), 0, (hipStream_t)stream_, x_dev, n, d,
                     assign_dev, sums_dev, counts_dev);
  EIOKU_LAUNCH_CHECK();
  return EIOKU_OK;
}

int eioku_kmeans_finalize(const long long* sums_dev, const int* counts_dev, int k, int d, float* centroids_dev,
                          void* stream_) {
  EIOKU_REQUIRE_INIT();
  EIOKU_REQUIRE(sums_dev && counts_dev && centroids_dev && d > 0 && k > 0, "bad argument");
  hipLaunchKernelGGL(k_kmeans_finalize, dim3((unsigned)(((long long)k * d + 255) / 256)), dim3(256), 0, (hipStream_t)stream_,
                     sums_dev, counts_dev, k, d, centroids_dev);
  EIOKU_LAUNCH_CHECK();
  return EIOKU_OK;
}

// Nearest PQ sub-centroid per (vector, sub-quantiser) of x (or of x - coarse[list] when coarse != NULL).
// codes_out [n][m] uint8 and/or resid_out [n][d] may be NULL.  d = m * dsub, dsub in {4, 8, 16}.
int eioku_pq_assign(const float* x_dev, long long n, int d, int m, const float* coarse_dev, const long long* list_dev,
                    const float* pq_dev, uint8_t* codes_out_dev, float* resid_out_dev, void* stream_) {
  EIOKU_REQUIRE_INIT();
  EIOKU_REQUIRE(x_dev && pq_dev && n >= 0 && d > 0 && m > 0 && d % m == 0, "bad argument");
  EIOKU_REQUIRE(!coarse_dev || list_dev, "coarse centroids need the list assignment");
  const int dsub = d / m;
  EIOKU_REQUIRE(dsub == 4 || dsub == 8 || dsub == 16, "sub-vector size %d not supported (4, 8, 16)", dsub);
  if (n == 0) return EIOKU_OK;
  hipStream_t stream = (hipStream_t)stream_;
  dim3 grid(grid_cap(n, 256) / 4 + 1, (unsigned)m);
  if (dsub == 4) hipLaunchKernelGGL(k_pq_assign<4>, grid, dim3(256), 0, stream, x_dev, n, d, m, coarse_dev, list_dev, pq_dev, codes_out_dev, resid_out_dev);
  else if (dsub == 8) hipLaunchKernelGGL(k_pq_assign<8>, grid, dim3(256), 0, stream, x_dev, n, d, m, coarse_dev, list_dev, pq_dev, codes_out_dev, resid_out_dev);
  else hipLaunchKernelGGL(k_pq_assign<16>, grid, dim3(256), 0, stream, x_dev, n, d, m, coarse_dev, list_dev, pq_dev, codes_out_dev, resid_out_dev);
  EIOKU_LAUNCH_CHECK();
  return EIOKU_OK;
}

int eioku_ivf_histogram(const long long* list_dev, long long n, int nlist, int* counts_dev, void* stream_) {
  EIOKU_REQUIRE_INIT();
  EIOKU_REQUIRE(list_dev && counts_dev && nlist > 0, "bad argument");
  hipStream_t stream = (hipStream_t)stream_;
  EIOKU_HIP_CHECK(hipMemsetAsync(counts_dev, 0, (size_t)nlist * 4, stream));
  if (n > 0) {
    hipLaunchKernelGGL(k_ivf_histogram, dim3(grid_cap(n, 256)), dim3(256), 0, stream, list_dev, n, counts_dev);
    EIOKU_LAUNCH_CHECK();
  }
  return EIOKU_OK;
}

// cursor_dev: per-list write positions (initialised by the caller to the lists' current ends); on return
// they have advanced by the number of vectors scattered into each list.
int eioku_ivf_scatter(const long long* list_dev, long long n, int m, const uint8_t* codes_dev, long long id_base,
                      int* cursor_dev, uint8_t* list_codes_dev, long long* list_ids_dev, void* stream_) {
  EIOKU_REQUIRE_INIT();
  EIOKU_REQUIRE(list_dev && codes_dev && cursor_dev && list_codes_dev && list_ids_dev && m > 0, "bad argument");
  if (n == 0) return EIOKU_OK;
  hipLaunchKernelGGL(k_ivf_scatter, dim3(grid_cap(n, 256)), dim3(256), 0, (hipStream_t)stream_, list_dev, n, m, codes_dev,
                     id_base, cursor_dev, list_codes_dev, list_ids_dev);
  EIOKU_LAUNCH_CHECK();
  return EIOKU_OK;
}

// out[v][j][c] = alpha ||pq[j][c]||^2 + beta (vecs[v]_j . pq[j][c]), [nvec][m][256] floats.  The two tables of the
// precomputed-table scan: per list (vecs = coarse centroids, alpha 1, beta 2; once per index) and per query
// (vecs = queries, alpha 0, beta -2; once per search).
int eioku_ivfpq_tables(const float* vecs_dev, int nvec, int d, int m, const float* pq_dev, float alpha, float beta,
                       float* out_dev, void* stream_) {
  EIOKU_REQUIRE_INIT();
  EIOKU_REQUIRE(vecs_dev && pq_dev && out_dev && nvec >= 0 && m > 0 && d % m == 0, "bad argument");
  const int dsub = d / m;
  EIOKU_REQUIRE(dsub == 4 || dsub == 8 || dsub == 16, "sub-vector size %d not supported (4, 8, 16)", dsub);
  if (nvec == 0) return EIOKU_OK;
  hipStream_t stream = (hipStream_t)stream_;
  const size_t lds = (size_t)d * 4;
  if (dsub == 4) hipLaunchKernelGGL(k_ivfpq_tables<4>, dim3(nvec), dim3(256), lds, stream, vecs_dev, d, m, pq_dev, alpha, beta, out_dev);
  else if (dsub == 8) hipLaunchKernelGGL(k_ivfpq_tables<8>, dim3(nvec), dim3(256), lds, stream, vecs_dev, d, m, pq_dev, alpha, beta, out_dev);
  else hipLaunchKernelGGL(k_ivfpq_tables<16>, dim3(nvec), dim3(256), lds, stream, vecs_dev, d, m, pq_dev, alpha, beta, out_dev);
  EIOKU_LAUNCH_CHECK();
  return EIOKU_OK;
}

}  // extern "C"

namespace {
int scan_launch(const float* q_dev, int nq, int d, int m, const long long* probes_dev, int nprobe,
                const float* coarse_dev, const float* pq_dev, const int* offsets_dev, const int* sizes_dev,
                const uint8_t* list_codes_dev, const long long* list_ids_dev, int k, float* pd_dev,
                long long* pi_dev, const float* t2_dev, const float* t3_dev, void* stream_) {
  EIOKU_REQUIRE_INIT();
  EIOKU_REQUIRE(q_dev && probes_dev && coarse_dev && pq_dev && offsets_dev && sizes_dev && pd_dev && pi_dev, "NULL buffer");
  EIOKU_REQUIRE(nq >= 0 && nprobe > 0 && k >= 1 && k <= 32 && d % m == 0, "bad argument");
  EIOKU_REQUIRE((t2_dev == nullptr) == (t3_dev == nullptr), "the list table and the query table come together");
  const int dsub = d / m;
  EIOKU_REQUIRE(dsub == 4 || dsub == 8 || dsub == 16, "sub-vector size %d not supported (4, 8, 16)", dsub);
  if (nq == 0) return EIOKU_OK;
  hipStream_t stream = (hipStream_t)stream_;
  const int K = k <= 16 ? 16 : 32;
  size_t lds = (size_t)m * 256 * 4 + (size_t)d * 4;  // LUT + residual query
  const size_t merge = (size_t)256 * K * 12;
  if (merge > lds) lds = merge;
  EIOKU_REQUIRE(lds <= 150 * 1024, "m=%d needs %zu bytes of LDS", m, lds);
  dim3 grid((unsigned)nq, (unsigned)nprobe);
  const bool pre = t2_dev != nullptr;
#define EIOKU_SCAN1(K_, D_, P_)                                                                                \
  {                                                                                                            \
    static bool attr = false;                                                                                  \
    if (!attr) {                                                                                               \
      EIOKU_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_ivfpq_scan<K_, D_, P_>),             \
                                          hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));            \
      attr = true;                                                                                             \
    }                                                                                                          \
    hipLaunchKernelGGL((k_ivfpq_scan<K_, D_, P_>), grid, dim3(256), lds, stream, q_dev, nq, d, m, probes_dev, nprobe, \
                       coarse_dev, pq_dev, offsets_dev, sizes_dev, list_codes_dev, list_ids_dev, pd_dev, pi_dev, \
                       t2_dev, t3_dev);                                                                        \
  }
#define EIOKU_SCAN(K_, D_) \
  if (pre) EIOKU_SCAN1(K_, D_, true) else EIOKU_SCAN1(K_, D_, false)
  if (K == 16) {
    if (dsub == 4) EIOKU_SCAN(16, 4) else if (dsub == 8) EIOKU_SCAN(16, 8) else EIOKU_SCAN(16, 16)
  } else {
    if (dsub == 4) EIOKU_SCAN(32, 4) else if (dsub == 8) EIOKU_SCAN(32, 8) else EIOKU_SCAN(32, 16)
  }
#undef EIOKU_SCAN
#undef EIOKU_SCAN1
  EIOKU_LAUNCH_CHECK();
  return EIOKU_OK;
}
}  // namespace

extern "C" {

// ADC scan of the probed lists.  Partial results pd/pi are [nprobe][nq][K] with K = 16 (k <= 16) or 32;
// merge them with eioku_topk_merge(pd, pi, nprobe, nq, K -> k ...).
int eioku_ivfpq_scan(const float* q_dev, int nq, int d, int m, const long long* probes_dev, int nprobe,
                     const float* coarse_dev, const float* pq_dev, const int* offsets_dev, const int* sizes_dev,
                     const uint8_t* list_codes_dev, const long long* list_ids_dev, int k, float* pd_dev,
                     long long* pi_dev, void* stream_) {
  return scan_launch(q_dev, nq, d, m, probes_dev, nprobe, coarse_dev, pq_dev, offsets_dev, sizes_dev, list_codes_dev,
                     list_ids_dev, k, pd_dev, pi_dev, nullptr, nullptr, stream_);
}

// the same scan with the look-up tables assembled from eioku_ivfpq_tables' outputs: list_tables [nlist][m][256] (alpha 1,
// beta 2 over the coarse centroids), query_tables [nq][m][256] (alpha 0, beta -2 over the queries)
int eioku_ivfpq_scan_tables(const float* q_dev, int nq, int d, int m, const long long* probes_dev, int nprobe,
                            const float* coarse_dev, const float* pq_dev, const int* offsets_dev, const int* sizes_dev,
                            const uint8_t* list_codes_dev, const long long* list_ids_dev, const float* list_tables_dev,
                            const float* query_tables_dev, int k, float* pd_dev, long long* pi_dev, void* stream_) {
  EIOKU_REQUIRE(list_tables_dev && query_tables_dev, "NULL table");
  return scan_launch(q_dev, nq, d, m, probes_dev, nprobe, coarse_dev, pq_dev, offsets_dev, sizes_dev, list_codes_dev,
                     list_ids_dev, k, pd_dev, pi_dev, list_tables_dev, query_tables_dev, stream_);
}

}  // extern "C"
