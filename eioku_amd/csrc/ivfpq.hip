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
#include <cstdlib>

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

// || q - c_l ||^2 of one (query, list) pair by a 256-thread workgroup: rq <- q - c_l (LDS, d floats), then thread t sums
// rq[t]^2, rq[t + 256]^2, ... in that order, wave_reduce_add, (w0 + w1) + (w2 + w3).  The list-major path (k_term1)
// reproduces exactly this order with one wave per pair so that both scans return the same bits.
__device__ __forceinline__ float term1_block(const float* __restrict__ qrow, const float* __restrict__ crow, int d,
                                             float* rq, float* s_part) {
  const int tid = threadIdx.x;
  for (int t = tid; t < d; t += 256) rq[t] = qrow[t] - crow[t];
  __syncthreads();
  float part = 0.f;
  for (int t = tid; t < d; t += 256) part += rq[t] * rq[t];
  part = wave_reduce_add(part);
  if ((tid & 63) == 0) s_part[tid >> 6] = part;
  __syncthreads();
  return (s_part[0] + s_part[1]) + (s_part[2] + s_part[3]);
}

// grid (nq, probes scanned <= nprobe); block 256.  probes: [nq][nprobe] list ids (-1 = none).
// out: pd/pi [nprobe][nq][K]  (the [list][nq][k] layout k_topk_merge takes)
// gate (optional): device word + per-query candidate counts, see the first statement (fallback of the list-major path)
template <int K, int DSUB, bool PRE = false>
__global__ __launch_bounds__(256) void k_ivfpq_scan(const float* __restrict__ q, int nq, int d, int m,
                                                    const long long* __restrict__ probes, int nprobe,
                                                    const float* __restrict__ coarse, const float* __restrict__ pq,
                                                    const int* __restrict__ offsets, const int* __restrict__ sizes,
                                                    const uint8_t* __restrict__ list_codes,
                                                    const long long* __restrict__ list_ids, float* __restrict__ pd,
                                                    long long* __restrict__ pi, const float* __restrict__ t2 = nullptr,
                                                    const float* __restrict__ t3 = nullptr,
                                                    const int* __restrict__ gate = nullptr, int max_rows = 0,
                                                    const int* __restrict__ qcnt = nullptr, int qcap = 0) {
  extern __shared__ __attribute__((aligned(16))) float lut[];  // [m][256], then merge area
  // fallback of the list-major path: runs for every query when bit 0 of *gate is set (a workgroup list overflowed), else
  // only for the queries whose own candidate list overflowed
  if (gate && !(*gate & 1) && !(qcnt && qcnt[blockIdx.x] > qcap)) return;  // uniform
  const int qi = blockIdx.x, pr = blockIdx.y, tid = threadIdx.x;
  const long long l = probes[(size_t)qi * nprobe + pr];
  SmallTop<K> top;
  top.init();
  if (l >= 0) {
    // residual query r = q - coarse[l] once, in LDS (behind the LUT; the merge area reuses both later)
    float* rq = lut + (size_t)m * 256;
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    const int nent = m * 256;
    float term1 = 0.f;  // PRE: || q - c_l ||^2, added to every distance of the list
    if (PRE) {
      // LUT[e] = T2[l][e] + T3[q][e]: two coalesced rows, 16 bytes per lane and load
      __shared__ float s_part[4];
      const f32x4* a = reinterpret_cast<const f32x4*>(t2 + (size_t)l * nent);
      const f32x4* b = reinterpret_cast<const f32x4*>(t3 + (size_t)qi * nent);
      for (int e4 = tid; e4 < nent / 4; e4 += 256) reinterpret_cast<f32x4*>(lut)[e4] = a[e4] + b[e4];
      term1 = term1_block(q + (size_t)qi * d, coarse + (size_t)l * d, d, rq, s_part);
    } else {
    for (int t = tid; t < d; t += 256) rq[t] = q[(size_t)qi * d + t] - coarse[(size_t)l * d + t];
    __syncthreads();
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
    // max_rows > 0: only the list's first rows (the bound of the list-major path: the k-th best of a subset)
    const int off = offsets[l], sz = max_rows > 0 && sizes[l] > max_rows ? max_rows : sizes[l];
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
                long long* pi_dev, const float* t2_dev, const float* t3_dev, void* stream_, int nscan = 0,
                const int* gate = nullptr, int max_rows = 0, const int* qcnt = nullptr, int qcap = 0) {
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
  dim3 grid((unsigned)nq, (unsigned)(nscan > 0 ? nscan : nprobe));  // nscan: only the first probes of every query
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
                       t2_dev, t3_dev, gate, max_rows, qcnt, qcap);                                            \
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

// =====================================================================================================================
// K10s: LIST-MAJOR ADC scan (round 3).
//
// The query-major scan above reads a list once per query that probes it: 1024 queries x 32 probes over 10 M x 48-byte
// codes stream 15.6 GB of codes per search, and every (query, code) pair costs 48 LDS look-ups with random bank
// conflicts.  Here the roles are swapped, exactly as in the flat index's scan path (knn.hip, k_l2_scan):
//
//   * the (query -> probes) relation is inverted into (list -> queries) by a counting sort per search;
//   * a workgroup owns a SEGMENT of one list (NW x RT x 32 codes), DECODES it once into MFMA A operands held in
//     registers - a lane's operand for k-step s is the 8 bf16 values of sub-quantiser 2s + half of its row's code, ONE
//     16-byte gather from the 196 KB bf16 copy of the PQ codebook (L1/L2 resident) - and multiplies the segment with
//     every 32-query tile of the list; the query tiles are gathered from the bf16 copy of the queries by LDS-DMA, one
//     tile ahead.  Codes cross HBM once per search (0.48 GB at 10 M) and the products run on the matrix pipe;
//   * ADC distance of (q, v in list l)  =  ||q - c_l||^2  +  ( ||p_v||^2 + 2 c_l.p_v )  -  2 q.p_v   (p_v = decoded
//     residual).  The bracket is a per-code scalar stored with the index (hx = half of it), the first term a per-pair
//     scalar, so the test  dist <= tau_q  becomes  q.p_v >= hx_v + 0.5 (t1 - tau_q)  on the MFMA accumulator;
//   * the scan does not rank, it FILTERS with one bf16 product term and a rigorous margin
//       |q.p - bf(q).bf(p)| <= (2^-7 + 2^-16) |q| |p|     (round-to-nearest bf16: unit roundoff 2^-8 per operand)
//     plus an fp32 slack; tau_q = the k-th best EXACT distance over the query's nearest list (query-major kernel on
//     probe 0: the k-th best of any subset bounds the k-th best of all).  Survivors are re-ranked with the query-major
//     kernel's own fp32 arithmetic (same tables, same summation order): the (D, I) that come back are bit-identical to
//     the query-major scan's.  A candidate list that overflows raises a flag that gates the query-major scan in.
// =====================================================================================================================
namespace {

typedef unsigned u32x4k __attribute__((ext_vector_type(4)));
typedef unsigned u32x2k __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) void lds_void_t;
typedef __attribute__((address_space(1))) const void gbl_cvoid_t;

__device__ __forceinline__ unsigned bf16_rne_bits(float f) {  // finite inputs
  unsigned u = __float_as_uint(f);
  u += 0x7FFFu + ((u >> 16) & 1u);
  return u >> 16;
}

// pq [m][256][8] fp32 -> pqh [m * 256] 16-byte units of 8 bf16 (round to nearest)
__global__ __launch_bounds__(256) void k_pq_bf16(const float* __restrict__ pq, int nent, u32x4k* __restrict__ pqh) {
  const int e = blockIdx.x * 256 + threadIdx.x;
  if (e >= nent) return;
  unsigned h[8];
#pragma unroll
  for (int t = 0; t < 8; ++t) h[t] = bf16_rne_bits(pq[(size_t)e * 8 + t]);
  pqh[e] = u32x4k{h[0] | (h[1] << 16), h[2] | (h[3] << 16), h[4] | (h[5] << 16), h[6] | (h[7] << 16)};
}

// per stored code (grid = lists): hx = 0.5 * sum_j T2[l][j][c_j]  (fp32, j ascending) and the list's largest |p_v|^2
__global__ __launch_bounds__(256) void k_list_aux(const uint8_t* __restrict__ codes, const int* __restrict__ offsets,
                                                  const int* __restrict__ sizes, int m, const float* __restrict__ t2,
                                                  const float* __restrict__ pq, float* __restrict__ hx,
                                                  float* __restrict__ pmax2) {
  const int l = blockIdx.x, off = offsets[l], sz = sizes[l];
  const float* tl = t2 + (size_t)l * m * 256;
  float mx = 0.f;
  for (int i = threadIdx.x; i < sz; i += 256) {
    const uint8_t* code = codes + (size_t)(off + i) * m;
    float s = 0.f, pn = 0.f;
    for (int j = 0; j < m; ++j) {
      const int e = j * 256 + code[j];
      s += tl[e];
      const float4 a = *reinterpret_cast<const float4*>(pq + (size_t)e * 8);
      const float4 b = *reinterpret_cast<const float4*>(pq + (size_t)e * 8 + 4);
      pn += a.x * a.x + a.y * a.y + a.z * a.z + a.w * a.w + b.x * b.x + b.y * b.y + b.z * b.z + b.w * b.w;
    }
    hx[off + i] = 0.5f * s;
    mx = fmaxf(mx, pn);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
  __shared__ float s_m[4];
  if ((threadIdx.x & 63) == 0) s_m[threadIdx.x >> 6] = mx;
  __syncthreads();
  if (threadIdx.x == 0) pmax2[l] = fmaxf(fmaxf(s_m[0], s_m[1]), fmaxf(s_m[2], s_m[3]));
}

// queries -> bf16 rows + |q| (one wave per query)
__global__ __launch_bounds__(256) void k_q_prep(const float* __restrict__ q, int nq, int d, unsigned short* __restrict__ qbf,
                                                float* __restrict__ qn) {
  const int qi = (blockIdx.x * 256 + threadIdx.x) >> 6, lane = threadIdx.x & 63;
  if (qi >= nq) return;
  float s = 0.f;
  for (int t = lane; t < d; t += 64) {
    const float v = q[(size_t)qi * d + t];
    qbf[(size_t)qi * d + t] = (unsigned short)bf16_rne_bits(v);
    s += v * v;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
  if (lane == 0) qn[qi] = sqrtf(s);
}

// t1[p] = || q - c_l ||^2 of pair p = q * nprobe + pr with term1_block's bits: a wave plays the four waves of that
// workgroup (virtual thread v = lane + 64 w sums t = v, v + 256, ...; per-wave wave_reduce_add; (w0 + w1) + (w2 + w3))
__global__ __launch_bounds__(256) void k_term1(const float* __restrict__ q, const float* __restrict__ coarse,
                                               const long long* __restrict__ probes, int npairs, int nprobe, int d,
                                               float* __restrict__ t1) {
  const int p = (blockIdx.x * 256 + threadIdx.x) >> 6, lane = threadIdx.x & 63;
  if (p >= npairs) return;
  const long long l = probes[p];
  if (l < 0) {
    if (lane == 0) t1[p] = 0.f;
    return;
  }
  const float* qrow = q + (size_t)(p / nprobe) * d;
  const float* crow = coarse + (size_t)l * d;
  float w[4];
#pragma unroll
  for (int wv = 0; wv < 4; ++wv) {
    float part = 0.f;
    for (int t = lane + 64 * wv; t < d; t += 256) {
      const float r = qrow[t] - crow[t];
      part += r * r;
    }
    w[wv] = wave_reduce_add(part);
  }
  if (lane == 0) t1[p] = (w[0] + w[1]) + (w[2] + w[3]);
}

// the list that bounds query q: its first probe that holds at least k codes (the k-th best of ANY k codes bounds the k-th
// best of all; the nearest list itself may be empty or tiny on unbalanced indexes).  -1: no such list, no bound.
__global__ __launch_bounds__(256) void k_tau_probe(const long long* __restrict__ probes, int nq, int nprobe,
                                                   const int* __restrict__ sizes, int k, long long* __restrict__ tprobe) {
  const int q = blockIdx.x * 256 + threadIdx.x;
  if (q >= nq) return;
  long long pick = -1;
  for (int pr = 0; pr < nprobe; ++pr) {
    const long long l = probes[(size_t)q * nprobe + pr];
    if (l >= 0 && sizes[l] >= k) {
      pick = l;
      break;
    }
  }
  tprobe[q] = pick;
}

// ---- inversion: (query -> probes) to (list -> queries) ----------------------------------------------------------
__global__ __launch_bounds__(256) void k_inv_count(const long long* __restrict__ probes, int npairs,
                                                   const int* __restrict__ sizes, int* __restrict__ lcnt) {
  const int p = blockIdx.x * 256 + threadIdx.x;
  if (p >= npairs) return;
  const long long l = probes[p];
  if (l >= 0 && sizes[l] > 0) atomicAdd(&lcnt[l], 1);
}

// one workgroup: loff (slots, every list padded to whole 32-query tiles), woff (work items: segments of the lists that
// have queries), totals; cursor zeroed
__global__ __launch_bounds__(1024) void k_inv_scan(const int* __restrict__ lcnt, const int* __restrict__ sizes, int nlist,
                                                   int seg, int* __restrict__ loff, int* __restrict__ woff,
                                                   int* __restrict__ cursor, int* __restrict__ nwork) {
  __shared__ int s_a[1024], s_b[1024];
  const int tid = threadIdx.x;
  const int per = (nlist + 1023) / 1024;
  const int lo = tid * per, hi = min(nlist, lo + per);
  int sa = 0, sb = 0;
  for (int l = lo; l < hi; ++l) {
    const int c = lcnt[l];
    sa += (c + 31) & ~31;
    sb += c > 0 ? (sizes[l] + seg - 1) / seg : 0;
  }
  s_a[tid] = sa;
  s_b[tid] = sb;
  __syncthreads();
  for (int o = 1; o < 1024; o <<= 1) {
    const int va = tid >= o ? s_a[tid - o] : 0, vb = tid >= o ? s_b[tid - o] : 0;
    __syncthreads();
    s_a[tid] += va;
    s_b[tid] += vb;
    __syncthreads();
  }
  int ra = s_a[tid] - sa, rb = s_b[tid] - sb;  // exclusive
  for (int l = lo; l < hi; ++l) {
    const int c = lcnt[l];
    loff[l] = ra;
    woff[l] = rb;
    cursor[l] = 0;
    ra += (c + 31) & ~31;
    rb += c > 0 ? (sizes[l] + seg - 1) / seg : 0;
  }
  if (tid == 1023) {
    loff[nlist] = s_a[1023];
    woff[nlist] = s_b[1023];
    *nwork = s_b[1023];
  }
}

// work item w -> {first code position, codes in the segment, first slot of the list, query tiles of the list}: the scan
// reads one 16-byte descriptor per item (prefetched an item ahead) instead of searching the prefix sums itself
__global__ __launch_bounds__(256) void k_inv_items(const int* __restrict__ woff, const int* __restrict__ loff,
                                                   const int* __restrict__ offsets, const int* __restrict__ sizes, int nlist,
                                                   int seg, int max_items, int4* __restrict__ items) {
  const int w = blockIdx.x * 256 + threadIdx.x;
  if (w >= woff[nlist] || w >= max_items) return;
  int lo = 0, hi = nlist;  // woff[lo] <= w < woff[hi]
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if (woff[mid] <= w) lo = mid; else hi = mid;
  }
  const int r0 = (w - woff[lo]) * seg;
  items[w] = make_int4(offsets[lo] + r0, min(seg, sizes[lo] - r0), loff[lo], (loff[lo + 1] - loff[lo]) >> 5);
}

__global__ __launch_bounds__(256) void k_lq_fill(int* __restrict__ lq_q, float* __restrict__ lq_thr, int n) {
  for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
    lq_q[i] = 0;
    lq_thr[i] = __builtin_inff();
  }
}

// pair p -> a slot of its list; the slot's filter threshold: candidate  <=>  bf16 product >= hx_v + thr
//   thr = 0.5 (t1 - tau) - [ 2^-7 (1 + 2^-9 + fp32 accumulation) |q| pmax_l  +  fp32 slack of the three scalar terms ]
__global__ __launch_bounds__(256) void k_inv_scatter(const long long* __restrict__ probes, int npairs, int nprobe,
                                                     const int* __restrict__ sizes, const int* __restrict__ loff,
                                                     int* __restrict__ cursor, const float* __restrict__ t1,
                                                     const float* __restrict__ taud, int K, int k,
                                                     const float* __restrict__ qn, const float* __restrict__ pmax2,
                                                     int* __restrict__ lq_q, int* __restrict__ lq_p,
                                                     float* __restrict__ lq_thr) {
  const int p = blockIdx.x * 256 + threadIdx.x;
  if (p >= npairs) return;
  const long long l = probes[p];
  if (l < 0 || sizes[l] <= 0) return;
  const int q = p / nprobe;
  const int slot = loff[l] + atomicAdd(&cursor[l], 1);
  const float tau = taud[(size_t)q * K + (k - 1)];  // FLT_MAX when the nearest list holds fewer than k codes
  const float tt = t1[p], nq_ = qn[q], pm = sqrtf(pmax2[l]) * 1.000001f;
  const float delta = 0.0078125f * 1.01f * nq_ * pm;
  const float mag = tt + pm * pm + 2.f * (nq_ + sqrtf(tt)) * pm + 2.f * nq_ * pm;  // >= |t1| + |t2_v| + 2 |q.p_v|
  lq_q[slot] = q;
  lq_p[slot] = p;
  lq_thr[slot] = tau >= 3.0e38f ? -__builtin_inff() : 0.5f * (tt - tau) - (delta + 2e-5f * mag);
}

struct LScanArgs {
  const u32x4k* pqh;          // [m * 256]
  const unsigned short* qbf;  // [nq][D]
  const uint8_t* codes;       // list_codes [ntotal][m]
  const float* hx;            // [ntotal]
  const int4* items;          // [nwork] {first code position, codes, first slot, query tiles}
  const int* nwork;
  const int* lq_q;
  const float* lq_thr;
  unsigned* wl;               // [gridDim.x][wl_cap][2]  (slot, code position)
  int* wl_cnt;                // [gridDim.x]
  int wl_cap;
  int ablate;                 // measurement builds only (EIOKU_LSCAN_ABLATE): 1 = no table gathers, 2 = no products
};

// NS = d / 16 k-steps (dsub = 8: m = 2 NS); NW waves x RT 32-code tiles per workgroup and work item
template <int NS, int NW, int RT>
__global__ __launch_bounds__(NW * 64) void k_lscan(LScanArgs a) {
  constexpr int PU = NS * 64, M = 2 * NS, CW = M / 4, D = NS * 16;
  constexpr int NDMA = (NS + NW - 1) / NW;
  extern __shared__ __attribute__((aligned(16))) unsigned char dyn_smem[];
  u32x4k* qbuf = reinterpret_cast<u32x4k*>(dyn_smem);                       // [2][PU]
  float* s_hx = reinterpret_cast<float*>(dyn_smem + (size_t)2 * PU * 16);   // [NW * RT * 32]
  int* s_cnt = reinterpret_cast<int*>(s_hx + NW * RT * 32);

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int col = lane & 31, half = lane >> 5;
  if (tid == 0) *s_cnt = 0;
  unsigned* const my_list = a.wl + (size_t)blockIdx.x * a.wl_cap * 2;
  const int nwork = *a.nwork;

  // one 32-query tile -> LDS in fragment order: DMA piece s is the 64 lanes' 16-byte B operands of k-step s, lane
  // (col, half) fetching dims 16 s + 8 half + [0, 8) of ITS query's bf16 row.  Every wave issues NDMA pieces.
  auto stage_q = [&](int qid, int buf) {
    const unsigned char* row = reinterpret_cast<const unsigned char*>(a.qbf) + (size_t)qid * (D * 2) + half * 16;
#pragma unroll
    for (int j = 0; j < NDMA; ++j) {
      int s = j * NW + wave;
      if (s >= NS) s = NS - 1;  // same bytes to the same place
      __builtin_amdgcn_global_load_lds((gbl_cvoid_t*)(row + s * 32), (lds_void_t*)(qbuf + (size_t)buf * PU + s * 64), 16, 0, 0);
    }
  };

  int4 it_n = blockIdx.x < nwork ? a.items[blockIdx.x] : make_int4(0, 0, 0, 0);
  for (int w = blockIdx.x; w < nwork; w += gridDim.x) {
    const int4 it = it_n;
    if (w + (int)gridDim.x < nwork) it_n = a.items[w + gridDim.x];
    const int pos0 = it.x, nrows = it.y, slot0 = it.z, nqt = it.w;
    __syncthreads();  // every wave is done with the previous item's query buffers

    int qid = a.lq_q[slot0 + col];
    float thr = a.lq_thr[slot0 + col];
    int qid1 = 0;      // tile 1's queries (tile t + 1 in the loop below: loaded a whole tile before they are needed)
    float thr1 = 0.f;
    if (nqt > 1) {
      qid1 = a.lq_q[slot0 + 32 + col];
      thr1 = a.lq_thr[slot0 + 32 + col];
    }

    // this wave's rows: code words, half table terms
    u32x4k xh[RT][NS];
    float hxmin[RT];
    bool act[RT];
#pragma unroll
    for (int i = 0; i < RT; ++i) {
      const int r0 = (wave * RT + i) * 32;
      act[i] = r0 < nrows;
      const int row = r0 + col;
      const int rowc = row < nrows ? row : nrows - 1;
      const unsigned* cp = reinterpret_cast<const unsigned*>(a.codes + (size_t)(pos0 + rowc) * M);
      unsigned cw[CW];
      if constexpr (CW % 4 == 0) {
#pragma unroll
        for (int c = 0; c < CW / 4; ++c) {
          const u32x4k v = *reinterpret_cast<const u32x4k*>(cp + 4 * c);
          cw[4 * c] = v[0]; cw[4 * c + 1] = v[1]; cw[4 * c + 2] = v[2]; cw[4 * c + 3] = v[3];
        }
      } else {
#pragma unroll
        for (int c = 0; c < CW / 2; ++c) {
          const u32x2k v = *reinterpret_cast<const u32x2k*>(cp + 2 * c);
          cw[2 * c] = v[0]; cw[2 * c + 1] = v[1];
        }
      }
      float h = row < nrows ? a.hx[pos0 + row] : __builtin_inff();
      if (half == 0) s_hx[(wave * RT + i) * 32 + col] = h;  // read back by this wave only
#pragma unroll
      for (int o = 16; o > 0; o >>= 1) h = fminf(h, __shfl_xor(h, o, 64));
      hxmin[i] = h;
#pragma unroll
      for (int s = 0; s < NS; ++s) {
        const unsigned byte = (a.ablate & 1) ? (unsigned)col : (cw[s >> 1] >> (16 * (s & 1) + 8 * half)) & 0xFFu;
        xh[i][s] = a.pqh[(2 * s + half) * 256 + byte];
      }
    }
    stage_q(qid, 0);

    for (int t = 0; t < nqt; ++t) {
      __builtin_amdgcn_s_waitcnt(0x0f70);  // vmcnt(0): this wave's pieces of tile t (and the operands at t = 0)
      __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0)
      asm volatile("" ::: "memory");
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
      if (t + 1 < nqt) stage_q(qid1, (t + 1) & 1);
      int qid2 = 0;      // tile t + 2's queries: in flight during this tile's products, waited for at the next meeting
      float thr2 = 0.f;
      if (t + 2 < nqt) {
        qid2 = a.lq_q[slot0 + 32 * (t + 2) + col];
        thr2 = a.lq_thr[slot0 + 32 * (t + 2) + col];
      }
      __builtin_amdgcn_sched_barrier(0);
      const u32x4k* qb = qbuf + (size_t)(t & 1) * PU;
      f32x16 acc[RT];
#pragma unroll
      for (int i = 0; i < RT; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
      constexpr int PF = 2;
      u32x4k bh[PF];
#pragma unroll
      for (int s = 0; s < PF; ++s) bh[s] = qb[s * 64 + lane];
      if (!(a.ablate & 2)) {
#pragma unroll
      for (int s = 0; s < NS; ++s) {
        const bf16x8 ch = __builtin_bit_cast(bf16x8, bh[s % PF]);
#pragma unroll
        for (int i = 0; i < RT; ++i)
          acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, xh[i][s]), ch, acc[i], 0, 0, 0);
        if (s + PF < NS) bh[s % PF] = qb[(s + PF) * 64 + lane];
      }
      } else {
#pragma unroll
        for (int i = 0; i < RT; ++i) acc[i][0] = __uint_as_float(xh[i][0][0] ^ xh[i][NS - 1][3] ^ bh[0][0]);
      }
      // filter (as k_l2_scan): tile maximum against the tile's smallest hx first, per row only where that passes
#pragma unroll
      for (int i = 0; i < RT; ++i) {
        const f32x16& c = acc[i];
        // the tile test on the products' bit patterns as signed integers (see k_l2_scan: 8 v_max3_i32, no canonicalisation);
        // T <= 0 or NaN (no bound for the query) takes the exact per-row test
        auto ib = [&](int r) { return __float_as_int(c[r]); };
        const int i01 = max(max(ib(0), ib(1)), ib(2)), i23 = max(max(ib(3), ib(4)), ib(5));
        const int i45 = max(max(ib(6), ib(7)), ib(8)), i67 = max(max(ib(9), ib(10)), ib(11));
        const int i89 = max(max(ib(12), ib(13)), ib(14));
        const int imx = max(max(max(i01, i23), i45), max(max(i67, i89), ib(15)));
        const float T = hxmin[i] + thr;
        if ((!(T > 0.f) || imx >= __float_as_int(T)) && act[i]) {
          const int r0 = pos0 + (wave * RT + i) * 32;
          int h4 = 4 * half;
          asm volatile("" : "+v"(h4));
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            const float4 hx4 = *reinterpret_cast<const float4*>(s_hx + (wave * RT + i) * 32 + 8 * g + h4);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              const float hxr = j == 0 ? hx4.x : (j == 1 ? hx4.y : (j == 2 ? hx4.z : hx4.w));
              if (c[4 * g + j] >= hxr + thr) {
                const int pos = atomicAdd(s_cnt, 1);  // LDS
                if (pos < a.wl_cap) {
                  my_list[2 * (size_t)pos] = (unsigned)(slot0 + 32 * t + col);
                  my_list[2 * (size_t)pos + 1] = (unsigned)(r0 + 8 * g + j + h4);
                }
              }
            }
          }
        }
      }
      qid = qid1;
      thr = thr1;
      qid1 = qid2;
      thr1 = thr2;
    }
  }
  __syncthreads();
  if (tid == 0) a.wl_cnt[blockIdx.x] = *s_cnt;
}

// workgroup lists -> per-query candidate lists
__global__ __launch_bounds__(256) void k_lbin(const unsigned* __restrict__ wl, const int* __restrict__ wl_cnt, int wl_cap,
                                              const int* __restrict__ lq_q, int* __restrict__ cand, int* __restrict__ cnt,
                                              int cap, int* __restrict__ overflow) {
  int c = wl_cnt[blockIdx.x];
  if (c > wl_cap) {
    if (threadIdx.x == 0) atomicOr(overflow, 1);
    c = wl_cap;
  }
  const unsigned* list = wl + (size_t)blockIdx.x * wl_cap * 2;
  for (int i = threadIdx.x; i < c; i += 256) {
    const unsigned slot = list[2 * i], row = list[2 * i + 1];
    const int qi = lq_q[slot];
    const int pos = atomicAdd(cnt + qi, 1);
    if (pos < cap) {
      cand[((size_t)qi * cap + pos) * 2] = (int)slot;
      cand[((size_t)qi * cap + pos) * 2 + 1] = (int)row;
    }
  }
}

// one workgroup per query: the candidates' ADC distances with the query-major kernel's arithmetic
// (sum_j (T2[l][j][c] + T3[q][j][c]) in j order, + t1), then the k best by (distance, id).  Candidates are scored in
// chunks of kRChunk; the best k so far ride along as extra entries of the next chunk, so the list a query may hold is
// bounded by the global buffer (cap), not by LDS.
constexpr int kRChunk = 2048;
__global__ __launch_bounds__(256) void k_lrerank(const int* __restrict__ cand, const int* __restrict__ cnt, int cap, int m,
                                                 const int* __restrict__ lq_p, const long long* __restrict__ probes,
                                                 const float* __restrict__ t1, const float* __restrict__ t2,
                                                 const float* __restrict__ t3, const uint8_t* __restrict__ codes,
                                                 const long long* __restrict__ ids, int k, float* __restrict__ Dout,
                                                 long long* __restrict__ Iout, int* __restrict__ overflow) {
  extern __shared__ __attribute__((aligned(16))) unsigned char dyn_smem[];
  const int nent = m * 256;
  float* s_t3 = reinterpret_cast<float*>(dyn_smem);            // [m * 256]
  long long* s_id = reinterpret_cast<long long*>(s_t3 + nent);  // [kRChunk + 32]
  float* s_d = reinterpret_cast<float*>(s_id + kRChunk + 32);   // [kRChunk + 32]
  __shared__ float s_bv[4], s_kv[32];
  __shared__ long long s_bi[4], s_ki[32];
  const int qi = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  int c = cnt[qi];
  if (tid == 0) {
    atomicMax(overflow + 2, c);
    atomicAdd(overflow + 3, c);
    atomicMax(reinterpret_cast<unsigned long long*>(overflow + 4), ((unsigned long long)c << 32) | (unsigned)qi);
  }
  if (c > cap) {  // this query is redone by the gated query-major launches that follow
    if (tid == 0) atomicOr(overflow, 2);
    c = cap;
  }
  typedef float f32x4 __attribute__((ext_vector_type(4)));
  for (int e4 = tid; e4 < nent / 4; e4 += 256)
    reinterpret_cast<f32x4*>(s_t3)[e4] = reinterpret_cast<const f32x4*>(t3 + (size_t)qi * nent)[e4];
  int kept = 0;  // entries of s_kv / s_ki: the best of the chunks so far
  __syncthreads();
  for (int c0 = 0; c0 == 0 || c0 < c; c0 += kRChunk) {
    const int cc = min(kRChunk, c - c0);
    for (int j = tid; j < cc; j += 256) {
      const int slot = cand[((size_t)qi * cap + c0 + j) * 2], pos = cand[((size_t)qi * cap + c0 + j) * 2 + 1];
      const int p = lq_p[slot];
      const long long l = probes[p];
      const float* tl = t2 + (size_t)l * nent;
      const uint8_t* code = codes + (size_t)pos * m;
      // 8 table entries in flight per thread (one load -> add -> next load chain per entry was 48 dependent L2 round
      // trips per candidate: a query holding thousands of candidates took longer than the whole scan); the sum itself
      // stays sequential in j, as in the query-major kernel
      float s = 0.f;
      for (int j0 = 0; j0 < m; j0 += 8) {  // m is a multiple of 8 (d / m = 8, d a multiple of 64)
        const u32x2k cw = *reinterpret_cast<const u32x2k*>(code + j0);
        float a[8], b[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const int e = (j0 + u) * 256 + (int)((cw[u >> 2] >> (8 * (u & 3))) & 0xFFu);
          a[u] = tl[e];
          b[u] = s_t3[e];
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) s += a[u] + b[u];
      }
      s_d[j] = s + t1[p];
      s_id[j] = ids[pos];
    }
    if (tid < kept) {
      s_d[cc + tid] = s_kv[tid];
      s_id[cc + tid] = s_ki[tid];
    }
    __syncthreads();
    const int tot = cc + kept;
    const bool last = c0 + kRChunk >= c;
    float lv = -__builtin_inff();
    long long li = -1;
    int found = 0;
    for (int r = 0; r < k; ++r) {
      float bv = FLT_MAX;
      long long bi = 0x7FFFFFFFFFFFFFFFll;
      for (int j = tid; j < tot; j += 256) {
        const float v = s_d[j];
        const long long i = s_id[j];
        const bool after = v > lv || (v == lv && i > li);
        if (after && (v < bv || (v == bv && i < bi))) {
          bv = v;
          bi = i;
        }
      }
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) {
        const float ov = __shfl_xor(bv, o, 64);
        const long long oi = __shfl_xor(bi, o, 64);
        if (ov < bv || (ov == bv && oi < bi)) {
          bv = ov;
          bi = oi;
        }
      }
      if (lane == 0) {
        s_bv[wave] = bv;
        s_bi[wave] = bi;
      }
      __syncthreads();
      bv = s_bv[0];
      bi = s_bi[0];
#pragma unroll
      for (int w = 1; w < 4; ++w)
        if (s_bv[w] < bv || (s_bv[w] == bv && s_bi[w] < bi)) {
          bv = s_bv[w];
          bi = s_bi[w];
        }
      __syncthreads();
      const bool none = bi == 0x7FFFFFFFFFFFFFFFll;
      if (tid == 0) {
        if (last) {
          Dout[(size_t)qi * k + r] = none ? FLT_MAX : bv;
          Iout[(size_t)qi * k + r] = none ? -1 : bi;
        } else if (!none) {
          s_kv[r] = bv;
          s_ki[r] = bi;
        }
      }
      if (!none) found = r + 1;
      lv = none ? FLT_MAX : bv;
      li = none ? 0x7FFFFFFFFFFFFFFFll : bi;
    }
    kept = found;
    __syncthreads();
  }
}

// fallback merge (gated): the query-major scan's [nprobe][nq][K] partial lists -> (D, I); one wave per query
__global__ __launch_bounds__(64) void k_probe_merge(const float* __restrict__ pd, const long long* __restrict__ pi, int nprobe,
                                                    int nq, int K, int k, float* __restrict__ Dout,
                                                    long long* __restrict__ Iout, const int* __restrict__ gate,
                                                    const int* __restrict__ qcnt, int qcap) {
  if (gate && !(*gate & 1) && !(qcnt && qcnt[blockIdx.x] > qcap)) return;
  const int qi = blockIdx.x, lane = threadIdx.x;
  const int n = nprobe * K;
  float lv = -__builtin_inff();
  long long li = -1;
  for (int r = 0; r < k; ++r) {
    float bv = FLT_MAX;
    long long bi = 0x7FFFFFFFFFFFFFFFll;
    for (int e = lane; e < n; e += 64) {
      const int pr = e / K, x = e - pr * K;
      const float v = pd[((size_t)pr * nq + qi) * K + x];
      const long long i = pi[((size_t)pr * nq + qi) * K + x];
      if (i < 0) continue;
      const bool after = v > lv || (v == lv && i > li);
      if (after && (v < bv || (v == bv && i < bi))) {
        bv = v;
        bi = i;
      }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const float ov = __shfl_xor(bv, o, 64);
      const long long oi = __shfl_xor(bi, o, 64);
      if (ov < bv || (ov == bv && oi < bi)) {
        bv = ov;
        bi = oi;
      }
    }
    const bool none = bi == 0x7FFFFFFFFFFFFFFFll;
    if (lane == 0) {
      Dout[(size_t)qi * k + r] = none ? FLT_MAX : bv;
      Iout[(size_t)qi * k + r] = none ? -1 : bi;
    }
    lv = none ? FLT_MAX : bv;
    li = none ? 0x7FFFFFFFFFFFFFFFll : bi;
  }
}

constexpr int kLRT = 2;  // 32-code tiles per wave; a work item = NW x RT x 32 codes
inline int lscan_nw() {  // waves per workgroup: 8 (one workgroup per CU) or 4 (two: one's operand gathers beside the other's products)
  static const int nw = getenv("EIOKU_LSCAN_NW") ? atoi(getenv("EIOKU_LSCAN_NW")) : 8;
  return nw == 4 ? 4 : 8;
}

inline size_t al256(size_t v) { return (v + 255) & ~(size_t)255; }

struct LWork {  // carve-up of the caller's workspace
  size_t qbf, qn, t1, tprobe, pd, pi, zero0, lcnt, cnt, overflow, nwork, zero1, cursor, loff, woff, lq_q, lq_p, lq_thr, wl, wl_cnt, cand,
      items, total;
  int nslots, grid, wl_cap, cap, K, max_items;
};

LWork lwork(int nq, int d, int nprobe, int nlist, int k, int cap, long long ntotal) {
  LWork w;
  const size_t npairs = (size_t)nq * nprobe;
  w.K = k <= 16 ? 16 : 32;
  w.cap = cap;
  w.nslots = (int)(npairs + 31 * (npairs < (size_t)nlist ? npairs : (size_t)nlist));
  w.grid = num_cus() * (8 / lscan_nw());
  const int seg = lscan_nw() * kLRT * 32;
  w.max_items = (int)(ntotal / seg + nlist);  // sum over lists of ceil(size / seg)
  long long wl_cap = ((64ll << 20) / 8) / w.grid;
  if (cap < 64) wl_cap = cap;  // tests shrink both kinds of list to force the overflow path
  w.wl_cap = (int)wl_cap;
  size_t o = 0;
  auto take = [&](size_t bytes) { const size_t at = o; o += al256(bytes); return at; };
  w.qbf = take((size_t)nq * d * 2);
  w.qn = take((size_t)nq * 4);
  w.t1 = take(npairs * 4);
  w.tprobe = take((size_t)nq * 8);
  w.pd = take(npairs * w.K * 4);
  w.pi = take(npairs * w.K * 8);
  w.zero0 = o;
  w.lcnt = take((size_t)nlist * 4);
  w.cnt = take((size_t)nq * 4);
  w.overflow = take(32);  // stats: overflow bits, work items, largest candidate list, candidates in all, (count << 32 | query) max
  w.nwork = w.overflow + 4;
  w.zero1 = o;
  w.cursor = take((size_t)nlist * 4);
  w.loff = take((size_t)(nlist + 1) * 4);
  w.woff = take((size_t)(nlist + 1) * 4);
  w.lq_q = take((size_t)w.nslots * 4);
  w.lq_p = take((size_t)w.nslots * 4);
  w.lq_thr = take((size_t)w.nslots * 4);
  w.wl = take((size_t)w.grid * w.wl_cap * 8);
  w.wl_cnt = take((size_t)w.grid * 4);
  w.cand = take((size_t)nq * cap * 8);
  w.items = take((size_t)w.max_items * 16);
  w.total = o;
  return w;
}

template <int NS, int NW>
int launch_lscan1(const LScanArgs& a, int grid, hipStream_t stream) {
  const size_t lds = (size_t)2 * NS * 64 * 16 + (size_t)NW * kLRT * 32 * 4 + 16;
  static bool attr = false;
  if (!attr) {
    EIOKU_HIP_CHECK(hipFuncSetAttribute((const void*)k_lscan<NS, NW, kLRT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    attr = true;
  }
  hipLaunchKernelGGL((k_lscan<NS, NW, kLRT>), dim3((unsigned)grid), dim3(NW * 64), lds, stream, a);
  EIOKU_LAUNCH_CHECK();
  return EIOKU_OK;
}
template <int NS>
int launch_lscan(const LScanArgs& a, int grid, hipStream_t stream) {
  return lscan_nw() == 4 ? launch_lscan1<NS, 4>(a, grid, stream) : launch_lscan1<NS, 8>(a, grid, stream);
}

bool lists_geometry_ok(int d, int m) { return m * 8 == d && (d == 64 || d == 128 || d == 256 || d == 384); }

}  // namespace

extern "C" {

// Index-side tables of the list-major scan (once per pack / codebook change): pqh_out [m * 256] 16-byte units (the bf16
// copy of the codebook), hx_out [ntotal] (half of each stored code's list-dependent scalar), pmax2_out [nlist].
int eioku_ivfpq_lists_aux(const uint8_t* list_codes_dev, const int* offsets_dev, const int* sizes_dev, int nlist, int d, int m,
                          const float* list_tables_dev, const float* pq_dev, void* pqh_out_dev, float* hx_out_dev,
                          float* pmax2_out_dev, void* stream_) {
  EIOKU_REQUIRE_INIT();
  EIOKU_REQUIRE(list_codes_dev && offsets_dev && sizes_dev && list_tables_dev && pq_dev && pqh_out_dev && hx_out_dev &&
                    pmax2_out_dev && nlist > 0,
                "bad argument");
  EIOKU_REQUIRE(lists_geometry_ok(d, m), "list-major scan: d/m must be 8 and d in {64, 128, 256, 384} (got d=%d m=%d)", d, m);
  hipStream_t stream = (hipStream_t)stream_;
  hipLaunchKernelGGL(k_pq_bf16, dim3((unsigned)(m * 256 + 255) / 256), dim3(256), 0, stream, pq_dev, m * 256, (u32x4k*)pqh_out_dev);
  EIOKU_LAUNCH_CHECK();
  hipLaunchKernelGGL(k_list_aux, dim3((unsigned)nlist), dim3(256), 0, stream, list_codes_dev, offsets_dev, sizes_dev, m,
                     list_tables_dev, pq_dev, hx_out_dev, pmax2_out_dev);
  EIOKU_LAUNCH_CHECK();
  return EIOKU_OK;
}

// bytes of workspace eioku_ivfpq_search_lists needs (cand_cap: per-query candidate capacity, 0 = default 8192)
long long eioku_ivfpq_lists_workspace(int nq, int d, int nprobe, int nlist, long long ntotal, int k, int cand_cap) {
  if (!initialised() || nq < 0 || nprobe <= 0 || nlist <= 0 || ntotal < 0 || k < 1 || k > 32) return -1;
  return (long long)lwork(nq, d, nprobe, nlist, k, cand_cap > 0 ? cand_cap : 8192, ntotal).total;
}

// The whole search behind one call (all pointers DEVICE, asynchronous on `stream`): probes [nq][nprobe] from the coarse
// quantiser, list_tables / query_tables as for eioku_ivfpq_scan_tables, pqh / hx / pmax2 from eioku_ivfpq_lists_aux.
// D [nq][k], I [nq][k]: bit-identical to eioku_ivfpq_scan_tables + eioku_topk_merge_ex.  stats_out (optional, device,
// 6 ints): overflow bits (1: a workgroup list overflowed, every query redone query-major; 2: some queries' own lists
// overflowed, those redone), work items, largest per-query candidate list, candidates of all queries, the query that
// holds the largest list, its size.
int eioku_ivfpq_search_lists(const float* q_dev, int nq, int d, int m, const long long* probes_dev, int nprobe, int nlist,
                             long long ntotal, const float* coarse_dev, const float* pq_dev, const int* offsets_dev, const int* sizes_dev,
                             const uint8_t* list_codes_dev, const long long* list_ids_dev, const float* list_tables_dev,
                             const float* query_tables_dev, const void* pqh_dev, const float* hx_dev,
                             const float* pmax2_dev, int k, int cand_cap, void* workspace_dev, long long workspace_bytes,
                             float* D_dev, long long* I_dev, int* stats_out_dev, void* stream_) {
  EIOKU_REQUIRE_INIT();
  EIOKU_REQUIRE(q_dev && probes_dev && coarse_dev && pq_dev && offsets_dev && sizes_dev && list_codes_dev && list_ids_dev &&
                    list_tables_dev && query_tables_dev && pqh_dev && hx_dev && pmax2_dev && workspace_dev && D_dev && I_dev,
                "NULL buffer");
  EIOKU_REQUIRE(nq >= 0 && nprobe > 0 && nlist > 0 && k >= 1 && k <= 32, "bad argument");
  EIOKU_REQUIRE(lists_geometry_ok(d, m), "list-major scan: d/m must be 8 and d in {64, 128, 256, 384} (got d=%d m=%d)", d, m);
  EIOKU_REQUIRE((long long)nq * nprobe < (1ll << 30), "nq x nprobe too large for one call");
  if (nq == 0) return EIOKU_OK;
  const int cap = cand_cap > 0 ? cand_cap : 8192;
  EIOKU_REQUIRE(ntotal >= 0 && ntotal < (1ll << 31), "ntotal out of range");
  const LWork w = lwork(nq, d, nprobe, nlist, k, cap, ntotal);
  EIOKU_REQUIRE(workspace_bytes >= (long long)w.total, "workspace: %lld bytes given, %zu needed", workspace_bytes, w.total);
  const size_t rr_lds = (size_t)m * 256 * 4 + (size_t)(kRChunk + 32) * 12;
  EIOKU_REQUIRE(cap <= (1 << 20), "cand_cap %d too large", cap);
  hipStream_t stream = (hipStream_t)stream_;
  unsigned char* ws = (unsigned char*)workspace_dev;
  const int npairs = nq * nprobe;
  unsigned short* qbf = (unsigned short*)(ws + w.qbf);
  float* qn = (float*)(ws + w.qn);
  float* t1 = (float*)(ws + w.t1);
  float* pd = (float*)(ws + w.pd);
  long long* pi = (long long*)(ws + w.pi);
  int* lcnt = (int*)(ws + w.lcnt);
  int* cnt = (int*)(ws + w.cnt);
  int* overflow = (int*)(ws + w.overflow);
  int* nwork = (int*)(ws + w.nwork);
  int* cursor = (int*)(ws + w.cursor);
  int* loff = (int*)(ws + w.loff);
  int* woff = (int*)(ws + w.woff);
  int* lq_q = (int*)(ws + w.lq_q);
  int* lq_p = (int*)(ws + w.lq_p);
  float* lq_thr = (float*)(ws + w.lq_thr);
  unsigned* wl = (unsigned*)(ws + w.wl);
  int* wl_cnt = (int*)(ws + w.wl_cnt);
  int* cand = (int*)(ws + w.cand);

  EIOKU_HIP_CHECK(hipMemsetAsync(ws + w.zero0, 0, w.zero1 - w.zero0, stream));
  hipLaunchKernelGGL(k_q_prep, dim3((unsigned)((nq + 3) / 4)), dim3(256), 0, stream, q_dev, nq, d, qbf, qn);
  hipLaunchKernelGGL(k_term1, dim3((unsigned)((npairs + 3) / 4)), dim3(256), 0, stream, q_dev, coarse_dev, probes_dev, npairs,
                     nprobe, d, t1);
  EIOKU_LAUNCH_CHECK();
  // tau: the exact scan of one list per query - its nearest one with at least k codes (plane 0 of pd / pi)
  long long* tprobe = (long long*)(ws + w.tprobe);
  hipLaunchKernelGGL(k_tau_probe, dim3((unsigned)((nq + 255) / 256)), dim3(256), 0, stream, probes_dev, nq, nprobe, sizes_dev, k,
                     tprobe);
  // ... of its first kTauRows rows: the k-th best of ANY k codes is a bound, and the whole list (9.9 k rows on average at
  // 10 M rows) cost 0.30 ms per search for a bound that left 35 candidates per query; 1024 rows: see DESIGN.md
  static const int tau_rows = getenv("EIOKU_LSCAN_TAU_ROWS") ? atoi(getenv("EIOKU_LSCAN_TAU_ROWS")) : 1024;
  int rc = scan_launch(q_dev, nq, d, m, tprobe, 1, coarse_dev, pq_dev, offsets_dev, sizes_dev, list_codes_dev,
                       list_ids_dev, k, pd, pi, list_tables_dev, query_tables_dev, stream_, 1, nullptr,
                       tau_rows > k ? tau_rows : k);
  if (rc) return rc;
  hipLaunchKernelGGL(k_lq_fill, dim3((unsigned)std::min(1024, (w.nslots + 255) / 256)), dim3(256), 0, stream, lq_q, lq_thr, w.nslots);
  hipLaunchKernelGGL(k_inv_count, dim3((unsigned)((npairs + 255) / 256)), dim3(256), 0, stream, probes_dev, npairs, sizes_dev, lcnt);
  const int seg = lscan_nw() * kLRT * 32;
  int4* items = (int4*)(ws + w.items);
  hipLaunchKernelGGL(k_inv_scan, dim3(1), dim3(1024), 0, stream, lcnt, sizes_dev, nlist, seg, loff, woff, cursor, nwork);
  hipLaunchKernelGGL(k_inv_items, dim3((unsigned)((w.max_items + 255) / 256)), dim3(256), 0, stream, woff, loff, offsets_dev,
                     sizes_dev, nlist, seg, w.max_items, items);
  hipLaunchKernelGGL(k_inv_scatter, dim3((unsigned)((npairs + 255) / 256)), dim3(256), 0, stream, probes_dev, npairs, nprobe,
                     sizes_dev, loff, cursor, t1, pd, w.K, k, qn, pmax2_dev, lq_q, lq_p, lq_thr);
  EIOKU_LAUNCH_CHECK();
  LScanArgs a;
  a.pqh = (const u32x4k*)pqh_dev;
  a.qbf = qbf;
  a.codes = list_codes_dev;
  a.hx = hx_dev;
  a.items = items;
  a.nwork = nwork;
  a.lq_q = lq_q;
  a.lq_thr = lq_thr;
  a.wl = wl;
  a.wl_cnt = wl_cnt;
  a.wl_cap = w.wl_cap;
  static const int ablate = getenv("EIOKU_LSCAN_ABLATE") ? atoi(getenv("EIOKU_LSCAN_ABLATE")) : 0;
  a.ablate = ablate;
  prof_start(EIOKU_PROF_IVFPQ, stream);
  switch (d / 16) {
    case 4: rc = launch_lscan<4>(a, w.grid, stream); break;
    case 8: rc = launch_lscan<8>(a, w.grid, stream); break;
    case 16: rc = launch_lscan<16>(a, w.grid, stream); break;
    default: rc = launch_lscan<24>(a, w.grid, stream); break;
  }
  prof_stop(EIOKU_PROF_IVFPQ, stream);
  if (rc) return rc;
  hipLaunchKernelGGL(k_lbin, dim3((unsigned)w.grid), dim3(256), 0, stream, wl, wl_cnt, w.wl_cap, lq_q, cand, cnt, cap, overflow);
  {
    static size_t attr_lds = 0;
    if (rr_lds > attr_lds) {
      EIOKU_HIP_CHECK(hipFuncSetAttribute((const void*)k_lrerank, hipFuncAttributeMaxDynamicSharedMemorySize, (int)rr_lds));
      attr_lds = rr_lds;
    }
  }
  hipLaunchKernelGGL(k_lrerank, dim3((unsigned)nq), dim3(256), rr_lds, stream, cand, cnt, cap, m, lq_p, probes_dev, t1,
                     list_tables_dev, query_tables_dev, list_codes_dev, list_ids_dev, k, D_dev, I_dev, overflow);
  EIOKU_LAUNCH_CHECK();
  // a query's candidate list overflowed (no contrast around it, or no list with k codes to bound it): the query-major
  // scan redoes THAT query; a workgroup list overflowed: it redoes all of them; both launches are no-ops otherwise
  rc = scan_launch(q_dev, nq, d, m, probes_dev, nprobe, coarse_dev, pq_dev, offsets_dev, sizes_dev, list_codes_dev,
                   list_ids_dev, k, pd, pi, list_tables_dev, query_tables_dev, stream_, 0, overflow, 0, cnt, cap);
  if (rc) return rc;
  hipLaunchKernelGGL(k_probe_merge, dim3((unsigned)nq), dim3(64), 0, stream, pd, pi, nprobe, nq, w.K, k, D_dev, I_dev, overflow,
                     cnt, cap);
  EIOKU_LAUNCH_CHECK();
  if (stats_out_dev) EIOKU_HIP_CHECK(hipMemcpyAsync(stats_out_dev, overflow, 24, hipMemcpyDeviceToDevice, stream));
  return EIOKU_OK;
}

}  // extern "C"
