// K9 l2_flat_topk: exact kNN over fp32 vectors (FAISS IndexFlatL2 semantics) on gfx950.
//
//   dist(q, x) = max(0, (|q|^2 + |x|^2) - 2 q.x)      squared L2, ascending, ids int64
//
// The dot products run on the exact-fp32 matrix core (v_mfma_f32_32x32x2_f32: a k-ordered fmaf
// chain, no reduced-precision fast path exists on gfx950), which is what the 1e-4 relative bar on
// distances needs.  One wave owns a 32-query tile whose 32 x d operand stays in registers for the
// whole pass; it streams its share of the database through a private, XOR-swizzled LDS ring
// (coalesced 256-B row segments from HBM, register-staged one chunk ahead) and reads each lane's
// four consecutive k-steps with one conflict-free ds_read_b128.  The 32x32 distance tile lands with
// the query on the lane, so every lane keeps a sorted top-K of the rows it has seen in registers:
// a row only costs 16 compares unless it beats the lane's current K-th best.
//
// Per-(workgroup, query) partial lists are merged by k_topk_merge (also used for the cross-rank
// merge after the RCCL all-gather).  Ties are broken by the smaller id everywhere.
//
// Algorithmic bytes: N*d*4 per 32-query pass; FLOPs 2*nq*N*d.
#include "common.h"

#include <cstdlib>

#include <cfloat>

using namespace eioku;

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int kQT = 32;        // queries per wave tile (MFMA N)
constexpr int kRT = 32;        // database rows per MFMA tile (MFMA M)
constexpr int kWaves = 4;

template <int K>
struct TopK {
  float v[K];
  int id[K];
  __device__ __forceinline__ void init() {
#pragma unroll
    for (int p = 0; p < K; ++p) {
      v[p] = FLT_MAX;
      id[p] = 0x7FFFFFFF;
    }
  }
  // Keep ascending by value.  A lane meets its rows in increasing id order, so on equal values the
  // element already in the list has the smaller id and stays ahead: strict compares on the value alone
  // give the (value, id) order without touching the ids (the merges across lanes do compare ids).
  __device__ __forceinline__ void insert(float x, int i) {
#pragma unroll
    for (int p = K - 1; p > 0; --p) {
      const bool shift = v[p - 1] > x;
      const bool here = !shift && (v[p] > x);
      v[p] = shift ? v[p - 1] : (here ? x : v[p]);
      id[p] = shift ? id[p - 1] : (here ? i : id[p]);
    }
    if (v[0] > x) {
      v[0] = x;
      id[0] = i;
    }
  }
  __device__ __forceinline__ bool beats_tail(float x, int) const { return x < v[K - 1]; }
};

struct KnnArgs {
  const float* db;      // [n][d]
  const float* dbnorm;  // [n]
  const float* q;       // [nq][d]
  const float* qnorm;   // [nq]
  long long n;
  int nq, d;
  long long rows_per_block;  // multiple of kRT * kWaves
  float* pd;                 // partial [gridDim.y][gridDim.x][kQT][K]
  int* pi;
  long long id_base;         // added to row ids at the very end (kept in merge)
  // optional exclusive lower bound per query (eioku_index_search_after): only rows with
  // (dist, id) > (lbd[q], lbi[q]) lexicographically are candidates.  NULL = no bound.
  const float* lbd;
  const long long* lbi;
  long long slab_stride;     // rows between slab starts (== rows_per_block for a full search; larger = strided sample)
  const int* gate;           // optional device word: the launch is a no-op when *gate == 0 (scan-path fallback)
};

// Query tile (32 x d) in registers, DB streamed through LDS.
//   narrow (WIDE = false, nq <= 32): grid.y = query tiles; the 4 waves of a workgroup share the queries
//     and split the slab's row tiles; each wave stages its own rows (wave-private LDS, no barriers).
//   wide (WIDE = true): grid.y = groups of 4 query tiles; wave w owns query tile 4*blockIdx.y + w and
//     all 4 waves multiply the SAME staged row tile, so the database is read from HBM once per 128
//     queries instead of once per 32.
// The (row tile, chunk) sequence is flat: the next chunk - also the first chunk of the next row tile -
// is in flight (registers) while the current one is multiplied; row norms ride along with chunk 0 so
// the epilogue issues no global load that would drain the in-order vmcnt queue.
template <int K, int D, bool WIDE>
__global__ __launch_bounds__(256, 1) void k_flat_l2(KnnArgs a) {
  if (a.gate && *a.gate == 0) return;  // uniform
  constexpr int KC = D < 128 ? D : 128;  // dims per LDS chunk (<= 512 B per row)
  constexpr int NCH = D / KC;            // chunks per row
  constexpr int UPR = KC / 4;          // 16-byte units per row chunk
  constexpr int TILE_U = kRT * UPR;     // units per staged chunk
  constexpr int NBUF = WIDE ? 1 : kWaves;
  constexpr int SPT = WIDE ? TILE_U / 256 : TILE_U / 64;  // staged units per thread
  constexpr int LISTS = WIDE ? 2 : 2 * kWaves;            // sorted lists per query to merge in the workgroup
  constexpr int NQ_WG = WIDE ? kQT * kWaves : kQT;        // queries per workgroup
  constexpr size_t STAGE_B = (size_t)NBUF * 2 * TILE_U * 16 + (size_t)NBUF * 2 * kRT * 4;
  constexpr size_t MERGE_B = (size_t)NQ_WG * LISTS * K * 8;
  __shared__ __attribute__((aligned(16))) unsigned char smem[STAGE_B > MERGE_B ? STAGE_B : MERGE_B];
  float4* lds = reinterpret_cast<float4*>(smem);
  float* lnorm = reinterpret_cast<float*>(smem + (size_t)NBUF * 2 * TILE_U * 16);  // [NBUF][2][kRT]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int col = lane & 31, half = lane >> 5;
  const int qt = WIDE ? blockIdx.y * kWaves + wave : blockIdx.y;
  const int qi = qt * kQT + col;
  const bool qvalid = qi < a.nq;

  // ---- query operand: Q[t][j] = q[col][8t + 4*half + j]  (same k permutation as the A reads) ----
  float4 Q[D / 8];
  {
    const float* qp = a.q + (size_t)(qvalid ? qi : 0) * D + 4 * half;
#pragma unroll
    for (int t = 0; t < D / 8; ++t) {
      Q[t] = *reinterpret_cast<const float4*>(qp + 8 * t);
      if (!qvalid) Q[t] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
  }
  const float qn = qvalid ? a.qnorm[qi] : 0.f;
  const bool bounded = a.lbd != nullptr;
  const float lbd = bounded && qvalid ? a.lbd[qi] : -1.f;
  const long long lbi = bounded && qvalid ? a.lbi[qi] : -1;

  TopK<K> top;
  top.init();

  const long long slab0 = (long long)blockIdx.x * a.slab_stride;
  const long long slab1 = min(a.n, slab0 + a.rows_per_block);
  float4* my = lds + (WIDE ? 0 : wave) * (2 * TILE_U);
  float* mynorm = lnorm + (WIDE ? 0 : wave) * (2 * kRT);
  const int sid = WIDE ? tid : lane;            // staging id inside the staging group
  const int sgroup = WIDE ? 256 : 64;
  const long long tile_step = WIDE ? kRT : (long long)kRT * kWaves;
  const long long first = slab0 + (WIDE ? 0 : (long long)wave * kRT);

  float4 stage[SPT];
  float stage_n = 0.f;
  auto issue = [&](long long row0, int ch) {
#pragma unroll
    for (int it = 0; it < SPT; ++it) {
      const int id = sid + sgroup * it;
      const long long r = row0 + (id / UPR);
      stage[it] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (r < slab1) stage[it] = *reinterpret_cast<const float4*>(a.db + (size_t)r * D + ch * KC + (id % UPR) * 4);
    }
    if (ch == 0) {
      stage_n = 0.f;
      if (sid < kRT && row0 + sid < slab1) stage_n = a.dbnorm[row0 + sid];
    }
  };
  auto commit = [&](int buf, int ch, int nbuf) {
#pragma unroll
    for (int it = 0; it < SPT; ++it) {
      const int id = sid + sgroup * it;
      const int row = id / UPR;
      my[buf * TILE_U + row * UPR + ((id % UPR) ^ (row & 15))] = stage[it];
    }
    if (ch == 0 && sid < kRT) mynorm[nbuf * kRT + sid] = stage_n;
  };
  auto sync = [&]() {
    if (WIDE) {
      __syncthreads();
    } else {  // wave-private buffers: LDS ops of one wave complete in order; only pin the compiler
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
  };

  // flat (tile, chunk) walk
  long long row0 = first;
  int buf = 0, nbuf = 0;  // data buffer / norm buffer parity
  if (row0 < slab1) {
    issue(row0, 0);
    commit(0, 0, 0);
  }
  sync();
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  while (row0 < slab1) {
#pragma unroll
    for (int ch = 0; ch < NCH; ++ch) {
      const bool last = ch == NCH - 1;
      const long long nrow = last ? row0 + tile_step : row0;
      const int nch = last ? 0 : ch + 1;
      const bool more = nrow < slab1;
      if (more) issue(nrow, nch);  // in flight during the MFMAs below
      // all LDS fragment reads of the chunk are issued before its MFMAs (counted lgkmcnt waits): a
      // wave alone on its SIMD has nobody to hide a read-then-use latency behind
      float4 xs[KC / 8];
#pragma unroll
      for (int t = 0; t < KC / 8; ++t) xs[t] = my[buf * TILE_U + col * UPR + ((2 * t + half) ^ (col & 15))];
#pragma unroll
      for (int t = 0; t < KC / 8; ++t) {
        const float4 x = xs[t];
        const float4 qq = Q[ch * (KC / 8) + t];
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(x.x, qq.x, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(x.y, qq.y, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(x.z, qq.z, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(x.w, qq.w, acc, 0, 0, 0);
      }
      // pin the issue order: 2 reads ahead, then 4 MFMAs per further read (hipcc otherwise sinks every read
      // to just before its first MFMA and waits lgkmcnt(0) on it)
      __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
#pragma unroll
      for (int t = 0; t < KC / 8; ++t) {
        __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
      }
      if (last) {
        // ---- distances + per-lane top-K: lane = query col, rows (r&3) + 8*(r>>2) + 4*half ----
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int lr = (r & 3) + 8 * (r >> 2) + 4 * half;
          const long long row = row0 + lr;
          if (row < slab1) {
            float dist = (qn + mynorm[nbuf * kRT + lr]) - 2.0f * acc[r];
            dist = dist < 0.f ? 0.f : dist;
            const int id = (int)(row - slab0);  // slab-local, fits 31 bits
            const bool after = !bounded || dist > lbd || (dist == lbd && row > lbi);
            if (after && top.beats_tail(dist, id)) top.insert(dist, id);
          }
          acc[r] = 0.f;
        }
      }
      if (more) commit(buf ^ 1, nch, nbuf ^ (last ? 1 : 0));
      sync();
      buf ^= 1;
      if (last) nbuf ^= 1;
    }
    row0 += tile_step;
  }

  // ---- merge the lists of each query inside the workgroup ----
  __syncthreads();  // every wave is done with the staging buffers before they are reused
  float (*s_mv)[LISTS][K] = reinterpret_cast<float (*)[LISTS][K]>(smem);
  int (*s_mi)[LISTS][K] = reinterpret_cast<int (*)[LISTS][K]>(smem + (size_t)NQ_WG * LISTS * K * 4);
  {
    const int ql = WIDE ? wave * kQT + col : col;
    const int li = WIDE ? half : wave * 2 + half;
#pragma unroll
    for (int p = 0; p < K; ++p) {
      s_mv[ql][li][p] = top.v[p];
      s_mi[ql][li][p] = top.id[p];
    }
  }
  __syncthreads();
  if (tid < NQ_WG) {
    int head[LISTS];
#pragma unroll
    for (int l = 0; l < LISTS; ++l) head[l] = 0;
    const int qtile = WIDE ? blockIdx.y * kWaves + tid / kQT : blockIdx.y;
    const size_t out = (((size_t)qtile * gridDim.x + blockIdx.x) * kQT + (tid % kQT)) * K;
    for (int p = 0; p < K; ++p) {
      float bv = FLT_MAX;
      int bi = 0x7FFFFFFF, bl = 0;
#pragma unroll
      for (int l = 0; l < LISTS; ++l) {
        if (head[l] < K) {
          const float v = s_mv[tid][l][head[l]];
          const int i = s_mi[tid][l][head[l]];
          if (v < bv || (v == bv && i < bi)) {
            bv = v;
            bi = i;
            bl = l;
          }
        }
      }
#pragma unroll
      for (int l = 0; l < LISTS; ++l)
        if (l == bl) head[l]++;
      a.pd[out + p] = bv;
      a.pi[out + p] = bi == 0x7FFFFFFF ? -1 : bi;  // -1 when fewer than K rows exist
    }
  }
}

// ---------------------------------------------------------------------------------------------
// Wide variant on the bf16 matrix pipe (nq > 64).  Exact-fp32 MFMA is 157 TFLOP/s: at 128 queries per database
// pass the wide kernel above is bound by it (60 % of that peak = 83 ms per 1024 queries over 10 M x 384), not
// by HBM.  Here every fp32 value is split into two bf16 terms, v = hi + lo + O(2^-17 |v|), and
//     q . x  ~  q_hi . x_hi  +  q_hi . x_lo  +  q_lo . x_hi          (3 v_mfma_f32_32x32x16_bf16 per 16 dims)
// accumulated in fp32: 96 matrix-pipe cycles per 16 dims instead of 512.  The dropped lo.lo term and the split
// remainders are ~2^-16 relative per product, ~1e-6 absolute on the distance of unit vectors: inside the path's
// 1e-4 tolerance (tests compare against a float64 brute force).  Queries are split once into registers; database
// rows are split by the staging threads on their way into LDS (hi and lo planes, 16-byte units of 8 dims).
// Norms stay the fp32 ones.
// ---------------------------------------------------------------------------------------------
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4k __attribute__((ext_vector_type(4)));
typedef unsigned u32x2k __attribute__((ext_vector_type(2)));

__device__ __forceinline__ void split_bf16x2(float a, float b, unsigned& hi, unsigned& lo) {
  const unsigned ua = __float_as_uint(a), ub = __float_as_uint(b);
  hi = (ua >> 16) | (ub & 0xFFFF0000u);  // truncation: the remainder below is then exact in fp32
  const float ra = a - __uint_as_float(ua & 0xFFFF0000u), rb = b - __uint_as_float(ub & 0xFFFF0000u);
  lo = (__float_as_uint(ra) >> 16) | (__float_as_uint(rb) & 0xFFFF0000u);
}

template <int K, int D>
__global__ __launch_bounds__(256, 1) void k_flat_l2_bf(KnnArgs a) {
  if (a.gate && *a.gate == 0) return;  // uniform
  constexpr int KC = D < 128 ? D : 128;  // dims per LDS chunk
  constexpr int NCH = D / KC;
  constexpr int NS = KC / 16;            // 16-dim MFMA steps per chunk
  constexpr int PPR = KC / 8;            // 8-dim units per row and plane
  constexpr int UPR = 2 * PPR;           // units per row: [hi plane | lo plane]
  constexpr int TILE_U = kRT * UPR;
  constexpr int SPT = kRT * (KC / 4) / 256;  // staged float4 per thread and chunk
  constexpr int LISTS = 2, NQ_WG = kQT * kWaves;
  // a whole row tile (NCH chunks) per buffer, two buffers: the NEXT tile's NCH x 16 KB are in flight (registers)
  // while this one is multiplied -- with one chunk ahead a CU had 16 KB outstanding and streamed 9 GB/s
  constexpr size_t STAGE_B = (size_t)2 * NCH * TILE_U * 16 + (size_t)2 * kRT * 4;
  constexpr size_t MERGE_B = (size_t)NQ_WG * LISTS * K * 8;
  static_assert(KC % 16 == 0 && D % KC == 0 && (kRT * (KC / 4)) % 256 == 0, "tile shape");
  __shared__ __attribute__((aligned(16))) unsigned char smem[STAGE_B > MERGE_B ? STAGE_B : MERGE_B];
  u32x4k* lds = reinterpret_cast<u32x4k*>(smem);
  float* lnorm = reinterpret_cast<float*>(smem + (size_t)2 * NCH * TILE_U * 16);  // [2][kRT]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int col = lane & 31, half = lane >> 5;
  const int qt = blockIdx.y * kWaves + wave;
  const int qi = qt * kQT + col;
  const bool qvalid = qi < a.nq;

  // query operand of step s: dims 16 s + 8 half + [0, 8), hi and lo terms
  u32x4k Qh[D / 16], Ql[D / 16];
  {
    const float* qp = a.q + (size_t)(qvalid ? qi : 0) * D + 8 * half;
#pragma unroll
    for (int s = 0; s < D / 16; ++s) {
      const float4 v0 = *reinterpret_cast<const float4*>(qp + 16 * s);
      const float4 v1 = *reinterpret_cast<const float4*>(qp + 16 * s + 4);
      unsigned h[4], l[4];
      split_bf16x2(v0.x, v0.y, h[0], l[0]);
      split_bf16x2(v0.z, v0.w, h[1], l[1]);
      split_bf16x2(v1.x, v1.y, h[2], l[2]);
      split_bf16x2(v1.z, v1.w, h[3], l[3]);
      Qh[s] = qvalid ? u32x4k{h[0], h[1], h[2], h[3]} : u32x4k{0, 0, 0, 0};
      Ql[s] = qvalid ? u32x4k{l[0], l[1], l[2], l[3]} : u32x4k{0, 0, 0, 0};
    }
  }
  const float qn = qvalid ? a.qnorm[qi] : 0.f;
  TopK<K> top;
  top.init();

  const long long slab0 = (long long)blockIdx.x * a.slab_stride;
  const long long slab1 = min(a.n, slab0 + a.rows_per_block);
  // register ring two tiles deep: tile t+2 is requested while tile t is multiplied (2 x NCH x 16 KB in flight per CU)
  float4 stage[2][NCH][SPT];
  float stage_n[2] = {0.f, 0.f};
  auto issue = [&](long long row0, int slot) {
#pragma unroll
    for (int ch = 0; ch < NCH; ++ch)
#pragma unroll
      for (int it = 0; it < SPT; ++it) {
        const int id = tid + 256 * it;
        long long r = row0 + id / (KC / 4);
        if (r >= slab1) r = slab1 - 1;  // rows past the slab: loaded, multiplied, never inserted
        stage[slot][ch][it] = *reinterpret_cast<const float4*>(a.db + (size_t)r * D + ch * KC + (id % (KC / 4)) * 4);
      }
    long long r = row0 + (tid < kRT ? tid : 0);
    if (r >= slab1) r = slab1 - 1;
    stage_n[slot] = a.dbnorm[r];
  };
  auto commit = [&](int buf, int slot) {
#pragma unroll
    for (int ch = 0; ch < NCH; ++ch)
#pragma unroll
      for (int it = 0; it < SPT; ++it) {
        const int id = tid + 256 * it;
        const int row = id / (KC / 4), fu = id % (KC / 4);  // float4 index inside the row chunk: dims 4 fu .. 4 fu + 3
        const int p = (fu >> 1) ^ (row & (PPR - 1) & 15), sub = fu & 1;
        unsigned h0, l0, h1, l1;
        split_bf16x2(stage[slot][ch][it].x, stage[slot][ch][it].y, h0, l0);
        split_bf16x2(stage[slot][ch][it].z, stage[slot][ch][it].w, h1, l1);
        const u32x2k hi = u32x2k{h0, h1}, lo = u32x2k{l0, l1};
        unsigned char* base = reinterpret_cast<unsigned char*>(lds + (buf * NCH + ch) * TILE_U + row * UPR);
        *reinterpret_cast<u32x2k*>(base + p * 16 + sub * 8) = hi;
        *reinterpret_cast<u32x2k*>(base + (PPR + p) * 16 + sub * 8) = lo;
      }
    if (tid < kRT) lnorm[buf * kRT + tid] = stage_n[slot];
  };
  auto multiply = [&](long long row0, int buf) {
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
    for (int ch = 0; ch < NCH; ++ch) {
      const u32x4k* rowp = lds + (buf * NCH + ch) * TILE_U + col * UPR;
      u32x4k xh[NS], xl[NS];
#pragma unroll
      for (int s = 0; s < NS; ++s) {
        const int p = (2 * s + half) ^ (col & (PPR - 1) & 15);
        xh[s] = rowp[p];
        xl[s] = rowp[PPR + p];
      }
#pragma unroll
      for (int s = 0; s < NS; ++s) {
        const bf16x8 h = __builtin_bit_cast(bf16x8, xh[s]), l = __builtin_bit_cast(bf16x8, xl[s]);
        const bf16x8 qh = __builtin_bit_cast(bf16x8, Qh[ch * NS + s]), ql = __builtin_bit_cast(bf16x8, Ql[ch * NS + s]);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(h, qh, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(l, qh, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(h, ql, acc, 0, 0, 0);
      }
      __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
#pragma unroll
      for (int s = 0; s < NS; ++s) {
        __builtin_amdgcn_sched_group_barrier(0x008, 3, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
      }
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int lr = (r & 3) + 8 * (r >> 2) + 4 * half;
      const long long row = row0 + lr;
      if (row < slab1) {
        float dist = (qn + lnorm[buf * kRT + lr]) - 2.0f * acc[r];
        dist = dist < 0.f ? 0.f : dist;
        const int id = (int)(row - slab0);
        if (top.beats_tail(dist, id)) top.insert(dist, id);
      }
    }
  };

  // tile t lives in LDS buffer t & 1 and travelled through register slot t & 1
  long long row0 = slab0;
  if (row0 < slab1) {
    issue(row0, 0);
    issue(row0 + kRT, 1);  // clamped inside the slab when it does not exist
    commit(0, 0);
  }
  __syncthreads();
  while (row0 < slab1) {
    // even tile: multiply buffer 0 while tile+2 loads into slot 0; tile+1 (slot 1) is committed to buffer 1
#pragma unroll
    for (int par = 0; par < 2; ++par) {
      if (row0 >= slab1) break;
      issue(row0 + 2 * kRT, par);
      multiply(row0, par);
      commit(par ^ 1, par ^ 1);
      __syncthreads();
      row0 += kRT;
    }
  }

  // ---- merge the two lists of each query (the two lane halves) inside the workgroup ----
  __syncthreads();
  float (*s_mv)[LISTS][K] = reinterpret_cast<float (*)[LISTS][K]>(smem);
  int (*s_mi)[LISTS][K] = reinterpret_cast<int (*)[LISTS][K]>(smem + (size_t)NQ_WG * LISTS * K * 4);
  {
    const int ql = wave * kQT + col;
#pragma unroll
    for (int p = 0; p < K; ++p) {
      s_mv[ql][half][p] = top.v[p];
      s_mi[ql][half][p] = top.id[p];
    }
  }
  __syncthreads();
  if (tid < NQ_WG) {
    int head[LISTS] = {0, 0};
    const int qtile = blockIdx.y * kWaves + tid / kQT;
    const size_t out = (((size_t)qtile * gridDim.x + blockIdx.x) * kQT + (tid % kQT)) * K;
    for (int p = 0; p < K; ++p) {
      float bv = FLT_MAX;
      int bi = 0x7FFFFFFF, bl = 0;
#pragma unroll
      for (int l = 0; l < LISTS; ++l) {
        if (head[l] < K) {
          const float v = s_mv[tid][l][head[l]];
          const int i = s_mi[tid][l][head[l]];
          if (v < bv || (v == bv && i < bi)) {
            bv = v;
            bi = i;
            bl = l;
          }
        }
      }
#pragma unroll
      for (int l = 0; l < LISTS; ++l)
        if (l == bl) head[l]++;
      a.pd[out + p] = bv;
      a.pi[out + p] = bi == 0x7FFFFFFF ? -1 : bi;
    }
  }
}

__global__ __launch_bounds__(256) void k_row_norms(const float* x, long long n, int d, float* out) {
  const int lane = threadIdx.x & 63;
  long long row = (blockIdx.x * 256ll + threadIdx.x) >> 6;
  const long long nw = ((long long)gridDim.x * 256) >> 6;
  for (; row < n; row += nw) {
    float s = 0.f;
    for (int c = lane * 4; c < d; c += 256) {
      const float4 v = *reinterpret_cast<const float4*>(x + (size_t)row * d + c);
      s += v.x * v.x;
      s += v.y * v.y;
      s += v.z * v.z;
      s += v.w * v.w;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
    if (lane == 0) out[row] = s;
  }
}

// Merge L sorted lists of Kin (value, local id) per query into the k best (value, id) ascending.
// One wave per query: lanes stride over the lists keeping a register top-K, then K rounds of
// wave-wide argmin pop the winners.  `list_base[l]` (optional) is added to the ids of list l.
template <int K>
__global__ __launch_bounds__(64) void k_topk_merge(const float* pd, const int* pi, const long long* pi64,
                                                   int L, int nq, int kin, int qstride_lists,
                                                   const long long* list_base, long long rows_per_list,
                                                   int k, float* D, long long* I, const int* gate = nullptr) {
  if (gate && *gate == 0) return;
  const int q = blockIdx.x, lane = threadIdx.x;
  // partial layout: [qtile][list][kQT][kin] when qstride_lists > 0 (search partials), else [list][nq][kin]
  TopK<K> top;
  top.init();
  long long gid[K];
#pragma unroll
  for (int p = 0; p < K; ++p) gid[p] = -1;
  for (int l = lane; l < L; l += 64) {
    size_t base;
    if (qstride_lists > 0) base = ((((size_t)(q / kQT) * L + l) * kQT) + (q % kQT)) * kin;
    else base = ((size_t)l * nq + q) * kin;
    for (int p = 0; p < kin; ++p) {
      const float v = pd[base + p];
      long long id;
      if (pi64) id = pi64[base + p];
      else {
        const int li = pi[base + p];
        id = li < 0 ? -1 : (long long)li + (list_base ? list_base[l] : (long long)l * rows_per_list);
      }
      if (id < 0) break;  // lists are sorted: the rest is padding
      // TopK keeps 31-bit ids for the tie rule; carry the 64-bit id alongside via the slot it lands in
      const bool better_tail = (v < top.v[K - 1]) || (v == top.v[K - 1] && id < gid[K - 1]) || gid[K - 1] < 0;
      if (!better_tail) break;  // ascending list: nothing later can enter either
#pragma unroll
      for (int s = K - 1; s > 0; --s) {
        const bool shift = gid[s - 1] < 0 || (top.v[s - 1] > v) || (top.v[s - 1] == v && gid[s - 1] > id);
        const bool here = !shift && (gid[s] < 0 || (top.v[s] > v) || (top.v[s] == v && gid[s] > id));
        const float nv = shift ? top.v[s - 1] : (here ? v : top.v[s]);
        const long long ni = shift ? gid[s - 1] : (here ? id : gid[s]);
        top.v[s] = nv;
        gid[s] = ni;
      }
      if (gid[0] < 0 || (top.v[0] > v) || (top.v[0] == v && gid[0] > id)) {
        top.v[0] = v;
        gid[0] = id;
      }
    }
  }
  // K rounds of wave argmin over the list heads
  for (int r = 0; r < k; ++r) {
    float bv = gid[0] < 0 ? FLT_MAX : top.v[0];
    long long bi = gid[0] < 0 ? 0x7FFFFFFFFFFFFFFFll : gid[0];
    int bl = lane;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      const float ov = __shfl_xor(bv, off, 64);
      const long long oi = __shfl_xor(bi, off, 64);
      const int ol = __shfl_xor(bl, off, 64);
      if (ov < bv || (ov == bv && oi < bi) || (ov == bv && oi == bi && ol < bl)) {
        bv = ov;
        bi = oi;
        bl = ol;
      }
    }
    const bool found = bi != 0x7FFFFFFFFFFFFFFFll;
    if (lane == 0) {
      D[(size_t)q * k + r] = found ? bv : FLT_MAX;
      I[(size_t)q * k + r] = found ? bi : -1;
    }
    if (found && lane == bl) {  // pop
#pragma unroll
      for (int s = 0; s < K - 1; ++s) {
        top.v[s] = top.v[s + 1];
        gid[s] = gid[s + 1];
      }
      top.v[K - 1] = FLT_MAX;
      gid[K - 1] = -1;
    }
  }
}


// The same merge for L <= 64 SORTED lists of 32-bit ids (every search partial is one): a lane owns one list and only ever
// looks at its head, so nothing is inserted - k rounds of wave argmin over the heads, the winner advances (its next
// entry is already in a register, the one after it is in flight).  The general kernel above builds a K-deep sorted list
// per lane first: 150 us for 1024 queries x 32 lists x 32 entries (the IVF probe selection), against ~10 us here.
// Same order as above: (distance, id) ascending; ids are unique across lists.
__global__ __launch_bounds__(64) void k_topk_merge_heads(const float* __restrict__ pd, const int* __restrict__ pi, int L, int kin,
                                                         long long rows_per_list, int k, float* __restrict__ D,
                                                         long long* __restrict__ I, const int* __restrict__ gate) {
  if (gate && *gate == 0) return;
  const int q = blockIdx.x, lane = threadIdx.x;
  // partial layout [qtile][list][kQT][kin]
  const bool have = lane < L;
  const size_t base = ((((size_t)(q / kQT) * L + (have ? lane : 0)) * kQT) + (q % kQT)) * kin;
  const long long idb = (long long)lane * rows_per_list;
  int h = 0;  // entries consumed
  float cv = FLT_MAX, nv = FLT_MAX;
  int ci = -1, ni = -1;
  if (have) {
    cv = pd[base];
    ci = pi[base];
    if (kin > 1) {
      nv = pd[base + 1];
      ni = pi[base + 1];
    }
  }
  for (int r = 0; r < k; ++r) {
    float bv = ci < 0 ? FLT_MAX : cv;
    long long bi = ci < 0 ? 0x7FFFFFFFFFFFFFFFll : (long long)ci + idb;
    int bl = lane;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      const float ov = __shfl_xor(bv, off, 64);
      const long long oi = __shfl_xor(bi, off, 64);
      const int ol = __shfl_xor(bl, off, 64);
      if (ov < bv || (ov == bv && oi < bi) || (ov == bv && oi == bi && ol < bl)) {
        bv = ov;
        bi = oi;
        bl = ol;
      }
    }
    const bool found = bi != 0x7FFFFFFFFFFFFFFFll;
    if (lane == 0) {
      D[(size_t)q * k + r] = found ? bv : FLT_MAX;
      I[(size_t)q * k + r] = found ? bi : -1;
    }
    if (found && lane == bl) {  // pop: the list is sorted and padded with id < 0 at its end
      ++h;
      cv = nv;
      ci = ni;
      if (h + 1 < kin && ni >= 0) {
        nv = pd[base + h + 1];
        ni = pi[base + h + 1];
      } else {
        ni = -1;
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------
// Scan path (large N, any nq): database tiles stationary in registers.
//
// The kernels above keep a 32-query tile in registers and stream the database past it, so a 1024-query
// search reads the database 8 times (once per 128 queries).  Here the roles are swapped: a wave keeps a tile
// of 32 database rows in registers as the MFMA A operand (a bf16 copy of the rows, fragment-tiled at add/attach
// time so that a lane's fragment is one coalesced 16-byte load) and ALL query tiles stream past it through
// LDS (LDS-DMA, one tile ahead; the query plane is <= 0.75 MB and lives in every XCD's L2).  HBM sees the bf16
// plane exactly once per search (half the bytes of the fp32 rows) and the kernel is bound by the bf16 matrix pipe.
//
// A row-stationary wave meets every query, so per-query top-k lists cannot live in its registers.  Instead a
// per-query upper bound tau_q on the k-th distance comes from an exact search of a strided SAMPLE of the rows
// (the k-th best of any subset bounds the k-th best of the whole), the scan keeps every (query, row) whose
// distance may be <= tau_q (expected k * N / S pairs per query), k_scan_bin sorts them into per-query lists and
// k_scan_select re-computes their distances in fp32 from the fp32 rows and picks the k best.  A list that
// overflows raises a device flag and the launch of the register-tile kernels that follows - gated on that flag -
// recomputes the search (tested by forcing tiny lists).
//
// The products are ONE bf16 term, q.x ~ qh.xh (a third of the matrix work of the split-bf16 kernel above).  bf16 keeps 8
// significant bits, so round-to-nearest has unit roundoff 2^-8 PER OPERAND: qh_i xh_i = q_i x_i (1 + a)(1 + b) with
// |a|, |b| <= 2^-8, i.e. |q.x - qh.xh| <= (2^-7 + 2^-16) sum|q_i x_i| <= 2^-7 (1 + 2^-9) |q| |x|, and the filter admits
// everything within that rigorous margin of tau_q (round 2 used half of it - ADVICE r2: a row just under a bf16
// midpoint in every coordinate could be filtered out; tests/test_knn_scan_gpu.py holds that case now); the distances
// that are returned never see bf16.
//
// Measured at 10 M x 384, nq 1024 (profiles/r02_knn_*): 7.4 ms for the scan = 1.06 PFLOP/s of algorithmic work;
// ablation builds put the matrix-only floor of this structure at 4.85 ms, the filter at +1.5, the workgroup barrier
// at +1.2, the query DMA at +1.15 and the LDS fragment reads at +0.7 (they add almost linearly: the 12 waves of a
// workgroup run their phases in lockstep behind the one barrier per query tile).
// ---------------------------------------------------------------------------------------------
typedef __attribute__((address_space(3))) void lds_void_t;
typedef __attribute__((address_space(1))) const void gbl_cvoid_t;

__device__ __forceinline__ unsigned bf16_rne_bits(float f) {  // finite inputs
  unsigned u = __float_as_uint(f);
  u += 0x7FFFu + ((u >> 16) & 1u);
  return u >> 16;
}

// rows [n][D] fp32 -> fragment-tiled bf16 plane: unit ((tile * NS + s) * 2 + half) * 32 + r holds dims
// 16 s + 8 half + [0, 8) of row 32 tile + r (the 32x32x16 A / B operand of lane 32 half + r), rounded to nearest.
// Optional per-row half norms (0.5 |x|^2, +inf for rows >= n) and per-tile max |x| (over valid rows).
// Tiles [first_tile, first_tile + gridDim.x / SPLIT); SPLIT workgroups share a tile (few tiles: the queries).
template <int D, int SPLIT>
__global__ __launch_bounds__(256) void k_split_planes(const float* __restrict__ x, long long n, long long first_tile,
                                                      const float* __restrict__ norms, u32x4k* __restrict__ hi,
                                                      float* __restrict__ hnorm, float* __restrict__ tmax) {
  constexpr int NS = D / 16, PU = NS * 64;
  const long long tile = first_tile + blockIdx.x / SPLIT;
  const int part = blockIdx.x % SPLIT;
  for (int o = part * (PU / SPLIT) + threadIdx.x; o < (part + 1) * (PU / SPLIT); o += 256) {
    const int s = o >> 6, h = (o >> 5) & 1, r = o & 31;
    const long long row = tile * 32 + r;
    float v[8];
    if (row < n) {
      const float4 a = *reinterpret_cast<const float4*>(x + (size_t)row * D + 16 * s + 8 * h);
      const float4 b = *reinterpret_cast<const float4*>(x + (size_t)row * D + 16 * s + 8 * h + 4);
      v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
    } else {
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = 0.f;
    }
    unsigned hb[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) hb[j] = bf16_rne_bits(v[j]);
    hi[(size_t)tile * PU + o] = u32x4k{hb[0] | (hb[1] << 16), hb[2] | (hb[3] << 16), hb[4] | (hb[5] << 16), hb[6] | (hb[7] << 16)};
  }
  if (hnorm && part == 0 && threadIdx.x < 64) {
    const int r = threadIdx.x & 31;
    const long long row = tile * 32 + r;
    const float nn = row < n ? norms[row] : 0.f;
    if (threadIdx.x < 32) hnorm[tile * 32 + r] = row < n ? 0.5f * nn : __builtin_inff();
    float m = sqrtf(nn);
#pragma unroll
    for (int off = 16; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off, 64));
    if (threadIdx.x == 0) tmax[tile] = m;
  }
}

struct ScanArgs {
  const u32x4k* xh;      // [ntiles][NS*64] units
  const float* hnorm;    // [ntiles*32] 0.5 |x|^2, +inf beyond n
  const float* tmax;     // [ntiles] max |x| of the tile's valid rows
  const float* xmax;     // [1] max |x| over all rows
  const u32x4k* qh;      // [nqt][NS*64]
  const float* qnorm;    // [nq]
  const float* tau;      // sample search result [nq][tau_k]; the bound is its last column
  int tau_k;
  long long n, ntiles;   // ntiles: row tiles THIS launch visits; tile t is physical tile t * tstride
  long long tstride;     // 1: every tile; > 1: the strided subset that tightens the bound first (scan_search)
  int nq, nqt;
  // candidates leave the scan as (query, row) pairs in a list private to the workgroup (an LDS counter hands out the
  // slots: no global atomic, nothing to wait for); k_scan_bin sorts them into the per-query lists afterwards
  unsigned* wl;          // [gridDim.x][wl_cap][2]
  int* wl_cnt;           // [gridDim.x]
  int wl_cap;
};

// NW waves per workgroup, each owning RT 32-row tiles (tiles (wt * NW + wave) * RT + [0, RT)); workgroups walk the
// workgroup tiles grid-stride.  Two staged query tiles: one in flight (LDS-DMA) while the other is multiplied.
// RT = 2 feeds two MFMAs from every query fragment read (half the LDS bytes and half the L2 -> LDS query traffic per
// product) at 2 x 96 operand registers: two waves per SIMD instead of three.
template <int D, int NW, int RT>
__global__ __launch_bounds__(NW * 64) void k_l2_scan(ScanArgs a) {
  constexpr int NS = D / 16, PU = NS * 64;  // steps; 16-B units per tile
  constexpr int NDMA = (PU / 64 + NW - 1) / NW;  // LDS-DMA instructions per wave and query tile
  extern __shared__ __attribute__((aligned(16))) unsigned char dyn_smem[];
  constexpr int NB = 4;                                                     // query-tile ring: two tiles per barrier
  u32x4k* qbuf = reinterpret_cast<u32x4k*>(dyn_smem);                      // [NB][PU]
  float* s_hc = reinterpret_cast<float*>(dyn_smem + (size_t)NB * PU * 16);  // [nqt*32] 0.5 (|q|^2 - tau')
  float* s_sq = s_hc + a.nqt * 32;                                         // [nqt*32] 2^-8-scaled |q|
  float* s_hx = s_sq + a.nqt * 32;                                         // [NW*RT*32] 0.5 |x|^2 of each wave's rows
  int* s_cnt = reinterpret_cast<int*>(s_hx + NW * RT * 32);                // slots handed out in this workgroup's list

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int col = lane & 31, half = lane >> 5;

  for (int q = tid; q < a.nqt * 32; q += NW * 64) {
    float hc = __builtin_inff(), sq = 0.f;
    if (q < a.nq) {
      const float t = a.tau[(size_t)q * a.tau_k + (a.tau_k - 1)];
      const float qn = a.qnorm[q];
      // slack: the sample's distances come from another kernel as |q|^2 + |x|^2 - 2 q.x in fp32 with split-bf16
      // products, so their error scales with the NORMS (a few eps (|q| + |x|)^2), not with t: un-normalised rows with
      // large norms and small distances need the second term
      const float nrm = sqrtf(qn) + a.xmax[0];
      hc = 0.5f * (qn - (t + 2e-5f * (1.0f + t) + 1e-6f * nrm * nrm));
      sq = 0.0078125f * 1.002f * sqrtf(qn);  // 2^-7 (1 + 2^-9) |q|, rounded up
    }
    s_hc[q] = hc;
    s_sq[q] = sq;
  }
  if (tid == 0) *s_cnt = 0;
  unsigned* const my_list = a.wl + (size_t)blockIdx.x * a.wl_cap * 2;

  const long long nwt = (a.ntiles + NW * RT - 1) / (NW * RT);
  // every wave issues exactly NDMA LDS-DMA instructions per query tile (pieces past the tile re-copy its last
  // piece: same bytes, same place)
  auto stage_q = [&](int qt, int buf) {
#pragma unroll
    for (int j = 0; j < NDMA; ++j) {
      int u = (j * NW + wave) * 64;
      if (u >= PU) u = PU - 64;
      __builtin_amdgcn_global_load_lds((gbl_cvoid_t*)(a.qh + (size_t)qt * PU + u + lane),
                                       (lds_void_t*)(qbuf + (size_t)buf * PU + u), 16, 0, 0);
    }
  };

  u32x4k xh[RT][NS];
  float sxm[RT], hxmin[RT];  // each tile's max |x| and min 0.5 |x|^2 (wave-uniform)
  long long wt = blockIdx.x;
  auto tile_of = [&](long long w, int i) {
    const long long t = (w * NW + wave) * RT + i;
    return (t < a.ntiles ? t : a.ntiles - 1) * a.tstride;  // clamped duplicates are masked through `own` below
  };
  // lanes 0..31 each read one row's half norm; the hot loop only needs the tile's minimum, the per-row values wait in
  // LDS for the rare tile that may hold a candidate (16 registers less in the loop)
  auto tile_min_hx = [&](long long t, int i) {
    float m = a.hnorm[t * 32 + col];
    if (half == 0) s_hx[(wave * RT + i) * 32 + col] = m;  // read back only by this wave: LDS operations of a wave stay in order
#pragma unroll
    for (int off = 16; off > 0; off >>= 1) m = fminf(m, __shfl_xor(m, off, 64));
    return m;
  };
  // flat sequence of staged query tiles over all of this workgroup's row tiles: entry e is query tile e % nqt
  const long long my_tiles = wt < nwt ? (nwt - 1 - wt) / gridDim.x + 1 : 0;
  const int total = (int)(my_tiles * a.nqt);
  if (wt < nwt) {
#pragma unroll
    for (int i = 0; i < RT; ++i) {
      const long long t = tile_of(wt, i);
#pragma unroll
      for (int s = 0; s < NS; ++s) xh[i][s] = a.xh[(size_t)t * PU + s * 64 + lane];
      sxm[i] = a.tmax[t];
      hxmin[i] = tile_min_hx(t, i);
    }
    // the first tile's operands are in registers before any LDS-DMA is issued: inside the loop the compiler must never
    // find a reason to wait for "all vector memory" in the middle of the products
    __builtin_amdgcn_s_waitcnt(0x0f70);
    __builtin_amdgcn_sched_barrier(0);
    stage_q(0, 0);
    if (total > 1) stage_q(1 % a.nqt, 1);
  }
  // Everything below is 32-bit and incremental: a 64-bit `% nqt` per iteration is a ~100-instruction scalar sequence
  int it = 0;                     // entries consumed so far: entry `it` lives in buffer it & (NB - 1)
  int stage_qt = 2 % a.nqt;       // query tile of the next entry to stage
  for (; wt < nwt; wt += gridDim.x) {
    bool own[RT];
    long long row0[RT], tn[RT];
    const long long nwt_next = wt + gridDim.x;
    const bool more_tiles = nwt_next < nwt;
#pragma unroll
    for (int i = 0; i < RT; ++i) {
      own[i] = (wt * NW + wave) * RT + i < a.ntiles;  // a clamped duplicate tile appends nothing
      row0[i] = tile_of(wt, i) * 32;
      tn[i] = more_tiles ? tile_of(nwt_next, i) : 0;
    }
    for (int qt = 0; qt < a.nqt; ++qt, ++it) {
      const bool last_qt = qt == a.nqt - 1;
      // Every second entry the workgroup meets: entries `it` and `it + 1` have landed (this wave's pieces: vmcnt;
      // everyone's: the barrier) - and so have the operand registers refilled during the previous iteration - and every
      // wave is done reading the two buffers that the DMAs issued below overwrite.  Between two meetings the waves run
      // free: one wave's filter and the next tile's first LDS reads overlap another wave's products on the same SIMD
      // (with a barrier per tile all waves sat in the same phase).  A raw s_barrier is not a memory fence to the
      // optimiser (__syncthreads' fence would do, at the price of its own waits): the empty asm statements keep LDS reads
      // from being hoisted across it, sched_barrier pins the machine order.
      const bool meet = (it & 1) == 0;
      if (meet || qt == 0) __builtin_amdgcn_s_waitcnt(0x0f70);  // vmcnt(0); qt == 0: the refilled row operands
      if (meet) {
        __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0)
        asm volatile("" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int e = 2; e < 4; ++e) {
          if (it + e < total) stage_q(stage_qt, (it + e) & (NB - 1));
          stage_qt = stage_qt + 1 == a.nqt ? 0 : stage_qt + 1;
        }
      }
      __builtin_amdgcn_sched_barrier(0);
      const u32x4k* qb = qbuf + (size_t)(it & (NB - 1)) * PU;
      const float hcq = s_hc[qt * 32 + col];
      const float sqq = s_sq[qt * 32 + col];
      f32x16 acc[RT];
#pragma unroll
      for (int i = 0; i < RT; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
      // query fragments PF steps ahead of their MFMAs (a lone ds_read -> wait -> MFMA chain exposes the LDS latency)
      constexpr int PF = RT == 2 ? 2 : 4;
      u32x4k bh[PF];
#pragma unroll
      for (int s = 0; s < PF; ++s) bh[s] = qb[s * 64 + lane];
#pragma unroll
      for (int s = 0; s < NS; ++s) {
        const bf16x8 ch = __builtin_bit_cast(bf16x8, bh[s % PF]);
#pragma unroll
        for (int i = 0; i < RT; ++i)
          acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, xh[i][s]), ch, acc[i], 0, 0, 0);
        if (s + PF < NS) bh[s % PF] = qb[(s + PF) * 64 + lane];
        if (last_qt && more_tiles) {
          // The next row tile replaces this one IN PLACE, step by step: the load is inline assembly with the operand
          // register as a read-write operand, so it lands in the very register the MFMA above just read (a plain
          // assignment made the compiler keep a second set of 96 operand registers: one wave per SIMD fewer).  The
          // compiler does not know these registers are pending: the vmcnt(0) + sched_barrier at the top of the next
          // iteration stands between the loads and their first use.  Address = wave-uniform SGPR base (tile, 4 KiB group
          // of steps) + lane * 16 in one VGPR + immediate: no per-step 64-bit address registers live across the loop.
#pragma unroll
          for (int i = 0; i < RT; ++i) {
            const unsigned char* base = reinterpret_cast<const unsigned char*>(a.xh) + ((size_t)tn[i] * PU + (s & ~3) * 64) * 16;
            asm volatile("global_load_dwordx4 %0, %1, %2 offset:%3" : "+v"(xh[i][s]) : "v"(lane * 16), "s"(base), "n"((s & 3) * 1024));
          }
        }
      }
      if (!last_qt) {  // pin the issue order of the steady-state body: PF reads ahead, then MFMAs and reads alternate
        __builtin_amdgcn_sched_group_barrier(0x100, PF, 0);
#pragma unroll
        for (int s = 0; s < NS; ++s) {
          __builtin_amdgcn_sched_group_barrier(0x008, RT, 0);
          __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        }
      }
      // filter: dist <= tau'  <=>  product >= 0.5 |x|^2 + 0.5 (|q|^2 - tau') - margin.  First the tile's largest product
      // against the tile's smallest row norm (8 max3 + one compare for 16 products); the per-row test runs only in the
      // rare tile that may hold a candidate.
#pragma unroll
      for (int i = 0; i < RT; ++i) {
        const f32x16& c = acc[i];
        const float thr_q = hcq - sqq * sxm[i];
        // "does any of the 16 products reach T = hxmin + thr_q?"  on the products' BIT PATTERNS as signed integers (r3):
        // for T > 0 a product >= T is a positive float >= T, and positive floats order like their bits; negative products
        // have the sign bit set = negative integers, below every positive T.  8 v_max3_i32 and no float canonicalisation
        // (fmaxf cost a v_max_f32 x, x per accumulator on top of the maxima: ~25 VALU per tile instead of 9; a
        // v_max3_f32 in inline asm was tried in r2 and reverted: the hazard recogniser does not see an asm's read of a
        // just-written MFMA result).  T <= 0 (no bound for the query) takes the per-row test, which is exact.
        auto ib = [&](int r) { return __float_as_int(c[r]); };
        const int i01 = max(max(ib(0), ib(1)), ib(2)), i23 = max(max(ib(3), ib(4)), ib(5));
        const int i45 = max(max(ib(6), ib(7)), ib(8)), i67 = max(max(ib(9), ib(10)), ib(11));
        const int i89 = max(max(ib(12), ib(13)), ib(14));
        const int imx = max(max(max(i01, i23), i45), max(max(i67, i89), ib(15)));
        const float T = hxmin[i] + thr_q;
        if ((!(T > 0.f) || imx >= __float_as_int(T)) && own[i]) {
          const int qi = qt * 32 + col;
          int h4 = 4 * half;
          asm volatile("" : "+v"(h4));  // keeps the 16 row offsets from being precomputed into registers that live across the loop
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            const float4 hx = *reinterpret_cast<const float4*>(s_hx + (wave * RT + i) * 32 + 8 * g + h4);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              const float hxr = j == 0 ? hx.x : (j == 1 ? hx.y : (j == 2 ? hx.z : hx.w));
              if (c[4 * g + j] >= hxr + thr_q) {
                const int pos = atomicAdd(s_cnt, 1);  // LDS
                if (pos < a.wl_cap) {
                  my_list[2 * (size_t)pos] = (unsigned)qi;
                  my_list[2 * (size_t)pos + 1] = (unsigned)((int)row0[i] + 8 * g + j + h4);
                }
              }
            }
          }
        }
      }
      if (last_qt && more_tiles) {
#pragma unroll
        for (int i = 0; i < RT; ++i) {
          sxm[i] = a.tmax[tn[i]];
          hxmin[i] = tile_min_hx(tn[i], i);
        }
      }
    }
  }
  __syncthreads();
  if (tid == 0) a.wl_cnt[blockIdx.x] = *s_cnt;
}

// (query, row) pairs of the scan's workgroup lists -> per-query candidate lists.  One workgroup per list.
__global__ __launch_bounds__(256) void k_scan_bin(const unsigned* __restrict__ wl, const int* __restrict__ wl_cnt, int wl_cap,
                                                  int* __restrict__ cand_i, int* __restrict__ cnt, int cap,
                                                  int* __restrict__ overflow) {
  int c = wl_cnt[blockIdx.x];
  if (c > wl_cap) {
    if (threadIdx.x == 0) atomicOr(overflow, 1);
    c = wl_cap;
  }
  const unsigned* list = wl + (size_t)blockIdx.x * wl_cap * 2;
  for (int i = threadIdx.x; i < c; i += 256) {
    const int qi = (int)list[2 * i], row = (int)list[2 * i + 1];
    const int pos = atomicAdd(cnt + qi, 1);
    if (pos < cap) cand_i[(size_t)qi * cap + pos] = row;
  }
}

// One workgroup per query: fp32 distances of its candidates, sum (q_i - x_i)^2 from the fp32 rows (no
// |q|^2 + |x|^2 - 2 q.x cancellation: an exact copy of the query scores exactly 0), then the k best by (dist, id).
template <int D>
__global__ __launch_bounds__(256) void k_scan_select(const float* __restrict__ db, const float* __restrict__ q,
                                                     const int* __restrict__ cand_i, const int* __restrict__ cnt, int cap, int k,
                                                     float* __restrict__ Dout, long long* __restrict__ Iout,
                                                     int* __restrict__ overflow) {
  extern __shared__ __attribute__((aligned(16))) unsigned char dyn_smem[];
  float* sd = reinterpret_cast<float*>(dyn_smem);  // [cap]
  int* si = reinterpret_cast<int*>(sd + cap);      // [cap]
  __shared__ float s_bv[4];
  __shared__ int s_bi[4];
  const int qi = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  int c = cnt[qi];
  if (c > cap) {
    if (tid == 0) atomicOr(overflow, 1);
    c = cap;
  }
  for (int j = tid; j < c; j += 256) si[j] = cand_i[(size_t)qi * cap + j];
  __syncthreads();
  {
    // a half wave per candidate, four candidates in flight: lane l of 32 owns dims 4 l + 128 t + [0, 4)
    constexpr int NV = D / 128, U = 4;
    const int l32 = lane & 31, hw = tid >> 5;
    float4 qv[NV];
#pragma unroll
    for (int t = 0; t < NV; ++t) qv[t] = *reinterpret_cast<const float4*>(q + (size_t)qi * D + 128 * t + 4 * l32);
    for (int j0 = hw; j0 < c; j0 += 8 * U) {
      float4 xv[U][NV];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int j = j0 + 8 * u;
        const int r = si[j < c ? j : j0];
#pragma unroll
        for (int t = 0; t < NV; ++t) xv[u][t] = *reinterpret_cast<const float4*>(db + (size_t)r * D + 128 * t + 4 * l32);
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        float sum = 0.f;
#pragma unroll
        for (int t = 0; t < NV; ++t) {
          float e;
          e = qv[t].x - xv[u][t].x; sum = fmaf(e, e, sum);
          e = qv[t].y - xv[u][t].y; sum = fmaf(e, e, sum);
          e = qv[t].z - xv[u][t].z; sum = fmaf(e, e, sum);
          e = qv[t].w - xv[u][t].w; sum = fmaf(e, e, sum);
        }
#pragma unroll
        for (int off = 16; off > 0; off >>= 1) sum += __shfl_xor(sum, off, 64);
        const int j = j0 + 8 * u;
        if (l32 == 0 && j < c) sd[j] = sum;
      }
    }
  }
  __syncthreads();
  float lv = -1.f;
  int li = -1;  // last emitted (dist, id): distances are >= 0
  for (int r = 0; r < k; ++r) {
    float bv = FLT_MAX;
    int bi = 0x7FFFFFFF;
    for (int j = tid; j < c; j += 256) {
      const float v = sd[j];
      const int i = si[j];
      const bool after = v > lv || (v == lv && i > li);
      if (after && (v < bv || (v == bv && i < bi))) {
        bv = v;
        bi = i;
      }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      const float ov = __shfl_xor(bv, off, 64);
      const int oi = __shfl_xor(bi, off, 64);
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
    if (tid == 0) {
      Dout[(size_t)qi * k + r] = bi == 0x7FFFFFFF ? FLT_MAX : bv;
      Iout[(size_t)qi * k + r] = bi == 0x7FFFFFFF ? -1 : (long long)bi;
    }
    lv = bv;
    li = bi;
    if (bi == 0x7FFFFFFF) lv = FLT_MAX;  // exhausted: the remaining rounds emit padding
  }
}

}  // namespace

// ---------------------------------------------------------------------------------------------
// C ABI
// ---------------------------------------------------------------------------------------------
struct eioku_index {
  int d = 0;
  long long n = 0, cap = 0;
  float* x = nullptr;      // owned unless attached
  float* norms = nullptr;  // owned
  long long norms_cap = 0;
  bool attached = false;
  // search workspace
  float* qbuf = nullptr; size_t qcap = 0;
  float* qnorm = nullptr; size_t qncap = 0;
  float* pd = nullptr; size_t pdcap = 0;
  int* pi = nullptr; size_t picap = 0;
  float* dout = nullptr; size_t dcap = 0;
  long long* iout = nullptr; size_t icap = 0;
  // scan path (see k_l2_scan): fragment-tiled bf16 plane of the rows, built lazily and extended on add()
  unsigned char* xh = nullptr; size_t xhcap = 0;
  float* hnorm = nullptr; size_t hncap = 0;
  float* tmax = nullptr; size_t tmcap = 0;
  float* xmax = nullptr;  // [1] max |x| over all rows (the norm-proportional slack of the scan's bound)
  long long planes_n = 0;      // rows covered by the plane
  unsigned char* qh = nullptr; size_t qhcap = 0;
  unsigned* wl = nullptr; size_t wlcap = 0;
  int* wl_cnt = nullptr; size_t wlccap = 0;
  float* tau = nullptr; size_t taucap = 0;
  long long* tau_i = nullptr; size_t tauicap = 0;
  float* tau1 = nullptr; size_t tau1cap = 0;       // the strided pre-scan's exact top-k (its last column tightens tau)
  long long* tau1_i = nullptr; size_t tau1icap = 0;
  int* cand_i = nullptr; size_t cicap = 0;
  int* cnt = nullptr; size_t cntcap = 0;  // [nq] counters + 1 overflow word
  // parameters (eioku_index_set_param)
  int scan_mode = 1;           // 0: register-tile kernels only; 1: scan path for wide searches over large indexes
  int scan_cap = 4096;         // candidate slots per query
  long long scan_min_rows = 262144;
  long long scan_sample = 0;   // sample rows for the bound (0: automatic)
  int scan_min_nq = 1;         // fewest queries that take the scan path.  One pass over the bf16 plane (half the bytes
                               // of the fp32 rows) beats the register-tile kernels at EVERY nq once N >= scan_min_rows:
                               // 10 M x 384: nq 1 / 64 = 1.53 / 1.70 ms against 3.79 / 7.92 ms; 262 144 rows: 0.29 / 0.40
                               // against 0.32 / 0.41 ms (tools/knn_nq_sweep.py, profiles/r02_knn_nq_sweep.jsonl)
  int scan_prescan = 32;       // stride of the pre-scan's row tiles (0: no pre-scan, the sample alone bounds the scan)
  int scan_rt = 2;             // row tiles per wave: 2 (8 waves per workgroup; 2-3 % faster at 10 M x 384) or 1 (12 waves)
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

int norms_for(const float* x, long long n, int d, float* out, hipStream_t stream) {
  if (n == 0) return EIOKU_OK;
  long long blocks = (n * 64 + 255) / 256;
  const long long cap = (long long)num_cus() * 16;
  if (blocks > cap) blocks = cap;
  hipLaunchKernelGGL(k_row_norms, dim3((unsigned)blocks), dim3(256), 0, stream, x, n, d, out);
  EIOKU_LAUNCH_CHECK();
  return EIOKU_OK;
}

template <int K>
int launch_search_bf(int d, const KnnArgs& a, dim3 grid, hipStream_t stream) {
  switch (d) {
    case 128: hipLaunchKernelGGL((k_flat_l2_bf<K, 128>), grid, dim3(256), 0, stream, a); break;
    case 256: hipLaunchKernelGGL((k_flat_l2_bf<K, 256>), grid, dim3(256), 0, stream, a); break;
    case 384: hipLaunchKernelGGL((k_flat_l2_bf<K, 384>), grid, dim3(256), 0, stream, a); break;
    default: return -1;
  }
  EIOKU_LAUNCH_CHECK();
  return EIOKU_OK;
}

template <int K, bool WIDE>
int launch_search_k(int d, const KnnArgs& a, dim3 grid, hipStream_t stream) {
  switch (d) {
    case 64: hipLaunchKernelGGL((k_flat_l2<K, 64, WIDE>), grid, dim3(256), 0, stream, a); break;
    case 128: hipLaunchKernelGGL((k_flat_l2<K, 128, WIDE>), grid, dim3(256), 0, stream, a); break;
    case 256: hipLaunchKernelGGL((k_flat_l2<K, 256, WIDE>), grid, dim3(256), 0, stream, a); break;
    case 384: hipLaunchKernelGGL((k_flat_l2<K, 384, WIDE>), grid, dim3(256), 0, stream, a); break;
    case 512: hipLaunchKernelGGL((k_flat_l2<K, 512, WIDE>), grid, dim3(256), 0, stream, a); break;
    default:
      set_error("dimension %d not supported (64, 128, 256, 384, 512)", d);
      return EIOKU_EINVAL;
  }
  EIOKU_LAUNCH_CHECK();
  return EIOKU_OK;
}

}  // namespace

extern "C" {

int eioku_index_flat_create(int d, eioku_index** out) {
  EIOKU_REQUIRE_INIT();
  EIOKU_REQUIRE(out, "NULL out");
  EIOKU_REQUIRE(d == 64 || d == 128 || d == 256 || d == 384 || d == 512,
                "dimension %d not supported (64, 128, 256, 384, 512)", d);
  auto* ix = new eioku_index();
  ix->d = d;
  *out = ix;
  return EIOKU_OK;
}

void eioku_index_destroy(eioku_index* ix) {
  if (!ix) return;
  (void)hipDeviceSynchronize();
  if (ix->x && !ix->attached) (void)hipFree(ix->x);
  void* bufs[] = {ix->norms, ix->qbuf, ix->qnorm, ix->pd, ix->pi, ix->dout, ix->iout, ix->xh, ix->hnorm,
                  ix->tmax, ix->xmax, ix->qh, ix->wl, ix->wl_cnt, ix->tau, ix->tau_i, ix->tau1, ix->tau1_i, ix->cand_i, ix->cnt};
  for (void* b : bufs)
    if (b) (void)hipFree(b);
  delete ix;
}

long long eioku_index_ntotal(const eioku_index* ix) { return ix ? ix->n : 0; }

int eioku_index_reset(eioku_index* ix) {
  EIOKU_REQUIRE(ix, "NULL index");
  if (ix->attached) {
    ix->x = nullptr;
    ix->attached = false;
    ix->cap = 0;
  }
  ix->n = 0;
  ix->planes_n = 0;
  return EIOKU_OK;
}

int eioku_index_add(eioku_index* ix, const float* x, long long n, int mem, void* stream_) {
  EIOKU_REQUIRE_INIT();
  EIOKU_REQUIRE(ix && (x || n == 0) && n >= 0, "bad argument");
  EIOKU_REQUIRE(!ix->attached, "index wraps an attached buffer; reset it before add()");
  EIOKU_REQUIRE(mem == EIOKU_MEM_HOST || mem == EIOKU_MEM_DEVICE, "bad mem flag %d", mem);
  if (n == 0) return EIOKU_OK;
  hipStream_t stream = (hipStream_t)stream_;
  const long long need = ix->n + n;
  if (need > ix->cap) {
    long long ncap = ix->cap ? ix->cap : 1024;
    while (ncap < need) ncap *= 2;
    // Earlier add() calls queued their copy + norms kernel on the caller's stream, which may be a
    // non-blocking one the NULL stream does not wait for: move the old rows ON that stream and drain it
    // before the old buffers are freed (a blocking hipMemcpy here could read rows that have not landed).
    float* nx = nullptr;
    float* nn = nullptr;
    EIOKU_HIP_CHECK(hipMalloc((void**)&nx, (size_t)ncap * ix->d * sizeof(float)));
    if (hipMalloc((void**)&nn, (size_t)ncap * sizeof(float)) != hipSuccess) {
      (void)hipFree(nx);
      set_error("hipMalloc of %lld norms failed", ncap);
      return EIOKU_ENOMEM;
    }
    if (ix->n) {
      EIOKU_HIP_CHECK(hipMemcpyAsync(nx, ix->x, (size_t)ix->n * ix->d * sizeof(float), hipMemcpyDeviceToDevice, stream));
      EIOKU_HIP_CHECK(hipMemcpyAsync(nn, ix->norms, (size_t)ix->n * sizeof(float), hipMemcpyDeviceToDevice, stream));
    }
    EIOKU_HIP_CHECK(hipStreamSynchronize(stream));
    if (ix->x) (void)hipFree(ix->x);
    if (ix->norms) (void)hipFree(ix->norms);
    ix->x = nx;
    ix->norms = nn;
    ix->norms_cap = ncap;
    ix->cap = ncap;
  }
  float* dst = ix->x + (size_t)ix->n * ix->d;
  EIOKU_HIP_CHECK(hipMemcpyAsync(dst, x, (size_t)n * ix->d * sizeof(float),
                                 mem == EIOKU_MEM_HOST ? hipMemcpyHostToDevice : hipMemcpyDeviceToDevice, stream));
  int rc = norms_for(dst, n, ix->d, ix->norms + ix->n, stream);
  if (rc) return rc;
  ix->n = need;
  if (mem == EIOKU_MEM_HOST) EIOKU_HIP_CHECK(hipStreamSynchronize(stream));
  return EIOKU_OK;
}

int eioku_index_attach(eioku_index* ix, float* x_dev, long long n, void* stream_) {
  EIOKU_REQUIRE_INIT();
  EIOKU_REQUIRE(ix && x_dev && n >= 0, "bad argument");
  EIOKU_REQUIRE(((uintptr_t)x_dev & 15) == 0, "attached buffer must be 16-byte aligned");
  if (ix->x && !ix->attached) (void)hipFree(ix->x);
  ix->x = x_dev;
  ix->attached = true;
  ix->n = n;
  ix->cap = n;
  ix->planes_n = 0;  // rebuilt by the next wide search
  if (ix->norms_cap < n) {
    if (ix->norms) (void)hipFree(ix->norms);
    ix->norms = nullptr;
    EIOKU_HIP_CHECK(hipMalloc((void**)&ix->norms, (size_t)(n ? n : 1) * sizeof(float)));
    ix->norms_cap = n;
  }
  return norms_for(x_dev, n, ix->d, ix->norms, (hipStream_t)stream_);
}

}  // extern "C"

namespace {

// The register-tile kernels (k_flat_l2 / k_flat_l2_bf) + k_topk_merge over rows of the index: a full search
// (sample_slabs == 0) or an exact search of `sample_slabs` strided slabs of `sample_rows` rows each (the scan
// path's bound).  All pointers are device pointers.  gate (optional): device word, the launches are no-ops when 0.
int legacy_search(eioku_index* ix, const float* dq, int nq, int k, const float* lbd, const long long* lbi, float* dD,
                  long long* dI, const int* gate, int sample_slabs, long long sample_rows, bool prof, hipStream_t stream) {
  const int d = ix->d;
  const int K = k == 1 ? 1 : (k <= 16 ? 16 : 32);  // k == 1: coarse assignment (k-means / IVF), a single compare per row
  const int qtiles = (nq + kQT - 1) / kQT;
  // wide: 4 query tiles share every staged row tile (one HBM pass per 128 queries)
  // two query tiles: two narrow passes (all 4 waves of every workgroup busy, 2 x 3.6 ms at 10 M x 384) beat one
  // wide pass with half of its waves idle (11.4 ms)
  const bool wide = qtiles > 2;
  const int ygroups = wide ? (qtiles + kWaves - 1) / kWaves : qtiles;
  long long rpb, slabs, stride;
  if (sample_slabs > 0) {
    rpb = sample_rows;
    slabs = sample_slabs;
    stride = ix->n / sample_slabs;
  } else {
    // One workgroup is resident per CU (the query tile fills the register file), and a slab that is too
    // short never leaves the phase where most rows still enter some lane's top-K (the insertion path runs
    // whenever ANY lane of the wave inserts): ~2 workgroups per CU over the whole grid.
    long long want = (long long)num_cus() * 2 / ygroups;
    if (want < 1) want = 1;
    rpb = (ix->n + want - 1) / want;
    rpb = ((rpb + kRT * kWaves - 1) / (kRT * kWaves)) * (kRT * kWaves);
    if (rpb < kRT * kWaves) rpb = kRT * kWaves;
    slabs = ix->n ? (ix->n + rpb - 1) / rpb : 1;
    stride = rpb;
  }
  EIOKU_REQUIRE(rpb < (1ll << 31), "slab too large");
  const size_t pn = (size_t)ygroups * (wide ? kWaves : 1) * slabs * kQT * K;
  int rc = grow(&ix->pd, &ix->pdcap, pn * sizeof(float));
  if (rc) return rc;
  rc = grow(&ix->pi, &ix->picap, pn * sizeof(int));
  if (rc) return rc;
  KnnArgs a;
  a.db = ix->x;
  a.dbnorm = ix->norms;
  a.q = dq;
  a.qnorm = ix->qnorm;
  a.n = ix->n;
  a.nq = nq;
  a.d = d;
  a.rows_per_block = rpb;
  a.pd = ix->pd;
  a.pi = ix->pi;
  a.id_base = 0;
  a.lbd = lbd;
  a.lbi = lbi;
  a.slab_stride = stride;
  a.gate = gate;
  dim3 grid((unsigned)slabs, (unsigned)ygroups);
  if (prof) prof_start(EIOKU_PROF_KNN, stream);
  // wide searches (nq > 64) with k <= 16 over d in {128, 256, 384}: split-bf16 kernel (see k_flat_l2_bf); a bounded
  // search (lbd) stays on the exact-fp32 kernels so that successive rounds see bit-identical distances
  rc = -1;
  if (wide && K == 16 && !lbd) rc = launch_search_bf<16>(d, a, grid, stream);
  if (rc != -1) {
  } else if (K == 1) rc = wide ? launch_search_k<1, true>(d, a, grid, stream) : launch_search_k<1, false>(d, a, grid, stream);
  else if (wide) rc = K == 16 ? launch_search_k<16, true>(d, a, grid, stream) : launch_search_k<32, true>(d, a, grid, stream);
  else rc = K == 16 ? launch_search_k<16, false>(d, a, grid, stream) : launch_search_k<32, false>(d, a, grid, stream);
  if (prof) prof_stop(EIOKU_PROF_KNN, stream);
  if (rc) return rc;
  if (slabs <= 64 && K > 1)  // a lane per sorted partial list: heads only
    hipLaunchKernelGGL(k_topk_merge_heads, dim3(nq), dim3(64), 0, stream, ix->pd, ix->pi, (int)slabs, K, stride, k, dD, dI, gate);
  else if (K == 1)
    hipLaunchKernelGGL((k_topk_merge<1>), dim3(nq), dim3(64), 0, stream, ix->pd, ix->pi, (const long long*)nullptr,
                       (int)slabs, nq, K, 1, (const long long*)nullptr, stride, k, dD, dI, gate);
  else if (K == 16)
    hipLaunchKernelGGL((k_topk_merge<16>), dim3(nq), dim3(64), 0, stream, ix->pd, ix->pi, (const long long*)nullptr,
                       (int)slabs, nq, K, 1, (const long long*)nullptr, stride, k, dD, dI, gate);
  else
    hipLaunchKernelGGL((k_topk_merge<32>), dim3(nq), dim3(64), 0, stream, ix->pd, ix->pi, (const long long*)nullptr,
                       (int)slabs, nq, K, 1, (const long long*)nullptr, stride, k, dD, dI, gate);
  EIOKU_LAUNCH_CHECK();
  return EIOKU_OK;
}

int split_planes(int d, const float* x, long long n, long long first_tile, long long tiles, const float* norms, void* hi,
                 float* hnorm, float* tmax, bool few_tiles, hipStream_t stream) {
  if (tiles <= 0) return EIOKU_OK;
#define EIOKU_SPLIT(D_)                                                                                                  \
  case D_:                                                                                                               \
    if (few_tiles)                                                                                                       \
      hipLaunchKernelGGL((k_split_planes<D_, 8>), dim3((unsigned)tiles * 8), dim3(256), 0, stream, x, n, first_tile, norms, \
                         (u32x4k*)hi, hnorm, tmax);                                                                      \
    else                                                                                                                 \
      hipLaunchKernelGGL((k_split_planes<D_, 1>), dim3((unsigned)tiles), dim3(256), 0, stream, x, n, first_tile, norms,   \
                         (u32x4k*)hi, hnorm, tmax);                                                                      \
    break;
  switch (d) {
    EIOKU_SPLIT(128)
    EIOKU_SPLIT(256)
    EIOKU_SPLIT(384)
    default:
      set_error("scan path: dimension %d not supported", d);
      return EIOKU_EINVAL;
  }
#undef EIOKU_SPLIT
  EIOKU_LAUNCH_CHECK();
  return EIOKU_OK;
}

// a buffer that keeps its contents when it grows (the plane is extended on add())
template <typename T>
int grow_keep(T** p, size_t* cap, size_t bytes, size_t keep, hipStream_t stream) {
  if (*cap >= bytes) return EIOKU_OK;
  size_t want = *cap ? *cap : bytes;
  while (want < bytes) want += want / 2 + 1;
  T* np = nullptr;
  EIOKU_HIP_CHECK(hipMalloc((void**)&np, want));
  if (*p) {
    if (keep) EIOKU_HIP_CHECK(hipMemcpyAsync(np, *p, keep, hipMemcpyDeviceToDevice, stream));
    EIOKU_HIP_CHECK(hipStreamSynchronize(stream));
    (void)hipDeviceSynchronize();
    (void)hipFree(*p);
  }
  *p = np;
  *cap = want;
  return EIOKU_OK;
}

__global__ __launch_bounds__(1024) void k_max_f32(const float* __restrict__ v, long long n, float* __restrict__ out) {
  __shared__ float s_m[16];
  float m = 0.f;
  for (long long i = threadIdx.x; i < n; i += 1024) m = fmaxf(m, v[i]);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
  if ((threadIdx.x & 63) == 0) s_m[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w = 1; w < 16; ++w) m = fmaxf(m, s_m[w]);
    out[0] = m;
  }
}

int ensure_planes(eioku_index* ix, hipStream_t stream) {
  const int d = ix->d;
  const size_t tile_b = (size_t)d * 32 * 2;  // bytes per 32-row tile
  const long long ntiles = (ix->n + 31) / 32;
  const long long keep_tiles = ix->planes_n / 32;  // complete tiles stay valid
  int rc = grow_keep(&ix->xh, &ix->xhcap, (size_t)ntiles * tile_b, (size_t)keep_tiles * tile_b, stream);
  if (rc) return rc;
  rc = grow_keep(&ix->hnorm, &ix->hncap, (size_t)ntiles * 32 * sizeof(float), (size_t)keep_tiles * 32 * sizeof(float), stream);
  if (rc) return rc;
  rc = grow_keep(&ix->tmax, &ix->tmcap, (size_t)ntiles * sizeof(float), (size_t)keep_tiles * sizeof(float), stream);
  if (rc) return rc;
  if (ix->planes_n != ix->n) {
    rc = split_planes(d, ix->x, ix->n, keep_tiles, ntiles - keep_tiles, ix->norms, ix->xh, ix->hnorm, ix->tmax, false, stream);
    if (rc) return rc;
    if (!ix->xmax) EIOKU_HIP_CHECK(hipMalloc((void**)&ix->xmax, sizeof(float)));
    hipLaunchKernelGGL(k_max_f32, dim3(1), dim3(1024), 0, stream, ix->tmax, ntiles, ix->xmax);
    EIOKU_LAUNCH_CHECK();
    ix->planes_n = ix->n;
  }
  return EIOKU_OK;
}

// the bound of query q = the smaller of two valid ones (the k-th best of ANY k rows bounds the k-th best of all)
__global__ void k_tau_min(float* __restrict__ tau, const float* __restrict__ other, int nq, int k) {
  const int q = blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= nq) return;
  const size_t j = (size_t)q * k + (k - 1);
  tau[j] = fminf(tau[j], other[j]);
}

template <int D>
int launch_scan(eioku_index* ix, ScanArgs& a, int rt, long long* grid_out, hipStream_t stream) {
  constexpr int PU = (D / 16) * 64;
  const int nw = rt == 2 ? 8 : 12;
  const size_t lds = (size_t)4 * PU * 16 + (size_t)a.nqt * 32 * 4 * 2 + (size_t)nw * rt * 32 * 4 + 16;
  const long long nwt = (a.ntiles + (long long)nw * rt - 1) / ((long long)nw * rt);
  // one workgroup per CU: its 12 (8) waves are the three (two) per SIMD that 96 (192) operand registers allow
  long long grid = num_cus();
  if (grid > nwt) grid = nwt;
  // workgroup lists: 64 MB of pairs shared out evenly (expected use: k N / sample rows per query in total)
  long long wl_cap = ((64ll << 20) / 8) / grid;
  if (wl_cap > (1 << 20)) wl_cap = 1 << 20;
  if (ix->scan_cap < 256) wl_cap = ix->scan_cap;  // tests shrink both kinds of list to force the overflow paths
  int rc = grow(&ix->wl, &ix->wlcap, (size_t)grid * wl_cap * 8);
  if (rc) return rc;
  rc = grow(&ix->wl_cnt, &ix->wlccap, (size_t)grid * sizeof(int));
  if (rc) return rc;
  a.wl = ix->wl;
  a.wl_cnt = ix->wl_cnt;
  a.wl_cap = (int)wl_cap;
  *grid_out = grid;
  if (rt == 2) {
    EIOKU_HIP_CHECK(hipFuncSetAttribute((const void*)k_l2_scan<D, 8, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL((k_l2_scan<D, 8, 2>), dim3((unsigned)grid), dim3(8 * 64), lds, stream, a);
  } else {
    EIOKU_HIP_CHECK(hipFuncSetAttribute((const void*)k_l2_scan<D, 12, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL((k_l2_scan<D, 12, 1>), dim3((unsigned)grid), dim3(12 * 64), lds, stream, a);
  }
  EIOKU_LAUNCH_CHECK();
  return EIOKU_OK;
}

template <int D>
int launch_select(const eioku_index* ix, const float* dq, int nq, int cap, int k, float* dD, long long* dI, int* overflow,
                  hipStream_t stream) {
  const size_t lds = (size_t)cap * 8;
  EIOKU_HIP_CHECK(hipFuncSetAttribute((const void*)k_scan_select<D>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipLaunchKernelGGL((k_scan_select<D>), dim3(nq), dim3(256), lds, stream, ix->x, dq, ix->cand_i, ix->cnt, cap, k, dD, dI,
                     overflow);
  EIOKU_LAUNCH_CHECK();
  return EIOKU_OK;
}

// nq <= 1024 device queries through the scan path; qnorm already holds their norms
int scan_search(eioku_index* ix, const float* dq, int nq, int k, float* dD, long long* dI, hipStream_t stream) {
  const int d = ix->d;
  int rc = ensure_planes(ix, stream);
  if (rc) return rc;
  const int nqt = (nq + 31) / 32;
  rc = grow(&ix->qh, &ix->qhcap, (size_t)nqt * d * 32 * 2);
  if (rc) return rc;
  rc = grow(&ix->tau, &ix->taucap, (size_t)nq * k * sizeof(float));
  if (rc) return rc;
  rc = grow(&ix->tau_i, &ix->tauicap, (size_t)nq * k * sizeof(long long));
  if (rc) return rc;
  const int cap = ix->scan_cap;
  rc = grow(&ix->cand_i, &ix->cicap, (size_t)nq * cap * sizeof(int));
  if (rc) return rc;
  rc = grow(&ix->cnt, &ix->cntcap, ((size_t)nq + 1) * sizeof(int));
  if (rc) return rc;
  int* overflow = ix->cnt + nq;
  EIOKU_HIP_CHECK(hipMemsetAsync(ix->cnt, 0, ((size_t)nq + 1) * sizeof(int), stream));
  rc = split_planes(d, dq, nq, 0, nqt, nullptr, ix->qh, nullptr, nullptr, true, stream);
  if (rc) return rc;
  // Bound, in two steps (the k-th best of ANY subset of the rows bounds the k-th best of all of them):
  //  1. exact register-tile search of a strided sample of N / sample_div rows, in slabs long enough to amortise each
  //     workgroup's 128-query operand load;
  //  2. (pre-scan) THIS scan path over every `stride`-th row tile with the bound of step 1, its candidates re-ranked
  //     exactly: the k-th best of N / stride rows.  With the sample alone (N / 128 rows) a query keeps ~128 k
  //     candidates in the full scan - 1.3 M list entries to bin and 1.2 GB of rows to gather and re-rank per 1024
  //     queries at 10 M rows (0.9 ms) on top of the sample search (0.9 ms); with a 512th sampled and a 32nd pre-scanned
  //     the three small steps cost about as much as one of those.
  const long long ntiles_all = (ix->n + 31) / 32;
  int stride = ix->scan_prescan;
  if (stride > 0 && ntiles_all / stride < 64) stride = 0;  // too few rows for a meaningful second bound
  long long sample = ix->scan_sample > 0 ? ix->scan_sample : ix->n / (stride ? 512 : 128);
  if (sample < 32768) sample = 32768;
  if (sample > 262144) sample = 262144;
  // exactly one round of workgroups on the chip: (query groups of 128) x slabs <= CUs - every workgroup of that kernel
  // pays ~50 us to load and split its 128-query operand before it sees a row
  const int ygroups = (nq + 127) / 128;
  int sslabs = num_cus() / ygroups;
  if (sslabs < 1) sslabs = 1;
  long long srows = ((sample / sslabs + 127) / 128) * 128;
  if (srows < 512) srows = 512;
  while (sslabs > 1 && (long long)sslabs * srows * 2 > ix->n) --sslabs;
  rc = legacy_search(ix, dq, nq, k, nullptr, nullptr, ix->tau, ix->tau_i, nullptr, sslabs, srows, false, stream);
  if (rc) return rc;
  ScanArgs a;
  a.xh = (const u32x4k*)ix->xh;
  a.hnorm = ix->hnorm;
  a.tmax = ix->tmax;
  a.xmax = ix->xmax;
  a.qh = (const u32x4k*)ix->qh;
  a.qnorm = ix->qnorm;
  a.tau = ix->tau;
  a.tau_k = k;
  a.n = ix->n;
  a.nq = nq;
  a.nqt = nqt;
  long long sgrid = 0;
  auto scan_bin_select = [&](long long ntiles, long long tstride, float* outD, long long* outI, int* oflow, bool prof) -> int {
    a.ntiles = ntiles;
    a.tstride = tstride;
    if (prof) prof_start(EIOKU_PROF_KNN, stream);
    int r;
    switch (d) {
      case 128: r = launch_scan<128>(ix, a, ix->scan_rt, &sgrid, stream); break;
      case 256: r = launch_scan<256>(ix, a, ix->scan_rt, &sgrid, stream); break;
      default: r = launch_scan<384>(ix, a, ix->scan_rt, &sgrid, stream); break;
    }
    if (prof) prof_stop(EIOKU_PROF_KNN, stream);
    if (r) return r;
    hipLaunchKernelGGL(k_scan_bin, dim3((unsigned)sgrid), dim3(256), 0, stream, ix->wl, ix->wl_cnt, a.wl_cap, ix->cand_i,
                       ix->cnt, cap, oflow);
    EIOKU_LAUNCH_CHECK();
    switch (d) {
      case 128: return launch_select<128>(ix, dq, nq, cap, k, outD, outI, oflow, stream);
      case 256: return launch_select<256>(ix, dq, nq, cap, k, outD, outI, oflow, stream);
      default: return launch_select<384>(ix, dq, nq, cap, k, outD, outI, oflow, stream);
    }
  };
  if (stride > 0) {
    rc = grow(&ix->tau1, &ix->tau1cap, (size_t)nq * k * sizeof(float));
    if (rc) return rc;
    rc = grow(&ix->tau1_i, &ix->tau1icap, (size_t)nq * k * sizeof(long long));
    if (rc) return rc;
    // a pre-scan list that overflows is merely truncated: the k best of the entries that did fit are k real rows, so
    // their k-th distance is still a valid bound (fewer than k entries: FLT_MAX, and the sample's bound stands).  Its
    // overflow word is the one the memset below clears again.
    rc = scan_bin_select((ntiles_all + stride - 1) / stride, stride, ix->tau1, ix->tau1_i, overflow, false);
    if (rc) return rc;
    hipLaunchKernelGGL(k_tau_min, dim3((unsigned)((nq + 255) / 256)), dim3(256), 0, stream, ix->tau, ix->tau1, nq, k);
    EIOKU_LAUNCH_CHECK();
    EIOKU_HIP_CHECK(hipMemsetAsync(ix->cnt, 0, ((size_t)nq + 1) * sizeof(int), stream));
  }
  rc = scan_bin_select(ntiles_all, 1, dD, dI, overflow, true);
  if (rc) return rc;
  // a list overflowed (adversarial data for the sample bound): the gated register-tile search redoes the group
  return legacy_search(ix, dq, nq, k, nullptr, nullptr, dD, dI, overflow, 0, 0, false, stream);
}

int search_impl(eioku_index* ix, const float* q, int nq, int k, const float* lbD, const int64_t* lbI, float* D,
                int64_t* I, int mem, void* stream_) {
  EIOKU_REQUIRE_INIT();
  EIOKU_REQUIRE(ix && nq >= 0 && k >= 1, "bad argument");
  EIOKU_REQUIRE(k <= 32, "k=%d not supported (k <= 32)", k);
  EIOKU_REQUIRE(mem == EIOKU_MEM_HOST || mem == EIOKU_MEM_DEVICE, "bad mem flag %d", mem);
  EIOKU_REQUIRE((lbD == nullptr) == (lbI == nullptr), "lower bound needs both distances and ids");
  if (nq == 0) return EIOKU_OK;
  EIOKU_REQUIRE(q && D && I, "NULL buffer");
  EIOKU_REQUIRE(ix->n < (1ll << 31), "index too large");
  hipStream_t stream = (hipStream_t)stream_;
  const int d = ix->d;
  int rc;
  const float* dq = q;
  const float* dlbD = lbD;
  const long long* dlbI = (const long long*)lbI;
  if (mem == EIOKU_MEM_HOST) {
    const size_t qb = (((size_t)nq * d * sizeof(float)) + 255) & ~(size_t)255;
    const size_t lb = lbD ? (size_t)nq * (sizeof(float) + sizeof(long long)) + 256 : 0;
    rc = grow(&ix->qbuf, &ix->qcap, qb + lb);
    if (rc) return rc;
    EIOKU_HIP_CHECK(hipMemcpyAsync(ix->qbuf, q, (size_t)nq * d * sizeof(float), hipMemcpyHostToDevice, stream));
    dq = ix->qbuf;
    if (lbD) {
      long long* li = (long long*)((unsigned char*)ix->qbuf + qb);
      float* ld = (float*)(li + nq);
      EIOKU_HIP_CHECK(hipMemcpyAsync(li, lbI, (size_t)nq * sizeof(long long), hipMemcpyHostToDevice, stream));
      EIOKU_HIP_CHECK(hipMemcpyAsync(ld, lbD, (size_t)nq * sizeof(float), hipMemcpyHostToDevice, stream));
      dlbD = ld;
      dlbI = li;
    }
  }
  EIOKU_REQUIRE(((uintptr_t)dq & 15) == 0, "queries must be 16-byte aligned");
  rc = grow(&ix->qnorm, &ix->qncap, (size_t)nq * sizeof(float));
  if (rc) return rc;
  rc = norms_for(dq, nq, d, ix->qnorm, stream);
  if (rc) return rc;
  float* dD = D;
  long long* dI = (long long*)I;
  if (mem == EIOKU_MEM_HOST) {
    rc = grow(&ix->dout, &ix->dcap, (size_t)nq * k * sizeof(float));
    if (rc) return rc;
    rc = grow(&ix->iout, &ix->icap, (size_t)nq * k * sizeof(long long));
    if (rc) return rc;
    dD = ix->dout;
    dI = ix->iout;
  }
  const bool scan = ix->scan_mode != 0 && !dlbD && nq >= ix->scan_min_nq && k <= 32 && ix->n >= ix->scan_min_rows &&
                    (d == 128 || d == 256 || d == 384);
  if (scan) {
    // groups of <= 1024 queries: their planes (<= 1.5 MB) stay in every XCD's L2 while the rows stream past
    for (int q0 = 0; q0 < nq; q0 += 1024) {
      const int g = nq - q0 < 1024 ? nq - q0 : 1024;
      if (q0) {
        rc = norms_for(dq + (size_t)q0 * d, g, d, ix->qnorm, stream);
        if (rc) return rc;
      }
      rc = scan_search(ix, dq + (size_t)q0 * d, g, k, dD + (size_t)q0 * k, dI + (size_t)q0 * k, stream);
      if (rc) return rc;
    }
  } else {
    rc = legacy_search(ix, dq, nq, k, dlbD, dlbI, dD, dI, nullptr, 0, 0, true, stream);
    if (rc) return rc;
  }
  if (mem == EIOKU_MEM_HOST) {
    EIOKU_HIP_CHECK(hipMemcpyAsync(D, dD, (size_t)nq * k * sizeof(float), hipMemcpyDeviceToHost, stream));
    EIOKU_HIP_CHECK(hipMemcpyAsync(I, dI, (size_t)nq * k * sizeof(long long), hipMemcpyDeviceToHost, stream));
    EIOKU_HIP_CHECK(hipStreamSynchronize(stream));
  }
  return EIOKU_OK;
}

}  // namespace

extern "C" {

int eioku_index_search(eioku_index* ix, const float* q, int nq, int k, float* D, int64_t* I, int mem,
                       void* stream_) {
  return search_impl(ix, q, nq, k, nullptr, nullptr, D, I, mem, stream_);
}

int eioku_index_search_after(eioku_index* ix, const float* q, int nq, int k, const float* after_D,
                             const int64_t* after_I, float* D, int64_t* I, int mem, void* stream_) {
  EIOKU_REQUIRE(after_D && after_I, "NULL lower bound");
  return search_impl(ix, q, nq, k, after_D, after_I, D, I, mem, stream_);
}

int eioku_index_set_param(eioku_index* ix, const char* name, long long value) {
  EIOKU_REQUIRE(ix && name, "bad argument");
  if (!strcmp(name, "scan_mode")) {
    EIOKU_REQUIRE(value == 0 || value == 1, "scan_mode is 0 (register-tile kernels only) or 1 (scan path for wide searches)");
    ix->scan_mode = (int)value;
  } else if (!strcmp(name, "scan_cap")) {
    EIOKU_REQUIRE(value >= 16 && value <= 16384, "scan_cap must be in [16, 16384]");
    ix->scan_cap = (int)value;
  } else if (!strcmp(name, "scan_min_rows")) {
    EIOKU_REQUIRE(value >= 4096, "scan_min_rows must be >= 4096");
    ix->scan_min_rows = value;
  } else if (!strcmp(name, "scan_sample")) {
    EIOKU_REQUIRE(value >= 0, "scan_sample must be >= 0");
    ix->scan_sample = value;
  } else if (!strcmp(name, "scan_min_nq")) {
    EIOKU_REQUIRE(value >= 1, "scan_min_nq must be >= 1");
    ix->scan_min_nq = (int)value;
  } else if (!strcmp(name, "scan_prescan")) {
    EIOKU_REQUIRE(value == 0 || (value >= 2 && value <= 1024), "scan_prescan is 0 (off) or a tile stride in [2, 1024]");
    ix->scan_prescan = (int)value;
  } else if (!strcmp(name, "scan_rt")) {
    EIOKU_REQUIRE(value == 1 || value == 2, "scan_rt must be 1 or 2");
    ix->scan_rt = (int)value;
  } else {
    set_error("unknown index parameter '%s'", name);
    return EIOKU_EINVAL;
  }
  return EIOKU_OK;
}

// Merge `nlists` per-shard results (each [nq][k] ascending, ids already global, -1 = empty) into the
// global top-k.  Device pointers; d_lists / i_lists are [nlists][nq][k] contiguous (the layout an
// all-gather of per-rank (D, I) produces).
int eioku_topk_merge_ex(const float* d_lists, const int64_t* i_lists, int nlists, int nq, int k_in, int k, float* D,
                        int64_t* I, void* stream_) {
  EIOKU_REQUIRE_INIT();
  EIOKU_REQUIRE(nlists >= 1 && nq >= 0 && k >= 1 && k <= 32 && k_in >= k, "bad argument");
  if (nq == 0) return EIOKU_OK;
  EIOKU_REQUIRE(d_lists && i_lists && D && I, "NULL buffer");
  hipStream_t stream = (hipStream_t)stream_;
  if (k <= 16)
    hipLaunchKernelGGL((k_topk_merge<16>), dim3(nq), dim3(64), 0, stream, d_lists, (const int*)nullptr,
                       (const long long*)i_lists, nlists, nq, k_in, 0, (const long long*)nullptr, 0ll, k, D, (long long*)I);
  else
    hipLaunchKernelGGL((k_topk_merge<32>), dim3(nq), dim3(64), 0, stream, d_lists, (const int*)nullptr,
                       (const long long*)i_lists, nlists, nq, k_in, 0, (const long long*)nullptr, 0ll, k, D, (long long*)I);
  EIOKU_LAUNCH_CHECK();
  return EIOKU_OK;
}

int eioku_topk_merge(const float* d_lists, const int64_t* i_lists, int nlists, int nq, int k, float* D,
                     int64_t* I, void* stream_) {
  return eioku_topk_merge_ex(d_lists, i_lists, nlists, nq, k, k, D, I, stream_);
}

}  // extern "C"
