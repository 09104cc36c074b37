// Retrieval blend of VC.vc (rvc/infer/pipeline.py:239-250): index.search(k=8) over the stored
// feature matrix, weights (1/d)^2 normalised, weighted sum of the neighbours, lerp by index_rate.
// faiss-cpu is not vendored in the reference; this implements exact squared-L2 search
// (IndexFlatL2 semantics: d = |q|^2 + |b|^2 - 2 q.b).
//
// Round 3: the N x T dot products (161 GFLOP per 30 s clip at N = 65 536: 2.2 ms of fp32 MFMA) are only a PRE-FILTER and
// run on the split-fp16 kernels (3x the rate, error <= 2^-20 |q||b| per dot).  Per query the 16 best rows by those
// approximate distances are re-scored with an exact fp32 dot product, the best 8 of them are the answer -- and the
// answer is CERTIFIED: every row outside the 16 has an approximate distance >= the 16th one, so if the 8th exact
// distance is below that minus the error bound no other row can belong to the top 8.  A query that fails the test
// (near-ties, duplicated rows) is searched again exhaustively with the same exact dot routine.  The ids are therefore
// those of an exact search whatever the pre-filter's arithmetic (tests: bit-exact against float64 brute force).
#include <algorithm>
#include <cstdio>
#include <cstdlib>

#include "gemm.h"
#include "h3_device.h"
#include "models.h"
#include "ops.h"

namespace rvcx {

constexpr int TOPK = 8;

std::unique_ptr<IndexData> index_load(Ctx& c, const float* big_npy, int64_t n, int dim, const float* centroids,
                                      int nlist, const int32_t* assign) {
  auto ix = std::make_unique<IndexData>();
  RegionScope scope(c, *ix->region);
  ix->n = n;
  ix->dim = dim;
  ix->mat = make_conv(c, big_npy, nullptr, (int)n, dim, 1, 1, true);    // + split-fp16 image: the pre-filter (see above)
  ix->rows = c.slab.upload(big_npy, (size_t)n * dim);
  std::vector<float> norms((size_t)n);
  for (int64_t i = 0; i < n; ++i) {
    float s = 0.f;
    for (int d = 0; d < dim; ++d) s += big_npy[i * dim + d] * big_npy[i * dim + d];
    norms[(size_t)i] = s;
  }
  ix->norms = c.slab.upload(norms);
  ix->max_norm = norms.empty() ? 0.f : *std::max_element(norms.begin(), norms.end());
  {
    const float zero = 0.f;
    ix->exhaustive = reinterpret_cast<int*>(c.slab.upload(&zero, 1));   // counter of exhaustively searched queries
  }
  if (centroids && nlist > 0 && assign) {
    ix->nlist = nlist;
    ix->cent = make_conv(c, centroids, nullptr, nlist, dim, 1, 1, false);
    std::vector<float> cn((size_t)nlist);
    for (int i = 0; i < nlist; ++i) {
      float s = 0.f;
      for (int d = 0; d < dim; ++d) s += centroids[(size_t)i * dim + d] * centroids[(size_t)i * dim + d];
      cn[(size_t)i] = s;
    }
    ix->cent_norms = c.slab.upload(cn);
    for (int64_t i = 0; i < n; ++i) RVCX_CHECK(assign[i] >= 0 && assign[i] < nlist, "index: list id out of range");
    static_assert(sizeof(int32_t) == sizeof(float), "list ids travel through the float upload path");
    ix->assign = reinterpret_cast<const int*>(c.slab.upload(reinterpret_cast<const float*>(assign), (size_t)n));
  }
  ix->region->seal();
  return ix;
}

constexpr int SPLITS = 32;
constexpr int NCAND = 16;        // rows re-scored exactly per query

size_t index_arena_bytes(const IndexData& ix, int T) {
  return ((size_t)ix.n * T + (size_t)ix.nlist * T + (size_t)T * (SPLITS * NCAND * 2 + 96 + ix.dim)) * sizeof(float) + (1 << 20);
}

namespace {

// |q|^2 per query: 64 queries x 16 dimension slices per workgroup, slices summed in a fixed order (one thread per query
// walking all 768 rows was a 0.47 ms chain of dependent loads per clip in the C3 trace of round 3)
__global__ __launch_bounds__(1024) void qnorm_kernel(const float* feats, float* qn, int dim, int T) {
  __shared__ float part[16][64];
  const int tx = threadIdx.x & 63, sl = threadIdx.x >> 6;
  const int t = blockIdx.x * 64 + tx;
  const int per = (dim + 15) / 16;
  float s = 0.f;
  if (t < T)
    for (int d = sl * per; d < min(dim, (sl + 1) * per); ++d) {
      const float v = feats[(long)d * T + t];
      s += v * v;
    }
  part[sl][tx] = s;
  __syncthreads();
  if (sl == 0 && t < T) {
    float a = 0.f;
#pragma unroll
    for (int k = 0; k < 16; ++k) a += part[k][tx];
    qn[t] = a;
  }
}

template <int K>
__device__ __forceinline__ void topk_insert(float (&bd)[K], int (&bi)[K], float d, int id) {
  if (d >= bd[K - 1]) return;
  bd[K - 1] = d;
  bi[K - 1] = id;
#pragma unroll
  for (int k = K - 1; k > 0; --k) {
    if (bd[k] < bd[k - 1]) {
      const float td = bd[k];
      bd[k] = bd[k - 1];
      bd[k - 1] = td;
      const int ti = bi[k];
      bi[k] = bi[k - 1];
      bi[k - 1] = ti;
    }
  }
}

// dots (N, T): block = 64 queries x 4 row slices, grid.y = row splits.  Partial top-K per (split, query).
template <int K>
__global__ __launch_bounds__(256) void topk_partial_kernel(const float* __restrict__ dots,
                                                           const float* __restrict__ bn,
                                                           const float* __restrict__ qn, float* pd, int* pi,
                                                           long N, int T, int splits,
                                                           const int* __restrict__ assign,
                                                           const int* __restrict__ qlist) {
  __shared__ float sd[4][64][K];
  __shared__ int si[4][64][K];
  const int tx = threadIdx.x & 63, part = threadIdx.x >> 6;
  const int t = blockIdx.x * 64 + tx;
  const long rows_per = (N + splits - 1) / splits;
  const long r0 = blockIdx.y * rows_per, r1 = min(N, r0 + rows_per);
  float bd[K];
  int bi[K];
#pragma unroll
  for (int k = 0; k < K; ++k) {
    bd[k] = INFINITY;
    bi[k] = -1;
  }
  if (t < T) {
    const float q = qn[t];
    const int mylist = qlist ? qlist[t] : -1;
    for (long r = r0 + part; r < r1; r += 4) {
      if (assign && assign[r] != mylist) continue;     // nprobe = 1: only the query's own inverted list is scanned
      float d = q + bn[r] - 2.f * dots[r * T + t];
      d = fmaxf(d, 0.f);
      topk_insert(bd, bi, d, (int)r);
    }
  }
#pragma unroll
  for (int k = 0; k < K; ++k) {
    sd[part][tx][k] = bd[k];
    si[part][tx][k] = bi[k];
  }
  __syncthreads();
  if (part == 0 && t < T) {
    for (int p = 1; p < 4; ++p)
#pragma unroll
      for (int k = 0; k < K; ++k) topk_insert(bd, bi, sd[p][tx][k], si[p][tx][k]);
    for (int k = 0; k < K; ++k) {
      pd[((long)blockIdx.y * T + t) * K + k] = bd[k];
      pi[((long)blockIdx.y * T + t) * K + k] = bi[k];
    }
  }
}

// coarse quantiser of an IVF index: nearest centroid per query (IndexFlatL2.search(k = 1) on the centroids; first
// minimum on ties)
__global__ void coarse_assign_kernel(const float* __restrict__ cdots, const float* __restrict__ cn,
                                     const float* __restrict__ qn, int* qlist, int nlist, int T) {
  const int t = blockIdx.x * 256 + threadIdx.x;
  if (t >= T) return;
  const float q = qn[t];
  float best = INFINITY;
  int bi = 0;
  for (int r = 0; r < nlist; ++r) {
    const float d = fmaxf(q + cn[r] - 2.f * cdots[(long)r * T + t], 0.f);
    if (d < best) {
      best = d;
      bi = r;
    }
  }
  qlist[t] = bi;
}

// (d, id) order: smaller distance first, smaller row id on exact ties -- what a stable sort of a sequential scan gives
template <int K>
__device__ __forceinline__ void ordered_insert(float (&bd)[K], int (&bi)[K], float d, int id) {
  if (!(d < bd[K - 1] || (d == bd[K - 1] && id < bi[K - 1]))) return;
  int pos = K - 1;
  while (pos > 0 && (d < bd[pos - 1] || (d == bd[pos - 1] && id < bi[pos - 1]))) {
    bd[pos] = bd[pos - 1];
    bi[pos] = bi[pos - 1];
    --pos;
  }
  bd[pos] = d;
  bi[pos] = id;
}

// exact fp32 dot product of the query (LDS) with a stored row by 16 lanes: lane l takes dimensions l, l + 16, .. in
// order, the partial sums meet in a fixed tree.  Both exact paths below use it, so a (query, row) pair has ONE distance.
__device__ __forceinline__ float dot16(const float* qs, const float* __restrict__ row, int dim, int l) {
  float a = 0.f;
  for (int d = l; d < dim; d += 16) a = fmaf(qs[d], row[d], a);
  a += __shfl_xor(a, 8);
  a += __shfl_xor(a, 4);
  a += __shfl_xor(a, 2);
  a += __shfl_xor(a, 1);
  return a;
}

constexpr int kMaxDim = 1024;

// One block per query: merge the per-split candidate lists (approximate distances), re-score the NCAND best exactly,
// take the best 8 and certify them (see the file header).  fd / fi: (T, 8); flag[t] = 1 when the query needs the
// exhaustive search.
__global__ __launch_bounds__(256) void rescore_kernel(const float* pd, const int* pi, int splits,
                                                      const float* __restrict__ rows, const float* __restrict__ bn,
                                                      const float* __restrict__ qn, const float* feats, int dim, int T,
                                                      float max_norm, float* fd, int* fi, int* flag) {
  __shared__ float qs[kMaxDim];
  __shared__ float cd[NCAND], ed[NCAND];
  __shared__ int ci[NCAND];
  __shared__ int nvalid;
  const int t = blockIdx.x, tid = threadIdx.x;
  for (int c = tid; c < dim; c += 256) qs[c] = feats[(long)c * T + t];
  if (tid == 0) {
    float bd[NCAND];
    int bi[NCAND];
    for (int k = 0; k < NCAND; ++k) {
      bd[k] = INFINITY;
      bi[k] = -1;
    }
    for (int s = 0; s < splits; ++s)
      for (int k = 0; k < NCAND; ++k) {
        const int id = pi[((long)s * T + t) * NCAND + k];
        if (id >= 0) ordered_insert(bd, bi, pd[((long)s * T + t) * NCAND + k], id);
      }
    int nv = 0;
    for (int k = 0; k < NCAND; ++k) {
      cd[k] = bd[k];
      ci[k] = bi[k];
      nv += bi[k] >= 0;
    }
    nvalid = nv;
  }
  __syncthreads();
  const int g = tid >> 4, l = tid & 15;
  if (g < nvalid) {
    const int r = ci[g];
    const float dot = dot16(qs, rows + (long)r * dim, dim, l);
    if (l == 0) ed[g] = fmaxf(qn[t] + bn[r] - 2.f * dot, 0.f);
  }
  __syncthreads();
  if (tid == 0) {
    float bd[TOPK];
    int bi[TOPK];
    for (int k = 0; k < TOPK; ++k) {
      bd[k] = INFINITY;
      bi[k] = -1;
    }
    for (int k = 0; k < nvalid; ++k) ordered_insert(bd, bi, ed[k], ci[k]);
    // error of an approximate distance: the split-fp16 dot (<= 2^-20 |q||b|), the fp32 accumulation of 768 terms and the
    // three fp32 additions -- bounded with a wide margin by 2^-16 (|q| + |b|max)^2
    const float sq = sqrtf(qn[t]) + sqrtf(max_norm);
    const float E = 1.52587890625e-5f * sq * sq;
    const bool certified = nvalid < NCAND || bd[TOPK - 1] < cd[NCAND - 1] - E;
    flag[t] = certified ? 0 : 1;
    for (int k = 0; k < TOPK; ++k) {
      fd[(long)t * TOPK + k] = bd[k];
      fi[(long)t * TOPK + k] = bi[k];
    }
  }
}

// The exact search of the queries rescore_kernel could not certify (ties, duplicated rows -- e.g. the silence frames a
// real RVC index repeats; one block per query, returns at once for the others).  Round 4: not every row is
// re-scored -- 200 MB of reads per query at 65 536 x 768 -- but only those that can still be among the best 8: a row whose
// exact distance is <= the 8th exact distance found so far has an APPROXIMATE distance <= that + E (E: the error bound
// rescore_kernel certifies with), and the approximate distances of all rows are still in memory (`dots`).  The scan of
// that column costs N x 4 bytes per query (768x less); the survivors are re-scored with the same dot routine and ordered
// by (distance, row id), so the result is what the exhaustive scan gives.  More than kMaxCand survivors: the full scan.
constexpr int kMaxCand = 4096;
__global__ __launch_bounds__(256) void exhaustive_kernel(const int* flag, const float* __restrict__ rows,
                                                         const float* __restrict__ bn, const float* __restrict__ qn,
                                                         const float* feats, int dim, int T, long N,
                                                         const int* __restrict__ assign, const int* __restrict__ qlist,
                                                         float* fd, int* fi, int* counter, const float* __restrict__ dots,
                                                         float max_norm) {
  const int t = blockIdx.x, tid = threadIdx.x;
  if (!flag[t]) return;
  if (tid == 0 && counter) atomicAdd(counter, 1);
  __shared__ float qs[kMaxDim];
  __shared__ float sd[16][TOPK];
  __shared__ int si[16][TOPK];
  __shared__ int cand[kMaxCand];
  __shared__ int ncand;
  for (int c = tid; c < dim; c += 256) qs[c] = feats[(long)c * T + t];
  if (tid == 0) ncand = 0;
  __syncthreads();
  const int g = tid >> 4, l = tid & 15;
  const int mylist = qlist ? qlist[t] : -1;
  float bd[TOPK];
  int bi[TOPK];
  for (int k = 0; k < TOPK; ++k) {
    bd[k] = INFINITY;
    bi[k] = -1;
  }
  const float q = qn[t];
  bool full = dots == nullptr;
  if (!full) {
    const float sq = sqrtf(q) + sqrtf(max_norm);
    const float thr = fd[(long)t * TOPK + TOPK - 1] + 1.52587890625e-5f * sq * sq;     // 8th exact distance so far + E
    for (long r = tid; r < N; r += 256) {
      if (assign && assign[r] != mylist) continue;
      const float ap = fmaxf(q + bn[r] - 2.f * dots[r * T + t], 0.f);
      if (ap <= thr) {
        const int k = atomicAdd(&ncand, 1);
        if (k < kMaxCand) cand[k] = (int)r;
      }
    }
    __syncthreads();
    full = ncand > kMaxCand;
  }
  if (full) {
    for (long r = g; r < N; r += 16) {
      if (assign && assign[r] != mylist) continue;
      const float dot = dot16(qs, rows + r * dim, dim, l);
      ordered_insert(bd, bi, fmaxf(q + bn[r] - 2.f * dot, 0.f), (int)r);
    }
  } else {
    for (int c = g; c < ncand; c += 16) {          // the order candidates were collected in does not matter: (d, id) is total
      const long r = cand[c];
      const float dot = dot16(qs, rows + r * dim, dim, l);
      ordered_insert(bd, bi, fmaxf(q + bn[r] - 2.f * dot, 0.f), (int)r);
    }
  }
  if (l == 0)
    for (int k = 0; k < TOPK; ++k) {
      sd[g][k] = bd[k];
      si[g][k] = bi[k];
    }
  __syncthreads();
  if (tid == 0) {
    for (int p = 1; p < 16; ++p)
      for (int k = 0; k < TOPK; ++k)
        if (si[p][k] >= 0) ordered_insert(bd, bi, sd[p][k], si[p][k]);
    for (int k = 0; k < TOPK; ++k) {
      fd[(long)t * TOPK + k] = bd[k];
      fi[(long)t * TOPK + k] = bi[k];
    }
  }
}

// weights (1/d)^2 normalised, weighted sum of the neighbours, lerp by index_rate (pipeline.py:243-250): one block per query
__global__ __launch_bounds__(256) void blend_kernel(const float* fd_g, const int* fi_g, const float* __restrict__ rows, long N,
                                                    float* feats, int dim, int T, float rate, float one_minus,
                                                    long long* ids_out, float* dist_out) {
  __shared__ int fi[TOPK];
  __shared__ float fw[TOPK];
  const int t = blockIdx.x;
  if (threadIdx.x == 0) {
    float bd[TOPK], w[TOPK];
    int bi[TOPK];
    for (int k = 0; k < TOPK; ++k) {
      bd[k] = fd_g[(long)t * TOPK + k];
      bi[k] = fi_g[(long)t * TOPK + k];
      // an inverted list with fewer than 8 vectors: faiss pads with id -1 / distance inf (3.4e38 in 1.7.3); the
      // reference then reads big_npy[-1] (numpy: the LAST row) with weight (1/inf)^2 = 0  (pipeline.py:243-245)
      if (bi[k] < 0) bi[k] = (int)(N - 1);
      const float inv = 1.f / bd[k];
      w[k] = inv * inv;                                  // np.square(1 / score)
    }
    const float ws = ((w[0] + w[1]) + (w[2] + w[3])) + ((w[4] + w[5]) + (w[6] + w[7]));  // numpy pairwise
    for (int k = 0; k < TOPK; ++k) {
      fi[k] = bi[k];
      fw[k] = w[k] / ws;
      if (ids_out) ids_out[(long)t * TOPK + k] = isinf(bd[k]) ? -1 : bi[k];
      if (dist_out) dist_out[(long)t * TOPK + k] = bd[k];
    }
  }
  __syncthreads();
  for (int c = threadIdx.x; c < dim; c += 256) {
    float acc = rows[(long)fi[0] * dim + c] * fw[0];
#pragma unroll
    for (int k = 1; k < TOPK; ++k) acc = acc + rows[(long)fi[k] * dim + c] * fw[k];
    const long o = (long)c * T + t;
    feats[o] = acc * rate + one_minus * feats[o];
  }
}

}  // namespace

void index_blend(Ctx& c, const IndexData& ix, float* feats_ct, int T, float index_rate, int64_t* ids, float* dist,
                 hipStream_t s) {
  RVCX_CHECK(ix.n >= TOPK, "index: fewer than 8 stored vectors");
  RVCX_CHECK(ix.dim <= kMaxDim, "index: feature dimension above 1024");
  Arena& A = c.arena;
  float* dots = A.alloc<float>((size_t)ix.n * T);
  // RVCX_INDEX_PREFILTER=0: the dot products on the exact-fp32 MFMA conv, as before round 3 (A/B runs)
  static const bool prefilter = !getenv("RVCX_INDEX_PREFILTER") || atoi(getenv("RVCX_INDEX_PREFILTER")) != 0;
  if (prefilter && conv_h3_ok(ix.mat) && gemm_h3_enabled() && ix.dim % 16 == 0 && ix.n % 4 == 0 &&
      (long)ix.mat.cin_gp * ix.mat.cout_gp * 4 < kH3Oob) {
    // pre-filter on the time-major GEMM kernel (gemm.hip): queries as split rows, dots stored channel-first (N, T)
    float* xs = A.alloc<float>((size_t)T * ix.dim);
    launch_cf_to_tm(feats_ct, (long)ix.dim * T, nullptr, 0, xs, (long)ix.dim * 4, 1, ix.dim, T, c.dev_err, ix.mat.ovf_word,
                    ++c.launch_seq, s);
    GemmArgs g = gemm_args(ix.mat, T, T);
    g.xs = xs;
    g.ld_xs = (long)ix.dim * 4;
    g.y_cf = dots;
    g.cf_bs = (long)ix.n * T;
    c.gemm_on(g, s);
  } else {
    ConvArgs a = conv1d_args(ix.mat, feats_ct, dots, 1, T, T);
    if (!prefilter) a.w_h3 = nullptr;
    c.conv_on(a, s);
  }
  float* qn = A.alloc<float>((size_t)T);
  hipLaunchKernelGGL(qnorm_kernel, dim3(cdiv(T, 64)), dim3(1024), 0, s, feats_ct, qn, ix.dim, T);
  int* qlist = nullptr;
  if (ix.nlist > 0) {
    float* cdots = A.alloc<float>((size_t)ix.nlist * T);
    ConvArgs a = conv1d_args(ix.cent, feats_ct, cdots, 1, T, T);
    c.conv_on(a, s);
    qlist = A.alloc<int>((size_t)T);
    hipLaunchKernelGGL(coarse_assign_kernel, dim3(cdiv(T, 256)), dim3(256), 0, s, cdots, ix.cent_norms, qn, qlist,
                       ix.nlist, T);
  }
  float* pd = A.alloc<float>((size_t)SPLITS * T * NCAND);
  int* pi = A.alloc<int>((size_t)SPLITS * T * NCAND);
  float* fd = A.alloc<float>((size_t)T * TOPK);
  int* fi = A.alloc<int>((size_t)T * TOPK);
  int* flag = A.alloc<int>((size_t)T);
  hipLaunchKernelGGL(topk_partial_kernel<NCAND>, dim3(cdiv(T, 64), SPLITS), dim3(256), 0, s, dots, ix.norms, qn, pd, pi,
                     (long)ix.n, T, SPLITS, ix.assign, qlist);
  hipLaunchKernelGGL(rescore_kernel, dim3(T), dim3(256), 0, s, pd, pi, SPLITS, ix.rows, ix.norms, qn, feats_ct, ix.dim, T,
                     ix.max_norm, fd, fi, flag);
  hipLaunchKernelGGL(exhaustive_kernel, dim3(T), dim3(256), 0, s, flag, ix.rows, ix.norms, qn, feats_ct, ix.dim, T,
                     (long)ix.n, ix.assign, qlist, fd, fi, ix.exhaustive, dots, ix.max_norm);
  hipLaunchKernelGGL(blend_kernel, dim3(T), dim3(256), 0, s, fd, fi, ix.rows, (long)ix.n, feats_ct, ix.dim, T, index_rate,
                     (float)(1.0 - (double)index_rate), reinterpret_cast<long long*>(ids), dist);
  RVCX_HIP(hipGetLastError());
}

}  // namespace rvcx
