// Retrieval blend of VC.vc (rvc/infer/pipeline.py:239-250): index.search(k=8) over the stored
// feature matrix, weights (1/d)^2 normalised, weighted sum of the neighbours, lerp by index_rate.
// faiss-cpu is not vendored in the reference; this implements exact squared-L2 search
// (IndexFlatL2 semantics: d = |q|^2 + |b|^2 - 2 q.b with the dot products on the fp32 MFMA GEMM).
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
  ix->mat = make_conv(c, big_npy, nullptr, (int)n, dim, 1, 1, false);   // exact fp32 products: neighbour ids are bit-exact
  ix->rows = c.slab.upload(big_npy, (size_t)n * dim);
  std::vector<float> norms((size_t)n);
  for (int64_t i = 0; i < n; ++i) {
    float s = 0.f;
    for (int d = 0; d < dim; ++d) s += big_npy[i * dim + d] * big_npy[i * dim + d];
    norms[(size_t)i] = s;
  }
  ix->norms = c.slab.upload(norms);
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

size_t index_arena_bytes(const IndexData& ix, int T) {
  return ((size_t)ix.n * T + (size_t)ix.nlist * T + (size_t)T * (SPLITS * TOPK * 2 + 72)) * sizeof(float) + (1 << 20);
}

namespace {

__global__ void qnorm_kernel(const float* feats, float* qn, int dim, int T) {
  const int t = blockIdx.x * 256 + threadIdx.x;
  if (t >= T) return;
  float s = 0.f;
  for (int d = 0; d < dim; ++d) {
    const float v = feats[(long)d * T + t];
    s += v * v;
  }
  qn[t] = s;
}

__device__ __forceinline__ void topk_insert(float (&bd)[TOPK], int (&bi)[TOPK], float d, int id) {
  if (d >= bd[TOPK - 1]) return;
  bd[TOPK - 1] = d;
  bi[TOPK - 1] = id;
#pragma unroll
  for (int k = TOPK - 1; k > 0; --k) {
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

// dots (N, T): block = 64 queries x 4 row slices, grid.y = row splits.  Partial top-8 per (split, query).
__global__ __launch_bounds__(256) void topk_partial_kernel(const float* __restrict__ dots,
                                                           const float* __restrict__ bn,
                                                           const float* __restrict__ qn, float* pd, int* pi,
                                                           long N, int T, int splits,
                                                           const int* __restrict__ assign,
                                                           const int* __restrict__ qlist) {
  __shared__ float sd[4][64][TOPK];
  __shared__ int si[4][64][TOPK];
  const int tx = threadIdx.x & 63, part = threadIdx.x >> 6;
  const int t = blockIdx.x * 64 + tx;
  const long rows_per = (N + splits - 1) / splits;
  const long r0 = blockIdx.y * rows_per, r1 = min(N, r0 + rows_per);
  float bd[TOPK];
  int bi[TOPK];
#pragma unroll
  for (int k = 0; k < TOPK; ++k) {
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
  for (int k = 0; k < TOPK; ++k) {
    sd[part][tx][k] = bd[k];
    si[part][tx][k] = bi[k];
  }
  __syncthreads();
  if (part == 0 && t < T) {
    for (int p = 1; p < 4; ++p)
#pragma unroll
      for (int k = 0; k < TOPK; ++k) topk_insert(bd, bi, sd[p][tx][k], si[p][tx][k]);
    for (int k = 0; k < TOPK; ++k) {
      pd[((long)blockIdx.y * T + t) * TOPK + k] = bd[k];
      pi[((long)blockIdx.y * T + t) * TOPK + k] = bi[k];
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

// merge the per-split candidates, then blend: one block per query
__global__ __launch_bounds__(256) void merge_blend_kernel(const float* pd, const int* pi, int splits,
                                                          const float* __restrict__ rows, long N, float* feats, int dim,
                                                          int T, float rate, float one_minus, long long* ids_out,
                                                          float* dist_out) {
  __shared__ float fd[TOPK];
  __shared__ int fi[TOPK];
  __shared__ float fw[TOPK];
  const int t = blockIdx.x;
  if (threadIdx.x == 0) {
    float bd[TOPK];
    int bi[TOPK];
    for (int k = 0; k < TOPK; ++k) {
      bd[k] = INFINITY;
      bi[k] = -1;
    }
    for (int s = 0; s < splits; ++s)
      for (int k = 0; k < TOPK; ++k) {
        const float d = pd[((long)s * T + t) * TOPK + k];
        const int id = pi[((long)s * T + t) * TOPK + k];
        if (id < 0) continue;
        // stable w.r.t. row id on exact ties (smaller id first), like a sequential scan
        if (d < bd[TOPK - 1] || (d == bd[TOPK - 1] && id < bi[TOPK - 1])) {
          int pos = TOPK - 1;
          while (pos > 0 && (d < bd[pos - 1] || (d == bd[pos - 1] && id < bi[pos - 1]))) {
            bd[pos] = bd[pos - 1];
            bi[pos] = bi[pos - 1];
            --pos;
          }
          bd[pos] = d;
          bi[pos] = id;
        }
      }
    float w[TOPK];
    for (int k = 0; k < TOPK; ++k) {
      // an inverted list with fewer than 8 vectors: faiss pads with id -1 / distance inf (3.4e38 in 1.7.3); the
      // reference then reads big_npy[-1] (numpy: the LAST row) with weight (1/inf)^2 = 0  (pipeline.py:243-245)
      if (bi[k] < 0) bi[k] = (int)(N - 1);
      const float inv = 1.f / bd[k];
      w[k] = inv * inv;                                  // np.square(1 / score)
    }
    const float ws = ((w[0] + w[1]) + (w[2] + w[3])) + ((w[4] + w[5]) + (w[6] + w[7]));  // numpy pairwise
    for (int k = 0; k < TOPK; ++k) {
      fd[k] = bd[k];
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
  Arena& A = c.arena;
  float* dots = A.alloc<float>((size_t)ix.n * T);
  {
    ConvArgs a = conv1d_args(ix.mat, feats_ct, dots, 1, T, T);
    c.conv_on(a, s);
  }
  float* qn = A.alloc<float>((size_t)T);
  hipLaunchKernelGGL(qnorm_kernel, dim3(cdiv(T, 256)), dim3(256), 0, s, feats_ct, qn, ix.dim, T);
  int* qlist = nullptr;
  if (ix.nlist > 0) {
    float* cdots = A.alloc<float>((size_t)ix.nlist * T);
    ConvArgs a = conv1d_args(ix.cent, feats_ct, cdots, 1, T, T);
    c.conv_on(a, s);
    qlist = A.alloc<int>((size_t)T);
    hipLaunchKernelGGL(coarse_assign_kernel, dim3(cdiv(T, 256)), dim3(256), 0, s, cdots, ix.cent_norms, qn, qlist,
                       ix.nlist, T);
  }
  float* pd = A.alloc<float>((size_t)SPLITS * T * TOPK);
  int* pi = A.alloc<int>((size_t)SPLITS * T * TOPK);
  hipLaunchKernelGGL(topk_partial_kernel, dim3(cdiv(T, 64), SPLITS), dim3(256), 0, s, dots, ix.norms, qn, pd, pi,
                     (long)ix.n, T, SPLITS, ix.assign, qlist);
  hipLaunchKernelGGL(merge_blend_kernel, dim3(T), dim3(256), 0, s, pd, pi, SPLITS, ix.rows, (long)ix.n, feats_ct, ix.dim, T,
                     index_rate, (float)(1.0 - (double)index_rate), reinterpret_cast<long long*>(ids), dist);
  RVCX_HIP(hipGetLastError());
}

}  // namespace rvcx
