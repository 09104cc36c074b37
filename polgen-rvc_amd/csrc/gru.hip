// Bidirectional GRU recurrence (RMVPE BiGRU(384 -> 2x256), rvc/lib/predictors/RMVPE.py:125-137).
// The input projections W_ih x + b_ih for every frame are one MFMA GEMM done by the conv kernel
// (transposed store, (B,T,6H)); what is left is the serial part h_t = f(gi_t, W_hh h_{t-1}).
//
// One workgroup per (direction, batch item): 3H threads, thread j owns gate row j.  W_hh is
// stored transposed (H x 3H) so the per-step read of column block k is coalesced across the
// threads; it streams from L2 every step (786 KB fp32 does not fit one CU's registers + LDS).
// Gate order r, z, n (torch.nn.GRU).
#include "ops.h"

namespace rvcx {

template <int H>
__global__ __launch_bounds__(3 * H) void bigru_kernel(const float* __restrict__ gi,
                                                      const float* __restrict__ whh_t,  // (2, H, 3H)
                                                      const float* __restrict__ bhh,    // (2, 3H)
                                                      float* __restrict__ y, int T) {
  __shared__ float hs[H];
  __shared__ float gh[3 * H];
  const int dir = blockIdx.x, b = blockIdx.y;
  const int j = threadIdx.x;
  const float* W = whh_t + (long)dir * H * 3 * H;
  const float bj = bhh[dir * 3 * H + j];
  const float* gib = gi + (long)b * T * 6 * H + dir * 3 * H;
  float* yb = y + ((long)b * 2 + dir) * H * T;
  if (j < H) hs[j] = 0.f;
  __syncthreads();
  for (int step = 0; step < T; ++step) {
    const int t = dir == 0 ? step : T - 1 - step;
    float acc0 = 0.f, acc1 = 0.f, acc2 = 0.f, acc3 = 0.f;
#pragma unroll 8
    for (int k = 0; k < H; k += 4) {
      acc0 = fmaf(W[(long)(k + 0) * 3 * H + j], hs[k + 0], acc0);
      acc1 = fmaf(W[(long)(k + 1) * 3 * H + j], hs[k + 1], acc1);
      acc2 = fmaf(W[(long)(k + 2) * 3 * H + j], hs[k + 2], acc2);
      acc3 = fmaf(W[(long)(k + 3) * 3 * H + j], hs[k + 3], acc3);
    }
    gh[j] = ((acc0 + acc1) + (acc2 + acc3)) + bj;
    __syncthreads();
    if (j < H) {
      const float* g = gib + (long)t * 6 * H;
      const float r = 1.f / (1.f + expf(-(g[j] + gh[j])));
      const float z = 1.f / (1.f + expf(-(g[H + j] + gh[H + j])));
      const float n = tanhf(g[2 * H + j] + r * gh[2 * H + j]);
      const float hn = (1.f - z) * n + z * hs[j];
      hs[j] = hn;
      yb[(long)j * T + t] = hn;
    }
    __syncthreads();
  }
}

void launch_bigru(const float* gi, const float* whh_t, const float* bhh, float* y, int B, int T, int H,
                  hipStream_t stream) {
  RVCX_CHECK(H == 256, "bigru: hidden size must be 256 (RMVPE)");
  hipLaunchKernelGGL(bigru_kernel<256>, dim3(2, B), dim3(768), 0, stream, gi, whh_t, bhh, y, T);
  RVCX_HIP(hipGetLastError());
}

}  // namespace rvcx
