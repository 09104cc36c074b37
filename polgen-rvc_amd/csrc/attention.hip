// Flash-style self-attention on the fp32 matrix cores for channel-first (B, H*D, T) tensors,
// with the optional windowed relative-position terms of the RVC TextEncoder
// (rvc/lib/algorithm/attentions.py:63-113) -- and, without them, the HuBERT encoder attention.
//
// Per wave: 32 queries.  S^T = K^T Q is computed "swapped" (rows = keys, cols = queries) so a
// lane owns one query column: the row-reduction of the online softmax is 16 in-register ops
// plus one cross-half shuffle, and the softmaxed registers are *already* the B fragment of
// O^T += V P^T (MFMA k-slot s of lane-half h <-> key (s&3)+8(s>>2)+4h; V's A fragment is read
// from LDS with the same permutation, so P never moves between lanes).
#include <cstdlib>

#include "conv.h"
#include "conv_device.h"
#include "ops.h"

namespace rvcx {

constexpr int KT = 32;     // keys per tile
constexpr int VROW = 33;   // odd pitch: conflict-free strided A-fragment reads of V

template <int DT>  // D <= 32*DT
__global__ __launch_bounds__(256) void attn_kernel(const float* __restrict__ q, const float* __restrict__ k,
                                                   const float* __restrict__ v, float* __restrict__ out,
                                                   float* __restrict__ m_out, float* __restrict__ l_out,
                                                   const float* __restrict__ relq, int H, int D, int T,
                                                   int ld, long in_bs, long out_bs, float scale, int window,
                                                   const int* lens, int nsplit, float* opart) {
  __shared__ float Ks[32 * DT * KT];
  __shared__ float Vs[32 * DT * VROW];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int i = lane & 31, h = lane >> 5;
  const int b = blockIdx.z, hd = blockIdx.y;
  const int qblk = blockIdx.x / nsplit, split = blockIdx.x - qblk * nsplit;
  const int q0 = qblk * 128 + wave * 32;
  const int len = lens ? lens[b] : T;
  const long base = (long)b * in_bs + (long)hd * D * ld;
  const long obase = (long)b * out_bs + (long)hd * D * ld;
  const float* qb = q + base;
  const float* kb = k + base;
  const float* vb = v + base;
  const int nsteps = D / 2;

  // Q fragment (B operand of S^T): lane (i,h), step s -> Q[2s+h][q0+i] * scale
  float qreg[16 * DT];
  const int qi = q0 + i;
#pragma unroll
  for (int s = 0; s < 16 * DT; ++s) {
    float val = 0.f;
    if (s < nsteps && qi < T) val = qb[(long)(2 * s + h) * ld + qi] * scale;
    qreg[s] = val;
  }

  f32x16 acc_o[DT];
#pragma unroll
  for (int dt = 0; dt < DT; ++dt)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc_o[dt][r] = 0.f;
  float m_run = -INFINITY, l_run = 0.f;
  const int nrel = 2 * window + 1;
  const float* relrow = relq ? relq + (((long)b * H + hd) * T + min(qi, T - 1)) * nrel : nullptr;

  // split-KV: this workgroup covers keys [klo, khi) (whole 32-key tiles); partial (O, m, l) are merged later
  const int tiles = (len + KT - 1) / KT;
  const int klo = (int)((long)tiles * split / nsplit) * KT;
  const int khi = min(len, (int)((long)tiles * (split + 1) / nsplit) * KT);
  for (int k0 = klo; k0 < khi; k0 += KT) {
    __syncthreads();
    // ---- stage K [D][32] and V [D][33] tiles (zero beyond D / len)
    for (int idx = tid; idx < 32 * DT * KT; idx += 256) {
      const int d = idx / KT, kk = idx % KT;
      const int key = k0 + kk;
      float kv = 0.f, vv = 0.f;
      if (d < D && key < len) {
        kv = kb[(long)d * ld + key];
        vv = vb[(long)d * ld + key];
      }
      Ks[d * KT + kk] = kv;
      Vs[d * VROW + kk] = vv;
    }
    __syncthreads();
    // ---- S^T tile: rows = keys, cols = queries
    f32x16 sacc;
#pragma unroll
    for (int r = 0; r < 16; ++r) sacc[r] = 0.f;
#pragma unroll
    for (int s = 0; s < 16 * DT; ++s) {
      if (s < nsteps) {
        const float a = Ks[(2 * s + h) * KT + i];
        sacc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, qreg[s], sacc, 0, 0, 0);
      }
    }
    // ---- relative-position logits on the diagonal band, key masking
    const bool band = relq && (k0 <= q0 + 31 + window) && (k0 + KT - 1 >= q0 - window);
    float mloc = -INFINITY;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int key = k0 + (r & 3) + 8 * (r >> 2) + 4 * h;
      float sv = sacc[r];
      if (band) {
        const int rel = key - qi + window;
        if (rel >= 0 && rel < nrel && qi < T) sv += relrow[rel];
      }
      if (key >= len) sv = -INFINITY;
      sacc[r] = sv;
      mloc = fmaxf(mloc, sv);
    }
    const float mtile = fmaxf(mloc, __shfl_xor(mloc, 32));
    const float m_new = fmaxf(m_run, mtile);
    const float alpha = (m_run == -INFINITY) ? 0.f : expf(m_run - m_new);
    float lloc = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float p = (sacc[r] == -INFINITY) ? 0.f : expf(sacc[r] - m_new);
      sacc[r] = p;
      lloc += p;
    }
    l_run = l_run * alpha + lloc + __shfl_xor(lloc, 32);
    m_run = m_new;
#pragma unroll
    for (int dt = 0; dt < DT; ++dt)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc_o[dt][r] *= alpha;
      // ---- O^T += V P^T
#pragma unroll
    for (int s = 0; s < 16; ++s) {
      const int kidx = (s & 3) + 8 * (s >> 2) + 4 * h;
#pragma unroll
      for (int dt = 0; dt < DT; ++dt) {
        const float a = Vs[(dt * 32 + i) * VROW + kidx];
        acc_o[dt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, sacc[s], acc_o[dt], 0, 0, 0);
      }
    }
  }
  // ---- normalise and store (coalesced along queries)
  if (nsplit > 1) {
    // partial results: opart[((b*H+hd)*nsplit + split)][DP+2][T]  (rows 0..D-1: unnormalised O, then m, l)
    if (qi < T) {
      const int DP = 32 * DT;
      float* pb = opart + (((long)b * H + hd) * nsplit + split) * (DP + 2) * T;
#pragma unroll
      for (int dt = 0; dt < DT; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int d = dt * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
          pb[(long)d * T + qi] = acc_o[dt][r];
        }
      if (h == 0) {
        pb[(long)DP * T + qi] = m_run;
        pb[(long)(DP + 1) * T + qi] = l_run;
      }
    }
    return;
  }
  if (qi < T) {
    const float inv = 1.f / l_run;
    float* ob = out + obase;
#pragma unroll
    for (int dt = 0; dt < DT; ++dt)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int d = dt * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
        if (d < D) ob[(long)d * ld + qi] = (qi < len) ? acc_o[dt][r] * inv : 0.f;
      }
    if (m_out && h == 0) {
      m_out[((long)b * H + hd) * T + qi] = m_run;
      l_out[((long)b * H + hd) * T + qi] = l_run;
    }
  }
}

// ---------------------------------------------------------------------------------------------------
// The same flash-style kernel on the fp16 matrix cores with fp32-grade products (hi/lo split, three
// v_mfma_f32_32x32x16_f16 per product block, scale S = 256; see conv_h3.hip for the arithmetic).
//   S^T = K^T Q : A = K tile, LDS image [k16-step][op {S kh, kh, S kl}][h][key] (16 B = 8 head dims),
//                 B = Q fragment {qh, S ql} in registers (loaded once per workgroup)
//   O^T += V P^T: the 16 score registers of a lane are keys (r&3)+8(r>>2)+4h, so registers 8s..8s+7 ARE the 8 key
//                 slots of k16-step s (the summation order over keys is free): P never moves between lanes;
//                 V's LDS image [dt][k16-step][op][h][d] stores the keys of each element in that same order.
typedef _Float16 ahalf8 __attribute__((ext_vector_type(8)));
constexpr float kAttS = 256.f;
// timing ablations (tools/build_variant.sh ... -DRVCX_ATT_ABL=n; results are garbage, only the clock counts):
//   1 no MFMAs   2 no global loads   4 no commit (LDS stores + conversion)   8 no exp   16 one barrier per tile
#ifndef RVCX_ATT_ABL
#define RVCX_ATT_ABL 0
#endif
constexpr int kAttAbl = RVCX_ATT_ABL;
__device__ __forceinline__ f32x16 att_mfma(ahalf8 a, ahalf8 b, f32x16 c) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
#else
  return c;
#endif
}

struct AttRsrc {
#if defined(__HIP_DEVICE_COMPILE__)
  __amdgpu_buffer_rsrc_t r;
#endif
};
__device__ __forceinline__ AttRsrc att_rsrc(const float* base, int bytes) {
  AttRsrc b;
#if defined(__HIP_DEVICE_COMPILE__)
  b.r = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(base), 0, bytes, 0x00020000);
#endif
  return b;
}
__device__ __forceinline__ float att_load(const AttRsrc& b, int off) {   // offset >= bytes reads 0
#if defined(__HIP_DEVICE_COMPILE__)
  return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(b.r, off, 0, 0));
#else
  return 0.f;
#endif
}
constexpr int kAttOob = 0x7ffffff0;

// K and V in the attention kernel's LDS layout, split once per call (round 3).  Before, every workgroup converted the
// key tiles it walked -- the 25 query blocks of the TextEncoder each re-split all of K and V, and the conversion plus the
// 32 dword loads per thread and stage made the kernel VALU-issue bound (rocprofv3 --pmc: SQ_ACTIVE_INST_VALU 29 % of the
// wave cycles with ONE wave per SIMD, 760 VALU instructions per 32-key stage against 36 MFMAs).  Images per
// (batch, head, 32-key tile):  K [s][op {S kh, S kl}][h][key],  V [dt][s2][op {S vh, S vl}][h][d]  (16 B = 8 halves).
template <int DT>
__global__ __launch_bounds__(256) void att_presplit_kernel(const float* __restrict__ k, const float* __restrict__ v,
                                                           uint4* __restrict__ kimg, uint4* __restrict__ vimg, int H, int D,
                                                           int T, int ld, long in_bs, const int* lens, int* ovf_word,
                                                           int* ovf_layer, int seq) {
  constexpr int NS = 2 * DT;
  constexpr int K_ELEMS = NS * 2 * 32, V_ELEMS = DT * 2 * 2 * 32;
  const int kt = blockIdx.x, hd = blockIdx.y, b = blockIdx.z, tid = threadIdx.x;
  const int len = lens ? lens[b] : T;
  const int k0 = kt * 32;
  const long base = (long)b * in_bs + (long)hd * D * ld;
  const float* kb = k + base;
  const float* vb = v + base;
  uint4* ko = kimg + (((long)b * H + hd) * gridDim.x + kt) * (2 * K_ELEMS);
  uint4* vo = vimg + (((long)b * H + hd) * gridDim.x + kt) * (2 * V_ELEMS);
  bool ovf = false;
  for (int e0 = tid; e0 < K_ELEMS; e0 += 256) {
    const int key = e0 & 31, hh = (e0 >> 5) & 1, s = e0 >> 6;
    ahalf8 a0, a2;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int d = 16 * s + 8 * hh + e;
      const float val = (d < D && k0 + key < len) ? kb[(long)d * ld + k0 + key] : 0.f;
      ovf |= !(fabsf(val) < kH3ActLimit / kAttS);
      const _Float16 vh = (_Float16)val;
      a0[e] = (_Float16)((float)vh * kAttS);
      a2[e] = (_Float16)((val - (float)vh) * kAttS);
    }
    ko[((s * 2 + 0) * 2 + hh) * 32 + key] = __builtin_bit_cast(uint4, a0);
    ko[((s * 2 + 1) * 2 + hh) * 32 + key] = __builtin_bit_cast(uint4, a2);
  }
  for (int e0 = tid; e0 < V_ELEMS; e0 += 256) {
    const int dd = e0 & 31, hh = (e0 >> 5) & 1, s2 = (e0 >> 6) & 1, dt = e0 >> 7;
    const int d = dt * 32 + dd;
    ahalf8 a0, a2;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int r = 8 * s2 + e;
      const int key = k0 + (r & 3) + 8 * (r >> 2) + 4 * hh;
      const float val = (d < D && key < len) ? vb[(long)d * ld + key] : 0.f;
      ovf |= !(fabsf(val) < kH3ActLimit / kAttS);
      const _Float16 vh = (_Float16)val;
      a0[e] = (_Float16)((float)vh * kAttS);
      a2[e] = (_Float16)((val - (float)vh) * kAttS);
    }
    vo[(((dt * 2 + s2) * 2 + 0) * 2 + hh) * 32 + dd] = __builtin_bit_cast(uint4, a0);
    vo[(((dt * 2 + s2) * 2 + 1) * 2 + hh) * 32 + dd] = __builtin_bit_cast(uint4, a2);
  }
  if (ovf) report_h3_overflow(ovf_word, ovf_layer, seq);
}

typedef unsigned att_u32x4 __attribute__((ext_vector_type(4)));    // a native vector: arrays of HIP's uint4 wrapper stayed in scratch
template <int NKE, int NVE>
__device__ __forceinline__ void att_fetch(att_u32x4 (&rk)[NKE], att_u32x4 (&rv)[NVE], const uint4* __restrict__ kp,
                                          const uint4* __restrict__ vp, int tid) {
#pragma unroll
  for (int j = 0; j < NKE; ++j) rk[j] = *reinterpret_cast<const att_u32x4*>(kp + tid + 256 * j);
#pragma unroll
  for (int j = 0; j < NVE; ++j) rv[j] = *reinterpret_cast<const att_u32x4*>(vp + tid + 256 * j);
}
template <int NKE, int NVE>
__device__ __forceinline__ void att_commit(const att_u32x4 (&rk)[NKE], const att_u32x4 (&rv)[NVE], uint4* Ks, uint4* Vs,
                                           int tid) {
#pragma unroll
  for (int j = 0; j < NKE; ++j) *reinterpret_cast<att_u32x4*>(Ks + tid + 256 * j) = rk[j];
#pragma unroll
  for (int j = 0; j < NVE; ++j) *reinterpret_cast<att_u32x4*>(Vs + tid + 256 * j) = rv[j];
}

template <int DT>  // D <= 32*DT
__global__ __launch_bounds__(256, 2) void attn_h3_kernel(const float* __restrict__ q, const uint4* __restrict__ kimg,
                                                      const uint4* __restrict__ vimg, float* __restrict__ out,
                                                      float* __restrict__ m_out, float* __restrict__ l_out,
                                                      const float* __restrict__ relq, int H, int D, int T,
                                                      int ld, long in_bs, long out_bs, float scale, int window,
                                                      const int* lens, int nsplit, float* opart, int* ovf_word,
                                                      int* ovf_layer, int seq) {
  constexpr int NS = 2 * DT;                       // k16-steps over the head dimension
  __shared__ uint4 Ks[NS * 2 * 2 * 32];            // [s][op {S kh, S kl}][h][key]   (kh = (S kh)/S in registers)
  __shared__ uint4 Vs[DT * 2 * 2 * 2 * 32];        // [dt][s2][op {S vh, S vl}][h][d]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int i = lane & 31, h = lane >> 5;
  const int b = blockIdx.z, hd = blockIdx.y;
  const int qblk = blockIdx.x / nsplit, split = blockIdx.x - qblk * nsplit;
  const int q0 = qblk * 128 + wave * 32;
  const int len = lens ? lens[b] : T;
  const long base = (long)b * in_bs + (long)hd * D * ld;
  const long obase = (long)b * out_bs + (long)hd * D * ld;
  const float* qb = q + base;

  // K and V arrive split (att_presplit_kernel checked their range); Q must stay below 65504 (conv.h: kH3ActLimit)
  bool ovf = false;
  // Q fragment: lane (query i, half h), step s, slot e -> Q[16 s + 8 h + e][q0 + i] * scale, split {qh, S ql}
  ahalf8 qh[NS], ql[NS];
  const int qi = q0 + i;
#pragma unroll
  for (int s = 0; s < NS; ++s)
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int d = 16 * s + 8 * h + e;
      const float val = (d < D && qi < T) ? qb[(long)d * ld + qi] * scale : 0.f;
      ovf |= !(fabsf(val) < kH3ActLimit);
      const _Float16 vh = (_Float16)val;
      qh[s][e] = vh;
      ql[s][e] = (_Float16)((val - (float)vh) * kAttS);
    }

  f32x16 acc_o[DT];
#pragma unroll
  for (int dt = 0; dt < DT; ++dt)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc_o[dt][r] = 0.f;
  float m_run = -INFINITY, l_run = 0.f;
  const int nrel = 2 * window + 1;
  const float* relrow = relq ? relq + (((long)b * H + hd) * T + min(qi, T - 1)) * nrel : nullptr;

  // staging: a key tile's two images are 2 NS x 128 elements of 16 bytes, already in LDS order: plain coalesced copies,
  // requested one tile ahead into registers
  constexpr float invS = 1.f / kAttS;
  constexpr int KSZ = NS * 2 * 2 * 32, VSZ = DT * 2 * 2 * 2 * 32;
  constexpr int NKE = KSZ / 256, NVE = VSZ / 256;
  static_assert(KSZ % 256 == 0 && VSZ % 256 == 0, "image sizes are multiples of the workgroup");
  const int ntile = (T + 31) / 32;
  const uint4* kt_base = kimg + ((long)b * H + hd) * ntile * KSZ;
  const uint4* vt_base = vimg + ((long)b * H + hd) * ntile * VSZ;
  // (plain arrays handed to force-inlined functions: as captures of [&] lambdas they stayed in scratch memory -- 80 bytes per
  // thread, a scratch store right behind every prefetch load and a scratch load in front of every commit -- so the
  // "prefetch under the MFMAs" waited for its loads at once; round 4, found in the disassembly)
  att_u32x4 rk[NKE], rv[NVE];
  auto fetch = [&](int k0) {
    att_fetch<NKE, NVE>(rk, rv, kt_base + (long)(k0 >> 5) * KSZ, vt_base + (long)(k0 >> 5) * VSZ, tid);
  };
  auto commit = [&]() { att_commit<NKE, NVE>(rk, rv, Ks, Vs, tid); };

  const int tiles = (len + 31) / 32;
  const int klo = (int)((long)tiles * split / nsplit) * 32;
  const int khi = min(len, (int)((long)tiles * (split + 1) / nsplit) * 32);
  if (klo < khi) fetch(klo);
  for (int k0 = klo; k0 < khi; k0 += 32) {
    if (!(kAttAbl & 16)) __syncthreads();
    if (!(kAttAbl & 4) || k0 == klo) commit();
    __syncthreads();
    // the next tile is requested BEHIND the barrier (__syncthreads() waits vmcnt(0): issued in front of it, the loads'
    // whole round trip sat on every stage -- round 3) and lands under this tile's 36 MFMAs and the softmax
    if (k0 + 32 < khi && !(kAttAbl & 2)) fetch(k0 + 32);
    // ---- S^T tile (scaled by S): rows = keys, cols = queries
    f32x16 sacc;
#pragma unroll
    for (int r = 0; r < 16; ++r) sacc[r] = 0.f;
#pragma unroll
    for (int s = 0; s < NS; ++s) {
      if (16 * s < D) {
        const ahalf8 a0 = __builtin_bit_cast(ahalf8, Ks[((s * 2 + 0) * 2 + h) * 32 + i]);
        const ahalf8 a2 = __builtin_bit_cast(ahalf8, Ks[((s * 2 + 1) * 2 + h) * 32 + i]);
        const ahalf8 a1 = a0 * (_Float16)invS;
        if (kAttAbl & 1) {
          sacc[s] += (float)a0[0] + (float)a2[1] + (float)a1[2];
          continue;
        }
        sacc = att_mfma(a0, qh[s], sacc);
        sacc = att_mfma(a1, ql[s], sacc);
        sacc = att_mfma(a2, qh[s], sacc);
      }
    }
    // Only two kinds of tile need per-element tests: the few on the relative-position band (TextEncoder) and the one that
    // holds an item's last key.  Every other tile takes the straight-line form (same arithmetic on the same values: the
    // tests it drops are all false there) -- round 4: 16 exec-masked branch blocks per tile came out of the common path.
    const bool band = relq && (k0 <= q0 + 31 + window) && (k0 + 31 >= q0 - window);
    const bool edge = band || k0 + 32 > len;
    float mloc = -INFINITY;
    if (edge) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int key = k0 + (r & 3) + 8 * (r >> 2) + 4 * h;
        float sv = sacc[r] * invS;
        if (band) {
          const int rel = key - qi + window;
          if (rel >= 0 && rel < nrel && qi < T) sv += relrow[rel];
        }
        if (key >= len) sv = -INFINITY;
        sacc[r] = sv;
        mloc = fmaxf(mloc, sv);
      }
    } else {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        sacc[r] *= invS;
        mloc = fmaxf(mloc, sacc[r]);
      }
    }
    const float mtile = fmaxf(mloc, __shfl_xor(mloc, 32));
    const float m_new = fmaxf(m_run, mtile);
    const float alpha = (m_run == -INFINITY) ? 0.f : __expf(m_run - m_new);
    float lloc = 0.f;
    ahalf8 ph[2], pl[2];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      float p = (kAttAbl & 8) ? sacc[r] - m_new : __expf(sacc[r] - m_new);
      if (edge) p = (sacc[r] == -INFINITY) ? 0.f : p;      // (exp(-inf - m) is 0 too, but m itself may be -inf there)
      lloc += p;
      const _Float16 vh = (_Float16)p;
      ph[r >> 3][r & 7] = vh;
      pl[r >> 3][r & 7] = (_Float16)((p - (float)vh) * kAttS);
    }
    l_run = l_run * alpha + lloc + __shfl_xor(lloc, 32);
    m_run = m_new;
#pragma unroll
    for (int dt = 0; dt < DT; ++dt)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc_o[dt][r] *= alpha;
    // ---- O^T (scaled by S) += V P^T
#pragma unroll
    for (int dt = 0; dt < DT; ++dt)
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        const ahalf8 a0 = __builtin_bit_cast(ahalf8, Vs[(((dt * 2 + s2) * 2 + 0) * 2 + h) * 32 + i]);
        const ahalf8 a2 = __builtin_bit_cast(ahalf8, Vs[(((dt * 2 + s2) * 2 + 1) * 2 + h) * 32 + i]);
        const ahalf8 a1 = a0 * (_Float16)invS;
        if (kAttAbl & 1) {
          acc_o[dt][s2] += (float)a0[0] + (float)a2[1] + (float)a1[2] + (float)ph[s2][0] + (float)pl[s2][1];
          continue;
        }
        acc_o[dt] = att_mfma(a0, ph[s2], acc_o[dt]);
        acc_o[dt] = att_mfma(a1, pl[s2], acc_o[dt]);
        acc_o[dt] = att_mfma(a2, ph[s2], acc_o[dt]);
      }
  }
  if (ovf) report_h3_overflow(ovf_word, ovf_layer, seq);
#pragma unroll
  for (int dt = 0; dt < DT; ++dt)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc_o[dt][r] *= invS;
  if (nsplit > 1) {
    if (qi < T) {
      const int DP = 32 * DT;
      float* pb = opart + (((long)b * H + hd) * nsplit + split) * (DP + 2) * T;
#pragma unroll
      for (int dt = 0; dt < DT; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int d = dt * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
          pb[(long)d * T + qi] = acc_o[dt][r];
        }
      if (h == 0) {
        pb[(long)DP * T + qi] = m_run;
        pb[(long)(DP + 1) * T + qi] = l_run;
      }
    }
    return;
  }
  if (qi < T) {
    const float inv = 1.f / l_run;
    float* ob = out + obase;
#pragma unroll
    for (int dt = 0; dt < DT; ++dt)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int d = dt * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
        if (d < D) ob[(long)d * ld + qi] = (qi < len) ? acc_o[dt][r] * inv : 0.f;
      }
    if (m_out && h == 0) {
      m_out[((long)b * H + hd) * T + qi] = m_run;
      l_out[((long)b * H + hd) * T + qi] = l_run;
    }
  }
}


// merge the split-KV partials: out = sum_s e^(m_s - m) O_s / sum_s e^(m_s - m) l_s
// grid.y = head * 4 + quarter of the head dimension: 4x the workgroups of a (query block, head) grid
__global__ __launch_bounds__(256) void attn_merge_kernel(const float* __restrict__ opart, float* __restrict__ out,
                                                         float* m_out, float* l_out, int H, int D, int DP, int T, int ld,
                                                         long out_bs, int nsplit, const int* lens) {
  const int t = blockIdx.x * 64 + (threadIdx.x & 63), part = threadIdx.x >> 6;
  const int hd = blockIdx.y >> 2, dq = blockIdx.y & 3, b = blockIdx.z;
  if (t >= T) return;
  const float* pb = opart + (((long)b * H + hd) * nsplit) * (DP + 2) * T;
  const long ss = (long)(DP + 2) * T;
  float ms[8], w[8];
  float m = -INFINITY, l = 0.f;
#pragma unroll
  for (int s = 0; s < 8; ++s) {
    ms[s] = s < nsplit ? pb[s * ss + (long)DP * T + t] : -INFINITY;
    m = fmaxf(m, ms[s]);
  }
#pragma unroll
  for (int s = 0; s < 8; ++s) {
    w[s] = (ms[s] == -INFINITY) ? 0.f : expf(ms[s] - m);
    if (s < nsplit) l += w[s] * pb[s * ss + (long)(DP + 1) * T + t];
  }
  const int len = lens ? lens[b] : T;
  const float inv = 1.f / l;
  float* ob = out + (long)b * out_bs + (long)hd * D * ld;
  const int dper = (D + 3) / 4, d0 = dq * dper, d1 = min(D, d0 + dper);
  for (int d = d0 + part; d < d1; d += 4) {
    float acc = 0.f;
#pragma unroll
    for (int s = 0; s < 8; ++s)
      if (s < nsplit) acc += w[s] * pb[s * ss + (long)d * T + t];
    ob[(long)d * ld + t] = t < len ? acc * inv : 0.f;
  }
  if (m_out && part == 0 && dq == 0) {
    m_out[((long)b * H + hd) * T + t] = m;
    l_out[((long)b * H + hd) * T + t] = l;
  }
}

// relq[b][h][t][r] = sum_d (q[d][t]*scale) * Ek[r][d]      (attentions.py:83-86)
// block = 64 queries x 4 slices of the relative offsets (r = part, part+4, ...)
__global__ __launch_bounds__(256) void rel_logits_kernel(const float* __restrict__ q, const float* __restrict__ ek,
                                                         float* __restrict__ relq, int H, int D, int T, int ld,
                                                         long in_bs, float scale, int nrel) {
  extern __shared__ float eks[];  // [nrel][D]
  for (int idx = threadIdx.x; idx < nrel * D; idx += 256) eks[idx] = ek[idx];
  __syncthreads();
  const int tq = threadIdx.x & 63, part = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int nmine = part < nrel ? (nrel - part + 3) / 4 : 0;       // rows r = part + 4 i < nrel
  const int t = blockIdx.x * 64 + tq;
  const int hd = blockIdx.y, b = blockIdx.z;
  if (t >= T) return;
  const float* qb = q + (long)b * in_bs + (long)hd * D * ld;
  float acc[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) acc[i] = 0.f;
  // the q loads are hoisted 8 at a time: one at a time they form a chain of D dependent-latency global loads
  int d = 0;
  for (; d + 8 <= D; d += 8) {
    float qv[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) qv[u] = qb[(long)(d + u) * ld + t] * scale;
    // a wave holds one `part`: its rows r = part + 4 i are the same for every lane -- a scalar count instead of an
    // exec-masked test around every FMA, and the embedding row read 16 bytes at a time (round 4; per output the order of
    // the sum is unchanged)
#pragma unroll
    for (int i = 0; i < 8; ++i)
      if (i < nmine) {
        const float* e = eks + (part + 4 * i) * D + d;
        if ((D & 3) == 0) {
          const float4 e0 = *reinterpret_cast<const float4*>(e), e1 = *reinterpret_cast<const float4*>(e + 4);
          acc[i] = fmaf(qv[0], e0.x, acc[i]);
          acc[i] = fmaf(qv[1], e0.y, acc[i]);
          acc[i] = fmaf(qv[2], e0.z, acc[i]);
          acc[i] = fmaf(qv[3], e0.w, acc[i]);
          acc[i] = fmaf(qv[4], e1.x, acc[i]);
          acc[i] = fmaf(qv[5], e1.y, acc[i]);
          acc[i] = fmaf(qv[6], e1.z, acc[i]);
          acc[i] = fmaf(qv[7], e1.w, acc[i]);
        } else {
#pragma unroll
          for (int u = 0; u < 8; ++u) acc[i] = fmaf(qv[u], e[u], acc[i]);
        }
      }
  }
  for (; d < D; ++d) {
    const float qv = qb[(long)d * ld + t] * scale;
#pragma unroll
    for (int i = 0; i < 8; ++i)
      if (i < nmine) acc[i] = fmaf(qv, eks[(part + 4 * i) * D + d], acc[i]);
  }
  float* o = relq + (((long)b * H + hd) * T + t) * nrel;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int r = part + 4 * i;
    if (r < nrel) o[r] = acc[i];
  }
}

// out[d][t] += sum_r p[t][t+r-w] * Ev[r][d], p recomputed on the band from (m, l)
// (attentions.py:106-111)
__global__ __launch_bounds__(256) void rel_values_kernel(const float* __restrict__ q, const float* __restrict__ k,
                                                         const float* __restrict__ relq,
                                                         const float* __restrict__ m_in,
                                                         const float* __restrict__ l_in,
                                                         const float* __restrict__ ev, float* __restrict__ out,
                                                         int H, int D, int T, int ld, long in_bs, long out_bs, float scale,
                                                         int window, const int* lens) {
  // block = 32 queries x 8 slices; the q tile and the k band (32 + 2*window keys) are staged in LDS once
  extern __shared__ float sm[];  // evs [nrel][D] ; ps [32][nrel+1] ; qs [D][32] ; ks [D][KW]
  const int nrel = 2 * window + 1;
  const int KW = 32 + 2 * window;
  float* evs = sm;
  float* ps = evs + nrel * D;
  float* qs = ps + 32 * (nrel + 1);
  float* ks = qs + D * 32;
  const int tq = threadIdx.x & 31, part = threadIdx.x >> 5;
  const int hd = blockIdx.y, b = blockIdx.z;
  const int tb = blockIdx.x * 32;
  const int t = tb + tq;
  const int len = lens ? lens[b] : T;
  const long base = (long)b * in_bs + (long)hd * D * ld;
  const long obase = (long)b * out_bs + (long)hd * D * ld;
  for (int idx = threadIdx.x; idx < nrel * D; idx += 256) evs[idx] = ev[idx];
  for (int idx = threadIdx.x; idx < D * 32; idx += 256) {
    const int d = idx >> 5, j = idx & 31;
    qs[idx] = tb + j < T ? q[base + (long)d * ld + tb + j] * scale : 0.f;
  }
  for (int idx = threadIdx.x; idx < D * KW; idx += 256) {
    const int d = idx / KW, j = idx - d * KW;
    const int key = tb - window + j;
    ks[idx] = (key >= 0 && key < T) ? k[base + (long)d * ld + key] : 0.f;
  }
  __syncthreads();
  if (t < T) {
    const float m = m_in[((long)b * H + hd) * T + t], l = l_in[((long)b * H + hd) * T + t];
    for (int r = part; r < nrel; r += 8) {
      const int key = t + r - window;
      float p = 0.f;
      if (key >= 0 && key < len) {
        float s = 0.f;
        for (int d = 0; d < D; ++d) s = fmaf(qs[d * 32 + tq], ks[d * KW + tq + r], s);
        s += relq[(((long)b * H + hd) * T + t) * nrel + r];
        p = expf(s - m) / l;
      }
      ps[tq * (nrel + 1) + r] = p;
    }
  }
  __syncthreads();
  if (t < T && t < len) {
    for (int d = part; d < D; d += 8) {
      float acc = 0.f;
      for (int r = 0; r < nrel; ++r) acc = fmaf(ps[tq * (nrel + 1) + r], evs[r * D + d], acc);
      out[obase + (long)d * ld + t] += acc;
    }
  }
}

void launch_attention(const float* q, const float* k, const float* v, float* out, int B, int H, int D, int T,
                      int ld, long in_bs, long out_bs, float scale, const float* emb_rel_k, const float* emb_rel_v, int window,
                      const int* lens, float* scratch, float* split_scratch, hipStream_t stream, int* ovf, int* ovf_layer,
                      int seq, bool allow_h3) {
  RVCX_CHECK(D % 2 == 0 && D <= 96, "attention: head dim must be even and <= 96");
  const int nrel = 2 * window + 1;
  float *relq = nullptr, *mb = nullptr, *lb = nullptr;
  if (emb_rel_k) {
    RVCX_CHECK(nrel <= 32 && scratch != nullptr, "attention: window too large / no scratch");
    relq = scratch;
    mb = relq + (size_t)B * H * T * nrel;
    lb = mb + (size_t)B * H * T;
    hipLaunchKernelGGL(rel_logits_kernel, dim3(cdiv(T, 64), H, B), dim3(256), nrel * D * sizeof(float), stream,
                       q, emb_rel_k, relq, H, D, T, ld, in_bs, scale, nrel);
  }
  const int DT = cdiv(D, 32);
  // split the key range over several workgroups when the (query block, head, batch) grid is too small
  // decided per batch item: the number of key splits changes the softmax merge order, and a batched conversion
  // must reproduce its single-utterance runs bit for bit
  const long blocks = (long)cdiv(T, 128) * H;
  (void)B;
  int nsplit = 1;
  // swept in serial mode (the three knobs are for that): TextEncoder (T 3198, 2 heads) 4 splits 2.27 ms / 8 splits
  // 2.45 / 16 splits 2.74 / 2 splits 2.83 for its six layers; HuBERT (T 1599, 12 heads) 2 and 4 splits equal
  static const int max_split = std::min(8, getenv("RVCX_ATT_MAXSPLIT") ? atoi(getenv("RVCX_ATT_MAXSPLIT")) : 8);   // scratch: 8
  static const int want_blocks = getenv("RVCX_ATT_BLOCKS") ? atoi(getenv("RVCX_ATT_BLOCKS")) : 200;
  static const int min_keys = getenv("RVCX_ATT_MINKEYS") ? atoi(getenv("RVCX_ATT_MINKEYS")) : 256;
  while (nsplit < max_split && blocks * nsplit < want_blocks && T / (nsplit * 2) >= min_keys) nsplit *= 2;
  float* opart = nullptr;
  if (nsplit > 1) {
    if (split_scratch) opart = split_scratch;
    else nsplit = 1;
  }
  float* mo = nsplit > 1 ? nullptr : mb;
  float* lo = nsplit > 1 ? nullptr : lb;
  dim3 grid(cdiv(T, 128) * nsplit, H, B);
  static const int h3_env = getenv("RVCX_ATT_H3") ? atoi(getenv("RVCX_ATT_H3")) : 1;   // 0: exact-fp32 MFMA attention
  const int h3 = h3_env && !g_force_fp32 && allow_h3;
  if (h3) {
    // K / V split once into the kernel's LDS images (behind the split-KV partials in split_scratch)
    RVCX_CHECK(split_scratch != nullptr, "attention: the split-fp16 kernel needs its scratch");
    const int ntile = cdiv(T, 32);
    uint4* kimg = reinterpret_cast<uint4*>(split_scratch + (size_t)B * H * 8 * 98 * T);
    uint4* vimg = kimg + (size_t)B * H * ntile * (DT * 256);
    const dim3 pg(ntile, H, B);
    if (DT == 1) {
      hipLaunchKernelGGL(att_presplit_kernel<1>, pg, dim3(256), 0, stream, k, v, kimg, vimg, H, D, T, ld, in_bs, lens, ovf, ovf_layer, seq);
      hipLaunchKernelGGL(attn_h3_kernel<1>, grid, dim3(256), 0, stream, q, kimg, vimg, out, mo, lo, relq, H, D, T, ld, in_bs,
                         out_bs, scale, window, lens, nsplit, opart, ovf, ovf_layer, seq);
    } else if (DT == 2) {
      hipLaunchKernelGGL(att_presplit_kernel<2>, pg, dim3(256), 0, stream, k, v, kimg, vimg, H, D, T, ld, in_bs, lens, ovf, ovf_layer, seq);
      hipLaunchKernelGGL(attn_h3_kernel<2>, grid, dim3(256), 0, stream, q, kimg, vimg, out, mo, lo, relq, H, D, T, ld, in_bs,
                         out_bs, scale, window, lens, nsplit, opart, ovf, ovf_layer, seq);
    } else {
      hipLaunchKernelGGL(att_presplit_kernel<3>, pg, dim3(256), 0, stream, k, v, kimg, vimg, H, D, T, ld, in_bs, lens, ovf, ovf_layer, seq);
      hipLaunchKernelGGL(attn_h3_kernel<3>, grid, dim3(256), 0, stream, q, kimg, vimg, out, mo, lo, relq, H, D, T, ld, in_bs,
                         out_bs, scale, window, lens, nsplit, opart, ovf, ovf_layer, seq);
    }
  }
  else if (DT == 1)
    hipLaunchKernelGGL(attn_kernel<1>, grid, dim3(256), 0, stream, q, k, v, out, mo, lo, relq, H, D, T, ld, in_bs,
                       out_bs, scale, window, lens, nsplit, opart);
  else if (DT == 2)
    hipLaunchKernelGGL(attn_kernel<2>, grid, dim3(256), 0, stream, q, k, v, out, mo, lo, relq, H, D, T, ld, in_bs,
                       out_bs, scale, window, lens, nsplit, opart);
  else
    hipLaunchKernelGGL(attn_kernel<3>, grid, dim3(256), 0, stream, q, k, v, out, mo, lo, relq, H, D, T, ld, in_bs,
                       out_bs, scale, window, lens, nsplit, opart);
  if (nsplit > 1)
    hipLaunchKernelGGL(attn_merge_kernel, dim3(cdiv(T, 64), H * 4, B), dim3(256), 0, stream, opart, out, mb, lb, H, D,
                       32 * DT, T, ld, out_bs, nsplit, lens);
  if (emb_rel_v) {
    size_t lds = ((size_t)nrel * D + 32 * (nrel + 1) + (size_t)D * 32 + (size_t)D * (32 + 2 * window)) * sizeof(float);
    hipLaunchKernelGGL(rel_values_kernel, dim3(cdiv(T, 32), H, B), dim3(256), lds, stream, q, k, relq, mb, lb,
                       emb_rel_v, out, H, D, T, ld, in_bs, out_bs, scale, window, lens);
  }
  RVCX_HIP(hipGetLastError());
}

size_t attention_scratch_floats(int B, int H, int T, int window) {
  return (size_t)B * H * T * (2 * window + 1) + 2 * (size_t)B * H * T;
}

// scratch for the split-KV partials (worst case 8 splits, head dim padded to 96, + m and l rows)
// + the split K / V images of the split-fp16 kernel: 2 x 96 x 4 bytes per (head, key), keys rounded up to tiles of 32
size_t attention_split_floats(int B, int H, int T) { return (size_t)B * H * (8 * 98 * (size_t)T + 2 * 96 * ((size_t)T + 32)); }

double attention_flops(int B, int H, int D, int T) { return 4.0 * B * H * (double)T * T * D; }

}  // namespace rvcx
