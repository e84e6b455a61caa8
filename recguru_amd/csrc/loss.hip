// K7 / K8: fused gather-dot-loss over the item catalogue.
//
// Replaces the [B,L,k,d] negative-embedding gather + batched matmul + cat + CrossEntropyLoss of
// AutoEnc4Rec_cross.py:201-215 / AutoEnc4Rec.py:218-227 / tools/utils.py:76-84 /
// tools/lossfunctions.py:36-49 (mode 0, label 0) and the BPR loss of tools/utils.py:114-126 /
// tools/lossfunctions.py:56-72 (mode 1).  Nothing of size B*L*k*d is ever materialised: one wave
// owns one position, keeps the decoder state in registers, streams the 1+k item rows (each row a
// contiguous 2*d or 4*d bytes across the lanes), wave-reduces the dots and keeps a running
// log-sum-exp.  Positions with mask == 0 are skipped entirely (they contribute 0 to the masked
// mean and get 0 gradient), so their rows are never read.
//   loss = sum_t mask_t * l_t / sum_t mask_t            (Q12)
// Backward recomputes the dots (rows come from L2 / Infinity Cache), forms
// c_j = dl_t/dlogit_j * mask_t * gout / sum(mask), accumulates dh_t = sum_j c_j E[j] in registers and
// scatters c_j * h_t into the dense f32 table gradient with full-row (256 B per instruction) atomics.
#include <stdlib.h>
#include "rg_common.hip.h"
#include "rg_det.hip.h"
#include "../../include/recguru_hip.h"

#define LW 4  // waves per block

// BPR forms.  RG_LOSS_BPR (tools/lossfunctions.py:56-72): -log sigmoid(p - nbar), aux = p - nbar.
// RG_LOSS_BPR_SAS (:79-96): -[log(sigmoid(p) + 1e-24) + log(1 - sigmoid(nbar) + 1e-24)], aux = nbar (p is
// recomputed by the backward before it needs it; nbar is needed by every negative's coefficient).
#define RG_SAS_EPS 1e-24f
__device__ __forceinline__ float sigmoidf_(float x) { return 1.f / (1.f + __expf(-x)); }
__device__ __forceinline__ void bpr_value(int mode, float p, float nbar, float& loss, float& aux) {
  if (mode == RG_LOSS_BPR_SAS) {
    aux = nbar;
    loss = -(__logf(sigmoidf_(p) + RG_SAS_EPS) + __logf(1.f - sigmoidf_(nbar) + RG_SAS_EPS));
  } else {
    aux = p - nbar;
    loss = -__logf(sigmoidf_(aux));
  }
}
// d loss / d logit of row idx (0 = positive), before the position weight w; p = the positive's logit (SAS only)
__device__ __forceinline__ float bpr_coef(int mode, int idx, float aux, float p, int k) {
  if (mode == RG_LOSS_BPR_SAS) {
    if (idx == 0) { const float sp = sigmoidf_(p); return -sp * (1.f - sp) / (sp + RG_SAS_EPS); }
    const float sn = sigmoidf_(aux);
    return sn * (1.f - sn) / (1.f - sn + RG_SAS_EPS) / (float)k;
  }
  const float sg = 1.f / (1.f + __expf(aux));          // sigmoid(-(p - nbar))
  return idx == 0 ? -sg : sg / (float)k;
}

template <typename T, int NPL>
__device__ __forceinline__ float row_dot(const T* __restrict__ row, const float* h, int lane, int d, float* keep) {
  float s = 0.f;
#pragma unroll
  for (int j = 0; j < NPL; ++j) {
    const int e = lane + 64 * j;
    const float v = e < d ? (float)row[e] : 0.f;
    if (keep) keep[j] = v;
    s += v * h[j];
  }
  return wave_sum(s);
}

template <typename T, int NPL>
__global__ __launch_bounds__(64 * LW) void item_loss_fwd_kernel(rg_item_loss_args a) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const T* __restrict__ H = reinterpret_cast<const T*>(a.h);
  const T* __restrict__ E = reinterpret_cast<const T*>(a.table);
  const int d = a.d, k = a.k;
  float lsum = 0.f, msum = 0.f;
  for (long long t = (long long)blockIdx.x * LW + wave; t < a.ntok; t += (long long)gridDim.x * LW) {
    const float m = a.mask[t];
    if (m == 0.f) { if (lane == 0 && a.aux_tok) a.aux_tok[t] = 0.f; continue; }
    float h[NPL];
#pragma unroll
    for (int j = 0; j < NPL; ++j) { const int e = lane + 64 * j; h[j] = e < d ? (float)H[(size_t)t * d + e] : 0.f; }
    const float l0 = row_dot<T, NPL>(E + (size_t)a.pos[t] * d, h, lane, d, nullptr);
    float loss, aux;
    if (a.mode == RG_LOSS_SAMPLED_CE) {
      float mx = l0, s = 1.f;
      for (int j = 0; j < k; ++j) {
        const float lj = row_dot<T, NPL>(E + (size_t)a.neg[t * k + j] * d, h, lane, d, nullptr);
        if (lj > mx) { s = s * __expf(mx - lj) + 1.f; mx = lj; } else s += __expf(lj - mx);
      }
      aux = mx + __logf(s);
      loss = aux - l0;
    } else {
      float ns = 0.f;
      for (int j = 0; j < k; ++j) ns += row_dot<T, NPL>(E + (size_t)a.neg[t * k + j] * d, h, lane, d, nullptr);
      bpr_value(a.mode, l0, ns / (float)k, loss, aux);
    }
    if (lane == 0 && a.aux_tok) a.aux_tok[t] = aux;
    lsum += loss * m;
    msum += m;
  }
  __shared__ float red[2][LW];
  if (lane == 0) { red[0][wave] = lsum; red[1][wave] = msum; }
  __syncthreads();
  if (threadIdx.x == 0) {
    float s0 = 0.f, s1 = 0.f;
    for (int w = 0; w < LW; ++w) { s0 += red[0][w]; s1 += red[1][w]; }
    if (s1 != 0.f) { rg_acc(a.sums, s0); rg_acc(a.sums + 1, s1); }
  }
}

template <typename T, int NPL>
__global__ __launch_bounds__(64 * LW) void item_loss_bwd_kernel(rg_item_loss_args a) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const T* __restrict__ H = reinterpret_cast<const T*>(a.h);
  const T* __restrict__ E = reinterpret_cast<const T*>(a.table);
  T* __restrict__ dH = reinterpret_cast<T*>(a.dh);
  const int d = a.d, k = a.k;
  const float gs = a.gout[0] / a.sums[1];
  for (long long t = (long long)blockIdx.x * LW + wave; t < a.ntok; t += (long long)gridDim.x * LW) {
    const float m = a.mask[t];
    float dh[NPL];
#pragma unroll
    for (int j = 0; j < NPL; ++j) dh[j] = 0.f;
    if (m != 0.f) {
      const float w = m * gs;
      const float aux = a.aux_tok[t];
      float h[NPL], row[NPL];
#pragma unroll
      for (int j = 0; j < NPL; ++j) { const int e = lane + 64 * j; h[j] = e < d ? (float)H[(size_t)t * d + e] : 0.f; }
      float l0 = 0.f;
      for (int j = -1; j < k; ++j) {
        const long long item = j < 0 ? a.pos[t] : a.neg[t * k + j];
        const float lj = row_dot<T, NPL>(E + (size_t)item * d, h, lane, d, row);
        float c;
        if (j < 0) l0 = lj;
        if (a.mode == RG_LOSS_SAMPLED_CE) c = (__expf(lj - aux) - (j < 0 ? 1.f : 0.f)) * w;
        else c = bpr_coef(a.mode, j + 1, aux, l0, k) * w;
#pragma unroll
        for (int q = 0; q < NPL; ++q) {
          const int e = lane + 64 * q;
          dh[q] += c * row[q];
          if (e < d && item != a.skip_row) rg_acc(a.dE + (size_t)item * d + e, c * h[q]);
        }
      }
    }
#pragma unroll
    for (int q = 0; q < NPL; ++q) { const int e = lane + 64 * q; if (e < d) dH[(size_t)t * d + e] = (T)dh[q]; }
  }
}

// ------------------------------------------------------------------------------------------------
// Row-group form for d = 64 / 128 / 256: a row of the table is ONE 16-byte load per lane across LPR = d/8
// lanes, so a wave instruction fetches G = 64/LPR rows of G different items at once and the 1+k rows of a
// position take (1+k)/G independent gather steps (batched RG_U at a time so that their latencies overlap)
// instead of 1+k dependent ones made of 2-byte loads.  The decoder state h sits in registers twice: in the
// load layout (8 consecutive features per lane) for the dots and dh, and in the transposed layout
// (feature li + LPR*j) for the table-gradient atomics, whose wave instruction then covers G rows x 64
// contiguous bytes -- the four 64-byte requests a 256-byte atomic instruction is split into anyway.
// ------------------------------------------------------------------------------------------------
#define RG_U 4

template <int LPR>
__device__ __forceinline__ float group_sum(float v) {
#pragma unroll
  for (int o = 1; o < LPR; o <<= 1) v += __shfl_xor(v, o);
  return v;
}

template <typename T, int LPR>
__global__ __launch_bounds__(64 * LW) void item_loss_fwd_rows_kernel(rg_item_loss_args a) {
  constexpr int G = 64 / LPR;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int gi = lane / LPR, li = lane % LPR;
  const T* __restrict__ H = reinterpret_cast<const T*>(a.h);
  const T* __restrict__ E = reinterpret_cast<const T*>(a.table);
  const int d = a.d, k = a.k, n = a.k + 1;
  float lsum = 0.f, msum = 0.f;
  for (long long t = (long long)blockIdx.x * LW + wave; t < a.ntok; t += (long long)gridDim.x * LW) {
    const float m = a.mask[t];
    if (m == 0.f) { if (lane == 0 && a.aux_tok) a.aux_tok[t] = 0.f; continue; }
    float h[8];
    load8(h, H + (size_t)t * d + 8 * li);
    const long long pos = a.pos[t];
    float mx = -INFINITY, sm = 0.f, ns = 0.f, l0 = 0.f;      // per row group
    for (int i0 = 0; i0 < n; i0 += G * RG_U) {
      long long item[RG_U];
      float e[RG_U][8];
#pragma unroll
      for (int u = 0; u < RG_U; ++u) {
        const int idx = i0 + u * G + gi;
        item[u] = (idx == 0 || idx >= n) ? pos : a.neg[t * k + idx - 1];
      }
#pragma unroll
      for (int u = 0; u < RG_U; ++u) load8(e[u], E + (size_t)item[u] * d + 8 * li);
#pragma unroll
      for (int u = 0; u < RG_U; ++u) {
        const int idx = i0 + u * G + gi;
        float dot = 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) dot += e[u][j] * h[j];
        dot = group_sum<LPR>(dot);
        if (idx < n) {
          if (idx == 0) l0 = dot;
          if (a.mode == RG_LOSS_SAMPLED_CE) {
            const float nm = fmaxf(mx, dot);
            sm = sm * __expf(mx - nm) + __expf(dot - nm);
            mx = nm;
          } else if (idx > 0) ns += dot;
        }
      }
    }
    // combine the G row groups
    l0 = __shfl(l0, 0);
    float loss, aux;
    if (a.mode == RG_LOSS_SAMPLED_CE) {
#pragma unroll
      for (int o = LPR; o < 64; o <<= 1) {
        const float omx = __shfl_xor(mx, o), osm = __shfl_xor(sm, o);
        const float nm = fmaxf(mx, omx);
        sm = (mx == -INFINITY ? 0.f : sm * __expf(mx - nm)) + (omx == -INFINITY ? 0.f : osm * __expf(omx - nm));
        mx = nm;
      }
      aux = mx + __logf(sm);
      loss = aux - l0;
    } else {
#pragma unroll
      for (int o = LPR; o < 64; o <<= 1) ns += __shfl_xor(ns, o);
      bpr_value(a.mode, l0, ns / (float)k, loss, aux);
    }
    if (lane == 0 && a.aux_tok) a.aux_tok[t] = aux;
    lsum += loss * m;
    msum += m;
  }
  __shared__ float red[2][LW];
  if (lane == 0) { red[0][wave] = lsum; red[1][wave] = msum; }
  __syncthreads();
  if (threadIdx.x == 0) {
    float s0 = 0.f, s1 = 0.f;
    for (int w = 0; w < LW; ++w) { s0 += red[0][w]; s1 += red[1][w]; }
    if (s1 != 0.f) { rg_acc(a.sums, s0); rg_acc(a.sums + 1, s1); }
  }
}

template <typename T, int LPR, bool CBUF>
__global__ __launch_bounds__(64 * LW) void item_loss_bwd_rows_kernel(rg_item_loss_args a, float* __restrict__ cbuf) {
  constexpr int G = 64 / LPR;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int gi = lane / LPR, li = lane % LPR;
  const T* __restrict__ H = reinterpret_cast<const T*>(a.h);
  const T* __restrict__ E = reinterpret_cast<const T*>(a.table);
  T* __restrict__ dH = reinterpret_cast<T*>(a.dh);
  const int d = a.d, k = a.k, n = a.k + 1;
  const float gs = a.gout[0] / a.sums[1];
  for (long long t = (long long)blockIdx.x * LW + wave; t < a.ntok; t += (long long)gridDim.x * LW) {
    const float m = a.mask[t];
    float dh[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) dh[j] = 0.f;
    if (m != 0.f) {
      const float w = m * gs;
      const float aux = a.aux_tok[t];
      float l0 = 0.f;                               // the positive's logit (BPR_SAS coefficient of row 0)
      float h[8], ht[8];
      load8(h, H + (size_t)t * d + 8 * li);
      if (!CBUF) {
#pragma unroll
        for (int j = 0; j < 8; ++j) ht[j] = (float)H[(size_t)t * d + li + LPR * j];
      }
      const long long pos = a.pos[t];
      for (int i0 = 0; i0 < n; i0 += G * RG_U) {
        long long item[RG_U];
        float e[RG_U][8];
#pragma unroll
        for (int u = 0; u < RG_U; ++u) {
          const int idx = i0 + u * G + gi;
          item[u] = (idx == 0 || idx >= n) ? pos : a.neg[t * k + idx - 1];
        }
#pragma unroll
        for (int u = 0; u < RG_U; ++u) load8(e[u], E + (size_t)item[u] * d + 8 * li);
#pragma unroll
        for (int u = 0; u < RG_U; ++u) {
          const int idx = i0 + u * G + gi;
          float dot = 0.f;
#pragma unroll
          for (int j = 0; j < 8; ++j) dot += e[u][j] * h[j];
          dot = group_sum<LPR>(dot);
          float c;
          if (idx == 0) l0 = dot;
          if (a.mode == RG_LOSS_SAMPLED_CE) c = (__expf(dot - aux) - (idx == 0 ? 1.f : 0.f)) * w;
          else c = bpr_coef(a.mode, idx, aux, l0, k) * w;
          if (idx >= n) c = 0.f;
#pragma unroll
          for (int j = 0; j < 8; ++j) dh[j] += c * e[u][j];
          if (CBUF) {                       // binned path: the table gradient is built from c by the kernels below
            if (idx < n && li == 0) cbuf[t * n + idx] = c;
          } else if (idx < n && item[u] != a.skip_row) {
            float* __restrict__ dst = a.dE + (size_t)item[u] * d + li;
#pragma unroll
            for (int j = 0; j < 8; ++j) rg_acc(dst + LPR * j, c * ht[j]);
          }
        }
      }
#pragma unroll
      for (int o = LPR; o < 64; o <<= 1)
#pragma unroll
        for (int j = 0; j < 8; ++j) dh[j] += __shfl_xor(dh[j], o);
    }
    if (gi == 0) store8(dH + (size_t)t * d + 8 * li, dh);
  }
}

// Training form (grad mode): the forward and the rows pass of the backward are ONE walk over the 1+k rows.
// The rows stay in registers between the log-sum-exp and the coefficient pass, so they are gathered once
// instead of twice (the second gather was the whole cost of K1 above).  The upstream gradient is not known yet:
// c and dh are produced for gout = 1 (w = mask_t / sums[1], sums[1] = the mask count, which the caller supplies
// BEFORE the launch) and the backward scales them (bin_fill multiplies c by gout, rg_scale_dev does dh; both
// are exact no-ops for gout == 1, the value a plain loss.backward() sends).  NIT batches of G * RG_U rows.
template <typename T, int LPR, int NIT>
__global__ __launch_bounds__(64 * LW) void item_loss_train_rows_kernel(rg_item_loss_args a, float* __restrict__ cbuf) {
  constexpr int G = 64 / LPR;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int gi = lane / LPR, li = lane % LPR;
  const T* __restrict__ H = reinterpret_cast<const T*>(a.h);
  const T* __restrict__ E = reinterpret_cast<const T*>(a.table);
  T* __restrict__ dH = reinterpret_cast<T*>(a.dh);
  const int d = a.d, k = a.k, n = a.k + 1;
  const float gs = 1.f / a.sums[1];
  float lsum = 0.f;
  for (long long t = (long long)blockIdx.x * LW + wave; t < a.ntok; t += (long long)gridDim.x * LW) {
    const float m = a.mask[t];
    float dh[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) dh[j] = 0.f;
    if (m != 0.f) {
      float h[8];
      load8(h, H + (size_t)t * d + 8 * li);
      const long long pos = a.pos[t];
      long long item[NIT][RG_U];
      float e[NIT][RG_U][8], dot[NIT][RG_U];
#pragma unroll
      for (int it = 0; it < NIT; ++it)
#pragma unroll
        for (int u = 0; u < RG_U; ++u) {
          const int idx = (it * RG_U + u) * G + gi;
          item[it][u] = (idx == 0 || idx >= n) ? pos : a.neg[t * k + idx - 1];
        }
#pragma unroll
      for (int it = 0; it < NIT; ++it)
#pragma unroll
        for (int u = 0; u < RG_U; ++u) load8(e[it][u], E + (size_t)item[it][u] * d + 8 * li);
      float mx = -INFINITY, sm = 0.f, ns = 0.f, l0 = 0.f;      // per row group
#pragma unroll
      for (int it = 0; it < NIT; ++it)
#pragma unroll
        for (int u = 0; u < RG_U; ++u) {
          const int idx = (it * RG_U + u) * G + gi;
          float s = 0.f;
#pragma unroll
          for (int j = 0; j < 8; ++j) s += e[it][u][j] * h[j];
          s = group_sum<LPR>(s);
          dot[it][u] = s;
          if (idx < n) {
            if (idx == 0) l0 = s;
            if (a.mode == RG_LOSS_SAMPLED_CE) {
              const float nm = fmaxf(mx, s);
              sm = sm * __expf(mx - nm) + __expf(s - nm);
              mx = nm;
            } else if (idx > 0) ns += s;
          }
        }
      l0 = __shfl(l0, 0);
      float loss, aux;
      if (a.mode == RG_LOSS_SAMPLED_CE) {
#pragma unroll
        for (int o = LPR; o < 64; o <<= 1) {
          const float omx = __shfl_xor(mx, o), osm = __shfl_xor(sm, o);
          const float nm = fmaxf(mx, omx);
          sm = (mx == -INFINITY ? 0.f : sm * __expf(mx - nm)) + (omx == -INFINITY ? 0.f : osm * __expf(omx - nm));
          mx = nm;
        }
        aux = mx + __logf(sm);
        loss = aux - l0;
      } else {
#pragma unroll
        for (int o = LPR; o < 64; o <<= 1) ns += __shfl_xor(ns, o);
        bpr_value(a.mode, l0, ns / (float)k, loss, aux);
      }
      lsum += loss * m;
      const float w = m * gs;
#pragma unroll
      for (int it = 0; it < NIT; ++it)
#pragma unroll
        for (int u = 0; u < RG_U; ++u) {
          const int idx = (it * RG_U + u) * G + gi;
          float c;
          if (a.mode == RG_LOSS_SAMPLED_CE) c = (__expf(dot[it][u] - aux) - (idx == 0 ? 1.f : 0.f)) * w;
          else c = bpr_coef(a.mode, idx, aux, l0, k) * w;
          if (idx >= n) c = 0.f;
#pragma unroll
          for (int j = 0; j < 8; ++j) dh[j] += c * e[it][u][j];
          if (idx < n && li == 0) cbuf[t * n + idx] = c;
        }
#pragma unroll
      for (int o = LPR; o < 64; o <<= 1)
#pragma unroll
        for (int j = 0; j < 8; ++j) dh[j] += __shfl_xor(dh[j], o);
    }
    if (gi == 0) store8(dH + (size_t)t * d + 8 * li, dh);
  }
  __shared__ float red[LW];
  if (lane == 0) red[wave] = lsum;
  __syncthreads();
  if (threadIdx.x == 0) {
    float s0 = 0.f;
    for (int w = 0; w < LW; ++w) s0 += red[w];
    if (s0 != 0.f) rg_acc(a.sums, s0);
  }
}


// Training form for ANY k (sampled softmax, mode RG_LOSS_SAMPLED_CE; config-5: k = 1024 rows of 512 B per position): the
// 1 + k rows cannot wait in registers, so the softmax is taken ONLINE -- per lane group a running maximum, a running sum and a
// running vector  A = sum_j exp(l_j - max) E[j]  that is rescaled whenever the maximum moves (the flash-attention
// recurrence with the item rows as V) -- and the rows are still gathered exactly ONCE per position:
//   loss_t = lse - l_0,    dh_t = (A / sum - E[pos]) * mask_t / count      (d loss / d h for label 0).
// The coefficients of the table gradient need the final lse, which is not known while the rows pass by: the kernel leaves
// the RAW logits in cbuf and lse in aux_tok, and the binned scatter (bin_fill) forms c = (exp(l - lse) - [j == 0]) mask / count
// from them while it sorts the pairs anyway.  The two-call form gathers the rows twice (0.47 TB each at config-5).
template <typename T, int LPR>
__global__ __launch_bounds__(64 * LW) void item_loss_train_online_kernel(rg_item_loss_args a, float* __restrict__ cbuf) {
  constexpr int G = 64 / LPR;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int gi = lane / LPR, li = lane % LPR;
  const T* __restrict__ H = reinterpret_cast<const T*>(a.h);
  const T* __restrict__ E = reinterpret_cast<const T*>(a.table);
  T* __restrict__ dH = reinterpret_cast<T*>(a.dh);
  const int d = a.d, k = a.k, n = a.k + 1;
  const float gs = 1.f / a.sums[1];
  float lsum = 0.f;
  for (long long t = (long long)blockIdx.x * LW + wave; t < a.ntok; t += (long long)gridDim.x * LW) {
    const float m = a.mask[t];
    float dh[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) dh[j] = 0.f;
    if (m != 0.f) {
      float h[8], e0[8], A[8];
      load8(h, H + (size_t)t * d + 8 * li);
      const long long pos = a.pos[t];
#pragma unroll
      for (int j = 0; j < 8; ++j) { A[j] = 0.f; e0[j] = 0.f; }
      float mx = -INFINITY, sm = 0.f, l0 = 0.f;          // per row group
      for (int i0 = 0; i0 < n; i0 += G * RG_U) {
        long long item[RG_U];
        float e[RG_U][8];
#pragma unroll
        for (int u = 0; u < RG_U; ++u) {
          const int idx = i0 + u * G + gi;
          item[u] = (idx == 0 || idx >= n) ? pos : a.neg[t * k + idx - 1];
        }
#pragma unroll
        for (int u = 0; u < RG_U; ++u) load8(e[u], E + (size_t)item[u] * d + 8 * li);
#pragma unroll
        for (int u = 0; u < RG_U; ++u) {
          const int idx = i0 + u * G + gi;
          float dot = 0.f;
#pragma unroll
          for (int j = 0; j < 8; ++j) dot += e[u][j] * h[j];
          dot = group_sum<LPR>(dot);
          if (idx < n) {
            if (idx == 0) {
              l0 = dot;
#pragma unroll
              for (int j = 0; j < 8; ++j) e0[j] = e[u][j];
            }
            if (li == 0) cbuf[t * n + idx] = dot;
            const float nm = fmaxf(mx, dot);
            const float r = __expf(mx - nm), p = __expf(dot - nm);     // (mx == -inf: r = 0, sm and A are still 0)
            sm = sm * r + p;
#pragma unroll
            for (int j = 0; j < 8; ++j) A[j] = A[j] * r + p * e[u][j];
            mx = nm;
          }
        }
      }
      // combine the G row groups (a group that saw no row has mx = -inf and weight 0)
      float M = mx;
#pragma unroll
      for (int o = LPR; o < 64; o <<= 1) M = fmaxf(M, __shfl_xor(M, o));
      const float wg = mx == -INFINITY ? 0.f : __expf(mx - M);
      sm *= wg;
#pragma unroll
      for (int j = 0; j < 8; ++j) A[j] *= wg;
#pragma unroll
      for (int o = LPR; o < 64; o <<= 1) {
        sm += __shfl_xor(sm, o);
#pragma unroll
        for (int j = 0; j < 8; ++j) A[j] += __shfl_xor(A[j], o);
      }
      l0 = __shfl(l0, 0);
      const float lse = M + __logf(sm);
      lsum += (lse - l0) * m;
      if (lane == 0) a.aux_tok[t] = lse;
      const float w = m * gs, inv = 1.f / sm;
#pragma unroll
      for (int j = 0; j < 8; ++j) dh[j] = (A[j] * inv - e0[j]) * w;       // e0 lives in group 0, which is the group that stores
    } else if (lane == 0) {
      a.aux_tok[t] = 0.f;
    }
    if (gi == 0) store8(dH + (size_t)t * d + 8 * li, dh);
  }
  __shared__ float red[LW];
  if (lane == 0) red[wave] = lsum;
  __syncthreads();
  if (threadIdx.x == 0) {
    float s0 = 0.f;
    for (int w = 0; w < LW; ++w) s0 += red[w];
    if (s0 != 0.f) rg_acc(a.sums, s0);
  }
}


// The same kernel with the gather PIPELINED (round 5): per 64 rows the item ids arrive by ONE coalesced load (lane l: row i0 + l; the next
// 64 are requested a block ahead) and reach the lane groups through ds_bpermute -- the id load was a second dependent memory latency in
// front of every batch of rows --, and the rows of batch s + 1 are requested BEFORE batch s is reduced (two register buffers), so a wave
// keeps 2 x RG_U x G rows in flight instead of RG_U x G.  The rows of a lane group pass through the online recurrence in the same order:
// bit-identical logits, lse, loss and dh (tests/test_kernels_gpu.py::test_item_loss_online_pipelined_equals_plain).  RG_ITEM_ONLINE_PLAIN=1
// selects the plain kernel above (A/B).
template <typename T, int LPR>
__global__ __launch_bounds__(64 * LW) void item_loss_train_online2_kernel(rg_item_loss_args a, float* __restrict__ cbuf) {
  constexpr int G = 64 / LPR, RS = G * RG_U, NS = 64 / RS;          // rows per step, steps per block of 64 rows (even)
  static_assert(NS % 2 == 0, "two register buffers alternate inside a block of 64 rows");
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int gi = lane / LPR, li = lane % LPR;
  const T* __restrict__ H = reinterpret_cast<const T*>(a.h);
  const T* __restrict__ E = reinterpret_cast<const T*>(a.table);
  T* __restrict__ dH = reinterpret_cast<T*>(a.dh);
  const int d = a.d, k = a.k, n = a.k + 1;
  const float gs = 1.f / a.sums[1];
  float lsum = 0.f;
  for (long long t = (long long)blockIdx.x * LW + wave; t < a.ntok; t += (long long)gridDim.x * LW) {
    const float m = a.mask[t];
    float dh[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) dh[j] = 0.f;
    if (m != 0.f) {
      float h[8], e0[8], A[8];
      load8(h, H + (size_t)t * d + 8 * li);
      const long long pos = a.pos[t];
      const int64_t* __restrict__ negt = a.neg + (size_t)t * k;
      // lane l: the item of row i0 + l (row 0 = the positive; rows >= n read the positive's row again and are ignored)
      auto load_ids = [&](int i0) -> long long {
        const int idx = i0 + lane;
        const long long v = negt[min(max(idx - 1, 0), k - 1)];       // unconditional load from a clamped address
        return (idx == 0 || idx >= n) ? pos : v;
      };
      Frag<T> e[2][RG_U];                                 // RAW rows (bf16: 4 registers each) until they are reduced
      auto issue = [&](Frag<T> (&eb)[RG_U], long long ids, int j0) {
        const int idlo = (int)(ids & 0xFFFFFFFFll), idhi = (int)(ids >> 32);
#pragma unroll
        for (int u = 0; u < RG_U; ++u) {
          const int src = j0 + u * G + gi;
          const long long item = ((long long)__shfl(idhi, src) << 32) | (unsigned int)__shfl(idlo, src);
          load_frag(eb[u], E + (size_t)item * d + 8 * li);
        }
      };
#pragma unroll
      for (int j = 0; j < 8; ++j) { A[j] = 0.f; e0[j] = 0.f; }
      float mx = -INFINITY, sm = 0.f, l0 = 0.f;          // per row group
      long long idn = load_ids(0);
      issue(e[0], idn, 0);
      for (int i0 = 0; i0 < n; i0 += 64) {
        const long long idc = idn;
        idn = load_ids(min(i0 + 64, n - 1));             // (unconditional; unused past the last block)
#pragma unroll 1
        for (int sp = 0; sp < NS; sp += 2) {               // two steps per trip: the register buffers keep compile-time indices
#pragma unroll
          for (int half = 0; half < 2; ++half) {
            const int s = sp + half;
            // the next batch of rows -- the next step of this block, or the first step of the next block -- before this one is
            // reduced; ONE unconditional load sequence from selected ids (a load under a branch is waited for at its join)
            const bool wrap = s + 1 >= NS;
            issue(e[(half + 1) & 1], wrap ? idn : idc, wrap ? 0 : (s + 1) * RS);
            Frag<T> (&eb)[RG_U] = e[half];
#pragma unroll
            for (int u = 0; u < RG_U; ++u) {
              const int idx = i0 + s * RS + u * G + gi;
              float ev[8];
#pragma unroll
              for (int j = 0; j < 8; ++j) ev[j] = (float)eb[u].v[j];
              float dot = 0.f;
#pragma unroll
              for (int j = 0; j < 8; ++j) dot += ev[j] * h[j];
              dot = group_sum<LPR>(dot);
              if (idx < n) {
                if (idx == 0) {
                  l0 = dot;
#pragma unroll
                  for (int j = 0; j < 8; ++j) e0[j] = ev[j];
                }
                if (li == 0) cbuf[t * n + idx] = dot;
                const float nm = fmaxf(mx, dot);
                const float r = __expf(mx - nm), p = __expf(dot - nm);     // (mx == -inf: r = 0, sm and A are still 0)
                sm = sm * r + p;
#pragma unroll
                for (int j = 0; j < 8; ++j) A[j] = A[j] * r + p * ev[j];
                mx = nm;
              }
            }
          }
        }
      }
      float M = mx;
#pragma unroll
      for (int o = LPR; o < 64; o <<= 1) M = fmaxf(M, __shfl_xor(M, o));
      const float wg = mx == -INFINITY ? 0.f : __expf(mx - M);
      sm *= wg;
#pragma unroll
      for (int j = 0; j < 8; ++j) A[j] *= wg;
#pragma unroll
      for (int o = LPR; o < 64; o <<= 1) {
        sm += __shfl_xor(sm, o);
#pragma unroll
        for (int j = 0; j < 8; ++j) A[j] += __shfl_xor(A[j], o);
      }
      l0 = __shfl(l0, 0);
      const float lse = M + __logf(sm);
      lsum += (lse - l0) * m;
      if (lane == 0) a.aux_tok[t] = lse;
      const float w = m * gs, inv = 1.f / sm;
#pragma unroll
      for (int j = 0; j < 8; ++j) dh[j] = (A[j] * inv - e0[j]) * w;
    } else if (lane == 0) {
      a.aux_tok[t] = 0.f;
    }
    if (gi == 0) store8(dH + (size_t)t * d + 8 * li, dh);
  }
  __shared__ float red[LW];
  if (lane == 0) red[wave] = lsum;
  __syncthreads();
  if (threadIdx.x == 0) {
    float s0 = 0.f;
    for (int w = 0; w < LW; ++w) s0 += red[w];
    if (s0 != 0.f) rg_acc(a.sums, s0);
  }
}


template <typename T>
static int launch(const rg_item_loss_args& a, bool bwd, hipStream_t s) {
  long long g = (a.ntok + LW - 1) / LW;
  if (g > 256 * 32) g = 256 * 32;
  dim3 grid((int)g), block(64 * LW);
#define RG_R(LPR)                                                                             \
  do {                                                                                        \
    if (bwd) hipLaunchKernelGGL((item_loss_bwd_rows_kernel<T, LPR, false>), grid, block, 0, s, a, (float*)nullptr);   \
    else hipLaunchKernelGGL((item_loss_fwd_rows_kernel<T, LPR>), grid, block, 0, s, a);       \
    RG_CHECK_LAUNCH();                                                                        \
    return 0;                                                                                 \
  } while (0)
  if (a.d == 64) RG_R(8);
  if (a.d == 128) RG_R(16);
  if (a.d == 256) RG_R(32);
#undef RG_R
#define RG_L(NPL)                                                                   \
  if (bwd) hipLaunchKernelGGL((item_loss_bwd_kernel<T, NPL>), grid, block, 0, s, a); \
  else hipLaunchKernelGGL((item_loss_fwd_kernel<T, NPL>), grid, block, 0, s, a)
  if (a.d <= 64) { RG_L(1); }
  else if (a.d <= 128) { RG_L(2); }
  else if (a.d <= 256) { RG_L(4); }
  else return rg_set_error_msg(RG_ERR_UNSUPPORTED, "item_loss: d > 256");
#undef RG_L
  RG_CHECK_LAUNCH();
  return 0;
}

// ------------------------------------------------------------------------------------------------
// Table gradient without memory-side atomics per (position, item) pair ("binned" backward).
// The atomic form adds (1+k) rows of d floats per position: 7.4 GB per launch at the bench shape, which runs at
// the chip-wide float-atomic rate (~1.3 TB/s) whatever the schedule.  Here the backward writes only the scalar
// c[t, j] = dloss/dlogit; the pairs are then counting-sorted by item BIN (RG_RPB consecutive table rows), and one
// workgroup per bin chunk accumulates  dE[item] += c * h[t]  in LDS (h rows gathered from L2 / Infinity Cache,
// 2 bytes per element instead of 4 bytes of atomic per element) and flushes its RG_RPB rows once.
//   K1 item_loss_bwd_rows_kernel<.., CBUF>   dh, c                       (no atomics)
//   K2 bin_count_kernel                       pairs per bin (LDS histogram per workgroup)
//   K3 bin_scan_kernel                        bin offsets, chunk offsets, cursors   (one workgroup)
//   K4 bin_fill_kernel                        (t, row-in-bin, c) triples grouped by bin
//   K5 bin_accumulate_kernel                  LDS accumulation per bin chunk, one flush of RG_RPB rows
// ------------------------------------------------------------------------------------------------
#define RG_RPB 64            // table rows per bin
#define RG_RPB_LOG 6
#define RG_CHUNK 2048        // entries per accumulate work item (item loss)
#define RG_CHUNK_EMB 512     // ... embedding scatter: a hot item's thousands of positions spread over many workgroups
#define RG_MAXBINS 8192
#define RG_PPW 8192          // (position, item) pairs per workgroup in count / fill ...
#define RG_PPW_WIDE 65536    // ... and with thousands of bins (256-row bins of a 2 M-row table: 7 813): a workgroup spends one GLOBAL
                             // atomic per non-empty bin in each of the two kernels -- at 8 192 pairs nearly one per pair (575 M per
                             // launch at config-5); at 65 536 pairs a bin gets ~8 entries per workgroup

struct BinWs {
  float* c; int* hist; int* start; int* cursor; int* chunk_start;
  int* chunk_bin;    // [chunks] the bin of each accumulate work item (written by the scan: no search per work item)
  int chunk;         // entries per accumulate work item
  uint2* ent;        // sorted entries: x = (position << RG_RPB_LOG) | row-in-bin, y = c as bits  (one 8-byte store / load)
  int nbins;
  int ppw;           // pairs per workgroup of bin_count / bin_fill
  int bin_log;       // log2 of the table rows per bin: RG_RPB_LOG (64 rows), or RG_WIDE_LOG (256 rows) for catalogues beyond RG_MAXBINS * 64 rows
  const float* cscale;  // non-null: the c values are for gout = 1 and bin_fill multiplies them by cscale[0]
  const float* lse;     // non-null (rg_item_loss_train's online form): c holds RAW logits; bin_fill forms
  const float* count;   //   c = (exp(l - lse[t]) - [j == 0]) * mask[t] / count[0]
};

// The items of pairs p, p+256, p+512, p+768 (-1: masked position, skip row, or p >= p1).  Branch-free: clamped
// indices and ONE id load per pair from a selected pointer, so that the 8 loads of a batch are in flight together
// (a pair at a time, mask -> id was two dependent latencies per loop iteration).
#define RG_PB 4
__device__ __forceinline__ void pair_items4(const rg_item_loss_args& a, long long p, long long p1, int n, long long (&it)[RG_PB]) {
  float mk[RG_PB];
#pragma unroll
  for (int u = 0; u < RG_PB; ++u) {
    const long long pc = min(p + 256 * u, p1 - 1);
    const long long t = (long long)((unsigned int)pc / (unsigned int)n);     // (pairs < 2^31 by the workspace contract: a 32-bit
    const int idx = (int)(pc - t * n);                                        //  division instead of the 64-bit expansion)
    mk[u] = a.mask[t];
    const int64_t* __restrict__ src = idx == 0 ? a.pos + t : a.neg + t * a.k + (idx - 1);
    it[u] = *src;
  }
#pragma unroll
  for (int u = 0; u < RG_PB; ++u)
    if (p + 256 * u >= p1 || mk[u] == 0.f || it[u] == a.skip_row) it[u] = -1;
}

__global__ __launch_bounds__(256) void bin_count_kernel(rg_item_loss_args a, BinWs w) {
  extern __shared__ int lh[];
  const int n = a.k + 1;
  const long long npairs = a.ntok * n;
  for (int i = threadIdx.x; i < w.nbins; i += 256) lh[i] = 0;
  __syncthreads();
  const long long p0 = (long long)blockIdx.x * w.ppw, p1 = min(p0 + w.ppw, npairs);
  for (long long p = p0 + threadIdx.x; p < p1; p += 256 * RG_PB) {
    long long it[RG_PB];
    pair_items4(a, p, p1, n, it);
#pragma unroll
    for (int u = 0; u < RG_PB; ++u)
      if (it[u] >= 0) atomicAdd(&lh[it[u] >> w.bin_log], 1);
  }
  __syncthreads();
  for (int i = threadIdx.x; i < w.nbins; i += 256)
    if (lh[i]) atomicAdd(&w.hist[i], lh[i]);
}

// one workgroup: start[b] = exclusive prefix of hist, cursor = start, chunk_start = exclusive prefix of ceil(hist/CHUNK)
__global__ __launch_bounds__(1024) void bin_scan_kernel(BinWs w) {
  __shared__ int wt[16], wtc[16];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int per = (w.nbins + 1023) / 1024;
  const int b0 = tid * per, b1 = min(b0 + per, w.nbins);
  int s = 0, sc = 0;
  for (int b = b0; b < b1; ++b) { s += w.hist[b]; sc += (w.hist[b] + w.chunk - 1) / w.chunk; }
  // exclusive scan of (s, sc) over the 1024 threads: shuffles inside a wave, 16 wave totals through LDS
  int is = s, isc = sc;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const int u = __shfl_up(is, o), uc = __shfl_up(isc, o);
    if (lane >= o) { is += u; isc += uc; }
  }
  if (lane == 63) { wt[wv] = is; wtc[wv] = isc; }
  __syncthreads();
  int base = 0, basec = 0;
  for (int i = 0; i < wv; ++i) { base += wt[i]; basec += wtc[i]; }
  if (tid == 1023) { w.start[w.nbins] = base + is; w.chunk_start[w.nbins] = basec + isc; }
  s = base + is - s; sc = basec + isc - sc;
  for (int b = b0; b < b1; ++b) {
    w.start[b] = s; w.cursor[b] = s; w.chunk_start[b] = sc;
    const int nc = (w.hist[b] + w.chunk - 1) / w.chunk;
    for (int j = 0; j < nc; ++j) w.chunk_bin[sc + j] = b;
    s += w.hist[b]; sc += nc;
  }
}

__global__ __launch_bounds__(256) void bin_fill_kernel(rg_item_loss_args a, BinWs w) {
  extern __shared__ int lh[];           // [nbins] counts, then running offsets; [nbins] reserved bases
  int* base = lh + w.nbins;
  const int n = a.k + 1;
  const long long npairs = a.ntok * n;
  const float gsc = w.cscale ? w.cscale[0] : 1.f;
  for (int i = threadIdx.x; i < w.nbins; i += 256) lh[i] = 0;
  __syncthreads();
  const long long p0 = (long long)blockIdx.x * w.ppw, p1 = min(p0 + w.ppw, npairs);
  for (long long p = p0 + threadIdx.x; p < p1; p += 256 * RG_PB) {
    long long it[RG_PB];
    pair_items4(a, p, p1, n, it);
#pragma unroll
    for (int u = 0; u < RG_PB; ++u)
      if (it[u] >= 0) atomicAdd(&lh[it[u] >> w.bin_log], 1);
  }
  __syncthreads();
  for (int i = threadIdx.x; i < w.nbins; i += 256) {
    const int cnt = lh[i];
    base[i] = cnt ? atomicAdd(&w.cursor[i], cnt) : 0;
    lh[i] = 0;
  }
  __syncthreads();
  for (long long p = p0 + threadIdx.x; p < p1; p += 256 * RG_PB) {
    long long it[RG_PB];
    float cv[RG_PB];
    pair_items4(a, p, p1, n, it);
#pragma unroll
    for (int u = 0; u < RG_PB; ++u) cv[u] = w.c[min(p + 256 * u, p1 - 1)];
    if (w.lse) {          // raw logits -> softmax coefficients (uniform branch)
      const float ic = 1.f / w.count[0];
#pragma unroll
      for (int u = 0; u < RG_PB; ++u) {
        const long long pc = min(p + 256 * u, p1 - 1);
        const long long t = (long long)((unsigned int)pc / (unsigned int)n);
        cv[u] = (__expf(cv[u] - w.lse[t]) - (pc - t * n == 0 ? 1.f : 0.f)) * a.mask[t] * ic;
      }
    }
#pragma unroll
    for (int u = 0; u < RG_PB; ++u) cv[u] *= gsc;
#pragma unroll
    for (int u = 0; u < RG_PB; ++u) {
      if (it[u] < 0) continue;
      const long long pu = p + 256 * u;
      const int b = (int)(it[u] >> w.bin_log);
      const int e = base[b] + atomicAdd(&lh[b], 1);
      w.ent[e] = make_uint2((((unsigned int)pu / (unsigned int)n) << w.bin_log) | (unsigned int)(it[u] & ((1 << w.bin_log) - 1)), __float_as_uint(cv[u]));
    }
  }
}

// One work item = up to RG_CHUNK entries of one bin.  LDS float atomics run at about one lane per clock per CU on
// gfx950 (measured: 128 ds_add_f32 lanes per entry made this kernel 14x slower than its gathers), so there are
// none: the chunk's entries are counting-sorted by row-in-bin inside LDS (integer LDS atomics only), each wave
// then owns whole rows, sums c * h[t] for a row in registers (LPR lanes per entry, G entries per wave
// instruction, U per lane group in flight), reduces the G lane groups by shuffles and parks the row in an LDS
// tile that is flushed once, coalesced, with 256-byte-shaped global atomics.
// DROP (the embedding scatter below): the rows are the gradient of an embedding output -- sparse bins, and under dropout
// element (t, e) of h is multiplied by the dropout multiplier of index t*D + e, regenerated here.
template <typename T, int LPR, bool DROP>
__global__ __launch_bounds__(256) void bin_accumulate_kernel(rg_item_loss_args a, BinWs w, long long table_rows, DropCfg drop) {
  constexpr int CH = DROP ? RG_CHUNK_EMB : RG_CHUNK;
  constexpr int G = 64 / LPR, D = LPR * 8, U = 4, EPT = CH / 256;
  extern __shared__ float sm[];                     // acc [RG_RPB][D] | sorted t [CH] | sorted c [CH]
  float* acc = sm;
  int* st = reinterpret_cast<int*>(sm + RG_RPB * D);
  float* sc = reinterpret_cast<float*>(st + CH);
  __shared__ int cnt[RG_RPB], rstart[RG_RPB + 1];
  __shared__ int s_bin, s_lo, s_hi, s_alone;
  __shared__ float part[4 * D];                     // hot rows: one partial row per wave
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int gi = lane / LPR, li = lane % LPR;
  const T* __restrict__ H = reinterpret_cast<const T*>(a.h);
  const int nchunks = w.chunk_start[w.nbins];
  for (int ch = blockIdx.x; ch < nchunks; ch += gridDim.x) {
    if (tid == 0) {
      const int lo = w.chunk_bin[ch];               // (a binary search over chunk_start here: 11 dependent L2 round trips per work item)
      const int bs = w.start[lo], be = w.start[lo + 1];
      const int e0 = bs + (ch - w.chunk_start[lo]) * CH;
      s_bin = lo; s_lo = e0; s_hi = min(e0 + CH, be);
      s_alone = be - bs <= CH;                // the bin's only work item: its rows belong to this workgroup alone
    }
    if (tid < RG_RPB) cnt[tid] = 0;
    __syncthreads();
    const int e0 = s_lo, e1 = s_hi, bin = s_bin;
    // ---- counting sort of the chunk by row-in-bin
    int myr[EPT], myp[EPT], myt[EPT];
    float myc[EPT];
#pragma unroll
    for (int i = 0; i < EPT; ++i) {
      const int e = e0 + tid + 256 * i;
      const uint2 en = w.ent[max(min(e, e1 - 1), 0)];       // unconditional (clamped): the 8 loads fly together
      myr[i] = e < e1 ? (int)(en.x & (RG_RPB - 1)) : -1;
      myt[i] = (int)(en.x >> RG_RPB_LOG);
      myc[i] = __uint_as_float(en.y);
    }
#pragma unroll
    for (int i = 0; i < EPT; ++i)
      if (myr[i] >= 0) myp[i] = atomicAdd(&cnt[myr[i]], 1);
    __syncthreads();
    if (tid < 64) {                                  // RG_RPB == 64: one wave scans the row counts
      const int v = cnt[tid];
      int incl = v;
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) { const int up = __shfl_up(incl, o); if (tid >= o) incl += up; }
      rstart[tid] = incl - v;
      if (tid == 63) rstart[64] = incl;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < EPT; ++i)
      if (myr[i] >= 0) { const int p = rstart[myr[i]] + myp[i]; st[p] = myt[i]; sc[p] = myc[i]; }
    __syncthreads();
    // ---- sparse bins (the embedding scatter: a few entries per row, some hot rows): rows of up to CAP entries are summed
    // by ONE LANE GROUP each, G rows of a wave in flight together and nothing reduced across groups (a row per wave made
    // 16 dependent gather rounds per chunk); longer rows take the wave-per-row loop below
    constexpr int CAP = 2 * U;
    if (DROP) {
      for (int r = wave * G + gi; r < RG_RPB; r += 4 * G) {
        const int lo = rstart[r], hi = rstart[r + 1];
        float s8[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) s8[j] = 0.f;
        const bool mine = hi - lo <= CAP;
#pragma unroll
        for (int it = 0; it < CAP / U; ++it) {
          const int eb = lo + it * U;
          int t[U];
          float c[U], h[U][8];
#pragma unroll
          for (int u = 0; u < U; ++u) {
            const int e = max(min(eb + u, hi - 1), 0);
            t[u] = st[e];
            c[u] = (mine && eb + u < hi) ? sc[e] : 0.f;
          }
#pragma unroll
          for (int u = 0; u < U; ++u) load8(h[u], H + (size_t)t[u] * D + 8 * li);
#pragma unroll
          for (int u = 0; u < U; ++u) {
            if (drop.thresh) {
              float kp[8];
              rg_keep8(drop, (unsigned int)t[u] * (unsigned int)D + 8u * (unsigned int)li, kp);
#pragma unroll
              for (int j = 0; j < 8; ++j) s8[j] += c[u] * kp[j] * h[u][j];
            } else {
#pragma unroll
              for (int j = 0; j < 8; ++j) s8[j] += c[u] * h[u][j];
            }
          }
        }
        if (mine) store8(acc + r * D + 8 * li, s8);
      }
    }
    // ---- hot rows (a popular item: up to the whole work item is ONE row): all four waves share
    // the row's entries and their partial sums meet in LDS (a single wave walked them 16 at a time)
    constexpr int HOT = DROP ? 64 : 512;
    {
      for (int r = 0; r < RG_RPB; ++r) {                     // (uniform over the workgroup)
        const int lo = rstart[r], hi = rstart[r + 1];
        if (hi - lo <= HOT) continue;
        float s8[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) s8[j] = 0.f;
        for (int eb = lo + (wave * G + gi) * U; eb < hi; eb += 4 * G * U) {
          int t[U];
          float c[U], h[U][8];
#pragma unroll
          for (int u = 0; u < U; ++u) {
            const int e = min(eb + u, hi - 1);
            t[u] = st[e];
            c[u] = (eb + u < hi) ? sc[e] : 0.f;
          }
#pragma unroll
          for (int u = 0; u < U; ++u) load8(h[u], H + (size_t)t[u] * D + 8 * li);
#pragma unroll
          for (int u = 0; u < U; ++u) {
            if (DROP && drop.thresh) {
              float kp[8];
              rg_keep8(drop, (unsigned int)t[u] * (unsigned int)D + 8u * (unsigned int)li, kp);
#pragma unroll
              for (int j = 0; j < 8; ++j) s8[j] += c[u] * kp[j] * h[u][j];
            } else {
#pragma unroll
              for (int j = 0; j < 8; ++j) s8[j] += c[u] * h[u][j];
            }
          }
        }
#pragma unroll
        for (int o = LPR; o < 64; o <<= 1)
#pragma unroll
          for (int j = 0; j < 8; ++j) s8[j] += __shfl_xor(s8[j], o);
        if (gi == 0) store8(part + wave * D + 8 * li, s8);
        __syncthreads();
        if (tid < D) acc[r * D + tid] = part[tid] + part[D + tid] + part[2 * D + tid] + part[3 * D + tid];
        __syncthreads();
      }
    }
    // ---- dense bins (the item loss) and the mid-length rows of sparse ones: each wave sums whole rows
    for (int r = wave; r < RG_RPB; r += 4) {
      const int lo = rstart[r], hi = rstart[r + 1];
      if ((DROP && hi - lo <= CAP) || hi - lo > HOT) continue;
      float s8[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) s8[j] = 0.f;
      for (int eb = lo + gi * U; eb < hi; eb += G * U) {
        int t[U];
        float c[U], h[U][8];
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const int e = min(eb + u, hi - 1);
          t[u] = st[e];
          c[u] = (eb + u < hi) ? sc[e] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < U; ++u) load8(h[u], H + (size_t)t[u] * D + 8 * li);
#pragma unroll
        for (int u = 0; u < U; ++u) {
          if (DROP && drop.thresh) {
            float kp[8];
            rg_keep8(drop, (unsigned int)t[u] * (unsigned int)D + 8u * (unsigned int)li, kp);
#pragma unroll
            for (int j = 0; j < 8; ++j) s8[j] += c[u] * kp[j] * h[u][j];
          } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) s8[j] += c[u] * h[u][j];
          }
        }
      }
#pragma unroll
      for (int o = LPR; o < 64; o <<= 1)
#pragma unroll
        for (int j = 0; j < 8; ++j) s8[j] += __shfl_xor(s8[j], o);
      if (gi == 0) store8(acc + r * D + 8 * li, s8);
    }
    __syncthreads();
    if (s_alone) {                                   // plain read-modify-write: no other work item of this launch adds to these rows
      for (int i = tid; i < RG_RPB * D; i += 256) {
        const long long row = (long long)bin * RG_RPB + i / D;
        const float v = acc[i];
        if (row < table_rows && v != 0.f) a.dE[row * D + (i % D)] += v;
      }
    } else {
      for (int i = tid; i < RG_RPB * D; i += 256) {
        const long long row = (long long)bin * RG_RPB + i / D;
        const float v = acc[i];
        if (row < table_rows && v != 0.f) atomicAdd(a.dE + row * D + (i % D), v);
      }
    }
    __syncthreads();
  }
}

// Large catalogues (table_rows > RG_MAXBINS * 64, e.g. 2 M items): bins of RG_WIDE_RPB = 256 rows keep the bin histograms of
// count / fill inside LDS, and a [256][D] f32 accumulation tile would not fit -- so a work item is a LONGER run of one bin's
// entries (RG_CHUNK_WIDE: ~32 entries per row at a uniform draw), counting-sorted by row inside LDS in two passes over the
// entries (no per-thread arrays), and every row's sum leaves straight from the wave that formed it: through a per-wave LDS
// row (so that the global adds are one dword per lane, 256 contiguous bytes per instruction: the shape the memory-side
// float-atomic unit takes at full rate).  Per (position, item) pair the table gradient then costs one gather of h[t]
// (2 bytes per element) plus 1 / (entries per row and chunk) of a 4-byte atomic per element.
#define RG_WIDE_LOG 8
#ifndef RG_WIDE_U_OF
#define RG_WIDE_U_OF(G) 4          // (round 5: 16 / 8 rows per lane group in flight -- 32 per wave -- measured 187 -> 264 ms per config-5 step)
#endif
#define RG_WIDE_RPB 256
#define RG_CHUNK_WIDE 8192
template <typename T, int LPR>
__global__ __launch_bounds__(256) void bin_accumulate_wide_kernel(rg_item_loss_args a, BinWs w, long long table_rows) {
  constexpr int CH = RG_CHUNK_WIDE, RB = RG_WIDE_RPB;
  // U rows of h in flight per lane group.  The kernel already moves 6.7 TB/s of L2 <-> fabric bytes (PMC, profiles/r05/c5: h rows are
  // re-read ~1 000 times, a quarter from the Infinity Cache): a deeper gather (U = 16: 32 rows per wave, the registers are free at two
  // workgroups per CU) made it SLOWER, 187 -> 264 ms per config-5 step -- it is at the memory system's rate, not short of loads
  constexpr int G = 64 / LPR, D = LPR * 8, U = RG_WIDE_U_OF(G);
  extern __shared__ float sm[];                     // sorted t [CH] | sorted c [CH] | row buffers [4][D]
  int* st = reinterpret_cast<int*>(sm);
  float* sc = sm + CH;
  float* rowbuf = sm + 2 * CH;
  __shared__ int cnt[RB], rstart[RB + 1], wsum[4];
  __shared__ int s_bin, s_lo, s_hi, s_alone;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int gi = lane / LPR, li = lane % LPR;
  const T* __restrict__ H = reinterpret_cast<const T*>(a.h);
  const int nchunks = w.chunk_start[w.nbins];
  for (int ch = blockIdx.x; ch < nchunks; ch += gridDim.x) {
    if (tid == 0) {
      const int lo = w.chunk_bin[ch];
      const int bs = w.start[lo], be = w.start[lo + 1];
      const int e0 = bs + (ch - w.chunk_start[lo]) * CH;
      s_bin = lo; s_lo = e0; s_hi = min(e0 + CH, be);
      s_alone = be - bs <= CH;
    }
    cnt[tid] = 0;                                    // RB == 256 == blockDim
    __syncthreads();
    const int e0 = s_lo, e1 = s_hi, bin = s_bin;
    const bool alone = s_alone != 0;
    // ---- counting sort of the run by row-in-bin: histogram, scan, placement (the second read of the entries hits L2)
    for (int e = e0 + tid; e < e1; e += 256) atomicAdd(&cnt[w.ent[e].x & (RB - 1)], 1);
    __syncthreads();
    {
      const int v = cnt[tid];
      int incl = v;
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) { const int up = __shfl_up(incl, o); if (lane >= o) incl += up; }
      if (lane == 63) wsum[wave] = incl;
      __syncthreads();
      int base = 0;
      for (int i = 0; i < wave; ++i) base += wsum[i];
      rstart[tid] = base + incl - v;
      if (tid == 255) rstart[RB] = base + incl;
      cnt[tid] = 0;
    }
    __syncthreads();
    for (int e = e0 + tid; e < e1; e += 256) {
      const uint2 en = w.ent[e];
      const int r = (int)(en.x & (RB - 1));
      const int p = rstart[r] + atomicAdd(&cnt[r], 1);
      st[p] = (int)(en.x >> RG_WIDE_LOG);
      sc[p] = __uint_as_float(en.y);
    }
    __syncthreads();
    // ---- each wave sums whole rows in registers and sends them off itself
    float* myrow = rowbuf + wave * D;
    for (int r = wave; r < RB; r += 4) {
      const int lo = rstart[r], hi = rstart[r + 1];
      if (hi == lo) continue;
      float s8[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) s8[j] = 0.f;
      for (int eb = lo + gi * U; eb < hi; eb += G * U) {
        int t[U];
        float c[U], h[U][8];
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const int e = min(eb + u, hi - 1);
          t[u] = st[e];
          c[u] = (eb + u < hi) ? sc[e] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < U; ++u) load8(h[u], H + (size_t)t[u] * D + 8 * li);
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
          for (int j = 0; j < 8; ++j) s8[j] += c[u] * h[u][j];
      }
#pragma unroll
      for (int o = LPR; o < 64; o <<= 1)
#pragma unroll
        for (int j = 0; j < 8; ++j) s8[j] += __shfl_xor(s8[j], o);
      if (gi == 0) store8(myrow + 8 * li, s8);
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");     // (same wave: LDS operations complete in order)
      __builtin_amdgcn_wave_barrier();
      const long long row = (long long)bin * RB + r;
      if (row < table_rows) {
        float* __restrict__ dst = a.dE + row * D;
#pragma unroll
        for (int i = 0; i < D / 64; ++i) {
          const float v = myrow[lane + 64 * i];
          if (v != 0.f) { if (alone) dst[lane + 64 * i] += v; else atomicAdd(dst + lane + 64 * i, v); }
        }
      }
      __builtin_amdgcn_wave_barrier();
    }
    __syncthreads();
  }
}

static size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

// bytes of workspace rg_item_loss_bwd_binned needs; 0 if the shape is not supported by the binned path
static int bin_log_of(long long table_rows) {      // rows per bin: 64, or 256 for catalogues of more than RG_MAXBINS * 64 rows; 0: too large
  if ((table_rows + RG_RPB - 1) / RG_RPB <= RG_MAXBINS) return RG_RPB_LOG;
  if ((table_rows + RG_WIDE_RPB - 1) / RG_WIDE_RPB <= RG_MAXBINS) return RG_WIDE_LOG;
  return 0;
}
extern "C" size_t rg_item_loss_bwd_binned_workspace(long long ntok, int k, int d, long long table_rows) {
  if (RG_DET) return 0;     // the binned kernels sum rows in the order the counting sort's cursors were claimed: not offered by the deterministic build
  const int bl = bin_log_of(table_rows);
  if (!bl) return 0;
  const long long nbins = (table_rows + (1LL << bl) - 1) >> bl;
  const long long npairs = ntok * (k + 1);
  if (!(d == 64 || d == 128 || d == 256) || npairs >= (1LL << 31) || ntok <= 0 || ntok >= (1LL << (32 - bl))) return 0;
  return align256(npairs * 4) + align256(npairs * 8) + align256((nbins + 1) * 4) * 4 + align256((nbins + npairs / RG_CHUNK_EMB + 1) * 4);
}

template <typename T>
static int launch_binned(const rg_item_loss_args& a, const float* coef, void* ws, size_t ws_bytes, long long table_rows, hipStream_t s,
                         const DropCfg* drop = nullptr) {
  const int n = a.k + 1;
  const long long npairs = a.ntok * n;
  const size_t need = rg_item_loss_bwd_binned_workspace(a.ntok, a.k, a.d, table_rows);
  if (!need) return rg_set_error_msg(RG_ERR_UNSUPPORTED, "item_loss_bwd_binned: needs d in {64,128,256}, <= 8192 bins of 64 (or 256) rows, < 2^31 pairs, < 2^26 (2^24) positions");
  if (!ws || ws_bytes < need) return rg_set_error_msg(RG_ERR_INVALID, "item_loss_bwd_binned: workspace too small");
  BinWs w;
  w.bin_log = bin_log_of(table_rows);
  const bool wide = w.bin_log == RG_WIDE_LOG;
  if (wide && drop) return rg_set_error_msg(RG_ERR_UNSUPPORTED, "embed_scatter_bwd_binned: tables of more than 524288 rows take the atomic form");
  w.nbins = (int)((table_rows + (1LL << w.bin_log) - 1) >> w.bin_log);
  char* p = reinterpret_cast<char*>(ws);
  w.c = coef ? const_cast<float*>(coef) : reinterpret_cast<float*>(p); p += align256(npairs * 4);
  w.cscale = coef ? a.gout : nullptr;
  w.lse = coef ? a.aux_tok : nullptr;           // rg_item_loss_scatter_binned after the online training form
  w.count = a.sums ? a.sums + 1 : nullptr;      // sums[1]: the mask count
  if (w.lse && !w.count) return rg_set_error_msg(RG_ERR_INVALID, "item_loss_scatter_binned: aux_tok (lse of raw logits) needs sums");
  w.chunk = drop ? RG_CHUNK_EMB : (wide ? RG_CHUNK_WIDE : RG_CHUNK);
  w.ent = reinterpret_cast<uint2*>(p); p += align256(npairs * 8);
  const size_t ib = align256((size_t)(w.nbins + 1) * 4);
  w.hist = reinterpret_cast<int*>(p); p += ib;
  w.start = reinterpret_cast<int*>(p); p += ib;
  w.cursor = reinterpret_cast<int*>(p); p += ib;
  w.chunk_start = reinterpret_cast<int*>(p); p += ib;
  w.chunk_bin = reinterpret_cast<int*>(p);
  hipError_t e = hipMemsetAsync(w.hist, 0, ib, s);
  if (e != hipSuccess) return rg_set_error(e, "item_loss_bwd_binned(memset)");
  // K1: dh and c (not when rg_item_loss_train already made them)
  long long g = (a.ntok + LW - 1) / LW;
  if (g > 256 * 32) g = 256 * 32;
  if (coef) {}
  else if (a.d == 64) hipLaunchKernelGGL((item_loss_bwd_rows_kernel<T, 8, true>), dim3((int)g), dim3(64 * LW), 0, s, a, w.c);
  else if (a.d == 128) hipLaunchKernelGGL((item_loss_bwd_rows_kernel<T, 16, true>), dim3((int)g), dim3(64 * LW), 0, s, a, w.c);
  else hipLaunchKernelGGL((item_loss_bwd_rows_kernel<T, 32, true>), dim3((int)g), dim3(64 * LW), 0, s, a, w.c);
  // K2..K4
  static const int ppw_wide_env = [] { const char* e = getenv("RG_PPW_WIDE"); return e ? atoi(e) : 0; }();      // (A/B: pairs per workgroup of count / fill)
  w.ppw = w.nbins > 4096 ? (ppw_wide_env >= 8192 ? ppw_wide_env : RG_PPW_WIDE) : RG_PPW;
  const int gp = (int)((npairs + w.ppw - 1) / w.ppw);
  hipLaunchKernelGGL(bin_count_kernel, dim3(gp), dim3(256), (size_t)w.nbins * 4, s, a, w);
  hipLaunchKernelGGL(bin_scan_kernel, dim3(1), dim3(1024), 0, s, w);
  hipFuncSetAttribute(reinterpret_cast<const void*>(bin_fill_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, w.nbins * 8);
  hipLaunchKernelGGL(bin_fill_kernel, dim3(gp), dim3(256), (size_t)w.nbins * 8, s, a, w);
  // K5
  if (wide) {
    const size_t smw = (size_t)RG_CHUNK_WIDE * 8 + (size_t)4 * a.d * 4;
#define RG_ACCW(LPR)                                                                                                 \
  do {                                                                                                               \
    hipFuncSetAttribute(reinterpret_cast<const void*>(bin_accumulate_wide_kernel<T, LPR>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smw); \
    hipLaunchKernelGGL((bin_accumulate_wide_kernel<T, LPR>), dim3(256 * 2), dim3(256), smw, s, a, w, table_rows);      \
  } while (0)
    if (a.d == 64) RG_ACCW(8);
    else if (a.d == 128) RG_ACCW(16);
    else RG_ACCW(32);
#undef RG_ACCW
    RG_CHECK_LAUNCH();
    return 0;
  }
  const size_t smem = (size_t)RG_RPB * a.d * 4 + (size_t)(drop ? RG_CHUNK_EMB : RG_CHUNK) * 8;
  const int ga = 256 * (int)(smem <= 40 * 1024 ? 4 : (smem <= 52 * 1024 ? 3 : (smem <= 80 * 1024 ? 2 : 1)));
#define RG_ACC1(LPR, DR)                                                                                            \
  do {                                                                                                              \
    hipFuncSetAttribute(reinterpret_cast<const void*>(bin_accumulate_kernel<T, LPR, DR>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem); \
    hipLaunchKernelGGL((bin_accumulate_kernel<T, LPR, DR>), dim3(ga), dim3(256), smem, s, a, w, table_rows, dc);      \
  } while (0)
#define RG_ACC(LPR) do { if (drop) RG_ACC1(LPR, true); else RG_ACC1(LPR, false); } while (0)
  const DropCfg dc = drop ? *drop : make_drop(0.f, 0);
  if (a.d == 64) RG_ACC(8);
  else if (a.d == 128) RG_ACC(16);
  else RG_ACC(32);
#undef RG_ACC1
#undef RG_ACC
  RG_CHECK_LAUNCH();
  return 0;
}

// Same results as rg_item_loss_bwd (dh, dE += table gradient), table gradient built by counting sort + LDS
// accumulation instead of one global atomic row per (position, item) pair.  workspace: device memory of at least
// rg_item_loss_bwd_binned_workspace() bytes, contents undefined on entry and exit.
extern "C" int rg_item_loss_bwd_binned(const rg_item_loss_args* a, long long table_rows, void* workspace, size_t workspace_bytes,
                                       int dtype, void* stream) {
  if (!a || a->ntok <= 0) return 0;
  if (dtype == RG_BF16) return launch_binned<__bf16>(*a, nullptr, workspace, workspace_bytes, table_rows, (hipStream_t)stream);
  if (dtype == RG_F32) return launch_binned<float>(*a, nullptr, workspace, workspace_bytes, table_rows, (hipStream_t)stream);
  return rg_set_error_msg(RG_ERR_INVALID, "item_loss_bwd_binned: bad dtype");
}

// ---- training form: rg_item_loss_train (forward, c and dh for gout = 1) + rg_item_loss_scatter_binned (backward) ----
static int train_nit(int k, int d) {           // row batches the kernel keeps in registers; 0 = not supported
  if (!(d == 64 || d == 128 || d == 256)) return 0;
  const int per = (64 / (d / 8)) * RG_U, n = k + 1;
  return n <= per ? 1 : (n <= 2 * per ? 2 : (n <= 4 * per ? 4 : 0));
}
// 1: the register form; 2: the online form (any k; sampled softmax only); 0: neither
extern "C" int rg_item_loss_train_supported(int k, int d) {
  if (!(d == 64 || d == 128 || d == 256)) return 0;
  return train_nit(k, d) != 0 ? 1 : 2;
}

template <typename T>
static int launch_train(const rg_item_loss_args& a, float* coef, hipStream_t s) {
  const int nit = train_nit(a.k, a.d);
  if (!(a.d == 64 || a.d == 128 || a.d == 256)) return rg_set_error_msg(RG_ERR_UNSUPPORTED, "item_loss_train: needs d in {64,128,256}");
  if (!coef || !a.dh || !a.sums) return rg_set_error_msg(RG_ERR_INVALID, "item_loss_train: coef, dh and sums are required");
  long long g = (a.ntok + LW - 1) / LW;
  if (g > 256 * 32) g = 256 * 32;
  dim3 grid((int)g), block(64 * LW);
  if (!nit) {       // more rows than four register batches: the online form
    if (a.mode != RG_LOSS_SAMPLED_CE || !a.aux_tok)
      return rg_set_error_msg(RG_ERR_UNSUPPORTED, "item_loss_train: 1+k beyond 4 row batches takes the online form: sampled softmax only, aux_tok [ntok] required");
    static const int plain = [] { const char* e = getenv("RG_ITEM_ONLINE_PLAIN"); return e ? atoi(e) : 0; }();
    if (plain) {
      if (a.d == 64) hipLaunchKernelGGL((item_loss_train_online_kernel<T, 8>), grid, block, 0, s, a, coef);
      else if (a.d == 128) hipLaunchKernelGGL((item_loss_train_online_kernel<T, 16>), grid, block, 0, s, a, coef);
      else hipLaunchKernelGGL((item_loss_train_online_kernel<T, 32>), grid, block, 0, s, a, coef);
    } else {
      if (a.d == 64) hipLaunchKernelGGL((item_loss_train_online2_kernel<T, 8>), grid, block, 0, s, a, coef);
      else if (a.d == 128) hipLaunchKernelGGL((item_loss_train_online2_kernel<T, 16>), grid, block, 0, s, a, coef);
      else hipLaunchKernelGGL((item_loss_train_online2_kernel<T, 32>), grid, block, 0, s, a, coef);
    }
    RG_CHECK_LAUNCH();
    return 0;
  }
#define RG_TR(LPR)                                                                                           \
  do {                                                                                                       \
    if (nit == 1) hipLaunchKernelGGL((item_loss_train_rows_kernel<T, LPR, 1>), grid, block, 0, s, a, coef);  \
    else if (nit == 2) hipLaunchKernelGGL((item_loss_train_rows_kernel<T, LPR, 2>), grid, block, 0, s, a, coef); \
    else hipLaunchKernelGGL((item_loss_train_rows_kernel<T, LPR, 4>), grid, block, 0, s, a, coef);           \
  } while (0)
  if (a.d == 64) RG_TR(8);
  else if (a.d == 128) RG_TR(16);
  else RG_TR(32);
#undef RG_TR
  RG_CHECK_LAUNCH();
  return 0;
}

// sums[1] must hold the mask count on entry (the divisor of the masked mean; under data parallelism the GLOBAL
// count); sums[0] += sum_t mask*loss_t.  coef [ntok*(1+k)] f32 and dh [ntok,d] come out for an upstream gradient
// of 1; aux_tok / gout / dE are not used.
extern "C" int rg_item_loss_train(const rg_item_loss_args* a, float* coef, int dtype, void* stream) {
  if (!a || a->ntok <= 0) return 0;
  if (dtype == RG_BF16) return launch_train<__bf16>(*a, coef, (hipStream_t)stream);
  if (dtype == RG_F32) return launch_train<float>(*a, coef, (hipStream_t)stream);
  return rg_set_error_msg(RG_ERR_INVALID, "item_loss_train: bad dtype");
}

// dE += gout[0] * (table gradient of the coefficients rg_item_loss_train left in coef): K2..K5 of the binned
// backward; dh is not touched (scale the saved one with rg_scale_dev).
extern "C" int rg_item_loss_scatter_binned(const rg_item_loss_args* a, const float* coef, long long table_rows, void* workspace,
                                           size_t workspace_bytes, int dtype, void* stream) {
  if (!a || a->ntok <= 0) return 0;
  if (!coef || !a->gout) return rg_set_error_msg(RG_ERR_INVALID, "item_loss_scatter_binned: coef and gout are required");
  if (dtype == RG_BF16) return launch_binned<__bf16>(*a, coef, workspace, workspace_bytes, table_rows, (hipStream_t)stream);
  if (dtype == RG_F32) return launch_binned<float>(*a, coef, workspace, workspace_bytes, table_rows, (hipStream_t)stream);
  return rg_set_error_msg(RG_ERR_INVALID, "item_loss_scatter_binned: bad dtype");
}

// ---- embedding backward through the same bins: dE[ids[t]] += dx[t] * mask[t] * dropout multipliers ----
// (elementwise.hip's rg_embed_scatter_bwd sends one atomic row per live position: 235 MB of float atomics per launch at
// the bench shape, bound by the chip-wide atomic rate; here a position is an entry with coefficient mask[t], and the
// rows of a bin are summed in LDS and flushed once)
extern "C" size_t rg_embed_scatter_binned_workspace(long long ntok, int d, long long table_rows) {
  if (bin_log_of(table_rows) != RG_RPB_LOG) return 0;       // wide bins: item loss only (the atomic scatter serves large tables)
  return rg_item_loss_bwd_binned_workspace(ntok, 0, d, table_rows);
}
extern "C" int rg_embed_scatter_bwd_binned(const void* dx, const int64_t* ids, const float* mask, float* dE, long long ntok, int d,
                                           long long table_rows, long long skip_row, float drop_p, unsigned long long seed,
                                           void* workspace, size_t workspace_bytes, int dtype, void* stream) {
  if (ntok <= 0) return 0;
  if (!dx || !ids || !mask || !dE) return rg_set_error_msg(RG_ERR_INVALID, "embed_scatter_bwd_binned: dx, ids, mask and dE are required");
  rg_item_loss_args a = {};
  a.h = dx; a.pos = ids; a.mask = mask; a.dE = dE; a.ntok = ntok; a.d = d; a.k = 0; a.skip_row = skip_row;
  const DropCfg drop = make_drop(drop_p, seed);
  if (dtype == RG_BF16) return launch_binned<__bf16>(a, mask, workspace, workspace_bytes, table_rows, (hipStream_t)stream, &drop);
  if (dtype == RG_F32) return launch_binned<float>(a, mask, workspace, workspace_bytes, table_rows, (hipStream_t)stream, &drop);
  return rg_set_error_msg(RG_ERR_INVALID, "embed_scatter_bwd_binned: bad dtype");
}

// ------------------------------------------------------------------------------------------------
// Ranking evaluation (gan_training.py:58-87 get_scores + the double argsort of evaluation_2 :129-132):
// per user, the score of the held-out target and of C sampled candidates against the last recommender-decoder
// state, and the 0-based rank of the target = number of candidates scoring strictly higher (ties: the reference's
// unstable argsort leaves their order unspecified).  One wave per user, row-group gathers as in the loss kernels;
// the [B, C, d] candidate-embedding tensor, the concatenation and both sorts are never materialised.
// ------------------------------------------------------------------------------------------------
template <typename T, int LPR>
__global__ __launch_bounds__(64 * LW) void rank_scores_kernel(rg_rank_args a) {
  constexpr int G = 64 / LPR;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int gi = lane / LPR, li = lane % LPR;
  const T* __restrict__ H = reinterpret_cast<const T*>(a.h);
  const T* __restrict__ E = reinterpret_cast<const T*>(a.table);
  const int d = a.d, n = a.C + 1;
  for (int b = blockIdx.x * LW + wave; b < a.B; b += gridDim.x * LW) {
    float h[8];
    load8(h, H + (size_t)b * d + 8 * li);
    const long long tgt = a.target[b];
    float s0;
    {
      float e[8];
      load8(e, E + (size_t)tgt * d + 8 * li);
      float dot = 0.f;
#pragma unroll
      for (int j = 0; j < 8; ++j) dot += e[j] * h[j];
      s0 = __shfl(group_sum<LPR>(dot), 0);
    }
    if (lane == 0 && a.scores) a.scores[(size_t)b * n] = s0;
    int higher = 0;
    for (int i0 = 0; i0 < a.C; i0 += G * RG_U) {
      long long item[RG_U];
      float e[RG_U][8];
#pragma unroll
      for (int u = 0; u < RG_U; ++u) {
        const int idx = i0 + u * G + gi;
        item[u] = idx < a.C ? a.cand[(size_t)b * a.C + idx] : tgt;
      }
#pragma unroll
      for (int u = 0; u < RG_U; ++u) load8(e[u], E + (size_t)item[u] * d + 8 * li);
#pragma unroll
      for (int u = 0; u < RG_U; ++u) {
        const int idx = i0 + u * G + gi;
        float dot = 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) dot += e[u][j] * h[j];
        dot = group_sum<LPR>(dot);
        if (idx < a.C) {
          if (li == 0) {
            higher += dot > s0;
            if (a.scores) a.scores[(size_t)b * n + 1 + idx] = dot;
          }
        }
      }
    }
    higher = (int)wave_sum((float)higher);          // exact below 2^24 candidates
    if (lane == 0 && a.rank) a.rank[b] = higher;
  }
}

extern "C" int rg_rank_scores(const rg_rank_args* a, int dtype, void* stream) {
  if (!a || a->B <= 0) return 0;
  if (a->C < 0 || a->C >= (1 << 24)) return rg_set_error_msg(RG_ERR_INVALID, "rank_scores: bad candidate count");
  if (!(a->d == 32 || a->d == 64 || a->d == 128 || a->d == 256))
    return rg_set_error_msg(RG_ERR_UNSUPPORTED, "rank_scores: d must be 32, 64, 128 or 256");
  hipStream_t s = (hipStream_t)stream;
  int g = (a->B + LW - 1) / LW;
  if (g > 256 * 32) g = 256 * 32;
#define RG_RK(T, LPR) hipLaunchKernelGGL((rank_scores_kernel<T, LPR>), dim3(g), dim3(64 * LW), 0, s, *a)
#define RG_RK_T(T)                  \
  do {                              \
    if (a->d == 32) RG_RK(T, 4);    \
    else if (a->d == 64) RG_RK(T, 8);    \
    else if (a->d == 128) RG_RK(T, 16); \
    else RG_RK(T, 32);              \
  } while (0)
  if (dtype == RG_BF16) RG_RK_T(__bf16);
  else if (dtype == RG_F32) RG_RK_T(float);
  else return rg_set_error_msg(RG_ERR_INVALID, "rank_scores: bad dtype");
#undef RG_RK_T
#undef RG_RK
  RG_CHECK_LAUNCH();
  return 0;
}

extern "C" int rg_item_loss_fwd(const rg_item_loss_args* a, int dtype, void* stream) {
  if (!a || a->ntok <= 0) return 0;
  if (dtype == RG_BF16) return launch<__bf16>(*a, false, (hipStream_t)stream);
  if (dtype == RG_F32) return launch<float>(*a, false, (hipStream_t)stream);
  return rg_set_error_msg(RG_ERR_INVALID, "item_loss_fwd: bad dtype");
}
extern "C" int rg_item_loss_bwd(const rg_item_loss_args* a, int dtype, void* stream) {
  if (!a || a->ntok <= 0) return 0;
  if (dtype == RG_BF16) return launch<__bf16>(*a, true, (hipStream_t)stream);
  if (dtype == RG_F32) return launch<float>(*a, true, (hipStream_t)stream);
  return rg_set_error_msg(RG_ERR_INVALID, "item_loss_bwd: bad dtype");
}
