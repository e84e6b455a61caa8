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
#include "rg_common.cuh"
#include "../../include/recguru_hip.h"

#define LW 4  // waves per block

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
      aux = l0 - ns / (float)k;
      loss = -__logf(1.f / (1.f + __expf(-aux)));
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
    if (s1 != 0.f) { atomicAdd(a.sums, s0); atomicAdd(a.sums + 1, s1); }
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
      const float sg = 1.f / (1.f + __expf(aux));  // sigmoid(-x), BPR only
      for (int j = -1; j < k; ++j) {
        const long long item = j < 0 ? a.pos[t] : a.neg[t * k + j];
        const float lj = row_dot<T, NPL>(E + (size_t)item * d, h, lane, d, row);
        float c;
        if (a.mode == RG_LOSS_SAMPLED_CE) c = (__expf(lj - aux) - (j < 0 ? 1.f : 0.f)) * w;
        else c = (j < 0 ? -sg : sg / (float)k) * w;
#pragma unroll
        for (int q = 0; q < NPL; ++q) {
          const int e = lane + 64 * q;
          dh[q] += c * row[q];
          if (e < d && item != a.skip_row) atomicAdd(a.dE + (size_t)item * d + e, c * h[q]);
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
      aux = l0 - ns / (float)k;
      loss = -__logf(1.f / (1.f + __expf(-aux)));
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
    if (s1 != 0.f) { atomicAdd(a.sums, s0); atomicAdd(a.sums + 1, s1); }
  }
}

template <typename T, int LPR>
__global__ __launch_bounds__(64 * LW) void item_loss_bwd_rows_kernel(rg_item_loss_args a) {
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
      const float sg = 1.f / (1.f + __expf(aux));  // sigmoid(-x), BPR only
      float h[8], ht[8];
      load8(h, H + (size_t)t * d + 8 * li);
#pragma unroll
      for (int j = 0; j < 8; ++j) ht[j] = (float)H[(size_t)t * d + li + LPR * j];
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
          if (a.mode == RG_LOSS_SAMPLED_CE) c = (__expf(dot - aux) - (idx == 0 ? 1.f : 0.f)) * w;
          else c = (idx == 0 ? -sg : sg / (float)k) * w;
          if (idx >= n) c = 0.f;
#pragma unroll
          for (int j = 0; j < 8; ++j) dh[j] += c * e[u][j];
          if (idx < n && item[u] != a.skip_row) {
            float* __restrict__ dst = a.dE + (size_t)item[u] * d + li;
#pragma unroll
            for (int j = 0; j < 8; ++j) atomicAdd(dst + LPR * j, c * ht[j]);
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

template <typename T>
static int launch(const rg_item_loss_args& a, bool bwd, hipStream_t s) {
  long long g = (a.ntok + LW - 1) / LW;
  if (g > 256 * 32) g = 256 * 32;
  dim3 grid((int)g), block(64 * LW);
#define RG_R(LPR)                                                                             \
  do {                                                                                        \
    if (bwd) hipLaunchKernelGGL((item_loss_bwd_rows_kernel<T, LPR>), grid, block, 0, s, a);   \
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
