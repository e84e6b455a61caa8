// Single-query attention for the LAST encoder layer.
//
// Every consumer of EncoderM's output on the hot path reads only position L-1
// (enc_outputs[:, -1, :]: AutoEnc4Rec_cross.py:122,154; AutoEnc4Rec.py:188; gan_training.py:157-161),
// so in the last layer only K and V are needed for all positions; Q, the softmax, the output
// projection and the FFN are needed for ONE query per sequence.  Results are identical to row L-1 of
// the full ScaledDotProductAttention (Transformer/transformer.py:119-129), incl. the -1e9 replace fill.
// One wave per (sequence, head); lanes stride the keys; f32 math.
#include "rg_common.hip.h"
#include "../../include/recguru_hip.h"

#define DK 32
#define NEG_FILL (-1e9f)
#define MASK_BIG (-1e30f)
#define MAXKPL 8       // keys per lane: L <= 512

template <typename T>
__device__ __forceinline__ void load_row32(float* o, const T* p) {
#pragma unroll
  for (int c = 0; c < 4; ++c) load8(o + 8 * c, p + 8 * c);
}
template <typename T>
__device__ __forceinline__ void store_row32(T* p, const float* v) {
#pragma unroll
  for (int c = 0; c < 4; ++c) store8(p + 8 * c, v + 8 * c);
}

// a 32-element row as it sits in memory: loads of several rows are issued back to back and converted where used
// (converting at the load made the compiler wait for every row before issuing the next)
template <typename T> struct Row32 {
  Frag<T> c[4];
  __device__ __forceinline__ void load(const T* p) {
#pragma unroll
    for (int k = 0; k < 4; ++k) load_frag(c[k], p + 8 * k);
  }
  __device__ __forceinline__ float at(int j) const { return (float)c[j >> 3].v[j & 7]; }
};

// q . bf16(bias row): what a key of the padded prefix scores (same summation order as the per-key dot products)
template <typename T>
__device__ __forceinline__ float lastq_bias_dot(const float* q, const float* __restrict__ bias32) {
  float d = 0.f;
#pragma unroll
  for (int j = 0; j < 32; ++j) d += q[j] * (float)(T)bias32[j];
  return d;
}

// scores of this lane's keys (raw dot * scale, replaced / -inf where masked), returns the row max.  KPL = keys per
// lane (64 KPL >= L): a template parameter, so that the loop is straight-line code with clamped, unconditional loads
template <typename T, int KPL>
__device__ __forceinline__ float lastq_scores(float (&s)[KPL], const float* q, const T* __restrict__ kv, int ldkv, int koff,
                                              const int64_t* __restrict__ ids, int64_t pad_value, int L, int lane, float scale,
                                              int first, float dzero) {
  // keys before `first` (the sequence's padded prefix, x_masked contract: their K rows all equal the bias row) are not
  // fetched: their loads are pointed at row `first` (one hot line for all of them) and their dot product is dzero = q . bk
  Row32<T> kr[KPL];
  int64_t id[KPL];
#pragma unroll
  for (int i = 0; i < KPL; ++i) {
    const int kc = max(min(lane + 64 * i, L - 1), first);
    kr[i].load(kv + (size_t)kc * ldkv + koff);
    id[i] = ids[min(lane + 64 * i, L - 1)];
  }
  float mx = -INFINITY;
#pragma unroll
  for (int i = 0; i < KPL; ++i) {
    float d = 0.f;
#pragma unroll
    for (int j = 0; j < 32; ++j) d += q[j] * kr[i].at(j);
    d = (lane + 64 * i < first) ? dzero : d;
    const float v = (lane + 64 * i < L) ? ((id[i] == pad_value) ? MASK_BIG : d * scale) : -INFINITY;
    s[i] = v;
    mx = fmaxf(mx, v);
  }
  return wave_max(mx);
}

template <typename T, int KPL>
__global__ __launch_bounds__(256) void attn_lastq_fwd_kernel(const T* __restrict__ qlast, const T* __restrict__ kv,
                                                             const int64_t* __restrict__ key_ids, int64_t pad_value,
                                                             T* __restrict__ ctx, int B, int L, int H, float scale, DropCfg drop,
                                                             const float* __restrict__ bkv, const int* __restrict__ first_live) {
  const int lane = threadIdx.x & 63;
  const int wg = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (wg >= B * H) return;
  const int b = wg / H, h = wg % H, P = H * DK;
  float q[32];
  load_row32(q, qlast + (size_t)b * P + h * DK);
  const T* kvb = kv + (size_t)b * L * 2 * P;
  const bool fold = bkv != nullptr && first_live != nullptr;
  const int first = fold ? min(first_live[b], L - 1) : 0;
  const float dzero = fold ? lastq_bias_dot<T>(q, bkv + h * DK) : 0.f;
  float s[KPL];
  const float mx = lastq_scores<T, KPL>(s, q, kvb, 2 * P, h * DK, key_ids + (size_t)b * L, pad_value, L, lane, scale, first, dzero);
  float sum = 0.f;
#pragma unroll
  for (int i = 0; i < KPL; ++i) { s[i] = __expf(s[i] - mx); sum += s[i]; }
  sum = wave_sum(sum);
  const float inv = 1.f / sum;
  const unsigned int dbase = (((unsigned int)b * H + h) * L + (L - 1)) * rg_lpad(L);   // same index space as the full kernel
  if (drop.thresh) {
#pragma unroll
    for (int i = 0; i < KPL; ++i) s[i] *= rg_keep(drop, dbase + lane + 64 * i);
  }
  float o[32];
#pragma unroll
  for (int j = 0; j < 32; ++j) o[j] = 0.f;
  Row32<T> vr[KPL];
#pragma unroll
  for (int i = 0; i < KPL; ++i) vr[i].load(kvb + (size_t)max(min(lane + 64 * i, L - 1), first) * 2 * P + P + h * DK);
  float pz = 0.f;                                   // probability mass of this lane's prefix keys: they all carry V = bv
#pragma unroll
  for (int i = 0; i < KPL; ++i) {
    const bool pre = lane + 64 * i < first;
    const float p0 = (lane + 64 * i < L) ? s[i] * inv : 0.f;      // keys past L: exp(-inf) = 0 already; the select guards NaN
    const float p = pre ? 0.f : p0;
    pz += pre ? p0 : 0.f;
#pragma unroll
    for (int j = 0; j < 32; ++j) o[j] += p * vr[i].at(j);
  }
  if (first > 0) {
#pragma unroll
    for (int j = 0; j < 32; ++j) o[j] += pz * (float)(T)bkv[P + h * DK + j];
  }
#pragma unroll
  for (int j = 0; j < 32; ++j) o[j] = wave_sum(o[j]);
  if (lane == 0) store_row32(ctx + (size_t)b * P + h * DK, o);
}

template <typename T, int KPL>
__global__ __launch_bounds__(256) void attn_lastq_bwd_kernel(const T* __restrict__ qlast, const T* __restrict__ kv,
                                                             const T* __restrict__ dctx, const int64_t* __restrict__ key_ids,
                                                             int64_t pad_value, T* __restrict__ dq, T* __restrict__ dkv,
                                                             int B, int L, int H, float scale, DropCfg drop,
                                                             const float* __restrict__ bkv, const int* __restrict__ first_live) {
  const int lane = threadIdx.x & 63;
  const int wg = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (wg >= B * H) return;
  const int b = wg / H, h = wg % H, P = H * DK;
  const unsigned int dbase = (((unsigned int)b * H + h) * L + (L - 1)) * rg_lpad(L);
  const bool fold = bkv != nullptr && first_live != nullptr;
  const int first = fold ? min(first_live[b], L - 1) : 0;
  float q[32], g[32];
  load_row32(q, qlast + (size_t)b * P + h * DK);
  load_row32(g, dctx + (size_t)b * P + h * DK);
  const T* kvb = kv + (size_t)b * L * 2 * P;
  T* dkvb = dkv + (size_t)b * L * 2 * P;
  const int64_t* ids = key_ids + (size_t)b * L;
  const float dzero = fold ? lastq_bias_dot<T>(q, bkv + h * DK) : 0.f;       // q . bk
  const float gzero = fold ? lastq_bias_dot<T>(g, bkv + P + h * DK) : 0.f;   // dO . bv
  float s[KPL];
  const float mx = lastq_scores<T, KPL>(s, q, kvb, 2 * P, h * DK, ids, pad_value, L, lane, scale, first, dzero);
  const bool full = mx < 0.5f * MASK_BIG;          // every key replaced: uniform row, no gradient to q / k (Q3)
  float sum = 0.f;
#pragma unroll
  for (int i = 0; i < KPL; ++i) { s[i] = __expf(s[i] - mx); sum += s[i]; }
  sum = wave_sum(sum);
  const float inv = 1.f / sum;
  float dp[KPL];
  float delta = 0.f;
  Row32<T> vr[KPL], kr[KPL];
  int64_t idk[KPL];
#pragma unroll
  for (int i = 0; i < KPL; ++i) {               // all V rows, then all K rows and ids: one batch of loads in flight
    const int kc = max(min(lane + 64 * i, L - 1), first);       // prefix keys: one hot line, data not used (see lastq_scores)
    vr[i].load(kvb + (size_t)kc * 2 * P + P + h * DK);
  }
#pragma unroll
  for (int i = 0; i < KPL; ++i) {
    const int kc = max(min(lane + 64 * i, L - 1), first);
    kr[i].load(kvb + (size_t)kc * 2 * P + h * DK);
    idk[i] = ids[min(lane + 64 * i, L - 1)];
  }
#pragma unroll
  for (int i = 0; i < KPL; ++i) {
    const int key = lane + 64 * i;
    dp[i] = 0.f;
    s[i] *= inv;
    if (key < L) {
      float dv[32];
      const float ks = drop.thresh ? rg_keep(drop, dbase + key) : 1.f;
      float d = 0.f;
#pragma unroll
      for (int j = 0; j < 32; ++j) { d += g[j] * vr[i].at(j); dv[j] = s[i] * ks * g[j]; }
      d = key < first ? gzero : d;
      dp[i] = d * ks;            // d(loss)/d(undropped probability)
      delta += s[i] * dp[i];
      store_row32(dkvb + (size_t)key * 2 * P + P + h * DK, dv);
    }
  }
  delta = wave_sum(delta);
  float dqa[32];
#pragma unroll
  for (int j = 0; j < 32; ++j) dqa[j] = 0.f;
  float dsz = 0.f;                                  // sum of dS over this lane's prefix keys: they all carry K = bk
#pragma unroll
  for (int i = 0; i < KPL; ++i) {
    const int key = lane + 64 * i;
    if (key < L) {
      const bool masked = full || idk[i] == pad_value;
      const float ds = masked ? 0.f : s[i] * (dp[i] - delta) * scale;
      const float dsk = key < first ? 0.f : ds;
      dsz += key < first ? ds : 0.f;
      float dk[32];
#pragma unroll
      for (int j = 0; j < 32; ++j) { dqa[j] += dsk * kr[i].at(j); dk[j] = ds * q[j]; }
      store_row32(dkvb + (size_t)key * 2 * P + h * DK, dk);
    }
  }
  if (first > 0) {
#pragma unroll
    for (int j = 0; j < 32; ++j) dqa[j] += dsz * (float)(T)bkv[h * DK + j];
  }
#pragma unroll
  for (int j = 0; j < 32; ++j) dqa[j] = wave_sum(dqa[j]);
  if (lane == 0) store_row32(dq + (size_t)b * P + h * DK, dqa);
}

extern "C" int rg_attn_lastq_fwd(const void* qlast, const void* kv, const int64_t* key_ids, int64_t pad_value, void* ctx,
                                 int B, int L, int H, float scale, float drop_p, unsigned long long seed, int dtype, void* stream,
                                 const float* bkv, const int* first_live) {
  if (B <= 0) return 0;
  const DropCfg drop = make_drop(drop_p, seed);
  if (L > 64 * MAXKPL) return rg_set_error_msg(RG_ERR_UNSUPPORTED, "attn_lastq: L > 512");
  hipStream_t s = (hipStream_t)stream;
  dim3 grid((B * H + 3) / 4), block(256);
#define RG_LQF(T, K) hipLaunchKernelGGL((attn_lastq_fwd_kernel<T, K>), grid, block, 0, s, (const T*)qlast, (const T*)kv, key_ids, pad_value, (T*)ctx, B, L, H, scale, drop, bkv, first_live)
#define RG_LQF_T(T)                 \
  do {                              \
    if (L <= 64) RG_LQF(T, 1);      \
    else if (L <= 128) RG_LQF(T, 2);\
    else if (L <= 256) RG_LQF(T, 4);\
    else RG_LQF(T, 8);              \
  } while (0)
  if (dtype == RG_BF16) RG_LQF_T(__bf16);
  else if (dtype == RG_F32) RG_LQF_T(float);
  else return rg_set_error_msg(RG_ERR_INVALID, "attn_lastq_fwd: bad dtype");
#undef RG_LQF_T
#undef RG_LQF
  RG_CHECK_LAUNCH();
  return 0;
}

extern "C" int rg_attn_lastq_bwd(const void* qlast, const void* kv, const void* dctx, const int64_t* key_ids, int64_t pad_value,
                                 void* dq, void* dkv, int B, int L, int H, float scale, float drop_p, unsigned long long seed,
                                 int dtype, void* stream, const float* bkv, const int* first_live) {
  if (B <= 0) return 0;
  const DropCfg drop = make_drop(drop_p, seed);
  if (L > 64 * MAXKPL) return rg_set_error_msg(RG_ERR_UNSUPPORTED, "attn_lastq: L > 512");
  hipStream_t s = (hipStream_t)stream;
  dim3 grid((B * H + 3) / 4), block(256);
#define RG_LQB(T, K) hipLaunchKernelGGL((attn_lastq_bwd_kernel<T, K>), grid, block, 0, s, (const T*)qlast, (const T*)kv, (const T*)dctx, key_ids, pad_value, (T*)dq, (T*)dkv, B, L, H, scale, drop, bkv, first_live)
#define RG_LQB_T(T)                 \
  do {                              \
    if (L <= 64) RG_LQB(T, 1);      \
    else if (L <= 128) RG_LQB(T, 2);\
    else if (L <= 256) RG_LQB(T, 4);\
    else RG_LQB(T, 8);              \
  } while (0)
  if (dtype == RG_BF16) RG_LQB_T(__bf16);
  else if (dtype == RG_F32) RG_LQB_T(float);
  else return rg_set_error_msg(RG_ERR_INVALID, "attn_lastq_bwd: bad dtype");
#undef RG_LQB_T
#undef RG_LQB
  RG_CHECK_LAUNCH();
  return 0;
}
