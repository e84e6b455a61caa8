// Single-query attention for the LAST encoder layer, computed from the layer INPUT (no K / V projection).
//
// Same result as attention_lastq.hip on K = x WK^T + bK, V = x WV^T + bV (row L-1 of ScaledDotProductAttention,
// Transformer/transformer.py:119-129, masked_fill -1e9 / all-masked rows uniform, attention-map dropout), but with
// the two projections absorbed into the single query and the single output:
//   score_j   = q_h . (WK_h x_j + bK_h)           = x_j . (WK_h^T q_h)  +  q_h . bK_h
//   context_h = sum_j p_j (WV_h x_j + bV_h)       = WV_h (sum_j p_j x_j) + bV_h sum_j p_j
// so that per sequence the work is  [L,128] x [128, 4 heads]  (scores) and  [4, L] x [L,128]  (weighted row sums)
// instead of the [L,128] x [128,256] K/V product: 28x fewer flops, x is read once (256 B per position) and the
// [B,L,256] K/V tensor (written, then read back by the single-query kernel) does not exist.  The backward mirrors it:
//   dxbar_h = WV_h^T dctx_h;  dp_j = x_j . dxbar_h + bV_h . dctx_h;  ds = softmax backward;
//   dq'_h = sum_j ds_j x_j;   dx_j = sum_h p_hj dxbar_h + ds_hj q'_h;  dq_h = WK_h dq'_h;
//   dWV_h = sum_b dctx_h (x) xbar_h,  dWK_h = sum_b q_h (x) dq'_h   (two small products over B*H rows, made by rg_gemm_tn
//   from the operands this kernel writes),  dbV_h = sum_b dctx_h * sum_j p_j,  dbK = 0 exactly.
//
// One workgroup (4 waves) per sequence, persistent over sequences; wave h owns head h and keeps WK_h and WV_h
// (32 x 128 bf16 each) in registers, each lane the two features (2*lane, 2*lane+1) of all 32 rows.  x rows are staged
// once in LDS (XOR-swizzled 16-byte chunks: conflict-free both for the MFMA row fragments and for the feature-pair
// reads).  Scores (and in the backward x . dxbar in the same pass) are ONE 16x16x32 MFMA column block: the B operand's
// 16 columns hold q'_h split into bf16 high + low parts (4 + 4 columns; backward: dxbar_h in the other 8), so the f32
// vectors lose nothing to the bf16 operand format.  bf16 tier, d_model = H*32 = 128, L <= 256.
#include "rg_common.hip.h"
#include "../../include/recguru_hip.h"

#define LX_D 128
#define LX_H 4
#define LX_DK 32
#define LX_QLD 136            // B-operand row stride (bf16): 272 B, 16-byte chunks rotate through the banks
#define LX_MASK_BIG (-1e30f)
#define LX_KPL 4              // keys per lane in the softmax stage: L <= 256

namespace {

__device__ __forceinline__ float bf_lo(unsigned int w) { return __uint_as_float(w << 16); }
__device__ __forceinline__ float bf_hi(unsigned int w) { return __uint_as_float(w & 0xffff0000u); }
__device__ __forceinline__ unsigned int pack_bf(float a, float b) {
  union { __bf16 h[2]; unsigned int u; } p;
  p.h[0] = (__bf16)a; p.h[1] = (__bf16)b;
  return p.u;
}
__device__ __forceinline__ float bcast(float v, int l) { return __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(v), l)); }

// sum over the 64 lanes of 32 per-lane values; lane l ends up with the total of value index l >> 1 (5 halving
// exchanges of 16, 8, 4, 2, 1 values and a last pair sum: 32 cross-lane moves instead of 32 x 6)
template <int W>
__device__ __forceinline__ void halve(float (&t)[32], int lane) {
  const bool up = lane & (2 * W);
#pragma unroll
  for (int i = 0; i < W; ++i) {
    const float send = up ? t[i] : t[W + i];
    const float keep = up ? t[W + i] : t[i];
    t[i] = keep + __shfl_xor(send, 2 * W);
  }
}
__device__ __forceinline__ float reduce32(float (&t)[32], int lane) {
  halve<16>(t, lane); halve<8>(t, lane); halve<4>(t, lane); halve<2>(t, lane); halve<1>(t, lane);
  return t[0] + __shfl_xor(t[0], 1);
}

struct Smem {
  unsigned int* xs;    // [LT*16 rows][64 dwords], chunk c of row r at chunk slot c ^ (r & 15)
  __bf16* qs;          // [16][LX_QLD]  B operand columns
  float* ss;           // [4][SLD]      scores -> dropped probabilities
  float* dp;           // [4][SLD]      backward: x . dxbar
  float* cf;           // [LT*16][8]    backward: (p~, ds) per head
  float* vec;          // [8][128]      backward: dxbar_h, q'_h (f32)
};

// x rows [rs, SLD) of sequence b into LDS by LDS-DMA (no staging registers, every row in flight at once): one wave
// instruction brings 4 rows; the chunk swizzle is applied on the SOURCE side (lane (row rr, slot s) fetches chunk
// s ^ (row & 15)).  Rows >= L are copies of row L-1 (finite; their scores are discarded and their weights are zero).
__device__ __forceinline__ void stage_x(const Smem& sm, const __bf16* __restrict__ xb, int rs, int L, int SLD, int wave, int lane) {
  const int rr = lane >> 4, sl = lane & 15;
  for (int r0 = rs + 4 * wave; r0 < SLD; r0 += 16) {
    const int r = r0 + rr;
    const __bf16* src = xb + (size_t)min(r, L - 1) * LX_D + 8 * (sl ^ (r & 15));
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                     (__attribute__((address_space(3))) void*)(sm.xs + (r0 << 6)), 16, 0, 0);
  }
}
__device__ __forceinline__ void stage_wait() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

__device__ __forceinline__ unsigned int x_pair(const Smem& sm, int j, int lane) {
  return sm.xs[(j << 6) + ((((lane >> 2) ^ (j & 15))) << 2) + (lane & 3)];
}

// the f32 pair (a0, a1) of features (2*lane, 2*lane+1) as bf16 high and low parts into B-operand columns ch, cl
__device__ __forceinline__ void put_cols(const Smem& sm, int ch, int cl, float a0, float a1, int lane) {
  const float h0 = (float)(__bf16)a0, h1 = (float)(__bf16)a1;
  reinterpret_cast<unsigned int*>(sm.qs + ch * LX_QLD)[lane] = pack_bf(h0, h1);
  reinterpret_cast<unsigned int*>(sm.qs + cl * LX_QLD)[lane] = pack_bf(a0 - h0, a1 - h1);
}

// [16 rows of tile t] x [16 operand columns]: lane (n = lane & 15, g = lane >> 4) gets rows 4g..4g+3 of column n
__device__ __forceinline__ f32x4 tile_product(const Smem& sm, int t, const Frag<__bf16> (&bq)[4], int lane) {
  const int i = lane & 15, g = lane >> 4;
  f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
  const unsigned int* row = sm.xs + ((16 * t + i) << 6);
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) {
    Frag<__bf16> a;
    a.v = *reinterpret_cast<const bf16x8_t*>(row + (((ks * 4 + g) ^ i) << 2));
    mma(a, bq[ks], acc);
  }
  return acc;
}

struct Head {                       // what wave h keeps for the whole launch
  unsigned int wk[32], wv[32];      // rows h*32 + c of WK / WV, features (2*lane, 2*lane+1)
  float bk, bv;                     // lane c < 32: bK / bV [h*32 + c]
};
__device__ __forceinline__ void load_head(Head& hd, const rg_lastq_x_args& a, int h, int lane) {
  const unsigned int* wk = reinterpret_cast<const unsigned int*>(a.wk) + (size_t)h * LX_DK * (LX_D / 2) + lane;
  const unsigned int* wv = reinterpret_cast<const unsigned int*>(a.wv) + (size_t)h * LX_DK * (LX_D / 2) + lane;
#pragma unroll
  for (int c = 0; c < 32; ++c) { hd.wk[c] = wk[c * (LX_D / 2)]; hd.wv[c] = wv[c * (LX_D / 2)]; }
  hd.bk = lane < 32 ? a.bk[h * LX_DK + lane] : 0.f;
  hd.bv = lane < 32 ? a.bv[h * LX_DK + lane] : 0.f;
}

// The packed rows are declared "possibly changed" once per sequence: without it the compiler hoists the 128 unpacked
// floats per matrix out of the sequence loop and spills.
__device__ __forceinline__ void pin_head(Head& hd) {
#pragma unroll
  for (int c = 0; c < 32; ++c) { asm volatile("" : "+v"(hd.wk[c])); asm volatile("" : "+v"(hd.wv[c])); }
}

// out pair = sum_c v[c] * W[c][pair], v[c] = lane c of `vl`
__device__ __forceinline__ void vec_times_rows(const unsigned int (&w)[32], float vl, float& o0, float& o1) {
  o0 = 0.f; o1 = 0.f;
#pragma unroll
  for (int c = 0; c < 32; ++c) {
    const float vc = bcast(vl, c);
    o0 += vc * bf_lo(w[c]);
    o1 += vc * bf_hi(w[c]);
  }
}
// lane l gets sum_e W[l >> 1][e] * v[e], v given as the pair (v0, v1) per lane
__device__ __forceinline__ float rows_times_vec(const unsigned int (&w)[32], float v0, float v1, int lane) {
  float t[32];
#pragma unroll
  for (int c = 0; c < 32; ++c) t[c] = bf_lo(w[c]) * v0 + bf_hi(w[c]) * v1;
  return reduce32(t, lane);
}

// softmax stage of wave h: keys lane + 64 i.  Leaves p (undropped), keep multiplier and the sum of p~.
__device__ __forceinline__ void softmax_keys(const Smem& sm, int SLD, int h, int lane, int L, int rs, float qb,
                                             const int64_t* __restrict__ ids, int64_t pad_value, const DropCfg& drop,
                                             unsigned int dbase, float (&p)[LX_KPL], float (&kp)[LX_KPL], bool (&msk)[LX_KPL],
                                             bool& full) {
  float s[LX_KPL];
  float mx = -INFINITY;
#pragma unroll
  for (int i = 0; i < LX_KPL; ++i) {
    const int j = lane + 64 * i;
    const int64_t id = ids[min(j, L - 1)];
    const float raw = (j >= rs && j < L ? sm.ss[h * SLD + j] : 0.f) + qb;
    msk[i] = id == pad_value;
    s[i] = j < L ? (msk[i] ? LX_MASK_BIG : raw) : -INFINITY;
    mx = fmaxf(mx, s[i]);
  }
  mx = wave_max(mx);
  full = mx < 0.5f * LX_MASK_BIG;
  float sum = 0.f;
#pragma unroll
  for (int i = 0; i < LX_KPL; ++i) { s[i] = __expf(s[i] - mx); sum += s[i]; }
  sum = wave_sum(sum);
  const float inv = 1.f / sum;
#pragma unroll
  for (int i = 0; i < LX_KPL; ++i) {
    const int j = lane + 64 * i;
    p[i] = j < L ? s[i] * inv : 0.f;
    kp[i] = drop.thresh ? rg_keep(drop, dbase + j) : 1.f;
  }
}

}  // namespace

__global__ __launch_bounds__(256, 2) void attn_lastq_x_fwd_kernel(rg_lastq_x_args a) {
  extern __shared__ __align__(16) unsigned char lx_smem[];
  const int tid = threadIdx.x, lane = tid & 63, h = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int L = a.L, LT = (L + 15) >> 4, SLD = LT * 16;
  Smem sm;
  sm.xs = reinterpret_cast<unsigned int*>(lx_smem);
  sm.qs = reinterpret_cast<__bf16*>(sm.xs + SLD * 64);
  sm.ss = reinterpret_cast<float*>(sm.qs + 16 * LX_QLD);
  for (int i = tid; i < 16 * LX_QLD / 2; i += 256) reinterpret_cast<unsigned int*>(sm.qs)[i] = 0u;
  Head hd;
  load_head(hd, a, h, lane);
  const DropCfg drop = make_drop(a.drop_p, a.seed);
  const __bf16* __restrict__ X = reinterpret_cast<const __bf16*>(a.x);
  const __bf16* __restrict__ Q = reinterpret_cast<const __bf16*>(a.qlast);
  __bf16* __restrict__ C = reinterpret_cast<__bf16*>(a.ctx);
  for (int b = blockIdx.x; b < a.B; b += gridDim.x) {
    const int first = a.first_live ? min(a.first_live[b], L - 1) : 0;
    const int rs = first & ~15;
    pin_head(hd);
    __syncthreads();                                            // the previous sequence is done with the LDS
    stage_x(sm, X + (size_t)b * L * LX_D, rs, L, SLD, h, lane);
    const float ql = lane < 32 ? (float)Q[(size_t)b * LX_D + h * LX_DK + lane] : 0.f;
    float q0, q1;
    vec_times_rows(hd.wk, ql, q0, q1);
    q0 *= a.scale; q1 *= a.scale;
    const float qb = wave_sum(ql * hd.bk) * a.scale;
    put_cols(sm, h, 4 + h, q0, q1, lane);
    stage_wait();
    __syncthreads();
    {                                                           // scores of all heads, tiles dealt to the waves
      Frag<__bf16> bq[4];
      const int n = lane & 15, g = lane >> 4;
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) bq[ks].v = *reinterpret_cast<const bf16x8_t*>(sm.qs + n * LX_QLD + ks * 32 + 8 * g);
#pragma unroll 1
      for (int t = (rs >> 4) + h; t < LT; t += 4) {
        const f32x4 acc = tile_product(sm, t, bq, lane);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float v = acc[r] + __shfl_down(acc[r], 4);
          if (n < 4) sm.ss[n * SLD + 16 * t + 4 * g + r] = v;
        }
      }
    }
    __syncthreads();
    float p[LX_KPL], kp[LX_KPL];
    bool msk[LX_KPL], full;
    const unsigned int dbase = (((unsigned int)b * LX_H + h) * L + (L - 1)) * rg_lpad(L);
    softmax_keys(sm, SLD, h, lane, L, rs, qb, a.key_ids + (size_t)b * L, a.pad_value, drop, dbase, p, kp, msk, full);
    float sp = 0.f;
#pragma unroll
    for (int i = 0; i < LX_KPL; ++i) {
      const int j = lane + 64 * i;
      const float pd = p[i] * kp[i];
      sp += pd;
      if (j < SLD) sm.ss[h * SLD + j] = pd;                    // this wave's own row: no barrier needed
    }
    sp = wave_sum(sp);
    float a0 = 0.f, a1 = 0.f;
#pragma unroll 2
    for (int j = rs; j < SLD; j += 4) {
      const float4 pw = *reinterpret_cast<const float4*>(sm.ss + h * SLD + j);
      const float pv[4] = {pw.x, pw.y, pw.z, pw.w};
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const unsigned int xw = x_pair(sm, j + u, lane);
        a0 += pv[u] * bf_lo(xw);
        a1 += pv[u] * bf_hi(xw);
      }
    }
    const float o = rows_times_vec(hd.wv, a0, a1, lane);
    const float bvl = __shfl(hd.bv, lane >> 1);
    if (!(lane & 1)) C[(size_t)b * LX_D + h * LX_DK + (lane >> 1)] = (__bf16)(o + bvl * sp);
  }
}

__global__ __launch_bounds__(256, 2) void attn_lastq_x_bwd_kernel(rg_lastq_x_args a) {
  extern __shared__ __align__(16) unsigned char lx_smem[];
  const int tid = threadIdx.x, lane = tid & 63, h = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int L = a.L, LT = (L + 15) >> 4, SLD = LT * 16;
  Smem sm;
  sm.xs = reinterpret_cast<unsigned int*>(lx_smem);
  sm.qs = reinterpret_cast<__bf16*>(sm.xs + SLD * 64);
  sm.ss = reinterpret_cast<float*>(sm.qs + 16 * LX_QLD);
  sm.dp = sm.ss + 4 * SLD;
  sm.cf = sm.dp + 4 * SLD;
  sm.vec = sm.cf + SLD * 8;
  Head hd;
  load_head(hd, a, h, lane);
  const DropCfg drop = make_drop(a.drop_p, a.seed);
  const __bf16* __restrict__ X = reinterpret_cast<const __bf16*>(a.x);
  const __bf16* __restrict__ Q = reinterpret_cast<const __bf16*>(a.qlast);
  const __bf16* __restrict__ G = reinterpret_cast<const __bf16*>(a.dctx);
  unsigned int* __restrict__ DX = reinterpret_cast<unsigned int*>(a.dx);
  __bf16* __restrict__ DQ = reinterpret_cast<__bf16*>(a.dq);
  unsigned int* __restrict__ YV = reinterpret_cast<unsigned int*>(a.ym_v);
  unsigned int* __restrict__ YQ = reinterpret_cast<unsigned int*>(a.ym_q);
  unsigned int* __restrict__ XB = reinterpret_cast<unsigned int*>(a.xbar);
  unsigned int* __restrict__ DP = reinterpret_cast<unsigned int*>(a.dqp);
  float dbv = 0.f;
  for (int b = blockIdx.x; b < a.B; b += gridDim.x) {
    const int first = a.first_live ? min(a.first_live[b], L - 1) : 0;
    const int rs = first & ~15;
    pin_head(hd);
    __syncthreads();
    stage_x(sm, X + (size_t)b * L * LX_D, rs, L, SLD, h, lane);
    const float ql = lane < 32 ? (float)Q[(size_t)b * LX_D + h * LX_DK + lane] : 0.f;
    const float gl = lane < 32 ? (float)G[(size_t)b * LX_D + h * LX_DK + lane] : 0.f;
    float q0, q1, d0, d1;
    vec_times_rows(hd.wk, ql, q0, q1);
    q0 *= a.scale; q1 *= a.scale;
    __builtin_amdgcn_sched_barrier(0);                          // one matrix unpacked at a time
    vec_times_rows(hd.wv, gl, d0, d1);                          // dxbar_h
    __builtin_amdgcn_sched_barrier(0);
    const float qb = wave_sum(ql * hd.bk) * a.scale;
    const float dsp = wave_sum(gl * hd.bv);                     // d(sum of p~)
    put_cols(sm, h, 4 + h, q0, q1, lane);
    put_cols(sm, 8 + h, 12 + h, d0, d1, lane);
    reinterpret_cast<float2*>(sm.vec + h * LX_D)[lane] = make_float2(d0, d1);
    reinterpret_cast<float2*>(sm.vec + (4 + h) * LX_D)[lane] = make_float2(q0, q1);
    {                                                           // the operands of the dWV / dWK products: other heads' blocks zero
      const int owner = lane >> 4;                              // features (2*lane, 2*lane+1) belong to head lane >> 4
      const unsigned int gq = reinterpret_cast<const unsigned int*>(G + (size_t)b * LX_D)[lane];
      const unsigned int qq = reinterpret_cast<const unsigned int*>(Q + (size_t)b * LX_D)[lane];
      YV[((size_t)b * LX_H + h) * (LX_D / 2) + lane] = owner == h ? gq : 0u;
      YQ[((size_t)b * LX_H + h) * (LX_D / 2) + lane] = owner == h ? qq : 0u;
    }
    stage_wait();
    __syncthreads();
    {                                                           // x . q' and x . dxbar, all heads, one pass
      Frag<__bf16> bq[4];
      const int n = lane & 15, g = lane >> 4;
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) bq[ks].v = *reinterpret_cast<const bf16x8_t*>(sm.qs + n * LX_QLD + ks * 32 + 8 * g);
#pragma unroll 1
      for (int t = (rs >> 4) + h; t < LT; t += 4) {
        const f32x4 acc = tile_product(sm, t, bq, lane);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float v = acc[r] + __shfl_down(acc[r], 4);
          if (n < 4) sm.ss[n * SLD + 16 * t + 4 * g + r] = v;
          else if (n >= 8 && n < 12) sm.dp[(n - 8) * SLD + 16 * t + 4 * g + r] = v;
        }
      }
    }
    __syncthreads();
    float p[LX_KPL], kp[LX_KPL];
    bool msk[LX_KPL], full;
    const unsigned int dbase = (((unsigned int)b * LX_H + h) * L + (L - 1)) * rg_lpad(L);
    softmax_keys(sm, SLD, h, lane, L, rs, qb, a.key_ids + (size_t)b * L, a.pad_value, drop, dbase, p, kp, msk, full);
    float dpk[LX_KPL];
    float delta = 0.f, sp = 0.f;
#pragma unroll
    for (int i = 0; i < LX_KPL; ++i) {
      const int j = lane + 64 * i;
      const float dpt = (j >= rs && j < L ? sm.dp[h * SLD + j] : 0.f) + dsp;
      dpk[i] = dpt * kp[i];                                     // d loss / d (undropped probability)
      delta += p[i] * dpk[i];
      sp += p[i] * kp[i];
    }
    delta = wave_sum(delta);
    sp = wave_sum(sp);
#pragma unroll
    for (int i = 0; i < LX_KPL; ++i) {
      const int j = lane + 64 * i;
      const float ds = (full || msk[i]) ? 0.f : p[i] * (dpk[i] - delta);
      if (j < SLD) *reinterpret_cast<float2*>(sm.cf + j * 8 + 2 * h) = make_float2(p[i] * kp[i], ds);
    }
    float a0 = 0.f, a1 = 0.f, g0 = 0.f, g1 = 0.f;               // xbar_h and dq'_h (before the scale)
#pragma unroll 2
    for (int j = rs; j < SLD; j += 4) {
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const float2 c = *reinterpret_cast<const float2*>(sm.cf + (j + u) * 8 + 2 * h);
        const unsigned int xw = x_pair(sm, j + u, lane);
        const float x0 = bf_lo(xw), x1 = bf_hi(xw);
        a0 += c.x * x0; a1 += c.x * x1;
        g0 += c.y * x0; g1 += c.y * x1;
      }
    }
    g0 *= a.scale; g1 *= a.scale;
    XB[((size_t)b * LX_H + h) * (LX_D / 2) + lane] = pack_bf(a0, a1);
    DP[((size_t)b * LX_H + h) * (LX_D / 2) + lane] = pack_bf(g0, g1);
    pin_head(hd);                                               // (no reuse of the unpacked rows from the top of the loop)
    __builtin_amdgcn_sched_barrier(0);
    const float dq = rows_times_vec(hd.wk, g0, g1, lane);
    if (!(lane & 1)) DQ[(size_t)b * LX_D + h * LX_DK + (lane >> 1)] = (__bf16)dq;
    dbv += gl * sp;
    __syncthreads();                                            // cf of every head is in place
    {                                                           // dx rows: wave w takes rows rs + w, rs + w + 4, ...
      float2 vd[4], vq[4];
#pragma unroll
      for (int hh = 0; hh < 4; ++hh) {
        vd[hh] = reinterpret_cast<const float2*>(sm.vec + hh * LX_D)[lane];
        vq[hh] = reinterpret_cast<const float2*>(sm.vec + (4 + hh) * LX_D)[lane];
      }
      unsigned int* dxb = DX + (size_t)b * L * (LX_D / 2);
      for (int j = h; j < rs; j += 4) dxb[(size_t)j * (LX_D / 2) + lane] = 0u;
#pragma unroll 2
      for (int j = rs + h; j < L; j += 4) {
        const float4 c0 = *reinterpret_cast<const float4*>(sm.cf + j * 8);
        const float4 c1 = *reinterpret_cast<const float4*>(sm.cf + j * 8 + 4);
        const float cc[8] = {c0.x, c0.y, c0.z, c0.w, c1.x, c1.y, c1.z, c1.w};
        float o0 = 0.f, o1 = 0.f;
#pragma unroll
        for (int hh = 0; hh < 4; ++hh) {
          o0 += cc[2 * hh] * vd[hh].x + cc[2 * hh + 1] * vq[hh].x;
          o1 += cc[2 * hh] * vd[hh].y + cc[2 * hh + 1] * vq[hh].y;
        }
        dxb[(size_t)j * (LX_D / 2) + lane] = pack_bf(o0, o1);
      }
    }
  }
  if (a.dbv && lane < 32) atomicAdd(a.dbv + h * LX_DK + lane, dbv);
}

static size_t lx_smem_bytes(int L, bool bwd) {
  const size_t SLD = (size_t)((L + 15) >> 4) * 16;
  size_t n = SLD * 256 + 16 * LX_QLD * 2 + 4 * SLD * 4;
  if (bwd) n += 4 * SLD * 4 + SLD * 8 * 4 + 8 * LX_D * 4;
  return n;
}

extern "C" int rg_attn_lastq_x_supported(int d, int P, int H, int L, int dtype) {
  return dtype == RG_BF16 && d == LX_D && P == LX_D && H == LX_H && L >= 1 && L <= 64 * LX_KPL;
}

static int lx_launch(const rg_lastq_x_args* a, bool bwd, hipStream_t s) {
  if (!a || a->B <= 0) return 0;
  if (a->L < 1 || a->L > 64 * LX_KPL) return rg_set_error_msg(RG_ERR_UNSUPPORTED, "attn_lastq_x: L must be in 1..256");
  if (!a->x || !a->qlast || !a->wk || !a->wv || !a->bk || !a->bv || !a->key_ids)
    return rg_set_error_msg(RG_ERR_INVALID, "attn_lastq_x: x, qlast, wk, wv, bk, bv and key_ids are required");
  if (!bwd && !a->ctx) return rg_set_error_msg(RG_ERR_INVALID, "attn_lastq_x_fwd: ctx is required");
  if (bwd && (!a->dctx || !a->dx || !a->dq || !a->ym_v || !a->ym_q || !a->xbar || !a->dqp))
    return rg_set_error_msg(RG_ERR_INVALID, "attn_lastq_x_bwd: dctx, dx, dq, ym_v, ym_q, xbar and dqp are required");
  const size_t smem = lx_smem_bytes(a->L, bwd);
  const int per_cu = smem <= 52 * 1024 ? 3 : (smem <= 80 * 1024 ? 2 : 1);
  const int grid = a->B < 256 * per_cu ? a->B : 256 * per_cu;
  const void* fn = bwd ? reinterpret_cast<const void*>(attn_lastq_x_bwd_kernel) : reinterpret_cast<const void*>(attn_lastq_x_fwd_kernel);
  hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
  if (e != hipSuccess) return rg_set_error(e, "attn_lastq_x(smem)");
  if (bwd) hipLaunchKernelGGL(attn_lastq_x_bwd_kernel, dim3(grid), dim3(256), smem, s, *a);
  else hipLaunchKernelGGL(attn_lastq_x_fwd_kernel, dim3(grid), dim3(256), smem, s, *a);
  RG_CHECK_LAUNCH();
  return 0;
}

extern "C" int rg_attn_lastq_x_fwd(const rg_lastq_x_args* a, void* stream) { return lx_launch(a, false, (hipStream_t)stream); }
extern "C" int rg_attn_lastq_x_bwd(const rg_lastq_x_args* a, void* stream) { return lx_launch(a, true, (hipStream_t)stream); }
