// Fused post-attention block, forward, for d_model == n_heads * 32 == 256 (BASELINE configs[4]: hidden 256, 8 heads) -- the
// d_model = 128 kernel of fused.hip on a wider machine (round 5):
//
//   y   = LayerNorm(ctx.Wo^T + bo + x)                      MultiHeadAttention tail  transformer.py:160-161
//  [y   = LayerNorm(y + o[b])                               collapsed decoder cross-attention (Q1), :259]
//   h1  = y.W1^T + b1 ; g = gelu_tanh(dropout(h1))          PositionWiseFeedForwardNet  transformer.py:181-184
//   out = LayerNorm(dropout(g.W2^T + b2) + y) * rowmask     :185-188 and the `* pad_mask` of :594 / :539
//
// replacing, per layer, gemm_ws<2,1> (+ bias + residual) + a LayerNorm row pass [+ rg_cross_add_ln] + gemm_ws<2,1> with the
// dropout / GELU epilogue (h1 and the activated operand, 2 x [M, 512]) + gemm_ws<4,1> + rg_add_drop_ln: per token 2 x 256 elements
// read and 256 written (+ 256 + 512 saved in training launches) instead of ~4.9 K elements moved by the five launches.
//
// Geometry: ONE 512-thread workgroup per CU, 64-token tiles.  Wave w owns the 32 output features 32 w .. 32 w + 31 of every
// 256-wide product (8 waves x 32), weight fragment = A operand, activation fragment = B operand (an accumulator register holds 4
// consecutive features of one token: intermediates leave as packed 8-byte stores).  An activation tile [64 x 256] is TWO
// [64 x 128] sub-tiles in the d_model = 128 kernel's layout (256-byte rows, 16-byte chunks XOR-swizzled by the row: conflict-free
// ds_read_b128 fragment reads); a product over K = 256 is two K halves, each with its own set of 8 weight fragments per wave
// (fragment-packed weights, rg_cast RG_CAST_PACK: contiguous 1 KB reads), loaded from L2 ONE HALF-STEP ahead of its use into
// alternating register sets.  d_ff is streamed in 256-wide chunks (each wave: 32 features of the chunk), so every GEMM step has
// the same shape: N = 256, K = 256.  Per 64-token tile a workgroup streams 655 KB of weights from L2 (10 KB per token).
// LDS: ctx (later: out) | x (later: g chunk) | y [| h1 chunk staging] = 3-4 x 32 KB + 10 KB of parameters + the LayerNorm exchange.
#include <stdlib.h>
#include <type_traits>
#include "rg_common.hip.h"
#include "../../include/recguru_hip.h"

namespace {

constexpr int D2 = 256;                 // d_model == P
constexpr int HW = 128;                 // sub-tile width
constexpr int NWV2 = 8;                 // waves per workgroup
constexpr int NT2 = 512;                // threads
constexpr int TM2 = 64;                 // tokens per tile
constexpr int RT2 = 4;                  // 16-row tiles per work tile
constexpr int SUB = TM2 * HW;           // elements of a sub-tile
typedef __bf16 T;

// element offset inside a [64 x 128] sub-tile: 256-byte rows, chunk c of row r at chunk c ^ (r & 15)
__device__ __forceinline__ int soff(int row, int col) { return row * HW + ((((col >> 3) ^ row) & 15) << 3) + (col & 7); }
// ... inside a [64 x 256] tile = two sub-tiles
__device__ __forceinline__ int toff(int row, int col) { return (col >> 7) * SUB + soff(row, col & 127); }

struct WSet2 { Frag<T> f[4][2]; };

template <typename U> __device__ __forceinline__ U* gofs2(U* base, unsigned int elem) {
  return reinterpret_cast<U*>(reinterpret_cast<char*>(base) + (size_t)(elem * (unsigned int)sizeof(U)));
}
template <typename U> __device__ __forceinline__ const U* gofs2(const U* base, unsigned int elem) {
  return reinterpret_cast<const U*>(reinterpret_cast<const char*>(base) + (size_t)(elem * (unsigned int)sizeof(U)));
}

// 8 fragments (4 k-steps x 2 feature tiles) of the fragment-packed weight W (logical [N][K], K = ldk): rows row0 .. row0 + 31,
// k0 .. k0 + 127
__device__ __forceinline__ void load_wset2(WSet2& w, const T* __restrict__ W, int ldk, int row0, int k0, int li, int lg) {
  const unsigned int nks = (unsigned int)ldk >> 5;
  const T* base = W + ((unsigned int)(row0 >> 4) * nks + (unsigned int)(k0 >> 5)) * 512u;
  const unsigned int sct = nks * 512u, lofs = (unsigned int)(lg * 16 + li) * 8u;
#pragma unroll
  for (int ks = 0; ks < 4; ++ks)
#pragma unroll
    for (int ct = 0; ct < 2; ++ct) load_frag(w.f[ks][ct], gofs2(base + (ct * sct + ks * 512u), lofs));
}

// acc[ct][rt] += W[row0 + ct*16 + i][k0 + k] . Act_sub[rt*16 + j][k]  over the 128 k of one sub-tile
__device__ __forceinline__ void mma_wset2(f32x4 (&acc)[2][RT2], const WSet2& w, const T* __restrict__ sub, int li, int lg) {
  Frag<T> af[2][RT2];
#pragma unroll
  for (int rt = 0; rt < RT2; ++rt) load_frag(af[0][rt], sub + soff(rt * 16 + li, 8 * lg));
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) {
    if (ks < 3) {
#pragma unroll
      for (int rt = 0; rt < RT2; ++rt) load_frag(af[(ks + 1) & 1][rt], sub + soff(rt * 16 + li, (ks + 1) * 32 + 8 * lg));
    }
#pragma unroll
    for (int rt = 0; rt < RT2; ++rt)
#pragma unroll
      for (int ct = 0; ct < 2; ++ct) mma(w.f[ks][ct], af[ks & 1][rt], acc[ct][rt]);
  }
}

__device__ __forceinline__ void init_acc2(f32x4 (&acc)[2][RT2], const float* __restrict__ bias_lds, int n0, int lg) {
#pragma unroll
  for (int ct = 0; ct < 2; ++ct) {
    float b[4];
    load4f(b, bias_lds + n0 + ct * 16 + 4 * lg);
#pragma unroll
    for (int rt = 0; rt < RT2; ++rt) acc[ct][rt] = (f32x4){b[0], b[1], b[2], b[3]};
  }
}

// LayerNorm over 256 features on the accumulator registers: lane (li, lg) of wave w holds, for token rt*16 + li, the features
// 32 w + ct*16 + 4 lg + r.  Per wave: sum and centred second moment of its 32 features of a row (in-lane + 2 shuffles), the eight
// (sum, M2) pairs of a row exchanged through LDS once and combined exactly (Chan et al.).  Every lane group stores the (identical)
// partial statistics: no lane-divergent branch in front of the exchange (recguru_amd/isa_screen.py, DESIGN.md 2a finding 2).
// (redA / redB carry NO __restrict__: the exchange crosses a workgroup barrier written as inline asm, and hipcc moves stores through a
// noalias pointer past an asm statement that does not name it -- the first build of this kernel read the partial statistics in front
// of the barrier, each wave with whatever the others had written so far)
__device__ __forceinline__ void ln_regs2(f32x4 (&v)[2][RT2], float (&rstd)[RT2], const float* gamma_lds,
                                         const float* beta_lds, float* redA, float* redB,
                                         float eps, int n0, int wave, int li, int lg) {
  float s[RT2], m2[RT2];
#pragma unroll
  for (int rt = 0; rt < RT2; ++rt) {
    float t = 0.f;
#pragma unroll
    for (int ct = 0; ct < 2; ++ct)
#pragma unroll
      for (int r = 0; r < 4; ++r) t += v[ct][rt][r];
    t += __shfl_xor(t, 16);
    t += __shfl_xor(t, 32);
    s[rt] = t;
    const float mw = t * (1.f / 32.f);
    float q = 0.f;
#pragma unroll
    for (int ct = 0; ct < 2; ++ct)
#pragma unroll
      for (int r = 0; r < 4; ++r) { const float dd = v[ct][rt][r] - mw; q += dd * dd; }
    q += __shfl_xor(q, 16);
    q += __shfl_xor(q, 32);
    m2[rt] = q;
  }
#pragma unroll
  for (int rt = 0; rt < RT2; ++rt) { redA[(rt * 16 + li) * NWV2 + wave] = s[rt]; redB[(rt * 16 + li) * NWV2 + wave] = m2[rt]; }
  lds_barrier();
  float mean[RT2];
#pragma unroll
  for (int rt = 0; rt < RT2; ++rt) {
    float p[8], q[8];
    load8(p, redA + (rt * 16 + li) * NWV2);
    load8(q, redB + (rt * 16 + li) * NWV2);
    float sm = 0.f, M2 = 0.f;
#pragma unroll
    for (int w = 0; w < NWV2; ++w) { sm += p[w]; M2 += q[w]; }
    const float mu = sm * (1.f / D2);
#pragma unroll
    for (int w = 0; w < NWV2; ++w) { const float dm = p[w] * (1.f / 32.f) - mu; M2 += 32.f * dm * dm; }
    mean[rt] = mu;
    rstd[rt] = __builtin_amdgcn_rsqf(M2 * (1.f / D2) + eps);      // (the bare v_rsq_f32: the argument is >= eps)
  }
#pragma unroll
  for (int ct = 0; ct < 2; ++ct) {
    float g[4], b[4];
    load4f(g, gamma_lds + n0 + ct * 16 + 4 * lg);
    load4f(b, beta_lds + n0 + ct * 16 + 4 * lg);
#pragma unroll
    for (int rt = 0; rt < RT2; ++rt)
#pragma unroll
      for (int r = 0; r < 4; ++r) v[ct][rt][r] = (v[ct][rt][r] - mean[rt]) * rstd[rt] * g[r] + b[r];
  }
}

__device__ __forceinline__ void regs_to_tile2(const f32x4 (&v)[2][RT2], T* __restrict__ tile, int n0, int li, int lg) {
#pragma unroll
  for (int ct = 0; ct < 2; ++ct)
#pragma unroll
    for (int rt = 0; rt < RT2; ++rt) {
      float t[4] = {v[ct][rt][0], v[ct][rt][1], v[ct][rt][2], v[ct][rt][3]};
      store4(tile + toff(rt * 16 + li, n0 + ct * 16 + 4 * lg), t);
    }
}

__device__ __forceinline__ void add_tile2(f32x4 (&acc)[2][RT2], const T* __restrict__ tile, int n0, int li, int lg) {
#pragma unroll
  for (int ct = 0; ct < 2; ++ct)
#pragma unroll
    for (int rt = 0; rt < RT2; ++rt) {
      float r4[4];
      load4t(r4, tile + toff(rt * 16 + li, n0 + ct * 16 + 4 * lg));
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[ct][rt][r] += r4[r];
    }
}

// cooperative, coalesced copy of a [64 x 256] LDS tile to its rows of a row-major HBM matrix: thread tid moves chunk
// (row 16 i + (tid >> 5), columns 8 (tid & 31) ..) of row tile i
template <bool NT>
__device__ __forceinline__ void tile_to_hbm2(const T* __restrict__ tile, T* __restrict__ dst, int ld, int col0, const int (&mb)[RT2], int M, int tid) {
#pragma unroll
  for (int i = 0; i < RT2; ++i) {
    const int r = 16 * i + (tid >> 5), c8 = (tid & 31) * 8, m = mb[i] + (tid >> 5);
    if (m < M) {
      T* g = gofs2(dst, (unsigned int)m * (unsigned int)ld + (unsigned int)(col0 + c8));
      const Frag<T> raw = *reinterpret_cast<const Frag<T>*>(tile + toff(r, c8));
      if constexpr (NT) frag_store_nt(g, raw);
      else *reinterpret_cast<Frag<T>*>(g) = raw;
    }
  }
}

__device__ __forceinline__ void zero_to_hbm2(T* __restrict__ dst, int ld, int col0, const int (&mb)[RT2], int M, int tid) {
  Frag<T> z;
  frag_zero(z);
#pragma unroll
  for (int i = 0; i < RT2; ++i) {
    const int c8 = (tid & 31) * 8, m = mb[i] + (tid >> 5);
    if (m < M) *reinterpret_cast<Frag<T>*>(gofs2(dst, (unsigned int)m * (unsigned int)ld + (unsigned int)(col0 + c8))) = z;
  }
}

// DM: dropout mode -- 0 none, 1 p == 0.5 (one hash bit per element), 2 generic p (16-bit hash fields); CROSS: the decoder form
// (collapsed cross-attention stage between the two LayerNorms); SAVE: a training launch (y / y2 / h1 / rstd* saved for the backward)
template <int DM, bool CROSS, bool SAVE>
__global__ __launch_bounds__(NT2, 2) void post_attn_fwd256_kernel(rg_post_attn_args a) {
  constexpr int ACT = TM2 * D2;                                     // elements of an activation tile
  extern __shared__ __align__(16) unsigned char smem2[];
  T* Actx = reinterpret_cast<T*>(smem2);                            // ctx tile; later: the output on its way to HBM
  T* Ag = Actx + ACT;                                               // x tile (residual), then the g chunks
  T* Ay = Ag + ACT;                                                 // LayerNorm-1 (or cross) output: operand and residual of the FFN
  float* prm = reinterpret_cast<float*>(Ay + ACT);                  // 8 x 256 + dff floats
  float* redA = prm + 8 * D2 + a.dff;                               // [64][8]
  float* redB = redA + TM2 * NWV2;
  float* klut = redB + TM2 * NWV2;                                  // [16][4]
  T* Ah = reinterpret_cast<T*>(klut + 64);                          // SAVE: h1 chunk staging
  float *p_bo = prm, *p_g1 = prm + D2, *p_be1 = prm + 2 * D2, *p_b2 = prm + 3 * D2, *p_g2 = prm + 4 * D2,
        *p_be2 = prm + 5 * D2, *p_gc = prm + 6 * D2, *p_bec = prm + 7 * D2, *p_b1 = prm + 8 * D2;

  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 15, lg = lane >> 4;
  const T* __restrict__ ctx = reinterpret_cast<const T*>(a.ctx);
  const T* __restrict__ x = reinterpret_cast<const T*>(a.x);
  const T* __restrict__ Wo = reinterpret_cast<const T*>(a.Wo);
  const T* __restrict__ W1 = reinterpret_cast<const T*>(a.W1);
  const T* __restrict__ W2 = reinterpret_cast<const T*>(a.W2);
  T* __restrict__ out = reinterpret_cast<T*>(a.out);
  T* __restrict__ ysave = SAVE ? reinterpret_cast<T*>(a.y_save) : nullptr;
  T* __restrict__ y2save = SAVE ? reinterpret_cast<T*>(a.y2_save) : nullptr;
  T* __restrict__ h1save = SAVE ? reinterpret_cast<T*>(a.h1_save) : nullptr;
  float* __restrict__ rstd1o = SAVE ? a.rstd1 : nullptr;
  float* __restrict__ rstd2o = SAVE ? a.rstd2 : nullptr;
  float* __restrict__ rstdco = SAVE ? a.rstd_c : nullptr;
  const int n0 = wave * 32;                 // this wave's 32 output features of every 256-wide block
  const int ntiles = (a.M + TM2 - 1) / TM2;
  const int nchunk = a.dff / D2;
  DropCfg drop1 = make_drop(a.drop_p, a.seed_h1), drop2 = make_drop(a.drop_p, a.seed_out);
  if constexpr (DM == 2) { drop1.onebit = 0u; drop2.onebit = 0u; }

  // ---- once per workgroup: parameters -> LDS
  for (int i = tid; i < D2; i += NT2) {
    p_bo[i] = a.bo[i]; p_g1[i] = a.g1[i]; p_be1[i] = a.be1[i]; p_b2[i] = a.b2[i]; p_g2[i] = a.g2[i]; p_be2[i] = a.be2[i];
    p_gc[i] = a.gc ? a.gc[i] : 1.f; p_bec[i] = a.bec ? a.bec[i] : 0.f;
  }
  for (int i = tid; i < a.dff; i += NT2) p_b1[i] = a.b1[i];
  if (tid < 64) klut[tid] = (((tid >> 2) >> (tid & 3)) & 1) ? drop1.inv_keep : 0.f;
  const unsigned int rot0 = (4u * lg + 28u) & 31u, rot1 = (4u * lg + 12u) & 31u;   // hash bit 4 lg + j (16 + 4 lg + j) -> bit 4 + j

  WSet2 wA, wB;                             // alternating weight sets: the K halves of consecutive GEMM steps
  Frag<T> cpre[RT2], xpre[RT2];             // ctx / x rows of the NEXT tile (staging prefetch)

  LiveWalk lw;
  lw.init(a.live16, a.M);
  const int nwork = a.live16 ? (lw.nlive + RT2 - 1) / RT2 : ntiles;
  int cur = (int)blockIdx.x, kcur = 0;
  auto next_group = [&](int (&g)[RT2]) -> bool {
    if (cur >= nwork) {
#pragma unroll
      for (int rt = 0; rt < RT2; ++rt) g[rt] = a.M;      // absent: prefetch_rows clamps, nothing is stored
      return false;
    }
    if (!a.live16) {
#pragma unroll
      for (int rt = 0; rt < RT2; ++rt) g[rt] = cur * TM2 + 16 * rt;
    } else {
      lw.template group_n<RT2>(kcur, g, a.M);
    }
    cur += gridDim.x;
    ++kcur;
    return true;
  };
  auto prefetch_rows = [&](const int (&g)[RT2]) {        // rows >= M: clamped address, no branch (never stored)
#pragma unroll
    for (int i = 0; i < RT2; ++i) {
      const int c8 = (tid & 31) * 8, m = min(g[i] + (tid >> 5), a.M - 1);
      load_frag(cpre[i], gofs2(ctx, (unsigned int)(m * D2 + c8)));
      load_frag(xpre[i], gofs2(x, (unsigned int)(m * D2 + c8)));
    }
  };
  // p == 0.5 / generic dropout of this lane's 2 x 4 accumulator elements of every row tile: element (row, col0 + ct*16 + 4 lg + r)
  // of a [M x ncol] index space; col0 is a multiple of 32, so both feature tiles of a lane sit in one hash word
  auto drop_acc = [&](f32x4 (&v)[2][RT2], const DropCfg& dc, const int (&mrow)[RT2], unsigned int ncol, unsigned int col0) {
#pragma unroll
    for (int rt = 0; rt < RT2; ++rt) {
      const unsigned int rb = (unsigned int)(mrow[rt] + li) * ncol + col0 + 4u * (unsigned int)lg;
      if constexpr (DM == 1) {
        const unsigned int w = rg_hash(dc.seed, rb >> 5);
        float k0[4], k1[4];
        load4f(k0, reinterpret_cast<const float*>(reinterpret_cast<const char*>(klut) + (__builtin_amdgcn_alignbit(w, w, rot0) & 0xF0u)));
        load4f(k1, reinterpret_cast<const float*>(reinterpret_cast<const char*>(klut) + (__builtin_amdgcn_alignbit(w, w, rot1) & 0xF0u)));
#pragma unroll
        for (int r = 0; r < 4; ++r) { v[0][rt][r] *= k0[r]; v[1][rt][r] *= k1[r]; }
      } else {
        float k0[4], k1[4];
        rg_keep4_pair(dc, rb, k0, k1);
#pragma unroll
        for (int r = 0; r < 4; ++r) { v[0][rt][r] *= k0[r]; v[1][rt][r] *= k1[r]; }
      }
    }
  };

  int mb[RT2], mbn[RT2];
  bool have = next_group(mb);
  if (have) {
    load_wset2(wA, Wo, D2, n0, 0, li, lg);
    prefetch_rows(mb);
  }
  for (; have;) {
    const bool have_next = next_group(mbn);
    // ---- ctx and x tiles: registers -> LDS (x parks in the g-chunk buffer, free until the FFN)
#pragma unroll
    for (int i = 0; i < RT2; ++i) {
      const int r = 16 * i + (tid >> 5), c8 = (tid & 31) * 8;
      *reinterpret_cast<Frag<T>*>(Actx + toff(r, c8)) = cpre[i];
      *reinterpret_cast<Frag<T>*>(Ag + toff(r, c8)) = xpre[i];
    }
    float rm4[RT2];
    bool any_live = false;
#pragma unroll
    for (int rt = 0; rt < RT2; ++rt) {
      const int m = mb[rt] + li;
      rm4[rt] = (a.rowmask && m < a.M) ? a.rowmask[m] : 1.f;
      any_live = any_live || rm4[rt] != 0.f;
    }
    if (a.rowmask && !a.live16 && __ballot(any_live) == 0ull) {      // (every wave sees all 64 rows: uniform across the workgroup)
      // 64 padded positions: out * rowmask = 0 whatever the arithmetic gives -- zeros (and finite placeholders for the backward)
      zero_to_hbm2(out, D2, 0, mb, a.M, tid);
      const bool cross = CROSS;
      if (ysave) zero_to_hbm2(ysave, D2, 0, mb, a.M, tid);
      if (cross && y2save) zero_to_hbm2(y2save, D2, 0, mb, a.M, tid);
      if (h1save)
        for (int ch = 0; ch < nchunk; ++ch) zero_to_hbm2(h1save, a.dff, ch * D2, mb, a.M, tid);
      if (tid < TM2 && mb[0] + tid < a.M) {             // plain tiles only: 64 consecutive rows
        if (rstd1o) rstd1o[mb[0] + tid] = 0.f;
        if (rstd2o) rstd2o[mb[0] + tid] = 0.f;
        if (rstdco) rstdco[mb[0] + tid] = 0.f;
      }
      lds_barrier();                                    // the staged ctx / x tile of this iteration is dropped
      prefetch_rows(mbn);
#pragma unroll
      for (int rt = 0; rt < RT2; ++rt) mb[rt] = mbn[rt];
      have = have_next;
      continue;
    }
    lds_barrier();
    // ---- attention output projection: K halves 0 (weights already in wA) and 1, bias folded into the accumulators
    f32x4 acc[2][RT2];
    init_acc2(acc, p_bo, n0, lg);
    load_wset2(wB, Wo, D2, n0, HW, li, lg);
    mma_wset2(acc, wA, Actx, li, lg);
    // FFN chunk 0, K half 0: hidden behind the second half + LayerNorm 1 (decoder form: requested after the cross stage's own loads
    // instead -- 32 more live registers across that stage were what spilled there)
    if constexpr (!CROSS) load_wset2(wA, W1, D2, n0, 0, li, lg);
    mma_wset2(acc, wB, Actx + SUB, li, lg);
    add_tile2(acc, Ag, n0, li, lg);                      // + residual x
    float rstd[RT2];
    ln_regs2(acc, rstd, p_g1, p_be1, redA, redB, a.eps, n0, wave, li, lg);
    // (past the barrier inside ln_regs2 every wave is done with the ctx tile and with x)
    if (rstd1o && wave == 0 && lg == 0) {
#pragma unroll
      for (int rt = 0; rt < RT2; ++rt)
        if (mb[rt] + li < a.M) rstd1o[mb[rt] + li] = rstd[rt];
    }
    regs_to_tile2(acc, Ay, n0, li, lg);
    if constexpr (CROSS) {
      // collapsed decoder cross-attention: y2 = LayerNorm(y1 + o[b]); y1 (ROUNDED, as the unfused path sees it) is saved first
      lds_barrier();
      if (ysave) tile_to_hbm2<true>(Ay, ysave, D2, 0, mb, a.M, tid);
      int mrow[RT2], brow[RT2];
#pragma unroll
      for (int rt = 0; rt < RT2; ++rt) { mrow[rt] = min(mb[rt] + li, a.M - 1); brow[rt] = mrow[rt] / a.L; }
      if (a.cross_s) {          // attention-map dropout: o = bo + sum_h s[m,h] * oh[b,h,:]   (H == 8 on this path)
        float bo4[2][4];
#pragma unroll
        for (int ct = 0; ct < 2; ++ct) load4f(bo4[ct], a.cross_bo + n0 + ct * 16 + 4 * lg);
#pragma unroll
        for (int rt = 0; rt < RT2; ++rt) {
          float sv[8];
          load8(sv, a.cross_s + (size_t)mrow[rt] * 8);
#pragma unroll
          for (int ct = 0; ct < 2; ++ct) {
            float o4[4] = {bo4[ct][0], bo4[ct][1], bo4[ct][2], bo4[ct][3]};
#pragma unroll
            for (int hh = 0; hh < 8; ++hh) {
              float w4[4];
              load4f(w4, a.cross_oh + ((size_t)brow[rt] * 8 + hh) * D2 + n0 + ct * 16 + 4 * lg);
#pragma unroll
              for (int r = 0; r < 4; ++r) o4[r] += sv[hh] * w4[r];
            }
            float y4[4];
            load4t(y4, Ay + toff(rt * 16 + li, n0 + ct * 16 + 4 * lg));
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[ct][rt][r] = y4[r] + o4[r];
          }
        }
      } else {
#pragma unroll
        for (int rt = 0; rt < RT2; ++rt)
#pragma unroll
          for (int ct = 0; ct < 2; ++ct) {
            float o4[4], y4[4];
            load4f(o4, a.o_bcast + (size_t)brow[rt] * D2 + n0 + ct * 16 + 4 * lg);
            load4t(y4, Ay + toff(rt * 16 + li, n0 + ct * 16 + 4 * lg));
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[ct][rt][r] = y4[r] + o4[r];
          }
      }
      lds_barrier();                                    // ysave copy and the y1 reads done before the tile is overwritten
      load_wset2(wA, W1, D2, n0, 0, li, lg);            // (see the out-projection above) hidden behind the cross LayerNorm
      ln_regs2(acc, rstd, p_gc, p_bec, redA, redB, a.eps, n0, wave, li, lg);
      if (rstdco && wave == 0 && lg == 0) {
#pragma unroll
        for (int rt = 0; rt < RT2; ++rt)
          if (mb[rt] + li < a.M) rstdco[mb[rt] + li] = rstd[rt];
      }
      regs_to_tile2(acc, Ay, n0, li, lg);
    }
    lds_barrier();                                      // y tile complete (and x no longer needed in Ag)
    {
      constexpr bool cross = CROSS;
      if (cross ? (y2save != nullptr) : (ysave != nullptr)) tile_to_hbm2<true>(Ay, cross ? y2save : ysave, D2, 0, mb, a.M, tid);
    }
    // ---- FFN: d_ff in 256-wide chunks; the second product accumulates across chunks
    f32x4 acc2[2][RT2];
    init_acc2(acc2, p_b2, n0, lg);
#pragma unroll 1
    for (int ch = 0; ch < nchunk; ++ch) {
      init_acc2(acc, p_b1 + ch * D2, n0, lg);
      load_wset2(wB, W1, D2, ch * D2 + n0, HW, li, lg);             // this chunk's second K half
      mma_wset2(acc, wA, Ay, li, lg);                               // h1 chunk = y . W1[chunk]^T + b1
      load_wset2(wA, W2, a.dff, n0, ch * D2, li, lg);               // second product, K half 0 (needed after the GELU below)
      mma_wset2(acc, wB, Ay + SUB, li, lg);
      if (ch > 0) lds_barrier();                                    // previous chunk's readers of Ag / Ah are done
      if constexpr (DM != 0) drop_acc(acc, drop1, mb, (unsigned int)a.dff, (unsigned int)(ch * D2 + n0));   // dropout BEFORE the GELU (Q4)
      if (h1save) regs_to_tile2(acc, Ah, n0, li, lg);
#pragma unroll
      for (int ct = 0; ct < 2; ++ct)
#pragma unroll
        for (int rt = 0; rt < RT2; ++rt)
#pragma unroll
          for (int r = 0; r < 4; r += 2) {
            const f32x2 gg = gelu2_fast((f32x2){acc[ct][rt][r], acc[ct][rt][r + 1]});
            acc[ct][rt][r] = gg.x;
            acc[ct][rt][r + 1] = gg.y;
          }
      regs_to_tile2(acc, Ag, n0, li, lg);
      lds_barrier();
      load_wset2(wB, W2, a.dff, n0, ch * D2 + HW, li, lg);
      if (h1save) tile_to_hbm2<true>(Ah, h1save, a.dff, ch * D2, mb, a.M, tid);
      mma_wset2(acc2, wA, Ag, li, lg);                              // out += g . W2[:, chunk]^T
      // next weight set: W1 chunk ch+1, or Wo for the next tile -- ONE unconditional load sequence from a selected pointer
      load_wset2(wA, (ch + 1 < nchunk) ? W1 : Wo, D2, (ch + 1 < nchunk) ? (ch + 1) * D2 + n0 : n0, 0, li, lg);
      mma_wset2(acc2, wB, Ag + SUB, li, lg);
    }
    prefetch_rows(mbn);                                 // the next tile's ctx / x rows, under the second LayerNorm
    if constexpr (DM != 0) drop_acc(acc2, drop2, mb, (unsigned int)D2, (unsigned int)n0);      // dropout on the l2 output (:186-188)
    add_tile2(acc2, Ay, n0, li, lg);                    // + residual y
    ln_regs2(acc2, rstd, p_g2, p_be2, redA, redB, a.eps, n0, wave, li, lg);
    if (rstd2o && wave == 0 && lg == 0) {
#pragma unroll
      for (int rt = 0; rt < RT2; ++rt)
        if (mb[rt] + li < a.M) rstd2o[mb[rt] + li] = rstd[rt];
    }
#pragma unroll
    for (int ct = 0; ct < 2; ++ct)
#pragma unroll
      for (int rt = 0; rt < RT2; ++rt)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc2[ct][rt][r] *= rm4[rt];
    regs_to_tile2(acc2, Actx, n0, li, lg);
    lds_barrier();
    tile_to_hbm2<false>(Actx, out, D2, 0, mb, a.M, tid);
    lds_barrier();                                      // before the next tile overwrites Actx / Ag / Ay
#pragma unroll
    for (int rt = 0; rt < RT2; ++rt) mb[rt] = mbn[rt];
    have = have_next;
  }
  if (a.live16) {
    // the padded row tiles (listed from the far end of live16): out rows = 0 -- and zeros / finite placeholders in everything a
    // backward pass reads --, 4 row tiles per step
    const int nrt = (a.M + 15) >> 4, ndead = nrt - a.live16[0];
    constexpr bool cross = CROSS;
    for (int j = RT2 * (int)blockIdx.x; j < ndead; j += RT2 * (int)gridDim.x) {
      int md[RT2];
#pragma unroll
      for (int rt = 0; rt < RT2; ++rt) md[rt] = j + rt < ndead ? a.live16[nrt - (j + rt)] * 16 : a.M;
      zero_to_hbm2(out, D2, 0, md, a.M, tid);
      if (a.skip_dead_saves) continue;
      if (ysave) zero_to_hbm2(ysave, D2, 0, md, a.M, tid);
      if (cross && y2save) zero_to_hbm2(y2save, D2, 0, md, a.M, tid);
      if (h1save)
        for (int ch = 0; ch < nchunk; ++ch) zero_to_hbm2(h1save, a.dff, ch * D2, md, a.M, tid);
      if (tid < TM2) {
        const int m = md[tid >> 4] + (tid & 15);
        if (m < a.M) {
          if (rstd1o) rstd1o[m] = 0.f;
          if (rstd2o) rstd2o[m] = 0.f;
          if (rstdco) rstdco[m] = 0.f;
        }
      }
    }
  }
}

}  // namespace

// called by rg_post_attn_fwd (fused.hip) for d == P == 256: bf16 tier, fragment-packed weights, d_ff a multiple of 256
int rg_post_attn_fwd256(const rg_post_attn_args* a, int dtype, hipStream_t s) {
  if (dtype != RG_BF16) return rg_set_error_msg(RG_ERR_UNSUPPORTED, "post_attn_fwd (d_model 256): bf16 tier only");
  if (a->P != D2 || a->dff <= 0 || (a->dff % D2) != 0 || !a->w_packed)
    return rg_set_error_msg(RG_ERR_UNSUPPORTED, "post_attn_fwd (d_model 256): needs n_heads*32 == 256, d_ff % 256 == 0 and fragment-packed weights");
  if (a->x_lo || a->out_lo) return rg_set_error_msg(RG_ERR_UNSUPPORTED, "post_attn_fwd (d_model 256): no split residual stream");
  if ((long long)a->M * a->dff * 2 >= (1ll << 32))
    return rg_set_error_msg(RG_ERR_UNSUPPORTED, "post_attn_fwd: M * d_ff * element size must be below 4 GiB (32-bit offsets)");
  if ((a->o_bcast || a->cross_s) && a->L <= 0) return rg_set_error_msg(RG_ERR_INVALID, "post_attn_fwd: cross stage needs L");
  if (a->cross_s && (!a->cross_oh || !a->cross_bo || a->H != 8)) return rg_set_error_msg(RG_ERR_INVALID, "post_attn_fwd (d_model 256): cross_s needs cross_oh, cross_bo, H == 8");
  const int ntiles = (a->M + TM2 - 1) / TM2;
  const int act = TM2 * D2 * 2;
  const int smem = 3 * act + (8 * D2 + a->dff) * 4 + 2 * TM2 * NWV2 * 4 + 64 * 4 + (a->h1_save ? act : 0);
  if (smem > 160 * 1024) return rg_set_error_msg(RG_ERR_UNSUPPORTED, "post_attn_fwd (d_model 256): d_ff too large for the parameter block in LDS");
  int grid = 256;
  if (grid > ntiles) grid = ntiles;
  const int dm = a->drop_p <= 0.f ? 0 : (a->drop_p == 0.5f ? 1 : 2);
  const bool cross = a->o_bcast || a->cross_s;
  const bool save = a->y_save || a->y2_save || a->h1_save || a->rstd1 || a->rstd2 || a->rstd_c;
#define RG_PA256(DM, C, S)                                                                                              \
  do {                                                                                                                  \
    hipFuncSetAttribute(reinterpret_cast<const void*>(post_attn_fwd256_kernel<DM, C, S>), hipFuncAttributeMaxDynamicSharedMemorySize, smem); \
    hipLaunchKernelGGL((post_attn_fwd256_kernel<DM, C, S>), dim3(grid), dim3(NT2), smem, s, *a);                        \
  } while (0)
#define RG_PA256_DM(DM)                                                           \
  do {                                                                            \
    if (cross) { if (save) RG_PA256(DM, true, true); else RG_PA256(DM, true, false); } \
    else { if (save) RG_PA256(DM, false, true); else RG_PA256(DM, false, false); }     \
  } while (0)
  if (dm == 0) RG_PA256_DM(0);
  else if (dm == 1) RG_PA256_DM(1);
  else RG_PA256_DM(2);
#undef RG_PA256_DM
#undef RG_PA256
  RG_CHECK_LAUNCH();
  return 0;
}
