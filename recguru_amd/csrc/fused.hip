// Fused post-attention block, forward:  one launch computes, for a 64-token tile,
//
//   y   = LayerNorm(ctx.Wo^T + bo + x)                      MultiHeadAttention tail  transformer.py:160-161
//  [y   = LayerNorm(y + o[b])                               collapsed decoder cross-attention (Q1), :259]
//   h1  = y.W1^T + b1 ; g = gelu_tanh(h1)                   PositionWiseFeedForwardNet  transformer.py:181-184
//   out = LayerNorm(g.W2^T + b2 + y) * rowmask              :185-188 and the `* pad_mask` of :594 / :539
//
// with every intermediate resident in LDS / registers: per token the kernel reads ctx and x
// (2*d*sizeof(T) bytes) and writes out (d*sizeof(T)); the [tile, d_ff] activation never reaches HBM
// unless the training path asks for it (h1_save) -- 2*(d*P + 2*d*d_ff) FLOP per token against
// 3*d*sizeof(T) bytes = 341 FLOP/B at d=128, d_ff=512, bf16: far above the HBM ridge, unlike the four separate
// GEMMs (43-102 FLOP/B each).  What bounds the kernel in practice is the VALU: 512 GELUs (two quarter-rate
// transcendentals each), the dropout masks and two LayerNorms per token are ~2x the issue cycles of the token's
// MFMAs (SQ counters: profiles/r01_final/sq_post_attn.txt).
//
// MFMA orientation: the WEIGHT fragment is the A operand (rows = output features) and the
// activation fragment the B operand (columns = tokens), so an accumulator register holds 4
// CONSECUTIVE FEATURES of one token: every LDS / global write of an intermediate is one packed
// 8- or 16-byte vector, never a 2-byte scatter.  The four waves split the output features; each
// weight element is fetched from L2 exactly once per tile, straight into its fragment.
// LDS (bf16): ctx tile + g chunk + y tile (+ the h1 staging tile when saving) = 3-4 x 16 KB, XOR-swizzled rows,
// + 8 KB of parameters and row statistics: 56-72 KB per workgroup, 2 workgroups per CU (256 VGPRs per wave allow
// two waves per SIMD in any case).
#include <type_traits>
#include "rg_common.hip.h"
#include "rg_det.hip.h"
#include "../../include/recguru_hip.h"

#define FT_M 64      // tokens per tile (the backward kernels; the forward block: 16 * RT)
#include <stdlib.h>
#define FD 128       // d_model == P == chunk width
#define FPAD 8
#define FLD (FD + FPAD)

// Activation tiles in LDS are [64 tokens x 128 features].  bf16: rows of exactly 256 B (all 64 banks) with the sixteen
// 16-byte chunks of a row XOR-swizzled by the row index -- chunk c of row r sits at chunk
// c ^ (r & 15) -- which makes the B-operand fragment reads (ds_read_b128: lane (li, lg) reads chunk 4 ks + lg of row li)
// conflict-free under the hardware's 16-lane service groups ({0-3, 12-15, 20-27}, ...: lanes of two lg values land in
// complementary chunk sets); a padded row (272 B) put two lanes of every group on one bank -- 43 % of the kernel's
// LDS cycles were bank-conflict cycles (SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE).  f32 (parity tier): padded rows.
// bf16x3 tier: T = x3p<PL>, two bf16 tiles (hi, lo) of exactly the bf16 tier's layout, PL = rows x 256 bytes apart.
template <typename T> struct Tile {
  static constexpr int LD = sizeof(T) == 2 ? FD : FLD;          // row pitch in elements
  static __device__ __forceinline__ int off(int row, int col) {  // col: any element whose 16-byte chunk holds it
    if constexpr (sizeof(T) == 2) return row * FD + ((((col >> 3) ^ row) & 15) << 3) + (col & 7);
    else return row * FLD + col;
  }
};
// raw f32 tile of the bf16x3 tier (rg_common.hip.h x3r): unpadded 512-byte rows -- the size of the split pair it may replace --,
// the sixteen 32-byte chunks of a row XOR-swizzled by the row index
template <> struct Tile<x3r> {
  static constexpr int LD = FD;
  static __device__ __forceinline__ int off(int row, int col) { return row * FD + ((((col >> 3) ^ row) & 15) << 3) + (col & 7); }
};

// One GEMM step = a set of 8 weight fragments (4 k-steps x 2 feature tiles of this wave), loaded
// straight from L2 one GEMM ahead of its use (software pipelining: the load of set s+1 is issued
// before the MFMAs of set s), against a [64 x 128] activation tile in LDS.
template <typename T> struct WSet { typename OpT<T>::type f[4][2]; };

// Global addresses are formed as (uniform base pointer) + (32-bit element offset): hipcc then keeps the base in scalar
// registers and one 32-bit offset per lane (the saddr form of global_load / global_store) instead of a 64-bit pointer
// per lane and access -- those pointer pairs were what spilled (27 VGPRs, each reload followed by a vmcnt(0) that also
// waited for the previous STORE to be acknowledged), and their 64-bit adds were a tenth of the FFN loop's VALU work.
// The launchers refuse shapes whose offsets do not fit 32 bits.
template <typename T> __device__ __forceinline__ T* gofs(T* base, unsigned int elem) {
  return reinterpret_cast<T*>(reinterpret_cast<char*>(base) + (size_t)(elem * (unsigned int)sizeof(T)));
}
template <typename T> __device__ __forceinline__ const T* gofs(const T* base, unsigned int elem) {
  return reinterpret_cast<const T*>(reinterpret_cast<const char*>(base) + (size_t)(elem * (unsigned int)sizeof(T)));
}

// packed: W is the fragment-packed copy (rg_cast RG_CAST_PACK; ldw = its logical K): the 8 fragments of a step are 8
// contiguous 1 KB reads.  From the row-major [out][in] layout a wave's fragment load touches 16 rows x 64 B, and the
// vector L1 spends a tag lookup per row and 16-lane pass: with 288 such loads per 64-token tile and two workgroups per CU
// the load pipe, not the VALU, was what the waves waited on.
template <typename T>
__device__ __forceinline__ void load_wset(WSet<T>& w, const T* __restrict__ W, int ldw, int row0, int k0, int li, int lg,
                                          int packed = 0) {
  // one code path for both layouts: fragment (ks, ct) sits at (uniform base) + (uniform step) + (lane offset)
  const unsigned int nks = (unsigned int)ldw >> 5;
  const T* base = W + (packed ? ((unsigned int)(row0 >> 4) * nks + (unsigned int)(k0 >> 5)) * 512u
                              : (unsigned int)row0 * (unsigned int)ldw + (unsigned int)k0);
  const unsigned int sct = packed ? nks * 512u : 16u * (unsigned int)ldw, sks = packed ? 512u : 32u;
  if constexpr (std::is_same<T, x3>::value) {
    if (packed) {
      // presplit fragment-packed copy (rg_cast, RG_X3 + RG_CAST_PACK): a fragment is 2 KB = 64 lanes x 16 B of hi parts, then
      // the same of lo parts -- 512 four-byte slots, so the fragment addressing is the packed f32 layout's
      const unsigned int lofs = (unsigned int)(lg * 16 + li) * 4u;
#pragma unroll
      for (int ks = 0; ks < 4; ++ks)
#pragma unroll
        for (int ct = 0; ct < 2; ++ct) {
          const x3* fp = gofs(base + (ct * sct + ks * sks), lofs);
          w.f[ks][ct].hi = *reinterpret_cast<const bf16x8_t*>(fp);
          w.f[ks][ct].lo = *reinterpret_cast<const bf16x8_t*>(fp + 256);
        }
      return;
    }
  }
  const unsigned int lofs = packed ? (unsigned int)(lg * 16 + li) * 8u : (unsigned int)li * (unsigned int)ldw + 8u * lg;
#pragma unroll
  for (int ks = 0; ks < 4; ++ks)
#pragma unroll
    for (int ct = 0; ct < 2; ++ct) load_frag(w.f[ks][ct], gofs(base + (ct * sct + ks * sks), lofs));
}

// acc[ct][rt] += W[row0 + ct*16 + i][k] . Act[rt*16 + j][k]   (weight = A operand, activation = B operand)
template <typename T, int RT, typename LT>
__device__ __forceinline__ void mma_wset(f32x4 (&acc)[2][RT], const WSet<T>& w, const LT* __restrict__ Act, int li, int lg) {
  // the RT token-tile fragments of k-step ks+1 are read while the MFMAs of k-step ks run (one LDS latency per GEMM
  // step instead of RT)
  typename OpT<T>::type af[2][RT];
#pragma unroll
  for (int rt = 0; rt < RT; ++rt) load_frag(af[0][rt], Act + Tile<LT>::off(rt * 16 + li, 8 * lg));
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) {
    if (ks < 3) {
#pragma unroll
      for (int rt = 0; rt < RT; ++rt) load_frag(af[(ks + 1) & 1][rt], Act + Tile<LT>::off(rt * 16 + li, (ks + 1) * 32 + 8 * lg));
    }
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
      for (int ct = 0; ct < 2; ++ct) mma(w.f[ks][ct], af[ks & 1][rt], acc[ct][rt]);
  }
}

// the same product with ONE set of activation fragments (16 VGPRs instead of 32): for the product that runs beside an epilogue's VALU
// work in the pipelined FFN loop, where the LDS latency of a k-step is covered by that work and the registers are what is short
template <typename T, int RT, typename LT>
__device__ __forceinline__ void mma_wset1(f32x4 (&acc)[2][RT], const WSet<T>& w, const LT* __restrict__ Act, int li, int lg) {
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) {
    typename OpT<T>::type af[RT];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) load_frag(af[rt], Act + Tile<LT>::off(rt * 16 + li, ks * 32 + 8 * lg));
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
      for (int ct = 0; ct < 2; ++ct) mma(w.f[ks][ct], af[rt], acc[ct][rt]);
  }
}

// accumulators start at the bias of their 4 features (bias add costs nothing afterwards)
template <int RT>
__device__ __forceinline__ void init_acc(f32x4 (&acc)[2][RT], const float* __restrict__ bias_lds, int n0, int lg) {
#pragma unroll
  for (int ct = 0; ct < 2; ++ct) {
    float b[4];
    load4f(b, bias_lds + n0 + ct * 16 + 4 * lg);
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) acc[ct][rt] = (f32x4){b[0], b[1], b[2], b[3]};
  }
}

// LayerNorm on the accumulator registers.  Lane (li, lg) of wave w holds, for token rt*16+li, the 8
// features w*32 + ct*16 + 4*lg + r.  Each wave forms the mean and the centred second moment of ITS 32 features of a
// row (in-lane sums + 2 shuffles across lg each), the four (sum, M2) pairs of a row are exchanged through LDS ONCE and
// combined exactly (Chan et al.: M2 = sum_w M2_w + 32 * sum_w (mean_w - mean)^2) -- as accurate as torch's two-pass
// LayerNorm with one workgroup barrier instead of two.  On return v holds (v - mean) * rstd * gamma + beta and
// rstd[rt] the row rstd.
// ALLST: every lane group stores the (identical) partial statistics instead of `if (lg == 0)` -- four times the (tiny) LDS
// write traffic for NO lane-divergent branch in the LayerNorm: in post_attn_fwd_kernel<x3, 2, true, false> hipcc parked eleven
// live VGPRs in AGPRs at the top of that branch's join block IN FRONT of its exec restore, i.e. for lanes 0-15 only
// (recguru_amd/isa_screen.py).  The bf16x3 instantiations (512 registers: the ones that spill to AGPRs) use it.
template <int RT, bool ALLST = false>
// (no __restrict__ on the LDS pointers: the exchange crosses the inline-asm workgroup barrier, and hipcc may move stores through a
// noalias pointer past an asm statement that does not name it -- it did in the first build of fused256.hip's copy of this function)
__device__ __forceinline__ void ln_regs(f32x4 (&v)[2][RT], float (&rstd)[RT], const float* gamma_lds,
                                        const float* beta_lds, float* redA, float* redB,
                                        float eps, int n0, int wave, int li, int lg) {
  float s[RT], m2[RT];
#pragma unroll
  for (int rt = 0; rt < RT; ++rt) {
    float t = 0.f;
#pragma unroll
    for (int ct = 0; ct < 2; ++ct)
#pragma unroll
      for (int r = 0; r < 4; ++r) t += v[ct][rt][r];
    t += __shfl_xor(t, 16);
    t += __shfl_xor(t, 32);
    s[rt] = t;                                  // sum over this wave's 32 features
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
  if (ALLST || lg == 0) {
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) { redA[(rt * 16 + li) * 4 + wave] = s[rt]; redB[(rt * 16 + li) * 4 + wave] = m2[rt]; }
  }
  lds_barrier();
  float mean[RT];
#pragma unroll
  for (int rt = 0; rt < RT; ++rt) {
    float p[4], q[4];
    load4f(p, redA + (rt * 16 + li) * 4);
    load4f(q, redB + (rt * 16 + li) * 4);
    mean[rt] = (p[0] + p[1] + p[2] + p[3]) * (1.f / FD);
    float M2 = (q[0] + q[1]) + (q[2] + q[3]);
#pragma unroll
    for (int w = 0; w < 4; ++w) { const float dm = p[w] * (1.f / 32.f) - mean[rt]; M2 += 32.f * dm * dm; }
    // the bare v_rsq_f32 (the argument is >= eps: no denormal range to rescale).  rsqrtf()'s expansion -- compare, scale,
    // v_rsq, rescale, select per row -- was compiled next to the LDS reads of gamma / beta into the same registers, and with
    // two workgroups per CU (two waves per SIMD) lanes 48-63 of those registers came out with the earlier VALU values in
    // ~4 % of the rows of EVERY launch (never with one wave per SIMD; tools/race_post_attn.py, tests/test_determinism_gpu.py)
    rstd[rt] = __builtin_amdgcn_rsqf(M2 * (1.f / FD) + eps);
  }
#pragma unroll
  for (int ct = 0; ct < 2; ++ct) {
    float g[4], b[4];
    load4f(g, gamma_lds + n0 + ct * 16 + 4 * lg);
    load4f(b, beta_lds + n0 + ct * 16 + 4 * lg);
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
      for (int r = 0; r < 4; ++r) v[ct][rt][r] = (v[ct][rt][r] - mean[rt]) * rstd[rt] * g[r] + b[r];
  }
}

// registers -> tile in LDS (4 consecutive features per store)
template <typename T, int RT>
__device__ __forceinline__ void regs_to_tile(const f32x4 (&v)[2][RT], T* __restrict__ tile, int n0, int li, int lg) {
#pragma unroll
  for (int ct = 0; ct < 2; ++ct)
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
      float t[4] = {v[ct][rt][0], v[ct][rt][1], v[ct][rt][2], v[ct][rt][3]};
      store4(tile + Tile<T>::off(rt * 16 + li, n0 + ct * 16 + 4 * lg), t);
    }
}

// the part of the registers that rounding to T loses: v - T(v), as a second tile (split residual stream, rg_post_attn_args.x_lo)
template <typename T, int RT>
__device__ __forceinline__ void regs_to_tile_lo(const f32x4 (&v)[2][RT], T* __restrict__ tile, int n0, int li, int lg) {
#pragma unroll
  for (int ct = 0; ct < 2; ++ct)
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
      float t[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) t[r] = v[ct][rt][r] - (float)(T)v[ct][rt][r];
      store4(tile + Tile<T>::off(rt * 16 + li, n0 + ct * 16 + 4 * lg), t);
    }
}

// acc += tile (this lane's own 8 x 4 positions)
template <typename T, int RT>
__device__ __forceinline__ void add_tile(f32x4 (&acc)[2][RT], const T* __restrict__ tile, int n0, int li, int lg) {
#pragma unroll
  for (int ct = 0; ct < 2; ++ct)
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
      float r4[4];
      load4t(r4, tile + Tile<T>::off(rt * 16 + li, n0 + ct * 16 + 4 * lg));
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[ct][rt][r] += r4[r];
    }
}

// A work tile is 4 row tiles of 16 consecutive rows each, at rows mb[0..3] (mb[rt] >= M marks an absent one): the 64
// rows of a plain tile (mb[rt] = m0 + 16 rt), or the next 4 LIVE row tiles of the workgroup's range when padded row
// tiles are compacted away.  Thread tid stages chunk i of the tile = row 16 i + (tid >> 4), columns 8 (tid & 15)..
// cooperative, coalesced copy of a [64 x 128] LDS tile to its rows of a row-major HBM matrix
template <typename T, bool NT = false, int RT = 4, typename LT>
__device__ __forceinline__ void tile_to_hbm(const LT* __restrict__ tile, T* __restrict__ dst, int ld, int col0, const int (&mb)[RT], int M, int tid) {
#pragma unroll
  for (int i = 0; i < RT; ++i) {
    const int r = 16 * i + (tid >> 4), c8 = (tid & 15) * 8, m = mb[i] + (tid >> 4);
    if (m < M) {
      T* g = gofs(dst, (unsigned int)m * (unsigned int)ld + (unsigned int)(col0 + c8));
      Frag<T> raw;
      unstage8(raw, tile + Tile<LT>::off(r, c8));
      if constexpr (NT) frag_store_nt(g, raw);
      else *reinterpret_cast<Frag<T>*>(g) = raw;
    }
  }
}

// the tile's rows of a row-major HBM matrix <- 0 (128 columns from col0), coalesced 16-byte stores
template <typename T, int RT>
__device__ __forceinline__ void zero_to_hbm(T* __restrict__ dst, int ld, int col0, const int (&mb)[RT], int M, int tid) {
  Frag<T> z;
  frag_zero(z);
#pragma unroll
  for (int i = 0; i < RT; ++i) {
    const int c8 = (tid & 15) * 8, m = mb[i] + (tid >> 4);
    if (m < M) *reinterpret_cast<Frag<T>*>(gofs(dst, (unsigned int)m * (unsigned int)ld + (unsigned int)(col0 + c8))) = z;
  }
}

#ifdef RG_STAMP
#define STAMP(i) do { unsigned long long t1__ = __builtin_amdgcn_s_memtime(); asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); \
    tacc[i] += t1__ - t0__; t0__ = t1__; } while (0)
#else
#define STAMP(i)
#endif

// DM: dropout mode -- 0 none, 1 p == 0.5 (one hash bit per element), 2 generic p (16-bit hash fields)
// CROSS: the decoder form (collapsed cross-attention stage between the two LayerNorms).  The encoder launches (three
// quarters of them) run the instantiation without it: its loads and registers were what spilled under dropout.
// SAVE: a training launch (y / y2 / h1 / rstd* saved for the backward); inference launches (the critic phase's encoder
// passes: most launches of a step) compile all of it away.
// RES: split residual stream (x = x + x_lo in, out / out_lo out; the LayerNorm outputs y / y2 keep their lo part in a free
// tile: y_lo parks in the ctx tile during the FFN, x_lo arrives in the y tile, out_lo leaves through the g-chunk tile).
// RT: 16-row tiles per work tile -- 4 (64 tokens) or 2 (32 tokens: half the accumulators and activation tiles per workgroup,
// so that three to four workgroups share a CU instead of two; the per-phase stamps show a wave at 2 per SIMD spending its
// time on exposed LDS / VALU latencies, not on the matrix pipe or the weight stream)
// PIPE (round 5; bf16, no saves, no cross stage, d_ff >= 256): the FFN chunk loop software-pipelined -- see the loop.
// (round 6, measured and NOT kept: this kernel alone without packed-f32 VALU instructions via the per-function target feature
// `__attribute__((target("no-packed-fp32-ops")))`.  The whole-library variant had shown the fused block 2 % faster and ffn_bwd_data 2 % slower
// (profiles/r06/ab/nopk_*); with the attribute on this kernel only, the always-inline device helpers -- compiled with the feature -- are no
// longer inline-compatible with it and become CALLS: 276 -> 605 us per inference launch, step 56.4 -> 70.0 ms, outputs bit-identical.)
template <typename T, int DM, bool CROSS, bool SAVE, bool RES = false, int RT = 4, bool PIPE = false>
__global__ __launch_bounds__(256, (sizeof(T) == 2 ? (RT == 2 ? 3 : 2) : ((std::is_same<T, x3>::value && RT == 2) ? 2 : 1)))
void post_attn_fwd_kernel(rg_post_attn_args a) {
  static_assert(!PIPE || (sizeof(T) == 2 && !SAVE && !CROSS && !RES && RT == 4), "PIPE: the bf16 inference launch");
  constexpr int FTM = 16 * RT;        // tokens per work tile
  constexpr int PL = FTM * FD * 2;    // bf16x3: bytes between the hi and the lo tile
  typedef typename LdsT<T, PL>::type LT;  // element type of the LDS tiles (T, or x3p: a hi and a lo bf16 tile)
#ifdef RG_STAMP
  unsigned long long tacc[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  unsigned long long t0__ = __builtin_amdgcn_s_memtime();
#endif
  // LDS: ctx tile (later: out staging) | x tile (later: g chunk) | y tile | params | row-stat exchange | [h1 chunk]
  constexpr int ACT_BYTES = FTM * Tile<LT>::LD * (int)sizeof(LT) * LdsT<T, PL>::PLANES;
  extern __shared__ __align__(16) unsigned char smem[];
  LT* Actx = reinterpret_cast<LT*>(smem);
  LT* Ag = reinterpret_cast<LT*>(smem + ACT_BYTES);
  LT* Ay = reinterpret_cast<LT*>(smem + 2 * ACT_BYTES);
  // tiles that are not matrix operands -- the residual x (parked in the g-chunk buffer) and the output on its way to HBM (in
  // the ctx tile): raw f32 in the bf16x3 tier (exact residual stream, no split work), the ordinary tile otherwise
  typedef typename ResT<T, PL>::type XT;
  XT* Ax = reinterpret_cast<XT*>(Ag);
  XT* Aout = reinterpret_cast<XT*>(Actx);
  constexpr bool KEEPY = std::is_same<T, x3>::value;     // the LayerNorm-1 output stays in registers as the FFN's residual (exact)
  float* prm = reinterpret_cast<float*>(smem + 3 * ACT_BYTES);           // 8*128 + dff floats
  float* redA = prm + 8 * FD + a.dff;                                     // [64][4]
  float* redB = redA + FTM * 4;
  // p == 0.5 dropout: nibble of hash bits -> 4 multipliers (0 or 1/(1-p)) from a 16-entry table: 2 address ops, one
  // ds_read_b128 and 4 multiplies per 4 elements instead of a bit extract, an AND and a multiply per element
  float* klut = redB + FTM * 4;                                          // [16][4]
  LT* Ah = reinterpret_cast<LT*>(klut + 64);                                // only when h1_save != NULL
  float *p_bo = prm, *p_g1 = prm + FD, *p_be1 = prm + 2 * FD, *p_b2 = prm + 3 * FD, *p_g2 = prm + 4 * FD,
        *p_be2 = prm + 5 * FD, *p_gc = prm + 6 * FD, *p_bec = prm + 7 * FD, *p_b1 = prm + 8 * FD;

  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // uniform: scalar address math
  const int li = lane & 15, lg = lane >> 4;
  const T* __restrict__ ctx = reinterpret_cast<const T*>(a.ctx);
  const T* __restrict__ x = reinterpret_cast<const T*>(a.x);
  const T* __restrict__ Wo = reinterpret_cast<const T*>(a.Wo);
  const T* __restrict__ W1 = reinterpret_cast<const T*>(a.W1);
  const T* __restrict__ W2 = reinterpret_cast<const T*>(a.W2);
  T* __restrict__ out = reinterpret_cast<T*>(a.out);
  const T* __restrict__ xlo = RES ? reinterpret_cast<const T*>(a.x_lo) : nullptr;
  T* __restrict__ outlo = RES ? reinterpret_cast<T*>(a.out_lo) : nullptr;
  T* __restrict__ ysave = SAVE ? reinterpret_cast<T*>(a.y_save) : nullptr;
  T* __restrict__ y2save = SAVE ? reinterpret_cast<T*>(a.y2_save) : nullptr;
  T* __restrict__ h1save = SAVE ? reinterpret_cast<T*>(a.h1_save) : nullptr;
  float* __restrict__ rstd1o = SAVE ? a.rstd1 : nullptr;
  float* __restrict__ rstd2o = SAVE ? a.rstd2 : nullptr;
  float* __restrict__ rstdco = SAVE ? a.rstd_c : nullptr;
  const int n0 = wave * 32;                 // this wave's 32 output features of every 128-wide block
  const int ntiles = (a.M + FTM - 1) / FTM;
  const int nchunk = a.dff / FD;
  DropCfg drop1 = make_drop(a.drop_p, a.seed_h1), drop2 = make_drop(a.drop_p, a.seed_out);
  if constexpr (DM == 2) { drop1.onebit = 0u; drop2.onebit = 0u; }

  // ---- once per workgroup: parameters -> LDS
  for (int i = tid; i < FD; i += 256) {
    p_bo[i] = a.bo[i]; p_g1[i] = a.g1[i]; p_be1[i] = a.be1[i]; p_b2[i] = a.b2[i]; p_g2[i] = a.g2[i]; p_be2[i] = a.be2[i];
    p_gc[i] = a.gc ? a.gc[i] : 1.f; p_bec[i] = a.bec ? a.bec[i] : 0.f;
  }
  for (int i = tid; i < a.dff; i += 256) p_b1[i] = a.b1[i];
  if (tid < 64) klut[tid] = (((tid >> 2) >> (tid & 3)) & 1) ? drop1.inv_keep : 0.f;
  const unsigned int rot0 = (4u * lg + 28u) & 31u, rot1 = (4u * lg + 12u) & 31u;   // hash bit 4 lg + j (16 + 4 lg + j) -> bit 4 + j

  WSet<T> wp, wq;                           // wp: Wo / W1 chunks, wq: W2 chunks
  Frag<T> cpre[RT], xpre[RT];               // ctx / x rows of the NEXT tile (staging prefetch)
  Frag<T> xlpre[RES ? RT : 1];              // ... and the lo part of x

  // ---- work-tile enumeration: work tile = blockIdx.x, += gridDim.x.  Plain: rows tile*64..  Compacted (a.live16):
  // 4 consecutive entries of the list of live 16-row tiles (balanced by construction: every
  // work tile is 64 live-ish rows wherever the padding sits).
  LiveWalk lw;                              // list entries by v_readlane (one vector load per 16 tiles), see rg_common.hip.h
  lw.init(a.live16, a.M);
  const int nwork = a.live16 ? (lw.nlive + RT - 1) / RT : ntiles;
  int cur = (int)blockIdx.x, kcur = 0;
  auto next_group = [&](int (&g)[RT]) -> bool {
    if (cur >= nwork) {
#pragma unroll
      for (int rt = 0; rt < RT; ++rt) g[rt] = a.M;      // absent: prefetch_rows clamps, nothing is stored
      return false;
    }
    if (!a.live16) {
#pragma unroll
      for (int rt = 0; rt < RT; ++rt) g[rt] = cur * FTM + 16 * rt;
    } else {
      lw.template group_n<RT>(kcur, g, a.M);
    }
    cur += gridDim.x;
    ++kcur;
    return true;
  };
  auto prefetch_rows = [&](const int (&g)[RT]) {        // rows >= M: clamped address, no branch (never stored)
#pragma unroll
    for (int i = 0; i < RT; ++i) {
      const int c8 = (tid & 15) * 8, m = min(g[i] + (tid >> 4), a.M - 1);
      load_frag(cpre[i], gofs(ctx, (unsigned int)(m * FD + c8)));
      load_frag(xpre[i], gofs(x, (unsigned int)(m * FD + c8)));
      if constexpr (RES) load_frag(xlpre[i], gofs(xlo, (unsigned int)(m * FD + c8)));
    }
  };
  int mb[RT], mbn[RT];
  bool have = next_group(mb);
  if (have) {
    load_wset(wp, Wo, FD, n0, 0, li, lg, a.w_packed);
    prefetch_rows(mb);
  }
  for (; have;) {
    const bool have_next = next_group(mbn);
    // ---- ctx and x tiles: registers -> LDS (x parks in the g-chunk buffer, free until the FFN)
#pragma unroll
    for (int i = 0; i < RT; ++i) {
      const int c = tid + 256 * i, r = c >> 4, c8 = (c & 15) * 8;
      stage8(Actx + Tile<LT>::off(r, c8), cpre[i]);
      stage8(Ax + Tile<XT>::off(r, c8), xpre[i]);
      if constexpr (RES) stage8(Ay + Tile<LT>::off(r, c8), xlpre[i]);     // the y tile is free until LayerNorm 1
    }
    float rm4[RT];
    bool any_live = false;
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
      const int m = mb[rt] + li;
      rm4[rt] = (a.rowmask && m < a.M) ? a.rowmask[m] : 1.f;
      any_live = any_live || rm4[rt] != 0.f;
    }
    if (a.rowmask && !a.live16 && __ballot(any_live) == 0ull) {
      // 64 padded positions: the block's output is out * rowmask = 0 whatever the arithmetic gives, and no gradient
      // comes back through these rows -- write the zeros (and finite placeholders for what backward reads) and move on
      zero_to_hbm<T>(out, FD, 0, mb, a.M, tid);
      if constexpr (RES) zero_to_hbm<T>(outlo, FD, 0, mb, a.M, tid);
      const bool cross = CROSS;
      if (ysave) zero_to_hbm<T>(ysave, FD, 0, mb, a.M, tid);
      if (cross && y2save) zero_to_hbm<T>(y2save, FD, 0, mb, a.M, tid);
      if (h1save)
        for (int ch = 0; ch < nchunk; ++ch) zero_to_hbm<T>(h1save, a.dff, ch * FD, mb, a.M, tid);
      if (tid < FTM && mb[0] + tid < a.M) {             // plain tiles only: FTM consecutive rows
        if (rstd1o) rstd1o[mb[0] + tid] = 0.f;
        if (rstd2o) rstd2o[mb[0] + tid] = 0.f;
        if (rstdco) rstdco[mb[0] + tid] = 0.f;
      }
      lds_barrier();                                    // the staged ctx / x tile of this iteration is dropped
      prefetch_rows(mbn);
#pragma unroll
      for (int rt = 0; rt < RT; ++rt) mb[rt] = mbn[rt];
      have = have_next;
      continue;
    }
    lds_barrier();
    STAMP(0);
    // ---- attention output projection (weights already in wp), bias folded into the accumulators
    f32x4 acc[2][RT];
    init_acc(acc, p_bo, n0, lg);
    mma_wset<T>(acc, wp, Actx, li, lg);
    load_wset(wp, W1, FD, n0, 0, li, lg, a.w_packed);   // prefetch FFN chunk 0 (hidden behind LN1)
    // + residual x (from LDS), LayerNorm 1 on the registers
    add_tile<XT>(acc, Ax, n0, li, lg);
    if constexpr (RES) add_tile<LT>(acc, Ay, n0, li, lg);          // x_lo
    STAMP(1);
    float rstd[RT];
    ln_regs<RT, KEEPY>(acc, rstd, p_g1, p_be1, redA, redB, a.eps, n0, wave, li, lg);
    // (past the barrier inside ln_regs every wave is done with the ctx tile, x and x_lo)
    if constexpr (RES) regs_to_tile_lo<LT>(acc, Actx, n0, li, lg);   // y_lo parks in the ctx tile (each lane: its own positions)
    if (rstd1o && wave == 0 && lg == 0) {
#pragma unroll
      for (int rt = 0; rt < RT; ++rt)
        if (mb[rt] + li < a.M) rstd1o[mb[rt] + li] = rstd[rt];
    }
    regs_to_tile<LT>(acc, Ay, n0, li, lg);
    STAMP(2);
    if constexpr (CROSS) {
      // collapsed decoder cross-attention: y2 = LayerNorm(y1 + o[b]); y1 is saved from its tile first
      if (ysave) { lds_barrier(); tile_to_hbm<T, true>(Ay, ysave, FD, 0, mb, a.M, tid); }
      // o rows: the (uniform) dropout / no-dropout switch sits outside the loops and the loads of a row tile are
      // issued together (inside the loops every (rt, ct, head) was a branch + load + wait: ~30 exposed L2 latencies
      // per tile, the decoder launches ran at half the rate of the encoder ones)
      int mrow[RT], brow[RT];
#pragma unroll
      for (int rt = 0; rt < RT; ++rt) { mrow[rt] = min(mb[rt] + li, a.M - 1); brow[rt] = mrow[rt] / a.L; }
      if (a.cross_s) {          // dropout: o = bo + sum_h s[m,h] * oh[b,h,:]   (H == P / 32 == 4 on this path)
        float sv[RT][4], bo4[2][4];
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) load4f(sv[rt], a.cross_s + (size_t)mrow[rt] * 4);
#pragma unroll
        for (int ct = 0; ct < 2; ++ct) load4f(bo4[ct], a.cross_bo + n0 + ct * 16 + 4 * lg);
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
          float w4[2][4][4];
#pragma unroll
          for (int ct = 0; ct < 2; ++ct)
#pragma unroll
            for (int hh = 0; hh < 4; ++hh)
              load4f(w4[ct][hh], a.cross_oh + ((size_t)brow[rt] * 4 + hh) * FD + n0 + ct * 16 + 4 * lg);
#pragma unroll
          for (int ct = 0; ct < 2; ++ct) {
            float y4[4];
            if constexpr (KEEPY) { y4[0] = acc[ct][rt][0]; y4[1] = acc[ct][rt][1]; y4[2] = acc[ct][rt][2]; y4[3] = acc[ct][rt][3]; }
            else load4t(y4, Ay + Tile<LT>::off(rt * 16 + li, n0 + ct * 16 + 4 * lg));     // the ROUNDED y1, as the unfused path sees it
            if constexpr (RES) {
              float yl[4];
              load4t(yl, Actx + Tile<LT>::off(rt * 16 + li, n0 + ct * 16 + 4 * lg));
#pragma unroll
              for (int r = 0; r < 4; ++r) y4[r] += yl[r];
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              float o = bo4[ct][r];
#pragma unroll
              for (int hh = 0; hh < 4; ++hh) o += sv[rt][hh] * w4[ct][hh][r];
              acc[ct][rt][r] = y4[r] + o;
            }
          }
        }
      } else {
        float o4[RT][2][4];
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
#pragma unroll
          for (int ct = 0; ct < 2; ++ct) load4f(o4[rt][ct], a.o_bcast + (size_t)brow[rt] * FD + n0 + ct * 16 + 4 * lg);
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
#pragma unroll
          for (int ct = 0; ct < 2; ++ct) {
            float y4[4];
            if constexpr (KEEPY) { y4[0] = acc[ct][rt][0]; y4[1] = acc[ct][rt][1]; y4[2] = acc[ct][rt][2]; y4[3] = acc[ct][rt][3]; }
            else load4t(y4, Ay + Tile<LT>::off(rt * 16 + li, n0 + ct * 16 + 4 * lg));
            if constexpr (RES) {
              float yl[4];
              load4t(yl, Actx + Tile<LT>::off(rt * 16 + li, n0 + ct * 16 + 4 * lg));
#pragma unroll
              for (int r = 0; r < 4; ++r) y4[r] += yl[r];
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[ct][rt][r] = y4[r] + o4[rt][ct][r];
          }
      }
      lds_barrier();                                    // ysave copy done before the tile is overwritten
      ln_regs<RT, KEEPY>(acc, rstd, p_gc, p_bec, redA, redB, a.eps, n0, wave, li, lg);
      if (rstdco && wave == 0 && lg == 0) {
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
          if (mb[rt] + li < a.M) rstdco[mb[rt] + li] = rstd[rt];
      }
      regs_to_tile<LT>(acc, Ay, n0, li, lg);
      if constexpr (RES) regs_to_tile_lo<LT>(acc, Actx, n0, li, lg);
    }
    f32x4 yres[KEEPY ? 2 : 1][KEEPY ? RT : 1];
    if constexpr (KEEPY) {
#pragma unroll
      for (int ct = 0; ct < 2; ++ct)
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) yres[ct][rt] = acc[ct][rt];
    }
    lds_barrier();                                      // y tile complete (and x no longer needed in Ag)
    {
      constexpr bool cross = CROSS;
      if (cross ? (y2save != nullptr) : (ysave != nullptr)) tile_to_hbm<T, true>(Ay, cross ? y2save : ysave, FD, 0, mb, a.M, tid);
    }
    STAMP(3);
    // word index of this lane ROW's row tile (row tile lg, row li) in the two dropout index spaces: (row * d_ff + column) >> 5 with
    // column = chunk * 128 + 32 * wave + 4 lg + j (j < 20) and (row * 128 + 32 * wave + 4 lg + j) >> 5
    // (bf16 kernels only: in the f32-storage kernels hipcc turns the select over lg into lane-divergent branches whose results pass through
    // AGPR copies under a narrowed exec -- the pattern the ISA screen refuses)
    constexpr bool SHARE = DM == 1 && RT == 4 && sizeof(T) == 2;
    unsigned int hrow1 = 0u, hrow2 = 0u;
    if constexpr (SHARE) {
      const int m01 = (lg & 1) ? mb[1] : mb[0], m23 = (lg & 1) ? mb[RT == 4 ? 3 : 1] : mb[RT == 4 ? 2 : 0];
      const unsigned int mrow_lg = (unsigned int)(((lg & 2) ? m23 : m01) + li);
      hrow1 = mrow_lg * ((unsigned int)a.dff >> 5) + (unsigned int)wave;
      hrow2 = mrow_lg * 4u + (unsigned int)wave;
    }
    // ---- FFN: stream d_ff in 128-wide chunks; the second GEMM accumulates across chunks
    f32x4 acc2[2][RT];
    init_acc(acc2, p_b2, n0, lg);
    if constexpr (PIPE) {
      // Software-pipelined form (MEASURED: no gain -- see the end of this comment).  In the plain loop below a wave alternates between a matrix phase (32 MFMAs) and a VALU phase
      // (dropout + GELU of 32 elements per lane: ~260 VALU instructions, 64 of them half-rate transcendentals) with a barrier between them, so
      // the matrix pipe idles through every VALU phase of the wave and the counters show the two waves of a SIMD adding their
      // phases up rather than overlapping them (SQ_VALU_MFMA_COEXEC_CYCLES = 14 % of the matrix pipe's busy cycles).  Here the
      // product that does NOT depend on the current epilogue -- out += g(ch - 1) W2 (g chunks alternate between the x tile and the ctx
      // tile, both free during the FFN) -- sits in the SAME straight-line block as the epilogue of chunk ch, where the scheduler
      // interleaves them.  (h1 of chunk ch + 1 into a second accumulator set as well: 256 VGPRs with 66 - 84 spilled; not kept.)
      // Outcome at the bench shape (profiles/r05/ab/post_attn_pipelined_ffn.txt): left to the scheduler the matrix instructions
      // all moved to the front of the block (no change, 10.9 -> 11.0 ms per step); in eight slices of 4 MFMAs + one epilogue piece
      // 10.9 -> 10.95; one MFMA per half GELU pair (this code) 10.9 -> 11.2.  The matrix pipe's 21 % of the SIMD time is not what
      // the kernel waits for in particular: matrix and vector cycles largely ADD on a gfx950 SIMD (tools/peaks.hip: one MFMA hides ~4
      // cycles of the same wave's VALU work, 44 % of its own cycles with a second wave) -- ~196 full-rate + 64 half-rate vector
      // instructions (~1 400 cycles) and 64 MFMAs (~1 300) per chunk and wave, whatever their order.  Same products, same order of accumulation, same
      // dropout words: bit-identical to the plain loop (tests/test_fused256_gpu.py::test_pipelined_ffn_loop_is_bit_identical).
      auto gbuf = [&](int c) -> LT* { return (c & 1) ? Actx : Ag; };
      // one (feature tile ect, row tile ert) piece of the epilogue of chunk ch: dropout before the GELU (quirk Q4), GELU, store to the g tile
      float kq[2][4];                                     // dropout multipliers of row tile ert: [0] feature tile 0, [1] feature tile 1
      auto epi_piece = [&](f32x4 (&h)[2][RT], int ch, LT* gdst, int ert, int ect) {
        if constexpr (DM != 0) {
          if (ect == 0) {
            const unsigned int rb = (unsigned int)(mb[ert] + li) * (unsigned int)a.dff + (unsigned int)(ch * FD + n0 + 4 * lg);
            if constexpr (DM == 1) {
              const unsigned int w = rg_hash(drop1.seed, rb >> 5);
              load4f(kq[0], reinterpret_cast<const float*>(reinterpret_cast<const char*>(klut) + (__builtin_amdgcn_alignbit(w, w, rot0) & 0xF0u)));
              load4f(kq[1], reinterpret_cast<const float*>(reinterpret_cast<const char*>(klut) + (__builtin_amdgcn_alignbit(w, w, rot1) & 0xF0u)));
            } else {
              rg_keep4_pair(drop1, rb, kq[0], kq[1]);
            }
          }
#pragma unroll
          for (int r = 0; r < 4; ++r) h[ect][ert][r] *= kq[ect][r];
        }
#pragma unroll
        for (int r = 0; r < 4; r += 2) {
          const f32x2 gg = gelu2_fast((f32x2){h[ect][ert][r], h[ect][ert][r + 1]});
          h[ect][ert][r] = gg.x;
          h[ect][ert][r + 1] = gg.y;
        }
        float t[4] = {h[ect][ert][0], h[ect][ert][1], h[ect][ert][2], h[ect][ert][3]};
        store4(gdst + Tile<LT>::off(ert * 16 + li, n0 + ect * 16 + 4 * lg), t);
      };
      // ---- chunk 0 (wp = W1 chunk 0, prefetched behind LayerNorm 1): nothing to run beside its epilogue yet
      init_acc(acc, p_b1, n0, lg);
      mma_wset<T>(acc, wp, Ay, li, lg);
      load_wset(wp, W1 + (unsigned int)(FD * FD), FD, n0, 0, li, lg, a.w_packed);
      load_wset(wq, W2, a.dff, n0, 0, li, lg, a.w_packed);
#pragma unroll
      for (int u = 0; u < 2 * RT; ++u) epi_piece(acc, 0, gbuf(0), u >> 1, u & 1);
      lds_barrier();
#pragma unroll 1
      for (int ch = 1; ch < nchunk; ++ch) {
        init_acc(acc, p_b1 + ch * FD, n0, lg);
        mma_wset<T>(acc, wp, Ay, li, lg);                                    // h1 of chunk ch
        load_wset(wp, (ch + 1 < nchunk) ? W1 + (unsigned int)(ch + 1) * (FD * FD) : Wo, FD, n0, 0, li, lg, a.w_packed);
        __builtin_amdgcn_sched_barrier(0);
        // out += g(ch - 1) . W2[:, chunk ch - 1]^T in eight slices of 4 MFMAs (k-step u >> 1, row tiles 2 (u & 1), + 1), each followed
        // by one piece of the epilogue of chunk ch; nothing crosses a slice boundary, so every group of 4 matrix instructions runs
        // under ~30 VALU instructions of the same wave (the fragments of the next slice are read a slice ahead)
        const LT* gprev = gbuf(ch - 1);
        LT* gcur = gbuf(ch);
        typename OpT<T>::type af[2][2];
        load_frag(af[0][0], gprev + Tile<LT>::off(li, 8 * lg));
        load_frag(af[0][1], gprev + Tile<LT>::off(16 + li, 8 * lg));
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const int ks = u >> 1, rh = (u & 1) * 2;
          if (u < 7) {
            const int ksn = (u + 1) >> 1, rhn = ((u + 1) & 1) * 2;
            load_frag(af[(u + 1) & 1][0], gprev + Tile<LT>::off(rhn * 16 + li, ksn * 32 + 8 * lg));
            load_frag(af[(u + 1) & 1][1], gprev + Tile<LT>::off((rhn + 1) * 16 + li, ksn * 32 + 8 * lg));
          }
          // ONE matrix instruction, then half a GELU pair (>= 40 cycles of VALU / transcendental work against the 16 the matrix pipe
          // is busy): a wave issues in order, so matrix instructions back to back would stall it on the pipe before its first VALU one
          const int ert = u >> 1, ect = u & 1;
          if constexpr (DM != 0) {
            if (ect == 0) {
              const unsigned int rb = (unsigned int)(mb[ert] + li) * (unsigned int)a.dff + (unsigned int)(ch * FD + n0 + 4 * lg);
              if constexpr (DM == 1) {
                const unsigned int w = rg_hash(drop1.seed, rb >> 5);
                load4f(kq[0], reinterpret_cast<const float*>(reinterpret_cast<const char*>(klut) + (__builtin_amdgcn_alignbit(w, w, rot0) & 0xF0u)));
                load4f(kq[1], reinterpret_cast<const float*>(reinterpret_cast<const char*>(klut) + (__builtin_amdgcn_alignbit(w, w, rot1) & 0xF0u)));
              } else {
                rg_keep4_pair(drop1, rb, kq[0], kq[1]);
              }
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[ect][ert][r] *= kq[ect][r];
          }
          f32x2 xa = (f32x2){acc[ect][ert][0], acc[ect][ert][1]}, xb = (f32x2){acc[ect][ert][2], acc[ect][ert][3]};
          __builtin_amdgcn_sched_barrier(0);
          mma(wq.f[ks][0], af[u & 1][0], acc2[0][rh]);
          __builtin_amdgcn_sched_barrier(0);
          const f32x2 ea = gelu2_fast_a(xa);
          __builtin_amdgcn_sched_barrier(0);
          mma(wq.f[ks][1], af[u & 1][0], acc2[1][rh]);
          __builtin_amdgcn_sched_barrier(0);
          xa = gelu2_fast_b(xa, ea);
          __builtin_amdgcn_sched_barrier(0);
          mma(wq.f[ks][0], af[u & 1][1], acc2[0][rh + 1]);
          __builtin_amdgcn_sched_barrier(0);
          const f32x2 eb = gelu2_fast_a(xb);
          __builtin_amdgcn_sched_barrier(0);
          mma(wq.f[ks][1], af[u & 1][1], acc2[1][rh + 1]);
          __builtin_amdgcn_sched_barrier(0);
          xb = gelu2_fast_b(xb, eb);
          float t[4] = {xa.x, xa.y, xb.x, xb.y};
          store4(gcur + Tile<LT>::off(ert * 16 + li, n0 + ect * 16 + 4 * lg), t);
          __builtin_amdgcn_sched_barrier(0);
        }
        load_wset(wq, W2, a.dff, n0, ch * FD, li, lg, a.w_packed);
        lds_barrier();                                                       // g(ch) visible; every reader of g(ch - 1) is done
      }
      mma_wset<T>(acc2, wq, gbuf(nchunk - 1), li, lg);
    } else {
#pragma unroll 1
    for (int ch = 0; ch < nchunk; ++ch) {
      load_wset(wq, W2, a.dff, n0, ch * FD, li, lg, a.w_packed);    // needed after the GELU below
      init_acc(acc, p_b1 + ch * FD, n0, lg);
      mma_wset<T>(acc, wp, Ay, li, lg);                 // h1 chunk = y . W1[chunk]^T + b1
      // next weight set: W1 chunk ch+1, or Wo for the next tile -- ONE unconditional load sequence from a selected
      // pointer (loads under a branch made hipcc drain vmcnt(0) at the join: the W2 fragments just issued above, i.e.
      // one exposed L2 latency per chunk; Wo is fetched needlessly after a workgroup's last tile, 32 KB once)
      load_wset(wp, (ch + 1 < nchunk) ? W1 + (unsigned int)(ch + 1) * (FD * FD) : Wo, FD, n0, 0, li, lg, a.w_packed);
      STAMP(4);
      if (ch > 0) lds_barrier();                        // previous chunk's readers of Ag / Ah are done
      STAMP(5);
      if constexpr (DM != 0) {  // dropout BEFORE the GELU (transformer.py:182-184, quirk Q4)
        // p == 0.5, four row tiles: the hash word of row tile rt is the same in the four lane rows lg -- lane row lg computes the word of
        // row tile lg only and the four are exchanged (rg_allgather_rows): one hash per lane and chunk instead of four
        unsigned int wv[4];
        if constexpr (SHARE) rg_allgather_rows(rg_hash(drop1.seed, hrow1 + 4u * (unsigned int)ch), wv);
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
          const unsigned int rb = (unsigned int)(mb[rt] + li) * (unsigned int)a.dff + (unsigned int)(ch * FD + n0 + 4 * lg);
          if constexpr (DM == 1) {       // rb & 31 == 4*lg: both feature tiles of this lane sit in one hash word
            const unsigned int w = SHARE ? wv[rt & 3] : rg_hash(drop1.seed, rb >> 5);
            float k0[4], k1[4];
            load4f(k0, reinterpret_cast<const float*>(reinterpret_cast<const char*>(klut) + (__builtin_amdgcn_alignbit(w, w, rot0) & 0xF0u)));
            load4f(k1, reinterpret_cast<const float*>(reinterpret_cast<const char*>(klut) + (__builtin_amdgcn_alignbit(w, w, rot1) & 0xF0u)));
#pragma unroll
            for (int r = 0; r < 4; ++r) { acc[0][rt][r] *= k0[r]; acc[1][rt][r] *= k1[r]; }
          } else {
            float k0[4], k1[4];
            rg_keep4_pair(drop1, rb, k0, k1);
#pragma unroll
            for (int r = 0; r < 4; ++r) { acc[0][rt][r] *= k0[r]; acc[1][rt][r] *= k1[r]; }
          }
        }
      }
      if (h1save) regs_to_tile<LT>(acc, Ah, n0, li, lg);
#pragma unroll
      for (int ct = 0; ct < 2; ++ct)
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
          if constexpr (Precise<T>::value) {
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[ct][rt][r] = gelu_t<true>(acc[ct][rt][r]);
          } else {
#pragma unroll
            for (int r = 0; r < 4; r += 2) {
              const f32x2 gg = gelu2_fast((f32x2){acc[ct][rt][r], acc[ct][rt][r + 1]});
              acc[ct][rt][r] = gg.x;
              acc[ct][rt][r + 1] = gg.y;
            }
          }
        }
      regs_to_tile<LT>(acc, Ag, n0, li, lg);
      STAMP(6);
      lds_barrier();
      STAMP(7);
      if (h1save) tile_to_hbm<T, true>(Ah, h1save, a.dff, ch * FD, mb, a.M, tid);
      mma_wset<T>(acc2, wq, Ag, li, lg);                // out += g . W2[:, chunk]^T
      STAMP(8);
    }
    }
    // prefetch the next tile's ctx / x rows while the second LayerNorm runs (unconditional: see next_group)
    prefetch_rows(mbn);
    if constexpr (DM != 0) {    // dropout on the l2 output, before the residual (transformer.py:186-188)
      unsigned int wv[4];
      if constexpr (SHARE) rg_allgather_rows(rg_hash(drop2.seed, hrow2), wv);
#pragma unroll
      for (int rt = 0; rt < RT; ++rt) {
        const unsigned int rb = (unsigned int)(mb[rt] + li) * (unsigned int)FD + (unsigned int)(n0 + 4 * lg);
        if constexpr (DM == 1) {
          const unsigned int w = SHARE ? wv[rt & 3] : rg_hash(drop2.seed, rb >> 5);
          float k0[4], k1[4];
          load4f(k0, reinterpret_cast<const float*>(reinterpret_cast<const char*>(klut) + (__builtin_amdgcn_alignbit(w, w, rot0) & 0xF0u)));
          load4f(k1, reinterpret_cast<const float*>(reinterpret_cast<const char*>(klut) + (__builtin_amdgcn_alignbit(w, w, rot1) & 0xF0u)));
#pragma unroll
          for (int r = 0; r < 4; ++r) { acc2[0][rt][r] *= k0[r]; acc2[1][rt][r] *= k1[r]; }
        } else {
          float k0[4], k1[4];
          rg_keep4_pair(drop2, rb, k0, k1);
#pragma unroll
          for (int r = 0; r < 4; ++r) { acc2[0][rt][r] *= k0[r]; acc2[1][rt][r] *= k1[r]; }
        }
      }
    }
    // ---- + residual y (LDS), LayerNorm 2 on the registers, * rowmask, out via the (free) ctx tile
    if constexpr (KEEPY) {
#pragma unroll
      for (int ct = 0; ct < 2; ++ct)
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) acc2[ct][rt] += yres[ct][rt];
    } else {
      add_tile<LT>(acc2, Ay, n0, li, lg);
    }
    if constexpr (RES) add_tile<LT>(acc2, Actx, n0, li, lg);       // y_lo
    ln_regs<RT, KEEPY>(acc2, rstd, p_g2, p_be2, redA, redB, a.eps, n0, wave, li, lg);
    if (rstd2o && wave == 0 && lg == 0) {
#pragma unroll
      for (int rt = 0; rt < RT; ++rt)
        if (mb[rt] + li < a.M) rstd2o[mb[rt] + li] = rstd[rt];
    }
#pragma unroll
    for (int ct = 0; ct < 2; ++ct)
#pragma unroll
      for (int rt = 0; rt < RT; ++rt)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc2[ct][rt][r] *= rm4[rt];
    regs_to_tile<XT>(acc2, Aout, n0, li, lg);
    if constexpr (RES) regs_to_tile_lo<LT>(acc2, Ag, n0, li, lg);  // (the barrier inside ln_regs: the last g chunk has been consumed)
    STAMP(9);
    lds_barrier();
    tile_to_hbm<T>(Aout, out, FD, 0, mb, a.M, tid);
    if constexpr (RES) tile_to_hbm<T>(Ag, outlo, FD, 0, mb, a.M, tid);
    lds_barrier();                                      // before the next tile overwrites Actx / Ag / Ay
    STAMP(10);
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) mb[rt] = mbn[rt];
    have = have_next;
  }
  if (a.live16) {
    // the padded row tiles (listed from the far end of live16): out rows = 0 -- and zeros / finite placeholders in
    // everything a backward pass reads --, 4 row tiles per step
    const int nrt = (a.M + 15) >> 4, ndead = nrt - a.live16[0];
    constexpr bool cross = CROSS;
    for (int j = RT * (int)blockIdx.x; j < ndead; j += RT * (int)gridDim.x) {
      int md[RT];
#pragma unroll
      for (int rt = 0; rt < RT; ++rt) md[rt] = j + rt < ndead ? a.live16[nrt - (j + rt)] * 16 : a.M;
      zero_to_hbm<T>(out, FD, 0, md, a.M, tid);
      if constexpr (RES) zero_to_hbm<T>(outlo, FD, 0, md, a.M, tid);
      if (a.skip_dead_saves) continue;      // the backward is list-driven too: it never reads the padded tiles' saves
      if (ysave) zero_to_hbm<T>(ysave, FD, 0, md, a.M, tid);
      if (cross && y2save) zero_to_hbm<T>(y2save, FD, 0, md, a.M, tid);
      if (h1save)
        for (int ch = 0; ch < nchunk; ++ch) zero_to_hbm<T>(h1save, a.dff, ch * FD, md, a.M, tid);
      if (tid < FTM) {
        const int m = md[tid >> 4] + (tid & 15);
        if (m < a.M) {
          if (rstd1o) rstd1o[m] = 0.f;
          if (rstd2o) rstd2o[m] = 0.f;
          if (rstdco) rstdco[m] = 0.f;
        }
      }
    }
  }
#ifdef RG_STAMP
  if (a.rstd_c == nullptr && a.o_bcast == nullptr && a.y2_save != nullptr && (tid & 63) == 0) {
    unsigned long long* dbg = reinterpret_cast<unsigned long long*>(a.y2_save) + (size_t)(blockIdx.x * 4 + wave) * 12;
    for (int i = 0; i < 12; ++i) dbg[i] = tacc[i];
  }
#endif
}

// bytes of one [rows x 128] activation tile in LDS: bf16 swizzled rows; f32 padded rows; bf16x3 a hi and a lo bf16 tile
static int act_tile_bytes(int dtype, int rows) {
  return dtype == RG_BF16 ? rows * FD * 2 : (dtype == RG_X3 ? rows * FD * 2 * 2 : rows * FLD * 4);
}

int rg_post_attn_fwd256(const rg_post_attn_args* a, int dtype, hipStream_t s);      // fused256.hip: d_model == n_heads * 32 == 256
int rg_post_attn_fwd_w8_try(const rg_post_attn_args* a, int dtype, hipStream_t s, int* rc);      // fused128w8.hip

extern "C" int rg_post_attn_fwd(const rg_post_attn_args* a, int dtype, void* stream) {
  if (!a || a->M <= 0) return 0;
  if (a->d == 256) return rg_post_attn_fwd256(a, dtype, (hipStream_t)stream);
  {
    int rc8 = 0;                      // RG_PA8=1: the eight-wave prototype takes the encoder-inference launches (fused128w8.hip, A/B timing)
    if (rg_post_attn_fwd_w8_try(a, dtype, (hipStream_t)stream, &rc8)) return rc8;
  }
  if (a->d != FD || a->P != FD || (a->dff % FD) != 0 || a->dff <= 0)
    return rg_set_error_msg(RG_ERR_UNSUPPORTED, "post_attn_fwd: needs d_model == n_heads*32 == 128 and d_ff % 128 == 0");
  if ((long long)a->M * a->dff * (dtype == RG_BF16 ? 2 : 4) >= (1ll << 32))
    return rg_set_error_msg(RG_ERR_UNSUPPORTED, "post_attn_fwd: M * d_ff * element size must be below 4 GiB (32-bit offsets)");
  if ((a->o_bcast || a->cross_s) && a->L <= 0) return rg_set_error_msg(RG_ERR_INVALID, "post_attn_fwd: cross stage needs L");
  if (a->cross_s && (!a->cross_oh || !a->cross_bo || a->H != 4)) return rg_set_error_msg(RG_ERR_INVALID, "post_attn_fwd: cross_s needs cross_oh, cross_bo, H == P / 32 == 4");
  hipStream_t s = (hipStream_t)stream;
  // work tile: 64 tokens.  The 32-token form (kernel template RT = 2: three workgroups per CU instead of two) was built and
  // measured SLOWER at the bench shape -- 277 -> 292 us (encoder inference), 337 -> 365, 413 -> 467 (decoder training): the
  // per-tile fixed costs (14 barriers, two LayerNorm exchanges, the weight sets) double per row, which the third wave per
  // SIMD does not buy back.  It is only instantiated in -DRG_PA_RT2 builds (RG_PA_RT=2 selects it there) for A/B timing.
  // bf16x3: the 64-token form holds 96 KB of split tiles -- ONE workgroup per CU, one wave per SIMD, nothing to overlap the
  // MFMA, VALU and LDS phases of a tile with; the 32-token form (48 KB) puts two workgroups on a CU (RG_X3_PA_RT=4: the 64-token
  // form, for A/B timing)
  static const int x3_rt = [] { const char* e = getenv("RG_X3_PA_RT"); return (e && atoi(e) == 4) ? 4 : 2; }();
#ifdef RG_PA_RT2
  static const int rt_env = [] { const char* e = getenv("RG_PA_RT"); return e ? atoi(e) : 0; }();
  const int rt = dtype == RG_X3 ? x3_rt : (dtype != RG_BF16 ? 4 : (rt_env == 2 ? 2 : 4));
#else
  const int rt = dtype == RG_X3 ? x3_rt : 4;
#endif
  const int ftm = 16 * rt;
  const int ntiles = (a->M + ftm - 1) / ftm;
  const int act = act_tile_bytes(dtype, ftm);
  const int smem = 3 * act + (8 * FD + a->dff) * 4 + 2 * ftm * 4 * 4 + 64 * 4 + (a->h1_save ? act : 0);
  const int per_cu = (160 * 1024) / smem;
  const int cap_cu = dtype == RG_X3 ? 2 : 3;            // (bf16 RT == 2: 3 waves per SIMD by registers; bf16x3: 256 VGPRs, two)
  int grid = 256 * (per_cu < 1 ? 1 : (per_cu > cap_cu ? cap_cu : per_cu));
  if (grid > ntiles) grid = ntiles;
  const int dm = a->drop_p <= 0.f ? 0 : (a->drop_p == 0.5f ? 1 : 2);
  const bool cross = a->o_bcast || a->cross_s;
  const bool save = a->y_save || a->y2_save || a->h1_save || a->rstd1 || a->rstd2 || a->rstd_c;
  const bool res = a->x_lo || a->out_lo;
  if (res && (!a->x_lo || !a->out_lo || dtype != RG_BF16))
    return rg_set_error_msg(RG_ERR_INVALID, "post_attn_fwd: the split residual stream needs x_lo AND out_lo, bf16 tier");
  // RG_PA_PIPE=1: the software-pipelined FFN loop (bf16 inference launches: no saves, no cross stage, one residual stream, d_ff >= 256) --
  // bit-identical, measured 0 ... 3 % SLOWER than the plain loop, kept for A/B timing
  static const int pipe_on = [] { const char* e = getenv("RG_PA_PIPE"); return e ? atoi(e) : 0; }();
  if (pipe_on && dtype == RG_BF16 && !cross && !save && !res && a->dff >= 2 * FD) {
#define RG_PAP(DM)                                                                                                        \
  do {                                                                                                                    \
    hipFuncSetAttribute(reinterpret_cast<const void*>(post_attn_fwd_kernel<__bf16, DM, false, false, false, 4, true>), hipFuncAttributeMaxDynamicSharedMemorySize, smem); \
    hipLaunchKernelGGL((post_attn_fwd_kernel<__bf16, DM, false, false, false, 4, true>), dim3(grid), dim3(256), smem, s, *a); \
  } while (0)
    if (dm == 0) RG_PAP(0);
    else if (dm == 1) RG_PAP(1);
    else RG_PAP(2);
#undef RG_PAP
    RG_CHECK_LAUNCH();
    return 0;
  }
#define RG_PA4(T, DM, C, S, R, RTV)                                                                                       \
  do {                                                                                                                    \
    hipFuncSetAttribute(reinterpret_cast<const void*>(post_attn_fwd_kernel<T, DM, C, S, R, RTV>), hipFuncAttributeMaxDynamicSharedMemorySize, smem); \
    hipLaunchKernelGGL((post_attn_fwd_kernel<T, DM, C, S, R, RTV>), dim3(grid), dim3(256), smem, s, *a);                  \
  } while (0)
#ifdef RG_PA_RT2
#define RG_PA3(T, DM, C, S, R)                                                              \
  do {                                                                                      \
    if constexpr (sizeof(T) == 2 || std::is_same<T, x3>::value) { if (rt == 2) RG_PA4(T, DM, C, S, R, 2); else RG_PA4(T, DM, C, S, R, 4); } \
    else RG_PA4(T, DM, C, S, R, 4);                                                         \
  } while (0)
#else
#define RG_PA3(T, DM, C, S, R)                                                              \
  do {                                                                                      \
    if constexpr (std::is_same<T, x3>::value) { if (rt == 2) RG_PA4(T, DM, C, S, R, 2); else RG_PA4(T, DM, C, S, R, 4); } \
    else RG_PA4(T, DM, C, S, R, 4);                                                         \
  } while (0)
#endif
#define RG_PA2(T, DM, C, S)                                                                       \
  do {                                                                                            \
    if constexpr (sizeof(T) == 2) { if (res) RG_PA3(T, DM, C, S, true); else RG_PA3(T, DM, C, S, false); } \
    else RG_PA3(T, DM, C, S, false);                                                              \
  } while (0)
#define RG_PA(T, DM)                                  \
  do {                                                \
    if (cross) { if (save) RG_PA2(T, DM, true, true); else RG_PA2(T, DM, true, false); }      \
    else { if (save) RG_PA2(T, DM, false, true); else RG_PA2(T, DM, false, false); }          \
  } while (0)
#define RG_PA_T(T)                   \
  do {                               \
    if (dm == 0) RG_PA(T, 0);        \
    else if (dm == 1) RG_PA(T, 1);   \
    else RG_PA(T, 2);                \
  } while (0)
  if (dtype == RG_BF16) RG_PA_T(__bf16);
  else if (dtype == RG_F32) RG_PA_T(float);
  else if (dtype == RG_X3) RG_PA_T(x3);
  else return rg_set_error_msg(RG_ERR_INVALID, "post_attn_fwd: bad dtype");
#undef RG_PA_T
#undef RG_PA
#undef RG_PA2
#undef RG_PA3
#undef RG_PA4
  RG_CHECK_LAUNCH();
  return 0;
}

// ---------------------------------------------------------------------------------------------------------------------
// FFN block, backward data path (rg_ffn_bwd_data): same tiling, weight streaming and LDS tiles as the forward kernel
// above.  Per 64-token tile: dl2 / dz rows and the h1 chunks arrive through coalesced 16-byte loads issued one stage
// ahead (registers -> LDS), the chunk's dg = dl2 . W2t[chunk]^T sits in the accumulators, is multiplied by gelu'(h1)
// (and the dropout mask read back from h1 != 0), goes to LDS as the B operand of the second product and to HBM as dh1
// for the weight-gradient kernel; dy accumulates across chunks and leaves with the residual gradient added.
// LDS: 4 tiles (dl2 -> dy staging | dz | dh1 chunk | h1 chunk) = 64 KB (bf16): two workgroups per CU.
// LNF: the LayerNorm backward (+ row mask + output dropout) in front of the block is computed here, on the rows as they
// arrive -- the staging layout (16 lanes x 8 features per row) is rg_ln_bwd's; dz never leaves the chip.
template <typename T, bool LNF>
__global__ __launch_bounds__(256, (sizeof(T) == 2 ? 2 : 1)) void ffn_bwd_data_kernel(rg_ffn_bwd_args a) {
  constexpr int PL = FT_M * FD * 2;
  typedef typename LdsT<T, PL>::type LT;
  constexpr int ACT_BYTES = FT_M * Tile<LT>::LD * (int)sizeof(LT) * LdsT<T, PL>::PLANES;
  extern __shared__ __align__(16) unsigned char smem[];
  LT* Adl = reinterpret_cast<LT*>(smem);
  // dz (the residual gradient), the h1 chunk (read element-wise only) and dy on its way out are not matrix operands: raw f32
  // tiles in the bf16x3 tier (rg_common.hip.h x3r), the ordinary tile otherwise
  typedef typename ResT<T, PL>::type XT;
  XT* Adz = reinterpret_cast<XT*>(smem + ACT_BYTES);
  LT* Adh = reinterpret_cast<LT*>(smem + 2 * ACT_BYTES);
  XT* Ah = reinterpret_cast<XT*>(smem + 3 * ACT_BYTES);
  XT* Ady = reinterpret_cast<XT*>(smem);                 // (the dl2 tile's space, after its last read)
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // uniform: scalar address math
  const int li = lane & 15, lg = lane >> 4;
  const T* __restrict__ dl2 = reinterpret_cast<const T*>(LNF ? a.ln_dout : a.dl2);      // LNF: dout and the saved LN output
  const T* __restrict__ dz = reinterpret_cast<const T*>(LNF ? a.ln_out : a.dz);
  T* __restrict__ dl2o = reinterpret_cast<T*>(a.dl2_out);
  const T* __restrict__ h1 = reinterpret_cast<const T*>(a.h1);
  const T* __restrict__ W2t = reinterpret_cast<const T*>(a.W2t);
  const T* __restrict__ W1t = reinterpret_cast<const T*>(a.W1t);
  T* __restrict__ dh1 = reinterpret_cast<T*>(a.dh1);
  T* __restrict__ dy = reinterpret_cast<T*>(a.dy);
  const int n0 = wave * 32;
  const int ntiles = (a.M + FT_M - 1) / FT_M;
  const int nchunk = a.dff / FD;
  const float nz = a.nz_scale;

  WSet<T> wp, wq;                           // wp: W2t chunks (first product), wq: W1t chunks (second product)
  Frag<T> cpre[4], xpre[4], hpre[4];        // dl2 / dz rows of the next tile, h1 rows of the next chunk
  float rs_pre[4], rm_pre[4];               // LNF: rstd and row mask of this thread's 4 rows of the next tile
  float dg[8], db[8];                       // LNF: this thread's column sums (features 8 (tid & 15) ..)
  const DropCfg ldrop = make_drop(a.ln_drop_p, a.ln_drop_seed);
  float* lnp = reinterpret_cast<float*>(smem + 4 * ACT_BYTES);      // LNF: gamma | beta | 1 / gamma
  if constexpr (LNF) {
#pragma unroll
    for (int j = 0; j < 8; ++j) { dg[j] = 0.f; db[j] = 0.f; }
    for (int i = tid; i < FD; i += 256) { lnp[i] = a.ln_gamma[i]; lnp[FD + i] = a.ln_beta[i]; lnp[2 * FD + i] = 1.f / a.ln_gamma[i]; }
  }

  LiveWalk lw;
  lw.init(a.live16, a.M);
  const int nwork = a.live16 ? (lw.nlive + 3) >> 2 : ntiles;
  int cur = (int)blockIdx.x, kcur = 0;
  auto next_group = [&](int (&g)[4]) -> bool {
    if (cur >= nwork) {
#pragma unroll
      for (int rt = 0; rt < 4; ++rt) g[rt] = a.M;
      return false;
    }
    if (!a.live16) {
#pragma unroll
      for (int rt = 0; rt < 4; ++rt) g[rt] = cur * FT_M + 16 * rt;
    } else {
      lw.group(kcur, g, a.M);
    }
    cur += gridDim.x;
    ++kcur;
    return true;
  };
  auto prefetch_h = [&](const int (&g)[4], int ch) {    // rows >= M: clamped address, never stored
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int c8 = (tid & 15) * 8, m = min(g[i] + (tid >> 4), a.M - 1);
      load_frag(hpre[i], gofs(h1, (unsigned int)m * (unsigned int)a.dff + (unsigned int)(ch * FD + c8)));
    }
  };
  auto prefetch_rows = [&](const int (&g)[4]) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int c8 = (tid & 15) * 8, m = min(g[i] + (tid >> 4), a.M - 1);
      load_frag(cpre[i], gofs(dl2, (unsigned int)(m * FD + c8)));
      load_frag(xpre[i], gofs(dz, (unsigned int)(m * FD + c8)));
      if constexpr (LNF) {
        // unconditional loads (a load under a condition is waited for at the join, i.e. with everything issued before it)
        const float* __restrict__ rmp = a.ln_rowmask ? a.ln_rowmask : a.ln_rstd;
        const float rv = rmp[m];
        rs_pre[i] = a.ln_rstd[m];
        rm_pre[i] = a.ln_rowmask ? rv : 1.f;
      }
    }
  };
  auto h_to_lds = [&]() {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int c = tid + 256 * i, r = c >> 4, c8 = (c & 15) * 8;
      stage8(Ah + Tile<XT>::off(r, c8), hpre[i]);
    }
  };
  int mb[4], mbn[4];
  bool have = next_group(mb);
  if (have) {
    load_wset(wp, W2t, FD, n0, 0, li, lg, a.w_packed);
    prefetch_rows(mb);
    prefetch_h(mb, 0);
  }
  for (; have;) {
    const bool have_next = next_group(mbn);
    if constexpr (LNF) {
      // LayerNorm backward of this thread's 4 row pieces (rg_ln_bwd's arithmetic, statistic by statistic):
      //   g = dy * rowmask * gamma ; dz = rstd * (g - mean(g) - xhat * mean(g * xhat)), xhat = (y / rowmask - beta) / gamma
      const int c8 = (tid & 15) * 8;
      float gam[8], bet[8], igam[8];
      load8(gam, lnp + c8);
      load8(bet, lnp + FD + c8);
      load8(igam, lnp + 2 * FD + c8);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int r = 16 * i + (tid >> 4), m = mb[i] + (tid >> 4);
        const bool live = m < a.M;
        const float rm = live ? rm_pre[i] : 0.f;
        float g[8], xh[8], s1 = 0.f, s2 = 0.f;
        if (rm != 0.f) {
          const float irm = 1.f / rm;
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            const float d = (float)cpre[i].v[j] * rm;
            xh[j] = ((float)xpre[i].v[j] * irm - bet[j]) * igam[j];
            g[j] = d * gam[j];
            dg[j] += d * xh[j];
            db[j] += d;
            s1 += g[j];
            s2 += g[j] * xh[j];
          }
        } else {
#pragma unroll
          for (int j = 0; j < 8; ++j) { g[j] = 0.f; xh[j] = 0.f; }
        }
#pragma unroll
        for (int o = 1; o < 16; o <<= 1) { s1 += __shfl_xor(s1, o); s2 += __shfl_xor(s2, o); }
        const float rstd = rm != 0.f ? rs_pre[i] : 0.f;
        s1 *= 1.f / FD;
        s2 *= 1.f / FD;
        float o8[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) o8[j] = rstd * (g[j] - s1 - xh[j] * s2);
        store8(Adz + Tile<XT>::off(r, c8), o8);
        if (ldrop.thresh != 0u || ldrop.onebit) {
          float k8[8];
          rg_keep8(ldrop, (unsigned int)m * (unsigned int)FD + (unsigned int)c8, k8);
#pragma unroll
          for (int j = 0; j < 8; ++j) o8[j] *= k8[j];
        }
        store8(Adl + Tile<LT>::off(r, c8), o8);
        if (live) store8(gofs(dl2o, (unsigned int)(m * FD + c8)), o8);
      }
    } else {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int c = tid + 256 * i, r = c >> 4, c8 = (c & 15) * 8;
        stage8(Adl + Tile<LT>::off(r, c8), cpre[i]);
        stage8(Adz + Tile<XT>::off(r, c8), xpre[i]);
      }
    }
    h_to_lds();
    lds_barrier();
    f32x4 acc[2][4], acc2[2][4];
#pragma unroll
    for (int ct = 0; ct < 2; ++ct)
#pragma unroll
      for (int rt = 0; rt < 4; ++rt) acc2[ct][rt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
    for (int ch = 0; ch < nchunk; ++ch) {
      load_wset(wq, W1t, a.dff, n0, ch * FD, li, lg, a.w_packed);
      {   // h1 rows of the next chunk, or chunk 0 of the next tile: one unconditional load sequence from selected rows
          // (a branch around loads drains vmcnt at the join)
        const bool last = ch + 1 == nchunk;
        int gs[4];
#pragma unroll
        for (int rt = 0; rt < 4; ++rt) gs[rt] = last ? mbn[rt] : mb[rt];
        prefetch_h(gs, last ? 0 : ch + 1);
      }
#pragma unroll
      for (int ct = 0; ct < 2; ++ct)
#pragma unroll
        for (int rt = 0; rt < 4; ++rt) acc[ct][rt] = (f32x4){0.f, 0.f, 0.f, 0.f};
      mma_wset<T>(acc, wp, Adl, li, lg);                 // dg chunk = dl2 . W2t[chunk]^T
      load_wset(wp, W2t + (unsigned int)((ch + 1 < nchunk) ? ch + 1 : 0) * (FD * FD), FD, n0, 0, li, lg, a.w_packed);
      if (ch > 0) lds_barrier();                         // readers of the previous dh1 chunk are done, h1 chunk visible
#pragma unroll
      for (int ct = 0; ct < 2; ++ct)
#pragma unroll
        for (int rt = 0; rt < 4; ++rt) {
          float x4[4];
          load4t(x4, Ah + Tile<XT>::off(rt * 16 + li, n0 + ct * 16 + 4 * lg));
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            float v = acc[ct][rt][r] * gelu_grad_t<Precise<T>::value>(x4[r]);
            if (nz > 0.f) v = x4[r] != 0.f ? v * nz : 0.f;
            acc[ct][rt][r] = v;
          }
        }
      regs_to_tile<LT>(acc, Adh, n0, li, lg);
      lds_barrier();                                     // dh1 chunk complete, h1 chunk no longer read
      if (ch + 1 < nchunk) h_to_lds();
      tile_to_hbm<T>(Adh, dh1, a.dff, ch * FD, mb, a.M, tid);
      mma_wset<T>(acc2, wq, Adh, li, lg);                // dy += dh1 chunk . W1t[:, chunk]^T
    }
    prefetch_rows(mbn);
#pragma unroll
    for (int ct = 0; ct < 2; ++ct)
#pragma unroll
      for (int rt = 0; rt < 4; ++rt) {
        float r4[4];
        load4t(r4, Adz + Tile<XT>::off(rt * 16 + li, n0 + ct * 16 + 4 * lg));
#pragma unroll
        for (int r = 0; r < 4; ++r) acc2[ct][rt][r] += r4[r];
      }
    regs_to_tile<XT>(acc2, Ady, n0, li, lg);              // every wave is past its last read of the dl2 tile
    lds_barrier();
    tile_to_hbm<T>(Ady, dy, FD, 0, mb, a.M, tid);
    lds_barrier();
#pragma unroll
    for (int rt = 0; rt < 4; ++rt) mb[rt] = mbn[rt];
    have = have_next;
  }
  if (a.live16) {                                        // padded row tiles: dy rows = 0
    const int nrt = (a.M + 15) >> 4, ndead = nrt - a.live16[0];
    for (int j = 4 * (int)blockIdx.x; j < ndead; j += 4 * (int)gridDim.x) {
      int md[4];
#pragma unroll
      for (int rt = 0; rt < 4; ++rt) md[rt] = j + rt < ndead ? a.live16[nrt - (j + rt)] * 16 : a.M;
      zero_to_hbm<T>(dy, FD, 0, md, a.M, tid);
    }
  }
  if constexpr (LNF) {
    // dgamma / dbeta: lanes with the same feature octet -> waves -> this workgroup's slice of the partials
#pragma unroll
    for (int o = 16; o < 64; o <<= 1)
#pragma unroll
      for (int j = 0; j < 8; ++j) { dg[j] += __shfl_xor(dg[j], o); db[j] += __shfl_xor(db[j], o); }
    float* red = reinterpret_cast<float*>(smem);          // [2][4][128] over the (dead) first tile
    __syncthreads();
    if (lane < 16) {
#pragma unroll
      for (int j = 0; j < 8; ++j) { red[(0 * 4 + wave) * FD + lane * 8 + j] = dg[j]; red[(1 * 4 + wave) * FD + lane * 8 + j] = db[j]; }
    }
    __syncthreads();
    const int w = tid >> 7, n = tid & 127;                // 256 threads = 2 x 128
    a.ln_partials[((size_t)blockIdx.x * 2 + w) * FD + n] = (red[(w * 4 + 0) * FD + n] + red[(w * 4 + 1) * FD + n]) + (red[(w * 4 + 2) * FD + n] + red[(w * 4 + 3) * FD + n]);
  }
}

// dgamma / dbeta += column sums of the workgroups' partials [nblocks][2][128]
__global__ __launch_bounds__(256) void ffn_bwd_ln_reduce_kernel(const float* __restrict__ partials, int nblocks,
                                                                float* __restrict__ dgamma, float* __restrict__ dbeta) {
  const int c = threadIdx.x;                              // 0 .. 255: [2][128]
  float* dst = c < FD ? dgamma : dbeta;
  if (!dst) return;
  float s0 = 0.f, s1 = 0.f;
  int b = blockIdx.x;
  for (; b + (int)gridDim.x < nblocks; b += 2 * gridDim.x) {
    s0 += partials[(size_t)b * 2 * FD + c];
    s1 += partials[(size_t)(b + gridDim.x) * 2 * FD + c];
  }
  if (b < nblocks) s0 += partials[(size_t)b * 2 * FD + c];
  rg_acc(dst + (c & (FD - 1)), s0 + s1);
}

extern "C" int rg_ffn_bwd_data_supported(int d, int dff) { return d == FD && dff > 0 && (dff % FD) == 0; }

static int ffn_bwd_grid(int M, int dtype) {               // persistent workgroups: two per CU (bf16), one (f32)
  const int ntiles = (M + FT_M - 1) / FT_M;
  const int smem = 4 * act_tile_bytes(dtype, FT_M);
  const int per_cu = (160 * 1024) / smem;
  const int grid = 256 * (per_cu < 1 ? 1 : (per_cu > 2 ? 2 : per_cu));
  return grid > ntiles ? ntiles : grid;
}
extern "C" size_t rg_ffn_bwd_ln_workspace(int M) { return (size_t)(M <= 0 ? 0 : 512) * 2 * FD * sizeof(float); }

extern "C" int rg_ffn_bwd_data(const rg_ffn_bwd_args* a, int dtype, void* stream) {
  if (!a || a->M <= 0) return 0;
  if (!rg_ffn_bwd_data_supported(a->d, a->dff))
    return rg_set_error_msg(RG_ERR_UNSUPPORTED, "ffn_bwd_data: needs d_model == 128 and d_ff % 128 == 0");
  if ((long long)a->M * a->dff * (dtype == RG_BF16 ? 2 : 4) >= (1ll << 32))
    return rg_set_error_msg(RG_ERR_UNSUPPORTED, "ffn_bwd_data: M * d_ff * element size must be below 4 GiB (32-bit offsets)");
  const bool lnf = a->ln_dout != nullptr;
  if ((!lnf && (!a->dl2 || !a->dz)) || !a->h1 || !a->W2t || !a->W1t || !a->dh1 || !a->dy)
    return rg_set_error_msg(RG_ERR_INVALID, "ffn_bwd_data: NULL operand");
  if (lnf && (!a->ln_out || !a->ln_rstd || !a->ln_gamma || !a->ln_beta || !a->dl2_out || !a->ln_partials))
    return rg_set_error_msg(RG_ERR_INVALID, "ffn_bwd_data: the fused LayerNorm backward needs ln_out, ln_rstd, ln_gamma, ln_beta, dl2_out, ln_partials");
  hipStream_t s = (hipStream_t)stream;
  const int ntiles = (a->M + FT_M - 1) / FT_M;
  const int smem = 4 * act_tile_bytes(dtype, FT_M) + (lnf ? 3 * FD * 4 : 0);
  const int grid = ffn_bwd_grid(a->M, dtype);
#define RG_FB(T, L)                                                                                                       \
  do {                                                                                                                    \
    hipFuncSetAttribute(reinterpret_cast<const void*>(ffn_bwd_data_kernel<T, L>), hipFuncAttributeMaxDynamicSharedMemorySize, smem); \
    hipLaunchKernelGGL((ffn_bwd_data_kernel<T, L>), dim3(grid), dim3(256), smem, s, *a);                                  \
  } while (0)
  if (dtype == RG_BF16) { if (lnf) RG_FB(__bf16, true); else RG_FB(__bf16, false); }
  else if (dtype == RG_F32) { if (lnf) RG_FB(float, true); else RG_FB(float, false); }
  else if (dtype == RG_X3) { if (lnf) RG_FB(x3, true); else RG_FB(x3, false); }
  else return rg_set_error_msg(RG_ERR_INVALID, "ffn_bwd_data: bad dtype");
#undef RG_FB
  if (lnf && (a->ln_dgamma || a->ln_dbeta))
    hipLaunchKernelGGL(ffn_bwd_ln_reduce_kernel, dim3(grid < 32 ? grid : 32), dim3(256), 0, s, a->ln_partials, grid, a->ln_dgamma, a->ln_dbeta);
  RG_CHECK_LAUNCH();
  return 0;
}

// ---------------------------------------------------------------------------------------------------------------------
// Attention block tail, backward (rg_attn_out_bwd): LayerNorm-1 backward on the staged rows (same staging layout and
// arithmetic as rg_ln_bwd / the FFN kernel above) -> dz to LDS (A operand) and HBM -> dctx = dz . Wo with Wo^T stationary
// in registers (one weight set, loaded once per workgroup).  HBM-bound: 2 tiles of LDS, ~100 VGPRs, 4 workgroups per CU.
template <typename T>
__global__ __launch_bounds__(256, 2) void attn_out_bwd_kernel(rg_attn_out_bwd_args a) {
  constexpr int PL = FT_M * FD * 2;
  typedef typename LdsT<T, PL>::type LT;
  constexpr int ACT_BYTES = FT_M * Tile<LT>::LD * (int)sizeof(LT) * LdsT<T, PL>::PLANES;
  extern __shared__ __align__(16) unsigned char smem[];
  LT* Adz = reinterpret_cast<LT*>(smem);
  typedef typename ResT<T, PL>::type XT;                 // dctx on its way out: not a matrix operand (raw f32 in the bf16x3 tier)
  XT* Aout = reinterpret_cast<XT*>(smem + ACT_BYTES);
  float* lnp = reinterpret_cast<float*>(smem + 2 * ACT_BYTES);      // gamma | beta | 1 / gamma
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 15, lg = lane >> 4;
  const T* __restrict__ dy = reinterpret_cast<const T*>(a.dy);
  const T* __restrict__ y = reinterpret_cast<const T*>(a.y);
  T* __restrict__ dz = reinterpret_cast<T*>(a.dz);
  T* __restrict__ dctx = reinterpret_cast<T*>(a.dctx);
  const int n0 = wave * 32;
  const int ntiles = (a.M + FT_M - 1) / FT_M;
  for (int i = tid; i < FD; i += 256) { lnp[i] = a.gamma[i]; lnp[FD + i] = a.beta[i]; lnp[2 * FD + i] = 1.f / a.gamma[i]; }
  WSet<T> w;
  load_wset(w, reinterpret_cast<const T*>(a.Wot), FD, n0, 0, li, lg, a.w_packed);
  Frag<T> cpre[4], xpre[4];
  float rs_pre[4], rm_pre[4];
  float dg[8], db[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) { dg[j] = 0.f; db[j] = 0.f; }
  LiveWalk lw;
  lw.init(a.live16, a.M);
  const int nwork = a.live16 ? (lw.nlive + 3) >> 2 : ntiles;
  int cur = (int)blockIdx.x, kcur = 0;
  auto next_group = [&](int (&g)[4]) -> bool {
    if (cur >= nwork) {
#pragma unroll
      for (int rt = 0; rt < 4; ++rt) g[rt] = a.M;
      return false;
    }
    if (!a.live16) {
#pragma unroll
      for (int rt = 0; rt < 4; ++rt) g[rt] = cur * FT_M + 16 * rt;
    } else {
      lw.group(kcur, g, a.M);
    }
    cur += gridDim.x;
    ++kcur;
    return true;
  };
  auto prefetch_rows = [&](const int (&g)[4]) {         // unconditional loads from clamped rows
    const float* __restrict__ rmp = a.rowmask ? a.rowmask : a.rstd;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int c8 = (tid & 15) * 8, m = min(g[i] + (tid >> 4), a.M - 1);
      load_frag(cpre[i], gofs(dy, (unsigned int)(m * FD + c8)));
      load_frag(xpre[i], gofs(y, (unsigned int)(m * FD + c8)));
      const float rv = rmp[m];
      rs_pre[i] = a.rstd[m];
      rm_pre[i] = a.rowmask ? rv : 1.f;
    }
  };
  int mb[4], mbn[4];
  bool have = next_group(mb);
  if (have) prefetch_rows(mb);
  __syncthreads();                                      // lnp
  for (; have;) {
    const bool have_next = next_group(mbn);
    {
      const int c8 = (tid & 15) * 8;
      float gam[8], bet[8], igam[8];
      load8(gam, lnp + c8);
      load8(bet, lnp + FD + c8);
      load8(igam, lnp + 2 * FD + c8);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int r = 16 * i + (tid >> 4), m = mb[i] + (tid >> 4);
        const bool live = m < a.M;
        const float rm = live ? rm_pre[i] : 0.f;
        float g[8], xh[8], s1 = 0.f, s2 = 0.f;
        if (rm != 0.f) {
          const float irm = 1.f / rm;
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            const float d = (float)cpre[i].v[j] * rm;
            xh[j] = ((float)xpre[i].v[j] * irm - bet[j]) * igam[j];
            g[j] = d * gam[j];
            dg[j] += d * xh[j];
            db[j] += d;
            s1 += g[j];
            s2 += g[j] * xh[j];
          }
        } else {
#pragma unroll
          for (int j = 0; j < 8; ++j) { g[j] = 0.f; xh[j] = 0.f; }
        }
#pragma unroll
        for (int o = 1; o < 16; o <<= 1) { s1 += __shfl_xor(s1, o); s2 += __shfl_xor(s2, o); }
        const float rstd = rm != 0.f ? rs_pre[i] : 0.f;
        s1 *= 1.f / FD;
        s2 *= 1.f / FD;
        float o8[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) o8[j] = rstd * (g[j] - s1 - xh[j] * s2);
        store8(Adz + Tile<LT>::off(r, c8), o8);
        if (live) store8(gofs(dz, (unsigned int)(m * FD + c8)), o8);
      }
    }
    prefetch_rows(mbn);                                 // next tile's rows: in flight under the product
    lds_barrier();
    f32x4 acc[2][4];
#pragma unroll
    for (int ct = 0; ct < 2; ++ct)
#pragma unroll
      for (int rt = 0; rt < 4; ++rt) acc[ct][rt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    mma_wset<T>(acc, w, Adz, li, lg);                   // dctx = dz . Wot^T
    regs_to_tile<XT>(acc, Aout, n0, li, lg);
    lds_barrier();
    tile_to_hbm<T>(Aout, dctx, FD, 0, mb, a.M, tid);
#pragma unroll
    for (int rt = 0; rt < 4; ++rt) mb[rt] = mbn[rt];
    have = have_next;
    // (the next iteration's stores to Adz come after this iteration's MFMA reads of it: every wave passed the second
    // barrier after its reads; Aout is rewritten only after the next iteration's first barrier)
  }
  // dgamma / dbeta: lanes with the same feature octet -> waves -> this workgroup's slice of the partials
#pragma unroll
  for (int o = 16; o < 64; o <<= 1)
#pragma unroll
    for (int j = 0; j < 8; ++j) { dg[j] += __shfl_xor(dg[j], o); db[j] += __shfl_xor(db[j], o); }
  float* red = reinterpret_cast<float*>(smem);
  __syncthreads();
  if (lane < 16) {
#pragma unroll
    for (int j = 0; j < 8; ++j) { red[(0 * 4 + wave) * FD + lane * 8 + j] = dg[j]; red[(1 * 4 + wave) * FD + lane * 8 + j] = db[j]; }
  }
  __syncthreads();
  const int wsel = tid >> 7, n = tid & 127;
  a.ln_partials[((size_t)blockIdx.x * 2 + wsel) * FD + n] = (red[(wsel * 4 + 0) * FD + n] + red[(wsel * 4 + 1) * FD + n]) + (red[(wsel * 4 + 2) * FD + n] + red[(wsel * 4 + 3) * FD + n]);
}

static int attn_out_bwd_grid(int M) {
  const int ntiles = (M + FT_M - 1) / FT_M;
  return ntiles < 512 ? ntiles : 512;                    // two persistent workgroups per CU (the LayerNorm state + weight set + prefetch need ~190 VGPRs)
}
extern "C" size_t rg_attn_out_bwd_workspace(int M) { return (size_t)(M <= 0 ? 0 : 1024) * 2 * FD * sizeof(float); }   // (grid <= 1024)

extern "C" int rg_attn_out_bwd(const rg_attn_out_bwd_args* a, int dtype, void* stream) {
  if (!a || a->M <= 0) return 0;
  if (a->d != FD || a->P != FD) return rg_set_error_msg(RG_ERR_UNSUPPORTED, "attn_out_bwd: needs d_model == n_heads*32 == 128");
  if (!a->dy || !a->y || !a->rstd || !a->gamma || !a->beta || !a->Wot || !a->dz || !a->dctx || !a->ln_partials)
    return rg_set_error_msg(RG_ERR_INVALID, "attn_out_bwd: NULL operand");
  if ((long long)a->M * FD * 4 >= (1ll << 32)) return rg_set_error_msg(RG_ERR_UNSUPPORTED, "attn_out_bwd: M too large for 32-bit offsets");
  hipStream_t s = (hipStream_t)stream;
  const int smem = 2 * act_tile_bytes(dtype, FT_M) + 3 * FD * 4;
  const int grid = attn_out_bwd_grid(a->M);
  if (dtype == RG_BF16) {
    hipFuncSetAttribute(reinterpret_cast<const void*>(attn_out_bwd_kernel<__bf16>), hipFuncAttributeMaxDynamicSharedMemorySize, smem);
    hipLaunchKernelGGL((attn_out_bwd_kernel<__bf16>), dim3(grid), dim3(256), smem, s, *a);
  } else if (dtype == RG_F32) {
    hipFuncSetAttribute(reinterpret_cast<const void*>(attn_out_bwd_kernel<float>), hipFuncAttributeMaxDynamicSharedMemorySize, smem);
    hipLaunchKernelGGL((attn_out_bwd_kernel<float>), dim3(grid), dim3(256), smem, s, *a);
  } else if (dtype == RG_X3) {
    hipFuncSetAttribute(reinterpret_cast<const void*>(attn_out_bwd_kernel<x3>), hipFuncAttributeMaxDynamicSharedMemorySize, smem);
    hipLaunchKernelGGL((attn_out_bwd_kernel<x3>), dim3(grid), dim3(256), smem, s, *a);
  } else return rg_set_error_msg(RG_ERR_INVALID, "attn_out_bwd: bad dtype");
  if (a->dgamma || a->dbeta)
    hipLaunchKernelGGL(ffn_bwd_ln_reduce_kernel, dim3(grid < 32 ? grid : 32), dim3(256), 0, s, a->ln_partials, grid, a->dgamma, a->dbeta);
  RG_CHECK_LAUNCH();
  return 0;
}
