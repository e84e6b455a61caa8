// Fused discriminator MLP + W-GAN gradient penalty (north-star component).
//
// Replaces, per critic update, netD(real), netD(fake), dis_loss.backward(), calc_gradient_penalty(...).backward()
// (reference GURU/tools/utils.py:41-57, GURU/gan_training.py:38-55,430-448) -- about a hundred small GEMM / elementwise
// launches over [B, <=1280] matrices -- by ONE row-parallel launch plus three weight-gradient GEMMs.
//
// Everything of the discriminator that depends on a single row stays inside one workgroup: a tile of TR rows walks the
// whole chain with its activations in LDS,
//
//   W rows  (stacked [real; fake], row r has d loss / d D(x_r) = coef_r):
//     x -> h1 -> h2 -> h3 -> out            Linear + ReLU + Dropout(0.2) x3, Linear        (tools/utils.py:41-57)
//     e3 = coef * w4 * m3 ;  e2 = (e3 W3) * m2 ;  e1 = (e2 W2) * m1 ;  [dx = e1 W1]        (backward of the same)
//   GP rows (xhat = alpha real + (1 - alpha) fake, gan_training.py:39-43):
//     xhat -> h1 -> h2 -> h3                 (masks m_i = [h_i > 0] / (1 - p) of the dropped activations)
//     u3 = w4 * m3 ;  u2 = (u3 W3) * m2 ;  u1 = (u2 W2) * m1 ;  g = u1 W1 = dD/dxhat      (closed form)
//     gp += lambda/B (|g| - 1)^2 ;  dg = lambda 2/B (|g| - 1)/|g| g
//     e1 = (dg W1^T) * m1 ;  e2 = (e1 W2^T) * m2 ;  e3 = (e2 W3^T) * m3                    (second-order chain)
//
// and leaves in HBM only what the weight gradients contract over ALL rows, as row-stacked operand pairs
//     dW1 += Y1^T X1,  dW2 += Y2^T X2,  dW3 += Y3^T X3        rows [0, 2B): (e_i, h_{i-1}),  rows [2B, 3B): (u_i, e_{i-1} | dg)
// (three rg_gemm_tn launches, which also produce db_i = column sums of the e_i rows: colsum_T = 2B); dw4, db4, the two
// output means and the penalty are accumulated by the row kernel itself (f32 atomics, n3 + 4 addresses).
//
// Tile vocabulary as everywhere (rg_common.hip.h): weight fragment = A operand (16 output features x 32 k), activation
// fragment = B operand (16 rows x 32 k from LDS), so an accumulator register holds 4 consecutive features of one row.
// The weights come in FRAGMENT-PACKED copies (rg_cast, RG_CAST_PACK): a tile of 32 rows gives each weight fragment only
// 2 x 2 MFMAs, so the fragment loads themselves are on the critical path -- from the row-major [out][in] layout a wave's
// load touches 16 rows x 64 B and the vector L1 spends one tag lookup per row and 16-lane pass (measured: 260 us per
// launch with the MFMAs and every other memory access ablated away); packed, it is one contiguous 1 KB read.  8 waves per workgroup; a wave takes PAIRS of
// 16-feature blocks so that every activation fragment read from LDS feeds two MFMAs per row block.
#include "rg_common.hip.h"
#include "rg_det.hip.h"
#include "../../include/recguru_hip.h"

#define DNW 8                // waves per workgroup
#define DTHREADS (64 * DNW)
#define DPAD 8               // LDS row padding (elements)
#define DKU 4                // k-steps per pipeline item

template <typename T> struct DiscCfg { static constexpr int NTB = sizeof(T) == 2 ? 2 : 1; };   // 32 rows (bf16) / 16 rows (f32)

// No epilogue touches global memory: vmcnt counts in order, so a wait for one small epilogue load (a bias vector, a mask)
// would be a wait for every weight fragment prefetched after it -- the pipeline drained at the end of every feature-block
// pair (measured: 2 500 cycles per pipeline item instead of ~300).  Biases and w4 are staged in LDS once per tile; the
// ReLU-and-kept masks m_i = [h_i > 0] live in LDS as BIT masks (one 32-bit word per row and feature-block pair), written
// by the forward epilogues and read by every later chain.
struct DiscEpi {
  int ablate;                // rg_disc_args.debug_ablate
  int kind;                  // 0: relu(+dropout) with bias, records the mask ; 1: mask multiply ; 2: plain f32 ; 3: plain T
  const float* bias;         // kind 0: LDS
  DropCfg dc; unsigned int drop_n; long long row0;     // kind 0: hash index = (row0 + t) * drop_n + f
  unsigned int* bits; int wpr;                         // kinds 0 / 1: LDS bit mask [TR][wpr], word fp = features 32 fp .. + 32
  float sc;                  // kind 1: factor of the kept elements (1 / (1 - p) under dropout)
};

// ---- one pipeline item = DKU k-steps of a pair of 16-feature blocks ----------------------------------------------------
template <typename T>
struct DiscItem { Frag<T> a[2][DKU]; };

template <typename T>
__device__ __forceinline__ void disc_issue(DiscItem<T>& b, const T* __restrict__ wl, int it, int total, int nch, int nks, int nfb,
                                           int wave, int ablate) {
  it = min(it, total - 1);                                     // unconditional, clamped (no vmcnt(0) at a join)
  const int pi = it / nch, c = it - pi * nch;
  const int fp = wave + pi * DNW;
  const int fb0 = 2 * fp, fb1 = min(2 * fp + 1, nfb - 1);
#pragma unroll
  for (int u = 0; u < DKU; ++u) {
    const int ks = (ablate & 2) ? 0 : min(c * DKU + u, nks - 1);
    load_frag(b.a[0][u], wl + ((size_t)fb0 * nks + ks) * 512);
    load_frag(b.a[1][u], wl + ((size_t)fb1 * nks + ks) * 512);
  }
}

template <typename T, int NTB>
__device__ __forceinline__ void disc_epilogue(const f32x4 (&acc)[2][NTB], int fp, int fb0, int fb1, bool two, T* out, int ldout,
                                              float* outf, int ldoutf, const DiscEpi& ep, int nvalid, int li, int lg) {
  // lane (lg, li) holds features f0 + 4 lg + r (r < 4) of row tb * 16 + li; the pair's 32 features are one mask word
#pragma unroll
  for (int tb = 0; tb < NTB; ++tb) {
    const int t = tb * 16 + li;
    unsigned int word = 0u;
    if (ep.kind == 1) word = ep.bits[t * ep.wpr + fp];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      if (q == 0 || two) {
        const int f = (q ? fb1 : fb0) * 16 + 4 * lg;
        float v[4] = {acc[q][tb][0], acc[q][tb][1], acc[q][tb][2], acc[q][tb][3]};
        if (ep.kind == 0) {
          const float4 b = *reinterpret_cast<const float4*>(ep.bias + f);
          v[0] = fmaxf(v[0] + b.x, 0.f); v[1] = fmaxf(v[1] + b.y, 0.f); v[2] = fmaxf(v[2] + b.z, 0.f); v[3] = fmaxf(v[3] + b.w, 0.f);
          if (ep.dc.thresh) {
            float k4[4];
            rg_keep4(ep.dc, (unsigned int)(ep.row0 + t) * ep.drop_n + (unsigned int)f, k4);
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] *= k4[r];
          }
          if (t >= nvalid) { v[0] = v[1] = v[2] = v[3] = 0.f; }
          const unsigned int nib = (v[0] > 0.f ? 1u : 0u) | (v[1] > 0.f ? 2u : 0u) | (v[2] > 0.f ? 4u : 0u) | (v[3] > 0.f ? 8u : 0u);
          word |= nib << (16 * q + 4 * lg);
          store4(out + t * ldout + f, v);
        } else if (ep.kind == 1) {
          const unsigned int nib = word >> (16 * q + 4 * lg);
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] = ((nib >> r) & 1u) ? v[r] * ep.sc : 0.f;
          store4(out + t * ldout + f, v);
        } else if (ep.kind == 2) {
          store4(outf + t * ldoutf + f, v);
        } else {
          store4(out + t * ldout + f, v);
        }
      }
    }
    if (ep.kind == 0) {                       // the four lanes of a row (lg = 0..3) hold 8 bits each of the pair's word
      word |= (unsigned int)__shfl_xor((int)word, 16);
      word |= (unsigned int)__shfl_xor((int)word, 32);
      if (lg == 0) ep.bits[t * ep.wpr + fp] = word;
    }
  }
}

template <typename T, int NTB>
__device__ __forceinline__ void disc_consume(const DiscItem<T>& b, f32x4 (&acc)[2][NTB], int it, int nch, int nks, int nfb, int wave,
                                             const T* in, int ldin, T* out, int ldout, float* outf, int ldoutf, const DiscEpi& ep,
                                             int nvalid, int li, int lg) {
  const int pi = it / nch, c = it - pi * nch;
  const int fp = wave + pi * DNW;
  if (c == 0) {
#pragma unroll
    for (int q = 0; q < 2; ++q)
#pragma unroll
      for (int tb = 0; tb < NTB; ++tb) acc[q][tb] = (f32x4){0.f, 0.f, 0.f, 0.f};
  }
#pragma unroll
  for (int u = 0; u < DKU; ++u) {
    const int ks = c * DKU + u;
    if (ks < nks && !(ep.ablate & 4)) {
#pragma unroll
      for (int tb = 0; tb < NTB; ++tb) {
        Frag<T> bf;
        load_frag(bf, in + (tb * 16 + li) * ldin + ks * 32 + 8 * lg);
        mma(b.a[0][u], bf, acc[0][tb]);
        mma(b.a[1][u], bf, acc[1][tb]);
      }
    }
  }
  if (c == nch - 1)
    disc_epilogue<T, NTB>(acc, fp, 2 * fp, min(2 * fp + 1, nfb - 1), 2 * fp + 1 < nfb, out, ldout, outf, ldoutf, ep, nvalid, li, lg);
}

// out[t][f] for the TR rows of the tile: acc = sum_k W[f][k] * in[t][k].
// A wave walks its feature-block pairs fp = wave, wave + DNW, ... and, inside a pair, the K axis in chunks of DKU k-steps.
// The (pair, chunk) items form ONE software pipeline with three register buffers: the loads of item it + 3 are issued
// while item it is in its MFMA phase, across pair boundaries too -- with a pipeline restarted per pair every pair paid a
// full L2 / Infinity-Cache round trip (measured: the kernel took 110 us per tile with every MFMA and store ablated away).
template <typename T, int NTB>
__device__ __forceinline__ void disc_stage(const T* __restrict__ W, int N, int K, const T* in, int ldin, T* out, int ldout,
                                           float* outf, int ldoutf, const DiscEpi& ep, int nvalid) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, li = lane & 15, lg = lane >> 4;
  const int nfb = N >> 4, nks = K >> 5;
  const int nch = (nks + DKU - 1) / DKU;
  const int npair = (nfb + 1) >> 1;
  const int mypairs = npair > wave ? (npair - wave + DNW - 1) / DNW : 0;
  const int total = mypairs * nch;
  if (total == 0) return;
  const T* wl = W + (size_t)lane * 8;
  DiscItem<T> b0, b1, b2;
  f32x4 acc[2][NTB];
  disc_issue<T>(b0, wl, 0, total, nch, nks, nfb, wave, ep.ablate);
  disc_issue<T>(b1, wl, 1, total, nch, nks, nfb, wave, ep.ablate);
  disc_issue<T>(b2, wl, 2, total, nch, nks, nfb, wave, ep.ablate);
  for (int it = 0; it < total; it += 3) {
    disc_consume<T, NTB>(b0, acc, it, nch, nks, nfb, wave, in, ldin, out, ldout, outf, ldoutf, ep, nvalid, li, lg);
    disc_issue<T>(b0, wl, it + 3, total, nch, nks, nfb, wave, ep.ablate);
    if (it + 1 < total) disc_consume<T, NTB>(b1, acc, it + 1, nch, nks, nfb, wave, in, ldin, out, ldout, outf, ldoutf, ep, nvalid, li, lg);
    disc_issue<T>(b1, wl, it + 4, total, nch, nks, nfb, wave, ep.ablate);
    if (it + 2 < total) disc_consume<T, NTB>(b2, acc, it + 2, nch, nks, nfb, wave, in, ldin, out, ldout, outf, ldoutf, ep, nvalid, li, lg);
    disc_issue<T>(b2, wl, it + 5, total, nch, nks, nfb, wave, ep.ablate);
  }
}

// LDS tile [TR][ld] -> global rows (16-byte vectors, consecutive threads on consecutive bytes); rows >= nvalid skipped
template <typename T>
__device__ __forceinline__ void disc_flush(const T* tile, int ld, T* dst, int lddst, int N, int TR, int nvalid) {
  if (!dst) return;
  constexpr int V = 16 / (int)sizeof(T);
  const int cpr = N / V;
  for (int c = threadIdx.x; c < TR * cpr; c += DTHREADS) {
    const int t = c / cpr, o = (c - t * cpr) * V;
    if (t < nvalid) *reinterpret_cast<float4*>(dst + (size_t)t * lddst + o) = *reinterpret_cast<const float4*>(tile + t * ld + o);
  }
}

#define DSTAMP(i) do { if (a.stamps && threadIdx.x == 0) a.stamps[(size_t)blockIdx.x * 16 + (i)] = (long long)__builtin_amdgcn_s_memtime(); } while (0)

template <typename T>
__global__ __launch_bounds__(DTHREADS, 1) void disc_rows_kernel(rg_disc_args a) {
  constexpr int NTB = DiscCfg<T>::NTB;
  constexpr int TR = 16 * NTB;
  extern __shared__ __align__(16) unsigned char dsm[];
  const int d = a.d, n1 = a.n1, n2 = a.n2, n3 = a.n3;
  const int ldA = n2 + DPAD, ldB = (n1 > n3 ? n1 : n3) + DPAD;
  const int w1 = (n1 + 31) >> 5, w2 = (n2 + 31) >> 5, w3 = (n3 + 31) >> 5;      // mask words per row
  T* bufA = reinterpret_cast<T*>(dsm);                       // [TR][ldA]: x, h2 / e2 / u2, g + dg, e2'
  T* bufB = bufA + TR * ldA;                                 // [TR][ldB]: h1, h3 / e3 / u3, e1 / u1, e1', e3'
  float* cst = reinterpret_cast<float*>(bufB + TR * ldB);    // b1 | b2 | b3 | w4
  float* rowf = cst + n1 + n2 + n3 + n3;                     // [TR] per-row coefficient / norm factor
  float* red = rowf + TR;                                    // [DNW] block reduction scratch
  unsigned int* bits1 = reinterpret_cast<unsigned int*>(red + DNW);   // [TR][w1]
  unsigned int* bits2 = bits1 + TR * w1;                              // [TR][w2]
  unsigned int* bits3 = bits2 + TR * w2;                              // [TR][w3]
  const float* sb1 = cst; const float* sb2 = cst + n1; const float* sb3 = cst + n1 + n2; const float* sw4 = cst + n1 + n2 + n3;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // the gradient-penalty tiles walk nine stages, the W tiles five: dispatched FIRST they overlap the second round of W
  // tiles instead of running alone on half of the CUs at the end of the launch
  const int n_gt = a.alpha ? (a.B + TR - 1) / TR : 0;
  const bool gp_tile = (int)blockIdx.x < n_gt;
  const int tile = gp_tile ? (int)blockIdx.x : (int)blockIdx.x - n_gt;
  const int nrows = gp_tile ? a.B : 2 * a.B;
  const int r0 = tile * TR;
  const int nvalid = min(TR, nrows - r0);
  const long long srow = (gp_tile ? 2LL * a.B : 0LL) + r0;   // first row of this tile in the stacked [3B, .] operands
  const T* W1 = reinterpret_cast<const T*>(a.W1);
  const T* W2 = reinterpret_cast<const T*>(a.W2);
  const T* W3 = reinterpret_cast<const T*>(a.W3);
  const T* W1t = reinterpret_cast<const T*>(a.W1t);
  const T* W2t = reinterpret_cast<const T*>(a.W2t);
  const T* W3t = reinterpret_cast<const T*>(a.W3t);
  T* Y1 = reinterpret_cast<T*>(a.Y1); T* X1 = reinterpret_cast<T*>(a.X1);
  T* Y2 = reinterpret_cast<T*>(a.Y2); T* X2 = reinterpret_cast<T*>(a.X2);
  T* Y3 = reinterpret_cast<T*>(a.Y3); T* X3 = reinterpret_cast<T*>(a.X3);
  const float sc = a.drop_p > 0.f ? 1.f / (1.f - a.drop_p) : 1.f;
  auto at = [](T* p, long long row, int ld) -> T* { return p ? p + (size_t)row * ld : nullptr; };
  const int nvf = (a.debug_ablate & 8) ? 0 : nvalid;         // rows the flushes write
  const bool wg = a.need_wgrad && !(a.debug_ablate & 1);     // dw4 / db4 atomics wanted

  DSTAMP(0);
  // ---- stage 0: constants -> LDS; the tile's input rows -> bufA[:, :d]  (+ X1 rows of the W tiles)
  for (int c = tid; c < n1; c += DTHREADS) cst[c] = a.b1[c];
  for (int c = tid; c < n2; c += DTHREADS) cst[n1 + c] = a.b2[c];
  for (int c = tid; c < n3; c += DTHREADS) { cst[n1 + n2 + c] = a.b3[c]; cst[n1 + n2 + n3 + c] = a.w4[c]; }
  {
    const T* real = reinterpret_cast<const T*>(a.real);
    const T* fake = reinterpret_cast<const T*>(a.fake);
    for (int c = tid; c < TR * d; c += DTHREADS) {
      const int t = c / d, k = c - t * d;
      float x = 0.f;
      if (t < nvalid) {
        const int r = r0 + t;
        if (gp_tile) {
          const float al = a.alpha[r];
          x = al * (float)real[(size_t)r * d + k] + (1.f - al) * (float)fake[(size_t)r * d + k];
        } else {
          x = r < a.B ? (float)real[(size_t)r * d + k] : (float)fake[(size_t)(r - a.B) * d + k];
        }
      }
      bufA[t * ldA + k] = (T)x;
    }
    if (tid < TR) {
      const int r = r0 + tid;
      rowf[tid] = (tid < nvalid && !gp_tile) ? (r < a.B ? a.coef_real : a.coef_fake) : 0.f;
    }
  }
  __syncthreads();
  if (!gp_tile) disc_flush(bufA, ldA, at(X1, srow, d), d, d, TR, nvf);
  if (a.debug_ablate & 32) return;                            // profiling: input staging only

  DiscEpi ep;
  ep.ablate = a.debug_ablate;
  ep.sc = sc; ep.bias = nullptr; ep.drop_n = 0; ep.row0 = r0; ep.bits = nullptr; ep.wpr = 0; ep.kind = 0;
  const unsigned long long seed0 = gp_tile ? a.seed_g[0] : a.seed_w[0], seed1 = gp_tile ? a.seed_g[1] : a.seed_w[1],
                           seed2 = gp_tile ? a.seed_g[2] : a.seed_w[2];
  float* G = reinterpret_cast<float*>(bufA);                  // GP rows, stage 5: g as [TR][d + 4] f32 ...
  const int ldg = d + 4;
  T* dgt = bufA + (size_t)TR * ldg * (sizeof(float) / sizeof(T));   // ... and dg behind it: [TR][d + DPAD]
  const int lddg = d + DPAD;
  // The chain is a TABLE of GEMM stages walked by one loop, so that the pipelined stage body exists once in the code
  // (inlined at eleven call sites it was 52 KB of ISA -- and this compiler's inliner crashes on it):
  //   stage  W tile                      GP tile
  //   0      h1 = relu(x W1^T + b1)      same                         A -> B
  //   1      h2                          same                         B -> A
  //   2      h3                          same                         A -> B   then out / e3 | u3 in place
  //   3      e2 = (e3 W3) * m2           u2                           B -> A
  //   4      e1 = (e2 W2) * m1           u1                           A -> B
  //   5      dx = e1 W1 (optional)       g = u1 W1 (f32), norms, dg   B -> A
  //   6..8   --                          e1' = (dg W1^T) m1, e2' = (e1' W2^T) m2, e3' = (e2' W3^T) m3
  const int nst = gp_tile ? 9 : (a.dx ? 6 : 5);
  for (int st = 0; st < nst; ++st) {
    const T* Wm = W1; int N = n1, K = d, ldin = ldA, ldout = ldB;
    const T* in = bufA; T* out = bufB;
    switch (st) {
      case 0: ep.kind = 0; ep.bias = sb1; ep.dc = make_drop(a.drop_p, seed0); ep.drop_n = (unsigned int)n1; ep.bits = bits1; ep.wpr = w1; break;
      case 1: Wm = W2; N = n2; K = n1; in = bufB; ldin = ldB; out = bufA; ldout = ldA;
              ep.bias = sb2; ep.dc = make_drop(a.drop_p, seed1); ep.drop_n = (unsigned int)n2; ep.bits = bits2; ep.wpr = w2; break;
      case 2: Wm = W3; N = n3; K = n2;
              ep.bias = sb3; ep.dc = make_drop(a.drop_p, seed2); ep.drop_n = (unsigned int)n3; ep.bits = bits3; ep.wpr = w3; break;
      case 3: Wm = W3t; N = n2; K = n3; in = bufB; ldin = ldB; out = bufA; ldout = ldA; ep.kind = 1; ep.bits = bits2; ep.wpr = w2; break;
      case 4: Wm = W2t; N = n1; K = n2; ep.kind = 1; ep.bits = bits1; ep.wpr = w1; break;
      case 5: Wm = W1t; N = d; K = n1; in = bufB; ldin = ldB; out = bufA; ldout = ldA; ep.kind = gp_tile ? 2 : 3; break;
      case 6: Wm = W1; N = n1; K = d; in = dgt; ldin = lddg; ep.kind = 1; ep.bits = bits1; ep.wpr = w1; break;
      case 7: Wm = W2; N = n2; K = n1; in = bufB; ldin = ldB; out = bufA; ldout = ldA; ep.kind = 1; ep.bits = bits2; ep.wpr = w2; break;
      default: Wm = W3; N = n3; K = n2; ep.kind = 1; ep.bits = bits3; ep.wpr = w3; break;
    }
    DSTAMP(1 + st);
    disc_stage<T, NTB>(Wm, N, K, in, ldin, out, ldout, G, ldg, ep, nvalid);
    __syncthreads();
    if (st == 0 && !gp_tile) disc_flush(bufB, ldB, at(X2, srow, n1), n1, n1, TR, nvf);     // h1, h2: X operands of dW2, dW3
    if (st == 1 && !gp_tile) disc_flush(bufA, ldA, at(X3, srow, n2), n2, n2, TR, nvf);
    if (st == 2) {
      if (a.debug_ablate & 16) return;                        // profiling: forward only
      // out_t = h3 . w4 + b4 and the two means (W tiles); e3 / u3 = coef * sc * w4 * [h3 > 0] in place over h3;
      // dw4 += sum_t coef_t h3[t], db4 += sum_t coef_t (W tiles).  Thread tid owns feature columns tid, tid + 512, ...
      if (!gp_tile) {
        float sr = 0.f, sf = 0.f;
        for (int t = wave; t < nvalid; t += DNW) {
          float s = 0.f;
          for (int f = lane; f < n3; f += 64) s += (float)bufB[t * ldB + f] * sw4[f];
          s = wave_sum(s) + a.b4[0];
          const int r = r0 + t;
          if (lane == 0) {
            if (a.out) a.out[r] = s;
            if (r < a.B) sr += s; else sf += s;
          }
        }
        if (lane == 0 && a.scalars) {
          if (sr != 0.f) rg_acc(a.scalars + 0, sr / (float)a.B);
          if (sf != 0.f) rg_acc(a.scalars + 1, sf / (float)a.B);
        }
        if (wg && tid == 0 && a.db4) {
          float s = 0.f;
          for (int t = 0; t < nvalid; ++t) s += rowf[t];
          if (s != 0.f) rg_acc(a.db4, s);
        }
      }
      __syncthreads();
      for (int f = tid; f < n3; f += DTHREADS) {
        const float wv = sw4[f] * sc;
        float dsum = 0.f;
        for (int t = 0; t < TR; ++t) {
          const float h = (float)bufB[t * ldB + f];
          const float cf = gp_tile ? (t < nvalid ? 1.f : 0.f) : rowf[t];
          dsum += cf * h;
          bufB[t * ldB + f] = (T)(h > 0.f ? cf * wv : 0.f);
        }
        if (!gp_tile && wg && a.dw4 && dsum != 0.f) rg_acc(a.dw4 + f, dsum);
      }
      __syncthreads();
      disc_flush(bufB, ldB, at(Y3, srow, n3), n3, n3, TR, nvf);
    }
    if (st == 3) disc_flush(bufA, ldA, at(Y2, srow, n2), n2, n2, TR, nvf);
    if (st == 4) disc_flush(bufB, ldB, at(Y1, srow, n1), n1, n1, TR, nvf);
    if (st == 5 && !gp_tile) disc_flush(bufA, ldA, at(reinterpret_cast<T*>(a.dx), r0, d), d, d, TR, nvf);
    if (st == 5 && gp_tile) {                                 // norms, penalty, dg = lambda 2/B (|g| - 1)/|g| g
      float pen = 0.f;
      for (int t = wave; t < TR; t += DNW) {
        float s = 0.f;
        for (int k = lane; k < d; k += 64) { const float v = G[t * ldg + k]; s += v * v; }
        const float nrm = sqrtf(wave_sum(s));
        const float c = (t < nvalid && nrm > 0.f) ? a.gp_coef * 2.f / (float)a.B * (nrm - 1.f) / nrm : 0.f;
        if (lane == 0) rowf[t] = c;
        if (t < nvalid) pen += (nrm - 1.f) * (nrm - 1.f);
      }
      if (lane == 0) red[wave] = pen;
      __syncthreads();
      if (tid == 0 && a.scalars) {
        float s = 0.f;
        for (int w = 0; w < DNW; ++w) s += red[w];
        if (s != 0.f) rg_acc(a.scalars + 2, s * a.gp_coef / (float)a.B);
      }
      for (int c = tid; c < TR * d; c += DTHREADS) {
        const int t = c / d, k = c - t * d;
        dgt[t * lddg + k] = (T)(rowf[t] * G[t * ldg + k]);
      }
      __syncthreads();
      disc_flush(dgt, lddg, at(X1, srow, d), d, d, TR, nvf);
    }
    if (st == 6) disc_flush(bufB, ldB, at(X2, srow, n1), n1, n1, TR, nvf);
    if (st == 7) disc_flush(bufA, ldA, at(X3, srow, n2), n2, n2, TR, nvf);
    if (st == 8 && wg && a.dw4) {                             // dw4 += column sums of e3' (rows >= nvalid hold zeros)
      for (int f = tid; f < n3; f += DTHREADS) {
        float s = 0.f;
        for (int t = 0; t < TR; ++t) s += (float)bufB[t * ldB + f];
        if (s != 0.f) rg_acc(a.dw4 + f, s);
      }
    }
  }
  DSTAMP(12);
}

static size_t disc_lds_bytes(const rg_disc_args& a, int dtype) {
  const size_t es = dtype == RG_BF16 ? 2 : 4;
  const int tr = dtype == RG_BF16 ? 32 : 16;
  const int ldA = a.n2 + DPAD, ldB = (a.n1 > a.n3 ? a.n1 : a.n3) + DPAD;
  const int words = ((a.n1 + 31) >> 5) + ((a.n2 + 31) >> 5) + ((a.n3 + 31) >> 5);
  return (size_t)tr * (ldA + ldB) * es + (size_t)(a.n1 + a.n2 + 2 * a.n3) * 4 + (size_t)(tr + DNW) * 4 + (size_t)tr * words * 4 + 16;
}

extern "C" int rg_disc_supported(int d, int n1, int n2, int n3, int dtype) {
  rg_disc_args a;
  a.d = d; a.n1 = n1; a.n2 = n2; a.n3 = n3;
  if (d <= 0 || (d & 31) || (n1 & 31) || (n2 & 31) || (n3 & 31)) return 0;
  if (n2 < n1 || n2 < n3 || n2 < d) return 0;                          // bufA is the wide buffer
  const size_t es = dtype == RG_BF16 ? 2 : 4;
  const int tr = dtype == RG_BF16 ? 32 : 16;
  // the g / dg stage lives in bufA: [TR][d + 4] f32 followed by [TR][d + DPAD] T
  if ((size_t)tr * (d + 4) * 4 + (size_t)tr * (d + DPAD) * es > (size_t)tr * (n2 + DPAD) * es) return 0;
  return disc_lds_bytes(a, dtype) <= 160 * 1024 ? 1 : 0;
}

extern "C" int rg_disc_rows(const rg_disc_args* a, int dtype, void* stream) {
  if (!a || a->B <= 0) return rg_set_error_msg(RG_ERR_INVALID, "disc_rows: empty problem");
  if (dtype != RG_BF16 && dtype != RG_F32 && dtype != RG_X3) return rg_set_error_msg(RG_ERR_INVALID, "disc_rows: bad dtype");
  if (!rg_disc_supported(a->d, a->n1, a->n2, a->n3, dtype))
    return rg_set_error_msg(RG_ERR_UNSUPPORTED, "disc_rows: widths must be multiples of 32 with n2 the widest and the tile must fit LDS");
  if (!a->real || !a->fake || !a->W1 || !a->W2 || !a->W3 || !a->W1t || !a->W2t || !a->W3t || !a->b1 || !a->b2 || !a->b3 ||
      !a->w4 || !a->b4)
    return rg_set_error_msg(RG_ERR_INVALID, "disc_rows: null operand");
  if (a->alpha && (!a->X1 || !a->X2 || !a->X3 || !a->Y1 || !a->Y2 || !a->Y3))
    return rg_set_error_msg(RG_ERR_INVALID, "disc_rows: the gradient-penalty rows need the X / Y operands");
  const int tr = dtype == RG_BF16 ? 32 : 16;
  const int n_wt = (2 * a->B + tr - 1) / tr, n_gt = a->alpha ? (a->B + tr - 1) / tr : 0;
  const size_t lds = disc_lds_bytes(*a, dtype);
  hipStream_t s = (hipStream_t)stream;
  if (dtype == RG_BF16) {
    static bool attr_b = false;
    if (!attr_b) { hipFuncSetAttribute((const void*)disc_rows_kernel<__bf16>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); attr_b = true; }
    hipLaunchKernelGGL(disc_rows_kernel<__bf16>, dim3(n_wt + n_gt), dim3(DTHREADS), lds, s, *a);
  } else if (dtype == RG_X3) {
    // bf16x3: the f32 tier's buffers, tiles and fragment-packed f32 weights; every fragment is split where it is used (a weight
    // fragment serves one 16-row tile here, so there is nothing to share a presplit copy with)
    static bool attr_x = false;
    if (!attr_x) { hipFuncSetAttribute((const void*)disc_rows_kernel<x3>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); attr_x = true; }
    hipLaunchKernelGGL(disc_rows_kernel<x3>, dim3(n_wt + n_gt), dim3(DTHREADS), lds, s, *a);
  } else {
    static bool attr_f = false;
    if (!attr_f) { hipFuncSetAttribute((const void*)disc_rows_kernel<float>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); attr_f = true; }
    hipLaunchKernelGGL(disc_rows_kernel<float>, dim3(n_wt + n_gt), dim3(DTHREADS), lds, s, *a);
  }
  RG_CHECK_LAUNCH();
  return 0;
}
