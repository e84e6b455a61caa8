// Persistent, weight-stationary GEMM for the hot Linear shapes of the path (bf16 tier):
//   C[M,N] = epi( A[M,K] . W[N,K]^T + bias ),  K, N multiples of 128, K*N <= 64K, (K==128 or N==128)
// i.e. QKV projection (128->384), FFN backward (128->512 with GELU', 512->128 + residual),
// attention backward-data (384->128 + residual, 128->128).
//
// Why a second GEMM kernel: at these shapes a tile has only 32..128 MFMAs per wave, so the generic
// tile-per-workgroup kernel (gemm.hip) is dominated by per-tile fixed costs (weight-fragment
// reloads, exposed staging latency, barrier drains).  Here a workgroup is persistent: every wave
// loads ITS slice of W into registers once (K*N/512 VGPRs) and then streams 64-token tiles:
// next tile's rows are prefetched into registers while the current tile is in the MFMA phase,
// barriers order LDS only (lds_barrier), and -- weight fragment as the A operand -- each
// accumulator register holds 4 consecutive features of one token, so results leave through a
// packed LDS tile and 16-byte coalesced stores (aux operands are read the same way).
//
// bf16x3 tier (T = x3): A and C are f32 in memory; the A tile is split ONCE while it is staged (a hi and a lo bf16 tile in
// LDS, rg_common.hip.h), the weight slice is split once per workgroup (K = 128: stationary) or per tile and chunk (K > 128:
// the slice of the current 128-wide chunk, re-read from L2 -- a stationary K = 384 slice would be 192 VGPRs), the C tile is raw
// f32.  One 128-column block per workgroup (gridDim.x): HBM-bound at twice the bf16 tier's bytes.
#define WS_M 64
#define WS_LD (128 + 8)
#define WS_PLANE (WS_M * WS_LD * 2)      // bf16x3: bytes between the hi and the lo image of the A tile
#include <type_traits>
#include <cstdlib>
#include "rg_common.hip.h"
#include "../../include/recguru_hip.h"

// AUX: the epilogue reads an aux operand (compile-time: a load under a runtime condition gets a vmcnt(0) at the join,
// which also drained the next tile's A prefetch -- one exposed HBM latency per tile even for the aux-free GEMMs)
// WP (bf16x3, K > 128): 0 the weight slice is f32 and split in the kernel; 1 it is the PRESPLIT fragment-packed copy (rg_gemm_nt_args.w_packed),
// streamed per tile and chunk like the f32 one; 2 (K = 256, no aux operand) presplit AND stationary -- 16 fragments = 128 VGPRs, affordable
// once the in-kernel split's temporaries are gone and the activation fragments are read one k-step at a time
template <typename T, int NKC, int NCB, bool AUX, int WP = 0>
__global__ __launch_bounds__(256, 2) void gemm_ws_kernel(rg_gemm_nt_args a) {
  constexpr bool PSTAT = WP == 2;
  static_assert(WP == 0 || (std::is_same<T, x3>::value && NKC == 2), "presplit weights: bf16x3, K = 256");
  static_assert(NKC == 1 || NCB == 1, "either K or N spans a single 128-wide block");
  constexpr bool X3 = std::is_same<T, x3>::value;
  static_assert(!X3 || NCB == 1, "bf16x3: one column block per workgroup");
  static_assert(!PSTAT || (X3 && NKC == 2), "PSTAT: bf16x3, K = 256");
  constexpr bool STREAM = X3 && NKC > 1 && !PSTAT;   // the weight slice of ONE chunk in registers, reloaded per tile and chunk
  typedef typename LdsT<T, WS_PLANE>::type LT;    // A tile: T, or a hi and a lo bf16 tile
  typedef typename ResT<T>::type XT;              // C tile: T, or raw f32
  typedef typename OpT<T>::type OP;
  __shared__ __align__(16) LT As[WS_M * WS_LD * LdsT<T, WS_PLANE>::PLANES];
  __shared__ __align__(16) XT Cs[WS_M * WS_LD];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 15, lg = lane >> 4;
  const T* __restrict__ A = reinterpret_cast<const T*>(a.A);
  const T* __restrict__ W = reinterpret_cast<const T*>(a.W);
  // (no __restrict__ on aux / C: the split-K dx product accumulates in place, aux == C -- a tile's aux rows are read before its
  // C rows are written, by the same threads)
  const T* aux = reinterpret_cast<const T*>(a.aux);
  T* C = reinterpret_cast<T*>(a.C);
  const int n0 = wave * 32;
  const DropCfg drop = make_drop(a.epilogue == RG_EPI_DROP_GELU ? a.drop_p : 0.f, a.drop_seed);
  // cby: first 128-column block of this workgroup (N > 128 with K > 128 -- the d_model = 256 shapes of config-5: a
  // workgroup keeps the K x 128 slice of W of ITS column block in registers and walks the row tiles; A is re-read per block
  // from L2 / Infinity Cache)
  // Grid (column blocks, walkers): the column blocks of one walker are CONSECUTIVE workgroups, so the N / 128 workgroups that
  // read the same rows of A run at the same time and all but the first find them in the Infinity Cache (with the walkers on the
  // fast axis the re-reads were a whole pass over A apart: 840 MB at config-5, 420 MB in the bf16x3 tier -- HBM every time)
  // XCD-aware (round 4): workgroups are dealt round-robin to the 8 XCDs (blocks b and b + 8 share one, each XCD has its own
  // 4 MB L2), so the ny column blocks of ONE walker are consecutive rounds of the SAME XCD: the first fetches a row tile of A over
  // the fabric, the other ny - 1 read it from that XCD's L2 (with consecutive block ids they sat on ny different XCDs and every one
  // of them fetched A over the fabric: FETCH_SIZE 2.3 x the algorithmic bytes on the 256 -> 768 projection of config-5).
  // Block i: xcd = i & 7, round r = i >> 3 -> column block r % ny of walker (r / ny) * 8 + xcd.  Only speed depends on it.
  const int ny = (a.N >> 7) / NCB, nwg = (int)gridDim.x / ny;
  int cby, wg;
  if (ny > 1 && (nwg & 7) == 0) {
    const int i = (int)blockIdx.x, r = i >> 3;
    cby = r % ny;
    wg = (r / ny) * 8 + (i & 7);
  } else {
    cby = (int)blockIdx.x % ny;
    wg = (int)blockIdx.x / ny;
  }
  const int ntiles = (a.M + WS_M - 1) / WS_M;
  // element offset of the 16-byte chunk (row m, block cb, columns c8..c8+7) of C: row-major, or head-major (c_hm_L: N = 3 * H * 32
  // columns q | k | v -> three tensors [B][H][L][32]; a 128-column block is four heads of one tensor -- block cb = tensor
  // cb / (H/4), heads 4 * (cb % (H/4)) + c8 / 32 -- and four consecutive rows of one head are 256 contiguous bytes)
  const int hmL = a.c_hm_L, hmB = hmL > 0 ? a.M / hmL : 0, hmBpt = a.N / 384;      // 128-column blocks per tensor = H / 4
  // m = mt + r with mt the first row of a 16-row tile (workgroup-uniform: its division by L is scalar work) and r < 16
  auto c_off = [&](int mt, int r, int cb, int c8) -> size_t {
    if (hmL > 0) {
      const int bt = __builtin_amdgcn_readfirstlane(mt) / hmL;
      int bb = bt, l = mt - bt * hmL + r;
      if (l >= hmL) { l -= hmL; ++bb; }                  // (a 16-row tile spans at most two sequences: L >= 16 is checked by the launcher)
      const int w = cb / hmBpt, hb = cb - w * hmBpt;     // (cb is workgroup-uniform: scalar)
      return ((size_t)(((w * hmB + bb) * hmBpt + hb) * 4 + (c8 >> 5)) * hmL + l) * 32 + (c8 & 31);
    }
    return (size_t)(mt + r) * a.ldc + cb * 128 + c8;
  };

  // ---- stationary weights: w[cb][kc][ks][ct]   (STREAM: [cb][0][ks][ct] = the current chunk's slice)
  OP w[NCB][STREAM ? 1 : NKC][4][2];
  auto load_w = [&](int cb, int kc, int slot) {
    if constexpr (WP != 0) {
      {
        // presplit fragment-packed copy (rg_cast RG_CAST_PACK | RG_CAST_SPLIT): fragment (row tile, k-step) is 2 KB -- 64 lanes x 16 B of
        // hi parts, then the same of lo parts; 512 four-byte slots, row tiles outermost.  Two 16-byte loads, no split arithmetic.
        const unsigned int nks = (unsigned int)a.ldw >> 5;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
#pragma unroll
          for (int ct = 0; ct < 2; ++ct) {
            const unsigned int rt16 = (unsigned int)(((cby + cb) * 128 + n0 + ct * 16) >> 4);
            const float* fp = reinterpret_cast<const float*>(W) + ((size_t)rt16 * nks + (unsigned int)(kc * 4 + ks)) * 512u + (unsigned int)(lg * 16 + li) * 4u;
            w[cb][slot][ks][ct].hi = *reinterpret_cast<const bf16x8_t*>(fp);
            w[cb][slot][ks][ct].lo = *reinterpret_cast<const bf16x8_t*>(fp + 256);
          }
        return;
      }
    }
#pragma unroll
    for (int ks = 0; ks < 4; ++ks)
#pragma unroll
      for (int ct = 0; ct < 2; ++ct)
        load_frag(w[cb][slot][ks][ct], W + (size_t)((cby + cb) * 128 + n0 + ct * 16 + li) * a.ldw + kc * 128 + ks * 32 + 8 * lg);
  };
  if constexpr (!STREAM) {
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
      for (int kc = 0; kc < NKC; ++kc) load_w(cb, kc, kc);
  }
  __shared__ __align__(16) float bs[NCB * 128];
  for (int i = tid; i < NCB * 128; i += 256) bs[i] = a.bias ? a.bias[cby * 128 + i] : 0.f;   // visible after the first barrier

  // A work tile is 4 row tiles of 16 rows at rows mb[0..3] (>= M: absent): 64 consecutive rows, or -- with the list
  // of live 16-row tiles a.live16 (rg_live_tiles) -- 4 consecutive list entries; rows of the padded tiles are not read
  // and their rows of C are written as zeros at the end.  Thread tid stages chunk i = row 16 i + (tid >> 4).
  LiveWalk lw;
  lw.init(a.live16, a.M, wg, nwg);
  const int nwork = a.live16 ? (lw.nlive + 3) >> 2 : ntiles;
  auto group = [&](int k, int (&g)[4]) {                 // k-th work tile of this workgroup
    if (!a.live16) {
      const int wt = wg + k * nwg;
#pragma unroll
      for (int rt = 0; rt < 4; ++rt) g[rt] = wt * WS_M + 16 * rt;
    } else {
      lw.group(k, g, a.M);
    }
  };
  // chunk (tile, kc) -> the tile's rows, columns kc*128.. of A ; each thread stages 4 x 16 bytes
  Frag<T> pre[4];
  auto prefetch = [&](const int (&g)[4], int kc) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {      // rows >= M: clamped address, no branch (their results are never stored)
      const int c8 = (tid & 15) * 8, m = min(g[i] + (tid >> 4), a.M - 1);
      load_frag(pre[i], A + (size_t)m * a.lda + kc * 128 + c8);
    }
  };
  // aux rows (epilogue operand) of the NEXT epilogue step -- the next 128-feature block of this tile, or block 0 of
  // the next tile -- are loaded a whole step ahead: loaded inside the step they cost one exposed HBM latency per
  // block (the MFMAs of one block cover a fifth of it)
  Frag<T> axn[4];
  auto aux_prefetch = [&](const int (&g)[4], int cb) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int c8 = (tid & 15) * 8, m = min(g[i] + (tid >> 4), a.M - 1);
      load_frag(axn[i], aux + (size_t)m * a.ldaux + (cby + cb) * 128 + c8);
    }
  };
  int tile = wg, kt_ = 0;                                 // kt_: index of `tile` among this workgroup's tiles
  int mb[4], mbn[4];
  if (tile < nwork) {
    group(0, mb);
    prefetch(mb, 0);
    if constexpr (AUX) aux_prefetch(mb, 0);
  }
  for (; tile < nwork; tile += nwg, ++kt_) {
    group(kt_ + 1, mbn);
    f32x4 acc[2][4];
#pragma unroll
    for (int kc = 0; kc < NKC; ++kc) {
      // registers -> LDS, then immediately put the next chunk in flight
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int c = tid + 256 * i, r = c >> 4, c8 = (c & 15) * 8;
        stage8(As + r * WS_LD + c8, pre[i]);
      }
      if constexpr (STREAM) load_w(0, kc, 0);             // (L2 hits; split in registers under the barrier below)
      // unconditional (rows of an absent next tile are clamped inside prefetch): loads under a runtime branch get a
      // vmcnt(0) at the join, which exposed one HBM latency per tile
      if (kc + 1 < NKC) prefetch(mb, kc + 1);
      else prefetch(mbn, 0);
      lds_barrier();
#pragma unroll
      for (int cb = 0; cb < NCB; ++cb) {
        Frag<T> axf[4];                                   // aux rows of this block (prefetched one step ago)
        if constexpr (AUX) if (kc == NKC - 1) {
#pragma unroll
          for (int i = 0; i < 4; ++i) axf[i] = axn[i];
          if (cb + 1 < NCB) aux_prefetch(mb, cb + 1);
          else aux_prefetch(mbn, 0);
        }
        if (kc == 0) {
#pragma unroll
          for (int ct = 0; ct < 2; ++ct) {
            float b4[4];
            load4f(b4, bs + cb * 128 + n0 + ct * 16 + 4 * lg);
#pragma unroll
            for (int rt = 0; rt < 4; ++rt) acc[ct][rt] = (f32x4){b4[0], b4[1], b4[2], b4[3]};
          }
        }
        if constexpr (PSTAT) {                            // one set of activation fragments (the stationary slice took the registers of the second)
#pragma unroll
          for (int ks = 0; ks < 4; ++ks) {
            OP af1[4];
#pragma unroll
            for (int rt = 0; rt < 4; ++rt) load_frag(af1[rt], As + (rt * 16 + li) * WS_LD + ks * 32 + 8 * lg);
#pragma unroll
            for (int rt = 0; rt < 4; ++rt)
#pragma unroll
              for (int ct = 0; ct < 2; ++ct) mma(w[cb][kc][ks][ct], af1[rt], acc[ct][rt]);
          }
        } else {
        OP af[2][4];                                      // k-step ks+1's fragments are read under the MFMAs of k-step ks
#pragma unroll
        for (int rt = 0; rt < 4; ++rt) load_frag(af[0][rt], As + (rt * 16 + li) * WS_LD + 8 * lg);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
          if (ks < 3) {
#pragma unroll
            for (int rt = 0; rt < 4; ++rt) load_frag(af[(ks + 1) & 1][rt], As + (rt * 16 + li) * WS_LD + (ks + 1) * 32 + 8 * lg);
          }
#pragma unroll
          for (int rt = 0; rt < 4; ++rt)
#pragma unroll
            for (int ct = 0; ct < 2; ++ct) mma(w[cb][STREAM ? 0 : kc][ks][ct], af[ks & 1][rt], acc[ct][rt]);
        }
        }
        if (kc == NKC - 1) {
          // ---- epilogue of this 128-feature block: packed tile -> coalesced 16-byte stores
#pragma unroll
          for (int ct = 0; ct < 2; ++ct)
#pragma unroll
            for (int rt = 0; rt < 4; ++rt) {
              float v[4] = {acc[ct][rt][0], acc[ct][rt][1], acc[ct][rt][2], acc[ct][rt][3]};
              store4(Cs + (rt * 16 + li) * WS_LD + n0 + ct * 16 + 4 * lg, v);
            }
          lds_barrier();
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const int r = 16 * i + (tid >> 4), c8 = (tid & 15) * 8;
            const int m = mb[i] + (tid >> 4);
            if (m < a.M) {
              const size_t off = c_off(mb[i], tid >> 4, cby + cb, c8);
              if constexpr (!AUX) {
                if (a.epilogue == RG_EPI_NONE) {
                  Frag<T> raw;
                  unstage8(raw, Cs + r * WS_LD + c8);
#ifdef RG_ABL_WS_NT      // timing experiment (tools/ab_round6.sh x3nt): the plain-epilogue output rows of the f32-storage tiers by nontemporal stores
                  if constexpr (sizeof(T) == 4) frag_store_nt(C + off, raw);
                  else *reinterpret_cast<Frag<T>*>(C + off) = raw;
#else
                  *reinterpret_cast<Frag<T>*>(C + off) = raw;
#endif
                } else if (a.epilogue == RG_EPI_DROP_GELU) {
                  // h1 = dropout(l1) as the backward reads it back, and the activated operand of the second product, from
                  // the value AS STORED (rounded to T): what rg_dropout_gelu does in a pass of its own
                  float v[8];
                  load8(v, Cs + r * WS_LD + c8);
                  if (drop.thresh) {
                    float k8[8];
                    rg_keep8(drop, (unsigned int)m * (unsigned int)a.N + (unsigned int)((cby + cb) * 128 + c8), k8);
#pragma unroll
                    for (int j = 0; j < 8; ++j) v[j] *= k8[j];
                  }
                  store8(C + off, v);
#pragma unroll
                  for (int j = 0; j < 8; ++j) v[j] = gelu_t<false>((float)(T)v[j]);
                  store8(reinterpret_cast<T*>(a.C2) + off, v);
                } else {              // RG_EPI_RELU
                  float v[8];
                  load8(v, Cs + r * WS_LD + c8);
#pragma unroll
                  for (int j = 0; j < 8; ++j) v[j] = fmaxf(v[j], 0.f);
                  store8(C + off, v);
                }
              } else {
                float v[8], x[8];
                load8(v, Cs + r * WS_LD + c8);
#pragma unroll
                for (int j = 0; j < 8; ++j) x[j] = (float)axf[i].v[j];
                // the (uniform) epilogue switch sits OUTSIDE the element loops: inside, every element became its own
                // basic block and the 8 dependent exp / rcp chains of a vector ran one after the other
                if (a.epilogue == RG_EPI_GELU_GRAD) {
                  const float nz = a.epi_nonzero_scale;
#pragma unroll
                  for (int j = 0; j < 8; ++j) v[j] *= gelu_grad_t<false>(x[j]);
                  if (nz > 0.f) {
#pragma unroll
                    for (int j = 0; j < 8; ++j) v[j] = x[j] != 0.f ? v[j] * nz : 0.f;
                  }
                } else if (a.epilogue == RG_EPI_ADD) {
#pragma unroll
                  for (int j = 0; j < 8; ++j) v[j] += x[j];
                } else {              // RG_EPI_MUL_POSMASK
                  const float sc = a.epi_scale > 0.f ? a.epi_scale : 1.f;
#pragma unroll
                  for (int j = 0; j < 8; ++j) v[j] = x[j] > 0.f ? v[j] * sc : 0.f;
                }
                store8(C + off, v);
              }
            }
          }
          if (NCB > 1 && cb + 1 < NCB) lds_barrier();     // Cs is rewritten by the next block
        }
      }
      if (NKC > 1 || true) lds_barrier();                 // As / Cs are rewritten by the next chunk / tile
    }
#pragma unroll
    for (int rt = 0; rt < 4; ++rt) mb[rt] = mbn[rt];
  }
  if (a.live16 && a.skip_dead_fill != 1) {                // rows of the padded tiles: C = 0, or C = bias (skip_dead_fill == 2:
    const int nrt = (a.M + 15) >> 4, ndead = nrt - a.live16[0];     // the caller guarantees that those rows of A are zero)
    Frag<T> z[NCB];
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb) {
      if (a.skip_dead_fill == 2 && a.epilogue == RG_EPI_NONE) {
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = bs[cb * 128 + (tid & 15) * 8 + j];
        store8(reinterpret_cast<T*>(&z[cb]), v);
      } else {
        frag_zero(z[cb]);
      }
    }
    for (int j = 4 * wg; j < ndead; j += 4 * nwg) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        if (j + i < ndead) {
          const int mt = a.live16[nrt - (j + i)] * 16, m = mt + (tid >> 4);
          if (m < a.M) {
#pragma unroll
            for (int cb = 0; cb < NCB; ++cb)
              *reinterpret_cast<Frag<T>*>(C + c_off(mt, tid >> 4, cby + cb, (tid & 15) * 8)) = z[cb];
          }
        }
      }
    }
  }
}

template <int NKC, int NCB, typename T = __bf16>
static int launch_ws(const rg_gemm_nt_args& a, hipStream_t s, int ny = 1) {
  const int ntiles = (a.M + WS_M - 1) / WS_M;
  // walkers per column block: ny * grid workgroups must fit the chip at once (persistent walkers: a second round of workgroups
  // would run with half the CUs idle).  bf16: 35 KB of LDS, <= 128 VGPRs -- up to four per CU (768 measured best); bf16x3: two A
  // planes + an f32 C tile = 70 KB -- two per CU, 512 slots
  const int slots = std::is_same<T, x3>::value ? 512 : 768;
  static const int slots_env = getenv("RG_WS_SLOTS") ? atoi(getenv("RG_WS_SLOTS")) : 0;
  int grid = ny > 1 ? ((slots_env ? slots_env : slots) / ny > 96 ? (slots_env ? slots_env : slots) / ny : 96) : 512;
  if (grid > ntiles) grid = ntiles;
  static const int no_xcd = getenv("RG_WS_NO_XCD") ? atoi(getenv("RG_WS_NO_XCD")) : 0;
  if (ny > 1 && grid >= 16) grid = (grid & ~7) - (no_xcd ? 1 : 0);     // walkers in multiples of 8: the XCD-aware block mapping (RG_WS_NO_XCD=1: one walker fewer = the plain mapping, for A/B)
  if constexpr (std::is_same<T, x3>::value && NKC == 2) {
    if (a.w_packed) {
      static const int pstat = [] { const char* e = getenv("RG_WS_PSTAT"); return e ? atoi(e) : 1; }();   // RG_WS_PSTAT=0: stream the presplit slice per tile (A/B)
      const bool aux_epi = a.epilogue != RG_EPI_NONE && a.epilogue != RG_EPI_RELU && a.epilogue != RG_EPI_DROP_GELU;
      if (aux_epi) hipLaunchKernelGGL((gemm_ws_kernel<T, NKC, NCB, true, 1>), dim3(ny * grid), dim3(256), 0, s, a);
      else if (pstat) hipLaunchKernelGGL((gemm_ws_kernel<T, NKC, NCB, false, 2>), dim3(ny * grid), dim3(256), 0, s, a);
      else hipLaunchKernelGGL((gemm_ws_kernel<T, NKC, NCB, false, 1>), dim3(ny * grid), dim3(256), 0, s, a);
      RG_CHECK_LAUNCH();
      return 0;
    }
  }
  if (a.epilogue != RG_EPI_NONE && a.epilogue != RG_EPI_RELU && a.epilogue != RG_EPI_DROP_GELU) hipLaunchKernelGGL((gemm_ws_kernel<T, NKC, NCB, true>), dim3(ny * grid), dim3(256), 0, s, a);
  else hipLaunchKernelGGL((gemm_ws_kernel<T, NKC, NCB, false>), dim3(ny * grid), dim3(256), 0, s, a);
  RG_CHECK_LAUNCH();
  return 0;
}

// 10 * (K/128) + N/128 of the instantiation that takes this problem, 0 if none does
int rg_gemm_ws_select(const rg_gemm_nt_args* a, int dtype) {
  if (a->w_packed && (dtype != RG_X3 || a->K != 256 || a->ldw != a->K)) return 0;      // presplit weights: the bf16x3 form at K = 256 only (at K = 384 / 512 the
                                                                                      // presplit loads of all chunks are hoisted and the kernel spills 36 - 73 VGPRs: no gain measured)
  if (dtype == RG_X3) {             // bf16x3: <K/128, 1> once per 128-column block; row-major f32 output, no head-major form
    if (a->prologue != RG_PRO_NONE || a->epilogue == RG_EPI_RESID_LN || a->c_hm_L > 0) return 0;
    if (a->epilogue == RG_EPI_DROP_GELU && (!a->C2 || a->aux || a->live16)) return 0;      // (round 5: the dropout + GELU epilogue on f32 tiles too)
    if (a->epilogue == RG_EPI_RELU && a->drop_p > 0.f) return 0;
    if ((a->K & 127) || (a->N & 127) || a->M < 4096 || a->K > 512 || a->N > 1024) return 0;
    if ((a->lda & 7) || (a->ldw & 7) || (a->ldc & 7) || (a->aux && (a->ldaux & 7))) return 0;
    if (a->epilogue != RG_EPI_NONE && a->epilogue != RG_EPI_RELU && a->epilogue != RG_EPI_DROP_GELU && !a->aux) return 0;
    return 10 * (a->K / 128) + 1;
  }
  if (dtype != RG_BF16 || a->c_is_f32 || a->prologue != RG_PRO_NONE || a->epilogue == RG_EPI_RESID_LN) return 0;
  if (a->epilogue == RG_EPI_RELU && a->drop_p > 0.f) return 0;   // dropout-after-ReLU lives in the generic kernel
  if ((a->K & 127) || (a->N & 127) || a->M < 4096) return 0;
  if ((a->lda & 7) || (a->ldw & 7) || (a->ldc & 7) || (a->aux && (a->ldaux & 7))) return 0;
  if (a->c_hm_L > 0 && ((a->M % a->c_hm_L) != 0 || a->c_hm_L < 16 || (a->N % 384) != 0)) return 0;
  if (a->epilogue == RG_EPI_DROP_GELU) { if (!a->C2 || a->aux || a->c_hm_L > 0 || a->live16) return 0; }
  else if (a->epilogue != RG_EPI_NONE && !a->aux) return 0;
  const int nkc = a->K / 128, ncb = a->N / 128;
  if ((nkc == 1 && ncb >= 1 && ncb <= 4) || (ncb == 1 && nkc >= 2 && nkc <= 4)) return 10 * nkc + ncb;
  // K and N both beyond 128 (d_model = 256: 256 -> 768 / 512 / 256, 512 -> 256): the <K/128, 1> instantiation once per
  // 128-column block (gridDim.x)
  if (nkc >= 2 && nkc <= 4 && ncb >= 2 && ncb <= 8) return 10 * nkc + 1;
  return 0;
}

// returns 1 if the shape is not handled here (caller falls back to the generic kernel)
int rg_gemm_ws_try(const rg_gemm_nt_args* a, int dtype, hipStream_t s) {
  if (!rg_gemm_ws_select(a, dtype)) return 1;
  const int nkc = a->K / 128, ncb = a->N / 128;
  if (dtype == RG_X3) {
    if (nkc == 1) return launch_ws<1, 1, x3>(*a, s, ncb);
    if (nkc == 2) return launch_ws<2, 1, x3>(*a, s, ncb);
    if (nkc == 3) return launch_ws<3, 1, x3>(*a, s, ncb);
    return launch_ws<4, 1, x3>(*a, s, ncb);
  }
  if (nkc == 1 && ncb == 1) return launch_ws<1, 1>(*a, s);
  if (nkc == 1 && ncb == 2) return launch_ws<1, 2>(*a, s);
  if (nkc == 2 && ncb == 1) return launch_ws<2, 1>(*a, s);
  if (nkc == 1 && ncb == 3) return launch_ws<1, 3>(*a, s);
  if (nkc == 1 && ncb == 4) return launch_ws<1, 4>(*a, s);
  if (nkc == 3 && ncb == 1) return launch_ws<3, 1>(*a, s);
  if (nkc == 4 && ncb == 1) return launch_ws<4, 1>(*a, s);
  if (nkc == 2) return launch_ws<2, 1>(*a, s, ncb);
  if (nkc == 3) return launch_ws<3, 1>(*a, s, ncb);
  if (nkc == 4) return launch_ws<4, 1>(*a, s, ncb);
  return 1;
}
