// HBM-bound kernels of the path: embedding gather / scatter, LayerNorm backward, the collapsed
// cross-attention broadcast+LayerNorm, gradient-penalty helpers, Adam, casts and reductions.
// All of them move 8-16 bytes per lane per access with lanes on consecutive addresses.
#include <stdlib.h>
#include "rg_common.hip.h"
#include "rg_det.hip.h"
#include "../../include/recguru_hip.h"

#define EW_BLOCK 256
static inline int ew_grid(long long work_items, int per_block) {
  long long g = (work_items + per_block - 1) / per_block;
  if (g < 1) g = 1;
  if (g > 256LL * 16) g = 256LL * 16;  // grid-stride beyond ~16 blocks per CU
  return (int)g;
}

// ------------------------------------------------------------------------------------------------
// K1: x[b,t,:] = (E[ids[b,t],:] + pe[t,:]) * mask[b,t]       (PE added BEFORE masking, quirk Q6)
// nn.Embedding lookup + PositionalEncoding.forward, transformer.py:104-106, AutoEnc4Rec_cross.py:98-102
// ------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(EW_BLOCK) void embed_pe_fwd_kernel(const T* __restrict__ table, const float* __restrict__ pe,
                                                               const int64_t* __restrict__ ids,
                                                               const float* __restrict__ mask, T* __restrict__ out,
                                                               long long ntok, int L, int d, DropCfg drop) {
  const int cpr = d >> 3;
  const long long total = ntok * cpr;
  for (long long i = (long long)blockIdx.x * EW_BLOCK + threadIdx.x; i < total; i += (long long)gridDim.x * EW_BLOCK) {
    const long long tok = i / cpr;
    const int c8 = (int)(i - tok * cpr) * 8;
    const float m = mask[tok];
    const int64_t id = ids[tok];          // with the mask, not behind it: mask -> id -> row was three dependent latencies
    float v[8];
    if (m != 0.f) {
      float p[8];
      load8(v, table + (size_t)id * d + c8);
      load8(p, pe + (size_t)(tok % L) * d + c8);
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = (v[j] + p[j]) * m;
      if (drop.thresh) {
        float k8[8];
        rg_keep8(drop, (unsigned int)tok * (unsigned int)d + (unsigned int)c8, k8);
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] *= k8[j];
      }
    } else {
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = 0.f;
    }
    store8(out + (size_t)tok * d + c8, v);
  }
}

// Row form for d_model 128 / 256 (round 5): a row is D / 8 lanes x 8 elements, a wave takes 64 consecutive tokens at a time -- ids and
// mask values arrive by ONE coalesced load per wave (lane t: token t0 + t) and reach the lane groups through ds_bpermute --, U rows
// per lane group are in flight together, and every index is 32-bit (the element-per-thread form above pays two 64-bit divisions
// per 16 bytes).  On-box ceiling for this access pattern (tools/peaks.hip, gather_copy_512B: random 512-B rows of a 1 GiB table copied
// out): 5.3 - 5.4 TB/s.  M2: a second, bf16 copy of the output rows (mixed tier: the operand copy the bf16 backward reads).
template <typename T, int D, int U, bool M2>
__global__ __launch_bounds__(EW_BLOCK) void embed_pe_fwd_rows_kernel(const T* __restrict__ table, const float* __restrict__ pe,
                                                                    const int64_t* __restrict__ ids, const float* __restrict__ mask,
                                                                    T* __restrict__ out, __bf16* __restrict__ out2, int ntok, int L, DropCfg drop) {
  constexpr int LPR = D / 8, RPW = 64 / LPR;           // lanes per row, rows per wave instruction
  const int lane = threadIdx.x & 63;
  const int wave = (int)((blockIdx.x * EW_BLOCK + threadIdx.x) >> 6), nwave = (int)gridDim.x * (EW_BLOCK / 64);
  const int lr = lane / LPR, c8 = (lane % LPR) * 8;
  for (int t0 = wave * 64; t0 < ntok; t0 += nwave * 64) {
    const int tl = min(t0 + lane, ntok - 1);
    const int64_t idl = ids[tl];
    const float ml = (t0 + lane < ntok) ? mask[tl] : 0.f;
    const int idlo = (int)(idl & 0xFFFFFFFFll), idhi = (int)(idl >> 32);
    const int p0 = __builtin_amdgcn_readfirstlane(t0) % L;           // (uniform: one scalar division per 64 tokens)
#pragma unroll 1
    for (int g = 0; g < 64; g += RPW * U) {
      float m[U];
      int64_t id[U];
      int pos[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int src = g + u * RPW + lr;
        m[u] = __shfl(ml, src);
        id[u] = ((int64_t)__shfl(idhi, src) << 32) | (unsigned int)__shfl(idlo, src);
        pos[u] = (p0 + src) % L;
      }
      // a group of U * RPW consecutive padded positions (left padding: 44 % of the groups at the bench's lengths) only writes zeros
      bool anyl = false;
#pragma unroll
      for (int u = 0; u < U; ++u) anyl = anyl || m[u] != 0.f;
      if (__ballot(anyl) == 0ull) {                    // (wave-uniform)
        float z8[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const int tok = t0 + g + u * RPW + lr;
          if (tok < ntok) {
            store8(out + (size_t)tok * D + c8, z8);
            if constexpr (M2) store8(out2 + (size_t)tok * D + c8, z8);
          }
        }
        continue;
      }
      // UNCONDITIONAL loads inside a live group (a load under a divergent condition gets a vmcnt(0) at its join: the U rows would
      // arrive one after the other): a padded position reads row 0 of the table -- one hot line set -- and its result is discarded
      float v[U][8], pp[U][8];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        load8(v[u], table + (size_t)(m[u] != 0.f ? id[u] : 0) * D + c8);
        load8(pp[u], pe + (unsigned int)pos[u] * D + c8);
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int tok = t0 + g + u * RPW + lr;
        if (m[u] != 0.f) {
#pragma unroll
          for (int j = 0; j < 8; ++j) v[u][j] = (v[u][j] + pp[u][j]) * m[u];
          if (drop.thresh) {
            float k8[8];
            rg_keep8(drop, (unsigned int)tok * (unsigned int)D + (unsigned int)c8, k8);
#pragma unroll
            for (int j = 0; j < 8; ++j) v[u][j] *= k8[j];
          }
        } else {
#pragma unroll
          for (int j = 0; j < 8; ++j) v[u][j] = 0.f;
        }
        if (tok < ntok) {
          store8(out + (size_t)tok * D + c8, v[u]);
          if constexpr (M2) store8(out2 + (size_t)tok * D + c8, v[u]);
        }
      }
    }
  }
}

// Position-major form (round 6; tools/embed_ladder.hip names what the two forms above lose against a bare gather-copy of the same rows on
// the 1 GiB config-5 table: (1) the f32 positional row -- TWICE the bf16 row's bytes through the vector L1 per position: 83 -> 99 us --
// and (2) ordinary stores, whose lines are allocated in the caches on the way out: 81 -> 67 us with nontemporal stores, 102 -> 49 us with
// the real 56 %-live mask).  Here a wave owns RPW consecutive POSITIONS (lane group lr: position pos0 + lr) and walks a range of
// sequences b, U of them in flight: its positional rows are read ONCE into 8 registers per lane and reused for every sequence; the row
// loads are unconditional (a padded position reads row 0 of the table, one hot line set, and its result is discarded); the output rows
// leave by nontemporal stores when NT (the launcher's choice, RG_EMBED_NT).  Consecutive waves take consecutive position blocks of the same
// sequence range, so the 32-B id / 16-B mask pieces of neighbouring waves share cache lines and a workgroup writes 4 KiB runs.
template <typename T> __device__ __forceinline__ void store8_nt(T* p, const float* v);
template <> __device__ __forceinline__ void store8_nt<float>(float* p, const float* v) {
  __builtin_nontemporal_store((rg_f4){v[0], v[1], v[2], v[3]}, reinterpret_cast<rg_f4*>(p));
  __builtin_nontemporal_store((rg_f4){v[4], v[5], v[6], v[7]}, reinterpret_cast<rg_f4*>(p + 4));
}
template <> __device__ __forceinline__ void store8_nt<__bf16>(__bf16* p, const float* v) {
  union { bf16x8_t b; rg_f4 x; } u;
#pragma unroll
  for (int j = 0; j < 8; ++j) u.b[j] = (__bf16)v[j];
  __builtin_nontemporal_store(u.x, reinterpret_cast<rg_f4*>(p));
}
template <typename T, int D, int U, bool NT>
__global__ __launch_bounds__(EW_BLOCK) void embed_pe_fwd_pos_kernel(const T* __restrict__ table, const float* __restrict__ pe,
                                                                   const int64_t* __restrict__ ids, const float* __restrict__ mask,
                                                                   T* __restrict__ out, int B, int L, int bs, DropCfg drop) {
  constexpr int LPR = D / 8, RPW = 64 / LPR;           // lanes per row, positions per wave
  const int lane = threadIdx.x & 63;
  const int wave = (int)((blockIdx.x * EW_BLOCK + threadIdx.x) >> 6);
  const int npb = (L + RPW - 1) / RPW;
  const int pb = wave % npb, b0 = (wave / npb) * bs;
  if (b0 >= B) return;
  const int b1 = min(B, b0 + bs);
  const int lr = lane / LPR, c8 = (lane % LPR) * 8;
  const int pos = pb * RPW + lr;
  const bool inr = pos < L;
  const int posc = inr ? pos : L - 1;
  float pp[8];
  load8(pp, pe + (unsigned int)posc * D + c8);
#pragma unroll 1
  for (int b = b0; b < b1; b += U) {
    int64_t id[U];
    float m[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int tok = min(b + u, b1 - 1) * L + posc;
      id[u] = ids[tok];
      m[u] = mask[tok];
    }
    float v[U][8];
#pragma unroll
    for (int u = 0; u < U; ++u) load8(v[u], table + (size_t)(m[u] != 0.f ? id[u] : 0) * D + c8);
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int tok = (b + u) * L + pos;
      if (m[u] != 0.f) {
#pragma unroll
        for (int j = 0; j < 8; ++j) v[u][j] = (v[u][j] + pp[j]) * m[u];
        if (drop.thresh) {
          float k8[8];
          rg_keep8(drop, (unsigned int)tok * (unsigned int)D + (unsigned int)c8, k8);
#pragma unroll
          for (int j = 0; j < 8; ++j) v[u][j] *= k8[j];
        }
      } else {
#pragma unroll
        for (int j = 0; j < 8; ++j) v[u][j] = 0.f;
      }
      if (b + u < b1 && inr) {
        if constexpr (NT) store8_nt<T>(out + (size_t)tok * D + c8, v[u]);
        else store8(out + (size_t)tok * D + c8, v[u]);
      }
    }
  }
}

// Split-residual form: f32 master rows in, value = hi + lo out as two bf16 tensors (rg_embed_pe_fwd_split).
__global__ __launch_bounds__(EW_BLOCK) void embed_pe_fwd_split_kernel(const float* __restrict__ table, const float* __restrict__ pe,
                                                                     const int64_t* __restrict__ ids,
                                                                     const float* __restrict__ mask, __bf16* __restrict__ out,
                                                                     __bf16* __restrict__ out_lo, long long ntok, int L, int d,
                                                                     DropCfg drop) {
  const int cpr = d >> 3;
  const long long total = ntok * cpr;
  for (long long i = (long long)blockIdx.x * EW_BLOCK + threadIdx.x; i < total; i += (long long)gridDim.x * EW_BLOCK) {
    const long long tok = i / cpr;
    const int c8 = (int)(i - tok * cpr) * 8;
    const float m = mask[tok];
    const int64_t id = ids[tok];
    float v[8], lo[8];
    if (m != 0.f) {
      float p[8];
      load8(v, table + (size_t)id * d + c8);
      load8(p, pe + (size_t)(tok % L) * d + c8);
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = (v[j] + p[j]) * m;
      if (drop.thresh) {
        float k8[8];
        rg_keep8(drop, (unsigned int)tok * (unsigned int)d + (unsigned int)c8, k8);
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] *= k8[j];
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) lo[j] = v[j] - (float)(__bf16)v[j];
    } else {
#pragma unroll
      for (int j = 0; j < 8; ++j) { v[j] = 0.f; lo[j] = 0.f; }
    }
    store8(out + (size_t)tok * d + c8, v);
    store8(out_lo + (size_t)tok * d + c8, lo);
  }
}

// dE[ids[b,t],:] += dx[b,t,:] * mask[b,t]  -- one wave per token row, 256 contiguous bytes per
// atomic wave-instruction (the shape the memory-side f32 atomic unit runs at full rate).
// Item popularity is heavy-tailed (the head item of a Zipf(1) catalogue is ~8 % of all tokens), and memory-side
// atomics onto one row serialise (MI355X_MICROARCH.md, Global float atomics: one row for everybody = 14x slower).
// So every workgroup keeps ES_SLOTS privately accumulated rows in LDS, claimed first-come by id through a CAS on
// the slot tag (2 probes): the popular ids show up early and take slots, their later occurrences are LDS adds,
// and each workgroup sends ONE global add per claimed row at the end.  Ids that find both probes taken go to
// global memory directly, as before.
#define ES_SLOTS 64
#define ES_UNROLL 4          // tokens in flight per wave: the id -> row -> add chain is pure latency otherwise
template <typename T, int NPL>
__global__ __launch_bounds__(EW_BLOCK) void embed_scatter_bwd_kernel(const T* __restrict__ dx, const int64_t* __restrict__ ids,
                                                                    const float* __restrict__ mask, float* __restrict__ dE,
                                                                    long long ntok, int d, long long skip_row, DropCfg drop) {
  extern __shared__ float es_acc[];                       // [ES_SLOTS][d]
  __shared__ int es_tag[ES_SLOTS];
  const int tid = threadIdx.x, lane = tid & 63;
  for (int i = tid; i < ES_SLOTS; i += EW_BLOCK) es_tag[i] = -1;
  for (int i = tid; i < ES_SLOTS * d; i += EW_BLOCK) es_acc[i] = 0.f;
  __syncthreads();
  const long long wave = ((long long)blockIdx.x * EW_BLOCK + tid) >> 6;
  const long long nwaves = ((long long)gridDim.x * EW_BLOCK) >> 6;
  for (long long t0 = wave * ES_UNROLL; t0 < ntok; t0 += nwaves * ES_UNROLL) {
    float m[ES_UNROLL];
    T g[ES_UNROLL][NPL];        // raw: converted where used, so that the loads of all tokens are in flight together
    long long row[ES_UNROLL];
#pragma unroll
    for (int u = 0; u < ES_UNROLL; ++u) {
      const long long tok = min(t0 + u, ntok - 1);
      m[u] = mask[tok];
      row[u] = ids[tok];
#pragma unroll
      for (int j = 0; j < NPL; ++j) g[u][j] = dx[(size_t)tok * d + min(lane + 64 * j, d - 1)];
    }
#pragma unroll
    for (int u = 0; u < ES_UNROLL; ++u)
      if (t0 + u >= ntok) m[u] = 0.f;
#pragma unroll
    for (int u = 0; u < ES_UNROLL; ++u) {
      if (m[u] == 0.f || row[u] == skip_row) continue;
      int slot = -1;
      if (!RG_DET && row[u] >= 0 && row[u] < 0x7fffffffLL) {     // (deterministic build: no first-come LDS slots, every add goes to the fixed-point shadow)
        const unsigned int h = ((unsigned int)row[u] * 2654435761u) >> 26;       // 6 bits
        int got = -1;
        if (lane == 0) {
          int old = atomicCAS(&es_tag[h], -1, (int)row[u]);
          if (old == -1 || old == (int)row[u]) got = (int)h;
          else {
            const unsigned int h2 = h ^ 1u;
            old = atomicCAS(&es_tag[h2], -1, (int)row[u]);
            if (old == -1 || old == (int)row[u]) got = (int)h2;
          }
        }
        slot = __builtin_amdgcn_readfirstlane(got);
      }
#pragma unroll
      for (int j = 0; j < NPL; ++j) {
        const int e = lane + 64 * j;
        if (e < d) {
          float v = (float)g[u][j] * m[u];
          if (drop.thresh) v *= rg_keep(drop, (unsigned int)(t0 + u) * (unsigned int)d + (unsigned int)e);
          if (slot >= 0) atomicAdd(es_acc + slot * d + e, v);
          else rg_acc(dE + (size_t)row[u] * d + e, v);
        }
      }
    }
  }
  __syncthreads();
  for (int sl = tid >> 6; sl < ES_SLOTS; sl += EW_BLOCK / 64) {
    const int row = es_tag[sl];
    if (row < 0) continue;
    for (int e = lane; e < d; e += 64) rg_acc(dE + (size_t)row * d + e, es_acc[sl * d + e]);
  }
}

// ------------------------------------------------------------------------------------------------
// LayerNorm backward from the saved OUTPUT: xhat = (y - beta) / gamma, rstd saved by the forward.
//   g = dy * rowmask * gamma ; dz = rstd * (g - mean(g) - xhat * mean(g * xhat))
//   dgamma += sum_rows dy*rowmask*xhat ; dbeta += sum_rows dy*rowmask
// nn.LayerNorm(eps=1e-8) backward at transformer.py:161,188; rows with rowmask == 0 (the
// `* pad_mask` of :594/:539) get dz = 0.
// ------------------------------------------------------------------------------------------------
// Row layout: LPR = N/8 lanes per row (8 features = one 16-byte bf16 / 32-byte f32 vector per lane),
// 64/LPR rows per wave pass; row statistics by log2(LPR) shuffles.
template <typename T, int LPR>
__global__ __launch_bounds__(EW_BLOCK) void ln_bwd_kernel(rg_ln_bwd_args a) {
  constexpr int RPW = 64 / LPR;                 // rows per wave pass
  constexpr int N = LPR * 8;
  __shared__ float red[3][4][N];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int rr = lane / LPR, c8 = (lane % LPR) * 8;
  const T* __restrict__ dy = reinterpret_cast<const T*>(a.dy);
  const T* __restrict__ y = reinterpret_cast<const T*>(a.y);
  T* __restrict__ dz = reinterpret_cast<T*>(a.dz);
  T* __restrict__ dzd = reinterpret_cast<T*>(a.dz_drop);
  const DropCfg drop = make_drop(a.drop_p, a.drop_seed);
  const float invn = 1.f / (float)N;
  float gam[8], bet[8], igam[8], dg[8], db[8];
  load8(gam, a.gamma + c8);
  load8(bet, a.beta + c8);
#pragma unroll
  for (int j = 0; j < 8; ++j) { dg[j] = 0.f; db[j] = 0.f; igam[j] = 1.f / gam[j]; }
  float dc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};        // column sums of dz (a.dz_colsum)
  const float* __restrict__ rmp = a.rowmask ? a.rowmask : a.rstd;      // any readable float array stands in
  const long long gw = (long long)blockIdx.x * 4 + wave, nw = (long long)gridDim.x * 4;
  // work unit = RPW consecutive rows: unit u of the matrix, or -- list-driven -- part (u % (16 / RPW)) of the
  // u / (16 / RPW)-th live 16-row tile (rows of unlisted tiles are neither read nor written)
  constexpr int UPT = 16 / RPW;
  const long long nunits = a.live16 ? (long long)a.live16[0] * UPT : (a.M + RPW - 1) / RPW;
  // list mode: the tile id of the NEXT iteration is fetched while this one runs (list entry -> row address -> rows is
  // two dependent latencies per iteration otherwise)
  const long long ntl = a.live16 ? (long long)a.live16[0] : 0;
  int tile_next = (a.live16 && gw < nunits) ? a.live16[1 + gw / UPT] : 0;
  for (long long u = gw; u < nunits; u += nw) {
    const int tile_cur = tile_next;
    if (a.live16) tile_next = a.live16[1 + min((u + nw) / UPT, ntl - 1)];       // unconditional (clamped)
    const long long m0 = a.live16 ? (long long)tile_cur * 16 + (u % UPT) * RPW : u * RPW;
    const long long m = m0 + rr;
    const bool live = m < a.M;
    const long long mc = live ? m : a.M - 1;
    float g[8], xh[8], d8[8], y8[8];
    float s1 = 0.f, s2 = 0.f, rm, rstd_m = 0.f;
    if (a.live16) {
      // list-driven: 9 of 10 rows of a live tile are live -- mask, dy, y and rstd are loaded together, unconditionally
      // (clamped row), instead of mask -> branch -> rows -> rstd as three dependent latencies
      const float rv = rmp[a.rowmask ? mc : 0];
      load8(d8, dy + (size_t)mc * a.ld + c8);
      load8(y8, y + (size_t)mc * a.ld + c8);
      rstd_m = a.rstd[mc];
      rm = live ? (a.rowmask ? rv : 1.f) : 0.f;
    } else {
      rm = live ? (a.rowmask ? a.rowmask[m] : 1.f) : 0.f;
      if (rm != 0.f) {                // 43 % of the rows are padding here: their dy / y are not read
        load8(d8, dy + (size_t)m * a.ld + c8);
        load8(y8, y + (size_t)m * a.ld + c8);
        rstd_m = a.rstd[m];
      }
    }
    if (rm != 0.f) {
      const float irm = 1.f / rm;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float d = d8[j] * rm;
        xh[j] = (y8[j] * irm - bet[j]) * igam[j];
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
    for (int o = 1; o < LPR; o <<= 1) { s1 += __shfl_xor(s1, o); s2 += __shfl_xor(s2, o); }
    if (live) {
      const float rstd = rm != 0.f ? rstd_m : 0.f;
      s1 *= invn;
      s2 *= invn;
      float o8[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) { o8[j] = rstd * (g[j] - s1 - xh[j] * s2); dc[j] += o8[j]; }
      store8(dz + (size_t)m * a.ld + c8, o8);
      if (dzd) {
        float k8[8];
        rg_keep8(drop, (unsigned int)m * (unsigned int)N + (unsigned int)c8, k8);
#pragma unroll
        for (int j = 0; j < 8; ++j) o8[j] *= k8[j];
        store8(dzd + (size_t)m * a.ld + c8, o8);
      }
    }
  }
  // column sums: lanes with the same c8 (RPW of them per wave) -> LDS -> atomics
#pragma unroll
  for (int o = LPR; o < 64; o <<= 1)
#pragma unroll
    for (int j = 0; j < 8; ++j) { dg[j] += __shfl_xor(dg[j], o); db[j] += __shfl_xor(db[j], o); dc[j] += __shfl_xor(dc[j], o); }
  if (rr == 0) {
#pragma unroll
    for (int j = 0; j < 8; ++j) { red[0][wave][c8 + j] = dg[j]; red[1][wave][c8 + j] = db[j]; red[2][wave][c8 + j] = dc[j]; }
  }
  __syncthreads();
  for (int n = threadIdx.x; n < N; n += EW_BLOCK) {
    const float sg = red[0][0][n] + red[0][1][n] + red[0][2][n] + red[0][3][n];
    const float sb = red[1][0][n] + red[1][1][n] + red[1][2][n] + red[1][3][n];
    const float sc = red[2][0][n] + red[2][1][n] + red[2][2][n] + red[2][3][n];
    if (a.partials) {           // two-stage column sums: [block][3][N] partials, summed by ln_bwd_reduce_kernel
      a.partials[((size_t)blockIdx.x * 3 + 0) * N + n] = sg;
      a.partials[((size_t)blockIdx.x * 3 + 1) * N + n] = sb;
      a.partials[((size_t)blockIdx.x * 3 + 2) * N + n] = sc;
    } else {
      if (a.dgamma) rg_acc(a.dgamma + n, sg);
      if (a.dbeta) rg_acc(a.dbeta + n, sb);
      if (a.dz_colsum) rg_acc(a.dz_colsum + n, sc);
    }
  }
}

// dgamma / dbeta += column sums of the per-block partials (4096 blocks x 2N same-address float atomics cost 20-40 us
// of a 130 us kernel: the memory-side atomic unit serialises adds to one row).  Block b sums rows b, b + grid, ...
__global__ __launch_bounds__(256) void ln_bwd_reduce_kernel(const float* __restrict__ partials, int nblocks, int N,
                                                            float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                            float* __restrict__ dzsum) {
  for (int c = threadIdx.x; c < 3 * N; c += 256) {
    float* dst = c < N ? dgamma : (c < 2 * N ? dbeta : dzsum);
    if (!dst) continue;
    float s0 = 0.f, s1 = 0.f;
    int b = blockIdx.x;
    for (; b + (int)gridDim.x < nblocks; b += 2 * gridDim.x) {
      s0 += partials[(size_t)b * 3 * N + c];
      s1 += partials[(size_t)(b + gridDim.x) * 3 * N + c];
    }
    if (b < nblocks) s0 += partials[(size_t)b * 3 * N + c];
    rg_acc(dst + (c % N), s0 + s1);
  }
}

// y[b,t,:] = LayerNorm(x[b,t,:] + o[b,:]) * rowmask : the collapsed decoder cross-attention (Q1):
// context = WV u + bV for every query, so MultiHeadAttention reduces to a per-sequence vector o.
template <typename T, int NPL>
__global__ __launch_bounds__(EW_BLOCK) void bcast_add_ln_kernel(const T* __restrict__ x, const float* __restrict__ o,
                                                               const float* __restrict__ gamma, const float* __restrict__ beta,
                                                               T* __restrict__ y, float* __restrict__ rstd_out, long long M,
                                                               int L, int N, float eps) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const float invn = 1.f / (float)N;
  const long long gw = (long long)blockIdx.x * 4 + wave, nw = (long long)gridDim.x * 4;
  for (long long m = gw; m < M; m += nw) {
    const long long b = m / L;
    float v[NPL];
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < NPL; ++j) {
      const int n = lane + 64 * j;
      v[j] = n < N ? (float)x[(size_t)m * N + n] + o[(size_t)b * N + n] : 0.f;
      s += v[j];
    }
    const float mean = wave_sum(s) * invn;
    float q = 0.f;
#pragma unroll
    for (int j = 0; j < NPL; ++j) {
      const int n = lane + 64 * j;
      const float dd = n < N ? v[j] - mean : 0.f;
      q += dd * dd;
    }
    const float rstd = __builtin_amdgcn_rsqf(wave_sum(q) * invn + eps);      // (argument >= eps: the bare v_rsq_f32, see fused.hip ln_regs)
#pragma unroll
    for (int j = 0; j < NPL; ++j) {
      const int n = lane + 64 * j;
      if (n < N) y[(size_t)m * N + n] = (T)((v[j] - mean) * rstd * gamma[n] + beta[n]);
    }
    if (lane == 0) rstd_out[m] = rstd;
  }
}

// the same for N == 64 * NPL exactly: a lane owns NPL CONTIGUOUS features (one 8 / 16-byte load per operand instead of NPL
// 2 / 4-byte ones; gamma and beta live in registers across the row loop)
template <typename T, int NPL>
__global__ __launch_bounds__(EW_BLOCK) void bcast_add_ln_vec_kernel(const T* __restrict__ x, const float* __restrict__ o,
                                                                   const float* __restrict__ gamma, const float* __restrict__ beta,
                                                                   T* __restrict__ y, float* __restrict__ rstd_out, long long M,
                                                                   int L, float eps) {
  constexpr int N = 64 * NPL;
  struct alignas(sizeof(T) * NPL) VT { T e[NPL]; };
  struct alignas(sizeof(float) * NPL) VF { float e[NPL]; };
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const float invn = 1.f / (float)N;
  const VF g = *reinterpret_cast<const VF*>(gamma + lane * NPL), be = *reinterpret_cast<const VF*>(beta + lane * NPL);
  const long long gw = (long long)blockIdx.x * 4 + wave, nw = (long long)gridDim.x * 4;
  for (long long m = gw; m < M; m += nw) {
    const long long b = m / L;
    const VT xv = *reinterpret_cast<const VT*>(x + (size_t)m * N + lane * NPL);
    const VF ov = *reinterpret_cast<const VF*>(o + (size_t)b * N + lane * NPL);
    float v[NPL];
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < NPL; ++j) {
      v[j] = (float)xv.e[j] + ov.e[j];
      s += v[j];
    }
    const float mean = wave_sum(s) * invn;
    float q = 0.f;
#pragma unroll
    for (int j = 0; j < NPL; ++j) {
      const float dd = v[j] - mean;
      q += dd * dd;
    }
    const float rstd = __builtin_amdgcn_rsqf(wave_sum(q) * invn + eps);      // (argument >= eps: the bare v_rsq_f32, see fused.hip ln_regs)
    VT yv;
#pragma unroll
    for (int j = 0; j < NPL; ++j) yv.e[j] = (T)((v[j] - mean) * rstd * g.e[j] + be.e[j]);
    *reinterpret_cast<VT*>(y + (size_t)m * N + lane * NPL) = yv;
    if (lane == 0) rstd_out[m] = rstd;
  }
}

// y[m,:] = LayerNorm(x[m,:] + bo + sum_h s[m,h] * oh[m / L, h, :]): rg_cross_rows and the LayerNorm row pass in one (the f32
// [M, N] matrix of per-row cross-attention outputs never reaches HBM); same accumulation order as the two launches
template <typename T, int NPL>
__global__ __launch_bounds__(EW_BLOCK) void cross_add_ln_kernel(const T* __restrict__ x, const float* __restrict__ s,
                                                               const float* __restrict__ oh, const float* __restrict__ bo,
                                                               const float* __restrict__ gamma, const float* __restrict__ beta,
                                                               T* __restrict__ y, float* __restrict__ rstd_out, long long M, int L,
                                                               int H, float eps) {
  constexpr int N = 64 * NPL;
  struct alignas(sizeof(T) * NPL) VT { T e[NPL]; };
  struct alignas(sizeof(float) * NPL) VF { float e[NPL]; };
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const float invn = 1.f / (float)N;
  const VF g = *reinterpret_cast<const VF*>(gamma + lane * NPL), be = *reinterpret_cast<const VF*>(beta + lane * NPL);
  const VF b0 = *reinterpret_cast<const VF*>(bo + lane * NPL);
  const long long gw = (long long)blockIdx.x * 4 + wave, nw = (long long)gridDim.x * 4;
  for (long long m = gw; m < M; m += nw) {
    const long long b = m / L;
    const VT xv = *reinterpret_cast<const VT*>(x + (size_t)m * N + lane * NPL);
    float acc[NPL];
#pragma unroll
    for (int j = 0; j < NPL; ++j) acc[j] = b0.e[j];
    for (int h = 0; h < H; ++h) {
      const float sh = s[m * H + h];
      const VF ov = *reinterpret_cast<const VF*>(oh + ((size_t)b * H + h) * N + lane * NPL);
#pragma unroll
      for (int j = 0; j < NPL; ++j) acc[j] += sh * ov.e[j];
    }
    float v[NPL];
    float sum = 0.f;
#pragma unroll
    for (int j = 0; j < NPL; ++j) {
      v[j] = (float)xv.e[j] + acc[j];
      sum += v[j];
    }
    const float mean = wave_sum(sum) * invn;
    float q = 0.f;
#pragma unroll
    for (int j = 0; j < NPL; ++j) {
      const float dd = v[j] - mean;
      q += dd * dd;
    }
    const float rstd = __builtin_amdgcn_rsqf(wave_sum(q) * invn + eps);
    VT yv;
#pragma unroll
    for (int j = 0; j < NPL; ++j) yv.e[j] = (T)((v[j] - mean) * rstd * g.e[j] + be.e[j]);
    *reinterpret_cast<VT*>(y + (size_t)m * N + lane * NPL) = yv;
    if (lane == 0) rstd_out[m] = rstd;
  }
}

// out[b,:] = sum_t x[b,t,:]   (f32 accumulation, tier-dtype result: it feeds the next GEMM)
template <typename T>
__global__ __launch_bounds__(EW_BLOCK) void seq_sum_kernel(const T* __restrict__ x, T* __restrict__ out, int L, int N) {
  const int b = blockIdx.x;
  for (int n = threadIdx.x; n < N; n += EW_BLOCK) {
    float s = 0.f;
    for (int t = 0; t < L; ++t) s += (float)x[((size_t)b * L + t) * N + n];
    out[(size_t)b * N + n] = (T)s;
  }
}

// out[n] += scale * sum_m coef[m] * x[m,n] * (aux ? [aux[m,n] > 0] : 1)
template <typename T>
__global__ __launch_bounds__(EW_BLOCK) void colsum_kernel(const T* __restrict__ x, const T* __restrict__ aux, const float* __restrict__ coef,
                                                         float* __restrict__ out, long long M, int N, int ld, float scale) {
  const int n = blockIdx.x * 64 + (threadIdx.x & 63);
  const int part = threadIdx.x >> 6;
  __shared__ float red[4][64];
  float s = 0.f;
  if (n < N)
    for (long long m = (long long)blockIdx.y * 4 + part; m < M; m += (long long)gridDim.y * 4) {
      float v = (float)x[(size_t)m * ld + n];
      if (aux && !((float)aux[(size_t)m * ld + n] > 0.f)) v = 0.f;
      if (coef) v *= coef[m];
      s += v;
    }
  red[part][threadIdx.x & 63] = s;
  __syncthreads();
  if (part == 0 && n < N) rg_acc(out + n, (red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x]) * scale);
}

// out[m,n] = coef[m] * w[n] * [aux[m,n] > 0]      (seed of the ReLU-mask chains, tools/utils.py:41-52)
template <typename T>
__global__ __launch_bounds__(EW_BLOCK) void outer_posmask_kernel(const float* __restrict__ coef, const float* __restrict__ w,
                                                                const T* __restrict__ aux, T* __restrict__ out, long long M, int N,
                                                                float scale) {
  const long long total = M * N;
  for (long long i = (long long)blockIdx.x * EW_BLOCK + threadIdx.x; i < total; i += (long long)gridDim.x * EW_BLOCK) {
    const long long m = i / N;
    const int n = (int)(i - m * N);
    const float c = (coef ? coef[m] : 1.f) * scale;
    out[i] = (T)(((float)aux[i] > 0.f) ? c * w[n] : 0.f);
  }
}

// xhat = alpha[b] * real + (1 - alpha[b]) * fake      gan_training.py:39-43
template <typename T>
__global__ __launch_bounds__(EW_BLOCK) void interpolate_kernel(const float* __restrict__ alpha, const T* __restrict__ real,
                                                              const T* __restrict__ fake, T* __restrict__ out, long long B, int d) {
  const long long total = B * d;
  for (long long i = (long long)blockIdx.x * EW_BLOCK + threadIdx.x; i < total; i += (long long)gridDim.x * EW_BLOCK) {
    const float al = alpha[i / d];
    out[i] = (T)(al * (float)real[i] + (1.f - al) * (float)fake[i]);
  }
}

// gp += lambda/B * sum_b (||g_b|| - 1)^2 ; dg_b = lambda * 2/B * (||g_b|| - 1)/||g_b|| * g_b
// gan_training.py:54 and its analytic derivative
template <typename T>
__global__ __launch_bounds__(EW_BLOCK) void gp_penalty_kernel(const float* __restrict__ g, T* __restrict__ dg, float* __restrict__ gp,
                                                             long long B, int d, float lambda) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long long gw = (long long)blockIdx.x * 4 + wave, nw = (long long)gridDim.x * 4;
  float acc = 0.f;
  for (long long b = gw; b < B; b += nw) {
    float s = 0.f;
    for (int e = lane; e < d; e += 64) { const float v = g[(size_t)b * d + e]; s += v * v; }
    const float nrm = sqrtf(wave_sum(s));
    const float c = lambda * 2.f / (float)B * (nrm - 1.f) / nrm;
    for (int e = lane; e < d; e += 64) dg[(size_t)b * d + e] = (T)(c * g[(size_t)b * d + e]);
    acc += (nrm - 1.f) * (nrm - 1.f);
  }
  if (lane == 0 && acc != 0.f) rg_acc(gp, acc * lambda / (float)B);
}

// out[0] += scale * sum(x)
__global__ __launch_bounds__(EW_BLOCK) void sum_kernel(const float* __restrict__ x, float* __restrict__ out, long long n, float scale) {
  float s = 0.f;
  for (long long i = (long long)blockIdx.x * EW_BLOCK + threadIdx.x; i < n; i += (long long)gridDim.x * EW_BLOCK) s += x[i];
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0 && s != 0.f) rg_acc(out, s * scale);
}

// torch.optim.Adam (no amsgrad, no weight decay), one tensor; optionally refreshes the operand-tier
// shadow copy (bf16) of the parameter in the same pass.     train_gan.py:126-134, gan_training.py:359
template <typename S>
__global__ __launch_bounds__(EW_BLOCK) void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                       float* __restrict__ v, S* __restrict__ shadow, long long n, float lr,
                                                       float b1, float b2, float eps, float bc1, float bc2_sqrt) {
  for (long long i = (long long)blockIdx.x * EW_BLOCK + threadIdx.x; i < n; i += (long long)gridDim.x * EW_BLOCK) {
    const float gi = g[i];
    const float mi = b1 * m[i] + (1.f - b1) * gi;
    const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
    m[i] = mi;
    v[i] = vi;
    const float denom = sqrtf(vi) / bc2_sqrt + eps;
    const float pi = p[i] - (lr / bc1) * (mi / denom);
    p[i] = pi;
    if (shadow) shadow[i] = (S)pi;
  }
}

// Multi-tensor form: one launch updates every (chunk of a) parameter listed in a device-resident segment table
// (222 per-tensor launches of ~5 us each per optimizer step otherwise).  Block b owns segs[b]; the per-parameter
// bias corrections (step counts may differ between parameters: torch skips parameters without gradient) ride in
// the segment.
__global__ __launch_bounds__(EW_BLOCK) void adam_multi_kernel(const rg_adam_seg* __restrict__ segs, float b1, float b2, float eps) {
  const rg_adam_seg sg = segs[blockIdx.x];
  float* __restrict__ p = sg.p;
  const float* __restrict__ g = sg.g;
  float* __restrict__ m = sg.m;
  float* __restrict__ v = sg.v;
  const long long n4 = (((uintptr_t)p | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v) & 15) ? 0 : (sg.n >> 2);
  for (long long i = threadIdx.x; i < n4; i += EW_BLOCK) {
    const float4 g4 = reinterpret_cast<const float4*>(g)[i];
    float4 m4 = reinterpret_cast<float4*>(m)[i], v4 = reinterpret_cast<float4*>(v)[i], p4 = reinterpret_cast<float4*>(p)[i];
    const float gg[4] = {g4.x, g4.y, g4.z, g4.w};
    float mm[4] = {m4.x, m4.y, m4.z, m4.w}, vv[4] = {v4.x, v4.y, v4.z, v4.w}, pp[4] = {p4.x, p4.y, p4.z, p4.w};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      mm[j] = b1 * mm[j] + (1.f - b1) * gg[j];
      vv[j] = b2 * vv[j] + (1.f - b2) * gg[j] * gg[j];
      pp[j] = pp[j] - sg.step_lr * (mm[j] / (sqrtf(vv[j]) * sg.inv_bc2_sqrt + eps));
    }
    reinterpret_cast<float4*>(m)[i] = make_float4(mm[0], mm[1], mm[2], mm[3]);
    reinterpret_cast<float4*>(v)[i] = make_float4(vv[0], vv[1], vv[2], vv[3]);
    reinterpret_cast<float4*>(p)[i] = make_float4(pp[0], pp[1], pp[2], pp[3]);
  }
  for (long long i = (n4 << 2) + threadIdx.x; i < sg.n; i += EW_BLOCK) {
    const float gi = g[i];
    const float mi = b1 * m[i] + (1.f - b1) * gi;
    const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
    m[i] = mi;
    v[i] = vi;
    p[i] = p[i] - sg.step_lr * (mi / (sqrtf(vi) * sg.inv_bc2_sqrt + eps));
  }
}

// The same with the step count living in the table (rg_adam_seg_dev): no per-step upload.  Every thread derives the two
// bias corrections from the block's step count (two pow() in double per thread -- nothing against the chunk's traffic).
__global__ __launch_bounds__(EW_BLOCK) void adam_multi_dev_kernel(rg_adam_seg_dev* __restrict__ segs, double lr, double b1d,
                                                                  double b2d, float eps) {
  const rg_adam_seg_dev sg = segs[blockIdx.x];
  const double st = (double)(sg.step + 1);
  const float step_lr = (float)(lr / (1.0 - pow(b1d, st)));
  const float inv_bc2_sqrt = (float)(1.0 / sqrt(1.0 - pow(b2d, st)));
  const float b1 = (float)b1d, b2 = (float)b2d;
  float* __restrict__ p = sg.p;
  const float* __restrict__ g = sg.g;
  float* __restrict__ m = sg.m;
  float* __restrict__ v = sg.v;
  const long long n4 = (((uintptr_t)p | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v) & 15) ? 0 : (sg.n >> 2);
  for (long long i = threadIdx.x; i < n4; i += EW_BLOCK) {
    const float4 g4 = reinterpret_cast<const float4*>(g)[i];
    float4 m4 = reinterpret_cast<float4*>(m)[i], v4 = reinterpret_cast<float4*>(v)[i], p4 = reinterpret_cast<float4*>(p)[i];
    const float gg[4] = {g4.x, g4.y, g4.z, g4.w};
    float mm[4] = {m4.x, m4.y, m4.z, m4.w}, vv[4] = {v4.x, v4.y, v4.z, v4.w}, pp[4] = {p4.x, p4.y, p4.z, p4.w};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      mm[j] = b1 * mm[j] + (1.f - b1) * gg[j];
      vv[j] = b2 * vv[j] + (1.f - b2) * gg[j] * gg[j];
      pp[j] = pp[j] - step_lr * (mm[j] / (sqrtf(vv[j]) * inv_bc2_sqrt + eps));
    }
    reinterpret_cast<float4*>(m)[i] = make_float4(mm[0], mm[1], mm[2], mm[3]);
    reinterpret_cast<float4*>(v)[i] = make_float4(vv[0], vv[1], vv[2], vv[3]);
    reinterpret_cast<float4*>(p)[i] = make_float4(pp[0], pp[1], pp[2], pp[3]);
  }
  for (long long i = (n4 << 2) + threadIdx.x; i < sg.n; i += EW_BLOCK) {
    const float gi = g[i];
    const float mi = b1 * m[i] + (1.f - b1) * gi;
    const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
    m[i] = mi;
    v[i] = vi;
    p[i] = p[i] - step_lr * (mi / (sqrtf(vi) * inv_bc2_sqrt + eps));
  }
  __syncthreads();                                               // every thread of the block has read its copy of sg
  if (threadIdx.x == 0) segs[blockIdx.x].step = sg.step + 1;
}

// dst[c,r] (or dst[r,c]) = (T) src[r,c]
// Fragment-packed destination (mode bit 1, RG_CAST_PACK): the logical operand M[n][k] (= src, or src^T with bit 0) is
// stored as consecutive MFMA A/B fragments -- block (n / 16, k / 32) is 64 lanes x 8 elements, lane 16 g + i holding
// M[16 nb + i][32 kb + 8 g .. + 8] -- so that a wave's fragment load is ONE contiguous 1 KB (bf16) read instead of 16
// rows x 64 B.  Writes the 32 x 32 source tile at (r0, c0) held in `tile` (two whole blocks).  nks = logical K / 32.
// split (mode bit 2, RG_CAST_SPLIT; f32 destination only): the bf16x3 tier's presplit operand -- the 2 KB slot of a fragment
// holds the 64 lanes' hi parts bf16(v) (1 KB), then their lo parts bf16(v - hi) (1 KB), so a kernel's fragment load is two
// contiguous 16-byte reads per lane with no conversion work behind them (fused.hip load_wset).
template <typename T>
__device__ __forceinline__ void cast_pack_tile(const float (&tile)[32][33], T* __restrict__ dst, int r0, int c0, int transpose,
                                               int n_off, int k_off, int nks, int split = 0) {
  const int n0 = (transpose ? c0 : r0) + n_off, k0 = (transpose ? r0 : c0) + k_off;
  for (int o = threadIdx.x; o < 1024; o += EW_BLOCK) {
    const int b2 = o >> 9, within = o & 511, lane = within >> 3, j = within & 7, lg = lane >> 4, li = lane & 15;
    const int nl = b2 * 16 + li, kl = lg * 8 + j;
    const float v = transpose ? tile[kl][nl] : tile[nl][kl];
    const size_t frag = (size_t)(n0 / 16 + b2) * nks + k0 / 32;
    if constexpr (sizeof(T) == 4) {
      if (split) {
        __bf16* d2 = reinterpret_cast<__bf16*>(dst) + frag * 1024;
        const __bf16 h = (__bf16)v;
        d2[within] = h;
        d2[512 + within] = (__bf16)(v - (float)h);
        continue;
      }
    }
    dst[frag * 512 + within] = (T)v;
  }
}

template <typename T>
__global__ __launch_bounds__(EW_BLOCK) void cast_kernel(const float* __restrict__ src, T* __restrict__ dst, int R, int C, int transpose) {
  if (transpose & 2) {
    __shared__ float tile[32][33];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const int tiles_c = C / 32, tiles_r = R / 32;
    for (int t = blockIdx.x; t < tiles_c * tiles_r; t += gridDim.x) {
      const int r0 = (t / tiles_c) * 32, c0 = (t % tiles_c) * 32;
      for (int k = ty; k < 32; k += 8) tile[k][tx] = src[(size_t)(r0 + k) * C + c0 + tx];
      __syncthreads();
      cast_pack_tile<T>(tile, dst, r0, c0, transpose & 1, 0, 0, ((transpose & 1) ? R : C) / 32, transpose & 4);
      __syncthreads();
    }
    return;
  }
  if (!transpose) {
    const long long total = (long long)R * C;
    for (long long i = (long long)blockIdx.x * EW_BLOCK + threadIdx.x; i < total; i += (long long)gridDim.x * EW_BLOCK) dst[i] = (T)src[i];
    return;
  }
  __shared__ float tile[32][33];
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
  const int tiles_c = (C + 31) / 32, tiles_r = (R + 31) / 32;
  for (int t = blockIdx.x; t < tiles_c * tiles_r; t += gridDim.x) {
    const int r0 = (t / tiles_c) * 32, c0 = (t % tiles_c) * 32;
    for (int k = ty; k < 32; k += 8) tile[k][tx] = (r0 + k < R && c0 + tx < C) ? src[(size_t)(r0 + k) * C + c0 + tx] : 0.f;
    __syncthreads();
    for (int k = ty; k < 32; k += 8)
      if (c0 + k < C && r0 + tx < R) dst[(size_t)(c0 + k) * R + r0 + tx] = (T)tile[tx][k];
    __syncthreads();
  }
}

// one workgroup = one 32 x 32 tile of one segment (rg_cast_seg), staged through LDS so that both orientations read and
// write 32 consecutive elements per row
template <typename T>
__global__ __launch_bounds__(EW_BLOCK) void cast_multi_kernel(const rg_cast_seg* __restrict__ segs, const int* __restrict__ tiles) {
  __shared__ float tile[32][33];
  const rg_cast_seg sg = segs[tiles[2 * blockIdx.x]];
  const int t = tiles[2 * blockIdx.x + 1];
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
  const int tiles_c = (sg.C + 31) / 32;
  const int r0 = (t / tiles_c) * 32, c0 = (t % tiles_c) * 32;
  T* __restrict__ dst = reinterpret_cast<T*>(sg.dst);
  for (int k = ty; k < 32; k += 8) tile[k][tx] = (r0 + k < sg.R && c0 + tx < sg.C) ? sg.src[(size_t)(r0 + k) * sg.C + c0 + tx] : 0.f;
  __syncthreads();
  if (sg.transpose & 2) {              // fragment-packed copy: R, C, row_off, col_off multiples of 32 / 16 (checked by the host)
    cast_pack_tile<T>(tile, dst, r0, c0, sg.transpose & 1, sg.row_off, sg.col_off, sg.ld / 32, sg.transpose & 4);
  } else if (sg.transpose) {
    for (int k = ty; k < 32; k += 8)
      if (c0 + k < sg.C && r0 + tx < sg.R) dst[(size_t)(c0 + k + sg.row_off) * sg.ld + r0 + tx + sg.col_off] = (T)tile[tx][k];
  } else {
    for (int k = ty; k < 32; k += 8)
      if (r0 + k < sg.R && c0 + tx < sg.C) dst[(size_t)(r0 + k + sg.row_off) * sg.ld + c0 + tx + sg.col_off] = (T)tile[k][tx];
  }
}

// s[b*L+q, h] = sum over live keys of keep(...) / n_live : the row sum of the dropped uniform attention
// map of the collapsed decoder cross-attention (Q1 + nn.Dropout of transformer.py:126-127).
// One workgroup per sequence: the live-key set becomes a bit vector in LDS; in the p == 0.5 mode a row
// sum is popcount(hash word & live word) over the row's L/32 words.
__global__ __launch_bounds__(EW_BLOCK) void cross_drop_scale_kernel(const int64_t* __restrict__ ids, int64_t pad, float* __restrict__ s,
                                                                   int B, int L, int H, DropCfg drop) {
  __shared__ unsigned int live[64];        // L <= 2048
  __shared__ int nlive_s;
  const int b = blockIdx.x, tid = threadIdx.x;
  const unsigned int nw = rg_lpad(L) >> 5;
  if (tid < 64) live[tid] = 0u;
  if (tid == 0) nlive_s = 0;
  __syncthreads();
  for (int j = tid; j < L; j += EW_BLOCK)
    if (ids[(size_t)b * L + j] != pad) { atomicOr(&live[j >> 5], 1u << (j & 31)); atomicAdd(&nlive_s, 1); }
  __syncthreads();
  const int n = nlive_s;
  const bool all_masked = n == 0;                        // replace-fill: uniform over all L keys
  if (all_masked) {
    __syncthreads();
    for (int j = tid; j < L; j += EW_BLOCK) atomicOr(&live[j >> 5], 1u << (j & 31));
    __syncthreads();
  }
  const float inv_n = 1.f / (float)(all_masked ? L : n);
  for (int i = tid; i < L * H; i += EW_BLOCK) {
    const int q = i / H, h = i - q * H;
    const unsigned int row = ((unsigned int)b * H + h) * L + q;                     // attention-map index space
    float acc = 0.f;
    if (drop.onebit) {
      int cnt = 0;
      for (unsigned int w = 0; w < nw; ++w) cnt += __popc(rg_hash(drop.seed, row * nw + w) & live[w]);
      acc = (float)cnt * drop.inv_keep;
    } else {
      for (int j = 0; j < L; ++j)
        if ((live[j >> 5] >> (j & 31)) & 1u) acc += rg_keep(drop, row * (nw << 5) + j);
    }
    s[((size_t)b * L + q) * H + h] = acc * inv_n;
  }
}

// out[b,h,:] = sum_q s[b*L+q, h] * x[b,q,:]     (one workgroup per sequence, x read once for all heads)
template <typename T, int MAXH>
__global__ __launch_bounds__(EW_BLOCK) void seq_wsum_kernel(const T* __restrict__ x, const float* __restrict__ s, T* __restrict__ out,
                                                           int L, int H, int N) {
  extern __shared__ float sw_smem[];       // s[b] : L*H floats, then the [rows][H][N] partials
  const int b = blockIdx.x, tid = threadIdx.x;
  const int tpr = N >> 2, rows = EW_BLOCK / tpr;       // threads per row (4 columns each), rows in flight
  float* sl = sw_smem;
  float* part = sw_smem + L * H;
  for (int i = tid; i < L * H; i += EW_BLOCK) sl[i] = s[(size_t)b * L * H + i];
  __syncthreads();
  const int c4 = (tid % tpr) * 4, r0 = tid / tpr;
  float acc[MAXH][4];
#pragma unroll
  for (int h = 0; h < MAXH; ++h)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[h][j] = 0.f;
  if (r0 < rows)
    for (int t = r0; t < L; t += rows) {
      float v[4];
      load4t(v, x + ((size_t)b * L + t) * N + c4);
#pragma unroll
      for (int h = 0; h < MAXH; ++h)
        if (h < H) {
          const float w = sl[t * H + h];
#pragma unroll
          for (int j = 0; j < 4; ++j) acc[h][j] += w * v[j];
        }
    }
  if (r0 < rows) {
#pragma unroll
    for (int h = 0; h < MAXH; ++h)
      if (h < H) {
#pragma unroll
        for (int j = 0; j < 4; ++j) part[((size_t)r0 * H + h) * N + c4 + j] = acc[h][j];
      }
  }
  __syncthreads();
  for (int i = tid; i < H * N; i += EW_BLOCK) {
    float t = 0.f;
    for (int r = 0; r < rows; ++r) t += part[(size_t)r * H * N + i];
    out[(size_t)b * H * N + i] = (T)t;
  }
}

// Live-tile list in two launches.  (1) fully parallel: one 64-bit mask per 64 tiles, bit = any(rowmask[16t..16t+15] != 0).
// (2) one workgroup: masks -> LDS (one coalesced load per thread), scan of their popcounts, then the waves write
// list[0] = number of live tiles, list[1 + i] = index of the i-th live one (ascending) and the dead ones from the far end
// backwards (list[nt - j] = j-th dead tile) with coalesced stores -- no global load inside a loop (the first version read the
// per-tile flags twice in 50-iteration loops per wave: 39 us for 51 200 tiles, 16 times per training step).
// The masks live in list[1 + nt ..] (scratch, 8-byte aligned part of it).
__global__ __launch_bounds__(EW_BLOCK) void live_flags_kernel(const float* __restrict__ rowmask, long long M, unsigned long long* __restrict__ masks) {
  const long long nt = (M + 15) >> 4, ntp = (nt + 63) & ~63ll;
  for (long long t = (long long)blockIdx.x * EW_BLOCK + threadIdx.x; t < ntp; t += (long long)gridDim.x * EW_BLOCK) {
    int live = 0;
    if (t < nt) {
      if (16 * t + 16 <= M && (reinterpret_cast<uintptr_t>(rowmask) & 15) == 0) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float4 v = reinterpret_cast<const float4*>(rowmask + 16 * t)[j];
          live |= (v.x != 0.f) | (v.y != 0.f) | (v.z != 0.f) | (v.w != 0.f);
        }
      } else {
        for (long long r = 16 * t; r < min(16 * t + 16, M); ++r) live |= rowmask[r] != 0.f;
      }
    }
    const unsigned long long m = __ballot(live != 0);          // t of lane 0 is a multiple of 64
    if ((threadIdx.x & 63) == 0) masks[t >> 6] = m;
  }
}

#define LIVE_MAXW 8192     // 64-tile words the single compaction workgroup can scan: M <= 8.4 M rows
__global__ __launch_bounds__(1024) void live_compact_kernel(long long nt, int* __restrict__ list, const unsigned long long* __restrict__ masks) {
  extern __shared__ __align__(8) unsigned char live_smem[];
  const int nwords = (int)((nt + 63) >> 6);
  unsigned long long* wm = reinterpret_cast<unsigned long long*>(live_smem);            // [nwords] masks
  int* woff = reinterpret_cast<int*>(live_smem + (size_t)nwords * 8);                   // [nwords] exclusive prefix of the popcounts
  __shared__ int part[1024];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int per = (nwords + 1023) / 1024;
  const int w0 = min(tid * per, nwords), w1 = min(w0 + per, nwords);
  int cnt = 0;
  for (int w = w0; w < w1; ++w) { const unsigned long long m = masks[w]; wm[w] = m; cnt += __popcll(m); }
  part[tid] = cnt;
  __syncthreads();
  for (int o = 1; o < 1024; o <<= 1) {                  // Hillis-Steele inclusive scan of the per-thread sums
    const int v = tid >= o ? part[tid - o] : 0;
    __syncthreads();
    part[tid] += v;
    __syncthreads();
  }
  int run = part[tid] - cnt;
  for (int w = w0; w < w1; ++w) { woff[w] = run; run += __popcll(wm[w]); }
  if (tid == 1023) list[0] = part[1023];
  __syncthreads();
  for (int w = wave; w < nwords; w += 16) {
    const long long t = 64ll * w + lane;
    const unsigned long long m = wm[w];
    const bool in = t < nt;
    const bool live = (m >> lane) & 1ull;
    const int rank = __popcll(m & ((1ull << lane) - 1ull));             // live tiles of this word before this lane
    if (live) list[1 + woff[w] + rank] = (int)t;
    else if (in) list[nt - ((64ll * w - woff[w]) + (lane - rank))] = (int)t;   // j-th dead tile goes to list[nt - j]
  }
}

// x[m, c] *= keep(m*N + c) in place: nn.Dropout for the block paths that are not fused (any width); same index
// space as the fused kernel / ln_bwd's dz_drop, so a backward regenerates the mask from the seed.
template <typename T>
__global__ __launch_bounds__(EW_BLOCK) void dropout_kernel(T* __restrict__ x, long long M, int N, DropCfg drop) {
  const int n8 = N >> 3;
  if ((N & 7) == 0) {
    const long long total = M * n8;
    for (long long i = (long long)blockIdx.x * EW_BLOCK + threadIdx.x; i < total; i += (long long)gridDim.x * EW_BLOCK) {
      const long long m = i / n8;
      const int c8 = (int)(i - m * n8) * 8;
      float v[8], k8[8];
      load8(v, x + (size_t)m * N + c8);
      rg_keep8(drop, (unsigned int)m * (unsigned int)N + (unsigned int)c8, k8);
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] *= k8[j];
      store8(x + (size_t)m * N + c8, v);
    }
  } else {
    const long long total = M * N;
    for (long long i = (long long)blockIdx.x * EW_BLOCK + threadIdx.x; i < total; i += (long long)gridDim.x * EW_BLOCK)
      x[i] = (T)((float)x[i] * rg_keep(drop, (unsigned int)i));
  }
}

// In-place dropout as above AND g = gelu(the value as stored): the activated operand of the second FFN product, for the
// weight-stationary GEMM (which has no prologue).  The weight-gradient product recomputes gelu() from the same stored x.
template <typename T, bool PRECISE>
__global__ __launch_bounds__(EW_BLOCK) void dropout_gelu_kernel(T* __restrict__ x, T* __restrict__ g, long long M, int N, DropCfg drop) {
  const int n8 = N >> 3;
  const long long total = M * n8;
  for (long long i = (long long)blockIdx.x * EW_BLOCK + threadIdx.x; i < total; i += (long long)gridDim.x * EW_BLOCK) {
    const long long m = i / n8;
    const int c8 = (int)(i - m * n8) * 8;
    float v[8];
    load8(v, x + (size_t)m * N + c8);
    if (drop.thresh) {
      float k8[8];
      rg_keep8(drop, (unsigned int)m * (unsigned int)N + (unsigned int)c8, k8);
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] *= k8[j];
      store8(x + (size_t)m * N + c8, v);
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = gelu_t<PRECISE>((float)(T)v[j]);
    store8(g + (size_t)m * N + c8, v);
  }
}

// y[m,:] = LayerNorm(x[m,:] + dropout(z)[m,:]) * rowmask[m]  (N == 64 * NPL: a lane owns NPL contiguous features).  The mask of
// z is the one rg_dropout(z, seed) would apply (index m*N + c), so rg_ln_bwd's dz_drop regenerates it; z itself is not rewritten.
template <typename T, int NPL>
__global__ __launch_bounds__(EW_BLOCK) void add_drop_ln_kernel(const T* __restrict__ x, const T* __restrict__ z,
                                                              const float* __restrict__ gamma, const float* __restrict__ beta,
                                                              const float* __restrict__ rowmask, T* __restrict__ y,
                                                              float* __restrict__ rstd_out, long long M, DropCfg drop, float eps) {
  constexpr int N = 64 * NPL;
  struct alignas(sizeof(T) * NPL) VT { T e[NPL]; };
  struct alignas(sizeof(float) * NPL) VF { float e[NPL]; };
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const float invn = 1.f / (float)N;
  const VF g = *reinterpret_cast<const VF*>(gamma + lane * NPL), be = *reinterpret_cast<const VF*>(beta + lane * NPL);
  const long long gw = (long long)blockIdx.x * 4 + wave, nw = (long long)gridDim.x * 4;
  for (long long m = gw; m < M; m += nw) {
    const VT xv = *reinterpret_cast<const VT*>(x + (size_t)m * N + lane * NPL);
    const VT zv = *reinterpret_cast<const VT*>(z + (size_t)m * N + lane * NPL);
    const float rm = rowmask ? rowmask[m] : 1.f;
    float v[NPL];
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < NPL; ++j) {
      const float k = drop.thresh ? rg_keep(drop, (unsigned int)m * (unsigned int)N + (unsigned int)(lane * NPL + j)) : 1.f;
      v[j] = (float)xv.e[j] + (float)(T)((float)zv.e[j] * k);      // the dropped addend rounded as rg_dropout stores it
      s += v[j];
    }
    const float mean = wave_sum(s) * invn;
    float q = 0.f;
#pragma unroll
    for (int j = 0; j < NPL; ++j) {
      const float dd = v[j] - mean;
      q += dd * dd;
    }
    const float rstd = __builtin_amdgcn_rsqf(wave_sum(q) * invn + eps);      // (argument >= eps: the bare v_rsq_f32, see fused.hip ln_regs)
    VT yv;
#pragma unroll
    for (int j = 0; j < NPL; ++j) yv.e[j] = (T)(((v[j] - mean) * rstd * g.e[j] + be.e[j]) * rm);
    *reinterpret_cast<VT*>(y + (size_t)m * N + lane * NPL) = yv;
    if (lane == 0) rstd_out[m] = rstd;
  }
}

// o[m, :] = bo + sum_h s[m, h] * oh[m / L, h, :]  : the collapsed decoder cross-attention output per ROW under
// attention-map dropout (what the fused kernel forms in registers), for the unfused block path.
__global__ __launch_bounds__(EW_BLOCK) void cross_rows_kernel(const float* __restrict__ s, const float* __restrict__ oh,
                                                             const float* __restrict__ bo, float* __restrict__ out, long long M, int L,
                                                             int H, int N) {
  const long long total = M * N;
  for (long long i = (long long)blockIdx.x * EW_BLOCK + threadIdx.x; i < total; i += (long long)gridDim.x * EW_BLOCK) {
    const long long m = i / N;
    const int n = (int)(i - m * N);
    float acc = bo[n];
    for (int h = 0; h < H; ++h) acc += s[m * H + h] * oh[((m / L) * H + h) * N + n];
    out[i] = acc;
  }
}

// ================================================================================================
// C ABI
// ================================================================================================
extern "C" int rg_cross_drop_scale(const int64_t* enc_ids, int64_t pad_value, float* s, int B, int L, int H, float drop_p,
                                   unsigned long long seed, void* stream) {
  if (B <= 0) return 0;
  if (L > 2048) return rg_set_error_msg(RG_ERR_UNSUPPORTED, "cross_drop_scale: L > 2048");
  hipLaunchKernelGGL(cross_drop_scale_kernel, dim3(B), dim3(EW_BLOCK), 0, (hipStream_t)stream,
                     enc_ids, pad_value, s, B, L, H, make_drop(drop_p, seed));
  RG_CHECK_LAUNCH();
  return 0;
}

extern "C" int rg_seq_wsum(const void* x, const float* s, void* out, int B, int L, int H, int N, int dtype, void* stream) {
  if (B <= 0) return 0;
  hipStream_t st = (hipStream_t)stream;
  const int tpr = N >> 2;
  if ((N & 3) || tpr < 1 || tpr > EW_BLOCK || (EW_BLOCK % tpr) || H > 8)
    return rg_set_error_msg(RG_ERR_UNSUPPORTED, "seq_wsum: needs N/4 a divisor of the block size and H <= 8");
  const size_t smem = ((size_t)L * H + (size_t)(EW_BLOCK / tpr) * H * N) * sizeof(float);
  if (smem > 64 * 1024) return rg_set_error_msg(RG_ERR_UNSUPPORTED, "seq_wsum: L*H + rows*H*N floats exceed 64 KB of LDS");
  if (dtype == RG_BF16) hipLaunchKernelGGL((seq_wsum_kernel<__bf16, 8>), dim3(B), dim3(EW_BLOCK), smem, st, (const __bf16*)x, s, (__bf16*)out, L, H, N);
  else if (dtype == RG_F32) hipLaunchKernelGGL((seq_wsum_kernel<float, 8>), dim3(B), dim3(EW_BLOCK), smem, st, (const float*)x, s, (float*)out, L, H, N);
  else return rg_set_error_msg(RG_ERR_INVALID, "seq_wsum: bad dtype");
  RG_CHECK_LAUNCH();
  return 0;
}

#define DISPATCH_T(dtype, CALL_BF16, CALL_F32, name)                                     \
  if ((dtype) == RG_BF16) { CALL_BF16; } else if ((dtype) == RG_F32) { CALL_F32; }        \
  else return rg_set_error_msg(RG_ERR_INVALID, name ": bad dtype");                       \
  RG_CHECK_LAUNCH(); return 0;

// first[b] = index of the first position of sequence b with rowmask != 0 (L if none): one wave per sequence
__global__ __launch_bounds__(EW_BLOCK) void first_live_kernel(const float* __restrict__ rowmask, int B, int L, int* __restrict__ first) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int b = blockIdx.x * (EW_BLOCK / 64) + wave; b < B; b += gridDim.x * (EW_BLOCK / 64)) {
    int f = L;
    for (int l0 = 0; l0 < L && f == L; l0 += 64) {
      const int l = l0 + lane;
      const unsigned long long m = __ballot(l < L && rowmask[(size_t)b * L + l] != 0.f);
      if (m) f = l0 + __builtin_ctzll(m);
    }
    if (lane == 0) first[b] = f;
  }
}
extern "C" int rg_first_live(const float* rowmask, int B, int L, int* first, void* stream) {
  if (B <= 0 || L <= 0) return 0;
  hipLaunchKernelGGL(first_live_kernel, dim3(ew_grid((long long)B * 64, EW_BLOCK)), dim3(EW_BLOCK), 0, (hipStream_t)stream, rowmask, B, L, first);
  RG_CHECK_LAUNCH();
  return 0;
}

extern "C" int rg_live_tiles(const float* rowmask, long long M, int* list, void* stream) {
  if (M <= 0) return 0;
  const long long nt = (M + 15) >> 4;
  const int nwords = (int)((nt + 63) >> 6);
  if (nwords > LIVE_MAXW) return rg_set_error_msg(RG_ERR_UNSUPPORTED, "live_tiles: more than 8.4 M rows");
  // masks: the scratch part of the list (nt + 4 ints), from its first 8-byte aligned int on
  unsigned long long* masks = reinterpret_cast<unsigned long long*>((reinterpret_cast<uintptr_t>(list + 1 + nt) + 7) & ~(uintptr_t)7);
  hipLaunchKernelGGL(live_flags_kernel, dim3(ew_grid(nt, EW_BLOCK)), dim3(EW_BLOCK), 0, (hipStream_t)stream, rowmask, M, masks);
  const int smem = nwords * 12;
  hipFuncSetAttribute(reinterpret_cast<const void*>(live_compact_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, smem);
  hipLaunchKernelGGL(live_compact_kernel, dim3(1), dim3(1024), smem, (hipStream_t)stream, nt, list, masks);
  RG_CHECK_LAUNCH();
  return 0;
}

// mask[i] = (ids[i] != pad) as f32: get_pad_mask (gan_training.py:347-350) and the inline copies -- one launch instead of
// a compare and a cast
__global__ __launch_bounds__(EW_BLOCK) void pad_mask_kernel(const int64_t* __restrict__ ids, int64_t pad, float* __restrict__ out, long long n) {
  for (long long i = (long long)blockIdx.x * EW_BLOCK + threadIdx.x; i < n; i += (long long)gridDim.x * EW_BLOCK)
    out[i] = ids[i] != pad ? 1.f : 0.f;
}

// x_last[b, :] = x[b, L-1, :], m_last[b] = rowmask[b * L + L - 1]: the one row per sequence the last encoder layer
// evaluates (EncoderLastLayerFn) -- one launch instead of two strided copies
template <typename T>
__global__ __launch_bounds__(EW_BLOCK) void last_rows_kernel(const T* __restrict__ x, const float* __restrict__ rowmask, T* __restrict__ x_last,
                                                            float* __restrict__ m_last, int B, int L, int d) {
  const long long total = (long long)B * d;
  for (long long i = (long long)blockIdx.x * EW_BLOCK + threadIdx.x; i < total; i += (long long)gridDim.x * EW_BLOCK) {
    const int b = (int)(i / d), c = (int)(i - (long long)b * d);
    x_last[i] = x[((size_t)b * L + L - 1) * d + c];
    if (c == 0 && m_last) m_last[b] = rowmask[(size_t)b * L + L - 1];
  }
}

extern "C" int rg_pad_mask(const int64_t* ids, int64_t pad, float* out, long long n, void* stream) {
  if (n <= 0) return 0;
  hipLaunchKernelGGL(pad_mask_kernel, dim3(ew_grid(n, EW_BLOCK)), dim3(EW_BLOCK), 0, (hipStream_t)stream, ids, pad, out, n);
  RG_CHECK_LAUNCH();
  return 0;
}

extern "C" int rg_last_rows(const void* x, const float* rowmask, void* x_last, float* m_last, int B, int L, int d, int dtype,
                            void* stream) {
  if (B <= 0 || L <= 0 || d <= 0) return 0;
  hipStream_t s = (hipStream_t)stream;
  const int grid = ew_grid((long long)B * d, EW_BLOCK);
  DISPATCH_T(dtype,
             hipLaunchKernelGGL(last_rows_kernel<__bf16>, dim3(grid), dim3(EW_BLOCK), 0, s, (const __bf16*)x, rowmask, (__bf16*)x_last, m_last, B, L, d),
             hipLaunchKernelGGL(last_rows_kernel<float>, dim3(grid), dim3(EW_BLOCK), 0, s, (const float*)x, rowmask, (float*)x_last, m_last, B, L, d),
             "last_rows")
}

extern "C" int rg_dropout(void* x, long long M, int N, float drop_p, unsigned long long seed, int dtype, void* stream) {
  if (M <= 0 || N <= 0 || drop_p <= 0.f) return 0;
  const DropCfg drop = make_drop(drop_p, seed);
  hipStream_t s = (hipStream_t)stream;
  const int grid = ew_grid(M * ((N & 7) ? N : (N >> 3)), EW_BLOCK);
  DISPATCH_T(dtype,
             hipLaunchKernelGGL(dropout_kernel<__bf16>, dim3(grid), dim3(EW_BLOCK), 0, s, (__bf16*)x, M, N, drop),
             hipLaunchKernelGGL(dropout_kernel<float>, dim3(grid), dim3(EW_BLOCK), 0, s, (float*)x, M, N, drop),
             "dropout")
}

extern "C" int rg_dropout_gelu(void* x, void* g, long long M, int N, float drop_p, unsigned long long seed, int dtype, void* stream) {
  if (M <= 0 || N <= 0) return 0;
  if (N & 7) return rg_set_error_msg(RG_ERR_UNSUPPORTED, "dropout_gelu: N % 8 != 0");
  const DropCfg drop = make_drop(drop_p, seed);
  hipStream_t s = (hipStream_t)stream;
  const int grid = ew_grid(M * (N >> 3), EW_BLOCK);
  DISPATCH_T(dtype,
             hipLaunchKernelGGL((dropout_gelu_kernel<__bf16, false>), dim3(grid), dim3(EW_BLOCK), 0, s, (__bf16*)x, (__bf16*)g, M, N, drop),
             hipLaunchKernelGGL((dropout_gelu_kernel<float, true>), dim3(grid), dim3(EW_BLOCK), 0, s, (float*)x, (float*)g, M, N, drop),
             "dropout_gelu")
}

template <typename T>
static int launch_add_drop_ln(const void* x, const void* z, const float* gamma, const float* beta, const float* rowmask, void* y,
                              float* rstd, long long M, int N, const DropCfg& drop, float eps, hipStream_t s) {
  const int grid = ew_grid(M, 16);
#define RG_ADL(NPL) hipLaunchKernelGGL((add_drop_ln_kernel<T, NPL>), dim3(grid), dim3(EW_BLOCK), 0, s, (const T*)x, (const T*)z, gamma, beta, rowmask, (T*)y, rstd, M, drop, eps)
  if (N == 128) RG_ADL(2);
  else if (N == 256) RG_ADL(4);
  else return rg_set_error_msg(RG_ERR_UNSUPPORTED, "add_drop_ln: N must be 128 or 256");
#undef RG_ADL
  RG_CHECK_LAUNCH();
  return 0;
}
extern "C" int rg_add_drop_ln(const void* x, const void* z, const float* gamma, const float* beta, const float* rowmask, void* y,
                              float* rstd, long long M, int N, float drop_p, unsigned long long seed, float eps, int dtype,
                              void* stream) {
  if (M <= 0) return 0;
  const DropCfg drop = make_drop(drop_p, seed);
  if (dtype == RG_BF16) return launch_add_drop_ln<__bf16>(x, z, gamma, beta, rowmask, y, rstd, M, N, drop, eps, (hipStream_t)stream);
  if (dtype == RG_F32) return launch_add_drop_ln<float>(x, z, gamma, beta, rowmask, y, rstd, M, N, drop, eps, (hipStream_t)stream);
  return rg_set_error_msg(RG_ERR_INVALID, "add_drop_ln: bad dtype");
}

template <typename T>
static int launch_cross_add_ln(const void* x, const float* s_, const float* oh, const float* bo, const float* gamma, const float* beta,
                               void* y, float* rstd, long long M, int L, int H, int N, float eps, hipStream_t st) {
  const int grid = ew_grid(M, 16);
#define RG_CAL(NPL) hipLaunchKernelGGL((cross_add_ln_kernel<T, NPL>), dim3(grid), dim3(EW_BLOCK), 0, st, (const T*)x, s_, oh, bo, gamma, beta, (T*)y, rstd, M, L, H, eps)
  if (N == 128) RG_CAL(2);
  else if (N == 256) RG_CAL(4);
  else return rg_set_error_msg(RG_ERR_UNSUPPORTED, "cross_add_ln: N must be 128 or 256");
#undef RG_CAL
  RG_CHECK_LAUNCH();
  return 0;
}
extern "C" int rg_cross_add_ln(const void* x, const float* s, const float* oh, const float* bo, const float* gamma, const float* beta,
                               void* y, float* rstd, long long M, int L, int H, int N, float eps, int dtype, void* stream) {
  if (M <= 0) return 0;
  if (L <= 0 || H <= 0) return rg_set_error_msg(RG_ERR_INVALID, "cross_add_ln: L and H must be positive");
  if (dtype == RG_BF16) return launch_cross_add_ln<__bf16>(x, s, oh, bo, gamma, beta, y, rstd, M, L, H, N, eps, (hipStream_t)stream);
  if (dtype == RG_F32) return launch_cross_add_ln<float>(x, s, oh, bo, gamma, beta, y, rstd, M, L, H, N, eps, (hipStream_t)stream);
  return rg_set_error_msg(RG_ERR_INVALID, "cross_add_ln: bad dtype");
}

extern "C" int rg_cross_rows(const float* s, const float* oh, const float* bo, float* out, long long M, int L, int H, int N, void* stream) {
  if (M <= 0) return 0;
  hipLaunchKernelGGL(cross_rows_kernel, dim3(ew_grid(M * N, EW_BLOCK)), dim3(EW_BLOCK), 0, (hipStream_t)stream, s, oh, bo, out, M, L, H, N);
  RG_CHECK_LAUNCH();
  return 0;
}

static int embed_pe_fwd_launch(const void* table, const float* pe, const int64_t* ids, const float* mask, void* out, void* out2,
                               long long ntok, int L, int d, const DropCfg& drop, int dtype, hipStream_t s, long long table_rows = 0) {
  // The row form is selected for the two-output call and by RG_EMBED_ROWS=1 (A/B).  Measured (round 5, bench shape, 56 % live
  // positions): 41 vs 49 us per launch when the same launch is repeated (ids / mask / table hot), 78.0 vs 77.0 us INSIDE the step --
  // both kernels sit at the memory system's rate for this read / write mix (0.55 of 8 TB/s = 0.89 of the box's measured copy rate),
  // so the simpler element-per-thread kernel stays the default.
  static const int rows_form = getenv("RG_EMBED_ROWS") ? atoi(getenv("RG_EMBED_ROWS")) : 0;
  // RG_EMBED_FORM: 2 = position-major (default since round 6), 0 = the element-per-thread kernel; RG_EMBED_NT: 1 = nontemporal stores
  static const int form = getenv("RG_EMBED_FORM") ? atoi(getenv("RG_EMBED_FORM")) : 2;
  // Store policy (measured, profiles/r06/ab/embed_forms.txt): with the bench table (25.6 MB) and its 210 MB of output rows the Infinity Cache
  // (256 MB) absorbs ordinary stores -- 68.5 us per launch in the step against 73.0 with nontemporal ones --, with the 1 GiB config-5 table it
  // cannot, and ordinary stores cost the gather its cache: 92.8 -> 78.3 us (uniform ids), 63.4 -> 48.6 (Zipf, 44 % padded).  So: nontemporal
  // stores when table + output rows exceed the Infinity Cache (table_rows = 0: unknown, the output alone decides); RG_EMBED_NT=0/1 forces.
  static const int nt_env = getenv("RG_EMBED_NT") ? atoi(getenv("RG_EMBED_NT")) : -1;
  const long long esz = dtype == RG_BF16 ? 2 : 4;
  const int use_nt = nt_env >= 0 ? nt_env : ((table_rows + ntok) * d * esz > (256ll << 20) ? 1 : 0);
  if (form == 2 && !rows_form && !out2 && (d == 128 || d == 256) && L > 0 && ntok % L == 0 && ntok < (1ll << 31) / d) {
    // waves = position blocks x sequence ranges; the ranges sized so that the launch is ~ 8 workgroups (32 waves) per CU, every wave
    // with the same number of sequences
    const int B = (int)(ntok / L), rpw = 64 / (d / 8), npb = (L + rpw - 1) / rpw;
    const long long cap = 256LL * 8 * (EW_BLOCK / 64);
    int chunks = (int)((cap + npb - 1) / npb);
    if (chunks > B) chunks = B;
    if (chunks < 1) chunks = 1;
    const int bs = (B + chunks - 1) / chunks;
    chunks = (B + bs - 1) / bs;
    const long long waves = (long long)npb * chunks;
    const int grid = (int)((waves + EW_BLOCK / 64 - 1) / (EW_BLOCK / 64));
#define RG_EMBP(T, D) do { if (use_nt) hipLaunchKernelGGL((embed_pe_fwd_pos_kernel<T, D, 4, true>), dim3(grid), dim3(EW_BLOCK), 0, s, (const T*)table, pe, ids, mask, (T*)out, B, L, bs, drop); \
                           else hipLaunchKernelGGL((embed_pe_fwd_pos_kernel<T, D, 4, false>), dim3(grid), dim3(EW_BLOCK), 0, s, (const T*)table, pe, ids, mask, (T*)out, B, L, bs, drop); } while (0)
#define RG_EMBP_T(T) do { if (d == 128) RG_EMBP(T, 128); else RG_EMBP(T, 256); } while (0)
    DISPATCH_T(dtype, RG_EMBP_T(__bf16), RG_EMBP_T(float), "embed_pe_fwd")
#undef RG_EMBP_T
#undef RG_EMBP
  }
  if ((d == 128 || d == 256) && L > 0 && ntok < (1ll << 31) / d && (rows_form || out2)) {
    // one wave per 64-token block, at most 8 workgroups (32 waves) per CU -- and every wave the SAME number of blocks: with 12 800
    // blocks on 8 192 waves a third of the waves took two blocks and the launch lasted two block times for 1.56 of work
    const long long nblk = (ntok + 63) / 64, cap = 256LL * 8 * (EW_BLOCK / 64);
    const long long iters = (nblk + cap - 1) / cap, waves = (nblk + iters - 1) / iters;
    const int grid = (int)((waves + EW_BLOCK / 64 - 1) / (EW_BLOCK / 64));
#define RG_EMB(T, D, M2) hipLaunchKernelGGL((embed_pe_fwd_rows_kernel<T, D, 4, M2>), dim3(grid), dim3(EW_BLOCK), 0, s, (const T*)table, pe, ids, mask, (T*)out, (__bf16*)out2, (int)ntok, L, drop)
#define RG_EMB_T(T) do { if (d == 128) { if (out2) RG_EMB(T, 128, true); else RG_EMB(T, 128, false); } else { if (out2) RG_EMB(T, 256, true); else RG_EMB(T, 256, false); } } while (0)
    DISPATCH_T(dtype, RG_EMB_T(__bf16), RG_EMB_T(float), "embed_pe_fwd")
#undef RG_EMB_T
#undef RG_EMB
  }
  const int grid = ew_grid(ntok * (d >> 3), EW_BLOCK);
  DISPATCH_T(dtype,
             hipLaunchKernelGGL(embed_pe_fwd_kernel<__bf16>, dim3(grid), dim3(EW_BLOCK), 0, s, (const __bf16*)table, pe, ids, mask, (__bf16*)out, ntok, L, d, drop),
             hipLaunchKernelGGL(embed_pe_fwd_kernel<float>, dim3(grid), dim3(EW_BLOCK), 0, s, (const float*)table, pe, ids, mask, (float*)out, ntok, L, d, drop),
             "embed_pe_fwd")
}

extern "C" int rg_embed_pe_fwd(const void* table, const float* pe, const int64_t* ids, const float* mask, void* out,
                               long long ntok, int L, int d, float drop_p, unsigned long long seed, int dtype, void* stream) {
  const DropCfg drop = make_drop(drop_p, seed);
  if (ntok <= 0) return 0;
  if (d & 7) return rg_set_error_msg(RG_ERR_UNSUPPORTED, "embed_pe_fwd: d must be a multiple of 8");
  return embed_pe_fwd_launch(table, pe, ids, mask, out, nullptr, ntok, L, d, drop, dtype, (hipStream_t)stream);
}

// ... told the number of table rows (the store policy of the launcher: see embed_pe_fwd_launch)
extern "C" int rg_embed_pe_fwd_rows(const void* table, long long table_rows, const float* pe, const int64_t* ids, const float* mask, void* out,
                                    long long ntok, int L, int d, float drop_p, unsigned long long seed, int dtype, void* stream) {
  if (ntok <= 0) return 0;
  if (d & 7) return rg_set_error_msg(RG_ERR_UNSUPPORTED, "embed_pe_fwd: d must be a multiple of 8");
  if (table_rows < 0) return rg_set_error_msg(RG_ERR_INVALID, "embed_pe_fwd_rows: table_rows < 0");
  return embed_pe_fwd_launch(table, pe, ids, mask, out, nullptr, ntok, L, d, make_drop(drop_p, seed), dtype, (hipStream_t)stream, table_rows);
}

// ... with a second, bf16 copy of the rows (rg_embed_pe_fwd2: the mixed tier's operand copy for the bf16 backward; out2 may be NULL)
extern "C" int rg_embed_pe_fwd2(const void* table, const float* pe, const int64_t* ids, const float* mask, void* out, void* out2,
                                long long ntok, int L, int d, float drop_p, unsigned long long seed, int dtype, void* stream) {
  const DropCfg drop = make_drop(drop_p, seed);
  if (ntok <= 0) return 0;
  if (d & 7) return rg_set_error_msg(RG_ERR_UNSUPPORTED, "embed_pe_fwd2: d must be a multiple of 8");
  if (out2 && !((d == 128 || d == 256) && L > 0 && ntok < (1ll << 31) / d))
    return rg_set_error_msg(RG_ERR_UNSUPPORTED, "embed_pe_fwd2: the second output needs d in {128, 256} and ntok * d < 2^31");
  return embed_pe_fwd_launch(table, pe, ids, mask, out, out2, ntok, L, d, drop, dtype, (hipStream_t)stream);
}

extern "C" int rg_embed_pe_fwd_split(const float* table_f32, const float* pe, const int64_t* ids, const float* mask, void* out,
                                     void* out_lo, long long ntok, int L, int d, float drop_p, unsigned long long seed,
                                     void* stream) {
  const DropCfg drop = make_drop(drop_p, seed);
  if (ntok <= 0) return 0;
  if (d & 7) return rg_set_error_msg(RG_ERR_UNSUPPORTED, "embed_pe_fwd_split: d must be a multiple of 8");
  if (!table_f32 || !out || !out_lo) return rg_set_error_msg(RG_ERR_INVALID, "embed_pe_fwd_split: table_f32, out and out_lo are required");
  const int grid = ew_grid(ntok * (d >> 3), EW_BLOCK);
  hipLaunchKernelGGL(embed_pe_fwd_split_kernel, dim3(grid), dim3(EW_BLOCK), 0, (hipStream_t)stream, table_f32, pe, ids, mask,
                     (__bf16*)out, (__bf16*)out_lo, ntok, L, d, drop);
  RG_CHECK_LAUNCH();
  return 0;
}

extern "C" int rg_embed_scatter_bwd(const void* dx, const int64_t* ids, const float* mask, float* dE, long long ntok, int d,
                                    long long skip_row, float drop_p, unsigned long long seed, int dtype, void* stream) {
  const DropCfg drop = make_drop(drop_p, seed);
  if (ntok <= 0) return 0;
  hipStream_t s = (hipStream_t)stream;
  const size_t smem = (size_t)ES_SLOTS * d * sizeof(float);
  if (d > 256) return rg_set_error_msg(RG_ERR_UNSUPPORTED, "embed_scatter_bwd: d > 256");
  int grid = ew_grid(ntok, 4 * ES_UNROLL);
  const int cap = 256 * (d <= 128 ? 4 : 2);                // persistent: few, long-lived workgroups = few flushes per hot row
  if (grid > cap) grid = cap;
#define RG_ES(T, NPL) hipLaunchKernelGGL((embed_scatter_bwd_kernel<T, NPL>), dim3(grid), dim3(EW_BLOCK), smem, s, (const T*)dx, ids, mask, dE, ntok, d, skip_row, drop)
#define RG_ES_T(T)                   \
  do {                               \
    if (d <= 64) RG_ES(T, 1);        \
    else if (d <= 128) RG_ES(T, 2);  \
    else RG_ES(T, 4);                \
  } while (0)
  DISPATCH_T(dtype, RG_ES_T(__bf16), RG_ES_T(float),
             "embed_scatter_bwd")
}

template <typename T>
static int launch_ln_bwd(const rg_ln_bwd_args& a, hipStream_t s) {
  const int grid = ew_grid(a.M, 64);
  if (a.N == 32) hipLaunchKernelGGL((ln_bwd_kernel<T, 4>), dim3(grid), dim3(EW_BLOCK), 0, s, a);
  else if (a.N == 64) hipLaunchKernelGGL((ln_bwd_kernel<T, 8>), dim3(grid), dim3(EW_BLOCK), 0, s, a);
  else if (a.N == 128) hipLaunchKernelGGL((ln_bwd_kernel<T, 16>), dim3(grid), dim3(EW_BLOCK), 0, s, a);
  else if (a.N == 256) hipLaunchKernelGGL((ln_bwd_kernel<T, 32>), dim3(grid), dim3(EW_BLOCK), 0, s, a);
  else return rg_set_error_msg(RG_ERR_UNSUPPORTED, "ln_bwd: N must be 32, 64, 128 or 256");
  if (a.partials && (a.dgamma || a.dbeta || a.dz_colsum))
    hipLaunchKernelGGL(ln_bwd_reduce_kernel, dim3(grid < 64 ? grid : 64), dim3(256), 0, s, a.partials, grid, a.N, a.dgamma, a.dbeta,
                       a.dz_colsum);
  RG_CHECK_LAUNCH();
  return 0;
}
extern "C" size_t rg_ln_bwd_workspace(long long M, int N) {
  return (size_t)ew_grid(M, 64) * 3 * N * sizeof(float);
}
extern "C" int rg_ln_bwd(const rg_ln_bwd_args* a, int dtype, void* stream) {
  if (!a || a->M <= 0) return 0;
  if (dtype == RG_BF16) return launch_ln_bwd<__bf16>(*a, (hipStream_t)stream);
  if (dtype == RG_F32) return launch_ln_bwd<float>(*a, (hipStream_t)stream);
  return rg_set_error_msg(RG_ERR_INVALID, "ln_bwd: bad dtype");
}

template <typename T>
static int launch_bcast_ln(const void* x, const float* o, const float* gamma, const float* beta, void* y, float* rstd,
                           long long M, int L, int N, float eps, hipStream_t s) {
  const int grid = ew_grid(M, 16);
#define RG_BV(NPL) hipLaunchKernelGGL((bcast_add_ln_vec_kernel<T, NPL>), dim3(grid), dim3(EW_BLOCK), 0, s, (const T*)x, o, gamma, beta, (T*)y, rstd, M, L, eps)
  if (N == 128 || N == 256) {                              // whole-wave rows of contiguous per-lane vectors
    if (N == 128) RG_BV(2); else RG_BV(4);
    RG_CHECK_LAUNCH();
    return 0;
  }
#undef RG_BV
#define RG_BL(NPL) hipLaunchKernelGGL((bcast_add_ln_kernel<T, NPL>), dim3(grid), dim3(EW_BLOCK), 0, s, (const T*)x, o, gamma, beta, (T*)y, rstd, M, L, N, eps)
  if (N <= 64) RG_BL(1);
  else if (N <= 128) RG_BL(2);
  else if (N <= 256) RG_BL(4);
  else return rg_set_error_msg(RG_ERR_UNSUPPORTED, "bcast_add_ln: N > 256");
#undef RG_BL
  RG_CHECK_LAUNCH();
  return 0;
}
extern "C" int rg_bcast_add_ln(const void* x, const float* o, const float* gamma, const float* beta, void* y, float* rstd,
                               long long M, int L, int N, float eps, int dtype, void* stream) {
  if (M <= 0) return 0;
  if (dtype == RG_BF16) return launch_bcast_ln<__bf16>(x, o, gamma, beta, y, rstd, M, L, N, eps, (hipStream_t)stream);
  if (dtype == RG_F32) return launch_bcast_ln<float>(x, o, gamma, beta, y, rstd, M, L, N, eps, (hipStream_t)stream);
  return rg_set_error_msg(RG_ERR_INVALID, "bcast_add_ln: bad dtype");
}

extern "C" int rg_seq_sum(const void* x, void* out, int B, int L, int N, int dtype, void* stream) {
  if (B <= 0) return 0;
  hipStream_t s = (hipStream_t)stream;
  DISPATCH_T(dtype,
             hipLaunchKernelGGL(seq_sum_kernel<__bf16>, dim3(B), dim3(EW_BLOCK), 0, s, (const __bf16*)x, (__bf16*)out, L, N),
             hipLaunchKernelGGL(seq_sum_kernel<float>, dim3(B), dim3(EW_BLOCK), 0, s, (const float*)x, (float*)out, L, N),
             "seq_sum")
}

extern "C" int rg_colsum(const void* x, const void* aux, const float* coef, float* out, long long M, int N, int ld, float scale,
                         int dtype, void* stream) {
  if (M <= 0) return 0;
  hipStream_t s = (hipStream_t)stream;
  long long gy = (M + 255) / 256; if (gy > 512) gy = 512; if (gy < 1) gy = 1;
  dim3 grid((N + 63) / 64, (int)gy);
  DISPATCH_T(dtype,
             hipLaunchKernelGGL(colsum_kernel<__bf16>, grid, dim3(EW_BLOCK), 0, s, (const __bf16*)x, (const __bf16*)aux, coef, out, M, N, ld, scale),
             hipLaunchKernelGGL(colsum_kernel<float>, grid, dim3(EW_BLOCK), 0, s, (const float*)x, (const float*)aux, coef, out, M, N, ld, scale),
             "colsum")
}

extern "C" int rg_outer_posmask(const float* coef, const float* w, const void* aux, void* out, long long M, int N, float scale,
                                int dtype, void* stream) {
  if (M <= 0) return 0;
  hipStream_t s = (hipStream_t)stream;
  const int grid = ew_grid(M * N, EW_BLOCK);
  DISPATCH_T(dtype,
             hipLaunchKernelGGL(outer_posmask_kernel<__bf16>, dim3(grid), dim3(EW_BLOCK), 0, s, coef, w, (const __bf16*)aux, (__bf16*)out, M, N, scale),
             hipLaunchKernelGGL(outer_posmask_kernel<float>, dim3(grid), dim3(EW_BLOCK), 0, s, coef, w, (const float*)aux, (float*)out, M, N, scale),
             "outer_posmask")
}

extern "C" int rg_interpolate(const float* alpha, const void* real, const void* fake, void* out, long long B, int d, int dtype, void* stream) {
  if (B <= 0) return 0;
  hipStream_t s = (hipStream_t)stream;
  const int grid = ew_grid(B * d, EW_BLOCK);
  DISPATCH_T(dtype,
             hipLaunchKernelGGL(interpolate_kernel<__bf16>, dim3(grid), dim3(EW_BLOCK), 0, s, alpha, (const __bf16*)real, (const __bf16*)fake, (__bf16*)out, B, d),
             hipLaunchKernelGGL(interpolate_kernel<float>, dim3(grid), dim3(EW_BLOCK), 0, s, alpha, (const float*)real, (const float*)fake, (float*)out, B, d),
             "interpolate")
}

extern "C" int rg_gp_penalty(const float* g, void* dg, float* gp, long long B, int d, float lambda, int dtype, void* stream) {
  if (B <= 0) return 0;
  hipStream_t s = (hipStream_t)stream;
  const int grid = ew_grid(B, 4);
  DISPATCH_T(dtype,
             hipLaunchKernelGGL(gp_penalty_kernel<__bf16>, dim3(grid), dim3(EW_BLOCK), 0, s, g, (__bf16*)dg, gp, B, d, lambda),
             hipLaunchKernelGGL(gp_penalty_kernel<float>, dim3(grid), dim3(EW_BLOCK), 0, s, g, (float*)dg, gp, B, d, lambda),
             "gp_penalty")
}

// x *= s[0], s on the device; every workgroup leaves at once when s[0] == 1 (the upstream gradient of a plain
// loss.backward()), so the common case costs one launch and no memory traffic.  n % 8 == 0.
template <typename T>
__global__ __launch_bounds__(EW_BLOCK) void scale_dev_kernel(T* __restrict__ x, long long n8, const float* __restrict__ sp) {
  const float s = sp[0];
  if (s == 1.f) return;
  for (long long i = (long long)blockIdx.x * EW_BLOCK + threadIdx.x; i < n8; i += (long long)gridDim.x * EW_BLOCK) {
    float v[8];
    load8(v, x + 8 * i);
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] *= s;
    store8(x + 8 * i, v);
  }
}

extern "C" int rg_scale_dev(void* x, long long n, const float* s, int dtype, void* stream) {
  if (n <= 0) return 0;
  if (n % 8 || !s) return rg_set_error_msg(RG_ERR_INVALID, "scale_dev: n must be a multiple of 8 and s non-null");
  hipStream_t st = (hipStream_t)stream;
  const int grid = ew_grid(n / 8, EW_BLOCK);
  DISPATCH_T(dtype,
             hipLaunchKernelGGL(scale_dev_kernel<__bf16>, dim3(grid), dim3(EW_BLOCK), 0, st, (__bf16*)x, n / 8, s),
             hipLaunchKernelGGL(scale_dev_kernel<float>, dim3(grid), dim3(EW_BLOCK), 0, st, (float*)x, n / 8, s),
             "scale_dev")
}

// nn.MSELoss()(a, b) of the overlapped-user term (gan_training.py:28-35,:494-507) with its gradient in the same pass:
// out[0] += scale * sum (a - b)^2 / n ;  da = 2 scale (a - b) / n = -db  (for an upstream gradient of 1).  n % 8 == 0.
template <typename T>
__global__ __launch_bounds__(EW_BLOCK) void mse_kernel(const T* __restrict__ a, const T* __restrict__ b, float* __restrict__ out,
                                                     T* __restrict__ da, T* __restrict__ db, long long n8, float inv_n) {
  float acc = 0.f;
  for (long long i = (long long)blockIdx.x * EW_BLOCK + threadIdx.x; i < n8; i += (long long)gridDim.x * EW_BLOCK) {
    float x[8], y[8], g[8];
    load8(x, a + 8 * i);
    load8(y, b + 8 * i);
#pragma unroll
    for (int j = 0; j < 8; ++j) { const float d = x[j] - y[j]; acc += d * d; g[j] = 2.f * inv_n * d; }
    if (da) store8(da + 8 * i, g);
    if (db) {
#pragma unroll
      for (int j = 0; j < 8; ++j) g[j] = -g[j];
      store8(db + 8 * i, g);
    }
  }
  acc = wave_sum(acc);
  __shared__ float red[EW_BLOCK / 64];
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    float t = 0.f;
    for (int w = 0; w < EW_BLOCK / 64; ++w) t += red[w];
    rg_acc(out, t * inv_n);
  }
}

extern "C" int rg_mse(const void* a, const void* b, float* out, void* da, void* db, long long n, int dtype, void* stream) {
  if (n <= 0) return 0;
  if (n % 8 || !a || !b || !out) return rg_set_error_msg(RG_ERR_INVALID, "mse: n must be a multiple of 8; a, b and out non-null");
  hipStream_t st = (hipStream_t)stream;
  int grid = ew_grid(n / 8, EW_BLOCK);
  if (grid > 1024) grid = 1024;
  const float inv_n = 1.f / (float)n;
  DISPATCH_T(dtype,
             hipLaunchKernelGGL(mse_kernel<__bf16>, dim3(grid), dim3(EW_BLOCK), 0, st, (const __bf16*)a, (const __bf16*)b, out, (__bf16*)da, (__bf16*)db, n / 8, inv_n),
             hipLaunchKernelGGL(mse_kernel<float>, dim3(grid), dim3(EW_BLOCK), 0, st, (const float*)a, (const float*)b, out, (float*)da, (float*)db, n / 8, inv_n),
             "mse")
}

extern "C" int rg_sum(const float* x, float* out, long long n, float scale, void* stream) {
  if (n <= 0) return 0;
  hipLaunchKernelGGL(sum_kernel, dim3(ew_grid(n, EW_BLOCK * 4)), dim3(EW_BLOCK), 0, (hipStream_t)stream, x, out, n, scale);
  RG_CHECK_LAUNCH();
  return 0;
}

extern "C" int rg_adam(float* p, const float* g, float* m, float* v, void* shadow, int shadow_dtype, long long n, float lr,
                       float beta1, float beta2, float eps, int step, void* stream) {
  if (n <= 0) return 0;
  if (step < 1) return rg_set_error_msg(RG_ERR_INVALID, "adam: step counts from 1");
  const float bc1 = 1.f - powf(beta1, (float)step);
  const float bc2s = sqrtf(1.f - powf(beta2, (float)step));
  hipStream_t s = (hipStream_t)stream;
  const int grid = ew_grid(n, EW_BLOCK);
  if (shadow && shadow_dtype == RG_BF16)
    hipLaunchKernelGGL(adam_kernel<__bf16>, dim3(grid), dim3(EW_BLOCK), 0, s, p, g, m, v, (__bf16*)shadow, n, lr, beta1, beta2, eps, bc1, bc2s);
  else
    hipLaunchKernelGGL(adam_kernel<float>, dim3(grid), dim3(EW_BLOCK), 0, s, p, g, m, v, (float*)nullptr, n, lr, beta1, beta2, eps, bc1, bc2s);
  RG_CHECK_LAUNCH();
  return 0;
}

extern "C" int rg_adam_multi(const rg_adam_seg* segs_device, int nsegs, float beta1, float beta2, float eps, void* stream) {
  if (nsegs <= 0) return 0;
  if (!segs_device) return rg_set_error_msg(RG_ERR_INVALID, "adam_multi: null segment table");
  hipLaunchKernelGGL(adam_multi_kernel, dim3(nsegs), dim3(EW_BLOCK), 0, (hipStream_t)stream, segs_device, beta1, beta2, eps);
  RG_CHECK_LAUNCH();
  return 0;
}

extern "C" int rg_adam_multi_dev(rg_adam_seg_dev* segs_device, int nsegs, double lr, double beta1, double beta2, double eps,
                                 void* stream) {
  if (nsegs <= 0) return 0;
  if (!segs_device) return rg_set_error_msg(RG_ERR_INVALID, "adam_multi_dev: null segment table");
  hipLaunchKernelGGL(adam_multi_dev_kernel, dim3(nsegs), dim3(EW_BLOCK), 0, (hipStream_t)stream, segs_device, lr, beta1, beta2,
                     (float)eps);
  RG_CHECK_LAUNCH();
  return 0;
}

extern "C" int rg_cast_multi(const rg_cast_seg* segs, const int* tiles, int ntiles, int dtype, void* stream) {
  if (ntiles <= 0) return 0;
  if (!segs || !tiles) return rg_set_error_msg(RG_ERR_INVALID, "cast_multi: null table");
  hipStream_t s = (hipStream_t)stream;
  DISPATCH_T(dtype,
             hipLaunchKernelGGL(cast_multi_kernel<__bf16>, dim3(ntiles), dim3(EW_BLOCK), 0, s, segs, tiles),
             hipLaunchKernelGGL(cast_multi_kernel<float>, dim3(ntiles), dim3(EW_BLOCK), 0, s, segs, tiles),
             "cast_multi")
}

extern "C" int rg_cast(const float* src, void* dst, int R, int C, int transpose, int dtype, void* stream) {
  if (R <= 0 || C <= 0) return 0;
  if ((transpose & 2) && ((R & 31) || (C & 31)))
    return rg_set_error_msg(RG_ERR_UNSUPPORTED, "cast: the fragment-packed layout needs R and C multiples of 32");
  if ((transpose & 4) && (!(transpose & 2) || dtype != RG_F32))
    return rg_set_error_msg(RG_ERR_UNSUPPORTED, "cast: RG_CAST_SPLIT is a form of the fragment-packed f32 copy (RG_CAST_PACK, RG_F32)");
  hipStream_t s = (hipStream_t)stream;
  const int grid = transpose ? min(((C + 31) / 32) * ((R + 31) / 32), 4096) : ew_grid((long long)R * C, EW_BLOCK);
  DISPATCH_T(dtype,
             hipLaunchKernelGGL(cast_kernel<__bf16>, dim3(grid), dim3(EW_BLOCK), 0, s, src, (__bf16*)dst, R, C, transpose),
             hipLaunchKernelGGL(cast_kernel<float>, dim3(grid), dim3(EW_BLOCK), 0, s, src, (float*)dst, R, C, transpose),
             "cast")
}
