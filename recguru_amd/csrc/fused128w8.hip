// PROTOTYPE (round 5): the d_model = 128 fused post-attention block on a WIDER machine per tile -- eight waves on a
// 64-token tile, each wave owning 16 output columns of every 128-wide product (half the weight and accumulator registers per wave of
// fused.hip's four-wave form), two such workgroups per CU = FOUR waves per SIMD instead of two.  Encoder INFERENCE launches only (no
// saves, no cross stage: 30 of the 48 launches of a step); selected by RG_PA8=1 for A/B timing against post_attn_fwd_kernel<bf16>
// (tools/kb_post_attn.py; DESIGN.md 6a has the measured outcome).  Same arithmetic and rounding points as the four-wave kernel:
//   y = LayerNorm(ctx.Wo^T + bo + x);  h1 = y.W1^T + b1;  g = gelu(dropout(h1));  out = LayerNorm(dropout(g.W2^T + b2) + y) * rowmask
// -- the K loop of every product runs in the same order; only the LayerNorm statistics are combined from eight partial sums of 16
// features instead of four of 32, so outputs agree with the four-wave form's to the last bf16 bit or two
// (tests/test_fused256_gpu.py::test_eight_wave_prototype_matches_the_four_wave_kernel).
// What changes per wave: 1 feature tile (16 columns) x 4 row tiles of accumulators (16 VGPRs), 4 weight fragments per GEMM step
// (16 VGPRs); what it costs: every wave still reads ALL activation fragments of the tile (B operand), so the LDS read traffic per
// tile doubles (8 waves x 16 KB per GEMM step).
#include <stdlib.h>
#include <type_traits>
#include "rg_common.hip.h"
#include "../../include/recguru_hip.h"

namespace {

constexpr int D8 = 128;
constexpr int NW8 = 8;
constexpr int NT8 = 512;
constexpr int TM8 = 64;
constexpr int RT8 = 4;
typedef __bf16 T;

__device__ __forceinline__ int soff8(int row, int col) { return row * D8 + ((((col >> 3) ^ row) & 15) << 3) + (col & 7); }

struct WSet8 { Frag<T> f[4]; };

template <typename U> __device__ __forceinline__ U* gofs8(U* base, unsigned int elem) {
  return reinterpret_cast<U*>(reinterpret_cast<char*>(base) + (size_t)(elem * (unsigned int)sizeof(U)));
}
template <typename U> __device__ __forceinline__ const U* gofs8(const U* base, unsigned int elem) {
  return reinterpret_cast<const U*>(reinterpret_cast<const char*>(base) + (size_t)(elem * (unsigned int)sizeof(U)));
}

// 4 fragments (4 k-steps of ONE 16-row feature tile) of the fragment-packed weight W (logical [N][K], K = ldk): rows row0 .. row0 + 15,
// k0 .. k0 + 127
__device__ __forceinline__ void load_w8(WSet8& w, const T* __restrict__ W, int ldk, int row0, int k0, int li, int lg) {
  const unsigned int nks = (unsigned int)ldk >> 5;
  const T* base = W + ((unsigned int)(row0 >> 4) * nks + (unsigned int)(k0 >> 5)) * 512u;
  const unsigned int lofs = (unsigned int)(lg * 16 + li) * 8u;
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) load_frag(w.f[ks], gofs8(base + ks * 512u, lofs));
}

__device__ __forceinline__ void mma_w8(f32x4 (&acc)[RT8], const WSet8& w, const T* __restrict__ tile, int li, int lg) {
  Frag<T> af[2][RT8];
#pragma unroll
  for (int rt = 0; rt < RT8; ++rt) load_frag(af[0][rt], tile + soff8(rt * 16 + li, 8 * lg));
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) {
    if (ks < 3) {
#pragma unroll
      for (int rt = 0; rt < RT8; ++rt) load_frag(af[(ks + 1) & 1][rt], tile + soff8(rt * 16 + li, (ks + 1) * 32 + 8 * lg));
    }
#pragma unroll
    for (int rt = 0; rt < RT8; ++rt) mma(w.f[ks], af[ks & 1][rt], acc[rt]);
  }
}

__device__ __forceinline__ void init8(f32x4 (&acc)[RT8], const float* __restrict__ bias_lds, int n0, int lg) {
  float b[4];
  load4f(b, bias_lds + n0 + 4 * lg);
#pragma unroll
  for (int rt = 0; rt < RT8; ++rt) acc[rt] = (f32x4){b[0], b[1], b[2], b[3]};
}

// LayerNorm over 128 features, eight partial (sum, M2) pairs per row (16 features each); no __restrict__ on the exchange pointers
// (fused256.hip: hipcc moves stores through a noalias pointer past the asm barrier)
__device__ __forceinline__ void ln8(f32x4 (&v)[RT8], float (&rstd)[RT8], const float* gamma_lds, const float* beta_lds, float* redA,
                                    float* redB, float eps, int n0, int wave, int li, int lg) {
  float s[RT8], m2[RT8];
#pragma unroll
  for (int rt = 0; rt < RT8; ++rt) {
    float t = (v[rt][0] + v[rt][1]) + (v[rt][2] + v[rt][3]);
    t += __shfl_xor(t, 16);
    t += __shfl_xor(t, 32);
    s[rt] = t;
    const float mw = t * (1.f / 16.f);
    float q = 0.f;
#pragma unroll
    for (int r = 0; r < 4; ++r) { const float dd = v[rt][r] - mw; q += dd * dd; }
    q += __shfl_xor(q, 16);
    q += __shfl_xor(q, 32);
    m2[rt] = q;
  }
#pragma unroll
  for (int rt = 0; rt < RT8; ++rt) { redA[(rt * 16 + li) * NW8 + wave] = s[rt]; redB[(rt * 16 + li) * NW8 + wave] = m2[rt]; }
  lds_barrier();
  float mean[RT8];
#pragma unroll
  for (int rt = 0; rt < RT8; ++rt) {
    float p[8], q[8];
    load8(p, redA + (rt * 16 + li) * NW8);
    load8(q, redB + (rt * 16 + li) * NW8);
    float sm = 0.f, M2 = 0.f;
#pragma unroll
    for (int w = 0; w < NW8; ++w) { sm += p[w]; M2 += q[w]; }
    const float mu = sm * (1.f / D8);
#pragma unroll
    for (int w = 0; w < NW8; ++w) { const float dm = p[w] * (1.f / 16.f) - mu; M2 += 16.f * dm * dm; }
    mean[rt] = mu;
    rstd[rt] = __builtin_amdgcn_rsqf(M2 * (1.f / D8) + eps);
  }
  float g[4], b[4];
  load4f(g, gamma_lds + n0 + 4 * lg);
  load4f(b, beta_lds + n0 + 4 * lg);
#pragma unroll
  for (int rt = 0; rt < RT8; ++rt)
#pragma unroll
    for (int r = 0; r < 4; ++r) v[rt][r] = (v[rt][r] - mean[rt]) * rstd[rt] * g[r] + b[r];
}

__device__ __forceinline__ void regs_to_tile8(const f32x4 (&v)[RT8], T* __restrict__ tile, int n0, int li, int lg) {
#pragma unroll
  for (int rt = 0; rt < RT8; ++rt) {
    float t[4] = {v[rt][0], v[rt][1], v[rt][2], v[rt][3]};
    store4(tile + soff8(rt * 16 + li, n0 + 4 * lg), t);
  }
}
__device__ __forceinline__ void add_tile8(f32x4 (&acc)[RT8], const T* __restrict__ tile, int n0, int li, int lg) {
#pragma unroll
  for (int rt = 0; rt < RT8; ++rt) {
    float r4[4];
    load4t(r4, tile + soff8(rt * 16 + li, n0 + 4 * lg));
#pragma unroll
    for (int r = 0; r < 4; ++r) acc[rt][r] += r4[r];
  }
}

// DM: dropout mode (0 none, 1 p == 0.5, 2 generic p)
template <int DM>
__global__ __launch_bounds__(NT8, 4) void post_attn_fwd_w8_kernel(rg_post_attn_args a) {
  constexpr int ACT = TM8 * D8;
  extern __shared__ __align__(16) unsigned char smem8[];
  T* Actx = reinterpret_cast<T*>(smem8);
  T* Ag = Actx + ACT;
  T* Ay = Ag + ACT;
  float* prm = reinterpret_cast<float*>(Ay + ACT);                  // 6 x 128 + dff floats
  float* redA = prm + 6 * D8 + a.dff;
  float* redB = redA + TM8 * NW8;
  float* klut = redB + TM8 * NW8;                                   // [16][4]
  float *p_bo = prm, *p_g1 = prm + D8, *p_be1 = prm + 2 * D8, *p_b2 = prm + 3 * D8, *p_g2 = prm + 4 * D8, *p_be2 = prm + 5 * D8,
        *p_b1 = prm + 6 * D8;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 15, lg = lane >> 4;
  const T* __restrict__ ctx = reinterpret_cast<const T*>(a.ctx);
  const T* __restrict__ x = reinterpret_cast<const T*>(a.x);
  const T* __restrict__ Wo = reinterpret_cast<const T*>(a.Wo);
  const T* __restrict__ W1 = reinterpret_cast<const T*>(a.W1);
  const T* __restrict__ W2 = reinterpret_cast<const T*>(a.W2);
  T* __restrict__ out = reinterpret_cast<T*>(a.out);
  const int n0 = wave * 16;                 // this wave's 16 output features of every 128-wide block
  const int ntiles = (a.M + TM8 - 1) / TM8;
  const int nchunk = a.dff / D8;
  DropCfg drop1 = make_drop(a.drop_p, a.seed_h1), drop2 = make_drop(a.drop_p, a.seed_out);
  if constexpr (DM == 2) { drop1.onebit = 0u; drop2.onebit = 0u; }
  for (int i = tid; i < D8; i += NT8) { p_bo[i] = a.bo[i]; p_g1[i] = a.g1[i]; p_be1[i] = a.be1[i]; p_b2[i] = a.b2[i]; p_g2[i] = a.g2[i]; p_be2[i] = a.be2[i]; }
  for (int i = tid; i < a.dff; i += NT8) p_b1[i] = a.b1[i];
  if (tid < 64) klut[tid] = (((tid >> 2) >> (tid & 3)) & 1) ? drop1.inv_keep : 0.f;

  WSet8 wp, wq;                             // wp: Wo / W1 chunks, wq: W2 chunks
  Frag<T> cpre[2], xpre[2];                 // 64 rows x 16 chunks = 1024 chunks per tensor: two per thread

  LiveWalk lw;
  lw.init(a.live16, a.M);
  const int nwork = a.live16 ? (lw.nlive + RT8 - 1) / RT8 : ntiles;
  int cur = (int)blockIdx.x, kcur = 0;
  auto next_group = [&](int (&g)[RT8]) -> bool {
    if (cur >= nwork) {
#pragma unroll
      for (int rt = 0; rt < RT8; ++rt) g[rt] = a.M;
      return false;
    }
    if (!a.live16) {
#pragma unroll
      for (int rt = 0; rt < RT8; ++rt) g[rt] = cur * TM8 + 16 * rt;
    } else {
      lw.template group_n<RT8>(kcur, g, a.M);
    }
    cur += gridDim.x;
    ++kcur;
    return true;
  };
  // thread tid stages chunk (row 32 i + (tid >> 4), columns 8 (tid & 15) ..), i < 2: row tile 2 i + (tid >> 8), row (tid >> 4) & 15 of it
  auto prefetch_rows = [&](const int (&g)[RT8]) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int rtl = 2 * i + (tid >> 8), c8 = (tid & 15) * 8;
      const int m = min((rtl == 0 ? g[0] : rtl == 1 ? g[1] : rtl == 2 ? g[2] : g[3]) + ((tid >> 4) & 15), a.M - 1);
      load_frag(cpre[i], gofs8(ctx, (unsigned int)(m * D8 + c8)));
      load_frag(xpre[i], gofs8(x, (unsigned int)(m * D8 + c8)));
    }
  };
  // dropout of this lane's 4 accumulator elements of every row tile: element (row, col0 + 4 lg + r) of a [M x ncol] index space
  auto drop_acc = [&](f32x4 (&v)[RT8], const DropCfg& dc, const int (&mrow)[RT8], unsigned int ncol, unsigned int col0) {
#pragma unroll
    for (int rt = 0; rt < RT8; ++rt) {
      const unsigned int rb = (unsigned int)(mrow[rt] + li) * ncol + col0 + 4u * (unsigned int)lg;     // col0 % 16 == 0
      float k4[4];
      if constexpr (DM == 1) {
        const unsigned int w = rg_hash(dc.seed, rb >> 5) >> (rb & 31u);
#pragma unroll
        for (int r = 0; r < 4; ++r) k4[r] = rg_bit(dc, w, r);
      } else {
        rg_keep4(dc, rb, k4);
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) v[rt][r] *= k4[r];
    }
  };

  int mb[RT8], mbn[RT8];
  bool have = next_group(mb);
  if (have) {
    load_w8(wp, Wo, D8, n0, 0, li, lg);
    prefetch_rows(mb);
  }
  for (; have;) {
    const bool have_next = next_group(mbn);
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int r = 32 * i + (tid >> 4), c8 = (tid & 15) * 8;
      *reinterpret_cast<Frag<T>*>(Actx + soff8(r, c8)) = cpre[i];
      *reinterpret_cast<Frag<T>*>(Ag + soff8(r, c8)) = xpre[i];
    }
    float rm4[RT8];
#pragma unroll
    for (int rt = 0; rt < RT8; ++rt) {
      const int m = mb[rt] + li;
      rm4[rt] = (a.rowmask && m < a.M) ? a.rowmask[m] : 1.f;
    }
    lds_barrier();
    f32x4 acc[RT8];
    init8(acc, p_bo, n0, lg);
    mma_w8(acc, wp, Actx, li, lg);
    load_w8(wp, W1, D8, n0, 0, li, lg);                 // FFN chunk 0 (hidden behind LayerNorm 1)
    add_tile8(acc, Ag, n0, li, lg);
    float rstd[RT8];
    ln8(acc, rstd, p_g1, p_be1, redA, redB, a.eps, n0, wave, li, lg);
    regs_to_tile8(acc, Ay, n0, li, lg);
    lds_barrier();                                      // y tile complete (and x no longer needed in Ag)
    f32x4 acc2[RT8];
    init8(acc2, p_b2, n0, lg);
#pragma unroll 1
    for (int ch = 0; ch < nchunk; ++ch) {
      load_w8(wq, W2, a.dff, n0, ch * D8, li, lg);
      init8(acc, p_b1 + ch * D8, n0, lg);
      mma_w8(acc, wp, Ay, li, lg);
      load_w8(wp, (ch + 1 < nchunk) ? W1 : Wo, D8, (ch + 1 < nchunk) ? (ch + 1) * D8 + n0 : n0, 0, li, lg);
      if (ch > 0) lds_barrier();
      if constexpr (DM != 0) drop_acc(acc, drop1, mb, (unsigned int)a.dff, (unsigned int)(ch * D8 + n0));
#pragma unroll
      for (int rt = 0; rt < RT8; ++rt)
#pragma unroll
        for (int r = 0; r < 4; r += 2) {
          const f32x2 gg = gelu2_fast((f32x2){acc[rt][r], acc[rt][r + 1]});
          acc[rt][r] = gg.x;
          acc[rt][r + 1] = gg.y;
        }
      regs_to_tile8(acc, Ag, n0, li, lg);
      lds_barrier();
      mma_w8(acc2, wq, Ag, li, lg);
    }
    prefetch_rows(mbn);
    if constexpr (DM != 0) drop_acc(acc2, drop2, mb, (unsigned int)D8, (unsigned int)n0);
    add_tile8(acc2, Ay, n0, li, lg);
    ln8(acc2, rstd, p_g2, p_be2, redA, redB, a.eps, n0, wave, li, lg);
#pragma unroll
    for (int rt = 0; rt < RT8; ++rt)
#pragma unroll
      for (int r = 0; r < 4; ++r) acc2[rt][r] *= rm4[rt];
    regs_to_tile8(acc2, Actx, n0, li, lg);
    lds_barrier();
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int r = 32 * i + (tid >> 4), c8 = (tid & 15) * 8, rtl = 2 * i + (tid >> 8);
      const int m = (rtl == 0 ? mb[0] : rtl == 1 ? mb[1] : rtl == 2 ? mb[2] : mb[3]) + ((tid >> 4) & 15);
      if (m < a.M) *reinterpret_cast<Frag<T>*>(gofs8(out, (unsigned int)m * (unsigned int)D8 + (unsigned int)c8)) = *reinterpret_cast<const Frag<T>*>(Actx + soff8(r, c8));
    }
    lds_barrier();
#pragma unroll
    for (int rt = 0; rt < RT8; ++rt) mb[rt] = mbn[rt];
    have = have_next;
  }
  if (a.live16) {           // rows of the padded tiles: out = 0
    const int nrt = (a.M + 15) >> 4, ndead = nrt - a.live16[0];
    Frag<T> z;
    frag_zero(z);
    for (int j = 2 * (int)blockIdx.x; j < ndead; j += 2 * (int)gridDim.x) {      // 512 threads: two 16-row tiles per step
      const int jj = j + (tid >> 8);
      if (jj < ndead) {
        const int m = a.live16[nrt - jj] * 16 + ((tid >> 4) & 15);
        if (m < a.M) *reinterpret_cast<Frag<T>*>(gofs8(out, (unsigned int)m * (unsigned int)D8 + (unsigned int)((tid & 15) * 8))) = z;
      }
    }
  }
}

}  // namespace

// 1 when the eight-wave prototype takes this launch (bf16, d == P == 128, packed weights, encoder inference: no saves, no cross stage,
// no split residual stream) and RG_PA8=1 asks for it; it then launches and sets *rc
int rg_post_attn_fwd_w8_try(const rg_post_attn_args* a, int dtype, hipStream_t s, int* rc) {
  static const int on = [] { const char* e = getenv("RG_PA8"); return e ? atoi(e) : 0; }();
  if (!on || dtype != RG_BF16 || a->d != D8 || a->P != D8 || !a->w_packed || (a->dff % D8) != 0) return 0;
  if (a->y_save || a->y2_save || a->h1_save || a->rstd1 || a->rstd2 || a->rstd_c || a->o_bcast || a->cross_s || a->x_lo || a->out_lo) return 0;
  if ((long long)a->M * a->dff * 2 >= (1ll << 32)) return 0;
  const int ntiles = (a->M + TM8 - 1) / TM8;
  const int smem = 3 * TM8 * D8 * 2 + (6 * D8 + a->dff) * 4 + 2 * TM8 * NW8 * 4 + 64 * 4;
  int grid = 256 * 2;
  if (grid > ntiles) grid = ntiles;
  const int dm = a->drop_p <= 0.f ? 0 : (a->drop_p == 0.5f ? 1 : 2);
#define RG_W8(DM)                                                                                                       \
  do {                                                                                                                  \
    hipFuncSetAttribute(reinterpret_cast<const void*>(post_attn_fwd_w8_kernel<DM>), hipFuncAttributeMaxDynamicSharedMemorySize, smem); \
    hipLaunchKernelGGL((post_attn_fwd_w8_kernel<DM>), dim3(grid), dim3(NT8), smem, s, *a);                              \
  } while (0)
  if (dm == 0) RG_W8(0);
  else if (dm == 1) RG_W8(1);
  else RG_W8(2);
#undef RG_W8
  hipError_t e = hipGetLastError();
  *rc = e == hipSuccess ? 0 : rg_set_error(e, "rg_post_attn_fwd (eight-wave prototype)");
  return 1;
}
