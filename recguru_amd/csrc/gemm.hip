// Generic MFMA GEMM kernels for the token-parallel parts of the path.
//
//  rg_gemm_nt : C[M,N] = epi( pro(A[M,K]) . W[N,K]^T + bias )     (Linear forward, dX, GP chains)
//  rg_gemm_tn : dW[N1,N2] += sum_t Y[t,n1] . pro(X[t,n2])          (weight gradients; + column sums)
//
// NT: one workgroup = 4 waves = a 64-row x (64*NTW)-column tile.  The activation chunk [64 x 128]
// is staged in LDS (padded rows -> conflict-free ds_read_b128 fragments); the four waves split the
// columns, so each weight element is fetched exactly once per workgroup, straight from L2 into its
// B fragment (torch Linear weights are [out,in] = [N][K]: 8 consecutive k are 16 contiguous bytes).
// The residual+LayerNorm epilogue needs whole rows and therefore a tile that spans N.
#include <stdio.h>
#include "rg_common.hip.h"
#include "rg_det.hip.h"
#include "../../include/recguru_hip.h"

#define TM 64
#define KC 128
#define APAD 8

// FULLK: K is a multiple of 128 -- every chunk is whole, so the prefetch of the next chunk is straight-line code (with
// run-time chunk lengths each guarded load got a vmcnt(0) at its join and the "prefetch" ran one load at a time)
template <typename T, int NTW, bool FULLK>
__global__ __launch_bounds__(256, 2) void gemm_nt_kernel(rg_gemm_nt_args a) {
  constexpr int TN = 64 * NTW;
  constexpr int LDA = KC + APAD;
  constexpr int A_BYTES = TM * LDA * (int)sizeof(T);
  constexpr int Z_LD = TN + 4;
  constexpr int Z_BYTES = TM * Z_LD * 4;
  constexpr int A2 = sizeof(T) == 2 ? 2 * A_BYTES : A_BYTES;     // bf16: two A tiles (one barrier per chunk)
  constexpr int SMEM = A2 > Z_BYTES ? A2 : Z_BYTES;             // A tiles and the f32 epilogue tile alias
  __shared__ __align__(16) unsigned char smem[SMEM];
  T* As = reinterpret_cast<T*>(smem);

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 15, lg = lane >> 4;
  const int m0 = blockIdx.x * TM;
  const int nb0 = blockIdx.y * TN + wave * NTW * 16;
  const T* __restrict__ A = reinterpret_cast<const T*>(a.A);
  const T* __restrict__ W = reinterpret_cast<const T*>(a.W);

  f32x4 acc[4][NTW];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < NTW; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // K is walked in 128-wide chunks, software-pipelined one chunk deep: while chunk c is in its MFMA phase the weight
  // fragments and the A rows of chunk c+1 are already on their way into a second set of registers (the un-pipelined
  // loop paid one exposed L2 / HBM latency per chunk: 10 of them for K = 1280, most of the kernel at M = 4096).
  const int nk = (a.K + KC - 1) / KC;
  auto load_chunk = [&](Frag<T> (&w)[4][NTW], Frag<T> (&pre)[4], int ci) {
    const int kc = ci * KC, klen = FULLK ? KC : min(KC, a.K - kc), ksteps = klen >> 5, cpr = klen >> 3;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks)
#pragma unroll
      for (int j = 0; j < NTW; ++j) {
        const int n = min(nb0 + j * 16 + li, a.N - 1);           // clamped: columns >= N are never stored
        if constexpr (FULLK) load_frag(w[ks][j], W + (size_t)n * a.ldw + kc + ks * 32 + 8 * lg);
        else if (ks < ksteps && !(a.debug_ablate & 8)) load_frag(w[ks][j], W + (size_t)n * a.ldw + kc + ks * 32 + 8 * lg);
        else frag_zero(w[ks][j]);
      }
#pragma unroll
    for (int i = 0; i < 4; ++i) {                                 // 64 rows x (klen / 8) 16-byte chunks over 256 threads
      const int c = tid + 256 * i;
      if (FULLK || c < TM * cpr) {
        const int r = c / cpr, c8 = (c - r * cpr) * 8;
        load_frag(pre[i], A + (size_t)min(m0 + r, a.M - 1) * a.lda + kc + c8);   // rows >= M: clamped, never stored
      }
    }
  };
  auto stage_chunk = [&](const Frag<T> (&pre)[4], int ci, T* As) {       // registers -> LDS, prologue applied on the way
    const int klen = FULLK ? KC : min(KC, a.K - ci * KC), cpr = klen >> 3;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int c = tid + 256 * i;
      if (FULLK || c < TM * cpr) {
        const int r = c / cpr, c8 = (c - r * cpr) * 8;
        if (a.prologue == RG_PRO_NONE) {
          *reinterpret_cast<Frag<T>*>(As + r * LDA + c8) = pre[i];
        } else {
          float v[8];
#pragma unroll
          for (int j = 0; j < 8; ++j) v[j] = gelu_t<Precise<T>::value>((float)pre[i].v[j]);
          store8(As + r * LDA + c8, v);
        }
      }
    }
  };
  auto mma_chunk = [&](const Frag<T> (&w)[4][NTW], int ci, const T* As) {
    const int ksteps = FULLK ? 4 : min(KC, a.K - ci * KC) >> 5;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      if (FULLK || (ks < ksteps && !(a.debug_ablate & 2))) {
#pragma unroll
        for (int rt = 0; rt < 4; ++rt) {
          Frag<T> af;
          load_frag(af, As + (rt * 16 + li) * LDA + ks * 32 + 8 * lg);
#pragma unroll
          for (int j = 0; j < NTW; ++j) mma(af, w[ks][j], acc[rt][j]);
        }
      }
    }
  };
  Frag<T> w0[4][NTW], p0[4];
  if constexpr (sizeof(T) == 4 || NTW >= 4) {     // f32 parity tier: fragments are twice as wide, one register set only; bf16 with
                                                  // a 256-column tile (N = 256 LayerNorm epilogues of d_model = 256): two weight sets
                                                  // of 16 fragments + 64 accumulators spilled 60 VGPRs per lane
    for (int ci = 0; ci < nk; ++ci) {
      load_chunk(w0, p0, ci);
      stage_chunk(p0, ci, As);
      __syncthreads();
      mma_chunk(w0, ci, As);
      __syncthreads();
    }
  } else {
  Frag<T> w1[4][NTW], p1[4];
  T* As1 = reinterpret_cast<T*>(smem + A_BYTES);
  load_chunk(w0, p0, 0);
  for (int ci = 0; ci < nk; ci += 2) {
    // the prefetch is UNCONDITIONAL (past the end it re-reads the last chunk): under a condition the compiler cannot
    // count the loads in flight at the join and waits for all of them before the MFMAs
    // two A tiles in LDS: chunk c+1 is staged into the other tile while the MFMAs of chunk c are in flight, one
    // barrier per chunk
    if (ci == 0) {
      stage_chunk(p0, 0, As);
      __syncthreads();
    }
    load_chunk(w1, p1, min(ci + 1, nk - 1));
    mma_chunk(w0, ci, As);
    stage_chunk(p1, min(ci + 1, nk - 1), As1);
    __syncthreads();
    if (ci + 1 >= nk) break;
    load_chunk(w0, p0, min(ci + 2, nk - 1));
    mma_chunk(w1, ci + 1, As1);
    stage_chunk(p0, min(ci + 2, nk - 1), As);
    __syncthreads();
  }
  }

  // ------------------------------------------------------------------ epilogue
  // The accumulator tile (+bias) goes through LDS as f32 so that every global access of the
  // epilogue (aux reads, C stores) is a 16-byte vector with consecutive lanes on consecutive bytes.
  const T* __restrict__ aux = reinterpret_cast<const T*>(a.aux);
  float* Z = reinterpret_cast<float*>(smem);
  const int ncol0 = blockIdx.y * TN;
  // aux vectors of this thread's epilogue iterations: all loaded here, ahead of the Z staging (inside the epilogue loop
  // each one was a load -> wait -> compute -> store round trip)
  constexpr int NEI = TM * (TN / 8) / 256;
  const bool vec_io = ((a.N & 7) == 0) && ((a.ldc & 7) == 0) && (a.aux == nullptr || (a.ldaux & 7) == 0);
  const bool aux_vec = vec_io && aux != nullptr && a.epilogue != RG_EPI_RESID_LN && a.epilogue != RG_EPI_RELU && a.epilogue != RG_EPI_NONE;
  Frag<T> axp[NEI];
  if (aux_vec) {
#pragma unroll
    for (int i = 0; i < NEI; ++i) {
      const int c = tid + 256 * i, row = c / (TN / 8), c8 = (c - row * (TN / 8)) * 8;
      const int m = min(m0 + row, a.M - 1), n = min(ncol0 + c8, a.N - 8);
      load_frag(axp[i], aux + (size_t)m * a.ldaux + n);
    }
  }
#pragma unroll
  for (int rt = 0; rt < 4; ++rt)
#pragma unroll
    for (int j = 0; j < NTW; ++j) {
      const int n = nb0 + j * 16 + li;
      const float b = (a.bias && n < a.N) ? a.bias[n] : 0.f;
#pragma unroll
      for (int r = 0; r < 4; ++r) Z[(rt * 16 + 4 * lg + r) * Z_LD + (n - ncol0)] = acc[rt][j][r] + b;
    }
  __syncthreads();
  if (a.epilogue != RG_EPI_RESID_LN) {
    constexpr int CPR = TN / 8;
    const bool vec = vec_io;
#pragma unroll
    for (int ei = 0; ei < NEI; ++ei) {
      const int c = tid + 256 * ei;
      const int row = c / CPR, c8 = (c - row * CPR) * 8;
      const int m = m0 + row, n = ncol0 + c8;
      if (m >= a.M || n >= a.N) continue;
      if ((a.debug_ablate & 4) && c != 0) continue;
      float v[8];
      load8(v, Z + row * Z_LD + c8);
      if (vec) {
        if (a.epilogue == RG_EPI_RELU) {
#pragma unroll
          for (int j = 0; j < 8; ++j) v[j] = fmaxf(v[j], 0.f);
          if (a.drop_p > 0.f) {
            const DropCfg dc = make_drop(a.drop_p, a.drop_seed);
            float k8[8];
            rg_keep8(dc, (unsigned int)m * (unsigned int)a.N + (unsigned int)n, k8);   // N % 8 == 0 here
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] *= k8[j];
          }
        } else if (a.epilogue != RG_EPI_NONE) {
          float x[8];
#pragma unroll
          for (int j = 0; j < 8; ++j) x[j] = (float)axp[ei].v[j];
          // uniform switch outside the element loops (inside, each element is its own basic block)
          if (a.epilogue == RG_EPI_MUL_POSMASK) {
            const float sc = a.epi_scale > 0.f ? a.epi_scale : 1.f;
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = x[j] > 0.f ? v[j] * sc : 0.f;
          } else if (a.epilogue == RG_EPI_GELU_GRAD) {
            const float nz = a.epi_nonzero_scale;
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] *= gelu_grad_t<Precise<T>::value>(x[j]);
            if (nz > 0.f) {
#pragma unroll
              for (int j = 0; j < 8; ++j) v[j] = x[j] != 0.f ? v[j] * nz : 0.f;
            }
          } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] += x[j];
          }
        }
        if (a.c_is_f32) store8(reinterpret_cast<float*>(a.C) + (size_t)m * a.ldc + n, v);
        else store8(reinterpret_cast<T*>(a.C) + (size_t)m * a.ldc + n, v);
      } else {
        for (int j = 0; j < 8 && n + j < a.N; ++j) {
          float y = v[j];
          switch (a.epilogue) {
            case RG_EPI_RELU:
              y = fmaxf(y, 0.f);
              if (a.drop_p > 0.f) y *= rg_keep(make_drop(a.drop_p, a.drop_seed), (unsigned int)m * (unsigned int)a.N + (unsigned int)(n + j));
              break;
            case RG_EPI_MUL_POSMASK:
              y = ((float)aux[(size_t)m * a.ldaux + n + j] > 0.f) ? y * (a.epi_scale > 0.f ? a.epi_scale : 1.f) : 0.f;
              break;
            case RG_EPI_GELU_GRAD: y *= gelu_grad_f((float)aux[(size_t)m * a.ldaux + n + j]); break;
            case RG_EPI_ADD: y += (float)aux[(size_t)m * a.ldaux + n + j]; break;
            default: break;
          }
          if (a.c_is_f32) reinterpret_cast<float*>(a.C)[(size_t)m * a.ldc + n + j] = y;
          else reinterpret_cast<T*>(a.C)[(size_t)m * a.ldc + n + j] = (T)y;
        }
      }
    }
    return;
  }
  // residual + LayerNorm(eps) [* rowmask]: whole rows live in this workgroup (gridDim.y == 1)
  const float invn = 1.f / (float)a.N;
  for (int i = 0; i < 16; ++i) {
    const int row = wave * 16 + i;
    const int m = m0 + row;
    if (m >= a.M) break;
    float x[NTW];
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < NTW; ++j) {
      const int n = lane + 64 * j;
      x[j] = (n < a.N) ? Z[row * Z_LD + n] + (float)aux[(size_t)m * a.ldaux + n] : 0.f;
      s += x[j];
    }
    const float mean = wave_sum(s) * invn;
    float q = 0.f;
#pragma unroll
    for (int j = 0; j < NTW; ++j) {
      const int n = lane + 64 * j;
      const float d = (n < a.N) ? x[j] - mean : 0.f;
      q += d * d;
    }
    const float rstd = __builtin_amdgcn_rsqf(wave_sum(q) * invn + a.ln_eps);   // (argument >= eps: the bare v_rsq_f32, see fused.hip ln_regs)
    const float rm = a.rowmask ? a.rowmask[m] : 1.f;
#pragma unroll
    for (int j = 0; j < NTW; ++j) {
      const int n = lane + 64 * j;
      if (n < a.N) {
        const float y = ((x[j] - mean) * rstd * a.gamma[n] + a.beta[n]) * rm;
        if (a.c_is_f32) reinterpret_cast<float*>(a.C)[(size_t)m * a.ldc + n] = y;
        else reinterpret_cast<T*>(a.C)[(size_t)m * a.ldc + n] = (T)y;
      }
    }
    if (lane == 0 && a.rstd_out) a.rstd_out[m] = rstd;
  }
}

static int nt_ntw(const rg_gemm_nt_args& a) {
  return (a.epilogue == RG_EPI_RESID_LN) ? (a.N <= 64 ? 1 : (a.N <= 128 ? 2 : 4)) : (a.N <= 64 ? 1 : 2);
}

template <typename T>
static int launch_nt(const rg_gemm_nt_args& a, hipStream_t s) {
  const int ntw = nt_ntw(a);
  const int tn = 64 * ntw;
  dim3 grid((a.M + TM - 1) / TM, (a.N + tn - 1) / tn);
  if (a.epilogue == RG_EPI_RESID_LN && grid.y != 1)
    return rg_set_error_msg(RG_ERR_UNSUPPORTED, "gemm_nt: RESID_LN needs N <= 256");
  const bool fullk = (a.K % KC) == 0 && !(a.debug_ablate & 15);
#define RG_NT(NTW_)                                                                              \
  do {                                                                                           \
    if (fullk) hipLaunchKernelGGL((gemm_nt_kernel<T, NTW_, true>), grid, dim3(256), 0, s, a);   \
    else hipLaunchKernelGGL((gemm_nt_kernel<T, NTW_, false>), grid, dim3(256), 0, s, a);        \
  } while (0)
  if (ntw == 1) RG_NT(1);
  else if (ntw == 2) RG_NT(2);
  else RG_NT(4);
#undef RG_NT
  RG_CHECK_LAUNCH();
  return 0;
}

int rg_gemm_ws_try(const rg_gemm_nt_args* a, int dtype, hipStream_t s);   // gemm_ws.hip
int rg_gemm_ws_select(const rg_gemm_nt_args* a, int dtype);

// Name of the kernel rg_gemm_nt() launches for these arguments (no launch): lets a profiler attribute time to
// the persistent weight-stationary kernels and the generic tile kernel separately.
extern "C" int rg_gemm_nt_plan(const rg_gemm_nt_args* a, int dtype, char* name, int cap) {
  if (!a || !name || cap <= 0) return rg_set_error_msg(RG_ERR_INVALID, "gemm_nt_plan: null argument");
  const int ws = (a->debug_ablate & 16) ? 0 : rg_gemm_ws_select(a, dtype);
  if (ws) snprintf(name, cap, "gemm_ws_kernel<%d,%d>", ws / 10, ws % 10);
  else snprintf(name, cap, "gemm_nt_kernel<%s,%d>", dtype == RG_BF16 ? "bf16" : (dtype == RG_X3 ? "x3" : "f32"), nt_ntw(*a));
  return 0;
}

extern "C" int rg_gemm_nt(const rg_gemm_nt_args* a, int dtype, void* stream) {
  if (!a || a->M <= 0 || a->N <= 0) return rg_set_error_msg(RG_ERR_INVALID, "gemm_nt: empty problem");
  if (a->K <= 0 || (a->K & 31)) return rg_set_error_msg(RG_ERR_UNSUPPORTED, "gemm_nt: K must be a multiple of 32");
  if ((a->lda & 7) || (a->ldw & 7)) return rg_set_error_msg(RG_ERR_UNSUPPORTED, "gemm_nt: lda/ldw must be multiples of 8");
  hipStream_t s = (hipStream_t)stream;
  if (!(a->debug_ablate & 16)) {             // bit 16: force the generic kernel (tools/kbench.py A/B)
    const int rc = rg_gemm_ws_try(a, dtype, s);
    if (rc <= 0) return rc;
  }
  if (a->w_packed)
    return rg_set_error_msg(RG_ERR_UNSUPPORTED, "gemm_nt: w_packed (presplit fragment-packed W) is a form of the bf16x3 weight-stationary kernel at K > 128");
  // the generic tile kernel reads EVERY row: a live-tile list means the padded tiles' rows of A / aux may never have
  // been written by their producers (rg_ln_bwd, skip_dead_fill), so falling through silently would compute on garbage
  if (a->epilogue == RG_EPI_DROP_GELU)
    return rg_set_error_msg(RG_ERR_UNSUPPORTED, "gemm_nt: RG_EPI_DROP_GELU is a form of the weight-stationary kernel (bf16, K and N "
                            "multiples of 128, M >= 4096, C2 given, no aux / list / head-major output)");
  if (a->c_hm_L > 0)
    return rg_set_error_msg(RG_ERR_UNSUPPORTED, "gemm_nt: c_hm_L (head-major output) is a form of the weight-stationary kernel "
                            "(bf16, K and N multiples of 128 up to 512, M >= 4096, M % c_hm_L == 0)");
  if (a->live16)
    return rg_set_error_msg(RG_ERR_UNSUPPORTED, "gemm_nt: live16 given but the list-driven kernel does not take this "
                            "problem (bf16, K and N multiples of 128 up to 512, M >= 4096 required)");
  if (dtype == RG_BF16) return launch_nt<__bf16>(*a, s);
  if (dtype == RG_F32) return launch_nt<float>(*a, s);
  if (dtype == RG_X3) return launch_nt<x3>(*a, s);
  return rg_set_error_msg(RG_ERR_INVALID, "gemm_nt: bad dtype");
}

// ------------------------------------------------------------------------------------------------
// TN (weight gradient).  Tile 64 (n1) x 64 (n2); tokens split over gridDim.z; f32 atomics at the
// end (one 256-byte-shaped add per accumulator register).  Row-major token chunks sit in LDS and
// the token-major fragments are read with the gfx950 transposing LDS read (bf16) or strided
// ds_read_b32 (f32 tier).  Column sums (bias gradients) come from one extra MFMA against a ones
// fragment in the n2-tile-0 workgroups.
// ------------------------------------------------------------------------------------------------
#define TT 64
#define TPAD 8

// fragment whose 8 slots (g, j) are rows t0 + 8g + j of column c of a row-major LDS tile
__device__ __forceinline__ void load_frag_tr(Frag<float>& f, const float* tile, int ld, int t0, int c0,
                                             int li, int lg, int use_tr) {
#pragma unroll
  for (int j = 0; j < 8; ++j) f.v[j] = tile[(t0 + 8 * lg + j) * ld + c0 + li];
}
__device__ __forceinline__ void load_frag_tr(Frag<x3>& f, const x3* tile, int ld, int t0, int c0,
                                             int li, int lg, int use_tr) {
#pragma unroll
  for (int j = 0; j < 8; ++j) f.v[j] = tile[(t0 + 8 * lg + j) * ld + c0 + li];
}
__device__ __forceinline__ void load_frag_tr(Frag<__bf16>& f, const __bf16* tile, int ld, int t0, int c0,
                                             int li, int lg, int use_tr) {
  if (use_tr) {
    // lane 4q+p of each 16-lane group addresses row q, columns 4p..4p+3 of a 4x16 block and
    // receives column (lane&15) of the 4 rows.
    typedef __attribute__((ext_vector_type(4))) short s16x4;
    typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
    const int q = li >> 2, p = li & 3;
    const __bf16* p0 = tile + (t0 + 8 * lg + q) * ld + c0 + 4 * p;
    const __bf16* p1 = p0 + 4 * ld;
    s16x4 r0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)p0);
    s16x4 r1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)p1);
    union { s16x4 s; bf16x4_t b; } u0, u1;
    u0.s = r0; u1.s = r1;
#pragma unroll
    for (int j = 0; j < 4; ++j) { f.v[j] = u0.b[j]; f.v[4 + j] = u1.b[j]; }
  } else {
#pragma unroll
    for (int j = 0; j < 8; ++j) f.v[j] = tile[(t0 + 8 * lg + j) * ld + c0 + li];
  }
}

template <typename T>
__global__ __launch_bounds__(256, 2) void gemm_tn_kernel(rg_gemm_tn_args a) {
  constexpr int LD = 64 + TPAD;
  __shared__ __align__(16) T Ys[TT * LD];
  __shared__ __align__(16) T Xs[TT * LD];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 15, lg = lane >> 4;
  const int n1_0 = blockIdx.x * 64, n2_0 = blockIdx.y * 64;
  const T* __restrict__ Y = reinterpret_cast<const T*>(a.Y);
  const T* __restrict__ X = reinterpret_cast<const T*>(a.X);
  const int per = (a.T + gridDim.z - 1) / gridDim.z;
  const int per64 = ((per + TT - 1) / TT) * TT;
  const int t_beg = blockIdx.z * per64;
  const int t_end = min(a.T, t_beg + per64);
  const bool do_colsum = (a.colsum != nullptr) && blockIdx.y == 0;

  f32x4 acc[4];
  f32x4 csum = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int j = 0; j < 4; ++j) acc[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  Frag<T> ones;
  frag_fill(ones, 1.f);

  // Token chunks are software-pipelined one deep: the rows of chunk c+1 are in flight (raw 16/32-byte loads, clamped
  // addresses, unconditional) while chunk c is in its MFMA phase; rows past the range become zeros at the LDS store,
  // columns past N1 / N2 only feed accumulators that are never written back.
  Frag<T> py[2], px[2];
  auto load_rows = [&](int t0) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int c = tid + 256 * i, r = c >> 3, c8 = (c & 7) * 8;
      const int t = min(t0 + r, a.T - 1);
      load_frag(py[i], Y + (size_t)t * a.ldy + min(n1_0 + c8, a.N1 - 8));
      load_frag(px[i], X + (size_t)t * a.ldx + min(n2_0 + c8, a.N2 - 8));
    }
  };
  if (t_beg < t_end) load_rows(t_beg);
  for (int t0 = t_beg; t0 < t_end; t0 += TT) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int c = tid + 256 * i, r = c >> 3, c8 = (c & 7) * 8;
      Frag<T> yv = py[i], xv = px[i];
      if (a.prologue_x == RG_PRO_GELU) {
        float w[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) w[j] = gelu_t<Precise<T>::value>((float)px[i].v[j]);
#pragma unroll
        for (int j = 0; j < 8; ++j) xv.v[j] = (T)w[j];
      }
      if (t0 + r >= t_end) { frag_zero(yv); frag_zero(xv); }
      *reinterpret_cast<Frag<T>*>(Ys + r * LD + c8) = yv;
      *reinterpret_cast<Frag<T>*>(Xs + r * LD + c8) = xv;
    }
    __syncthreads();
    load_rows(min(t0 + TT, a.T - 1));               // next chunk (past the end: a harmless re-read)
#pragma unroll
    for (int ks = 0; ks < TT / 32; ++ks) {
      Frag<T> af;
      load_frag_tr(af, Ys, LD, ks * 32, wave * 16, li, lg, a.use_tr);
      if (do_colsum) {
        if (a.colsum_T > 0) {              // slot (g, j) of the ones fragment is row t0 + ks * 32 + 8 g + j
#pragma unroll
          for (int j = 0; j < 8; ++j) ones.v[j] = (T)((t0 + ks * 32 + 8 * lg + j < a.colsum_T) ? 1.f : 0.f);
        }
        mma(af, ones, csum);
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        Frag<T> bf;
        load_frag_tr(bf, Xs, LD, ks * 32, j * 16, li, lg, a.use_tr);
        mma(af, bf, acc[j]);
      }
    }
    __syncthreads();
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int n2 = n2_0 + j * 16 + li;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int n1 = n1_0 + wave * 16 + 4 * lg + r;
      if (n1 < a.N1 && n2 < a.N2) rg_acc(a.dW + (size_t)n1 * a.lddw + n2, acc[j][r] * a.scale);
    }
  }
  if (do_colsum && li == 0) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int n1 = n1_0 + wave * 16 + 4 * lg + r;
      if (n1 < a.N1) rg_acc(a.colsum + n1, csum[r] * a.scale);
    }
  }
}

int rg_gemm_tn_big_try(const rg_gemm_tn_args* a, int dtype, hipStream_t s);   // gemm_tn_big.hip
int rg_gemm_tn_big_select(const rg_gemm_tn_args* a, int dtype);

size_t rg_gemm_tn_big_workspace(const rg_gemm_tn_args* a, int dtype);
const char* rg_gemm_tn_big_name(const rg_gemm_tn_args* a);

extern "C" size_t rg_gemm_tn_workspace(const rg_gemm_tn_args* a, int dtype) {
  if (!a || a->splits != 0) return 0;
  return rg_gemm_tn_big_workspace(a, dtype);
}

extern "C" int rg_gemm_tn_plan(const rg_gemm_tn_args* a, int dtype, char* name, int cap) {
  if (!a || !name || cap <= 0) return rg_set_error_msg(RG_ERR_INVALID, "gemm_tn_plan: null argument");
  if (a->splits == 0 && rg_gemm_tn_big_select(a, dtype))
    snprintf(name, cap, "%s<%s%d,%d>", dtype == RG_X3 ? "gemm_tn_big_kernel" : rg_gemm_tn_big_name(a), dtype == RG_X3 ? "x3," : "", a->N1, a->N2);
  else snprintf(name, cap, "gemm_tn_kernel<%s>", dtype == RG_BF16 ? "bf16" : (dtype == RG_X3 ? "x3" : "f32"));
  return 0;
}

extern "C" int rg_gemm_tn(const rg_gemm_tn_args* a, int dtype, void* stream) {
  if (!a || a->T <= 0 || a->N1 <= 0 || a->N2 <= 0) return rg_set_error_msg(RG_ERR_INVALID, "gemm_tn: empty problem");
  if ((a->N1 & 7) || (a->N2 & 7) || (a->ldy & 7) || (a->ldx & 7))
    return rg_set_error_msg(RG_ERR_UNSUPPORTED, "gemm_tn: N1/N2/ld must be multiples of 8");
  hipStream_t s = (hipStream_t)stream;
  if (a->splits == 0) {                        // an explicit split count forces the generic kernel
    const int rc = rg_gemm_tn_big_try(a, dtype, s);
    if (rc <= 0) return rc;
  }
  // same contract as rg_gemm_nt: the generic kernel sums every row, so it must not be handed a list
  if (a->live16)
    return rg_set_error_msg(RG_ERR_UNSUPPORTED, "gemm_tn: live16 given but the list-driven kernel does not take this "
                            "problem (bf16, T >= 8192, splits == 0, supported N1 x N2 required)");
  const int g1 = (a->N1 + 63) / 64, g2 = (a->N2 + 63) / 64;
  int splits = a->splits;
  if (splits <= 0) {
    const int chunks = (a->T + TT - 1) / TT;
    splits = max(1, min(chunks, 2048 / (g1 * g2)));
  }
  dim3 grid(g1, g2, splits);
  if (dtype == RG_BF16) hipLaunchKernelGGL((gemm_tn_kernel<__bf16>), grid, dim3(256), 0, s, *a);
  else if (dtype == RG_F32) hipLaunchKernelGGL((gemm_tn_kernel<float>), grid, dim3(256), 0, s, *a);
  else if (dtype == RG_X3) hipLaunchKernelGGL((gemm_tn_kernel<x3>), grid, dim3(256), 0, s, *a);
  else return rg_set_error_msg(RG_ERR_INVALID, "gemm_tn: bad dtype");
  RG_CHECK_LAUNCH();
  return 0;
}
