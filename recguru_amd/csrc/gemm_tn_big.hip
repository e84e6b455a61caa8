// Weight-gradient GEMM for the hot shapes (bf16 tier):  dW[N1,N2] += sum_t Y[t,n1] * pro(X[t,n2])
// with (N1,N2) in {(512,128), (128,512), (384,128), (128,128)} -- FFN l1 / l2, fused QKV, attention
// out-projection -- plus the bias gradient colsum[n1] += sum_t Y[t,n1].
//
// One workgroup (8 waves) owns the WHOLE dW tile in its accumulators (<= 128 VGPRs per lane) and
// walks a contiguous token range in 32-token chunks, so Y and X are read from HBM exactly once in
// total (the 64x64-tile kernel in gemm.hip re-reads X N1/64 times and Y N2/64 times).  Chunks are
// staged row-major in LDS (raw 16-byte copies, next chunk prefetched in registers) and the
// token-major MFMA fragments come from the transposing LDS read.  The waves split the larger of
// N1 / N2.  Partial results are combined with f32 atomics (256 workgroups x |dW|, a few MB).
#include "rg_common.cuh"
#include "../../include/recguru_hip.h"

#define TB_T 32

// fragment whose 8 slots (g, j) are rows 8g + j of column c0 + (lane & 15) of a row-major LDS tile
__device__ __forceinline__ void frag_tr16(Frag<__bf16>& f, const __bf16* tile, int ld, int c0, int li, int lg) {
  typedef __attribute__((ext_vector_type(4))) short s16x4;
  typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
  const int q = li >> 2, p = li & 3;
  const __bf16* p0 = tile + (8 * lg + q) * ld + c0 + 4 * p;
  const __bf16* p1 = p0 + 4 * ld;
  union { s16x4 s; bf16x4_t b; } u0, u1;
  u0.s = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)p0);
  u1.s = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)p1);
#pragma unroll
  for (int j = 0; j < 4; ++j) { f.v[j] = u0.b[j]; f.v[4 + j] = u1.b[j]; }
}

template <int N1, int N2, bool GELU_X>
__global__ __launch_bounds__(512) void gemm_tn_big_kernel(rg_gemm_tn_args a) {
  typedef __bf16 T;
  constexpr bool SPLIT1 = N1 >= N2;
  constexpr int MT = SPLIT1 ? N1 / 128 : N1 / 16;      // n1 tiles per wave
  constexpr int NT = SPLIT1 ? N2 / 16 : N2 / 128;      // n2 tiles per wave
  constexpr int LDY = N1 + 8, LDX = N2 + 8;
  constexpr int CY = TB_T * N1 / 8, CX = TB_T * N2 / 8;          // 16-byte chunks per staged tile
  constexpr int PY = (CY + 511) / 512, PX = (CX + 511) / 512;    // per-thread prefetch registers
  __shared__ __align__(16) T Ys[TB_T * LDY];
  __shared__ __align__(16) T Xs[TB_T * LDX];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 15, lg = lane >> 4;
  const T* __restrict__ Y = reinterpret_cast<const T*>(a.Y);
  const T* __restrict__ X = reinterpret_cast<const T*>(a.X);
  const int m1 = SPLIT1 ? wave * MT * 16 : 0;
  const int m2 = SPLIT1 ? 0 : wave * NT * 16;
  // a.live16 (optional): the list of live 16-row tiles (rg_live_tiles) -- a chunk is then 2 consecutive LIST entries
  // instead of 32 consecutive tokens; rows of padded tiles carry zero upstream gradient and are never read
  const int nchunks = a.live16 ? (a.live16[0] + 1) >> 1 : (a.T + TB_T - 1) / TB_T;
  const int per = (nchunks + gridDim.x - 1) / gridDim.x;
  const int c_beg = blockIdx.x * per, c_end = min(nchunks, c_beg + per);
  const bool do_cs = a.colsum != nullptr && (SPLIT1 || wave == 0);

  f32x4 acc[MT][NT];
  f32x4 cs[MT];
#pragma unroll
  for (int i = 0; i < MT; ++i) {
    cs[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < NT; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  }
  Frag<T> ones;
  frag_fill(ones, 1.f);
  Frag<T> py[PY], px[PX];
  auto prefetch = [&](int chunk) {
    int b0 = chunk * TB_T, b1 = chunk * TB_T + 16;      // first rows of the chunk's two 16-row halves
    if (a.live16) {
      const int nl = a.live16[0];
      b0 = a.live16[1 + 2 * chunk] * 16;
      b1 = 2 * chunk + 1 < nl ? a.live16[2 + 2 * chunk] * 16 : a.T;
    }
#pragma unroll
    for (int i = 0; i < PY; ++i) {
      const int c = tid + 512 * i;
      if (c < CY) {
        const int r = c / (N1 / 8), c8 = (c % (N1 / 8)) * 8;
        const int t = (r < 16 ? b0 : b1 - 16) + r;
        if (t < a.T) load_frag(py[i], Y + (size_t)t * a.ldy + c8);
        else frag_zero(py[i]);
      }
    }
#pragma unroll
    for (int i = 0; i < PX; ++i) {
      const int c = tid + 512 * i;
      if (c < CX) {
        const int r = c / (N2 / 8), c8 = (c % (N2 / 8)) * 8;
        const int t = (r < 16 ? b0 : b1 - 16) + r;
        if (t < a.T) load_frag(px[i], X + (size_t)t * a.ldx + c8);
        else frag_zero(px[i]);
      }
    }
  };
  if (c_beg < c_end) prefetch(c_beg);
  for (int chunk = c_beg; chunk < c_end; ++chunk) {
#pragma unroll
    for (int i = 0; i < PY; ++i) {
      const int c = tid + 512 * i;
      if (c < CY) {
        const int r = c / (N1 / 8), c8 = (c % (N1 / 8)) * 8;
        *reinterpret_cast<Frag<T>*>(Ys + r * LDY + c8) = py[i];
      }
    }
#pragma unroll
    for (int i = 0; i < PX; ++i) {
      const int c = tid + 512 * i;
      if (c < CX) {
        const int r = c / (N2 / 8), c8 = (c % (N2 / 8)) * 8;
        if (GELU_X) {
          float v[8];
#pragma unroll
          for (int j = 0; j < 8; ++j) v[j] = gelu_t<false>((float)px[i].v[j]);
          store8(Xs + r * LDX + c8, v);
        } else {
          *reinterpret_cast<Frag<T>*>(Xs + r * LDX + c8) = px[i];
        }
      }
    }
    if (chunk + 1 < c_end) prefetch(chunk + 1);
    lds_barrier();
    Frag<T> af[MT];
#pragma unroll
    for (int i = 0; i < MT; ++i) {
      frag_tr16(af[i], Ys, LDY, m1 + i * 16, li, lg);
      if (do_cs) mma(af[i], ones, cs[i]);
    }
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      Frag<T> bf;
      frag_tr16(bf, Xs, LDX, m2 + j * 16, li, lg);
#pragma unroll
      for (int i = 0; i < MT; ++i) mma(af[i], bf, acc[i][j]);
    }
    lds_barrier();
  }
  if (c_beg >= c_end) return;
#pragma unroll
  for (int i = 0; i < MT; ++i) {
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r)
        atomicAdd(a.dW + (size_t)(m1 + i * 16 + 4 * lg + r) * a.lddw + m2 + j * 16 + li, acc[i][j][r] * a.scale);
    if (do_cs && li == 0) {
#pragma unroll
      for (int r = 0; r < 4; ++r) atomicAdd(a.colsum + m1 + i * 16 + 4 * lg + r, cs[i][r] * a.scale);
    }
  }
}

template <int N1, int N2>
static int launch_big(const rg_gemm_tn_args& a, hipStream_t s) {
  const int nchunks = (a.T + TB_T - 1) / TB_T;
  int grid = nchunks < 256 ? nchunks : 256;
  if (a.prologue_x == RG_PRO_GELU) hipLaunchKernelGGL((gemm_tn_big_kernel<N1, N2, true>), dim3(grid), dim3(512), 0, s, a);
  else hipLaunchKernelGGL((gemm_tn_big_kernel<N1, N2, false>), dim3(grid), dim3(512), 0, s, a);
  RG_CHECK_LAUNCH();
  return 0;
}

// 1 if an instantiation takes this problem
int rg_gemm_tn_big_select(const rg_gemm_tn_args* a, int dtype) {
  if (dtype != RG_BF16 || !a->use_tr || a->T < 8192 || (a->ldy & 7) || (a->ldx & 7)) return 0;
  return (a->N2 == 128 && (a->N1 == 512 || a->N1 == 384 || a->N1 == 256 || a->N1 == 128)) || (a->N1 == 128 && a->N2 == 512);
}

// returns 1 if the shape is not handled here (caller falls back to the generic kernel)
int rg_gemm_tn_big_try(const rg_gemm_tn_args* a, int dtype, hipStream_t s) {
  if (!rg_gemm_tn_big_select(a, dtype)) return 1;
  if (a->N1 == 512 && a->N2 == 128) return launch_big<512, 128>(*a, s);
  if (a->N1 == 128 && a->N2 == 512) return launch_big<128, 512>(*a, s);
  if (a->N1 == 384 && a->N2 == 128) return launch_big<384, 128>(*a, s);
  if (a->N1 == 256 && a->N2 == 128) return launch_big<256, 128>(*a, s);
  if (a->N1 == 128 && a->N2 == 128) return launch_big<128, 128>(*a, s);
  return 1;
}
