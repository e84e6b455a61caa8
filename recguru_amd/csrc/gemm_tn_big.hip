// Weight-gradient GEMM for the hot shapes (bf16 tier):  dW[N1,N2] += sum_t Y[t,n1] * pro(X[t,n2])
// with (N1,N2) in {(512,128), (128,512), (384,128), (128,128)} -- FFN l1 / l2, fused QKV, attention
// out-projection -- plus the bias gradient colsum[n1] += sum_t Y[t,n1].
//
// One workgroup (8 waves) owns the WHOLE dW tile in its accumulators (<= 128 VGPRs per lane) and
// walks a contiguous token range in 64-token chunks, so Y and X are read from HBM exactly once in
// total (the 64x64-tile kernel in gemm.hip re-reads X N1/64 times and Y N2/64 times).  Chunks are
// staged row-major in LDS (raw 16-byte copies, next chunk prefetched in registers) and the
// token-major MFMA fragments come from the transposing LDS read.  The waves split the larger of
// N1 / N2.  Partial results: with a.partials each workgroup stores its accumulators register-major (every store
// instruction = 256 contiguous bytes) and tn_big_reduce_kernel sums the slices into dW; without it, f32 atomics
// (a 16x16 accumulator register is 4 x 64-byte segments per instruction: 256 x |dW| of those cost more than the GEMM).
#include "rg_common.hip.h"
#include "../../include/recguru_hip.h"

#define TB_T 64      // tokens per chunk: two MFMA k-steps per barrier pair and >= 80 KB of loads in flight per CU

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
  extern __shared__ __align__(16) unsigned char smem_tb[];
  T* Ys = reinterpret_cast<T*>(smem_tb);                // [TB_T][LDY]
  T* Xs = Ys + TB_T * LDY;                              // [TB_T][LDX]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 15, lg = lane >> 4;
  const T* __restrict__ Y = reinterpret_cast<const T*>(a.Y);
  const T* __restrict__ X = reinterpret_cast<const T*>(a.X);
  const int m1 = SPLIT1 ? wave * MT * 16 : 0;
  const int m2 = SPLIT1 ? 0 : wave * NT * 16;
  // a.live16 (optional): the list of live 16-row tiles (rg_live_tiles) -- a chunk is then 4 consecutive LIST entries
  // instead of 64 consecutive tokens; rows of padded tiles carry zero upstream gradient and are never read
  const int nchunks = a.live16 ? (a.live16[0] + 3) >> 2 : (a.T + TB_T - 1) / TB_T;
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
  // First rows of a chunk's 16-row sub-tiles.  With a live list they are read ONE CHUNK AHEAD of the prefetch that
  // needs them (list entry -> row address -> data is two dependent memory latencies otherwise, once per chunk), by a
  // per-lane VECTOR load (lane q & 3 holds entry q) queued behind the chunk's data loads: it has landed by the time
  // those are stored to LDS, and it does not drain the LDS queue the way a scalar load's lgkmcnt(0) would.
  // The load is unconditional (clamped index; validity is decided when the value is used): a load under a divergent
  // condition makes the compiler drain vmcnt(0) right behind it -- and with it the whole data prefetch.
  const int nlive = a.live16 ? a.live16[0] : 0;
  const int nt16 = (a.T + 15) >> 4;
  int nb = 0;
  auto rows_of = [&](int chunk) {
    if (a.live16) nb = a.live16[1 + min((TB_T / 16) * chunk + (lane & (TB_T / 16 - 1)), nt16 - 1)];
  };
  auto prefetch = [&](int chunk) {                      // expects nb = rows_of(chunk); leaves nb = rows_of(chunk + 1)
    int bs[TB_T / 16];
#pragma unroll
    for (int q = 0; q < TB_T / 16; ++q) {
      if (a.live16) bs[q] = (TB_T / 16) * chunk + q < nlive ? __builtin_amdgcn_readlane(nb, q) * 16 : a.T;
      else bs[q] = chunk * TB_T + 16 * q;
    }
#pragma unroll
    for (int i = 0; i < PY; ++i) {
      const int c = tid + 512 * i;
      if (c < CY) {
        const int r = c / (N1 / 8), c8 = (c % (N1 / 8)) * 8;
        const int t = bs[r >> 4] + (r & 15);
        if (t < a.T) load_frag(py[i], Y + (size_t)t * a.ldy + c8);
        else frag_zero(py[i]);
      }
    }
#pragma unroll
    for (int i = 0; i < PX; ++i) {
      const int c = tid + 512 * i;
      if (c < CX) {
        const int r = c / (N2 / 8), c8 = (c % (N2 / 8)) * 8;
        const int t = bs[r >> 4] + (r & 15);
        if (t < a.T) load_frag(px[i], X + (size_t)t * a.ldx + c8);
        else frag_zero(px[i]);
      }
    }
    rows_of(chunk + 1);
  };
  if (c_beg < c_end) { rows_of(c_beg); prefetch(c_beg); }
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
#pragma unroll
    for (int k = 0; k < TB_T / 32; ++k) {               // MFMA k-steps of 32 tokens
      if constexpr (MT <= NT) {                         // hold the shorter side's fragments, stream the longer
        Frag<T> af[MT];
#pragma unroll
        for (int i = 0; i < MT; ++i) {
          frag_tr16(af[i], Ys + k * 32 * LDY, LDY, m1 + i * 16, li, lg);
          if (do_cs) mma(af[i], ones, cs[i]);
        }
#pragma unroll
        for (int j = 0; j < NT; ++j) {
          Frag<T> bf;
          frag_tr16(bf, Xs + k * 32 * LDX, LDX, m2 + j * 16, li, lg);
#pragma unroll
          for (int i = 0; i < MT; ++i) mma(af[i], bf, acc[i][j]);
        }
      } else {
        Frag<T> bf[NT];
#pragma unroll
        for (int j = 0; j < NT; ++j) frag_tr16(bf[j], Xs + k * 32 * LDX, LDX, m2 + j * 16, li, lg);
#pragma unroll
        for (int i = 0; i < MT; ++i) {
          Frag<T> af;
          frag_tr16(af, Ys + k * 32 * LDY, LDY, m1 + i * 16, li, lg);
          if (do_cs) mma(af, ones, cs[i]);
#pragma unroll
          for (int j = 0; j < NT; ++j) mma(af, bf[j], acc[i][j]);
        }
      }
    }
    lds_barrier();
  }
  if (c_beg >= c_end) return;
  float* __restrict__ part = a.partials ? a.partials + (size_t)blockIdx.x * (N1 * N2) + tid : nullptr;
#pragma unroll
  for (int i = 0; i < MT; ++i) {
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        if (part) part[((i * NT + j) * 4 + r) * 512] = acc[i][j][r];     // element e = slot * 512 + tid of this workgroup's slice
        else atomicAdd(a.dW + (size_t)(m1 + i * 16 + 4 * lg + r) * a.lddw + m2 + j * 16 + li, acc[i][j][r] * a.scale);
      }
    if (do_cs && li == 0) {
#pragma unroll
      for (int r = 0; r < 4; ++r) atomicAdd(a.colsum + m1 + i * 16 + 4 * lg + r, cs[i][r] * a.scale);
    }
  }
}

// dW += scale * sum over the active workgroups' slices.  Thread = one element e of the register-major slice layout
// (coalesced across threads for every slice); blockIdx.y splits the slices so that small dW still fill the chip.
template <int N1, int N2>
__global__ __launch_bounds__(256) void tn_big_reduce_kernel(rg_gemm_tn_args a, int grid1) {
  constexpr bool SPLIT1 = N1 >= N2;
  constexpr int NT = SPLIT1 ? N2 / 16 : N2 / 128;
  const int nchunks = a.live16 ? (a.live16[0] + 3) >> 2 : (a.T + TB_T - 1) / TB_T;
  if (nchunks <= 0) return;
  const int per = (nchunks + grid1 - 1) / grid1;
  const int nact = (nchunks + per - 1) / per;          // workgroups of the first launch that had a token range
  const int e = blockIdx.x * 256 + threadIdx.x;
  const int b0 = (int)((long long)nact * blockIdx.y / gridDim.y), b1 = (int)((long long)nact * (blockIdx.y + 1) / gridDim.y);
  const float* __restrict__ p = a.partials + e;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  int b = b0;
  for (; b + 4 <= b1; b += 4) {
    s0 += p[(size_t)b * (N1 * N2)];
    s1 += p[(size_t)(b + 1) * (N1 * N2)];
    s2 += p[(size_t)(b + 2) * (N1 * N2)];
    s3 += p[(size_t)(b + 3) * (N1 * N2)];
  }
  for (; b < b1; ++b) s0 += p[(size_t)b * (N1 * N2)];
  const int tid = e & 511, slot = e >> 9, r = slot & 3, j = (slot >> 2) % NT, i = (slot >> 2) / NT;
  const int wave = tid >> 6, li = tid & 15, lg = (tid >> 4) & 3;
  constexpr int MT = SPLIT1 ? N1 / 128 : N1 / 16;
  const int m1 = SPLIT1 ? wave * MT * 16 : 0, m2 = SPLIT1 ? 0 : wave * NT * 16;
  atomicAdd(a.dW + (size_t)(m1 + i * 16 + 4 * lg + r) * a.lddw + m2 + j * 16 + li, ((s0 + s1) + (s2 + s3)) * a.scale);
}

template <int N1, int N2>
static int launch_big(const rg_gemm_tn_args& a, hipStream_t s) {
  const int nchunks = (a.T + TB_T - 1) / TB_T;
  int grid = nchunks < 256 ? nchunks : 256;
  const int smem = TB_T * (N1 + 8 + N2 + 8) * 2;
  if (a.prologue_x == RG_PRO_GELU) {
    hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_tn_big_kernel<N1, N2, true>), hipFuncAttributeMaxDynamicSharedMemorySize, smem);
    hipLaunchKernelGGL((gemm_tn_big_kernel<N1, N2, true>), dim3(grid), dim3(512), smem, s, a);
  } else {
    hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_tn_big_kernel<N1, N2, false>), hipFuncAttributeMaxDynamicSharedMemorySize, smem);
    hipLaunchKernelGGL((gemm_tn_big_kernel<N1, N2, false>), dim3(grid), dim3(512), smem, s, a);
  }
  if (a.partials) {
    const int gx = N1 * N2 / 256;
    const int gy = gx >= 256 ? 4 : (gx >= 128 ? 8 : 16);
    hipLaunchKernelGGL((tn_big_reduce_kernel<N1, N2>), dim3(gx, gy), dim3(256), 0, s, a, grid);
  }
  RG_CHECK_LAUNCH();
  return 0;
}

// 1 if an instantiation takes this problem
int rg_gemm_tn_big_select(const rg_gemm_tn_args* a, int dtype) {
  if (dtype != RG_BF16 || !a->use_tr || a->T < 8192 || (a->ldy & 7) || (a->ldx & 7) || a->colsum_T > 0) return 0;
  return (a->N2 == 128 && (a->N1 == 512 || a->N1 == 384 || a->N1 == 256 || a->N1 == 128)) || (a->N1 == 128 && a->N2 == 512);
}

// bytes of partial-sum scratch the big kernel can use for this problem (0 if it does not take it)
size_t rg_gemm_tn_big_workspace(const rg_gemm_tn_args* a, int dtype) {
  if (!rg_gemm_tn_big_select(a, dtype)) return 0;
  return (size_t)256 * a->N1 * a->N2 * sizeof(float);
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
