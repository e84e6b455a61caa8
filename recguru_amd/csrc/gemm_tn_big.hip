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
#include <cstdlib>
#include <type_traits>
#include "rg_common.hip.h"
#include "rg_det.hip.h"
#include "../../include/recguru_hip.h"

#define TB_T_BF16 64 // tokens per chunk (bf16): two MFMA k-steps per barrier pair and >= 80 KB of loads in flight per CU
#define TB_T_X3 32   // bf16x3: f32 rows, the same bytes per chunk

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

// operand fragment of one column tile, token-major, from the staged image(s): bf16 -- one image; bf16x3 -- the hi image and,
// `lo_ofs` elements behind it, the lo image (the rows were split while they were staged)
__device__ __forceinline__ void frag_tr_op(Frag<__bf16>& f, const __bf16* tile, int ld, int c0, int li, int lg, int lo_ofs) {
  frag_tr16(f, tile, ld, c0, li, lg);
}
__device__ __forceinline__ void frag_tr_op(FragX3& f, const __bf16* tile, int ld, int c0, int li, int lg, int lo_ofs) {
  Frag<__bf16> h, l;
  frag_tr16(h, tile, ld, c0, li, lg);
  frag_tr16(l, tile + lo_ofs, ld, c0, li, lg);
  f.hi = h.v;
  f.lo = l.v;
}
__device__ __forceinline__ void mma_ones(const Frag<__bf16>& a, const Frag<__bf16>& ones, f32x4& c) { mma(a, ones, c); }
__device__ __forceinline__ void mma_ones(const FragX3& a, const Frag<__bf16>& ones, f32x4& c) {      // column sums: (hi + lo) . 1
  c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.lo, ones.v, c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.hi, ones.v, c, 0, 0, 0);
}

// T = __bf16, or x3 (the bf16x3 tier: Y and X are f32 in memory; a chunk's rows are split into a hi and a lo bf16 image while
// they are staged -- 2.5 VALU operations per element, once -- and every product is three MFMAs; chunks of 32 tokens, the bytes
// of the bf16 tier's 64)
template <typename T, int N1, int N2, bool GELU_X>
__device__ __forceinline__ void tn_big_body(const rg_gemm_tn_args& a, const int bid, const int nwg) {      // workgroup bid of nwg
  constexpr bool X3 = std::is_same<T, x3>::value;
  constexpr int TB_T = X3 ? TB_T_X3 : TB_T_BF16;
  typedef typename OpT<T>::type OP;
  constexpr bool SPLIT1 = N1 >= N2;
  constexpr int MT = SPLIT1 ? N1 / 128 : N1 / 16;      // n1 tiles per wave
  constexpr int NT = SPLIT1 ? N2 / 16 : N2 / 128;      // n2 tiles per wave
  constexpr int LDY = N1 + 8, LDX = N2 + 8;
  constexpr int CY = TB_T * N1 / 8, CX = TB_T * N2 / 8;          // 8-element pieces per staged tile
  constexpr int PY = (CY + 511) / 512, PX = (CX + 511) / 512;    // per-thread prefetch registers
  constexpr int YLO = X3 ? TB_T * LDY : 0, XLO = X3 ? TB_T * LDX : 0;   // element offset of the lo image behind the hi image
  extern __shared__ __align__(16) unsigned char smem_tb[];
  __bf16* Ys = reinterpret_cast<__bf16*>(smem_tb);      // [TB_T][LDY] (x3: hi | lo)
  __bf16* Xs = Ys + TB_T * LDY + YLO;                   // [TB_T][LDX] (x3: hi | lo)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 15, lg = lane >> 4;
  const T* __restrict__ Y = reinterpret_cast<const T*>(a.Y);
  const T* __restrict__ X = reinterpret_cast<const T*>(a.X);
  const int m1 = SPLIT1 ? wave * MT * 16 : 0;
  const int m2 = SPLIT1 ? 0 : wave * NT * 16;
  // a.live16 (optional): the list of live 16-row tiles (rg_live_tiles) -- a chunk is then 4 consecutive LIST entries
  // instead of 64 consecutive tokens; rows of padded tiles carry zero upstream gradient and are never read
  const int nchunks = a.live16 ? (a.live16[0] + TB_T / 16 - 1) / (TB_T / 16) : (a.T + TB_T - 1) / TB_T;
  const int per = (nchunks + nwg - 1) / nwg;
  const int c_beg = bid * per, c_end = min(nchunks, c_beg + per);
  const bool do_cs = a.colsum != nullptr && (SPLIT1 || wave == 0);

  f32x4 acc[MT][NT];
  f32x4 cs[MT];
#pragma unroll
  for (int i = 0; i < MT; ++i) {
    cs[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < NT; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  }
  Frag<__bf16> ones;
  frag_fill(ones, 1.f);
  Frag<T> py[PY], px[PX];
  // raw rows -> the staged image(s)
  auto put = [&](__bf16* dst, int lo_ofs, const Frag<T>& raw) {
    if constexpr (X3) {
      bf16x8_t hi, lo;
      split_x3(raw.v, hi, lo);
      *reinterpret_cast<bf16x8_t*>(dst) = hi;
      *reinterpret_cast<bf16x8_t*>(dst + lo_ofs) = lo;
    } else {
      *reinterpret_cast<Frag<T>*>(dst) = raw;
    }
  };
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
        put(Ys + r * LDY + c8, YLO, py[i]);
      }
    }
#pragma unroll
    for (int i = 0; i < PX; ++i) {
      const int c = tid + 512 * i;
      if (c < CX) {
        const int r = c / (N2 / 8), c8 = (c % (N2 / 8)) * 8;
        if (GELU_X) {
          Frag<T> g;
#pragma unroll
          for (int j = 0; j < 8; ++j) g.v[j] = (T)gelu_t<false>((float)px[i].v[j]);
          put(Xs + r * LDX + c8, XLO, g);
        } else {
          put(Xs + r * LDX + c8, XLO, px[i]);
        }
      }
    }
    if (chunk + 1 < c_end) prefetch(chunk + 1);
    lds_barrier();
#pragma unroll
    for (int k = 0; k < TB_T / 32; ++k) {               // MFMA k-steps of 32 tokens
      if constexpr (MT <= NT) {                         // hold the shorter side's fragments, stream the longer
        OP af[MT];
#pragma unroll
        for (int i = 0; i < MT; ++i) {
          frag_tr_op(af[i], Ys + k * 32 * LDY, LDY, m1 + i * 16, li, lg, YLO);
          if (do_cs) mma_ones(af[i], ones, cs[i]);
        }
#pragma unroll
        for (int j = 0; j < NT; ++j) {
          OP bf;
          frag_tr_op(bf, Xs + k * 32 * LDX, LDX, m2 + j * 16, li, lg, XLO);
#pragma unroll
          for (int i = 0; i < MT; ++i) mma(af[i], bf, acc[i][j]);
        }
      } else {
        OP bf[NT];
#pragma unroll
        for (int j = 0; j < NT; ++j) frag_tr_op(bf[j], Xs + k * 32 * LDX, LDX, m2 + j * 16, li, lg, XLO);
#pragma unroll
        for (int i = 0; i < MT; ++i) {
          OP af;
          frag_tr_op(af, Ys + k * 32 * LDY, LDY, m1 + i * 16, li, lg, YLO);
          if (do_cs) mma_ones(af, ones, cs[i]);
#pragma unroll
          for (int j = 0; j < NT; ++j) mma(af, bf[j], acc[i][j]);
        }
      }
    }
    lds_barrier();
  }
  if (c_beg >= c_end) return;
  float* __restrict__ part = a.partials ? a.partials + (size_t)bid * (N1 * N2) + tid : nullptr;
#pragma unroll
  for (int i = 0; i < MT; ++i) {
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        if (part) part[((i * NT + j) * 4 + r) * 512] = acc[i][j][r];     // element e = slot * 512 + tid of this workgroup's slice
        else rg_acc(a.dW + (size_t)(m1 + i * 16 + 4 * lg + r) * a.lddw + m2 + j * 16 + li, acc[i][j][r] * a.scale);
      }
    if (do_cs && li == 0) {
#pragma unroll
      for (int r = 0; r < 4; ++r) rg_acc(a.colsum + m1 + i * 16 + 4 * lg + r, cs[i][r] * a.scale);
    }
  }
}

template <typename T, int N1, int N2, bool GELU_X>
__global__ __launch_bounds__(512) void gemm_tn_big_kernel(rg_gemm_tn_args a) {
  tn_big_body<T, N1, N2, GELU_X>(a, (int)blockIdx.x, (int)gridDim.x);
}

// ---------------------------------------------------------------------------------------------------------------------
// The same product with the token chunks brought in by LDS-DMA (global_load_lds_dwordx4): no staging registers, so a
// ring of NST 32-token stages fits beside the 128 accumulator registers and NST - 1 chunks (80 KB and more per CU) are in
// flight at ALL times -- the register-staged kernel above has one 64-token chunk in flight, and only between its LDS
// store and the next iteration's wait: its chunk time was the HBM latency, 40-46 % of the HBM rate on executed rows.
//   * a stage is the row-major image [32][N1] | [32][N2] with NO padding (a wave's DMA instruction writes 1 KB
//     contiguously: wave-uniform base + lane x 16 B); bank conflicts of the transposing fragment reads are avoided by an XOR
//     swizzle of the 16-byte chunks of a row by 2 (row & 3), applied to the SOURCE address of the DMA and to the reads;
//   * one raw barrier per chunk: wait (counted vmcnt) for this wave's DMAs of chunk i -> barrier (every wave's have landed,
//     and everyone is done reading the stage of chunk i - 1) -> issue the DMAs of chunk i + NST - 1 into that stage ->
//     fragments + MFMAs of chunk i;
//   * rows past T / absent list entries are DMA'd from clamped (valid, finite) rows so that every chunk is exactly NPER
//     instructions per wave, and their Y rows are zeroed in LDS behind one extra barrier (last chunk of the kernel only);
//   * the GELU of the l2 weight gradient (dW2 += dl2^T gelu(h1)) is applied to the X fragments after the read.
template <int N> __device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
template <int NPER, int K> __device__ __forceinline__ void wait_chunks(int k) {      // at most k chunks stay in flight
  if constexpr (K == 0) wait_vmcnt<0>();
  else { if (k >= K) wait_vmcnt<NPER * K>(); else wait_chunks<NPER, K - 1>(k); }
}

template <int N1, int N2> struct TnDma {
  static constexpr int CT = (N1 + N2 <= 384) ? 64 : 32;     // tokens per chunk: two k-steps per barrier at the narrow shapes
  static constexpr int YB = CT * N1 * 2, XB = CT * N2 * 2, STG = YB + XB;
  static constexpr int KY = YB / 8192, KX = XB / 8192, NPER = KY + KX;
  static constexpr int NSTF = (159 * 1024) / STG;
  static constexpr int NST = NSTF > 8 ? 8 : NSTF;
};

template <int N1, int N2, bool GELU_X>
__device__ __forceinline__ void tn_dma_body(const rg_gemm_tn_args& a, const int bid, const int nb) {      // workgroup bid of nb
  typedef __bf16 T;
  typedef TnDma<N1, N2> C;
  constexpr bool SPLIT1 = N1 >= N2;
  constexpr int MT = SPLIT1 ? N1 / 128 : N1 / 16;      // n1 tiles per wave
  constexpr int NT = SPLIT1 ? N2 / 16 : N2 / 128;      // n2 tiles per wave
  constexpr int CT = C::CT, KY = C::KY, NPER = C::NPER, NST = C::NST, STG = C::STG;
  constexpr int EPC = CT / 16, KS = CT / 32;            // list entries (16-row tiles) and MFMA k-steps per chunk
  static_assert(NST >= 3 && (NST - 1) * NPER < 64, "ring depth / vmcnt range");
  extern __shared__ __align__(16) unsigned char smem_td[];                       // the ONLY LDS object (hipcc waits vmcnt(0) before reads otherwise)
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 15, lg = lane >> 4;
  const char* __restrict__ Y = reinterpret_cast<const char*>(a.Y);
  const char* __restrict__ X = reinterpret_cast<const char*>(a.X);
  const int m1 = SPLIT1 ? wave * MT * 16 : 0;
  const int m2 = SPLIT1 ? 0 : wave * NT * 16;
  const int nlive = a.live16 ? a.live16[0] : 0;
  const int nchunks = a.live16 ? (nlive + EPC - 1) / EPC : (a.T + CT - 1) / CT;
  const int per = (nchunks + nb - 1) / nb;
  const int c_beg = bid * per, c_end = min(nchunks, c_beg + per);
  if (c_beg >= c_end) return;
  const bool do_cs = a.colsum != nullptr && (SPLIT1 || wave == 0);

  // per-lane source of DMA instruction k: row r of the chunk (r >> 4: which of its two 16-row tiles), swizzled 16-byte chunk
  int srow[NPER];                   // row inside the chunk (0 .. CT-1)
  unsigned int scb[NPER];           // byte offset of the (swizzled) 16-byte chunk inside the row
#pragma unroll
  for (int k = 0; k < NPER; ++k) {
    const bool isy = k < KY;
    const int n = isy ? N1 : N2;
    const int o = (wave + 8 * (isy ? k : k - KY)) * 1024 + lane * 16;      // byte offset inside the Y (X) image
    const int r = o / (2 * n), c = (o % (2 * n)) >> 4;
    srow[k] = r;
    scb[k] = (unsigned int)((c ^ ((r & 3) << 1)) * 16);
  }
  // This workgroup's slice of the live-tile list goes to LDS once, BEFORE the first DMA: beside LDS-DMAs in flight hipcc
  // waits vmcnt(0) for any ordinary (register-destination) global load, i.e. a list read per chunk drained the ring
  int* lst = reinterpret_cast<int*>(smem_td + NST * STG);           // [EPC * (c_end - c_beg)] first rows (>= T: absent)
  if (a.live16) {
    for (int i = tid; i < EPC * (c_end - c_beg); i += 512) {
      const int e = EPC * c_beg + i;
      lst[i] = e < nlive ? a.live16[1 + min(e, nlive - 1)] * 16 : a.T;
    }
    __syncthreads();
  }
  // LDS reads inside the chunk loop are inline asm: for an LDS read it can see, hipcc waits until EVERY LDS-DMA issued
  // before it has completed (it cannot tell the stages of the ring apart) -- vmcnt(0) in front of each chunk's fragments.
  // Ordering is by hand instead: counted vmcnt + barrier before a stage is read (above), lgkmcnt(0) before use.
  auto bases = [&](int chunk, int (&bb)[EPC]) {          // first rows of the chunk's 16-row tiles (>= T: absent)
    if (a.live16) {
      const unsigned int ad = (unsigned int)(size_t)(lst + EPC * (chunk - c_beg));
      if constexpr (EPC == 2) {
        long long v;
        asm volatile("ds_read_b64 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(ad) : "memory");
        bb[0] = (int)(v & 0xFFFFFFFFll);
        bb[1] = (int)(v >> 32);
      } else {
        typedef __attribute__((ext_vector_type(4))) int i32x4;
        i32x4 v;
        asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(ad) : "memory");
#pragma unroll
        for (int e = 0; e < EPC; ++e) bb[e] = v[e];
      }
    } else {
#pragma unroll
      for (int e = 0; e < EPC; ++e) bb[e] = chunk * CT + 16 * e;
    }
  };
  auto issue = [&](int chunk) {
    int bb[EPC];
    bases(chunk, bb);
    unsigned char* stage = smem_td + (chunk % NST) * STG;
#pragma unroll
    for (int k = 0; k < NPER; ++k) {
      const bool isy = k < KY;
      const unsigned int ld2 = (unsigned int)(isy ? a.ldy : a.ldx) * 2u;
      // rows past the end: clamped to the last row (finite data; their Y rows are zeroed before use)
      int bsel = bb[0];
#pragma unroll
      for (int e = 1; e < EPC; ++e) bsel = (srow[k] >> 4) == e ? bb[e] : bsel;
      const int row = min(bsel + (srow[k] & 15), a.T - 1);
      const char* src = (isy ? Y : X) + (size_t)(unsigned int)row * ld2 + scb[k];
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                       (__attribute__((address_space(3))) void*)(stage + (isy ? 0 : C::YB) + (wave + 8 * (isy ? k : k - KY)) * 1024),
                                       16, 0, 0);
    }
  };

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
  // fragment (token-major, see frag_tr16) of the 16 columns of column tile ti from the swizzled row-major image [32][n]:
  // rows 8 lg + q (+ 4), 16-byte chunk (2 ti + (pp >> 1)) ^ (q << 1) = 2 (ti ^ q) + (pp >> 1), 8-byte half pp & 1
  const int q = li >> 2, pp = li & 3;
  const unsigned int smem0 = (unsigned int)(size_t)smem_td;
  const unsigned int ly0 = (unsigned int)((8 * lg + q) * (2 * N1) + ((pp >> 1) << 4) + (pp & 1) * 8);
  const unsigned int lx0 = (unsigned int)((8 * lg + q) * (2 * N2) + ((pp >> 1) << 4) + (pp & 1) * 8) + (unsigned int)C::YB;
  typedef long long i64;
  auto rd_y = [&](i64& lo, i64& hi, unsigned int stage_ofs, int ti) {
    const unsigned int ad = smem0 + stage_ofs + ly0 + (((unsigned int)ti ^ (unsigned int)q) << 5);
    asm volatile("ds_read_b64_tr_b16 %0, %2\n\tds_read_b64_tr_b16 %1, %2 offset:%3" : "=&v"(lo), "=&v"(hi) : "v"(ad), "n"(8 * N1) : "memory");
  };
  auto rd_x = [&](i64& lo, i64& hi, unsigned int stage_ofs, int ti) {
    const unsigned int ad = smem0 + stage_ofs + lx0 + (((unsigned int)ti ^ (unsigned int)q) << 5);
    asm volatile("ds_read_b64_tr_b16 %0, %2\n\tds_read_b64_tr_b16 %1, %2 offset:%3" : "=&v"(lo), "=&v"(hi) : "v"(ad), "n"(8 * N2) : "memory");
  };
  auto to_frag = [&](Frag<T>& f, i64 lo, i64 hi, bool gelu) {
    union { i64 l; bf16x4_t b; } u0, u1;
    u0.l = lo; u1.l = hi;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      f.v[j] = gelu ? (T)gelu_t<false>((float)u0.b[j]) : u0.b[j];
      f.v[4 + j] = gelu ? (T)gelu_t<false>((float)u1.b[j]) : u1.b[j];
    }
  };

  // prologue: NST - 1 chunks in flight
#pragma unroll 1
  for (int c = c_beg; c < c_end && c < c_beg + NST - 1; ++c) issue(c);
#pragma unroll 1
  for (int chunk = c_beg; chunk < c_end; ++chunk) {
    wait_chunks<NPER, NST - 2>(min(NST - 2, c_end - 1 - chunk));      // this wave's DMAs of `chunk` have landed
    lds_barrier();                                                      // ... and everyone else's; stage of chunk - 1 is free
    if (chunk + NST - 1 < c_end) issue(chunk + NST - 1);
    unsigned char* stage = smem_td + (chunk % NST) * STG;
    if (chunk == nchunks - 1) {                       // only the kernel's last chunk can hold rows that do not exist
      int bb[EPC];
      bases(chunk, bb);
      bool partial = false;
#pragma unroll
      for (int e = 0; e < EPC; ++e) { bb[e] = __builtin_amdgcn_readfirstlane(bb[e]); partial = partial || bb[e] + 16 > a.T; }
      if (partial) {                                  // zero their Y rows
        for (int i = tid; i < CT * N1 / 8; i += 512) {
          const int r = i / (N1 / 8);
          int bsel = bb[0];
#pragma unroll
          for (int e = 1; e < EPC; ++e) bsel = (r >> 4) == e ? bb[e] : bsel;
          const int row = bsel + (r & 15);
          if (row >= a.T) *reinterpret_cast<uint4*>(stage + i * 16) = make_uint4(0u, 0u, 0u, 0u);
        }
        lds_barrier();
      }
    }
    const unsigned int sofs0 = (unsigned int)((chunk % NST) * STG);
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {                   // k-steps of 32 tokens
    // Y fragments and the first X fragment up front; X fragment j + 1 is read while the MFMAs of fragment j run.  LDS reads
    // return in order: with the two reads of fragment j + 1 outstanding, lgkmcnt(2) says everything before them has landed.
    // The registers pass through the wait statements, so no MFMA moves above the wait that covers its operands.
    i64 ya[MT][2], xb[NT][2];
#pragma unroll
    for (int i = 0; i < MT; ++i) rd_y(ya[i][0], ya[i][1], sofs0 + ks * 32 * 2 * N1, (m1 >> 4) + i);
    rd_x(xb[0][0], xb[0][1], sofs0 + ks * 32 * 2 * N2, m2 >> 4);
    Frag<T> af[MT];
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      if (j + 1 < NT) {
        rd_x(xb[j + 1][0], xb[j + 1][1], sofs0 + ks * 32 * 2 * N2, (m2 >> 4) + j + 1);
        asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(xb[j][0]), "+v"(xb[j][1]));
      } else {
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(xb[j][0]), "+v"(xb[j][1]));
      }
      if (j == 0) {
#pragma unroll
        for (int i = 0; i < MT; ++i) {
          asm volatile("" : "+v"(ya[i][0]), "+v"(ya[i][1]));          // (landed with x_0: read before it)
          to_frag(af[i], ya[i][0], ya[i][1], false);
          if (do_cs) mma(af[i], ones, cs[i]);
        }
      }
      Frag<T> bf;
      to_frag(bf, xb[j][0], xb[j][1], GELU_X);
#pragma unroll
      for (int i = 0; i < MT; ++i) mma(af[i], bf, acc[i][j]);
    }
    }
  }
  float* __restrict__ part = a.partials ? a.partials + (size_t)bid * (N1 * N2) + tid : nullptr;
#pragma unroll
  for (int i = 0; i < MT; ++i) {
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        if (part) part[((i * NT + j) * 4 + r) * 512] = acc[i][j][r];     // element e = slot * 512 + tid of this workgroup's slice
        else rg_acc(a.dW + (size_t)(m1 + i * 16 + 4 * lg + r) * a.lddw + m2 + j * 16 + li, acc[i][j][r] * a.scale);
      }
    if (do_cs && li == 0) {
#pragma unroll
      for (int r = 0; r < 4; ++r) rg_acc(a.colsum + m1 + i * 16 + 4 * lg + r, cs[i][r] * a.scale);
    }
  }
}

template <int N1, int N2, bool GELU_X>
__global__ __launch_bounds__(512) void gemm_tn_dma_kernel(rg_gemm_tn_args a) {
  tn_dma_body<N1, N2, GELU_X>(a, (int)blockIdx.x, (int)gridDim.x);
}

// ---------------------------------------------------------------------------------------------------------------------
// The FOUR weight-gradient products of a transformer layer's backward in ONE launch (rg_gemm_tn_layer): slot 0 dW2 +=
// dl2^T gelu(h1) [128 x 512], slot 1 dW1 += dh1^T y [512 x 128], slot 2 dWqkv += dqkv^T x [384 x 128], slot 3 dWo += dz^T ctx
// [128 x 128].  Each call of the single-product kernel pays ~35 us that do not scale with T -- the ramp of 192 one-workgroup
// CUs, the tail, 192 partial dW tiles written and read back by a reduce launch -- 56 calls per step.  Here the workgroups of
// one grid are dealt to the slots in proportion to their bytes (so every slot's token ranges are longer and its partial tiles
// fewer), the four ramps and tails coincide, and one reduce launch sums all partial tiles.
struct rg_tn_layer_args {
  rg_gemm_tn_args p[4];
  int end[4];                 // workgroups [end[i-1], end[i]) work on slot i (an empty slot has no workgroups)
};
// bf16x3 tier: the register-staged bodies (rows split into hi / lo images on the way to LDS)
__global__ __launch_bounds__(512) void gemm_tn_layer_x3_kernel(rg_tn_layer_args m) {
  const int bid = (int)blockIdx.x;
  if (bid < m.end[0]) tn_big_body<x3, 128, 512, true>(m.p[0], bid, m.end[0]);
  else if (bid < m.end[1]) tn_big_body<x3, 512, 128, false>(m.p[1], bid - m.end[0], m.end[1] - m.end[0]);
  else if (bid < m.end[2]) tn_big_body<x3, 384, 128, false>(m.p[2], bid - m.end[1], m.end[2] - m.end[1]);
  else tn_big_body<x3, 128, 128, false>(m.p[3], bid - m.end[2], m.end[3] - m.end[2]);
}
__global__ __launch_bounds__(512) void gemm_tn_layer_kernel(rg_tn_layer_args m) {
  const int bid = (int)blockIdx.x;
  if (bid < m.end[0]) tn_dma_body<128, 512, true>(m.p[0], bid, m.end[0]);
  else if (bid < m.end[1]) tn_dma_body<512, 128, false>(m.p[1], bid - m.end[0], m.end[1] - m.end[0]);
  else if (bid < m.end[2]) tn_dma_body<384, 128, false>(m.p[2], bid - m.end[1], m.end[2] - m.end[1]);
  else tn_dma_body<128, 128, false>(m.p[3], bid - m.end[2], m.end[3] - m.end[2]);
}

// dW += scale * sum over the active workgroups' slices.  Thread = one element e of the register-major slice layout
// (coalesced across threads for every slice); blockIdx.y splits the slices so that small dW still fill the chip.
template <int N1, int N2>
__device__ __forceinline__ void tn_reduce_body(const rg_gemm_tn_args& a, int grid1, int ct, const int bx, const int by, const int ny) {
  constexpr bool SPLIT1 = N1 >= N2;
  constexpr int NT = SPLIT1 ? N2 / 16 : N2 / 128;
  const int nchunks = a.live16 ? (a.live16[0] + ct / 16 - 1) / (ct / 16) : (a.T + ct - 1) / ct;
  if (nchunks <= 0) return;
  const int per = (nchunks + grid1 - 1) / grid1;
  const int nact = (nchunks + per - 1) / per;          // workgroups of the first launch that had a token range
  const int e = bx * 256 + threadIdx.x;
  const int b0 = (int)((long long)nact * by / ny), b1 = (int)((long long)nact * (by + 1) / ny);
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
  rg_acc(a.dW + (size_t)(m1 + i * 16 + 4 * lg + r) * a.lddw + m2 + j * 16 + li, ((s0 + s1) + (s2 + s3)) * a.scale);
}

template <int N1, int N2>
__global__ __launch_bounds__(256) void tn_big_reduce_kernel(rg_gemm_tn_args a, int grid1, int ct) {     // ct: tokens per chunk of the first launch
  tn_reduce_body<N1, N2>(a, grid1, ct, (int)blockIdx.x, (int)blockIdx.y, (int)gridDim.y);
}
// one reduce launch for the four slots of gemm_tn_layer_kernel: blockIdx.x walks the slots' (element block, slice group) pairs
struct rg_tn_layer_reduce_args {
  rg_gemm_tn_args p[4];
  int grid1[4], ct[4], gx[4], gy[4], end[4];
};
__global__ __launch_bounds__(256) void tn_layer_reduce_kernel(rg_tn_layer_reduce_args m) {
  const int bid = (int)blockIdx.x;
  const int i = (bid >= m.end[0]) + (bid >= m.end[1]) + (bid >= m.end[2]);
  const int r = bid - (i ? m.end[i - 1] : 0);
  const int bx = r % m.gx[i], by = r / m.gx[i];
  if (i == 0) tn_reduce_body<128, 512>(m.p[0], m.grid1[0], m.ct[0], bx, by, m.gy[0]);
  else if (i == 1) tn_reduce_body<512, 128>(m.p[1], m.grid1[1], m.ct[1], bx, by, m.gy[1]);
  else if (i == 2) tn_reduce_body<384, 128>(m.p[2], m.grid1[2], m.ct[2], bx, by, m.gy[2]);
  else tn_reduce_body<128, 128>(m.p[3], m.grid1[3], m.ct[3], bx, by, m.gy[3]);
}

static int tn_use_dma() {            // RG_TN_REGSTAGE=1: the register-staged kernel (kept for A/B timing, tools/kb_tn.py)
  static int v = -1;
  if (v < 0) { const char* e = getenv("RG_TN_REGSTAGE"); v = (e && e[0] == '1') ? 0 : 1; }
  return v;
}

template <int N1, int N2, typename T = __bf16>
static int launch_big(const rg_gemm_tn_args& a, hipStream_t s) {
  constexpr bool X3 = std::is_same<T, x3>::value;           // bf16x3: the register-staged kernel (its rows are split on the way to LDS)
  // the DMA kernel keeps its slice of the live-tile list in LDS behind the ring: T up to ~4 M rows (64 K chunks per workgroup)
  const bool dma = !X3 && tn_use_dma() != 0 && (long long)TnDma<N1, N2>::NST * TnDma<N1, N2>::STG + ((a.T + 31) / 32 / 256 + 2) * 8 <= 160 * 1024;
  const int ct = dma ? TnDma<N1, N2>::CT : (X3 ? TB_T_X3 : TB_T_BF16);
  const int nchunks = (a.T + ct - 1) / ct;
  // 192 workgroups, not one per CU: every workgroup leaves a partial dW tile (written, then read by the reduce launch), and
  // three quarters of the CUs already keep the memory system full (512x128 / 384x128 / 128x128 on the live rows of the bench
  // shape: 150.7 / 125.7 / 65.3 us at 256, 145.2 / 118.8 / 64.8 at 192, 145.7 / 117.1 / 67.8 at 160).  RG_TN_GRID overrides.
  static const int grid_cap = [] { const char* e = getenv("RG_TN_GRID"); const int v = e ? atoi(e) : 0; return v > 0 && v <= 256 ? v : 192; }();
  int grid = nchunks < grid_cap ? nchunks : grid_cap;
  const int per = (nchunks + grid - 1) / grid;
  const int smem = dma ? TnDma<N1, N2>::NST * TnDma<N1, N2>::STG + (TnDma<N1, N2>::CT / 16) * per * 4
                       : (X3 ? TB_T_X3 * (N1 + 8 + N2 + 8) * 2 * 2 : TB_T_BF16 * (N1 + 8 + N2 + 8) * 2);
#define RG_TNB(KERN)                                                                                                     \
  do {                                                                                                                    \
    hipFuncSetAttribute(reinterpret_cast<const void*>(KERN), hipFuncAttributeMaxDynamicSharedMemorySize, smem);           \
    hipLaunchKernelGGL((KERN), dim3(grid), dim3(512), smem, s, a);                                                        \
  } while (0)
  if constexpr (X3) {
    if (a.prologue_x == RG_PRO_GELU) RG_TNB((gemm_tn_big_kernel<T, N1, N2, true>));
    else RG_TNB((gemm_tn_big_kernel<T, N1, N2, false>));
  } else if (a.prologue_x == RG_PRO_GELU) {
    if (dma) RG_TNB((gemm_tn_dma_kernel<N1, N2, true>));
    else RG_TNB((gemm_tn_big_kernel<T, N1, N2, true>));
  } else {
    if (dma) RG_TNB((gemm_tn_dma_kernel<N1, N2, false>));
    else RG_TNB((gemm_tn_big_kernel<T, N1, N2, false>));
  }
#undef RG_TNB
  if (a.partials) {
    const int gx = N1 * N2 / 256;
    const int gy = gx >= 256 ? 4 : (gx >= 128 ? 8 : 16);
    hipLaunchKernelGGL((tn_big_reduce_kernel<N1, N2>), dim3(gx, gy), dim3(256), 0, s, a, grid, ct);
  }
  RG_CHECK_LAUNCH();
  return 0;
}

static bool tn_native(int N1, int N2) {
  return (N2 == 128 && (N1 == 512 || N1 == 384 || N1 == 256 || N1 == 128)) || (N1 == 128 && N2 == 512);
}

// Wider problems (d_model = 256: 512 x 256, 256 x 512, 768 x 256, 256 x 256) run as a grid of native blocks, one launch per
// block on the same stream and scratch: block (i, j) reads columns [i*b1, +b1) of Y and [j*b2, +b2) of X, so Y is read
// N2 / b2 times and X N1 / b1 times -- the generic 64 x 64 kernel re-reads them N2 / 64 and N1 / 64 times.
static bool tn_blocks(int N1, int N2, int* b1, int* b2) {
  if (tn_native(N1, N2)) { *b1 = N1; *b2 = N2; return true; }
  if ((N1 & 127) || (N2 & 127) || N1 > 1024 || N2 > 512) return false;
  if (N2 == 512 && N1 <= 512) { *b1 = 128; *b2 = 512; return true; }       // X (the wide operand) read N1 / 128 times
  *b2 = 128;
  *b1 = (N1 % 512 == 0) ? 512 : (N1 % 384 == 0) ? 384 : (N1 % 256 == 0) ? 256 : 128;
  return true;
}

// 1 if an instantiation (or a grid of them) takes this problem
int rg_gemm_tn_big_select(const rg_gemm_tn_args* a, int dtype) {
  if ((dtype != RG_BF16 && dtype != RG_X3) || !a->use_tr || a->T < 8192 || (a->ldy & 7) || (a->ldx & 7) || a->colsum_T > 0) return 0;
  int b1, b2;
  return tn_blocks(a->N1, a->N2, &b1, &b2) ? 1 : 0;
}

// kernel family that runs the big shapes (profiler names: rg_gemm_tn_plan)
const char* rg_gemm_tn_big_name(const rg_gemm_tn_args* a) {
  const bool fits = (long long)3 * 40 * 1024 + ((a->T + 31) / 32 / 256 + 2) * 8 <= 160 * 1024;
  return tn_use_dma() && fits ? "gemm_tn_dma_kernel" : "gemm_tn_big_kernel";
}

// bytes of partial-sum scratch the big kernel can use for this problem (0 if it does not take it)
size_t rg_gemm_tn_big_workspace(const rg_gemm_tn_args* a, int dtype) {
  if (!rg_gemm_tn_big_select(a, dtype)) return 0;
  int b1, b2;
  tn_blocks(a->N1, a->N2, &b1, &b2);
  return (size_t)256 * b1 * b2 * sizeof(float);
}

// returns 1 if the shape is not handled here (caller falls back to the generic kernel)
static int tn_big_native(const rg_gemm_tn_args* a, int dtype, hipStream_t s);

int rg_gemm_tn_big_try(const rg_gemm_tn_args* a, int dtype, hipStream_t s) {
  if (!rg_gemm_tn_big_select(a, dtype)) return 1;
  int b1, b2;
  tn_blocks(a->N1, a->N2, &b1, &b2);
  if (b1 == a->N1 && b2 == a->N2) return tn_big_native(a, dtype, s);
  const size_t esz = dtype == RG_BF16 ? 2 : 4;
  for (int i = 0; i < a->N1 / b1; ++i)
    for (int j = 0; j < a->N2 / b2; ++j) {
      rg_gemm_tn_args t = *a;
      t.Y = (const char*)a->Y + (size_t)i * b1 * esz;
      t.X = (const char*)a->X + (size_t)j * b2 * esz;
      t.dW = a->dW + (size_t)i * b1 * a->lddw + (size_t)j * b2;
      t.colsum = (a->colsum && j == 0) ? a->colsum + (size_t)i * b1 : nullptr;      // the bias gradient once per Y block
      t.N1 = b1;
      t.N2 = b2;
      const int rc = tn_big_native(&t, dtype, s);
      if (rc) return rc < 0 ? rc : rg_set_error_msg(RG_ERR_INVALID, "gemm_tn: block launch refused");
    }
  return 0;
}

static int tn_big_native(const rg_gemm_tn_args* a, int dtype, hipStream_t s) {
  if (dtype == RG_X3) {
    if (a->N1 == 512 && a->N2 == 128) return launch_big<512, 128, x3>(*a, s);
    if (a->N1 == 128 && a->N2 == 512) return launch_big<128, 512, x3>(*a, s);
    if (a->N1 == 384 && a->N2 == 128) return launch_big<384, 128, x3>(*a, s);
    if (a->N1 == 256 && a->N2 == 128) return launch_big<256, 128, x3>(*a, s);
    if (a->N1 == 128 && a->N2 == 128) return launch_big<128, 128, x3>(*a, s);
    return 1;
  }
  if (a->N1 == 512 && a->N2 == 128) return launch_big<512, 128>(*a, s);
  if (a->N1 == 128 && a->N2 == 512) return launch_big<128, 512>(*a, s);
  if (a->N1 == 384 && a->N2 == 128) return launch_big<384, 128>(*a, s);
  if (a->N1 == 256 && a->N2 == 128) return launch_big<256, 128>(*a, s);
  if (a->N1 == 128 && a->N2 == 128) return launch_big<128, 128>(*a, s);
  return 1;
}

// ---- the four weight-gradient products of a layer in one launch ---------------------------------------------------------------
static const int LAYER_N1[4] = {128, 512, 384, 128}, LAYER_N2[4] = {512, 128, 128, 128};

static int layer_ct(int i, int dtype = RG_BF16) { return dtype == RG_X3 ? TB_T_X3 : i == 3 ? TnDma<128, 128>::CT : (i == 2 ? TnDma<384, 128>::CT : (i == 1 ? TnDma<512, 128>::CT : TnDma<128, 512>::CT)); }
static int layer_ring(int i) {
  return i == 3 ? TnDma<128, 128>::NST * TnDma<128, 128>::STG : (i == 2 ? TnDma<384, 128>::NST * TnDma<384, 128>::STG :
         (i == 1 ? TnDma<512, 128>::NST * TnDma<512, 128>::STG : TnDma<128, 512>::NST * TnDma<128, 512>::STG));
}
// LDS of slot i with g workgroups: the DMA ring + this workgroup's slice of the live-tile list behind it
static int layer_lds(const rg_gemm_tn_args& a, int i, int g, int dtype) {
  if (dtype == RG_X3) return TB_T_X3 * (a.N1 + 8 + a.N2 + 8) * 2 * 2;        // hi | lo images of the Y and X chunks (the list rides in registers)
  const int ct = layer_ct(i), nchunks = (a.T + ct - 1) / ct;
  if (g > nchunks) g = nchunks;
  const int per = (nchunks + g - 1) / g;
  return layer_ring(i) + (ct / 16) * per * 4;
}

extern "C" int rg_gemm_tn_layer_supported(const rg_gemm_tn_args* p, const int* wgs, int dtype) {
  if (!p || !wgs || (dtype != RG_BF16 && dtype != RG_X3) || (dtype == RG_BF16 && !tn_use_dma())) return 0;
  int tot = 0;
  for (int i = 0; i < 4; ++i) {
    const rg_gemm_tn_args& a = p[i];
    if (a.T == 0) continue;                                          // empty slot
    if (a.N1 != LAYER_N1[i] || a.N2 != LAYER_N2[i] || (a.prologue_x == RG_PRO_GELU) != (i == 0)) return 0;
    if (!rg_gemm_tn_big_select(&a, dtype) || !a.partials || !a.Y || !a.X || !a.dW || wgs[i] <= 0) return 0;
    if (layer_lds(a, i, wgs[i], dtype) > 160 * 1024) return 0;
    tot += wgs[i];
  }
  return tot <= 256;
}

// bytes of partial-tile scratch slot i needs when the launch deals `wgs` workgroups to it
extern "C" size_t rg_gemm_tn_layer_workspace(int slot, int wgs) { return (size_t)wgs * LAYER_N1[slot] * LAYER_N2[slot] * sizeof(float); }

// wgs[i]: workgroups for slot i (0 for an empty slot; sum <= 256); p[i].partials: >= rg_gemm_tn_layer_workspace(i, wgs[i]) bytes each
extern "C" int rg_gemm_tn_layer(const rg_gemm_tn_args* p, const int* wgs, int dtype, void* stream) {
  if (!rg_gemm_tn_layer_supported(p, wgs, dtype)) return rg_set_error_msg(RG_ERR_UNSUPPORTED, "gemm_tn_layer: needs the bf16 / bf16x3 tier and the four layer "
                                                                     "shapes (128x512 gelu, 512x128, 384x128, 128x128; T >= 8192; partials)");
  hipStream_t s = (hipStream_t)stream;
  rg_tn_layer_args m;
  rg_tn_layer_reduce_args r;
  int tot = 0, rtot = 0, smem = 0;
  for (int i = 0; i < 4; ++i) {
    m.p[i] = p[i];
    r.p[i] = p[i];
    const int ct = layer_ct(i, dtype);
    int g = p[i].T > 0 ? wgs[i] : 0;
    if (p[i].T > 0) {
      const int nchunks = (p[i].T + ct - 1) / ct;
      if (g > nchunks) g = nchunks;
      const int need = layer_lds(p[i], i, g, dtype);
      if (need > smem) smem = need;
    }
    tot += g;
    m.end[i] = tot;
    r.grid1[i] = g;
    r.ct[i] = ct;
    r.gx[i] = LAYER_N1[i] * LAYER_N2[i] / 256;
    r.gy[i] = g == 0 ? 0 : (r.gx[i] >= 256 ? 4 : (r.gx[i] >= 128 ? 8 : 16));
    rtot += r.gx[i] * r.gy[i];
    r.end[i] = rtot;
  }
  if (tot == 0) return 0;
  if (tot > 256 || smem > 160 * 1024) return rg_set_error_msg(RG_ERR_INVALID, "gemm_tn_layer: more than 256 workgroups, or a token range too long for the list slice in LDS");
  if (dtype == RG_X3) {
    hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_tn_layer_x3_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, smem);
    hipLaunchKernelGGL(gemm_tn_layer_x3_kernel, dim3(tot), dim3(512), smem, s, m);
  } else {
    hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_tn_layer_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, smem);
    hipLaunchKernelGGL(gemm_tn_layer_kernel, dim3(tot), dim3(512), smem, s, m);
  }
  hipLaunchKernelGGL(tn_layer_reduce_kernel, dim3(rtot), dim3(256), 0, s, r);
  RG_CHECK_LAUNCH();
  return 0;
}
