// Scaled-dot-product attention core for d_k = d_v = 32 (hard-coded in the reference,
// config_auto4rec.py:35-36), forward and backward, one workgroup per (sequence, head).
//
// Replaces ScaledDotProductAttention.forward (Transformer/transformer.py:119-129) and the mask
// materialisation around it (:54-78, :157; AutoEnc4Rec_cross.py:103-107,130-134): the key-pad /
// causal mask is evaluated in-kernel from the key ids, nothing of size L x L ever reaches HBM, and
// the returned attention map (never consumed on the hot path) is not produced.
//
// Semantics kept bit-for-bit in spirit: scores/sqrt(d_k), then REPLACE-fill -1e9 where masked
// (so a fully masked row is uniform over all L keys, quirk Q3), softmax in f32.
//
// Layout: qkv is [B, L, 3P] row-major (P = H*32): Q | K | V column blocks, head h at h*32.
// Whole key range of one head lives on chip (L <= 16*NKT): S^T = K.Q^T accumulators (key on the
// register axis, query on the lane) are exponentiated in place and fed straight back as the B
// operand of O^T = V^T.P^T (rg_common.hip.h, stacked-accumulator mapping) -- P never touches LDS.
#include <stdlib.h>
#include <type_traits>
#include "rg_common.hip.h"
#include "../../include/recguru_hip.h"

#define DK 32
#define NEG_FILL (-1e9f)

#ifdef RG_STAMP
#define ASTAMP(i) do { unsigned long long t1__ = __builtin_amdgcn_s_memtime(); asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); \
    tacc[i] += t1__ - t0__; t0__ = t1__; } while (0)
#else
#define ASTAMP(i)
#endif

// V operand of O^T = V^T.P^T for the stacked-accumulator slot map: lane (i = dv, g) needs
// V[key = k0 + 4g + j][dv] (j<4) and V[key = k0 + 16 + 4g + j][dv].
//  * bf16: V stays ROW-MAJOR in LDS ([key][32+8], filled with raw 16-byte copies) and the gfx950
//    transposing read ds_read_b64_tr_b16 delivers exactly "4 consecutive rows of one column".
//  * f32 : V is staged transposed ([dv][keys]) and read as two 4-runs.
template <typename T> struct VStage;
template <> struct VStage<__bf16> {
  static constexpr bool TRANSPOSED = false;
  static __device__ __forceinline__ void frag(Frag<__bf16>& f, const __bf16* V, int ldv, int k0, int dv0, int li, int lg) {
    typedef __attribute__((ext_vector_type(4))) short s16x4;
    typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
    const int q = li >> 2, p = li & 3;           // lane 4q+p of the 16-lane group -> row q, columns 4p..4p+3
    const __bf16* p0 = V + (k0 + 4 * lg + q) * ldv + dv0 + 4 * p;
    const __bf16* p1 = p0 + 16 * ldv;
    union { s16x4 s; bf16x4_t b; } u0, u1;
    u0.s = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)p0);
    u1.s = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)p1);
#pragma unroll
    for (int j = 0; j < 4; ++j) { f.v[j] = u0.b[j]; f.v[4 + j] = u1.b[j]; }
  }
  // the two 16-key halves of the k-step at independent key offsets kA, kB (compacted key-tile lists)
  static __device__ __forceinline__ void frag2(Frag<__bf16>& f, const __bf16* V, int ldv, int kA, int kB, int dv0, int li, int lg) {
    typedef __attribute__((ext_vector_type(4))) short s16x4;
    typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
    const int q = li >> 2, p = li & 3;
    union { s16x4 s; bf16x4_t b; } u0, u1;
    u0.s = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(V + (kA + 4 * lg + q) * ldv + dv0 + 4 * p));
    u1.s = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(V + (kB + 4 * lg + q) * ldv + dv0 + 4 * p));
#pragma unroll
    for (int j = 0; j < 4; ++j) { f.v[j] = u0.b[j]; f.v[4 + j] = u1.b[j]; }
  }
};
// Head-major form (rg_attn_args.qkv_hm): the V tile is filled by LDS-DMA, which writes 1 KB per wave instruction CONTIGUOUSLY --
// no row pad.  Rows of exactly 64 bytes with the 16-byte chunks XOR-swizzled by 2 * ((row >> 2) & 1): the 32 lanes of a
// ds_read_b64_tr_b16 service group cover 8 consecutive rows x 2 chunks, and rows r, r + 4 (same banks unswizzled) then use
// disjoint chunk pairs.  The swizzle is applied to the DMA's per-lane SOURCE address and to these reads.
struct VStageHM {
  static __device__ __forceinline__ int off(int row, int col) {       // element offset of (row, col); col % 4 == 0
    return row * DK + ((((col >> 3) ^ (((row >> 2) & 1) << 1)) & 3) << 3) + (col & 7);
  }
  static __device__ __forceinline__ void frag2(Frag<__bf16>& f, const __bf16* V, int kA, int kB, int dv0, int li, int lg) {
    typedef __attribute__((ext_vector_type(4))) short s16x4;
    typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
    const int q = li >> 2, p = li & 3;
    union { s16x4 s; bf16x4_t b; } u0, u1;
    u0.s = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(V + off(kA + 4 * lg + q, dv0 + 4 * p)));
    u1.s = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(V + off(kB + 4 * lg + q, dv0 + 4 * p)));
#pragma unroll
    for (int j = 0; j < 4; ++j) { f.v[j] = u0.b[j]; f.v[4 + j] = u1.b[j]; }
  }
};
template <> struct VStage<float> {
  static constexpr bool TRANSPOSED = true;
  static __device__ __forceinline__ void frag(Frag<float>& f, const float* Vt, int ldv, int k0, int dv0, int li, int lg) {
    const float* vp = Vt + (dv0 + li) * ldv + k0 + 4 * lg;
    load_frag_2x4(f, vp, vp + 16);
  }
  static __device__ __forceinline__ void frag2(Frag<float>& f, const float* Vt, int ldv, int kA, int kB, int dv0, int li, int lg) {
    const float* vp = Vt + (dv0 + li) * ldv + 4 * lg;
    load_frag_2x4(f, vp + kA, vp + kB);
  }
};

template <> struct VStage<x3> {             // f32 in LDS, transposed like the f32 tier's
  static constexpr bool TRANSPOSED = true;
  static __device__ __forceinline__ void frag(Frag<x3>& f, const x3* Vt, int ldv, int k0, int dv0, int li, int lg) {
    const x3* vp = Vt + (dv0 + li) * ldv + k0 + 4 * lg;
    load_frag_2x4(f, vp, vp + 16);
  }
  static __device__ __forceinline__ void frag2(Frag<x3>& f, const x3* Vt, int ldv, int kA, int kB, int dv0, int li, int lg) {
    const x3* vp = Vt + (dv0 + li) * ldv + 4 * lg;
    load_frag_2x4(f, vp + kA, vp + kB);
  }
};

// Masking is branch-free: kbias[key] in LDS is 0 for a live key, -1e30 for a replaced (pad) key and
// -inf beyond L.  s + (-1e30) == -1e30 exactly in f32, so every replaced score is the same value --
// the "replace-fill" semantics of masked_fill_(-1e9): a fully masked row is uniform over all L keys
// (Q3).  Only the reported lse maps the sentinel back to the reference's -1e9.
#define MASK_BIG (-0x1p100f)   // ~ -1.27e30; a power of two so that MASK_BIG * c is exact for any float c

// qlive[t] = 1 iff the 16-query tile t of sequence b holds a row with rowmask != 0 (all 1 without a rowmask).  One
// coalesced load per row and a ballot per wave, in two halves so that the caller can put its own global loads
// between the mask load and its first use; visible after the caller's next LDS barrier.
template <int NT16, int NTH = 256>        // NTH: threads of the workgroup
struct QLive {
  static constexpr int NR = (NT16 * 16 + NTH - 1) / NTH;
  float v[NR];
  bool has_mask;
  int len;
  // dummy: any readable float array, read at [0] when there is no rowmask (== rowmask otherwise)
  __device__ __forceinline__ void load(const float* __restrict__ rowmask, const float* __restrict__ dummy, int b, int L, int tid) {
#pragma unroll
    for (int i = 0; i < NR; ++i) {      // unconditional loads from clamped addresses, selected afterwards (no branch, no wait)
      const int r = i * NTH + tid;
      v[i] = dummy[rowmask ? (size_t)b * L + min(r, L - 1) : 0];      // raw value; interpreted in publish()
    }
    has_mask = rowmask != nullptr;
    len = L;
  }
  __device__ __forceinline__ void publish(int* __restrict__ qlive, int tid) const {
    const int lane = tid & 63;
#pragma unroll
    for (int i = 0; i < NR; ++i) {
      const unsigned long long m = __ballot(has_mask ? (i * NTH + tid < len && v[i] != 0.f) : true);
      if (lane < 4) {
        const int t = (i * NTH + (tid & ~63)) / 16 + lane;          // this wave's 4 tiles
        if (t < NT16) qlive[t] = ((m >> (16 * lane)) & 0xFFFFull) != 0ull;
      }
    }
  }
};

// One (b, h) head's dropout bits, p == 0.5 mode: word w of query row q covers keys 32w..32w+31 and equals
// rg_hash(seed, idx >> 5) for idx = ((b*H + h)*L + q) * LPAD + key -- exactly what rg_keep() would hash,
// computed once per head instead of once per lane and element.  Layout [word][query].
// row0 (a multiple of 16): query rows in front of it belong to 16-row tiles of padded positions only (the forward skips those
// tiles, the backward's dctx rows there are zeros by contract) -- their words are never looked at and are not hashed: 44 % of the
// rows at the bench's lengths.  A group of 32 lanes takes 32 consecutive rows of one word index (consecutive LDS banks); the
// (word, row block) units of the live rows are dealt round-robin to the NTH / 32 groups.
template <int NW, int LPK, int NTH = 256>
__device__ __forceinline__ void fill_dmask(unsigned int* __restrict__ dmask, const DropCfg& drop, int b, int h, int H, int L, int tid,
                                           int row0 = 0) {
  const unsigned int nw = rg_lpad(L) >> 5;
  const unsigned int wbase = ((unsigned int)b * H + h) * L * nw;
  const int nblk = (L - row0 + 31) >> 5;
  constexpr int G = NTH / 32, DW = G % NW, DB = G / NW;      // unit u = blk * NW + w, u += G per round: no division in the loop
  int blk = (tid >> 5) / NW, w = (tid >> 5) - blk * NW;
  while (blk < nblk) {
    const int row = row0 + 32 * blk + (tid & 31);
    if (row < LPK)
      dmask[w * LPK + row] = (row < L && (unsigned int)w < nw) ? rg_hash(drop.seed, wbase + __umul24((unsigned int)row, nw) + w) : 0u;
    w += DW;
    blk += DB;
    if (w >= NW) { w -= NW; ++blk; }
  }
}

// operand helpers of the kernels below whose LDS tiles are bf16: one image per staged tile, or (bf16x3) a hi and a lo image PL
// elements apart; the f32 tier (raw f32 tiles) goes through the Frag<float> overloads
template <int PL> __device__ __forceinline__ void stage_op(__bf16* dst, const Frag<__bf16>& raw) { *reinterpret_cast<Frag<__bf16>*>(dst) = raw; }
template <int PL> __device__ __forceinline__ void stage_op(__bf16* dst, const Frag<x3>& raw) {
  bf16x8_t hi, lo;
  split_x3(raw.v, hi, lo);
  *reinterpret_cast<bf16x8_t*>(dst) = hi;
  *reinterpret_cast<bf16x8_t*>(dst + PL) = lo;
}
template <int PL> __device__ __forceinline__ void load_op(Frag<__bf16>& f, const __bf16* p) { load_frag(f, p); }
template <int PL> __device__ __forceinline__ void load_op(FragX3& f, const __bf16* p) {
  f.hi = *reinterpret_cast<const bf16x8_t*>(p);
  f.lo = *reinterpret_cast<const bf16x8_t*>(p + PL);
}
template <int PL> __device__ __forceinline__ void vstage_op(Frag<__bf16>& f, const __bf16* V, int ldv, int k0, int dv0, int li, int lg) {
  VStage<__bf16>::frag(f, V, ldv, k0, dv0, li, lg);
}
template <int PL> __device__ __forceinline__ void vstage_op(FragX3& f, const __bf16* V, int ldv, int k0, int dv0, int li, int lg) {
  Frag<__bf16> h, l;
  VStage<__bf16>::frag(h, V, ldv, k0, dv0, li, lg);
  VStage<__bf16>::frag(l, V + PL, ldv, k0, dv0, li, lg);
  f.hi = h.v;
  f.lo = l.v;
}
// a raw row fragment as loaded from memory -> operand fragment (bf16x3: split on the way)
__device__ __forceinline__ void raw_to_op(Frag<__bf16>& f, const Frag<__bf16>& raw) { f = raw; }
__device__ __forceinline__ void raw_to_op(FragX3& f, const Frag<x3>& raw) { split_x3(raw.v, f.hi, f.lo); }
__device__ __forceinline__ void acc_to_op(Frag<__bf16>& f, const f32x4& lo, const f32x4& hi) { acc_to_frag(f, lo, hi); }
__device__ __forceinline__ void acc_to_op(FragX3& f, const f32x4& lo, const f32x4& hi) {
  const float v[8] = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  split_x3(v, f.hi, f.lo);
}

template <int PL> __device__ __forceinline__ void stage_op(float* dst, const Frag<float>& raw) { *reinterpret_cast<Frag<float>*>(dst) = raw; }
template <int PL> __device__ __forceinline__ void load_op(Frag<float>& f, const float* p) { load_frag(f, p); }
__device__ __forceinline__ void acc_to_op(Frag<float>& f, const f32x4& lo, const f32x4& hi) { acc_to_frag(f, lo, hi); }
// the two 16-key halves of a V k-step at independent key offsets (forward kernel)
template <int PL> __device__ __forceinline__ void vstage2_op(Frag<__bf16>& f, const __bf16* V, int ldv, int kA, int kB, int dv0, int li, int lg) {
  VStage<__bf16>::frag2(f, V, ldv, kA, kB, dv0, li, lg);
}
template <int PL> __device__ __forceinline__ void vstage2_op(Frag<float>& f, const float* V, int ldv, int kA, int kB, int dv0, int li, int lg) {
  VStage<float>::frag2(f, V, ldv, kA, kB, dv0, li, lg);
}
template <int PL> __device__ __forceinline__ void vstage2_op(FragX3& f, const __bf16* V, int ldv, int kA, int kB, int dv0, int li, int lg) {
  Frag<__bf16> h, l;
  VStage<__bf16>::frag2(h, V, ldv, kA, kB, dv0, li, lg);
  VStage<__bf16>::frag2(l, V + PL, ldv, kA, kB, dv0, li, lg);
  f.hi = h.v;
  f.lo = l.v;
}
template <typename T> __device__ __forceinline__ void op_zero(Frag<T>& f) { frag_zero(f); }
__device__ __forceinline__ void op_zero(FragX3& f) {
#pragma unroll
  for (int j = 0; j < 8; ++j) { f.hi[j] = (__bf16)0.f; f.lo[j] = (__bf16)0.f; }
}
// row sums of P: ones . P (bf16x3: ones . (hi + lo), two MFMAs)
template <typename T> __device__ __forceinline__ void ones_mma(const Frag<T>& ones, const Frag<T>& p, f32x4& c) { mma(ones, p, c); }
__device__ __forceinline__ void ones_mma(const Frag<__bf16>& ones, const FragX3& p, f32x4& c) {
  c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones.v, p.lo, c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones.v, p.hi, c, 0, 0, 0);
}
// p == 0.5 dropout on the PACKED P fragment: 16-bit AND masks per element (m0: elements 0..3, m1: 4..7)
__device__ __forceinline__ void and_frag(bf16x8_t& v, const uint2& m0, const uint2& m1) {
  uint4 pu = __builtin_bit_cast(uint4, v);
  pu.x &= m0.x; pu.y &= m0.y; pu.z &= m1.x; pu.w &= m1.y;
  v = __builtin_bit_cast(bf16x8_t, pu);
}
__device__ __forceinline__ void and_op(Frag<__bf16>& f, const uint2& m0, const uint2& m1) { and_frag(f.v, m0, m1); }
__device__ __forceinline__ void and_op(FragX3& f, const uint2& m0, const uint2& m1) { and_frag(f.hi, m0, m1); and_frag(f.lo, m0, m1); }
__device__ __forceinline__ void and_op(Frag<float>& f, const uint2& m0, const uint2& m1) {}      // (never instantiated: PLUT needs bf16 tiles)

// DM: dropout mode -- 0 none, 1 p == 0.5 (bit table in LDS, AND masks), 2 generic p (16-bit hash fields)
// (256, 2): a register budget of 256 makes hipcc select the VGPR form of the MFMAs; with the default budget
// of 512 it parks every score accumulator in AGPRs and copies it out and back (~6 v_accvgpr moves per score).
// XIN: the kernel gets the layer INPUT x [B, L, d] and the fused projection weight [3P, d] + bias instead of qkv and
// projects its head's K, V (into the LDS tiles) and Q itself -- inference passes only (nothing is saved for a backward):
// the separate Q/K/V GEMM, its [M, 3P] write and this kernel's read of it disappear (the matrix pipe is idle > 90 % of
// this kernel's time, so the 24 extra MFMAs per 16-key tile are free).  A Q tile goes to the wave's own rows of the ctx
// output (which it overwrites with the context later) and comes back through the same prefetch that read qkv.
// HM: head-major qkv (rg_attn_args.qkv_hm; bf16): a head's K / V / Q tiles are contiguous runs, K and V arrive by LDS-DMA.
template <typename T, int NKT, bool CAUSAL, int DM, bool XIN = false, bool HM = false>
__global__ __launch_bounds__(256, 2) void attn_fwd_kernel(rg_attn_args a) {
#ifdef RG_STAMP
  unsigned long long tacc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  unsigned long long t0__ = __builtin_amdgcn_s_memtime();
#endif
  constexpr int LPK = NKT * 16;       // padded key count (multiple of 32)
  // K rows [key][dk].  bf16: 64-byte rows, no pad, the four 16-byte chunks of a row XOR-swizzled by f(row) =
  // -(row >> 2) & 3: the ds_read_b128 fragment reads (lane (li, lg): chunk lg of row li) are conflict-free under the
  // hardware's lane groups, and without the pad the head's tiles fit FOUR workgroups per CU instead of three under
  // dropout (39.4 KB with the bit table; the kernel is VALU-bound and the fourth wave per SIMD is worth 12-19 %).
  // bf16x3 tier (T = x3, f32 in memory): the K and V tiles are bf16 tiles in the bf16 tier's layouts, twice -- a hi and a lo image,
  // PLK / PLV elements apart, split while they are staged -- and Q is split when it is loaded, P when it leaves the accumulators.
  constexpr bool X3 = std::is_same<T, x3>::value;
  typedef typename std::conditional<X3, __bf16, T>::type E;        // element of the LDS tiles
  typedef typename OpT<T>::type OP;                                 // operand fragment
  constexpr int LDK = sizeof(E) == 2 ? DK : DK + 8;
  auto kofs = [](int row, int chunk) { return sizeof(E) == 2 ? row * DK + ((chunk ^ ((-(row >> 2)) & 3)) << 3) : row * (DK + 8) + chunk * 8; };
  constexpr bool VT = VStage<E>::TRANSPOSED;
  static_assert(!HM || (sizeof(T) == 2 && !XIN), "head-major staging: bf16 qkv form");
  static_assert(!X3 || (!XIN && !HM), "bf16x3: token-major qkv");
  constexpr int LDV = VT ? LPK + 8 : (HM ? DK : DK + 8);
  constexpr int VELEMS = VT ? DK * LDV : LPK * LDV;
  constexpr int PLK = X3 ? LPK * LDK : 0, PLV = X3 ? VELEMS : 0;
  __shared__ __align__(16) E Ks[LPK * LDK * (X3 ? 2 : 1)];
  __shared__ __align__(16) E Vs[VELEMS * (X3 ? 2 : 1)];
  __shared__ __align__(16) float kbias[LPK];
  constexpr int NW = (NKT + 1) / 2;   // 32-key hash words per attention row
  __shared__ __align__(16) unsigned int dmask[DM == 1 ? NW * LPK : 4];   // [word][query]: the head's dropout bits (p == 0.5 mode)
  // p == 0.5, bf16: the dropout mask is applied to the PACKED P fragment -- nibble of hash bits -> 4 x 16-bit masks from a
  // 16-entry table (2 VALU + 1 ds_read_b64 + 2 ANDs per 4 elements instead of a bit extract + AND per element), and the
  // f32 probabilities stay undropped, so the row sums come out of the ones-MFMA exactly as without dropout
  constexpr bool PLUT = DM == 1 && sizeof(E) == 2;
  constexpr bool MSUM = DM == 0 || PLUT;
  __shared__ __align__(8) unsigned int dlut[PLUT ? 32 : 2];
  __shared__ int klo_s;               // first key that is not replaced by the pad mask (L if none)
  __shared__ int zpre_s;              // ZKEYS: length of the leading run of zero-input keys (L if all of them are)
  __shared__ int fix_s;               // HM: a row with rowmask == 0 at or behind first_live (not a left-padded sequence)

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 15, lg = lane >> 4;
  int b, h;
  if (!rg_head_of_block((int)blockIdx.x, a.B, a.H, b, h)) return;     // XCD-aware head pairing (rg_common.hip.h)
  const int L = a.L, P = a.H * DK, ld = 3 * P;
  const T* __restrict__ qkv = XIN ? nullptr : reinterpret_cast<const T*>(a.qkv) + (HM ? (size_t)0 : (size_t)b * L * ld);
  // head-major: tensor w of head (b, h) starts at ((w * B + b) * H + h) * L * 32
  const size_t hm_head = ((size_t)b * a.H + h) * L * DK, hm_tensor = (size_t)a.B * a.H * L * DK;
  // first position with rowmask != 0 (loaded FIRST: the DMA addresses and the dropout-word fill need it); HM + x_masked == 2: rows
  // before it are bias rows
  int first_q = 0, first_hm = 0;
  if (a.rowmask != nullptr && a.first_live) first_q = min(a.first_live[b], L);
  if constexpr (HM) {
    if (a.x_masked == 2) first_hm = first_q;
  }
  const int nkt = (L + 31) / 32 * 2;  // live key tiles (wave-uniform)
  const int nqt = (L + 15) / 16;
  DropCfg drop = make_drop(a.drop_p, a.seed);
  if constexpr (DM == 2) drop.onebit = 0u;      // compile the bit-mode branches of the helpers away
  if (tid == 0) { klo_s = L; zpre_s = L; fix_s = 0; }
  if (PLUT && tid < 16) {
    dlut[2 * tid] = ((tid & 1) ? 0xFFFFu : 0u) | ((tid & 2) ? 0xFFFF0000u : 0u);
    dlut[2 * tid + 1] = ((tid & 4) ? 0xFFFFu : 0u) | ((tid & 8) ? 0xFFFF0000u : 0u);
  }
  __syncthreads();                    // klo_s initialised (nothing in flight yet: a cheap barrier)
  // Zero-input keys (x_masked: the caller guarantees that the layer input rows at positions with rowmask == 0 are all
  // zero, so their K rows all equal bk and their V rows bv).  A non-causal head whose FIRST nz >= 2 key tiles consist of
  // such keys only (left padding) computes the scores of tile 0 and treats tiles 1 .. nz-1 as (nz-1)*16 more copies of
  // key 0: same max, (nz-1)*16 (kept: popcount of the dropout bits) more terms exp(s_0 - m) in the row sum, that
  // many times bv in the context -- no MFMA, no exp, no dropout work for 40 % of the key tiles at the bench's lengths.
  constexpr bool ZKEYS = !CAUSAL && !XIN && DM != 2 && sizeof(T) == 2;     // (the f32 tier evaluates every key)
  const bool zkeys = ZKEYS && a.x_masked && a.rowmask != nullptr;

  // This wave's query tiles are wave, wave+4, ...; a tile made of padded positions only is skipped (the layer
  // multiplies those rows by the pad mask, nothing downstream reads their context).  The wave finds its live tiles
  // itself -- lane group g of round rd loads the 16 mask values of tile wave + 4 (g + 4 rd), one ballot per round --
  // with loads issued here, ahead of the K / V staging loads, and consumed after them.
  constexpr int NRD = (NKT + 15) / 16;             // rounds of 4 tiles per wave
  float rmw[NRD];
#pragma unroll
  for (int rd = 0; rd < NRD; ++rd) {
    const int t = wave + 4 * (lg + 4 * rd), row = t * 16 + li;
    // unconditional load from a clamped address (a load under a condition makes hipcc wait for it at the join,
    // i.e. BEFORE the staging loads below are issued); without a rowmask any readable float stands in
    const float* __restrict__ rmp = a.rowmask ? a.rowmask : reinterpret_cast<const float*>(XIN ? a.x : a.qkv);
    const float v = rmp[a.rowmask ? (size_t)b * L + min(row, L - 1) : 0];
    rmw[rd] = (t < nqt && row < L) ? (a.rowmask ? v : 1.f) : 0.f;
  }
  constexpr int NKR = (LPK + 255) / 256;
  bool padk_r[NKR];                                // key ids: loaded ahead of the staging loads as well
  float rmk_r[NKR];                                // the row mask at this thread's keys (zero-input keys)
#pragma unroll
  for (int i = 0; i < NKR; ++i) {
    const int key = i * 256 + tid;
    const int64_t kid = a.key_ids[(size_t)b * L + min(key, L - 1)];
    padk_r[i] = key < L && kid == a.pad_value;
    rmk_r[i] = 1.f;
    if constexpr (ZKEYS || HM) {
      const float* __restrict__ rmp2 = a.rowmask ? a.rowmask : reinterpret_cast<const float*>(a.qkv);
      rmk_r[i] = rmp2[a.rowmask ? (size_t)b * L + min(key, L - 1) : 0];
    }
  }
  if constexpr (HM) {
    // ---- K and V tiles of this head by LDS-DMA: 16 rows (1 KB, contiguous in LDS) per wave instruction, the chunk
    // swizzles of the two images applied to the per-lane SOURCE address.  Rows the kernel must not read from qkv come from
    // pad_rows instead: before first_live (x_masked == 2: such rows may be unwritten) the head's bias row, beyond L zeros.
    const T* __restrict__ padr = reinterpret_cast<const T*>(a.pad_rows);
    const T* __restrict__ kh = qkv + hm_tensor + hm_head;
    const T* __restrict__ vh = qkv + 2 * hm_tensor + hm_head;
    const T* __restrict__ kb_row = padr + (a.H + h) * DK;
    const T* __restrict__ vb_row = padr + (2 * a.H + h) * DK;
    const T* __restrict__ z_row = padr + 3 * a.H * DK;
    // A piece = 16 rows = one wave instruction per tensor.  The per-lane part of the source address does not depend on the piece
    // (pieces start at multiples of 16 rows; the swizzles look at (row >> 2) & 3 only), and all but the (at most two) pieces that
    // straddle first_live or L read ONE run -- the head's rows, the bias row or the zero row: scalar base + per-lane offset, no
    // per-lane selects (they were 35 VALU instructions per piece: with the hash fill, half of this kernel's VALU work per head)
    const int wv = __builtin_amdgcn_readfirstlane(wave);
    const int lr = lane >> 2, cp = lane & 3;
    const int ck = cp ^ ((-(lr >> 2)) & 3), cv = cp ^ (((lr >> 2) & 1) << 1);
    const unsigned int o_row = (unsigned int)lr * DK * (unsigned int)sizeof(T);
#pragma unroll
    for (int j0 = 0; j0 < NKT; j0 += 4) {
      const int j = j0 + wv;                         // 16-row piece of the tiles (wave-uniform)
      if (j < NKT) {
        const int r0 = j * 16;
        const bool beyond = r0 >= L, biasrun = r0 + 16 <= first_hm, liverun = r0 >= first_hm && r0 + 16 <= L;
        if (beyond || biasrun || liverun) {
          const char* kb = reinterpret_cast<const char*>(beyond ? z_row : biasrun ? kb_row : kh + (size_t)r0 * DK);
          const char* vb = reinterpret_cast<const char*>(beyond ? z_row : biasrun ? vb_row : vh + (size_t)r0 * DK);
          const unsigned int ro = liverun ? o_row : 0u;
          __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(kb + (size_t)(ro + 16u * ck)),
                                           (__attribute__((address_space(3))) void*)(Ks + j * 16 * DK), 16, 0, 0);
          __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(vb + (size_t)(ro + 16u * cv)),
                                           (__attribute__((address_space(3))) void*)(Vs + j * 16 * DK), 16, 0, 0);
        } else {
          const int row = r0 + lr;
          const T* ksrc = row >= L ? z_row : (row < first_hm ? kb_row : kh + (size_t)row * DK);
          const T* vsrc = row >= L ? z_row : (row < first_hm ? vb_row : vh + (size_t)row * DK);
          __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(ksrc + 8 * ck),
                                           (__attribute__((address_space(3))) void*)(Ks + j * 16 * DK), 16, 0, 0);
          __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(vsrc + 8 * cv),
                                           (__attribute__((address_space(3))) void*)(Vs + j * 16 * DK), 16, 0, 0);
        }
      }
    }
  }
  // the head's dropout words: hashed HERE, under the latency of the mask / key-id loads above (at the top of the kernel,
  // with nothing in flight, the fill was exposed time: 1 us per head)
#ifndef RG_ABL_NO_DMASK    // (timing-only ablation: the per-head hash of the dropout words)
  if constexpr (DM == 1) fill_dmask<NW, LPK>(dmask, drop, b, h, a.H, L, tid, first_q & ~15);
#endif
  unsigned int wl = 0u;                            // bit i: tile wave + 4 i is live
#pragma unroll
  for (int rd = 0; rd < NRD; ++rd) {
    const unsigned long long m = __ballot(rmw[rd] != 0.f);
#pragma unroll
    for (int g = 0; g < 4; ++g)
      if ((m >> (16 * g)) & 0xFFFFull) wl |= 1u << (g + 4 * rd);
  }
  if constexpr (XIN) {
    // ---- project this head's K, V and Q from x: wave w owns the 16-row tiles w, w + 4, ...
    //   D[n][row] = sum_c W[n][c] x[row][c]  (weight fragment = A operand, x fragment = B operand), so that a lane holds 4
    //   consecutive head features of one row: 8-byte stores into the K / V tiles and into the Q rows.
    constexpr int XD = 128, XKS = XD / 32;             // d_model of this instantiation
    const T* __restrict__ xb = reinterpret_cast<const T*>(a.x) + (size_t)b * L * XD;
    const T* __restrict__ Wp = reinterpret_cast<const T*>(a.wqkv);
    T* __restrict__ qrows = reinterpret_cast<T*>(a.ctx) + (size_t)b * L * P + h * DK;
    constexpr int NOWN = (NKT + 3) / 4;
    // K pass, then V pass (one weight slice in registers at a time: with both, the projection phase -- not the score
    // loop -- set the kernel's register count, 130 > 128, and a workgroup per CU was lost)
#pragma unroll
    for (int pv = 0; pv < 2; ++pv) {
      Frag<T> wf[2][XKS];
#pragma unroll
      for (int nb = 0; nb < 2; ++nb)
#pragma unroll
        for (int ks = 0; ks < XKS; ++ks)
          load_frag(wf[nb][ks], Wp + (size_t)((1 + pv) * P + h * DK + nb * 16 + li) * XD + ks * 32 + 8 * lg);
      float br[2][4];
#pragma unroll
      for (int nb = 0; nb < 2; ++nb) load4f(br[nb], a.bqkv + (1 + pv) * P + h * DK + nb * 16 + 4 * lg);
#pragma unroll
      for (int i = 0; i < NOWN; ++i) {
        const int t = wave + 4 * i;
        if (t >= NKT) break;
        const int key = t * 16 + li;
        f32x4 ac[2] = {(f32x4){0.f, 0.f, 0.f, 0.f}, (f32x4){0.f, 0.f, 0.f, 0.f}};
        // a tile of padded positions only has all-zero x rows when the caller says so (x_masked): K = bk, V = bv
        const bool proj = t < nqt && (!a.x_masked || ((wl >> i) & 1u));
        if (proj) {
          Frag<T> xf[XKS];
#pragma unroll
          for (int ks = 0; ks < XKS; ++ks) load_frag(xf[ks], xb + (size_t)min(key, L - 1) * XD + ks * 32 + 8 * lg);
#pragma unroll
          for (int ks = 0; ks < XKS; ++ks)
#pragma unroll
            for (int nb = 0; nb < 2; ++nb) mma(wf[nb][ks], xf[ks], ac[nb]);
        }
#pragma unroll
        for (int nb = 0; nb < 2; ++nb) {
          float o4[4];
#pragma unroll
          for (int r = 0; r < 4; ++r) o4[r] = key < L ? ac[nb][r] + br[nb][r] : 0.f;
          if (pv == 0) {
            store4(Ks + kofs(key, 2 * nb + (lg >> 1)) + 4 * (lg & 1), o4);
          } else if (VT) {
#pragma unroll
            for (int r = 0; r < 4; ++r) Vs[(nb * 16 + 4 * lg + r) * LDV + key] = (E)o4[r];
          } else {
            store4(Vs + key * LDV + nb * 16 + 4 * lg, o4);
          }
        }
      }
    }
    {
      Frag<T> wq[2][XKS];
#pragma unroll
      for (int nb = 0; nb < 2; ++nb)
#pragma unroll
        for (int ks = 0; ks < XKS; ++ks)
          load_frag(wq[nb][ks], Wp + (size_t)(h * DK + nb * 16 + li) * XD + ks * 32 + 8 * lg);
      float bqr[2][4];
#pragma unroll
      for (int nb = 0; nb < 2; ++nb) load4f(bqr[nb], a.bqkv + h * DK + nb * 16 + 4 * lg);
#pragma unroll
      for (int i = 0; i < NOWN; ++i) {
        const int t = wave + 4 * i;
        if (t >= nqt || !((wl >> i) & 1u)) continue;           // Q only for the query tiles that will be processed
        const int q = t * 16 + li;
        f32x4 aq[2] = {(f32x4){0.f, 0.f, 0.f, 0.f}, (f32x4){0.f, 0.f, 0.f, 0.f}};
        Frag<T> xf[XKS];
#pragma unroll
        for (int ks = 0; ks < XKS; ++ks) load_frag(xf[ks], xb + (size_t)min(q, L - 1) * XD + ks * 32 + 8 * lg);
#pragma unroll
        for (int ks = 0; ks < XKS; ++ks)
#pragma unroll
          for (int nb = 0; nb < 2; ++nb) mma(wq[nb][ks], xf[ks], aq[nb]);
        if (q < L) {
#pragma unroll
          for (int nb = 0; nb < 2; ++nb) {
            float v[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = aq[nb][r] + bqr[nb][r];
            store4(qrows + (size_t)q * P + nb * 16 + 4 * lg, v);
          }
        }
      }
      // the Q rows are read back by this same wave (global memory, its own rows of ctx): stores complete first
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
  } else if constexpr (!HM) {
  // ---- stage K and V of this head (raw 16/32-byte copies) and the key bias row
  // all of a batch's global loads are issued before its first LDS store: one HBM latency per batch of 4 chunks
  // per thread, not one per chunk (the rolled copy loop spent as long staging as computing)
  constexpr int NCH = (LPK * 4 + 255) / 256;
  // x_masked == 2: rows at positions with rowmask == 0 may be unwritten -- the bias rows stand in (this thread always
  // stages chunk tid & 3 of a row: its 8 K-bias and 8 V-bias values are converted once)
  const bool sub = a.x_masked == 2 && a.rowmask != nullptr && a.bqkv != nullptr;
  // rows before the sequence's first live position are bias rows for sure: their loads are pointed at that first live row
  // (hot lines) instead of at unwritten, cold ones
  const int first = (sub && a.first_live) ? min(a.first_live[b], L - 1) : 0;
  Frag<T> kbf, vbf;
  frag_zero(kbf);
  frag_zero(vbf);
  if (sub) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      kbf.v[j] = (T)a.bqkv[P + h * DK + (tid & 3) * 8 + j];
      vbf.v[j] = (T)a.bqkv[2 * P + h * DK + (tid & 3) * 8 + j];
    }
  }
#pragma unroll
  for (int i0 = 0; i0 < NCH; i0 += 4) {
    Frag<T> kr[4], vr[4];
    float rms[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int c = tid + 256 * (i0 + i), key = c >> 2, c8 = (c & 3) * 8;
      const int kc = max(min(key, L - 1), first);     // clamped address, zeroed / replaced below: no branch between the loads
      rms[i] = 1.f;
      if (i0 + i < NCH) {
        load_frag(kr[i], qkv + (size_t)kc * ld + P + h * DK + c8);
        load_frag(vr[i], qkv + (size_t)kc * ld + 2 * P + h * DK + c8);
        const float* __restrict__ rmp3 = a.rowmask ? a.rowmask : reinterpret_cast<const float*>(a.qkv);
        rms[i] = rmp3[a.rowmask ? (size_t)b * L + kc : 0];       // unconditional (see above)
      }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int key = (tid + 256 * (i0 + i)) >> 2;
      if (i0 + i >= NCH || key >= L) { frag_zero(kr[i]); frag_zero(vr[i]); }
      else if (sub && (rms[i] == 0.f || key < first)) { kr[i] = kbf; vr[i] = vbf; }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int c = tid + 256 * (i0 + i), key = c >> 2, c8 = (c & 3) * 8;
      if (i0 + i < NCH && c < LPK * 4) {
        stage_op<PLK>(Ks + kofs(key, c8 >> 3), kr[i]);
        if constexpr (VT) {
#pragma unroll
          for (int j = 0; j < 8; ++j) Vs[(c8 + j) * LDV + key] = vr[i].v[j];
        } else {
          stage_op<PLV>(Vs + key * LDV + c8, vr[i]);
        }
      }
    }
  }
  }
#pragma unroll
  for (int i = 0; i < NKR; ++i) {
    const int key = i * 256 + tid;
    if (key >= LPK) break;
    const bool pad = padk_r[i];
    kbias[key] = key >= L ? -INFINITY : (pad ? MASK_BIG : 0.f);
    if (CAUSAL && key < L && !pad) atomicMin(&klo_s, key);
    if (ZKEYS && key < L && (pad || rmk_r[i] != 0.f)) atomicMin(&zpre_s, key);      // the prefix ends at the first other key
    if (HM && a.x_masked == 2 && a.rowmask != nullptr && key < L && key >= first_hm && rmk_r[i] == 0.f) fix_s = 1;
  }
  // first live tile's Q fragment: in flight across the barrier.  Q rows: columns h*32.. of qkv, or (XIN) this wave's
  // own rows of ctx
  const T* __restrict__ qsrc = XIN ? reinterpret_cast<const T*>(a.ctx) + (size_t)b * L * P + h * DK
                                   : (HM ? qkv + hm_head : qkv + h * DK);
  const int qld = XIN ? P : (HM ? DK : ld);
  OP qnext;
  op_zero(qnext);
  if (wl) {
    const int q = (wave + 4 * __builtin_ctz(wl)) * 16 + li;
    if (q < L) load_frag(qnext, qsrc + (size_t)q * qld + 8 * lg);
  }
  const unsigned int wl0 = wl;
  if constexpr (HM) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // this wave's DMA pieces have landed
  lds_barrier();
  if constexpr (HM) {
    if (fix_s) {
      // rare: rowmask == 0 at or behind first_live (not a left-padded sequence) -- those rows of qkv may be unwritten too:
      // the bias rows are written over what the DMA brought
      const T* __restrict__ padr = reinterpret_cast<const T*>(a.pad_rows);
      for (int c = tid; c < L * 4; c += 256) {
        const int row = c >> 2, ch = c & 3;
        if (row >= first_hm && a.rowmask[(size_t)b * L + row] == 0.f) {
          *reinterpret_cast<Frag<E>*>(Ks + kofs(row, ch)) = *reinterpret_cast<const Frag<E>*>(padr + (a.H + h) * DK + 8 * ch);
          *reinterpret_cast<Frag<E>*>(Vs + VStageHM::off(row, 8 * ch)) = *reinterpret_cast<const Frag<E>*>(padr + (2 * a.H + h) * DK + 8 * ch);
        }
      }
      lds_barrier();
    }
  }
  const int klo = klo_s;
  const int nz = zkeys ? (__builtin_amdgcn_readfirstlane(zpre_s) >> 4) : 0;        // leading key tiles made of zero-input keys only
  const int nskip = nz >= 2 ? (nz - 1) * 16 : 0;   // keys of tiles 1 .. nz-1, folded into key 0
  ASTAMP(0);

  // scores stay RAW dot products; the reference's 1/sqrt(d_k) and log2(e) are folded into the exp2 argument
  const float c2 = a.scale * 1.4426950408889634f;
  while (wl) {
    const int qt = wave + 4 * __builtin_ctz(wl);
    wl &= wl - 1u;                         // next live tile, if any: its Q fragment is prefetched under this tile's work
    const int q = qt * 16 + li;            // this lane's query (column of S^T)
    const int qrel = q - 4 * lg;           // key (= kt*16 + 4*lg + r) > q  <=>  kt*16 + r > qrel
    const OP qf = qnext;
    if (wl) {
      const int q2 = (wave + 4 * __builtin_ctz(wl)) * 16 + li;
      if (q2 < L) load_frag(qnext, qsrc + (size_t)q2 * qld + 8 * lg);
      else op_zero(qnext);
    }
    f32x4 o[2];
    float sum, mx;
    {
      // ---- Key tiles are taken as a LIST (entry j -> tile t(j), all compile-time) cut into regions with one scalar branch
      // per region and loop, branch-free inside a region: the K fragment and bias of entry j + 1 are read while the MFMA of
      // entry j runs (a branch per tile kept every LDS round trip on the wave's critical path: 262 -> 220 us at the bench
      // shape).
      //  * non-causal: every query row sees every key; order 0, NKT-1, NKT-2, ..., 1.  Zero-input keys (see above) sit at
      //    the FRONT of a left-padded sequence, i.e. at the END of the list: its tail is cut into NTR regions of 4 tiles and
      //    a region whose tiles are all zero-input keys is skipped -- its 64 keys are (kept: popcount of the dropout bits)
      //    more copies of key 0.
      //  * causal: ascending order, regions of 4; a region entirely in the future of every row of this query tile
      //    contributes exact zeros and is skipped -- unless a row is FULLY masked (no live key at or before it: uniform
      //    over all L keys, Q3), which can only happen for rows before the first live key (klo).  Inside the last evaluated
      //    region the future keys are masked element-wise (its own instantiation of the region body).
      constexpr int TR = 4;
      constexpr int NTR = CAUSAL ? (NKT + TR - 1) / TR - 1 : (NKT >= 12 ? 2 : (NKT >= 8 ? 1 : 0));
      constexpr int F = CAUSAL ? (NKT < TR ? NKT : TR) : NKT - TR * NTR;
      auto tile_of = [](int j) { return CAUSAL ? j : (j == 0 ? 0 : NKT - j); };
      const int nkq = (CAUSAL && qt * 16 >= klo) ? min(nkt, qt + 1) : nkt;       // causal: tiles 0 .. nkq-1 matter
      const int nfr = CAUSAL ? 0 : min(NTR, (nz - 1) >> 2);     // folded regions (nz == 0: (nz - 1) >> 2 == -1 -> see nrun)
      const int nrun = CAUSAL ? (nkq <= F ? 0 : (nkq - F + TR - 1) / TR) : NTR - max(nfr, 0);     // tail regions that are evaluated
      const int nskip = 16 * TR * max(nfr, 0);            // folded keys: tiles 1 .. TR * nfr
      // (measured and reverted, round 3: skipping the causal forward's LEADING regions made of masked left-padding keys --
      // exact zeros for query tiles behind the first live key -- needs the first evaluated region to issue its own first LDS
      // read under a run-time condition; that lost the read-ahead of entry 0 for every tile: 228 -> 276 us per launch)
      f32x4 s[NKT];
      mx = -INFINITY;
      {
        OP kfb[2];
        float kbb[2][4];
        auto issue = [&](int j, int buf) {
          load_op<PLK>(kfb[buf], Ks + kofs(tile_of(j) * 16 + li, lg));
          load4f(kbb[buf], kbias + tile_of(j) * 16 + 4 * lg);
        };
        issue(0, 0);
#pragma unroll
        for (int rg = 0; rg <= NTR; ++rg)
          if (rg <= nrun) {
            const int j0 = rg == 0 ? 0 : F + (rg - 1) * TR, j1 = min(rg == 0 ? F : j0 + TR, NKT);
            auto finish = [&](int j, auto MK) {           // entry j's scores are complete: future keys masked (MK), row max
              if constexpr (decltype(MK)::value) {
#pragma unroll
                for (int r = 0; r < 4; ++r) s[j][r] = (j * 16 + r > qrel) ? fminf(s[j][r], MASK_BIG) : s[j][r];   // keeps -inf beyond L
              }
              mx = fmaxf(fmaxf(mx, s[j][0]), s[j][1]);
              mx = fmaxf(fmaxf(mx, s[j][2]), s[j][3]);
            };
            auto body = [&](auto MK) {
#pragma unroll
              for (int j = j0; j < j1; ++j) {
                if (j + 1 < NKT) issue(j + 1, (j + 1) & 1);
                s[j] = (f32x4){kbb[j & 1][0], kbb[j & 1][1], kbb[j & 1][2], kbb[j & 1][3]};   // the key bias rides in the accumulator
                mma(kfb[j & 1], qf, s[j]);
                if (j > j0) finish(j - 1, MK);
                __builtin_amdgcn_sched_barrier(0);        // keeps the reads one entry ahead (hoisting them all costs a wave per SIMD)
              }
              finish(j1 - 1, MK);
            };
            if (CAUSAL && j1 - 1 >= qt) body(std::true_type{});      // (uniform) the region reaches the diagonal or beyond
            else body(std::false_type{});
          }
      }
      ASTAMP(1);
      mx = fmaxf(mx, __shfl_xor(mx, 16));
      mx = fmaxf(mx, __shfl_xor(mx, 32));
      const float nmx = -mx * c2;
      sum = 0.f;
      f32x2 sum2 = (f32x2){0.f, 0.f};
#pragma unroll
      for (int rg = 0; rg <= NTR; ++rg)
        if (rg <= nrun) {
          const int j0 = rg == 0 ? 0 : F + (rg - 1) * TR, j1 = min(rg == 0 ? F : j0 + TR, NKT);
#pragma unroll
          for (int j = j0; j < j1; ++j)
#pragma unroll
            for (int r = 0; r < 4; r += 2) {
              const f32x2 arg = (f32x2){s[j][r], s[j][r + 1]} * (f32x2){c2, c2} + (f32x2){nmx, nmx};
              f32x2 p;
              p.x = __builtin_amdgcn_exp2f(arg.x);
              p.y = __builtin_amdgcn_exp2f(arg.y);
              s[j][r] = p.x;
              s[j][r + 1] = p.y;
              if constexpr (!MSUM) sum2 += p;
            }
        }
      // zero-input keys: p of key 0 (every key of tile 0 has it) stands for each of the nskip folded keys
      const float pz = nskip ? s[0][0] : 0.f;
      const float pzr = (float)(T)pz;               // as the P operand of the MFMAs sees it (bf16 tier: rounded)
      if constexpr (!MSUM) {
        sum = sum2.x + sum2.y;
        if (nskip) sum += (float)(nskip >> 2) * pz;                 // a quarter in each of the row's four lanes
        sum += __shfl_xor(sum, 16);
        sum += __shfl_xor(sum, 32);
      }
      float zcnt = (float)nskip;                    // folded keys that survive the attention-map dropout
      if constexpr (DM == 1) {
        if (nskip) {
          int c = 0;
#pragma unroll
          for (int w = 0; w < NW; ++w) {
            const int lo = max(16, 32 * w), hi = min(16 + nskip, 32 * w + 32);      // folded keys covered by hash word w
            if (lo < hi) {
              const unsigned int m = (hi - 32 * w == 32 ? 0xFFFFFFFFu : ((1u << (hi - 32 * w)) - 1u)) & ~((1u << (lo - 32 * w)) - 1u);
              c += __popc(dmask[w * LPK + min(q, L - 1)] & m);
            }
          }
          zcnt = (float)c;
        }
      }
      // dropout on the f32 probabilities (f32 tier, generic p): s[j] holds tile t(j)
      auto entry_of = [](int t) { return CAUSAL ? t : (t == 0 ? 0 : NKT - t); };       // inverse of tile_of
      auto region_on = [&](int j) { return j < F || (j - F) / TR + 1 <= nrun; };        // (uniform) entry j was evaluated
      if constexpr (DM == 1 && !PLUT) {
#pragma unroll
        for (int j = 0; j < NKT; ++j)
          if (!CAUSAL || region_on(j)) {
            const int t = tile_of(j);
            const unsigned int w = dmask[(t >> 1) * LPK + q] >> (4 * lg + 16 * (t & 1));
#pragma unroll
            for (int r = 0; r < 4; ++r) s[j][r] = rg_and(s[j][r], rg_bitmask(w, r));
          }
      } else if constexpr (DM == 2) {
        const unsigned int base = (((unsigned int)b * a.H + h) * L + min(q, L - 1)) * rg_lpad(L) + 4u * lg;
#pragma unroll
        for (int t = 0; t < NKT; t += 2) {            // hash words cover the key-tile pair (t, t + 1): list entries ja, jb
          const int ja = entry_of(t), jb = entry_of(t + 1);
          if (!CAUSAL || region_on(ja)) {             // (causal: a pair never straddles two regions)
            float k0[4], k1[4];
            rg_keep4_pair(drop, base + t * 16, k0, k1);
#pragma unroll
            for (int r = 0; r < 4; ++r) { s[ja][r] *= k0[r]; s[jb][r] *= k1[r]; }
          }
          __builtin_amdgcn_sched_barrier(0);        // one pair's hashes at a time (all of them in flight cost a wave per SIMD)
        }
      }
      ASTAMP(2);
      o[0] = (f32x4){0.f, 0.f, 0.f, 0.f}; o[1] = (f32x4){0.f, 0.f, 0.f, 0.f};
      f32x4 osum = (f32x4){0.f, 0.f, 0.f, 0.f};
      Frag<E> ones;
      frag_fill(ones, 1.f);
#pragma unroll
      for (int rg = 0; rg <= NTR; ++rg)
        if (rg <= nrun) {
          const int j0 = rg == 0 ? 0 : F + (rg - 1) * TR, j1 = min(rg == 0 ? F : j0 + TR, NKT);
#pragma unroll
          for (int ks = j0 / 2; ks < j1 / 2; ++ks) {          // F, TR and NKT are even
            const int tA = tile_of(2 * ks), tB = tile_of(2 * ks + 1);
            OP pf;
            acc_to_op(pf, s[2 * ks], s[2 * ks + 1]);
            // row sums of P as a third product against a ones tile: every accumulator row of a lane is the complete sum
            // over the keys, and it is the sum of exactly the (rounded, undropped) P
            if constexpr (MSUM) ones_mma(ones, pf, osum);
            if constexpr (PLUT) {
              const unsigned int wA = dmask[(tA >> 1) * LPK + q], wB = dmask[(tB >> 1) * LPK + q];
              const unsigned int a0 = __builtin_amdgcn_alignbit(wA, wA, (4 * lg + 29 + 16 * (tA & 1)) & 31) & 0x78u;   // 8 x its nibble
              const unsigned int a1 = __builtin_amdgcn_alignbit(wB, wB, (4 * lg + 29 + 16 * (tB & 1)) & 31) & 0x78u;
              const uint2 m0 = *reinterpret_cast<const uint2*>(reinterpret_cast<const char*>(dlut) + a0);
              const uint2 m1 = *reinterpret_cast<const uint2*>(reinterpret_cast<const char*>(dlut) + a1);
              and_op(pf, m0, m1);
            }
#pragma unroll
            for (int dt = 0; dt < 2; ++dt) {
              OP vf;
              if constexpr (HM) VStageHM::frag2(vf, Vs, tA * 16, tB * 16, dt * 16, li, lg);
              else vstage2_op<PLV>(vf, Vs, LDV, tA * 16, tB * 16, dt * 16, li, lg);
              mma(vf, pf, o[dt]);
            }
          }
        }
      if constexpr (MSUM) sum = osum[0] + (float)nskip * pzr;     // the MFMA row sum adds the ROUNDED P
      if (nskip) {                                  // + (kept folded keys) x p_0 x bv: V row of key 0
        const float wz = pzr * zcnt;
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int dv = dt * 16 + 4 * lg + r;
            o[dt][r] += wz * (float)(VT ? Vs[dv * LDV] : Vs[dv]);
          }
      }
    }
    float inv = __builtin_amdgcn_rcpf(sum);
    if constexpr (DM == 1) inv *= drop.inv_keep;       // dropped entries were ANDed to zero, the 1/(1-p) rides on the normaliser
    ASTAMP(3);
    if (q < L) {
      T* __restrict__ ctx = reinterpret_cast<T*>(a.ctx) + ((size_t)b * L + q) * P + h * DK;
#pragma unroll
      for (int dt = 0; dt < 2; ++dt) {
        float v[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = o[dt][r] * inv;
        store4(ctx + dt * 16 + 4 * lg, v);
      }
#ifndef RG_STAMP
      if (lg == 0 && a.lse) a.lse[((size_t)b * a.H + h) * L + q] = (mx < 0.5f * MASK_BIG ? NEG_FILL : mx * a.scale) + __logf(sum);
#endif
    }
    ASTAMP(4);
  }
  // rows of this wave's padded tiles: context = 0 (finite placeholders for a backward that never reads them); last,
  // so that no load of the loop above waits behind these stores
  for (int i = 0; wave + 4 * i < nqt; ++i) {
    if ((wl0 >> i) & 1u) continue;
    const int q = (wave + 4 * i) * 16 + li;
    if (q < L) {
      T* __restrict__ ctxz = reinterpret_cast<T*>(a.ctx) + ((size_t)b * L + q) * P + h * DK;
      const float z4[4] = {0.f, 0.f, 0.f, 0.f};
      store4(ctxz + 4 * lg, z4);
      store4(ctxz + 16 + 4 * lg, z4);
#ifndef RG_STAMP
      if (lg == 0 && a.lse) a.lse[((size_t)b * a.H + h) * L + q] = 0.f;
#endif
    }
  }
#ifdef RG_STAMP
  if (a.lse != nullptr && blockIdx.x >= 8192 && blockIdx.x < 9216 && lane == 0) {   // diagnostic build: lse doubles as the stamp buffer (steady-state window)
    unsigned long long* dbg = reinterpret_cast<unsigned long long*>(a.lse) + (size_t)((blockIdx.x - 8192) * 4 + wave) * 8;
    for (int i = 0; i < 8; ++i) dbg[i] = tacc[i];
  }
#endif
}

// ------------------------------------------------------------------------------------------------
// Backward.  Phase 1 (dK, dV): waves own key tiles and sweep query pairs with S = Q.K^T (query on
// the register axis).  Phase 2 (dQ): waves own query tiles and sweep key pairs with S^T = K.Q^T.
// Recomputing S/dP in both orientations (7 products instead of 5) removes every LDS transpose of
// P/dS and every cross-workgroup reduction.  Gradients do not flow through replaced (masked)
// scores: dS = 0 there, exactly like masked_fill_ in the reference.
// ------------------------------------------------------------------------------------------------
template <typename T, int NKT>
__global__ __launch_bounds__(256, 2) void attn_bwd_kernel(rg_attn_bwd_args a) {
  constexpr int LPK = NKT * 16;
  constexpr int LDT = LPK + 8;        // transposed images [32][LPK]
  // two transposed images at a time: phase 1 reads Q^T and dO^T, phase 2 K^T -- which is staged into Q^T's space between the
  // phases (three images of [32][L + 8] floats are 163 KB at L = 400: with two, the exact-f32 backward reaches L <= 416)
  __shared__ __align__(16) T Qt[DK * LDT];
  __shared__ __align__(16) T dOt[DK * LDT];
  T* const Kt = Qt;
  __shared__ float lse_s[LPK];
  __shared__ float dl_s[LPK];
  __shared__ unsigned char kpad[LPK];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 15, lg = lane >> 4;
  int b, h;
  if (!rg_head_of_block((int)blockIdx.x, a.B, a.H, b, h)) return;     // XCD-aware head pairing (rg_common.hip.h)
  const int L = a.L, P = a.H * DK, ld = 3 * P;
  const T* __restrict__ qkv = reinterpret_cast<const T*>(a.qkv) + (size_t)b * L * ld;
  const T* __restrict__ dO = reinterpret_cast<const T*>(a.dctx) + (size_t)b * L * P + h * DK;
  const T* __restrict__ O = reinterpret_cast<const T*>(a.ctx) + (size_t)b * L * P + h * DK;
  T* __restrict__ dqkv = reinterpret_cast<T*>(a.dqkv) + (size_t)b * L * ld;
  const int nt = (L + 31) / 32 * 2;   // live 16-row tiles (keys and queries), wave-uniform
  const DropCfg gdrop = make_drop(a.drop_p, a.seed);
  const unsigned int gbase = ((unsigned int)b * a.H + h) * L;
  const unsigned int glp4 = rg_lpad(L);

  for (int c = tid; c < LPK * 4; c += 256) {
    const int row = c >> 2, c8 = (c & 3) * 8;
    float qv[8], gv[8], ov[8];
    if (row < L) {
      load8(qv, qkv + (size_t)row * ld + h * DK + c8);
      load8(gv, dO + (size_t)row * P + c8);
      load8(ov, O + (size_t)row * P + c8);
    } else {
#pragma unroll
      for (int j = 0; j < 8; ++j) { qv[j] = 0.f; gv[j] = 0.f; ov[j] = 0.f; }
    }
    float d = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      Qt[(c8 + j) * LDT + row] = (T)qv[j];
      dOt[(c8 + j) * LDT + row] = (T)gv[j];
      d += gv[j] * ov[j];
    }
    d += __shfl_xor(d, 1);
    d += __shfl_xor(d, 2);
    if ((c & 3) == 0) dl_s[row] = d;
  }
  for (int r = tid; r < LPK; r += 256) {
    lse_s[r] = (r < L) ? a.lse[((size_t)b * a.H + h) * L + r] : 0.f;
    kpad[r] = (r < L && a.key_ids[(size_t)b * L + r] == a.pad_value) ? 1 : 0;
  }
  __syncthreads();

  // ---------------------------------------------------------------- phase 1: dK, dV
  for (int kt = wave; kt < nt; kt += 4) {
    const int key = kt * 16 + li;      // this lane's key (column of S)
    Frag<T> kf, vf;
    if (key < L) {
      load_frag(kf, qkv + (size_t)key * ld + P + h * DK + 8 * lg);
      load_frag(vf, qkv + (size_t)key * ld + 2 * P + h * DK + 8 * lg);
    } else { frag_zero(kf); frag_zero(vf); }
    const bool kmask = (key >= L) || kpad[key];
    f32x4 dk[2] = {(f32x4){0.f, 0.f, 0.f, 0.f}, (f32x4){0.f, 0.f, 0.f, 0.f}};
    f32x4 dv[2] = {(f32x4){0.f, 0.f, 0.f, 0.f}, (f32x4){0.f, 0.f, 0.f, 0.f}};
    for (int qs = 0; qs < nt / 2; ++qs) {
      f32x4 p[2], ds[2];
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int qrow = qs * 32 + u * 16 + li;     // A-operand row of this lane
        Frag<T> qf, gf;
        if (qrow < L) {
          load_frag(qf, qkv + (size_t)qrow * ld + h * DK + 8 * lg);
          load_frag(gf, dO + (size_t)qrow * P + 8 * lg);
        } else { frag_zero(qf); frag_zero(gf); }
        f32x4 sv = (f32x4){0.f, 0.f, 0.f, 0.f}, dp = (f32x4){0.f, 0.f, 0.f, 0.f};
        mma(qf, kf, sv);
        mma(gf, vf, dp);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int q = qs * 32 + u * 16 + 4 * lg + r;   // accumulator row
          const bool masked = kmask || (a.causal && key > q);
          const float sc = masked ? NEG_FILL : sv[r] * a.scale;
          // fully masked row: lse = -1e9 + log L rounds to -1e9 in f32, the row is uniform 1/L (Q3)
          const float lq = lse_s[min(q, LPK - 1)];
          float pv = (q < L && key < L) ? (lq < -5e8f ? 1.f / (float)L : __expf(sc - lq)) : 0.f;
          const float ks = gdrop.thresh ? rg_keep(gdrop, (gbase + min(q, L - 1)) * glp4 + key) : 1.f;
          p[u][r] = pv * ks;
          ds[u][r] = masked ? 0.f : pv * (dp[r] * ks - dl_s[q]) * a.scale;
        }
      }
      Frag<T> pf, dsf;
      acc_to_frag(pf, p[0], p[1]);
      acc_to_frag(dsf, ds[0], ds[1]);
#pragma unroll
      for (int dt = 0; dt < 2; ++dt) {
        Frag<T> gtf, qtf;
        const T* gp = dOt + (dt * 16 + li) * LDT + qs * 32 + 4 * lg;
        const T* qp = Qt + (dt * 16 + li) * LDT + qs * 32 + 4 * lg;
        load_frag_2x4(gtf, gp, gp + 16);
        load_frag_2x4(qtf, qp, qp + 16);
        mma(pf, gtf, dv[dt]);     // dV[key][dv] += sum_q P[q][key] dO[q][dv]
        mma(dsf, qtf, dk[dt]);    // dK[key][dk] += sum_q dS[q][key] Q[q][dk]
      }
    }
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int krow = kt * 16 + 4 * lg + r;
        if (krow < L) {
          dqkv[(size_t)krow * ld + P + h * DK + dt * 16 + li] = (T)dk[dt][r];
          dqkv[(size_t)krow * ld + 2 * P + h * DK + dt * 16 + li] = (T)dv[dt][r];
        }
      }
  }

  // ---------------------------------------------------------------- phase 2: dQ
  __syncthreads();                     // every wave is done with Q^T
  for (int c = tid; c < LPK * 4; c += 256) {
    const int row = c >> 2, c8 = (c & 3) * 8;
    float kv[8];
    if (row < L) load8(kv, qkv + (size_t)row * ld + P + h * DK + c8);
    else {
#pragma unroll
      for (int j = 0; j < 8; ++j) kv[j] = 0.f;
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) Kt[(c8 + j) * LDT + row] = (T)kv[j];
  }
  __syncthreads();
  for (int qt = wave; qt < nt; qt += 4) {
    const int q = qt * 16 + li;        // this lane's query (column of S^T)
    Frag<T> qf, gf;
    if (q < L) {
      load_frag(qf, qkv + (size_t)q * ld + h * DK + 8 * lg);
      load_frag(gf, dO + (size_t)q * P + 8 * lg);
    } else { frag_zero(qf); frag_zero(gf); }
    const float lse_q = lse_s[min(q, LPK - 1)], dl_q = dl_s[min(q, LPK - 1)];
    f32x4 dq[2] = {(f32x4){0.f, 0.f, 0.f, 0.f}, (f32x4){0.f, 0.f, 0.f, 0.f}};
    for (int ks = 0; ks < nt / 2; ++ks) {
      f32x4 ds[2];
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int krow = ks * 32 + u * 16 + li;
        Frag<T> kf, vf;
        if (krow < L) {
          load_frag(kf, qkv + (size_t)krow * ld + P + h * DK + 8 * lg);
          load_frag(vf, qkv + (size_t)krow * ld + 2 * P + h * DK + 8 * lg);
        } else { frag_zero(kf); frag_zero(vf); }
        f32x4 sv = (f32x4){0.f, 0.f, 0.f, 0.f}, dp = (f32x4){0.f, 0.f, 0.f, 0.f};
        mma(kf, qf, sv);          // S^T[key][q]
        mma(vf, gf, dp);          // dP^T[key][q]
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int key = ks * 32 + u * 16 + 4 * lg + r;
          const bool masked = (key >= L) || kpad[key] || (a.causal && key > q);
          const float pv = (q < L && key < L && !masked) ? __expf(sv[r] * a.scale - lse_q) : 0.f;
          const float ks = gdrop.thresh ? rg_keep(gdrop, (gbase + min(q, L - 1)) * glp4 + key) : 1.f;
          ds[u][r] = pv * (dp[r] * ks - dl_q) * a.scale;
        }
      }
      Frag<T> dsf;
      acc_to_frag(dsf, ds[0], ds[1]);
#pragma unroll
      for (int dt = 0; dt < 2; ++dt) {
        Frag<T> ktf;
        const T* kp = Kt + (dt * 16 + li) * LDT + ks * 32 + 4 * lg;
        load_frag_2x4(ktf, kp, kp + 16);
        mma(dsf, ktf, dq[dt]);    // dQ[q][dk] += sum_key dS[q][key] K[key][dk]
      }
    }
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int qrow = qt * 16 + 4 * lg + r;
        if (qrow < L) dqkv[(size_t)qrow * ld + h * DK + dt * 16 + li] = (T)dq[dt][r];
      }
  }
}

// ------------------------------------------------------------------------------------------------
// Backward, bf16 tier: same two-phase algorithm, but every operand of the head lives in LDS
// (Q, K, V, dO row-major, filled by raw 16-byte copies) -- no global loads inside the loops --,
// stacked-slot operands come from the transposing LDS read, masking is the additive key-bias row,
// and the products are oriented so that dK^T / dV^T / dQ^T land with 4 consecutive features per
// lane: gradients leave as packed 8-byte stores.
// ------------------------------------------------------------------------------------------------
//
// ONEPASS (L <= 256): dQ is accumulated in the SAME sweep instead of a second one that recomputes S^T, dP^T and the
// softmax -- the kernel is bound by the VALU (exp2, masks, converts: 14 issue slots per score in phase 1 + 9.5 in
// phase 2 against 1/4 of an MFMA), not by the matrix pipe.  A wave keeps dQ^T for ALL queries of the head in
// registers (NKT x 2 accumulator tiles, partial over the wave's key tiles); each 16 x 16 dS tile is written as bf16
// into the (otherwise unused) pad columns of the wave's own rows of the Q / K tiles and read back with the
// transposing LDS read as the [key][query] operand of a 16x16x16 MFMA against K^T; the four partial dQ^T are summed
// through the (by then dead) Q/K/V/dO tiles at the end.
// NTH: 256 threads, two workgroups per CU -- or 512 (two-phase form only) where the four staged tiles leave room for ONE
// workgroup per CU (L = 400: 133 KB + the dropout bit table): eight waves keep two per SIMD for the issue-bound softmax work
// G: the type of qkv / dctx / ctx / dqkv in memory -- __bf16, or x3 (the bf16x3 tier: f32 in memory; the four staged tiles are
// split into hi and lo bf16 images while they are staged, PL elements apart, P and dS are split when they leave the
// accumulators, every product is three MFMAs; two-phase form, eight waves, L <= 224: 153 KB of LDS at L = 200).
// RESTAGE (round 5, bf16x3 beyond L = 224: config-5's L = 400): only TWO of the four operand tiles live in LDS at a time -- phase 1 (dK, dV)
// sweeps Q and dO from LDS and takes the K / V fragments of a wave's own key tile straight from memory; K and V are then staged INTO THE
// SAME SPACE and phase 2 (dQ) sweeps them with the Q / dO fragments of the wave's own query tile from memory.  hi + lo images of two
// [416, 40] tiles are 133 KB (with the dropout bit table 161.5 KB: one workgroup per CU) where four are 266 KB -- the split-operand
// product at L = 400 instead of the generic exact-f32-layout kernel (860 of 2 425 ms of config-5's bf16x3 step).
template <int NKT, bool CAUSAL, int DM, bool ONEPASS, int NTH = 256, typename G = __bf16, bool RESTAGE = false>
__global__ __launch_bounds__(NTH, NTH == 512 ? 1 : 2) void attn_bwd_bf16_kernel(rg_attn_bwd_args a) {
  static_assert(!ONEPASS || NKT >= 8, "the dS scratch tiles need 4 x 32 rows of pad columns");
  static_assert(!ONEPASS || NTH == 256, "the one-pass dQ reduction is written for four waves");
  constexpr bool X3 = std::is_same<G, x3>::value;
  static_assert(!X3 || !ONEPASS, "bf16x3: two-phase form");
  static_assert(!RESTAGE || !ONEPASS, "restaging between the phases: two-phase form");
  constexpr int NWV = NTH / 64;
  typedef __bf16 T;                                           // element of the LDS tiles
  typedef typename OpT<G>::type OP;                           // operand fragment
  constexpr int LPK = NKT * 16;
  constexpr int LDR = DK + 8;
  constexpr int NTILE = RESTAGE ? 2 : 4;                      // operand tiles resident at a time
  constexpr int PL = X3 ? NTILE * LPK * LDR : 0;              // bf16x3: the lo images sit PL elements behind the hi images
  __shared__ __align__(16) T QKVG[NTILE * LPK * LDR * (X3 ? 2 : 1)];   // one block: reused as f32 scratch by the ONEPASS reduction
  T* const Qs = QKVG;
  T* const Ks = RESTAGE ? QKVG : QKVG + LPK * LDR;                     // RESTAGE: K takes Q's place, V takes dO's
  T* const Vs = RESTAGE ? QKVG + LPK * LDR : QKVG + 2 * LPK * LDR;
  T* const Gs = RESTAGE ? QKVG + LPK * LDR : QKVG + 3 * LPK * LDR;     // dO
  __shared__ __align__(16) float lse2_s[LPK];   // lse * log2(e)   (+inf marks a fully masked row)
  __shared__ __align__(16) float dl_s[LPK];     // delta = rowsum(dO * O)
  __shared__ __align__(16) float rowp_s[LPK];   // 1/L for fully masked rows, else 0
  __shared__ __align__(16) float kbias[LPK];
  constexpr int NW = (NKT + 1) / 2;
  __shared__ __align__(16) unsigned int dmask[DM == 1 ? NW * LPK : 4];   // [word][query] dropout bits (p == 0.5 mode), see fill_dmask
  __shared__ int qlive[NKT];                    // 16-query tile has a row with rowmask != 0 (all 1 without a rowmask)
  __shared__ int klo_s;                         // first key not replaced by the pad mask (causal tile skipping, see forward)

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 15, lg = lane >> 4;
  int b, h;
  if (!rg_head_of_block((int)blockIdx.x, a.B, a.H, b, h)) return;     // XCD-aware head pairing (rg_common.hip.h)
  const int L = a.L, P = a.H * DK, ld = 3 * P;
  // q / k / v rows of this head: token-major qkv [B, L, 3P] (row pitch ld) or head-major q | k | v [3][B][H][L][32] (qkv_hm: a
  // head's rows are contiguous 64-byte runs -- the staging loads of a wave cover whole lines)
  const size_t hm_tensor = (size_t)a.B * a.H * L * DK;
  const G* __restrict__ qrow = reinterpret_cast<const G*>(a.qkv) + (a.qkv_hm ? ((size_t)b * a.H + h) * L * DK : (size_t)b * L * ld + h * DK);
  const G* __restrict__ krow = qrow + (a.qkv_hm ? hm_tensor : (size_t)P);
  const G* __restrict__ vrow = qrow + (a.qkv_hm ? 2 * hm_tensor : (size_t)(2 * P));
  const int qld = a.qkv_hm ? DK : ld;
  const G* __restrict__ dO = reinterpret_cast<const G*>(a.dctx) + (size_t)b * L * P + h * DK;
  const G* __restrict__ O = reinterpret_cast<const G*>(a.ctx) + (size_t)b * L * P + h * DK;
  G* __restrict__ dqkv = reinterpret_cast<G*>(a.dqkv) + (size_t)b * L * ld;
  const int nt = (L + 31) / 32 * 2;   // live 16-row tiles (keys and queries), wave-uniform
  const float c2 = a.scale * 1.4426950408889634f;
  DropCfg drop = make_drop(a.drop_p, a.seed);
  if constexpr (DM == 2) drop.onebit = 0u;      // compile the bit-mode branches of the helpers away
  const unsigned int dbase = ((unsigned int)b * a.H + h) * L;
  const unsigned int lp4 = rg_lpad(L);
  QLive<NKT, NTH> ql;
  const float* __restrict__ rmp = a.rowmask ? a.rowmask : a.lse;
  ql.load(a.rowmask, rmp, b, L, tid); // consumed after the staging loads below are in flight
  float lse_r[QLive<NKT, NTH>::NR];        // same for the per-row softmax statistics and key ids
  bool padk_r[QLive<NKT, NTH>::NR];
#pragma unroll
  for (int i = 0; i < QLive<NKT, NTH>::NR; ++i) {
    const int r = i * NTH + tid;
    const float lv = a.lse[((size_t)b * a.H + h) * L + min(r, L - 1)];
    const int64_t kid = a.key_ids[(size_t)b * L + min(r, L - 1)];
    lse_r[i] = (r < L) ? lv : 0.f;
    padk_r[i] = r < L && kid == a.pad_value;
  }
  if (tid == 0) klo_s = L;
  __syncthreads();

  // a batch's 5 x 4 global loads are all issued before its first LDS store (one HBM latency per 4 chunks per thread)
  constexpr int NCH = (LPK * 4 + NTH - 1) / NTH;
  // RESTAGE: a pair of tiles at a time -- (Q, dO) with the row deltas, later (K, V): plain rows, zeros beyond L (no bias-row substitution
  // in this tier); dO rows of positions with rowmask == 0 are TAKEN as zero, and carry the 1 / (1 - p) in the p == 0.5 mode, as below
  auto stage_pair = [&](const G* __restrict__ r0, int ld0, const G* __restrict__ r1, int ld1, T* __restrict__ t0, T* __restrict__ t1, bool grad) {
#pragma unroll
    for (int i0 = 0; i0 < NCH; i0 += 4) {
      Frag<G> ar[4], br[4], orow[4];
      float rmr[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int c = tid + NTH * (i0 + i), row = c >> 2, c8 = (c & 3) * 8;
        const int rc = min(row, L - 1);
        rmr[i] = 1.f;
        if (i0 + i < NCH) {
          load_frag(ar[i], r0 + (size_t)rc * ld0 + c8);
          load_frag(br[i], r1 + (size_t)rc * ld1 + c8);
          if (grad) {
            load_frag(orow[i], O + (size_t)rc * P + c8);
            const float x = rmp[a.rowmask ? (size_t)b * L + rc : 0];
            rmr[i] = a.rowmask ? x : 1.f;
          }
        }
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int c = tid + NTH * (i0 + i), row = c >> 2, c8 = (c & 3) * 8;
        if (i0 + i < NCH && c < LPK * 4) {
          Frag<G> zf;
          frag_zero(zf);
          const bool in = row < L;
          if (!in) { ar[i] = zf; br[i] = zf; }
          if (grad) {
            if (!in || rmr[i] == 0.f) { br[i] = zf; orow[i] = zf; }
            float d = 0.f;
#pragma unroll
            for (int j = 0; j < 8; ++j) d += (float)br[i].v[j] * (float)orow[i].v[j];
            d += __shfl_xor(d, 1);
            d += __shfl_xor(d, 2);
            if ((c & 3) == 0) dl_s[row] = d;
            if constexpr (DM == 1) {
#pragma unroll
              for (int j = 0; j < 8; ++j) br[i].v[j] = (G)((float)br[i].v[j] * drop.inv_keep);
            }
          }
          stage_op<PL>(t0 + row * LDR + c8, ar[i]);
          stage_op<PL>(t1 + row * LDR + c8, br[i]);
        }
      }
    }
  };
  if constexpr (RESTAGE) {
    if constexpr (DM == 1) fill_dmask<NW, LPK, NTH>(dmask, drop, b, h, a.H, L, tid, 0);
    stage_pair(qrow, qld, dO, P, Qs, Gs, true);
  } else {
  const bool sub = a.x_masked == 2 && a.rowmask != nullptr && a.bqkv != nullptr;
  const int first = (sub && a.first_live) ? min(a.first_live[b], L - 1) : 0;      // see the forward
  Frag<G> qbf, kbf, vbf;              // bias rows of this head, this thread's chunk (tid & 3)
  frag_zero(qbf);
  frag_zero(kbf);
  frag_zero(vbf);
  if (sub) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      qbf.v[j] = (G)a.bqkv[h * DK + (tid & 3) * 8 + j];
      kbf.v[j] = (G)a.bqkv[P + h * DK + (tid & 3) * 8 + j];
      vbf.v[j] = (G)a.bqkv[2 * P + h * DK + (tid & 3) * 8 + j];
    }
  }
#pragma unroll
  for (int i0 = 0; i0 < NCH; i0 += 4) {
    Frag<G> qr[4], kr[4], vr[4], gr[4], orow[4];
    float rmr[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int c = tid + NTH * (i0 + i), row = c >> 2, c8 = (c & 3) * 8;
      const int rc = max(min(row, L - 1), first);     // clamped address, zeroed / replaced below: no branch between the loads
      rmr[i] = 1.f;
      if (i0 + i < NCH) {
        load_frag(qr[i], qrow + (size_t)rc * qld + c8);
        load_frag(kr[i], krow + (size_t)rc * qld + c8);
        load_frag(vr[i], vrow + (size_t)rc * qld + c8);
        load_frag(gr[i], dO + (size_t)rc * P + c8);
        load_frag(orow[i], O + (size_t)rc * P + c8);
        const float x = rmp[a.rowmask ? (size_t)b * L + rc : 0];     // unconditional (see QLive::load)
        rmr[i] = a.rowmask ? x : 1.f;
      }
    }
    // the head's dropout words are hashed under the latency of the staging loads just issued
    if constexpr (DM == 1) { if (i0 == 0) fill_dmask<NW, LPK, NTH>(dmask, drop, b, h, a.H, L, tid, first & ~15); }
    // dctx rows with rowmask == 0 are zero by contract (rg_attn_bwd_args.rowmask) and are TAKEN as zero whatever the
    // buffer holds: its producer may leave the rows of padded 16-row tiles unwritten (rg_gemm_nt_args.skip_dead_fill).
    // The qkv / ctx rows of such positions are real data (a padded position is still a key unless its id is pad_value).
    // A wave's 64 chunks of one round are 16 consecutive rows.  Three wave-uniform cases take no per-lane selects at all -- all
    // 16 rows live (raw rows, delta, scaled dO), all 16 dead (rowmask == 0 or in front of first_live: bias rows or raw rows,
    // delta = 0, dO = 0 -- no products), all 16 beyond L (zeros); only a piece that straddles first_live or L takes the general
    // path.  (The selects, the delta of rows whose dO is zero and the scaling of zeros were a quarter of this kernel's VALU
    // work per head.)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int c = tid + NTH * (i0 + i), row = c >> 2, c8 = (c & 3) * 8;
      if (i0 + i < NCH && c < LPK * 4) {
        const bool in = row < L, dead = rmr[i] == 0.f || row < first;
        const bool all_live = __all(in && !dead), all_dead = __all(in && dead), all_beyond = __all(!in);
        Frag<G> zf;
        frag_zero(zf);
        if (all_live || all_dead) {
          if (all_dead && sub) { qr[i] = qbf; kr[i] = kbf; vr[i] = vbf; }
          stage_op<PL>(Qs + row * LDR + c8, qr[i]);
          stage_op<PL>(Ks + row * LDR + c8, kr[i]);
          stage_op<PL>(Vs + row * LDR + c8, vr[i]);
          if (all_dead) {
            if ((c & 3) == 0) dl_s[row] = 0.f;
            stage_op<PL>(Gs + row * LDR + c8, zf);
            continue;
          }
        } else if (all_beyond) {
          stage_op<PL>(Qs + row * LDR + c8, zf);
          stage_op<PL>(Ks + row * LDR + c8, zf);
          stage_op<PL>(Vs + row * LDR + c8, zf);
          if ((c & 3) == 0) dl_s[row] = 0.f;
          stage_op<PL>(Gs + row * LDR + c8, zf);
          continue;
        } else {
          if (!in) { qr[i] = zf; kr[i] = zf; vr[i] = zf; gr[i] = zf; orow[i] = zf; }
          else if (dead) {
            gr[i] = zf;
            // x_masked == 2: the qkv rows of such positions may be unwritten -- bias rows; ctx rows of padded query tiles are
            // placeholders either way and only meet dO = 0
            if (sub) { qr[i] = qbf; kr[i] = kbf; vr[i] = vbf; orow[i] = zf; }
          }
          stage_op<PL>(Qs + row * LDR + c8, qr[i]);
          stage_op<PL>(Ks + row * LDR + c8, kr[i]);
          stage_op<PL>(Vs + row * LDR + c8, vr[i]);
        }
        float d = 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) d += (float)gr[i].v[j] * (float)orow[i].v[j];
        d += __shfl_xor(d, 1);
        d += __shfl_xor(d, 2);
        if ((c & 3) == 0) dl_s[row] = d;
        if constexpr (DM == 1) {      // p == 0.5 mode: the 1/(1-p) = 2 rides (exactly) on the staged dO, masks are ANDs
#pragma unroll
          for (int j = 0; j < 8; ++j) gr[i].v[j] = (G)((float)gr[i].v[j] * drop.inv_keep);
        }
        stage_op<PL>(Gs + row * LDR + c8, gr[i]);
      }
    }
  }
  }       // (!RESTAGE)
  ql.publish(qlive, tid);
#pragma unroll
  for (int i = 0; i < QLive<NKT, NTH>::NR; ++i) {
    const int r = i * NTH + tid;
    if (r >= LPK) break;
    const float lse = lse_r[i];
    const bool full = lse < -5e8f;                  // -1e9 + log L rounds to -1e9: row was uniform 1/L (Q3)
    // a query row with rowmask == 0 has dO = 0 by contract: its P and dS only ever meet zeros.  Making its P exactly 0
    // changes nothing -- and keeps NaNs out when its lse / ctx / Q were never written (rows of padded tiles under
    // x_masked == 2: the forward skips or mis-feeds them, nothing list-driven reads them)
    const bool deadq = ql.has_mask && ql.v[i] == 0.f;
    lse2_s[r] = (r < L && !full && !deadq) ? lse * 1.4426950408889634f : INFINITY;     // exp2(x - inf) = 0
    rowp_s[r] = (r < L && full && !deadq) ? 1.f / (float)L : 0.f;
    const bool padk = padk_r[i];
    kbias[r] = r >= L ? -INFINITY : (padk ? MASK_BIG : 0.f);
    if (CAUSAL && r < L && !padk) atomicMin(&klo_s, r);
  }
  lds_barrier();
  const int klo = __builtin_amdgcn_readfirstlane(klo_s);
  // bit t: query tile t is live -- as a SCALAR (read from LDS per step, the skip test was a full LDS round trip and a
  // vector compare in front of every step's first MFMA)
  const unsigned int qmask = (unsigned int)__builtin_amdgcn_readfirstlane(
      (int)(unsigned int)__ballot(lane < NKT && qlive[min(lane, NKT - 1)] != 0));

  typedef __attribute__((ext_vector_type(4))) short s16x4;
  typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
  f32x4 dq1[ONEPASS ? NKT : 1][2];                  // ONEPASS: this wave's partial dQ^T, all query tiles
  if constexpr (ONEPASS) {
#pragma unroll
    for (int i = 0; i < NKT; ++i) { dq1[i][0] = (f32x4){0.f, 0.f, 0.f, 0.f}; dq1[i][1] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
  }
  // ---------------------------------------------------------------- phase 1: dK^T, dV^T
  const int nkeyt = (L + 15) >> 4;                  // key tiles that hold a key (nt rounds up to a pair: at L = 200 the 14th
                                                    // tile is keys 208 .. 223 -- nothing to differentiate)
  for (int kt = wave; kt < nkeyt; kt += NWV) {
    const int key = kt * 16 + li;                   // this lane's key (column of S)
    OP kf, vf;
    if constexpr (RESTAGE) {          // this wave's key tile straight from memory (K / V are not in LDS during phase 1)
      Frag<G> kr, vr;
      load_frag(kr, krow + (size_t)min(key, L - 1) * qld + 8 * lg);
      load_frag(vr, vrow + (size_t)min(key, L - 1) * qld + 8 * lg);
      if (key >= L) { frag_zero(kr); frag_zero(vr); }
      raw_to_op(kf, kr);
      raw_to_op(vf, vr);
    } else {
      load_op<PL>(kf, Ks + key * LDR + 8 * lg);
      load_op<PL>(vf, Vs + key * LDR + 8 * lg);
    }
    const float kb = kbias[key];
    f32x4 dkt[2] = {(f32x4){0.f, 0.f, 0.f, 0.f}, (f32x4){0.f, 0.f, 0.f, 0.f}};
    f32x4 dvt[2] = {(f32x4){0.f, 0.f, 0.f, 0.f}, (f32x4){0.f, 0.f, 0.f, 0.f}};
    s16x4 ktf16[2];                                 // ONEPASS: K^T of this key tile as the [dk][key] operand (k = 16 keys)
    if constexpr (ONEPASS) {
#pragma unroll
      for (int dt = 0; dt < 2; ++dt)
        ktf16[dt] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(Ks + (kt * 16 + 4 * lg + (li >> 2)) * LDR + dt * 16 + 4 * (li & 3)));
    }
    auto qstep = [&](const int qs) {
      if (!((qmask >> (2 * qs)) & 3u)) return;             // padded query rows only: dO = 0 there, nothing to add to dK / dV
      if (CAUSAL && 2 * qs + 1 < kt && qs * 32 >= klo) return;     // keys entirely in the future of both query tiles: P = dS = 0
      // a key tile entirely BEFORE the first live key (the decoder's masked left padding): every row at or behind klo has a
      // live key, so its P and dS are exact zeros on these replaced scores; only rows before klo (fully masked: uniform 1/L over
      // ALL keys, quirk Q3 -- with the decoder's shifted inputs the first two live rows of a sequence) reach such a tile
      if (CAUSAL && kt * 16 + 16 <= klo && qs * 32 >= klo) return;
      f32x4 p[2], ds[2];
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int q0 = qs * 32 + u * 16;
        const bool diag = kt * 16 + 15 > q0;        // (uniform) this key tile reaches past the first query of the tile
        OP qf, gf;
        load_op<PL>(qf, Qs + (q0 + li) * LDR + 8 * lg);
        load_op<PL>(gf, Gs + (q0 + li) * LDR + 8 * lg);
        f32x4 sv = (f32x4){kb, kb, kb, kb}, dp = (f32x4){0.f, 0.f, 0.f, 0.f};     // key bias rides in the accumulator
        mma(qf, kf, sv);
        mma(gf, vf, dp);
        float l4[4], d4[4], r4[4];
        load4f(l4, lse2_s + q0 + 4 * lg);
        load4f(d4, dl_s + q0 + 4 * lg);
        load4f(r4, rowp_s + q0 + 4 * lg);
        float ks4[4] = {1.f, 1.f, 1.f, 1.f};
        unsigned int km4[4] = {~0u, ~0u, ~0u, ~0u};
        if constexpr (DM == 1) {        // dO is pre-scaled by 1/(1-p): a dropped entry is an AND with 0
          const uint4 w4 = *reinterpret_cast<const uint4*>(dmask + (key >> 5) * LPK + q0 + 4 * lg);
          km4[0] = rg_bitmask(w4.x, key & 31); km4[1] = rg_bitmask(w4.y, key & 31);     // one v_bfe_i32 each
          km4[2] = rg_bitmask(w4.z, key & 31); km4[3] = rg_bitmask(w4.w, key & 31);
        } else if constexpr (DM == 2) {
#pragma unroll
          for (int r = 0; r < 4; ++r) ks4[r] = rg_keep(drop, (dbase + min(q0 + 4 * lg + r, L - 1)) * lp4 + key);
        }
        if (CAUSAL && diag) {       // its own block AHEAD of the element loop (a uniform branch inside would split it)
#pragma unroll
          for (int r = 0; r < 4; ++r) sv[r] = (key > q0 + 4 * lg + r) ? fminf(sv[r], MASK_BIG) : sv[r];
        }
#pragma unroll
        for (int r = 0; r < 4; r += 2) {          // element pairs: the fma / add / sub / mul as packed v_pk_* instructions
          const f32x2 arg = (f32x2){sv[r], sv[r + 1]} * (f32x2){c2, c2} - (f32x2){l4[r], l4[r + 1]};   // scores: -2^100 / -inf where replaced
          f32x2 pe;                                                     // 0 where masked or row fully masked
          pe.x = __builtin_amdgcn_exp2f(arg.x);
          pe.y = __builtin_amdgcn_exp2f(arg.y);
          // uniform 1/L rows (Q3); columns key >= L only reach dV rows that are never stored
          const f32x2 pu = pe + (f32x2){r4[r], r4[r + 1]};
          const f32x2 dl = (f32x2){d4[r], d4[r + 1]};
          f32x2 dd, pp;
          if constexpr (DM == 1) {
            dd = (f32x2){rg_and(dp[r], km4[r]), rg_and(dp[r + 1], km4[r + 1])};
            pp = (f32x2){rg_and(pu.x, km4[r]), rg_and(pu.y, km4[r + 1])};      // the dropped map feeds dV
          } else if constexpr (DM == 2) {
            dd = (f32x2){dp[r], dp[r + 1]} * (f32x2){ks4[r], ks4[r + 1]};
            pp = pu * (f32x2){ks4[r], ks4[r + 1]};
          } else {
            dd = (f32x2){dp[r], dp[r + 1]};
            pp = pu;
          }
          const f32x2 dsv = pe * (dd - dl);       // the 1/sqrt(d_k) is applied to dK / dQ on the way out
          ds[u][r] = dsv.x; ds[u][r + 1] = dsv.y;
          p[u][r] = pp.x; p[u][r + 1] = pp.y;
        }
      }
      OP pf, dsf;
      acc_to_op(pf, p[0], p[1]);
      acc_to_op(dsf, ds[0], ds[1]);
#pragma unroll
      for (int dt = 0; dt < 2; ++dt) {
        OP gtf, qtf;
        vstage_op<PL>(gtf, Gs, LDR, qs * 32, dt * 16, li, lg);
        vstage_op<PL>(qtf, Qs, LDR, qs * 32, dt * 16, li, lg);
        mma(gtf, pf, dvt[dt]);     // dV^T[dv][key] += sum_q dO[q][dv] P[q][key]
        mma(qtf, dsf, dkt[dt]);    // dK^T[dk][key] += sum_q Q[q][dk] dS[q][key]
      }
#ifndef RG_ABL_NO_DS      // (timing-only ablation, tools/ab_round5.sh: the whole in-sweep dQ path -- dS scratch write, transposing read, 16-deep MFMAs)
      if constexpr (ONEPASS) {
        // dS[q][key] (accumulator layout: lane = key, registers = 4 queries) -> bf16 -> the wave's scratch tile
        // T[key][q] in the pad columns (32..39) of ITS rows of Qs (q 0..7) and Ks (q 8..15), one 8-byte store per
        // lane; the transposing read hands lane (q, lg) the keys 4 lg .. 4 lg + 3 of column q: the [key][q] operand
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          bf16x4_t pk;
#pragma unroll
          for (int r = 0; r < 4; ++r) pk[r] = (T)ds[u][r];
          T* wr = (lg < 2 ? Qs : Ks) + (wave * 32 + u * 16 + li) * LDR + DK + 4 * (lg & 1);
          *reinterpret_cast<bf16x4_t*>(wr) = pk;
        }
        asm volatile("" ::: "memory");              // LDS operations of one wave are executed in order
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          const int pq = li & 3;
          const T* rd = (pq < 2 ? Qs : Ks) + (wave * 32 + u * 16 + 4 * lg + (li >> 2)) * LDR + DK + 4 * (pq & 1);
          const s16x4 dst16 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)rd);
#pragma unroll
          for (int dt = 0; dt < 2; ++dt)        // dQ^T[dk][q] += sum_key K[key][dk] dS[q][key]
            dq1[2 * qs + u][dt] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(ktf16[dt], dst16, dq1[2 * qs + u][dt], 0, 0, 0);
        }
        asm volatile("" ::: "memory");
      }
#endif
    };
    if constexpr (ONEPASS) {
#pragma unroll
      for (int qs = 0; qs < NKT / 2; ++qs)
        if (qs < nt / 2) qstep(qs);
    } else {
      for (int qs = 0; qs < nt / 2; ++qs) qstep(qs);
    }
    if (key < L) {
#pragma unroll
      for (int dt = 0; dt < 2; ++dt) {
        float v[4] = {dkt[dt][0] * a.scale, dkt[dt][1] * a.scale, dkt[dt][2] * a.scale, dkt[dt][3] * a.scale};
        float w[4] = {dvt[dt][0], dvt[dt][1], dvt[dt][2], dvt[dt][3]};
        store4(dqkv + (size_t)key * ld + P + h * DK + dt * 16 + 4 * lg, v);
        store4(dqkv + (size_t)key * ld + 2 * P + h * DK + dt * 16 + 4 * lg, w);
      }
    }
  }

  if constexpr (!ONEPASS) {
  if constexpr (RESTAGE) {
    __syncthreads();                                // every wave is done with Q / dO
    stage_pair(krow, qld, vrow, qld, Ks, Vs, false);
    __syncthreads();
  }
  // ---------------------------------------------------------------- phase 2: dQ^T
  for (int qt = wave; qt < nt; qt += NWV) {
    const int q = qt * 16 + li;                     // this lane's query (column of S^T)
    OP qf, gf;
    if constexpr (RESTAGE) {          // this wave's query tile straight from memory, dO as staged in phase 1 (zero where rowmask == 0, x 1/(1-p))
      Frag<G> qr, gr;
      const int qc = min(q, L - 1);
      load_frag(qr, qrow + (size_t)qc * qld + 8 * lg);
      load_frag(gr, dO + (size_t)qc * P + 8 * lg);
      const float rmq = a.rowmask ? a.rowmask[(size_t)b * L + qc] : 1.f;
      if (q >= L) frag_zero(qr);
      if (q >= L || rmq == 0.f) frag_zero(gr);
      if constexpr (DM == 1) {
#pragma unroll
        for (int j = 0; j < 8; ++j) gr.v[j] = (G)((float)gr.v[j] * drop.inv_keep);
      }
      raw_to_op(qf, qr);
      raw_to_op(gf, gr);
    } else {
      load_op<PL>(qf, Qs + q * LDR + 8 * lg);
      load_op<PL>(gf, Gs + q * LDR + 8 * lg);
    }
    const float lse_q = lse2_s[q], dl_q = dl_s[q];
    const int qrel = q - 4 * lg;
    f32x4 dqt[2] = {(f32x4){0.f, 0.f, 0.f, 0.f}, (f32x4){0.f, 0.f, 0.f, 0.f}};
    const int nks = !((qmask >> qt) & 1u) ? 0 : ((CAUSAL && qt * 16 >= klo) ? min(nt / 2, qt / 2 + 1) : nt / 2);
    for (int ks = 0; ks < nks; ++ks) {      // padded query tile: dQ rows stay 0; causal: future key pairs contribute 0
      f32x4 ds[2];
      float kd[2][4] = {{1.f, 1.f, 1.f, 1.f}, {1.f, 1.f, 1.f, 1.f}};
      unsigned int km[2][4] = {{~0u, ~0u, ~0u, ~0u}, {~0u, ~0u, ~0u, ~0u}};
      if constexpr (DM == 1) {
        const unsigned int w = dmask[ks * LPK + q] >> (4 * lg);
#pragma unroll
        for (int r = 0; r < 4; ++r) { km[0][r] = rg_bitmask(w, r); km[1][r] = rg_bitmask(w, 16 + r); }
      } else if constexpr (DM == 2) {
        rg_keep4_pair(drop, (dbase + min(q, L - 1)) * lp4 + ks * 32 + 4 * lg, kd[0], kd[1]);
      }
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int k0 = ks * 32 + u * 16;
        OP kf, vf;
        load_op<PL>(kf, Ks + (k0 + li) * LDR + 8 * lg);
        load_op<PL>(vf, Vs + (k0 + li) * LDR + 8 * lg);
        float kb4[4];
        load4f(kb4, kbias + k0 + 4 * lg);
        f32x4 sv = (f32x4){kb4[0], kb4[1], kb4[2], kb4[3]}, dp = (f32x4){0.f, 0.f, 0.f, 0.f};
        mma(kf, qf, sv);           // S^T[key][q] + key bias
        mma(vf, gf, dp);           // dP^T[key][q]
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float sc = sv[r];
          if (CAUSAL) sc = (k0 + r > qrel) ? fminf(sc, MASK_BIG) : sc;
          const float pe = __builtin_amdgcn_exp2f(fmaf(sc, c2, -lse_q));
          if constexpr (DM == 1) ds[u][r] = pe * (rg_and(dp[r], km[u][r]) - dl_q);
          else if constexpr (DM == 2) ds[u][r] = pe * (dp[r] * kd[u][r] - dl_q);
          else ds[u][r] = pe * (dp[r] - dl_q);
        }
      }
      OP dsf;
      acc_to_op(dsf, ds[0], ds[1]);
#pragma unroll
      for (int dt = 0; dt < 2; ++dt) {
        OP ktf;
        vstage_op<PL>(ktf, Ks, LDR, ks * 32, dt * 16, li, lg);
        mma(ktf, dsf, dqt[dt]);    // dQ^T[dk][q] += sum_key K[key][dk] dS^T[key][q]
      }
    }
    if (q < L) {
#pragma unroll
      for (int dt = 0; dt < 2; ++dt) {
        float v[4] = {dqt[dt][0] * a.scale, dqt[dt][1] * a.scale, dqt[dt][2] * a.scale, dqt[dt][3] * a.scale};
        store4(dqkv + (size_t)q * ld + h * DK + dt * 16 + 4 * lg, v);
      }
    }
  }
  } else {
    // ---- ONEPASS: sum the four partial dQ^T through the dead operand tiles, half of the query tiles per round:
    // every wave parks its tiles ([wave][tile][dt] x 1 KB, lane-major: conflict-free 16-byte accesses), then each wave
    // sums and stores a quarter of them
    float* red = reinterpret_cast<float*>(QKVG);
    constexpr int HT = NKT / 2;                     // tiles per round
    // a query tile of padded positions only never entered a step: its partial sums are exact zeros in every wave, so it
    // is neither parked nor summed (its rows are written as zeros) -- 39 % of the tiles at the bench's lengths, and with
    // left padding the whole first round
#pragma unroll
    for (int rnd = 0; rnd < 2; ++rnd) {
      const unsigned int lm = (qmask >> (rnd * HT)) & ((1u << HT) - 1u);      // (scalar) live tiles of this round
      if (lm) {
        __syncthreads();                            // first executed round: every wave is done with Q / K / V / dO; then: readers done
#pragma unroll
        for (int i = 0; i < HT; ++i)
          if ((lm >> i) & 1u) {
#pragma unroll
            for (int dt = 0; dt < 2; ++dt)
              *reinterpret_cast<f32x4*>(red + (((wave * HT + i) * 2 + dt) * 64 + lane) * 4) = dq1[rnd * HT + i][dt];
          }
        __syncthreads();
      }
      for (int i = wave; i < HT; i += 4) {
        const int qt = rnd * HT + i, q = qt * 16 + li;
        if (qt >= nt || q >= L) continue;
        const bool livet = (lm >> i) & 1u;
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) {
          f32x4 sacc = (f32x4){0.f, 0.f, 0.f, 0.f};
          if (livet) {
            sacc = *reinterpret_cast<const f32x4*>(red + (((0 * HT + i) * 2 + dt) * 64 + lane) * 4);
#pragma unroll
            for (int w = 1; w < 4; ++w) sacc += *reinterpret_cast<const f32x4*>(red + (((w * HT + i) * 2 + dt) * 64 + lane) * 4);
          }
          float v[4] = {sacc[0] * a.scale, sacc[1] * a.scale, sacc[2] * a.scale, sacc[3] * a.scale};
          store4(dqkv + (size_t)q * ld + h * DK + dt * 16 + 4 * lg, v);
        }
      }
    }
  }
}

// the x-input (fused projection) form: bf16, d_model 128, dropout off or 0.5
static int launch_fwd_x(const rg_attn_args& a, hipStream_t s) {
  const int nkt = (a.L + 31) / 32 * 2;
  dim3 grid(rg_head_grid(a.B, a.H)), block(256);
  const int dm = a.drop_p <= 0.f ? 0 : 1;
#define RG_FWDX2(N, C)                                                                                       \
  do {                                                                                                       \
    if (dm == 0) hipLaunchKernelGGL((attn_fwd_kernel<__bf16, N, C, 0, true>), grid, block, 0, s, a);         \
    else hipLaunchKernelGGL((attn_fwd_kernel<__bf16, N, C, 1, true>), grid, block, 0, s, a);                 \
  } while (0)
#define RG_FWDX(N)                       \
  do {                                   \
    if (a.causal) RG_FWDX2(N, true);     \
    else RG_FWDX2(N, false);             \
  } while (0)
  if (nkt <= 4) RG_FWDX(4);
  else if (nkt <= 8) RG_FWDX(8);
  else if (nkt <= 14) RG_FWDX(14);
  else if (nkt <= 16) RG_FWDX(16);
  else if (nkt <= 26) RG_FWDX(26);
  else return rg_set_error_msg(RG_ERR_UNSUPPORTED, "attn_fwd: L > 416 not supported yet");
#undef RG_FWDX
#undef RG_FWDX2
  RG_CHECK_LAUNCH();
  return 0;
}

// head-major qkv (bf16): K / V tiles by LDS-DMA
static int launch_fwd_hm(const rg_attn_args& a, hipStream_t s) {
  const int nkt = (a.L + 31) / 32 * 2;
  dim3 grid(rg_head_grid(a.B, a.H)), block(256);
  const int dm = a.drop_p <= 0.f ? 0 : (a.drop_p == 0.5f ? 1 : 2);
#define RG_FWDH2(N, C)                                                                                          \
  do {                                                                                                          \
    if (dm == 0) hipLaunchKernelGGL((attn_fwd_kernel<__bf16, N, C, 0, false, true>), grid, block, 0, s, a);     \
    else if (dm == 1) hipLaunchKernelGGL((attn_fwd_kernel<__bf16, N, C, 1, false, true>), grid, block, 0, s, a); \
    else hipLaunchKernelGGL((attn_fwd_kernel<__bf16, N, C, 2, false, true>), grid, block, 0, s, a);             \
  } while (0)
#define RG_FWDH(N)                       \
  do {                                   \
    if (a.causal) RG_FWDH2(N, true);     \
    else RG_FWDH2(N, false);             \
  } while (0)
  if (nkt <= 4) RG_FWDH(4);
  else if (nkt <= 8) RG_FWDH(8);
  else if (nkt <= 14) RG_FWDH(14);
  else if (nkt <= 16) RG_FWDH(16);
  else if (nkt <= 26) RG_FWDH(26);
  else return rg_set_error_msg(RG_ERR_UNSUPPORTED, "attn_fwd: L > 416 not supported yet");
#undef RG_FWDH
#undef RG_FWDH2
  RG_CHECK_LAUNCH();
  return 0;
}

extern "C" int rg_attn_fwd_x_supported(int d, int dtype, float drop_p) {
  return dtype == RG_BF16 && d == 128 && (drop_p <= 0.f || drop_p == 0.5f);
}

template <typename T>
static int launch_fwd(const rg_attn_args& a, hipStream_t s) {
  const int nkt = (a.L + 31) / 32 * 2;
  dim3 grid(rg_head_grid(a.B, a.H)), block(256);
  const int dm = a.drop_p <= 0.f ? 0 : (a.drop_p == 0.5f ? 1 : 2);
#define RG_FWD2(N, C)                                                                              \
  do {                                                                                             \
    if (dm == 0) hipLaunchKernelGGL((attn_fwd_kernel<T, N, C, 0>), grid, block, 0, s, a);          \
    else if (dm == 1) hipLaunchKernelGGL((attn_fwd_kernel<T, N, C, 1>), grid, block, 0, s, a);     \
    else hipLaunchKernelGGL((attn_fwd_kernel<T, N, C, 2>), grid, block, 0, s, a);                  \
  } while (0)
#define RG_FWD(N)                       \
  do {                                  \
    if (a.causal) RG_FWD2(N, true);     \
    else RG_FWD2(N, false);             \
  } while (0)
  if (nkt <= 2) RG_FWD(2);
  else if (nkt <= 4) RG_FWD(4);
  else if (nkt <= 8) RG_FWD(8);
  else if (nkt <= 14) RG_FWD(14);
  else if (nkt <= 16) RG_FWD(16);
  else if (nkt <= 26) RG_FWD(26);
  else return rg_set_error_msg(RG_ERR_UNSUPPORTED, "attn_fwd: L > 416 not supported yet");
#undef RG_FWD
#undef RG_FWD2
  RG_CHECK_LAUNCH();
  return 0;
}
template <typename T>
static int launch_bwd(const rg_attn_bwd_args& a, hipStream_t s) {
  const int nkt = (a.L + 31) / 32 * 2;
  dim3 grid(rg_head_grid(a.B, a.H)), block(256);
  if constexpr (sizeof(T) == 2) {
    const int dm = a.drop_p <= 0.f ? 0 : (a.drop_p == 0.5f ? 1 : 2);
    // test hook: RG_ATTN_BWD_TWO_PHASE=1 runs the two-phase (softmax recomputed for dQ) form at every length, so that
    // tests can hold the one-pass form against it on identical inputs and seeds
    const bool two_phase = getenv("RG_ATTN_BWD_TWO_PHASE") != nullptr;
#define RG_BWD16_2(N, C)                                                                              \
  do {                                                                                                \
    if (two_phase) {                                                                                  \
      if (dm == 0) hipLaunchKernelGGL((attn_bwd_bf16_kernel<N, C, 0, false>), grid, block, 0, s, a);    \
      else if (dm == 1) hipLaunchKernelGGL((attn_bwd_bf16_kernel<N, C, 1, false>), grid, block, 0, s, a); \
      else hipLaunchKernelGGL((attn_bwd_bf16_kernel<N, C, 2, false>), grid, block, 0, s, a);            \
    } else if (dm == 0) hipLaunchKernelGGL((attn_bwd_bf16_kernel<N, C, 0, (N >= 8 && N <= 16)>), grid, block, 0, s, a);  \
    else if (dm == 1) hipLaunchKernelGGL((attn_bwd_bf16_kernel<N, C, 1, (N >= 8 && N <= 16)>), grid, block, 0, s, a);      \
    else hipLaunchKernelGGL((attn_bwd_bf16_kernel<N, C, 2, (N >= 8 && N <= 16)>), grid, block, 0, s, a);                   \
  } while (0)
#define RG_BWD16(N)                        \
  do {                                     \
    if (a.causal) RG_BWD16_2(N, true);     \
    else RG_BWD16_2(N, false);             \
  } while (0)
    if (nkt <= 2) RG_BWD16(2);
    else if (nkt <= 4) RG_BWD16(4);
    else if (nkt <= 8) RG_BWD16(8);
    else if (nkt <= 14) RG_BWD16(14);
    else if (nkt <= 16) RG_BWD16(16);
    else if (nkt <= 26) {
      // 26 key tiles: 161 KB of LDS = one workgroup per CU -- with eight waves (RG_ATTN_BWD_256=1: the four-wave form, for A/B)
      static const bool four = getenv("RG_ATTN_BWD_256") != nullptr;
      if (four) RG_BWD16(26);
      else {
        const dim3 block8(512);
#define RG_BWD26(C)                                                                                          \
  do {                                                                                                       \
    if (dm == 0) hipLaunchKernelGGL((attn_bwd_bf16_kernel<26, C, 0, false, 512>), grid, block8, 0, s, a);      \
    else if (dm == 1) hipLaunchKernelGGL((attn_bwd_bf16_kernel<26, C, 1, false, 512>), grid, block8, 0, s, a); \
    else hipLaunchKernelGGL((attn_bwd_bf16_kernel<26, C, 2, false, 512>), grid, block8, 0, s, a);              \
  } while (0)
        if (a.causal) RG_BWD26(true);
        else RG_BWD26(false);
#undef RG_BWD26
      }
    } else return rg_set_error_msg(RG_ERR_UNSUPPORTED, "attn_bwd: L > 416 not supported yet");
#undef RG_BWD16
#undef RG_BWD16_2
  } else {
    if constexpr (std::is_same<T, x3>::value) {
      // bf16x3, 128 < L <= 224 (round 5): the two-tiles-at-a-time form (RESTAGE) -- its 81.6 KB of LDS let TWO workgroups share a CU
      // (four waves per SIMD) where the four-tile form's 153 KB allow one: 18.9 -> 14.9 ms per bench-shape step
      // (profiles/r05/ab/bench_bf16x3_attention_backward_*.json); RG_ATTN_BWD_X3_NO_RESTAGE=1: the four-tile form (A/B)
      static const bool restage_short = getenv("RG_ATTN_BWD_X3_NO_RESTAGE") == nullptr;
      if (nkt > 8 && nkt <= 14 && !a.qkv_hm && restage_short) {
        const int dm = a.drop_p <= 0.f ? 0 : (a.drop_p == 0.5f ? 1 : 2);
        const dim3 block8(512);
#define RG_BWDS2(C)                                                                                                      \
  do {                                                                                                                  \
    if (dm == 0) hipLaunchKernelGGL((attn_bwd_bf16_kernel<14, C, 0, false, 512, x3, true>), grid, block8, 0, s, a);       \
    else if (dm == 1) hipLaunchKernelGGL((attn_bwd_bf16_kernel<14, C, 1, false, 512, x3, true>), grid, block8, 0, s, a);  \
    else hipLaunchKernelGGL((attn_bwd_bf16_kernel<14, C, 2, false, 512, x3, true>), grid, block8, 0, s, a);               \
  } while (0)
        if (a.causal) RG_BWDS2(true); else RG_BWDS2(false);
#undef RG_BWDS2
        RG_CHECK_LAUNCH();
        return 0;
      }
      if (nkt <= 14 && !a.qkv_hm) {            // (16 key tiles: 168 KB of split tiles -- beyond a CU's LDS; the generic form below)
        const int dm = a.drop_p <= 0.f ? 0 : (a.drop_p == 0.5f ? 1 : 2);
        const dim3 block8(512);
#define RG_BWDX2(N, C)                                                                                            \
  do {                                                                                                            \
    if (dm == 0) hipLaunchKernelGGL((attn_bwd_bf16_kernel<N, C, 0, false, 512, x3>), grid, block8, 0, s, a);        \
    else if (dm == 1) hipLaunchKernelGGL((attn_bwd_bf16_kernel<N, C, 1, false, 512, x3>), grid, block8, 0, s, a);   \
    else hipLaunchKernelGGL((attn_bwd_bf16_kernel<N, C, 2, false, 512, x3>), grid, block8, 0, s, a);                \
  } while (0)
#define RG_BWDX(N)                        \
  do {                                    \
    if (a.causal) RG_BWDX2(N, true);      \
    else RG_BWDX2(N, false);              \
  } while (0)
        if (nkt <= 2) RG_BWDX(2);
        else if (nkt <= 4) RG_BWDX(4);
        else if (nkt <= 8) RG_BWDX(8);
        else RG_BWDX(14);
#undef RG_BWDX
#undef RG_BWDX2
        RG_CHECK_LAUNCH();
        return 0;
      }
      // bf16x3, 224 < L <= 416 (round 5): the same kernel with two of the four split tiles resident at a time (RESTAGE);
      // RG_ATTN_BWD_X3_GENERIC=1: the generic two-image kernel below (A/B)
      static const bool generic_long = getenv("RG_ATTN_BWD_X3_GENERIC") != nullptr;
      if (nkt <= 26 && !a.qkv_hm && !generic_long) {
        const int dm = a.drop_p <= 0.f ? 0 : (a.drop_p == 0.5f ? 1 : 2);
        const dim3 block8(512);
#define RG_BWDR2(N, C)                                                                                                  \
  do {                                                                                                                  \
    if (dm == 0) hipLaunchKernelGGL((attn_bwd_bf16_kernel<N, C, 0, false, 512, x3, true>), grid, block8, 0, s, a);        \
    else if (dm == 1) hipLaunchKernelGGL((attn_bwd_bf16_kernel<N, C, 1, false, 512, x3, true>), grid, block8, 0, s, a);   \
    else hipLaunchKernelGGL((attn_bwd_bf16_kernel<N, C, 2, false, 512, x3, true>), grid, block8, 0, s, a);                \
  } while (0)
        if (nkt <= 16) { if (a.causal) RG_BWDR2(16, true); else RG_BWDR2(16, false); }
        else { if (a.causal) RG_BWDR2(26, true); else RG_BWDR2(26, false); }
#undef RG_BWDR2
        RG_CHECK_LAUNCH();
        return 0;
      }
    }
#define RG_BWD(N) hipLaunchKernelGGL((attn_bwd_kernel<T, N>), grid, block, 0, s, a)
    if (nkt <= 2) RG_BWD(2);
    else if (nkt <= 4) RG_BWD(4);
    else if (nkt <= 8) RG_BWD(8);
    else if (nkt <= 14) RG_BWD(14);
    else if (nkt <= 16) RG_BWD(16);
    else if (nkt <= 26) RG_BWD(26);
    else return rg_set_error_msg(RG_ERR_UNSUPPORTED, "attn_bwd: L > 416 not supported yet");
#undef RG_BWD
  }
  RG_CHECK_LAUNCH();
  return 0;
}

extern "C" int rg_attn_fwd(const rg_attn_args* a, int dtype, void* stream) {
  if (!a || a->B <= 0 || a->L <= 0 || a->H <= 0) return rg_set_error_msg(RG_ERR_INVALID, "attn_fwd: empty problem");
  if (a->dk != DK) return rg_set_error_msg(RG_ERR_UNSUPPORTED, "attn_fwd: d_k must be 32");
  if (a->x) {
    if (!rg_attn_fwd_x_supported(a->d, dtype, a->drop_p) || !a->wqkv || !a->bqkv)
      return rg_set_error_msg(RG_ERR_UNSUPPORTED, "attn_fwd: the x-input form needs bf16, d_model 128, dropout 0 or 0.5, wqkv and bqkv");
    if (a->lse) return rg_set_error_msg(RG_ERR_INVALID, "attn_fwd: the x-input form is inference only (lse must be NULL)");
    return launch_fwd_x(*a, (hipStream_t)stream);
  }
  if (a->qkv_hm) {
    if (dtype != RG_BF16 || !a->pad_rows) return rg_set_error_msg(RG_ERR_UNSUPPORTED, "attn_fwd: qkv_hm is a bf16 form and needs pad_rows");
    return launch_fwd_hm(*a, (hipStream_t)stream);
  }
  if (dtype == RG_BF16) return launch_fwd<__bf16>(*a, (hipStream_t)stream);
  if (dtype == RG_F32) return launch_fwd<float>(*a, (hipStream_t)stream);
  if (dtype == RG_X3) return launch_fwd<x3>(*a, (hipStream_t)stream);
  return rg_set_error_msg(RG_ERR_INVALID, "attn_fwd: bad dtype");
}
extern "C" int rg_attn_bwd(const rg_attn_bwd_args* a, int dtype, void* stream) {
  if (!a || a->B <= 0 || a->L <= 0 || a->H <= 0) return rg_set_error_msg(RG_ERR_INVALID, "attn_bwd: empty problem");
  if (a->dk != DK) return rg_set_error_msg(RG_ERR_UNSUPPORTED, "attn_bwd: d_k must be 32");
  if (a->qkv_hm && dtype != RG_BF16) return rg_set_error_msg(RG_ERR_UNSUPPORTED, "attn_bwd: qkv_hm is a bf16 form");
  if (dtype == RG_BF16) return launch_bwd<__bf16>(*a, (hipStream_t)stream);
  if (dtype != RG_BF16 && a->x_masked == 2)
    return rg_set_error_msg(RG_ERR_UNSUPPORTED, "attn_bwd: bias-row substitution (x_masked == 2) is a bf16-tier form");
  if (dtype == RG_F32) return launch_bwd<float>(*a, (hipStream_t)stream);
  if (dtype == RG_X3) return launch_bwd<x3>(*a, (hipStream_t)stream);
  return rg_set_error_msg(RG_ERR_INVALID, "attn_bwd: bad dtype");
}
