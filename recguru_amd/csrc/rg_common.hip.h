// Common device helpers for the RecGURU gfx950 kernels.
//
// One tile vocabulary for the three precision tiers:
//   * T = __bf16 : v_mfma_f32_16x16x32_bf16 (perf tier; operands bf16, accumulation f32)
//   * T = float  : 8 x v_mfma_f32_16x16x4_f32 (parity tier; bit-exact f32 fma chain)
//   * T = x3     : f32 in memory and in every elementwise step, but each MFMA operand is split into a bf16 pair
//                  hi + lo (~16 significant bits) and a product is THREE bf16 MFMAs, lo.hi + hi.lo + hi.hi, accumulated
//                  in f32 (the dropped lo.lo term is 2^-16 of the product): 48 matrix-pipe cycles per k-step against 256
//                  for the exact-f32 form, error ~1e-5 of the operand instead of bf16's 4e-3 (tests/emulate_tiers.py).
// A "k-step" is always 32 contraction elements.  Within a k-step lane l = 16*g + i owns the 8
// slots (g, j), j = 0..7 of row/column i.  Which physical k a slot means is up to the caller as
// long as the A and the B fragment agree:
//   * contiguous mapping : slot (g, j) <-> k = 8*g + j                    (load_frag)
//   * stacked-accumulator mapping: slot (g, j) <-> k = 16*(j>>2) + 4*g + (j&3), which is what
//     two vertically stacked 16x16 accumulator tiles already hold (rows 4g+r of tile j>>2), so an
//     accumulator can be fed back as an operand with no data movement (acc_to_frag, load_frag_2x4).
// Accumulator tile map (both MFMAs): reg r of lane l holds D[row 4*(l>>4) + r][col l & 15].
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4_t;

#define RG_F32 0
#define RG_BF16 1
#define RG_X3 2
#define RG_WAVE 64

// x3: a float in memory (same size, same layout: any f32 buffer is an x3 buffer) -- a distinct type only so that the kernel
// templates get a third instantiation whose mma() is the split-operand product.
struct x3 {
  float f;
  x3() = default;
  __host__ __device__ __forceinline__ x3(float v) : f(v) {}
  __host__ __device__ __forceinline__ operator float() const { return f; }
};
static_assert(sizeof(x3) == 4 && alignof(x3) == 4, "x3 is layout-identical to float");
__device__ __forceinline__ const float* fptr(const x3* p) { return reinterpret_cast<const float*>(p); }
__device__ __forceinline__ float* fptr(x3* p) { return reinterpret_cast<float*>(p); }

template <typename T> struct Frag;
template <> struct Frag<float> { float v[8]; };
template <> struct Frag<__bf16> { bf16x8_t v; };
template <> struct Frag<x3> { float v[8]; };

// hi = bf16(x) and lo = bf16(x - hi), both round-to-nearest, two elements per step: v_cvt_pk_bf16_f32, shift / and back to
// f32, v_pk_add_f32 (exact: x - hi has at most 16 significant bits), v_cvt_pk_bf16_f32 -- 2.5 VALU operations per element.
// The compiler shares the split of a fragment between all the mma() calls that use it (common subexpressions of one
// scope), so a kernel pays it once per fragment it loads.
typedef __attribute__((ext_vector_type(2))) float rg_f2;
typedef __attribute__((ext_vector_type(2))) __bf16 rg_bf2;
__device__ __forceinline__ void split_x3(const float (&v)[8], bf16x8_t& hi, bf16x8_t& lo) {
#pragma unroll
  for (int j = 0; j < 8; j += 2) {
    const rg_f2 x = (rg_f2){v[j], v[j + 1]};
    const rg_bf2 h = __builtin_convertvector(x, rg_bf2);
    union { rg_bf2 b; unsigned int u; } hu;
    hu.b = h;
    const rg_f2 hf = (rg_f2){__uint_as_float(hu.u << 16), __uint_as_float(hu.u & 0xFFFF0000u)};
    const rg_bf2 l = __builtin_convertvector(x - hf, rg_bf2);
    hi[j] = h[0]; hi[j + 1] = h[1];
    lo[j] = l[0]; lo[j + 1] = l[1];
  }
}

__device__ __forceinline__ void mma(const Frag<__bf16>& a, const Frag<__bf16>& b, f32x4& c) {
  c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.v, b.v, c, 0, 0, 0);
}
__device__ __forceinline__ void mma(const Frag<float>& a, const Frag<float>& b, f32x4& c) {
#pragma unroll
  for (int j = 0; j < 8; ++j) c = __builtin_amdgcn_mfma_f32_16x16x4f32(a.v[j], b.v[j], c, 0, 0, 0);
}

__device__ __forceinline__ void mma(const Frag<x3>& a, const Frag<x3>& b, f32x4& c) {
  bf16x8_t ah, al, bh, bl;
  split_x3(a.v, ah, al);
  split_x3(b.v, bh, bl);
  c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, bh, c, 0, 0, 0);      // small terms first
  c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bl, c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bh, c, 0, 0, 0);
}

template <typename T> __device__ __forceinline__ void frag_zero(Frag<T>& f);
template <> __device__ __forceinline__ void frag_zero<x3>(Frag<x3>& f) {
#pragma unroll
  for (int j = 0; j < 8; ++j) f.v[j] = 0.f;
}
template <> __device__ __forceinline__ void frag_zero<float>(Frag<float>& f) {
#pragma unroll
  for (int j = 0; j < 8; ++j) f.v[j] = 0.f;
}
template <> __device__ __forceinline__ void frag_zero<__bf16>(Frag<__bf16>& f) {
#pragma unroll
  for (int j = 0; j < 8; ++j) f.v[j] = (__bf16)0.f;
}
template <typename T> __device__ __forceinline__ void frag_fill(Frag<T>& f, float x) {
#pragma unroll
  for (int j = 0; j < 8; ++j) f.v[j] = (T)x;
}

// 8 contiguous elements starting at p (16-byte aligned for bf16, 32-byte for f32).
__device__ __forceinline__ void load_frag(Frag<__bf16>& f, const __bf16* p) {
  f.v = *reinterpret_cast<const bf16x8_t*>(p);
}
__device__ __forceinline__ void load_frag(Frag<float>& f, const float* p) {
  const float4 a = *reinterpret_cast<const float4*>(p);
  const float4 b = *reinterpret_cast<const float4*>(p + 4);
  f.v[0] = a.x; f.v[1] = a.y; f.v[2] = a.z; f.v[3] = a.w;
  f.v[4] = b.x; f.v[5] = b.y; f.v[6] = b.z; f.v[7] = b.w;
}
__device__ __forceinline__ void load_frag(Frag<x3>& f, const x3* p) {
  const float4 a = *reinterpret_cast<const float4*>(p);
  const float4 b = *reinterpret_cast<const float4*>(fptr(p) + 4);
  f.v[0] = a.x; f.v[1] = a.y; f.v[2] = a.z; f.v[3] = a.w;
  f.v[4] = b.x; f.v[5] = b.y; f.v[6] = b.z; f.v[7] = b.w;
}
// two runs of 4 contiguous elements: slots j<4 from p0, j>=4 from p1 (stacked-accumulator mapping)
__device__ __forceinline__ void load_frag_2x4(Frag<__bf16>& f, const __bf16* p0, const __bf16* p1) {
  const bf16x4_t a = *reinterpret_cast<const bf16x4_t*>(p0);
  const bf16x4_t b = *reinterpret_cast<const bf16x4_t*>(p1);
  f.v[0] = a[0]; f.v[1] = a[1]; f.v[2] = a[2]; f.v[3] = a[3];
  f.v[4] = b[0]; f.v[5] = b[1]; f.v[6] = b[2]; f.v[7] = b[3];
}
__device__ __forceinline__ void load_frag_2x4(Frag<float>& f, const float* p0, const float* p1) {
  const float4 a = *reinterpret_cast<const float4*>(p0);
  const float4 b = *reinterpret_cast<const float4*>(p1);
  f.v[0] = a.x; f.v[1] = a.y; f.v[2] = a.z; f.v[3] = a.w;
  f.v[4] = b.x; f.v[5] = b.y; f.v[6] = b.z; f.v[7] = b.w;
}
__device__ __forceinline__ void load_frag_2x4(Frag<x3>& f, const x3* p0, const x3* p1) {
  const float4 a = *reinterpret_cast<const float4*>(p0);
  const float4 b = *reinterpret_cast<const float4*>(p1);
  f.v[0] = a.x; f.v[1] = a.y; f.v[2] = a.z; f.v[3] = a.w;
  f.v[4] = b.x; f.v[5] = b.y; f.v[6] = b.z; f.v[7] = b.w;
}
// two stacked accumulator tiles (rows 0..15 = lo, 16..31 = hi of a 32-deep k-step) -> operand
template <typename T>
__device__ __forceinline__ void acc_to_frag(Frag<T>& f, const f32x4& lo, const f32x4& hi) {
#pragma unroll
  for (int j = 0; j < 4; ++j) { f.v[j] = (T)lo[j]; f.v[4 + j] = (T)hi[j]; }
}

// 8 contiguous elements <-> float[8]
__device__ __forceinline__ void load8(float* o, const float* p) {
  const float4 a = *reinterpret_cast<const float4*>(p);
  const float4 b = *reinterpret_cast<const float4*>(p + 4);
  o[0] = a.x; o[1] = a.y; o[2] = a.z; o[3] = a.w; o[4] = b.x; o[5] = b.y; o[6] = b.z; o[7] = b.w;
}
__device__ __forceinline__ void load8(float* o, const __bf16* p) {
  const bf16x8_t a = *reinterpret_cast<const bf16x8_t*>(p);
#pragma unroll
  for (int j = 0; j < 8; ++j) o[j] = (float)a[j];
}
__device__ __forceinline__ void load8(float* o, const x3* p) { load8(o, fptr(p)); }
__device__ __forceinline__ void store8(float* p, const float* v) {
  *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
  *reinterpret_cast<float4*>(p + 4) = make_float4(v[4], v[5], v[6], v[7]);
}
__device__ __forceinline__ void store8(__bf16* p, const float* v) {
  bf16x8_t a;
#pragma unroll
  for (int j = 0; j < 8; ++j) a[j] = (__bf16)v[j];
  *reinterpret_cast<bf16x8_t*>(p) = a;
}
__device__ __forceinline__ void store8(x3* p, const float* v) {
  *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
  *reinterpret_cast<float4*>(fptr(p) + 4) = make_float4(v[4], v[5], v[6], v[7]);
}
__device__ __forceinline__ void load4f(float* o, const float* p) {
  const float4 a = *reinterpret_cast<const float4*>(p);
  o[0] = a.x; o[1] = a.y; o[2] = a.z; o[3] = a.w;
}
__device__ __forceinline__ void store4(float* p, const float* v) {
  *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
}
__device__ __forceinline__ void store4(x3* p, const float* v) {
  *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
}
__device__ __forceinline__ void store4(__bf16* p, const float* v) {
  bf16x4_t a;
#pragma unroll
  for (int j = 0; j < 4; ++j) a[j] = (__bf16)v[j];
  *reinterpret_cast<bf16x4_t*>(p) = a;
}

// Nontemporal 16-byte store of a fragment: for activations SAVED for the backward pass (read tens of milliseconds
// later) -- written with plain stores they evict the layer output the next kernel is about to read from the
// Infinity Cache.
typedef __attribute__((ext_vector_type(4))) float rg_f4;
__device__ __forceinline__ void frag_store_nt(float* p, const Frag<float>& f) {
  __builtin_nontemporal_store((rg_f4){f.v[0], f.v[1], f.v[2], f.v[3]}, reinterpret_cast<rg_f4*>(p));
  __builtin_nontemporal_store((rg_f4){f.v[4], f.v[5], f.v[6], f.v[7]}, reinterpret_cast<rg_f4*>(p + 4));
}
__device__ __forceinline__ void frag_store_nt(x3* p, const Frag<x3>& f) {
  __builtin_nontemporal_store((rg_f4){f.v[0], f.v[1], f.v[2], f.v[3]}, reinterpret_cast<rg_f4*>(p));
  __builtin_nontemporal_store((rg_f4){f.v[4], f.v[5], f.v[6], f.v[7]}, reinterpret_cast<rg_f4*>(fptr(p) + 4));
}
__device__ __forceinline__ void frag_store_nt(__bf16* p, const Frag<__bf16>& f) {
  union { bf16x8_t b; rg_f4 x; } u;
  u.b = f.v;
  __builtin_nontemporal_store(u.x, reinterpret_cast<rg_f4*>(p));
}

__device__ __forceinline__ float wave_sum(float x) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) x += __shfl_xor(x, o);
  return x;
}
__device__ __forceinline__ float wave_max(float x) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) x = fmaxf(x, __shfl_xor(x, o));
  return x;
}

// tanh-approximation GELU (reference transformer.py:81-84) and its derivative.
// PRECISE (f32 parity tier): libm tanhf.  Fast (bf16 tier): 0.5*(1+tanh u) == sigmoid(2u), one v_exp.
template <bool PRECISE>
__device__ __forceinline__ float gelu_t(float x) {
  const float c = 0.7978845608028654f;
  if (PRECISE) {
    const float u = c * (x + 0.044715f * x * x * x);
    return 0.5f * x * (1.f + tanhf(u));
  }
  // x * sigmoid(2u) with every constant folded: 7 VALU (mul, fma, mul, exp2, add, rcp, mul)
  const float k1 = -2.f * 1.4426950408889634f * c;
  const float k3 = k1 * 0.044715f;
  const float e = __builtin_amdgcn_exp2f(x * fmaf(x * x, k3, k1));
  return x * __builtin_amdgcn_rcpf(1.f + e);
}
// two elements at a time, fast tier: the five simple operations as packed v_pk_* instructions (one issue slot for
// both elements), exp2 / rcp per element -- same arithmetic as gelu_t<false>
typedef __attribute__((ext_vector_type(2))) float f32x2;
__device__ __forceinline__ f32x2 gelu2_fast(f32x2 x) {
  const float c = 0.7978845608028654f;
  const float k1 = -2.f * 1.4426950408889634f * c;
  const float k3 = k1 * 0.044715f;
  const f32x2 t = x * x * (f32x2){k3, k3} + (f32x2){k1, k1};
  const f32x2 arg = x * t;
  f32x2 e;
  e.x = __builtin_amdgcn_exp2f(arg.x);
  e.y = __builtin_amdgcn_exp2f(arg.y);
  const f32x2 den = e + (f32x2){1.f, 1.f};
  f32x2 r;
  r.x = __builtin_amdgcn_rcpf(den.x);
  r.y = __builtin_amdgcn_rcpf(den.y);
  return x * r;
}
// gelu2_fast in two halves (the pipelined FFN loop of fused.hip puts a matrix instruction between them): the exponentials, then the rest
__device__ __forceinline__ f32x2 gelu2_fast_a(f32x2 x) {
  const float c = 0.7978845608028654f;
  const float k1 = -2.f * 1.4426950408889634f * c;
  const float k3 = k1 * 0.044715f;
  const f32x2 t = x * x * (f32x2){k3, k3} + (f32x2){k1, k1};
  const f32x2 arg = x * t;
  f32x2 e;
  e.x = __builtin_amdgcn_exp2f(arg.x);
  e.y = __builtin_amdgcn_exp2f(arg.y);
  return e;
}
__device__ __forceinline__ f32x2 gelu2_fast_b(f32x2 x, f32x2 e) {
  const f32x2 den = e + (f32x2){1.f, 1.f};
  f32x2 r;
  r.x = __builtin_amdgcn_rcpf(den.x);
  r.y = __builtin_amdgcn_rcpf(den.y);
  return x * r;
}
template <bool PRECISE>
__device__ __forceinline__ float gelu_grad_t(float x) {
  const float c = 0.7978845608028654f;
  const float x2 = x * x;
  if (PRECISE) {
    const float u = c * (x + 0.044715f * x * x2);
    const float t = tanhf(u);
    return 0.5f * (1.f + t) + 0.5f * x * (1.f - t * t) * c * (1.f + 3.f * 0.044715f * x2);
  }
  // s = sigmoid(2u); d/dx [x s] = s + x s (1 - s) * 2 u'  with u' = c (1 + 3*0.044715 x^2)
  const float k1 = -2.f * 1.4426950408889634f * c;
  const float k3 = k1 * 0.044715f;
  const float sg = __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(x * fmaf(x2, k3, k1)));
  return fmaf(x * sg * (1.f - sg), fmaf(x2, 6.f * 0.044715f * c, 2.f * c), sg);
}
__device__ __forceinline__ float gelu_f(float x) { return gelu_t<true>(x); }
__device__ __forceinline__ float gelu_grad_f(float x) { return gelu_grad_t<true>(x); }
// 4 consecutive elements -> float[4]
__device__ __forceinline__ void load4t(float* o, const float* p) { load4f(o, p); }
__device__ __forceinline__ void load4t(float* o, const x3* p) { load4f(o, fptr(p)); }
__device__ __forceinline__ void load4t(float* o, const __bf16* p) {
  const bf16x4_t a = *reinterpret_cast<const bf16x4_t*>(p);
#pragma unroll
  for (int j = 0; j < 4; ++j) o[j] = (float)a[j];
}

template <typename T> struct Precise { static constexpr bool value = true; };
template <> struct Precise<__bf16> { static constexpr bool value = false; };
// x3: the 7-instruction exp2 / rcp GELU (v_exp_f32 and v_rcp_f32 are good to ~1 ulp: 1e-7, two orders below the operand split)
template <> struct Precise<x3> { static constexpr bool value = false; };

// ---- bf16x3 with operands split ONCE (the kernels that stage their operands through LDS) --------------------------------
// An LDS tile of the x3 tier is a PAIR of bf16 tiles, hi and lo, laid out exactly like the bf16 tier's tile (same offsets, same
// swizzles, same bank behaviour); the lo tile sits PL bytes behind the hi tile (a template argument of the tile's address
// type x3p<PL>, so that kernels with different tile sizes share these helpers).  An element is split when it is WRITTEN to the tile (staging from HBM, an accumulator leaving
// the registers) -- once -- and every wave's operand-fragment read is two plain 16-byte LDS reads with no VALU work behind
// them; weights come from the fragment-packed presplit copy rg_cast writes (hi fragment, then lo fragment, 1 KB each).
template <int PL> struct x3p { unsigned short u; };     // one bf16 slot of a split tile whose lo image sits PL BYTES behind the hi image
struct FragX3 { bf16x8_t hi, lo; };                    // operand fragment, already split
template <typename T, int PL = 0> struct LdsT { typedef T type; static constexpr int PLANES = 1; };
template <int PL> struct LdsT<x3, PL> { typedef x3p<PL> type; static constexpr int PLANES = 2; };
template <typename T> struct OpT { typedef Frag<T> type; };
template <> struct OpT<x3> { typedef FragX3 type; };
__device__ __forceinline__ void mma(const FragX3& a, const FragX3& b, f32x4& c) {
  c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.lo, b.hi, c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.hi, b.lo, c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.hi, b.hi, c, 0, 0, 0);
}
// operand fragment straight from f32 memory (split on the way: for operands a workgroup loads once)
__device__ __forceinline__ void load_frag(FragX3& f, const x3* p) {
  float v[8];
  load8(v, reinterpret_cast<const float*>(p));
  split_x3(v, f.hi, f.lo);
}
template <int PL> __device__ __forceinline__ const x3p<PL>* lo_of(const x3p<PL>* p) { return reinterpret_cast<const x3p<PL>*>(reinterpret_cast<const char*>(p) + PL); }
template <int PL> __device__ __forceinline__ x3p<PL>* lo_of(x3p<PL>* p) { return reinterpret_cast<x3p<PL>*>(reinterpret_cast<char*>(p) + PL); }
template <int PL> __device__ __forceinline__ void load_frag(FragX3& f, const x3p<PL>* p) {
  f.hi = *reinterpret_cast<const bf16x8_t*>(p);
  f.lo = *reinterpret_cast<const bf16x8_t*>(lo_of(p));
}
// 8 raw elements (as loaded from HBM) -> the tile
template <int PL> __device__ __forceinline__ void stage8(x3p<PL>* dst, const Frag<x3>& raw) {
  bf16x8_t hi, lo;
  split_x3(raw.v, hi, lo);
  *reinterpret_cast<bf16x8_t*>(dst) = hi;
  *reinterpret_cast<bf16x8_t*>(lo_of(dst)) = lo;
}
// ... and back: hi + lo (exact in f32: the two parts do not overlap)
template <int PL> __device__ __forceinline__ void unstage8(Frag<x3>& raw, const x3p<PL>* src) {
  const bf16x8_t hi = *reinterpret_cast<const bf16x8_t*>(src), lo = *reinterpret_cast<const bf16x8_t*>(lo_of(src));
#pragma unroll
  for (int j = 0; j < 8; ++j) raw.v[j] = (float)hi[j] + (float)lo[j];
}
template <int PL> __device__ __forceinline__ void store8(x3p<PL>* p, const float* v) {
  Frag<x3> r;
#pragma unroll
  for (int j = 0; j < 8; ++j) r.v[j] = v[j];
  stage8(p, r);
}
template <int PL> __device__ __forceinline__ void load8(float* o, const x3p<PL>* p) {
  Frag<x3> r;
  unstage8(r, p);
#pragma unroll
  for (int j = 0; j < 8; ++j) o[j] = r.v[j];
}
template <int PL> __device__ __forceinline__ void store4(x3p<PL>* p, const float* v) {
  bf16x4_t hi, lo;
#pragma unroll
  for (int j = 0; j < 4; j += 2) {
    const rg_f2 x = (rg_f2){v[j], v[j + 1]};
    const rg_bf2 h = __builtin_convertvector(x, rg_bf2);
    union { rg_bf2 b; unsigned int u; } hu;
    hu.b = h;
    const rg_f2 hf = (rg_f2){__uint_as_float(hu.u << 16), __uint_as_float(hu.u & 0xFFFF0000u)};
    const rg_bf2 l = __builtin_convertvector(x - hf, rg_bf2);
    hi[j] = h[0]; hi[j + 1] = h[1];
    lo[j] = l[0]; lo[j + 1] = l[1];
  }
  *reinterpret_cast<bf16x4_t*>(p) = hi;
  *reinterpret_cast<bf16x4_t*>(lo_of(p)) = lo;
}
template <int PL> __device__ __forceinline__ void load4t(float* o, const x3p<PL>* p) {
  const bf16x4_t hi = *reinterpret_cast<const bf16x4_t*>(p), lo = *reinterpret_cast<const bf16x4_t*>(lo_of(p));
#pragma unroll
  for (int j = 0; j < 4; ++j) o[j] = (float)hi[j] + (float)lo[j];
}
// x3r: a raw f32 slot of the x3 tier -- for tiles that are NOT matrix operands (residuals, outputs on their way to HBM):
// splitting those would cost VALU work and round a residual to 16 bits for nothing.  A [64 x 128] tile of them is exactly as
// large as the split pair, so it can take the place of one.
struct x3r { float f; };
template <typename T, int PL = 0> struct ResT { typedef typename LdsT<T, PL>::type type; };
template <int PL> struct ResT<x3, PL> { typedef x3r type; };
__device__ __forceinline__ void stage8(x3r* dst, const Frag<x3>& raw) {
  *reinterpret_cast<float4*>(dst) = make_float4(raw.v[0], raw.v[1], raw.v[2], raw.v[3]);
  *reinterpret_cast<float4*>(dst + 4) = make_float4(raw.v[4], raw.v[5], raw.v[6], raw.v[7]);
}
__device__ __forceinline__ void unstage8(Frag<x3>& raw, const x3r* src) { load8(raw.v, reinterpret_cast<const float*>(src)); }
__device__ __forceinline__ void store8(x3r* p, const float* v) { store8(reinterpret_cast<float*>(p), v); }
__device__ __forceinline__ void load8(float* o, const x3r* p) { load8(o, reinterpret_cast<const float*>(p)); }
__device__ __forceinline__ void store4(x3r* p, const float* v) { store4(reinterpret_cast<float*>(p), v); }
__device__ __forceinline__ void load4t(float* o, const x3r* p) { load4f(o, reinterpret_cast<const float*>(p)); }
// the same two for the tiers whose tile holds the raw element
template <typename T> __device__ __forceinline__ void stage8(T* dst, const Frag<T>& raw) { *reinterpret_cast<Frag<T>*>(dst) = raw; }
template <typename T> __device__ __forceinline__ void unstage8(Frag<T>& raw, const T* src) { raw = *reinterpret_cast<const Frag<T>*>(src); }

// Dropout masks are stateless: keep(seed, idx) is a hash of a per-call seed and the element index,
// so the backward kernels regenerate exactly the forward's mask (nn.Dropout semantics: element kept
// with probability 1-p and scaled by 1/(1-p); only the random stream differs from torch's Philox).
struct DropCfg {
  unsigned int seed;         // folded 64-bit call seed
  unsigned int thresh;       // 16-bit mode: drop iff field < thresh (= p * 65536); 0 = dropout off
  float inv_keep;            // 1 / (1 - p)
  unsigned int onebit;       // p == 0.5 exactly: one hash BIT per element (32 elements per hash)
};
// Element idx (a 32-bit, wrapping index in a per-site index space) is decided
//   * p == 0.5 : by bit (idx & 31) of hash(idx >> 5) -- a 2-multiply lowbias32 hash serves 32 elements,
//   * otherwise: by the (idx & 1)-th 16-bit half of hash(idx >> 1) compared with p * 65536.
// The group helpers (rg_keep4 / rg_keep8 / rg_keep4_pair) return the same decisions as rg_keep()
// element by element; they only share the hash words.
__device__ __forceinline__ unsigned int rg_hash(unsigned int seed, unsigned int x) {
  x ^= seed;
  x ^= x >> 16; x *= 0x21f0aaadu;
  x ^= x >> 15; x *= 0x735a2d97u;
  x ^= x >> 15;
  return x;
}
// Every lane (column li of row r of the wave's four 16-lane rows) receives the values that the four lanes (li, row 0 .. 3) hold: one
// v_permlane32_swap and two v_permlane16_swap (gfx950; order checked on hardware by tools/permlane_probe.hip).  Used to compute ONE
// dropout hash word per lane where every lane needs the words of four row tiles: a hash is 6 simple + 2 v_mul_lo_u32 instructions.
__device__ __forceinline__ void rg_allgather_rows(unsigned int v, unsigned int (&w)[4]) {
  typedef __attribute__((ext_vector_type(2))) unsigned int u2_t;
  const u2_t pq = __builtin_amdgcn_permlane32_swap(v, v, false, false);
  const u2_t a = __builtin_amdgcn_permlane16_swap(pq.x, pq.x, false, false);
  const u2_t b = __builtin_amdgcn_permlane16_swap(pq.y, pq.y, false, false);
  w[0] = a.x; w[1] = a.y; w[2] = b.x; w[3] = b.y;
}
__device__ __forceinline__ float rg_bit(const DropCfg& c, unsigned int w, int j) { return (w >> j) & 1u ? c.inv_keep : 0.f; }
// bit j of w as an AND mask (all ones = keep) and its use on a float: 2 VALU ops per element, scale applied elsewhere
__device__ __forceinline__ unsigned int rg_bitmask(unsigned int w, int j) { return (unsigned int)__builtin_amdgcn_sbfe((int)w, j, 1); }
__device__ __forceinline__ float rg_and(float x, unsigned int m) { return __uint_as_float(__float_as_uint(x) & m); }
__device__ __forceinline__ float rg_field(const DropCfg& c, unsigned int f16) { return f16 < c.thresh ? 0.f : c.inv_keep; }
__device__ __forceinline__ float rg_keep(const DropCfg& c, unsigned int idx) {
  if (c.onebit) return (rg_hash(c.seed, idx >> 5) >> (idx & 31u)) & 1u ? c.inv_keep : 0.f;
  const unsigned int h = rg_hash(c.seed, idx >> 1);
  return rg_field(c, (idx & 1u) ? (h >> 16) : (h & 0xFFFFu));
}
// base % 4 == 0
__device__ __forceinline__ void rg_keep4(const DropCfg& c, unsigned int base, float (&k)[4]) {
  if (c.onebit) {
    const unsigned int w = rg_hash(c.seed, base >> 5) >> (base & 31u);
#pragma unroll
    for (int j = 0; j < 4; ++j) k[j] = rg_bit(c, w, j);
  } else {
    const unsigned int h0 = rg_hash(c.seed, base >> 1), h1 = rg_hash(c.seed, (base >> 1) + 1u);
    k[0] = rg_field(c, h0 & 0xFFFFu); k[1] = rg_field(c, h0 >> 16);
    k[2] = rg_field(c, h1 & 0xFFFFu); k[3] = rg_field(c, h1 >> 16);
  }
}
// base % 8 == 0
__device__ __forceinline__ void rg_keep8(const DropCfg& c, unsigned int base, float (&k)[8]) {
  if (c.onebit) {
    const unsigned int w = rg_hash(c.seed, base >> 5) >> (base & 31u);
#pragma unroll
    for (int j = 0; j < 8; ++j) k[j] = rg_bit(c, w, j);
  } else {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const unsigned int h = rg_hash(c.seed, (base >> 1) + j);
      k[2 * j] = rg_field(c, h & 0xFFFFu); k[2 * j + 1] = rg_field(c, h >> 16);
    }
  }
}
// elements base..base+3 and base+16..base+19, base % 4 == 0 and (base & 31) < 16 (both groups in one hash word)
__device__ __forceinline__ void rg_keep4_pair(const DropCfg& c, unsigned int base, float (&k0)[4], float (&k1)[4]) {
  if (c.onebit) {
    const unsigned int w = rg_hash(c.seed, base >> 5) >> (base & 31u);
#pragma unroll
    for (int j = 0; j < 4; ++j) { k0[j] = rg_bit(c, w, j); k1[j] = rg_bit(c, w, 16 + j); }
  } else {
    rg_keep4(c, base, k0);
    rg_keep4(c, base + 16u, k1);
  }
}
__host__ __device__ inline DropCfg make_drop(float p, unsigned long long seed) {
  DropCfg c;
  unsigned int s = (unsigned int)seed * 0x9E3779B1u ^ ((unsigned int)(seed >> 32) * 0x85EBCA77u + 0x165667B1u);
  s ^= s >> 15; s *= 0x2c1b3c6du; s ^= s >> 12;
  c.seed = s;
  c.thresh = p > 0.f ? (unsigned int)((double)p * 65536.0 + 0.5) : 0u;
  c.inv_keep = p > 0.f ? 1.f / (1.f - p) : 1.f;
  c.onebit = p == 0.5f ? 1u : 0u;
  return c;
}
// attention-map index space: ((b*H + h)*L + q) * LPAD + key, LPAD = L rounded up to a multiple of 32
// (a row starts on a hash-word boundary)
__host__ __device__ inline unsigned int rg_lpad(int L) { return (unsigned int)((L + 31) & ~31); }

// XCD-aware (b, h) assignment for one-workgroup-per-head kernels.  Workgroups are dealt round-robin to the 8 XCDs,
// each with its own L2; heads 2p and 2p+1 of a sequence share every 128-byte line of the [.., H*32] bf16 rows
// (64 bytes each), so with blockIdx = b*H + h the two halves of every line are fetched by two different L2s
// (rocprofv3 FETCH_SIZE: 2x the algorithmic bytes).  Here consecutive ROUNDS of the same XCD (blockIdx and
// blockIdx + 8) take the two heads of a pair, so the second one hits the lines the first brought in.  Only speed
// depends on the dispatch order.  Returns false for the padding blocks of the grid (rg_head_grid).
__device__ __forceinline__ bool rg_head_of_block(int bid, int B, int H, int& b, int& h) {
  const int hp = (H + 1) >> 1;
  const int xcd = bid & 7, rnd = bid >> 3;
  const int slot = (rnd >> 1) * 8 + xcd;
  b = slot / hp;
  h = 2 * (slot - b * hp) + (rnd & 1);
  return b < B && h < H;
}
static inline int rg_head_grid(int B, int H) {
  const int slots = B * ((H + 1) >> 1);
  return ((slots + 7) / 8) * 16;
}

// Workgroup barrier that orders LDS traffic only.  __syncthreads() carries a workgroup-scope release
// fence, for which hipcc drains vmcnt(0) whenever a global store is outstanding -- and loads share that
// counter, so every software-prefetched global load would be waited for at every barrier.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// The work tiles of a persistent workgroup (k-th tile = blockIdx.x + k * gridDim.x) under a live-tile list
// (rg_live_tiles): a work tile is 4 consecutive list entries.  The entries of 16 consecutive k's sit in ONE register
// (lane 4 (k & 15) + rt), fetched by a single unconditional vector load per 16 tiles and handed out with v_readlane.
// Read from memory tile by tile -- a dependent load and a wait per entry, behind per-entry branches -- the list cost five
// L2 round trips per tile, each behind a vmcnt(0) that also drained the kernel's prefetches and waited for its stores.
struct LiveWalk {
  const int* list;
  int nlive, cap, v, blk;
  int wg, nwg;                          // this workgroup's index among the walkers and their number (default: blockIdx.x of gridDim.x)
  __device__ __forceinline__ void init(const int* live16, int M) { init(live16, M, (int)blockIdx.x, (int)gridDim.x); }
  __device__ __forceinline__ void init(const int* live16, int M, int wg_, int nwg_) {
    wg = wg_;
    nwg = nwg_;
    list = live16;
    nlive = live16 ? live16[0] : 0;
    cap = ((M + 15) >> 4) - 1;          // last valid entry index (entries past nlive are never used)
    v = 0;
    blk = -1;
  }
  // first rows of the 4 row tiles of work tile `wt` = blockIdx.x + k * gridDim.x (>= M: absent)
  __device__ __forceinline__ void group(int k, int (&g)[4], int M) {
    const int wt = wg + k * nwg;
    if ((k >> 4) != blk) {              // (uniform) next block of 16 tiles
      blk = k >> 4;
      const int lane = threadIdx.x & 63;
      const long long idx = 4ll * ((long long)wg + (long long)nwg * (16 * blk + (lane >> 2))) + (lane & 3);
      v = list[1 + (int)(idx < (long long)cap ? idx : (long long)cap)];
    }
#pragma unroll
    for (int rt = 0; rt < 4; ++rt) {
      const int e = __builtin_amdgcn_readlane(v, 4 * (k & 15) + rt);
      g[rt] = (4 * wt + rt < nlive) ? e * 16 : M;
    }
  }
  // the same for work tiles of RT row tiles (RT = 1, 2, 4): one vector load serves 64 / RT work tiles
  template <int RT>
  __device__ __forceinline__ void group_n(int k, int (&g)[RT], int M) {
    constexpr int PER = 64 / RT, SH = RT == 4 ? 4 : (RT == 2 ? 5 : 6);
    const int wt = wg + k * nwg;
    if ((k >> SH) != blk) {             // (uniform) next block of PER work tiles
      blk = k >> SH;
      const int lane = threadIdx.x & 63;
      const long long idx = (long long)RT * ((long long)wg + (long long)nwg * (PER * blk + lane / RT)) + (lane % RT);
      v = list[1 + (int)(idx < (long long)cap ? idx : (long long)cap)];
    }
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
      const int e = __builtin_amdgcn_readlane(v, RT * (k & (PER - 1)) + rt);
      g[rt] = (RT * wt + rt < nlive) ? e * 16 : M;
    }
  }
};

#define RG_CHECK_LAUNCH()                                   \
  do {                                                      \
    hipError_t e__ = hipGetLastError();                     \
    if (e__ != hipSuccess) return rg_set_error(e__, __func__); \
  } while (0)

int rg_set_error(hipError_t e, const char* where);
int rg_set_error_msg(int code, const char* msg);
