// Single-query attention for the LAST encoder layer, computed from the layer INPUT (no K / V projection).
//
// Same result as attention_lastq.hip on K = x WK^T + bK, V = x WV^T + bV (row L-1 of ScaledDotProductAttention,
// Transformer/transformer.py:119-129, masked_fill -1e9 / all-masked rows uniform, attention-map dropout), but with
// the two projections absorbed into the single query and the single output:
//   score_j   = q_h . (WK_h x_j + bK_h)           = x_j . (WK_h^T q_h)  +  q_h . bK_h
//   context_h = sum_j p_j (WV_h x_j + bV_h)       = WV_h (sum_j p_j x_j) + bV_h sum_j p_j
// so that per sequence the work is  [L,128] x [128, 4 heads]  (scores) and  [4, L] x [L,128]  (weighted row sums)
// instead of the [L,128] x [128,256] K/V product: 28x fewer flops, x is read once (256 B per position) and the
// [B,L,256] K/V tensor (written, then read back by the single-query kernel) does not exist.  The backward mirrors it:
//   dxbar_h = WV_h^T dctx_h;  dp_j = x_j . dxbar_h + bV_h . dctx_h;  ds = softmax backward;
//   dq'_h = sum_j ds_j x_j;   dx_j = sum_h p_hj dxbar_h + ds_hj q'_h;  dq_h = WK_h dq'_h;
//   dWV_h = sum_b dctx_h (x) xbar_h,  dWK_h = sum_b q_h (x) dq'_h   (two small products over B*H rows, made by rg_gemm_tn
//   from the operands this kernel writes),  dbV_h = sum_b dctx_h * sum_j p_j,  dbK = 0 exactly.
//
// One workgroup (4 waves) per sequence, persistent over sequences; wave h owns head h and keeps WK_h and WV_h
// (32 x 128 bf16 each) in registers, each lane the two features (2*lane, 2*lane+1) of all 32 rows.  x rows are staged
// once in LDS (XOR-swizzled 16-byte chunks: conflict-free both for the MFMA row fragments and for the feature-pair
// reads).  Scores (and in the backward x . dxbar in the same pass) are ONE 16x16x32 MFMA column block: the B operand's
// 16 columns hold q'_h split into bf16 high + low parts (4 + 4 columns; backward: dxbar_h in the other 8), so the f32
// vectors lose nothing to the bf16 operand format.  bf16 tier, d_model = H*32 = 128, L <= 256.
#include "rg_common.hip.h"
#include "rg_det.hip.h"
#include "../../include/recguru_hip.h"

#define LX_D 128
#define LX_H 4
#define LX_DK 32
#define LX_QLD 136            // B-operand row stride (bf16): 272 B, 16-byte chunks rotate through the banks
#define LX_MASK_BIG (-1e30f)
#define LX_KPL 4              // keys per lane in the softmax stage: L <= 256

namespace {

__device__ __forceinline__ float bf_lo(unsigned int w) { return __uint_as_float(w << 16); }
__device__ __forceinline__ float bf_hi(unsigned int w) { return __uint_as_float(w & 0xffff0000u); }
__device__ __forceinline__ unsigned int pack_bf(float a, float b) {
  union { __bf16 h[2]; unsigned int u; } p;
  p.h[0] = (__bf16)a; p.h[1] = (__bf16)b;
  return p.u;
}
__device__ __forceinline__ float bcast(float v, int l) { return __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(v), l)); }

// wave reductions by DPP (row-local steps, then row_bcast15 / row_bcast31 into the last lane): 6 VALU operations and a
// v_readlane instead of 6 dependent ds_bpermute round trips -- the four reductions of a sequence sit on its critical path
template <int CTRL, int ROWS>
__device__ __forceinline__ float dpp_move(float idle, float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(idle), __float_as_int(v), CTRL, ROWS, 0xf, false));
}
__device__ __forceinline__ float wave_sum_dpp(float v) {
  v += dpp_move<0xB1, 0xf>(0.f, v);          // quad_perm [1,0,3,2]
  v += dpp_move<0x4E, 0xf>(0.f, v);          // quad_perm [2,3,0,1]
  v += dpp_move<0x141, 0xf>(0.f, v);         // row_half_mirror
  v += dpp_move<0x140, 0xf>(0.f, v);         // row_mirror: every lane holds its row's sum
  v += dpp_move<0x142, 0xa>(0.f, v);         // row_bcast15 into rows 1 and 3
  v += dpp_move<0x143, 0xc>(0.f, v);         // row_bcast31 into rows 2 and 3
  return bcast(v, 63);
}
__device__ __forceinline__ float wave_max_dpp(float v) {
  v = fmaxf(v, dpp_move<0xB1, 0xf>(v, v));
  v = fmaxf(v, dpp_move<0x4E, 0xf>(v, v));
  v = fmaxf(v, dpp_move<0x141, 0xf>(v, v));
  v = fmaxf(v, dpp_move<0x140, 0xf>(v, v));
  v = fmaxf(v, dpp_move<0x142, 0xa>(v, v));
  v = fmaxf(v, dpp_move<0x143, 0xc>(v, v));
  return bcast(v, 63);
}

// sum over the 64 lanes of 32 per-lane values; lane l ends up with the total of value index l >> 1 (5 halving
// exchanges of 16, 8, 4, 2, 1 values and a last pair sum: 32 cross-lane moves instead of 32 x 6)
template <int W>
__device__ __forceinline__ void halve(float (&t)[32], int lane) {
  const bool up = lane & (2 * W);
#pragma unroll
  for (int i = 0; i < W; ++i) {
    const float send = up ? t[i] : t[W + i];
    const float keep = up ? t[W + i] : t[i];
    t[i] = keep + __shfl_xor(send, 2 * W);
  }
}
__device__ __forceinline__ float reduce32(float (&t)[32], int lane) {
  halve<16>(t, lane); halve<8>(t, lane); halve<4>(t, lane); halve<2>(t, lane); halve<1>(t, lane);
  return t[0] + __shfl_xor(t[0], 1);
}

struct Smem {
  unsigned int* xs;    // [LT*16 rows][64 dwords], chunk c of row r at chunk slot c ^ (r & 15)
  __bf16* qs;          // [16][LX_QLD]  B operand columns
  float* ss;           // [4][SLD]      scores -> dropped probabilities
  float* dp;           // [4][SLD]      backward: x . dxbar
  float* cf;           // [LT*16][8]    backward: (p~, ds) per head
  float* vec;          // [8][128]      backward: dxbar_h, q'_h (f32)
  float* xb;           // [8][128]      weighted row sums of all heads (forward: 4 rows used)
};

// x rows [rs, SLD) of sequence b into LDS by LDS-DMA (no staging registers, every row in flight at once): one wave
// instruction brings 4 rows; the chunk swizzle is applied on the SOURCE side (lane (row rr, slot s) fetches chunk
// s ^ (row & 15)).  Rows >= L are copies of row L-1 (finite; their scores are discarded and their weights are zero).
__device__ __forceinline__ void stage_x(const Smem& sm, const __bf16* __restrict__ xb, int rs, int L, int SLD, int wave, int lane) {
  const int rr = lane >> 4, sl = lane & 15;
  for (int r0 = rs + 4 * wave; r0 < SLD; r0 += 16) {
    const int r = r0 + rr;
    const __bf16* src = xb + (size_t)min(r, L - 1) * LX_D + 8 * (sl ^ (r & 15));
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                     (__attribute__((address_space(3))) void*)(sm.xs + (r0 << 6)), 16, 0, 0);
  }
}
__device__ __forceinline__ void stage_wait() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

__device__ __forceinline__ unsigned int x_pair(const Smem& sm, int j, int lane) {
  return sm.xs[(j << 6) + ((((lane >> 2) ^ (j & 15))) << 2) + (lane & 3)];
}

// the f32 pair (a0, a1) of features (2*lane, 2*lane+1) as bf16 high and low parts into B-operand columns ch, cl
__device__ __forceinline__ void put_cols(const Smem& sm, int ch, int cl, float a0, float a1, int lane) {
  const float h0 = (float)(__bf16)a0, h1 = (float)(__bf16)a1;
  reinterpret_cast<unsigned int*>(sm.qs + ch * LX_QLD)[lane] = pack_bf(h0, h1);
  reinterpret_cast<unsigned int*>(sm.qs + cl * LX_QLD)[lane] = pack_bf(a0 - h0, a1 - h1);
}

// [16 rows of tile t] x [16 operand columns]: lane (n = lane & 15, g = lane >> 4) gets rows 4g..4g+3 of column n
__device__ __forceinline__ f32x4 tile_product(const Smem& sm, int t, const Frag<__bf16> (&bq)[4], int lane) {
  const int i = lane & 15, g = lane >> 4;
  f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
  const unsigned int* row = sm.xs + ((16 * t + i) << 6);
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) {
    Frag<__bf16> a;
    a.v = *reinterpret_cast<const bf16x8_t*>(row + (((ks * 4 + g) ^ i) << 2));
    mma(a, bq[ks], acc);
  }
  return acc;
}

// B operand of a weighted row sum: slots (g, jj) = rows j0 + 8g + jj of feature 16 ft + (lane & 15), straight from the
// swizzled row-major image by two transposing reads (4 rows x 16 features per 16 lanes each)
__device__ __forceinline__ void x_rows_frag(Frag<__bf16>& f, const Smem& sm, int j0, int ft, int lane) {
  typedef __attribute__((ext_vector_type(4))) short s16x4;
  typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
  const int li = lane & 15, g = lane >> 4, q = li >> 2, pp = li & 3;
  const int r0 = 8 * g + q, ch = 2 * ft + (pp >> 1);
  const unsigned char* base = reinterpret_cast<const unsigned char*>(sm.xs) + ((pp & 1) << 3);
  const unsigned char* p0 = base + (j0 + r0) * 256 + ((ch ^ (r0 & 15)) << 4);
  const unsigned char* p1 = base + (j0 + r0 + 4) * 256 + ((ch ^ ((r0 + 4) & 15)) << 4);
  union { s16x4 s; bf16x4_t b; } u0, u1;
  u0.s = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)p0);
  u1.s = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)p1);
#pragma unroll
  for (int j = 0; j < 4; ++j) { f.v[j] = u0.b[j]; f.v[4 + j] = u1.b[j]; }
}
// A operand rows from f32 weights w[0..7] (this lane's 8 slots): bf16 high part where `hi`, low part where `lo`, else 0
__device__ __forceinline__ void weights_frag(Frag<__bf16>& f, const float (&w)[8], bool hi, bool lo) {
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const float h = (float)(__bf16)w[j];
    f.v[j] = (__bf16)(hi ? h : (lo ? w[j] - h : 0.f));
  }
}

struct Head {                       // what wave h keeps for the whole launch
  unsigned int wk[32], wv[32];      // rows h*32 + c of WK / WV, features (2*lane, 2*lane+1)
  float bk, bv;                     // lane c < 32: bK / bV [h*32 + c]
};
__device__ __forceinline__ void load_head(Head& hd, const rg_lastq_x_args& a, int h, int lane) {
  const unsigned int* wk = reinterpret_cast<const unsigned int*>(a.wk) + (size_t)h * LX_DK * (LX_D / 2) + lane;
  const unsigned int* wv = reinterpret_cast<const unsigned int*>(a.wv) + (size_t)h * LX_DK * (LX_D / 2) + lane;
#pragma unroll
  for (int c = 0; c < 32; ++c) { hd.wk[c] = wk[c * (LX_D / 2)]; hd.wv[c] = wv[c * (LX_D / 2)]; }
  hd.bk = lane < 32 ? a.bk[h * LX_DK + lane] : 0.f;
  hd.bv = lane < 32 ? a.bv[h * LX_DK + lane] : 0.f;
}

// The packed rows are declared "possibly changed" once per sequence: without it the compiler hoists the 128 unpacked
// floats per matrix out of the sequence loop and spills.
__device__ __forceinline__ void pin_head(Head& hd) {
#pragma unroll
  for (int c = 0; c < 32; ++c) { asm volatile("" : "+v"(hd.wk[c])); asm volatile("" : "+v"(hd.wv[c])); }
}

// out pair = sum_c v[c] * W[c][pair], v[c] = lane c of `vl`
__device__ __forceinline__ void vec_times_rows(const unsigned int (&w)[32], float vl, float& o0, float& o1) {
  o0 = 0.f; o1 = 0.f;
#pragma unroll
  for (int c = 0; c < 32; ++c) {
    const float vc = bcast(vl, c);
    o0 += vc * bf_lo(w[c]);
    o1 += vc * bf_hi(w[c]);
  }
}
// lane l gets sum_e W[l >> 1][e] * v[e], v given as the pair (v0, v1) per lane
__device__ __forceinline__ float rows_times_vec(const unsigned int (&w)[32], float v0, float v1, int lane) {
  float t[32];
#pragma unroll
  for (int c = 0; c < 32; ++c) t[c] = bf_lo(w[c]) * v0 + bf_hi(w[c]) * v1;
  return reduce32(t, lane);
}

// softmax stage of wave h: keys lane + 64 i.  Leaves p (undropped), keep multiplier and the sum of p~.
__device__ __forceinline__ void softmax_keys(const Smem& sm, int SLD, int h, int lane, int L, int rs, float qb,
                                             const int64_t* __restrict__ ids, int64_t pad_value, const DropCfg& drop,
                                             unsigned int dbase, float (&p)[LX_KPL], float (&kp)[LX_KPL], bool (&msk)[LX_KPL],
                                             bool& full) {
  float s[LX_KPL];
  float mx = -INFINITY;
#pragma unroll
  for (int i = 0; i < LX_KPL; ++i) {
    const int j = lane + 64 * i;
    const int64_t id = ids[min(j, L - 1)];
    const float raw = (j >= rs && j < L ? sm.ss[h * SLD + j] : 0.f) + qb;
    msk[i] = id == pad_value;
    s[i] = j < L ? (msk[i] ? LX_MASK_BIG : raw) : -INFINITY;
    mx = fmaxf(mx, s[i]);
  }
  mx = wave_max_dpp(mx);
  full = mx < 0.5f * LX_MASK_BIG;
  float sum = 0.f;
#pragma unroll
  for (int i = 0; i < LX_KPL; ++i) { s[i] = __expf(s[i] - mx); sum += s[i]; }
  sum = wave_sum_dpp(sum);
  const float inv = 1.f / sum;
#pragma unroll
  for (int i = 0; i < LX_KPL; ++i) {
    const int j = lane + 64 * i;
    p[i] = j < L ? s[i] * inv : 0.f;
    kp[i] = drop.thresh ? rg_keep(drop, dbase + j) : 1.f;
  }
}

}  // namespace

__global__ __launch_bounds__(256, 2) void attn_lastq_x_fwd_kernel(rg_lastq_x_args a) {
  extern __shared__ __align__(16) unsigned char lx_smem[];
  const int tid = threadIdx.x, lane = tid & 63, h = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int L = a.L, LT = (L + 15) >> 4, SLD = ((L + 31) >> 5) << 5;      // rows held in LDS: whole 32-row k-steps
  Smem sm;
  sm.xs = reinterpret_cast<unsigned int*>(lx_smem);
  sm.qs = reinterpret_cast<__bf16*>(sm.xs + SLD * 64);
  sm.ss = reinterpret_cast<float*>(sm.qs + 16 * LX_QLD);
  sm.xb = sm.ss + 4 * SLD;
  for (int i = tid; i < 16 * LX_QLD / 2; i += 256) reinterpret_cast<unsigned int*>(sm.qs)[i] = 0u;
  Head hd;
  load_head(hd, a, h, lane);
  const DropCfg drop = make_drop(a.drop_p, a.seed);
  const __bf16* __restrict__ X = reinterpret_cast<const __bf16*>(a.x);
  const __bf16* __restrict__ Q = reinterpret_cast<const __bf16*>(a.qlast);
  __bf16* __restrict__ C = reinterpret_cast<__bf16*>(a.ctx);
  // the x rows of a sequence are requested as soon as the previous sequence's last reader of the LDS image is done (one
  // barrier before its epilogue), so the HBM latency runs under the epilogue and the next query's projection
  auto first_row = [&](int b) { return (a.first_live ? min(a.first_live[b], L - 1) : 0) & ~31; };
  // ... and what the next sequence needs from global memory BEFORE its rows land (query, first row) is fetched one stage
  // earlier still and pinned in registers ahead of the request: a load issued after the LDS-DMAs could only be waited
  // for together with them (vmcnt counts in order)
  // (raw values: any arithmetic on them would make the compiler wait for the load where it is issued)
  const unsigned short* Qr = reinterpret_cast<const unsigned short*>(a.qlast) + h * LX_DK + (lane & 31);
  auto as_q = [&](unsigned int raw) { return lane < 32 ? __uint_as_float(raw << 16) : 0.f; };
  int rs = blockIdx.x < a.B ? first_row(blockIdx.x) : 0;
  float ql_next = blockIdx.x < a.B ? as_q(Qr[(size_t)blockIdx.x * LX_D]) : 0.f;
  asm volatile("" : "+v"(ql_next));
  pin_head(hd);                                                 // (the weight loads are waited for HERE, not at the loop head)
  __syncthreads();                                              // (qs zeroed)
  if (blockIdx.x < a.B) stage_x(sm, X + (size_t)blockIdx.x * L * LX_D, rs, L, SLD, h, lane);
  for (int b = blockIdx.x; b < a.B; b += gridDim.x) {
    pin_head(hd);
    const float ql = ql_next;
    const int nb = b + gridDim.x;
    float q0, q1;
    vec_times_rows(hd.wk, ql, q0, q1);
    q0 *= a.scale; q1 *= a.scale;
    const float qb = wave_sum_dpp(ql * hd.bk) * a.scale;
    put_cols(sm, h, 4 + h, q0, q1, lane);
    stage_wait();
    __syncthreads();
    {                                                           // scores of all heads, tiles dealt to the waves
      Frag<__bf16> bq[4];
      const int n = lane & 15, g = lane >> 4;
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) bq[ks].v = *reinterpret_cast<const bf16x8_t*>(sm.qs + n * LX_QLD + ks * 32 + 8 * g);
#pragma unroll 1
      for (int t = (rs >> 4) + h; t < LT; t += 4) {
        const f32x4 acc = tile_product(sm, t, bq, lane);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float v = acc[r] + dpp_move<0x104, 0xf>(0.f, acc[r]);          // row_shl:4: column n + 4 (the low parts)
          if (n < 4) sm.ss[n * SLD + 16 * t + 4 * g + r] = v;
        }
      }
    }
    __syncthreads();
    const int nbc = min(nb, a.B - 1);
    int first_raw = a.first_live ? a.first_live[nbc] : 0;       // in flight under the softmax and the row sums
    unsigned int q_raw = Qr[(size_t)nbc * LX_D];
    float p[LX_KPL], kp[LX_KPL];
    bool msk[LX_KPL], full;
    const unsigned int dbase = (((unsigned int)b * LX_H + h) * L + (L - 1)) * rg_lpad(L);
    softmax_keys(sm, SLD, h, lane, L, rs, qb, a.key_ids + (size_t)b * L, a.pad_value, drop, dbase, p, kp, msk, full);
    float sp = 0.f;
#pragma unroll
    for (int i = 0; i < LX_KPL; ++i) {
      const int j = lane + 64 * i;
      const float pd = p[i] * kp[i];
      sp += pd;
      if (j < SLD) sm.ss[h * SLD + j] = pd;                    // this wave's own row: no barrier needed
    }
    sp = wave_sum_dpp(sp);
    __syncthreads();                                            // p~ of every head is in place
    {                                                           // xbar[h][f] = sum_j p~[h][j] x[j][f]: wave w makes features 32w .. 32w+31
      const int i = lane & 15, g = lane >> 4;
      f32x4 acc[2] = {(f32x4){0.f, 0.f, 0.f, 0.f}, (f32x4){0.f, 0.f, 0.f, 0.f}};
#pragma unroll 1
      for (int j0 = rs; j0 < SLD; j0 += 32) {
        const float* src = sm.ss + (i & 3) * SLD + j0 + 8 * g;
        const float4 w0 = *reinterpret_cast<const float4*>(src), w1 = *reinterpret_cast<const float4*>(src + 4);
        const float w[8] = {w0.x, w0.y, w0.z, w0.w, w1.x, w1.y, w1.z, w1.w};
        Frag<__bf16> af, b0, b1;
        weights_frag(af, w, i < 4, i >= 4 && i < 8);
        x_rows_frag(b0, sm, j0, 2 * h, lane);
        x_rows_frag(b1, sm, j0, 2 * h + 1, lane);
        mma(af, b0, acc[0]);
        mma(af, b1, acc[1]);
      }
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {                           // rows 0-3: high parts, rows 4-7 (lanes 16-31): low parts
          const float v = acc[t][r] + __shfl_down(acc[t][r], 16);
          if (g == 0) sm.xb[r * LX_D + 16 * (2 * h + t) + i] = v;
        }
    }
    __syncthreads();                                            // xb complete; nobody reads the x image any more
    const float2 xp = reinterpret_cast<const float2*>(sm.xb + h * LX_D)[lane];
    const float a0 = xp.x, a1 = xp.y;
    asm volatile("" : "+v"(q_raw), "+v"(first_raw));             // both have arrived: nothing after the request waits on them
    ql_next = as_q(q_raw);
    if (nb < a.B) { rs = __builtin_amdgcn_readfirstlane(min(first_raw, L - 1) & ~31); stage_x(sm, X + (size_t)nb * L * LX_D, rs, L, SLD, h, lane); }
    const float o = rows_times_vec(hd.wv, a0, a1, lane);
    const float bvl = __shfl(hd.bv, lane >> 1);
    if (!(lane & 1)) C[(size_t)b * LX_D + h * LX_DK + (lane >> 1)] = (__bf16)(o + bvl * sp);
  }
}

__global__ __launch_bounds__(256, 2) void attn_lastq_x_bwd_kernel(rg_lastq_x_args a) {
  extern __shared__ __align__(16) unsigned char lx_smem[];
  const int tid = threadIdx.x, lane = tid & 63, h = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int L = a.L, LT = (L + 15) >> 4, SLD = ((L + 31) >> 5) << 5;      // rows held in LDS: whole 32-row k-steps
  Smem sm;
  sm.xs = reinterpret_cast<unsigned int*>(lx_smem);
  sm.qs = reinterpret_cast<__bf16*>(sm.xs + SLD * 64);
  sm.ss = reinterpret_cast<float*>(sm.qs + 16 * LX_QLD);
  sm.dp = sm.ss + 4 * SLD;
  sm.cf = sm.ss + (8 * SLD > 8 * LX_D ? 8 * SLD : 8 * LX_D);
  sm.vec = sm.cf + SLD * 8;
  sm.xb = sm.ss;                                                // [8][128] <= ss | dp: both dead once cf is written
  Head hd;
  load_head(hd, a, h, lane);
  const DropCfg drop = make_drop(a.drop_p, a.seed);
  const __bf16* __restrict__ X = reinterpret_cast<const __bf16*>(a.x);
  const __bf16* __restrict__ Q = reinterpret_cast<const __bf16*>(a.qlast);
  const __bf16* __restrict__ G = reinterpret_cast<const __bf16*>(a.dctx);
  unsigned int* __restrict__ DX = reinterpret_cast<unsigned int*>(a.dx);
  __bf16* __restrict__ DQ = reinterpret_cast<__bf16*>(a.dq);
  unsigned int* __restrict__ YV = reinterpret_cast<unsigned int*>(a.ym_v);
  unsigned int* __restrict__ YQ = reinterpret_cast<unsigned int*>(a.ym_q);
  unsigned int* __restrict__ XB = reinterpret_cast<unsigned int*>(a.xbar);
  unsigned int* __restrict__ DP = reinterpret_cast<unsigned int*>(a.dqp);
  float dbv = 0.f;
  auto first_row = [&](int b) { return (a.first_live ? min(a.first_live[b], L - 1) : 0) & ~31; };
  const unsigned short* Qr = reinterpret_cast<const unsigned short*>(a.qlast) + h * LX_DK + (lane & 31);
  const unsigned short* Gr = reinterpret_cast<const unsigned short*>(a.dctx) + h * LX_DK + (lane & 31);
  auto as_q = [&](unsigned int raw) { return lane < 32 ? __uint_as_float(raw << 16) : 0.f; };
  int rs_next = blockIdx.x < a.B ? first_row(blockIdx.x) : 0;
  float ql_next = blockIdx.x < a.B ? as_q(Qr[(size_t)blockIdx.x * LX_D]) : 0.f;
  float gl_next = blockIdx.x < a.B ? as_q(Gr[(size_t)blockIdx.x * LX_D]) : 0.f;
  asm volatile("" : "+v"(ql_next), "+v"(gl_next));
  pin_head(hd);
  if (blockIdx.x < a.B) stage_x(sm, X + (size_t)blockIdx.x * L * LX_D, rs_next, L, SLD, h, lane);
  for (int b = blockIdx.x; b < a.B; b += gridDim.x) {
    const int rs = rs_next;
    const int nb = b + gridDim.x;
    pin_head(hd);
    __syncthreads();                                            // the previous sequence's dx stage is done with cf / vec
    const float ql = ql_next, gl = gl_next;
    float q0, q1, d0, d1;
    vec_times_rows(hd.wk, ql, q0, q1);
    q0 *= a.scale; q1 *= a.scale;
    __builtin_amdgcn_sched_barrier(0);                          // one matrix unpacked at a time
    vec_times_rows(hd.wv, gl, d0, d1);                          // dxbar_h
    __builtin_amdgcn_sched_barrier(0);
    const float qb = wave_sum_dpp(ql * hd.bk) * a.scale;
    const float dsp = wave_sum_dpp(gl * hd.bv);                     // d(sum of p~)
    put_cols(sm, h, 4 + h, q0, q1, lane);
    put_cols(sm, 8 + h, 12 + h, d0, d1, lane);
    reinterpret_cast<float2*>(sm.vec + h * LX_D)[lane] = make_float2(d0, d1);
    reinterpret_cast<float2*>(sm.vec + (4 + h) * LX_D)[lane] = make_float2(q0, q1);
    {                                                           // the operands of the dWV / dWK products: other heads' blocks zero
      const bool own = (lane >> 4) == h;                        // features (2*lane, 2*lane+1) belong to head lane >> 4
      const int c = (2 * lane) & 31;                            // ... and are elements c, c+1 of this head's 32 (held by lanes c, c+1)
      const unsigned int gq = pack_bf(__shfl(gl, c), __shfl(gl, c + 1));      // (exact: the values came from bf16)
      const unsigned int qq = pack_bf(__shfl(ql, c), __shfl(ql, c + 1));
      YV[((size_t)b * LX_H + h) * (LX_D / 2) + lane] = own ? gq : 0u;
      YQ[((size_t)b * LX_H + h) * (LX_D / 2) + lane] = own ? qq : 0u;
    }
    stage_wait();
    __syncthreads();
    {                                                           // x . q' and x . dxbar, all heads, one pass
      Frag<__bf16> bq[4];
      const int n = lane & 15, g = lane >> 4;
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) bq[ks].v = *reinterpret_cast<const bf16x8_t*>(sm.qs + n * LX_QLD + ks * 32 + 8 * g);
#pragma unroll 1
      for (int t = (rs >> 4) + h; t < LT; t += 4) {
        const f32x4 acc = tile_product(sm, t, bq, lane);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float v = acc[r] + dpp_move<0x104, 0xf>(0.f, acc[r]);          // row_shl:4: column n + 4 (the low parts)
          if (n < 4) sm.ss[n * SLD + 16 * t + 4 * g + r] = v;
          else if (n >= 8 && n < 12) sm.dp[(n - 8) * SLD + 16 * t + 4 * g + r] = v;
        }
      }
    }
    __syncthreads();
    const int nbc = min(nb, a.B - 1);
    int first_raw = a.first_live ? a.first_live[nbc] : 0;       // (raw: see the forward kernel)
    unsigned int q_raw = Qr[(size_t)nbc * LX_D], g_raw = Gr[(size_t)nbc * LX_D];
    float p[LX_KPL], kp[LX_KPL];
    bool msk[LX_KPL], full;
    const unsigned int dbase = (((unsigned int)b * LX_H + h) * L + (L - 1)) * rg_lpad(L);
    softmax_keys(sm, SLD, h, lane, L, rs, qb, a.key_ids + (size_t)b * L, a.pad_value, drop, dbase, p, kp, msk, full);
    float dpk[LX_KPL];
    float delta = 0.f, sp = 0.f;
#pragma unroll
    for (int i = 0; i < LX_KPL; ++i) {
      const int j = lane + 64 * i;
      const float dpt = (j >= rs && j < L ? sm.dp[h * SLD + j] : 0.f) + dsp;
      dpk[i] = dpt * kp[i];                                     // d loss / d (undropped probability)
      delta += p[i] * dpk[i];
      sp += p[i] * kp[i];
    }
    delta = wave_sum_dpp(delta);
    sp = wave_sum_dpp(sp);
#pragma unroll
    for (int i = 0; i < LX_KPL; ++i) {
      const int j = lane + 64 * i;
      const float ds = (full || msk[i]) ? 0.f : p[i] * (dpk[i] - delta);
      if (j < SLD) *reinterpret_cast<float2*>(sm.cf + j * 8 + 2 * h) = make_float2(p[i] * kp[i], ds);
    }
    __syncthreads();                                            // cf of every head is in place; scores / dp are dead
    {                                                           // xbar[h][f] = sum_j p~ x, dq'[h][f] = sum_j ds x: wave w makes features 32w .. 32w+31
      const int i = lane & 15, g = lane >> 4;
      f32x4 acc[2] = {(f32x4){0.f, 0.f, 0.f, 0.f}, (f32x4){0.f, 0.f, 0.f, 0.f}};
      const float* src = sm.cf + 2 * (i & 3) + (i >> 3);        // operand rows: p~ high, p~ low, ds high, ds low (4 heads each)
#pragma unroll 1
      for (int j0 = rs; j0 < SLD; j0 += 32) {
        float w[8];
#pragma unroll
        for (int jj = 0; jj < 8; ++jj) w[jj] = src[(j0 + 8 * g + jj) * 8];
        Frag<__bf16> af, b0, b1;
        weights_frag(af, w, !(i & 4), (i & 4) != 0);
        x_rows_frag(b0, sm, j0, 2 * h, lane);
        x_rows_frag(b1, sm, j0, 2 * h + 1, lane);
        mma(af, b0, acc[0]);
        mma(af, b1, acc[1]);
      }
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {                           // lanes 0-15: xbar (high + low rows), lanes 32-47: dq'
          const float v = acc[t][r] + __shfl_down(acc[t][r], 16);
          if (!(g & 1)) sm.xb[((g >> 1) * 4 + r) * LX_D + 16 * (2 * h + t) + i] = v;
        }
    }
    __syncthreads();                                            // xb complete; nobody reads the x image any more
    asm volatile("" : "+v"(q_raw), "+v"(g_raw), "+v"(first_raw));
    ql_next = as_q(q_raw); gl_next = as_q(g_raw);
    if (nb < a.B) { rs_next = __builtin_amdgcn_readfirstlane(min(first_raw, L - 1) & ~31); stage_x(sm, X + (size_t)nb * L * LX_D, rs_next, L, SLD, h, lane); }   // in flight under the dx stage
    const float2 xp = reinterpret_cast<const float2*>(sm.xb + h * LX_D)[lane];
    const float2 gp = reinterpret_cast<const float2*>(sm.xb + (4 + h) * LX_D)[lane];
    const float a0 = xp.x, a1 = xp.y;
    float g0 = gp.x, g1 = gp.y;
    g0 *= a.scale; g1 *= a.scale;
    XB[((size_t)b * LX_H + h) * (LX_D / 2) + lane] = pack_bf(a0, a1);
    DP[((size_t)b * LX_H + h) * (LX_D / 2) + lane] = pack_bf(g0, g1);
    pin_head(hd);                                               // (no reuse of the unpacked rows from the top of the loop)
    __builtin_amdgcn_sched_barrier(0);
    const float dq = rows_times_vec(hd.wk, g0, g1, lane);
    if (!(lane & 1)) DQ[(size_t)b * LX_D + h * LX_DK + (lane >> 1)] = (__bf16)dq;
    dbv += gl * sp;
    {                                                           // dx rows: wave w takes rows rs + w, rs + w + 4, ...
      float2 vd[4], vq[4];
#pragma unroll
      for (int hh = 0; hh < 4; ++hh) {
        vd[hh] = reinterpret_cast<const float2*>(sm.vec + hh * LX_D)[lane];
        vq[hh] = reinterpret_cast<const float2*>(sm.vec + (4 + hh) * LX_D)[lane];
      }
      unsigned int* dxb = DX + (size_t)b * L * (LX_D / 2);
      for (int j = h; j < rs; j += 4) dxb[(size_t)j * (LX_D / 2) + lane] = 0u;
#pragma unroll 2
      for (int j = rs + h; j < L; j += 4) {
        const float4 c0 = *reinterpret_cast<const float4*>(sm.cf + j * 8);
        const float4 c1 = *reinterpret_cast<const float4*>(sm.cf + j * 8 + 4);
        const float cc[8] = {c0.x, c0.y, c0.z, c0.w, c1.x, c1.y, c1.z, c1.w};
        float o0 = 0.f, o1 = 0.f;
#pragma unroll
        for (int hh = 0; hh < 4; ++hh) {
          o0 += cc[2 * hh] * vd[hh].x + cc[2 * hh + 1] * vq[hh].x;
          o1 += cc[2 * hh] * vd[hh].y + cc[2 * hh + 1] * vq[hh].y;
        }
        dxb[(size_t)j * (LX_D / 2) + lane] = pack_bf(o0, o1);
      }
    }
  }
  if (a.dbv && lane < 32) rg_acc(a.dbv + h * LX_DK + lane, dbv);
}


// =====================================================================================================================================
// f32 form (round 6; the f32 and bf16x3 tiers: x, q, dctx and the weights are f32).  Same algebra, EXACT f32 arithmetic on the vector
// ALU -- per sequence the work is 4 x [L,128] dot products and 4 weighted row sums, ~2 k FMAs per lane, far below what fetching
// the sequence's 100 KB of rows costs, so there is nothing for the matrix pipe to win and no operand split to pay.  Replaces, in those
// tiers, the K | V projection GEMM ([B L,128] -> [B L,256] f32, written and read back) + rg_attn_lastq_fwd/bwd + the dkv -> dx GEMM
// and the dWK | dWV product over B L rows (8.5 ms of the bf16x3 step against 1.3 ms in the bf16 tier).
// One workgroup per sequence (persistent), wave h = head h.  The x rows [rs, SLD) live in LDS as raw f32 (512 B per row, 16-byte chunk
// c of row r in slot c ^ (r & 31): conflict-free both for "lane = row" reads of one chunk and for "lane = feature pair" reads of one
// row), staged by LDS-DMA with the swizzle on the source side.  L <= 256 (160 KB of LDS: one workgroup per CU).
struct SmemF {
  float* xs;   // [SLD][128]
  float* vec;  // [8][128]   q'_h (rows 0-3), dxbar_h (rows 4-7, backward)
  float* ss;   // [4][SLD]   scores -> dropped probabilities
  float* dp;   // [4][SLD]   backward: x . dxbar
  float* cf;   // [SLD][8]   backward: (p~, ds) per head
};
__device__ __forceinline__ void stage_xf(const SmemF& sm, const float* __restrict__ xb, int rs, int L, int SLD, int wave, int lane) {
  const int rr = lane >> 5, sl = lane & 31;
  for (int r0 = rs + 2 * wave; r0 < SLD; r0 += 8) {
    const int r = r0 + rr;
    const float* src = xb + (size_t)min(r, L - 1) * LX_D + 4 * (sl ^ (r & 31));
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                     (__attribute__((address_space(3))) void*)(sm.xs + r0 * LX_D), 16, 0, 0);
  }
}
__device__ __forceinline__ float4 xf_chunk(const SmemF& sm, int r, int c) {
  return *reinterpret_cast<const float4*>(sm.xs + r * LX_D + 4 * (c ^ (r & 31)));
}
__device__ __forceinline__ float2 xf_pair(const SmemF& sm, int j, int lane) {
  return *reinterpret_cast<const float2*>(sm.xs + j * LX_D + 4 * ((lane >> 1) ^ (j & 31)) + 2 * (lane & 1));
}
struct HeadF {                      // wave h: rows h*32 + c of WK / WV, features 2*lane (w?0) and 2*lane+1 (w?1)
  float wk0[32], wk1[32], wv0[32], wv1[32];
  float bk, bv;
};
__device__ __forceinline__ void load_headf(HeadF& hd, const rg_lastq_x_args& a, int h, int lane) {
  const float2* wk = reinterpret_cast<const float2*>(a.wk) + (size_t)h * LX_DK * (LX_D / 2) + lane;
  const float2* wv = reinterpret_cast<const float2*>(a.wv) + (size_t)h * LX_DK * (LX_D / 2) + lane;
#pragma unroll
  for (int c = 0; c < 32; ++c) {
    const float2 k2 = wk[c * (LX_D / 2)], v2 = wv[c * (LX_D / 2)];
    hd.wk0[c] = k2.x; hd.wk1[c] = k2.y; hd.wv0[c] = v2.x; hd.wv1[c] = v2.y;
  }
  hd.bk = lane < 32 ? a.bk[h * LX_DK + lane] : 0.f;
  hd.bv = lane < 32 ? a.bv[h * LX_DK + lane] : 0.f;
}
// out pair = sum_c v[c] * W[c][pair], v[c] = lane c of `vl`
__device__ __forceinline__ void vec_times_rowsf(const float (&w0)[32], const float (&w1)[32], float vl, float& o0, float& o1) {
  o0 = 0.f; o1 = 0.f;
#pragma unroll
  for (int c = 0; c < 32; ++c) {
    const float vc = bcast(vl, c);
    o0 = fmaf(vc, w0[c], o0);
    o1 = fmaf(vc, w1[c], o1);
  }
}
// lane l gets sum_e W[l >> 1][e] * v[e], v given as the pair (v0, v1) per lane
__device__ __forceinline__ float rows_times_vecf(const float (&w0)[32], const float (&w1)[32], float v0, float v1, int lane) {
  float t[32];
#pragma unroll
  for (int c = 0; c < 32; ++c) t[c] = fmaf(w0[c], v0, w1[c] * v1);
  return reduce32(t, lane);
}
// dot products of the rows lane + 64 i with vec row va (and, BWD, vec row vb): one pass over the row's 32 chunks
template <bool BWD>
__device__ __forceinline__ void row_dots(const SmemF& sm, int va, int vb, int lane, int rs, int L, int SLD, float (&sa)[LX_KPL], float (&sb)[LX_KPL]) {
  int jr[LX_KPL];
#pragma unroll
  for (int i = 0; i < LX_KPL; ++i) { jr[i] = min(max(lane + 64 * i, rs), SLD - 1); sa[i] = 0.f; sb[i] = 0.f; }
  const int ni = (L + 63) >> 6;                                 // key groups that hold a row < L (wave-uniform)
#pragma unroll 8      // one wave per SIMD: the LDS latency of a chunk hides only behind the other chunks in flight (512 VGPRs to hold them)
  for (int c = 0; c < 32; ++c) {
    const float4 qa = *reinterpret_cast<const float4*>(sm.vec + va * LX_D + 4 * c);
    float4 qb4 = qa;
    if (BWD) qb4 = *reinterpret_cast<const float4*>(sm.vec + vb * LX_D + 4 * c);
#pragma unroll
    for (int i = 0; i < LX_KPL; ++i) {
      if (i < ni) {
        const float4 xv = xf_chunk(sm, jr[i], c);
        sa[i] = fmaf(xv.x, qa.x, fmaf(xv.y, qa.y, fmaf(xv.z, qa.z, fmaf(xv.w, qa.w, sa[i]))));
        if (BWD) sb[i] = fmaf(xv.x, qb4.x, fmaf(xv.y, qb4.y, fmaf(xv.z, qb4.z, fmaf(xv.w, qb4.w, sb[i]))));
      }
    }
  }
}
// softmax stage on register scores (the bf16 form's softmax_keys reads them from LDS): same masks, same dropout indices
__device__ __forceinline__ void load_key_ids(const int64_t* __restrict__ ids, int lane, int L, int64_t (&id)[LX_KPL]) {
#pragma unroll
  for (int i = 0; i < LX_KPL; ++i) id[i] = ids[min(lane + 64 * i, L - 1)];
}
// (the key ids arrive as registers: they are requested at the top of a sequence, in front of the wait for its rows -- one wave per SIMD has
//  nothing to hide a 2 us global-load latency behind)
__device__ __forceinline__ void softmax_keysf(const float (&sc)[LX_KPL], int lane, int L, int rs, float qb, const int64_t (&ids)[LX_KPL],
                                              int64_t pad_value, const DropCfg& drop, unsigned int dbase, float (&p)[LX_KPL],
                                              float (&kp)[LX_KPL], bool (&msk)[LX_KPL], bool& full) {
  float s[LX_KPL];
  float mx = -INFINITY;
#pragma unroll
  for (int i = 0; i < LX_KPL; ++i) {
    const int j = lane + 64 * i;
    const int64_t id = ids[i];
    const float raw = (j >= rs && j < L ? sc[i] : 0.f) + qb;
    msk[i] = id == pad_value;
    s[i] = j < L ? (msk[i] ? LX_MASK_BIG : raw) : -INFINITY;
    mx = fmaxf(mx, s[i]);
  }
  mx = wave_max_dpp(mx);
  full = mx < 0.5f * LX_MASK_BIG;
  float sum = 0.f;
#pragma unroll
  for (int i = 0; i < LX_KPL; ++i) { s[i] = __expf(s[i] - mx); sum += s[i]; }
  sum = wave_sum_dpp(sum);
  const float inv = 1.f / sum;
#pragma unroll
  for (int i = 0; i < LX_KPL; ++i) {
    const int j = lane + 64 * i;
    p[i] = j < L ? s[i] * inv : 0.f;
    kp[i] = drop.thresh ? rg_keep(drop, dbase + j) : 1.f;
  }
}

__global__ __launch_bounds__(256, 1) void attn_lastq_xf_fwd_kernel(rg_lastq_x_args a) {
  extern __shared__ __align__(16) unsigned char lx_smem[];
  const int tid = threadIdx.x, lane = tid & 63, h = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int L = a.L, SLD = ((L + 31) >> 5) << 5;
  SmemF sm;
  sm.xs = reinterpret_cast<float*>(lx_smem);
  sm.vec = sm.xs + SLD * LX_D;
  sm.ss = sm.vec + 8 * LX_D;
  HeadF hd;
  load_headf(hd, a, h, lane);
  const DropCfg drop = make_drop(a.drop_p, a.seed);
  const float* __restrict__ X = reinterpret_cast<const float*>(a.x);
  const float* __restrict__ Q = reinterpret_cast<const float*>(a.qlast);
  float* __restrict__ C = reinterpret_cast<float*>(a.ctx);
  auto first_row = [&](int b) { return (a.first_live ? min(a.first_live[b], L - 1) : 0) & ~31; };
  int rs = blockIdx.x < a.B ? first_row(blockIdx.x) : 0;
  float ql_next = (blockIdx.x < a.B && lane < 32) ? Q[(size_t)blockIdx.x * LX_D + h * LX_DK + lane] : 0.f;
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");              // weights, query, first row: here, not behind the first row requests
  if (blockIdx.x < a.B) stage_xf(sm, X + (size_t)blockIdx.x * L * LX_D, rs, L, SLD, h, lane);
  for (int b = blockIdx.x; b < a.B; b += gridDim.x) {
    const float ql = ql_next;
    const int nb = b + gridDim.x;
    float q0, q1;
    vec_times_rowsf(hd.wk0, hd.wk1, ql, q0, q1);
    q0 *= a.scale; q1 *= a.scale;
    const float qb = wave_sum_dpp(ql * hd.bk) * a.scale;
    reinterpret_cast<float2*>(sm.vec + h * LX_D)[lane] = make_float2(q0, q1);     // this wave's own row (read back by this wave only)
    int64_t kid[LX_KPL];
    load_key_ids(a.key_ids + (size_t)b * L, lane, L, kid);
    stage_wait();
    __syncthreads();                                            // x rows of this sequence have landed
    float sc[LX_KPL], unused[LX_KPL];
    row_dots<false>(sm, h, h, lane, rs, L, SLD, sc, unused);
    float p[LX_KPL], kp[LX_KPL];
    bool msk[LX_KPL], full;
    const unsigned int dbase = (((unsigned int)b * LX_H + h) * L + (L - 1)) * rg_lpad(L);
    softmax_keysf(sc, lane, L, rs, qb, kid, a.pad_value, drop, dbase, p, kp, msk, full);
    float sp = 0.f;
#pragma unroll
    for (int i = 0; i < LX_KPL; ++i) {
      const int j = lane + 64 * i;
      const float pd = p[i] * kp[i];
      sp += pd;
      if (j < SLD) sm.ss[h * SLD + j] = pd;
    }
    sp = wave_sum_dpp(sp);
    float a0 = 0.f, a1 = 0.f;                                   // xbar_h[f] = sum_j p~_j x_j[f], this lane's feature pair
#pragma unroll 4
    for (int j0 = rs; j0 < SLD; j0 += 4) {
      const float4 w4 = *reinterpret_cast<const float4*>(sm.ss + h * SLD + j0);
      const float w[4] = {w4.x, w4.y, w4.z, w4.w};
#pragma unroll
      for (int jj = 0; jj < 4; ++jj) {
        const float2 xv = xf_pair(sm, j0 + jj, lane);
        a0 = fmaf(w[jj], xv.x, a0);
        a1 = fmaf(w[jj], xv.y, a1);
      }
    }
    // what the next sequence needs from global memory, then its rows (in flight under the epilogue below)
    const int nbc = min(nb, a.B - 1);
    const int first_raw = a.first_live ? a.first_live[nbc] : 0;
    ql_next = lane < 32 ? Q[(size_t)nbc * LX_D + h * LX_DK + lane] : 0.f;
    __syncthreads();                                            // every wave is done with the x image
    if (nb < a.B) { rs = __builtin_amdgcn_readfirstlane(min(first_raw, L - 1) & ~31); stage_xf(sm, X + (size_t)nb * L * LX_D, rs, L, SLD, h, lane); }
    const float o = rows_times_vecf(hd.wv0, hd.wv1, a0, a1, lane);
    const float bvl = __shfl(hd.bv, lane >> 1);
    if (!(lane & 1)) C[(size_t)b * LX_D + h * LX_DK + (lane >> 1)] = o + bvl * sp;
  }
}

__global__ __launch_bounds__(256, 1) void attn_lastq_xf_bwd_kernel(rg_lastq_x_args a) {
  extern __shared__ __align__(16) unsigned char lx_smem[];
  const int tid = threadIdx.x, lane = tid & 63, h = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int L = a.L, SLD = ((L + 31) >> 5) << 5;
  SmemF sm;
  sm.xs = reinterpret_cast<float*>(lx_smem);
  sm.vec = sm.xs + SLD * LX_D;
  sm.cf = sm.vec + 8 * LX_D;
  HeadF hd;
  load_headf(hd, a, h, lane);
  const DropCfg drop = make_drop(a.drop_p, a.seed);
  const float* __restrict__ X = reinterpret_cast<const float*>(a.x);
  const float* __restrict__ Q = reinterpret_cast<const float*>(a.qlast);
  const float* __restrict__ G = reinterpret_cast<const float*>(a.dctx);
  float2* __restrict__ DX = reinterpret_cast<float2*>(a.dx);
  float* __restrict__ DQ = reinterpret_cast<float*>(a.dq);
  float2* __restrict__ YV = reinterpret_cast<float2*>(a.ym_v);
  float2* __restrict__ YQ = reinterpret_cast<float2*>(a.ym_q);
  float2* __restrict__ XB = reinterpret_cast<float2*>(a.xbar);
  float2* __restrict__ DP = reinterpret_cast<float2*>(a.dqp);
  float dbv = 0.f;
  auto first_row = [&](int b) { return (a.first_live ? min(a.first_live[b], L - 1) : 0) & ~31; };
  int rs_next = blockIdx.x < a.B ? first_row(blockIdx.x) : 0;
  float ql_next = (blockIdx.x < a.B && lane < 32) ? Q[(size_t)blockIdx.x * LX_D + h * LX_DK + lane] : 0.f;
  float gl_next = (blockIdx.x < a.B && lane < 32) ? G[(size_t)blockIdx.x * LX_D + h * LX_DK + lane] : 0.f;
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (blockIdx.x < a.B) stage_xf(sm, X + (size_t)blockIdx.x * L * LX_D, rs_next, L, SLD, h, lane);
  for (int b = blockIdx.x; b < a.B; b += gridDim.x) {
    const int rs = rs_next;
    const int nb = b + gridDim.x;
    const float ql = ql_next, gl = gl_next;
    float q0, q1, d0, d1;
    vec_times_rowsf(hd.wk0, hd.wk1, ql, q0, q1);
    q0 *= a.scale; q1 *= a.scale;
    vec_times_rowsf(hd.wv0, hd.wv1, gl, d0, d1);                // dxbar_h
    const float qb = wave_sum_dpp(ql * hd.bk) * a.scale;
    const float dsp = wave_sum_dpp(gl * hd.bv);                 // d(sum of p~)
    __syncthreads();                                            // the previous sequence's dx stage is done with cf / vec
    reinterpret_cast<float2*>(sm.vec + h * LX_D)[lane] = make_float2(q0, q1);
    reinterpret_cast<float2*>(sm.vec + (4 + h) * LX_D)[lane] = make_float2(d0, d1);
    {                                                           // the operands of the dWV / dWK products: other heads' blocks zero
      const bool own = (lane >> 4) == h;
      const int c = (2 * lane) & 31;
      const float2 gq = make_float2(__shfl(gl, c), __shfl(gl, c + 1));
      const float2 qq = make_float2(__shfl(ql, c), __shfl(ql, c + 1));
      const float2 z2 = make_float2(0.f, 0.f);
      YV[((size_t)b * LX_H + h) * (LX_D / 2) + lane] = own ? gq : z2;
      YQ[((size_t)b * LX_H + h) * (LX_D / 2) + lane] = own ? qq : z2;
    }
    int64_t kid[LX_KPL];
    load_key_ids(a.key_ids + (size_t)b * L, lane, L, kid);
    stage_wait();
    __syncthreads();                                            // x rows landed; vec complete
    float sc[LX_KPL], dpr[LX_KPL];
    row_dots<true>(sm, h, 4 + h, lane, rs, L, SLD, sc, dpr);
    float p[LX_KPL], kp[LX_KPL];
    bool msk[LX_KPL], full;
    const unsigned int dbase = (((unsigned int)b * LX_H + h) * L + (L - 1)) * rg_lpad(L);
    softmax_keysf(sc, lane, L, rs, qb, kid, a.pad_value, drop, dbase, p, kp, msk, full);
    float dpk[LX_KPL];
    float delta = 0.f, sp = 0.f;
#pragma unroll
    for (int i = 0; i < LX_KPL; ++i) {
      const int j = lane + 64 * i;
      const float dpt = (j >= rs && j < L ? dpr[i] : 0.f) + dsp;
      dpk[i] = dpt * kp[i];
      delta += p[i] * dpk[i];
      sp += p[i] * kp[i];
    }
    delta = wave_sum_dpp(delta);
    sp = wave_sum_dpp(sp);
#pragma unroll
    for (int i = 0; i < LX_KPL; ++i) {
      const int j = lane + 64 * i;
      const float ds = (full || msk[i]) ? 0.f : p[i] * (dpk[i] - delta);
      if (j < SLD) *reinterpret_cast<float2*>(sm.cf + j * 8 + 2 * h) = make_float2(p[i] * kp[i], ds);
    }
    float a0 = 0.f, a1 = 0.f, g0 = 0.f, g1 = 0.f;               // xbar_h and dq'_h, this lane's feature pair (own head's cf column: no barrier)
#pragma unroll 16
    for (int j = rs; j < SLD; ++j) {
      const float2 cw = *reinterpret_cast<const float2*>(sm.cf + j * 8 + 2 * h);
      const float2 xv = xf_pair(sm, j, lane);
      a0 = fmaf(cw.x, xv.x, a0); a1 = fmaf(cw.x, xv.y, a1);
      g0 = fmaf(cw.y, xv.x, g0); g1 = fmaf(cw.y, xv.y, g1);
    }
    g0 *= a.scale; g1 *= a.scale;
    XB[((size_t)b * LX_H + h) * (LX_D / 2) + lane] = make_float2(a0, a1);
    DP[((size_t)b * LX_H + h) * (LX_D / 2) + lane] = make_float2(g0, g1);
    const int nbc = min(nb, a.B - 1);
    const int first_raw = a.first_live ? a.first_live[nbc] : 0;
    ql_next = lane < 32 ? Q[(size_t)nbc * LX_D + h * LX_DK + lane] : 0.f;
    gl_next = lane < 32 ? G[(size_t)nbc * LX_D + h * LX_DK + lane] : 0.f;
    __syncthreads();                                            // cf of every head in place; nobody reads the x image any more
    if (nb < a.B) { rs_next = __builtin_amdgcn_readfirstlane(min(first_raw, L - 1) & ~31); stage_xf(sm, X + (size_t)nb * L * LX_D, rs_next, L, SLD, h, lane); }
    const float dq = rows_times_vecf(hd.wk0, hd.wk1, g0, g1, lane);
    if (!(lane & 1)) DQ[(size_t)b * LX_D + h * LX_DK + (lane >> 1)] = dq;
    dbv += gl * sp;
    {                                                           // dx rows: wave w takes rows rs + w, rs + w + 4, ...
      float2 vd[4], vq[4];
#pragma unroll
      for (int hh = 0; hh < 4; ++hh) {
        vq[hh] = reinterpret_cast<const float2*>(sm.vec + hh * LX_D)[lane];
        vd[hh] = reinterpret_cast<const float2*>(sm.vec + (4 + hh) * LX_D)[lane];
      }
      float2* dxb = DX + (size_t)b * L * (LX_D / 2);
      for (int j = h; j < rs; j += 4) dxb[(size_t)j * (LX_D / 2) + lane] = make_float2(0.f, 0.f);
#pragma unroll 8
      for (int j = rs + h; j < L; j += 4) {
        const float4 c0 = *reinterpret_cast<const float4*>(sm.cf + j * 8);
        const float4 c1 = *reinterpret_cast<const float4*>(sm.cf + j * 8 + 4);
        const float cc[8] = {c0.x, c0.y, c0.z, c0.w, c1.x, c1.y, c1.z, c1.w};
        float o0 = 0.f, o1 = 0.f;
#pragma unroll
        for (int hh = 0; hh < 4; ++hh) {
          o0 += cc[2 * hh] * vd[hh].x + cc[2 * hh + 1] * vq[hh].x;
          o1 += cc[2 * hh] * vd[hh].y + cc[2 * hh + 1] * vq[hh].y;
        }
        dxb[(size_t)j * (LX_D / 2) + lane] = make_float2(o0, o1);
      }
    }
  }
  if (a.dbv && lane < 32) rg_acc(a.dbv + h * LX_DK + lane, dbv);
}

static size_t lxf_smem_bytes(int L, bool bwd) {
  const size_t SLD = (size_t)((L + 31) >> 5) * 32;
  return SLD * LX_D * 4 + 8 * LX_D * 4 + (bwd ? SLD * 8 * 4 : 4 * SLD * 4);
}

static size_t lx_smem_bytes(int L, bool bwd) {
  const size_t SLD = (size_t)((L + 31) >> 5) * 32;
  const size_t sc = 4 * SLD * 4;                                  // one [4][SLD] f32 image
  size_t n = SLD * 256 + 16 * LX_QLD * 2;
  if (!bwd) n += sc + 8 * LX_D * 4;                               // ss | xb
  else n += (2 * sc > 8 * LX_D * 4 ? 2 * sc : 8 * LX_D * 4) + SLD * 8 * 4 + 8 * LX_D * 4;   // ss | dp (xb aliases them) | cf | vec
  return n;
}

extern "C" int rg_attn_lastq_x_supported(int d, int P, int H, int L, int dtype) {
  return (dtype == RG_BF16 || dtype == RG_F32 || dtype == RG_X3) && d == LX_D && P == LX_D && H == LX_H && L >= 1 && L <= 64 * LX_KPL;
}

static int lx_launch(const rg_lastq_x_args* a, bool bwd, hipStream_t s, bool f32 = false) {
  if (!a || a->B <= 0) return 0;
  if (a->L < 1 || a->L > 64 * LX_KPL) return rg_set_error_msg(RG_ERR_UNSUPPORTED, "attn_lastq_x: L must be in 1..256");
  if (!a->x || !a->qlast || !a->wk || !a->wv || !a->bk || !a->bv || !a->key_ids)
    return rg_set_error_msg(RG_ERR_INVALID, "attn_lastq_x: x, qlast, wk, wv, bk, bv and key_ids are required");
  if (!bwd && !a->ctx) return rg_set_error_msg(RG_ERR_INVALID, "attn_lastq_x_fwd: ctx is required");
  if (bwd && (!a->dctx || !a->dx || !a->dq || !a->ym_v || !a->ym_q || !a->xbar || !a->dqp))
    return rg_set_error_msg(RG_ERR_INVALID, "attn_lastq_x_bwd: dctx, dx, dq, ym_v, ym_q, xbar and dqp are required");
  if (f32) {
    const size_t smf = lxf_smem_bytes(a->L, bwd);
    const int gridf = a->B < 256 ? a->B : 256;
    const void* fnf = bwd ? reinterpret_cast<const void*>(attn_lastq_xf_bwd_kernel) : reinterpret_cast<const void*>(attn_lastq_xf_fwd_kernel);
    hipError_t ef = hipFuncSetAttribute(fnf, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smf);
    if (ef != hipSuccess) return rg_set_error(ef, "attn_lastq_xf(smem)");
    if (bwd) hipLaunchKernelGGL(attn_lastq_xf_bwd_kernel, dim3(gridf), dim3(256), smf, s, *a);
    else hipLaunchKernelGGL(attn_lastq_xf_fwd_kernel, dim3(gridf), dim3(256), smf, s, *a);
    RG_CHECK_LAUNCH();
    return 0;
  }
  const size_t smem = lx_smem_bytes(a->L, bwd);
  const int per_cu = smem <= 52 * 1024 ? 3 : (smem <= 80 * 1024 ? 2 : 1);
  const int grid = a->B < 256 * per_cu ? a->B : 256 * per_cu;
  const void* fn = bwd ? reinterpret_cast<const void*>(attn_lastq_x_bwd_kernel) : reinterpret_cast<const void*>(attn_lastq_x_fwd_kernel);
  hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
  if (e != hipSuccess) return rg_set_error(e, "attn_lastq_x(smem)");
  if (bwd) hipLaunchKernelGGL(attn_lastq_x_bwd_kernel, dim3(grid), dim3(256), smem, s, *a);
  else hipLaunchKernelGGL(attn_lastq_x_fwd_kernel, dim3(grid), dim3(256), smem, s, *a);
  RG_CHECK_LAUNCH();
  return 0;
}

extern "C" int rg_attn_lastq_x_fwd(const rg_lastq_x_args* a, void* stream) { return lx_launch(a, false, (hipStream_t)stream); }
extern "C" int rg_attn_lastq_x_bwd(const rg_lastq_x_args* a, void* stream) { return lx_launch(a, true, (hipStream_t)stream); }
// f32 form: every tensor of the argument block (x, qlast, wk, wv, ctx, dctx, dx, dq, ym_*, xbar, dqp) is f32 -- the f32 and bf16x3 tiers
extern "C" int rg_attn_lastq_xf_fwd(const rg_lastq_x_args* a, void* stream) { return lx_launch(a, false, (hipStream_t)stream, true); }
extern "C" int rg_attn_lastq_xf_bwd(const rg_lastq_x_args* a, void* stream) { return lx_launch(a, true, (hipStream_t)stream, true); }
