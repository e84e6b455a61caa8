// Input side of the hot path: batch assembly and negative sampling on the device.
//
// The reference does both per sample on the host (GURU/data/data_loader.py): seq_padding (:25-36) left-pads a
// user's item sequence, appends the EOS id and derives the shifted decoder input / target; __getitem__ (:276-316)
// allocates a length-V weight vector, zeroes the user's own items and draws L*k negatives with
// torch.multinomial(weights, num, replacement=True) -- V = 2M and k = 1024 make that the slowest stage by far.
// Here a user is a CSR row of item ids plus a CSR row of its sorted, unique exclusion set; one thread per output
// element does the padding arithmetic, one thread per draw maps a counter-based hash to the u-th ALLOWED item
// (exact uniform over 1..V minus the exclusions, no rejection loop) or, for frequency^0.75 sampling, walks an
// alias table and rejects excluded draws.
#include "rg_common.hip.h"
#include "../../include/recguru_hip.h"

#define SB 256

// enc_in = leftpad0(last L_enc-1 items) + [eos];  dec_in[t] = ([0,0] + enc_in[:-2])[-L_dec:];  dec_out likewise
// from enc_in[1:-1]   (data_loader.py:25-36)
__global__ __launch_bounds__(SB) void assemble_batch_kernel(const int64_t* __restrict__ items, const int64_t* __restrict__ offsets,
                                                           const int64_t* __restrict__ users, int B, int Le, int Ld, int64_t eos,
                                                           int64_t* __restrict__ enc_in, int64_t* __restrict__ dec_in,
                                                           int64_t* __restrict__ dec_out) {
  const long long total = (long long)B * (Le + 2 * Ld);
  for (long long i = (long long)blockIdx.x * SB + threadIdx.x; i < total; i += (long long)gridDim.x * SB) {
    const int b = (int)(i / (Le + 2 * Ld));
    const int c = (int)(i - (long long)b * (Le + 2 * Ld));
    const int64_t u = users[b];
    const int64_t s0 = offsets[u], n = offsets[u + 1] - s0;
    // enc(t): t in [0, Le); position Le-1 is eos, positions before hold the last min(n, Le-1) items, right-aligned
    auto enc = [&](int t) -> int64_t {
      if (t < 0) return 0;
      if (t == Le - 1) return eos;
      const int64_t k = (int64_t)t - (Le - 1) + n;      // index into the user's sequence
      return k >= 0 ? items[s0 + k] : 0;
    };
    if (c < Le) {
      enc_in[(size_t)b * Le + c] = enc(c);
    } else if (c < Le + Ld) {
      const int t = c - Le;                              // dec_in has length Le before the [-Ld:] cut; element j = enc(j-2)
      const int j = t + (Le - Ld);
      dec_in[(size_t)b * Ld + t] = j < 2 ? 0 : enc(j - 2);
    } else {
      const int t = c - Le - Ld;
      const int j = t + (Le - Ld);
      dec_out[(size_t)b * Ld + t] = j < 2 ? 0 : enc(j - 1);
    }
  }
}

__device__ __forceinline__ unsigned int draw32(unsigned long long seed, unsigned long long ctr) {
  unsigned int lo = (unsigned int)ctr, hi = (unsigned int)(ctr >> 32);
  unsigned int x = rg_hash((unsigned int)seed, lo);
  x = rg_hash((unsigned int)(seed >> 32) ^ 0x9E3779B9u, x ^ hi);
  return x;
}

// number of exclusions <= the candidate once shifted: smallest j with ex[j] - j >= u  (ex sorted, unique, in 1..V)
__device__ __forceinline__ int64_t nth_allowed(const int64_t* __restrict__ ex, int m, int64_t u) {
  int lo = 0, hi = m;                                    // invariant: ex[lo-1] - (lo-1) < u <= ... ; answer j in [lo, hi]
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    if (ex[mid] - mid >= u + 1) hi = mid; else lo = mid + 1;   // ex[mid] - mid > u  <=>  the u-th allowed item is below ex[mid]
  }
  return u + lo;
}

__global__ __launch_bounds__(SB) void sample_uniform_kernel(const int64_t* __restrict__ excl, const int64_t* __restrict__ excl_off,
                                                           const int64_t* __restrict__ users, int B, int n, int64_t V,
                                                           unsigned long long seed, int64_t* __restrict__ out) {
  const long long total = (long long)B * n;
  for (long long i = (long long)blockIdx.x * SB + threadIdx.x; i < total; i += (long long)gridDim.x * SB) {
    const int b = (int)(i / n);
    const int64_t usr = users[b];
    const int64_t e0 = excl_off[usr];
    const int m = (int)(excl_off[usr + 1] - e0);
    const unsigned long long range = (unsigned long long)(V - m);        // allowed items; host guarantees >= 1
    // 64-bit multiply-shift of two hashes: uniform on [0, range) up to 2^-32 relative bias
    const unsigned long long r = ((unsigned long long)draw32(seed, 2ull * (unsigned long long)i) << 32) | draw32(seed, 2ull * (unsigned long long)i + 1);
    const int64_t u = (int64_t)__umul64hi(r, range) + 1;                  // 1..range
    out[i] = nth_allowed(excl + e0, m, u);
  }
}

// Walker alias draw over ids 0..V (prob[j] = acceptance of slot j, alias[j] its alternative), redrawn while the id is
// excluded or 0; after 64 rejections falls back to the uniform-over-allowed draw (unreachable in practice).
__global__ __launch_bounds__(SB) void sample_alias_kernel(const float* __restrict__ prob, const int* __restrict__ alias, int64_t slots,
                                                         const int64_t* __restrict__ excl, const int64_t* __restrict__ excl_off,
                                                         const int64_t* __restrict__ users, int B, int n, int64_t V,
                                                         unsigned long long seed, int64_t* __restrict__ out) {
  const long long total = (long long)B * n;
  for (long long i = (long long)blockIdx.x * SB + threadIdx.x; i < total; i += (long long)gridDim.x * SB) {
    const int b = (int)(i / n);
    const int64_t usr = users[b];
    const int64_t e0 = excl_off[usr];
    const int m = (int)(excl_off[usr + 1] - e0);
    int64_t id = -1;
    for (int tr = 0; tr < 64 && id < 0; ++tr) {
      const unsigned long long c = ((unsigned long long)i * 64ull + tr) * 2ull;
      const unsigned int h0 = draw32(seed, c), h1 = draw32(seed, c + 1);
      const int64_t slot = (int64_t)(((unsigned long long)h0 * (unsigned long long)slots) >> 32);
      const float f = (float)(h1 >> 8) * (1.f / 16777216.f);
      const int64_t cand = f < prob[slot] ? slot : (int64_t)alias[slot];
      if (cand < 1 || cand > V) continue;
      int lo = 0, hi = m;                                 // member of the sorted exclusion set?
      while (lo < hi) { const int mid = (lo + hi) >> 1; if (excl[e0 + mid] < cand) lo = mid + 1; else hi = mid; }
      if (lo < m && excl[e0 + lo] == cand) continue;
      id = cand;
    }
    if (id < 0) {
      const unsigned long long r = ((unsigned long long)draw32(seed ^ 0xA5A5A5A5ull, 2ull * (unsigned long long)i) << 32) | draw32(seed ^ 0xA5A5A5A5ull, 2ull * (unsigned long long)i + 1);
      id = nth_allowed(excl + e0, m, (int64_t)__umul64hi(r, (unsigned long long)(V - m)) + 1);
    }
    out[i] = id;
  }
}

static int sgrid(long long n) {
  long long g = (n + SB - 1) / SB;
  return (int)(g < 1 ? 1 : (g > 256 * 16 ? 256 * 16 : g));
}

extern "C" int rg_assemble_batch(const int64_t* items, const int64_t* offsets, const int64_t* users, int B, int L_enc, int L_dec,
                                 int64_t eos, int64_t* enc_in, int64_t* dec_in, int64_t* dec_out, void* stream) {
  if (B <= 0) return 0;
  if (L_enc < 1 || L_dec < 1 || L_dec > L_enc) return rg_set_error_msg(RG_ERR_INVALID, "assemble_batch: need 1 <= L_dec <= L_enc");
  hipLaunchKernelGGL(assemble_batch_kernel, dim3(sgrid((long long)B * (L_enc + 2 * L_dec))), dim3(SB), 0, (hipStream_t)stream,
                     items, offsets, users, B, L_enc, L_dec, eos, enc_in, dec_in, dec_out);
  RG_CHECK_LAUNCH();
  return 0;
}

extern "C" int rg_sample_negatives(const int64_t* excl, const int64_t* excl_off, const int64_t* users, int B, int n, int64_t V,
                                   unsigned long long seed, int64_t* out, void* stream) {
  if (B <= 0 || n <= 0) return 0;
  if (V < 1) return rg_set_error_msg(RG_ERR_INVALID, "sample_negatives: empty catalogue");
  hipLaunchKernelGGL(sample_uniform_kernel, dim3(sgrid((long long)B * n)), dim3(SB), 0, (hipStream_t)stream, excl, excl_off, users, B, n, V, seed, out);
  RG_CHECK_LAUNCH();
  return 0;
}

extern "C" int rg_sample_negatives_alias(const float* prob, const int* alias, int64_t slots, const int64_t* excl, const int64_t* excl_off,
                                         const int64_t* users, int B, int n, int64_t V, unsigned long long seed, int64_t* out, void* stream) {
  if (B <= 0 || n <= 0) return 0;
  if (V < 1 || slots < 1) return rg_set_error_msg(RG_ERR_INVALID, "sample_negatives_alias: empty catalogue");
  hipLaunchKernelGGL(sample_alias_kernel, dim3(sgrid((long long)B * n)), dim3(SB), 0, (hipStream_t)stream, prob, alias, slots, excl, excl_off, users,
                     B, n, V, seed, out);
  RG_CHECK_LAUNCH();
  return 0;
}
