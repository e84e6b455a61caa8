// ffn_shape_probe.hip -- round 6, VERDICT r5 item 1b: what would the fused block's two FFN products gain from v_mfma_f32_32x32x16_bf16
// (an MFMA that holds the SIMD's vector issue for 8 of its 32 cycles) instead of v_mfma_f32_16x16x32_bf16 (8 of its 16)?
//
// Not the product kernel: a stand-alone kernel with the product's FFN STRUCTURE and nothing else -- 64-token tiles, four waves, wave w owns
// output features 32 w .. 32 w + 31, d_ff streamed in 128-wide chunks; per chunk  h = y W1c^T + b1  (activation fragments from an LDS
// tile, weight fragments from L2), dropout (p = 0.5, one hash word per 32 elements) + tanh-GELU (exp2 + rcp) + bf16 conversion +
// ds_write of the g tile, barrier, out += g W2c^T -- instantiated with both MFMA shapes on the same data.  Per chunk and wave: 64 MFMAs of
// 16x16x32 or 32 of 32x32x16, the same 32 activation fragments read from LDS, the same 16 weight fragments, the same 32 GELUs per lane.
// Outputs of the two shapes are compared (same products, different accumulation grouping: equal to f32 rounding).
//
// Build + run:  hipcc --offload-arch=gfx950 -O3 -o tools/ffn_shape_probe_bin tools/ffn_shape_probe.hip && tools/ffn_shape_probe_bin
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <stdlib.h>

#include <algorithm>
#include <cmath>
#include <vector>

#define CK(x)                                                                      \
  do {                                                                             \
    hipError_t e_ = (x);                                                           \
    if (e_ != hipSuccess) {                                                        \
      fprintf(stderr, "%s:%d: %s\n", __FILE__, __LINE__, hipGetErrorString(e_));   \
      exit(1);                                                                     \
    }                                                                              \
  } while (0)

typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4_t;

constexpr int D = 128, DFF = 512, TM = 64, LDR = 136;       // LDS row stride in bf16: 272 B (16-byte chunks rotate through the banks)

__device__ __forceinline__ unsigned int hash32(unsigned int seed, unsigned int x) {
  x ^= seed;
  x ^= x >> 16; x *= 0x21f0aaadu;
  x ^= x >> 15; x *= 0x735a2d97u;
  x ^= x >> 15;
  return x;
}
__device__ __forceinline__ float gelu_fast(float x) {
  const float c = 0.7978845608028654f, k1 = -2.f * 1.4426950408889634f * c, k3 = k1 * 0.044715f;
  const float e = __builtin_amdgcn_exp2f(x * fmaf(x * x, k3, k1));
  return x * __builtin_amdgcn_rcpf(1.f + e);
}
__device__ __forceinline__ void lds_barrier() { __syncthreads(); }

// VWORK: 1 = the product's per-element work (dropout + GELU), 0 = conversion only (how much of the time is the VALU work at all)
template <bool BIG, int VWORK>
__global__ void __launch_bounds__(256, 2) ffn_kernel(const __bf16* __restrict__ Y, const __bf16* __restrict__ W1, const float* __restrict__ b1,
                                                     const __bf16* __restrict__ W2, const float* __restrict__ b2, __bf16* __restrict__ out, int ntiles,
                                                     unsigned int seed) {
  __shared__ __align__(16) __bf16 Ay[TM * LDR];
  __shared__ __align__(16) __bf16 Ag[TM * LDR];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, n0 = 32 * wave;
  for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const int m0 = tile * TM;
    lds_barrier();                                             // previous tile's readers of Ay / Ag are done
    for (int i = tid; i < TM * (D / 8); i += 256) {            // y tile: 64 rows x 16 chunks of 16 B
      const int r = i >> 4, c = i & 15;
      *reinterpret_cast<bf16x8_t*>(Ay + r * LDR + 8 * c) = *reinterpret_cast<const bf16x8_t*>(Y + (size_t)(m0 + r) * D + 8 * c);
    }
    lds_barrier();
    if constexpr (!BIG) {
      const int li = lane & 15, lg = lane >> 4;
      f32x4 acc2[2][4];
#pragma unroll
      for (int ct = 0; ct < 2; ++ct)
#pragma unroll
        for (int rt = 0; rt < 4; ++rt)
#pragma unroll
          for (int r = 0; r < 4; ++r) acc2[ct][rt][r] = b2[n0 + ct * 16 + 4 * lg + r];
#pragma unroll 1
      for (int ch = 0; ch < DFF / 128; ++ch) {
        bf16x8_t w1f[4][2], w2f[4][2];
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
#pragma unroll
          for (int ct = 0; ct < 2; ++ct) {
            w1f[ks][ct] = *reinterpret_cast<const bf16x8_t*>(W1 + (size_t)(ch * 128 + n0 + ct * 16 + li) * D + ks * 32 + 8 * lg);
            w2f[ks][ct] = *reinterpret_cast<const bf16x8_t*>(W2 + (size_t)(n0 + ct * 16 + li) * DFF + ch * 128 + ks * 32 + 8 * lg);
          }
        f32x4 h[2][4];
#pragma unroll
        for (int ct = 0; ct < 2; ++ct)
#pragma unroll
          for (int rt = 0; rt < 4; ++rt)
#pragma unroll
            for (int r = 0; r < 4; ++r) h[ct][rt][r] = b1[ch * 128 + n0 + ct * 16 + 4 * lg + r];
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
#pragma unroll
          for (int rt = 0; rt < 4; ++rt) {
            const bf16x8_t bf = *reinterpret_cast<const bf16x8_t*>(Ay + (rt * 16 + li) * LDR + ks * 32 + 8 * lg);
            h[0][rt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w1f[ks][0], bf, h[0][rt], 0, 0, 0);
            h[1][rt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w1f[ks][1], bf, h[1][rt], 0, 0, 0);
          }
        if (ch > 0) lds_barrier();                             // readers of the previous g chunk are done
#pragma unroll
        for (int rt = 0; rt < 4; ++rt) {
          const unsigned int row = (unsigned int)(m0 + rt * 16 + li);
          const unsigned int w = VWORK ? hash32(seed, row * (DFF / 32) + ch * 4 + wave) : 0u;
#pragma unroll
          for (int ct = 0; ct < 2; ++ct) {
            bf16x4_t o;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              float v = h[ct][rt][r];
              if (VWORK) {
                v *= ((w >> (ct * 16 + 4 * lg + r)) & 1u) ? 2.f : 0.f;
                v = gelu_fast(v);
              }
              o[r] = (__bf16)v;
            }
            *reinterpret_cast<bf16x4_t*>(Ag + (rt * 16 + li) * LDR + n0 + ct * 16 + 4 * lg) = o;
          }
        }
        lds_barrier();
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
#pragma unroll
          for (int rt = 0; rt < 4; ++rt) {
            const bf16x8_t gf = *reinterpret_cast<const bf16x8_t*>(Ag + (rt * 16 + li) * LDR + ks * 32 + 8 * lg);
            acc2[0][rt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w2f[ks][0], gf, acc2[0][rt], 0, 0, 0);
            acc2[1][rt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w2f[ks][1], gf, acc2[1][rt], 0, 0, 0);
          }
      }
#pragma unroll
      for (int ct = 0; ct < 2; ++ct)
#pragma unroll
        for (int rt = 0; rt < 4; ++rt) {
          bf16x4_t o;
#pragma unroll
          for (int r = 0; r < 4; ++r) o[r] = (__bf16)acc2[ct][rt][r];
          *reinterpret_cast<bf16x4_t*>(out + (size_t)(m0 + rt * 16 + li) * D + n0 + ct * 16 + 4 * lg) = o;
        }
    } else {
      // 32x32x16: A = weights [32 features][16 k] (lane: feature lane % 32, k 8 (lane / 32) ..), B = activations [16 k][32 tokens]
      // (lane: token lane % 32, same k); C [32 features][32 tokens]: c[j] = feature 8 (j / 4) + 4 (lane / 32) + j % 4, token lane % 32
      const int l32 = lane & 31, lh = lane >> 5;
      f32x16 acc2[2];
#pragma unroll
      for (int t2 = 0; t2 < 2; ++t2)
#pragma unroll
        for (int j = 0; j < 16; ++j) acc2[t2][j] = b2[n0 + 8 * (j >> 2) + 4 * lh + (j & 3)];
#pragma unroll 1
      for (int ch = 0; ch < DFF / 128; ++ch) {
        bf16x8_t w1f[8], w2f[8];
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) {
          w1f[ks] = *reinterpret_cast<const bf16x8_t*>(W1 + (size_t)(ch * 128 + n0 + l32) * D + ks * 16 + 8 * lh);
          w2f[ks] = *reinterpret_cast<const bf16x8_t*>(W2 + (size_t)(n0 + l32) * DFF + ch * 128 + ks * 16 + 8 * lh);
        }
        f32x16 h[2];
#pragma unroll
        for (int t2 = 0; t2 < 2; ++t2)
#pragma unroll
          for (int j = 0; j < 16; ++j) h[t2][j] = b1[ch * 128 + n0 + 8 * (j >> 2) + 4 * lh + (j & 3)];
#pragma unroll
        for (int ks = 0; ks < 8; ++ks)
#pragma unroll
          for (int t2 = 0; t2 < 2; ++t2) {
            const bf16x8_t bf = *reinterpret_cast<const bf16x8_t*>(Ay + (t2 * 32 + l32) * LDR + ks * 16 + 8 * lh);
            h[t2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w1f[ks], bf, h[t2], 0, 0, 0);
          }
        if (ch > 0) lds_barrier();
#pragma unroll
        for (int t2 = 0; t2 < 2; ++t2) {
          const unsigned int row = (unsigned int)(m0 + t2 * 32 + l32);
          const unsigned int w = VWORK ? hash32(seed, row * (DFF / 32) + ch * 4 + wave) : 0u;
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            bf16x4_t o;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              float v = h[t2][4 * q + r];
              if (VWORK) {
                v *= ((w >> (8 * q + 4 * lh + r)) & 1u) ? 2.f : 0.f;
                v = gelu_fast(v);
              }
              o[r] = (__bf16)v;
            }
            *reinterpret_cast<bf16x4_t*>(Ag + (t2 * 32 + l32) * LDR + n0 + 8 * q + 4 * lh) = o;
          }
        }
        lds_barrier();
#pragma unroll
        for (int ks = 0; ks < 8; ++ks)
#pragma unroll
          for (int t2 = 0; t2 < 2; ++t2) {
            const bf16x8_t gf = *reinterpret_cast<const bf16x8_t*>(Ag + (t2 * 32 + l32) * LDR + ks * 16 + 8 * lh);
            acc2[t2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w2f[ks], gf, acc2[t2], 0, 0, 0);
          }
      }
#pragma unroll
      for (int t2 = 0; t2 < 2; ++t2)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          bf16x4_t o;
#pragma unroll
          for (int r = 0; r < 4; ++r) o[r] = (__bf16)acc2[t2][4 * q + r];
          *reinterpret_cast<bf16x4_t*>(out + (size_t)(m0 + t2 * 32 + l32) * D + n0 + 8 * q + 4 * lh) = o;
        }
    }
  }
}

template <typename F>
static double time_us(F launch, int reps = 9) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  launch();
  CK(hipDeviceSynchronize());
  std::vector<float> ts;
  for (int r = 0; r < reps; ++r) {
    CK(hipEventRecord(e0, 0));
    launch();
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    ts.push_back(ms);
  }
  std::sort(ts.begin(), ts.end());
  return ts[ts.size() / 2] * 1e3;
}

static unsigned short f2bf(float f) {
  unsigned int u;
  memcpy(&u, &f, 4);
  return (unsigned short)((u + 0x7FFF + ((u >> 16) & 1)) >> 16);
}
static float bf2f(unsigned short h) {
  unsigned int u = (unsigned int)h << 16;
  float f;
  memcpy(&f, &u, 4);
  return f;
}

int main() {
  const int M = 4096 * 200 * 56 / 100 / TM * TM;            // the live rows of one bench launch (56 % of B L = 819 200 positions)
  const int ntiles = M / TM;
  std::vector<unsigned short> hy((size_t)M * D), hw1((size_t)DFF * D), hw2((size_t)D * DFF);
  std::vector<float> hb1(DFF), hb2(D);
  uint64_t st = 0x9E3779B97F4A7C15ull;
  auto rnd = [&]() { st ^= st << 13; st ^= st >> 7; st ^= st << 17; return (float)((st >> 11) * (1.0 / 9007199254740992.0)) * 2.f - 1.f; };
  for (auto& v : hy) v = f2bf(rnd());
  for (auto& v : hw1) v = f2bf(rnd() * 0.09f);
  for (auto& v : hw2) v = f2bf(rnd() * 0.045f);
  for (auto& v : hb1) v = rnd() * 0.05f;
  for (auto& v : hb2) v = rnd() * 0.05f;
  __bf16 *Y, *W1, *W2, *o16, *o32;
  float *b1, *b2;
  CK(hipMalloc(&Y, hy.size() * 2)); CK(hipMalloc(&W1, hw1.size() * 2)); CK(hipMalloc(&W2, hw2.size() * 2));
  CK(hipMalloc(&o16, hy.size() * 2)); CK(hipMalloc(&o32, hy.size() * 2));
  CK(hipMalloc(&b1, DFF * 4)); CK(hipMalloc(&b2, D * 4));
  CK(hipMemcpy(Y, hy.data(), hy.size() * 2, hipMemcpyHostToDevice));
  CK(hipMemcpy(W1, hw1.data(), hw1.size() * 2, hipMemcpyHostToDevice));
  CK(hipMemcpy(W2, hw2.data(), hw2.size() * 2, hipMemcpyHostToDevice));
  CK(hipMemcpy(b1, hb1.data(), DFF * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(b2, hb2.data(), D * 4, hipMemcpyHostToDevice));
  const int grid = 512;
  const double flops = 2.0 * 2.0 * M * D * DFF;
  printf("FFN structure of the fused block, %d live rows (%d tiles of 64 tokens), d_model 128, d_ff 512, grid %d x 256 threads, 2 workgroups per CU\n", M, ntiles, grid);
  for (int rep = 0; rep < 2; ++rep) {
    const double a = time_us([&] { ffn_kernel<false, 1><<<grid, 256>>>(Y, W1, b1, W2, b2, o16, ntiles, 7u); });
    const double b = time_us([&] { ffn_kernel<true, 1><<<grid, 256>>>(Y, W1, b1, W2, b2, o32, ntiles, 7u); });
    const double a0 = time_us([&] { ffn_kernel<false, 0><<<grid, 256>>>(Y, W1, b1, W2, b2, o16, ntiles, 7u); });
    const double b0 = time_us([&] { ffn_kernel<true, 0><<<grid, 256>>>(Y, W1, b1, W2, b2, o32, ntiles, 7u); });
    printf("round %d  dropout + GELU : 16x16x32 %7.1f us (%6.1f TFLOP/s)   32x32x16 %7.1f us (%6.1f TFLOP/s)   ratio %.3f\n", rep, a, flops / a * 1e-6, b,
           flops / b * 1e-6, b / a);
    printf("round %d  conversion only: 16x16x32 %7.1f us (%6.1f TFLOP/s)   32x32x16 %7.1f us (%6.1f TFLOP/s)   ratio %.3f\n", rep, a0, flops / a0 * 1e-6, b0,
           flops / b0 * 1e-6, b0 / a0);
  }
  // same values from both shapes (with the VALU work)
  ffn_kernel<false, 1><<<grid, 256>>>(Y, W1, b1, W2, b2, o16, ntiles, 7u);
  ffn_kernel<true, 1><<<grid, 256>>>(Y, W1, b1, W2, b2, o32, ntiles, 7u);
  CK(hipDeviceSynchronize());
  std::vector<unsigned short> r16(hy.size()), r32(hy.size());
  CK(hipMemcpy(r16.data(), o16, hy.size() * 2, hipMemcpyDeviceToHost));
  CK(hipMemcpy(r32.data(), o32, hy.size() * 2, hipMemcpyDeviceToHost));
  double worst = 0, scale = 0;
  size_t differ = 0;
  for (size_t i = 0; i < r16.size(); ++i) {
    const double x = bf2f(r16[i]), y = bf2f(r32[i]);
    worst = std::max(worst, std::fabs(x - y));
    scale = std::max(scale, std::fabs(x));
    differ += r16[i] != r32[i];
  }
  printf("outputs of the two shapes: max |difference| %.3g of max |value| %.3g; %zu of %zu bf16 outputs differ (accumulation grouping)\n", worst, scale, differ,
         r16.size());
  return 0;
}
