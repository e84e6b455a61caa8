// peaks.hip -- the denominators of bench.py's rooflines re-derived ON THE BOX:
//
//   * HBM: device-to-device copy (hipMemcpyAsync and a float4 copy kernel), read-only and write-only streams and a triad
//     over buffers far beyond the 256 MiB Infinity Cache;
//   * random row gather: uniformly drawn rows of 256 B / 512 B / 1 KiB from a 1 GiB table (the config-5 embedding table is
//     2 M x 256 bf16 = 1 GiB of 512-B rows), as a read-only gather (rows summed in registers) and as the embedding kernel's
//     gather-copy (row read + contiguous row write + 8-B id read);
//   * matrix pipe: v_mfma_f32_16x16x32_bf16 issue loop, 4 independent accumulators per wave, at 1 / 2 / 4 waves per SIMD.
//
// Build + run (tools/peaks.py does both):   hipcc --offload-arch=gfx950 -O3 -o tools/peaks_probe tools/peaks.hip && tools/peaks_probe
// Prints one "name value unit" line per probe and a JSON object on the last line.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include <algorithm>
#include <string>
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
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;

// ---- streams -------------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) copy_kernel(const float4* __restrict__ a, float4* __restrict__ b, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) b[i] = a[i];
}
__global__ void __launch_bounds__(256) read_kernel(const float4* __restrict__ a, float* __restrict__ out, size_t n) {
  float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const float4 v = a[i];
    s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
  }
  if (s.x + s.y + s.z + s.w == 12345.678f) out[0] = s.x;      // (never true: keeps the loads)
}
__global__ void __launch_bounds__(256) write_kernel(float4* __restrict__ b, size_t n, float x) {
  const float4 v = make_float4(x, x, x, x);
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) b[i] = v;
}
__global__ void __launch_bounds__(256) triad_kernel(const float4* __restrict__ b, const float4* __restrict__ c, float4* __restrict__ a,
                                                    size_t n, float s) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const float4 x = b[i], y = c[i];
    a[i] = make_float4(x.x + s * y.x, x.y + s * y.y, x.z + s * y.z, x.w + s * y.w);
  }
}

// ---- tuned streams (round 6): U float4 per lane in flight (block-contiguous chunks of U * 256 float4, every
// wave instruction a full 1 KiB line run), optional nontemporal loads / stores, swept over workgroups per CU by the harness.  The best
// of each family is the ceiling bench.py prices kernels against ("hbm_*_best").
template <int U, bool NT>
__global__ void __launch_bounds__(256) copy_u_kernel(const f32x4* __restrict__ a, f32x4* __restrict__ b, size_t n) {
  const size_t chunk = (size_t)U * 256;
  for (size_t base = (size_t)blockIdx.x * chunk; base + chunk <= n; base += (size_t)gridDim.x * chunk) {
    f32x4 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) v[u] = NT ? __builtin_nontemporal_load(a + base + u * 256 + threadIdx.x) : a[base + u * 256 + threadIdx.x];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (NT) __builtin_nontemporal_store(v[u], b + base + u * 256 + threadIdx.x);
      else b[base + u * 256 + threadIdx.x] = v[u];
    }
  }
}
template <int U, bool NT>
__global__ void __launch_bounds__(256) read_u_kernel(const f32x4* __restrict__ a, float* __restrict__ out, size_t n) {
  const size_t chunk = (size_t)U * 256;
  f32x4 s = {0.f, 0.f, 0.f, 0.f};
  for (size_t base = (size_t)blockIdx.x * chunk; base + chunk <= n; base += (size_t)gridDim.x * chunk) {
    f32x4 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) v[u] = NT ? __builtin_nontemporal_load(a + base + u * 256 + threadIdx.x) : a[base + u * 256 + threadIdx.x];
#pragma unroll
    for (int u = 0; u < U; ++u) s += v[u];
  }
  if (s[0] + s[1] + s[2] + s[3] == 12345.678f) out[0] = s[0];
}
template <int U, bool NT>
__global__ void __launch_bounds__(256) write_u_kernel(f32x4* __restrict__ b, size_t n, float x) {
  const size_t chunk = (size_t)U * 256;
  const f32x4 v = {x, x, x, x};
  for (size_t base = (size_t)blockIdx.x * chunk; base + chunk <= n; base += (size_t)gridDim.x * chunk) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (NT) __builtin_nontemporal_store(v, b + base + u * 256 + threadIdx.x);
      else b[base + u * 256 + threadIdx.x] = v;
    }
  }
}
template <int U, bool NT>
__global__ void __launch_bounds__(256) triad_u_kernel(const f32x4* __restrict__ b, const f32x4* __restrict__ c, f32x4* __restrict__ a, size_t n, float s) {
  const size_t chunk = (size_t)U * 256;
  for (size_t base = (size_t)blockIdx.x * chunk; base + chunk <= n; base += (size_t)gridDim.x * chunk) {
    f32x4 x[U], y[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      x[u] = NT ? __builtin_nontemporal_load(b + base + u * 256 + threadIdx.x) : b[base + u * 256 + threadIdx.x];
      y[u] = NT ? __builtin_nontemporal_load(c + base + u * 256 + threadIdx.x) : c[base + u * 256 + threadIdx.x];
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const f32x4 r = x[u] + s * y[u];
      if (NT) __builtin_nontemporal_store(r, a + base + u * 256 + threadIdx.x);
      else a[base + u * 256 + threadIdx.x] = r;
    }
  }
}

// ---- random row gather ---------------------------------------------------------------------------------------------------
// A row of RB bytes is RB / 16 lanes x 16 B; a wave instruction therefore fetches 64 * 16 / RB rows.  Every lane group keeps U
// rows in flight.  COPY: the row is written to out[t] (the embedding kernel's shape); otherwise it is summed in registers.
template <int RB, int U, bool COPY>
__global__ void __launch_bounds__(256) gather_kernel(const float4* __restrict__ table, const int* __restrict__ ids, float4* __restrict__ out,
                                                     float* __restrict__ sink, int n) {
  constexpr int LPR = RB / 16;                 // lanes per row
  const int lane_in_row = threadIdx.x % LPR;
  const int groups = (gridDim.x * blockDim.x) / LPR;
  const int g = (blockIdx.x * blockDim.x + threadIdx.x) / LPR;
  float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int t0 = g * U; t0 < n; t0 += groups * U) {
    int id[U];
#pragma unroll
    for (int u = 0; u < U; ++u) id[u] = ids[min(t0 + u, n - 1)];
    float4 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) v[u] = table[(size_t)id[u] * LPR + lane_in_row];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (COPY) {
        if (t0 + u < n) out[(size_t)(t0 + u) * LPR + lane_in_row] = v[u];
      } else {
        s.x += v[u].x; s.y += v[u].y; s.z += v[u].z; s.w += v[u].w;
      }
    }
  }
  if (!COPY && s.x + s.y + s.z + s.w == 12345.678f) sink[0] = s.x;
}

// ---- MFMA issue loop -----------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) mfma_kernel(float* __restrict__ out, int iters) {
  bf16x8_t a, b;
  for (int j = 0; j < 8; ++j) { a[j] = (__bf16)(0.001f * (threadIdx.x & 7)); b[j] = (__bf16)(0.002f * (j + 1)); }
  f32x4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
  for (int i = 0; i < iters; ++i) {
    c0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c0, 0, 0, 0);
    c1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c1, 0, 0, 0);
    c2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c2, 0, 0, 0);
    c3 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c3, 0, 0, 0);
  }
  const f32x4 c = c0 + c1 + c2 + c3;
  if (c[0] == 12345.678f) out[threadIdx.x] = c[0] + c[1] + c[2] + c[3];
}

// ---- VALU port: full-rate fma, packed fma, transcendental (exp2 / rcp), and a matrix instruction with 0 / 2 / 4 / 8 independent VALU
// instructions of the SAME wave behind it (what one wave can hide under its own MFMAs) --------------------------------------------------
template <int KIND>      // 0: v_fma_f32   1: v_pk_fma_f32   2: v_exp_f32   3: v_rcp_f32
__global__ void __launch_bounds__(256) valu_kernel(float* __restrict__ out, int iters) {
  float x[8];
  for (int j = 0; j < 8; ++j) x[j] = 0.5f + 0.001f * (threadIdx.x & 31) + 0.01f * j;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      if (KIND == 0) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(x[j]) : "v"(0.999f));
      if (KIND == 2) asm volatile("v_exp_f32 %0, %0" : "+v"(x[j]));
      if (KIND == 3) asm volatile("v_rcp_f32 %0, %0" : "+v"(x[j]));
    }
    if (KIND == 1) {
      typedef __attribute__((ext_vector_type(2))) float f2;
#pragma unroll
      for (int j = 0; j < 8; j += 2) {
        f2 v = {x[j], x[j + 1]};
        const f2 k = {0.999f, 0.999f};
        asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(v) : "v"(k));
        asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(v) : "v"(k));
        x[j] = v.x; x[j + 1] = v.y;
      }
    }
  }
  float s = 0.f;
  for (int j = 0; j < 8; ++j) s += x[j];
  if (s == 12345.678f) out[threadIdx.x] = s;
}
template <int NV>        // per iteration: 4 MFMAs (independent accumulators), each followed by NV independent v_fma_f32
__global__ void __launch_bounds__(256) mfma_valu_kernel(float* __restrict__ out, int iters) {
  bf16x8_t a, b;
  for (int j = 0; j < 8; ++j) { a[j] = (__bf16)(0.001f * (threadIdx.x & 7)); b[j] = (__bf16)(0.002f * (j + 1)); }
  f32x4 c[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
  float x[8];
  for (int j = 0; j < 8; ++j) x[j] = 0.5f + 0.001f * (threadIdx.x & 31) + 0.01f * j;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(c[m]) : "v"(a), "v"(b));
#pragma unroll
      for (int j = 0; j < NV; ++j) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(x[j & 7]) : "v"(0.999f));
    }
  }
  float s = c[0][0] + c[1][0] + c[2][0] + c[3][0];
  for (int j = 0; j < 8; ++j) s += x[j];
  if (s == 12345.678f) out[threadIdx.x] = s;
}

// (round 6; MI355X_MICROARCH.md "price of one filler beside MFMAs" and "vector-instruction ISSUE cost")
// FILL: 0 v_fma_f32, 1 v_pk_fma_f32, 2 v_pk_mul_f32, 3 v_pk_add_f32, 4 v_exp_f32, 5 v_cvt_pk_bf16_f32;  BIG: v_mfma_f32_32x32x16_bf16
// (twice the FLOPs of a 16x16x32 per instruction) instead of v_mfma_f32_16x16x32_bf16.  4 independent accumulators either way.
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(2))) float pf2;
template <int NV, int FILL, bool BIG>
__global__ void __launch_bounds__(256) coexec_kernel(float* __restrict__ out, int iters) {
  bf16x8_t a, b;
  for (int j = 0; j < 8; ++j) { a[j] = (__bf16)(0.001f * (threadIdx.x & 7)); b[j] = (__bf16)(0.002f * (j + 1)); }
  f32x4 c[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
  f32x16 d[4];
  for (int m = 0; m < 4; ++m) for (int j = 0; j < 16; ++j) d[m][j] = 0.f;
  float x[8];
  pf2 xp[8];
  unsigned int cv[8];
  for (int j = 0; j < 8; ++j) { x[j] = 0.5f + 0.001f * (threadIdx.x & 31) + 0.01f * j; xp[j] = (pf2){x[j], x[j] + 0.1f}; cv[j] = 0; }
  const pf2 kp = {0.999f, 0.999f};
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      if (BIG) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(d[m]) : "v"(a), "v"(b));
      else asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(c[m]) : "v"(a), "v"(b));
#pragma unroll
      for (int j = 0; j < NV; ++j) {
        if (FILL == 0) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(x[j & 7]) : "v"(0.999f));
        if (FILL == 1) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(xp[j & 7]) : "v"(kp));
        if (FILL == 2) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(xp[j & 7]) : "v"(kp));
        if (FILL == 3) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(xp[j & 7]) : "v"(kp));
        if (FILL == 4) asm volatile("v_exp_f32 %0, %0" : "+v"(x[j & 7]));
        if (FILL == 5) asm volatile("v_cvt_pk_bf16_f32 %0, %1, %1" : "=v"(cv[j & 7]) : "v"(x[j & 7]));
      }
    }
  }
  float s = c[0][0] + c[1][0] + c[2][0] + c[3][0] + d[0][0] + d[1][0] + d[2][0] + d[3][0];
  for (int j = 0; j < 8; ++j) s += x[j] + xp[j].x + xp[j].y + __uint_as_float(cv[j]);
  if (s == 12345.678f) out[threadIdx.x] = s;
}

// ---- harness -------------------------------------------------------------------------------------------------------------
struct Res { std::string name; double value; std::string unit; std::string note; };
static std::vector<Res> results;

template <typename F>
static double time_ms(F launch, int reps = 7) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  launch();                                         // warm-up
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
  CK(hipEventDestroy(e0));
  CK(hipEventDestroy(e1));
  return ts[ts.size() / 2];                         // median
}

static void report(const char* name, double v, const char* unit, const char* note) {
  printf("%-44s %10.1f %-8s %s\n", name, v, unit, note);
  fflush(stdout);
  results.push_back({name, v, unit, note});
}

template <int RB, int U>
static void gather_probe(const float4* table, size_t table_bytes, const int* ids, float4* out, float* sink, int n, int grid) {
  char nm[96], note[160];
  const double rows_b = (double)n * RB;
  double ms = time_ms([&] { gather_kernel<RB, U, false><<<grid, 256>>>(table, ids, out, sink, n); });
  snprintf(nm, sizeof nm, "gather_read_%dB_rows_u%d", RB, U);
  snprintf(note, sizeof note, "%d uniformly random %d-B rows of a %.2f GiB table summed in registers, %d rows in flight per lane group", n, RB,
           table_bytes / 1073741824.0, U);
  report(nm, (rows_b + 4.0 * n) / ms * 1e-6, "GB/s", note);
  ms = time_ms([&] { gather_kernel<RB, U, true><<<grid, 256>>>(table, ids, out, sink, n); });
  snprintf(nm, sizeof nm, "gather_copy_%dB_rows_u%d", RB, U);
  snprintf(note, sizeof note, "the same rows copied to a contiguous [n, %d B] output: row read + row write + id read", RB);
  report(nm, (2.0 * rows_b + 4.0 * n) / ms * 1e-6, "GB/s", note);
}

int main(int argc, char** argv) {
  hipDeviceProp_t prop;
  CK(hipGetDeviceProperties(&prop, 0));
  printf("device: %s, %d CUs, clock %.0f MHz, memory clock %.0f MHz, bus %d bit\n", prop.name, prop.multiProcessorCount,
         prop.clockRate * 1e-3, prop.memoryClockRate * 1e-3, prop.memoryBusWidth);
  const int CUS = prop.multiProcessorCount;
  const size_t GB = (size_t)1 << 30;
  const size_t NB = 2 * GB;                      // 2 GiB per stream buffer: 8 x the Infinity Cache
  float4 *a, *b, *c;
  float* sink;
  CK(hipMalloc(&a, NB));
  CK(hipMalloc(&b, NB));
  CK(hipMalloc(&c, NB));
  CK(hipMalloc(&sink, 4096));
  CK(hipMemset(a, 0, NB));
  CK(hipMemset(b, 0, NB));
  CK(hipMemset(c, 0, NB));
  const size_t n4 = NB / 16;
  const int grid = CUS * 8;
  double ms;
  ms = time_ms([&] { CK(hipMemcpyAsync(b, a, NB, hipMemcpyDeviceToDevice, 0)); });
  report("hbm_memcpy_d2d", 2.0 * NB / ms * 1e-6, "GB/s", "hipMemcpyAsync device-to-device, 2 GiB; read + write bytes");
  ms = time_ms([&] { copy_kernel<<<grid, 256>>>(a, b, n4); });
  report("hbm_copy_kernel", 2.0 * NB / ms * 1e-6, "GB/s", "float4 grid-stride copy kernel, 2 GiB; read + write bytes");
  ms = time_ms([&] { read_kernel<<<grid, 256>>>(a, sink, n4); });
  report("hbm_read_only", 1.0 * NB / ms * 1e-6, "GB/s", "float4 read-only stream (register sum), 2 GiB");
  ms = time_ms([&] { write_kernel<<<grid, 256>>>(b, n4, 1.f); });
  report("hbm_write_only", 1.0 * NB / ms * 1e-6, "GB/s", "float4 write-only stream, 2 GiB");
  ms = time_ms([&] { triad_kernel<<<grid, 256>>>(b, c, a, n4, 0.5f); });
  report("hbm_triad", 3.0 * NB / ms * 1e-6, "GB/s", "a = b + s c over 2 GiB buffers; two reads + one write");
  // ---- tuned streams: U in flight x workgroups per CU x nontemporal; the best of each family is the ceiling
  {
    const f32x4* av = reinterpret_cast<const f32x4*>(a);
    const f32x4* cvp = reinterpret_cast<const f32x4*>(c);
    f32x4* bv = reinterpret_cast<f32x4*>(b);
    f32x4* aw = reinterpret_cast<f32x4*>(a);
    double best[4] = {0, 0, 0, 0};
    char bestcfg[4][64] = {"", "", "", ""};
    auto upd = [&](int k, double gbs, int U, int nt, int wpc) {
      if (gbs > best[k]) { best[k] = gbs; snprintf(bestcfg[k], 64, "U=%d %s %d workgroups per CU", U, nt ? "nontemporal" : "plain", wpc); }
    };
#define SWEEP(U)                                                                                                          \
    for (int wpc : {2, 4, 8, 16, 32}) {                                                                                     \
      const int g = CUS * wpc;                                                                                              \
      ms = time_ms([&] { copy_u_kernel<U, false><<<g, 256>>>(av, bv, n4); }, 5);   upd(0, 2.0 * NB / ms * 1e-6, U, 0, wpc);  \
      ms = time_ms([&] { copy_u_kernel<U, true><<<g, 256>>>(av, bv, n4); }, 5);    upd(0, 2.0 * NB / ms * 1e-6, U, 1, wpc);  \
      ms = time_ms([&] { read_u_kernel<U, false><<<g, 256>>>(av, sink, n4); }, 5); upd(1, 1.0 * NB / ms * 1e-6, U, 0, wpc);  \
      ms = time_ms([&] { read_u_kernel<U, true><<<g, 256>>>(av, sink, n4); }, 5);  upd(1, 1.0 * NB / ms * 1e-6, U, 1, wpc);  \
      ms = time_ms([&] { write_u_kernel<U, false><<<g, 256>>>(bv, n4, 1.f); }, 5); upd(2, 1.0 * NB / ms * 1e-6, U, 0, wpc);  \
      ms = time_ms([&] { write_u_kernel<U, true><<<g, 256>>>(bv, n4, 1.f); }, 5);  upd(2, 1.0 * NB / ms * 1e-6, U, 1, wpc);  \
      ms = time_ms([&] { triad_u_kernel<U, false><<<g, 256>>>(bv, cvp, aw, n4, 0.5f); }, 5); upd(3, 3.0 * NB / ms * 1e-6, U, 0, wpc); \
      ms = time_ms([&] { triad_u_kernel<U, true><<<g, 256>>>(bv, cvp, aw, n4, 0.5f); }, 5);  upd(3, 3.0 * NB / ms * 1e-6, U, 1, wpc); \
    }
    SWEEP(1) SWEEP(2) SWEEP(4) SWEEP(8)
#undef SWEEP
    // the same copy with source + destination INSIDE the 256 MiB Infinity Cache (2 x 96 MiB, repeated): the ceiling of a kernel whose inputs were
    // written by the launch before it (in the step most activations are: a consumer follows its producer immediately)
    {
      const size_t n96 = ((size_t)96 << 20) / 16;
      double mb = 0;
      char cfg[64] = "";
      for (int wpc : {8, 16, 32})
        for (int nt = 0; nt < 2; ++nt) {
          auto go = [&] { for (int r = 0; r < 4; ++r) { if (nt) copy_u_kernel<4, true><<<CUS * wpc, 256>>>(av, bv, n96); else copy_u_kernel<4, false><<<CUS * wpc, 256>>>(av, bv, n96); } };
          ms = time_ms(go, 5);
          const double gbs = 4 * 2.0 * n96 * 16 / ms * 1e-6;
          if (gbs > mb) { mb = gbs; snprintf(cfg, sizeof cfg, "U=4 %s %d workgroups per CU", nt ? "nontemporal" : "plain", wpc); }
        }
      char note[200];
      snprintf(note, sizeof note, "float4 copy of 96 MiB to 96 MiB, four times back to back (source and destination stay in the 256 MiB Infinity Cache): %s", cfg);
      report("mall_copy_best", mb, "GB/s", note);
    }
    CK(hipMemset(a, 0, NB));
    const char* nm[4] = {"hbm_copy_best", "hbm_read_best", "hbm_write_best", "hbm_triad_best"};
    for (int k = 0; k < 4; ++k) {
      char note[200];
      snprintf(note, sizeof note, "best of U in {1,2,4,8} float4 in flight per lane x {2..32} workgroups per CU x plain / nontemporal: %s", bestcfg[k]);
      report(nm[k], best[k], "GB/s", note);
    }
  }
  // ---- random rows of a 1 GiB table (buffer a); ids in c, outputs in b
  const int n = 819200 * 2;                       // two bench batches of B * L = 4096 * 200 positions
  std::vector<int> h(n);
  uint64_t st = 0x9E3779B97F4A7C15ull;
  auto rnd = [&]() { st ^= st << 13; st ^= st >> 7; st ^= st << 17; return st; };
  int* ids = reinterpret_cast<int*>(c);
  const size_t TB = GB;
  for (int rb : {256, 512, 1024}) {
    const uint64_t rows = TB / rb;
    for (int i = 0; i < n; ++i) h[i] = (int)(rnd() % rows);
    CK(hipMemcpy(ids, h.data(), (size_t)n * 4, hipMemcpyHostToDevice));
    if (rb == 256) { gather_probe<256, 2>(a, TB, ids, b, sink, n, CUS * 8); gather_probe<256, 4>(a, TB, ids, b, sink, n, CUS * 8); gather_probe<256, 8>(a, TB, ids, b, sink, n, CUS * 8); }
    if (rb == 512) { gather_probe<512, 2>(a, TB, ids, b, sink, n, CUS * 8); gather_probe<512, 4>(a, TB, ids, b, sink, n, CUS * 8); gather_probe<512, 8>(a, TB, ids, b, sink, n, CUS * 8); }
    if (rb == 1024) { gather_probe<1024, 2>(a, TB, ids, b, sink, n, CUS * 8); gather_probe<1024, 4>(a, TB, ids, b, sink, n, CUS * 8); }
  }
  // the bench table: 100 002 rows x 256 B = 25.6 MB (Infinity-Cache / L2 resident)
  {
    const uint64_t rows = 100002;
    for (int i = 0; i < n; ++i) h[i] = (int)(rnd() % rows);
    CK(hipMemcpy(ids, h.data(), (size_t)n * 4, hipMemcpyHostToDevice));
    char note[160];
    ms = time_ms([&] { gather_kernel<256, 4, true><<<CUS * 8, 256>>>(a, ids, b, sink, n); });
    snprintf(note, sizeof note, "%d random 256-B rows of a 25.6 MB table (cache resident) copied out: row read + write + id", n);
    report("gather_copy_256B_rows_25MB_table", (2.0 * n * 256 + 4.0 * n) / ms * 1e-6, "GB/s", note);
  }
  // ---- matrix pipe
  for (int wps : {1, 2, 4}) {
    const int iters = 20000;
    ms = time_ms([&] { mfma_kernel<<<CUS * wps, 256>>>(sink, iters); }, 5);
    const double flops = (double)CUS * wps * 4 /*waves*/ * iters * 4.0 * (2.0 * 16 * 16 * 32);
    char nm[64], note[128];
    snprintf(nm, sizeof nm, "mfma_16x16x32_bf16_%dwave_per_simd", wps);
    snprintf(note, sizeof note, "issue loop, 4 independent accumulators per wave, %d workgroups of 256 threads per CU", wps);
    report(nm, flops / ms * 1e-9, "TFLOP/s", note);
  }
  // ---- VALU port (wave-instructions per cycle per SIMD at the clock the probe runs at is not observable from here: report the rate in
  // G wave-instructions per second over the chip, and the ratios)
  {
    const int iters = 20000, wps = 4;
    const double n8 = (double)CUS * wps * 4 * iters * 8.0;
    char note[160];
    double g[4];
    ms = time_ms([&] { valu_kernel<0><<<CUS * wps, 256>>>(sink, iters); }, 5); g[0] = n8 / ms * 1e-6;
    ms = time_ms([&] { valu_kernel<1><<<CUS * wps, 256>>>(sink, iters); }, 5); g[1] = n8 / ms * 1e-6;
    ms = time_ms([&] { valu_kernel<2><<<CUS * wps, 256>>>(sink, iters); }, 5); g[2] = n8 / ms * 1e-6;
    ms = time_ms([&] { valu_kernel<3><<<CUS * wps, 256>>>(sink, iters); }, 5); g[3] = n8 / ms * 1e-6;
    snprintf(note, sizeof note, "8 independent chains per wave, 4 waves per SIMD; per SIMD and cycle at 2.4 GHz: %.3f", g[0] * 1e9 / (CUS * 4 * 2.4e9));
    report("valu_fma_f32", g[0], "G wave-instr/s", note);
    snprintf(note, sizeof note, "the same count of v_pk_fma_f32 (two results per lane each): %.2f x the time of v_fma_f32", g[0] / g[1]);
    report("valu_pk_fma_f32", g[1], "G wave-instr/s", note);
    snprintf(note, sizeof note, "transcendental: %.2f x the time of v_fma_f32", g[0] / g[2]);
    report("valu_exp_f32", g[2], "G wave-instr/s", note);
    snprintf(note, sizeof note, "transcendental: %.2f x the time of v_fma_f32", g[0] / g[3]);
    report("valu_rcp_f32", g[3], "G wave-instr/s", note);
  }
  {
    const int iters = 20000;
    for (int wps : {1, 2}) {
      double t[4];
      t[0] = time_ms([&] { mfma_valu_kernel<0><<<CUS * wps, 256>>>(sink, iters); }, 5);
      t[1] = time_ms([&] { mfma_valu_kernel<2><<<CUS * wps, 256>>>(sink, iters); }, 5);
      t[2] = time_ms([&] { mfma_valu_kernel<4><<<CUS * wps, 256>>>(sink, iters); }, 5);
      t[3] = time_ms([&] { mfma_valu_kernel<8><<<CUS * wps, 256>>>(sink, iters); }, 5);
      char nm[64], note[200];
      snprintf(nm, sizeof nm, "mfma_plus_own_valu_%dwave_per_simd", wps);
      snprintf(note, sizeof note, "time of (1 MFMA 16x16x32 bf16 + N independent v_fma_f32 of the same wave) relative to the MFMA alone: N = 2: %.2f, 4: %.2f, 8: %.2f",
               t[1] / t[0], t[2] / t[0], t[3] / t[0]);
      report(nm, (double)CUS * wps * 4 * iters * 4.0 * (2.0 * 16 * 16 * 32) / t[0] * 1e-9, "TFLOP/s", note);
    }
  }
  {
    // per MFMA gap: time relative to the bare MFMA loop of the same shape.  A pk filler carries two lane results: compare N pk with 2 N fma.
    const int iters = 20000;
    for (int wps : {1, 2}) {
      for (int big = 0; big < 2; ++big) {
        auto run = [&](auto kern) { return time_ms([&] { kern<<<CUS * wps, 256>>>(sink, iters); }, 5); };
        double t0, f2, f4, f8, p1, p2, p4, m2, m4, a2, a4, e1, e2, cv2, cv4;
        if (big) {
          t0 = run(coexec_kernel<0, 0, true>); f2 = run(coexec_kernel<2, 0, true>); f4 = run(coexec_kernel<4, 0, true>); f8 = run(coexec_kernel<8, 0, true>);
          p1 = run(coexec_kernel<1, 1, true>); p2 = run(coexec_kernel<2, 1, true>); p4 = run(coexec_kernel<4, 1, true>);
          m2 = run(coexec_kernel<2, 2, true>); m4 = run(coexec_kernel<4, 2, true>); a2 = run(coexec_kernel<2, 3, true>); a4 = run(coexec_kernel<4, 3, true>);
          e1 = run(coexec_kernel<1, 4, true>); e2 = run(coexec_kernel<2, 4, true>); cv2 = run(coexec_kernel<2, 5, true>); cv4 = run(coexec_kernel<4, 5, true>);
        } else {
          t0 = run(coexec_kernel<0, 0, false>); f2 = run(coexec_kernel<2, 0, false>); f4 = run(coexec_kernel<4, 0, false>); f8 = run(coexec_kernel<8, 0, false>);
          p1 = run(coexec_kernel<1, 1, false>); p2 = run(coexec_kernel<2, 1, false>); p4 = run(coexec_kernel<4, 1, false>);
          m2 = run(coexec_kernel<2, 2, false>); m4 = run(coexec_kernel<4, 2, false>); a2 = run(coexec_kernel<2, 3, false>); a4 = run(coexec_kernel<4, 3, false>);
          e1 = run(coexec_kernel<1, 4, false>); e2 = run(coexec_kernel<2, 4, false>); cv2 = run(coexec_kernel<2, 5, false>); cv4 = run(coexec_kernel<4, 5, false>);
        }
        char nm[64], note[400];
        snprintf(nm, sizeof nm, "coexec_%s_%dwave_per_simd", big ? "32x32x16" : "16x16x32", wps);
        snprintf(note, sizeof note,
                 "gap time / bare MFMA: v_fma_f32 x2 %.2f x4 %.2f x8 %.2f | v_pk_fma_f32 x1 %.2f x2 %.2f x4 %.2f | v_pk_mul_f32 x2 %.2f x4 %.2f | "
                 "v_pk_add_f32 x2 %.2f x4 %.2f | v_exp_f32 x1 %.2f x2 %.2f | v_cvt_pk_bf16_f32 x2 %.2f x4 %.2f",
                 f2 / t0, f4 / t0, f8 / t0, p1 / t0, p2 / t0, p4 / t0, m2 / t0, m4 / t0, a2 / t0, a4 / t0, e1 / t0, e2 / t0, cv2 / t0, cv4 / t0);
        const double fl = big ? 2.0 * 32 * 32 * 16 : 2.0 * 16 * 16 * 32;
        report(nm, (double)CUS * wps * 4 * iters * 4.0 * fl / t0 * 1e-9, "TFLOP/s", note);
      }
    }
  }
  printf("{");
  for (size_t i = 0; i < results.size(); ++i)
    printf("%s\"%s\": {\"value\": %.1f, \"unit\": \"%s\"}", i ? ", " : "", results[i].name.c_str(), results[i].value, results[i].unit.c_str());
  printf("}\n");
  return 0;
}
