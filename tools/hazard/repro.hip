// Minimal experiment for DESIGN 2a: does an LDS load return land in a VGPR BEFORE an earlier VALU / transcendental
// instruction of the same wave has written / read that register, when two waves share a SIMD?
//   hipcc --offload-arch=gfx950 -O2 tools/hazard/repro.hip -o /tmp/hazard_repro && /tmp/hazard_repro [variant] [wgs] [lds_kb]
// Every lane keeps (1, 2, 3, 4) at its own 16 bytes of LDS.  The asm block replays the instruction mix of the failing
// LayerNorm code (rsqrtf()'s expansion followed by ds_read_b128 into the registers the expansion used as temporaries) and
// then checks that the four registers hold (1, 2, 3, 4).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

template <int VARIANT>
__global__ __launch_bounds__(256) void k(unsigned long long* errs, int iters, float seed) {
  extern __shared__ float lds[];
  const int tid = threadIdx.x;
  lds[tid * 4 + 0] = 1.f; lds[tid * 4 + 1] = 2.f; lds[tid * 4 + 2] = 3.f; lds[tid * 4 + 3] = 4.f;
  __syncthreads();
  const unsigned addr = tid * 16;
  float x = seed + 0.001f * (float)(tid & 63);
  unsigned long long bad = 0, lanes_hi = 0;
  if (VARIANT == 2 && (blockIdx.x & 1)) {
    // aggressor workgroups: keep the matrix pipe, the transcendental unit and the LDS busy next to the testers
    typedef __attribute__((ext_vector_type(4))) float f4;
    typedef __attribute__((ext_vector_type(8))) __bf16 b8;
    f4 acc = {0.f, 0.f, 0.f, 0.f};
    b8 av, bv;
    for (int j = 0; j < 8; ++j) { av[j] = (__bf16)(x + j); bv[j] = (__bf16)(x - j); }
    float t = x;
    for (int it = 0; it < iters * 2; ++it) {
      acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av, bv, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bv, av, acc, 0, 0, 0);
      t = __builtin_amdgcn_rsqf(t + 1.f) + lds[(tid * 4 + (it & 3)) & 1023];
      t += __builtin_amdgcn_exp2f(t * 1e-3f);
    }
    if (t + acc[0] == 12345.f) errs[3] = 1;
    return;
  }
  for (int it = 0; it < iters; ++it) {
    float r0, r1, r2, r3, keep;
    if (VARIANT == 0 || VARIANT == 2) {
      asm volatile(
          "v_mov_b32 v70, %6\n\t"
          "v_mov_b32 v71, %6\n\t"
          "v_mul_f32 v72, 0x4b800000, v71\n\t"
          "v_cmp_gt_f32 vcc, 0x0da24260, v71\n\t"
          "v_mul_f32 v73, 0x4b800000, v70\n\t"
          "s_nop 0\n\t"
          "v_cndmask_b32 v71, v71, v72, vcc\n\t"
          "v_rsq_f32 v71, v71\n\t"
          "v_mul_f32 v72, 0x45800000, v71\n\t"
          "v_cndmask_b32 v70, v70, v73, vcc\n\t"
          "v_mul_f32 v72, 0x45800000, v71\n\t"
          "v_rsq_f32 v79, v70\n\t"
          "v_cndmask_b32 v82, v71, v72, vcc\n\t"
          "ds_read_b128 v[70:73], %5\n\t"
          "s_waitcnt lgkmcnt(0)\n\t"
          "v_mov_b32 %0, v70\n\t"
          "v_mov_b32 %1, v71\n\t"
          "v_mov_b32 %2, v72\n\t"
          "v_mov_b32 %3, v73\n\t"
          "v_add_f32 %4, v79, v82\n\t"
          : "=v"(r0), "=v"(r1), "=v"(r2), "=v"(r3), "=v"(keep)
          : "v"(addr), "v"(x)
          : "v70", "v71", "v72", "v73", "v79", "v82", "vcc", "memory");
    } else {
      // the same with several transcendental instructions queued in front (the unit is shared by the SIMD's waves)
      asm volatile(
          "v_mov_b32 v70, %6\n\t"
          "v_mov_b32 v71, %6\n\t"
          "v_rsq_f32 v74, v70\n\t"
          "v_rsq_f32 v75, v71\n\t"
          "v_exp_f32 v76, v70\n\t"
          "v_rcp_f32 v77, v71\n\t"
          "v_mul_f32 v72, 0x4b800000, v71\n\t"
          "v_cmp_gt_f32 vcc, 0x0da24260, v71\n\t"
          "v_mul_f32 v73, 0x4b800000, v70\n\t"
          "s_nop 0\n\t"
          "v_cndmask_b32 v71, v71, v72, vcc\n\t"
          "v_rsq_f32 v71, v71\n\t"
          "v_mul_f32 v72, 0x45800000, v71\n\t"
          "v_cndmask_b32 v70, v70, v73, vcc\n\t"
          "v_mul_f32 v72, 0x45800000, v71\n\t"
          "v_rsq_f32 v79, v70\n\t"
          "v_cndmask_b32 v82, v71, v72, vcc\n\t"
          "ds_read_b128 v[70:73], %5\n\t"
          "s_waitcnt lgkmcnt(0)\n\t"
          "v_mov_b32 %0, v70\n\t"
          "v_mov_b32 %1, v71\n\t"
          "v_mov_b32 %2, v72\n\t"
          "v_mov_b32 %3, v73\n\t"
          "v_add_f32 %4, v79, v82\n\t"
          "v_add_f32 %4, %4, v74\n\t"
          "v_add_f32 %4, %4, v75\n\t"
          "v_add_f32 %4, %4, v76\n\t"
          "v_add_f32 %4, %4, v77\n\t"
          : "=v"(r0), "=v"(r1), "=v"(r2), "=v"(r3), "=v"(keep)
          : "v"(addr), "v"(x)
          : "v70", "v71", "v72", "v73", "v74", "v75", "v76", "v77", "v79", "v82", "vcc", "memory");
    }
    const bool b = (r0 != 1.f) || (r1 != 2.f) || (r2 != 3.f) || (r3 != 4.f);
    bad += b;
    lanes_hi += b && (tid & 63) >= 48;
    x = x * 1.0000001f + keep * 1e-30f;
  }
  if (bad) { atomicAdd(errs, bad); atomicAdd(errs + 1, lanes_hi); }
  if (x == 12345.f) errs[2] = 1;
}

int main(int argc, char** argv) {
  const int variant = argc > 1 ? atoi(argv[1]) : 0, wgs = argc > 2 ? atoi(argv[2]) : 512, lds_kb = argc > 3 ? atoi(argv[3]) : 70;
  unsigned long long* errs;
  hipMalloc(&errs, 32);
  hipMemset(errs, 0, 32);
  const int iters = 200000;
  const size_t smem = (size_t)lds_kb * 1024;
  if (variant == 2) {
    hipFuncSetAttribute(reinterpret_cast<const void*>(k<2>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    hipLaunchKernelGGL(k<2>, dim3(wgs), dim3(256), smem, 0, errs, iters, 1.5f);
  } else if (variant == 0) {
    hipFuncSetAttribute(reinterpret_cast<const void*>(k<0>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    hipLaunchKernelGGL(k<0>, dim3(wgs), dim3(256), smem, 0, errs, iters, 1.5f);
  } else {
    hipFuncSetAttribute(reinterpret_cast<const void*>(k<1>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    hipLaunchKernelGGL(k<1>, dim3(wgs), dim3(256), smem, 0, errs, iters, 1.5f);
  }
  hipDeviceSynchronize();
  unsigned long long h[4];
  hipMemcpy(h, errs, 32, hipMemcpyDeviceToHost);
  printf("variant %d, %d workgroups x 256 threads, %d KB LDS each, %d iterations per lane: %llu wrong register quadruples (%llu in lanes 48-63)\n",
         variant, wgs, lds_kb, iters, h[0], h[1]);
  return 0;
}
