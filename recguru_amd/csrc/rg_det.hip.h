// rg_acc(p, v): every floating-point accumulation into global memory that more than one wave may reach goes through here.
//
// The shipped library (librecguru_hip.so) compiles it to the hardware float atomic: sums whose ORDER follows the workgroup
// schedule, so two runs of one process differ in their last bits (parameter gradients, loss sums, the discriminator's scalars --
// SURVEY.md 5.2, DESIGN.md 2 "loss curves").  The second library (librecguru_hip_det.so: the same sources with -DRG_DETERMINISTIC,
// loaded by RG_DETERMINISTIC=1) compiles it to an INTEGER atomic on a 64-bit fixed-point shadow of the destination: integer
// addition is associative, so the sum is the same bits in any order -- across workgroup schedules, across runs, and with any number
// of contributions already in the slot.  No kernel argument changes: the host hands these kernels destinations that live in one of
// two ARENAS (recguru_amd/hip.py `_DetArena`) -- a float buffer the kernel may still READ (the mask count next to a loss sum) and a
// long long buffer twice its size at a fixed byte offset rule
//       shadow address = sbase + 2 * (p - fbase)
// -- and converts shadow -> float, added into the real destination, when the launch is done (one element-wise pass: ordered).
// Arena 0 (gradients): 2^-46 units, |sum| < 131072.  Arena 1 (loss sums, scalars): 2^-30 units, |sum| < 8.6e9.
// A destination outside both arenas is a host-side omission: the add still happens (float atomic) and the fault flag is raised,
// which rg_det_fault() reports -- tests/test_det_gpu.py holds it at 0 over whole training steps.
#pragma once
#include <hip/hip_runtime.h>

#ifdef RG_DETERMINISTIC
struct RgDetArena { unsigned long long fbase, sbase, bytes; int bits; int pad; };
static __device__ RgDetArena rg_det_arena[2];
static __device__ int rg_det_fault_flag;

// rg_error.hip; internal to the library (hidden: not part of the C ABI of include/recguru_hip.h)
extern "C" __attribute__((visibility("hidden"))) void rg_det_register_tu(int (*set)(const void* arenas), int (*fault)(int* out, int clear));

namespace {
struct RgDetTu {
  RgDetTu() { rg_det_register_tu(&set, &fault); }
  static int set(const void* arenas) {
    return (int)hipMemcpyToSymbol(HIP_SYMBOL(rg_det_arena), arenas, sizeof(RgDetArena) * 2, 0, hipMemcpyHostToDevice);
  }
  static int fault(int* out, int clear) {
    int v = 0;
    hipError_t e = hipMemcpyFromSymbol(&v, HIP_SYMBOL(rg_det_fault_flag), sizeof(int), 0, hipMemcpyDeviceToHost);
    if (e == hipSuccess && clear && v) {
      const int z = 0;
      e = hipMemcpyToSymbol(HIP_SYMBOL(rg_det_fault_flag), &z, sizeof(int), 0, hipMemcpyHostToDevice);
    }
    if (v > *out) *out = v;                              // the worst flag of any unit (0 / 1 / 2 as documented), not their sum
    return (int)e;
  }
};
static RgDetTu rg_det_tu;
}   // namespace

__device__ __forceinline__ void rg_acc(float* p, float v) {
  const unsigned long long a = (unsigned long long)p;
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const unsigned long long off = a - rg_det_arena[i].fbase;
    if (off < rg_det_arena[i].bytes) {
      const double s = (double)v * __longlong_as_double((long long)(1023 + rg_det_arena[i].bits) << 52);     // v * 2^bits, exact
      if (!(fabs(s) < 4.0e18)) rg_det_fault_flag = 2;                                                       // out of range (or NaN)
      atomicAdd(reinterpret_cast<unsigned long long*>(rg_det_arena[i].sbase + 2ull * off), (unsigned long long)__double2ll_rn(s));
      return;
    }
  }
  rg_det_fault_flag = 1;
  atomicAdd(p, v);
}
#define RG_DET_ONLY(x) x
#define RG_DET 1
#else
__device__ __forceinline__ void rg_acc(float* p, float v) { atomicAdd(p, v); }
#define RG_DET_ONLY(x)
#define RG_DET 0
#endif
