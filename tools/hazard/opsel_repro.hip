// Stand-alone reproducer of DESIGN 2a finding 1: on gfx950 (MI355X)
//     v_pk_add_f32 vD, vA, vB op_sel:[0,1]          (a packed-f32 add whose LOW result takes the HIGH half of src1)
// returns  src0.lo + 0  as its low result in lanes 48-63 when other waves on the same SIMD are issuing MFMAs.
//   hipcc --offload-arch=gfx950 -O2 tools/hazard/opsel_repro.hip -o tools/hazard/opsel_repro && ./tools/hazard/opsel_repro [iterations]
// Every lane of the probing waves repeats  d = a - b.hi  with the packed form (inline asm) and with two v_sub_f32 and counts the
// results that differ, per lane and per half.  "neighbours": what else runs on the SIMD -- 0 nothing, 1 an LDS store that reads the
// destination pair in front of the instruction and an LDS load behind it (what stood around it in the fused block), 2 = 1 +
// transcendental loops in the odd waves of the workgroup, 3 = 1 + back-to-back MFMAs in the odd waves, 4 = 1 + MFMAs in waves 1-3
// (wave 0 probes), 5 = 3 without the LDS traffic.  Measured (profiles/r04/determinism/opsel_repro.txt): nothing without MFMA
// neighbours at any occupancy; with them wrong LOW results in lanes 48-63 ONLY, never a wrong high result, never lanes 0-47, growing
// with the MFMA density (neighbours 4: 80 - 176 at two workgroups per CU, 3e4 at four, 9e4 - 3e7 at eight, of 6e8 ... 2.5e9 operations);
// every one of them equals src0.lo + 0 exactly.
// The same with v_pk_mul_f32 op_sel:[0,1] and v_pk_fma_f32 op_sel:[0,1,0]: it is the SECOND source operand.  Clean under the same
// neighbours: high-half selects on src0 (v_pk_add_f32 op_sel:[1,0], v_pk_mov_b32 op_sel:[1,0], v_pk_fma_f32 op_sel:[1,0,0]) and on src2
// (v_pk_fma_f32 op_sel:[0,0,1]), op_sel_hi:[1,0] (the LOW half to both results), no selection.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f2 __attribute__((ext_vector_type(2)));

template <int MODE, int FORM>
__global__ __launch_bounds__(256) void probe(const f2* __restrict__ a, const f2* __restrict__ b, unsigned int* __restrict__ bad, int iters, float* sink) {
  __shared__ f2 lds[256 * 2];
  const int t = threadIdx.x, g = blockIdx.x * 256 + t;
  f2 av = a[g], bv = b[g];
  unsigned int c0 = 0, c1 = 0, cz = 0;
  float acc = 0.f;
  if (MODE == 2 && ((t >> 6) & 1)) {            // odd waves: transcendental traffic on the SIMD
    float x = av.x;
    for (int i = 0; i < iters * 8; ++i) x = __builtin_amdgcn_exp2f(x * 0.5f) + __builtin_amdgcn_rcpf(x + 2.f);
    sink[g] = x;
    return;
  }
  if (((MODE == 3 || MODE == 5) && ((t >> 6) & 1)) || (MODE == 4 && (t >> 6) != 0)) {     // odd waves (4: all but wave 0): matrix-pipe traffic on the SIMD
    typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
    typedef __attribute__((ext_vector_type(4))) float f32x4;
    bf16x8_t x, y;
    for (int j = 0; j < 8; ++j) { x[j] = (__bf16)(av.x + j); y[j] = (__bf16)(bv.y - j); }
    f32x4 c0v = {0.f, 0.f, 0.f, 0.f}, c1v = {1.f, 1.f, 1.f, 1.f};
    for (int i = 0; i < iters * 2; ++i) {
      c0v = __builtin_amdgcn_mfma_f32_16x16x32_bf16(x, y, c0v, 0, 0, 0);
      c1v = __builtin_amdgcn_mfma_f32_16x16x32_bf16(y, x, c1v, 0, 0, 0);
    }
    sink[g] = c0v[0] + c1v[1];
    return;
  }
  f2 d = {0.f, 0.f};
  for (int i = 0; i < iters; ++i) {
    if (MODE >= 1 && MODE != 5) lds[t] = d;                   // LDS store reading the pair the next instruction writes
    // FORM 0: the form under test (LOW result from the HIGH half of src1); controls under the same conditions -- FORM 1: the LOW half to
    // both results (op_sel_hi:[1,0]); FORM 2: the high half selected on src0 of a v_pk_fma_f32 (op_sel:[1,0,0]); FORM 3: no selection
    const f2 m1 = {-1.f, -1.f};
    if (FORM == 4) {                             // FORM 4: FORM 0 in the registers it had in the fused block (dst v[38:39], src0 v[44:45], src1 v[78:79])
      asm volatile("v_mov_b32 v44, %2\n\tv_mov_b32 v45, %3\n\tv_mov_b32 v78, %4\n\tv_mov_b32 v79, %5\n\ts_nop 1\n\t"
                   "v_pk_add_f32 v[38:39], v[44:45], v[78:79] op_sel:[0,1] neg_lo:[0,1] neg_hi:[0,1]\n\ts_nop 1\n\t"
                   "v_mov_b32 %0, v38\n\tv_mov_b32 %1, v39"
                   : "=v"(d.x), "=v"(d.y) : "v"(av.x), "v"(av.y), "v"(bv.x), "v"(bv.y) : "v38", "v39", "v44", "v45", "v78", "v79");
    } else if (FORM == 0) asm volatile("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] neg_lo:[0,1] neg_hi:[0,1]" : "=v"(d) : "v"(av), "v"(bv));
    else if (FORM == 1) asm volatile("v_pk_add_f32 %0, %1, %2 op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]" : "=v"(d) : "v"(av), "v"(bv));
    else if (FORM == 2) asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,0,0]" : "=v"(d) : "v"(bv), "v"(m1), "v"(av));
    else if (FORM == 5) asm volatile("v_pk_mov_b32 %0, %1, %2 op_sel:[1,0]" : "=v"(d) : "v"(av), "v"(bv));          // D.lo = src0.hi, D.hi = src1.lo
    else if (FORM == 6) asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[0,1]" : "=v"(d) : "v"(av), "v"(bv));          // D.lo = a.lo * b.hi, D.hi = a.hi * b.hi
    else if (FORM == 7) asm volatile("v_pk_add_f32 %0, %1, %2 op_sel:[1,0]" : "=v"(d) : "v"(av), "v"(bv));          // D.lo = a.hi + b.lo, D.hi = a.hi + b.hi
    else if (FORM == 8) asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,1,0]" : "=v"(d) : "v"(m1), "v"(bv), "v"(av));  // D.lo = -1 * b.hi + a.lo, D.hi = -1 * b.hi + a.hi
    else if (FORM == 9) asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,0,1]" : "=v"(d) : "v"(m1), "v"(av), "v"(bv));  // D.lo = -a.lo + b.hi, D.hi = -a.hi + b.hi
    else asm volatile("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(d) : "v"(av), "v"(bv));
    float e0, e1;                                // (plain C here is vectorised by hipcc into the very instruction under test)
    const float s0 = FORM == 1 ? bv.x : bv.y, s1 = FORM == 1 ? bv.x : (FORM == 3 ? bv.y : bv.y);
    if (FORM == 5) { asm volatile("v_mov_b32 %0, %1" : "=v"(e0) : "v"(av.y)); asm volatile("v_mov_b32 %0, %1" : "=v"(e1) : "v"(bv.x)); }
    else if (FORM == 6) { asm volatile("v_mul_f32 %0, %1, %2" : "=v"(e0) : "v"(av.x), "v"(bv.y)); asm volatile("v_mul_f32 %0, %1, %2" : "=v"(e1) : "v"(av.y), "v"(bv.y)); }
    else if (FORM == 7) { asm volatile("v_add_f32 %0, %1, %2" : "=v"(e0) : "v"(av.y), "v"(bv.x)); asm volatile("v_add_f32 %0, %1, %2" : "=v"(e1) : "v"(av.y), "v"(bv.y)); }
    else if (FORM == 9) { asm volatile("v_sub_f32 %0, %1, %2" : "=v"(e0) : "v"(bv.y), "v"(av.x)); asm volatile("v_sub_f32 %0, %1, %2" : "=v"(e1) : "v"(bv.y), "v"(av.y)); }
    else {
      asm volatile("v_sub_f32 %0, %1, %2" : "=v"(e0) : "v"(av.x), "v"(FORM == 3 ? bv.x : s0));
      asm volatile("v_sub_f32 %0, %1, %2" : "=v"(e1) : "v"(av.y), "v"(s1));
    }
    c0 += d.x != e0;
    cz += d.x != e0 && d.x == av.x;              // ... and equal to src0 + 0: the selected source read as zero
    c1 += d.y != e1;
    if (MODE >= 1 && MODE != 5) acc += lds[(t + 64) & 255].x;
    av.x += 1.f;
    av.y -= 0.5f;
    bv.y = bv.y * 1.0009765625f + 0.25f;
    bv.x += 0.125f;
  }
  sink[g] = acc + d.x;
  if (c0) atomicAdd(bad + (t & 63), c0);
  if (c1) atomicAdd(bad + 64 + (t & 63), c1);
  if (cz) atomicAdd(bad + 128, cz);
}

template <int MODE, int FORM = 0>
static void run(int wgs_per_cu, int iters) {
  const int cus = 256, grid = cus * wgs_per_cu, n = grid * 256;
  std::vector<f2> ha(n), hb(n);
  for (int i = 0; i < n; ++i) { ha[i] = (f2){(float)(i % 97) * 0.25f, (float)(i % 31) - 7.f}; hb[i] = (f2){(float)(i % 13) + 100.f, (float)(i % 29) * 0.125f + 1.f}; }
  f2 *a, *b; unsigned int* bad; float* sink;
  hipMalloc(&a, n * sizeof(f2)); hipMalloc(&b, n * sizeof(f2)); hipMalloc(&bad, 129 * 4); hipMalloc(&sink, n * 4);
  hipMemcpy(a, ha.data(), n * sizeof(f2), hipMemcpyHostToDevice); hipMemcpy(b, hb.data(), n * sizeof(f2), hipMemcpyHostToDevice);
  hipMemset(bad, 0, 129 * 4);
  for (int rep = 0; rep < 20; ++rep) hipLaunchKernelGGL((probe<MODE, FORM>), dim3(grid), dim3(256), 0, 0, a, b, bad, iters, sink);
  hipDeviceSynchronize();
  unsigned int h[129];
  hipMemcpy(h, bad, sizeof(h), hipMemcpyDeviceToHost);
  unsigned long long lo = 0, hi = 0, q[4] = {0, 0, 0, 0};
  for (int l = 0; l < 64; ++l) { lo += h[l]; hi += h[64 + l]; q[l >> 4] += h[l] + h[64 + l]; }
  static const char* forms[10] = {"v_pk_add_f32 op_sel:[0,1]", "v_pk_add_f32 op_sel_hi:[1,0]", "v_pk_fma_f32 op_sel:[1,0,0]", "v_pk_add_f32 (no select)", "op_sel:[0,1] in v38/v44/v78", "v_pk_mov_b32 op_sel:[1,0]", "v_pk_mul_f32 op_sel:[0,1]", "v_pk_add_f32 op_sel:[1,0]", "v_pk_fma_f32 op_sel:[0,1,0]", "v_pk_fma_f32 op_sel:[0,0,1]"};
  printf("%-30s neighbours %d, %d workgroup(s) per CU, 20 launches x %d per lane: wrong LOW results %llu (%u of them = src0 + 0), wrong HIGH results %llu; by lane quarter %llu %llu %llu %llu\n",
         forms[FORM], MODE, wgs_per_cu, iters, lo, h[128], hi, q[0], q[1], q[2], q[3]);
  (void)hipFree(a); (void)hipFree(b); (void)hipFree(bad); (void)hipFree(sink);
}

int main(int argc, char** argv) {
  const int iters = argc > 1 ? atoi(argv[1]) : 20000;
  printf("neighbours: 0 none, 1 LDS store / load around the instruction, 2 = 1 + transcendental loops in the odd waves, 3 = 1 + MFMAs in the odd waves, 4 = 1 + MFMAs in waves 1-3, 5 = MFMAs in the odd waves, no LDS\n");
  for (int w : {1, 2, 3, 4, 5, 6, 8}) { run<0>(w, iters); run<1>(w, iters); run<2>(w, iters); run<3>(w, iters); run<4>(w, iters); run<5>(w, iters); run<3, 4>(w, iters); run<5, 4>(w, iters); }
  printf("other forms with a HIGH half feeding the LOW result\n");
  for (int w : {1, 4, 8}) { run<0, 5>(w, iters); run<4, 5>(w, iters); run<0, 6>(w, iters); run<4, 6>(w, iters); run<0, 7>(w, iters); run<4, 7>(w, iters); run<4, 2>(w, iters);
                          run<0, 8>(w, iters); run<4, 8>(w, iters); run<0, 9>(w, iters); run<4, 9>(w, iters); }
  printf("controls\n");
  for (int w : {2, 3, 4, 5, 6}) { run<3, 1>(w, iters); run<3, 2>(w, iters); run<3, 3>(w, iters); run<5, 1>(w, iters); run<5, 3>(w, iters); }
  return 0;
}
