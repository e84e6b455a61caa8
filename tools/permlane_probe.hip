// gfx950 row exchanges used by csrc/rg_common.hip.h rg_allgather_rows(): after one v_permlane32_swap and two v_permlane16_swap every
// lane (li, row r) holds the values that the lanes (li, row 0..3) started with.   hipcc --offload-arch=gfx950 tools/permlane_probe.hip -o /tmp/pp && /tmp/pp
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(2))) unsigned int u2;
__global__ void k(unsigned int* out) {
  const unsigned int v = threadIdx.x * 7u + 3u;
  const u2 pq = __builtin_amdgcn_permlane32_swap(v, v, false, false);
  const u2 a = __builtin_amdgcn_permlane16_swap(pq.x, pq.x, false, false);
  const u2 b = __builtin_amdgcn_permlane16_swap(pq.y, pq.y, false, false);
  out[threadIdx.x * 4 + 0] = a.x; out[threadIdx.x * 4 + 1] = a.y; out[threadIdx.x * 4 + 2] = b.x; out[threadIdx.x * 4 + 3] = b.y;
}
int main() {
  unsigned int *d, h[256];
  hipMalloc(&d, sizeof h);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
  hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
  int bad = 0;
  for (int l = 0; l < 64; ++l)
    for (int r = 0; r < 4; ++r) {
      const unsigned int want = (unsigned int)(r * 16 + (l & 15)) * 7u + 3u;
      if (h[l * 4 + r] != want) { if (bad < 8) printf("lane %d slot %d: got %u want %u (lane %u)\n", l, r, h[l * 4 + r], want, (h[l * 4 + r] - 3u) / 7u); ++bad; }
    }
  printf("permlane all-gather over the 4 rows: %s (%d mismatches)\n", bad ? "DIFFERENT ORDER" : "slot r = row r", bad);
  return 0;
}
