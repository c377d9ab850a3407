// Probe: does gfx950 execute scalar memory atomics (s_atomic_add ... glc), and what is their round trip?
// build: hipcc --offload-arch=gfx950 -O3 -o tools/probe/satomic tools/probe/satomic.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
__global__ void draw(unsigned* ctr, unsigned* out, unsigned long long* cyc, int reps) {
  const unsigned wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  unsigned last = 0;
  for (int r = 0; r < reps; ++r) {
    unsigned t = 1;
    asm volatile("s_atomic_add %0, %1, 0x0 glc\n\ts_waitcnt lgkmcnt(0)" : "+s"(t) : "s"(ctr) : "memory");
    last = t;
    if ((threadIdx.x & 63) == 0) out[t] = wave;
  }
  unsigned long long t1 = __builtin_amdgcn_s_memrealtime();
  if ((threadIdx.x & 63) == 0) cyc[wave] = t1 - t0 + (last & 0);
}
int main() {
  const int blocks = 256, threads = 512, reps = 16, waves = blocks * threads / 64;
  unsigned *ctr, *out; unsigned long long* cyc;
  hipMalloc(&ctr, 256); hipMalloc(&out, sizeof(unsigned) * waves * reps); hipMalloc(&cyc, 8 * waves);
  hipMemset(ctr, 0, 256); hipMemset(out, 0xff, sizeof(unsigned) * waves * reps);
  hipLaunchKernelGGL(draw, dim3(blocks), dim3(threads), 0, 0, ctr, out, cyc, reps);
  if (hipDeviceSynchronize() != hipSuccess) { printf("FAILED\n"); return 1; }
  unsigned c; hipMemcpy(&c, ctr, 4, hipMemcpyDeviceToHost);
  std::vector<unsigned> o(waves * reps); hipMemcpy(o.data(), out, 4 * o.size(), hipMemcpyDeviceToHost);
  std::vector<unsigned long long> cy(waves); hipMemcpy(cy.data(), cyc, 8 * waves, hipMemcpyDeviceToHost);
  size_t missing = 0; for (unsigned v : o) missing += v == 0xffffffffu;
  std::sort(cy.begin(), cy.end());
  printf("counter %u (expect %d), tickets never handed out %zu, per-draw 100MHz ticks median %.2f max %.2f\n", c, waves * reps, missing,
         cy[waves / 2] / double(reps), cy[waves - 1] / double(reps));
  return (c == unsigned(waves * reps) && missing == 0) ? 0 : 2;
}
