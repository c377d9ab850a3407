// Probe: how fast does a wave pair per SIMD issue the filter kernel's MFMA sequence (4 x 32x32x8_1k + 16 x 32x32x16 bf16 per
// block, four accumulators) when nothing else is in the loop?  Variants (argv[1]): 0 = accumulators carried across blocks,
// 1 = fresh accumulators per block (C = 0) + one use of each at the end of the block, 2 = 1 + sixteen v_mov of the B operand.
// build: hipcc --offload-arch=gfx950 -O3 -o tools/probe/mfma_rate tools/probe/mfma_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ __launch_bounds__(256, 2) void k(const u32x4* __restrict__ in, float* __restrict__ out, int nblocks, unsigned long long* cyc) {
  const int tid = threadIdx.x;
  u32x4 a[4][4], b[4];
#pragma unroll
  for (int m = 0; m < 4; ++m)
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) a[m][ks] = in[(m * 4 + ks) * 256 + tid];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) b[ks] = in[(16 + ks) * 256 + tid];
  s16x4 at = {0x3F80, 0, 0, 0}, ones = {0x3F80, 0x3F80, 0, 0};
  f32x16 acc[4];
#pragma unroll
  for (int m = 0; m < 4; ++m)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[m][r] = 0.f;
  float sink = 0.f;
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int j = 0; j < nblocks; ++j) {
    if (MODE >= 2) {
      u32x4 t[4];
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) { t[ks] = b[ks]; asm volatile("" : "+v"(t[ks])); }
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) b[ks] = t[ks];
    }
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      f32x16 c = acc[m];
      if (MODE >= 1) {
#pragma unroll
        for (int r = 0; r < 16; ++r) c[r] = 0.f;
      }
      c = __builtin_amdgcn_mfma_f32_32x32x8bf16_1k(at, ones, c, 0, 0, 0);
#pragma unroll
      for (int ks = 0; ks < 4; ++ks)
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a[m][ks]), __builtin_bit_cast(bf16x8, b[ks]), c, 0, 0, 0);
      acc[m] = c;
    }
    if (MODE >= 1) {
#pragma unroll
      for (int m = 0; m < 4; ++m) sink += acc[m][m];
    }
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
#pragma unroll
  for (int m = 0; m < 4; ++m)
#pragma unroll
    for (int r = 0; r < 16; ++r) sink += acc[m][r];
  out[blockIdx.x * 256 + tid] = sink;
  if (tid == 0) cyc[blockIdx.x] = t1 - t0;
}

int main(int argc, char** argv) {
  const int mode = argc > 1 ? atoi(argv[1]) : 0, wgs = argc > 2 ? atoi(argv[2]) : 512, nblocks = 2000;
  u32x4* in; float* out; unsigned long long* cyc;
  hipMalloc(&in, 20 * 256 * 16); hipMemset(in, 0x3c, 20 * 256 * 16);
  hipMalloc(&out, wgs * 256 * 4); hipMalloc(&cyc, wgs * 8);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(e0);
    if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(wgs), dim3(256), 0, 0, in, out, nblocks, cyc);
    else if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(wgs), dim3(256), 0, 0, in, out, nblocks, cyc);
    else hipLaunchKernelGGL(k<2>, dim3(wgs), dim3(256), 0, 0, in, out, nblocks, cyc);
    hipEventRecord(e1);
    if (hipDeviceSynchronize() != hipSuccess) { printf("FAILED\n"); return 1; }
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long c0; hipMemcpy(&c0, cyc, 8, hipMemcpyDeviceToHost);
    const double mfma_per_simd = double(wgs) * 4 * nblocks * 20 / 1024.0;  // waves x blocks x 20 / SIMDs
    printf("mode %d, %d workgroups: %.1f us, %.1f ns per MFMA and SIMD (32 cycles at 2.4 GHz = 13.3 ns); wave 0: %.1f cycle-counter ticks per MFMA of its own\n",
           mode, wgs, ms * 1e3, ms * 1e6 / mfma_per_simd, double(c0) / (nblocks * 20.0));
  }
  return 0;
}
