// Issue rate of v_pk_fma_f32 against v_fma_f32 on gfx950 (developer probe):  hipcc --offload-arch=gfx950 -O3 tools/probe/pkfma.hip -o /tmp/pkfma && /tmp/pkfma
// Each wave runs N dependent-free fma instructions per loop trip on 16 accumulators (scalar) or 8 accumulator pairs (packed).
// MI355X, round 4: v_fma_f32 99.8 / 110.3 / 116.7 TFLOP/s at 1 / 2 / 4 waves per SIMD, v_pk_fma_f32 93.1 / 113.3 / 122.8 -- the
// scalar instruction already issues at the full fp32 rate (a wave64 v_fma_f32 every 2 cycles), the packed form buys nothing.
// (Tried in slsh64_kernel's projections with the planes pair-interleaved in LDS: 147 VGPRs instead of 116 and no faster.)
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float v2f __attribute__((ext_vector_type(2)));
template <bool PK>
__global__ __launch_bounds__(256) void k(float* out, int trips, float a, float b) {
  float acc[16];
  for (int i = 0; i < 16; ++i) acc[i] = threadIdx.x + i;
  for (int t = 0; t < trips; ++t) {
    if (PK) {
#pragma unroll
      for (int rep = 0; rep < 4; ++rep)
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          v2f v = {acc[2 * i], acc[2 * i + 1]};
          v = __builtin_elementwise_fma(v, v2f{a, a}, v2f{b, b});
          acc[2 * i] = v.x;
          acc[2 * i + 1] = v.y;
        }
    } else {
#pragma unroll
      for (int rep = 0; rep < 4; ++rep)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = __builtin_fmaf(acc[i], a, b);
    }
  }
  float s = 0;
  for (int i = 0; i < 16; ++i) s += acc[i];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
int main() {
  float* out;
  hipMalloc(&out, 4096 * 256 * 4);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  const int trips = 4096;
  for (int pk = 0; pk < 2; ++pk)
    for (int waves = 1; waves <= 4; waves *= 2) {  // workgroups per CU (4 waves each): 1, 2, 4 waves per SIMD
      const int grid = 256 * waves;
      float ms = 0;
      for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        if (pk) k<true><<<grid, 256>>>(out, trips, 1.0001f, 0.5f);
        else k<false><<<grid, 256>>>(out, trips, 1.0001f, 0.5f);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
      }
      const double fma = double(grid) * 256 * trips * 64;  // scalar fmas (a packed instruction counts two per lane)
      printf("%s  %d waves/SIMD: %.3f ms, %.1f TFLOP/s (2 flop per fma)\n", pk ? "v_pk_fma_f32" : "v_fma_f32   ", waves, ms, 2 * fma / ms / 1e9);
    }
  return 0;
}
