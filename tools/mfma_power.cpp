// Developer microbenchmark: sustained bf16 matrix rate of the whole chip on RANDOM operands held in registers,
// v_mfma_f32_32x32x16_bf16 against v_mfma_f32_16x16x32_bf16 (is the clock that gives way under the first any
// kinder to the second?).   hipcc -O2 --offload-arch=gfx950 tools/mfma_power.cpp -o tools/mfma_power
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
__device__ inline uint32_t mix(uint32_t x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }
__device__ inline bf16x8 rnd(uint32_t s, int zero) {
  u32x4 v;
  for (int i = 0; i < 4; ++i) {
    uint32_t r = mix(s * 4 + i);
    // two bf16 in [-2, 2): sign + exponent 0x3F/0x3E + random mantissa
    v[i] = zero ? 0u : ((r & 0x807F807Fu) | 0x3F003F00u | ((r >> 3) & 0x00800080u));
  }
  return __builtin_bit_cast(bf16x8, v);
}
template <int SHAPE>
__global__ __launch_bounds__(256) void k(float* out, int iters, int zero) {
  const uint32_t t = blockIdx.x * 256 + threadIdx.x;
  bf16x8 a[4], b[4];
  for (int i = 0; i < 4; ++i) { a[i] = rnd(t * 8 + i, zero); b[i] = rnd(t * 8 + 4 + i, zero); }
  if (SHAPE == 0) {
    f32x16 acc[8];
    for (int i = 0; i < 8; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[(i + j) & 3], b[j], acc[i], 0, 0, 0);
    }
    float s = 0;
    for (int i = 0; i < 8; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[t] = s;
  } else if (SHAPE == 2) {
    f32x4 acc[32];
    f16x8 ah[4], bh[4];  // the same random bits read as f16: sign, exponent 0x0F/0x0E-ish, random mantissa
    for (int i = 0; i < 4; ++i) {
      u32x4 ua = __builtin_bit_cast(u32x4, a[i]), ub = __builtin_bit_cast(u32x4, b[i]);
      for (int j = 0; j < 4; ++j) { ua[j] = (ua[j] & 0x83FF83FFu) | 0x3C003C00u; ub[j] = (ub[j] & 0x83FF83FFu) | 0x3C003C00u; if (zero) { ua[j] = 0; ub[j] = 0; } }
      ah[i] = __builtin_bit_cast(f16x8, ua); bh[i] = __builtin_bit_cast(f16x8, ub);
    }
    for (int i = 0; i < 32; ++i) for (int r = 0; r < 4; ++r) acc[i][r] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int i = 0; i < 32; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[(i + j) & 3], bh[((i >> 2) + j) & 3], acc[i], 0, 0, 0);
    }
    float s = 0;
    for (int i = 0; i < 32; ++i) for (int r = 0; r < 4; ++r) s += acc[i][r];
    out[t] = s;
  } else {
    f32x4 acc[32];
    for (int i = 0; i < 32; ++i) for (int r = 0; r < 4; ++r) acc[i][r] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int i = 0; i < 32; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[(i + j) & 3], b[((i >> 2) + j) & 3], acc[i], 0, 0, 0);
    }
    float s = 0;
    for (int i = 0; i < 32; ++i) for (int r = 0; r < 4; ++r) s += acc[i][r];
    out[t] = s;
  }
}
int main(int argc, char** argv) {
  const int iters = argc > 1 ? atoi(argv[1]) : 2000, wgs = argc > 2 ? atoi(argv[2]) : 512;
  float* out;
  hipMalloc(&out, (size_t)wgs * 256 * 4);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int zero = 0; zero < 2; ++zero)
    for (int shape = 0; shape < 3; ++shape) {
      for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        if (shape == 0) k<0><<<wgs, 256>>>(out, iters, zero); else if (shape == 1) k<1><<<wgs, 256>>>(out, iters, zero); else k<2><<<wgs, 256>>>(out, iters, zero);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        // per wave and iteration: shape 0: 32 MFMAs x 32768 flop; shape 1: 64 MFMAs x 16384 flop: the same
        const double flop = (double)wgs * 4 * iters * 32 * 32768.0;
        if (rep == 2) printf("%s operands, %s: %.3f ms  %.0f TFLOP/s\n", zero ? "zero  " : "random", shape == 0 ? "bf16 32x32x16" : shape == 1 ? "bf16 16x16x32" : "f16  16x16x32", ms, flop / ms / 1e9);
      }
    }
  return 0;
}
