// VALU issue-rate probe for gfx950: how many cycles does a wave64 spend per instruction of each kind?
// Each kernel runs ITER x 16 instructions over 8 independent accumulators (no memory traffic), with
// 4 waves per SIMD resident like the lsh64 kernel.  Prints ns per instruction per wave and the implied
// issue cycles at the measured clock.  Build: hipcc -O3 --offload-arch=gfx950 tools/valu_rate.hip -o tools/valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef float v2f __attribute__((ext_vector_type(2)));
constexpr int ITER = 4096;

#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)

template <int KIND>
__global__ __launch_bounds__(256) void probe(float* out, float seed) {
  v2f a[8], m = v2f{seed, seed * 0.5f}, c = v2f{1.0f, 2.0f};
  for (int i = 0; i < 8; ++i) a[i] = v2f{seed + i, seed - i};
  for (int it = 0; it < ITER; ++it) {
    if (KIND == 0) {  // v_fma_f32, 16 per iteration
#define X(i) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a[i].x) : "v"(m.x), "v"(c.x)); \
             asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a[i].y) : "v"(m.y), "v"(c.y));
      REP8(X)
#undef X
    } else if (KIND == 1) {  // v_pk_fma_f32, 16 per iteration
#define X(i) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(a[i]) : "v"(m), "v"(c)); \
             asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(a[i]) : "v"(c), "v"(m));
      REP8(X)
#undef X
    } else if (KIND == 2) {  // v_pk_fma_f32 with a broadcast source (op_sel_hi:[0,1,1])
#define X(i) asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[0,1,1]" : "+v"(a[i]) : "v"(m), "v"(c)); \
             asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,0,0]" : "+v"(a[i]) : "v"(c), "v"(m));
      REP8(X)
#undef X
    } else if (KIND == 3) {  // v_add_f32_dpp row_ror on independent registers (no RAW hazard inside a group of 8)
#define X(i) asm volatile("v_add_f32_dpp %0, %1, %1 row_ror:8 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "=v"(a[i].x) : "v"(a[i].y));
      REP8(X)
#undef X
#define X(i) asm volatile("v_add_f32_dpp %0, %1, %1 row_ror:4 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "=v"(a[i].y) : "v"(a[i].x));
      REP8(X)
#undef X
    } else if (KIND == 4) {  // v_pk_mul_f32 + v_pk_add_f32
#define X(i) asm volatile("v_pk_mul_f32 %0, %1, %0" : "+v"(a[i]) : "v"(c)); \
             asm volatile("v_pk_add_f32 %0, %1, %0" : "+v"(a[i]) : "v"(m));
      REP8(X)
#undef X
    } else if (KIND == 5) {  // v_cndmask + v_cmp pairs
#define X(i) asm volatile("v_cmp_gt_f32 vcc, 0, %1\n v_cndmask_b32 %0, 1.0, %2, vcc" : "=v"(a[i].x) : "v"(a[i].y), "v"(m.x) : "vcc");
      REP8(X)
#undef X
    } else if (KIND == 6) {  // dependent chain: v_fma_f32 on ONE accumulator (latency bound for a single wave)
#define X(i) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a[0].x) : "v"(m.x), "v"(c.x)); \
             asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a[0].x) : "v"(m.y), "v"(c.y));
      REP8(X)
#undef X
    } else if (KIND == 7) {  // dependent chain of v_pk_fma_f32
#define X(i) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(a[0]) : "v"(m), "v"(c)); \
             asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(a[0]) : "v"(c), "v"(m));
      REP8(X)
#undef X
    } else if (KIND == 8) {  // v_rcp_f32
#define X(i) asm volatile("v_rcp_f32 %0, %0" : "+v"(a[i].x)); asm volatile("v_rcp_f32 %0, %0" : "+v"(a[i].y));
      REP8(X)
#undef X
    }
  }
  float s = 0.f;
  for (int i = 0; i < 8; ++i) s += a[i].x + a[i].y;
  if (s == 12345.678f) out[0] = s;
}

template <int KIND>
static void run(const char* name, int waves_per_simd, float* d, double ghz) {
  const int blocks = 256 * waves_per_simd;  // 256 CUs x (4 waves = 1 per SIMD) x waves_per_simd
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(probe<KIND>, dim3(blocks), dim3(256), 0, 0, d, 1.0f);
  hipEventRecord(e0);
  const int reps = 5;
  for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(probe<KIND>, dim3(blocks), dim3(256), 0, 0, d, 1.0f);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  const double ns_per_instr_simd = ms * 1e6 / reps / (double(ITER) * 16 * waves_per_simd);
  printf("%-34s waves/SIMD %d : %.3f ns per wave-instruction on a SIMD = %.2f cycles @ %.2f GHz\n", name, waves_per_simd,
         ns_per_instr_simd, ns_per_instr_simd * ghz, ghz);
}

int main(int argc, char** argv) {
  const double ghz = argc > 1 ? atof(argv[1]) : 2.4;
  float* d;
  hipMalloc(&d, 4096);
  for (int w : {1, 4}) {
    run<0>("v_fma_f32", w, d, ghz);
    run<1>("v_pk_fma_f32", w, d, ghz);
    run<2>("v_pk_fma_f32 op_sel broadcast", w, d, ghz);
    run<3>("v_add_f32_dpp row_ror", w, d, ghz);
    run<4>("v_pk_mul_f32 / v_pk_add_f32", w, d, ghz);
    run<5>("v_cmp_gt_f32 + v_cndmask_b32 (x2)", w, d, ghz);
    run<6>("v_fma_f32 dependent chain", w, d, ghz);
    run<7>("v_pk_fma_f32 dependent chain", w, d, ghz);
    run<8>("v_rcp_f32", w, d, ghz);
  }
  return 0;
}
