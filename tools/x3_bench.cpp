// Developer harness (not part of the product): times mi_oov_linear_x3 (and mi_oov_linear_act) on one layer shape through
// the C ABI of a dlopen'ed library, so that variants built with different knobs can be compared in one GPU session.
//   hipcc -O2 tools/x3_bench.cpp -o tools/x3_bench -ldl
//   tools/x3_bench <libmi_oov.so> [B=65536] [K=1024] [N=512] [act=1] [iters=20]
// Prints the time per launch and a checksum of the output bits (schedule variants must agree bit for bit).
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include <vector>

#define CK(x)                                                               \
  do {                                                                      \
    hipError_t e_ = (x);                                                    \
    if (e_ != hipSuccess) {                                                 \
      printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); \
      exit(1);                                                              \
    }                                                                       \
  } while (0)

typedef int64_t (*bytes_fn)(int64_t, int64_t);
typedef int (*prep_fn)(const float*, int64_t, int64_t, void*, void*);
typedef int (*x3_fn)(const float*, int64_t, int64_t, const void*, const float*, int64_t, int, float*, void*);
typedef int (*lin_fn)(const float*, int64_t, int64_t, const float*, const float*, int64_t, int, float*, void*);

__device__ inline uint64_t mix(uint64_t x) {
  x ^= x >> 33; x *= 0xff51afd7ed558ccdULL; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL; x ^= x >> 33;
  return x;
}
__global__ void fill_f32(float* p, size_t n, uint64_t seed, float scale) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const uint64_t h = mix(i + seed * 0x9E3779B97F4A7C15ULL);
    p[i] = scale * (float)((int64_t)(h >> 40) - (1 << 23)) * (1.0f / (1 << 23));
  }
}
__global__ void checksum(const uint32_t* p, size_t n, unsigned long long* out) {
  unsigned long long acc = 0;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    acc += (unsigned long long)p[i] * (i % 1000003 + 1);
  atomicAdd(out, acc);
}

int main(int argc, char** argv) {
  if (argc < 2) return 1;
  const int64_t B = argc > 2 ? atoll(argv[2]) : 65536, K = argc > 3 ? atoll(argv[3]) : 1024, N = argc > 4 ? atoll(argv[4]) : 512;
  const int act = argc > 5 ? atoi(argv[5]) : 1, iters = argc > 6 ? atoi(argv[6]) : 20;
  void* h = dlopen(argv[1], RTLD_NOW);
  if (!h) { printf("dlopen: %s\n", dlerror()); return 1; }
  bytes_fn wbytes = (bytes_fn)dlsym(h, "mi_oov_linear_x3_weights_bytes");
  prep_fn prep = (prep_fn)dlsym(h, "mi_oov_linear_x3_prepare");
  x3_fn x3 = (x3_fn)dlsym(h, "mi_oov_linear_x3");
  lin_fn lin = (lin_fn)dlsym(h, "mi_oov_linear_act");
  if (!wbytes || !prep || !x3 || !lin) { printf("missing symbol\n"); return 1; }
  float *X, *W, *bias, *Y;
  void* ws;
  unsigned long long* cs;
  CK(hipMalloc(&X, B * K * 4)); CK(hipMalloc(&W, N * K * 4)); CK(hipMalloc(&bias, N * 4)); CK(hipMalloc(&Y, B * N * 4));
  CK(hipMalloc(&ws, wbytes(N, K))); CK(hipMalloc(&cs, 8));
  const float zs = getenv("XB_ZERO") ? 0.f : 1.f;  // zero operands: what the matrix pipe does when no bit toggles
  fill_f32<<<2048, 256>>>(X, B * K, 1, 1.0f * zs);
  fill_f32<<<256, 256>>>(W, N * K, 2, 0.03f * zs);
  fill_f32<<<1, 256>>>(bias, N, 3, 0.1f);
  CK(hipDeviceSynchronize());
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int which = 0; which < 2; ++which) {
    if (which == 1 && getenv("XB_SKIP_F32")) break;
    auto run = [&]() { return which == 0 ? x3(X, B, K, ws, bias, N, act, Y, nullptr) : lin(X, B, K, W, bias, N, act, Y, nullptr); };
    if (prep(W, N, K, ws, nullptr)) { printf("prepare failed\n"); return 1; }
    for (int i = 0; i < 3; ++i) if (int rc = run()) { printf("rc %d\n", rc); return 1; }
    CK(hipDeviceSynchronize());
    float best = 1e30f, sum = 0;
    for (int rep = 0; rep < 3; ++rep) {
      CK(hipEventRecord(e0));
      for (int i = 0; i < iters; ++i) run();
      CK(hipEventRecord(e1));
      CK(hipEventSynchronize(e1));
      float ms;
      CK(hipEventElapsedTime(&ms, e0, e1));
      best = ms < best ? ms : best;
      sum += ms;
    }
    CK(hipMemset(cs, 0, 8));
    checksum<<<1024, 256>>>((const uint32_t*)Y, B * N, cs);
    unsigned long long hcs;
    CK(hipMemcpy(&hcs, cs, 8, hipMemcpyDeviceToHost));
    const double us = best * 1e3 / iters, tf = 2.0 * B * K * N / us / 1e6;
    printf("%s %s B=%lld K=%lld N=%lld act=%d: %.1f us (mean %.1f)  %.1f TFLOP/s of the layer  checksum %016llx\n", argv[1],
           which == 0 ? "linear_x3 " : "linear_act", (long long)B, (long long)K, (long long)N, act, us, sum * 1e3 / iters / 3, tf, hcs);
  }
  return 0;
}
