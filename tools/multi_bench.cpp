// Developer harness (not part of the product): times mi_oov_lsh_embed_score_multi against K single-batch
// launches of mi_oov_lsh_embed_score on the headline shape and checks that the two agree bit for bit.
//   hipcc -O2 tools/multi_bench.cpp -o tools/multi_bench -ldl
//   tools/multi_bench <libmi_oov.so> [K=64] [launches=20] [ring_MiB=1024] [N=10000000] [B=65536] [H=8] [distinct=512]
// `distinct` id batches exist in all (>= K); launch i takes batches [i*K, i*K+K) modulo that, so that a row of the
// table gathered by one launch is not found in the Infinity Cache by a later one (512 batches = 8.6 GB of rows).
// The library is dlopen'ed so that variants built with different knobs can be compared in one GPU session.
// Environment: MB_MODE = score (default) | rows | lookup_score | lookup_rows selects which of the four per-batch calls is
// queued (mi_oov_lsh_multi; the single launches it is compared with are mi_oov_lsh_embed_score / _embed / _lookup_score /
// _lookup; lookups take the first half of the table as the in-vocabulary table); MB_PREP=1 hands the launches a
// prepared table of aggregates (mi_oov_lsh_table_prepare).
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>

#include <algorithm>
#include <vector>

#define CK(x)                                                                  \
  do {                                                                         \
    hipError_t e_ = (x);                                                       \
    if (e_ != hipSuccess) {                                                    \
      printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__);    \
      exit(1);                                                                 \
    }                                                                          \
  } while (0)

typedef int (*multi_fn)(const int64_t* const*, const float* const*, float* const*, int64_t, int64_t, const float*, int64_t,
                        int64_t, const float*, int64_t, const float*, int64_t, void*);
typedef int (*single_fn)(const int64_t*, int64_t, const float*, int64_t, int64_t, const float*, int64_t, const float*,
                         int64_t, const float*, float*, float*, void*);
typedef int (*multi2_fn)(int, const int64_t* const*, const float* const*, void* const*, int64_t, int64_t, const float*, int64_t,
                         const float*, int64_t, int64_t, const float*, int64_t, const float*, int64_t, const float*, void*);
typedef int (*embed_fn)(const int64_t*, int64_t, const float*, int64_t, int64_t, const float*, int64_t, const float*, int64_t,
                        float*, uint8_t*, void*);
typedef int (*lookup_fn)(const int64_t*, int64_t, const float*, int64_t, const float*, int64_t, int64_t, const float*, int64_t,
                         const float*, int64_t, float*, void*);
typedef int (*lookup_score_fn)(const int64_t*, int64_t, const float*, int64_t, const float*, int64_t, int64_t, const float*,
                               int64_t, const float*, int64_t, const float*, float*, float*, void*);
typedef int (*prep_fn)(const float*, int64_t, int64_t, float*, void*);

__device__ inline uint64_t mix(uint64_t x) {
  x ^= x >> 33; x *= 0xff51afd7ed558ccdULL; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL; x ^= x >> 33;
  return x;
}
__global__ void fill_f32(float* p, size_t n, uint64_t seed) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const uint64_t h = mix(i + seed * 0x9E3779B97F4A7C15ULL);
    p[i] = (float)((int64_t)(h >> 40) - (1 << 23)) * (1.0f / (1 << 23));  // uniform in [-1, 1)
  }
}
__global__ void fill_ids(int64_t* p, size_t n, uint64_t seed, int64_t N) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    p[i] = (int64_t)(mix(i + seed * 0x9E3779B97F4A7C15ULL) % (uint64_t)N);
}

int main(int argc, char** argv) {
  if (argc < 2) { printf("usage: %s lib [K] [launches] [ring_MiB] [N] [B] [H]\n", argv[0]); return 2; }
  const int64_t K = argc > 2 ? atoll(argv[2]) : 64;
  const int L = argc > 3 ? atoi(argv[3]) : 20;
  const int64_t ring_mib = argc > 4 ? atoll(argv[4]) : 1024;
  const int64_t N = argc > 5 ? atoll(argv[5]) : 10000000;
  const int64_t B = argc > 6 ? atoll(argv[6]) : 65536;
  const int64_t H = argc > 7 ? atoll(argv[7]) : 8;
  const int64_t T = std::max<int64_t>(K, argc > 8 ? atoll(argv[8]) : 512) / K * K;  // distinct batches, multiple of K
  void* lib = dlopen(argv[1], RTLD_LAZY);
  if (!lib) { printf("dlopen: %s\n", dlerror()); return 2; }
  multi_fn multi = (multi_fn)dlsym(lib, "mi_oov_lsh_embed_score_multi");
  single_fn single = (single_fn)dlsym(lib, "mi_oov_lsh_embed_score");
  if (!multi || !single) { printf("missing symbol\n"); return 2; }
  multi2_fn multi2 = (multi2_fn)dlsym(lib, "mi_oov_lsh_multi");
  embed_fn embed1 = (embed_fn)dlsym(lib, "mi_oov_lsh_embed");
  lookup_fn lookup1 = (lookup_fn)dlsym(lib, "mi_oov_lsh_lookup");
  lookup_score_fn lookup_score1 = (lookup_score_fn)dlsym(lib, "mi_oov_lsh_lookup_score");
  prep_fn prep = (prep_fn)dlsym(lib, "mi_oov_lsh_table_prepare");
  const char* mode_s = getenv("MB_MODE") ? getenv("MB_MODE") : "score";
  const bool rows = !strcmp(mode_s, "rows") || !strcmp(mode_s, "lookup_rows");
  const bool lookup = !strncmp(mode_s, "lookup", 6);
  const bool want_prep = getenv("MB_PREP") && atoi(getenv("MB_PREP"));
  if ((rows || lookup || want_prep) && (!multi2 || !prep)) { printf("library has no mi_oov_lsh_multi\n"); return 2; }

  const int64_t R = std::max<int64_t>(1, (ring_mib << 20) / (B * 256));
  float *feat, *planes, *buckets, *users, *sc_m, *sc_s;
  int64_t* ids;
  CK(hipMalloc(&feat, (size_t)N * 256));
  CK(hipMalloc(&planes, H * 256));
  CK(hipMalloc(&buckets, H * 256 + 4096 * 4 * 8));  // + room for a stamps build's per-wave timestamps
  CK(hipMalloc(&users, (size_t)R * B * 256));
  CK(hipMalloc(&ids, (size_t)T * B * 8));
  const size_t osz = rows ? 256 : 4;  // bytes of output per lookup
  const int64_t OR = rows ? std::max<int64_t>(K, R) : T;  // output buffers of the queued launches (rows: a ring)
  CK(hipMalloc(&sc_m, (size_t)OR * B * osz));
  CK(hipMalloc(&sc_s, (size_t)K * B * osz));
  float* tabp = nullptr;
  CK(hipMalloc(&tabp, 256 * 256));
  fill_f32<<<4096, 256>>>(feat, (size_t)N * 64, 1);
  fill_f32<<<64, 256>>>(planes, H * 64, 2);
  fill_f32<<<64, 256>>>(buckets, H * 64, 3);
  fill_f32<<<4096, 256>>>(users, (size_t)R * B * 64, 4);
  fill_ids<<<4096, 256>>>(ids, (size_t)T * B, 5, N);
  CK(hipMemset(sc_m, 0xFF, (size_t)OR * B * osz));
  CK(hipMemset(sc_s, 0xEE, (size_t)K * B * osz));
  std::vector<const int64_t*> h_ids(T);
  std::vector<const float*> h_oth(T);
  std::vector<float*> h_sc(T);
  for (int64_t k = 0; k < T; ++k) {
    h_ids[k] = ids + k * B;
    h_oth[k] = users + (k % R) * B * 64;
    h_sc[k] = reinterpret_cast<float*>(reinterpret_cast<char*>(sc_m) + (size_t)(k % OR) * B * osz);
  }
  const int64_t** d_ids; const float** d_oth; float** d_sc;
  CK(hipMalloc(&d_ids, T * 8)); CK(hipMalloc(&d_oth, T * 8)); CK(hipMalloc(&d_sc, T * 8));
  CK(hipMemcpy(d_ids, h_ids.data(), T * 8, hipMemcpyHostToDevice));
  CK(hipMemcpy(d_oth, h_oth.data(), T * 8, hipMemcpyHostToDevice));
  CK(hipMemcpy(d_sc, h_sc.data(), T * 8, hipMemcpyHostToDevice));
  hipStream_t st;
  CK(hipStreamCreate(&st));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  CK(hipDeviceSynchronize());

  const float* vtab = lookup ? feat : nullptr;
  const int64_t nvoc = lookup ? N / 2 : 0;
  if (want_prep) {
    int rc = prep(buckets, H, 64, tabp, st);
    if (rc) { printf("prepare rc=%d\n", rc); return 1; }
  }
  printf("mode %s%s\n", mode_s, want_prep ? ", prepared table" : "");
  int64_t cursor = 0;
  auto run_multi = [&]() {
    int rc = multi2 ? multi2(rows ? 1 : 0, d_ids + cursor, rows ? nullptr : d_oth + cursor, (void* const*)(d_sc + cursor), K, B, vtab, nvoc,
                             feat, N, 64, planes, H, buckets, 64, want_prep ? tabp : nullptr, st)
                    : multi(d_ids + cursor, d_oth + cursor, d_sc + cursor, K, B, feat, N, 64, planes, H, buckets, 64, st);
    cursor = (cursor + K) % T;
    if (rc) { printf("multi rc=%d\n", rc); exit(1); }
  };
  auto run_single = [&]() {
    for (int64_t k = 0; k < K; ++k) {
      float* o = reinterpret_cast<float*>(reinterpret_cast<char*>(sc_s) + (size_t)k * B * osz);
      int rc = rows ? (lookup ? lookup1(h_ids[k], B, vtab, nvoc, feat, N, 64, planes, H, buckets, 64, o, st)
                              : embed1(h_ids[k], B, feat, N, 64, planes, H, buckets, 64, o, nullptr, st))
                    : (lookup ? lookup_score1(h_ids[k], B, vtab, nvoc, feat, N, 64, planes, H, buckets, 64, h_oth[k], o, nullptr, st)
                              : single(h_ids[k], B, feat, N, 64, planes, H, buckets, 64, h_oth[k], o, nullptr, st));
      if (rc) { printf("single rc=%d\n", rc); exit(1); }
    }
  };
  // correctness first
  run_multi();
  run_single();
  CK(hipStreamSynchronize(st));
  {
    std::vector<uint32_t> a((size_t)K * B * (osz / 4)), b((size_t)K * B * (osz / 4));
    CK(hipMemcpy(a.data(), sc_m, a.size() * 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(b.data(), sc_s, b.size() * 4, hipMemcpyDeviceToHost));
    size_t bad = 0, nan = 0;
    for (size_t i = 0; i < a.size(); ++i) {
      const bool na = (a[i] & 0x7FFFFFFF) > 0x7F800000, nb = (b[i] & 0x7FFFFFFF) > 0x7F800000;
      nan += nb;
      if (na && nb) continue;
      if (a[i] != b[i]) { if (bad < 5) printf("  mismatch at %zu: %08x vs %08x\n", i, a[i], b[i]); ++bad; }
    }
    printf("check: %zu values, %zu NaN (all-zero codes), %zu mismatches vs single-batch launches\n", a.size(), nan, bad);
    if (bad) return 1;
  }
  // clock ramp
  for (int i = 0; i < 30; ++i) run_multi();
  CK(hipStreamSynchronize(st));
  const double bytes = (rows ? 520.0 : 532.0) * B;
  for (int rep = 0; rep < 3; ++rep) {
    CK(hipEventRecord(e0, st));
    for (int i = 0; i < L; ++i) run_multi();
    CK(hipEventRecord(e1, st));
    CK(hipStreamSynchronize(st));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    const double us = ms * 1e3 / (L * K);
    printf("multi : K=%lld x %d launches: %.3f us per batch, %.2f TB/s (%d B/lookup), frac %.3f\n", (long long)K, L, us,
           bytes / us * 1e-6, rows ? 520 : 532, bytes / us * 1e-6 / 8.0);
  }
  if (getenv("MB_ISOLATED")) {  // every launch on its own, the stream drained (and the host asleep) before it
    std::vector<float> t;
    const int idle_us = atoi(getenv("MB_ISOLATED"));
    for (int i = 0; i < 40; ++i) {
      CK(hipStreamSynchronize(st));
      if (idle_us > 0) usleep(idle_us);
      CK(hipEventRecord(e0, st));
      run_multi();
      CK(hipEventRecord(e1, st));
      CK(hipStreamSynchronize(st));
      float ms;
      CK(hipEventElapsedTime(&ms, e0, e1));
      t.push_back(ms * 1e3f / K);
    }
    std::sort(t.begin(), t.end());
    printf("isolated launches (idle %d us before each): per batch min %.3f  p25 %.3f  median %.3f  p75 %.3f  max %.3f us\n",
           idle_us, t[0], t[10], t[20], t[30], t[39]);
  }
  if (getenv("MB_STAMPS")) {  // library built with -DMI_PSTAMPS: per-wave timeline of ONE isolated launch (100 MHz clock)
    CK(hipStreamSynchronize(st));
    CK(hipMemset(reinterpret_cast<char*>(buckets) + H * 256, 0, 4096 * 4 * 8));
    run_multi();
    CK(hipStreamSynchronize(st));
    std::vector<uint64_t> s4(4096 * 4);
    CK(hipMemcpy(s4.data(), reinterpret_cast<char*>(buckets) + H * 256, s4.size() * 8, hipMemcpyDeviceToHost));
    uint64_t t0 = ~0ull;
    int nw = 0;
    for (int w = 0; w < 4096; ++w) if (s4[w * 4 + 3]) { t0 = std::min(t0, s4[w * 4]); ++nw; }
    std::vector<double> a, b, c, d;
    for (int w = 0; w < 4096; ++w) if (s4[w * 4 + 3]) {
      a.push_back((s4[w * 4] - t0) * 0.01); b.push_back((s4[w * 4 + 1] - t0) * 0.01);
      c.push_back((s4[w * 4 + 2] - t0) * 0.01); d.push_back((s4[w * 4 + 3] - t0) * 0.01);
    }
    auto pr = [&](const char* nm, std::vector<double>& v) {
      std::sort(v.begin(), v.end());
      printf("  %-28s min %7.2f  p50 %7.2f  p90 %7.2f  max %7.2f us\n", nm, v[0], v[v.size() / 2], v[v.size() * 9 / 10], v.back());
    };
    printf("stamps of %d waves (us after the first wave's start):\n", nw);
    pr("wave start", a); pr("weights + table ready", b); pr("first tile finished", c); pr("wave end", d);
    // where do the slow waves sit?  workgroup b runs on XCD b % 8 (round-robin dispatch); 8 waves per workgroup
    const int wpb = nw / 256 > 0 ? nw / 256 : 1;
    double xs[8] = {0}, xm[8] = {0}; int xn[8] = {0};
    double in_wg = 0;
    for (int blk = 0; blk * wpb < nw; ++blk) {
      double lo = 1e30, hi = 0;
      for (int w = 0; w < wpb; ++w) {
        const double e = (s4[(blk * wpb + w) * 4 + 3] - t0) * 0.01;
        lo = std::min(lo, e); hi = std::max(hi, e);
        xs[blk % 8] += e; xm[blk % 8] = std::max(xm[blk % 8], e); ++xn[blk % 8];
      }
      in_wg += hi - lo;
    }
    printf("  wave end by XCD (workgroup %% 8): mean");
    for (int x = 0; x < 8; ++x) printf(" %6.1f", xs[x] / std::max(1, xn[x]));
    printf("\n                                   max ");
    for (int x = 0; x < 8; ++x) printf(" %6.1f", xm[x]);
    printf("\n  mean spread of wave ends INSIDE a workgroup: %.2f us\n", in_wg / (nw / wpb));
  }
  for (int rep = 0; rep < 2; ++rep) {
    CK(hipEventRecord(e0, st));
    for (int i = 0; i < 3; ++i) run_single();
    CK(hipEventRecord(e1, st));
    CK(hipStreamSynchronize(st));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    const double us = ms * 1e3 / (3 * K);
    printf("single: %.3f us per batch, %.2f TB/s, frac %.3f (host-launched)\n", us, bytes / us * 1e-6, bytes / us * 1e-6 / 8.0);
  }
  return 0;
}
