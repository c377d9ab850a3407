// Integer hashing of the path: SipHash-2-4 (dhe) and the random-mapper mixers.  Bit-exact work.
//   DeepHashEmbedder._get_hashes/_hash_ids       R/inductive/dh_embedder.py:140-170
//   RandomOOVInductiveMapper._*_hash / map_*_ids R/inductive/random_mapper.py:70-130
// SipHash-2-4 is the published algorithm (Aumasson & Bernstein, 2012) that the reference's
// third-party csiphash==0.0.5 implements; it is restated here, not taken from any source tree.
#include "common.hpp"

namespace mi_oov {

__device__ __forceinline__ uint64_t rotl64(uint64_t x, int b) { return (x << b) | (x >> (64 - b)); }

#define MI_SIPROUND        \
  do {                     \
    v0 += v1;              \
    v1 = rotl64(v1, 13);   \
    v1 ^= v0;              \
    v0 = rotl64(v0, 32);   \
    v2 += v3;              \
    v3 = rotl64(v3, 16);   \
    v3 ^= v2;              \
    v0 += v3;              \
    v3 = rotl64(v3, 21);   \
    v3 ^= v0;              \
    v2 += v1;              \
    v1 = rotl64(v1, 17);   \
    v1 ^= v2;              \
    v2 = rotl64(v2, 32);   \
  } while (0)

// SipHash-2-4 of the 8-byte little-endian encoding of m under key (k0,k1).
__device__ __forceinline__ uint64_t siphash24_u64(uint64_t k0, uint64_t k1, uint64_t m) {
  uint64_t v0 = k0 ^ 0x736f6d6570736575ULL;
  uint64_t v1 = k1 ^ 0x646f72616e646f6dULL;
  uint64_t v2 = k0 ^ 0x6c7967656e657261ULL;
  uint64_t v3 = k1 ^ 0x7465646279746573ULL;
  v3 ^= m;
  MI_SIPROUND;
  MI_SIPROUND;
  v0 ^= m;
  const uint64_t last = 8ULL << 56;  // message length 8, no tail bytes
  v3 ^= last;
  MI_SIPROUND;
  MI_SIPROUND;
  v0 ^= last;
  v2 ^= 0xff;
  MI_SIPROUND;
  MI_SIPROUND;
  MI_SIPROUND;
  MI_SIPROUND;
  return v0 ^ v1 ^ v2 ^ v3;
}

// out[b, j] for j fastest: a wave writes 256 contiguous bytes; the id is wave-uniform for
// K >= 64 and the key pair comes from LDS (K*16 B, staged once per workgroup).
template <bool KEYS_IN_LDS>
__global__ __launch_bounds__(kBlock) void siphash_kernel(const int64_t* __restrict__ ids, int64_t B,
                                                         const uint8_t* __restrict__ keys, int64_t K,
                                                         uint32_t mask, float* __restrict__ out) {
  extern __shared__ __attribute__((aligned(16))) uint64_t skeys[];
  const uint64_t* kp = reinterpret_cast<const uint64_t*>(keys);  // little-endian host & device
  if (KEYS_IN_LDS) {
    for (int64_t i = threadIdx.x; i < 2 * K; i += kBlock) skeys[i] = kp[i];
    __syncthreads();
  }
  const int64_t total = B * K;
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x; i < total;
       i += static_cast<int64_t>(gridDim.x) * kBlock) {
    const int64_t b = i / K;
    const int64_t j = i - b * K;
    const uint64_t k0 = KEYS_IN_LDS ? skeys[2 * j] : kp[2 * j];
    const uint64_t k1 = KEYS_IN_LDS ? skeys[2 * j + 1] : kp[2 * j + 1];
    const uint64_t h = siphash24_u64(k0, k1, static_cast<uint64_t>(ids[b]));
    out[i] = static_cast<float>(static_cast<uint32_t>(h) & mask);  // < 2^24: exact in f32
  }
}

// ---- random mapper ---------------------------------------------------------------------------
__device__ __forceinline__ int64_t mul_wrap(int64_t a, uint64_t c) {
  return static_cast<int64_t>(static_cast<uint64_t>(a) * c);
}

// random_mapper.py:70-76 (signed int64 tensor ops: arithmetic >>, wrapping *)
__device__ __forceinline__ int64_t fast_int_hash(int64_t x) {
  x ^= (x >> 16);
  x = mul_wrap(x, 0x21f0aaadULL);
  x ^= (x >> 15);
  x = mul_wrap(x, 0xd35a2d97ULL);
  x ^= (x >> 15);
  return x;
}

// random_mapper.py:78-86
__device__ __forceinline__ int64_t three_round_int_hash(int64_t x) {
  x ^= (x >> 17);
  x = mul_wrap(x, 0xed5ad4bbULL);
  x ^= (x >> 11);
  x = mul_wrap(x, 0xac4c1b51ULL);
  x ^= (x >> 15);
  x = mul_wrap(x, 0x31848babULL);
  x ^= (x >> 14);
  return x;
}

// random_mapper.py:95-102: numpy uint64; the constants are the splitmix64 multipliers decoded
// with byteorder='little', i.e. byte-reversed: 0xb9e5e41c6d4758bf and 0xeb113113bb49d094.
__device__ __forceinline__ uint64_t big_64bit_hash(uint64_t x) {
  x = (x ^ (x >> 30)) * 0xb9e5e41c6d4758bfULL;
  x = (x ^ (x >> 27)) * 0xeb113113bb49d094ULL;
  x = x ^ (x >> 31);
  return x;
}

__device__ __forceinline__ int64_t pymod(int64_t x, int64_t n) {
  int64_t r = x % n;
  return (r < 0) ? r + n : r;
}

__device__ __forceinline__ int64_t raw_hash(int kind, int64_t x) {
  switch (kind) {
    case MI_OOV_HASH_FAST: return fast_int_hash(x);
    case MI_OOV_HASH_3ROUND: return three_round_int_hash(x);
    case MI_OOV_HASH_64BIT: return static_cast<int64_t>(big_64bit_hash(static_cast<uint64_t>(x)));
    default: return x;
  }
}

__device__ __forceinline__ int64_t bucket_of(int kind, int64_t x, int64_t n) {
  if (kind == MI_OOV_HASH_64BIT)
    return static_cast<int64_t>(big_64bit_hash(static_cast<uint64_t>(x)) % static_cast<uint64_t>(n));
  return pymod(raw_hash(kind, x), n);
}

__global__ __launch_bounds__(kBlock) void mapper_kernel(const int64_t* __restrict__ ids, int64_t B, int kind,
                                                        int64_t n_orig, int64_t n_buckets, bool map,
                                                        int64_t* __restrict__ out) {
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x; i < B;
       i += static_cast<int64_t>(gridDim.x) * kBlock) {
    const int64_t id = ids[i];
    int64_t r;
    if (!map) {
      r = raw_hash(kind, id);
    } else {
      r = (id < n_orig) ? id : bucket_of(kind, id - n_orig, n_buckets) + n_orig;
    }
    out[i] = r;
  }
}

}  // namespace mi_oov

extern "C" int mi_oov_siphash24_mod(const int64_t* ids, int64_t B, const uint8_t* keys, int64_t K, uint32_t mod,
                                    float* out, void* stream) {
  using namespace mi_oov;
  if (B < 0 || K <= 0) return MI_OOV_ERR_SHAPE;
  if (mod == 0 || (mod & (mod - 1)) != 0 || mod > (1u << 24)) return MI_OOV_ERR_SHAPE;
  if (B == 0) return MI_OOV_OK;
  if (!ids || !keys || !out) return MI_OOV_ERR_NULL;
  if ((reinterpret_cast<uintptr_t>(keys) & 7u) != 0) return MI_OOV_ERR_ALIGN;
  hipStream_t st = static_cast<hipStream_t>(stream);
  const int grid = grid_for(B * K, kBlock * 4);
  const size_t lds = static_cast<size_t>(K) * 16;
  if (lds <= 48 * 1024) {
    hipLaunchKernelGGL(siphash_kernel<true>, dim3(grid), dim3(kBlock), lds, st, ids, B, keys, K, mod - 1, out);
  } else {
    hipLaunchKernelGGL(siphash_kernel<false>, dim3(grid), dim3(kBlock), 0, st, ids, B, keys, K, mod - 1, out);
  }
  return check_launch();
}

static int run_mapper(const int64_t* ids, int64_t B, int kind, int64_t n_orig, int64_t n_buckets, bool map,
                      int64_t* out, void* stream) {
  using namespace mi_oov;
  if (kind < MI_OOV_HASH_MOD || kind > MI_OOV_HASH_64BIT) return MI_OOV_ERR_KIND;
  if (B < 0 || (map && n_buckets <= 0)) return MI_OOV_ERR_SHAPE;
  if (B == 0) return MI_OOV_OK;
  if (!ids || !out) return MI_OOV_ERR_NULL;
  hipStream_t st = static_cast<hipStream_t>(stream);
  const int grid = grid_for(B, kBlock * 4);
  hipLaunchKernelGGL(mapper_kernel, dim3(grid), dim3(kBlock), 0, st, ids, B, kind, n_orig, n_buckets, map, out);
  return check_launch();
}

extern "C" int mi_oov_mapper_hash(const int64_t* ids, int64_t B, int kind, int64_t* out, void* stream) {
  return run_mapper(ids, B, kind, 0, 1, false, out, stream);
}

extern "C" int mi_oov_mapper_map(const int64_t* ids, int64_t B, int kind, int64_t n_orig, int64_t n_buckets,
                                 int64_t* out, void* stream) {
  return run_mapper(ids, B, kind, n_orig, n_buckets, true, out, stream);
}
