// Integer hashing of the path: SipHash-2-4 (dhe) and the random-mapper mixers.  Bit-exact work.
//   DeepHashEmbedder._get_hashes/_hash_ids       R/inductive/dh_embedder.py:140-170
//   RandomOOVInductiveMapper._*_hash / map_*_ids R/inductive/random_mapper.py:70-130
// SipHash-2-4 is the published algorithm (Aumasson & Bernstein, 2012) that the reference's
// third-party csiphash==0.0.5 implements; it is restated here, not taken from any source tree.
#include "common.hpp"

namespace mi_oov {

// 64-bit lanes are kept as (lo, hi) dword pairs: on gfx950 a 64-bit rotate by r < 32 is two
// v_alignbit_b32, a rotate by 32 is a register rename, add is v_add_co/v_addc, xor is two v_xor.
struct u64p {
  uint32_t lo, hi;
};
__device__ __forceinline__ u64p mk(uint64_t v) { return {static_cast<uint32_t>(v), static_cast<uint32_t>(v >> 32)}; }
// The carry pair is written out: as a uint64_t sum the compiler picks v_lshl_add_u64, whose operands must sit in aligned
// register pairs -- after every rotate by 32 (the rename) it then MOVES both halves back into place: 48 v_mov and 15
// v_perm of the 252 vector instructions a hash took.  Round 3: ~205 per hash, 360 -> 350 us per 65536 x 1024 hashes --
// the moves were cheap; what is left is 62 carry adds + 61 v_alignbit + 72 v_xor per hash at the vector issue rate
// (67 M hashes x 205 / 64 lanes / 1024 SIMDs x 4 cycles = 341 us at 2.4 GHz).
__device__ __forceinline__ u64p add64(u64p a, u64p b) {
  u64p r;
  asm("v_add_co_u32 %0, vcc, %2, %4\n\tv_addc_co_u32 %1, vcc, %3, %5, vcc"
      : "=&v"(r.lo), "=v"(r.hi)
      : "v"(a.lo), "v"(a.hi), "v"(b.lo), "v"(b.hi)
      : "vcc");
  return r;
}
__device__ __forceinline__ u64p xor64(u64p a, u64p b) { return {a.lo ^ b.lo, a.hi ^ b.hi}; }
template <int R>  // 0 < R < 32
__device__ __forceinline__ u64p rotl(u64p x) {
  // alignbit(a, b, s) = low 32 bits of ((a:b) >> s)
  return {__builtin_amdgcn_alignbit(x.lo, x.hi, 32 - R), __builtin_amdgcn_alignbit(x.hi, x.lo, 32 - R)};
}
__device__ __forceinline__ u64p rotl32(u64p x) { return {x.hi, x.lo}; }

// one SipRound = the key-only quarter (v0 += v1; v1 <<<= 13; v1 ^= v0; v0 <<<= 32) + the rest
#define MI_SIPROUND_HEAD           \
  do {                             \
    v0 = add64(v0, v1);            \
    v1 = rotl<13>(v1);             \
    v1 = xor64(v1, v0);            \
    v0 = rotl32(v0);               \
  } while (0)
#define MI_SIPROUND_REST           \
  do {                             \
    v2 = add64(v2, v3);            \
    v3 = rotl<16>(v3);             \
    v3 = xor64(v3, v2);            \
    v0 = add64(v0, v3);            \
    v3 = rotl<21>(v3);             \
    v3 = xor64(v3, v0);            \
    v2 = add64(v2, v1);            \
    v1 = rotl<17>(v1);             \
    v1 = xor64(v1, v2);            \
    v2 = rotl32(v2);               \
  } while (0)
#define MI_SIPROUND                \
  do {                             \
    MI_SIPROUND_HEAD;              \
    MI_SIPROUND_REST;              \
  } while (0)

// What of a hash depends on the key alone: the four initial lanes k ^ "somepseudorandomlygeneratedbytes" and the first
// quarter of round 1 (the message enters through v3 only, and v3 is not touched before the second quarter).  Made once
// per key and workgroup (8 dwords per key in LDS), it takes 14 of the ~210 vector instructions off every hash.
struct SipKey {
  u64p v0, v1, v2, v3;
};
__device__ __forceinline__ SipKey sip_key_state(u64p k0, u64p k1) {
  u64p v0 = xor64(k0, mk(0x736f6d6570736575ULL));
  u64p v1 = xor64(k1, mk(0x646f72616e646f6dULL));
  const u64p v2 = xor64(k0, mk(0x6c7967656e657261ULL));
  const u64p v3 = xor64(k1, mk(0x7465646279746573ULL));
  MI_SIPROUND_HEAD;
  return {v0, v1, v2, v3};
}

// SipHash-2-4 of the 8-byte little-endian encoding of m under the key whose state is ks; returns the low dword
// of the 64-bit hash (all the path needs: `% 16777216`).
__device__ __forceinline__ uint32_t siphash24_lo(const SipKey& ks, u64p m) {
  u64p v0 = ks.v0, v1 = ks.v1, v2 = ks.v2, v3 = xor64(ks.v3, m);
  MI_SIPROUND_REST;
  MI_SIPROUND;
  v0 = xor64(v0, m);
  const u64p last = mk(8ULL << 56);  // message length 8, no tail bytes
  v3 = xor64(v3, last);
  MI_SIPROUND;
  MI_SIPROUND;
  v0 = xor64(v0, last);
  v2.lo ^= 0xff;
  MI_SIPROUND;
  MI_SIPROUND;
  MI_SIPROUND;
  MI_SIPROUND;
  return v0.lo ^ v1.lo ^ v2.lo ^ v3.lo;
}

// Thread -> (id slot, key): KP = min(256, next pow2 >= K) keys per id slot, 256/KP id slots per
// workgroup, so no 64-bit division is needed; j fastest -> a wave writes contiguous floats.  The id
// is wave-uniform for K >= 64; key pairs come from LDS (K*16 B, staged once per workgroup).
template <bool KEYS_IN_LDS>
__global__ __launch_bounds__(kBlock) void siphash_kernel(const int64_t* __restrict__ ids, int64_t B,
                                                         const uint8_t* __restrict__ keys, int K, int log2kp,
                                                         uint32_t mask, float* __restrict__ out, int64_t ld) {
  extern __shared__ __attribute__((aligned(16))) uint32_t skeys[];  // [K][8]: the key states (sip_key_state)
  const uint32_t* kp = reinterpret_cast<const uint32_t*>(keys);    // little-endian host & device
  if (KEYS_IN_LDS) {
    for (int j = threadIdx.x; j < K; j += kBlock) {
      const SipKey ks = sip_key_state({kp[4 * j], kp[4 * j + 1]}, {kp[4 * j + 2], kp[4 * j + 3]});
      *reinterpret_cast<uint4*>(skeys + 8 * j) = make_uint4(ks.v0.lo, ks.v0.hi, ks.v1.lo, ks.v1.hi);
      *reinterpret_cast<uint4*>(skeys + 8 * j + 4) = make_uint4(ks.v2.lo, ks.v2.hi, ks.v3.lo, ks.v3.hi);
    }
    __syncthreads();
  }
  const int KP = 1 << log2kp;
  const int jl = threadIdx.x & (KP - 1);
  const int slot = threadIdx.x >> log2kp;
  const int slots = kBlock >> log2kp;
  for (int64_t b = static_cast<int64_t>(blockIdx.x) * slots + slot; b < B;
       b += static_cast<int64_t>(gridDim.x) * slots) {
    const u64p m = mk(static_cast<uint64_t>(ids[b]));
    for (int j = jl; j < K; j += KP) {
      SipKey ks;
      if (KEYS_IN_LDS) {
        const uint4 a = *reinterpret_cast<const uint4*>(skeys + 8 * j), c = *reinterpret_cast<const uint4*>(skeys + 8 * j + 4);
        ks = {{a.x, a.y}, {a.z, a.w}, {c.x, c.y}, {c.z, c.w}};
      } else {
        ks = sip_key_state({kp[4 * j], kp[4 * j + 1]}, {kp[4 * j + 2], kp[4 * j + 3]});
      }
      const uint32_t h = siphash24_lo(ks, m);
      out[b * ld + j] = static_cast<float>(h & mask);  // < 2^24: exact in f32
    }
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

extern "C" int mi_oov_siphash24_mod_ld(const int64_t* ids, int64_t B, const uint8_t* keys, int64_t K, uint32_t mod,
                                       float* out, int64_t ld, void* stream) {
  using namespace mi_oov;
  if (B < 0 || K <= 0 || ld < K) return MI_OOV_ERR_SHAPE;
  if (mod == 0 || (mod & (mod - 1)) != 0 || mod > (1u << 24)) return MI_OOV_ERR_SHAPE;
  if (B == 0) return MI_OOV_OK;
  if (!ids || !keys || !out) return MI_OOV_ERR_NULL;
  if ((reinterpret_cast<uintptr_t>(keys) & 3u) != 0) return MI_OOV_ERR_ALIGN;
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (K > (1 << 20)) return MI_OOV_ERR_SHAPE;
  int log2kp = 0;
  while ((1 << log2kp) < K && log2kp < 8) ++log2kp;  // KP = min(256, next pow2 >= K)
  const int slots = kBlock >> log2kp;
  const int64_t per_block = static_cast<int64_t>(slots) * (K > 256 ? 1 : 4);
  const int grid = grid_for(B, per_block);
  const size_t lds = static_cast<size_t>(K) * 32;
  if (lds <= 64 * 1024) {
    hipLaunchKernelGGL(siphash_kernel<true>, dim3(grid), dim3(kBlock), lds, st, ids, B, keys, static_cast<int>(K),
                       log2kp, mod - 1, out, ld);
  } else {
    hipLaunchKernelGGL(siphash_kernel<false>, dim3(grid), dim3(kBlock), 0, st, ids, B, keys, static_cast<int>(K),
                       log2kp, mod - 1, out, ld);
  }
  return check_launch();
}

extern "C" int mi_oov_siphash24_mod(const int64_t* ids, int64_t B, const uint8_t* keys, int64_t K, uint32_t mod,
                                    float* out, void* stream) {
  return mi_oov_siphash24_mod_ld(ids, B, keys, K, mod, out, K, stream);
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
