// Device helpers shared by the F = D = 64 lsh kernels (lsh64.hip: one launch per batch; lsh64p.hip: the
// persistent multi-batch launch).  Every kernel that includes this file performs the same additions in the same
// order, which is what keeps them bit-identical to each other and to oracle/oov_oracle.c.
#pragma once
#include "common.hpp"

namespace mi_oov {

// ---- eight row sums at once ---------------------------------------------------------------------------
// row16_sum's tree is (l, l+8), (l, l+4), (l, l+2), (l, l+1).  After level 1 a sum needs only 8 lanes, after
// level 2 only 4: bank_mask (a DPP row = 4 banks of 4 lanes) lets a second plane's level-1 result be written
// into lanes 8-15 of the same register, and two such registers be folded into one at level 2 (row_shl:4
// feeds banks 0 and 2, row_shr:4 banks 1 and 3).  16 DPP adds instead of 32, every individual addition the
// same as in row16_sum (IEEE add is commutative), so the totals are bit-identical.  Result layout:
//   t0: bank 0 plane 0 | bank 1 plane 2 | bank 2 plane 1 | bank 3 plane 3       (all 4 lanes of a bank equal)
//   t1: bank 0 plane 4 | bank 1 plane 6 | bank 2 plane 5 | bank 3 plane 7
// Inline asm because the compiler cannot express a partially masked v_add_f32_dpp; the s_nop's are the
// "VALU write -> DPP read" wait states (2) the hazard recognizer would otherwise insert.
__device__ __forceinline__ void rows8_sum(const float (&p)[8], float& t0, float& t1) {
  float r0, r1, r2, r3;
  asm volatile(
      "s_nop 1\n\t"
      "v_add_f32_dpp %2, %6, %6 row_ror:8 row_mask:0xf bank_mask:0xf\n\t"
      "v_add_f32_dpp %3, %8, %8 row_ror:8 row_mask:0xf bank_mask:0xf\n\t"
      "v_add_f32_dpp %4, %10, %10 row_ror:8 row_mask:0xf bank_mask:0xf\n\t"
      "v_add_f32_dpp %5, %12, %12 row_ror:8 row_mask:0xf bank_mask:0xf\n\t"
      "v_add_f32_dpp %2, %7, %7 row_ror:8 row_mask:0xf bank_mask:0xc\n\t"
      "v_add_f32_dpp %3, %9, %9 row_ror:8 row_mask:0xf bank_mask:0xc\n\t"
      "v_add_f32_dpp %4, %11, %11 row_ror:8 row_mask:0xf bank_mask:0xc\n\t"
      "v_add_f32_dpp %5, %13, %13 row_ror:8 row_mask:0xf bank_mask:0xc\n\t"
      "v_add_f32_dpp %0, %2, %2 row_shl:4 row_mask:0xf bank_mask:0x5\n\t"
      "v_add_f32_dpp %1, %4, %4 row_shl:4 row_mask:0xf bank_mask:0x5\n\t"
      "v_add_f32_dpp %0, %3, %3 row_shr:4 row_mask:0xf bank_mask:0xa\n\t"
      "v_add_f32_dpp %1, %5, %5 row_shr:4 row_mask:0xf bank_mask:0xa\n\t"
      "s_nop 0\n\t"
      "v_add_f32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
      "v_add_f32_dpp %1, %1, %1 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 0\n\t"
      "v_add_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
      "v_add_f32_dpp %1, %1, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 0"
      : "=&v"(t0), "=&v"(t1), "=&v"(r0), "=&v"(r1), "=&v"(r2), "=&v"(r3)
      : "v"(p[0]), "v"(p[1]), "v"(p[2]), "v"(p[3]), "v"(p[4]), "v"(p[5]), "v"(p[6]), "v"(p[7]));
}

// bit of plane h for every lane of the row, from the banked registers of rows8_sum (row_newbcast:lane)
template <int CTRL>
__device__ __forceinline__ float dpp_all_f32(float v) {  // every lane is written: no `old` value to preserve
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, true));
}
__device__ __forceinline__ float bcast8(int h, float b0, float b1) {
  switch (h) {
    case 0: return dpp_all_f32<0x150>(b0);   // t0 bank 0
    case 1: return dpp_all_f32<0x158>(b0);   // t0 bank 2
    case 2: return dpp_all_f32<0x154>(b0);   // t0 bank 1
    case 3: return dpp_all_f32<0x15C>(b0);   // t0 bank 3
    case 4: return dpp_all_f32<0x150>(b1);
    case 5: return dpp_all_f32<0x158>(b1);
    case 6: return dpp_all_f32<0x154>(b1);
    default: return dpp_all_f32<0x15C>(b1);
  }
}

// emb = acc / cnt, correctly rounded (0/0 -> NaN row, lsh_embedder.py:178).  cnt is a small integer (0..32),
// so instead of four IEEE division sequences (~9 VALU each) one reciprocal rc = RN(1/cnt) is shared and each
// quotient is refined once:  q = a rc;  e = fma(-q, cnt, a);  q' = fma(e, rc, q).
// Verified EXHAUSTIVELY on the CPU (same IEEE fma) for cnt = 1..32 and all 2^32 values of a: q' equals
// RN(a / cnt) except for a = -0 (acc is never -0: it starts at +0 and every step is RN(bit*w + acc)) and for a
// handful of |a| < 2^-120 whose quotient is subnormal.  Lanes whose |acc| is below 2^-100 (including exact
// zeros, i.e. the cnt = 0 rows) or infinite take the IEEE division.
// rc itself: v_rcp_f32 is accurate to 1 ulp, and one Newton step y' = fma(fma(-cnt, y, 1), y, y) from a 1-ulp
// estimate is the correctly rounded reciprocal unless the significand of cnt is all ones (Markstein); cnt = 0
// gives NaN here, and that row takes the IEEE branch anyway.
__device__ __forceinline__ float4 masked_mean(float4 acc, float cnt) {
  const float y0 = __builtin_amdgcn_rcpf(cnt);
  const float rc = __builtin_fmaf(__builtin_fmaf(-cnt, y0, 1.0f), y0, y0);
  float4 emb;
  {
    float q;
    q = acc.x * rc; emb.x = __builtin_fmaf(__builtin_fmaf(-q, cnt, acc.x), rc, q);
    q = acc.y * rc; emb.y = __builtin_fmaf(__builtin_fmaf(-q, cnt, acc.y), rc, q);
    q = acc.z * rc; emb.z = __builtin_fmaf(__builtin_fmaf(-q, cnt, acc.z), rc, q);
    q = acc.w * rc; emb.w = __builtin_fmaf(__builtin_fmaf(-q, cnt, acc.w), rc, q);
  }
  const float amin = fminf(fminf(fabsf(acc.x), fabsf(acc.y)), fminf(fabsf(acc.z), fabsf(acc.w)));
  const float amax = fmaxf(fmaxf(fabsf(acc.x), fabsf(acc.y)), fmaxf(fabsf(acc.z), fabsf(acc.w)));
  if (!(amin >= 0x1p-100f) || !(amax < __builtin_inff()) || !(cnt <= 32.f)) {  // divisors above 32: not verified
    emb.x = acc.x / cnt;
    emb.y = acc.y / cnt;
    emb.z = acc.z / cnt;
    emb.w = acc.w / cnt;
  }
  return emb;
}


// The lsh embedding of ONE gathered feature row (lsh_embedder.py:133-179): the lane's 4-float slice of
// emb = (bits @ W) / popcount(bits), bits_h = (x . P_h >= 0).  Same operations in the same order as lsh64_tile
// (lsh64.hip): dot4_fma chain per lane, stride-halving 16-lane tree (bank-masked for H = 8), fmaf chain over the
// bucket rows in plane order, one correctly rounded division.
template <int H>
__device__ __forceinline__ float4 lsh64_embed_row(const float4& x, const float4 (&pw)[H], const float4 (&bw)[H]) {
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  float cnt = 0.f;
  if constexpr (H == 8) {
    float p[8];
#pragma unroll
    for (int h = 0; h < 8; ++h) p[h] = dot4_fma(x, pw[h], 0.f);
    float t0, t1;
    rows8_sum(p, t0, t1);
    const float b0 = (t0 < 0.f) ? 0.f : 1.f;  // >= 0, +-0 and NaN -> 1 (torch_hash.py:57-59)
    const float b1 = (t1 < 0.f) ? 0.f : 1.f;
    cnt = b0 + b1;
    cnt = cnt + dpp_f32<0x124>(cnt);
    cnt = cnt + dpp_f32<0x128>(cnt);
#pragma unroll
    for (int h = 0; h < 8; ++h) {
      const float bit = bcast8(h, b0, b1);
      acc.x = __builtin_fmaf(bit, bw[h].x, acc.x);
      acc.y = __builtin_fmaf(bit, bw[h].y, acc.y);
      acc.z = __builtin_fmaf(bit, bw[h].z, acc.z);
      acc.w = __builtin_fmaf(bit, bw[h].w, acc.w);
    }
  } else {
#pragma unroll
    for (int h = 0; h < H; ++h) {
      const float s = row16_sum(dot4_fma(x, pw[h], 0.f));
      const float bit = (s < 0.f) ? 0.f : 1.f;
      cnt = cnt + bit;
      acc.x = __builtin_fmaf(bit, bw[h].x, acc.x);
      acc.y = __builtin_fmaf(bit, bw[h].y, acc.y);
      acc.z = __builtin_fmaf(bit, bw[h].z, acc.z);
      acc.w = __builtin_fmaf(bit, bw[h].w, acc.w);
    }
  }
  return masked_mean(acc, cnt);
}

}  // namespace mi_oov
