// Linear layers of the hash nets on the bf16 matrix cores at f32 accuracy ("split-bf16").
//
//   Y = act(X W^T + bias)      (R/inductive/dh_embedder.py:70-89, feat_dh_embedder.py:108-127, dnn_embedder.py:65-90)
//
// The f32 matrix instruction (v_mfma_f32_32x32x2_f32, mi_oov_linear_act) runs at 1/16 of the bf16 one's rate, and the
// 1024-512-512-512-64 net of dhe is 142 GFLOP per 65536 lookups: 1.28 ms at 70 % of the f32 matrix peak.  Here every
// f32 operand is held as THREE bf16 values h + m + l (round-to-nearest pieces of the running remainder: 3 x 8 = 24
// significand bits, so h + m + l is the f32 value exactly) and six of the nine cross products are accumulated in f32 --
// l*h, h*l, m*m, m*h, h*m, h*h; the three left out (m*l, l*m, l*l) are below 2^-24 of |x||w| each.  Every bf16 x bf16
// product is exact in f32, so what is lost is lost in the matrix instruction's own accumulation (not a single rounding
// of the exact 32-term sum: a CPU model of that does not reproduce its bits) -- measured against f64 the same error
// as the f32 kernel's or smaller (tests/test_gpu_parity.py::test_linear_x3_*).  It is NOT the oracle's summation order: parity is within a tolerance
// written in the test, not bit for bit -- mi_oov_linear_act stays the bit-exact form (MI_OOV_LINEAR_X3=0 on the host).
//
// The arithmetic, shared by both kernels of this file (they agree bit for bit): v_mfma_f32_16x16x32_bf16 with TWO planes
// side by side in its 32 k, so that 16 k of the operands take three instructions per 16 x 16 output block,
//   [x_h | x_l] . [w_l | w_h],   [x_h | x_m] . [w_m | w_h],   [x_h | x_m] . [w_h | w_m]      (in this order),
// the product taken transposed (W rows as the A operand: a lane ends with four consecutive output columns of one row).
//   X  : f32 rows straight from the producer -- split into the three planes while staging (5.5 vector operations per
//        element, v_cvt_pk_bf16_f32), so no layer has to write anything but plain f32;
//   W  : split ONCE per weight update by linear_x3_split_kernel into [K/16][N padded][3 planes][16] bf16 -- a stage of a
//        workgroup is one contiguous piece of it, and the pipelined kernel's LDS image as it stands.
// LDS row = the three planes of 16 k side by side (96 B, no padding): conflict-free for the lane groups of the
// ds_read_b128 that fetches a fragment (lane l: row l & 15, k half (l >> 4) & 1, plane-of-the-pair l >> 5).
// Kernels: linear_x3_kernel (any shape; register-staged, double-buffered LDS; also the split-K form of training's
// weight gradients) and linear_x3_fast_kernel (K % 16 == 0, wide outputs, enough tiles: persistent, W by LDS-DMA).
#include "common.hpp"

namespace mi_oov {

typedef __bf16 l3_bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 l3_bf16x2 __attribute__((ext_vector_type(2)));
typedef float l3_f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int l3_u32x4 __attribute__((ext_vector_type(4)));

constexpr int kL3NPad = 256;  // rows of the split weights are padded to a multiple of this (any tile shape fits)

__host__ __device__ constexpr int64_t l3_np(int64_t N) { return (N + kL3NPad - 1) / kL3NPad * kL3NPad; }
__host__ __device__ constexpr int64_t l3_chunks(int64_t K) { return (K + 15) / 16; }

// Two f32 -> their three bf16 planes, packed (low half = the first value).  Finite values only: an infinite operand (or
// a finite one above the largest bf16, 3.39e38, which rounds to inf in h) leaves inf - inf = NaN in the lower planes, so
// where the f32 product would hold +-inf or NaN this one holds NaN -- non-finite in, non-finite out.
__device__ __forceinline__ void split3_pair(float x0, float x1, uint32_t& h, uint32_t& m, uint32_t& l) {
  const l3_bf16x2 hb = __builtin_convertvector(l3_f32x2{x0, x1}, l3_bf16x2);
  h = __builtin_bit_cast(uint32_t, hb);
  const float h0 = __uint_as_float(h << 16), h1 = __uint_as_float(h & 0xFFFF0000u);
  const float r0 = x0 - h0, r1 = x1 - h1;
  const l3_bf16x2 mb = __builtin_convertvector(l3_f32x2{r0, r1}, l3_bf16x2);
  m = __builtin_bit_cast(uint32_t, mb);
  const float m0 = __uint_as_float(m << 16), m1 = __uint_as_float(m & 0xFFFF0000u);
  const l3_bf16x2 lb = __builtin_convertvector(l3_f32x2{r0 - m0, r1 - m1}, l3_bf16x2);
  l = __builtin_bit_cast(uint32_t, lb);
}

__device__ __forceinline__ void split3_x8(const float (&x)[8], l3_u32x4& h, l3_u32x4& m, l3_u32x4& l) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    uint32_t a, b, c;
    split3_pair(x[2 * i], x[2 * i + 1], a, b, c);
    h[i] = a, m[i] = b, l[i] = c;
  }
}

// 8 consecutive k of one row (zeros beyond the matrix)
template <bool VEC>
__device__ __forceinline__ void load8(const float* __restrict__ A, int64_t row, int64_t rows, int64_t K, int64_t k0, float (&x)[8]) {
  if (row < rows && VEC && k0 + 8 <= K) {
    const float4 a = *reinterpret_cast<const float4*>(A + row * K + k0);
    const float4 b = *reinterpret_cast<const float4*>(A + row * K + k0 + 4);
    x[0] = a.x, x[1] = a.y, x[2] = a.z, x[3] = a.w, x[4] = b.x, x[5] = b.y, x[6] = b.z, x[7] = b.w;
  } else {
#pragma unroll
    for (int i = 0; i < 8; ++i) x[i] = (row < rows && k0 + i < K) ? A[row * K + k0 + i] : 0.f;
  }
}

// W f32[N,K] -> split [K/16][Np][3][16] bf16: a row of a chunk is the pipelined kernel's LDS row as it stands (6 units of
// 16 bytes: unit (chunk * Np + n) * 6 + plane * 2 + half), so a stage of a workgroup is one contiguous piece that an
// LDS-DMA copies without a register in between.
template <bool TRANSPOSED>  // TRANSPOSED: W is given as its transpose, f32[K,N] row-major (training: dW's and dX's operands as they lie)
__global__ __launch_bounds__(kBlock) void linear_x3_split_kernel(const float* __restrict__ W, int64_t N, int64_t K,
                                                                 l3_u32x4* __restrict__ out) {
  const int64_t Np = l3_np(N), units = l3_chunks(K) * Np * 2;
  for (int64_t u = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x; u < units; u += static_cast<int64_t>(gridDim.x) * kBlock) {
    // plain: (half, n, chunk) with the half fastest -- a pair of threads reads 64 contiguous bytes of a row;
    // transposed: (n, half, chunk) with n fastest -- a wave reads 64 consecutive floats of one k row at a time
    const int half = TRANSPOSED ? static_cast<int>((u / Np) & 1) : static_cast<int>(u & 1);
    const int64_t n = TRANSPOSED ? u % Np : (u >> 1) % Np, chunk = (u >> 1) / Np;
    float x[8];
    if (TRANSPOSED) {
      const int64_t k0 = chunk * 16 + half * 8;
#pragma unroll
      for (int i = 0; i < 8; ++i) x[i] = (n < N && k0 + i < K) ? W[(k0 + i) * N + n] : 0.f;
    } else {
      load8<false>(W, n, N, K, chunk * 16 + half * 8, x);
    }
    l3_u32x4 h, m, l;
    split3_x8(x, h, m, l);
    l3_u32x4* dst = out + (chunk * Np + n) * 6 + half;
    dst[0] = h, dst[2] = m, dst[4] = l;
  }
}

template <int ACT>
__device__ __forceinline__ float l3_act(float v) {
  if (ACT == MI_OOV_ACT_SIGMOID) return 1.0f / (1.0f + expf(-v));  // (GELU: l3_gelu2 below)
  return v;
}

// nn.GELU() (erf form) of two values, 0.5 v (1 + erf(v / sqrt 2)), for the tiles' epilogues: the device library's erff is
// two polynomial branches that a wave of mixed arguments executes both of (~35 instructions per value; a 256 x 256 tile
// has 128 values per lane and nothing else runs on the CU meanwhile).  Here erfc(t) = 2^(-t Q(t)) on t = min(|v| / sqrt 2, 4),
// Q of degree 8 (weighted minimax fit, tools/fit_gelu.py), one v_exp_f32, packed f32 arithmetic: |erf error| <= 1.0e-7, and
// over v in [-8, 8] the result is within 2.5e-7 (8.1e-8 max(|v|, 1)) of the exact GELU -- the reference's own expression
// evaluated in f32 with a perfect erf: 4.5e-7 (1.1e-7 max(|v|, 1)); its 1 + erf cancels for negative v.
typedef float l3_f2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ l3_f2 l3_gelu2(l3_f2 v) {
  const l3_f2 t = __builtin_elementwise_min(__builtin_elementwise_abs(v * 0.70710678118654752440f), l3_f2{4.f, 4.f});
  l3_f2 q = l3_f2{1.160470219e-05f, 1.160470219e-05f};  // -Q: the exponent comes out negated
  q = __builtin_elementwise_fma(q, t, l3_f2{-1.529632864e-04f, -1.529632864e-04f});
  q = __builtin_elementwise_fma(q, t, l3_f2{8.482300327e-04f, 8.482300327e-04f});
  q = __builtin_elementwise_fma(q, t, l3_f2{-2.274776343e-03f, -2.274776343e-03f});
  q = __builtin_elementwise_fma(q, t, l3_f2{8.479427197e-05f, 8.479427197e-05f});
  q = __builtin_elementwise_fma(q, t, l3_f2{2.772448398e-02f, 2.772448398e-02f});
  q = __builtin_elementwise_fma(q, t, l3_f2{-1.483079195e-01f, -1.483079195e-01f});
  q = __builtin_elementwise_fma(q, t, l3_f2{-9.184429049e-01f, -9.184429049e-01f});
  q = __builtin_elementwise_fma(q, t, l3_f2{-1.627907276e+00f, -1.627907276e+00f});
  const l3_f2 a = t * q;
  const l3_f2 e = l3_f2{__builtin_amdgcn_exp2f(a.x), __builtin_amdgcn_exp2f(a.y)};  // erfc(t)
  const l3_f2 h = (v * 0.5f) * e;
  const l3_f2 pos = v - h;  // v > 0: 0.5 v (2 - erfc t)
  return l3_f2{v.x > 0.f ? pos.x : h.x, v.y > 0.f ? pos.y : h.y};
}

// The generic tile kernel: any K (zero-padded to 16), any alignment, (64 WM) x (32 NB WN) tiles, both operands staged
// through registers into a double-buffered LDS stage (one barrier per stage).  The SAME arithmetic as the pipelined
// kernel below, in the same order -- v_mfma_f32_16x16x32_bf16 with two planes side by side in its 32 k, three
// instructions per 16 x 16 output block and 16 k, product taken transposed -- so the two agree bit for bit and a result
// does not depend on which of them a call's shape selects.
typedef float l3_f32x4 __attribute__((ext_vector_type(4)));
constexpr int kL3Row = 48;  // bf16 elements per LDS row: 3 planes x 16 k (96 B: conflict-free for the 16x16x32 fragment reads)

template <int WM, int WN, int NB, int ACT, bool VEC>
__global__ __launch_bounds__(64 * WM * WN) void linear_x3_kernel(const float* __restrict__ X, int64_t B, int64_t K,
                                                                 const l3_u32x4* __restrict__ Wp, int64_t N,
                                                                 const float* __restrict__ bias, float* __restrict__ Y,
                                                                 int n_nblk, int n_mblk, int ksplit) {
  constexpr int T = 64 * WM * WN, BMt = 64 * WM, BNt = 32 * NB * WN;
  constexpr int XU = (BMt * 2 + T - 1) / T;  // 8-float units of X per thread and stage
  constexpr int WU = (BNt * 6 + T - 1) / T;  // 16-byte units of the split W per thread and stage
  constexpr int kStage = (BMt + BNt) * kL3Row;  // bf16 elements
  constexpr int NQ = 2 * NB;                    // 16-column blocks of a wave
  extern __shared__ __attribute__((aligned(16))) unsigned short l3_lds[];

  // Workgroups that share rows of X (the n-blocks of one m-block) sit next to each other on ONE XCD (consecutive
  // workgroup ids go round the eight XCDs): the X tile comes out of HBM / the Infinity Cache once per L2.
  const int id = blockIdx.x, xcd = id & 7, slot = id >> 3;
  const int nb_i = slot % n_nblk, mb_i = (slot / n_nblk) * 8 + xcd;
  if (mb_i >= n_mblk) return;
  const int64_t b0 = static_cast<int64_t>(mb_i) * BMt, n0 = static_cast<int64_t>(nb_i) * BNt;
  const int64_t Np = l3_np(N);
  // split K (training's dW = dZ^T X: a few output tiles, K = the batch): workgroup (x, y) takes the y-th share of the
  // 16-k stages and leaves its sums, no bias, no activation, in slab y of Y = the workspace [ksplit][B][N];
  // linear_x3_reduce_kernel adds the slabs in order.  ksplit = 1: the whole K, Y the output.
  const int nst_all = static_cast<int>(l3_chunks(K));
  const int per = (nst_all + ksplit - 1) / ksplit;
  const int st0 = static_cast<int>(blockIdx.y) * per;
  const int nst = (st0 + per < nst_all) ? st0 + per : nst_all;  // stages [st0, nst)
  if (ksplit > 1) Y += static_cast<int64_t>(blockIdx.y) * B * N;

  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int wm = wv / WN, wn = wv % WN;
  const int r16 = lane & 15, kh = (lane >> 4) & 1, ps = lane >> 5, q4 = lane >> 4;

  l3_f32x4 acc[NQ][4];  // [16-column block][16-row block]
#pragma unroll
  for (int n = 0; n < NQ; ++n)
#pragma unroll
    for (int m = 0; m < 4; ++m) acc[n][m] = l3_f32x4{0.f, 0.f, 0.f, 0.f};

  float xr[XU][8];
  l3_u32x4 wr[WU];
  auto gload = [&](int s) {
#pragma unroll
    for (int j = 0; j < XU; ++j) {
      const int u = tid + T * j;
      if (BMt * 2 % T == 0 || u < BMt * 2) load8<VEC>(X, b0 + (u >> 1), B, K, static_cast<int64_t>(s) * 16 + (u & 1) * 8, xr[j]);
    }
    const l3_u32x4* src = Wp + (static_cast<int64_t>(s) * Np + n0) * 6;
#pragma unroll
    for (int j = 0; j < WU; ++j) {
      const int u = tid + T * j;
      if (BNt * 6 % T == 0 || u < BNt * 6) wr[j] = src[u];
    }
  };
  auto lstore = [&](int buf) {
    unsigned short* sX = l3_lds + buf * kStage;
    unsigned short* sW = sX + BMt * kL3Row;
#pragma unroll
    for (int j = 0; j < XU; ++j) {
      const int u = tid + T * j;
      if (BMt * 2 % T == 0 || u < BMt * 2) {
        l3_u32x4 h, m, l;
        split3_x8(xr[j], h, m, l);
        unsigned short* d = sX + (u >> 1) * kL3Row + (u & 1) * 8;
        *reinterpret_cast<l3_u32x4*>(d) = h;
        *reinterpret_cast<l3_u32x4*>(d + 16) = m;
        *reinterpret_cast<l3_u32x4*>(d + 32) = l;
      }
    }
#pragma unroll
    for (int j = 0; j < WU; ++j) {
      const int u = tid + T * j;
      if (BNt * 6 % T == 0 || u < BNt * 6) *reinterpret_cast<l3_u32x4*>(sW + u * 8) = wr[j];  // (the image's rows are the LDS rows)
    }
  };
  // fragment of 16 rows x [plane p0 | plane p1]: lane (r16, kh, ps) reads 8 k of plane (ps ? p1 : p0)
  auto frag = [&](const unsigned short* base, int row0, int p0, int p1) {
    return __builtin_bit_cast(l3_bf16x8, *reinterpret_cast<const l3_u32x4*>(base + (row0 + r16) * kL3Row + (ps ? p1 : p0) * 16 + kh * 8));
  };

  if (st0 < nst) {
    gload(st0);
    lstore(st0 & 1);
  }
  __syncthreads();
  for (int s = st0; s < nst; ++s) {
    if (s + 1 < nst) gload(s + 1);
    const unsigned short* sX = l3_lds + (s & 1) * kStage;
    const unsigned short* sW = sX + BMt * kL3Row;
    l3_bf16x8 xhm[4], xhl[4];
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      xhm[m] = frag(sX, wm * 64 + m * 16, 0, 1);
      xhl[m] = frag(sX, wm * 64 + m * 16, 0, 2);
    }
#pragma unroll
    for (int n = 0; n < NQ; ++n) {
      const l3_bf16x8 wlh = frag(sW, wn * 32 * NB + n * 16, 2, 0);  // [w_l | w_h]
      const l3_bf16x8 wmh = frag(sW, wn * 32 * NB + n * 16, 1, 0);  // [w_m | w_h]
      const l3_bf16x8 whm = frag(sW, wn * 32 * NB + n * 16, 0, 1);  // [w_h | w_m]
#pragma unroll
      for (int m = 0; m < 4; ++m) {  // small terms first
        acc[n][m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wlh, xhl[m], acc[n][m], 0, 0, 0);
        acc[n][m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wmh, xhm[m], acc[n][m], 0, 0, 0);
        acc[n][m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(whm, xhm[m], acc[n][m], 0, 0, 0);
      }
    }
    if (s + 1 < nst) lstore((s + 1) & 1);
    __syncthreads();
  }

  // C/D map of the 16x16 shapes, product taken transposed: lane (r16, q4) holds X row r16 of its m-block and the
  // four output columns 4 q4 .. 4 q4 + 3 of its n-block
  const bool vec4 = (N % 4 == 0) && ((reinterpret_cast<uintptr_t>(Y) & 15u) == 0);
#pragma unroll
  for (int n = 0; n < NQ; ++n) {
    const int64_t col = n0 + wn * 32 * NB + n * 16 + q4 * 4;
    float bc[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) bc[r] = (ksplit == 1 && col + r < N) ? bias[col + r] : 0.f;
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      const int64_t row = b0 + wm * 64 + m * 16 + r16;
      l3_f2 y0 = l3_f2{acc[n][m][0] + bc[0], acc[n][m][1] + bc[1]}, y1 = l3_f2{acc[n][m][2] + bc[2], acc[n][m][3] + bc[3]};
      if (ACT == MI_OOV_ACT_GELU) y0 = l3_gelu2(y0), y1 = l3_gelu2(y1);
      if (ACT == MI_OOV_ACT_SIGMOID) y0 = l3_f2{l3_act<ACT>(y0.x), l3_act<ACT>(y0.y)}, y1 = l3_f2{l3_act<ACT>(y1.x), l3_act<ACT>(y1.y)};
      if (row < B) {
        float* dst = Y + row * N + col;
        if (vec4 && col + 3 < N) {
          *reinterpret_cast<l3_f32x4*>(dst) = l3_f32x4{y0.x, y0.y, y1.x, y1.y};
        } else {
          if (col + 0 < N) dst[0] = y0.x;
          if (col + 1 < N) dst[1] = y0.y;
          if (col + 2 < N) dst[2] = y1.x;
          if (col + 3 < N) dst[3] = y1.y;
        }
      }
    }
  }
}

// ---- the pipelined form: K % 16 == 0, X 16-byte aligned ----------------------------------------------------------------
// 8 waves, a 256 x 256 tile (each wave 64 x 128), one persistent workgroup per CU, two waves per SIMD.  Bytes that
// reach a CU per matrix instruction are what bounds a tile (an XCD's L2 serves ~30 B/clk/CU with every CU reading):
// 16 KB of X + 24 KB of W per 16 k = 13 B/clk/CU (a 128 x 256 tile: 21).
//   W : LDS-DMA (global_load_lds_dwordx4: 1 KiB per wave instruction, no register in between) TWO stages ahead into a
//       ring of three stage images -- the split weights are stored as the LDS image (24 pieces per stage, 3 per wave);
//   X : one register set per thread (8 floats = half of one row's 16 k): split and written into the NEXT stage's image
//       (ring of two) while the matrix instructions of this stage run, then re-issued at once for the stage after.
// One raw s_barrier per stage; the waits are counted by hand (the DMA is not in the compiler's bookkeeping): before the
// barrier of stage s everything but this wave's loads for stage s + 2 has landed.
//
// The matrix instruction is v_mfma_f32_16x16x32_bf16 with TWO planes side by side in its 32 k: for 16 k of the operands
//   [x_h | x_l] . [w_l | w_h] = x_h w_l + x_l w_h      [x_h | x_m] . [w_m | w_h] = x_h w_m + x_m w_h
//   [x_h | x_m] . [w_h | w_m] = x_h w_h + x_m w_m
// -- the same six products in three instructions of the 16 x 16 shape.  With every CU multiplying random operands the
// chip's clock gives way (power), and it gives way less for this shape: 2.1 PFLOP/s against 1.8 for 32x32x16 in a bare
// register loop (tools/mfma_power.cpp; both 2.46 on zeros).  Lane l of a fragment holds row l & 15, k half (l >> 4) & 1
// of plane-of-the-pair l >> 5: one ds_read_b128 at row * 96 + plane * 32 + half * 16 -- 96-byte rows (no padding) are
// conflict-free for the lane groups of that read.  The product is taken transposed (W rows as the A operand), so a
// lane ends with four CONSECUTIVE output columns of one row: 16-byte stores.
constexpr int kFastM = 256, kFastN = 256, kFastT = 512;
constexpr int kFastRow = 48;                      // bf16 elements per LDS / image row: 3 planes x 16 k
constexpr int kFastWStage = kFastN * kFastRow * 2;  // bytes of a W stage image (24 576 = 24 DMA pieces of 1 KiB)
constexpr int kFastXStage = kFastM * kFastRow * 2;
constexpr int kFastLds = 3 * kFastWStage + 2 * kFastXStage;  // 122 880 B

__device__ __forceinline__ void glds16(const void* gsrc, uint32_t lds_dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep)
               : "v"(gsrc), "s"(lds_dst)
               : "memory");
}

template <int ACT, bool VEC4>
__global__ __launch_bounds__(kFastT) void linear_x3_fast_kernel(const float* __restrict__ X, int64_t B, int64_t K,
                                                               const l3_u32x4* __restrict__ Wp, int64_t N,
                                                               const float* __restrict__ bias, float* __restrict__ Y,
                                                               int n_nblk, int n_mblk) {
  extern __shared__ __attribute__((aligned(16))) unsigned short l3_lds[];  // [3][W stage][2][X stage]
  const int64_t Np = l3_np(N);
  const int nst = static_cast<int>(K / 16);  // >= 2 (host)
  const int total = (n_mblk + 7) / 8 * 8 * n_nblk, step = gridDim.x;  // step % 8 == 0: a workgroup's tiles stay on its XCD

  const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wv >> 1, wn = wv & 1;
  const int r16 = lane & 15, kh = (lane >> 4) & 1, ps = lane >> 5, q4 = lane >> 4;
  const uint32_t lds_base = static_cast<uint32_t>(reinterpret_cast<uintptr_t>((__attribute__((address_space(3))) unsigned short*)l3_lds));

  // Tiles: id = slot * 8 + xcd; the n-blocks of one m-block are consecutive slots of ONE XCD (consecutive workgroup
  // ids go round the eight XCDs), so the X rows they share come out of HBM / the Infinity Cache once per L2.  A
  // workgroup walks ids blockIdx.x, + gridDim.x, ... (persistent: one workgroup per CU) and treats its tiles as ONE
  // stream of stages -- the loads of the next tile's first two stages are issued during the last two of this one, the
  // accumulators are stored in between, nothing restarts.
  struct Tile { int64_t b0, n0; const float* xsrc; const char* wsrc; };
  const int64_t wstep = Np * (kFastRow * 2);
  auto valid = [&](int id) { return id < total && ((id >> 3) / n_nblk) * 8 + (id & 7) < n_mblk; };
  auto next_valid = [&](int id) { while (id < total && !valid(id)) id += step; return id; };
  auto decode = [&](int id) {
    Tile t;
    const int slot = id >> 3;
    t.b0 = static_cast<int64_t>((slot / n_nblk) * 8 + (id & 7)) * kFastM;
    t.n0 = static_cast<int64_t>(slot % n_nblk) * kFastN;
    const int64_t xrow = (t.b0 + (tid >> 1) < B) ? t.b0 + (tid >> 1) : B - 1;  // rows beyond B: the last row again (never stored)
    t.xsrc = X + xrow * K + (tid & 1) * 8;
    t.wsrc = reinterpret_cast<const char*>(Wp) + t.n0 * (kFastRow * 2) + wv * 1024 + lane * 16;
    return t;
  };
  int id = next_valid(blockIdx.x);
  if (id >= total) return;
  Tile cur = decode(id);
  int nid = next_valid(id + step);
  Tile nxt = decode(nid < total ? nid : id);

  unsigned short* xdst = l3_lds + 3 * (kFastWStage / 2) + (tid >> 1) * kFastRow + (tid & 1) * 8;
  auto dma_w = [&](const Tile& t, int s, int g) {  // stage s of tile t = stage g of the stream: W ring slot g % 3
    const uint32_t dst = lds_base + static_cast<uint32_t>(g % 3) * kFastWStage + wv * 1024;
    const char* src = t.wsrc + static_cast<int64_t>(s) * wstep;
#pragma unroll
    for (int j = 0; j < 3; ++j) glds16(src + j * 8192, dst + j * 8192);  // pieces wv, wv + 8, wv + 16 of the image's 24
  };
  auto dma_w_piece = [&](const Tile& t, int s, int g, int j) {
    glds16(t.wsrc + static_cast<int64_t>(s) * wstep + j * 8192, lds_base + static_cast<uint32_t>(g % 3) * kFastWStage + wv * 1024 + j * 8192);
  };
  float4 xa, xb;
  auto load_x = [&](const Tile& t, int s) {
    xa = *reinterpret_cast<const float4*>(t.xsrc + static_cast<int64_t>(s) * 16);
    xb = *reinterpret_cast<const float4*>(t.xsrc + static_cast<int64_t>(s) * 16 + 4);
  };
  auto store_x = [&](int g) {
    const float x[8] = {xa.x, xa.y, xa.z, xa.w, xb.x, xb.y, xb.z, xb.w};
    l3_u32x4 h, m, l;
    split3_x8(x, h, m, l);
    unsigned short* d = xdst + (g & 1) * (kFastXStage / 2);
    *reinterpret_cast<l3_u32x4*>(d) = h;
    *reinterpret_cast<l3_u32x4*>(d + 16) = m;
    *reinterpret_cast<l3_u32x4*>(d + 32) = l;
  };
  // fragment of 16 rows x [plane p0 | plane p1]: lane (r16, kh, ps) reads 8 k of plane (ps ? p1 : p0)
  auto frag = [&](const unsigned short* base, int row0, int p0, int p1) {
    return __builtin_bit_cast(l3_bf16x8, *reinterpret_cast<const l3_u32x4*>(base + (row0 + r16) * kFastRow + (ps ? p1 : p0) * 16 + kh * 8));
  };

  int g = 0;  // stage of the stream
  dma_w(cur, 0, 0);
  dma_w(cur, 1, 1);
  load_x(cur, 0);
  store_x(0);
  load_x(cur, 1);
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();

  while (true) {
    const bool has_next = nid < total;
    l3_f32x4 acc[8][4];  // [n-block of 16 columns][m-block of 16 rows]
#pragma unroll
    for (int n = 0; n < 8; ++n)
#pragma unroll
      for (int m = 0; m < 4; ++m) acc[n][m] = l3_f32x4{0.f, 0.f, 0.f, 0.f};

    for (int s = 0; s < nst; ++s, ++g) {
      // stage g + 2 of the stream: stage s + 2 of this tile, or stage s + 2 - nst of the next one
      const bool wrap2 = s + 2 >= nst, more2 = !wrap2 || has_next;
      const bool wrap1 = s + 1 >= nst, more1 = !wrap1 || has_next;
      const Tile& t2 = wrap2 ? nxt : cur;
      const int s2 = wrap2 ? s + 2 - nst : s + 2;
      const unsigned short* sW = l3_lds + (g % 3) * (kFastWStage / 2);
      const unsigned short* sX = l3_lds + 3 * (kFastWStage / 2) + (g & 1) * (kFastXStage / 2);
      l3_bf16x8 xhm[4], xhl[4], w[2][3];
#pragma unroll
      for (int m = 0; m < 4; ++m) {
        xhm[m] = frag(sX, wm * 64 + m * 16, 0, 1);
        xhl[m] = frag(sX, wm * 64 + m * 16, 0, 2);
      }
      auto read_w = [&](int n, l3_bf16x8 (&d)[3]) {
        d[0] = frag(sW, wn * 128 + n * 16, 2, 0);  // [w_l | w_h]
        d[1] = frag(sW, wn * 128 + n * 16, 1, 0);  // [w_m | w_h]
        d[2] = frag(sW, wn * 128 + n * 16, 0, 1);  // [w_h | w_m]
      };
      read_w(0, w[0]);
      l3_u32x4 ch, cm, cl;  // the split of stage g + 1's X, made two pairs at a time between the matrix instructions
      const float cx[8] = {xa.x, xa.y, xa.z, xa.w, xb.x, xb.y, xb.z, xb.w};
#pragma unroll
      for (int n = 0; n < 8; ++n) {
        // The stage's one barrier sits behind n-block 4, not at its end: by then the next stage's X image is written and
        // this wave's W pieces for it (issued behind the barrier of the stage before) have landed, so the barrier
        // publishes stage g + 1 -- and the last matrix instruction of a stage runs straight into the next stage's
        // fragment reads, the two waves of a SIMD free to drift apart there.  Behind it: the DMA of W two stages ahead
        // (its ring slot was last read in stage g - 1, which every wave has left once it is past this barrier).
        // sched_barrier(0) pins the pieces between the n-blocks: left to itself the compiler makes one block of the
        // ~60 vector instructions of the split, during which neither wave of the SIMD issues a matrix instruction.
        __builtin_amdgcn_sched_barrier(0);
        if (n + 1 < 8) read_w(n + 1, w[(n + 1) & 1]);
        if (more1 && (n == 1 || n == 2)) {
#pragma unroll
          for (int i = (n - 1) * 2; i < (n - 1) * 2 + 2; ++i) {
            uint32_t a, b, c;
            split3_pair(cx[2 * i], cx[2 * i + 1], a, b, c);
            ch[i] = a, cm[i] = b, cl[i] = c;
          }
        }
        if (more1 && n == 3) {
          unsigned short* d = xdst + ((g + 1) & 1) * (kFastXStage / 2);
          *reinterpret_cast<l3_u32x4*>(d) = ch;
          *reinterpret_cast<l3_u32x4*>(d + 16) = cm;
          *reinterpret_cast<l3_u32x4*>(d + 32) = cl;
          if (more2) load_x(t2, s2);
        }
        if (n == 5) {
          if (more2) asm volatile("s_waitcnt vmcnt(2) lgkmcnt(0)" ::: "memory");  // (younger: the two X loads just issued)
          else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
          __builtin_amdgcn_s_barrier();
        }
        if (more2 && n >= 5) dma_w_piece(t2, s2, g + 2, n - 5);
        __builtin_amdgcn_sched_barrier(0);
        const l3_bf16x8* ww = w[n & 1];
#pragma unroll
        for (int m = 0; m < 4; ++m) {  // small terms first
          acc[n][m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ww[0], xhl[m], acc[n][m], 0, 0, 0);
          acc[n][m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ww[1], xhm[m], acc[n][m], 0, 0, 0);
          acc[n][m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ww[2], xhm[m], acc[n][m], 0, 0, 0);
        }
      }
    }

    // C/D map of the 16x16 shapes, product taken transposed: lane (r16, q4) holds X row r16 of its m-block and the
    // four output columns 4 q4 .. 4 q4 + 3 of its n-block
#pragma unroll
    for (int n = 0; n < 8; ++n) {
      const int64_t col = cur.n0 + wn * 128 + n * 16 + q4 * 4;
      float bc[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) bc[r] = (col + r < N) ? bias[col + r] : 0.f;
#pragma unroll
      for (int m = 0; m < 4; ++m) {
        const int64_t row = cur.b0 + wm * 64 + m * 16 + r16;
        l3_f2 y0 = l3_f2{acc[n][m][0] + bc[0], acc[n][m][1] + bc[1]}, y1 = l3_f2{acc[n][m][2] + bc[2], acc[n][m][3] + bc[3]};
        if (ACT == MI_OOV_ACT_GELU) y0 = l3_gelu2(y0), y1 = l3_gelu2(y1);
        if (ACT == MI_OOV_ACT_SIGMOID) y0 = l3_f2{l3_act<ACT>(y0.x), l3_act<ACT>(y0.y)}, y1 = l3_f2{l3_act<ACT>(y1.x), l3_act<ACT>(y1.y)};
        if (row < B) {
          float* dst = Y + row * N + col;
          if (VEC4 && col + 3 < N) {
            *reinterpret_cast<l3_f32x4*>(dst) = l3_f32x4{y0.x, y0.y, y1.x, y1.y};
          } else {
            if (col + 0 < N) dst[0] = y0.x;
            if (col + 1 < N) dst[1] = y0.y;
            if (col + 2 < N) dst[2] = y1.x;
            if (col + 3 < N) dst[3] = y1.y;
          }
        }
      }
    }
    if (!has_next) break;
    id = nid;
    cur = nxt;
    nid = next_valid(id + step);
    nxt = decode(nid < total ? nid : id);
  }
}

static int l3_cus() {  // compute units of the current device, rounded down to a multiple of 8 (the XCDs)
  static const int cus = [] {
    int dev = 0, n = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n < 8) n = 256;
    return n / 8 * 8;
  }();
  return cus;
}

template <int ACT>
static int launch_x3_fast(const float* X, int64_t B, int64_t K, const void* wsplit, const float* bias, int64_t N, float* Y, hipStream_t st) {
  const int64_t n_mblk = (B + kFastM - 1) / kFastM, n_nblk = (N + kFastN - 1) / kFastN;
  const int64_t total = (n_mblk + 7) / 8 * 8 * n_nblk;
  if (total > 0x7FFFFFFF) return MI_OOV_ERR_SHAPE;
  const int cus = l3_cus();
  const int64_t grid = total < cus ? total : cus;  // one workgroup per CU (143 KB of LDS each); total % 8 == 0
  const bool vec4 = (N % 4 == 0) && aligned16(Y);  // rows of Y 16-byte aligned: one store per four columns
  auto k = vec4 ? linear_x3_fast_kernel<ACT, true> : linear_x3_fast_kernel<ACT, false>;
  if (int rc = set_lds(k, kFastLds)) return rc;
  hipLaunchKernelGGL(k, dim3(static_cast<unsigned>(grid)), dim3(kFastT), kFastLds, st, X, B, K, static_cast<const l3_u32x4*>(wsplit), N, bias, Y,
                     static_cast<int>(n_nblk), static_cast<int>(n_mblk));
  return check_launch();
}

template <int WM, int WN, int NB, int ACT, bool VEC>
static int launch_x3(const float* X, int64_t B, int64_t K, const void* wsplit, const float* bias, int64_t N, float* Y, hipStream_t st,
                     int ksplit = 1) {
  constexpr int T = 64 * WM * WN, BMt = 64 * WM, BNt = 32 * NB * WN;
  const int64_t n_mblk = (B + BMt - 1) / BMt, n_nblk = (N + BNt - 1) / BNt;
  const int64_t grid = (n_mblk + 7) / 8 * 8 * n_nblk;
  if (grid > 0x7FFFFFFF || n_mblk > 0x7FFFFFFF || ksplit < 1 || ksplit > 65535) return MI_OOV_ERR_SHAPE;
  const size_t lds = 2 * static_cast<size_t>(BMt + BNt) * kL3Row * sizeof(unsigned short);
  auto k = linear_x3_kernel<WM, WN, NB, ACT, VEC>;
  if (int rc = set_lds(k, lds)) return rc;
  hipLaunchKernelGGL(k, dim3(static_cast<unsigned>(grid), static_cast<unsigned>(ksplit)), dim3(T), lds, st, X, B, K,
                     static_cast<const l3_u32x4*>(wsplit), N, bias, Y, static_cast<int>(n_nblk), static_cast<int>(n_mblk), ksplit);
  return check_launch();
}

// Y = act(slab 0 + slab 1 + ... (in this order) + bias): the second half of a split-K product
template <int ACT>
__global__ __launch_bounds__(kBlock) void linear_x3_reduce_kernel(const float* __restrict__ part, int ksplit, int64_t total, int64_t N,
                                                                  const float* __restrict__ bias, float* __restrict__ Y) {
  for (int64_t i = (static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x) * 2; i < total; i += static_cast<int64_t>(gridDim.x) * kBlock * 2) {
    const bool two = i + 1 < total;
    l3_f2 v = l3_f2{0.f, 0.f};
    for (int z = 0; z < ksplit; ++z) {
      const float* p = part + static_cast<int64_t>(z) * total + i;
      v = v + l3_f2{p[0], two ? p[1] : 0.f};
    }
    v = v + l3_f2{bias[i % N], two ? bias[(i + 1) % N] : 0.f};
    if (ACT == MI_OOV_ACT_GELU) v = l3_gelu2(v);
    if (ACT == MI_OOV_ACT_SIGMOID) v = l3_f2{l3_act<ACT>(v.x), l3_act<ACT>(v.y)};
    Y[i] = v.x;
    if (two) Y[i + 1] = v.y;
  }
}

template <int WM, int WN, int NB>
static int launch_x3_act(const float* X, int64_t B, int64_t K, const void* wsplit, const float* bias, int64_t N, int act, float* Y, hipStream_t st) {
  const bool vec = (K % 4 == 0) && aligned16(X);
#define MI_L3(A)                                                                                   \
  return vec ? launch_x3<WM, WN, NB, A, true>(X, B, K, wsplit, bias, N, Y, st)                     \
             : launch_x3<WM, WN, NB, A, false>(X, B, K, wsplit, bias, N, Y, st);
  switch (act) {
    case MI_OOV_ACT_NONE: MI_L3(MI_OOV_ACT_NONE)
    case MI_OOV_ACT_GELU: MI_L3(MI_OOV_ACT_GELU)
    default: MI_L3(MI_OOV_ACT_SIGMOID)
  }
#undef MI_L3
}

}  // namespace mi_oov

using namespace mi_oov;

extern "C" int64_t mi_oov_linear_x3_weights_bytes(int64_t N_out, int64_t K) {
  if (N_out <= 0 || K <= 0) return MI_OOV_ERR_SHAPE;
  return l3_chunks(K) * l3_np(N_out) * 96;
}

static int x3_prepare(const float* W, int64_t N_out, int64_t K, void* wsplit, void* stream, bool transposed) {
  if (N_out <= 0 || K <= 0) return MI_OOV_ERR_SHAPE;
  if (!W || !wsplit) return MI_OOV_ERR_NULL;
  if (!aligned16(wsplit)) return MI_OOV_ERR_ALIGN;
  const int64_t units = l3_chunks(K) * l3_np(N_out) * 2;
  const int64_t grid = (units + kBlock - 1) / kBlock;
  const dim3 g(static_cast<unsigned>(grid < kMaxGrid ? grid : kMaxGrid));
  if (transposed)
    hipLaunchKernelGGL(linear_x3_split_kernel<true>, g, dim3(kBlock), 0, static_cast<hipStream_t>(stream), W, N_out, K, static_cast<l3_u32x4*>(wsplit));
  else
    hipLaunchKernelGGL(linear_x3_split_kernel<false>, g, dim3(kBlock), 0, static_cast<hipStream_t>(stream), W, N_out, K, static_cast<l3_u32x4*>(wsplit));
  return check_launch();
}

extern "C" int mi_oov_linear_x3_prepare(const float* W, int64_t N_out, int64_t K, void* wsplit, void* stream) {
  return x3_prepare(W, N_out, K, wsplit, stream, false);
}

extern "C" int mi_oov_linear_x3_prepare_t(const float* Wt, int64_t N_out, int64_t K, void* wsplit, void* stream) {
  return x3_prepare(Wt, N_out, K, wsplit, stream, true);
}

extern "C" int mi_oov_linear_x3(const float* X, int64_t B, int64_t K, const void* wsplit, const float* bias, int64_t N_out,
                                int act, float* Y, void* stream) {
  if (B < 0 || K <= 0 || N_out <= 0) return MI_OOV_ERR_SHAPE;
  if (act < 0 || act > 2) return MI_OOV_ERR_KIND;
  if (B == 0) return MI_OOV_OK;
  if (!X || !wsplit || !bias || !Y) return MI_OOV_ERR_NULL;
  if (!aligned16(wsplit)) return MI_OOV_ERR_ALIGN;
  hipStream_t st = static_cast<hipStream_t>(stream);
  // developer knob, read per call (the tests force the pipelined kernel onto small shapes with it): 0 = by shape,
  // 1 / 2 / 3 = the generic kernel with 128 x 128 / 256 x 256 / 128 x 64 tiles, 4 = the pipelined kernel where it applies.
  // Every form does the same arithmetic in the same order: results do not depend on it.
  const int shape = static_cast<int>(env_knob("MI_OOV_X3_SHAPE", 0, 0, 4));
  // the pipelined 256 x 256 form: K a multiple of 16 (no tail chunk), rows of X 16-byte aligned, outputs wider than 128
  // ... when its 256 x 256 tiles fill more than half of the CUs: a tile is 122 us of one CU at K = 1024 however few
  // there are, so smaller batches are quicker as four times as many 128 x 128 tiles (65536 x 1024 -> 512, us by rows,
  // pipelined / 128 x 128: 4096 126 / 76, 16384 136 / 117, 24576 153 / 184, 65536 337 / 433)
  const int64_t tiles256 = ((B + kFastM - 1) / kFastM) * ((N_out + kFastN - 1) / kFastN);
  if ((shape == 4 || (shape == 0 && tiles256 > l3_cus() / 2)) && K % 16 == 0 && K >= 32 && aligned16(X) && N_out > 128) {
    switch (act) {
      case MI_OOV_ACT_NONE: return launch_x3_fast<MI_OOV_ACT_NONE>(X, B, K, wsplit, bias, N_out, Y, st);
      case MI_OOV_ACT_GELU: return launch_x3_fast<MI_OOV_ACT_GELU>(X, B, K, wsplit, bias, N_out, Y, st);
      default: return launch_x3_fast<MI_OOV_ACT_SIGMOID>(X, B, K, wsplit, bias, N_out, Y, st);
    }
  }
  // narrow outputs (the last layer of the nets, 64 wide): a 128 x 64 tile; otherwise 128 x 256
  // ... and 128 x 64 ones when even the 128 x 128 tiles would leave three CUs in four idle (1024 rows x 1024 -> 512: 51 us
  // against 58; a small batch is bound by the serial walk of a tile over K, not by its matrix instructions)
  const int64_t tiles128 = ((B + 127) / 128) * ((N_out + 127) / 128);
  if ((N_out <= 64 || tiles128 <= l3_cus() / 4) && shape == 0) return launch_x3_act<2, 2, 1>(X, B, K, wsplit, bias, N_out, act, Y, st);
  if ((N_out <= 128 || tiles256 <= l3_cus() / 2) && shape == 0) return launch_x3_act<2, 2, 2>(X, B, K, wsplit, bias, N_out, act, Y, st);
  switch (shape) {
    case 1: return launch_x3_act<2, 2, 2>(X, B, K, wsplit, bias, N_out, act, Y, st);
    case 2: return launch_x3_act<4, 2, 4>(X, B, K, wsplit, bias, N_out, act, Y, st);
    case 3: return launch_x3_act<2, 2, 1>(X, B, K, wsplit, bias, N_out, act, Y, st);
    default: return launch_x3_act<2, 2, 4>(X, B, K, wsplit, bias, N_out, act, Y, st);
  }
}

extern "C" int64_t mi_oov_linear_x3_splitk_workspace(int64_t B, int64_t N_out, int64_t ksplit) {
  if (B < 0 || N_out <= 0 || ksplit < 1 || ksplit > 65535) return MI_OOV_ERR_SHAPE;
  return ksplit * B * N_out * static_cast<int64_t>(sizeof(float));
}

extern "C" int mi_oov_linear_x3_splitk(const float* X, int64_t B, int64_t K, const void* wsplit, const float* bias, int64_t N_out,
                                       int act, float* Y, int64_t ksplit, void* workspace, void* stream) {
  if (B < 0 || K <= 0 || N_out <= 0 || ksplit < 1 || ksplit > 65535) return MI_OOV_ERR_SHAPE;
  if (act < 0 || act > 2) return MI_OOV_ERR_KIND;
  if (B == 0) return MI_OOV_OK;
  if (!X || !wsplit || !bias || !Y || !workspace) return MI_OOV_ERR_NULL;
  if (!aligned16(wsplit) || !aligned16(workspace)) return MI_OOV_ERR_ALIGN;
  hipStream_t st = static_cast<hipStream_t>(stream);
  float* part = static_cast<float*>(workspace);
  const bool vec = (K % 4 == 0) && aligned16(X);
  const int rc = vec ? launch_x3<2, 2, 2, MI_OOV_ACT_NONE, true>(X, B, K, wsplit, bias, N_out, part, st, static_cast<int>(ksplit))
                     : launch_x3<2, 2, 2, MI_OOV_ACT_NONE, false>(X, B, K, wsplit, bias, N_out, part, st, static_cast<int>(ksplit));
  if (rc) return rc;
  const int64_t total = B * N_out, units = (total + 2 * kBlock - 1) / (2 * kBlock);
  const dim3 grid(static_cast<unsigned>(units < kMaxGrid ? units : kMaxGrid));
  const int ks = static_cast<int>(ksplit);
  switch (act) {
    case MI_OOV_ACT_NONE: hipLaunchKernelGGL(linear_x3_reduce_kernel<MI_OOV_ACT_NONE>, grid, dim3(kBlock), 0, st, part, ks, total, N_out, bias, Y); break;
    case MI_OOV_ACT_GELU: hipLaunchKernelGGL(linear_x3_reduce_kernel<MI_OOV_ACT_GELU>, grid, dim3(kBlock), 0, st, part, ks, total, N_out, bias, Y); break;
    default: hipLaunchKernelGGL(linear_x3_reduce_kernel<MI_OOV_ACT_SIGMOID>, grid, dim3(kBlock), 0, st, part, ks, total, N_out, bias, Y); break;
  }
  return check_launch();
}
