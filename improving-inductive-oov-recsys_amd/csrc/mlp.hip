// Training-side pieces of the dhe / fdhe / dnn hash nets (R/inductive/dh_embedder.py:70-89,191-217,
// feat_dh_embedder.py:108-127, dnn_embedder.py:65-109): what torch autograd does for
//   nn.Sequential(Linear, GELU, Linear, GELU, Linear, GELU, Linear, Sigmoid)
// restated on this library's own kernels.  The three GEMMs of a Linear's backward
//   dX = dZ W            dW = dZ^T X            db = 1^T dZ
// all run on mi_oov_full_sort_scores (the f32-MFMA tiled kernel of score.hip: one fmaf chain per element over
// increasing k) once the operands are laid out as [rows, k]; this file supplies the layout change (transpose) and the
// two elementwise passes around the activation:
//   forward (training)   Z = X W^T + b  (mi_oov_linear_act, act = identity)   Y = act(Z)   (mi_oov_act_forward)
//   backward             dZ = dY * act'(Z)                                                  (mi_oov_act_backward)
// act'(z):  GELU (erf form, nn.GELU())   Phi(z) + z phi(z)      Sigmoid   y (1 - y), y = 1 / (1 + exp(-z))
// Memory-bound elementwise kernels; the activation expressions are those of the fused forward epilogue (score.hip).
#include "common.hpp"

namespace mi_oov {

__device__ __forceinline__ float act_fwd(float v, int act) {
  if (act == MI_OOV_ACT_GELU) return 0.5f * v * (1.0f + erff(v * 0.70710678118654752440f));
  if (act == MI_OOV_ACT_SIGMOID) return 1.0f / (1.0f + expf(-v));
  return v;
}

__device__ __forceinline__ float act_grad(float z, int act) {
  if (act == MI_OOV_ACT_GELU) {
    const float cdf = 0.5f * (1.0f + erff(z * 0.70710678118654752440f));
    const float pdf = expf(-0.5f * z * z) * 0.39894228040143267794f;  // 1 / sqrt(2 pi)
    return cdf + z * pdf;
  }
  if (act == MI_OOV_ACT_SIGMOID) {
    const float y = 1.0f / (1.0f + expf(-z));
    return (1.0f - y) * y;
  }
  return 1.0f;
}

__global__ __launch_bounds__(256) void act_forward_kernel(const float* __restrict__ Z, int64_t n, int act, float* __restrict__ Y) {
  const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x * 4;
  for (int64_t i = (static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x) * 4; i < n; i += stride) {
    if (i + 4 <= n) {
      float4 z = *reinterpret_cast<const float4*>(Z + i);
      z.x = act_fwd(z.x, act); z.y = act_fwd(z.y, act); z.z = act_fwd(z.z, act); z.w = act_fwd(z.w, act);
      *reinterpret_cast<float4*>(Y + i) = z;
    } else {
      for (int64_t j = i; j < n; ++j) Y[j] = act_fwd(Z[j], act);
    }
  }
}

__global__ __launch_bounds__(256) void act_backward_kernel(const float* __restrict__ dY, const float* __restrict__ Z, int64_t n,
                                                          int act, float* __restrict__ dZ) {
  const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x * 4;
  for (int64_t i = (static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x) * 4; i < n; i += stride) {
    if (i + 4 <= n) {
      const float4 z = *reinterpret_cast<const float4*>(Z + i);
      float4 g = *reinterpret_cast<const float4*>(dY + i);
      g.x = g.x * act_grad(z.x, act); g.y = g.y * act_grad(z.y, act);
      g.z = g.z * act_grad(z.z, act); g.w = g.w * act_grad(z.w, act);
      *reinterpret_cast<float4*>(dZ + i) = g;
    } else {
      for (int64_t j = i; j < n; ++j) dZ[j] = dY[j] * act_grad(Z[j], act);
    }
  }
}

// At[c, r] = A[r, c]: 64 x 64 tiles through LDS (padded rows: conflict-free both ways), coalesced 256-B reads and writes
__global__ __launch_bounds__(256) void transpose_kernel(const float* __restrict__ A, int64_t R, int64_t C, float* __restrict__ At) {
  __shared__ float tile[64][65];
  const int64_t r0 = static_cast<int64_t>(blockIdx.y) * 64, c0 = static_cast<int64_t>(blockIdx.x) * 64;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;  // 64 x 4
  for (int j = ty; j < 64; j += 4)
    if (r0 + j < R && c0 + tx < C) tile[j][tx] = A[(r0 + j) * C + c0 + tx];
  __syncthreads();
  for (int j = ty; j < 64; j += 4)
    if (c0 + j < C && r0 + tx < R) At[(c0 + j) * R + r0 + tx] = tile[tx][j];
}

}  // namespace mi_oov

extern "C" int mi_oov_act_forward(const float* Z, int64_t n, int act, float* Y, void* stream) {
  using namespace mi_oov;
  if (n < 0) return MI_OOV_ERR_SHAPE;
  if (act < 0 || act > 2) return MI_OOV_ERR_KIND;
  if (n == 0) return MI_OOV_OK;
  if (!Z || !Y) return MI_OOV_ERR_NULL;
  if (!aligned16(Z) || !aligned16(Y)) return MI_OOV_ERR_ALIGN;
  hipLaunchKernelGGL(act_forward_kernel, dim3(grid_for(n, 1024)), dim3(256), 0, static_cast<hipStream_t>(stream), Z, n, act, Y);
  return check_launch();
}

extern "C" int mi_oov_act_backward(const float* dY, const float* Z, int64_t n, int act, float* dZ, void* stream) {
  using namespace mi_oov;
  if (n < 0) return MI_OOV_ERR_SHAPE;
  if (act < 0 || act > 2) return MI_OOV_ERR_KIND;
  if (n == 0) return MI_OOV_OK;
  if (!dY || !Z || !dZ) return MI_OOV_ERR_NULL;
  if (!aligned16(dY) || !aligned16(Z) || !aligned16(dZ)) return MI_OOV_ERR_ALIGN;
  hipLaunchKernelGGL(act_backward_kernel, dim3(grid_for(n, 1024)), dim3(256), 0, static_cast<hipStream_t>(stream), dY, Z, n, act, dZ);
  return check_launch();
}

extern "C" int mi_oov_transpose(const float* A, int64_t R, int64_t C, float* At, void* stream) {
  using namespace mi_oov;
  if (R < 0 || C < 0) return MI_OOV_ERR_SHAPE;
  if (R == 0 || C == 0) return MI_OOV_OK;
  if (!A || !At) return MI_OOV_ERR_NULL;
  const int64_t gx = (C + 63) / 64, gy = (R + 63) / 64;
  if (gy > 65535) return MI_OOV_ERR_SHAPE;  // R <= 4 M rows
  hipLaunchKernelGGL(transpose_kernel, dim3(static_cast<unsigned>(gx), static_cast<unsigned>(gy)), dim3(256), 0,
                     static_cast<hipStream_t>(stream), A, R, C, At);
  return check_launch();
}
