// Overlap-add ("fold") and its adjoint ("unfold") of the channels-last transposed convolution.
//
// reference models.py:254-258,277: weight_norm(ConvTranspose1d(c_in, c_out, k, u, padding=(k-u)//2)).
// The upsampler is computed as  P = X . W  (a 1x1 channels-last convolution with k*c_out output
// columns, on the matrix cores: vits_conv1d_cl)  followed by
//     Y[b][to][co] = bias[co] + sum_{j : (to + pad - j) % u == 0} P[b][(to + pad - j)/u][j*c_out + co]
// which touches k/u (= 2 for every VITS upsampler) rows of P per output row.  Pure bandwidth work:
// 16-byte vectors along the channel dimension, one output vector per thread.
#include "common.h"

namespace {

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float to_f(float v) { return v; }
__device__ __forceinline__ float to_f(__bf16 v) { return (float)v; }
template <typename T> __device__ __forceinline__ T from_f(float v);
template <> __device__ __forceinline__ float from_f<float>(float v) { return v; }
template <> __device__ __forceinline__ __bf16 from_f<__bf16>(float v) { return (__bf16)v; }

template <typename T>
__global__ void fold_kernel(const T* __restrict__ P, const float* __restrict__ bias, T* __restrict__ Y,
                            int B, int Tin, int Tout, int C, int k, int u, int pad) {
  constexpr int V = 16 / sizeof(T);
  const int vpr = C / V;
  const size_t n = (size_t)B * Tout * vpr;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const int vc = (int)(i % vpr);
    const int to = (int)((i / vpr) % Tout);
    const int b = (int)(i / ((size_t)vpr * Tout));
    float acc[V];
#pragma unroll
    for (int e = 0; e < V; ++e) acc[e] = bias ? bias[vc * V + e] : 0.f;
    for (int j = (to + pad) % u; j < k; j += u) {
      const int ti = (to + pad - j) / u;
      if (to + pad - j < 0 || ti >= Tin) continue;
      union { u32x4 u4; T e[V]; } v;
      v.u4 = *reinterpret_cast<const u32x4*>(P + ((size_t)b * Tin + ti) * (size_t)k * C + (size_t)j * C + vc * V);
#pragma unroll
      for (int e = 0; e < V; ++e) acc[e] += to_f(v.e[e]);
    }
    union { u32x4 u4; T e[V]; } o;
#pragma unroll
    for (int e = 0; e < V; ++e) o.e[e] = from_f<T>(acc[e]);
    *reinterpret_cast<u32x4*>(Y + ((size_t)b * Tout + to) * C + vc * V) = o.u4;
  }
}

template <typename T>
__global__ void unfold_kernel(const T* __restrict__ dY, T* __restrict__ dP, int B, int Tin, int Tout, int C, int k, int u, int pad) {
  constexpr int V = 16 / sizeof(T);
  const int vpr = C / V;
  const size_t n = (size_t)B * Tin * k * vpr;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const int vc = (int)(i % vpr);
    const int j = (int)((i / vpr) % k);
    const int ti = (int)((i / ((size_t)vpr * k)) % Tin);
    const int b = (int)(i / ((size_t)vpr * k * Tin));
    const int to = ti * u + j - pad;
    u32x4 v = {0u, 0u, 0u, 0u};
    if (to >= 0 && to < Tout) v = *reinterpret_cast<const u32x4*>(dY + ((size_t)b * Tout + to) * C + vc * V);
    *reinterpret_cast<u32x4*>(dP + ((size_t)b * Tin + ti) * (size_t)k * C + (size_t)j * C + vc * V) = v;
  }
}

template <typename T>
int run(bool fold, const void* src, const float* bias, void* dst, int B, int Tin, int C, int k, int u, int pad, hipStream_t s) {
  const int Tout = (Tin - 1) * u - 2 * pad + k;
  const size_t n = fold ? (size_t)B * Tout * (C / (16 / sizeof(T))) : (size_t)B * Tin * k * (C / (16 / sizeof(T)));
  unsigned blocks = (unsigned)((n + 255) / 256);
  if (blocks > 256u * 16u) blocks = 256u * 16u;
  if (fold) hipLaunchKernelGGL(fold_kernel<T>, dim3(blocks), dim3(256), 0, s, static_cast<const T*>(src), bias, static_cast<T*>(dst), B, Tin, Tout, C, k, u, pad);
  else hipLaunchKernelGGL(unfold_kernel<T>, dim3(blocks), dim3(256), 0, s, static_cast<const T*>(src), static_cast<T*>(dst), B, Tin, Tout, C, k, u, pad);
  return vits::check_launch(fold ? "vits_convt_fold_cl" : "vits_convt_unfold_cl");
}

int common(bool fold, int dtype, const void* src, const float* bias, void* dst, int b, int t_in, int c, int k, int u, int pad, void* stream) {
  if (!src || !dst || b <= 0 || t_in <= 0 || c <= 0 || k <= 0 || u <= 0 || pad < 0) return VITS_E_BADARG;
  if ((t_in - 1) * u - 2 * pad + k <= 0) return VITS_E_BADARG;
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (dtype == VITS_DT_BF16) return (c % 8) ? VITS_E_UNSUPPORTED : run<__bf16>(fold, src, bias, dst, b, t_in, c, k, u, pad, s);
  if (dtype == VITS_DT_F32) return (c % 4) ? VITS_E_UNSUPPORTED : run<float>(fold, src, bias, dst, b, t_in, c, k, u, pad, s);
  return VITS_E_UNSUPPORTED;
}

}  // namespace

extern "C" int vits_convt_fold_cl(int dtype, const void* p, const float* bias, void* y, int b, int t_in, int c_out, int k,
                                  int u, int pad, void* stream) {
  return common(true, dtype, p, bias, y, b, t_in, c_out, k, u, pad, stream);
}

extern "C" int vits_convt_unfold_cl(int dtype, const void* dy, void* dp, int b, int t_in, int c_out, int k, int u, int pad,
                                    void* stream) {
  return common(false, dtype, dy, nullptr, dp, b, t_in, c_out, k, u, pad, stream);
}
