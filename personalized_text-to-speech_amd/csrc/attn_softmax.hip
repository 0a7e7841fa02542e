// Row kernels of the windowed relative-position attention (reference attentions.py:150-182):
// everything between the QK^T product and the PV product, and its backward.
//
//   forward : logit[i][j] = scale * ( S[i][j] + (|j-i| <= w ? R[i][j-i+w] : 0) ),   S = Q K^T, R = Q E_k^T
//             masked_fill(query or key beyond the item's length, -1e4); P = softmax_j(logit);
//             Pd = P * keep (attention dropout, keep = 0 or 1/(1-p), optional);
//             Pband[i][m] = Pd[i][i+m-w]   (operand of the relative-value product, attentions.py:173-176)
//   backward: dP' = (dPd + band(dPband)) * keep;  dS' = P * (dP' - <P, dP'>);  masked logits get no gradient;
//             dS = scale * dS';  dSband[i][m] = dS[i][i+m-w]
// The matrix products themselves run on vits_conv1d_cl (k = 1, per-item operands) — see attention_cl.py.
// One wave per query row; keys on the lanes (up to 16 per lane => t <= 1024).
#include "common.h"

namespace {

constexpr int NCMAX = 16;
constexpr int BAND = 16;              // columns of the band tensors (2w+1 used)

__device__ __forceinline__ float to_f(float v) { return v; }
__device__ __forceinline__ float to_f(__bf16 v) { return (float)v; }
template <typename T> __device__ __forceinline__ T from_f(float v);
template <> __device__ __forceinline__ float from_f<float>(float v) { return v; }
template <> __device__ __forceinline__ __bf16 from_f<__bf16>(float v) { return (__bf16)v; }

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
  return v;
}

template <typename T>
__global__ __launch_bounds__(256) void relsoftmax_fwd(const T* __restrict__ S, const T* __restrict__ R, const T* __restrict__ keep,
                                                      const int* __restrict__ lengths, T* __restrict__ P, T* __restrict__ Pd,
                                                      T* __restrict__ Pband, int rows, int Tn, int ld, int w, float scale) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int b = row / Tn, i = row % Tn;
  const int len = lengths ? lengths[b] : Tn;
  const bool qvalid = i < len;
  float v[NCMAX], m = -3.0e38f;
#pragma unroll
  for (int c = 0; c < NCMAX; ++c) {
    const int j = lane + 64 * c;
    float x = -3.0e38f;
    if (j < Tn) {
      if (qvalid && j < len) {
        x = to_f(S[(size_t)row * ld + j]);
        const int rel = j - i + w;
        if (R && rel >= 0 && rel <= 2 * w) x += to_f(R[(size_t)row * BAND + rel]);
        x *= scale;
      } else {
        x = -1e4f;
      }
    }
    v[c] = x;
    m = fmaxf(m, x);
  }
  m = wave_max(m);
  float sum = 0.f;
#pragma unroll
  for (int c = 0; c < NCMAX; ++c) {
    const int j = lane + 64 * c;
    v[c] = (j < Tn) ? expf(v[c] - m) : 0.f;
    sum += v[c];
  }
  const float inv = 1.f / wave_sum(sum);
#pragma unroll
  for (int c = 0; c < NCMAX; ++c) {
    const int j = lane + 64 * c;
    if (j < ld) {
      const float p = (j < Tn) ? v[c] * inv : 0.f;
      P[(size_t)row * ld + j] = from_f<T>(p);
      float pd = p;
      if (keep) pd *= to_f(keep[(size_t)row * ld + j]);
      if (Pd) Pd[(size_t)row * ld + j] = from_f<T>(pd);
      const int rel = j - i + w;
      if (Pband && j < Tn && rel >= 0 && rel <= 2 * w) Pband[(size_t)row * BAND + rel] = from_f<T>(pd);
    }
  }
  // band columns that fall outside [0, t) and the unused columns 2w+1..15
  if (Pband && lane < BAND) {
    const int j = i + lane - w;
    if (lane > 2 * w || j < 0 || j >= Tn) Pband[(size_t)row * BAND + lane] = from_f<T>(0.f);
  }
}

template <typename T>
__global__ __launch_bounds__(256) void relsoftmax_bwd(const T* __restrict__ P, const T* __restrict__ dPd, const T* __restrict__ dPband,
                                                      const T* __restrict__ keep, const int* __restrict__ lengths, T* __restrict__ dS,
                                                      T* __restrict__ dSband, int rows, int Tn, int ld, int w, float scale) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int b = row / Tn, i = row % Tn;
  const int len = lengths ? lengths[b] : Tn;
  const bool qvalid = i < len;
  float p[NCMAX], g[NCMAX], dot = 0.f;
#pragma unroll
  for (int c = 0; c < NCMAX; ++c) {
    const int j = lane + 64 * c;
    p[c] = 0.f; g[c] = 0.f;
    if (j < Tn) {
      p[c] = to_f(P[(size_t)row * ld + j]);
      float d = to_f(dPd[(size_t)row * ld + j]);
      const int rel = j - i + w;
      if (dPband && rel >= 0 && rel <= 2 * w) d += to_f(dPband[(size_t)row * BAND + rel]);
      if (keep) d *= to_f(keep[(size_t)row * ld + j]);
      g[c] = d;
      dot += p[c] * d;
    }
  }
  dot = wave_sum(dot);
#pragma unroll
  for (int c = 0; c < NCMAX; ++c) {
    const int j = lane + 64 * c;
    if (j < ld) {
      float d = 0.f;
      if (j < Tn && qvalid && j < len) d = scale * p[c] * (g[c] - dot);
      dS[(size_t)row * ld + j] = from_f<T>(d);
      const int rel = j - i + w;
      if (dSband && j < Tn && rel >= 0 && rel <= 2 * w) dSband[(size_t)row * BAND + rel] = from_f<T>(d);
    }
  }
  if (dSband && lane < BAND) {
    const int j = i + lane - w;
    if (lane > 2 * w || j < 0 || j >= Tn) dSband[(size_t)row * BAND + lane] = from_f<T>(0.f);
  }
}

}  // namespace

extern "C" int vits_relsoftmax(int dtype, const void* s, const void* r, const void* keep, const int32_t* lengths, void* p, void* pd,
                               void* pband, int b, int t, int ld, int window, float scale, void* stream) {
  if (!s || !p || b <= 0 || t <= 0 || ld < t || window < 0 || 2 * window + 1 > 16) return VITS_E_BADARG;
  if (t > 64 * 16) return VITS_E_UNSUPPORTED;
  const int rows = b * t;
  hipStream_t st = static_cast<hipStream_t>(stream);
  dim3 grid((rows + 3) / 4), block(256);
  if (dtype == VITS_DT_BF16)
    hipLaunchKernelGGL(relsoftmax_fwd<__bf16>, grid, block, 0, st, (const __bf16*)s, (const __bf16*)r, (const __bf16*)keep, lengths, (__bf16*)p, (__bf16*)pd, (__bf16*)pband, rows, t, ld, window, scale);
  else if (dtype == VITS_DT_F32)
    hipLaunchKernelGGL(relsoftmax_fwd<float>, grid, block, 0, st, (const float*)s, (const float*)r, (const float*)keep, lengths, (float*)p, (float*)pd, (float*)pband, rows, t, ld, window, scale);
  else return VITS_E_UNSUPPORTED;
  return vits::check_launch("vits_relsoftmax");
}

extern "C" int vits_relsoftmax_bwd(int dtype, const void* p, const void* dpd, const void* dpband, const void* keep,
                                   const int32_t* lengths, void* ds, void* dsband, int b, int t, int ld, int window, float scale,
                                   void* stream) {
  if (!p || !dpd || !ds || b <= 0 || t <= 0 || ld < t || window < 0 || 2 * window + 1 > 16) return VITS_E_BADARG;
  if (t > 64 * 16) return VITS_E_UNSUPPORTED;
  const int rows = b * t;
  hipStream_t st = static_cast<hipStream_t>(stream);
  dim3 grid((rows + 3) / 4), block(256);
  if (dtype == VITS_DT_BF16)
    hipLaunchKernelGGL(relsoftmax_bwd<__bf16>, grid, block, 0, st, (const __bf16*)p, (const __bf16*)dpd, (const __bf16*)dpband, (const __bf16*)keep, lengths, (__bf16*)ds, (__bf16*)dsband, rows, t, ld, window, scale);
  else if (dtype == VITS_DT_F32)
    hipLaunchKernelGGL(relsoftmax_bwd<float>, grid, block, 0, st, (const float*)p, (const float*)dpd, (const float*)dpband, (const float*)keep, lengths, (float*)ds, (float*)dsband, rows, t, ld, window, scale);
  else return VITS_E_UNSUPPORTED;
  return vits::check_launch("vits_relsoftmax_bwd");
}
