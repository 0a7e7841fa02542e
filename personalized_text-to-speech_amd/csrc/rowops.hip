// Row-wise channels-last kernels of the duration predictor / text encoder:
//   vits_ln_act_cl      y = [res +] act( LayerNorm_c(x) * gamma + beta )          (fwd, bwd)
//   vits_dwconv_cl      depth-wise dilated convolution over time, masked input    (fwd, bwd)
//
// Replaces (reference): modules.LayerNorm (modules.py:20-32: F.layer_norm over the channel dim after
// two transposes), F.gelu, the `x = x + y` residual and the depth-separable conv of modules.DDSConv
// (modules.py:95-108), and the post-norm residual LayerNorms of attentions.Encoder (attentions.py:41-46).
//
// Tensors are [rows][C] with C <= 1024; one workgroup walks a chunk of rows with ONE THREAD PER
// CHANNEL (coalesced row reads; the per-row mean/variance is a block reduction, the per-channel
// parameter gradients are plain per-thread sums over the chunk, written as per-workgroup partials
// and summed by a second tiny kernel in a fixed order => reproducible, no float atomics).
#include "common.h"

namespace {

__device__ __forceinline__ float to_f(float v) { return v; }
__device__ __forceinline__ float to_f(__bf16 v) { return (float)v; }
template <typename T> __device__ __forceinline__ T from_f(float v);
template <> __device__ __forceinline__ float from_f<float>(float v) { return v; }
template <> __device__ __forceinline__ __bf16 from_f<__bf16>(float v) { return (__bf16)v; }

constexpr float kInvSqrt2 = 0.70710678118654752440f;
constexpr float kInvSqrt2Pi = 0.39894228040143267794f;
__device__ __forceinline__ float gelu_f(float u) { return 0.5f * u * (1.0f + erff(u * kInvSqrt2)); }
__device__ __forceinline__ float gelu_grad(float u) {
  return 0.5f * (1.0f + erff(u * kInvSqrt2)) + u * kInvSqrt2Pi * __expf(-0.5f * u * u);
}

// sum over the workgroup (blockDim.x multiple of 64, <= 1024); every thread gets the result
__device__ __forceinline__ float block_sum(float v, float* red) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
  __syncthreads();
  if (lane == 0) red[wave] = v;
  __syncthreads();
  float s = 0.f;
  for (int w = 0; w < nw; ++w) s += red[w];
  return s;
}

// One WAVE per row (lane l owns channels l, l+64, ...: up to 16 per lane), 4 rows per workgroup step: the
// row statistics are wave shuffles, no workgroup barrier inside the row loop.
constexpr int kNcMax = 16;      // up to 1024 channels; the kernels are instantiated for 4 (<= 256 channels) and 16 column slots per lane

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

// 4 adjacent channels in one access (8 bytes of bf16 / 16 bytes of float); p must be aligned to that size
template <typename T>
__device__ __forceinline__ void load4(const T* p, bool ok, float (&o)[4]) {
  if (!ok) { o[0] = o[1] = o[2] = o[3] = 0.f; return; }
  if constexpr (sizeof(T) == 2) {
    union { uint2 u; T e[4]; } v;
    v.u = *reinterpret_cast<const uint2*>(p);
#pragma unroll
    for (int i = 0; i < 4; ++i) o[i] = to_f(v.e[i]);
  } else {
    const float4 f = *reinterpret_cast<const float4*>(p);
    o[0] = f.x; o[1] = f.y; o[2] = f.z; o[3] = f.w;
  }
}
template <typename T>
__device__ __forceinline__ void store4(T* p, const float (&o)[4]) {
  if constexpr (sizeof(T) == 2) {
    union { uint2 u; T e[4]; } v;
#pragma unroll
    for (int i = 0; i < 4; ++i) v.e[i] = from_f<T>(o[i]);
    *reinterpret_cast<uint2*>(p) = v.u;
  } else {
    *reinterpret_cast<float4*>(p) = make_float4(o[0], o[1], o[2], o[3]);
  }
}

template <typename T, int NCMAX>
__global__ __launch_bounds__(256) void ln_act_fwd(const T* __restrict__ x, const float* __restrict__ gamma, const float* __restrict__ beta,
                           const T* __restrict__ res, T* __restrict__ y, int rows, int C, float eps, int act, int rows_per_wg) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int nc = (C + 63) >> 6;
  float g[NCMAX], bt[NCMAX];
#pragma unroll
  for (int i = 0; i < NCMAX; ++i) {
    const int c = (NCMAX == 4) ? lane * 4 + i : lane + 64 * i;   // <= 256 channels: 4 adjacent channels per lane (one 8-byte access)
    g[i] = ((NCMAX == 4 || i < nc) && c < C) ? gamma[c] : 0.f;
    bt[i] = ((NCMAX == 4 || i < nc) && c < C) ? beta[c] : 0.f;
  }
  const int r0 = blockIdx.x * rows_per_wg;
  for (int r = r0 + wave; r < r0 + rows_per_wg && r < rows; r += 4) {
    float v[NCMAX], sum = 0.f;
    if constexpr (NCMAX == 4) {
      float q[4];
      load4<T>(x + (size_t)r * C + lane * 4, lane * 4 < C, q);
#pragma unroll
      for (int i = 0; i < 4; ++i) { v[i] = q[i]; sum += q[i]; }
    } else {
#pragma unroll
      for (int i = 0; i < NCMAX; ++i) {
        const int c = lane + 64 * i;
        v[i] = (i < nc && c < C) ? to_f(x[(size_t)r * C + c]) : 0.f;
        sum += v[i];
      }
    }
    const float mean = wave_sum(sum) / C;
    float sq = 0.f;
#pragma unroll
    for (int i = 0; i < NCMAX; ++i) {
      const int c = (NCMAX == 4) ? lane * 4 + i : lane + 64 * i;   // <= 256 channels: 4 adjacent channels per lane (one 8-byte access)
      v[i] = ((NCMAX == 4 || i < nc) && c < C) ? v[i] - mean : 0.f;
      sq += v[i] * v[i];
    }
    const float rstd = rsqrtf(wave_sum(sq) / C + eps);
    if constexpr (NCMAX == 4) {
      if (lane * 4 < C) {
        float rq[4] = {0.f, 0.f, 0.f, 0.f}, u[4];
        if (res) load4<T>(res + (size_t)r * C + lane * 4, true, rq);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          float w_ = v[i] * rstd * g[i] + bt[i];
          if (act == 1) w_ = gelu_f(w_);
          u[i] = w_ + rq[i];
        }
        store4<T>(y + (size_t)r * C + lane * 4, u);
      }
    } else {
#pragma unroll
      for (int i = 0; i < NCMAX; ++i) {
        const int c = lane + 64 * i;
        if (i < nc && c < C) {
          float u = v[i] * rstd * g[i] + bt[i];
          if (act == 1) u = gelu_f(u);
          if (res) u += to_f(res[(size_t)r * C + c]);
          y[(size_t)r * C + c] = from_f<T>(u);
        }
      }
    }
  }
}

template <typename T, int NCMAX>
__global__ __launch_bounds__(256) void ln_act_bwd(const T* __restrict__ x, const float* __restrict__ gamma, const float* __restrict__ beta,
                           const T* __restrict__ dy, T* __restrict__ dx, float* __restrict__ part, int rows, int C, float eps,
                           int act, int rows_per_wg) {
  extern __shared__ float red_s[];                 // [4 waves][2][C] cross-wave sums of dgamma / dbeta
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int nc = (C + 63) >> 6;
  float g[NCMAX], bt[NCMAX], dg[NCMAX], db[NCMAX];
#pragma unroll
  for (int i = 0; i < NCMAX; ++i) {
    const int c = (NCMAX == 4) ? lane * 4 + i : lane + 64 * i;   // <= 256 channels: 4 adjacent channels per lane (one 8-byte access)
    g[i] = ((NCMAX == 4 || i < nc) && c < C) ? gamma[c] : 0.f;
    bt[i] = ((NCMAX == 4 || i < nc) && c < C) ? beta[c] : 0.f;
    dg[i] = 0.f; db[i] = 0.f;
  }
  const int r0 = blockIdx.x * rows_per_wg;
  const int r_end = (r0 + rows_per_wg < rows) ? r0 + rows_per_wg : rows;
  // software pipeline over this wave's rows: the loads of row r + 4 are in flight while row r is reduced (the row chain is
  // otherwise one exposed load latency per row)
  float vn[NCMAX], an[NCMAX];
  auto load_row = [&](int rr) {
    if constexpr (NCMAX == 4) {
      const bool ok = lane * 4 < C && rr < r_end;
      float q[4], p[4];
      load4<T>(x + (size_t)rr * C + lane * 4, ok, q);
      load4<T>(dy + (size_t)rr * C + lane * 4, ok, p);
#pragma unroll
      for (int i = 0; i < 4; ++i) { vn[i] = q[i]; an[i] = p[i]; }
    } else {
#pragma unroll
      for (int i = 0; i < NCMAX; ++i) {
        const int c = lane + 64 * i;
        const bool ok = i < nc && c < C && rr < r_end;
        vn[i] = ok ? to_f(x[(size_t)rr * C + c]) : 0.f;
        an[i] = ok ? to_f(dy[(size_t)rr * C + c]) : 0.f;
      }
    }
  };
  load_row(r0 + wave);
  for (int r = r0 + wave; r < r_end; r += 4) {
    float v[NCMAX], a[NCMAX], sum = 0.f;
#pragma unroll
    for (int i = 0; i < NCMAX; ++i) {
      v[i] = vn[i];
      a[i] = an[i];
      sum += v[i];
    }
    load_row(r + 4);
    const float mean = wave_sum(sum) / C;
    float sq = 0.f;
#pragma unroll
    for (int i = 0; i < NCMAX; ++i) {
      const int c = (NCMAX == 4) ? lane * 4 + i : lane + 64 * i;   // <= 256 channels: 4 adjacent channels per lane (one 8-byte access)
      v[i] = ((NCMAX == 4 || i < nc) && c < C) ? v[i] - mean : 0.f;
      sq += v[i] * v[i];
    }
    const float rstd = rsqrtf(wave_sum(sq) / C + eps);
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int i = 0; i < NCMAX; ++i) {
      const float xh = v[i] * rstd;
      float du = a[i];
      if (act == 1) du *= gelu_grad(xh * g[i] + bt[i]);
      dg[i] += du * xh;
      db[i] += du;
      v[i] = xh;
      a[i] = du * g[i];
      s1 += a[i];
      s2 += a[i] * xh;
    }
    const float m1 = wave_sum(s1) / C, m2 = wave_sum(s2) / C;
    if constexpr (NCMAX == 4) {
      if (lane * 4 < C) {
        float u[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) u[i] = rstd * (a[i] - m1 - v[i] * m2);
        store4<T>(dx + (size_t)r * C + lane * 4, u);
      }
    } else {
#pragma unroll
      for (int i = 0; i < NCMAX; ++i) {
        const int c = lane + 64 * i;
        if (i < nc && c < C) dx[(size_t)r * C + c] = from_f<T>(rstd * (a[i] - m1 - v[i] * m2));
      }
    }
  }
  // combine the 4 waves' per-channel sums in a fixed order, one partial row per workgroup
#pragma unroll
  for (int i = 0; i < NCMAX; ++i) {
    const int c = (NCMAX == 4) ? lane * 4 + i : lane + 64 * i;   // <= 256 channels: 4 adjacent channels per lane (one 8-byte access)
    if ((NCMAX == 4 || i < nc) && c < C) { red_s[(wave * 2 + 0) * C + c] = dg[i]; red_s[(wave * 2 + 1) * C + c] = db[i]; }
  }
  __syncthreads();
  for (int j = threadIdx.x; j < 2 * C; j += 256) {
    const int which = j / C, c = j % C;
    part[(size_t)blockIdx.x * 2 * C + j] = red_s[(0 * 2 + which) * C + c] + red_s[(1 * 2 + which) * C + c] +
                                           red_s[(2 * 2 + which) * C + c] + red_s[(3 * 2 + which) * C + c];
  }
}

// out[j] (+)= sum_w part[w * stride + j], j < n.  1024 threads = 64 columns x 16 slices of w; the slice sums
// are combined in a fixed order => reproducible.
__global__ __launch_bounds__(1024) void reduce_partials(const float* __restrict__ part, float* __restrict__ out, int n, int W,
                                                        int stride, int accumulate) {
  __shared__ float sm[16][64];
  const int col = threadIdx.x & 63, slice = threadIdx.x >> 6;
  const int j = blockIdx.x * 64 + col;
  float s = 0.f;
  if (j < n)
    for (int w = slice; w < W; w += 16) s += part[(size_t)w * stride + j];
  sm[slice][col] = s;
  __syncthreads();
  if (slice == 0 && j < n) {
    float t = accumulate ? out[j] : 0.f;
#pragma unroll
    for (int k = 0; k < 16; ++k) t += sm[k][col];
    out[j] = t;
  }
}

// Two outputs in one launch: partial rows are [nA | nB] wide; blocks [0, ceil(nA/64)) reduce into outA, the others into outB.
__global__ __launch_bounds__(1024) void reduce_partials2(const float* __restrict__ part, float* __restrict__ outA, int nA,
                                                         float* __restrict__ outB, int nB, int W, int stride, int accumulate) {
  __shared__ float sm[16][64];
  const int blocksA = (nA + 63) / 64;
  const bool isB = (int)blockIdx.x >= blocksA;
  const int col = threadIdx.x & 63, slice = threadIdx.x >> 6;
  const int j = (isB ? (int)blockIdx.x - blocksA : (int)blockIdx.x) * 64 + col;
  const int n = isB ? nB : nA;
  const float* src = part + (isB ? nA : 0);
  float* out = isB ? outB : outA;
  float s = 0.f;
  if (j < n) {
#pragma unroll 4
    for (int w = slice; w < W; w += 16) s += src[(size_t)w * stride + j];
  }
  sm[slice][col] = s;
  __syncthreads();
  if (slice == 0 && j < n) {
    float t = accumulate ? out[j] : 0.f;
#pragma unroll
    for (int k = 0; k < 16; ++k) t += sm[k][col];
    out[j] = t;
  }
}

// ---------------------------------------------------------------- depth-wise conv
template <typename T, int KC>          // KC = compile-time tap count (3 for the VITS duration predictor), 0 = run-time k
__global__ void dwconv_fwd(const T* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias,
                           const int* __restrict__ lengths, T* __restrict__ y, int B, int Tn, int C, int k, int dil,
                           int rows_per_wg) {
  const int c = threadIdx.x;
  if (c >= C) return;
  if (KC > 0) k = KC;
  float wk[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) wk[j] = j < k ? w[c * k + j] : 0.f;
  const float bs = bias ? bias[c] : 0.f;
  const int half = (k - 1) / 2;
  const int r0 = blockIdx.x * rows_per_wg;
#pragma unroll 2
  for (int r = r0; r < r0 + rows_per_wg && r < B * Tn; ++r) {
    const int b = r / Tn, t = r % Tn;
    const int len = lengths ? lengths[b] : Tn;
    float acc = bs;
#pragma unroll
    for (int j = 0; j < (KC > 0 ? KC : 8); ++j) {
      const int ti = t + (j - half) * dil;
      if (j < k && ti >= 0 && ti < Tn && ti < len) acc += wk[j] * to_f(x[((size_t)b * Tn + ti) * C + c]);
    }
    y[(size_t)r * C + c] = from_f<T>(acc);
  }
}

template <typename T, int KC>          // KC = compile-time tap count (3 for the VITS duration predictor), 0 = run-time k
__global__ void dwconv_bwd(const T* __restrict__ x, const float* __restrict__ w, const int* __restrict__ lengths,
                           const T* __restrict__ dy, T* __restrict__ dx, float* __restrict__ part, int B, int Tn, int C, int k,
                           int dil, int rows_per_wg) {
  const int c = threadIdx.x;
  if (c >= C) return;
  if (KC > 0) k = KC;
  float wk[8], dw[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) { wk[j] = j < k ? w[c * k + j] : 0.f; dw[j] = 0.f; }
  float db = 0.f;
  const int half = (k - 1) / 2;
  const int r0 = blockIdx.x * rows_per_wg;
#pragma unroll 2
  for (int r = r0; r < r0 + rows_per_wg && r < B * Tn; ++r) {
    const int b = r / Tn, t = r % Tn;
    const int len = lengths ? lengths[b] : Tn;
    // dx[t] = mask[t] * sum_j w[j] * dy[t - (j-half)*dil]
    float acc = 0.f;
    if (t < len) {
#pragma unroll
      for (int j = 0; j < (KC > 0 ? KC : 8); ++j) {
        const int to = t - (j - half) * dil;
        if (j < k && to >= 0 && to < Tn) acc += wk[j] * to_f(dy[((size_t)b * Tn + to) * C + c]);
      }
    }
    dx[(size_t)r * C + c] = from_f<T>(acc);
    // dw[j] += dy[t] * xm[t + (j-half)*dil]
    const float g = to_f(dy[(size_t)r * C + c]);
    db += g;
#pragma unroll
    for (int j = 0; j < (KC > 0 ? KC : 8); ++j) {
      const int ti = t + (j - half) * dil;
      if (j < k && ti >= 0 && ti < Tn && ti < len) dw[j] += g * to_f(x[((size_t)b * Tn + ti) * C + c]);
    }
  }
  float* P = part + (size_t)blockIdx.x * (k + 1) * C;
#pragma unroll
  for (int j = 0; j < 8; ++j) if (j < k) P[c * k + j] = dw[j];
  P[k * C + c] = db;
}

// ---- 8-channel vector form of the depth-wise convolution (k = 3, C % 8 == 0): a thread owns 8 adjacent channels (16-byte
// accesses for bf16) and one of RL = 256 / (C/8) row lanes; the backward sums its per-lane weight / bias partials over the row
// lanes through LDS in a fixed order before writing the workgroup's partial row.
template <typename T>
__device__ __forceinline__ void load8(const T* p, bool ok, float (&o)[8]) {
  if (!ok) {
#pragma unroll
    for (int i = 0; i < 8; ++i) o[i] = 0.f;
    return;
  }
  if constexpr (sizeof(T) == 2) {
    union { uint4 u; T e[8]; } v;
    v.u = *reinterpret_cast<const uint4*>(p);
#pragma unroll
    for (int i = 0; i < 8; ++i) o[i] = to_f(v.e[i]);
  } else {
    const float4 a = *reinterpret_cast<const float4*>(p), b = *reinterpret_cast<const float4*>(p + 4);
    o[0] = a.x; o[1] = a.y; o[2] = a.z; o[3] = a.w; o[4] = b.x; o[5] = b.y; o[6] = b.z; o[7] = b.w;
  }
}
template <typename T>
__device__ __forceinline__ void store8(T* p, const float (&o)[8]) {
  if constexpr (sizeof(T) == 2) {
    union { uint4 u; T e[8]; } v;
#pragma unroll
    for (int i = 0; i < 8; ++i) v.e[i] = from_f<T>(o[i]);
    *reinterpret_cast<uint4*>(p) = v.u;
  } else {
    *reinterpret_cast<float4*>(p) = make_float4(o[0], o[1], o[2], o[3]);
    *reinterpret_cast<float4*>(p + 4) = make_float4(o[4], o[5], o[6], o[7]);
  }
}

template <typename T>
__global__ __launch_bounds__(256) void dwconv3_fwd_v8(const T* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias,
                                                      const int* __restrict__ lengths, T* __restrict__ y, int B, int Tn, int C, int dil,
                                                      int rows_per_wg) {
  const int cg = C >> 3, RL = 256 / cg;
  const int g = threadIdx.x % cg, rl = threadIdx.x / cg;
  if (rl >= RL) return;
  const int c0 = g * 8;
  float wk[8][3], bs[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
#pragma unroll
    for (int j = 0; j < 3; ++j) wk[i][j] = w[(c0 + i) * 3 + j];
    bs[i] = bias ? bias[c0 + i] : 0.f;
  }
  const int r0 = blockIdx.x * rows_per_wg;
  const int r_end = (r0 + rows_per_wg < B * Tn) ? r0 + rows_per_wg : B * Tn;
  for (int r = r0 + rl; r < r_end; r += RL) {
    const int b = r / Tn, t = r - b * Tn;
    const int len = lengths ? lengths[b] : Tn;
    float acc[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = bs[i];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      const int ti = t + (j - 1) * dil;
      float v[8];
      load8<T>(x + ((size_t)b * Tn + ti) * C + c0, ti >= 0 && ti < Tn && ti < len, v);
#pragma unroll
      for (int i = 0; i < 8; ++i) acc[i] += wk[i][j] * v[i];
    }
    store8<T>(y + (size_t)r * C + c0, acc);
  }
}

template <typename T>
__global__ __launch_bounds__(256) void dwconv3_bwd_v8(const T* __restrict__ x, const float* __restrict__ w, const int* __restrict__ lengths,
                                                      const T* __restrict__ dy, T* __restrict__ dx, float* __restrict__ part, int B, int Tn,
                                                      int C, int dil, int rows_per_wg) {
  extern __shared__ float red_v8[];                 // [256 threads][32]: per-thread dw[8][3] + db[8]
  const int cg = C >> 3, RL = 256 / cg;
  const int g = threadIdx.x % cg, rl = threadIdx.x / cg;
  const bool active = rl < RL;
  const int c0 = g * 8;
  float wk[8][3], dw[8][3], db[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
#pragma unroll
    for (int j = 0; j < 3; ++j) { wk[i][j] = active ? w[(c0 + i) * 3 + j] : 0.f; dw[i][j] = 0.f; }
    db[i] = 0.f;
  }
  const int r0 = blockIdx.x * rows_per_wg;
  const int r_end = (r0 + rows_per_wg < B * Tn) ? r0 + rows_per_wg : B * Tn;
  if (active) {
    for (int r = r0 + rl; r < r_end; r += RL) {
      const int b = r / Tn, t = r - b * Tn;
      const int len = lengths ? lengths[b] : Tn;
      float acc[8], gq[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) acc[i] = 0.f;
      load8<T>(dy + (size_t)r * C + c0, true, gq);
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        // dx[t] = mask[t] * sum_j w[j] * dy[t - (j-1)*dil];   dw[j] += dy[t] * xm[t + (j-1)*dil]
        const int to = t - (j - 1) * dil, ti = t + (j - 1) * dil;
        float v[8], xv[8];
        load8<T>(dy + ((size_t)b * Tn + to) * C + c0, t < len && to >= 0 && to < Tn, v);
        load8<T>(x + ((size_t)b * Tn + ti) * C + c0, ti >= 0 && ti < Tn && ti < len, xv);
#pragma unroll
        for (int i = 0; i < 8; ++i) { acc[i] += wk[i][j] * v[i]; dw[i][j] += gq[i] * xv[i]; }
      }
#pragma unroll
      for (int i = 0; i < 8; ++i) db[i] += gq[i];
      store8<T>(dx + (size_t)r * C + c0, acc);
    }
  }
  // sum over the row lanes in lane order (fixed => reproducible), then one partial row per workgroup: [c][3] then [c]
  float* mine = red_v8 + (size_t)threadIdx.x * 32;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
#pragma unroll
    for (int j = 0; j < 3; ++j) mine[i * 3 + j] = dw[i][j];
    mine[24 + i] = db[i];
  }
  __syncthreads();
  float* P = part + (size_t)blockIdx.x * 4 * C;
  for (int idx = threadIdx.x; idx < cg * 32; idx += 256) {
    const int gg = idx / 32, e = idx - gg * 32;
    float sum = 0.f;
    for (int l = 0; l < RL; ++l) sum += red_v8[((size_t)(l * cg + gg)) * 32 + e];
    if (e < 24) P[(gg * 8 + e / 3) * 3 + (e % 3)] = sum;
    else P[3 * C + gg * 8 + (e - 24)] = sum;
  }
}

// forward kernels: up to 512 workgroups; backward kernels write one partial row per workgroup, so
// fewer (<= 96) keeps the second-stage sum short
int pick_rows_per_wg(int rows, int max_wgs = 512) {
  int wgs = rows < max_wgs ? rows : max_wgs;
  if (wgs < 1) wgs = 1;
  return (rows + wgs - 1) / wgs;
}
constexpr int kBwdWgs = 384;

}  // namespace

extern "C" size_t vits_rowops_workspace(int rows, int c, int k) {
  const int rpw = pick_rows_per_wg(rows, kBwdWgs);
  const int wgs = (rows + rpw - 1) / rpw;
  const int per = (k + 1) > 2 ? (k + 1) : 2;
  return (size_t)wgs * per * c * sizeof(float);
}

extern "C" int vits_ln_act_cl(int dtype, const void* x, const float* gamma, const float* beta, const void* res, void* y,
                              int rows, int c, float eps, int act, void* stream) {
  if (!x || !gamma || !beta || !y || rows <= 0 || c <= 0) return VITS_E_BADARG;
  if (c > 1024 || act < 0 || act > 1) return VITS_E_UNSUPPORTED;
  const int threads = 256, rpw = pick_rows_per_wg(rows, 1024), wgs = (rows + rpw - 1) / rpw;
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (dtype == VITS_DT_BF16)
    if (c <= 256 && c % 4 == 0) hipLaunchKernelGGL((ln_act_fwd<__bf16, 4>), dim3(wgs), dim3(threads), 0, s, (const __bf16*)x, gamma, beta, (const __bf16*)res, (__bf16*)y, rows, c, eps, act, rpw);
    else hipLaunchKernelGGL((ln_act_fwd<__bf16, kNcMax>), dim3(wgs), dim3(threads), 0, s, (const __bf16*)x, gamma, beta, (const __bf16*)res, (__bf16*)y, rows, c, eps, act, rpw);
  else if (dtype == VITS_DT_F32)
    if (c <= 256 && c % 4 == 0) hipLaunchKernelGGL((ln_act_fwd<float, 4>), dim3(wgs), dim3(threads), 0, s, (const float*)x, gamma, beta, (const float*)res, (float*)y, rows, c, eps, act, rpw);
    else hipLaunchKernelGGL((ln_act_fwd<float, kNcMax>), dim3(wgs), dim3(threads), 0, s, (const float*)x, gamma, beta, (const float*)res, (float*)y, rows, c, eps, act, rpw);
  else return VITS_E_UNSUPPORTED;
  return vits::check_launch("vits_ln_act_cl");
}

extern "C" int vits_ln_act_cl_bwd(int dtype, const void* x, const float* gamma, const float* beta, const void* dy, void* dx,
                                  float* dgamma, float* dbeta, void* workspace, size_t workspace_bytes, int rows, int c,
                                  float eps, int act, int accumulate, void* stream) {
  if (!x || !gamma || !beta || !dy || !dx || !dgamma || !dbeta || !workspace || rows <= 0 || c <= 0) return VITS_E_BADARG;
  if (c > 1024 || act < 0 || act > 1) return VITS_E_UNSUPPORTED;
  if (workspace_bytes < vits_rowops_workspace(rows, c, 1)) return VITS_E_BADARG;
  const int threads = 256, rpw = pick_rows_per_wg(rows, kBwdWgs), wgs = (rows + rpw - 1) / rpw;
  const size_t lds = (size_t)8 * c * sizeof(float);
  hipStream_t s = static_cast<hipStream_t>(stream);
  float* part = static_cast<float*>(workspace);
  if (dtype == VITS_DT_BF16)
    if (c <= 256 && c % 4 == 0) hipLaunchKernelGGL((ln_act_bwd<__bf16, 4>), dim3(wgs), dim3(threads), lds, s, (const __bf16*)x, gamma, beta, (const __bf16*)dy, (__bf16*)dx, part, rows, c, eps, act, rpw);
    else hipLaunchKernelGGL((ln_act_bwd<__bf16, kNcMax>), dim3(wgs), dim3(threads), lds, s, (const __bf16*)x, gamma, beta, (const __bf16*)dy, (__bf16*)dx, part, rows, c, eps, act, rpw);
  else if (dtype == VITS_DT_F32)
    if (c <= 256 && c % 4 == 0) hipLaunchKernelGGL((ln_act_bwd<float, 4>), dim3(wgs), dim3(threads), lds, s, (const float*)x, gamma, beta, (const float*)dy, (float*)dx, part, rows, c, eps, act, rpw);
    else hipLaunchKernelGGL((ln_act_bwd<float, kNcMax>), dim3(wgs), dim3(threads), lds, s, (const float*)x, gamma, beta, (const float*)dy, (float*)dx, part, rows, c, eps, act, rpw);
  else return VITS_E_UNSUPPORTED;
  // partials are [wg][2][c]: dgamma then dbeta
  hipLaunchKernelGGL(reduce_partials2, dim3(2 * ((c + 63) / 64)), dim3(1024), 0, s, part, dgamma, c, dbeta, c, wgs, 2 * c, accumulate);
  return vits::check_launch("vits_ln_act_cl_bwd");
}

extern "C" int vits_dwconv_cl(int dtype, const void* x, const float* w, const float* bias, const int32_t* lengths, void* y,
                              int b, int t, int c, int k, int dil, void* stream) {
  if (!x || !w || !y || b <= 0 || t <= 0 || c <= 0 || k <= 0 || dil <= 0) return VITS_E_BADARG;
  if (c > 1024 || k > 8 || (k % 2) == 0) return VITS_E_UNSUPPORTED;
  const int rows = b * t, threads = ((c + 63) / 64) * 64, rpw = pick_rows_per_wg(rows), wgs = (rows + rpw - 1) / rpw;
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (k == 3 && c % 8 == 0 && c >= 64) {                     // 8-channel vector form
    if (dtype == VITS_DT_BF16) hipLaunchKernelGGL(dwconv3_fwd_v8<__bf16>, dim3(wgs), dim3(256), 0, s, (const __bf16*)x, w, bias, lengths, (__bf16*)y, b, t, c, dil, rpw);
    else if (dtype == VITS_DT_F32) hipLaunchKernelGGL(dwconv3_fwd_v8<float>, dim3(wgs), dim3(256), 0, s, (const float*)x, w, bias, lengths, (float*)y, b, t, c, dil, rpw);
    else return VITS_E_UNSUPPORTED;
    return vits::check_launch("vits_dwconv_cl");
  }
  if (dtype == VITS_DT_BF16)
    if (k == 3) hipLaunchKernelGGL((dwconv_fwd<__bf16, 3>), dim3(wgs), dim3(threads), 0, s, (const __bf16*)x, w, bias, lengths, (__bf16*)y, b, t, c, k, dil, rpw);
    else hipLaunchKernelGGL((dwconv_fwd<__bf16, 0>), dim3(wgs), dim3(threads), 0, s, (const __bf16*)x, w, bias, lengths, (__bf16*)y, b, t, c, k, dil, rpw);
  else if (dtype == VITS_DT_F32)
    if (k == 3) hipLaunchKernelGGL((dwconv_fwd<float, 3>), dim3(wgs), dim3(threads), 0, s, (const float*)x, w, bias, lengths, (float*)y, b, t, c, k, dil, rpw);
    else hipLaunchKernelGGL((dwconv_fwd<float, 0>), dim3(wgs), dim3(threads), 0, s, (const float*)x, w, bias, lengths, (float*)y, b, t, c, k, dil, rpw);
  else return VITS_E_UNSUPPORTED;
  return vits::check_launch("vits_dwconv_cl");
}

extern "C" int vits_dwconv_cl_bwd(int dtype, const void* x, const float* w, const int32_t* lengths, const void* dy, void* dx,
                                  float* dw, float* dbias, void* workspace, size_t workspace_bytes, int b, int t, int c, int k,
                                  int dil, int accumulate, void* stream) {
  if (!x || !w || !dy || !dx || !dw || !dbias || !workspace || b <= 0 || t <= 0 || c <= 0 || k <= 0 || dil <= 0) return VITS_E_BADARG;
  if (c > 1024 || k > 8 || (k % 2) == 0) return VITS_E_UNSUPPORTED;
  const int rows = b * t;
  if (workspace_bytes < vits_rowops_workspace(rows, c, k)) return VITS_E_BADARG;
  const int threads = ((c + 63) / 64) * 64, rpw = pick_rows_per_wg(rows, kBwdWgs), wgs = (rows + rpw - 1) / rpw;
  hipStream_t s = static_cast<hipStream_t>(stream);
  float* part = static_cast<float*>(workspace);
  if (k == 3 && c % 8 == 0 && c >= 64) {                     // 8-channel vector form
    const size_t lds = 256 * 32 * sizeof(float);
    if (dtype == VITS_DT_BF16) hipLaunchKernelGGL(dwconv3_bwd_v8<__bf16>, dim3(wgs), dim3(256), lds, s, (const __bf16*)x, w, lengths, (const __bf16*)dy, (__bf16*)dx, part, b, t, c, dil, rpw);
    else if (dtype == VITS_DT_F32) hipLaunchKernelGGL(dwconv3_bwd_v8<float>, dim3(wgs), dim3(256), lds, s, (const float*)x, w, lengths, (const float*)dy, (float*)dx, part, b, t, c, dil, rpw);
    else return VITS_E_UNSUPPORTED;
    hipLaunchKernelGGL(reduce_partials2, dim3((k * c + 63) / 64 + (c + 63) / 64), dim3(1024), 0, s, part, dw, k * c, dbias, c, wgs, (k + 1) * c, accumulate);
    return vits::check_launch("vits_dwconv_cl_bwd");
  }
  if (dtype == VITS_DT_BF16)
    if (k == 3) hipLaunchKernelGGL((dwconv_bwd<__bf16, 3>), dim3(wgs), dim3(threads), 0, s, (const __bf16*)x, w, lengths, (const __bf16*)dy, (__bf16*)dx, part, b, t, c, k, dil, rpw);
    else hipLaunchKernelGGL((dwconv_bwd<__bf16, 0>), dim3(wgs), dim3(threads), 0, s, (const __bf16*)x, w, lengths, (const __bf16*)dy, (__bf16*)dx, part, b, t, c, k, dil, rpw);
  else if (dtype == VITS_DT_F32)
    if (k == 3) hipLaunchKernelGGL((dwconv_bwd<float, 3>), dim3(wgs), dim3(threads), 0, s, (const float*)x, w, lengths, (const float*)dy, (float*)dx, part, b, t, c, k, dil, rpw);
    else hipLaunchKernelGGL((dwconv_bwd<float, 0>), dim3(wgs), dim3(threads), 0, s, (const float*)x, w, lengths, (const float*)dy, (float*)dx, part, b, t, c, k, dil, rpw);
  else return VITS_E_UNSUPPORTED;
  // partial rows are [(k+1)*c]: first k*c = dw[c][k], then c = dbias
  hipLaunchKernelGGL(reduce_partials2, dim3((k * c + 63) / 64 + (c + 63) / 64), dim3(1024), 0, s, part, dw, k * c, dbias, c, wgs, (k + 1) * c, accumulate);
  return vits::check_launch("vits_dwconv_cl_bwd");
}
