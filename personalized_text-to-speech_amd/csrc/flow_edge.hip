// The single-channel input layer of the duration predictor's flows as row kernels (gfx950).
//
//   h[r][c] = x[r] * w[c] + bias[c] (+ g[r][c])         x = one channel of a [rows][xs] fp32 tensor (xs = 1 | 2)
//
// Replaces: modules.ConvFlow.pre = Conv1d(1, filter_channels, 1) applied to the conditioning half of the two-channel flow
//           state, plus the `x + g` of the DDSConv stack it feeds (modules.py:83-96, 366-372), and
//           StochasticDurationPredictor.post_pre = Conv1d(1, filter_channels, 1) on the durations (models.py:43,62) — a rank-1
//           product that has no GEMM shape: on the matrix cores it needs the channel padded to 8 (a pad, a cast, the product,
//           an add; and a data-gradient, a weight-gradient launch with its second stage and a slice in the backward).
// Backward: dx[r] = sum_c dh[r][c] w[c] (written into channel c0 of a zeroed [rows][xs] gradient), dw[c] = sum_r dh[r][c] x[r],
// db[c] = sum_r dh[r][c]: wave per row, lane = channels l, l + 64, ...; per-workgroup partial rows, summed in fixed order by a
// second launch (no atomics).
#include "common.h"

namespace {

constexpr int kThreads = 256;
constexpr int kMaxPerLane = 8;          // channels <= 512

__device__ __forceinline__ float ldv(const float* p, size_t i) { return p[i]; }
__device__ __forceinline__ float ldv(const __bf16* p, size_t i) { return (float)p[i]; }
__device__ __forceinline__ void stv(float* p, size_t i, float v) { p[i] = v; }
__device__ __forceinline__ void stv(__bf16* p, size_t i, float v) { p[i] = (__bf16)v; }

template <typename T>
__global__ __launch_bounds__(kThreads) void front_fwd_kernel(const float* __restrict__ x, int xs, int c0, const float* __restrict__ w,
                                                             const float* __restrict__ bias, const T* __restrict__ g, T* __restrict__ h,
                                                             int rows, int C) {
  const size_t n = (size_t)rows * C;
  for (size_t i = (size_t)blockIdx.x * kThreads + threadIdx.x; i < n; i += (size_t)gridDim.x * kThreads) {
    const size_t r = i / C;
    const int c = (int)(i - r * C);
    float v = x[r * xs + c0] * w[c] + (bias ? bias[c] : 0.f);
    if (g) v += ldv(g, i);
    stv(h, i, v);
  }
}

// workgroup = 4 waves; wave = rows r0 + wave, r0 + wave + 4, ...; partial[wg][2][C] = (dw, db) of the workgroup's rows
template <typename T>
__global__ __launch_bounds__(kThreads) void front_bwd_kernel(const float* __restrict__ x, int xs, int c0, const float* __restrict__ w,
                                                             const T* __restrict__ dh, float* __restrict__ dx, float* __restrict__ part,
                                                             int rows, int C, int rows_per_wg) {
  extern __shared__ float red[];                 // [4 waves][2][C]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float wl[kMaxPerLane], dwl[kMaxPerLane], dbl[kMaxPerLane];
#pragma unroll
  for (int j = 0; j < kMaxPerLane; ++j) {
    const int c = lane + 64 * j;
    wl[j] = c < C ? w[c] : 0.f;
    dwl[j] = 0.f; dbl[j] = 0.f;
  }
  const int r0 = blockIdx.x * rows_per_wg;
  const int r1 = r0 + rows_per_wg < rows ? r0 + rows_per_wg : rows;
  for (int r = r0 + wave; r < r1; r += 4) {
    const float xv = x[(size_t)r * xs + c0];
    float dot = 0.f;
#pragma unroll
    for (int j = 0; j < kMaxPerLane; ++j) {
      const int c = lane + 64 * j;
      if (c < C) {
        const float d = ldv(dh, (size_t)r * C + c);
        dot += d * wl[j];
        dwl[j] += d * xv;
        dbl[j] += d;
      }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) dot += __shfl_xor(dot, o, 64);
    if (lane < xs) dx[(size_t)r * xs + lane] = (lane == c0) ? dot : 0.f;
  }
#pragma unroll
  for (int j = 0; j < kMaxPerLane; ++j) {
    const int c = lane + 64 * j;
    if (c < C) { red[(wave * 2 + 0) * C + c] = dwl[j]; red[(wave * 2 + 1) * C + c] = dbl[j]; }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 2 * C; i += kThreads)
    part[(size_t)blockIdx.x * 2 * C + i] = red[i] + red[2 * C + i] + red[4 * C + i] + red[6 * C + i];
}

// out[i] = sum over the workgroups' partial rows (i < C -> dw, else db): block = 64 columns x 16 slices of the workgroup index,
// the slices summed through LDS in a fixed order
__global__ __launch_bounds__(1024) void front_final_kernel(const float* __restrict__ part, int wgs, int C, float* __restrict__ dw,
                                                           float* __restrict__ db) {
  __shared__ float sm[16][64];
  const int col = threadIdx.x & 63, slice = threadIdx.x >> 6;
  const int i = blockIdx.x * 64 + col;
  float s = 0.f;
  if (i < 2 * C)
    for (int g = slice; g < wgs; g += 16) s += part[(size_t)g * 2 * C + i];
  sm[slice][col] = s;
  __syncthreads();
  if (slice == 0 && i < 2 * C) {
    float tot = 0.f;
#pragma unroll
    for (int k = 0; k < 16; ++k) tot += sm[k][col];
    if (i < C) dw[i] = tot; else db[i - C] = tot;
  }
}

// ---- mean-only residual coupling layer: the element-wise tail (modules.py:330-343) with the channel flip that follows it
// (modules.py:273-279) folded in:   y = flip([x0, m + x1 * mask])   i.e.   y[r][C-1-c] = c < half ? x[r][c] : stats[r][c-half] + x[r][c] * mask[r]
// backward:  dx[r][c] = c < half ? dy[r][C-1-c] : dy[r][C-1-c] * mask[r],   dstats[r][j] = dy[r][C-1-half-j]
template <typename T, bool BWD>
__global__ __launch_bounds__(kThreads) void coupling_tail_kernel(const T* __restrict__ x, const T* __restrict__ stats, const int* __restrict__ lengths,
                                                                 T* __restrict__ y, T* __restrict__ dstats, int t, int C, int half, int flip,
                                                                 size_t n) {
  for (size_t i = (size_t)blockIdx.x * kThreads + threadIdx.x; i < n; i += (size_t)gridDim.x * kThreads) {
    const size_t r = i / C;
    const int c = (int)(i - r * C);
    const int b = (int)(r / t), tt = (int)(r - (size_t)b * t);
    const float mv = (lengths == nullptr || tt < lengths[b]) ? 1.f : 0.f;
    const size_t j = r * C + (flip ? C - 1 - c : c);          // position of channel c in the (flipped) output
    if (!BWD) {
      float v = ldv(x, i);
      if (c >= half) v = ldv(stats, r * (C - half) + (c - half)) + v * mv;
      stv(y, j, v);
    } else {                                                   // x = dy (flipped layout), y = dx
      const float d = ldv(x, j);
      stv(y, i, c < half ? d : d * mv);
      if (c >= half) stv(dstats, r * (C - half) + (c - half), d);
    }
  }
}

// ---- modules.ElementwiseAffine (modules.py:280-295) on a channels-last [b][t][C] state, C <= 8 ------------------------------------
// forward: y = (m[pc] + exp(logs[pc]) x) mask, logdet[b] = (sum_c logs[c]) len[b];  inverse: y = (x - m[pc]) exp(-logs[pc]) mask
// (pc = C-1-c when `swap`: the state is virtually flipped, see vits_flow_spline).
__global__ __launch_bounds__(kThreads) void affine_fwd_kernel(const float* __restrict__ x, const float* __restrict__ m, const float* __restrict__ logs,
                                                              const int* __restrict__ lengths, float* __restrict__ y, float* __restrict__ logdet,
                                                              int B, int t, int C, int swap, int inverse) {
  const size_t n = (size_t)B * t * C;
  for (size_t i = (size_t)blockIdx.x * kThreads + threadIdx.x; i < n; i += (size_t)gridDim.x * kThreads) {
    const size_t r = i / C;
    const int c = (int)(i - r * C), pc = swap ? C - 1 - c : c;
    const int b = (int)(r / t), tt = (int)(r - (size_t)b * t);
    const float mv = tt < lengths[b] ? 1.f : 0.f;
    y[i] = inverse ? (x[i] - m[pc]) * expf(-logs[pc]) * mv : (m[pc] + expf(logs[pc]) * x[i]) * mv;
  }
  if (logdet && blockIdx.x == 0 && (int)threadIdx.x < B) {
    float sl = 0.f;
    for (int c = 0; c < C; ++c) sl += logs[c];
    const int len = lengths[threadIdx.x] < t ? lengths[threadIdx.x] : t;
    logdet[threadIdx.x] = sl * (float)len;
  }
}

// one workgroup: dx = dy exp(logs) mask;  dm[pc] = sum dy mask;  dlogs[pc] = sum dy exp(logs) x mask + sum_b dlogdet[b] len[b]
__global__ __launch_bounds__(kThreads) void affine_bwd_kernel(const float* __restrict__ x, const float* __restrict__ logs, const int* __restrict__ lengths,
                                                              const float* __restrict__ dy, const float* __restrict__ dlogdet, float* __restrict__ dx,
                                                              float* __restrict__ dm, float* __restrict__ dlogs, int B, int t, int C, int swap) {
  __shared__ float red[kThreads][16];
  float am[8], al[8];
#pragma unroll
  for (int c = 0; c < 8; ++c) { am[c] = 0.f; al[c] = 0.f; }
  const size_t rows = (size_t)B * t;
  for (size_t r = threadIdx.x; r < rows; r += kThreads) {
    const int b = (int)(r / t), tt = (int)(r - (size_t)b * t);
    const float mv = tt < lengths[b] ? 1.f : 0.f;
#pragma unroll
    for (int c = 0; c < 8; ++c)
      if (c < C) {
        const int pc = swap ? C - 1 - c : c;
        const float d = (dy ? dy[r * C + c] : 0.f) * mv, e = expf(logs[pc]);
        dx[r * C + c] = d * e;
        am[c] += d;
        al[c] += d * e * x[r * C + c];
      }
  }
#pragma unroll
  for (int c = 0; c < 8; ++c) { red[threadIdx.x][c] = am[c]; red[threadIdx.x][8 + c] = al[c]; }
  __syncthreads();
  if ((int)threadIdx.x < 2 * C) {
    const int c = threadIdx.x % C, which = threadIdx.x / C;
    float s = 0.f;
    for (int k = 0; k < kThreads; ++k) s += red[k][which * 8 + c];
    const int pc = swap ? C - 1 - c : c;
    if (which == 0) dm[pc] = s;
    else {
      float ld = 0.f;
      if (dlogdet)
        for (int b = 0; b < B; ++b) ld += dlogdet[b] * (float)(lengths[b] < t ? lengths[b] : t);
      dlogs[pc] = s + ld;
    }
  }
}

// ---- the dequantisation + Log step between the two flow chains of the stochastic duration predictor (models.py:71-80) ---------
//   u = sigmoid(z_u) m;  z0 = (w - u) m;  s1[b] = sum_t (logsigmoid(z_u) + logsigmoid(-z_u)) m
//   z0l = log(max(z0, 1e-5)) m  (modules.Log);  s2[b] = sum_t -z0l;   out[r] = [z0l, z1]            (z_q [rows][2] = [z_u, z1])
// one workgroup per item (the per-item sums are its block reductions, fixed order)
__device__ __forceinline__ float logsigmoid_f(float v) { return fminf(v, 0.f) - log1pf(expf(-fabsf(v))); }

template <bool BWD>
__global__ __launch_bounds__(kThreads) void dequant_log_kernel(const float* __restrict__ zq, const float* __restrict__ w, const int* __restrict__ lengths,
                                                               float* __restrict__ out, float* __restrict__ s1, float* __restrict__ s2,
                                                               const float* __restrict__ dout, const float* __restrict__ ds1, const float* __restrict__ ds2,
                                                               float* __restrict__ dzq, int t) {
  __shared__ float red[2][4];
  const int b = blockIdx.x, len = lengths[b] < t ? lengths[b] : t;
  float a1 = 0.f, a2 = 0.f;
  const float g1 = BWD && ds1 ? ds1[b] : 0.f, g2 = BWD && ds2 ? ds2[b] : 0.f;
  for (int tt = threadIdx.x; tt < t; tt += kThreads) {
    const size_t r = (size_t)b * t + tt;
    const float mv = tt < len ? 1.f : 0.f;
    const float zu = zq[2 * r], z1 = zq[2 * r + 1];
    const float sg = 1.f / (1.f + expf(-zu));
    const float z0 = (w[r] - sg * mv) * mv;
    const bool above = z0 > 1e-5f;
    const float z0l = logf(above ? z0 : 1e-5f) * mv;
    if (!BWD) {
      out[2 * r] = z0l; out[2 * r + 1] = z1;
      a1 += (logsigmoid_f(zu) + logsigmoid_f(-zu)) * mv;
      a2 -= z0l;
    } else {
      const float d0 = (dout ? dout[2 * r] : 0.f) - g2;              // gradient of z0l (its direct use and the -z0l sum)
      const float dz0 = above ? d0 * mv / z0 : 0.f;                    // log(clamp_min(z0, 1e-5)): no gradient below the clamp
      const float du = -dz0 * mv;                                      // z0 = (w - u) m
      dzq[2 * r] = du * mv * sg * (1.f - sg) + g1 * mv * (1.f - 2.f * sg);
      dzq[2 * r + 1] = dout ? dout[2 * r + 1] : 0.f;
    }
  }
  if (!BWD) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { a1 += __shfl_xor(a1, o, 64); a2 += __shfl_xor(a2, o, 64); }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) { red[0][wave] = a1; red[1][wave] = a2; }
    __syncthreads();
    if (threadIdx.x == 0) {
      s1[b] = red[0][0] + red[0][1] + red[0][2] + red[0][3];
      s2[b] = red[1][0] + red[1][1] + red[1][2] + red[1][3];
    }
  }
}

int front_wgs(int rows) { int w = (rows + 31) / 32; return w < 1 ? 1 : (w > 128 ? 128 : w); }

}  // namespace

extern "C" int vits_flow_front(int dtype, const float* x, int xs, int c0, const float* w, const float* bias, const void* g, void* h,
                               int rows, int c, void* stream) {
  if (!x || !w || !h || rows <= 0 || c <= 0 || (xs != 1 && xs != 2) || c0 < 0 || c0 >= xs) return VITS_E_BADARG;
  hipStream_t s = static_cast<hipStream_t>(stream);
  const size_t n = (size_t)rows * c;
  unsigned blocks = (unsigned)((n + kThreads * 4 - 1) / (kThreads * 4));
  if (blocks > 1024) blocks = 1024;
  if (dtype == VITS_DT_BF16)
    hipLaunchKernelGGL(front_fwd_kernel<__bf16>, dim3(blocks), dim3(kThreads), 0, s, x, xs, c0, w, bias, (const __bf16*)g, (__bf16*)h, rows, c);
  else if (dtype == VITS_DT_F32)
    hipLaunchKernelGGL(front_fwd_kernel<float>, dim3(blocks), dim3(kThreads), 0, s, x, xs, c0, w, bias, (const float*)g, (float*)h, rows, c);
  else return VITS_E_UNSUPPORTED;
  return vits::check_launch("vits_flow_front");
}

extern "C" size_t vits_flow_front_workspace(int rows, int c) { return (size_t)front_wgs(rows) * 2 * c * sizeof(float); }

extern "C" int vits_flow_front_bwd(int dtype, const float* x, int xs, int c0, const float* w, const void* dh, float* dx, float* dw, float* db,
                                   void* workspace, size_t workspace_bytes, int rows, int c, void* stream) {
  if (!x || !w || !dh || !dx || !dw || !db || !workspace || rows <= 0 || c <= 0 || (xs != 1 && xs != 2) || c0 < 0 || c0 >= xs)
    return VITS_E_BADARG;
  if (c > 64 * kMaxPerLane) return VITS_E_UNSUPPORTED;
  if (workspace_bytes < vits_flow_front_workspace(rows, c)) return VITS_E_BADARG;
  hipStream_t s = static_cast<hipStream_t>(stream);
  const int wgs = front_wgs(rows), rpw = (rows + wgs - 1) / wgs;
  const int wgs_used = (rows + rpw - 1) / rpw;
  float* part = static_cast<float*>(workspace);
  const size_t lds = (size_t)8 * c * sizeof(float);
  if (dtype == VITS_DT_BF16)
    hipLaunchKernelGGL(front_bwd_kernel<__bf16>, dim3(wgs_used), dim3(kThreads), lds, s, x, xs, c0, w, (const __bf16*)dh, dx, part, rows, c, rpw);
  else if (dtype == VITS_DT_F32)
    hipLaunchKernelGGL(front_bwd_kernel<float>, dim3(wgs_used), dim3(kThreads), lds, s, x, xs, c0, w, (const float*)dh, dx, part, rows, c, rpw);
  else return VITS_E_UNSUPPORTED;
  hipLaunchKernelGGL(front_final_kernel, dim3((2 * c + 63) / 64), dim3(1024), 0, s, part, wgs_used, c, dw, db);
  return vits::check_launch("vits_flow_front_bwd");
}

extern "C" int vits_coupling_tail(int dtype, const void* x, const void* stats, const int32_t* lengths, void* y, int b, int t, int c, int half,
                                  int flip, void* stream) {
  if (!x || !stats || !y || b <= 0 || t <= 0 || c <= 0 || half <= 0 || half >= c) return VITS_E_BADARG;
  hipStream_t s = static_cast<hipStream_t>(stream);
  const size_t n = (size_t)b * t * c;
  unsigned blocks = (unsigned)((n + kThreads * 4 - 1) / (kThreads * 4));
  if (blocks > 2048) blocks = 2048;
  if (dtype == VITS_DT_BF16)
    hipLaunchKernelGGL((coupling_tail_kernel<__bf16, false>), dim3(blocks), dim3(kThreads), 0, s, (const __bf16*)x, (const __bf16*)stats, lengths, (__bf16*)y, (__bf16*)nullptr, t, c, half, flip, n);
  else if (dtype == VITS_DT_F32)
    hipLaunchKernelGGL((coupling_tail_kernel<float, false>), dim3(blocks), dim3(kThreads), 0, s, (const float*)x, (const float*)stats, lengths, (float*)y, (float*)nullptr, t, c, half, flip, n);
  else return VITS_E_UNSUPPORTED;
  return vits::check_launch("vits_coupling_tail");
}

extern "C" int vits_coupling_tail_bwd(int dtype, const void* dy, const int32_t* lengths, void* dx, void* dstats, int b, int t, int c, int half,
                                      int flip, void* stream) {
  if (!dy || !dx || !dstats || b <= 0 || t <= 0 || c <= 0 || half <= 0 || half >= c) return VITS_E_BADARG;
  hipStream_t s = static_cast<hipStream_t>(stream);
  const size_t n = (size_t)b * t * c;
  unsigned blocks = (unsigned)((n + kThreads * 4 - 1) / (kThreads * 4));
  if (blocks > 2048) blocks = 2048;
  if (dtype == VITS_DT_BF16)
    hipLaunchKernelGGL((coupling_tail_kernel<__bf16, true>), dim3(blocks), dim3(kThreads), 0, s, (const __bf16*)dy, (const __bf16*)nullptr, lengths, (__bf16*)dx, (__bf16*)dstats, t, c, half, flip, n);
  else if (dtype == VITS_DT_F32)
    hipLaunchKernelGGL((coupling_tail_kernel<float, true>), dim3(blocks), dim3(kThreads), 0, s, (const float*)dy, (const float*)nullptr, lengths, (float*)dx, (float*)dstats, t, c, half, flip, n);
  else return VITS_E_UNSUPPORTED;
  return vits::check_launch("vits_coupling_tail_bwd");
}

extern "C" int vits_flow_affine(const float* x, const float* m, const float* logs, const int32_t* lengths, float* y, float* logdet, int b, int t,
                                int c, int swap, int inverse, void* stream) {
  if (!x || !m || !logs || !lengths || !y || b <= 0 || t <= 0 || c <= 0) return VITS_E_BADARG;
  if (c > 8 || b > kThreads) return VITS_E_UNSUPPORTED;
  const size_t n = (size_t)b * t * c;
  unsigned blocks = (unsigned)((n + kThreads * 4 - 1) / (kThreads * 4));
  if (blocks > 1024) blocks = 1024;
  hipLaunchKernelGGL(affine_fwd_kernel, dim3(blocks), dim3(kThreads), 0, static_cast<hipStream_t>(stream), x, m, logs, lengths, y, logdet, b, t, c, swap,
                     inverse);
  return vits::check_launch("vits_flow_affine");
}

extern "C" int vits_flow_affine_bwd(const float* x, const float* logs, const int32_t* lengths, const float* dy, const float* dlogdet, float* dx,
                                    float* dm, float* dlogs, int b, int t, int c, int swap, void* stream) {
  if (!x || !logs || !lengths || !dx || !dm || !dlogs || b <= 0 || t <= 0 || c <= 0) return VITS_E_BADARG;
  if (c > 8) return VITS_E_UNSUPPORTED;
  hipLaunchKernelGGL(affine_bwd_kernel, dim3(1), dim3(kThreads), 0, static_cast<hipStream_t>(stream), x, logs, lengths, dy, dlogdet, dx, dm, dlogs, b, t,
                     c, swap);
  return vits::check_launch("vits_flow_affine_bwd");
}

extern "C" int vits_flow_dequant_log(const float* zq, const float* w, const int32_t* lengths, float* out, float* s1, float* s2, int b, int t,
                                     void* stream) {
  if (!zq || !w || !lengths || !out || !s1 || !s2 || b <= 0 || t <= 0) return VITS_E_BADARG;
  hipLaunchKernelGGL(dequant_log_kernel<false>, dim3(b), dim3(kThreads), 0, static_cast<hipStream_t>(stream), zq, w, lengths, out, s1, s2, nullptr, nullptr,
                     nullptr, nullptr, t);
  return vits::check_launch("vits_flow_dequant_log");
}

extern "C" int vits_flow_dequant_log_bwd(const float* zq, const float* w, const int32_t* lengths, const float* dout, const float* ds1, const float* ds2,
                                         float* dzq, int b, int t, void* stream) {
  if (!zq || !w || !lengths || !dzq || b <= 0 || t <= 0) return VITS_E_BADARG;
  hipLaunchKernelGGL(dequant_log_kernel<true>, dim3(b), dim3(kThreads), 0, static_cast<hipStream_t>(stream), zq, w, lengths, nullptr, nullptr, nullptr, dout,
                     ds1, ds2, dzq, t);
  return vits::check_launch("vits_flow_dequant_log_bwd");
}
