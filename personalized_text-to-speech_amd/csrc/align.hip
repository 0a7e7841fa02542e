// Kernels around the alignment step of SynthesizerTrn.forward / infer (gfx950):
//   vits_neg_cent            the (frame, token) negative cross-entropy matrix that feeds the alignment DP   (models.py:470-477)
//   vits_slice_segments[_bwd] per-item segment slices and their gradient                                      (commons.py:48-67)
//   vits_generate_path       durations -> hard monotonic path                                               (commons.py:131-146)
#include "common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int kThreads = 256;
constexpr float kLog2Pi = 1.8378770664093453f;

__device__ __forceinline__ float ld_f(const float* p) { return *p; }
__device__ __forceinline__ float ld_f(const __bf16* p) { return (float)*p; }

// ---------------------------------------------------------------------------------------------------------------------
// neg_cent[b][t][s] = sum_c ( -0.5 log(2 pi) - logs[s][c] )  +  sum_c -0.5 z[t][c]^2 r[s][c]  +  sum_c z[t][c] m[s][c] r[s][c]
//                     + sum_c -0.5 m[s][c]^2 r[s][c],      r = exp(-2 logs)
// = one K = 2C product [z^2 | z] x [-0.5 r | m r]^T plus a per-token bias, in fp32 on the matrix cores
// (v_mfma_f32_32x32x2_f32: a k-ordered fmaf chain).  The reference materialises r, z^2, m r and four [b, t, s] tensors.
// Workgroup = 64 frames x 64 tokens, 4 waves (2 x 2), K walked in chunks of 32 channels: z rows and the two derived token
// operands are staged in LDS (144-byte pitch: conflict-free 16-byte fragment reads); each thread derives 8 channels of one
// token per chunk and keeps that token's bias partial in a register.
template <typename TZ, typename TS>
__global__ __launch_bounds__(kThreads) void neg_cent_kernel(const TZ* __restrict__ Z, long ldz, const TS* __restrict__ Mp,
                                                            const TS* __restrict__ Lp, long lds_, float* __restrict__ NC,
                                                            int T_t, int T_s, int C) {
  constexpr int PITCH = 36;                     // floats per LDS row (32 + 4)
  __shared__ __attribute__((aligned(16))) float Zs[64 * PITCH], B1[64 * PITCH], B2[64 * PITCH];
  __shared__ float bias_s[64];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave & 1, wn = wave >> 1, r = lane & 31, h = lane >> 5;
  const int s0 = blockIdx.x * 64, t0 = blockIdx.y * 64, b = blockIdx.z;
  const int row = tid >> 2, q = tid & 3;
  const bool t_ok = t0 + row < T_t, s_ok = s0 + row < T_s;
  const TZ* zrow = Z + ((size_t)b * T_t + (t_ok ? t0 + row : 0)) * ldz + q * 8;
  const TS* mrow = Mp + ((size_t)b * T_s + (s_ok ? s0 + row : 0)) * lds_ + q * 8;
  const TS* lrow = Lp + ((size_t)b * T_s + (s_ok ? s0 + row : 0)) * lds_ + q * 8;
  f32x16 acc;
#pragma unroll
  for (int e = 0; e < 16; ++e) acc[e] = 0.f;
  float bias = 0.f;
  for (int c0 = 0; c0 < C; c0 += 32) {
    float zv[8], b1[8], b2[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const bool c_ok = c0 + q * 8 + e < C;        // (a last partial chunk contributes zeros)
      zv[e] = (t_ok && c_ok) ? ld_f(zrow + c0 + e) : 0.f;
      const float m = (s_ok && c_ok) ? ld_f(mrow + c0 + e) : 0.f, l = (s_ok && c_ok) ? ld_f(lrow + c0 + e) : 0.f;
      const float rr = c_ok ? expf(-2.0f * l) : 0.f;
      b1[e] = -0.5f * rr;
      b2[e] = m * rr;
      bias += c_ok ? (-0.5f * kLog2Pi - l) - 0.5f * m * m * rr : 0.f;
    }
    __syncthreads();                             // previous chunk's fragments have been read
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      Zs[row * PITCH + q * 8 + e] = zv[e];
      B1[row * PITCH + q * 8 + e] = b1[e];
      B2[row * PITCH + q * 8 + e] = b2[e];
    }
    __syncthreads();
#pragma unroll
    for (int mm = 0; mm < 4; ++mm) {
      const float4 a = *reinterpret_cast<const float4*>(&Zs[(wm * 32 + r) * PITCH + mm * 8 + h * 4]);
      const float4 u = *reinterpret_cast<const float4*>(&B1[(wn * 32 + r) * PITCH + mm * 8 + h * 4]);
      const float4 v = *reinterpret_cast<const float4*>(&B2[(wn * 32 + r) * PITCH + mm * 8 + h * 4]);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x * a.x, u.x, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y * a.y, u.y, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z * a.z, u.z, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w * a.w, u.w, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, v.x, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, v.y, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, v.z, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, v.w, acc, 0, 0, 0);
    }
  }
  bias += __shfl_xor(bias, 1, 64);               // the four threads of a token are adjacent lanes
  bias += __shfl_xor(bias, 2, 64);
  if (q == 0) bias_s[row] = bias;
  __syncthreads();
  const int s = s0 + wn * 32 + r;
  const float bs = bias_s[wn * 32 + r];
  if (s < T_s) {
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int t = t0 + wm * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
      if (t < T_t) NC[((size_t)b * T_t + t) * T_s + s] = acc[e] + bs;
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// Segment slices.  rows layout:  x [b][T][C] -> y [b][seg][C]  (channels-last activations);  time-inner layout:
// x [b][D][T] -> y [b][D][seg] (the reference's layout: mel, waveform).  Elements are moved as raw 2- or 4-byte words.
template <typename W>
__global__ void slice_rows_kernel(const W* __restrict__ x, const long* __restrict__ ids, W* __restrict__ y, int T, int C, int seg,
                                  long mul) {
  const int b = blockIdx.y;
  const long start = ids[b] * mul;
  const size_t n = (size_t)seg * C;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const long t = start + (long)(i / C);
    y[(size_t)b * n + i] = (t >= 0 && t < T) ? x[((size_t)b * T + t) * C + i % C] : W(0);
  }
}

template <typename W>
__global__ void slice_rows_bwd_kernel(const W* __restrict__ dy, const long* __restrict__ ids, W* __restrict__ dx, int T, int C,
                                      int seg, long mul) {
  const int b = blockIdx.y;
  const long start = ids[b] * mul;
  const size_t n = (size_t)T * C;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const long t = (long)(i / C) - start;
    dx[(size_t)b * n + i] = (t >= 0 && t < seg) ? dy[((size_t)b * seg + t) * C + i % C] : W(0);
  }
}

template <typename W>
__global__ void slice_time_kernel(const W* __restrict__ x, const long* __restrict__ ids, W* __restrict__ y, int D, int T, int seg,
                                  long mul) {
  const int b = blockIdx.y;
  const long start = ids[b] * mul;
  const size_t n = (size_t)D * seg;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const long t = start + (long)(i % seg);
    y[(size_t)b * n + i] = (t >= 0 && t < T) ? x[((size_t)b * D + i / seg) * T + t] : W(0);
  }
}

template <typename W>
__global__ void slice_time_bwd_kernel(const W* __restrict__ dy, const long* __restrict__ ids, W* __restrict__ dx, int D, int T,
                                      int seg, long mul) {
  const int b = blockIdx.y;
  const long start = ids[b] * mul;
  const size_t n = (size_t)D * T;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const long t = (long)(i % T) - start;
    dx[(size_t)b * n + i] = (t >= 0 && t < seg) ? dy[((size_t)b * D + i / T) * seg + t] : W(0);
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// generate_path: path[b][y][x] = ((y < cum[x]) - (y < cum[x-1])) * mask  (cum = inclusive prefix sum of the durations), i.e. 1 on
// the frames of token x.  Workgroup = 32 frames of one item; every workgroup redoes the item's prefix sum (t_x <= 15 k values,
// 256-wide wave scans with a serial carry) instead of a second launch.
__global__ __launch_bounds__(kThreads) void generate_path_kernel(const float* __restrict__ dur, const float* __restrict__ mask,
                                                                 float* __restrict__ path, int t_y, int t_x) {
  extern __shared__ float cum[];                 // [t_x]
  __shared__ float carry_s, wave_tot[4];
  const int b = blockIdx.x, tid = threadIdx.x;
  if (tid == 0) carry_s = 0.f;
  __syncthreads();
  for (int x0 = 0; x0 < t_x; x0 += kThreads) {
    const int x = x0 + tid;
    float v = x < t_x ? dur[(size_t)b * t_x + x] : 0.f;
    for (int o = 1; o < 64; o <<= 1) {           // inclusive scan inside the wave
      const float u = __shfl_up(v, o, 64);
      if ((tid & 63) >= o) v += u;
    }
    if ((tid & 63) == 63) wave_tot[tid >> 6] = v;
    __syncthreads();
    float base = carry_s;
    for (int w = 0; w < (tid >> 6); ++w) base += wave_tot[w];
    if (x < t_x) cum[x] = v + base;
    __syncthreads();
    if (tid == 0) carry_s += wave_tot[0] + wave_tot[1] + wave_tot[2] + wave_tot[3];
    __syncthreads();
  }
  const int y0 = blockIdx.y * 32;
  const int rows = t_y - y0 < 32 ? t_y - y0 : 32;
  const size_t n = (size_t)rows * t_x, base = ((size_t)b * t_y + y0) * t_x;
  for (size_t i = tid; i < n; i += kThreads) {
    const int y = y0 + (int)(i / t_x), x = (int)(i % t_x);
    const float on = ((float)y < cum[x] ? 1.f : 0.f) - ((x > 0 && (float)y < cum[x - 1]) ? 1.f : 0.f);
    path[base + i] = on * mask[base + i];
  }
}

template <typename TZ, typename TS>
int launch_neg_cent(const void* z, long ldz, const void* m, const void* logs, long lds_, float* nc, int b, int t_t, int t_s, int c,
                    hipStream_t s) {
  dim3 grid(vits::ceil_div(t_s, 64), vits::ceil_div(t_t, 64), b);
  hipLaunchKernelGGL((neg_cent_kernel<TZ, TS>), grid, dim3(kThreads), 0, s, static_cast<const TZ*>(z), ldz, static_cast<const TS*>(m),
                     static_cast<const TS*>(logs), lds_, nc, t_t, t_s, c);
  return vits::check_launch("vits_neg_cent");
}

}  // namespace

extern "C" int vits_neg_cent(int z_dtype, const void* z, long ldz, int s_dtype, const void* m, const void* logs, long lds_, float* nc,
                             int b, int t_t, int t_s, int c, void* stream) {
  if (!z || !m || !logs || !nc || b <= 0 || t_t <= 0 || t_s <= 0 || c <= 0 || ldz < c || lds_ < c) return VITS_E_BADARG;
  if (b > 65535 || vits::ceil_div(t_t, 64) > 65535) return VITS_E_UNSUPPORTED;
  hipStream_t s = static_cast<hipStream_t>(stream);
  const bool zb = z_dtype == VITS_DT_BF16, sb = s_dtype == VITS_DT_BF16;
  if ((!zb && z_dtype != VITS_DT_F32) || (!sb && s_dtype != VITS_DT_F32)) return VITS_E_UNSUPPORTED;
  if (zb && sb) return launch_neg_cent<__bf16, __bf16>(z, ldz, m, logs, lds_, nc, b, t_t, t_s, c, s);
  if (zb) return launch_neg_cent<__bf16, float>(z, ldz, m, logs, lds_, nc, b, t_t, t_s, c, s);
  if (sb) return launch_neg_cent<float, __bf16>(z, ldz, m, logs, lds_, nc, b, t_t, t_s, c, s);
  return launch_neg_cent<float, float>(z, ldz, m, logs, lds_, nc, b, t_t, t_s, c, s);
}

extern "C" int vits_slice_segments(int elem_bytes, int time_inner, const void* x, const int64_t* ids, long ids_mul, void* y, int b,
                                   int d, int t, int seg, int backward, void* stream) {
  if (!x || !ids || !y || b <= 0 || d <= 0 || t <= 0 || seg <= 0 || b > 65535) return VITS_E_BADARG;
  if (elem_bytes != 2 && elem_bytes != 4) return VITS_E_UNSUPPORTED;
  hipStream_t s = static_cast<hipStream_t>(stream);
  const size_t n = (size_t)d * (backward ? t : seg);
  int blocks = (int)((n + kThreads * 4 - 1) / (kThreads * 4));
  blocks = blocks < 1 ? 1 : (blocks > 1024 ? 1024 : blocks);
  const dim3 grid(blocks, b);
  const long* idp = reinterpret_cast<const long*>(ids);
#define VITS_SLICE(W)                                                                                                          \
  do {                                                                                                                         \
    const W* xs = static_cast<const W*>(x);                                                                                    \
    W* ys = static_cast<W*>(y);                                                                                                \
    if (!time_inner && !backward) hipLaunchKernelGGL(slice_rows_kernel<W>, grid, dim3(kThreads), 0, s, xs, idp, ys, t, d, seg, ids_mul); \
    else if (!time_inner) hipLaunchKernelGGL(slice_rows_bwd_kernel<W>, grid, dim3(kThreads), 0, s, xs, idp, ys, t, d, seg, ids_mul);     \
    else if (!backward) hipLaunchKernelGGL(slice_time_kernel<W>, grid, dim3(kThreads), 0, s, xs, idp, ys, d, t, seg, ids_mul);           \
    else hipLaunchKernelGGL(slice_time_bwd_kernel<W>, grid, dim3(kThreads), 0, s, xs, idp, ys, d, t, seg, ids_mul);                      \
  } while (0)
  if (elem_bytes == 2) VITS_SLICE(uint16_t); else VITS_SLICE(uint32_t);
#undef VITS_SLICE
  return vits::check_launch("vits_slice_segments");
}

extern "C" int vits_generate_path(const float* duration, const float* mask, float* path, int b, int t_y, int t_x, void* stream) {
  if (!duration || !mask || !path || b <= 0 || t_y <= 0 || t_x <= 0) return VITS_E_BADARG;
  if ((size_t)t_x * 4 > 60 * 1024) return VITS_E_UNSUPPORTED;
  if (vits::ceil_div(t_y, 32) > 65535) return VITS_E_UNSUPPORTED;
  hipLaunchKernelGGL(generate_path_kernel, dim3(b, vits::ceil_div(t_y, 32)), dim3(kThreads), (size_t)t_x * 4,
                     static_cast<hipStream_t>(stream), duration, mask, path, t_y, t_x);
  return vits::check_launch("vits_generate_path");
}
