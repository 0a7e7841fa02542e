// The two "edge" layers of every discriminator (reference models.py:299-361) — the single-input-channel first convolution
// and the single-output-channel conv_post — as bandwidth kernels instead of 8-channel-padded matrix-core launches:
//
//   first layer   DiscriminatorP: view [n,1,T/p,p] of the reflect-padded waveform -> Conv2d(1, 32, (5,1), (3,1), pad (2,0)) -> lrelu
//                 DiscriminatorS: Conv1d(1, 16, 15, 1, pad 7) -> lrelu            (p = 1)
//                 One output element costs k MACs: HBM-bound.  The kernels read the raw fp32 waveform [n][T] directly
//                 (fold into (item, column) pairs, reflect pad and zero pad are index arithmetic), so the pad / view /
//                 transpose / cast launches of the reference graph disappear together with the 8x padded input tensor.
//   conv_post     Conv2d(1024, 1, (3,1), pad (1,0)) / Conv1d(1024, 1, 3, pad 1): a 3 x 1024 dot product per output row.
//
// Layouts are the library's channels-last ones: h1 [(n,w)][r1][c_out], h_last [(n,w)][r][c_in], logits y8 [(n,w)][r][8]
// (channel 0 live; 8 = the vector width the feature-matching kernels expect).  Weights come from the weight arena:
// first layer w [k][c_out][8] (input channel 0 live), conv_post w [k][8][c_in] (output channel 0 live); weight gradients are
// written to the same positions of the arena's fp32 dw.  All reductions are two-stage, fixed order (no atomics).
#include "common.h"

namespace {

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
constexpr int kThreads = 256;

__device__ __forceinline__ float to_f(float v) { return v; }
__device__ __forceinline__ float to_f(__bf16 v) { return (float)v; }
template <typename T> __device__ __forceinline__ T from_f(float v);
template <> __device__ __forceinline__ float from_f<float>(float v) { return v; }
template <> __device__ __forceinline__ __bf16 from_f<__bf16>(float v) { return (__bf16)v; }

struct FirstGeom {
  int n, T, p, R0, R1, k, s1, pad, c_out;      // R0 = ceil(T/p) rows of the folded input, R1 rows of the output
};

// element (item j = n*p + w, row r) of the folded, reflect-padded, zero-padded waveform
__device__ __forceinline__ float fold_x(const float* __restrict__ x, const FirstGeom& g, int ni, int w, int r) {
  if (r < 0 || r >= g.R0) return 0.f;
  int s = r * g.p + w;
  if (s >= g.T) s = 2 * (g.T - 1) - s;         // F.pad(..., "reflect") on the right (models.py:319-322)
  return x[(size_t)ni * g.T + s];
}

// ---------------------------------------------------------------------------------------------------------------------
// first layer, forward: thread = one output row x 8 channels
template <typename T>
__global__ __launch_bounds__(kThreads) void first_fwd(const float* __restrict__ x, const T* __restrict__ w, const float* __restrict__ bias,
                                                      T* __restrict__ y, FirstGeom g, float slope) {
  __shared__ float ws[16 * 32 + 32];
  const int kc = g.k * g.c_out;
  for (int i = threadIdx.x; i < kc; i += kThreads) ws[i] = to_f(w[(size_t)i * 8]);
  for (int i = threadIdx.x; i < g.c_out; i += kThreads) ws[kc + i] = bias ? bias[i] : 0.f;
  __syncthreads();
  const int groups = g.c_out / 8;
  const long total = (long)g.n * g.p * g.R1 * groups;
  const long idx = (long)blockIdx.x * kThreads + threadIdx.x;
  if (idx >= total) return;
  const int cg = (int)(idx % groups);
  const long m = idx / groups;
  const int r1 = (int)(m % g.R1);
  const int j = (int)(m / g.R1);
  const int ni = j / g.p, wcol = j - ni * g.p;
  float acc[8];
#pragma unroll
  for (int c = 0; c < 8; ++c) acc[c] = ws[kc + cg * 8 + c];
  for (int tap = 0; tap < g.k; ++tap) {
    const float xv = fold_x(x, g, ni, wcol, r1 * g.s1 + tap - g.pad);
#pragma unroll
    for (int c = 0; c < 8; ++c) acc[c] = fmaf(ws[tap * g.c_out + cg * 8 + c], xv, acc[c]);
  }
  T out[8];
#pragma unroll
  for (int c = 0; c < 8; ++c) out[c] = from_f<T>(acc[c] > 0.f ? acc[c] : acc[c] * slope);
  T* dst = y + (size_t)m * g.c_out + cg * 8;
  if constexpr (sizeof(T) == 2) {
    *reinterpret_cast<u32x4*>(dst) = *reinterpret_cast<const u32x4*>(out);
  } else {
    *reinterpret_cast<u32x4*>(dst) = *reinterpret_cast<const u32x4*>(out);
    *reinterpret_cast<u32x4*>(dst + 4) = *reinterpret_cast<const u32x4*>(out + 4);
  }
}

// first layer, weight + bias gradient, stage 1: lane = (channel, row chunk); a chunk is RC consecutive output rows of one item.
// partial[block][(K+1)*c_out]: [tap][co] then the bias sums.
constexpr int RC = 16;
template <typename T, int KMAX>
__global__ __launch_bounds__(kThreads) void first_wgrad(const float* __restrict__ x, const T* __restrict__ dy, float* __restrict__ partial,
                                                        FirstGeom g, int j_lo) {
  __shared__ float red[kThreads];
  const int lanes_per_chunk = g.c_out;                       // 16 or 32
  const int chunks_per_block = kThreads / lanes_per_chunk;
  const int co = threadIdx.x % lanes_per_chunk, cl = threadIdx.x / lanes_per_chunk;
  const int chunks_per_item = (g.R1 + RC - 1) / RC;
  const long n_chunks = (long)(g.n * g.p - j_lo) * chunks_per_item;
  const long chunk = (long)blockIdx.x * chunks_per_block + cl;
  float acc[KMAX + 1];
#pragma unroll
  for (int i = 0; i <= KMAX; ++i) acc[i] = 0.f;
  if (chunk < n_chunks) {
    const int j = j_lo + (int)(chunk / chunks_per_item);
    const int r_lo = (int)(chunk % chunks_per_item) * RC;
    const int r_hi = r_lo + RC < g.R1 ? r_lo + RC : g.R1;
    const int ni = j / g.p, wcol = j - ni * g.p;
    const T* d = dy + ((size_t)j * g.R1) * g.c_out + co;
#pragma unroll 4
    for (int r1 = r_lo; r1 < r_hi; ++r1) {
      const float dv = to_f(d[(size_t)r1 * g.c_out]);
      acc[KMAX] += dv;
#pragma unroll
      for (int tap = 0; tap < KMAX; ++tap)
        if (tap < g.k) acc[tap] = fmaf(dv, fold_x(x, g, ni, wcol, r1 * g.s1 + tap - g.pad), acc[tap]);
    }
  }
  // sum the block's chunks per (tap, channel) in a fixed order
  float* out = partial + (size_t)blockIdx.x * (size_t)((g.k + 1) * g.c_out);
#pragma unroll
  for (int i = 0; i <= KMAX; ++i) {
    if (i < g.k || i == KMAX) {
      __syncthreads();
      red[threadIdx.x] = acc[i];
      __syncthreads();
      if (threadIdx.x < lanes_per_chunk) {
        float s = 0.f;
        for (int c = 0; c < chunks_per_block; ++c) s += red[c * lanes_per_chunk + threadIdx.x];
        out[(i == KMAX ? g.k : i) * g.c_out + threadIdx.x] = s;
      }
    }
  }
}

// stage 2 of every reduction in this file: out[map(i)] (+)= sum_b partial[b][i], fixed order.
// map(i) = (i / n_in) * s_out + (i % n_in) * s_in for i < n_main; the following n_tail elements go to `tail` densely.
// Workgroup = 32 outputs x 8 slices of the block range (slice g sums blocks g, g+8, ...; the 8 slice sums are then added in
// slice order): 8x the parallelism of one thread per output, same result for every launch.
__global__ __launch_bounds__(256) void reduce_final(const float* __restrict__ partial, int blocks, int n_main, int n_tail, int n_in,
                                                    long s_in, long s_out, float* __restrict__ out, float* __restrict__ tail, int accumulate) {
  __shared__ float red[8][32];
  const int ii = threadIdx.x & 31, g = threadIdx.x >> 5;
  const int i = blockIdx.x * 32 + ii;
  const int n = n_main + n_tail;
  float s = 0.f;
  if (i < n) {
    int b = g;
    for (; b + 24 < blocks; b += 32) {                        // 4 independent loads in flight
      const float v0 = partial[(size_t)b * n + i], v1 = partial[(size_t)(b + 8) * n + i];
      const float v2 = partial[(size_t)(b + 16) * n + i], v3 = partial[(size_t)(b + 24) * n + i];
      s += v0; s += v1; s += v2; s += v3;
    }
    for (; b < blocks; b += 8) s += partial[(size_t)b * n + i];
  }
  red[g][ii] = s;
  __syncthreads();
  if (g != 0 || i >= n) return;
  float t = red[0][ii];
#pragma unroll
  for (int q = 1; q < 8; ++q) t += red[q][ii];
  float* dst = i < n_main ? out + (long)(i / n_in) * s_out + (long)(i % n_in) * s_in : (tail ? tail + (i - n_main) : nullptr);
  if (!dst) return;
  *dst = accumulate ? *dst + t : t;
}

// first layer, data gradient wrt the waveform (generator step): thread = one sample of one item in [n_lo, n)
template <typename T>
__global__ __launch_bounds__(kThreads) void first_dgrad(const T* __restrict__ dy, const T* __restrict__ w, float* __restrict__ dx,
                                                        FirstGeom g, int n_lo, int accumulate) {
  __shared__ float ws[16 * 32];
  const int kc = g.k * g.c_out;
  for (int i = threadIdx.x; i < kc; i += kThreads) ws[i] = to_f(w[(size_t)i * 8]);
  __syncthreads();
  const long total = (long)(g.n - n_lo) * g.T;
  const long idx = (long)blockIdx.x * kThreads + threadIdx.x;
  if (idx >= total) return;
  // lanes run along the folded rows of one column (coalesced dy reads): idx -> (item, column w, row r)
  const int per_item = g.T;
  const int ni = n_lo + (int)(idx / per_item);
  const int q = (int)(idx % per_item);
  // columns w < (T % p or p) have one row more than the others inside [0, T): enumerate samples column-major
  const int full_rows = g.T / g.p, rem = g.T - full_rows * g.p;       // columns [0, rem) hold full_rows + 1 samples
  int wcol, r;
  if (q < rem * (full_rows + 1)) { wcol = q / (full_rows + 1); r = q - wcol * (full_rows + 1); }
  else { const int q2 = q - rem * (full_rows + 1); wcol = rem + q2 / full_rows; r = q2 - (q2 / full_rows) * full_rows; }
  const int s = r * g.p + wcol;
  auto grad_at = [&](int rr, int wc) -> float {               // d/d(folded element (ni*p + wc, rr))
    float a = 0.f;
    const T* d = dy + ((size_t)(ni * g.p + wc) * g.R1) * g.c_out;
    for (int tap = 0; tap < g.k; ++tap) {
      const int num = rr + g.pad - tap;
      if (num < 0) continue;
      const int r1 = num / g.s1;
      if (r1 * g.s1 != num || r1 >= g.R1) continue;
      const T* row = d + (size_t)r1 * g.c_out;
      for (int c = 0; c < g.c_out; c += 8) {
        union { u32x4 u[2]; T e[8]; } v;
        if constexpr (sizeof(T) == 2) { v.u[0] = *reinterpret_cast<const u32x4*>(row + c); }
        else { v.u[0] = *reinterpret_cast<const u32x4*>(row + c); v.u[1] = *reinterpret_cast<const u32x4*>(row + c + 4); }
#pragma unroll
        for (int e = 0; e < 8; ++e) a = fmaf(to_f(v.e[e]), ws[tap * g.c_out + c + e], a);
      }
    }
    return a;
  };
  float v = grad_at(r, wcol);
  const int sm = 2 * (g.T - 1) - s;                            // the reflect-padded sample that mirrors this one, if any
  if (sm >= g.T && sm < g.R0 * g.p && sm != s) v += grad_at(sm / g.p, sm % g.p);
  float* dst = dx + (size_t)(ni - n_lo) * g.T + s;
  *dst = accumulate ? *dst + v : v;
}

// ---------------------------------------------------------------------------------------------------------------------
// conv_post forward: wave = 4 consecutive output rows; lane = 16 bytes of the channel axis per step
template <typename T>
__global__ __launch_bounds__(kThreads) void post_fwd(const T* __restrict__ h, const T* __restrict__ w, const float* __restrict__ bias,
                                                     T* __restrict__ y8, int J, int R, int c_in, int k, int pad) {
  constexpr int V = 16 / sizeof(T);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long M = (long)J * R;
  const long m0 = ((long)blockIdx.x * 4 + wave) * 4;
  float acc[4] = {0.f, 0.f, 0.f, 0.f};
  for (int tap = 0; tap < k; ++tap) {
    const T* wt = w + (size_t)tap * 8 * c_in;                   // output channel 0 of tap `tap`
    for (int c = lane * V; c < c_in; c += 64 * V) {
      union { u32x4 u; T e[V]; } wv;
      wv.u = *reinterpret_cast<const u32x4*>(wt + c);
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const long m = m0 + q;
        if (m >= M) continue;
        const int t = (int)(m % R) + tap - pad;
        if (t < 0 || t >= R) continue;
        union { u32x4 u; T e[V]; } xv;
        xv.u = *reinterpret_cast<const u32x4*>(h + (size_t)(m + tap - pad) * c_in + c);
#pragma unroll
        for (int e = 0; e < V; ++e) acc[q] = fmaf(to_f(wv.e[e]), to_f(xv.e[e]), acc[q]);
      }
    }
  }
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    float v = acc[q];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
    const long m = m0 + q;
    if (lane == 0 && m < M) {
      T out[8];
      out[0] = from_f<T>(v + (bias ? bias[0] : 0.f));
#pragma unroll
      for (int c = 1; c < 8; ++c) out[c] = from_f<T>(0.f);
      T* dst = y8 + (size_t)m * 8;
      *reinterpret_cast<u32x4*>(dst) = *reinterpret_cast<const u32x4*>(out);
      if constexpr (sizeof(T) == 4) *reinterpret_cast<u32x4*>(dst + 4) = *reinterpret_cast<const u32x4*>(out + 4);
    }
  }
}

// conv_post data gradient: dh[m][c] = (sum_tap dy8[m - tap + pad][0] * w[tap][0][c] + res[m][c]) * lrelu'(h[m][c]);
// thread = one row x 16 bytes of channels; rows enumerate items [j_lo, J) only
template <typename T>
__global__ __launch_bounds__(kThreads) void post_dgrad(const T* __restrict__ dy8, const T* __restrict__ w, const T* __restrict__ res,
                                                       const T* __restrict__ h, T* __restrict__ dh, int J, int R, int c_in, int k, int pad,
                                                       int j_lo, float slope) {
  constexpr int V = 16 / sizeof(T);
  const int vpr = c_in / V;
  const long total = (long)(J - j_lo) * R * vpr;
  const long idx = (long)blockIdx.x * kThreads + threadIdx.x;
  if (idx >= total) return;
  const int vc = (int)(idx % vpr);
  const long m = (long)j_lo * R + idx / vpr;
  const int t = (int)(m % R);
  float acc[V];
#pragma unroll
  for (int e = 0; e < V; ++e) acc[e] = 0.f;
  for (int tap = 0; tap < k; ++tap) {
    const int ty = t - tap + pad;
    if (ty < 0 || ty >= R) continue;
    const float d = to_f(dy8[(size_t)(m - tap + pad) * 8]);
    union { u32x4 u; T e[V]; } wv;
    wv.u = *reinterpret_cast<const u32x4*>(w + (size_t)tap * 8 * c_in + vc * V);
#pragma unroll
    for (int e = 0; e < V; ++e) acc[e] = fmaf(d, to_f(wv.e[e]), acc[e]);
  }
  const size_t o = (size_t)m * c_in + vc * V;
  union { u32x4 u; T e[V]; } hv, rv, out;
  hv.u = *reinterpret_cast<const u32x4*>(h + o);
  if (res) rv.u = *reinterpret_cast<const u32x4*>(res + o);
#pragma unroll
  for (int e = 0; e < V; ++e) {
    float v = acc[e] + (res ? to_f(rv.e[e]) : 0.f);
    out.e[e] = from_f<T>(to_f(hv.e[e]) > 0.f ? v : v * slope);
  }
  *reinterpret_cast<u32x4*>(dh + o) = out.u;
}

// conv_post weight + bias gradient, stage 1: block = WR consecutive rows of dy; thread = 16 bytes of channels.
// partial[block][k*c_in + 1]
constexpr int WR = 16;
template <typename T, int KMAX>
__global__ __launch_bounds__(128) void post_wgrad(const T* __restrict__ dy8, const T* __restrict__ h, float* __restrict__ partial,
                                                  int J, int R, int c_in, int k, int pad, int j_lo) {
  constexpr int V = 16 / sizeof(T);
  const long M = (long)J * R;
  const long m_lo = (long)j_lo * R + (long)blockIdx.x * WR;
  const long m_hi = m_lo + WR < M ? m_lo + WR : M;
  const int n_out = k * c_in + 1;
  float* out = partial + (size_t)blockIdx.x * n_out;
  for (int c = threadIdx.x * V; c < c_in; c += 128 * V) {
    float acc[KMAX][V];
#pragma unroll
    for (int tap = 0; tap < KMAX; ++tap)
#pragma unroll
      for (int e = 0; e < V; ++e) acc[tap][e] = 0.f;
    for (long m = m_lo; m < m_hi; ++m) {
      const float d = to_f(dy8[(size_t)m * 8]);
      const int t = (int)(m % R);
#pragma unroll
      for (int tap = 0; tap < KMAX; ++tap) {
        if (tap >= k) continue;
        const int ti = t + tap - pad;
        if (ti < 0 || ti >= R) continue;
        union { u32x4 u; T e[V]; } xv;
        xv.u = *reinterpret_cast<const u32x4*>(h + (size_t)(m + tap - pad) * c_in + c);
#pragma unroll
        for (int e = 0; e < V; ++e) acc[tap][e] = fmaf(d, to_f(xv.e[e]), acc[tap][e]);
      }
    }
#pragma unroll
    for (int tap = 0; tap < KMAX; ++tap)
      if (tap < k)
#pragma unroll
        for (int e = 0; e < V; ++e) out[(size_t)tap * c_in + c + e] = acc[tap][e];
  }
  if (threadIdx.x == 0) {
    float s = 0.f;
    for (long m = m_lo; m < m_hi; ++m) s += to_f(dy8[(size_t)m * 8]);
    out[(size_t)k * c_in] = s;
  }
}

bool geom_ok(const FirstGeom& g) {
  return g.n > 0 && g.T > 1 && g.p >= 1 && g.p <= g.T && g.k >= 1 && g.k <= 16 && g.s1 >= 1 && g.pad >= 0 && g.c_out % 8 == 0 &&
         g.c_out >= 8 && g.c_out <= 32 && g.k * g.c_out <= 16 * 32;
}

FirstGeom make_geom(int n, int T, int p, int k, int s1, int pad, int c_out) {
  FirstGeom g{n, T, p, 0, 0, k, s1, pad, c_out};
  g.R0 = (T + p - 1) / p;
  g.R1 = (g.R0 + 2 * pad - k) / s1 + 1;
  return g;
}

}  // namespace

extern "C" {

int vits_disc_first_rows(int T, int p, int k, int s1, int pad) {
  if (T <= 0 || p <= 0 || k <= 0 || s1 <= 0 || pad < 0) return VITS_E_BADARG;
  return ((T + p - 1) / p + 2 * pad - k) / s1 + 1;
}

int vits_disc_first_fwd(int dtype, const float* x, const void* w, const float* bias, void* y, int n, int T, int p, int k, int s1,
                        int pad, int c_out, float slope, void* stream) {
  if (!x || !w || !y) return VITS_E_BADARG;
  const FirstGeom g = make_geom(n, T, p, k, s1, pad, c_out);
  if (!geom_ok(g) || g.R1 <= 0) return VITS_E_UNSUPPORTED;
  const long total = (long)n * p * g.R1 * (c_out / 8);
  const unsigned blocks = (unsigned)((total + kThreads - 1) / kThreads);
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (dtype == VITS_DT_BF16)
    hipLaunchKernelGGL(first_fwd<__bf16>, dim3(blocks), dim3(kThreads), 0, s, x, static_cast<const __bf16*>(w), bias, static_cast<__bf16*>(y), g, slope);
  else if (dtype == VITS_DT_F32)
    hipLaunchKernelGGL(first_fwd<float>, dim3(blocks), dim3(kThreads), 0, s, x, static_cast<const float*>(w), bias, static_cast<float*>(y), g, slope);
  else
    return VITS_E_UNSUPPORTED;
  return vits::check_launch("vits_disc_first_fwd");
}

size_t vits_disc_first_wgrad_workspace(int n, int T, int p, int k, int s1, int pad, int c_out) {
  const FirstGeom g = make_geom(n, T, p, k, s1, pad, c_out);
  if (!geom_ok(g) || g.R1 <= 0) return 0;
  const long chunks = (long)n * p * ((g.R1 + RC - 1) / RC);
  const long blocks = (chunks + kThreads / c_out - 1) / (kThreads / c_out);
  return (size_t)blocks * (size_t)((k + 1) * c_out) * sizeof(float);
}

/* dw: the arena's fp32 [k][c_out][8] (element [tap][co][0] written, the rest untouched); dbias float32[c_out].
 * Items n_lo.. only (the generated half in the generator step passes n_lo = n/2 and no weight gradient is taken there,
 * so n_lo is 0 in practice; kept for symmetry). */
int vits_disc_first_wgrad(int dtype, const float* x, const void* dy, float* dw, float* dbias, void* workspace, size_t workspace_bytes,
                          int n, int T, int p, int k, int s1, int pad, int c_out, int accumulate, void* stream) {
  if (!x || !dy || !dw || !workspace) return VITS_E_BADARG;
  const FirstGeom g = make_geom(n, T, p, k, s1, pad, c_out);
  if (!geom_ok(g) || g.R1 <= 0) return VITS_E_UNSUPPORTED;
  if (workspace_bytes < vits_disc_first_wgrad_workspace(n, T, p, k, s1, pad, c_out)) return VITS_E_BADARG;
  const long chunks = (long)n * p * ((g.R1 + RC - 1) / RC);
  const int cpb = kThreads / c_out;
  const unsigned blocks = (unsigned)((chunks + cpb - 1) / cpb);
  hipStream_t s = static_cast<hipStream_t>(stream);
  float* ws = static_cast<float*>(workspace);
  if (dtype == VITS_DT_BF16)
    hipLaunchKernelGGL((first_wgrad<__bf16, 16>), dim3(blocks), dim3(kThreads), 0, s, x, static_cast<const __bf16*>(dy), ws, g, 0);
  else if (dtype == VITS_DT_F32)
    hipLaunchKernelGGL((first_wgrad<float, 16>), dim3(blocks), dim3(kThreads), 0, s, x, static_cast<const float*>(dy), ws, g, 0);
  else
    return VITS_E_UNSUPPORTED;
  const int n_main = k * c_out, n_tail = c_out;
  hipLaunchKernelGGL(reduce_final, dim3((n_main + n_tail + 31) / 32), dim3(256), 0, s, ws, (int)blocks, n_main, n_tail, n_main, 8L, 0L,
                     dw, dbias, accumulate);
  return vits::check_launch("vits_disc_first_wgrad");
}

/* dx float32 [n - n_lo][T]: gradient wrt the waveforms of items n_lo..n-1 (dy, w as in the forward). */
int vits_disc_first_dgrad(int dtype, const void* dy, const void* w, float* dx, int n, int n_lo, int T, int p, int k, int s1, int pad,
                          int c_out, int accumulate, void* stream) {
  if (!dy || !w || !dx || n_lo < 0 || n_lo >= n) return VITS_E_BADARG;
  const FirstGeom g = make_geom(n, T, p, k, s1, pad, c_out);
  if (!geom_ok(g) || g.R1 <= 0) return VITS_E_UNSUPPORTED;
  const long total = (long)(n - n_lo) * T;
  const unsigned blocks = (unsigned)((total + kThreads - 1) / kThreads);
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (dtype == VITS_DT_BF16)
    hipLaunchKernelGGL(first_dgrad<__bf16>, dim3(blocks), dim3(kThreads), 0, s, static_cast<const __bf16*>(dy), static_cast<const __bf16*>(w), dx, g, n_lo, accumulate);
  else if (dtype == VITS_DT_F32)
    hipLaunchKernelGGL(first_dgrad<float>, dim3(blocks), dim3(kThreads), 0, s, static_cast<const float*>(dy), static_cast<const float*>(w), dx, g, n_lo, accumulate);
  else
    return VITS_E_UNSUPPORTED;
  return vits::check_launch("vits_disc_first_dgrad");
}

int vits_disc_post_fwd(int dtype, const void* h, const void* w, const float* bias, void* y8, int J, int R, int c_in, int k, int pad,
                       void* stream) {
  if (!h || !w || !y8 || J <= 0 || R <= 0) return VITS_E_BADARG;
  const int V = dtype == VITS_DT_BF16 ? 8 : 4;
  if ((dtype != VITS_DT_BF16 && dtype != VITS_DT_F32) || c_in % V != 0 || k < 1 || k > 4 || pad < 0) return VITS_E_UNSUPPORTED;
  const long M = (long)J * R;
  const unsigned blocks = (unsigned)((M + 15) / 16);
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (dtype == VITS_DT_BF16)
    hipLaunchKernelGGL(post_fwd<__bf16>, dim3(blocks), dim3(kThreads), 0, s, static_cast<const __bf16*>(h), static_cast<const __bf16*>(w), bias,
                       static_cast<__bf16*>(y8), J, R, c_in, k, pad);
  else
    hipLaunchKernelGGL(post_fwd<float>, dim3(blocks), dim3(kThreads), 0, s, static_cast<const float*>(h), static_cast<const float*>(w), bias,
                       static_cast<float*>(y8), J, R, c_in, k, pad);
  return vits::check_launch("vits_disc_post_fwd");
}

/* dh rows of items j_lo..J-1 (same indexing as h; rows of earlier items are not touched). */
int vits_disc_post_dgrad(int dtype, const void* dy8, const void* w, const void* res, const void* h, void* dh, int J, int R, int c_in,
                         int k, int pad, int j_lo, float slope, void* stream) {
  if (!dy8 || !w || !h || !dh || J <= 0 || R <= 0 || j_lo < 0 || j_lo >= J) return VITS_E_BADARG;
  const int V = dtype == VITS_DT_BF16 ? 8 : 4;
  if ((dtype != VITS_DT_BF16 && dtype != VITS_DT_F32) || c_in % V != 0 || k < 1 || k > 4 || pad < 0) return VITS_E_UNSUPPORTED;
  const long total = (long)(J - j_lo) * R * (c_in / V);
  const unsigned blocks = (unsigned)((total + kThreads - 1) / kThreads);
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (dtype == VITS_DT_BF16)
    hipLaunchKernelGGL(post_dgrad<__bf16>, dim3(blocks), dim3(kThreads), 0, s, static_cast<const __bf16*>(dy8), static_cast<const __bf16*>(w),
                       static_cast<const __bf16*>(res), static_cast<const __bf16*>(h), static_cast<__bf16*>(dh), J, R, c_in, k, pad, j_lo, slope);
  else
    hipLaunchKernelGGL(post_dgrad<float>, dim3(blocks), dim3(kThreads), 0, s, static_cast<const float*>(dy8), static_cast<const float*>(w),
                       static_cast<const float*>(res), static_cast<const float*>(h), static_cast<float*>(dh), J, R, c_in, k, pad, j_lo, slope);
  return vits::check_launch("vits_disc_post_dgrad");
}

size_t vits_disc_post_wgrad_workspace(int J, int R, int c_in, int k) {
  const long M = (long)J * R;
  return (size_t)((M + WR - 1) / WR) * (size_t)(k * c_in + 1) * sizeof(float);
}

/* dw: the arena's fp32 [k][8][c_in] (row [tap][0][:] written); dbias float32[1]. */
int vits_disc_post_wgrad(int dtype, const void* dy8, const void* h, float* dw, float* dbias, void* workspace, size_t workspace_bytes,
                         int J, int R, int c_in, int k, int pad, int accumulate, void* stream) {
  if (!dy8 || !h || !dw || !workspace || J <= 0 || R <= 0) return VITS_E_BADARG;
  const int V = dtype == VITS_DT_BF16 ? 8 : 4;
  if ((dtype != VITS_DT_BF16 && dtype != VITS_DT_F32) || c_in % V != 0 || k < 1 || k > 4 || pad < 0) return VITS_E_UNSUPPORTED;
  if (workspace_bytes < vits_disc_post_wgrad_workspace(J, R, c_in, k)) return VITS_E_BADARG;
  const long M = (long)J * R;
  const unsigned blocks = (unsigned)((M + WR - 1) / WR);
  hipStream_t s = static_cast<hipStream_t>(stream);
  float* ws = static_cast<float*>(workspace);
  if (dtype == VITS_DT_BF16)
    hipLaunchKernelGGL((post_wgrad<__bf16, 4>), dim3(blocks), dim3(128), 0, s, static_cast<const __bf16*>(dy8), static_cast<const __bf16*>(h), ws, J, R, c_in, k, pad, 0);
  else
    hipLaunchKernelGGL((post_wgrad<float, 4>), dim3(blocks), dim3(128), 0, s, static_cast<const float*>(dy8), static_cast<const float*>(h), ws, J, R, c_in, k, pad, 0);
  const int n_main = k * c_in;
  hipLaunchKernelGGL(reduce_final, dim3((n_main + 1 + 31) / 32), dim3(256), 0, s, ws, (int)blocks, n_main, 1, c_in, 1L, 8L * c_in, dw, dbias, accumulate);
  return vits::check_launch("vits_disc_post_wgrad");
}

}  // extern "C"
