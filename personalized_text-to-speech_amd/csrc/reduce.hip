// Deterministic two-stage reductions for the loss terms of the fine-tune step (reference losses.py:7-61 and the
// [1,2]-sums of the duration predictor, models.py:71-102).
//
// Why not torch.sum / torch.mean: their multi-block path zeroes a semaphore with hipMemsetAsync before every launch, and
// device memset nodes are the one node type the captured step cannot rely on.  These kernels need no
// zero-initialised memory: stage 1 writes one partial per workgroup, stage 2 sums the partials in a fixed order.
//   vits_absdiff_sum   out (+)= scale * sum |a - b|           (feature-matching loss: a = real half, b = generated half)
//   vits_absdiff_bwd   db = -sign(a - b) * scale * g,  da = 0 (the reference detaches the real half, losses.py:11)
//   vits_segsum_f32    out[s] = sum x[s][:]                    (per-item sums, whole-tensor sums with n_seg = 1)
#include "common.h"

namespace {

constexpr int kThreads = 256;
constexpr int kMaxSplits = 256;

__device__ __forceinline__ float block_sum(float v, float* red) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) red[wave] = v;
  __syncthreads();
  float s = 0.f;
  for (int w = 0; w < (int)(blockDim.x >> 6); ++w) s += red[w];
  return s;
}

template <typename T> __device__ __forceinline__ float ld(const T* p, size_t i);
template <> __device__ __forceinline__ float ld<float>(const float* p, size_t i) { return p[i]; }
template <> __device__ __forceinline__ float ld<__bf16>(const __bf16* p, size_t i) { return (float)p[i]; }

// stage 1: partial[seg][split] = sum over the split's strided slice of f(seg, i)
template <typename T, bool ABSDIFF>
__global__ __launch_bounds__(kThreads) void partial_kernel(const T* __restrict__ a, const T* __restrict__ b, size_t seg_len,
                                                          float* __restrict__ partial) {
  __shared__ float red[4];
  const size_t base = (size_t)blockIdx.y * seg_len;
  float acc = 0.f;
#pragma unroll 4
  for (size_t i = (size_t)blockIdx.x * kThreads + threadIdx.x; i < seg_len; i += (size_t)gridDim.x * kThreads) {
    const float x = ld<T>(a, base + i);
    acc += ABSDIFF ? fabsf(x - ld<T>(b, base + i)) : x;
  }
  acc = block_sum(acc, red);
  if (threadIdx.x == 0) partial[(size_t)blockIdx.y * gridDim.x + blockIdx.x] = acc;
}

// stage 2: out[seg] (+)= scale * sum_s partial[seg][s]   (one wave per segment, fixed order)
__global__ __launch_bounds__(64) void final_kernel(const float* __restrict__ partial, int splits, float scale, float* __restrict__ out,
                                                   int accumulate) {
  const int seg = blockIdx.x;
  float acc = 0.f;
  for (int s = threadIdx.x; s < splits; s += 64) acc += partial[(size_t)seg * splits + s];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
  if (threadIdx.x == 0) out[seg] = (accumulate ? out[seg] : 0.f) + scale * acc;
}

template <typename T>
__global__ __launch_bounds__(kThreads) void absdiff_bwd_kernel(const T* __restrict__ a, const T* __restrict__ b, size_t n,
                                                              const float* __restrict__ g, float scale, T* __restrict__ da,
                                                              T* __restrict__ db) {
  const float gs = g[0] * scale;
  for (size_t i = (size_t)blockIdx.x * kThreads + threadIdx.x; i < n; i += (size_t)gridDim.x * kThreads) {
    const float d = ld<T>(a, i) - ld<T>(b, i);
    const float v = d > 0.f ? -gs : (d < 0.f ? gs : 0.f);          // d|a-b|/db = -sign(a-b)
    db[i] = (T)v;
    if (da) da[i] = (T)0.f;
  }
}

// dy' = dy * (y > 0 ? 1 : slope) (y optional) with rows >= lengths[b] zeroed (lengths optional): the chain rule of a fused output
// leaky-relu and of an output mask in ONE launch (replaces gt + where + 2 scalar fills + mul + cast, and arange + lt + mul)
template <typename T>
__global__ __launch_bounds__(kThreads) void lrelu_mask_bwd_kernel(const T* __restrict__ dy, const T* __restrict__ y, float slope,
                                                                 const int* __restrict__ lengths, int t, int c, size_t n,
                                                                 T* __restrict__ out) {
  for (size_t i = (size_t)blockIdx.x * kThreads + threadIdx.x; i < n; i += (size_t)gridDim.x * kThreads) {
    float v = ld<T>(dy, i);
    if (y && !(ld<T>(y, i) > 0.f)) v *= slope;
    if (lengths) {
      const size_t row = i / c;
      const int b = (int)(row / t), tt = (int)(row - (size_t)b * t);
      if (tt >= lengths[b]) v = 0.f;
    }
    out[i] = (T)v;
  }
}

// column sums of a channels-last tensor: partial[(seg*S + s)*C + c] = sum over the split's rows of x[seg][row][c]
// block = 64 columns x 4 row lanes; coalesced 128/256-byte row segments
template <typename T>
__global__ __launch_bounds__(kThreads) void colsum_kernel(const T* __restrict__ x, int rows, int C, float* __restrict__ dst) {
  __shared__ float red[4][64];
  const int col = blockIdx.x * 64 + (threadIdx.x & 63), rl = threadIdx.x >> 6;
  const int S = gridDim.y, s = blockIdx.y, seg = blockIdx.z;
  const int per = (rows + S - 1) / S, r0 = s * per, r1 = (r0 + per < rows) ? r0 + per : rows;
  const T* base = x + (size_t)seg * rows * C;
  float acc = 0.f;
  if (col < C) {
#pragma unroll 8
    for (int r = r0 + rl; r < r1; r += 4) acc += ld<T>(base, (size_t)r * C + col);
  }
  red[rl][threadIdx.x & 63] = acc;
  __syncthreads();
  if (rl == 0 && col < C) dst[((size_t)seg * S + s) * C + col] = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
}

// out[seg][c] = sum_s partial[(seg*S + s)*C + c]
__global__ __launch_bounds__(kThreads) void colsum_final_kernel(const float* __restrict__ partial, int S, int C, int n_seg, float* __restrict__ out) {
  const int i = blockIdx.x * kThreads + threadIdx.x;
  if (i >= n_seg * C) return;
  const int seg = i / C, c = i - seg * C;
  float acc = 0.f;
  for (int s = 0; s < S; ++s) acc += partial[((size_t)seg * S + s) * C + c];
  out[i] = acc;
}

int colsum_splits(int n_seg, int rows, int C) {
  const int wgs = n_seg * ((C + 63) / 64);
  int S = (512 + wgs - 1) / wgs;
  const int max_s = rows / 64 > 0 ? rows / 64 : 1;
  if (S > max_s) S = max_s;
  if (S > 64) S = 64;
  return S < 1 ? 1 : S;
}

// ---- bf16, 8 elements (16 bytes) per thread per step: the element-wise passes over the discriminators' feature maps are
// instruction-bound with 2-byte loads
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
union Bf8 { u32x4 u; __bf16 e[8]; };

__global__ __launch_bounds__(kThreads) void partial_absdiff_bf16x8(const u32x4* __restrict__ a, const u32x4* __restrict__ b, size_t nvec,
                                                                  float* __restrict__ partial) {
  __shared__ float red[4];
  float acc = 0.f;
#pragma unroll 2
  for (size_t i = (size_t)blockIdx.x * kThreads + threadIdx.x; i < nvec; i += (size_t)gridDim.x * kThreads) {
    Bf8 x, y;
    x.u = a[i]; y.u = b[i];
#pragma unroll
    for (int j = 0; j < 8; ++j) acc += fabsf((float)x.e[j] - (float)y.e[j]);
  }
  acc = block_sum(acc, red);
  if (threadIdx.x == 0) partial[blockIdx.x] = acc;
}

__global__ __launch_bounds__(kThreads) void absdiff_bwd_bf16x8(const u32x4* __restrict__ a, const u32x4* __restrict__ b, size_t nvec,
                                                              const float* __restrict__ g, float scale, u32x4* __restrict__ da,
                                                              u32x4* __restrict__ db) {
  const float gs = g[0] * scale;
  for (size_t i = (size_t)blockIdx.x * kThreads + threadIdx.x; i < nvec; i += (size_t)gridDim.x * kThreads) {
    Bf8 x, y, o;
    x.u = a[i]; y.u = b[i];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float d = (float)x.e[j] - (float)y.e[j];
      o.e[j] = (__bf16)(d > 0.f ? -gs : (d < 0.f ? gs : 0.f));
    }
    db[i] = o.u;
    if (da) da[i] = u32x4{0u, 0u, 0u, 0u};
  }
}

__global__ __launch_bounds__(kThreads) void lrelu_mask_bwd_bf16x8(const u32x4* __restrict__ dy, const u32x4* __restrict__ y, float slope,
                                                                 const int* __restrict__ lengths, int t, int cvec, size_t nvec,
                                                                 u32x4* __restrict__ out) {
  for (size_t i = (size_t)blockIdx.x * kThreads + threadIdx.x; i < nvec; i += (size_t)gridDim.x * kThreads) {
    Bf8 v, o;
    v.u = dy[i];
    bool dead = false;
    if (lengths) {                                       // all 8 elements of a vector share their row (c % 8 == 0)
      const size_t row = i / cvec;
      const int b = (int)(row / t), tt = (int)(row - (size_t)b * t);
      dead = tt >= lengths[b];
    }
    if (y) {
      Bf8 g;
      g.u = y[i];
#pragma unroll
      for (int j = 0; j < 8; ++j) o.e[j] = dead ? (__bf16)0.f : ((float)g.e[j] > 0.f ? v.e[j] : (__bf16)((float)v.e[j] * slope));
    } else {
#pragma unroll
      for (int j = 0; j < 8; ++j) o.e[j] = dead ? (__bf16)0.f : v.e[j];
    }
    out[i] = o.u;
  }
}

__host__ inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

int pick_splits(size_t seg_len, int n_seg) {
  long s = (long)((seg_len + (size_t)kThreads * 8 - 1) / ((size_t)kThreads * 8));     // >= 8 elements per thread
  const long fill = (1024 + n_seg - 1) / n_seg;                                      // ~4 workgroups per CU in total
  if (s > fill) s = fill;
  if (s > kMaxSplits) s = kMaxSplits;
  if (s < 1) s = 1;
  return (int)s;
}

}  // namespace

namespace {

// ---- feature-matching loss over ALL feature maps of all discriminators in one pass (losses.py:7-15) ----------------------------
// entry e: h [2 n_e] (first half real, second half generated), scale_e = 2 / (elements of one half without padding channels).
constexpr int kFeatMax = 48, kFeatSplits = 64;
struct FeatTable { const void* h[kFeatMax]; void* dh[kFeatMax]; unsigned long long n[kFeatMax]; float scale[kFeatMax]; int count; };

// grid (kFeatSplits, entries): partial[e * kFeatSplits + split] = scale_e * sum over the split's slice of |a - b|
template <typename T>
__global__ __launch_bounds__(kThreads) void feat_partial_kernel(FeatTable tab, float* __restrict__ partial) {
  __shared__ float red[4];
  const int e = blockIdx.y;
  const size_t n = tab.n[e];
  const T* a = static_cast<const T*>(tab.h[e]);
  const T* b = a + n;
  float acc = 0.f;
  if (sizeof(T) == 2 && n % 8 == 0 && (reinterpret_cast<uintptr_t>(a) & 15) == 0) {
    const u32x4* av = reinterpret_cast<const u32x4*>(a);
    const u32x4* bv = reinterpret_cast<const u32x4*>(b);
    const size_t nvec = n / 8;
#pragma unroll 2
    for (size_t i = (size_t)blockIdx.x * kThreads + threadIdx.x; i < nvec; i += (size_t)kFeatSplits * kThreads) {
      Bf8 x, y;
      x.u = av[i]; y.u = bv[i];
#pragma unroll
      for (int j = 0; j < 8; ++j) acc += fabsf((float)x.e[j] - (float)y.e[j]);
    }
  } else {
    for (size_t i = (size_t)blockIdx.x * kThreads + threadIdx.x; i < n; i += (size_t)kFeatSplits * kThreads) acc += fabsf(ld<T>(a, i) - ld<T>(b, i));
  }
  acc = block_sum(acc, red);
  if (threadIdx.x == 0) partial[(size_t)e * kFeatSplits + blockIdx.x] = acc * tab.scale[e];
}

// out = sum of all partials, entry by entry, split by split (fixed order)
__global__ __launch_bounds__(64) void feat_final_kernel(const float* __restrict__ partial, int count, float* __restrict__ out) {
  float acc = 0.f;
  for (int i = threadIdx.x; i < count * kFeatSplits; i += 64) acc += partial[i];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
  if (threadIdx.x == 0) out[0] = acc;
}

// d(generated half) = -sign(a - b) * scale_e * g, d(real half) = 0 (the reference detaches it, losses.py:11); every element written
template <typename T>
__global__ __launch_bounds__(kThreads) void feat_bwd_kernel(FeatTable tab, const float* __restrict__ g) {
  const int e = blockIdx.y;
  const size_t n = tab.n[e];
  const T* a = static_cast<const T*>(tab.h[e]);
  const T* b = a + n;
  T* da = static_cast<T*>(tab.dh[e]);
  T* db = da + n;
  const float gs = g[0] * tab.scale[e];
  if (sizeof(T) == 2 && n % 8 == 0 && ((reinterpret_cast<uintptr_t>(a) | reinterpret_cast<uintptr_t>(da)) & 15) == 0) {
    const u32x4* av = reinterpret_cast<const u32x4*>(a);
    const u32x4* bv = reinterpret_cast<const u32x4*>(b);
    u32x4* dav = reinterpret_cast<u32x4*>(da);
    u32x4* dbv = reinterpret_cast<u32x4*>(db);
    const size_t nvec = n / 8;
    for (size_t i = (size_t)blockIdx.x * kThreads + threadIdx.x; i < nvec; i += (size_t)gridDim.x * kThreads) {
      Bf8 x, y, o;
      x.u = av[i]; y.u = bv[i];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float d = (float)x.e[j] - (float)y.e[j];
        o.e[j] = (__bf16)(d > 0.f ? -gs : (d < 0.f ? gs : 0.f));
      }
      dbv[i] = o.u;
      dav[i] = u32x4{0u, 0u, 0u, 0u};
    }
  } else {
    for (size_t i = (size_t)blockIdx.x * kThreads + threadIdx.x; i < n; i += (size_t)gridDim.x * kThreads) {
      const float d = ld<T>(a, i) - ld<T>(b, i);
      db[i] = (T)(d > 0.f ? -gs : (d < 0.f ? gs : 0.f));
      da[i] = (T)0.f;
    }
  }
}

int feat_table(const vits_feat_item* items, int n_items, FeatTable& tab, bool need_dh) {
  if (!items || n_items <= 0 || n_items > kFeatMax) return VITS_E_BADARG;
  tab.count = n_items;
  for (int i = 0; i < n_items; ++i) {
    if (!items[i].h || items[i].n == 0 || (need_dh && !items[i].dh)) return VITS_E_BADARG;
    tab.h[i] = items[i].h; tab.dh[i] = items[i].dh; tab.n[i] = items[i].n; tab.scale[i] = items[i].scale;
  }
  return VITS_OK;
}

}  // namespace

extern "C" size_t vits_feature_l1_workspace(int n_items) { return (size_t)n_items * kFeatSplits * sizeof(float); }

extern "C" int vits_feature_l1(int dtype, const vits_feat_item* host_items, int n_items, float* out, void* workspace, size_t workspace_bytes,
                               void* stream) {
  FeatTable tab;
  const int rc = feat_table(host_items, n_items, tab, false);
  if (rc != VITS_OK) return rc;
  if (!out || !workspace || workspace_bytes < vits_feature_l1_workspace(n_items)) return VITS_E_BADARG;
  hipStream_t s = static_cast<hipStream_t>(stream);
  float* partial = static_cast<float*>(workspace);
  const dim3 grid(kFeatSplits, n_items);
  if (dtype == VITS_DT_BF16) hipLaunchKernelGGL(feat_partial_kernel<__bf16>, grid, dim3(kThreads), 0, s, tab, partial);
  else if (dtype == VITS_DT_F32) hipLaunchKernelGGL(feat_partial_kernel<float>, grid, dim3(kThreads), 0, s, tab, partial);
  else return VITS_E_UNSUPPORTED;
  hipLaunchKernelGGL(feat_final_kernel, dim3(1), dim3(64), 0, s, partial, n_items, out);
  return vits::check_launch("vits_feature_l1");
}

extern "C" int vits_feature_l1_bwd(int dtype, const vits_feat_item* host_items, int n_items, const float* g, void* stream) {
  FeatTable tab;
  const int rc = feat_table(host_items, n_items, tab, true);
  if (rc != VITS_OK) return rc;
  if (!g) return VITS_E_BADARG;
  hipStream_t s = static_cast<hipStream_t>(stream);
  const dim3 grid(96, n_items);
  if (dtype == VITS_DT_BF16) hipLaunchKernelGGL(feat_bwd_kernel<__bf16>, grid, dim3(kThreads), 0, s, tab, g);
  else if (dtype == VITS_DT_F32) hipLaunchKernelGGL(feat_bwd_kernel<float>, grid, dim3(kThreads), 0, s, tab, g);
  else return VITS_E_UNSUPPORTED;
  return vits::check_launch("vits_feature_l1_bwd");
}

namespace {

// ---- least-squares GAN losses over the logits of all discriminators (losses.py:18-43) -----------------------------------
// item d: y8 [J][R][8], channel 0 live, items j < J/2 real, the rest generated.
//   mode 0 (discriminator_loss): sum_d mean_real (1 - y)^2 + mean_gen y^2       mode 1 (generator_loss): sum_d mean_gen (1 - y)^2
constexpr int kLsganMax = 8, kLsganSplits = 16;
struct LsganTable { const void* y8[kLsganMax]; void* dy8[kLsganMax]; int J[kLsganMax], R[kLsganMax]; int count; };

// grid (kLsganSplits, 2 halves, items): partial[(d*2 + half) * kLsganSplits + split]
template <typename T>
__global__ __launch_bounds__(kThreads) void lsgan_partial_kernel(LsganTable tab, int mode, float* __restrict__ partial) {
  __shared__ float red[4];
  const int d = blockIdx.z, half = blockIdx.y;
  const size_t n = (size_t)(tab.J[d] / 2) * tab.R[d];
  const T* y = static_cast<const T*>(tab.y8[d]) + (size_t)half * n * 8;
  float acc = 0.f;
  if (!(mode == 1 && half == 0)) {
    const bool one_minus = (mode == 1) || half == 0;
    for (size_t i = (size_t)blockIdx.x * kThreads + threadIdx.x; i < n; i += (size_t)kLsganSplits * kThreads) {
      const float v = ld<T>(y, i * 8);
      const float u = one_minus ? 1.0f - v : v;
      acc += u * u;
    }
  }
  acc = block_sum(acc, red);
  if (threadIdx.x == 0) partial[((size_t)d * 2 + half) * kLsganSplits + blockIdx.x] = acc;
}

// out[0] = total, out[1 + 2d + half] = the term of discriminator d (mean over its half); one wave, fixed order
__global__ __launch_bounds__(64) void lsgan_final_kernel(LsganTable tab, const float* __restrict__ partial, float* __restrict__ out,
                                                         float* __restrict__ total_out) {
  if (threadIdx.x != 0) return;
  float total = 0.f;
  for (int d = 0; d < tab.count; ++d)
    for (int half = 0; half < 2; ++half) {
      float s = 0.f;
      for (int k = 0; k < kLsganSplits; ++k) s += partial[((size_t)d * 2 + half) * kLsganSplits + k];
      s /= (float)((size_t)(tab.J[d] / 2) * tab.R[d]);
      out[1 + 2 * d + half] = s;
      total += s;
    }
  out[0] = total;
  if (total_out) *total_out = total;
}

// dy8[j][r][0] = g * 2 (y - 1) / n  (or 2 y / n), channels 1..7 and the halves without a term = 0; every element written
template <typename T>
__global__ __launch_bounds__(kThreads) void lsgan_bwd_kernel(LsganTable tab, int mode, const float* __restrict__ g) {
  const int d = blockIdx.y;
  const size_t n = (size_t)(tab.J[d] / 2) * tab.R[d];
  const T* y = static_cast<const T*>(tab.y8[d]);
  T* dy = static_cast<T*>(tab.dy8[d]);
  const float scale = 2.0f * g[0] / (float)n;
  for (size_t i = (size_t)blockIdx.x * kThreads + threadIdx.x; i < 2 * n; i += (size_t)gridDim.x * kThreads) {
    const int half = i >= n;
    float v = 0.f;
    if (!(mode == 1 && half == 0)) {
      const float yv = ld<T>(y, i * 8);
      v = scale * (((mode == 1) || half == 0) ? yv - 1.0f : yv);
    }
    T row[8];
    row[0] = (T)v;
#pragma unroll
    for (int c = 1; c < 8; ++c) row[c] = (T)0.f;
    T* dst = dy + i * 8;
    if constexpr (sizeof(T) == 2) *reinterpret_cast<uint4*>(dst) = *reinterpret_cast<const uint4*>(row);
    else { *reinterpret_cast<uint4*>(dst) = *reinterpret_cast<const uint4*>(row); *reinterpret_cast<uint4*>(dst + 4) = *reinterpret_cast<const uint4*>(row + 4); }
  }
}

int lsgan_table(const vits_lsgan_item* items, int n_items, LsganTable& tab, bool need_dy) {
  if (!items || n_items <= 0 || n_items > kLsganMax) return VITS_E_BADARG;
  tab.count = n_items;
  for (int i = 0; i < n_items; ++i) {
    if (!items[i].y8 || items[i].J <= 0 || items[i].J % 2 != 0 || items[i].R <= 0 || (need_dy && !items[i].dy8)) return VITS_E_BADARG;
    tab.y8[i] = items[i].y8; tab.dy8[i] = items[i].dy8; tab.J[i] = items[i].J; tab.R[i] = items[i].R;
  }
  return VITS_OK;
}

}  // namespace

extern "C" size_t vits_lsgan_workspace(int n_items) { return (size_t)n_items * 2 * kLsganSplits * sizeof(float); }

extern "C" int vits_lsgan_loss(int dtype, const vits_lsgan_item* host_items, int n_items, int mode, float* out, float* total,
                               void* workspace, size_t workspace_bytes, void* stream) {
  LsganTable tab;
  int rc = lsgan_table(host_items, n_items, tab, false);
  if (rc != VITS_OK) return rc;
  if (!out || !workspace || workspace_bytes < vits_lsgan_workspace(n_items) || (mode != 0 && mode != 1)) return VITS_E_BADARG;
  hipStream_t s = static_cast<hipStream_t>(stream);
  float* partial = static_cast<float*>(workspace);
  const dim3 grid(kLsganSplits, 2, n_items);
  if (dtype == VITS_DT_BF16) hipLaunchKernelGGL(lsgan_partial_kernel<__bf16>, grid, dim3(kThreads), 0, s, tab, mode, partial);
  else if (dtype == VITS_DT_F32) hipLaunchKernelGGL(lsgan_partial_kernel<float>, grid, dim3(kThreads), 0, s, tab, mode, partial);
  else return VITS_E_UNSUPPORTED;
  hipLaunchKernelGGL(lsgan_final_kernel, dim3(1), dim3(64), 0, s, tab, partial, out, total);
  return vits::check_launch("vits_lsgan_loss");
}

extern "C" int vits_lsgan_loss_bwd(int dtype, const vits_lsgan_item* host_items, int n_items, int mode, const float* g, void* stream) {
  LsganTable tab;
  int rc = lsgan_table(host_items, n_items, tab, true);
  if (rc != VITS_OK) return rc;
  if (!g || (mode != 0 && mode != 1)) return VITS_E_BADARG;
  hipStream_t s = static_cast<hipStream_t>(stream);
  const dim3 grid(64, n_items);
  if (dtype == VITS_DT_BF16) hipLaunchKernelGGL(lsgan_bwd_kernel<__bf16>, grid, dim3(kThreads), 0, s, tab, mode, g);
  else if (dtype == VITS_DT_F32) hipLaunchKernelGGL(lsgan_bwd_kernel<float>, grid, dim3(kThreads), 0, s, tab, mode, g);
  else return VITS_E_UNSUPPORTED;
  return vits::check_launch("vits_lsgan_loss_bwd");
}

extern "C" size_t vits_reduce_workspace(int n_seg) { return (size_t)n_seg * kMaxSplits * sizeof(float); }

extern "C" int vits_absdiff_sum(int dtype, const void* a, const void* b, size_t n, float scale, float* out, int accumulate,
                                void* workspace, size_t workspace_bytes, void* stream) {
  if (!a || !b || !out || !workspace || n == 0) return VITS_E_BADARG;
  if (workspace_bytes < vits_reduce_workspace(1)) return VITS_E_BADARG;
  hipStream_t s = static_cast<hipStream_t>(stream);
  const int splits = pick_splits(n, 1);
  float* part = static_cast<float*>(workspace);
  if (dtype == VITS_DT_BF16 && n % 8 == 0 && aligned16(a) && aligned16(b))
    hipLaunchKernelGGL(partial_absdiff_bf16x8, dim3(splits, 1), dim3(kThreads), 0, s, static_cast<const u32x4*>(a),
                       static_cast<const u32x4*>(b), n / 8, part);
  else if (dtype == VITS_DT_BF16)
    hipLaunchKernelGGL((partial_kernel<__bf16, true>), dim3(splits, 1), dim3(kThreads), 0, s, static_cast<const __bf16*>(a),
                       static_cast<const __bf16*>(b), n, part);
  else if (dtype == VITS_DT_F32)
    hipLaunchKernelGGL((partial_kernel<float, true>), dim3(splits, 1), dim3(kThreads), 0, s, static_cast<const float*>(a),
                       static_cast<const float*>(b), n, part);
  else
    return VITS_E_UNSUPPORTED;
  hipLaunchKernelGGL(final_kernel, dim3(1), dim3(64), 0, s, part, splits, scale, out, accumulate);
  return vits::check_launch("vits_absdiff_sum");
}

extern "C" int vits_absdiff_bwd(int dtype, const void* a, const void* b, size_t n, const float* g, float scale, void* da, void* db,
                                void* stream) {
  if (!a || !b || !g || !db || n == 0) return VITS_E_BADARG;
  hipStream_t s = static_cast<hipStream_t>(stream);
  size_t blocks = (n + (size_t)kThreads * 8 - 1) / ((size_t)kThreads * 8);
  if (blocks > 2048) blocks = 2048;
  if (dtype == VITS_DT_BF16 && n % 8 == 0 && aligned16(a) && aligned16(b) && aligned16(db) && (!da || aligned16(da)))
    hipLaunchKernelGGL(absdiff_bwd_bf16x8, dim3((unsigned)((n / 8 + kThreads - 1) / kThreads > 2048 ? 2048 : (n / 8 + kThreads - 1) / kThreads)),
                       dim3(kThreads), 0, s, static_cast<const u32x4*>(a), static_cast<const u32x4*>(b), n / 8, g, scale,
                       static_cast<u32x4*>(da), static_cast<u32x4*>(db));
  else if (dtype == VITS_DT_BF16)
    hipLaunchKernelGGL(absdiff_bwd_kernel<__bf16>, dim3((unsigned)blocks), dim3(kThreads), 0, s, static_cast<const __bf16*>(a),
                       static_cast<const __bf16*>(b), n, g, scale, static_cast<__bf16*>(da), static_cast<__bf16*>(db));
  else if (dtype == VITS_DT_F32)
    hipLaunchKernelGGL(absdiff_bwd_kernel<float>, dim3((unsigned)blocks), dim3(kThreads), 0, s, static_cast<const float*>(a),
                       static_cast<const float*>(b), n, g, scale, static_cast<float*>(da), static_cast<float*>(db));
  else
    return VITS_E_UNSUPPORTED;
  return vits::check_launch("vits_absdiff_bwd");
}

extern "C" int vits_segsum_f32(const float* x, int n_seg, size_t seg_len, float* out, void* workspace, size_t workspace_bytes,
                               void* stream) {
  if (!x || !out || !workspace || n_seg <= 0 || seg_len == 0) return VITS_E_BADARG;
  if (workspace_bytes < vits_reduce_workspace(n_seg)) return VITS_E_BADARG;
  hipStream_t s = static_cast<hipStream_t>(stream);
  const int splits = pick_splits(seg_len, n_seg);
  float* part = static_cast<float*>(workspace);
  hipLaunchKernelGGL((partial_kernel<float, false>), dim3(splits, n_seg), dim3(kThreads), 0, s, x, x, seg_len, part);
  hipLaunchKernelGGL(final_kernel, dim3(n_seg), dim3(64), 0, s, part, splits, 1.0f, out, 0);
  return vits::check_launch("vits_segsum_f32");
}

extern "C" int vits_lrelu_mask_bwd(int dtype, const void* dy, const void* y, float slope, const int32_t* lengths, int b, int t, int c,
                                   void* out, void* stream) {
  if (!dy || !out || b <= 0 || t <= 0 || c <= 0) return VITS_E_BADARG;
  const size_t n = (size_t)b * t * c;
  size_t blocks = (n + (size_t)kThreads * 4 - 1) / ((size_t)kThreads * 4);
  if (blocks > 4096) blocks = 4096;
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (dtype == VITS_DT_BF16 && c % 8 == 0 && aligned16(dy) && aligned16(out) && (!y || aligned16(y)))
    hipLaunchKernelGGL(lrelu_mask_bwd_bf16x8, dim3((unsigned)((n / 8 + kThreads - 1) / kThreads > 4096 ? 4096 : (n / 8 + kThreads - 1) / kThreads)),
                       dim3(kThreads), 0, s, static_cast<const u32x4*>(dy), static_cast<const u32x4*>(y), slope, lengths, t, c / 8, n / 8,
                       static_cast<u32x4*>(out));
  else if (dtype == VITS_DT_BF16)
    hipLaunchKernelGGL(lrelu_mask_bwd_kernel<__bf16>, dim3((unsigned)blocks), dim3(kThreads), 0, s, static_cast<const __bf16*>(dy),
                       static_cast<const __bf16*>(y), slope, lengths, t, c, n, static_cast<__bf16*>(out));
  else if (dtype == VITS_DT_F32)
    hipLaunchKernelGGL(lrelu_mask_bwd_kernel<float>, dim3((unsigned)blocks), dim3(kThreads), 0, s, static_cast<const float*>(dy),
                       static_cast<const float*>(y), slope, lengths, t, c, n, static_cast<float*>(out));
  else
    return VITS_E_UNSUPPORTED;
  return vits::check_launch("vits_lrelu_mask_bwd");
}

extern "C" size_t vits_colsum_workspace(int n_seg, int rows, int c) {
  return (size_t)n_seg * colsum_splits(n_seg, rows, c) * c * sizeof(float);
}

extern "C" int vits_colsum(int dtype, const void* x, int n_seg, int rows, int c, float* out, void* workspace, size_t workspace_bytes,
                           void* stream) {
  if (!x || !out || n_seg <= 0 || rows <= 0 || c <= 0) return VITS_E_BADARG;
  const int S = colsum_splits(n_seg, rows, c);
  if (S > 1 && (!workspace || workspace_bytes < vits_colsum_workspace(n_seg, rows, c))) return VITS_E_BADARG;
  hipStream_t s = static_cast<hipStream_t>(stream);
  float* dst = S > 1 ? static_cast<float*>(workspace) : out;
  dim3 grid((c + 63) / 64, S, n_seg);
  if (dtype == VITS_DT_BF16)
    hipLaunchKernelGGL(colsum_kernel<__bf16>, grid, dim3(kThreads), 0, s, static_cast<const __bf16*>(x), rows, c, dst);
  else if (dtype == VITS_DT_F32)
    hipLaunchKernelGGL(colsum_kernel<float>, grid, dim3(kThreads), 0, s, static_cast<const float*>(x), rows, c, dst);
  else
    return VITS_E_UNSUPPORTED;
  if (S > 1)
    hipLaunchKernelGGL(colsum_final_kernel, dim3((n_seg * c + kThreads - 1) / kThreads), dim3(kThreads), 0, s, dst, S, c, n_seg, out);
  return vits::check_launch("vits_colsum");
}
