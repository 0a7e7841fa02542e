// Multi-tensor weight preparation: ONE launch turns the fp32 master parameters of a whole group of
// convolutions into the matrix-core kernels' operands, and ONE launch maps the weight gradients back.
//
// Replaces, per convolution and per step, the reference's weight-norm recomputation
// (torch.nn.utils.weight_norm's forward pre-hook: w = g * v / ||v||, norm over every dim but 0 —
// reference models.py:254, modules.py:128,135,145,191-206) and its autograd, plus the layout/precision
// changes this implementation needs:
//   forward  : w_fwd [k][c_out_p][c_in_p]  (tap-major, channels padded to the vector width)
//              w_bwd [k][c_in_p][c_out_p]  with reversed taps (the data-gradient operand)
//              in the compute dtype (bf16 or f32);
//   backward : dv = (g/||v||) * (dW - v * <dW, v>/||v||^2),  dg = <dW, v>/||v||   (plain layers: dW re-laid-out)
// One workgroup per weight-norm row (output channel of a Conv1d, INPUT channel of a ConvTranspose1d,
// whose legacy weight_norm dim 0 is c_in).  ConvTranspose1d [c_in][c_out][k] is emitted as the 1x1
// operand [1][k*c_out][c_in] that vits_convt_fold_cl expects.
#include "common.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
constexpr int kRowMax = 6144;            // floats of a parameter row staged in LDS (1024 x 5 and 512 x 11 fit); longer rows walk global memory twice

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

__device__ __forceinline__ int find_entry(const vits_prep_entry* e, int n, int row) {
  int lo = 0, hi = n - 1;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (e[mid].row0 <= row) lo = mid; else hi = mid - 1;
  }
  return lo;
}

template <typename T> __device__ __forceinline__ void put(T* p, size_t i, float v);
template <> __device__ __forceinline__ void put<float>(float* p, size_t i, float v) { p[i] = v; }
template <> __device__ __forceinline__ void put<__bf16>(__bf16* p, size_t i, float v) { p[i] = (__bf16)v; }

// element (row, inner) of the torch-layout parameter -> (tap, co, ci)
struct Idx { int tap, co, ci; };
__device__ __forceinline__ Idx locate(const vits_prep_entry& e, int r, int inner) {
  Idx x;
  if (e.layout != 1) {            // Conv1d [c_out][c_in][k] (layouts 0, 2, 3, 4): row = co (within the entry), inner = ci*k + tap
    x.co = r; x.ci = inner / e.k; x.tap = inner % e.k;
    if (e.layout == 3) x.ci += (r / (e.c_out / e.groups)) * (e.c_in / e.groups);   // grouped: channel inside the dense operand
  } else {                        // ConvTranspose1d [c_in][c_out][k]: row = ci, inner = co*k + j
    x.ci = r; x.co = inner / e.k; x.tap = inner % e.k;
  }
  return x;
}


template <typename T> __device__ __forceinline__ void put8(T* p, const float (&v)[8]);
template <> __device__ __forceinline__ void put8<float>(float* p, const float (&v)[8]) {
  *reinterpret_cast<f32x4*>(p) = f32x4{v[0], v[1], v[2], v[3]};
  *reinterpret_cast<f32x4*>(p + 4) = f32x4{v[4], v[5], v[6], v[7]};
}
template <> __device__ __forceinline__ void put8<__bf16>(__bf16* p, const float (&v)[8]) {
  bf16x8 o;
#pragma unroll
  for (int j = 0; j < 8; ++j) o[j] = (__bf16)v[j];
  *reinterpret_cast<bf16x8*>(p) = o;
}

// The row is read from memory ONCE (16-byte loads) into LDS; the norm and the operands come out of LDS.  The forward operand of
// a plain convolution (layout 4: w_fwd[tap][co][ci], ci contiguous) is written tap by tap as 16-byte stores of 8 consecutive
// input channels — the row walked with stride k in LDS (k is odd for every VITS convolution: conflict-free) — instead of 2-byte
// stores scattered over the k tap planes.  Summation order of the norm and every rounding are those of the first version.
template <typename T>
__global__ __launch_bounds__(256) void prep_fwd(const vits_prep_entry* __restrict__ ents, int n_ents, T* __restrict__ w_fwd,
                                                T* __restrict__ w_bwd) {
  __shared__ float red[4];
  __shared__ __attribute__((aligned(16))) float row[kRowMax];
  const vits_prep_entry e = ents[find_entry(ents, n_ents, blockIdx.x)];
  const int r = blockIdx.x - e.row0;                         // row within the entry
  const int inner = (e.layout == 1) ? e.c_out * e.k : (e.layout == 3 ? (e.c_in / e.groups) * e.k : e.c_in * e.k);
  const float* v = e.v + ((size_t)(e.row_lo + r)) * inner;
  const bool staged = inner <= kRowMax;
  if (staged) {
    if ((inner & 3) == 0 && (reinterpret_cast<uintptr_t>(v) & 15) == 0) {
      for (int i = threadIdx.x; i < (inner >> 2); i += blockDim.x) reinterpret_cast<f32x4*>(row)[i] = reinterpret_cast<const f32x4*>(v)[i];
    } else {
      for (int i = threadIdx.x; i < inner; i += blockDim.x) row[i] = v[i];
    }
    __syncthreads();
  }
  const float* src = staged ? row : v;                        // (generic address: LDS or global)
  float scale = 1.f;
  if (e.g) {
    float ss = 0.f;
    for (int i = threadIdx.x; i < inner; i += blockDim.x) { const float a = src[i]; ss += a * a; }
    ss = block_sum(ss, red);
    scale = e.g[e.row_lo + r] / sqrtf(ss);
  }
  if (e.layout == 4 && staged && (e.c_in & 7) == 0 && (e.c_in_p & 7) == 0) {
    const int c8n = e.c_in >> 3, k = e.k;
    T* dst = w_fwd + e.off + (size_t)r * e.c_in_p;
    const size_t plane = (size_t)e.c_out_p * e.c_in_p;
    for (int tap = 0; tap < k; ++tap)
      for (int c8 = threadIdx.x; c8 < c8n; c8 += blockDim.x) {
        float o[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = row[(c8 * 8 + j) * k + tap] * scale;
        put8<T>(dst + tap * plane + c8 * 8, o);
      }
    return;
  }
  for (int i = threadIdx.x; i < inner; i += blockDim.x) {
    const Idx x = locate(e, r, i);
    const float val = src[i] * scale;
    if (e.layout == 2) {                  // torch layout kept (consumer is a MIOpen convolution): only weight-norm + dtype
      put<T>(w_fwd, e.off + (size_t)r * inner + i, val);
    } else if (e.layout == 0 || e.layout == 3 || e.layout == 4) {
      put<T>(w_fwd, e.off + ((size_t)x.tap * e.c_out_p + x.co) * e.c_in_p + x.ci, val);
      // the transposed copy: a strided 2-byte scatter from here; layout 4 leaves it to transpose_tiles (coalesced both ways)
      if (e.layout != 4) put<T>(w_bwd, e.off + ((size_t)(e.k - 1 - x.tap) * e.c_in_p + x.ci) * e.c_out_p + x.co, val);
    } else {
      const size_t col = (size_t)x.tap * e.c_out + x.co;                   // column of the 1x1 operand
      put<T>(w_fwd, e.off + col * e.c_in_p + x.ci, val);                   // [1][k*c_out][c_in_p]
      put<T>(w_bwd, e.off + (size_t)x.ci * ((size_t)e.k * e.c_out) + col, val);   // [1][c_in][k*c_out]
    }
  }
}

__global__ __launch_bounds__(256) void prep_bwd(const vits_prep_entry* __restrict__ ents, int n_ents,
                                                const float* __restrict__ dw, float* __restrict__ dparam) {
  __shared__ float red[4];
  const vits_prep_entry e = ents[find_entry(ents, n_ents, blockIdx.x)];
  const int r = blockIdx.x - e.row0;
  const int inner = (e.layout == 1) ? e.c_out * e.k : (e.layout == 3 ? (e.c_in / e.groups) * e.k : e.c_in * e.k);
  const float* v = e.v + ((size_t)(e.row_lo + r)) * inner;
  float* dv = dparam + e.off_dv + ((size_t)(e.row_lo + r)) * inner;
  auto dw_at = [&](int i) -> float {
    if (e.layout == 2) return dw[e.off + (size_t)r * inner + i];
    if (e.layout == 3) return dw[e.off + ((size_t)(i % e.k) * e.c_out + r) * (e.c_in / e.groups) + i / e.k];   // compact [k][c_out][Ig]
    const Idx x = locate(e, r, i);
    if (e.layout == 0 || e.layout == 4) return dw[e.off + ((size_t)x.tap * e.c_out_p + x.co) * e.c_in_p + x.ci];
    return dw[e.off + ((size_t)x.tap * e.c_out + x.co) * e.c_in_p + x.ci];
  };
  // (Measured and not kept: the passes tap-major — scattered dv writes, 1.7x slower; the row of dW staged in LDS by 16-byte
  // tap-plane loads — 265 against 228 us per launch, the 24 KB per workgroup cost more occupancy than the gathers cost; the
  // thread's elements of v and dW kept in registers between the passes, one trip to memory — no change in an A/B of the step;
  // one WAVE per row, four rows per workgroup, butterfly reductions without LDS or barriers — 255 us.  What is left is the
  // gather itself: a wave's 64 consecutive parameter elements lie in k tap planes of dW, 5-10 cache lines per load.)
  if (!e.g) {
    for (int i = threadIdx.x; i < inner; i += blockDim.x) dv[i] = dw_at(i);
    return;
  }
  float ss = 0.f, dot = 0.f;
  for (int i = threadIdx.x; i < inner; i += blockDim.x) { const float a = v[i]; ss += a * a; dot += a * dw_at(i); }
  ss = block_sum(ss, red);
  dot = block_sum(dot, red);
  const float norm = sqrtf(ss), gval = e.g[e.row_lo + r];
  const float s = gval / norm, c = dot / ss;
  for (int i = threadIdx.x; i < inner; i += blockDim.x) dv[i] = s * (dw_at(i) - v[i] * c);
  if (threadIdx.x == 0) dparam[e.off_dg + e.row_lo + r] = dot / norm;
}

// w_bwd[k-1-tap][ci][co] = w_fwd[tap][co][ci] for 64 x 64 tiles listed in a table (layout-4 entries): LDS transpose, 128-byte
// rows on both sides
template <typename T>
__global__ __launch_bounds__(256) void transpose_tiles(const vits_prep_tile* __restrict__ tiles, const vits_prep_entry* __restrict__ ents,
                                                       const T* __restrict__ w_fwd, T* __restrict__ w_bwd) {
  __shared__ T tile[64][66];
  const vits_prep_tile t = tiles[blockIdx.x];
  const vits_prep_entry e = ents[t.entry];
  const int c = threadIdx.x & 63, r4 = threadIdx.x >> 6;
  const T* src = w_fwd + e.off + (size_t)t.tap * e.c_out_p * e.c_in_p;
  T* dst = w_bwd + e.off + (size_t)(e.k - 1 - t.tap) * e.c_in_p * e.c_out_p;
#pragma unroll 4
  for (int r = r4; r < 64; r += 4) {
    const int co = t.co0 + r, ci = t.ci0 + c;
    tile[r][c] = (co < e.c_out_p && ci < e.c_in_p) ? src[(size_t)co * e.c_in_p + ci] : (T)0.f;
  }
  __syncthreads();
#pragma unroll 4
  for (int r = r4; r < 64; r += 4) {
    const int ci = t.ci0 + r, co = t.co0 + c;
    if (ci < e.c_in_p && co < e.c_out_p) dst[(size_t)ci * e.c_out_p + co] = tile[c][r];
  }
}

}  // namespace

extern "C" int vits_weight_prep_transpose(const vits_prep_tile* tiles, int n_tiles, const vits_prep_entry* entries, int dtype,
                                          const void* w_fwd, void* w_bwd, void* stream) {
  if (!tiles || n_tiles <= 0 || !entries || !w_fwd || !w_bwd) return VITS_E_BADARG;
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (dtype == VITS_DT_BF16)
    hipLaunchKernelGGL(transpose_tiles<__bf16>, dim3(n_tiles), dim3(256), 0, s, tiles, entries, static_cast<const __bf16*>(w_fwd), static_cast<__bf16*>(w_bwd));
  else if (dtype == VITS_DT_F32)
    hipLaunchKernelGGL(transpose_tiles<float>, dim3(n_tiles), dim3(256), 0, s, tiles, entries, static_cast<const float*>(w_fwd), static_cast<float*>(w_bwd));
  else
    return VITS_E_UNSUPPORTED;
  return vits::check_launch("vits_weight_prep_transpose");
}

extern "C" int vits_weight_prep(const vits_prep_entry* entries, int n_entries, int total_rows, int dtype, void* w_fwd,
                                void* w_bwd, void* stream) {
  if (!entries || n_entries <= 0 || total_rows <= 0 || !w_fwd || !w_bwd) return VITS_E_BADARG;
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (dtype == VITS_DT_BF16)
    hipLaunchKernelGGL(prep_fwd<__bf16>, dim3(total_rows), dim3(256), 0, s, entries, n_entries, static_cast<__bf16*>(w_fwd), static_cast<__bf16*>(w_bwd));
  else if (dtype == VITS_DT_F32)
    hipLaunchKernelGGL(prep_fwd<float>, dim3(total_rows), dim3(256), 0, s, entries, n_entries, static_cast<float*>(w_fwd), static_cast<float*>(w_bwd));
  else
    return VITS_E_UNSUPPORTED;
  return vits::check_launch("vits_weight_prep");
}

extern "C" int vits_weight_prep_bwd(const vits_prep_entry* entries, int n_entries, int total_rows, const float* dw,
                                    float* dparam, void* stream) {
  if (!entries || n_entries <= 0 || total_rows <= 0 || !dw || !dparam) return VITS_E_BADARG;
  hipLaunchKernelGGL(prep_bwd, dim3(total_rows), dim3(256), 0, static_cast<hipStream_t>(stream), entries, n_entries, dw, dparam);
  return vits::check_launch("vits_weight_prep_bwd");
}
