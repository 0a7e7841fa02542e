// "Flat-row" variant of the channels-last MFMA convolution: the GEMM M dimension enumerates (item, time)
// pairs jointly instead of tiling each item's time axis, and every tap's input row is GATHERED per output row.
//
// Why: the period discriminators (reference models.py:299-335) fold the period into the batch, so their deep
// layers have 10..51 rows per item and hundreds of items — per-item tiles of 64/128 rows would be mostly empty —
// and their convolutions are strided (stride 3).  Same MFMA core, LDS pitch and epilogue as conv1d_cl.hip; what
// differs is the staging:  ldsX[tap][row][chunk]  with  row m -> (b, t),  input time = (t*stride + tap*dil - pad) / in_div.
//   forward, stride s            : in_div = 1, phases = 1
//   data gradient of a stride-s  : call on dY with stride = 1, in_div = s, pad' = dil*(k-1) - pad, phases = s:
//     rows are enumerated phase-major (t mod s), so inside a tile only the taps with
//     (phase + tap*dil - pad') % s == 0 are non-zero and the others are skipped entirely (no zero-insertion,
//     no wasted MFMAs).
#include "common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

constexpr int ROWB = 128;
constexpr int PITCH = ROWB + 16;
constexpr int kThreads = 256;
constexpr int XVF = 8;        // 16-byte vectors of gathered X per thread per stage
constexpr int WVF = 8;        // ... of W
constexpr int KMAX = 48;      // taps

template <typename T> struct Elem;
template <> struct Elem<__bf16> { static constexpr int VEC = 8; static constexpr int KC = 64; };
template <> struct Elem<float> { static constexpr int VEC = 4; static constexpr int KC = 32; };

struct FlatArgs {
  vits_conv_desc d;
  int Tout, G, phases, Q, tiles_per_phase;     // Q = rows of time per phase, per item
};

__device__ __forceinline__ float to_f(float v) { return v; }
__device__ __forceinline__ float to_f(__bf16 v) { return (float)v; }
template <typename T> __device__ __forceinline__ T from_f(float v);
template <> __device__ __forceinline__ float from_f<float>(float v) { return v; }
template <> __device__ __forceinline__ __bf16 from_f<__bf16>(float v) { return (__bf16)v; }

template <typename T>
__device__ __forceinline__ u32x4 lrelu_vec(u32x4 raw, float slope) {
  constexpr int V = Elem<T>::VEC;
  union { u32x4 u; T e[V]; } in, out;
  in.u = raw;
#pragma unroll
  for (int i = 0; i < V; ++i) {
    float f = to_f(in.e[i]);
    out.e[i] = from_f<T>(f > 0.f ? f : f * slope);
  }
  return out.u;
}

// WR = 32-row blocks per wave: WR = 2 gives every wave a 64 x (32*NT) tile, so one A and one B fragment read from LDS feed
// two MFMAs each (LDS reads per MFMA: (WR + NT) / (WR * NT) — 1.5 for 1x2, 1.0 for 2x2): the large discriminator layers are
// LDS-read-bound, not MFMA-bound.
template <typename T, int NT, int WM, int WR>
__global__ __launch_bounds__(kThreads) void conv1d_flat_kernel(FlatArgs args) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  __shared__ int taplist[KMAX];
  __shared__ int n_taps_s;
  const vits_conv_desc& a = args.d;
  constexpr int V = Elem<T>::VEC;
  constexpr int KC = Elem<T>::KC;
  constexpr int WN = 4 / WM, TMW = 32 * WM * WR, TNW = 32 * NT, TN = TNW * WN;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave % WM, wn = wave / WM;
  const int r = lane & 31, h = lane >> 5;
  const int Tout = args.Tout;
  const int phase = blockIdx.x / args.tiles_per_phase;
  const int row0 = (blockIdx.x % args.tiles_per_phase) * TMW;        // first row of this tile inside its phase
  const int rows_in_phase = a.b * args.Q;
  const int co0 = blockIdx.y * TN;
  const int in_div = a.in_div > 1 ? a.in_div : 1;

  // taps that can be non-zero for this phase
  if (tid == 0) {
    int n = 0;
    for (int tap = 0; tap < a.k; ++tap) {
      const int num = phase * a.stride + tap * a.dil - a.pad;
      const int md = ((num % in_div) + in_div) % in_div;
      if (args.phases == 1 || md == 0) taplist[n++] = tap;
    }
    n_taps_s = n;
  }
  __syncthreads();
  const int n_taps = n_taps_s;

  // decode a row of the tile: -> item b, output time t (or t = -1)
  auto decode = [&](int row, int& b, int& t) {
    const int m = row0 + row;
    b = 0; t = -1;
    if (m < rows_in_phase) {
      b = m / args.Q;
      const int q = m - b * args.Q;
      const int tt = q * args.phases + phase;
      if (tt < Tout) t = tt;
    }
  };

  unsigned char* ldsX = smem;                                    // [G][TMW][PITCH]
  unsigned char* ldsW = smem + (size_t)args.G * TMW * PITCH;     // [G][TN][PITCH]
  const T* X = static_cast<const T*>(a.x);
  const T* W = static_cast<const T*>(a.w);

  // per-thread staging slots: slot i handles vector idx = tid + i*256 of [G][TMW][8]
  int xs_b[XVF], xs_t[XVF];
#pragma unroll
  for (int i = 0; i < XVF; ++i) {
    const int idx = tid + i * kThreads;
    const int row = (idx >> 3) % TMW;
    decode(row, xs_b[i], xs_t[i]);
  }

  f32x16 acc[WR][NT];
#pragma unroll
  for (int q = 0; q < WR; ++q)
#pragma unroll
    for (int n = 0; n < NT; ++n)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[q][n][i] = 0.f;

  const int n_groups = (n_taps + args.G - 1) / args.G;
  // grouped convolution on dense block-diagonal operands: this tile of output channels only sees the input
  // channels of its own groups
  int ci_begin = 0, ci_end = a.c_in;
  if (a.groups > 1) {
    const int og = a.c_out / a.groups, ig = a.c_in / a.groups;
    const int co_last = (co0 + TN < a.c_out ? co0 + TN : a.c_out) - 1;
    ci_begin = ((co0 / og) * ig / KC) * KC;
    ci_end = (co_last / og + 1) * ig;
  }
  const int n_chunks = (ci_end - ci_begin + KC - 1) / KC;
  const int n_stages = n_groups * n_chunks;

  u32x4 xr[XVF], wr[WVF];
  auto load_stage = [&](int ci0, int g0) {
    const int ntap = (n_taps - g0 < args.G) ? (n_taps - g0) : args.G;
#pragma unroll
    for (int i = 0; i < XVF; ++i) {
      const int idx = tid + i * kThreads;
      u32x4 v = {0u, 0u, 0u, 0u};
      const int tl = (idx >> 3) / TMW, ch = idx & 7;
      if (tl < ntap && xs_t[i] >= 0) {
        const int tap = taplist[g0 + tl];
        const int num = xs_t[i] * a.stride + tap * a.dil - a.pad;
        const int ci = ci0 + ch * V;
        int ti = num;
        bool on_grid = true;
        if (in_div > 1) {                    // (uniform branch: keeps the integer division out of the common case)
          ti = num / in_div;
          on_grid = ti * in_div == num;
        }
        if (num >= 0 && on_grid && ci < a.c_in) {
          int hi = a.t;
          if (a.flags & VITS_CONV_MASK_IN) { const int len = a.lengths[xs_b[i]]; hi = len < a.t ? len : a.t; }
          if (ti < hi) v = *reinterpret_cast<const u32x4*>(X + ((size_t)xs_b[i] * a.t + ti) * a.ldx + ci);
        }
      }
      xr[i] = v;
    }
#pragma unroll
    for (int i = 0; i < WVF; ++i) {
      const int idx = tid + i * kThreads;
      u32x4 v = {0u, 0u, 0u, 0u};
      if (idx < ntap * TN * 8) {
        const int ch = idx & 7, col = (idx >> 3) % TN, tl = (idx >> 3) / TN;
        const int co = co0 + col, ci = ci0 + ch * V;
        if (co < a.c_out && ci < a.c_in)
          v = *reinterpret_cast<const u32x4*>(W + ((size_t)taplist[g0 + tl] * a.c_out + co) * a.ldw + ci);
      }
      wr[i] = v;
    }
  };
  auto store_stage = [&](int g0) {
    const int ntap = (n_taps - g0 < args.G) ? (n_taps - g0) : args.G;
#pragma unroll
    for (int i = 0; i < XVF; ++i) {
      const int idx = tid + i * kThreads;
      // (the fused input leaky-relu runs here, after the stage's MFMAs: the loads stay in flight during them)
      if (idx < ntap * TMW * 8) *reinterpret_cast<u32x4*>(ldsX + (idx >> 3) * PITCH + (idx & 7) * 16) = (a.in_slope != 1.0f) ? lrelu_vec<T>(xr[i], a.in_slope) : xr[i];
    }
#pragma unroll
    for (int i = 0; i < WVF; ++i) {
      const int idx = tid + i * kThreads;
      if (idx < ntap * TN * 8) *reinterpret_cast<u32x4*>(ldsW + (idx >> 3) * PITCH + (idx & 7) * 16) = wr[i];
    }
  };

  if (n_stages > 0) { load_stage(ci_begin, 0); store_stage(0); }
  __syncthreads();
  for (int s = 0; s < n_stages; ++s) {
    const int g0 = (s % n_groups) * args.G;
    const int nxt = s + 1;
    const bool has_next = nxt < n_stages;
    const int ng0 = (nxt % n_groups) * args.G, nci0 = ci_begin + (nxt / n_groups) * KC;
    if (has_next) load_stage(nci0, ng0);
    const int ntap = (n_taps - g0 < args.G) ? (n_taps - g0) : args.G;
    for (int tl = 0; tl < ntap; ++tl) {
      const unsigned char* xa = ldsX + (tl * TMW + wm * 32 * WR + r) * PITCH + 16 * h;
      const unsigned char* wb = ldsW + (tl * TN + wn * TNW + r) * PITCH + 16 * h;
#pragma unroll
      for (int m = 0; m < 4; ++m) {
        u32x4 av[WR];
#pragma unroll
        for (int q = 0; q < WR; ++q) av[q] = *reinterpret_cast<const u32x4*>(xa + q * 32 * PITCH + 32 * m);
#pragma unroll
        for (int n = 0; n < NT; ++n) {
          const u32x4 bv = *reinterpret_cast<const u32x4*>(wb + n * 32 * PITCH + 32 * m);
#pragma unroll
          for (int q = 0; q < WR; ++q) {
            if constexpr (sizeof(T) == 2) {
              union { u32x4 u; bf16x8 v; } ua, ub;
              ua.u = av[q]; ub.u = bv;
              acc[q][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ua.v, ub.v, acc[q][n], 0, 0, 0);
            } else {
              union { u32x4 u; float f[4]; } ua, ub;
              ua.u = av[q]; ub.u = bv;
#pragma unroll
              for (int j = 0; j < 4; ++j)
                acc[q][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(ua.f[j], ub.f[j], acc[q][n], 0, 0, 0);
            }
          }
        }
      }
    }
    if (has_next) {
      __syncthreads();
      store_stage(ng0);
      __syncthreads();
    }
  }

  // ---- epilogue (subset of conv1d_cl's: bias, per-item bias, residual, scale, lrelu' multiplier, out lrelu, masks, accumulate)
  T* Y = static_cast<T*>(a.y);
  const T* R = static_cast<const T*>(a.res);
  const T* MG = static_cast<const T*>(a.mg_src);
#pragma unroll
  for (int q = 0; q < WR; ++q)
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    int b, t;
    decode((wm * WR + q) * 32 + (i & 3) + 8 * (i >> 2) + 4 * h, b, t);
    if (t < 0) continue;
    const int len = a.lengths ? a.lengths[b] : Tout;
#pragma unroll
    for (int n = 0; n < NT; ++n) {
      const int co = co0 + (wn * NT + n) * 32 + r;
      if (co >= a.c_out) continue;
      const size_t o = ((size_t)b * Tout + t) * a.ldy + co;
      float v = acc[q][n][i];
      if (a.bias) v += a.bias[co];
      if (a.bias_b) v += a.bias_b[(size_t)b * a.c_out + co];
      const bool res_after = (a.flags & VITS_CONV_RES_AFTER) != 0;
      if (R && !res_after) v += to_f(R[o]);
      v *= a.out_scale;
      if (MG) v *= (to_f(MG[o]) > 0.f) ? 1.0f : a.mg_slope;
      if (R && res_after) v += to_f(R[o]);
      if (a.flags & VITS_CONV_OUT_LRELU) v = v > 0.f ? v : v * a.out_slope;
      if (a.flags & VITS_CONV_TANH) v = tanhf(v);
      if ((a.flags & VITS_CONV_MASK_OUT) && t >= len) v = 0.f;
      if (a.flags & VITS_CONV_ACCUM) v += to_f(Y[o]);
      Y[o] = from_f<T>(v);
    }
  }
}

template <typename T, int NT, int WM, int WR = 1>
int launch_flat(const vits_conv_desc& d, int t_out, hipStream_t s) {
  constexpr int WN = 4 / WM, TMW = 32 * WM * WR, TN = 32 * NT * WN;
  FlatArgs args{d, t_out, 1, 1, 1, 1};
  const int in_div = d.in_div > 1 ? d.in_div : 1;
  args.phases = (in_div > 1 && d.stride == 1) ? in_div : 1;
  args.Q = vits::ceil_div(t_out, args.phases);
  args.tiles_per_phase = vits::ceil_div(d.b * args.Q, TMW);
  int G = (kThreads * XVF) / (TMW * 8);               // taps per stage the gather registers can hold
  const int gw = (kThreads * WVF) / (TN * 8);
  if (gw < G) G = gw;
  if (G < 1) return VITS_E_UNSUPPORTED;
  if (G > d.k) G = d.k;
  args.G = G;
  const size_t lds = (size_t)G * (TMW + TN) * PITCH;
  if (lds > (size_t)vits::kLdsBytesMax - 1024) return VITS_E_UNSUPPORTED;
  auto kern = conv1d_flat_kernel<T, NT, WM, WR>;
  { const hipError_t e = vits::ensure_max_dynamic_lds(reinterpret_cast<const void*>(kern), 1024); if (e != hipSuccess) return vits::note_hip_error(e, "vits_conv1d_flat/attr"); }   // minus the static taplist
  dim3 grid(args.phases * args.tiles_per_phase, vits::ceil_div(d.c_out, TN), 1);
  hipLaunchKernelGGL(kern, grid, dim3(kThreads), lds, s, args);
  return vits::check_launch("vits_conv1d_cl(flat)");
}

}  // namespace

namespace vits {

// Called by vits_conv1d_cl for strided / divided / short-sequence launches.  `d` is validated and defaulted.
int conv1d_flat_dispatch(const vits_conv_desc& d, int t_out, hipStream_t s) {
  if (d.k > KMAX) return VITS_E_UNSUPPORTED;
  const long rows = (long)d.b * t_out;
  if (d.dtype == VITS_DT_BF16) {
    // (measured and dropped: a 64x64-per-wave variant, launch_flat<__bf16, 2, 2, 2>, was slower for the step — 53.2 vs 52.1 ms)
    if (d.c_out > 64 && rows * ((d.c_out + 127) / 128) >= 128 * 512) return launch_flat<__bf16, 4, 4>(d, t_out, s);   // 128 x 128
    // 64 x 64 tiles when 64 x 128 would leave CUs idle, and for grouped layers (one 64-channel chunk per tile, 41 taps: more
    // taps per stage and twice the workgroups)
    const bool small = true;
    const long wgs_64x128 = ((rows + 63) / 64) * ((d.c_out + 127) / 128);
    if (d.c_out > 64 && small && (d.groups > 1 || wgs_64x128 < 320)) return launch_flat<__bf16, 1, 2>(d, t_out, s);
    if (d.c_out > 64) return launch_flat<__bf16, 2, 2>(d, t_out, s);                                               // 64 x 128
    if (d.c_out > 32) return launch_flat<__bf16, 1, 2>(d, t_out, s);                                               // 64 x 64
    return launch_flat<__bf16, 1, 4>(d, t_out, s);                                                                 // 128 x 32
  }
  // fp32 (the DFT products of the mel, 512 rows): 64 x 64 tiles when 64 x 128 would leave most CUs idle
  if (d.c_out > 64 && ((rows + 63) / 64) * ((d.c_out + 127) / 128) >= 320) return launch_flat<float, 2, 2>(d, t_out, s);
  if (d.c_out > 32) return launch_flat<float, 1, 2>(d, t_out, s);
  return launch_flat<float, 1, 4>(d, t_out, s);
}

}  // namespace vits
