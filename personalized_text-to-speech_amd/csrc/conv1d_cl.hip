// Channels-last 1-D convolution on the CDNA4 matrix cores (implicit GEMM), forward / data-gradient.
//
// Serves every stride-1 Conv1d of the generator (reference modules.py ResBlock1/2 :187-256,
// WN :111-184, attentions.FFN :257-303, models.Generator conv_pre/ups(1x1 part) :244-289, the 1x1
// projections) and, with tap-flipped transposed weights, their data gradients.
//
// Layout (MI355X-first, not the reference's [b, c, t]): activations are [b][t][c] ("channels
// last") so that the GEMM reduction index (input channel) is contiguous for BOTH MFMA operands:
//     Y[b][t][co] = epilogue( sum_{tap, ci} W[tap][co][ci] * act(X[b][t + tap*dil - pad][ci]) )
//   A operand = X rows  (M = 32 time steps per wave),  lane (r, h) holds 16 contiguous bytes of row r
//   B operand = W rows  (N = 32 output channels),      lane (c, h) holds 16 contiguous bytes of row c
// and a tap is just a row offset into the staged X tile.  One source serves two element types:
//   bf16 : v_mfma_f32_32x32x16_bf16, 8 elements per lane per instruction  (performance mode)
//   f32  : v_mfma_f32_32x32x2_f32,  4 instructions per 16-byte fragment   (exact-fp32 parity mode:
//          the matrix core's f32 path is a k-ordered fmaf chain, MI355X_MICROARCH.md)
// because a 16-byte fragment at byte offset 32*m + 16*h of a 128-byte channel chunk means
// "elements k = 8h..8h+7 of MFMA m" for bf16 and "k = h of MFMAs 4m..4m+3" for f32.
//
// Workgroup = 4 waves arranged WM (time) x WN (channels), each wave 32 rows x 32*NT output channels;
// the tile is chosen per launch so that small layers still put >= 2 workgroups on every CU.  Per 128-byte
// input-channel chunk the X rows (with halo, input activation and row mask applied once) and a group of
// taps of W are staged in LDS with a 16-byte row pad (pitch 144 B: conflict-free ds_read_b128); the
// global loads of the NEXT stage are issued into registers before the current stage's MFMAs and written
// to LDS after them (register-prefetch software pipeline, cdna_hip_programming.md T14).
//
// Fused epilogue (flags): + bias[co] + bias_b[b][co] (speaker conditioning) + residual, * scale,
// * leaky-relu'(src) (chain rule of a fused input activation, for the data-gradient call),
// residual after the multiplier (skip connection of a data gradient), row mask, tanh, accumulate.
#include "common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));


constexpr int ROWB = 128;          // bytes of input channels staged per chunk
// LDS row pitch = ROWB * NC + 16 (PITCHK inside the kernel)
constexpr int kThreads = 256;

template <typename T> struct Elem;
template <> struct Elem<__bf16> { static constexpr int VEC = 8; static constexpr int KC = 64; };
template <> struct Elem<float> { static constexpr int VEC = 4; static constexpr int KC = 32; };

struct ConvArgs {
  vits_conv_desc d;
  int Tout, G;            // output rows; taps staged per group
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

__device__ __forceinline__ float sigmoidf_(float v) { return 1.0f / (1.0f + __expf(-v)); }
// tanh of the WaveNet gate: libm's tanhf costs about as much as the layer's MFMA loop per element; bf16 mode takes
// 1 - 2 / (1 + e^{2v}) on the fast exponential (absolute error ~1e-7, far below bf16's rounding), fp32 parity mode keeps tanhf
template <typename T> __device__ __forceinline__ float gate_tanh(float v) {
  if constexpr (sizeof(T) == 2) return 1.0f - 2.0f / (1.0f + __expf(2.0f * v));
  else return tanhf(v);
}

constexpr int XV_MAX = 6;     // 16-byte vectors of the X tile a thread may hold in flight (xrows*8 <= 256*XV_MAX)
constexpr int WV_MAX = 9;     // ... of the W slab (G*TN*8 <= 256*WV_MAX)

// WM waves along time x WN = 4/WM waves along output channels; each wave owns 32 rows x 32*NT columns.
// NC = 128-byte channel chunks staged per pipeline stage.  NC = 3 (bf16: 192 channels) makes the 1x1 layers of the WaveNet
// stacks, the text encoder and the duration predictor SINGLE-stage: one round of global loads, then all the MFMAs — these
// launches are bound by the load latency of their 3 sequential stages, not by bandwidth or math.
template <typename T, int NT, int WM, int NC>
__global__ __launch_bounds__(kThreads) void conv1d_cl_kernel(ConvArgs args) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const vits_conv_desc& a = args.d;
  constexpr int V = Elem<T>::VEC;
  constexpr int KC = Elem<T>::KC * NC;    // channels per stage
  constexpr int VPR = 8 * NC;             // 16-byte vectors per staged row
  constexpr int PITCHK = ROWB * NC + 16;  // LDS row pitch (same bank pattern for NC = 1 and 3: 36 and 100 dwords, both 4 mod 32)
  constexpr int WN = 4 / WM;
  constexpr int TMW = 32 * WM;            // time rows per workgroup
  constexpr int TNW = 32 * NT;            // columns per wave
  constexpr int TN = TNW * WN;            // columns per workgroup
  constexpr int XV = NC == 1 ? XV_MAX : (TMW * VPR) / kThreads + 1;      // NC > 1: k = 1, stride 1, so the X tile is exactly TMW rows
  constexpr int WV = NC == 1 ? WV_MAX : (TN * VPR + kThreads - 1) / kThreads;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave % WM, wn = wave / WM;
  const int r = lane & 31, h = lane >> 5;
  const int t0 = blockIdx.x * TMW;
  const int b = blockIdx.z;
  const int Tout = args.Tout;
  const bool gate = (a.flags & VITS_CONV_GATE) != 0;
  // output-channel origin of this workgroup: in GATE mode every wave's NT tiles are split in two halves
  // (columns c.. of the tanh half and c + gate_h.. of the sigmoid half), so a workgroup covers TN/2 gate channels.
  const int co0 = blockIdx.y * (gate ? TN / 2 : TN);
  const int xrows = (TMW - 1) * a.stride + (a.k - 1) * a.dil + 1;
  unsigned char* ldsX = smem;
  unsigned char* ldsW = smem + (size_t)xrows * PITCHK;

  const T* X = static_cast<const T*>(a.x) + (size_t)b * a.t * a.ldx;
  const T* W = static_cast<const T*>(a.w) + (size_t)b * a.w_batch_stride;     // per-item operand (attention products) or shared weights
  const int len = (a.lengths != nullptr) ? a.lengths[b] : a.t;
  const int t_in_hi = (a.flags & VITS_CONV_MASK_IN) ? (len < a.t ? len : a.t) : a.t;

  // output channel of LDS column `col` (0 <= col < TN) of this workgroup, or -1
  auto col_to_co = [&](int col) -> int {
    const int w_ = col / TNW, n = (col % TNW) / 32, c = col % 32;
    if (!gate) { const int co = co0 + col; return co < a.c_out ? co : -1; }
    const int half = n / (NT / 2 > 0 ? NT / 2 : 1), sub = n % (NT / 2 > 0 ? NT / 2 : 1);
    const int g = co0 + (w_ * (NT / 2) + sub) * 32 + c;            // gate channel
    return g < a.gate_h ? g + half * a.gate_h : -1;
  };

  f32x16 acc[NT];
#pragma unroll
  for (int n = 0; n < NT; ++n)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[n][i] = 0.f;

  const int n_groups = (a.k + args.G - 1) / args.G;
  const int n_chunks = (a.c_in + KC - 1) / KC;
  const int n_stages = n_groups * n_chunks;
  const int xvec = xrows * VPR;
  const bool x_in_regs = xvec <= kThreads * XV;        // else: stage X synchronously (long strided tiles)
  const bool lrelu_in = a.in_slope != 1.0f;

  u32x4 xr[XV], wr[WV];
  auto load_x = [&](int ci0) {
#pragma unroll
    for (int i = 0; i < XV; ++i) {
      const int idx = tid + i * kThreads;
      u32x4 v = {0u, 0u, 0u, 0u};
      if (idx < xvec) {
        const int row = idx / VPR, ch = idx % VPR;
        const int t = t0 * a.stride - a.pad + row, ci = ci0 + ch * V;
        if (t >= 0 && t < t_in_hi && ci < a.c_in) v = *reinterpret_cast<const u32x4*>(X + (size_t)t * a.ldx + ci);
      }
      xr[i] = v;
    }
  };
  auto store_x = [&]() {
#pragma unroll
    for (int i = 0; i < XV; ++i) {
      const int idx = tid + i * kThreads;
      // the fused input leaky-relu runs here, after the stage's MFMAs: the loads stay in flight during them
      if (idx < xvec) *reinterpret_cast<u32x4*>(ldsX + (idx / VPR) * PITCHK + (idx % VPR) * 16) = lrelu_in ? lrelu_vec<T>(xr[i], a.in_slope) : xr[i];
    }
  };
  auto stage_x_direct = [&](int ci0) {
    for (int idx = tid; idx < xvec; idx += kThreads) {
      const int row = idx / VPR, ch = idx % VPR;
      const int t = t0 * a.stride - a.pad + row, ci = ci0 + ch * V;
      u32x4 v = {0u, 0u, 0u, 0u};
      if (t >= 0 && t < t_in_hi && ci < a.c_in) {
        v = *reinterpret_cast<const u32x4*>(X + (size_t)t * a.ldx + ci);
        if (a.in_slope != 1.0f) v = lrelu_vec<T>(v, a.in_slope);
      }
      *reinterpret_cast<u32x4*>(ldsX + row * PITCHK + ch * 16) = v;
    }
  };
  auto load_w = [&](int ci0, int tg) {
    const int ntap = (a.k - tg < args.G) ? (a.k - tg) : args.G;
#pragma unroll
    for (int i = 0; i < WV; ++i) {
      const int idx = tid + i * kThreads;
      u32x4 v = {0u, 0u, 0u, 0u};
      if (idx < ntap * TN * VPR) {
        const int ch = idx % VPR, col = (idx / VPR) % TN, tl = (idx / VPR) / TN;
        const int co = col_to_co(col), ci = ci0 + ch * V;
        if (co >= 0 && ci < a.c_in) v = *reinterpret_cast<const u32x4*>(W + ((size_t)(tg + tl) * a.c_out + co) * a.ldw + ci);
      }
      wr[i] = v;
    }
  };
  auto store_w = [&](int tg) {
    const int ntap = (a.k - tg < args.G) ? (a.k - tg) : args.G;
#pragma unroll
    for (int i = 0; i < WV; ++i) {
      const int idx = tid + i * kThreads;
      if (idx < ntap * TN * VPR) *reinterpret_cast<u32x4*>(ldsW + (idx / VPR) * PITCHK + (idx % VPR) * 16) = wr[i];
    }
  };

  // one tap of the staged chunk: 4*NC k-steps of 32x32 MFMAs
  auto mma_tap = [&](int tap_abs, int tl) {
    const unsigned char* xa = ldsX + ((wm * 32 + r) * a.stride + tap_abs * a.dil) * PITCHK + 16 * h;
    const unsigned char* wb = ldsW + (tl * TN + wn * TNW + r) * PITCHK + 16 * h;
#pragma unroll
    for (int m = 0; m < 4 * NC; ++m) {
      const u32x4 av = *reinterpret_cast<const u32x4*>(xa + 32 * m);
#pragma unroll
      for (int n = 0; n < NT; ++n) {
        const u32x4 bv = *reinterpret_cast<const u32x4*>(wb + n * 32 * PITCHK + 32 * m);
        if constexpr (sizeof(T) == 2) {
          union { u32x4 u; bf16x8 v; } ua, ub;
          ua.u = av; ub.u = bv;
          acc[n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ua.v, ub.v, acc[n], 0, 0, 0);
        } else {
          union { u32x4 u; float f[4]; } ua, ub;
          ua.u = av; ub.u = bv;
#pragma unroll
          for (int j = 0; j < 4; ++j)
            acc[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(ua.f[j], ub.f[j], acc[n], 0, 0, 0);
        }
      }
    }
  };

  // ---- 1x1 layers with at most 3 channel chunks and a 64-row tile: ALL chunks' global loads are issued at once into the
  // prefetch registers (the same registers the pipeline below uses), then staged and multiplied chunk by chunk — one load
  // round trip instead of three for the ~430 small launches per step that are bound by exactly that latency.
  constexpr int XS = TMW / 32, WS = TN / 32;                 // 16-byte vectors per thread per chunk (k = 1, stride 1: xrows = TMW)
  constexpr bool CAN_PRELOAD = NC == 1 && 3 * XS <= XV && 3 * WS <= WV;
  bool done = false;
  if constexpr (CAN_PRELOAD) {
    if (a.k == 1 && a.stride == 1 && n_chunks <= 3 && args.G == 1) {
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        if (c < n_chunks) {
#pragma unroll
          for (int j = 0; j < XS; ++j) {
            const int idx = tid + j * kThreads;
            const int row = idx >> 3, ch = idx & 7;
            const int t = t0 - a.pad + row, ci = c * KC + ch * V;
            u32x4 v = {0u, 0u, 0u, 0u};
            if (t >= 0 && t < t_in_hi && ci < a.c_in) v = *reinterpret_cast<const u32x4*>(X + (size_t)t * a.ldx + ci);
            xr[c * XS + j] = v;
          }
#pragma unroll
          for (int j = 0; j < WS; ++j) {
            const int idx = tid + j * kThreads;
            const int ch = idx & 7, col = idx >> 3;
            const int co = col_to_co(col), ci = c * KC + ch * V;
            u32x4 v = {0u, 0u, 0u, 0u};
            if (co >= 0 && ci < a.c_in) v = *reinterpret_cast<const u32x4*>(W + (size_t)co * a.ldw + ci);
            wr[c * WS + j] = v;
          }
        }
      }
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        if (c < n_chunks) {
          if (c > 0) __syncthreads();                          // every wave is done reading the previous chunk's LDS
#pragma unroll
          for (int j = 0; j < XS; ++j) {
            const int idx = tid + j * kThreads;
            *reinterpret_cast<u32x4*>(ldsX + (idx >> 3) * PITCHK + (idx & 7) * 16) = lrelu_in ? lrelu_vec<T>(xr[c * XS + j], a.in_slope) : xr[c * XS + j];
          }
#pragma unroll
          for (int j = 0; j < WS; ++j) {
            const int idx = tid + j * kThreads;
            *reinterpret_cast<u32x4*>(ldsW + (idx >> 3) * PITCHK + (idx & 7) * 16) = wr[c * WS + j];
          }
          __syncthreads();
          mma_tap(0, 0);
        }
      }
      done = true;
    }
  }

  // ---- software pipeline: the global loads of stage s+1 are in flight while stage s is multiplied
  if (!done) {
  load_w(0, 0);
  if (x_in_regs) { load_x(0); store_x(); } else stage_x_direct(0);
  store_w(0);
  __syncthreads();
  }
  for (int s = 0; s < (done ? 0 : n_stages); ++s) {
    const int tg = (s % n_groups) * args.G;
    const int nxt = s + 1;
    const int nci0 = (nxt / n_groups) * KC, ntg = (nxt % n_groups) * args.G;
    const bool has_next = nxt < n_stages, new_chunk = has_next && (nxt % n_groups == 0);
    if (has_next) {
      load_w(nci0, ntg);
      if (new_chunk && x_in_regs) load_x(nci0);
    }
    const int ntap = (a.k - tg < args.G) ? (a.k - tg) : args.G;
    for (int tl = 0; tl < ntap; ++tl) mma_tap(tg + tl, tl);
    if (has_next) {
      __syncthreads();                       // every wave is done reading this stage's LDS
      store_w(ntg);
      if (new_chunk) { if (x_in_regs) store_x(); else stage_x_direct(nci0); }
      __syncthreads();
    }
  }

  // ---- epilogue: C/D layout of the 32x32 MFMA: col = lane & 31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)
  T* Y = static_cast<T*>(a.y) + (size_t)b * Tout * a.ldy;
  const T* R = a.res ? static_cast<const T*>(a.res) + (size_t)b * Tout * a.ldy : nullptr;
  const T* MG = a.mg_src ? static_cast<const T*>(a.mg_src) + (size_t)b * Tout * a.ldy : nullptr;

  if (gate) {
    // WaveNet gate (reference commons.py:103-110): tiles [0, NT/2) hold a = x_in[:, co], tiles [NT/2, NT) hold
    // b = x_in[:, co + H];  y = tanh(a) * sigmoid(b);  optionally y2 keeps the pre-activations (for backward).
    if constexpr (NT >= 2) {
      T* Y2 = a.y2 ? static_cast<T*>(a.y2) + (size_t)b * Tout * a.ldy2 : nullptr;
#pragma unroll
      for (int n = 0; n < NT / 2; ++n) {
        const int co = co0 + (wn * (NT / 2) + n) * 32 + r;
        if (co >= a.gate_h) continue;
        float ba = 0.f, bb = 0.f;
        if (a.bias) { ba += a.bias[co]; bb += a.bias[co + a.gate_h]; }
        if (a.bias_b) { ba += a.bias_b[(size_t)b * a.c_out + co]; bb += a.bias_b[(size_t)b * a.c_out + co + a.gate_h]; }
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          const int t = t0 + wm * 32 + (i & 3) + 8 * (i >> 2) + 4 * h;
          if (t >= Tout) continue;
          const float va = acc[n][i] + ba, vb = acc[n + NT / 2][i] + bb;
          if (Y2) {
            Y2[(size_t)t * a.ldy2 + co] = from_f<T>(va);
            Y2[(size_t)t * a.ldy2 + co + a.gate_h] = from_f<T>(vb);
          }
          Y[(size_t)t * a.ldy + co] = from_f<T>(gate_tanh<T>(va) * sigmoidf_(vb));
        }
      }
    }
    return;
  }

  if (a.flags & VITS_CONV_RES_SKIP) {
    // WaveNet res_skip layer: the first gate_h columns are the residual update, the rest is added into the skip sum (y2)
    T* Y2 = static_cast<T*>(a.y2) + (size_t)b * Tout * a.ldy2;
    const bool accum = (a.flags & VITS_CONV_ACCUM) != 0;
#pragma unroll
    for (int n = 0; n < NT; ++n) {
      const int co = co0 + (wn * NT + n) * 32 + r;
      if (co >= a.c_out) continue;
      const float bsum = a.bias ? a.bias[co] : 0.f;
      const bool is_res = co < a.gate_h;
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int t = t0 + wm * 32 + (i & 3) + 8 * (i >> 2) + 4 * h;
        if (t >= Tout) continue;
        float v = acc[n][i] + bsum;
        if (is_res) {
          const size_t o = (size_t)t * a.ldy + co;
          v += to_f(R[o]);
          Y[o] = from_f<T>(t < len ? v : 0.f);
        } else {
          const size_t o = (size_t)t * a.ldy2 + (co - a.gate_h);
          v = t < len ? v : 0.f;
          if (accum) v += to_f(Y2[o]);
          Y2[o] = from_f<T>(v);
        }
      }
    }
    return;
  }

  // the common epilogue (bias, scale, output leaky-relu, output mask: no memory operand besides the bias) without a flag test
  // per element — most of the step's ~400 small launches end here, and the generic loop below spends a branch per flag per element
  if (!R && !MG && !a.bias_b && !(a.flags & (VITS_CONV_ACCUM | VITS_CONV_TANH | VITS_CONV_GATE_BWD))) {
    const float oslope = (a.flags & VITS_CONV_OUT_LRELU) ? a.out_slope : 1.0f;
    const int t_hi = ((a.flags & VITS_CONV_MASK_OUT) && len < Tout) ? len : Tout;     // rows >= t_hi (and < Tout) are written as zero
#pragma unroll
    for (int n = 0; n < NT; ++n) {
      const int co = co0 + (wn * NT + n) * 32 + r;
      if (co >= a.c_out) continue;
      const float bsum = a.bias ? a.bias[co] : 0.f;
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int t = t0 + wm * 32 + (i & 3) + 8 * (i >> 2) + 4 * h;
        if (t >= Tout) continue;
        float v = (acc[n][i] + bsum) * a.out_scale;
        v = v > 0.f ? v : v * oslope;
        Y[(size_t)t * a.ldy + co] = from_f<T>(t < t_hi ? v : 0.f);
      }
    }
    return;
  }
#pragma unroll
  for (int n = 0; n < NT; ++n) {
    const int co = co0 + (wn * NT + n) * 32 + r;
    if (co >= a.c_out) continue;
    float bsum = 0.f;
    if (a.bias) bsum += a.bias[co];
    if (a.bias_b) bsum += a.bias_b[(size_t)b * a.c_out + co];
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int t = t0 + wm * 32 + (i & 3) + 8 * (i >> 2) + 4 * h;
      if (t >= Tout) continue;
      const size_t o = (size_t)t * a.ldy + co;
      float v = acc[n][i] + bsum;
      const bool res_after = (a.flags & VITS_CONV_RES_AFTER) != 0;
      if (R && !res_after) v += to_f(R[o]);
      v *= a.out_scale;
      if (a.flags & VITS_CONV_GATE_BWD) {
        // chain rule of the gate: v = d(acts[:, co]);  mg_src = saved pre-activations [.., 2H]
        const float ta = gate_tanh<T>(to_f(MG[o])), sb = sigmoidf_(to_f(MG[o + a.gate_h]));
        const bool dead = (a.flags & VITS_CONV_MASK_OUT) && t >= len;
        Y[o] = from_f<T>(dead ? 0.f : v * sb * (1.0f - ta * ta));
        Y[o + a.gate_h] = from_f<T>(dead ? 0.f : v * ta * sb * (1.0f - sb));
        continue;
      }
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

template <typename T, int NT, int WM, int NC = 1>
int launch_conv(const vits_conv_desc& d, int t_out, hipStream_t s) {
  ConvArgs args{d, t_out, 1};
  constexpr int WN = 4 / WM, TMW = 32 * WM, TN = 32 * NT * WN;
  constexpr int PITCHK = ROWB * NC + 16;
  const int xrows = (TMW - 1) * d.stride + (d.k - 1) * d.dil + 1;
  int G = (kThreads * WV_MAX) / (TN * 8);        // taps per W stage: what one prefetch round can hold (<= 41 KB)
  if (G < 1 || NC > 1) G = 1;
  if (G > d.k) G = d.k;
  args.G = G;
  const size_t lds = (size_t)xrows * PITCHK + (size_t)G * TN * PITCHK;
  if (lds > (size_t)vits::kLdsBytesMax) return VITS_E_UNSUPPORTED;
  auto kern = conv1d_cl_kernel<T, NT, WM, NC>;
  { const hipError_t e = vits::ensure_max_dynamic_lds(reinterpret_cast<const void*>(kern)); if (e != hipSuccess) return vits::note_hip_error(e, "vits_conv1d_cl/attr"); }
  const bool gate = (d.flags & VITS_CONV_GATE) != 0;
  const int cols = gate ? d.gate_h : d.c_out;
  dim3 grid(vits::ceil_div(t_out, TMW), vits::ceil_div(cols, gate ? TN / 2 : TN), d.b);
  hipLaunchKernelGGL(kern, grid, dim3(kThreads), lds, s, args);
  return vits::check_launch("vits_conv1d_cl");
}

// Tile choice: the largest tile that still gives the chip >= 2 workgroups per CU, else the smallest.
// (time rows x channels): 128x128, 128x64, 64x128(2x2 waves), 64x64, 128x32
template <typename T>
int dispatch_tile(const vits_conv_desc& d, int t_out, hipStream_t s) {
  const bool gate = (d.flags & VITS_CONV_GATE) != 0;
  const int cols = gate ? 2 * d.gate_h : d.c_out;
  auto wgs = [&](int tm, int tn) { return (long)vits::ceil_div(t_out, tm) * vits::ceil_div(cols, tn) * d.b; };
  const long want = 2 * 256;
  if (gate) {                                    // needs an even number of tiles per wave
    if (wgs(128, 128) >= want || cols <= 64) return launch_conv<T, 4, 4>(d, t_out, s);
    return launch_conv<T, 2, 2>(d, t_out, s);    // 64 rows x (2 waves x 64 columns)
  }
  if (cols > 64 && wgs(128, 128) >= want) return launch_conv<T, 4, 4>(d, t_out, s);
  if (cols > 64 && wgs(64, 128) >= want) return launch_conv<T, 2, 2>(d, t_out, s);
  if (cols > 32 && wgs(128, 64) >= want) return launch_conv<T, 2, 4>(d, t_out, s);
  if (cols > 32) return launch_conv<T, 1, 2>(d, t_out, s);           // 64 x 64
  return launch_conv<T, 1, 4>(d, t_out, s);                           // 128 x 32
}


// ---- products with very few rows and a long reduction (the conditioning layers' data gradient: [16] x [6144] -> [256], K = 6144):
// the tiled kernels would put 4 workgroups on the chip and walk ~100 channel chunks each.  Here one WORKGROUP owns one output
// channel: its 256 lanes stride over the reduction (16-byte loads of W and of every row of X), fp32 sums per row, a fixed-order
// butterfly across the lanes of a wave, the four waves' sums added in wave order through LDS; lane m writes row m.
// Deterministic; bias and scale as in the common epilogue.
constexpr int SMALLM_ROWS = 16;
__global__ __launch_bounds__(256) void small_m_kernel(const __bf16* __restrict__ X, const __bf16* __restrict__ W, const float* __restrict__ bias,
                                                      __bf16* __restrict__ Y, int M, int N, int Kc, int ldx, int ldw, int ldy, float scale) {
  __shared__ float red[4][SMALLM_ROWS];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int n = blockIdx.x;
  float acc[SMALLM_ROWS];
#pragma unroll
  for (int m = 0; m < SMALLM_ROWS; ++m) acc[m] = 0.f;
  const __bf16* wrow = W + (size_t)n * ldw;
  for (int kk = threadIdx.x * 8; kk < Kc; kk += 256 * 8) {
    union { u32x4 u; __bf16 e[8]; } wv;
    wv.u = *reinterpret_cast<const u32x4*>(wrow + kk);
    float wf[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) wf[j] = (float)wv.e[j];
#pragma unroll
    for (int m = 0; m < SMALLM_ROWS; ++m) {
      if (m < M) {
        union { u32x4 u; __bf16 e[8]; } xv;
        xv.u = *reinterpret_cast<const u32x4*>(X + (size_t)m * ldx + kk);
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[m] = fmaf(wf[j], (float)xv.e[j], acc[m]);
      }
    }
  }
#pragma unroll
  for (int m = 0; m < SMALLM_ROWS; ++m) {
    float v = acc[m];
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
    if (lane == 0) red[wave][m] = v;
  }
  __syncthreads();
  if (threadIdx.x < M) {
    const int m = threadIdx.x;
    const float v = ((red[0][m] + red[1][m]) + red[2][m]) + red[3][m];
    Y[(size_t)m * ldy + n] = (__bf16)((v + (bias ? bias[n] : 0.f)) * scale);
  }
}

bool small_m_applies(const vits_conv_desc& d, int t_out) {
  return d.dtype == VITS_DT_BF16 && d.k == 1 && d.stride == 1 && d.in_div <= 1 && d.groups <= 1 && d.pad == 0 && d.w_batch_stride == 0 &&
         (long)d.b * t_out <= SMALLM_ROWS && t_out == d.t && d.c_in >= 1024 && d.c_in % 8 == 0 && d.ldx % 8 == 0 && d.ldw % 8 == 0 &&
         d.flags == 0 && d.in_slope == 1.0f && !d.res && !d.mg_src && !d.bias_b && !d.y2;
}

}  // namespace

namespace {
// Validation and defaults shared by the entry points: VITS_OK and the output length, or the error to return.
int normalize_desc(vits_conv_desc& d, int& t_out) {
  if (!d.x || !d.w || !d.y || d.b <= 0 || d.t <= 0 || d.c_in <= 0 || d.c_out <= 0 || d.k <= 0 || d.dil <= 0 || d.pad < 0)
    return VITS_E_BADARG;
  if (d.stride <= 0) d.stride = 1;
  const int in_div = d.in_div > 1 ? d.in_div : 1;
  if (in_div > 1) {                         // data gradient of a strided convolution: the caller states the output length
    if (d.stride != 1 || d.t_out_override <= 0) return VITS_E_BADARG;
    t_out = d.t_out_override;
  } else {
    const int span = d.t + 2 * d.pad - d.dil * (d.k - 1) - 1;
    if (span < 0) return VITS_E_BADARG;
    t_out = span / d.stride + 1;
  }
  if (((d.flags & (VITS_CONV_MASK_IN | VITS_CONV_MASK_OUT)) != 0) && !d.lengths) return VITS_E_BADARG;
  const bool gate = (d.flags & VITS_CONV_GATE) != 0, gate_bwd = (d.flags & VITS_CONV_GATE_BWD) != 0;
  if (gate && (d.gate_h <= 0 || d.c_out != 2 * d.gate_h)) return VITS_E_BADARG;
  if (gate_bwd && (d.gate_h != d.c_out || !d.mg_src)) return VITS_E_BADARG;
  const bool res_skip = (d.flags & VITS_CONV_RES_SKIP) != 0;
  if (res_skip && (d.gate_h <= 0 || d.c_out != 2 * d.gate_h || !d.y2 || !d.res || !d.lengths || gate || gate_bwd || d.mg_src || d.bias_b)) return VITS_E_BADARG;
  if (d.ldx <= 0) d.ldx = d.c_in;
  if (d.ldw <= 0) d.ldw = d.c_in;
  if (d.w_batch_stride < 0) return VITS_E_BADARG;
  if (d.ldy <= 0) d.ldy = (gate || res_skip) ? d.gate_h : (gate_bwd ? 2 * d.gate_h : d.c_out);
  if (d.ldy2 <= 0) d.ldy2 = res_skip ? d.gate_h : d.c_out;
  const int vec = d.dtype == VITS_DT_BF16 ? 8 : (d.dtype == VITS_DT_F32 ? 4 : 0);
  if (vec == 0) return VITS_E_UNSUPPORTED;
  if (d.c_in % vec != 0 || d.ldx % vec != 0 || d.ldw % vec != 0 || d.w_batch_stride % vec != 0) return VITS_E_UNSUPPORTED;
  if (d.groups > 1 && (d.c_out % d.groups != 0 || d.c_in % d.groups != 0)) return VITS_E_BADARG;
  return VITS_OK;
}

// the launches the deep-prefetch ring kernel is tried for (csrc/conv1d_ring.hip decides the rest)
bool ring_candidate(const vits_conv_desc& d, int t_out) {
  const bool plain = !(d.flags & (VITS_CONV_GATE | VITS_CONV_GATE_BWD | VITS_CONV_RES_SKIP)) && d.w_batch_stride == 0 && d.y2 == nullptr;
  return d.dtype == VITS_DT_BF16 && plain && d.groups <= 1 && d.k >= 2 && d.c_in >= 128 && d.c_out >= 96 && !small_m_applies(d, t_out) &&
         !(d.flags & VITS_CONV_FLAT);
}
}  // namespace

extern "C" int vits_conv1d_cl_multi(const vits_conv_desc* descs, int count, void* stream) {
  if (!descs || count <= 0) return VITS_E_BADARG;
  if (count < 2 || count > 8) return VITS_E_UNSUPPORTED;
  vits_conv_desc d[8];
  int t_out[8];
  for (int i = 0; i < count; ++i) {
    d[i] = descs[i];
    const int rc = normalize_desc(d[i], t_out[i]);
    if (rc != VITS_OK) return rc;
    if (!ring_candidate(d[i], t_out[i])) return VITS_E_UNSUPPORTED;
  }
  return vits::conv1d_ring_multi_dispatch(d, t_out, count, static_cast<hipStream_t>(stream));
}

extern "C" int vits_conv1d_cl(const vits_conv_desc* desc, void* stream) {
  if (!desc) return VITS_E_BADARG;
  vits_conv_desc d = *desc;
  int t_out;
  { const int rc = normalize_desc(d, t_out); if (rc != VITS_OK) return rc; }
  hipStream_t s = static_cast<hipStream_t>(stream);
  const int in_div = d.in_div > 1 ? d.in_div : 1;
  const bool gate = (d.flags & VITS_CONV_GATE) != 0, gate_bwd = (d.flags & VITS_CONV_GATE_BWD) != 0;
  const bool res_skip = (d.flags & VITS_CONV_RES_SKIP) != 0;
  if (small_m_applies(d, t_out)) {
    hipLaunchKernelGGL(small_m_kernel, dim3(d.c_out), dim3(256), 0, s, static_cast<const __bf16*>(d.x), static_cast<const __bf16*>(d.w),
                       d.bias, static_cast<__bf16*>(d.y), d.b * t_out, d.c_out, d.c_in, d.ldx, d.ldw, d.ldy, d.out_scale);
    return vits::check_launch("vits_conv1d_cl/small_m");
  }
  // flat-row kernel: strided / divided launches, and short sequences spread over many items (most of a per-item
  // time tile would be empty).  It has no gate epilogues and no per-item operands.
  const bool flat_ok = !gate && !gate_bwd && !res_skip && d.w_batch_stride == 0 && d.y2 == nullptr;
  const bool must_flat = in_div > 1 || (d.flags & VITS_CONV_FLAT) != 0 || d.groups > 1;
  if (must_flat && !flat_ok) return VITS_E_UNSUPPORTED;
  const bool auto_flat = true;
  // deep-prefetch ring kernel (csrc/conv1d_ring.hip) for layers with >= 128 input channels and k >= 2: measured faster than or
  // equal to the one-stage-ahead kernels on every such shape of the step (tools/ubench_conv.py; 2.3x on the 1024-channel layers)
  if (d.dtype == VITS_DT_BF16 && flat_ok && d.groups <= 1 && d.k >= 2 && d.c_in >= 128 && d.c_out >= 96) {
    const int rc = vits::conv1d_ring_dispatch(d, t_out, s);
    if (rc != VITS_E_UNSUPPORTED) return rc;
  }
  if (must_flat || (flat_ok && (d.stride > 1 || (auto_flat && t_out <= 80 && d.b >= 8)))) {
    const int rc = vits::conv1d_flat_dispatch(d, t_out, s);
    if (rc != VITS_E_UNSUPPORTED || must_flat) return rc;
  }
  if (d.dtype == VITS_DT_BF16) return dispatch_tile<__bf16>(d, t_out, s);
  return dispatch_tile<float>(d, t_out, s);
}
