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
// Workgroup = 4 waves = 128 time rows x (32*NT) output channels; per 128-byte input-channel chunk
// the X rows (with halo, input activation and row mask applied once) and a group of taps of W are
// staged in LDS with a 16-byte row pad (pitch 144 B: conflict-free ds_read_b128).
//
// Fused epilogue (flags): + bias[co] + bias_b[b][co] (speaker conditioning) + residual, * scale,
// * leaky-relu'(src) (chain rule of a fused input activation, for the data-gradient call),
// residual after the multiplier (skip connection of a data gradient), row mask, tanh, accumulate.
#include "common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

constexpr int TM = 128;            // time rows per workgroup
constexpr int ROWB = 128;          // bytes of input channels staged per chunk
constexpr int PITCH = ROWB + 16;   // LDS row pitch
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

template <typename T, int NT>
__global__ __launch_bounds__(kThreads) void conv1d_cl_kernel(ConvArgs args) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const vits_conv_desc& a = args.d;
  constexpr int V = Elem<T>::VEC;
  constexpr int KC = Elem<T>::KC;
  constexpr int TN = 32 * NT;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int r = lane & 31, h = lane >> 5;
  const int t0 = blockIdx.x * TM;
  const int b = blockIdx.z;
  const int Tout = args.Tout;
  const bool gate = (a.flags & VITS_CONV_GATE) != 0;
  // output-channel origin of this workgroup: in GATE mode the NT tiles are split in two halves
  // (columns co0.. of the tanh half and co0 + gate_h.. of the sigmoid half).
  const int co0 = blockIdx.y * (gate ? TN / 2 : TN);
  const int xrows = (TM - 1) * a.stride + (a.k - 1) * a.dil + 1;
  unsigned char* ldsX = smem;
  unsigned char* ldsW = smem + (size_t)xrows * PITCH;

  const T* X = static_cast<const T*>(a.x) + (size_t)b * a.t * a.ldx;
  const T* W = static_cast<const T*>(a.w);
  const int len = (a.lengths != nullptr) ? a.lengths[b] : a.t;
  const int t_in_hi = (a.flags & VITS_CONV_MASK_IN) ? (len < a.t ? len : a.t) : a.t;

  f32x16 acc[NT];
#pragma unroll
  for (int n = 0; n < NT; ++n)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[n][i] = 0.f;

  for (int ci0 = 0; ci0 < a.c_in; ci0 += KC) {
    for (int tg = 0; tg < a.k; tg += args.G) {
      __syncthreads();                       // previous compute finished with ldsW (and ldsX)
      if (tg == 0) {
        // ---- stage X rows [t0*stride - pad, ... + xrows) x channels [ci0, ci0 + KC)
        for (int idx = tid; idx < xrows * 8; idx += kThreads) {
          const int row = idx >> 3, ch = idx & 7;
          const int t = t0 * a.stride - a.pad + row;
          const int ci = ci0 + ch * V;
          u32x4 v = {0u, 0u, 0u, 0u};
          if (t >= 0 && t < t_in_hi && ci < a.c_in) {
            v = *reinterpret_cast<const u32x4*>(X + (size_t)t * a.ldx + ci);
            if (a.in_slope != 1.0f) v = lrelu_vec<T>(v, a.in_slope);
          }
          *reinterpret_cast<u32x4*>(ldsX + row * PITCH + ch * 16) = v;
        }
      }
      // ---- stage W[tg .. tg+G)[this workgroup's TN output channels][ci0 .. ci0+KC)
      const int ntap = (a.k - tg < args.G) ? (a.k - tg) : args.G;
      for (int idx = tid; idx < ntap * TN * 8; idx += kThreads) {
        const int ch = idx & 7;
        const int col = (idx >> 3) % TN;
        const int tl = (idx >> 3) / TN;
        int co = co0 + col;
        bool ok = co < a.c_out;
        if (gate) {
          const int half = col / (TN / 2), sub = col % (TN / 2);
          co = co0 + sub + half * a.gate_h;
          ok = (co0 + sub) < a.gate_h;
        }
        const int ci = ci0 + ch * V;
        u32x4 v = {0u, 0u, 0u, 0u};
        if (ok && ci < a.c_in)
          v = *reinterpret_cast<const u32x4*>(W + ((size_t)(tg + tl) * a.c_out + co) * a.c_in + ci);
        *reinterpret_cast<u32x4*>(ldsW + (tl * TN + col) * PITCH + ch * 16) = v;
      }
      __syncthreads();
      // ---- MFMA over the staged taps and the chunk's 4 macro-steps
      for (int tl = 0; tl < ntap; ++tl) {
        const unsigned char* xa = ldsX + ((wave * 32 + r) * a.stride + (tg + tl) * a.dil) * PITCH + 16 * h;
        const unsigned char* wb = ldsW + (tl * TN + r) * PITCH + 16 * h;
#pragma unroll
        for (int m = 0; m < 4; ++m) {
          const u32x4 av = *reinterpret_cast<const u32x4*>(xa + 32 * m);
#pragma unroll
          for (int n = 0; n < NT; ++n) {
            const u32x4 bv = *reinterpret_cast<const u32x4*>(wb + n * 32 * PITCH + 32 * m);
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
      }
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
        const int co = co0 + n * 32 + r;
        if (co >= a.gate_h) continue;
        float ba = 0.f, bb = 0.f;
        if (a.bias) { ba += a.bias[co]; bb += a.bias[co + a.gate_h]; }
        if (a.bias_b) { ba += a.bias_b[(size_t)b * a.c_out + co]; bb += a.bias_b[(size_t)b * a.c_out + co + a.gate_h]; }
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          const int t = t0 + wave * 32 + (i & 3) + 8 * (i >> 2) + 4 * h;
          if (t >= Tout) continue;
          const float va = acc[n][i] + ba, vb = acc[n + NT / 2][i] + bb;
          if (Y2) {
            Y2[(size_t)t * a.ldy2 + co] = from_f<T>(va);
            Y2[(size_t)t * a.ldy2 + co + a.gate_h] = from_f<T>(vb);
          }
          Y[(size_t)t * a.ldy + co] = from_f<T>(tanhf(va) * sigmoidf_(vb));
        }
      }
    }
    return;
  }

#pragma unroll
  for (int n = 0; n < NT; ++n) {
    const int co = co0 + n * 32 + r;
    if (co >= a.c_out) continue;
    float bsum = 0.f;
    if (a.bias) bsum += a.bias[co];
    if (a.bias_b) bsum += a.bias_b[(size_t)b * a.c_out + co];
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int t = t0 + wave * 32 + (i & 3) + 8 * (i >> 2) + 4 * h;
      if (t >= Tout) continue;
      const size_t o = (size_t)t * a.ldy + co;
      float v = acc[n][i] + bsum;
      const bool res_after = (a.flags & VITS_CONV_RES_AFTER) != 0;
      if (R && !res_after) v += to_f(R[o]);
      v *= a.out_scale;
      if (a.flags & VITS_CONV_GATE_BWD) {
        // chain rule of the gate: v = d(acts[:, co]);  mg_src = saved pre-activations [.., 2H]
        const float ta = tanhf(to_f(MG[o])), sb = sigmoidf_(to_f(MG[o + a.gate_h]));
        const bool dead = (a.flags & VITS_CONV_MASK_OUT) && t >= len;
        Y[o] = from_f<T>(dead ? 0.f : v * sb * (1.0f - ta * ta));
        Y[o + a.gate_h] = from_f<T>(dead ? 0.f : v * ta * sb * (1.0f - sb));
        continue;
      }
      if (MG) v *= (to_f(MG[o]) > 0.f) ? 1.0f : a.mg_slope;
      if (R && res_after) v += to_f(R[o]);
      if (a.flags & VITS_CONV_TANH) v = tanhf(v);
      if ((a.flags & VITS_CONV_MASK_OUT) && t >= len) v = 0.f;
      if (a.flags & VITS_CONV_ACCUM) v += to_f(Y[o]);
      Y[o] = from_f<T>(v);
    }
  }
}

template <typename T, int NT>
int launch_conv(const vits_conv_desc& d, int t_out, hipStream_t s) {
  ConvArgs args{d, t_out, 1};
  const int TN = 32 * NT;
  const int xrows = (TM - 1) * d.stride + (d.k - 1) * d.dil + 1;
  int G = (40 * 1024) / (TN * PITCH);            // taps per W stage: keep the W slab under ~40 KB
  if (G < 1) G = 1;
  if (G > d.k) G = d.k;
  args.G = G;
  const size_t lds = (size_t)xrows * PITCH + (size_t)G * TN * PITCH;
  if (lds > (size_t)vits::kLdsBytesMax) return VITS_E_UNSUPPORTED;
  auto kern = conv1d_cl_kernel<T, NT>;
  if (lds > 64 * 1024) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return vits::note_hip_error(e, "vits_conv1d_cl/attr");
  }
  const bool gate = (d.flags & VITS_CONV_GATE) != 0;
  const int cols = gate ? d.gate_h : d.c_out;
  dim3 grid(vits::ceil_div(t_out, TM), vits::ceil_div(cols, gate ? TN / 2 : TN), d.b);
  hipLaunchKernelGGL(kern, grid, dim3(kThreads), lds, s, args);
  return vits::check_launch("vits_conv1d_cl");
}

template <typename T>
int dispatch_nt(const vits_conv_desc& d, int t_out, hipStream_t s) {
  if (d.flags & VITS_CONV_GATE) return launch_conv<T, 4>(d, t_out, s);
  if (d.c_out > 64) return launch_conv<T, 4>(d, t_out, s);
  if (d.c_out > 32) return launch_conv<T, 2>(d, t_out, s);
  return launch_conv<T, 1>(d, t_out, s);
}

}  // namespace

extern "C" int vits_conv1d_cl(const vits_conv_desc* desc, void* stream) {
  if (!desc) return VITS_E_BADARG;
  vits_conv_desc d = *desc;
  if (!d.x || !d.w || !d.y || d.b <= 0 || d.t <= 0 || d.c_in <= 0 || d.c_out <= 0 || d.k <= 0 || d.dil <= 0 || d.pad < 0)
    return VITS_E_BADARG;
  if (d.stride <= 0) d.stride = 1;
  const int span = d.t + 2 * d.pad - d.dil * (d.k - 1) - 1;
  if (span < 0) return VITS_E_BADARG;
  const int t_out = span / d.stride + 1;
  if (((d.flags & (VITS_CONV_MASK_IN | VITS_CONV_MASK_OUT)) != 0) && !d.lengths) return VITS_E_BADARG;
  const bool gate = (d.flags & VITS_CONV_GATE) != 0, gate_bwd = (d.flags & VITS_CONV_GATE_BWD) != 0;
  if (gate && (d.gate_h <= 0 || d.c_out != 2 * d.gate_h)) return VITS_E_BADARG;
  if (gate_bwd && (d.gate_h != d.c_out || !d.mg_src)) return VITS_E_BADARG;
  if (d.ldx <= 0) d.ldx = d.c_in;
  if (d.ldy <= 0) d.ldy = gate ? d.gate_h : (gate_bwd ? 2 * d.gate_h : d.c_out);
  if (d.ldy2 <= 0) d.ldy2 = d.c_out;
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (d.dtype == VITS_DT_BF16) {
    if (d.c_in % 8 != 0 || d.ldx % 8 != 0) return VITS_E_UNSUPPORTED;
    return dispatch_nt<__bf16>(d, t_out, s);
  }
  if (d.dtype == VITS_DT_F32) {
    if (d.c_in % 4 != 0 || d.ldx % 4 != 0) return VITS_E_UNSUPPORTED;
    return dispatch_nt<float>(d, t_out, s);
  }
  return VITS_E_UNSUPPORTED;
}
