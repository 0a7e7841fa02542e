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
  const void* x; const void* w; const float* bias; const float* bias_b; const void* res;
  const void* mg_src; void* y; const int* lengths;
  int B, T, Tout, Cin, Cout, K, dil, pad, G;   // G = taps staged per group
  float in_slope, mg_slope, out_scale;
  int flags;
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

template <typename T, int NT>
__global__ __launch_bounds__(kThreads) void conv1d_cl_kernel(ConvArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int V = Elem<T>::VEC;
  constexpr int KC = Elem<T>::KC;
  constexpr int TN = 32 * NT;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int r = lane & 31, h = lane >> 5;
  const int t0 = blockIdx.x * TM;
  const int co0 = blockIdx.y * TN;
  const int b = blockIdx.z;
  const int xrows = TM + (a.K - 1) * a.dil;
  unsigned char* ldsX = smem;
  unsigned char* ldsW = smem + (size_t)xrows * PITCH;

  const T* X = static_cast<const T*>(a.x) + (size_t)b * a.T * a.Cin;
  const T* W = static_cast<const T*>(a.w);
  const int len = (a.lengths != nullptr) ? a.lengths[b] : a.T;
  const int t_in_hi = (a.flags & VITS_CONV_MASK_IN) ? (len < a.T ? len : a.T) : a.T;

  f32x16 acc[NT];
#pragma unroll
  for (int n = 0; n < NT; ++n)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[n][i] = 0.f;

  for (int ci0 = 0; ci0 < a.Cin; ci0 += KC) {
    for (int tg = 0; tg < a.K; tg += a.G) {
      __syncthreads();                       // previous compute finished with ldsW (and ldsX)
      if (tg == 0) {
        // ---- stage X rows [t0 - pad, t0 - pad + xrows) x channels [ci0, ci0 + KC)
        for (int idx = tid; idx < xrows * 8; idx += kThreads) {
          const int row = idx >> 3, ch = idx & 7;
          const int t = t0 - a.pad + row;
          const int ci = ci0 + ch * V;
          u32x4 v = {0u, 0u, 0u, 0u};
          if (t >= 0 && t < t_in_hi && ci < a.Cin) {
            v = *reinterpret_cast<const u32x4*>(X + (size_t)t * a.Cin + ci);
            if (a.in_slope != 1.0f) v = lrelu_vec<T>(v, a.in_slope);
          }
          *reinterpret_cast<u32x4*>(ldsX + row * PITCH + ch * 16) = v;
        }
      }
      // ---- stage W[tg .. tg+G)[co0 .. co0+TN)[ci0 .. ci0+KC)
      const int ntap = (a.K - tg < a.G) ? (a.K - tg) : a.G;
      for (int idx = tid; idx < ntap * TN * 8; idx += kThreads) {
        const int ch = idx & 7;
        const int col = (idx >> 3) % TN;
        const int tl = (idx >> 3) / TN;
        const int co = co0 + col, ci = ci0 + ch * V;
        u32x4 v = {0u, 0u, 0u, 0u};
        if (co < a.Cout && ci < a.Cin)
          v = *reinterpret_cast<const u32x4*>(W + ((size_t)(tg + tl) * a.Cout + co) * a.Cin + ci);
        *reinterpret_cast<u32x4*>(ldsW + (tl * TN + col) * PITCH + ch * 16) = v;
      }
      __syncthreads();
      // ---- MFMA over the staged taps and the chunk's 4 macro-steps
      for (int tl = 0; tl < ntap; ++tl) {
        const unsigned char* xa = ldsX + (wave * 32 + r + (tg + tl) * a.dil) * PITCH + 16 * h;
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
  T* Y = static_cast<T*>(a.y) + (size_t)b * a.Tout * a.Cout;
  const T* R = a.res ? static_cast<const T*>(a.res) + (size_t)b * a.Tout * a.Cout : nullptr;
  const T* MG = a.mg_src ? static_cast<const T*>(a.mg_src) + (size_t)b * a.Tout * a.Cout : nullptr;
#pragma unroll
  for (int n = 0; n < NT; ++n) {
    const int co = co0 + n * 32 + r;
    if (co >= a.Cout) continue;
    float bsum = 0.f;
    if (a.bias) bsum += a.bias[co];
    if (a.bias_b) bsum += a.bias_b[(size_t)b * a.Cout + co];
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int t = t0 + wave * 32 + (i & 3) + 8 * (i >> 2) + 4 * h;
      if (t >= a.Tout) continue;
      const size_t o = (size_t)t * a.Cout + co;
      float v = acc[n][i] + bsum;
      const bool res_after = (a.flags & VITS_CONV_RES_AFTER) != 0;
      if (R && !res_after) v += to_f(R[o]);
      v *= a.out_scale;
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
int launch_conv(const ConvArgs& a, hipStream_t s) {
  ConvArgs args = a;
  const int TN = 32 * NT;
  const int xrows = TM + (a.K - 1) * a.dil;
  // taps per W stage: keep the W slab under ~40 KB
  int G = (40 * 1024) / (TN * PITCH);
  if (G < 1) G = 1;
  if (G > a.K) G = a.K;
  args.G = G;
  const size_t lds = (size_t)xrows * PITCH + (size_t)G * TN * PITCH;
  if (lds > (size_t)vits::kLdsBytesMax) return VITS_E_UNSUPPORTED;
  auto kern = conv1d_cl_kernel<T, NT>;
  if (lds > 64 * 1024) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return vits::note_hip_error(e, "vits_conv1d_cl/attr");
  }
  dim3 grid(vits::ceil_div(a.Tout, TM), vits::ceil_div(a.Cout, TN), a.B);
  hipLaunchKernelGGL(kern, grid, dim3(kThreads), lds, s, args);
  return vits::check_launch("vits_conv1d_cl");
}

template <typename T>
int dispatch_nt(const ConvArgs& a, hipStream_t s) {
  if (a.Cout > 64) return launch_conv<T, 4>(a, s);
  if (a.Cout > 32) return launch_conv<T, 2>(a, s);
  return launch_conv<T, 1>(a, s);
}

}  // namespace

extern "C" int vits_conv1d_cl(int dtype, const void* x, const void* w, const float* bias, const float* bias_b,
                              const void* res, const void* mg_src, void* y, const int32_t* lengths,
                              int b, int t, int c_in, int c_out, int k, int dil, int pad,
                              float in_slope, float mg_slope, float out_scale, int flags, void* stream) {
  if (!x || !w || !y || b <= 0 || t <= 0 || c_in <= 0 || c_out <= 0 || k <= 0 || dil <= 0 || pad < 0) return VITS_E_BADARG;
  const int t_out = t + 2 * pad - dil * (k - 1);
  if (t_out <= 0) return VITS_E_BADARG;
  if (((flags & (VITS_CONV_MASK_IN | VITS_CONV_MASK_OUT)) != 0) && !lengths) return VITS_E_BADARG;
  if ((res || mg_src) && t_out != t) return VITS_E_UNSUPPORTED;
  ConvArgs a{x, w, bias, bias_b, res, mg_src, y, lengths, b, t, t_out, c_in, c_out, k, dil, pad, 1,
             in_slope, mg_slope, out_scale, flags};
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (dtype == VITS_DT_BF16) {
    if (c_in % 8 != 0) return VITS_E_UNSUPPORTED;
    return dispatch_nt<__bf16>(a, s);
  }
  if (dtype == VITS_DT_F32) {
    if (c_in % 4 != 0) return VITS_E_UNSUPPORTED;
    return dispatch_nt<float>(a, s);
  }
  return VITS_E_UNSUPPORTED;
}
