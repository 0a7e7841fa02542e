// Piecewise rational-quadratic spline with linear tails, forward / inverse and the full backward,
// one thread per element with the element's 29 parameters in registers.
//
// Replaces reference transforms.py:12-193 (piecewise_rational_quadratic_transform ->
// unconstrained_rational_quadratic_spline -> rational_quadratic_spline) as called from
// modules.ConvFlow.forward (modules.py:375-384) with tails='linear', num_bins=10: ~40 small torch
// kernels plus boolean-mask gathers with host synchronisation per call there, one launch here
// (and one for the backward, which torch autograd spreads over ~100 more).
//
// Per element (row n of h holds 10 widths, 10 heights, 9 interior derivatives, as produced by the
// ConvFlow projection; widths/heights are scaled by `hscale` = 1/sqrt(filter_channels)):
//   p = softmax(u);  size = 1e-3 + (1 - 1e-2) p;  knots = [-B, 2B cumsum(size) - B ..., B] (ends pinned)
//   d[0] = d[10] = 1e-3 + softplus(log(exp(1 - 1e-3) - 1)),  d[j] = 1e-3 + softplus(ud[j-1])
//   bin = #(x >= knot_j, last knot + 1e-6) - 1                      (transforms.py:47-52 searchsorted)
//   forward  (transforms.py:178-193) / inverse (transforms.py:152-176: root = 2c / (-b - sqrt(b^2 - 4ac)))
//   outside [-B, B] (inclusive test, :65): y = x, logabsdet = 0.
// The backward is the hand-derived reverse-mode of exactly these formulas.
#include "common.h"

namespace {

constexpr int NB = 10;
constexpr float kMin = 1e-3f;

__device__ __forceinline__ float to_f(float v) { return v; }
__device__ __forceinline__ float to_f(__bf16 v) { return (float)v; }
template <typename T> __device__ __forceinline__ T from_f(float v);
template <> __device__ __forceinline__ float from_f<float>(float v) { return v; }
template <> __device__ __forceinline__ __bf16 from_f<__bf16>(float v) { return (__bf16)v; }

// Transcendental VALU instructions with an explicit wait behind them.  gfx950 issues v_exp / v_log / v_rcp / v_sqrt at a quarter
// of the VALU rate (four 16-lane passes) and does NOT interlock their result against the next VALU instruction (tools/hazard/
// trans_hazard.hip: a dependent instruction directly behind v_rcp_f32 reads the stale register in every lane); hipcc pads a
// dependent instruction with `s_nop 0`.  In this kernel that was not always enough: identical launches disagreed in lanes 48..63
// — the last pass of a transcendental — of single waves, only while kernels of another stream shared the SIMDs (DESIGN.md §6b,
// tools/dbg_spline_tap.py).  Every transcendental of this file therefore goes through these wrappers, which the compiler can
// neither reorder nor pair up (the vectoriser had turned two divisions into back-to-back v_rcp_f32 feeding one v_pk_fma_f32),
// and which keep VITS_TRANS_NOPS + 1 wait states between the instruction and any consumer.
#ifndef VITS_TRANS_NOPS
#define VITS_TRANS_NOPS 4
#endif
#define VITS_STR2(x) #x
#define VITS_STR(x) VITS_STR2(x)
#define VITS_TRANS(op, dst, src) asm volatile(op " %0, %1\n\ts_nop " VITS_STR(VITS_TRANS_NOPS) : "=v"(dst) : "v"(src))
__device__ __forceinline__ float t_rcp(float x) { float r; VITS_TRANS("v_rcp_f32", r, x); return r; }
__device__ __forceinline__ float t_exp2(float x) { float r; VITS_TRANS("v_exp_f32", r, x); return r; }
__device__ __forceinline__ float t_log2(float x) { float r; VITS_TRANS("v_log_f32", r, x); return r; }
__device__ __forceinline__ float t_sqrt(float x) { float r; VITS_TRANS("v_sqrt_f32", r, x); return r; }
// e^x, ln x on the hardware's base-2 instructions (1 ulp each; arguments here are far from the denormal range: softmax exponents
// are <= 0 and the smallest term is flushed harmlessly, logs take dnum / den > 0 of order 1e-3 .. 1e3)
__device__ __forceinline__ float t_exp(float x) { return t_exp2(x * 1.44269504088896341f); }
__device__ __forceinline__ float t_log(float x) { return t_log2(x) * 0.693147180559945309f; }

// a / b as reciprocal estimate + one Newton step + one residual correction (error < 1 ulp for operands in the normal range, which
// is all this kernel divides) instead of the v_div_scale / v_div_fmas / v_div_fixup sequence: 24 divisions per element, a third of
// the kernel's instructions.
__device__ __forceinline__ float fdiv(float a, float b) {
  float r = t_rcp(b);
  r = fmaf(fmaf(-b, r, 1.0f), r, r);
  const float q = a * r;
  return fmaf(fmaf(-b, q, a), r, q);
}

// log(1 + e^v): v > 20 returns v (as torch's softplus threshold); e^v < 2^-24 adds nothing to 1 in fp32, the result is then e^v
__device__ __forceinline__ float softplus_f(float v) {
  if (v > 20.f) return v;
  const float e = t_exp(v);
  return e < 5.9604645e-8f ? e : t_log(1.f + e);
}
__device__ __forceinline__ float sigmoid_f(float v) { return fdiv(1.f, 1.f + t_exp(-v)); }

struct Knots {
  float p[NB];        // softmax probabilities
  float c[NB + 1];    // knot positions
};

__device__ __forceinline__ void make_knots(const float* u, float B, Knots& k) {
  float m = u[0];
#pragma unroll
  for (int j = 1; j < NB; ++j) m = fmaxf(m, u[j]);
  float s = 0.f;
#pragma unroll
  for (int j = 0; j < NB; ++j) { k.p[j] = t_exp(u[j] - m); s += k.p[j]; }
  const float inv = fdiv(1.f, s);
  float cum = 0.f;
  k.c[0] = -B;
#pragma unroll
  for (int j = 0; j < NB; ++j) {
    k.p[j] *= inv;
    cum += kMin + (1.f - kMin * NB) * k.p[j];
    k.c[j + 1] = 2.f * B * cum - B;
  }
  k.c[NB] = B;
}

// adjoint of the knots (only knots i and i+1 carry one) -> adjoint of the unnormalised logits
__device__ __forceinline__ void knots_bwd(const Knots& k, int i, float g_lo, float g_hi, float B, float* g_u) {
  // knot j (1 <= j <= NB-1) = 2B * sum_{m<j} size[m] - B; knots 0 and NB are constants
  const float glo = (i >= 1) ? g_lo : 0.f;
  const float ghi = (i + 1 <= NB - 1) ? g_hi : 0.f;
  float gp[NB], dot = 0.f;
  float fi = (float)i;
  asm volatile("" : "+v"(fi));          // opaque to the optimiser: it would prove fi integral and rebuild the compare + select
#pragma unroll
  for (int m = 0; m < NB; ++m) {
#ifdef VITS_SPLINE_MASK_SELECTS
    float gs = 0.f;
    if (m < i) gs += glo;
    if (m < i + 1) gs += ghi;
#else
    // [m < i] and [m <= i] as 0 / 1 factors clamp(i - m, 0, 1), clamp(i + 1 - m, 0, 1) on v_med3_f32: no per-lane select through
    // an SGPR mask (the compiler turns every integer formulation of this back into v_cmp + v_cndmask)
    const float lt = __builtin_amdgcn_fmed3f(fi - (float)m, 0.f, 1.f), le = __builtin_amdgcn_fmed3f(fi - (float)(m - 1), 0.f, 1.f);
    const float gs = glo * lt + ghi * le;
#endif
    gp[m] = 2.f * B * gs * (1.f - kMin * NB);
    dot += gp[m] * k.p[m];
  }
#pragma unroll
  for (int m = 0; m < NB; ++m) g_u[m] = k.p[m] * (gp[m] - dot);
}

// FLOW = the element lives in a two-channel flow state (modules.ConvFlow, modules.py:346-390): x / y / gy / gx are [n][2] tensors,
// channel c1 is transformed, channel 1 - c1 passes through, everything is multiplied by the row mask m[n], and the
// log-determinant leaves as lad[n] * m[n] (its per-item sum is a separate launch).  The gradient of that sum arrives per ITEM
// (dlogdet[e / t]).  This replaces the slice / cat / mask / cast glue around the spline (and, with c1 alternating from layer to
// layer, the physical channel flips between the layers).
struct FlowIO { const float* m; const float* dlogdet; int t; int c1; };

template <typename TH, bool BWD, bool FLOW>
__global__ void spline_kernel(const float* __restrict__ x, const TH* __restrict__ h, int ldh, float hscale, int inverse, float B,
                              float* __restrict__ y, float* __restrict__ lad, const float* __restrict__ gy,
                              const float* __restrict__ gl, float* __restrict__ gx, TH* __restrict__ gh, int n, FlowIO io) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= n) return;
  const int ix = FLOW ? 2 * e + io.c1 : e, ip = FLOW ? 2 * e + 1 - io.c1 : e;      // transformed / pass-through element
  const float mv = FLOW ? io.m[e] : 1.f;
  const float xv = x[ix];
  const TH* hr = h + (size_t)e * ldh;
  const bool inside = (xv >= -B) && (xv <= B);
  if (!inside) {
    if (!BWD) {
      y[ix] = xv * mv; lad[e] = 0.f;
      if (FLOW) y[ip] = x[ip] * mv;
    } else {
      gx[ix] = gy[ix] * mv;
      if (FLOW) gx[ip] = gy[ip] * mv;
      for (int j = 0; j < ldh; ++j) gh[(size_t)e * ldh + j] = from_f<TH>(0.f);
    }
    return;
  }
  float uw[NB], uh[NB], ud[NB - 1];
#pragma unroll
  for (int j = 0; j < NB; ++j) { uw[j] = to_f(hr[j]) * hscale; uh[j] = to_f(hr[NB + j]) * hscale; }
#pragma unroll
  for (int j = 0; j < NB - 1; ++j) ud[j] = to_f(hr[2 * NB + j]);
  Knots kw, kh;
  make_knots(uw, B, kw);
  make_knots(uh, B, kh);
  // the pinned end derivatives: min_derivative + softplus(log(exp(1 - min_derivative) - 1)) = 1 (transforms.py:79-82), folded at
  // compile time with libm (no instruction is emitted for it)
  const float d_edge = kMin + log1pf(expf(logf(expf(1.f - kMin) - 1.f)));
  // bin search on the widths' knots (forward) or the heights' knots (inverse); last edge + 1e-6
  const float* loc = inverse ? kh.c : kw.c;
  int i = -1;
#pragma unroll
  for (int j = 0; j <= NB; ++j) i += (xv >= (j == NB ? loc[j] + 1e-6f : loc[j])) ? 1 : 0;
  i = i < 0 ? 0 : (i > NB - 1 ? NB - 1 : i);
  float CW = 0.f, CW1 = 0.f, CH = 0.f, CH1 = 0.f, ud0 = 0.f, ud1 = 0.f;
#pragma unroll
  for (int j = 0; j < NB; ++j)
    if (j == i) { CW = kw.c[j]; CW1 = kw.c[j + 1]; CH = kh.c[j]; CH1 = kh.c[j + 1]; }
#pragma unroll
  for (int j = 0; j < NB - 1; ++j) { if (j == i - 1) ud0 = ud[j]; if (j == i) ud1 = ud[j]; }
  const float W = CW1 - CW, Hh = CH1 - CH;
  const float d0 = (i == 0) ? d_edge : kMin + softplus_f(ud0);
  const float d1 = (i == NB - 1) ? d_edge : kMin + softplus_f(ud1);
  const float delta = fdiv(Hh, W);
  const float s = d0 + d1 - 2.f * delta;

  float theta, out;
  float a = 0.f, b = 0.f, c = 0.f, sd = 0.f, D = 0.f, u = 0.f;
  if (!inverse) {
    theta = fdiv(xv - CW, W);
  } else {
    u = xv - CH;
    a = u * s + Hh * (delta - d0);
    b = Hh * d0 - u * s;
    c = -delta * u;
    sd = t_sqrt(b * b - 4.f * a * c);
    D = -b - sd;
    theta = fdiv(2.f * c, D);
  }
  const float q = theta * (1.f - theta);
  const float den = delta + s * q;
  const float E = d1 * theta * theta + 2.f * delta * q + d0 * (1.f - theta) * (1.f - theta);
  const float dnum = delta * delta * E;
  const float num = Hh * (delta * theta * theta + d0 * q);
  const float l = t_log(dnum) - 2.f * t_log(den);
  if (!inverse) out = CH + fdiv(num, den); else out = theta * W + CW;
  if (!BWD) {
    y[ix] = out * mv;
    lad[e] = (inverse ? -l : l) * mv;
    if (FLOW) y[ip] = x[ip] * mv;
    return;
  }
  // ------------------------------------------------------------------ reverse mode
  const float gyv = gy[ix] * mv, glv = FLOW ? io.dlogdet[e / io.t] * mv : gl[e];
  float g_theta = 0.f, g_q = 0.f, g_delta = 0.f, g_s = 0.f, g_d0 = 0.f, g_d1 = 0.f, g_H = 0.f, g_W = 0.f, g_CW = 0.f, g_CH = 0.f;
  float g_x = 0.f;
  float g_dnum, g_den;
  if (!inverse) {
    const float g_num = fdiv(gyv, den);
    g_den = -fdiv(gyv * num, den * den) - fdiv(2.f * glv, den);
    g_dnum = fdiv(glv, dnum);
    g_CH += gyv;
    g_H += g_num * (delta * theta * theta + d0 * q);
    g_delta += g_num * Hh * theta * theta;
    g_theta += g_num * Hh * 2.f * delta * theta;
    g_d0 += g_num * Hh * q;
    g_q += g_num * Hh * d0;
  } else {
    g_theta += gyv * W;
    g_W += gyv * theta;
    g_CW += gyv;
    g_dnum = -fdiv(glv, dnum);
    g_den = fdiv(2.f * glv, den);
  }
  // den = delta + s q
  g_delta += g_den; g_s += g_den * q; g_q += g_den * s;
  // dnum = delta^2 E
  g_delta += g_dnum * 2.f * delta * E;
  const float g_E = g_dnum * delta * delta;
  g_d1 += g_E * theta * theta;
  g_theta += g_E * (2.f * d1 * theta - 2.f * d0 * (1.f - theta));
  g_q += g_E * 2.f * delta;
  g_delta += g_E * 2.f * q;
  g_d0 += g_E * (1.f - theta) * (1.f - theta);
  // q = theta (1 - theta)
  g_theta += g_q * (1.f - 2.f * theta);
  if (!inverse) {
    g_x = fdiv(g_theta, W);
    g_CW -= fdiv(g_theta, W);
    g_W -= fdiv(g_theta * theta, W);
  } else {
    // theta = 2c / D, D = -b - sqrt(b^2 - 4ac)
    float g_c = fdiv(g_theta * 2.f, D);
    const float g_D = -fdiv(g_theta * theta, D);
    float g_b = -g_D;
    const float g_disc = -fdiv(g_D, 2.f * sd);
    g_b += g_disc * 2.f * b;
    const float g_a = -4.f * c * g_disc;
    g_c += -4.f * a * g_disc;
    float g_u = 0.f;
    g_u += g_a * s; g_s += g_a * u; g_H += g_a * (delta - d0); g_delta += g_a * Hh; g_d0 -= g_a * Hh;
    g_H += g_b * d0; g_d0 += g_b * Hh; g_u -= g_b * s; g_s -= g_b * u;
    g_delta -= g_c * u; g_u -= g_c * delta;
    g_x = g_u;
    g_CH -= g_u;
  }
  // s = d0 + d1 - 2 delta ; delta = H / W
  g_d0 += g_s; g_d1 += g_s; g_delta -= 2.f * g_s;
  g_H += fdiv(g_delta, W);
  g_W -= fdiv(g_delta * Hh, W * W);
  // W = CW1 - CW, H = CH1 - CH
  float g_uw[NB], g_uh[NB];
  knots_bwd(kw, i, g_CW - g_W, g_W, B, g_uw);
  knots_bwd(kh, i, g_CH - g_H, g_H, B, g_uh);
  gx[ix] = g_x;
  if (FLOW) gx[ip] = gy[ip] * mv;
#ifdef VITS_SPLINE_VOLATILE_STORE
  volatile TH* gr = gh + (size_t)e * ldh;
#else
  TH* gr = gh + (size_t)e * ldh;
#endif
#ifdef VITS_SPLINE_STORE_NOPS
  asm volatile("s_nop 7\n\ts_nop 7" ::: "memory");
#endif
#pragma unroll
  for (int j = 0; j < NB; ++j) { gr[j] = from_f<TH>(g_uw[j] * hscale); gr[NB + j] = from_f<TH>(g_uh[j] * hscale); }
#ifdef VITS_SPLINE_STORE_NOPS
  asm volatile("s_nop 7\n\ts_nop 7" ::: "memory");
#endif
#pragma unroll
  for (int j = 0; j < NB - 1; ++j) {
    float g = 0.f;
    if (j == i - 1) g = g_d0 * sigmoid_f(ud[j]);     // d[i]   = 1e-3 + softplus(ud[i-1])
    if (j == i) g = g_d1 * sigmoid_f(ud[j]);         // d[i+1] = 1e-3 + softplus(ud[i])
    gr[2 * NB + j] = from_f<TH>(g);
  }
  for (int j = 3 * NB - 1; j < ldh; ++j) gr[j] = from_f<TH>(0.f);
}

}  // namespace

extern "C" int vits_rq_spline(int h_dtype, const float* x, const void* h, int ldh, float hscale, int inverse, float tail_bound,
                              float* y, float* logabsdet, int n, void* stream) {
  if (!x || !h || !y || !logabsdet || n <= 0 || ldh < 29 || !(tail_bound > 0.f)) return VITS_E_BADARG;
  hipStream_t s = static_cast<hipStream_t>(stream);
  dim3 grid((n + 127) / 128), block(128);
  if (h_dtype == VITS_DT_F32)
    hipLaunchKernelGGL((spline_kernel<float, false, false>), grid, block, 0, s, x, (const float*)h, ldh, hscale, inverse, tail_bound, y, logabsdet, nullptr, nullptr, nullptr, (float*)nullptr, n, FlowIO{});
  else if (h_dtype == VITS_DT_BF16)
    hipLaunchKernelGGL((spline_kernel<__bf16, false, false>), grid, block, 0, s, x, (const __bf16*)h, ldh, hscale, inverse, tail_bound, y, logabsdet, nullptr, nullptr, nullptr, (__bf16*)nullptr, n, FlowIO{});
  else return VITS_E_UNSUPPORTED;
  return vits::check_launch("vits_rq_spline");
}

extern "C" int vits_rq_spline_bwd(int h_dtype, const float* x, const void* h, int ldh, float hscale, int inverse, float tail_bound,
                                  const float* gy, const float* glogabsdet, float* gx, void* gh, int n, void* stream) {
  if (!x || !h || !gy || !glogabsdet || !gx || !gh || n <= 0 || ldh < 29 || !(tail_bound > 0.f)) return VITS_E_BADARG;
  hipStream_t s = static_cast<hipStream_t>(stream);
  dim3 grid((n + 127) / 128), block(128);
  if (h_dtype == VITS_DT_F32)
    hipLaunchKernelGGL((spline_kernel<float, true, false>), grid, block, 0, s, x, (const float*)h, ldh, hscale, inverse, tail_bound, nullptr, nullptr, gy, glogabsdet, gx, (float*)gh, n, FlowIO{});
  else if (h_dtype == VITS_DT_BF16)
    hipLaunchKernelGGL((spline_kernel<__bf16, true, false>), grid, block, 0, s, x, (const __bf16*)h, ldh, hscale, inverse, tail_bound, nullptr, nullptr, gy, glogabsdet, gx, (__bf16*)gh, n, FlowIO{});
  else return VITS_E_UNSUPPORTED;
  return vits::check_launch("vits_rq_spline_bwd");
}

// ---- the spline inside a two-channel flow state (see FlowIO above) -------------------------------------------------------
extern "C" int vits_flow_spline(int h_dtype, const float* x2, const void* h, int ldh, float hscale, int inverse, float tail_bound,
                                const float* mask, int c1, float* y2, float* lad_masked, int n, void* stream) {
  if (!x2 || !h || !mask || !y2 || !lad_masked || n <= 0 || ldh < 29 || !(tail_bound > 0.f) || (c1 != 0 && c1 != 1)) return VITS_E_BADARG;
  hipStream_t s = static_cast<hipStream_t>(stream);
  dim3 grid((n + 127) / 128), block(128);
  const FlowIO io{mask, nullptr, 1, c1};
  if (h_dtype == VITS_DT_F32)
    hipLaunchKernelGGL((spline_kernel<float, false, true>), grid, block, 0, s, x2, (const float*)h, ldh, hscale, inverse, tail_bound, y2, lad_masked, nullptr, nullptr, nullptr, (float*)nullptr, n, io);
  else if (h_dtype == VITS_DT_BF16)
    hipLaunchKernelGGL((spline_kernel<__bf16, false, true>), grid, block, 0, s, x2, (const __bf16*)h, ldh, hscale, inverse, tail_bound, y2, lad_masked, nullptr, nullptr, nullptr, (__bf16*)nullptr, n, io);
  else return VITS_E_UNSUPPORTED;
  return vits::check_launch("vits_flow_spline");
}

extern "C" int vits_flow_spline_bwd(int h_dtype, const float* x2, const void* h, int ldh, float hscale, int inverse, float tail_bound,
                                    const float* mask, int c1, const float* dy2, const float* dlogdet, int t, float* dx2, void* gh, int n,
                                    void* stream) {
  if (!x2 || !h || !mask || !dy2 || !dlogdet || !dx2 || !gh || n <= 0 || t <= 0 || n % t != 0 || ldh < 29 || !(tail_bound > 0.f) ||
      (c1 != 0 && c1 != 1))
    return VITS_E_BADARG;
  hipStream_t s = static_cast<hipStream_t>(stream);
  dim3 grid((n + 127) / 128), block(128);
  const FlowIO io{mask, dlogdet, t, c1};
  if (h_dtype == VITS_DT_F32)
    hipLaunchKernelGGL((spline_kernel<float, true, true>), grid, block, 0, s, x2, (const float*)h, ldh, hscale, inverse, tail_bound, nullptr, nullptr, dy2, nullptr, dx2, (float*)gh, n, io);
  else if (h_dtype == VITS_DT_BF16)
    hipLaunchKernelGGL((spline_kernel<__bf16, true, true>), grid, block, 0, s, x2, (const __bf16*)h, ldh, hscale, inverse, tail_bound, nullptr, nullptr, dy2, nullptr, dx2, (__bf16*)gh, n, io);
  else return VITS_E_UNSUPPORTED;
  return vits::check_launch("vits_flow_spline_bwd");
}
