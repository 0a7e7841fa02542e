// Grouped strided convolutions of DiscriminatorS (reference models.py:343-349: Conv1d(16,64,41,4,groups=4) ...
// Conv1d(1024,1024,41,4,groups=256): 4 input channels per group, 16 or 4 output channels per group) as direct kernels.
//
// On the tiled matrix-core kernels these layers run as dense block-diagonal tiles: 1/16 of the multiplies are real and every stage
// gathers 41 taps x 64 rows (105 us per layer at batch 32).  A group's reduction is only 4 x 41 = 164 deep, so here
//   * workgroup = 4 waves = 4 adjacent groups x 64 consecutive output times (forward) / 64 values of t / stride (data gradient) of
//     one item; wave = one group; the tile's input (dY) window of the 4 groups is staged once in LDS;
//   * 16 output channels per group, even stride (layers 2-4): v_mfma_f32_16x16x32_bf16 with the reduction index (tap, channel) —
//     the group's weights are gathered ONCE per wave from the arena's dense operand [k][c_out][c_in] into 24 registers (A), the
//     window supplies 16-byte B fragments, 6 MFMAs per 16 times: 16 us per layer.  The data gradient is the same product with the
//     4 phases x 4 input channels of one t / 4 as rows and (dY row offset, output channel) as the reduction: 19 us;
//   * other group shapes (4 output channels per group: layer 5; odd strides): lane = one time, the group's weights wave-uniform
//     through the scalar cache into SGPRs (v_fma with scalar operands), the window de-interleaved by (group, row mod stride) so
//     that the per-tap read is lane-consecutive.  This form is bound by the scalar cache (one 8-byte piece per line: its misses
//     are serialised — 120 us on the 16-channel layers before the matrix-core form), fine for the small layer 5;
//   * fp32 accumulation, bias + leaky-relu epilogue (forward), (acc + res) * lrelu'(mg_src) (data gradient) as in vits_conv1d_cl.
// The weight gradient (16 output channels per group) is a matrix-core product too, see grouped_wgrad_mfma_kernel: 24 us per layer;
// a VALU form (thread = (tap, channel), dY rows broadcast from LDS) was built first and measured no faster than the tiled kernel
// (152 vs 112 us).
#include "common.h"

namespace {

typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

constexpr int TT = 64;            // output (fwd) / input-phase (dgrad) times per tile = lanes
constexpr int IG = 4;             // input channels per group
constexpr int KMAXG = 64, SMAXG = 4;
constexpr int XROWS = (TT - 1) * SMAXG + KMAXG;          // rows of the forward input window
constexpr int XPLANE = XROWS / 1 + 8;                    // entries per (group) plane, all phases
constexpr int DYROWS = TT + KMAXG;                       // rows of the data gradient's dY window (>= TT + taps per phase)

__device__ __forceinline__ float bf_lo(unsigned int u) { return __uint_as_float(u << 16); }
__device__ __forceinline__ float bf_hi(unsigned int u) { return __uint_as_float(u & 0xffff0000u); }

struct GArgs {
  const __bf16* x; const __bf16* w; const float* bias; __bf16* y;
  const __bf16* res; const __bf16* mg;
  int T_in, T_out, C_in, C_out, k, stride, pad, groups;
  float slope;
};

// ---- forward: y[n][t][g*OG + o] = lrelu(bias + sum_{tap, c} x[n][t*stride + tap - pad][g*4 + c] * w[tap][g*OG + o][g*4 + c])
template <int OG>
__global__ __launch_bounds__(256) void grouped_fwd_kernel(GArgs a) {
  __shared__ u32x2 xs[4][XPLANE];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);      // (uniform: the weights' addresses must be)
  const int t0 = blockIdx.x * TT, g4 = blockIdx.y * 4, n = blockIdx.z;
  const int g = g4 + wave;
  const int s = a.stride;
  const int nrows = (TT - 1) * s + a.k;
  const int per_phase = (nrows + s - 1) / s;               // entries per phase inside a plane
  // stage: row r of the window = input time t0*s - pad + r; 16 B = the 4 channels of two adjacent groups
  const __bf16* X = a.x + (size_t)n * a.T_in * a.C_in + (size_t)g4 * IG;
  for (int idx = tid; idx < nrows * 2; idx += 256) {
    const int r = idx >> 1, half = idx & 1;
    const int tin = t0 * s - a.pad + r;
    u32x4 v = {0u, 0u, 0u, 0u};
    if (tin >= 0 && tin < a.T_in) v = *reinterpret_cast<const u32x4*>(X + (size_t)tin * a.C_in + half * 8);
    const int e = (r % s) * per_phase + r / s;
    xs[half * 2][e] = u32x2{v.x, v.y};
    xs[half * 2 + 1][e] = u32x2{v.z, v.w};
  }
  __syncthreads();
  float acc[OG];
#pragma unroll
  for (int o = 0; o < OG; ++o) acc[o] = a.bias ? a.bias[g * OG + o] : 0.f;
  const __bf16* W = a.w + (size_t)(g * OG) * a.C_in + (size_t)g * IG;      // (tap 0, first channel of the group): wave-uniform
  const size_t wtap = (size_t)a.C_out * a.C_in;
  for (int tap = 0; tap < a.k; ++tap) {
    const int r = lane * s + tap;
    const u32x2 xv = xs[wave][(r % s) * per_phase + r / s];
    const float x0 = bf_lo(xv.x), x1 = bf_hi(xv.x), x2 = bf_lo(xv.y), x3 = bf_hi(xv.y);
    const __bf16* wt = W + tap * wtap;
#pragma unroll
    for (int o = 0; o < OG; ++o) {
      const u32x2 wq = *reinterpret_cast<const u32x2*>(wt + (size_t)o * a.C_in);       // scalar load: uniform address
      acc[o] = fmaf(x3, bf_hi(wq.y), fmaf(x2, bf_lo(wq.y), fmaf(x1, bf_hi(wq.x), fmaf(x0, bf_lo(wq.x), acc[o]))));
    }
  }
  const int t = t0 + lane;
  if (t >= a.T_out) return;
  __bf16* Y = a.y + ((size_t)n * a.T_out + t) * a.C_out + (size_t)g * OG;
  union { __bf16 e[OG]; u32x2 q[OG / 4]; } out;
#pragma unroll
  for (int o = 0; o < OG; ++o) { const float v = acc[o]; out.e[o] = (__bf16)(v > 0.f ? v : v * a.slope); }
#pragma unroll
  for (int i = 0; i < OG / 4; ++i) reinterpret_cast<u32x2*>(Y)[i] = out.q[i];
}

// ---- forward on the matrix cores (16 output channels per group, even stride): per group the layer is a [16 o] x [k*4] x [t]
// product, which v_mfma_f32_16x16x32_bf16 takes directly with the reduction index kk = tap*4 + c:
//   A (weights)  lane (o = l%16, q = l/16), step ks: kk = 32 ks + 8 q .. +7 = taps 8 ks + 2 q, +1 x 4 channels: two 8-byte pieces of the
//                dense operand, loaded ONCE per wave into 24 registers (taps >= k are zero);
//   B (input)    lane (t = l%16, q): the same kk of time t = rows t*stride + tap, tap + 1 of the staged window — adjacent rows of
//                the [row][4] plane, one 16-byte LDS read;
//   D            lane holds channels 4 q .. 4 q + 3 of time l%16: an 8-byte store per lane after bias + leaky-relu.
// 6 MFMAs per 16 times instead of 41 x 64 scalar-operand FMAs per time.
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef float f32x4_t __attribute__((ext_vector_type(4)));
constexpr int KSTEPS = 6;                                  // 6 x 32 = 192 >= 4 * 41 (taps 0..47)
constexpr int MROWS = (TT - 1) * SMAXG + 8 * KSTEPS;       // rows of the window incl. the zero-weight taps

__global__ __launch_bounds__(256) void grouped_fwd_mfma_kernel(GArgs a) {
  __shared__ __attribute__((aligned(16))) u32x2 xs[4][MROWS + 4];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int t0 = blockIdx.x * TT, g4 = blockIdx.y * 4, n = blockIdx.z;
  const int g = g4 + wave;
  const int s = a.stride;
  const int l16 = lane & 15, q = lane >> 4;
  // weights of the group: A fragments
  union Frag { bf16x8_t v; u32x2 h[2]; };
  Frag wa[KSTEPS];
  {
    const __bf16* W = a.w + (size_t)(g * 16 + l16) * a.C_in + (size_t)g * IG;
    const size_t wtap = (size_t)a.C_out * a.C_in;
#pragma unroll
    for (int ks = 0; ks < KSTEPS; ++ks) {
      const int tap = 8 * ks + 2 * q;
      wa[ks].h[0] = tap < a.k ? *reinterpret_cast<const u32x2*>(W + (size_t)tap * wtap) : u32x2{0u, 0u};
      wa[ks].h[1] = tap + 1 < a.k ? *reinterpret_cast<const u32x2*>(W + (size_t)(tap + 1) * wtap) : u32x2{0u, 0u};
    }
  }
  // input window: row r = input time t0*s - pad + r, rows past the layer's last tap are real (finite) data or zeros
  const int nrows = (TT - 1) * s + 8 * KSTEPS;
  const __bf16* X = a.x + (size_t)n * a.T_in * a.C_in + (size_t)g4 * IG;
  for (int idx = tid; idx < nrows * 2; idx += 256) {
    const int r = idx >> 1, half = idx & 1;
    const int tin = t0 * s - a.pad + r;
    u32x4 v = {0u, 0u, 0u, 0u};
    if (tin >= 0 && tin < a.T_in) v = *reinterpret_cast<const u32x4*>(X + (size_t)tin * a.C_in + half * 8);
    xs[half * 2][r] = u32x2{v.x, v.y};
    xs[half * 2 + 1][r] = u32x2{v.z, v.w};
  }
  __syncthreads();
  float bias4[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) bias4[i] = a.bias ? a.bias[g * 16 + 4 * q + i] : 0.f;
#pragma unroll
  for (int sub = 0; sub < TT / 16; ++sub) {
    f32x4_t acc = {0.f, 0.f, 0.f, 0.f};
    const u32x2* xrow = &xs[wave][(sub * 16 + l16) * s + 2 * q];
#pragma unroll
    for (int ks = 0; ks < KSTEPS; ++ks) {
      union { bf16x8_t v; u32x4 u; } xb;
      xb.u = *reinterpret_cast<const u32x4*>(xrow + 8 * ks);              // rows tap, tap + 1 (16-byte aligned: even stride, even tap)
      acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa[ks].v, xb.v, acc, 0, 0, 0);
    }
    const int t = t0 + sub * 16 + l16;
    if (t < a.T_out) {
      union { __bf16 e[4]; u32x2 qv; } out;
#pragma unroll
      for (int i = 0; i < 4; ++i) { const float v = acc[i] + bias4[i]; out.e[i] = (__bf16)(v > 0.f ? v : v * a.slope); }
      *reinterpret_cast<u32x2*>(a.y + ((size_t)n * a.T_out + t) * a.C_out + (size_t)g * 16 + 4 * q) = out.qv;
    }
  }
}

// ---- data gradient: dx[n][ti][g*4 + c] = (sum_{tap, o} dy[n][(ti + pad - tap) / stride][g*OG + o] * w[tap][g*OG + o][g*4 + c] + res) * lrelu'(mg)
// over the taps with (ti + pad - tap) divisible by stride.  Tile = one phase (ti mod stride) x 64 values of q = ti / stride.
template <int OG>
__global__ __launch_bounds__(256) void grouped_dgrad_kernel(GArgs a) {
  constexpr int EPR = OG / 4;                               // 8-byte entries per dY row of one group
  __shared__ u32x2 ds[4][DYROWS * EPR];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);      // (uniform: the weights' addresses must be)
  const int s = a.stride;
  const int q_tiles = gridDim.x / s;
  const int phase = blockIdx.x / q_tiles, q0 = (blockIdx.x - phase * q_tiles) * TT;
  const int g4 = blockIdx.y * 4, n = blockIdx.z;
  const int g = g4 + wave;
  // taps of this phase: tap = tap0 + s*j, output time to = q + d0 - j
  const int tap0 = (phase + a.pad) % s;
  const int J = tap0 < a.k ? (a.k - tap0 + s - 1) / s : 0;
  const int d0 = (phase + a.pad - tap0) / s;
  const int row_lo = q0 + d0 - (J - 1);                     // first dY row the tile reads
  const int nrows = TT + J - 1;
  const __bf16* DY = a.x + (size_t)n * a.T_out * a.C_out + (size_t)g4 * OG;
  constexpr int VPR = 4 * OG / 8;                           // 16-byte vectors per row of the 4 groups
  for (int idx = tid; idx < nrows * VPR; idx += 256) {
    const int r = idx / VPR, vc = idx - r * VPR;
    const int to = row_lo + r;
    u32x4 v = {0u, 0u, 0u, 0u};
    if (to >= 0 && to < a.T_out) v = *reinterpret_cast<const u32x4*>(DY + (size_t)to * a.C_out + vc * 8);
    const int grp = (vc * 8) / OG, e0 = ((vc * 8) % OG) / 4;
    if constexpr (OG >= 8) {
      ds[grp][r * EPR + e0] = u32x2{v.x, v.y};
      ds[grp][r * EPR + e0 + 1] = u32x2{v.z, v.w};
    } else {                                                // OG = 4: a 16-byte vector holds two groups
      ds[grp][r] = u32x2{v.x, v.y};
      ds[grp + 1][r] = u32x2{v.z, v.w};
    }
  }
  __syncthreads();
  float acc[IG] = {0.f, 0.f, 0.f, 0.f};
  const __bf16* W = a.w + (size_t)(g * OG) * a.C_in + (size_t)g * IG;
  const size_t wtap = (size_t)a.C_out * a.C_in;
  for (int j = 0; j < J; ++j) {
    const int r = lane + (J - 1) - j;                       // row (q + d0 - j) - row_lo
    const __bf16* wt = W + (size_t)(tap0 + s * j) * wtap;
#pragma unroll
    for (int e = 0; e < EPR; ++e) {
      const u32x2 dv = ds[wave][r * EPR + e];
      const float d[4] = {bf_lo(dv.x), bf_hi(dv.x), bf_lo(dv.y), bf_hi(dv.y)};
#pragma unroll
      for (int oo = 0; oo < 4; ++oo) {
        const u32x2 wq = *reinterpret_cast<const u32x2*>(wt + (size_t)(e * 4 + oo) * a.C_in);   // scalar load
        acc[0] = fmaf(d[oo], bf_lo(wq.x), acc[0]);
        acc[1] = fmaf(d[oo], bf_hi(wq.x), acc[1]);
        acc[2] = fmaf(d[oo], bf_lo(wq.y), acc[2]);
        acc[3] = fmaf(d[oo], bf_hi(wq.y), acc[3]);
      }
    }
  }
  const int ti = (q0 + lane) * s + phase;
  if (ti >= a.T_in) return;
  const size_t o = ((size_t)n * a.T_in + ti) * a.C_in + (size_t)g * IG;
  if (a.res) {
    const u32x2 rv = *reinterpret_cast<const u32x2*>(a.res + o);
    acc[0] += bf_lo(rv.x); acc[1] += bf_hi(rv.x); acc[2] += bf_lo(rv.y); acc[3] += bf_hi(rv.y);
  }
  if (a.mg) {
    const u32x2 mv = *reinterpret_cast<const u32x2*>(a.mg + o);
    acc[0] *= bf_lo(mv.x) > 0.f ? 1.f : a.slope; acc[1] *= bf_hi(mv.x) > 0.f ? 1.f : a.slope;
    acc[2] *= bf_lo(mv.y) > 0.f ? 1.f : a.slope; acc[3] *= bf_hi(mv.y) > 0.f ? 1.f : a.slope;
  }
  union { __bf16 e[4]; u32x2 q; } out;
#pragma unroll
  for (int c = 0; c < 4; ++c) out.e[c] = (__bf16)acc[c];
  *reinterpret_cast<u32x2*>(a.y + o) = out.q;
}


// ---- data gradient on the matrix cores (16 output channels per group, stride 4): the 4 phases x 4 input channels of one q = t / 4
// are the 16 rows of the product, the reduction index is kk = (dY row offset r, o): dx[4 q + p][c] = sum_{r, o} dy[q + r][o] *
// w[p + pad - 4 r][o][c] (taps outside [0, k) are zero).  A = those weights (gathered once per wave), B = 16-byte pieces of the staged
// dY window, D: lane (q = l%16, p = l/16) holds the 4 channels of input time 4 q + p.
constexpr int DROWS = TT + 2 * KSTEPS;                     // dY rows of the window: 64 q + 12 row offsets
__global__ __launch_bounds__(256) void grouped_dgrad_mfma_kernel(GArgs a, int r_min) {
  __shared__ __attribute__((aligned(16))) u32x4 ds[4][DROWS][2];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int q0 = blockIdx.x * TT, g4 = blockIdx.y * 4, n = blockIdx.z;
  const int g = g4 + wave;
  const int l16 = lane & 15, kq = lane >> 4;
  union Frag { bf16x8_t v; unsigned short e[8]; };
  Frag wa[KSTEPS];
  {
    const int p = l16 >> 2, c = l16 & 3;
    const unsigned short* W = reinterpret_cast<const unsigned short*>(a.w) + (size_t)(g * 16 + (kq & 1) * 8) * a.C_in + (size_t)g * IG + c;
    const size_t wtap = (size_t)a.C_out * a.C_in;
#pragma unroll
    for (int ks = 0; ks < KSTEPS; ++ks) {
      const int r = r_min + 2 * ks + (kq >> 1);
      const int tap = p + a.pad - 4 * r;
      const bool ok = tap >= 0 && tap < a.k;
#pragma unroll
      for (int i = 0; i < 8; ++i) wa[ks].e[i] = ok ? W[(size_t)tap * wtap + (size_t)i * a.C_in] : (unsigned short)0;
    }
  }
  const __bf16* DY = a.x + (size_t)n * a.T_out * a.C_out + (size_t)g4 * 16;
  for (int idx = tid; idx < DROWS * 8; idx += 256) {
    const int r = idx >> 3, vc = idx & 7;
    const int to = q0 + r_min + r;
    u32x4 v = {0u, 0u, 0u, 0u};
    if (to >= 0 && to < a.T_out) v = *reinterpret_cast<const u32x4*>(DY + (size_t)to * a.C_out + vc * 8);
    ds[vc >> 1][r][vc & 1] = v;
  }
  __syncthreads();
#pragma unroll
  for (int sub = 0; sub < TT / 16; ++sub) {
    f32x4_t acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < KSTEPS; ++ks) {
      union { bf16x8_t v; u32x4 u; } db;
      db.u = ds[wave][sub * 16 + l16 + 2 * ks + (kq >> 1)][kq & 1];
      acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa[ks].v, db.v, acc, 0, 0, 0);
    }
    const int ti = (q0 + sub * 16 + l16) * 4 + kq;
    if (ti < a.T_in) {
      const size_t o = ((size_t)n * a.T_in + ti) * a.C_in + (size_t)g * IG;
      float v[4] = {acc[0], acc[1], acc[2], acc[3]};
      if (a.res) {
        const u32x2 rv = *reinterpret_cast<const u32x2*>(a.res + o);
        v[0] += bf_lo(rv.x); v[1] += bf_hi(rv.x); v[2] += bf_lo(rv.y); v[3] += bf_hi(rv.y);
      }
      if (a.mg) {
        const u32x2 mv = *reinterpret_cast<const u32x2*>(a.mg + o);
        v[0] *= bf_lo(mv.x) > 0.f ? 1.f : a.slope; v[1] *= bf_hi(mv.x) > 0.f ? 1.f : a.slope;
        v[2] *= bf_lo(mv.y) > 0.f ? 1.f : a.slope; v[3] *= bf_hi(mv.y) > 0.f ? 1.f : a.slope;
      }
      union { __bf16 e[4]; u32x2 qv; } out;
#pragma unroll
      for (int i = 0; i < 4; ++i) out.e[i] = (__bf16)v[i];
      *reinterpret_cast<u32x2*>(a.y + o) = out.qv;
    }
  }
}

// ---- weight gradient on the matrix cores (16 output channels per group): dw[tap][g*16 + o][c] = sum_{n, t} dy[n][t][o] * x[n][t*stride + tap - pad][c]
// is, per group, the product [16 o] x [t] x [(tap, c)]: rows = o, the reduction runs over the output times, the columns are the
// 4 k (<= 176) pairs kk = tap*4 + c in 11 tiles of 16.  Both operands are needed TIME-major per lane (8 consecutive t), so the tile's
// dY rows are staged transposed ([o][t]) and the input window de-interleaved and transposed ([row mod stride][c][row / stride]:
// for a fixed tap the times t, t+1, .. are consecutive entries).  A = one 16-byte read, B = 8 two-byte reads per fragment; the bias
// gradient is one more MFMA against a vector of ones.  Workgroup = 4 adjacent groups (they share the staged rows) x one split of the
// (item, 64-time tile) list; every split writes its fp32 slab (compact [k][c_out][4] + [c_out]), summed in split order by
// vits_wgrad_reduce_pending (bitwise reproducible).
constexpr int NT_W = 11;                                   // column tiles: 11 x 16 = 176 >= 4 * 41 + (padding pairs, skipped)
constexpr int WIDX = TT + KMAXG / 1 + 8;                   // entries per (phase, c) row of the transposed window (row / stride < 64 + k)
struct GWArgs {
  const __bf16* x; const __bf16* dy; float* partial; float* partial_db;
  int n, T_in, T_out, C_in, C_out, k, stride, pad, S, tiles_per_item;
  size_t slab;
};

__global__ __launch_bounds__(256) void grouped_wgrad_mfma_kernel(GWArgs a) {
  __shared__ __attribute__((aligned(16))) unsigned short dyT[4][16][TT + 8];           // [group][o][t]
  __shared__ unsigned short xT[4][SMAXG][IG][WIDX];                                    // [group][row % s][c][row / s]
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int split = blockIdx.x, g4 = blockIdx.y * 4;
  const int g = g4 + wave;
  const int s = a.stride;
  const int l16 = lane & 15, kq = lane >> 4;
  f32x4_t acc[NT_W], accb = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int j = 0; j < NT_W; ++j) acc[j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  // column (tap, c) of this lane in every tile, as the offset into xT of time 0: tap % s plane, channel c, entry tap / s
  int xoff[NT_W];
#pragma unroll
  for (int j = 0; j < NT_W; ++j) {
    const int kk = j * 16 + l16, tap = kk >> 2, c = kk & 3;
    xoff[j] = tap < a.k ? ((tap % s) * IG + c) * WIDX + tap / s : -1;
  }
  union { bf16x8_t v; unsigned short e[8]; } ones;
#pragma unroll
  for (int i = 0; i < 8; ++i) ones.e[i] = 0x3F80;          // bf16 1.0
  const int n_tiles = a.n * a.tiles_per_item;
  const int nrows = (TT - 1) * s + a.k;
  for (int tile = split; tile < n_tiles; tile += a.S) {
    const int n = tile / a.tiles_per_item, t0 = (tile - n * a.tiles_per_item) * TT;
    const int rows = (a.T_out - t0 < TT) ? (a.T_out - t0) : TT;
    __syncthreads();                                        // the previous tile's staged rows have been consumed
    const __bf16* X = a.x + (size_t)n * a.T_in * a.C_in + (size_t)g4 * IG;
    for (int idx = tid; idx < nrows * 2; idx += 256) {
      const int r = idx >> 1, half = idx & 1;
      const int tin = t0 * s - a.pad + r;
      union { u32x4 u; unsigned short e[8]; } v;
      v.u = u32x4{0u, 0u, 0u, 0u};
      if (tin >= 0 && tin < a.T_in) v.u = *reinterpret_cast<const u32x4*>(X + (size_t)tin * a.C_in + half * 8);
      const int ph = r % s, ix = r / s;
#pragma unroll
      for (int e = 0; e < 8; ++e) xT[half * 2 + (e >> 2)][ph][e & 3][ix] = v.e[e];
    }
    const __bf16* DY = a.dy + ((size_t)n * a.T_out + t0) * a.C_out + (size_t)g4 * 16;
    for (int idx = tid; idx < TT * 8; idx += 256) {
      const int r = idx >> 3, vc = idx & 7;
      union { u32x4 u; unsigned short e[8]; } v;
      v.u = u32x4{0u, 0u, 0u, 0u};
      if (r < rows) v.u = *reinterpret_cast<const u32x4*>(DY + (size_t)r * a.C_out + vc * 8);      // rows past the item's end count as zero
#pragma unroll
      for (int e = 0; e < 8; ++e) dyT[vc >> 1][(vc & 1) * 8 + e][r] = v.e[e];
    }
    __syncthreads();
    const unsigned short* xg = &xT[wave][0][0][0];
#pragma unroll
    for (int ks = 0; ks < TT / 32; ++ks) {
      const int tb = ks * 32 + kq * 8;                      // first of this lane's 8 times
      union { bf16x8_t v; u32x4 u; } fa;
      fa.u = *reinterpret_cast<const u32x4*>(&dyT[wave][l16][tb]);
      accb = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa.v, ones.v, accb, 0, 0, 0);
#pragma unroll
      for (int j = 0; j < NT_W; ++j) {
        union { bf16x8_t v; unsigned short e[8]; } fb;
        const unsigned short* xp = xg + (xoff[j] >= 0 ? xoff[j] : 0) + tb;
#pragma unroll
        for (int i = 0; i < 8; ++i) fb.e[i] = xp[i];
        acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa.v, fb.v, acc[j], 0, 0, 0);
      }
    }
  }
  // slab of this split: lane holds rows o = 4 kq + i of column kk = 16 j + l16
  float* P = a.partial + (size_t)split * a.slab;
#pragma unroll
  for (int j = 0; j < NT_W; ++j) {
    const int kk = j * 16 + l16, tap = kk >> 2, c = kk & 3;
    if (tap < a.k) {
#pragma unroll
      for (int i = 0; i < 4; ++i) P[((size_t)tap * a.C_out + g * 16 + 4 * kq + i) * IG + c] = acc[j][i];
    }
  }
  if (a.partial_db && l16 == 0) {
#pragma unroll
    for (int i = 0; i < 4; ++i) a.partial_db[(size_t)split * a.slab + g * 16 + 4 * kq + i] = accb[i];
  }
}

int grouped_wgrad_splits(int n, int t_out, int groups) {
  const int tiles = n * vits::ceil_div(t_out, TT);
  int S = vits::ceil_div(2048, groups);                     // ~512 workgroups of 4 groups
  if (S > tiles) S = tiles;
  if (S < 1) S = 1;
  return S;
}

int check_geom(int dtype, int n, int t_in, int c_in, int c_out, int k, int stride, int pad, int groups, int* t_out) {
  if (dtype != VITS_DT_BF16) return VITS_E_UNSUPPORTED;
  if (n <= 0 || t_in <= 0 || c_in <= 0 || c_out <= 0 || k <= 0 || stride <= 0 || pad < 0 || groups <= 0) return VITS_E_BADARG;
  if (c_in % groups != 0 || c_out % groups != 0) return VITS_E_BADARG;
  const int ig = c_in / groups, og = c_out / groups;
  if (ig != IG || (og != 16 && og != 4) || groups % 4 != 0 || k > KMAXG || stride > SMAXG || n > 65535) return VITS_E_UNSUPPORTED;
  const int span = t_in + 2 * pad - k;
  if (span < 0) return VITS_E_BADARG;
  *t_out = span / stride + 1;
  return VITS_OK;
}

}  // namespace

extern "C" int vits_grouped_conv_fwd(int dtype, const void* x, const void* w, const float* bias, void* y, int n, int t_in, int c_in,
                                     int c_out, int k, int stride, int pad, int groups, float out_slope, void* stream) {
  if (!x || !w || !y) return VITS_E_BADARG;
  int t_out = 0;
  const int rc = check_geom(dtype, n, t_in, c_in, c_out, k, stride, pad, groups, &t_out);
  if (rc != VITS_OK) return rc;
  GArgs a{static_cast<const __bf16*>(x), static_cast<const __bf16*>(w), bias, static_cast<__bf16*>(y), nullptr, nullptr,
          t_in, t_out, c_in, c_out, k, stride, pad, groups, out_slope};
  const dim3 grid(vits::ceil_div(t_out, TT), groups / 4, n);
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (c_out / groups == 16 && stride % 2 == 0 && k <= 8 * KSTEPS) hipLaunchKernelGGL(grouped_fwd_mfma_kernel, grid, dim3(256), 0, s, a);
  else if (c_out / groups == 16) hipLaunchKernelGGL(grouped_fwd_kernel<16>, grid, dim3(256), 0, s, a);
  else hipLaunchKernelGGL(grouped_fwd_kernel<4>, grid, dim3(256), 0, s, a);
  return vits::check_launch("vits_grouped_conv_fwd");
}

extern "C" int vits_grouped_conv_dgrad(int dtype, const void* dy, const void* w, const void* res, const void* mg_src, void* dx, int n,
                                       int t_in, int c_in, int c_out, int k, int stride, int pad, int groups, float mg_slope,
                                       void* stream) {
  if (!dy || !w || !dx) return VITS_E_BADARG;
  int t_out = 0;
  const int rc = check_geom(dtype, n, t_in, c_in, c_out, k, stride, pad, groups, &t_out);
  if (rc != VITS_OK) return rc;
  GArgs a{static_cast<const __bf16*>(dy), static_cast<const __bf16*>(w), nullptr, static_cast<__bf16*>(dx),
          static_cast<const __bf16*>(res), static_cast<const __bf16*>(mg_src), t_in, t_out, c_in, c_out, k, stride, pad, groups, mg_slope};
  const int q_rows = vits::ceil_div(t_in, stride);
  const dim3 grid(vits::ceil_div(q_rows, TT) * stride, groups / 4, n);
  hipStream_t s = static_cast<hipStream_t>(stream);
  // row offsets r of the matrix-core form: tap = p + pad - 4 r in [0, k) for some phase p
  const int r_max = (3 + pad) / 4;                                                     // p = 3, tap >= 0
  const int r_lo = pad - (k - 1) >= 0 ? (pad - (k - 1) + 3) / 4 : -((k - 1 - pad) / 4);  // p = 0, tap <= k - 1: ceil((pad - (k - 1)) / 4)
  if (c_out / groups == 16 && stride == 4 && r_max - r_lo + 1 <= 2 * KSTEPS) {
    const dim3 gridm(vits::ceil_div(q_rows, TT), groups / 4, n);
    hipLaunchKernelGGL(grouped_dgrad_mfma_kernel, gridm, dim3(256), 0, s, a, r_lo);
  } else if (c_out / groups == 16) hipLaunchKernelGGL(grouped_dgrad_kernel<16>, grid, dim3(256), 0, s, a);
  else hipLaunchKernelGGL(grouped_dgrad_kernel<4>, grid, dim3(256), 0, s, a);
  return vits::check_launch("vits_grouped_conv_dgrad");
}

extern "C" size_t vits_grouped_conv_wgrad_workspace(int n, int t_out, int c_out, int k, int groups) {
  if (n <= 0 || t_out <= 0 || c_out <= 0 || k <= 0 || groups <= 0) return 0;
  return (size_t)grouped_wgrad_splits(n, t_out, groups) * ((size_t)k * c_out * IG + c_out) * sizeof(float);
}

extern "C" int vits_grouped_conv_wgrad(int dtype, const void* x, const void* dy, float* dw, float* dbias, void* workspace,
                                       size_t workspace_bytes, int n, int t_in, int c_in, int c_out, int k, int stride, int pad, int groups,
                                       int accumulate, vits_wgrad_pending* pending, void* stream) {
  if (!x || !dy || !dw || !workspace || !pending) return VITS_E_BADARG;
  int t_out = 0;
  const int rc = check_geom(dtype, n, t_in, c_in, c_out, k, stride, pad, groups, &t_out);
  if (rc != VITS_OK) return rc;
  if (c_out / groups != 16 || k * IG > NT_W * 16) return VITS_E_UNSUPPORTED;
  const int S = grouped_wgrad_splits(n, t_out, groups);
  const size_t nd = (size_t)k * c_out * IG, nb = dbias ? (size_t)c_out : 0, slab = nd + nb;
  if ((size_t)S * slab * sizeof(float) > workspace_bytes) return VITS_E_BADARG;
  float* ws = static_cast<float*>(workspace);
  GWArgs a{static_cast<const __bf16*>(x), static_cast<const __bf16*>(dy), ws, dbias ? ws + nd : nullptr,
           n, t_in, t_out, c_in, c_out, k, stride, pad, S, vits::ceil_div(t_out, TT), slab};
  hipLaunchKernelGGL(grouped_wgrad_mfma_kernel, dim3(S, groups / 4), dim3(256), 0, static_cast<hipStream_t>(stream), a);
  *pending = vits_wgrad_pending{ws, dw, dbias, nd, nb, slab, S, accumulate ? 1 : 0};
  return vits::check_launch("vits_grouped_conv_wgrad");
}
