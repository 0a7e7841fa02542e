// One WaveNet layer of modules.WN as ONE launch per direction (reference modules.py:157-176, commons.py:103-110).
//
// forward  (vits_wn_layer_fwd), per 64-row time tile of one item, 4 waves:
//     pre  = conv_k(h; W_in) + b_in (+ cond[item])                       [2H]   dilated k-tap convolution on the matrix cores
//     acts = tanh(pre[:H]) * sigmoid(pre[H:])                            [H]    gate in the epilogue; the tile stays in LDS
//     rs   = acts . W_rs^T + b_rs                                        [2H | H for the last layer]   second product from that tile
//     h'   = (h + rs[:H]) * mask ;  out (+)= rs[H:] * mask               residual / skip epilogues
//   `pre` and `acts` are also written to memory (the backward's saved tensors).
// backward (vits_wn_layer_bwd), same tiling:
//     d_acts = [d_h | d_o] . W_rs                                        1x1 data gradient, recomputed on the k-1 halo rows too
//     d_pre  = gate'(pre) * d_acts * mask                                [2H]   written to memory (the weight gradients read it) and kept in LDS
//     d_h'   = d_h + conv_k^T(d_pre; W_in)                               k-tap data gradient + the residual path
//
// Operand flow (MI355X-first): the 64(+halo)-row activation tile is staged ONCE in LDS for the whole reduction depth and is shared
// by the four waves; every wave owns its own output columns, so the weights are shared by nobody inside a workgroup and do NOT go
// through LDS: each lane loads its 16-byte MFMA fragments straight from the arena operand ([tap][c_out][c_in], c_in contiguous)
// into a ring of D x NT register fragments — D steps in flight, every load unconditional so that the compiler counts vmcnt
// instead of draining it.  No barrier inside the reduction loops.  The same fragment walk serves bf16 (v_mfma_f32_32x32x16_bf16,
// 8 k per fragment) and exact fp32 (v_mfma_f32_32x32x2_f32, 4 k per fragment = 4 instructions).
//
// Gate columns: output column tile j of the first product holds tanh channels 16j..16j+15 in lanes 0-15 and THEIR sigmoid partners
// (rows H+16j.. of W_in) in lanes 16-31, so the two pre-activations of a channel sit in lanes c and c+16 of one accumulator
// register: one ds_bpermute exchange per register pair, and each lane evaluates 8 of the tile's 16 rows.
#include "common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

constexpr int kThreads = 256;
constexpr int TM = 64;            // time rows per workgroup
constexpr int RT = TM / 32;       // row tiles per wave (every wave covers all rows)
constexpr int D = 6;              // weight-fragment steps in flight per wave

__device__ __forceinline__ float to_f(float v) { return v; }
__device__ __forceinline__ float to_f(__bf16 v) { return (float)v; }
template <typename T> __device__ __forceinline__ T from_f(float v);
template <> __device__ __forceinline__ float from_f<float>(float v) { return v; }
template <> __device__ __forceinline__ __bf16 from_f<__bf16>(float v) { return (__bf16)v; }

__device__ __forceinline__ float sigmoidf_(float v) { return 1.0f / (1.0f + __expf(-v)); }
// as in conv1d_cl.hip: bf16 mode takes the fast exponential, fp32 parity mode keeps libm's tanhf
template <typename T> __device__ __forceinline__ float gate_tanh(float v) {
  if constexpr (sizeof(T) == 2) return 1.0f - 2.0f / (1.0f + __expf(2.0f * v));
  else return tanhf(v);
}

template <typename T>
__device__ __forceinline__ void mma_frag(f32x16& acc, const u32x4& av, const u32x4& bv) {
  if constexpr (sizeof(T) == 2) {
    union { u32x4 u; bf16x8 v; } ua, ub;
    ua.u = av; ub.u = bv;
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ua.v, ub.v, acc, 0, 0, 0);
  } else {
    union { u32x4 u; float f[4]; } ua, ub;
    ua.u = av; ub.u = bv;
#pragma unroll
    for (int j = 0; j < 4; ++j) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ua.f[j], ub.f[j], acc, 0, 0, 0);
  }
}

struct FwdArgs { vits_wn_layer_desc d; int pitch, xrows, spt, P1, P2; };

// cooperative copy of `rows` rows x `rowbytes` bytes (16-byte vectors) from an LDS tile to global rows [t0, t0+rows) ∩ [0, t_hi)
__device__ __forceinline__ void tile_to_global(const unsigned char* lds, int pitch, unsigned char* g, size_t ldg_bytes, int rowbytes,
                                               int rows, int t0, int t_hi) {
  const int vpr = rowbytes / 16;
  for (int idx = threadIdx.x; idx < rows * vpr; idx += kThreads) {
    const int row = idx / vpr, v = idx % vpr;
    if (t0 + row < t_hi) *reinterpret_cast<u32x4*>(g + (size_t)(t0 + row) * ldg_bytes + v * 16) = *reinterpret_cast<const u32x4*>(lds + row * pitch + v * 16);
  }
}

template <typename T, int NT>
__global__ __launch_bounds__(kThreads) void wn_layer_fwd_kernel(FwdArgs args) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const vits_wn_layer_desc& a = args.d;
  constexpr int ES = sizeof(T);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int c = lane & 31, h = lane >> 5;
  const int H = a.h, k = a.k, dil = a.dil;
  const int padr = (k - 1) * dil / 2;
  const int t0 = blockIdx.x * TM, b = blockIdx.y;
  const int T_ = a.t;
  const int len = a.lengths ? (a.lengths[b] < T_ ? a.lengths[b] : T_) : T_;
  const int pitch = args.pitch, spt = args.spt;          // spt = 32-byte fragment steps per tap = H * ES / 32
  const int rowbytes = H * ES;
  unsigned char* ldsX = smem;                            // [xrows][pitch]  h rows t0 - padr ...
  unsigned char* ldsA = smem + (size_t)args.xrows * pitch;   // [TM][pitch]  gate outputs

  const int c_rs = a.last ? H : 2 * H;                   // rows of W_rs
  const int S1 = k * spt, S2 = spt;
  const int P1 = args.P1;                                // S1 rounded up to a multiple of D (dummy steps load, do not multiply)

  // ---- per-lane weight row pointers (bytes): first product (gate-interleaved columns), second product (natural order)
  const unsigned char* w1[NT];
  const unsigned char* w2[NT];
  bool live1[NT], live2[NT];
#pragma unroll
  for (int n = 0; n < NT; ++n) {
    const int tile = wave * NT + n;
    const int ch = 16 * tile + (c & 15);
    live1[n] = 16 * tile < H;
    const int row1 = live1[n] ? (c < 16 ? ch : H + ch) : 0;
    w1[n] = static_cast<const unsigned char*>(a.w_in) + (size_t)row1 * rowbytes + 16 * h;
    const int co = 32 * tile + c;
    live2[n] = 32 * tile < c_rs;
    w2[n] = static_cast<const unsigned char*>(a.w_rs) + (size_t)(co < c_rs ? co : 0) * rowbytes + 16 * h;
  }
  const size_t tap_stride = (size_t)2 * H * rowbytes;

  // fragment of step s (unified numbering: [0, P1) first product incl. dummies, [P1, P1 + P2) second product)
  auto load_step = [&](u32x4 (&dst)[NT], int s) {
    if (s < P1) {
      const int sc = s < S1 ? s : S1 - 1;
      const size_t off = (size_t)(sc / spt) * tap_stride + (size_t)(sc % spt) * 32;
#pragma unroll
      for (int n = 0; n < NT; ++n) dst[n] = *reinterpret_cast<const u32x4*>(w1[n] + off);
    } else {
      int s2 = s - P1;
      s2 = s2 < S2 ? s2 : S2 - 1;
#pragma unroll
      for (int n = 0; n < NT; ++n) dst[n] = *reinterpret_cast<const u32x4*>(w2[n] + (size_t)s2 * 32);
    }
  };

  u32x4 bq[D][NT];
#pragma unroll
  for (int j = 0; j < D; ++j) load_step(bq[j], j);

  // ---- stage the activation tile (rows outside [0, len) read as zero: the input is masked, modules.py:157 x * x_mask upstream)
  {
    const T* X = static_cast<const T*>(a.x) + (size_t)b * T_ * a.ldx;
    const int vpr = rowbytes / 16;
    for (int idx = tid; idx < args.xrows * vpr; idx += kThreads) {
      const int row = idx / vpr, v = idx % vpr;
      const int t = t0 - padr + row;
      u32x4 val = {0u, 0u, 0u, 0u};
      if (t >= 0 && t < len) val = *reinterpret_cast<const u32x4*>(reinterpret_cast<const unsigned char*>(X + (size_t)t * a.ldx) + v * 16);
      *reinterpret_cast<u32x4*>(ldsX + row * pitch + v * 16) = val;
    }
  }
  __syncthreads();

  f32x16 acc[RT][NT];
#pragma unroll
  for (int r = 0; r < RT; ++r)
#pragma unroll
    for (int n = 0; n < NT; ++n)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[r][n][i] = 0.f;

  // ---- first product: pre = conv_k(h)
  for (int s0 = 0; s0 < P1; s0 += D) {
#pragma unroll
    for (int j = 0; j < D; ++j) {
      const int s = s0 + j;
      if (s < S1) {
        const int tap = s / spt, m = s % spt;
        const unsigned char* xa = ldsX + (c + tap * dil) * pitch + 32 * m + 16 * h;
        u32x4 av[RT];
#pragma unroll
        for (int r = 0; r < RT; ++r) av[r] = *reinterpret_cast<const u32x4*>(xa + r * 32 * pitch);
#pragma unroll
        for (int n = 0; n < NT; ++n)
#pragma unroll
          for (int r = 0; r < RT; ++r) mma_frag<T>(acc[r][n], av[r], bq[j][n]);
      }
      load_step(bq[j], s + D);
    }
  }

  // ---- gate epilogue: lanes c and c + 16 hold the tanh / sigmoid pre-activations of one channel
  {
    T* PRE = a.pre ? static_cast<T*>(a.pre) + (size_t)b * T_ * a.ldpre : nullptr;
#pragma unroll
    for (int n = 0; n < NT; ++n) {
      if (!live1[n]) continue;
      const int ch = 16 * (wave * NT + n) + (c & 15);
      const bool lo = c < 16;
      float ba = a.b_in ? a.b_in[ch] : 0.f, bb = a.b_in ? a.b_in[H + ch] : 0.f;
      if (a.cond) { ba += a.cond[(size_t)b * 2 * H + ch]; bb += a.cond[(size_t)b * 2 * H + H + ch]; }
#pragma unroll
      for (int r = 0; r < RT; ++r) {
        if (PRE) {                                        // pre-activations as they stand (bias included), own column
          const float bias_own = lo ? ba : bb;
          const int col = lo ? ch : H + ch;
#pragma unroll
          for (int i = 0; i < 16; ++i) {
            const int t = t0 + r * 32 + (i & 3) + 8 * (i >> 2) + 4 * h;
            if (t < T_) PRE[(size_t)t * a.ldpre + col] = from_f<T>(acc[r][n][i] + bias_own);
          }
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          // low lanes evaluate registers 0..7 (they need the partner's register i), high lanes registers 8..15
          const float send = lo ? acc[r][n][i + 8] : acc[r][n][i];
          const float recv = __shfl_xor(send, 16, 64);
          const float va = (lo ? acc[r][n][i] : recv) + ba;
          const float vb = (lo ? recv : acc[r][n][i + 8]) + bb;
          // the memory image of `pre` is in T: gate on the rounded values so that the backward (which re-reads `pre`) sees the same
          const float ra = to_f(from_f<T>(va)), rb = to_f(from_f<T>(vb));
          const int ii = lo ? i : i + 8;
          const int row = r * 32 + (ii & 3) + 8 * (ii >> 2) + 4 * h;
          *reinterpret_cast<T*>(ldsA + row * pitch + ch * ES) = from_f<T>(gate_tanh<T>(ra) * sigmoidf_(rb));
        }
      }
    }
  }
  __syncthreads();
  if (a.acts)
    tile_to_global(ldsA, pitch, reinterpret_cast<unsigned char*>(static_cast<T*>(a.acts) + (size_t)b * T_ * a.ldacts), (size_t)a.ldacts * ES, rowbytes,
                   TM, t0, T_);

  // ---- second product: rs = acts . W_rs^T (the first D fragment steps are already in flight)
  f32x16 acc2[RT][NT];
#pragma unroll
  for (int r = 0; r < RT; ++r)
#pragma unroll
    for (int n = 0; n < NT; ++n)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc2[r][n][i] = 0.f;
  for (int s0 = 0; s0 < args.P2; s0 += D) {
#pragma unroll
    for (int j = 0; j < D; ++j) {
      const int s = s0 + j;
      if (s < S2) {
        const unsigned char* xa = ldsA + c * pitch + 32 * s + 16 * h;
        u32x4 av[RT];
#pragma unroll
        for (int r = 0; r < RT; ++r) av[r] = *reinterpret_cast<const u32x4*>(xa + r * 32 * pitch);
#pragma unroll
        for (int n = 0; n < NT; ++n)
#pragma unroll
          for (int r = 0; r < RT; ++r) mma_frag<T>(acc2[r][n], av[r], bq[j][n]);
      }
      load_step(bq[j], P1 + s + D);
    }
  }

  // ---- residual / skip epilogues
  T* HO = a.h_out ? static_cast<T*>(a.h_out) + (size_t)b * T_ * a.ldh : nullptr;
  T* SK = static_cast<T*>(a.skip) + (size_t)b * T_ * a.ldskip;
#pragma unroll
  for (int n = 0; n < NT; ++n) {
    if (!live2[n]) continue;
    const int co = 32 * (wave * NT + n) + c;
    if (co >= c_rs) continue;
    const float bias = a.b_rs ? a.b_rs[co] : 0.f;
    const bool is_res = !a.last && co < H;
    const int ch = is_res ? co : (a.last ? co : co - H);
#pragma unroll
    for (int r = 0; r < RT; ++r)
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int row = r * 32 + (i & 3) + 8 * (i >> 2) + 4 * h;
        const int t = t0 + row;
        if (t >= T_) continue;
        const float v = acc2[r][n][i] + bias;
        if (is_res) {
          const float xin = to_f(*reinterpret_cast<const T*>(ldsX + (row + padr) * pitch + ch * ES));
          if (HO) HO[(size_t)t * a.ldh + ch] = from_f<T>(t < len ? xin + v : 0.f);
        } else {
          const size_t o = (size_t)t * a.ldskip + ch;
          const float m = t < len ? v : 0.f;
          SK[o] = from_f<T>(a.accumulate ? to_f(SK[o]) + m : m);
        }
      }
  }
}

template <typename T, int NT>
int launch_fwd(const vits_wn_layer_desc& d, hipStream_t s) {
  FwdArgs args;
  args.d = d;
  const int es = sizeof(T);
  args.pitch = d.h * es + 16;
  args.xrows = TM + (d.k - 1) * d.dil;
  args.spt = d.h * es / 32;
  args.P1 = vits::ceil_div(d.k * args.spt, D) * D;
  args.P2 = vits::ceil_div(args.spt, D) * D;
  const size_t lds = (size_t)(args.xrows + TM) * args.pitch;
  if (lds > (size_t)vits::kLdsBytesMax) return VITS_E_UNSUPPORTED;
  auto kern = wn_layer_fwd_kernel<T, NT>;
  { const hipError_t e = vits::ensure_max_dynamic_lds(reinterpret_cast<const void*>(kern)); if (e != hipSuccess) return vits::note_hip_error(e, "vits_wn_layer_fwd/attr"); }
  dim3 grid(vits::ceil_div(d.t, TM), d.b);
  hipLaunchKernelGGL(kern, grid, dim3(kThreads), lds, s, args);
  return vits::check_launch("vits_wn_layer_fwd");
}


// =============================================================================================================================
// backward: 16x16 tiles so that the four waves split H = 192 output channels evenly (48 = 3 column tiles each) in BOTH products.
//   rows: the workgroup holds R = 16 * RTILES rows of d_pre, t = t0 - pad + r, and emits R - (k-1)*dil rows of d_h
//   (the k-tap data gradient of row t reads d_pre rows t - pad .. t + pad): the halo is recomputed, nothing is exchanged.
// A 16x16 fragment step covers 64 bytes of a row (bf16: 32 k of v_mfma_f32_16x16x32_bf16; f32: 16 k = 4 x v_mfma_f32_16x16x4_f32);
// lane (r = l & 15, q = l >> 4) holds bytes [64 m + 16 q, +16) of row r.  LDS pitch = row bytes + 32 (conflict-free ds_read_b128
// for this lane map: pitch = 8 mod 16 dwords).
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <typename T>
__device__ __forceinline__ void mma16(f32x4& acc, const u32x4& av, const u32x4& bv) {
  if constexpr (sizeof(T) == 2) {
    union { u32x4 u; bf16x8 v; } ua, ub;
    ua.u = av; ub.u = bv;
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ua.v, ub.v, acc, 0, 0, 0);
  } else {
    union { u32x4 u; float f[4]; } ua, ub;
    ua.u = av; ub.u = bv;
#pragma unroll
    for (int j = 0; j < 4; ++j) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(ua.f[j], ub.f[j], acc, 0, 0, 0);
  }
}

struct BwdArgs { vits_wn_layer_bwd_desc d; int pitch, P1, P2, S1, spt; };

template <typename T, int RTILES, int NT>
__global__ __launch_bounds__(kThreads) void wn_layer_bwd_kernel(BwdArgs args) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const vits_wn_layer_bwd_desc& a = args.d;
  constexpr int ES = sizeof(T);
  constexpr int R = 16 * RTILES;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int c = lane & 15, q = lane >> 4;
  const int H = a.h, k = a.k, dil = a.dil;
  const int halo = (k - 1) * dil, pad = halo / 2, TOUT = R - halo;
  const int t0 = blockIdx.x * TOUT, b = blockIdx.y;
  const int T_ = a.t;
  const int len = a.lengths ? (a.lengths[b] < T_ ? a.lengths[b] : T_) : T_;
  const int pitch = args.pitch;
  const int K1 = a.last ? H : 2 * H;                      // reduction depth of the first product (channels of [d_h | d_o])
  const int k1bytes = K1 * ES, rowbytes2 = 2 * H * ES;
  unsigned char* ldsD = smem;                               // [R][pitch]          [d_h | d_o] rows t0 - pad ...
  unsigned char* ldsP = smem + (size_t)R * pitch;           // [R + halo][pitch]   pre -> d_pre (rows >= R zero)

  const int S1 = args.S1, spt = args.spt, P1 = args.P1;     // S1 = ceil(k1bytes / 64); spt = rowbytes2 / 64 steps per tap
  const int S2 = k * spt;

  const unsigned char* w1[NT];
  const unsigned char* w2[NT];
  bool live[NT];
#pragma unroll
  for (int n = 0; n < NT; ++n) {
    const int ch = (wave * NT + n) * 16 + c;
    live[n] = (wave * NT + n) * 16 < H;
    const int row = live[n] ? ch : 0;
    w1[n] = static_cast<const unsigned char*>(a.w_rs_t) + (size_t)row * k1bytes + 16 * q;
    w2[n] = static_cast<const unsigned char*>(a.w_in_t) + (size_t)row * rowbytes2 + 16 * q;
  }
  const size_t tap_stride = (size_t)H * rowbytes2;

  auto load_step = [&](u32x4 (&dst)[NT], int s) {
    if (s < P1) {
      const int sc = s < S1 ? s : S1 - 1;
      const bool ok = 64 * sc + 16 * q < k1bytes;           // a row of W_rs^T may end inside the last 64-byte step (H = 16, bf16)
      const size_t off = ok ? (size_t)sc * 64 : 0;
#pragma unroll
      for (int n = 0; n < NT; ++n) {
        const u32x4 v = *reinterpret_cast<const u32x4*>(w1[n] + off - (ok ? 0 : 16 * q));
        const u32x4 z = {0u, 0u, 0u, 0u};
        dst[n] = ok ? v : z;
      }
    } else {
      int s2 = s - P1;
      s2 = s2 < S2 ? s2 : S2 - 1;
      const size_t off = (size_t)(s2 / spt) * tap_stride + (size_t)(s2 % spt) * 64;
#pragma unroll
      for (int n = 0; n < NT; ++n) dst[n] = *reinterpret_cast<const u32x4*>(w2[n] + off);
    }
  };

  u32x4 bq[D][NT];
#pragma unroll
  for (int j = 0; j < D; ++j) load_step(bq[j], j);

  // ---- stage [d_h | d_o] and pre (rows outside [0, T) zero)
  {
    const int vh = H * ES / 16;                             // 16-byte vectors per H-channel half-row
    const unsigned char* DH = a.d_h ? reinterpret_cast<const unsigned char*>(static_cast<const T*>(a.d_h) + (size_t)b * T_ * a.ld_dh) : nullptr;
    const unsigned char* DO = reinterpret_cast<const unsigned char*>(static_cast<const T*>(a.d_o) + (size_t)b * T_ * a.ld_do);
    const unsigned char* PR = reinterpret_cast<const unsigned char*>(static_cast<const T*>(a.pre) + (size_t)b * T_ * a.ldpre);
    const int vpr = 2 * vh;
    for (int idx = tid; idx < (R + halo) * vpr; idx += kThreads) {
      const int row = idx / vpr, v = idx % vpr;
      const int t = t0 - pad + row;
      const bool in = row < R && t >= 0 && t < T_;
      u32x4 pv = {0u, 0u, 0u, 0u}, dv = {0u, 0u, 0u, 0u};
      if (in) {
        pv = *reinterpret_cast<const u32x4*>(PR + (size_t)t * a.ldpre * ES + v * 16);
        if (a.last) { if (v < vh) dv = *reinterpret_cast<const u32x4*>(DO + (size_t)t * a.ld_do * ES + v * 16); }
        else if (v < vh) dv = *reinterpret_cast<const u32x4*>(DH + (size_t)t * a.ld_dh * ES + v * 16);
        else dv = *reinterpret_cast<const u32x4*>(DO + (size_t)t * a.ld_do * ES + (v - vh) * 16);
      }
      *reinterpret_cast<u32x4*>(ldsP + row * pitch + v * 16) = pv;
      if (row < R) *reinterpret_cast<u32x4*>(ldsD + row * pitch + v * 16) = dv;
    }
  }
  __syncthreads();

  f32x4 acc[RTILES][NT];
#pragma unroll
  for (int r = 0; r < RTILES; ++r)
#pragma unroll
    for (int n = 0; n < NT; ++n) acc[r][n] = f32x4{0.f, 0.f, 0.f, 0.f};

  // ---- first product: d_acts = [d_h | d_o] . W_rs
  for (int s0 = 0; s0 < P1; s0 += D) {
#pragma unroll
    for (int j = 0; j < D; ++j) {
      const int s = s0 + j;
      if (s < S1) {
        const unsigned char* xa = ldsD + c * pitch + 64 * s + 16 * q;
        u32x4 av[RTILES];
#pragma unroll
        for (int r = 0; r < RTILES; ++r) av[r] = *reinterpret_cast<const u32x4*>(xa + r * 16 * pitch);
#pragma unroll
        for (int n = 0; n < NT; ++n)
#pragma unroll
          for (int r = 0; r < RTILES; ++r) mma16<T>(acc[r][n], av[r], bq[j][n]);
      }
      load_step(bq[j], s + D);
    }
  }

  // ---- gate chain rule, in place on the staged pre tile: lane (c, q) holds rows 4q .. 4q+3 of each 16-row tile, column ch
#pragma unroll
  for (int n = 0; n < NT; ++n) {
    if (!live[n]) continue;
    const int ch = (wave * NT + n) * 16 + c;
#pragma unroll
    for (int r = 0; r < RTILES; ++r)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int row = 16 * r + 4 * q + i;
        const int t = t0 - pad + row;
        T* pa = reinterpret_cast<T*>(ldsP + row * pitch) + ch;
        T* pb = pa + H;
        const float ta = gate_tanh<T>(to_f(*pa)), sb = sigmoidf_(to_f(*pb));
        const float dv = (t >= 0 && t < len) ? acc[r][n][i] : 0.f;
        *pa = from_f<T>(dv * sb * (1.0f - ta * ta));
        *pb = from_f<T>(dv * ta * sb * (1.0f - sb));
      }
  }
  __syncthreads();
  // d_pre (centre rows only: every row is written by exactly one workgroup) -> memory, for the weight gradients
  {
    const int vpr = rowbytes2 / 16;
    unsigned char* DP = reinterpret_cast<unsigned char*>(static_cast<T*>(a.d_pre) + (size_t)b * T_ * a.lddpre);
    for (int idx = tid; idx < TOUT * vpr; idx += kThreads) {
      const int j = idx / vpr, v = idx % vpr;
      const int t = t0 + j;
      if (t < T_) *reinterpret_cast<u32x4*>(DP + (size_t)t * a.lddpre * ES + v * 16) = *reinterpret_cast<const u32x4*>(ldsP + (j + pad) * pitch + v * 16);
    }
  }

  // ---- second product: conv^T(d_pre; W_in): output row j reads d_pre tile rows j + tap * dil
  f32x4 acc2[RTILES][NT];
#pragma unroll
  for (int r = 0; r < RTILES; ++r)
#pragma unroll
    for (int n = 0; n < NT; ++n) acc2[r][n] = f32x4{0.f, 0.f, 0.f, 0.f};
  for (int s0 = 0; s0 < args.P2; s0 += D) {
#pragma unroll
    for (int j = 0; j < D; ++j) {
      const int s = s0 + j;
      if (s < S2) {
        const int tap = s / spt, m = s % spt;
        const unsigned char* xa = ldsP + (c + tap * dil) * pitch + 64 * m + 16 * q;
        u32x4 av[RTILES];
#pragma unroll
        for (int r = 0; r < RTILES; ++r) av[r] = *reinterpret_cast<const u32x4*>(xa + r * 16 * pitch);
#pragma unroll
        for (int n = 0; n < NT; ++n)
#pragma unroll
          for (int r = 0; r < RTILES; ++r) mma16<T>(acc2[r][n], av[r], bq[j][n]);
      }
      if (s + D < args.P2) load_step(bq[j], P1 + s + D);
    }
  }

  // ---- d_h' = (d_h + conv^T) * mask
  T* OUT = static_cast<T*>(a.d_h_out) + (size_t)b * T_ * a.ldout;
#pragma unroll
  for (int n = 0; n < NT; ++n) {
    if (!live[n]) continue;
    const int ch = (wave * NT + n) * 16 + c;
#pragma unroll
    for (int r = 0; r < RTILES; ++r)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int j = 16 * r + 4 * q + i;
        const int t = t0 + j;
        if (j >= TOUT || t >= T_) continue;
        float v = acc2[r][n][i];
        if (!a.last) v += to_f(*(reinterpret_cast<const T*>(ldsD + (j + pad) * pitch) + ch));
        OUT[(size_t)t * a.ldout + ch] = from_f<T>(t < len ? v : 0.f);
      }
  }
}

template <typename T, int RTILES, int NT>
int launch_bwd(const vits_wn_layer_bwd_desc& d, hipStream_t s) {
  BwdArgs args;
  args.d = d;
  const int es = sizeof(T), R = 16 * RTILES, halo = (d.k - 1) * d.dil;
  if (halo >= R) return VITS_E_UNSUPPORTED;
  args.pitch = 2 * d.h * es + 32;
  args.spt = 2 * d.h * es / 64;
  args.S1 = vits::ceil_div((d.last ? d.h : 2 * d.h) * es, 64);
  args.P1 = vits::ceil_div(args.S1, D) * D;
  args.P2 = vits::ceil_div(d.k * args.spt, D) * D;
  const size_t lds = (size_t)(2 * R + halo) * args.pitch;
  if (lds > (size_t)vits::kLdsBytesMax) return VITS_E_UNSUPPORTED;
  auto kern = wn_layer_bwd_kernel<T, RTILES, NT>;
  { const hipError_t e = vits::ensure_max_dynamic_lds(reinterpret_cast<const void*>(kern)); if (e != hipSuccess) return vits::note_hip_error(e, "vits_wn_layer_bwd/attr"); }
  dim3 grid(vits::ceil_div(d.t, R - halo), d.b);
  hipLaunchKernelGGL(kern, grid, dim3(kThreads), lds, s, args);
  return vits::check_launch("vits_wn_layer_bwd");
}

}  // namespace

extern "C" int vits_wn_layer_fwd(const vits_wn_layer_desc* desc, void* stream) {
  if (!desc) return VITS_E_BADARG;
  vits_wn_layer_desc d = *desc;
  if (!d.x || !d.w_in || !d.w_rs || !d.skip || d.b <= 0 || d.t <= 0 || d.h <= 0 || d.k <= 0 || (d.k & 1) == 0 || d.dil <= 0) return VITS_E_BADARG;
  if (!d.last && !d.h_out) return VITS_E_BADARG;
  if (d.h % 16 != 0 || d.h > 192) return VITS_E_UNSUPPORTED;        // gate interleave granularity; 4 waves x 3 column tiles
  if (d.ldx <= 0) d.ldx = d.h;
  if (d.ldh <= 0) d.ldh = d.h;
  if (d.ldskip <= 0) d.ldskip = d.h;
  if (d.ldacts <= 0) d.ldacts = d.h;
  if (d.ldpre <= 0) d.ldpre = 2 * d.h;
  const int vec = d.dtype == VITS_DT_BF16 ? 8 : (d.dtype == VITS_DT_F32 ? 4 : 0);
  if (vec == 0) return VITS_E_UNSUPPORTED;
  if (d.ldx % vec != 0 || d.ldacts % vec != 0) return VITS_E_UNSUPPORTED;
  hipStream_t s = static_cast<hipStream_t>(stream);
  const int nt = vits::ceil_div(d.h, 64);
  if (d.dtype == VITS_DT_BF16) {
    if (nt == 1) return launch_fwd<__bf16, 1>(d, s);
    if (nt == 2) return launch_fwd<__bf16, 2>(d, s);
    return launch_fwd<__bf16, 3>(d, s);
  }
  if (nt == 1) return launch_fwd<float, 1>(d, s);
  if (nt == 2) return launch_fwd<float, 2>(d, s);
  return launch_fwd<float, 3>(d, s);
}

extern "C" int vits_wn_layer_bwd(const vits_wn_layer_bwd_desc* desc, void* stream) {
  if (!desc) return VITS_E_BADARG;
  vits_wn_layer_bwd_desc d = *desc;
  if (!d.d_o || !d.pre || !d.w_rs_t || !d.w_in_t || !d.d_pre || !d.d_h_out || d.b <= 0 || d.t <= 0 || d.h <= 0 || d.k <= 0 || (d.k & 1) == 0 ||
      d.dil <= 0)
    return VITS_E_BADARG;
  if (!d.last && !d.d_h) return VITS_E_BADARG;
  if (d.h % 16 != 0 || d.h > 192) return VITS_E_UNSUPPORTED;
  if (d.ld_dh <= 0) d.ld_dh = d.h;
  if (d.ld_do <= 0) d.ld_do = d.h;
  if (d.ldout <= 0) d.ldout = d.h;
  if (d.ldpre <= 0) d.ldpre = 2 * d.h;
  if (d.lddpre <= 0) d.lddpre = 2 * d.h;
  const int vec = d.dtype == VITS_DT_BF16 ? 8 : (d.dtype == VITS_DT_F32 ? 4 : 0);
  if (vec == 0) return VITS_E_UNSUPPORTED;
  if (d.ld_dh % vec != 0 || d.ld_do % vec != 0 || d.ldpre % vec != 0 || d.lddpre % vec != 0) return VITS_E_UNSUPPORTED;
  hipStream_t s = static_cast<hipStream_t>(stream);
  const int nt = vits::ceil_div(d.h, 64);
  if (d.dtype == VITS_DT_BF16) {
    if (nt == 1) return launch_bwd<__bf16, 4, 1>(d, s);
    if (nt == 2) return launch_bwd<__bf16, 4, 2>(d, s);
    return launch_bwd<__bf16, 4, 3>(d, s);
  }
  if (nt == 1) return launch_bwd<float, 2, 1>(d, s);
  if (nt == 2) return launch_bwd<float, 2, 2>(d, s);
  return launch_bwd<float, 2, 3>(d, s);
}
