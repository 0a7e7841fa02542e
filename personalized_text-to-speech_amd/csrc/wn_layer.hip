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

struct FwdArgs { vits_wn_layer_desc d; int pitch, pitchP, xrows, spt, P1, P2; };

// 16-byte vectors a thread holds when a workgroup stages `nvec` of them in one batch (all loads issued, then all LDS stores)
constexpr int kStageMax = 10;

// cooperative copy of `rows` rows x `rowbytes` bytes (16-byte vectors) from an LDS tile to global rows [t0, t0+rows) ∩ [0, t_hi)
__device__ __forceinline__ void tile_to_global(const unsigned char* lds, int pitch, unsigned char* g, size_t ldg_bytes, int rowbytes,
                                               int rows, int t0, int t_hi) {
  const int vpr = rowbytes / 16;
  for (int idx = threadIdx.x; idx < rows * vpr; idx += kThreads) {
    const int row = idx / vpr, v = idx % vpr;
    if (t0 + row < t_hi) *reinterpret_cast<u32x4*>(g + (size_t)(t0 + row) * ldg_bytes + v * 16) = *reinterpret_cast<const u32x4*>(lds + row * pitch + v * 16);
  }
}

// 16 bytes of T as floats and back
template <typename T> struct Vec16 {
  static constexpr int N = 16 / sizeof(T);
  union { u32x4 u; T e[N]; };
};

// RT row tiles of 32 rows per workgroup: 2 in bf16; 1 in fp32, whose tiles are twice as large in LDS.
template <typename T, int NT, int RT>
__global__ __launch_bounds__(kThreads) void wn_layer_fwd_kernel(FwdArgs args) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const vits_wn_layer_desc& a = args.d;
  constexpr int ES = sizeof(T);
  constexpr int TM = 32 * RT;
  constexpr int VN = 16 / ES;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int c = lane & 31, h = lane >> 5;
  const int H = a.h, k = a.k, dil = a.dil;
  const int padr = (k - 1) * dil / 2;
  const int t0 = blockIdx.x * TM, b = blockIdx.y;
  const int T_ = a.t;
  const int len = a.lengths ? (a.lengths[b] < T_ ? a.lengths[b] : T_) : T_;
  const int pitch = args.pitch, pitchP = args.pitchP, spt = args.spt;      // spt = 32-byte fragment steps per tap = H * ES / 32
  const int rowbytes = H * ES;
  unsigned char* ldsX = smem;                                   // [xrows][pitch]   h rows t0 - padr ...
  unsigned char* ldsA = smem + (size_t)args.xrows * pitch;      // [TM][pitch]      gate outputs
  unsigned char* ldsP = ldsA + (size_t)TM * pitch;              // [TM][pitchP]     pre-activations (T), later res | skip (fp32)

  const int c_rs = a.last ? H : 2 * H;                          // rows of W_rs
  const int S1 = k * spt, S2 = spt;
  const int P1 = args.P1;                                       // S1 rounded up to a multiple of D (dummy steps load, do not multiply)

  // ---- per-lane fragment streams of the PACKED operands (vits_wn_pack: [column tile][step][lane][16 bytes], dead tiles zero):
  // every wave-instruction below reads 1 KiB of consecutive memory
  const unsigned char* w1[NT];
  const unsigned char* w2[NT];
  bool live1[NT], live2[NT];
#pragma unroll
  for (int n = 0; n < NT; ++n) {
    const int tile = wave * NT + n;
    live1[n] = 16 * tile < H;
    live2[n] = 32 * tile < c_rs;
    w1[n] = static_cast<const unsigned char*>(a.w_in) + ((size_t)tile * S1 * 64 + lane) * 16;
    w2[n] = static_cast<const unsigned char*>(a.w_rs) + ((size_t)tile * S2 * 64 + lane) * 16;
  }

  // fragment of step s (unified numbering: [0, P1) first product incl. dummies, [P1, P1 + P2) second product)
  auto load_step = [&](u32x4 (&dst)[NT], int s) {
    if (s < P1) {
      const size_t off = (size_t)(s < S1 ? s : S1 - 1) * 1024;
#pragma unroll
      for (int n = 0; n < NT; ++n) dst[n] = *reinterpret_cast<const u32x4*>(w1[n] + off);
    } else {
      const int s2 = s - P1;
      const size_t off = (size_t)(s2 < S2 ? s2 : S2 - 1) * 1024;
#pragma unroll
      for (int n = 0; n < NT; ++n) dst[n] = *reinterpret_cast<const u32x4*>(w2[n] + off);
    }
  };

  // ---- stage the activation tile: all of a thread's loads first, then its LDS stores (rows outside [0, len) read as zero: the
  // input is masked, modules.py:157 x * x_mask upstream)
  {
    const unsigned char* X = reinterpret_cast<const unsigned char*>(static_cast<const T*>(a.x) + (size_t)b * T_ * a.ldx);
    const int vpr = rowbytes / 16, nvec = args.xrows * vpr;
    u32x4 xv[kStageMax];
#pragma unroll
    for (int i = 0; i < kStageMax; ++i) {
      const int idx = tid + i * kThreads, row = idx / vpr, v = idx % vpr;
      const int t = t0 - padr + row;
      u32x4 val = {0u, 0u, 0u, 0u};
      if (idx < nvec && t >= 0 && t < len) val = *reinterpret_cast<const u32x4*>(X + (size_t)t * a.ldx * ES + v * 16);
      xv[i] = val;
    }
#pragma unroll
    for (int i = 0; i < kStageMax; ++i) {
      const int idx = tid + i * kThreads;
      if (idx < nvec) *reinterpret_cast<u32x4*>(ldsX + (idx / vpr) * pitch + (idx % vpr) * 16) = xv[i];
    }
  }
  u32x4 bq[D][NT];
#pragma unroll
  for (int j = 0; j < D; ++j) load_step(bq[j], j);
  __syncthreads();

  f32x16 acc[RT][NT];
#pragma unroll
  for (int r = 0; r < RT; ++r)
#pragma unroll
    for (int n = 0; n < NT; ++n)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[r][n][i] = 0.f;

  // ---- first product: pre = conv_k(h).  The input fragments of step s + 1 are read from LDS before step s multiplies (the LDS
  // latency hides behind six MFMAs); (tap, m) of the next read advance as counters (no division by the runtime steps-per-tap).
  static_assert(D % 2 == 0, "the fragment double buffer alternates with the unrolled step index");
  {
    const unsigned char* xbase = ldsX + c * pitch + 16 * h;
    int tap_n = 0, m_n = 0;                               // (tap, 32-byte step inside the row) of the next fragment read
    u32x4 av[2][RT];
    auto read_a = [&](u32x4 (&dst)[RT]) {
      const unsigned char* xa = xbase + tap_n * dil * pitch + 32 * m_n;
#pragma unroll
      for (int r = 0; r < RT; ++r) dst[r] = *reinterpret_cast<const u32x4*>(xa + r * 32 * pitch);
      if (++m_n == spt) { m_n = 0; ++tap_n; }
    };
    read_a(av[0]);
    for (int s0 = 0; s0 < P1; s0 += D) {
#pragma unroll
      for (int j = 0; j < D; ++j) {
        const int s = s0 + j;
        if (s + 1 < S1) read_a(av[(j + 1) & 1]);
        if (s < S1) {
#pragma unroll
          for (int n = 0; n < NT; ++n)
#pragma unroll
            for (int r = 0; r < RT; ++r) mma_frag<T>(acc[r][n], av[j & 1][r], bq[j][n]);
        }
        load_step(bq[j], s + D);
      }
    }
  }

  // ---- gate epilogue: lanes c and c + 16 hold the tanh / sigmoid pre-activations of one channel.  Everything goes to LDS tiles
  // (pre as T, acts as T) and leaves for memory as 16-byte vectors afterwards.
#pragma unroll
  for (int n = 0; n < NT; ++n) {
    if (!live1[n]) continue;
    const int ch = 16 * (wave * NT + n) + (c & 15);
    const bool lo = c < 16;
    float ba = a.b_in ? a.b_in[ch] : 0.f, bb = a.b_in ? a.b_in[H + ch] : 0.f;
    if (a.cond) { ba += a.cond[(size_t)b * 2 * H + ch]; bb += a.cond[(size_t)b * 2 * H + H + ch]; }
    const float bias_own = lo ? ba : bb;
    const int col = lo ? ch : H + ch;
#pragma unroll
    for (int r = 0; r < RT; ++r) {
#pragma unroll
      for (int i = 0; i < 16; ++i) {                     // pre-activations as they stand (bias included), own column
        const int row = r * 32 + (i & 3) + 8 * (i >> 2) + 4 * h;
        *reinterpret_cast<T*>(ldsP + row * pitchP + col * ES) = from_f<T>(acc[r][n][i] + bias_own);
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
  __syncthreads();
  if (a.pre)
    tile_to_global(ldsP, pitchP, reinterpret_cast<unsigned char*>(static_cast<T*>(a.pre) + (size_t)b * T_ * a.ldpre), (size_t)a.ldpre * ES, 2 * rowbytes,
                   TM, t0, T_);
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
  {
    const unsigned char* abase = ldsA + c * pitch + 16 * h;
    u32x4 av[2][RT];
#pragma unroll
    for (int r = 0; r < RT; ++r) av[0][r] = *reinterpret_cast<const u32x4*>(abase + r * 32 * pitch);
    for (int s0 = 0; s0 < args.P2; s0 += D) {
#pragma unroll
      for (int j = 0; j < D; ++j) {
        const int s = s0 + j;
        if (s + 1 < S2) {
#pragma unroll
          for (int r = 0; r < RT; ++r) av[(j + 1) & 1][r] = *reinterpret_cast<const u32x4*>(abase + 32 * (s + 1) + r * 32 * pitch);
        }
        if (s < S2) {
#pragma unroll
          for (int n = 0; n < NT; ++n)
#pragma unroll
            for (int r = 0; r < RT; ++r) mma_frag<T>(acc2[r][n], av[j & 1][r], bq[j][n]);
        }
        if (s + D < args.P2) load_step(bq[j], P1 + s + D);
      }
    }
  }

  // ---- residual / skip epilogues: rs + bias as fp32 into the (now free) pre tile, then 16-byte vectors: h' = (h + res) * mask
  // from the staged input rows, skip (+)= skip-half * mask with all the old values loaded before any is needed
  __syncthreads();                                        // every thread is done copying the pre tile out
  float* rsT = reinterpret_cast<float*>(ldsP);
  const int pitchF = pitchP / 4;
#pragma unroll
  for (int n = 0; n < NT; ++n) {
    if (!live2[n]) continue;
    const int co = 32 * (wave * NT + n) + c;
    if (co >= c_rs) continue;
    const float bias = a.b_rs ? a.b_rs[co] : 0.f;
#pragma unroll
    for (int r = 0; r < RT; ++r)
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int row = r * 32 + (i & 3) + 8 * (i >> 2) + 4 * h;
        rsT[row * pitchF + co] = acc2[r][n][i] + bias;
      }
  }
  __syncthreads();
  {
    const int vpr = rowbytes / 16, nvec = TM * vpr;       // vectors of one H-channel half of the tile
    constexpr int NV = (TM * 12 * (int)sizeof(T) + kThreads - 1) / kThreads;        // H <= 192: at most TM * 24 (bf16) / 48 (f32) vectors
    if (!a.last && a.h_out) {
      unsigned char* HO = reinterpret_cast<unsigned char*>(static_cast<T*>(a.h_out) + (size_t)b * T_ * a.ldh);
#pragma unroll
      for (int i = 0; i < NV; ++i) {
        const int idx = tid + i * kThreads, row = idx / vpr, v = idx % vpr, t = t0 + row;
        if (idx < nvec && t < T_) {
          Vec16<T> xin, out;
          xin.u = *reinterpret_cast<const u32x4*>(ldsX + (row + padr) * pitch + v * 16);
          const float* rp = rsT + row * pitchF + v * VN;
#pragma unroll
          for (int e = 0; e < VN; ++e) out.e[e] = from_f<T>(t < len ? to_f(xin.e[e]) + rp[e] : 0.f);
          *reinterpret_cast<u32x4*>(HO + (size_t)t * a.ldh * ES + v * 16) = out.u;
        }
      }
    }
    unsigned char* SK = reinterpret_cast<unsigned char*>(static_cast<T*>(a.skip) + (size_t)b * T_ * a.ldskip);
    const int sk0 = a.last ? 0 : H;                       // first column of the skip half in the rs tile
    Vec16<T> old[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int idx = tid + i * kThreads, row = idx / vpr, v = idx % vpr, t = t0 + row;
      old[i].u = u32x4{0u, 0u, 0u, 0u};
      if (a.accumulate && idx < nvec && t < T_) old[i].u = *reinterpret_cast<const u32x4*>(SK + (size_t)t * a.ldskip * ES + v * 16);
    }
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int idx = tid + i * kThreads, row = idx / vpr, v = idx % vpr, t = t0 + row;
      if (idx < nvec && t < T_) {
        const float* rp = rsT + row * pitchF + sk0 + v * VN;
        Vec16<T> out;
#pragma unroll
        for (int e = 0; e < VN; ++e) out.e[e] = from_f<T>(to_f(old[i].e[e]) + (t < len ? rp[e] : 0.f));
        *reinterpret_cast<u32x4*>(SK + (size_t)t * a.ldskip * ES + v * 16) = out.u;
      }
    }
  }
}

template <typename T, int NT, int RT>
int launch_fwd(const vits_wn_layer_desc& d, hipStream_t s) {
  FwdArgs args;
  args.d = d;
  const int es = sizeof(T), TM = 32 * RT;
  args.pitch = d.h * es + 16;
  args.pitchP = 2 * d.h * 4 + 16;                          // fp32 [2H] rows (the pre tile in T fits inside)
  args.xrows = TM + (d.k - 1) * d.dil;
  args.spt = d.h * es / 32;
  args.P1 = vits::ceil_div(d.k * args.spt, D) * D;
  args.P2 = vits::ceil_div(args.spt, D) * D;
  if (args.xrows * (d.h * es / 16) > kStageMax * kThreads) return VITS_E_UNSUPPORTED;
  const size_t lds = (size_t)(args.xrows + TM) * args.pitch + (size_t)TM * args.pitchP;
  if (lds > (size_t)vits::kLdsBytesMax) return VITS_E_UNSUPPORTED;
  auto kern = wn_layer_fwd_kernel<T, NT, RT>;
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
  const int rowbytes2 = 2 * H * ES;
  unsigned char* ldsD = smem;                               // [R][pitch]          [d_h | d_o] rows t0 - pad ...
  unsigned char* ldsP = smem + (size_t)R * pitch;           // [R + halo][pitch]   pre -> d_pre (rows >= R zero)

  const int S1 = args.S1, spt = args.spt, P1 = args.P1;     // S1 = ceil(k1bytes / 64); spt = rowbytes2 / 64 steps per tap
  const int S2 = k * spt;

  // packed data-gradient operands (vits_wn_pack mode 2: [16-channel tile][step][lane][16 bytes], tails and dead tiles zero)
  const unsigned char* w1[NT];
  const unsigned char* w2[NT];
  bool live[NT];
#pragma unroll
  for (int n = 0; n < NT; ++n) {
    const int tile = wave * NT + n;
    live[n] = tile * 16 < H;
    w1[n] = static_cast<const unsigned char*>(a.w_rs_t) + ((size_t)tile * S1 * 64 + lane) * 16;
    w2[n] = static_cast<const unsigned char*>(a.w_in_t) + ((size_t)tile * S2 * 64 + lane) * 16;
  }

  auto load_step = [&](u32x4 (&dst)[NT], int s) {
    if (s < P1) {
      const size_t off = (size_t)(s < S1 ? s : S1 - 1) * 1024;
#pragma unroll
      for (int n = 0; n < NT; ++n) dst[n] = *reinterpret_cast<const u32x4*>(w1[n] + off);
    } else {
      const int s2 = s - P1;
      const size_t off = (size_t)(s2 < S2 ? s2 : S2 - 1) * 1024;
#pragma unroll
      for (int n = 0; n < NT; ++n) dst[n] = *reinterpret_cast<const u32x4*>(w2[n] + off);
    }
  };

  u32x4 bq[D][NT];
#pragma unroll
  for (int j = 0; j < D; ++j) load_step(bq[j], j);

  // ---- stage [d_h | d_o] and pre (rows outside [0, T) zero): batches of kStageMax vectors per thread, loads first, LDS stores after
  {
    const int vh = H * ES / 16;                             // 16-byte vectors per H-channel half-row
    const unsigned char* DH = a.d_h ? reinterpret_cast<const unsigned char*>(static_cast<const T*>(a.d_h) + (size_t)b * T_ * a.ld_dh) : nullptr;
    const unsigned char* DO = reinterpret_cast<const unsigned char*>(static_cast<const T*>(a.d_o) + (size_t)b * T_ * a.ld_do);
    const unsigned char* PR = reinterpret_cast<const unsigned char*>(static_cast<const T*>(a.pre) + (size_t)b * T_ * a.ldpre);
    const int vpr = 2 * vh, nvec = (R + halo) * vpr;
    for (int base = 0; base < nvec; base += kStageMax / 2 * kThreads) {
      u32x4 pv[kStageMax / 2], dv[kStageMax / 2];
#pragma unroll
      for (int i = 0; i < kStageMax / 2; ++i) {
        const int idx = base + tid + i * kThreads, row = idx / vpr, v = idx % vpr;
        const int t = t0 - pad + row;
        const bool in = idx < nvec && row < R && t >= 0 && t < T_;
        pv[i] = dv[i] = u32x4{0u, 0u, 0u, 0u};
        if (in) {
          pv[i] = *reinterpret_cast<const u32x4*>(PR + (size_t)t * a.ldpre * ES + v * 16);
          if (a.last) { if (v < vh) dv[i] = *reinterpret_cast<const u32x4*>(DO + (size_t)t * a.ld_do * ES + v * 16); }
          else if (v < vh) dv[i] = *reinterpret_cast<const u32x4*>(DH + (size_t)t * a.ld_dh * ES + v * 16);
          else dv[i] = *reinterpret_cast<const u32x4*>(DO + (size_t)t * a.ld_do * ES + (v - vh) * 16);
        }
      }
#pragma unroll
      for (int i = 0; i < kStageMax / 2; ++i) {
        const int idx = base + tid + i * kThreads, row = idx / vpr, v = idx % vpr;
        if (idx < nvec) {
          *reinterpret_cast<u32x4*>(ldsP + row * pitch + v * 16) = pv[i];
          if (row < R) *reinterpret_cast<u32x4*>(ldsD + row * pitch + v * 16) = dv[i];
        }
      }
    }
  }
  __syncthreads();

  f32x4 acc[RTILES][NT];
#pragma unroll
  for (int r = 0; r < RTILES; ++r)
#pragma unroll
    for (int n = 0; n < NT; ++n) acc[r][n] = f32x4{0.f, 0.f, 0.f, 0.f};

  // ---- first product: d_acts = [d_h | d_o] . W_rs (fragments of step s + 1 read from LDS before step s multiplies)
  {
    const unsigned char* dbase = ldsD + c * pitch + 16 * q;
    u32x4 av[2][RTILES];
#pragma unroll
    for (int r = 0; r < RTILES; ++r) av[0][r] = *reinterpret_cast<const u32x4*>(dbase + r * 16 * pitch);
    for (int s0 = 0; s0 < P1; s0 += D) {
#pragma unroll
      for (int j = 0; j < D; ++j) {
        const int s = s0 + j;
        if (s + 1 < S1) {
#pragma unroll
          for (int r = 0; r < RTILES; ++r) av[(j + 1) & 1][r] = *reinterpret_cast<const u32x4*>(dbase + 64 * (s + 1) + r * 16 * pitch);
        }
        if (s < S1) {
#pragma unroll
          for (int n = 0; n < NT; ++n)
#pragma unroll
            for (int r = 0; r < RTILES; ++r) mma16<T>(acc[r][n], av[j & 1][r], bq[j][n]);
        }
        load_step(bq[j], s + D);
      }
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
  {
    const unsigned char* pbase = ldsP + c * pitch + 16 * q;
    int tap_n = 0, m_n = 0;                                // (tap, 64-byte step inside the row) of the next fragment read
    u32x4 av[2][RTILES];
    auto read_a = [&](u32x4 (&dst)[RTILES]) {
      const unsigned char* xa = pbase + tap_n * dil * pitch + 64 * m_n;
#pragma unroll
      for (int r = 0; r < RTILES; ++r) dst[r] = *reinterpret_cast<const u32x4*>(xa + r * 16 * pitch);
      if (++m_n == spt) { m_n = 0; ++tap_n; }
    };
    read_a(av[0]);
    for (int s0 = 0; s0 < args.P2; s0 += D) {
#pragma unroll
      for (int j = 0; j < D; ++j) {
        const int s = s0 + j;
        if (s + 1 < S2) read_a(av[(j + 1) & 1]);
        if (s < S2) {
#pragma unroll
          for (int n = 0; n < NT; ++n)
#pragma unroll
            for (int r = 0; r < RTILES; ++r) mma16<T>(acc2[r][n], av[j & 1][r], bq[j][n]);
        }
        if (s + D < args.P2) load_step(bq[j], P1 + s + D);
      }
    }
  }

  // ---- d_h' = (d_h + conv^T) * mask: summed in fp32, rounded once, into the (now free) d_pre tile, then 16-byte vector stores
  __syncthreads();                                         // every wave is done reading the d_pre tile
#pragma unroll
  for (int n = 0; n < NT; ++n) {
    if (!live[n]) continue;
    const int ch = (wave * NT + n) * 16 + c;
#pragma unroll
    for (int r = 0; r < RTILES; ++r)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int j = 16 * r + 4 * q + i;
        if (j >= TOUT) continue;
        float v = acc2[r][n][i];
        if (!a.last) v += to_f(*(reinterpret_cast<const T*>(ldsD + (j + pad) * pitch) + ch));
        *(reinterpret_cast<T*>(ldsP + j * pitch) + ch) = from_f<T>(t0 + j < len ? v : 0.f);
      }
  }
  __syncthreads();
  tile_to_global(ldsP, pitch, reinterpret_cast<unsigned char*>(static_cast<T*>(a.d_h_out) + (size_t)b * T_ * a.ldout), (size_t)a.ldout * ES, H * ES, TOUT, t0,
                 T_);
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


// =============================================================================================================================
// Operand packing: re-orders the arena's operands of a stack's layers into the order the kernels above consume them — per column
// tile and reduction step the 64 lanes' 16-byte MFMA fragments back to back — so that every weight load of a wave is 1 KiB of
// consecutive memory (loaded straight from the [c_out][c_in] rows a fragment load touches 32 cache lines for 32 bytes each, and
// the four k-steps that share those lines are a whole prefetch ring apart: measured 77 us per layer instead of ~10).
// One launch per stack per forward (the operands change with every optimizer step).
constexpr int kMaxPack = 64;
struct PackTable { vits_wn_pack_seg e[kMaxPack]; unsigned first[kMaxPack + 1]; int n; };

__global__ void wn_pack_kernel(PackTable tab) {
  const unsigned v = blockIdx.x * blockDim.x + threadIdx.x;                // destination 16-byte vector, over all segments
  if (v >= tab.first[tab.n]) return;
  int ei = 0;
#pragma unroll 1
  for (int i = 1; i < tab.n; ++i) if (v >= tab.first[i]) ei = i;
  const vits_wn_pack_seg& g = tab.e[ei];
  const unsigned lv = v - tab.first[ei];
  const int lane = lv & 63;
  const unsigned rest = lv >> 6;
  const int steps = g.taps * g.spt;
  const int s = rest % steps, tile = rest / steps;
  const int tap = s / g.spt, m = s % g.spt;
  int row, off;
  bool ok;
  if (g.mode == 2) {                                                        // 16-row tiles, 64-byte steps
    const int c = lane & 15, q = lane >> 4;
    row = 16 * tile + c; off = 64 * m + 16 * q;
    ok = row < g.rows && off < g.rowbytes;
  } else {
    const int c = lane & 31, h = lane >> 5;
    off = 32 * m + 16 * h;
    if (g.mode == 0) {                                                      // gate interleave: 16 tanh rows | their 16 sigmoid rows
      const int ch = 16 * tile + (c & 15);
      ok = 16 * tile < g.h; row = c < 16 ? ch : g.h + ch;
    } else {
      row = 32 * tile + c; ok = row < g.rows;
    }
    ok = ok && off < g.rowbytes;
  }
  u32x4 val = {0u, 0u, 0u, 0u};
  if (ok) val = *reinterpret_cast<const u32x4*>(static_cast<const unsigned char*>(g.src) + (size_t)tap * g.rows * g.rowbytes + (size_t)row * g.rowbytes + off);
  *reinterpret_cast<u32x4*>(static_cast<unsigned char*>(g.dst) + (size_t)lv * 16) = val;
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
    // 64-row tiles only when they still give every CU a workgroup; else 32-row tiles (twice the workgroups, half the LDS)
    const bool small = (long)vits::ceil_div(d.t, 64) * d.b < 256;
    if (nt == 1) return small ? launch_fwd<__bf16, 1, 1>(d, s) : launch_fwd<__bf16, 1, 2>(d, s);
    if (nt == 2) return small ? launch_fwd<__bf16, 2, 1>(d, s) : launch_fwd<__bf16, 2, 2>(d, s);
    return small ? launch_fwd<__bf16, 3, 1>(d, s) : launch_fwd<__bf16, 3, 2>(d, s);
  }
  if (nt == 1) return launch_fwd<float, 1, 1>(d, s);
  if (nt == 2) return launch_fwd<float, 2, 1>(d, s);
  return launch_fwd<float, 3, 1>(d, s);
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
    const int halo = (d.k - 1) * d.dil;
    const bool small = halo < 32 && (long)vits::ceil_div(d.t, 64 - halo) * d.b < 256;
    if (nt == 1) return small ? launch_bwd<__bf16, 2, 1>(d, s) : launch_bwd<__bf16, 4, 1>(d, s);
    if (nt == 2) return small ? launch_bwd<__bf16, 2, 2>(d, s) : launch_bwd<__bf16, 4, 2>(d, s);
    return small ? launch_bwd<__bf16, 2, 3>(d, s) : launch_bwd<__bf16, 4, 3>(d, s);
  }
  if (nt == 1) return launch_bwd<float, 2, 1>(d, s);
  if (nt == 2) return launch_bwd<float, 2, 2>(d, s);
  return launch_bwd<float, 2, 3>(d, s);
}

extern "C" size_t vits_wn_pack_bytes(int mode, int dtype, int h, int rows, int k_elems, int taps) {
  const int es = dtype == VITS_DT_BF16 ? 2 : 4;
  const int nt = vits::ceil_div(h, 64);
  const int spt = mode == 2 ? vits::ceil_div(k_elems * es, 64) : k_elems * es / 32;
  (void)rows;
  return (size_t)4 * nt * taps * spt * 1024;
}

extern "C" int vits_wn_pack(const vits_wn_pack_seg* segs, int count, void* stream) {
  if (!segs || count <= 0) return VITS_E_BADARG;
  hipStream_t s = static_cast<hipStream_t>(stream);
  for (int i0 = 0; i0 < count; i0 += kMaxPack) {
    PackTable tab;
    tab.n = count - i0 < kMaxPack ? count - i0 : kMaxPack;
    unsigned total = 0;
    for (int j = 0; j < tab.n; ++j) {
      const vits_wn_pack_seg& g = segs[i0 + j];
      if (!g.src || !g.dst || g.mode < 0 || g.mode > 2 || g.h <= 0 || g.h % 16 != 0 || g.h > 192 || g.rows <= 0 || g.rowbytes <= 0 || g.taps <= 0 ||
          g.spt <= 0 || g.rowbytes % 16 != 0)
        return VITS_E_BADARG;
      tab.e[j] = g;
      tab.first[j] = total;
      total += (unsigned)(4 * vits::ceil_div(g.h, 64) * g.taps * g.spt * 64);
    }
    tab.first[tab.n] = total;
    hipLaunchKernelGGL(wn_pack_kernel, dim3((total + 255) / 256), dim3(256), 0, s, tab);
  }
  return vits::check_launch("vits_wn_pack");
}
