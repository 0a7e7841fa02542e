// Weight gradient of the channels-last 1-D convolution (csrc/conv1d_cl.hip) on the matrix cores.
//
//   dW[tap][co][ci] = sum_{b,t} dY[b][t][co] * act(X[b][t + tap*dil - pad][ci])
//
// GEMM view: M = co, N = ci, reduction = (b, t).  Both operands are reduced over their ROW index
// in the channels-last tiles, which is exactly what the MFMA wants once the tile is read
// column-wise:
//   bf16 : ds_read_b64_tr_b16 (gfx950's transposing LDS read) turns a 4(t) x 16(c) block into
//          per-lane columns — two reads make one 8-deep v_mfma_f32_32x32x16_bf16 fragment, and a
//          tap is a row offset of the X tile (any shift: only the COLUMN offset must be 8-byte
//          aligned, and it is);
//   f32  : v_mfma_f32_32x32x2_f32 takes one k per lane, so lane (i, h) simply reads row 2s+h,
//          column i: a conflict-free ds_read_b32.
// Workgroup = 4 waves = 64 co x 64 ci x a group of <= 4 taps (wave (i,j) owns the 32x32 block (i,j) and one
// accumulator per tap); the next chunk's tiles are prefetched into registers during the MFMAs.  The (b, t) reduction is split over blockIdx.x; every split writes its own
// fp32 slab and a second kernel sums the slabs in a fixed order (bitwise reproducible — no float
// atomics, cdna_hip_programming.md Guideline 12).
#include "common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

constexpr int TK = 128;          // time rows per staged chunk
constexpr int CT = 64;           // channels per tile (both co and ci)
constexpr int kThreads = 256;

struct WgradArgs {
  const void* x; const void* dy; float* partial; const int* lengths;
  int B, T, Tout, Cin, Cout, K, dil, pad, S, chunks_per_item;
  float in_slope;
  int flags;
  int ldx, lddy, stride;
  float inv_tout;         // 1 / Tout
  int groups;             // > 1: block-diagonal only, compact dw [k][c_out][c_in/groups]
  int* counters;          // non-null: the last split to finish a tile sums the slabs itself (fixed order), see below
  float* dw_final; float* db_final; int accumulate;
  float* partial_db;      // per-split bias-gradient sums (slab pitch `slab`), or null
  size_t slab;            // floats per split in `partial` (dw slab, optionally followed by the db slab)
};

__device__ __forceinline__ float to_f(float v) { return v; }
__device__ __forceinline__ float to_f(__bf16 v) { return (float)v; }
template <typename T> __device__ __forceinline__ T from_f(float v);
template <> __device__ __forceinline__ float from_f<float>(float v) { return v; }
template <> __device__ __forceinline__ __bf16 from_f<__bf16>(float v) { return (__bf16)v; }

template <typename T>
__device__ __forceinline__ u32x4 lrelu_vec(u32x4 raw, float slope) {
  constexpr int V = 16 / sizeof(T);
  union { u32x4 u; T e[V]; } in, out;
  in.u = raw;
#pragma unroll
  for (int i = 0; i < V; ++i) {
    float f = to_f(in.e[i]);
    out.e[i] = from_f<T>(f > 0.f ? f : f * slope);
  }
  return out.u;
}

template <typename T> struct Pitch;
template <> struct Pitch<__bf16> { static constexpr int value = CT * 2 + 64; };   // 192 B: tr-reads conflict-free
template <> struct Pitch<float> { static constexpr int value = CT * 4 + 16; };

constexpr int DV_MAX = 8;      // 16-byte vectors of the dY tile per thread (TK*VPR <= 256*DV_MAX)
constexpr int XV_MAX = 10;     // ... of the X tile (xrows*VPR <= 256*XV_MAX, else staged synchronously)
constexpr int XV_FLAT = 12;    // ... of the gathered per-tap X tiles of the flat-row variant (3 taps x 128 rows x 8 vectors)
constexpr int KT_FLAT = 3;

// One workgroup = one 64(co) x 64(ci) tile x one GROUP of at most KT taps x one split of the (b,t) reduction.
// Splitting the taps over workgroups multiplies the parallelism of large-k layers without any extra slab
// traffic (the tiles of x and dy are re-read from L2), and keeps the accumulators at KT*16 registers.
// SMALL = true: c_out <= 32 and c_in <= 32 (last decoder stage): the four waves share the single 32x32 block
// and split each chunk's 128 rows among themselves; their accumulators are summed through LDS at the end.
template <typename T, int KT, bool SMALL, bool FLAT>
__global__ __launch_bounds__(kThreads) void wgrad_kernel(WgradArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int V = 16 / sizeof(T);
  constexpr int PITCH = Pitch<T>::value;
  constexpr int VPR = CT / V;                 // 16-byte vectors per tile row
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wi = SMALL ? 0 : (wave >> 1), wj = SMALL ? 0 : (wave & 1);
  const int r = lane & 31, h = lane >> 5;
  const int n_ci_tiles = a.groups > 1 ? 1 : (a.Cin + CT - 1) / CT;
  const int bx = blockIdx.x, by = blockIdx.y, bz = blockIdx.z;      // (an XCD-aware remap of the ids was measured: no gain)
  const int co0 = by * CT;
  const int og = a.groups > 1 ? a.Cout / a.groups : 1, ig = a.groups > 1 ? a.Cin / a.groups : 1;
  // grouped: the one ci tile that holds this co tile's diagonal blocks (host guarantees it is a single tile)
  const int ci0 = a.groups > 1 ? ((co0 / og) * ig / CT) * CT : (bz % n_ci_tiles) * CT;
  const int tap0 = (bz / n_ci_tiles) * KT;
  const int ntap = (a.K - tap0 < KT) ? (a.K - tap0) : KT;
  // FLAT: chunks are TK consecutive rows of the joint (item, time) index and every tap has its own gathered
  // [TK][CT] tile (period-discriminator shapes: many short items, strided) — see csrc/conv1d_flat.hip.
  const int xrows = FLAT ? ntap * TK : (TK - 1) * a.stride + (ntap - 1) * a.dil + 1;
  constexpr int XV = FLAT ? XV_FLAT : XV_MAX;
  unsigned char* ldsD = smem;                               // [TK][CT] of dY
  unsigned char* ldsX = smem + (size_t)TK * PITCH;          // [xrows][CT] of act(X), first row = tap0's
  const int dvec = TK * VPR, xvec = xrows * VPR;
  const bool x_in_regs = xvec <= kThreads * XV;
  const bool lrelu = a.in_slope != 1.0f;

  f32x16 acc[KT];
#pragma unroll
  for (int k = 0; k < KT; ++k)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[k][i] = 0.f;

  u32x4 dr[DV_MAX], xr[XV];
  auto chunk_info = [&](int ch, int& b, int& t0, int& t_out_hi, int& t_in_hi) {
    if constexpr (FLAT) { b = 0; t0 = ch * TK; t_out_hi = a.B * a.Tout; t_in_hi = a.T; return; }
    b = ch / a.chunks_per_item;
    t0 = (ch % a.chunks_per_item) * TK;
    const int len = a.lengths ? a.lengths[b] : a.T;
    t_out_hi = (a.flags & VITS_CONV_MASK_OUT) ? (len < a.Tout ? len : a.Tout) : a.Tout;
    t_in_hi = (a.flags & VITS_CONV_MASK_IN) ? (len < a.T ? len : a.T) : a.T;
  };
  auto load_x_vec = [&](const T* X, int idx, int t0, int t_in_hi) -> u32x4 {
    const int row = idx / VPR, vc = idx % VPR;
    int t = t0 * a.stride - a.pad + tap0 * a.dil + row;
    const int ci = ci0 + vc * V;
    u32x4 v = {0u, 0u, 0u, 0u};
    if constexpr (FLAT) {                    // t0 = first flat row of the chunk; X = base of the whole tensor
      const int kk = row / TK, m = t0 + (row - kk * TK);
      if (m >= a.B * a.Tout) return v;
      int b = (int)((float)m * a.inv_tout);            // m / Tout without the integer-division sequence (m < 2^24: exact after the fix-up)
      if (b * a.Tout > m) --b;
      if ((b + 1) * a.Tout <= m) ++b;
      t = (m - b * a.Tout) * a.stride + (tap0 + kk) * a.dil - a.pad;
      if (a.flags & VITS_CONV_MASK_IN) { const int len = a.lengths[b]; t_in_hi = len < a.T ? len : a.T; }
      X += (size_t)b * a.T * a.ldx;
    }
    if (t >= 0 && t < t_in_hi && ci < a.Cin) {
      v = *reinterpret_cast<const u32x4*>(X + (size_t)t * a.ldx + ci);
    }
    return v;
  };
  auto load_chunk = [&](int ch) {
    int b, t0, t_out_hi, t_in_hi;
    chunk_info(ch, b, t0, t_out_hi, t_in_hi);
    const T* X = static_cast<const T*>(a.x) + (size_t)b * a.T * a.ldx;
    const T* DY = static_cast<const T*>(a.dy) + (size_t)b * a.Tout * a.lddy;
#pragma unroll
    for (int i = 0; i < DV_MAX; ++i) {
      const int idx = tid + i * kThreads;
      u32x4 v = {0u, 0u, 0u, 0u};
      if (idx < dvec) {
        const int row = idx / VPR, vc = idx % VPR;
        const int t = t0 + row, co = co0 + vc * V;
        bool ok = t < t_out_hi && co < a.Cout;
        if constexpr (FLAT) {
          if (ok && (a.flags & VITS_CONV_MASK_OUT)) { const int bb = t / a.Tout; ok = (t - bb * a.Tout) < a.lengths[bb]; }
        }
        if (ok) v = *reinterpret_cast<const u32x4*>(DY + (size_t)t * a.lddy + co);
      }
      dr[i] = v;
    }
    if (x_in_regs) {
#pragma unroll
      for (int i = 0; i < XV; ++i) {
        const int idx = tid + i * kThreads;
        xr[i] = (idx < xvec) ? load_x_vec(X, idx, t0, t_in_hi) : u32x4{0u, 0u, 0u, 0u};
      }
    }
  };
  auto store_chunk = [&](int ch) {
#pragma unroll
    for (int i = 0; i < DV_MAX; ++i) {
      const int idx = tid + i * kThreads;
      if (idx < dvec) *reinterpret_cast<u32x4*>(ldsD + (idx / VPR) * PITCH + (idx % VPR) * 16) = dr[i];
    }
    if (x_in_regs) {
#pragma unroll
      for (int i = 0; i < XV; ++i) {
        const int idx = tid + i * kThreads;
        // the fused leaky-relu is applied here, AFTER the chunk's MFMAs, so that the loads stay in flight during them
        if (idx < xvec) *reinterpret_cast<u32x4*>(ldsX + (idx / VPR) * PITCH + (idx % VPR) * 16) = lrelu ? lrelu_vec<T>(xr[i], a.in_slope) : xr[i];
      }
    } else {
      int b, t0, t_out_hi, t_in_hi;
      chunk_info(ch, b, t0, t_out_hi, t_in_hi);
      const T* X = static_cast<const T*>(a.x) + (size_t)b * a.T * a.ldx;
      for (int idx = tid; idx < xvec; idx += kThreads)
      {
        const u32x4 v = load_x_vec(X, idx, t0, t_in_hi);
        *reinterpret_cast<u32x4*>(ldsX + (idx / VPR) * PITCH + (idx % VPR) * 16) = lrelu ? lrelu_vec<T>(v, a.in_slope) : v;
      }
    }
  };

  // bias gradient = column sums of dY: done by the workgroups of ci-tile 0 / tap-group 0 on their staged tiles
  const bool do_db = (a.partial_db != nullptr) && (bz == 0);
  float db_acc = 0.f;

  const int n_chunks = FLAT ? (a.B * a.Tout + TK - 1) / TK : a.B * a.chunks_per_item;
  int ch = bx;
  if (ch < n_chunks) { load_chunk(ch); store_chunk(ch); }
  __syncthreads();
  for (; ch < n_chunks; ch += a.S) {
    const int nxt = ch + a.S;
    if (nxt < n_chunks) load_chunk(nxt);                  // in flight during this chunk's MFMAs
    if (do_db) {
      const int col = tid & 63, q4 = tid >> 6;
      const T* dcol = reinterpret_cast<const T*>(ldsD) + col;
#pragma unroll 8
      for (int rr = q4 * (TK / 4); rr < (q4 + 1) * (TK / 4); ++rr)
        db_acc += to_f(*reinterpret_cast<const T*>(reinterpret_cast<const unsigned char*>(dcol) + (size_t)rr * PITCH));
    }

    if constexpr (sizeof(T) == 2) {
      // transposing reads: within a 16-lane group, lane 4q+p addresses row q, columns 4p..4p+3 and
      // lane i receives column i of the 4 rows.  Group g = lane>>4 covers columns 16*(g&1).. of the
      // wave's 32, and the k half h = g>>1 (rows 8h..8h+7 of the 16-row step, two reads of 4 rows).
      const int i16 = lane & 15, q = i16 >> 2, p = i16 & 3, g = lane >> 4;
      const int colA = (wi * 32 + 16 * (g & 1) + 4 * p) * 2;
      const int colB = (wj * 32 + 16 * (g & 1) + 4 * p) * 2;
      const int rowk = 8 * (g >> 1) + q;
      const int s_lo = SMALL ? wave * (TK / 64) : 0, s_hi = SMALL ? (wave + 1) * (TK / 64) : TK / 16;
#pragma unroll 2
      for (int s = s_lo; s < s_hi; ++s) {
        union { s16x4 half[2]; bf16x8 v; } fa;
#pragma unroll
        for (int rd = 0; rd < 2; ++rd) {
          auto pa = reinterpret_cast<__attribute__((address_space(3))) s16x4*>(
              (__attribute__((address_space(3))) unsigned char*)ldsD + (16 * s + rowk + 4 * rd) * PITCH + colA);
          fa.half[rd] = __builtin_amdgcn_ds_read_tr16_b64_v4i16(pa);
        }
        // every tap of the group unconditionally (no branch between the LDS reads and the MFMAs: the reads of a whole step are
        // issued together).  A short last group (ntap < KT) multiplies stale LDS rows into accumulators that are never written.
#pragma unroll
        for (int k = 0; k < KT; ++k) {
          union { s16x4 half[2]; bf16x8 v; } fb;
#pragma unroll
          for (int rd = 0; rd < 2; ++rd) {
            auto pb = reinterpret_cast<__attribute__((address_space(3))) s16x4*>(
                (__attribute__((address_space(3))) unsigned char*)ldsX +
                (FLAT ? (k * TK + 16 * s + rowk + 4 * rd) : ((16 * s + rowk + 4 * rd) * a.stride + k * a.dil)) * PITCH + colB);
            fb.half[rd] = __builtin_amdgcn_ds_read_tr16_b64_v4i16(pb);
          }
          acc[k] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa.v, fb.v, acc[k], 0, 0, 0);
        }
      }
    } else {
      const float* dA = reinterpret_cast<const float*>(ldsD) + wi * 32 + r;
      const float* xB = reinterpret_cast<const float*>(ldsX) + wj * 32 + r;
      constexpr int PF = PITCH / 4;
      const int f_lo = SMALL ? wave * (TK / 8) : 0, f_hi = SMALL ? (wave + 1) * (TK / 8) : TK / 2;
#pragma unroll 4
      for (int s = f_lo; s < f_hi; ++s) {
        const float av = dA[(2 * s + h) * PF];
#pragma unroll
        for (int k = 0; k < KT; ++k) {
          const float bv = xB[(FLAT ? (k * TK + 2 * s + h) : ((2 * s + h) * a.stride + k * a.dil)) * PF];
          acc[k] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[k], 0, 0, 0);
        }
      }
    }
    if (nxt < n_chunks) {
      __syncthreads();
      store_chunk(nxt);
      __syncthreads();
    }
  }

  if (do_db) {                                            // 4 row-quarters -> one sum per column, fixed order
    __syncthreads();
    float* red = reinterpret_cast<float*>(smem);
    red[tid] = db_acc;
    __syncthreads();
    if (tid < 64 && co0 + tid < a.Cout)
      a.partial_db[(size_t)bx * a.slab + co0 + tid] = red[tid] + red[tid + 64] + red[tid + 128] + red[tid + 192];
    __syncthreads();
  }
  if constexpr (SMALL) {                                  // sum the four waves' accumulators (fixed order) into wave 0
    __syncthreads();
    float* red = reinterpret_cast<float*>(smem);          // [3 waves][KT][16][64] floats <= 48 KB
    if (wave > 0) {
#pragma unroll
      for (int k = 0; k < KT; ++k)
#pragma unroll
        for (int i = 0; i < 16; ++i) red[(((wave - 1) * KT + k) * 16 + i) * 64 + lane] = acc[k][i];
    }
    __syncthreads();
    if (wave == 0) {
#pragma unroll
      for (int w = 0; w < 3; ++w)
#pragma unroll
        for (int k = 0; k < KT; ++k)
#pragma unroll
          for (int i = 0; i < 16; ++i) acc[k][i] += red[((w * KT + k) * 16 + i) * 64 + lane];
    }
  }
  // slab of this split: partial[split][tap][co][ci]
  float* P = a.partial + (size_t)bx * a.slab;
  const int ci = ci0 + wj * 32 + r;
  const bool writer = ci < a.Cin && (!SMALL || wave == 0);
  // element index of accumulator (k, i) in dw's layout (dense, or compact for grouped layers); -1 = not an element of dw
  auto index_of = [&](int k, int i) -> long {
    const int co = co0 + wi * 32 + (i & 3) + 8 * (i >> 2) + 4 * h;
    if (co >= a.Cout) return -1;
    if (a.groups > 1) {
      const int ci_lo = (co / og) * ig;
      if (ci < ci_lo || ci >= ci_lo + ig) return -1;
      return ((long)(tap0 + k) * a.Cout + co) * ig + (ci - ci_lo);
    }
    return ((long)(tap0 + k) * a.Cout + co) * a.Cin + ci;
  };
  if (writer) {
#pragma unroll
    for (int k = 0; k < KT; ++k)
      if (k < ntap) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          const long idx = index_of(k, i);
          if (idx >= 0) P[idx] = acc[k][i];
        }
      }
  }
  if (a.counters == nullptr) return;

  // Fused second stage: every split publishes its slab (release), the LAST split to arrive at this tile sums the S slabs
  // in split order — the same fixed order as reduce_slabs, so the result is bitwise independent of which split is last —
  // and re-arms the counter for the next launch (no zero-fill between launches, nothing for a graph replay to forget).
  __shared__ int is_last;
  __threadfence();
  __syncthreads();
  const int tile = by * gridDim.z + bz;
  if (tid == 0) is_last = (atomicAdd(&a.counters[tile], 1) == a.S - 1);
  __syncthreads();
  if (!is_last) return;
  __threadfence();
  const float* base = a.partial;
  if (writer) {
#pragma unroll
    for (int k = 0; k < KT; ++k)
      if (k < ntap) {
#pragma unroll 4
        for (int i = 0; i < 16; ++i) {
          const long idx = index_of(k, i);
          if (idx < 0) continue;
          float sum = a.accumulate ? a.dw_final[idx] : 0.f;
          for (int sp = 0; sp < a.S; ++sp) sum += __builtin_nontemporal_load(base + (size_t)sp * a.slab + idx);
          a.dw_final[idx] = sum;
        }
      }
  }
  if (do_db && tid < 64 && co0 + tid < a.Cout) {
    float sum = a.accumulate ? a.db_final[co0 + tid] : 0.f;
    const float* pb = a.partial_db + co0 + tid;
    for (int sp = 0; sp < a.S; ++sp) sum += __builtin_nontemporal_load(pb + (size_t)sp * a.slab);
    a.db_final[co0 + tid] = sum;
  }
  if (tid == 0) a.counters[tile] = 0;
}

// out[i] (+)= sum_s partial[s * slab + i] in a fixed order; elements i < n go to dw, the following nb to db.
// One thread per 4 consecutive elements (n and nb are multiples of 4).
__global__ void reduce_slabs(const float* __restrict__ partial, float* __restrict__ dw, float* __restrict__ db, size_t n, size_t nb,
                             size_t slab, int S, int accumulate) {
  const size_t i = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;
  if (i >= n + nb) return;
  float* dst = (i < n) ? dw + i : db + (i - n);
  float4 acc = accumulate ? *reinterpret_cast<const float4*>(dst) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll 8
  for (int k = 0; k < S; ++k) {
    const float4 v = *reinterpret_cast<const float4*>(partial + (size_t)k * slab + i);
    acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
  }
  *reinterpret_cast<float4*>(dst) = acc;
}

// All second stages of a group of weight-gradient calls in ONE launch (vits_wgrad_reduce_pending): the table travels by value
// in the kernel arguments (no device table to upload, nothing for a captured graph to keep alive).
constexpr int kMaxPending = 48;
struct PendingTable { vits_wgrad_pending e[kMaxPending]; };

__global__ void reduce_pending_kernel(PendingTable tab) {
  const vits_wgrad_pending& p = tab.e[blockIdx.y];
  const size_t total = p.n + p.nb;
  for (size_t i = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * 4; i < total; i += (size_t)gridDim.x * blockDim.x * 4) {
    float* dst = (i < p.n) ? p.dw + i : p.dbias + (i - p.n);
    float4 acc = p.accumulate ? *reinterpret_cast<const float4*>(dst) : make_float4(0.f, 0.f, 0.f, 0.f);
    // (loads of 8 splits in flight at a time; the additions keep the split order, so the bits do not change)
#pragma unroll 8
    for (int k = 0; k < p.splits; ++k) {
      const float4 v = *reinterpret_cast<const float4*>(p.partial + (size_t)k * p.slab + i);
      acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
    }
    *reinterpret_cast<float4*>(dst) = acc;
  }
}

// upper bound of the (b,t) splits: narrow layers (one or a few 64 x 64 tiles: the last decoder stages) need many splits to put
// 2-3 workgroups on every CU — their chunks are latency-bound — and their slabs are tiny (k * 32 * 32 floats)
constexpr int kMaxSplits = 256;

int taps_per_group(int k) { return k <= 4 ? k : (k <= 8 ? (k + 1) / 2 : 4); }

// Number of (b,t)-reduction splits.  The grid should be ONE resident round of workgroups: a second, partly filled round
// doubles the kernel's time (measured, tools/ubench_wgrad.py: 592 workgroups 57 us, 512 workgroups 39 us on the decoder's
// 128-channel k = 7 layers).  Resident workgroups per CU: 2 by registers (acc + two prefetched tiles), 1 where the LDS tiles of
// the flat-row variant with 3 taps (98 KB) or the registers of the SMALL variant with 4 taps bound it.  Never more slab traffic
// than ~4x the reads of x and dy (each split writes and re-reads one dW).
int pick_splits(int b, int t_out, int c_in, int c_out, int k, bool flat = false) {
  const int kt = flat ? (k < KT_FLAT ? k : KT_FLAT) : taps_per_group(k);
  const int groups = vits::ceil_div(c_out, CT) * vits::ceil_div(c_in, CT) * vits::ceil_div(k, kt);
  const int chunks = flat ? vits::ceil_div(b * t_out, TK) : b * vits::ceil_div(t_out, TK);
  const bool small = !flat && c_out <= 32 && c_in <= 32;
  const int slots = ((flat && kt == 3) || (small && kt == 4)) ? 256 : 512;
  int s = groups <= slots ? slots / groups : vits::ceil_div(768, groups);
  const double io_elems = (double)b * t_out * (c_in + c_out);          // read once, 2 B (bf16) each
  const double dw_elems = (double)k * c_out * c_in;                    // each split writes + re-reads it in fp32: 8 B each
  const int s_traffic = (int)(io_elems / dw_elems) + 1;
  if (s > s_traffic) s = s_traffic;
  if (s > kMaxSplits) s = kMaxSplits;
  if (s > chunks) s = chunks;
  if (s < 1) s = 1;
  return s;
}

template <typename T, int KT, bool SMALL, bool FLAT = false>
int launch(const WgradArgs& a, hipStream_t s) {
  constexpr int PITCH = Pitch<T>::value;
  size_t lds = FLAT ? (size_t)(1 + KT) * TK * PITCH : (size_t)(TK + (TK - 1) * a.stride + (KT - 1) * a.dil + 1) * PITCH;
  if (SMALL && lds < (size_t)3 * KT * 16 * 64 * 4) lds = (size_t)3 * KT * 16 * 64 * 4;
  if (lds > (size_t)vits::kLdsBytesMax - 256) return VITS_E_UNSUPPORTED;      // minus the kernel's static LDS
  auto kern = wgrad_kernel<T, KT, SMALL, FLAT>;
  { const hipError_t e = vits::ensure_max_dynamic_lds(reinterpret_cast<const void*>(kern), 256); if (e != hipSuccess) return vits::note_hip_error(e, "vits_conv1d_cl_wgrad/attr"); }
  dim3 grid(a.S, vits::ceil_div(a.Cout, CT), (a.groups > 1 ? 1 : vits::ceil_div(a.Cin, CT)) * vits::ceil_div(a.K, KT));
  hipLaunchKernelGGL(kern, grid, dim3(kThreads), lds, s, a);
  return vits::check_launch("vits_conv1d_cl_wgrad");
}

template <typename T>
int dispatch_k(const WgradArgs& a, hipStream_t s, bool flat) {
  if (flat) {
    switch (a.K < KT_FLAT ? a.K : KT_FLAT) {
      case 1: return launch<T, 1, false, true>(a, s);
      case 2: return launch<T, 2, false, true>(a, s);
      default: return launch<T, 3, false, true>(a, s);
    }
  }
  const bool small = a.Cout <= 32 && a.Cin <= 32;
  switch (taps_per_group(a.K)) {
    case 1: return small ? launch<T, 1, true>(a, s) : launch<T, 1, false>(a, s);
    case 2: return small ? launch<T, 2, true>(a, s) : launch<T, 2, false>(a, s);
    case 3: return small ? launch<T, 3, true>(a, s) : launch<T, 3, false>(a, s);
    default: return small ? launch<T, 4, true>(a, s) : launch<T, 4, false>(a, s);
  }
}

}  // namespace

extern "C" size_t vits_conv1d_cl_wgrad_workspace(int b, int t_out, int c_in, int c_out, int k) {
  // the larger of the two variants' split counts: the caller does not know which kernel the launcher will take
  const int s0 = pick_splits(b, t_out, c_in, c_out, k, false), s1 = pick_splits(b, t_out, c_in, c_out, k, true);
  return (size_t)(s0 > s1 ? s0 : s1) * ((size_t)k * c_out * c_in + c_out) * sizeof(float);
}

static int wgrad_impl(const vits_wgrad_desc* desc, void* stream, vits_wgrad_pending* pending);

extern "C" int vits_conv1d_cl_wgrad(const vits_wgrad_desc* desc, void* stream) { return wgrad_impl(desc, stream, nullptr); }

extern "C" int vits_conv1d_cl_wgrad_deferred(const vits_wgrad_desc* desc, void* stream, vits_wgrad_pending* pending) {
  if (!pending) return VITS_E_BADARG;
  pending->splits = 0;
  return wgrad_impl(desc, stream, pending);
}

extern "C" int vits_wgrad_reduce_pending(const vits_wgrad_pending* list, int count, void* stream) {
  if (!list || count < 0) return VITS_E_BADARG;
  hipStream_t s = static_cast<hipStream_t>(stream);
  for (int i0 = 0; i0 < count; i0 += kMaxPending) {
    PendingTable tab;
    int m = 0;
    size_t longest = 0;
    for (int i = i0; i < count && m < kMaxPending; ++i) {
      if (list[i].splits <= 0) continue;               // that call wrote dw itself
      tab.e[m++] = list[i];
      if (list[i].n + list[i].nb > longest) longest = list[i].n + list[i].nb;
    }
    if (m == 0) continue;
    unsigned bx = (unsigned)((longest / 4 + 255) / 256);
    if (bx > 256) bx = 256;
    hipLaunchKernelGGL(reduce_pending_kernel, dim3(bx, m), dim3(256), 0, s, tab);
  }
  return vits::check_launch("vits_wgrad_reduce_pending");
}

static int wgrad_impl(const vits_wgrad_desc* desc, void* stream, vits_wgrad_pending* pending) {
  if (!desc) return VITS_E_BADARG;
  vits_wgrad_desc d = *desc;
  if (!d.x || !d.dy || !d.dw || !d.workspace || d.b <= 0 || d.t <= 0 || d.c_in <= 0 || d.c_out <= 0 || d.k <= 0 ||
      d.dil <= 0 || d.pad < 0)
    return VITS_E_BADARG;
  if (d.stride <= 0) d.stride = 1;
  const int span = d.t + 2 * d.pad - d.dil * (d.k - 1) - 1;
  if (span < 0) return VITS_E_BADARG;
  const int t_out = span / d.stride + 1;
  if (((d.flags & (VITS_CONV_MASK_IN | VITS_CONV_MASK_OUT)) != 0) && !d.lengths) return VITS_E_BADARG;
  // grouped: splits and slabs are sized by the compact dw, i.e. call the workspace function with c_in / groups
  const int c_in_eff = d.groups > 1 ? d.c_in / d.groups : d.c_in;
  if (d.workspace_bytes < vits_conv1d_cl_wgrad_workspace(d.b, t_out, c_in_eff, d.c_out, d.k)) return VITS_E_BADARG;
  if (d.ldx <= 0) d.ldx = d.c_in;
  if (d.lddy <= 0) d.lddy = d.c_out;
  // flat-row variant: strided layers and many short items (same rule as vits_conv1d_cl); never more splits than the
  // per-item variant, so the workspace bound above holds for both
  const bool auto_flat = true;
  if (d.groups > 1) {
    // block-diagonal tiles only: a 64-wide co tile must map into ONE 64-wide ci tile
    if (d.c_out % d.groups != 0 || d.c_in % d.groups != 0) return VITS_E_BADARG;
    const int og = d.c_out / d.groups, ig = d.c_in / d.groups;
    if (ig > og || CT % og != 0 || CT % ((CT / og) * ig) != 0) return VITS_E_UNSUPPORTED;
  }
  const bool flat = d.groups > 1 || d.stride > 1 || (d.flags & VITS_CONV_FLAT) != 0 || (auto_flat && t_out <= 80 && d.b >= 8);
  int S = pick_splits(d.b, t_out, c_in_eff, d.c_out, d.k, flat);
  // large-tile deep-prefetch kernel (csrc/conv1d_wgrad_ring.hip) where it applies: never more splits than the bound above
  const vits::WgradRingPlan ring = vits::wgrad_ring_plan(d, t_out, S);
  if (ring.TC) S = ring.S;
  const size_t n = (size_t)d.k * d.c_out * (d.groups > 1 ? d.c_in / d.groups : d.c_in), nb = d.dbias ? (size_t)d.c_out : 0;   // multiples of 4
  if ((size_t)S * (n + nb) * sizeof(float) > d.workspace_bytes) return VITS_E_BADARG;      // never write past the caller's slabs
  const bool accumulate = (d.flags & VITS_CONV_ACCUM) != 0;
  const bool direct = (S == 1) && !accumulate;           // a single split writes dw / db itself: no second launch
  float* ws = static_cast<float*>(d.workspace);
  // fused second stage (the last split of a tile sums the slabs): needs one zero-initialised, self-re-arming counter per tile
  const bool flat_k = flat;
  const int kt = flat_k ? (d.k < KT_FLAT ? d.k : KT_FLAT) : taps_per_group(d.k);
  const size_t tiles = (size_t)vits::ceil_div(d.c_out, CT) * (d.groups > 1 ? 1 : vits::ceil_div(d.c_in, CT)) * vits::ceil_div(d.k, kt);
  const bool fused = !direct && d.counters != nullptr && d.counters_len >= tiles;
  WgradArgs a{d.x, d.dy, direct ? d.dw : ws, d.lengths, d.b, d.t, t_out, d.c_in, d.c_out, d.k, d.dil, d.pad,
              S, vits::ceil_div(t_out, TK), d.in_slope, d.flags, d.ldx, d.lddy, d.stride, 1.0f / (float)t_out, d.groups,
              fused ? d.counters : nullptr, d.dw, d.dbias, accumulate ? 1 : 0,
              d.dbias ? (direct ? d.dbias : ws + n) : nullptr, direct ? 0 : n + nb};
  hipStream_t s = static_cast<hipStream_t>(stream);
  int rc;
  if (ring.TC) {
    rc = vits::wgrad_ring_launch(d, t_out, ring, a.partial, a.partial_db, a.slab, s);
  } else if (d.dtype == VITS_DT_BF16) {
    if (d.c_in % 8 != 0 || d.c_out % 8 != 0 || d.ldx % 8 != 0 || d.lddy % 8 != 0) return VITS_E_UNSUPPORTED;
    rc = dispatch_k<__bf16>(a, s, flat);
  } else if (d.dtype == VITS_DT_F32) {
    if (d.c_in % 4 != 0 || d.c_out % 4 != 0 || d.ldx % 4 != 0 || d.lddy % 4 != 0) return VITS_E_UNSUPPORTED;
    rc = dispatch_k<float>(a, s, flat);
  } else {
    return VITS_E_UNSUPPORTED;
  }
  if (rc != VITS_OK) return rc;
  if (direct || fused) return VITS_OK;
  if (pending) {                                         // second stage deferred to vits_wgrad_reduce_pending
    *pending = vits_wgrad_pending{a.partial, d.dw, d.dbias, n, nb, a.slab, a.S, accumulate ? 1 : 0};
    return VITS_OK;
  }
  hipLaunchKernelGGL(reduce_slabs, dim3((unsigned)(((n + nb) / 4 + 255) / 256)), dim3(256), 0, s, a.partial, d.dw, d.dbias, n, nb,
                     a.slab, a.S, accumulate ? 1 : 0);
  return vits::check_launch("vits_conv1d_cl_wgrad/reduce");
}
