// Weight gradient of the channels-last convolution with large tiles and a deep register prefetch (bf16, gfx950).
//
//   dW[tap][co][ci] = sum_m dY[m][co] * act(X[b][t*stride + tap*dil - pad][ci]),   m = b*Tout + t  ("flat" rows)
//
// Why a second weight-gradient kernel: conv1d_cl_wgrad.hip works on 64 x 64 (co x ci) tiles of <= 4 taps with ONE stage of loads in
// flight.  On the layers that dominate the step that is bound twice over:
//   * the 512/1024-channel discriminator layers re-read a 128-row chunk of dY and X per 64 x 64 tile: 21 bytes of L2 traffic per
//     1000 MACs — 250 us for 37 GFLOP (147 TFLOP/s), L2-bandwidth-bound;
//   * the decoder's 128/256-channel layers run 3 workgroups per CU that each wait a full load round trip per chunk.
// Here a workgroup owns TC x TC channels (128 x 128, or 64 x 64 for narrow layers) x KT taps: 4 waves as 2 x 2, each wave
// (TC/2) x (TC/2) = RB x RB MFMA blocks per tap; the (b, t) reduction is walked in stages of TK flat rows with NR stages of global
// loads in flight in registers (every load unconditional — out-of-range rows come back as zeros from the buffer descriptor — so
// the compiler can count vmcnt instead of draining it); the X rows of a stage are staged ONCE with their halo, for all the
// item segments the stage touches, and every tap reads them at a row offset (as csrc/conv1d_ring.hip does).  MFMA operands are
// read column-wise with ds_read_b64_tr_b16.  The (b, t) reduction is split over blockIdx.x into fp32 slabs exactly like the
// other kernel (same slab layout, same second stage: reduce_slabs / vits_wgrad_reduce_pending, fixed order).
#include "common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

constexpr int kThreads = 256;

struct RArgs {
  const void* x; const void* dy; float* partial; float* partial_db;
  int B, T, Tout, M, Cin, Cout, K, dil, pad, stride, S;
  int ldx, lddy;
  int XR;                  // LDS rows of the X tile
  int n_ci_tiles;
  float in_slope, inv_tout;
  size_t slab;
};

template <int TC> struct Geo {
  static constexpr int PITCH = TC * 2 + 64;          // bytes per LDS row: transposing reads of 4 consecutive rows hit 4 x 16 distinct banks
  static constexpr int VPR = TC / 8;                 // 16-byte vectors per row
  static constexpr int RPS = kThreads / VPR;         // rows covered by one slot of all threads
  static constexpr int RB = TC / 64;                 // 32-wide MFMA blocks per wave and side
};

template <int TC, int KT, int TK, int XIM, int NR>
__global__ __launch_bounds__(kThreads) void wgrad_ring_kernel(RArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  __shared__ int xrow_s[TK];
  using G = Geo<TC>;
  constexpr int PITCH = G::PITCH, VPR = G::VPR, RPS = G::RPS, RB = G::RB;
  constexpr int DI = TK / RPS;                       // dY slots per thread
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wi = wave >> 1, wj = wave & 1;
  const int r = lane & 31, h = lane >> 5;
  const int split = blockIdx.x, by = blockIdx.y, bz = blockIdx.z;
  const int co0 = by * TC;
  const int ci0 = (bz % a.n_ci_tiles) * TC;
  const int tap0 = (bz / a.n_ci_tiles) * KT;
  const int ntap = (a.K - tap0 < KT) ? (a.K - tap0) : KT;
  const int s = a.stride, Tout = a.Tout, M = a.M;
  const int halo = (ntap - 1) * a.dil + 1;
  const int Lfull = (Tout - 1) * s + halo;
  const float inv_lfull = 1.0f / (float)Lfull;

  unsigned char* const ldsD = smem;                                  // [TK][TC] of dY
  unsigned char* const ldsX = smem + (size_t)TK * PITCH;             // [XR][TC] of act(X); row 0 = tap0 of the stage's first row

  const __amdgpu_buffer_rsrc_t rd = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.dy), 0, (int)((size_t)M * a.lddy * 2), 0x00020000);
  const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.x), 0, (int)((size_t)a.B * a.T * a.ldx * 2), 0x00020000);

  const int frow = tid / VPR, fcol = tid % VPR;                      // slot i of a thread: tile row i * RPS + frow, 16-byte column fcol
  const bool d_col_ok = co0 + fcol * 8 < a.Cout, x_col_ok = ci0 + fcol * 8 < a.Cin;
  const int XI = (a.XR + RPS - 1) / RPS;

  f32x16 acc[KT][RB][RB];
#pragma unroll
  for (int k = 0; k < KT; ++k)
#pragma unroll
    for (int i = 0; i < RB; ++i)
#pragma unroll
      for (int j = 0; j < RB; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[k][i][j][e] = 0.f;

  const int n_chunks = (M + TK - 1) / TK;
  const int n_stages = split < n_chunks ? (n_chunks - 1 - split) / a.S + 1 : 0;        // chunks split, split + S, ...

  // m / Tout without the integer-division sequence (m < 2^24: exact after the fix-up)
  auto div_tout = [&](int m) { int b = (int)((float)m * a.inv_tout); if (b * Tout > m) --b; if ((b + 1) * Tout <= m) ++b; return b; };

  u32x4 dreg[NR][DI], xreg[NR][XIM];
  auto load_stage = [&](int set, int q) {
    const bool live = q < n_stages;
    const int m0 = (split + q * a.S) * TK;
#pragma unroll
    for (int i = 0; i < DI; ++i) {
      const int m = m0 + i * RPS + frow;
      const unsigned off = (live && d_col_ok && m < M) ? (unsigned)(((size_t)m * a.lddy + co0 + fcol * 8) * 2) : 0x80000000u;
      dreg[set][i] = __builtin_amdgcn_raw_buffer_load_b128(rd, (int)off, 0, 0);
    }
    // item segments of the stage: segment 0 = rows [t0, t0 + n0) of item b0, then whole items
    const int b0 = div_tout(m0), t0 = m0 - b0 * Tout;
    const int rows_here = (M - m0 < TK) ? (M - m0) : TK;
    const int n0 = (Tout - t0 < rows_here) ? (Tout - t0) : rows_here;
    const int L0 = (n0 - 1) * s + halo;
    const int rows_left = rows_here - n0;
#pragma unroll
    for (int i = 0; i < XIM; ++i) {
      const int l = i * RPS + frow;
      unsigned off = 0x80000000u;
      if (i < XI && live && x_col_ok) {
        int j, o, nj;
        if (l < L0) { j = 0; o = l; nj = n0; }
        else {
          const int u = l - L0;
          int jj = (int)((float)u * inv_lfull);
          if (jj * Lfull > u) --jj;
          if ((jj + 1) * Lfull <= u) ++jj;
          j = 1 + jj;
          o = u - jj * Lfull;
          nj = rows_left - jj * Tout;
          if (nj > Tout) nj = Tout;
        }
        const int item = b0 + j;
        const int tin = (j == 0 ? t0 * s : 0) - a.pad + tap0 * a.dil + o;
        if (nj > 0 && o < (nj - 1) * s + halo && item < a.B && tin >= 0 && tin < a.T)
          off = (unsigned)((((size_t)item * a.T + tin) * a.ldx + ci0 + fcol * 8) * 2);
      }
      xreg[set][i] = __builtin_amdgcn_raw_buffer_load_b128(rx, (int)off, 0, 0);
    }
  };
  auto store_stage = [&](int set, int q) {
#pragma unroll
    for (int i = 0; i < DI; ++i) *reinterpret_cast<u32x4*>(ldsD + (size_t)(i * RPS + frow) * PITCH + fcol * 16) = dreg[set][i];
#pragma unroll
    for (int i = 0; i < XIM; ++i)
      if (i < XI) {
        u32x4 v = xreg[set][i];
        if (a.in_slope != 1.0f) {                                     // fused input leaky-relu: once per staged element
          union { u32x4 u; __bf16 e[8]; } t;
          t.u = v;
#pragma unroll
          for (int e = 0; e < 8; ++e) { const float f = (float)t.e[e]; t.e[e] = (__bf16)(f > 0.f ? f : f * a.in_slope); }
          v = t.u;
        }
        const int l = i * RPS + frow;
        if (l < a.XR) *reinterpret_cast<u32x4*>(ldsX + (size_t)l * PITCH + fcol * 16) = v;
      }
    // LDS row of tap0's input row of every row of the stage
    if (tid < TK) {
      const int m0 = (split + q * a.S) * TK;
      int m = m0 + tid;
      int row = 0;
      if (m < M) {
        const int b0 = div_tout(m0), t0 = m0 - b0 * Tout;
        const int rows_here = (M - m0 < TK) ? (M - m0) : TK;
        const int n0 = (Tout - t0 < rows_here) ? (Tout - t0) : rows_here;
        const int L0 = (n0 - 1) * s + halo;
        const int b = div_tout(m), t = m - b * Tout, j = b - b0;
        row = (j == 0) ? (t - t0) * s : L0 + (j - 1) * Lfull + t * s;
      }
      xrow_s[tid] = row;
    }
  };

  // bias gradient = column sums of dY, by the workgroups of ci tile 0 / tap group 0: thread = one column x one slice of the rows
  const bool do_db = a.partial_db != nullptr && bz == 0;
  constexpr int DBP = kThreads / TC;                                  // row slices
  float db_acc = 0.f;

  const int i16 = lane & 15, qq = i16 >> 2, p = i16 & 3, g = lane >> 4;
  const int rowk = 8 * (g >> 1) + qq;
  int colA[RB], colB[RB];
#pragma unroll
  for (int i = 0; i < RB; ++i) {
    colA[i] = (wi * (TC / 2) + i * 32 + 16 * (g & 1) + 4 * p) * 2;
    colB[i] = (wj * (TC / 2) + i * 32 + 16 * (g & 1) + 4 * p) * 2;
  }

#pragma unroll
  for (int d = 0; d < NR; ++d) load_stage(d, d);

  for (int q0 = 0; q0 < n_stages; q0 += NR) {
#pragma unroll
    for (int d = 0; d < NR; ++d) {
      const int q = q0 + d;
      if (q >= n_stages) break;
      __syncthreads();                                                // the previous stage's fragments have been read
      store_stage(d, q);
      load_stage(d, q + NR);                                          // in flight during the next NR stages' MFMAs
      __syncthreads();
      if (do_db) {
        const int col = tid % TC, part = tid / TC;
        const __bf16* dcol = reinterpret_cast<const __bf16*>(ldsD) + col;
#pragma unroll 8
        for (int rr = part * (TK / DBP); rr < (part + 1) * (TK / DBP); ++rr)
          db_acc += (float)*reinterpret_cast<const __bf16*>(reinterpret_cast<const unsigned char*>(dcol) + (size_t)rr * PITCH);
      }
#pragma unroll 2
      for (int st = 0; st < TK / 16; ++st) {
        union { s16x4 half[2]; bf16x8 v; } fa[RB];
        int xr2[2];
#pragma unroll
        for (int rd2 = 0; rd2 < 2; ++rd2) {
          const int row = 16 * st + rowk + 4 * rd2;
          xr2[rd2] = xrow_s[row];
#pragma unroll
          for (int i = 0; i < RB; ++i) {
            auto pa = reinterpret_cast<__attribute__((address_space(3))) s16x4*>(
                (__attribute__((address_space(3))) unsigned char*)ldsD + row * PITCH + colA[i]);
            fa[i].half[rd2] = __builtin_amdgcn_ds_read_tr16_b64_v4i16(pa);
          }
        }
#pragma unroll
        for (int k = 0; k < KT; ++k) {
          if (k < ntap) {
            union { s16x4 half[2]; bf16x8 v; } fb[RB];
#pragma unroll
            for (int rd2 = 0; rd2 < 2; ++rd2)
#pragma unroll
              for (int j = 0; j < RB; ++j) {
                auto pb = reinterpret_cast<__attribute__((address_space(3))) s16x4*>(
                    (__attribute__((address_space(3))) unsigned char*)ldsX + (xr2[rd2] + k * a.dil) * PITCH + colB[j]);
                fb[j].half[rd2] = __builtin_amdgcn_ds_read_tr16_b64_v4i16(pb);
              }
#pragma unroll
            for (int i = 0; i < RB; ++i)
#pragma unroll
              for (int j = 0; j < RB; ++j) acc[k][i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i].v, fb[j].v, acc[k][i][j], 0, 0, 0);
          }
        }
      }
    }
  }

  if (do_db) {                                                        // row slices -> one sum per column, fixed order
    __syncthreads();
    float* red = reinterpret_cast<float*>(smem);
    red[tid] = db_acc;
    __syncthreads();
    if (tid < TC && co0 + tid < a.Cout) {
      float sum = 0.f;
#pragma unroll
      for (int pp = 0; pp < DBP; ++pp) sum += red[tid + pp * TC];
      a.partial_db[(size_t)split * a.slab + co0 + tid] = sum;
    }
  }
  // slab of this split: partial[split][tap][co][ci]
  float* P = a.partial + (size_t)split * a.slab;
#pragma unroll
  for (int k = 0; k < KT; ++k)
    if (k < ntap) {
#pragma unroll
      for (int i = 0; i < RB; ++i)
#pragma unroll
        for (int j = 0; j < RB; ++j) {
          const int ci = ci0 + wj * (TC / 2) + j * 32 + r;
          if (ci >= a.Cin) continue;
#pragma unroll
          for (int e = 0; e < 16; ++e) {
            const int co = co0 + wi * (TC / 2) + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
            if (co < a.Cout) P[((size_t)(tap0 + k) * a.Cout + co) * a.Cin + ci] = acc[k][i][j][e];
          }
        }
    }
}

template <int TC, int KT, int TK, int XIM, int NR>
int launch_wr(const RArgs& a, hipStream_t s) {
  using G = Geo<TC>;
  const size_t lds = (size_t)(TK + a.XR) * G::PITCH;
  auto kern = wgrad_ring_kernel<TC, KT, TK, XIM, NR>;
  const hipError_t e = vits::ensure_max_dynamic_lds(reinterpret_cast<const void*>(kern), 1024);
  if (e != hipSuccess) return vits::note_hip_error(e, "vits_conv1d_cl_wgrad(ring)/attr");
  const dim3 grid(a.S, vits::ceil_div(a.Cout, TC), a.n_ci_tiles * vits::ceil_div(a.K, KT));
  hipLaunchKernelGGL(kern, grid, dim3(kThreads), lds, s, a);
  return vits::check_launch("vits_conv1d_cl_wgrad(ring)");
}

}  // namespace

namespace vits {

// Geometry the large-tile kernel would use for this layer (TC = 0: not eligible).  `s_max`: the split bound the caller's
// workspace was sized for.
WgradRingPlan wgrad_ring_plan(const vits_wgrad_desc& d, int t_out, int s_max) {
  WgradRingPlan p{0, 0, 0, 0, 0};
  if (d.dtype != VITS_DT_BF16 || d.groups > 1 || (d.flags & (VITS_CONV_MASK_IN | VITS_CONV_MASK_OUT)) || d.counters != nullptr) return p;
  if (d.c_in % 8 != 0 || d.c_out % 8 != 0 || d.ldx % 8 != 0 || d.lddy % 8 != 0) return p;
  if (d.c_in < 64 || d.c_out < 64 || d.k < 2) return p;
  // Measured (tools/ubench_wgrad.py): the large tiles pay off where the layer has MANY of them — the 1024-channel layers
  // (250 -> 110 us).  Narrow layers (<= 256 channels: one to four 128 x 128 tiles) get their parallelism from (b, t) splits, and
  // every split costs a slab, so the 64 x 64 kernel's 4x more tiles win there (decoder 128/256-channel layers: 66 -> 101 us with
  // this kernel); strided layers need stride x more staged X rows per stage and stay L2-bound either way (123 -> 128 us).
  // (VITS_CONV_BIG_TILES asks for this kernel wherever it can run: tests cover every instantiation that way)
  if (!(d.flags & VITS_CONV_BIG_TILES) && (d.stride != 1 || d.c_in < 512 || d.c_out < 512)) return p;
  const long M = (long)d.b * t_out;
  if (M >= (1l << 24) || M < 1024) return p;
  if ((size_t)M * d.lddy * 2 >= (1ull << 31) || (size_t)d.b * d.t * d.ldx * 2 >= (1ull << 31)) return p;
  const int TC = (d.c_in >= 128 && d.c_out >= 128) ? 128 : 64;
  const int TK = (d.stride > 1 && TC == 128) ? 64 : 128;
  const int KT = TC == 128 ? (d.k < 3 ? d.k : 3) : (d.k <= 4 ? d.k : (d.k <= 8 ? (d.k + 1) / 2 : 4));
  const int halo = (KT - 1) * d.dil + 1;
  int nseg = (TK - 1) / t_out + 2;
  if (nseg > d.b) nseg = d.b;
  int XR = d.stride * TK + nseg * (halo - d.stride);
  if (XR < halo) XR = halo;
  const int rps = kThreads / (TC / 8);
  XR = (XR + rps - 1) / rps * rps;
  const int xi = XR / rps;
  if (xi > (TC == 128 ? 14 : 8)) return p;
  if ((size_t)(TK + XR) * (TC * 2 + 64) > (size_t)kLdsBytesMax - 1024) return p;
  // splits: enough workgroups for two rounds of the chip, a multiple of 8 (the workgroups that share a chunk then share an XCD and
  // its L2), never more than the caller's workspace allows or than there are chunks
  const int tiles = ceil_div(d.c_out, TC) * ceil_div(d.c_in, TC) * ceil_div(d.k, KT);
  const int chunks = (int)((M + TK - 1) / TK);
  int S = ceil_div(512, tiles);
  if (S > 8) S = (S + 7) & ~7;
  if (S > s_max) S = s_max >= 8 ? (s_max & ~7) : s_max;
  if (S > chunks) S = chunks;
  if (S < 1) S = 1;
  p.TC = TC; p.TK = TK; p.KT = KT; p.XR = XR; p.S = S;
  return p;
}

int wgrad_ring_launch(const vits_wgrad_desc& d, int t_out, const WgradRingPlan& p, float* partial, float* partial_db, size_t slab, hipStream_t s) {
  RArgs a{d.x, d.dy, partial, partial_db, d.b, d.t, t_out, d.b * t_out, d.c_in, d.c_out, d.k, d.dil, d.pad, d.stride, p.S,
          d.ldx, d.lddy, p.XR, ceil_div(d.c_in, p.TC), d.in_slope, 1.0f / (float)t_out, slab};
  const int xi = p.XR / (kThreads / (p.TC / 8));
  if (p.TC == 128) {
    if (p.TK == 128) {
      if (p.KT == 3) return xi <= 10 ? launch_wr<128, 3, 128, 10, 2>(a, s) : launch_wr<128, 3, 128, 14, 2>(a, s);
      return xi <= 10 ? launch_wr<128, 2, 128, 10, 2>(a, s) : launch_wr<128, 2, 128, 14, 2>(a, s);
    }
    if (p.KT == 3) return launch_wr<128, 3, 64, 14, 2>(a, s);
    return launch_wr<128, 2, 64, 14, 2>(a, s);
  }
  switch (p.KT) {
    case 2: return launch_wr<64, 2, 128, 8, 3>(a, s);
    case 3: return launch_wr<64, 3, 128, 8, 3>(a, s);
    default: return launch_wr<64, 4, 128, 8, 3>(a, s);
  }
}

}  // namespace vits
