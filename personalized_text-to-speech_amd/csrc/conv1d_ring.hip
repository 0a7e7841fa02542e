// Channels-last convolution as an implicit GEMM with a deep register-prefetch ring (bf16, gfx950).
//
//   Y[m][co] = epilogue( sum_{tap, ci} W[tap][co][ci] * X[b][t*stride + tap*dil - pad][ci] ),   m = b*Tout + t  ("flat" rows)
//
// Why a third convolution kernel: the other two (conv1d_cl.hip, conv1d_flat.hip) keep ONE stage of global loads in flight
// per workgroup and run at one workgroup per CU on the layers that matter (<= 256 tiles), so every stage exposes a full
// L2 / Infinity-Cache round trip (~1-1.5 us against ~0.4 us of MFMA work): the 1024-channel discriminator layers sat at
// 10 % of the matrix-core peak, latency-bound.  Here
//   * NR = 3 stages of W (one tap of one 64-channel chunk each) are in flight in registers — a workgroup that owns its CU
//     may spend 100+ VGPRs on that — loaded by `buffer_load_dwordx4` (the descriptor's range check returns zeros for rows
//     outside an item, so halo rows need no branches), written to a double-buffered LDS image one stage ahead of their use;
//     ONE barrier per stage.  (An LDS-DMA ring, `buffer_load ... lds` with counted vmcnt, was built and measured first: its
//     ~100-cycle issue cost per 1 KB piece made the 21 pieces of a stage as expensive as the stage's 16 MFMAs per wave —
//     62 us on the 1024-channel layers against 136 us before; kept in the history of this file.)
//   * the X rows a tile needs are staged ONCE per 64-channel chunk — the rows of every item segment the tile touches with
//     their halo — and every tap reads them at a row offset, instead of re-gathering the rows per tap (5x less X traffic, k = 5);
//   * LDS rows are 128 B with the 16-byte column XOR-swizzled by (row >> 1) & 7: conflict-free ds_read_b128 for the 32x32x16
//     MFMA fragments at any row stride, no padding;
//   * workgroup ids are remapped so that the workgroups of one XCD walk the row tiles of one column tile: its weights stay in
//     that XCD's L2 (MI355X_MICROARCH.md: 8 XCDs, private L2s; T1).
// Tile: BM (128 | 64) flat rows x 128 output channels, 4 waves as 2 x 2, K step = one tap of one 64-channel chunk.
// Fused epilogue: bias, residual (before or after the multiplier), scale, lrelu' multiplier, output leaky-relu, output mask.
#include "common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

constexpr int kThreads = 256;
constexpr int BN = 128;                 // output channels per workgroup
constexpr int ROWB = 128;               // bytes per LDS row = 64 bf16 channels
constexpr int WSTAGE = BN * ROWB;       // bytes of one W buffer
constexpr int XI_MAX = 14;              // X rows per buffer <= 32 * XI_MAX (kernel instances for <= 6 and <= 14)
constexpr int NR = 3;                   // W stages in flight in registers

struct RingArgs {
  vits_conv_desc d;
  int Tout, M;                          // rows per item of the launch's row space (phases > 1: ceil(Tstore / phases)); flat rows B * Tout
  int XR;                               // LDS rows per X buffer (multiple of 32)
  int n_row_tiles, n_col_tiles;
  int phases, Tstore;                   // data gradient of a stride-`phases` convolution (in_div): see the kernel; Tstore = output rows per item
};

__device__ __forceinline__ float to_f(__bf16 v) { return (float)v; }

// The workgroup `id` of `nwg` of one problem (the kernels below: one problem per launch, or several side by side).
template <int BM, int XIM>
__device__ __forceinline__ void ring_body(const RingArgs& args, const int nwg, const int id) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const vits_conv_desc& a = args.d;
  constexpr int RI = BM / 64;                         // 32-row blocks per wave (wave tile = BM/2 rows x 64 columns)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave & 1, wn = wave >> 1;
  const int r = lane & 31, h = lane >> 5;
  const int M = args.M, dil = a.dil;

  // ---- tile of this workgroup (XCD-aware: ids that share an XCD walk the row tiles of one column tile)
  const int q8 = nwg >> 3, r8 = nwg & 7, xcd = id & 7;
  const int nid = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (id >> 3);
  const int col_tile = nid / args.n_row_tiles, row_all = nid - col_tile * args.n_row_tiles;
  const int co0 = col_tile * BN;
  // Data gradient of a stride-P convolution (in_div = P, dil = 1): output row u = phase + P q only sees the taps
  // tap0 + P j (tap0 = (pad - phase) mod P), whose input row is q + j + delta — i.e. every phase is a stride-1 convolution over q
  // with its own tap subset, and a tile never mixes phases.  P = 1 is the plain case (phase 0, all taps).
  const int P = args.phases;
  const int tiles_per_phase = args.n_row_tiles / P;
  const int phase = row_all / tiles_per_phase, row_tile = row_all - phase * tiles_per_phase;
  int tap0 = 0, k = a.k, pad = a.pad, s = a.stride;
  const int Tout = args.Tout;                              // rows per item in this launch's row space (P > 1: ceil(Tstore / P))
  if (P > 1) {
    tap0 = ((a.pad - phase) % P + P) % P;
    k = tap0 < a.k ? (a.k - tap0 + P - 1) / P : 0;
    pad = -((phase + tap0 - a.pad) / P);                   // exact division
    s = 1;
  }
  const int m0 = row_tile * BM;

  // ---- item segments of the tile: segment 0 = rows [t_first, t_first + n0) of item b_first, then whole items, then a tail
  const int b_first = m0 / Tout, t_first = m0 - b_first * Tout;
  const int rows_here = (M - m0 < BM) ? (M - m0) : BM;
  const int n0 = (Tout - t_first < rows_here) ? (Tout - t_first) : rows_here;
  const int halo = (k - 1) * dil + 1;
  const int L0 = (n0 - 1) * s + halo;
  const int rows_left = rows_here - n0;
  const int Lfull = (Tout - 1) * s + halo;

  unsigned char* const Xb = smem;                                  // [2][XR][128]
  unsigned char* const Wb = smem + (size_t)2 * args.XR * ROWB;     // [2][128][128]

  const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.x), 0, (int)((size_t)a.b * a.t * a.ldx * 2), 0x00020000);
  const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.w), 0, (int)((size_t)a.k * a.c_out * a.ldw * 2), 0x00020000);

  // ---- X fill: slot i of a thread covers LDS row 32 i + tid / 8, 16-byte column tid % 8 (source offset; >= 2^31 = zeros)
  const int XI = args.XR >> 5;
  const int frow = tid >> 3, c16 = tid & 7;
  unsigned xoff[XIM], xdst[XIM];
#pragma unroll
  for (int i = 0; i < XIM; ++i) {
    unsigned off = 0x80000000u;
    const int l = i * 32 + frow;
    if (i < XI) {
      int j, o, nj;
      if (l < L0) { j = 0; o = l; nj = n0; }
      else {
        const int u = l - L0;
        j = 1 + u / Lfull;
        o = u - (j - 1) * Lfull;
        nj = rows_left - (j - 1) * Tout;                           // rows of the tile left for segment j
        if (nj > Tout) nj = Tout;
      }
      const int item = b_first + j;
      const int tin = (j == 0 ? t_first * s : 0) - pad + o;
      bool ok = nj > 0 && o < (nj - 1) * s + halo && item < a.b && tin >= 0;
      if (ok) {
        int hi = a.t;
        if (a.flags & VITS_CONV_MASK_IN) { const int len = a.lengths[item]; hi = len < hi ? len : hi; }
        ok = tin < hi;
      }
      if (ok) off = (unsigned)(((size_t)item * a.t + tin) * a.ldx * 2) + (unsigned)(c16 << 4);
    }
    xoff[i] = off;
    xdst[i] = (unsigned)l * ROWB + (unsigned)((c16 ^ ((l >> 1) & 7)) << 4);
  }
  // ---- W fill: slot i covers column 32 i + tid / 8 of the tile
  unsigned woff[4], wdst[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int e = i * 32 + frow;
    int co = co0 + e;
    if (co >= a.c_out) co = a.c_out - 1;                           // columns beyond c_out are computed on a valid row and never stored
    woff[i] = (unsigned)((size_t)co * a.ldw * 2) + (unsigned)(c16 << 4);
    wdst[i] = (unsigned)e * ROWB + (unsigned)((c16 ^ ((e >> 1) & 7)) << 4);
  }
  // ---- LDS row of every MFMA row of this lane (tap 0), and the swizzle of its B columns
  int abase[RI];
#pragma unroll
  for (int i = 0; i < RI; ++i) {
    int m = m0 + wm * (BM / 2) + i * 32 + r;
    if (m >= M) m = M - 1;
    const int b = m / Tout, t = m - b * Tout, j = b - b_first;
    abase[i] = (j == 0) ? (t - t_first) * s : L0 + (j - 1) * Lfull + t * s;
  }
  unsigned bcol[2], bsw[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int col = wn * 64 + j * 32 + r;
    bcol[j] = (unsigned)col * ROWB;
    bsw[j] = (unsigned)((col >> 1) & 7) << 4;
  }

  f32x16 acc[RI][2];
#pragma unroll
  for (int i = 0; i < RI; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  const int n_chunks = a.c_in / 64;
  const int Q = n_chunks * k;                              // (k = 0: a phase no tap reaches — zero rows, straight to the epilogue)
  const unsigned wtap = (unsigned)((size_t)a.c_out * a.ldw * 2);   // bytes between taps of W

  // Every global load below is UNCONDITIONAL (stages / chunks / X slots that do not exist load from an out-of-range offset and
  // get zeros): with a load behind a branch the compiler can no longer count how many younger loads are in flight and falls
  // back to s_waitcnt vmcnt(0) before every LDS write — which would drain the whole ring each stage.
  u32x4 wreg[NR][4], xreg[XIM];
  auto load_w = [&](int set, int c, int tap) {
    const unsigned sb = c < n_chunks ? (unsigned)(tap0 + P * tap) * wtap + (unsigned)c * 128u : 0x80000000u;
#pragma unroll
    for (int i = 0; i < 4; ++i) wreg[set][i] = __builtin_amdgcn_raw_buffer_load_b128(rw, (int)(woff[i] + sb), 0, 0);
  };
  auto store_w = [&](int set, int buf) {
    unsigned char* dst = Wb + (size_t)buf * WSTAGE;
#pragma unroll
    for (int i = 0; i < 4; ++i) *reinterpret_cast<u32x4*>(dst + wdst[i]) = wreg[set][i];
  };
  auto load_x = [&](int c) {
    const unsigned cb = c < n_chunks ? (unsigned)c * 128u : 0x80000000u;
#pragma unroll
    for (int i = 0; i < XIM; ++i) xreg[i] = __builtin_amdgcn_raw_buffer_load_b128(rx, (int)((xoff[i] | (cb & 0x80000000u)) + (cb & 0x7FFFFFFFu)), 0, 0);
  };
  auto store_x = [&](int buf) {
    unsigned char* dst = Xb + (size_t)buf * args.XR * ROWB;
#pragma unroll
    for (int i = 0; i < XIM; ++i)
      if (i < XI) {
        u32x4 v = xreg[i];
        if (a.in_slope != 1.0f) {                                 // fused input leaky-relu / relu: once per staged element, not per tap
          union { u32x4 u; __bf16 e[8]; } t;
          t.u = v;
#pragma unroll
          for (int e = 0; e < 8; ++e) { const float f = (float)t.e[e]; t.e[e] = (__bf16)(f > 0.f ? f : f * a.in_slope); }
          v = t.u;
        }
        *reinterpret_cast<u32x4*>(dst + xdst[i]) = v;
      }
  };

  // ---- prologue: stage 0 and X(0) into LDS; stages 1 .. NR and X(1) in flight
  load_x(0);
  load_w(0, 0, 0);
  store_x(0);
  store_w(0, 0);
  int ci = 0, tapi = 0;                  // chunk / tap of the next stage to load
  auto advance_i = [&]() { if (++tapi == k) { tapi = 0; ++ci; } };
  advance_i();
#pragma unroll
  for (int d = 1; d <= NR; ++d) {
    load_w(d % NR, ci, tapi);
    advance_i();
  }
  load_x(1);                             // (ci, tapi) now names stage NR + 1
  int c = 0, tap = 0;                    // chunk / tap of the stage being computed

  // Q is rounded up to a multiple of NR: the extra stages multiply zero weights (the register sets are named statically,
  // so the stage loop is unrolled NR times and carries no conditional stage)
  const int Qp = (Q + NR - 1) / NR * NR;
  for (int q0 = 0; q0 < Qp; q0 += NR) {
#pragma unroll
    for (int d = 0; d < NR; ++d) {
      const int q = q0 + d;
      __syncthreads();
      // ---- MFMAs of stage q: tap `tap` of chunk c.  The staging work for the next stages (registers -> LDS for stage q + 1, the
      // global loads of stage q + 1 + NR) is placed BEHIND the first k-step's MFMAs: it then issues while the matrix pipe is busy
      const unsigned char* xb = Xb + (size_t)(c & 1) * args.XR * ROWB;
      const unsigned char* wb = Wb + (size_t)(q & 1) * WSTAGE;
      unsigned arow[RI], asw[RI];
#pragma unroll
      for (int i = 0; i < RI; ++i) {
        const int row = abase[i] + tap * dil;
        arow[i] = (unsigned)row * ROWB;
        asw[i] = (unsigned)((row >> 1) & 7) << 4;
      }
#pragma unroll
      for (int mm = 0; mm < 4; ++mm) {
        const unsigned lc = (unsigned)(2 * mm + h) << 4;
        union { u32x4 u; bf16x8 v; } fa[RI], fb[2];
#pragma unroll
        for (int i = 0; i < RI; ++i) fa[i].u = *reinterpret_cast<const u32x4*>(xb + arow[i] + (lc ^ asw[i]));
#pragma unroll
        for (int j = 0; j < 2; ++j) fb[j].u = *reinterpret_cast<const u32x4*>(wb + bcol[j] + (lc ^ bsw[j]));
#pragma unroll
        for (int i = 0; i < RI; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i].v, fb[j].v, acc[i][j], 0, 0, 0);
        if (mm == 0) {
          // stage q + 1: registers -> LDS (its loads were issued NR stages ago), then its register set takes stage q + 1 + NR
          __builtin_amdgcn_sched_barrier(0);
          store_w((d + 1) % NR, (q + 1) & 1);
          if (tap == k - 1) {                                  // stage q + 1 opens chunk c + 1
            store_x((c + 1) & 1);
            load_x(c + 2);
          }
          load_w((d + 1) % NR, ci, tapi);
          advance_i();
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      if (++tap == k) { tap = 0; ++c; }
    }
  }

  // ---- epilogue.  The accumulators go through LDS (fp32, 64 rows x 128 columns at a time) so that every thread finishes
  // 8 adjacent channels of one row: 16-byte loads of res / mg_src, one division per row piece, one 16-byte store — the
  // per-element form (2-byte stores, a flag test per element) cost 30 of this kernel's 68 us on the 1024-channel layers.
  //   y = mask( lrelu_out( ((acc + bias [+ res]) * scale) * lrelu'(mg_src) [+ res] ) )      (res before or after, by RES_AFTER)
  constexpr int EP = 128 * 4 + 16;                     // fp32 row pitch of the staging tile (bytes)
  float* const stage = reinterpret_cast<float*>(smem);
  __bf16* Y = static_cast<__bf16*>(a.y);
  const __bf16* R = static_cast<const __bf16*>(a.res);
  const __bf16* MG = static_cast<const __bf16*>(a.mg_src);
  const float res_before = (R && !(a.flags & VITS_CONV_RES_AFTER)) ? 1.f : 0.f, res_after = (R && (a.flags & VITS_CONV_RES_AFTER)) ? 1.f : 0.f;
  const float oslope = (a.flags & VITS_CONV_OUT_LRELU) ? a.out_slope : 1.0f;
  const bool mask_out = (a.flags & VITS_CONV_MASK_OUT) != 0;
  const int cg = tid & 15, prow = tid >> 4;            // phase 2: thread = 8 columns x one of 16 rows per iteration
  const int co8 = co0 + cg * 8;
  float bias8[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) bias8[e] = (a.bias && co8 + e < a.c_out) ? a.bias[co8 + e] : 0.f;
  constexpr int PASSES = BM / 64;
#pragma unroll
  for (int pass = 0; pass < PASSES; ++pass) {
    __syncthreads();                                    // the staging tile (and, first, the operand buffers under it) is free
    if (BM == 64 || wm == pass) {
#pragma unroll
      for (int i = 0; i < RI; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int e = 0; e < 16; ++e) {
            const int row = (BM == 64 ? wm * 32 : i * 32) + (e & 3) + 8 * (e >> 2) + 4 * h;
            const int col = wn * 64 + j * 32 + r;
            *reinterpret_cast<float*>(reinterpret_cast<unsigned char*>(stage) + (size_t)row * EP + col * 4) = acc[i][j][e];
          }
    }
    __syncthreads();
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const int row = it * 16 + prow;
      const int m = m0 + pass * 64 + row;
      if (m >= M || co8 >= a.c_out) continue;
      const int bq = m / Tout, q_ = m - bq * Tout;
      const int u = phase + P * q_;                          // output row inside the item
      if (u >= args.Tstore) continue;
      const unsigned char* src = reinterpret_cast<const unsigned char*>(stage) + (size_t)row * EP + cg * 32;
      union { u32x4 u[2]; float f[8]; } v;
      v.u[0] = *reinterpret_cast<const u32x4*>(src);
      v.u[1] = *reinterpret_cast<const u32x4*>(src + 16);
      const size_t o = ((size_t)bq * args.Tstore + u) * a.ldy + co8;
      union { u32x4 u; __bf16 e[8]; } rv, gv, out;
      if (R) rv.u = *reinterpret_cast<const u32x4*>(R + o);
      if (MG) gv.u = *reinterpret_cast<const u32x4*>(MG + o);
      bool dead = false;
      if (mask_out) dead = u >= a.lengths[bq];
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float rr = R ? to_f(rv.e[e]) : 0.f;
        float x = (v.f[e] + bias8[e] + res_before * rr) * a.out_scale;
        if (MG) x *= (to_f(gv.e[e]) > 0.f) ? 1.0f : a.mg_slope;
        x += res_after * rr;
        x = x > 0.f ? x : x * oslope;
        out.e[e] = (__bf16)(dead ? 0.f : x);
      }
      if (co8 + 8 <= a.c_out) *reinterpret_cast<u32x4*>(Y + o) = out.u;
      else
        for (int e = 0; e < 8; ++e) if (co8 + e < a.c_out) Y[o + e] = out.e[e];
    }
  }
}

template <int BM, int XIM>
__global__ __launch_bounds__(kThreads) void conv1d_ring_kernel(RingArgs args) {
  ring_body<BM, XIM>(args, gridDim.x, blockIdx.x);
}

// Several independent problems of the same kernel instance in ONE launch (the same layer of the five period discriminators:
// same channels and taps, other row counts and other weights).  A discriminator layer alone is 208-304 workgroups on 256 CUs —
// one round at 81 % of the CUs, or two rounds with the second a fifth full; side by side the problems fill each other's tails,
// and four of five launch boundaries go.  Every problem's workgroup ids start at a multiple of 8, so that blockIdx & 7 — the XCD —
// is also its local id & 7 and the XCD-aware tile walk of the body holds; the padding workgroups exit at once.
constexpr int kMaxMulti = 8;
struct MultiArgs { RingArgs a[kMaxMulti]; int start[kMaxMulti + 1]; int n; };

template <int BM, int XIM>
__global__ __launch_bounds__(kThreads) void conv1d_ring_multi_kernel(MultiArgs m) {
  int p = 0;
#pragma unroll 1
  for (int i = 1; i < m.n; ++i) if ((int)blockIdx.x >= m.start[i]) p = i;
  const RingArgs& args = m.a[p];
  const int id = blockIdx.x - m.start[p], nwg = args.n_row_tiles * args.n_col_tiles;
  if (id >= nwg) return;                                  // (uniform over the workgroup)
  ring_body<BM, XIM>(args, nwg, id);
}

template <int BM, int XIM>
int launch_ring_multi(const MultiArgs& m, size_t lds, hipStream_t s) {
  auto kern = conv1d_ring_multi_kernel<BM, XIM>;
  hipError_t e = vits::ensure_max_dynamic_lds(reinterpret_cast<const void*>(kern));
  if (e != hipSuccess) return vits::note_hip_error(e, "vits_conv1d_cl_multi(ring)/attr");
  hipLaunchKernelGGL(kern, dim3(m.start[m.n]), dim3(kThreads), lds, s, m);
  return vits::check_launch("vits_conv1d_cl_multi(ring)");
}

template <int BM, int XIM>
int launch_ring(const RingArgs& args, size_t lds, hipStream_t s) {
  auto kern = conv1d_ring_kernel<BM, XIM>;
  hipError_t e = vits::ensure_max_dynamic_lds(reinterpret_cast<const void*>(kern));
  if (e != hipSuccess) return vits::note_hip_error(e, "vits_conv1d_cl(ring)/attr");
  hipLaunchKernelGGL(kern, dim3(args.n_row_tiles * args.n_col_tiles), dim3(kThreads), lds, s, args);
  return vits::check_launch("vits_conv1d_cl(ring)");
}

}  // namespace

// Called by vits_conv1d_cl (d validated and defaulted, bf16).  Returns VITS_E_UNSUPPORTED when the shape is outside what this
// kernel handles; the caller then takes the other kernels.
namespace {
struct RingPlan { RingArgs args; int BM; bool wide; size_t lds; };

// VITS_OK and the plan, or VITS_E_UNSUPPORTED when the shape is outside what this kernel handles.
int ring_plan(const vits_conv_desc& d, int t_out, RingPlan* out) {
  if (d.dtype != VITS_DT_BF16 || d.c_in % 64 != 0 || d.k < 2 || d.groups > 1 || d.w_batch_stride != 0) return VITS_E_UNSUPPORTED;
  const int P = d.in_div > 1 ? d.in_div : 1;
  // data gradient of a strided convolution: dil 1, every phase keeps at least one tap, no length masks (the two grids differ)
  if (P > 1 && (d.stride != 1 || d.dil != 1 || d.k < P || (d.flags & (VITS_CONV_MASK_IN | VITS_CONV_MASK_OUT)))) return VITS_E_UNSUPPORTED;
  if (d.flags & (VITS_CONV_GATE | VITS_CONV_GATE_BWD)) return VITS_E_UNSUPPORTED;
  if (d.y2 != nullptr || d.bias_b != nullptr) return VITS_E_UNSUPPORTED;
  if (d.flags & (VITS_CONV_TANH | VITS_CONV_ACCUM)) return VITS_E_UNSUPPORTED;                 // (rare epilogues stay on the other kernels)
  if (d.ldy % 8 != 0 || d.c_out % 8 != 0) return VITS_E_UNSUPPORTED;
  if (d.ldx % 8 != 0 || d.ldw % 8 != 0) return VITS_E_UNSUPPORTED;
  const size_t xbytes = (size_t)d.b * d.t * d.ldx * 2, wbytes = (size_t)d.k * d.c_out * d.ldw * 2;
  if (xbytes >= (1ull << 31) || wbytes >= (1ull << 31)) return VITS_E_UNSUPPORTED;
  const int t_rows = (t_out + P - 1) / P;                     // rows per item in one phase
  const long M = (long)d.b * t_rows;
  if (M * P >= (1l << 30)) return VITS_E_UNSUPPORTED;
  const int n_col = vits::ceil_div(d.c_out, BN);
  // 128-row tiles when they still fill the chip, else 64-row tiles; too few tiles: the other kernels' smaller tiles win
  // (threshold 200 re-measured in round 3 against 120 / 260 / 520 in the whole step: each of them +0.1 to +0.3 ms)
  int BM = ((M + 127) / 128) * n_col * P >= 200 ? 128 : 64;
  if (((M + 63) / 64) * n_col * P < 100) return VITS_E_UNSUPPORTED;
  const int halo = P > 1 ? (d.k + P - 1) / P : (d.k - 1) * d.dil + 1;      // (phases: the longest tap subset)
  auto xrows = [&](int bm) {
    int nseg = (bm - 1) / t_rows + 2;
    if (nseg > d.b) nseg = d.b;
    int rows = d.stride * bm + nseg * (halo - d.stride);
    if (rows < halo) rows = halo;
    return (rows + 31) & ~31;
  };
  int XR = xrows(BM);
  auto fits = [&](int xr) { return (size_t)2 * xr * ROWB + (size_t)2 * WSTAGE <= (size_t)vits::kLdsBytesMax && xr <= 32 * XI_MAX; };
  if (!fits(XR) && BM == 128) { BM = 64; XR = xrows(BM); }
  if (!fits(XR)) return VITS_E_UNSUPPORTED;
  out->args = RingArgs{d, t_rows, (int)M, XR, P * (int)((M + BM - 1) / BM), n_col, P, t_out};
  out->BM = BM;
  out->wide = XR > 32 * 6;
  out->lds = (size_t)2 * XR * ROWB + (size_t)2 * WSTAGE;
  return VITS_OK;
}
}  // namespace

namespace vits {

int conv1d_ring_dispatch(const vits_conv_desc& d, int t_out, hipStream_t s) {
  RingPlan p;
  const int rc = ring_plan(d, t_out, &p);
  if (rc != VITS_OK) return rc;
  if (!p.wide) return p.BM == 128 ? launch_ring<128, 6>(p.args, p.lds, s) : launch_ring<64, 6>(p.args, p.lds, s);
  return p.BM == 128 ? launch_ring<128, 14>(p.args, p.lds, s) : launch_ring<64, 14>(p.args, p.lds, s);
}

// `count` problems (validated and defaulted by the caller like conv1d_ring_dispatch's) in one launch: only when every one of
// them is this kernel's and they agree on the instance; else VITS_E_UNSUPPORTED and nothing is launched.
int conv1d_ring_multi_dispatch(const vits_conv_desc* d, const int* t_out, int count, hipStream_t s) {
  if (count < 2 || count > kMaxMulti) return VITS_E_UNSUPPORTED;
  MultiArgs m;
  RingPlan p0;
  size_t lds = 0;
  int start = 0;
  for (int i = 0; i < count; ++i) {
    RingPlan p;
    const int rc = ring_plan(d[i], t_out[i], &p);
    if (rc != VITS_OK) return rc;
    if (i == 0) p0 = p;
    else if (p.BM != p0.BM) return VITS_E_UNSUPPORTED;
    if (p.wide) p0.wide = true;          // (the wide instance also runs the narrow problems: its extra X slots stay empty)
    lds = p.lds > lds ? p.lds : lds;
    m.a[i] = p.args;
    m.start[i] = start;
    start += (p.args.n_row_tiles * p.args.n_col_tiles + 7) & ~7;
  }
  m.start[count] = start;
  m.n = count;
  if (!p0.wide) return p0.BM == 128 ? launch_ring_multi<128, 6>(m, lds, s) : launch_ring_multi<64, 6>(m, lds, s);
  return p0.BM == 128 ? launch_ring_multi<128, 14>(m, lds, s) : launch_ring_multi<64, 14>(m, lds, s);
}

}  // namespace vits
