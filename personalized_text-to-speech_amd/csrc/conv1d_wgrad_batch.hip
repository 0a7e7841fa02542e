// Weight (+ bias) gradients of a whole GROUP of stride-1 channels-last convolutions in one launch (vits_conv1d_cl_wgrad_batch).
//
//   dW_e[tap][co][ci] = sum_{b,t} dY_e[b][t][co] * X_e[b][t + tap*dil - pad][ci]          for every entry e of the table
//
// Why a table: the weight gradients of a layer stack (the 16 WaveNet layers of the posterior encoder, the 4 of a coupling layer,
// a DDSConv stack ...) are independent of each other once the stack's data gradients exist, and each of them alone has far too
// few 64 x 64 output tiles to fill 256 CUs — vits_conv1d_cl_wgrad therefore splits the (b, t) reduction of every layer over
// workgroups and pays for it with per-split fp32 slabs (4x the activation bytes, profiles/r02_pmc_wgrad.txt).  Launched
// together, the layers ARE the parallelism: every workgroup owns one (entry, co tile, ci tile, tap group) and walks the whole
// reduction, nothing is written but dW itself (S = 1).  Groups that still have too few tiles fall back to slabs (S > 1) and the
// caller's deferred second stage (vits_wgrad_reduce_pending) — same fixed summation order, same bits as the per-layer call.
//
// Inner loop (what the per-layer kernel measured as its bound: ~1 k MFMA cycles against ~5 k cycles of index arithmetic, range
// checks and exposed load latency per 128-row chunk):
//   * raw buffer loads with per-thread offsets computed ONCE; the descriptor's range check zero-fills rows outside the item
//     (the halo in front of row 0 wraps to a huge unsigned offset) and rows >= lengths[b]; no branch, no per-element test;
//   * TWO chunks of loads in flight in registers (the per-layer kernel keeps one), stored to LDS after the current chunk's MFMAs;
//   * operand fragments as there: ds_read_b64_tr_b16 (bf16) / ds_read_b32 (exact fp32), 192-byte pitch.
#include "common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

constexpr int TK = 128;          // rows per chunk
constexpr int CT = 64;           // channels per tile (co and ci)
constexpr int kThreads = 256;
constexpr int kMaxBatch = 20;
constexpr int XROWS_MAX = 160;   // staged X rows per chunk: TK + (KT - 1) * dil
constexpr int XROWS_WIDE = 184;  // ... of the all-taps-in-one-pass instances (KT = 7, 11: the decoder's k = 7 / 11 layers at dilation <= 5)
constexpr int xrows_of(int kt) { return kt > 4 ? XROWS_WIDE : XROWS_MAX; }

struct Entry {
  const void* x; const void* dy; float* dw; float* db; float* partial; const int* lengths;
  unsigned long long slab;       // floats per split slab (dw then db); 0 when S == 1
  int B, T, Cin, Cout, K, dil, pad, ldx, lddy, flags;
  int tiles_co, tiles_ci, tap_groups, S, first_block, accumulate;
  float in_slope;                // fused leaky-relu on x (1 = none), applied when the chunk is written to LDS
};
struct Table { Entry e[kMaxBatch]; int n; };

template <typename T> struct Pitch;
template <> struct Pitch<__bf16> { static constexpr int value = CT * 2 + 64; };   // 192 B: tr-reads conflict-free
template <> struct Pitch<float> { static constexpr int value = CT * 4 + 16; };

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
    const float f = to_f(in.e[i]);
    out.e[i] = from_f<T>(f > 0.f ? f : f * slope);
  }
  return out.u;
}

template <typename T, int KT>
__global__ __launch_bounds__(kThreads, (KT > 8 ? 1 : 2)) void wgrad_batch_kernel(Table tab) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int ES = sizeof(T);
  constexpr int V = 16 / ES;
  constexpr int PITCH = Pitch<T>::value;
  constexpr int VPR = CT / V;                         // 16-byte vectors per tile row
  constexpr int DV = TK * VPR / kThreads;             // dY vectors per thread per chunk (4 bf16 / 8 f32)
  constexpr int XR = xrows_of(KT);
  constexpr int XV = (XR * VPR + kThreads - 1) / kThreads;          // X vectors per thread per chunk (5 or 6 / 10)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wi = wave >> 1, wj = wave & 1;
  const int r = lane & 31, h = lane >> 5;

  // ---- which entry / tile / split
  int ei = 0;
#pragma unroll 1
  for (int i = 1; i < tab.n; ++i) if ((int)blockIdx.x >= tab.e[i].first_block) ei = i;
  const Entry& a = tab.e[ei];
  int local = blockIdx.x - a.first_block;
  const int split = local % a.S; local /= a.S;
  const int by = local % a.tiles_co; local /= a.tiles_co;
  const int bzc = local % a.tiles_ci, tg = local / a.tiles_ci;
  const int co0 = by * CT, ci0 = bzc * CT, tap0 = tg * KT;
  const int ntap = (a.K - tap0 < KT) ? (a.K - tap0) : KT;
  const int xrows = TK + (ntap - 1) * a.dil;
  const int cpi = (a.T + TK - 1) / TK;                // chunks per item
  const int n_chunks = a.B * cpi;

  unsigned char* ldsD = smem;                         // [TK][PITCH]      dY rows
  unsigned char* ldsX = smem + (size_t)TK * PITCH;    // [XROWS_MAX][PITCH]  X rows, first = tap0's

  // ---- per-thread byte offsets inside an item, computed once; 0xFFFFFFFF = never in range (columns beyond the tensor's channels)
  unsigned doff[DV], xoff[XV];
#pragma unroll
  for (int i = 0; i < DV; ++i) {
    const int idx = tid + i * kThreads, row = idx / VPR, vc = idx % VPR;
    doff[i] = (co0 + vc * V < a.Cout) ? (unsigned)((row * a.lddy + co0 + vc * V) * ES) : 0xFFFFFFFFu;
  }
#pragma unroll
  for (int i = 0; i < XV; ++i) {
    const int idx = tid + i * kThreads, row = idx / VPR, vc = idx % VPR;
    xoff[i] = (row < xrows && ci0 + vc * V < a.Cin) ? (unsigned)((row * a.ldx + ci0 + vc * V) * ES) : 0xFFFFFFFFu;
  }

  const bool lrelu = a.in_slope != 1.0f;
  u32x4 dr[2][DV], xr[2][XV];
  auto load_chunk = [&](int ch, u32x4 (&d)[DV], u32x4 (&x)[XV]) {
    // a chunk index beyond the end loads chunk 0's addresses with an empty range: every load stays unconditional
    const bool live = ch < n_chunks;
    const int b = live ? ch / cpi : 0, t0 = live ? (ch % cpi) * TK : 0;
    int len = a.lengths ? a.lengths[b] : a.T;
    len = len < a.T ? len : a.T;
    const int t_out_hi = live ? ((a.flags & VITS_CONV_MASK_OUT) ? len : a.T) : 0;
    const int t_in_hi = live ? ((a.flags & VITS_CONV_MASK_IN) ? len : a.T) : 0;
    const __amdgpu_buffer_rsrc_t rd = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<unsigned char*>(static_cast<const unsigned char*>(a.dy)) + (size_t)b * a.T * a.lddy * ES, 0, t_out_hi * a.lddy * ES, 0x00020000);
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<unsigned char*>(static_cast<const unsigned char*>(a.x)) + (size_t)b * a.T * a.ldx * ES, 0, t_in_hi * a.ldx * ES, 0x00020000);
    const unsigned dbase = (unsigned)(t0 * a.lddy * ES);
    const unsigned xbase = (unsigned)((t0 - a.pad + tap0 * a.dil) * a.ldx * ES);       // negative in front of the item: wraps out of range
#pragma unroll
    for (int i = 0; i < DV; ++i) {
      const unsigned o = doff[i] == 0xFFFFFFFFu ? 0xFFFFFFFFu : doff[i] + dbase;
      d[i] = __builtin_amdgcn_raw_buffer_load_b128(rd, (int)o, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < XV; ++i) {
      const unsigned o = xoff[i] == 0xFFFFFFFFu ? 0xFFFFFFFFu : xoff[i] + xbase;
      x[i] = __builtin_amdgcn_raw_buffer_load_b128(rx, (int)o, 0, 0);
    }
  };
  auto store_chunk = [&](const u32x4 (&d)[DV], const u32x4 (&x)[XV]) {
#pragma unroll
    for (int i = 0; i < DV; ++i) {
      const int idx = tid + i * kThreads;
      *reinterpret_cast<u32x4*>(ldsD + (idx / VPR) * PITCH + (idx % VPR) * 16) = d[i];
    }
#pragma unroll
    for (int i = 0; i < XV; ++i) {
      const int idx = tid + i * kThreads;
      if (idx < XR * VPR) *reinterpret_cast<u32x4*>(ldsX + (idx / VPR) * PITCH + (idx % VPR) * 16) = lrelu ? lrelu_vec<T>(x[i], a.in_slope) : x[i];
    }
  };

  f32x16 acc[KT];
#pragma unroll
  for (int k = 0; k < KT; ++k)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[k][i] = 0.f;
  const bool do_db = a.db != nullptr && bzc == 0 && tg == 0;
  float db_acc = 0.f;

  auto mma_chunk = [&]() {
    if (do_db) {
      const int col = tid & 63, q4 = tid >> 6;
      const T* dcol = reinterpret_cast<const T*>(ldsD) + col;
#pragma unroll 8
      for (int rr = q4 * (TK / 4); rr < (q4 + 1) * (TK / 4); ++rr)
        db_acc += to_f(*reinterpret_cast<const T*>(reinterpret_cast<const unsigned char*>(dcol) + (size_t)rr * PITCH));
    }
    if constexpr (sizeof(T) == 2) {
      // transposing reads (see conv1d_cl_wgrad.hip): within a 16-lane group lane 4q+p addresses row q, columns 4p..4p+3
      const int i16 = lane & 15, q = i16 >> 2, p = i16 & 3, g = lane >> 4;
      const int colA = (wi * 32 + 16 * (g & 1) + 4 * p) * 2;
      const int colB = (wj * 32 + 16 * (g & 1) + 4 * p) * 2;
      const int rowk = 8 * (g >> 1) + q;
#pragma unroll 2
      for (int s = 0; s < TK / 16; ++s) {
        union { s16x4 half[2]; bf16x8 v; } fa;
#pragma unroll
        for (int rd = 0; rd < 2; ++rd) {
          auto pa = reinterpret_cast<__attribute__((address_space(3))) s16x4*>(
              (__attribute__((address_space(3))) unsigned char*)ldsD + (16 * s + rowk + 4 * rd) * PITCH + colA);
          fa.half[rd] = __builtin_amdgcn_ds_read_tr16_b64_v4i16(pa);
        }
#pragma unroll
        for (int k = 0; k < KT; ++k) {
          union { s16x4 half[2]; bf16x8 v; } fb;
#pragma unroll
          for (int rd = 0; rd < 2; ++rd) {
            auto pb = reinterpret_cast<__attribute__((address_space(3))) s16x4*>(
                (__attribute__((address_space(3))) unsigned char*)ldsX + (16 * s + rowk + 4 * rd + k * a.dil) * PITCH + colB);
            fb.half[rd] = __builtin_amdgcn_ds_read_tr16_b64_v4i16(pb);
          }
          acc[k] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa.v, fb.v, acc[k], 0, 0, 0);
        }
      }
    } else {
      const float* dA = reinterpret_cast<const float*>(ldsD) + wi * 32 + r;
      const float* xB = reinterpret_cast<const float*>(ldsX) + wj * 32 + r;
      constexpr int PF = PITCH / 4;
#pragma unroll 4
      for (int s = 0; s < TK / 2; ++s) {
        const float av = dA[(2 * s + h) * PF];
#pragma unroll
        for (int k = 0; k < KT; ++k) {
          const float bv = xB[(2 * s + h + k * a.dil) * PF];
          acc[k] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[k], 0, 0, 0);
        }
      }
    }
  };

  // ---- chunk loop: chunks split, split + S, ...; two chunks of loads in flight.  (A tap beyond ntap multiplies rows that belong
  // to the next tap group — finite data — into an accumulator that is never written.)
  int ch = split;
  load_chunk(ch, dr[0], xr[0]);
  load_chunk(ch + a.S, dr[1], xr[1]);
  store_chunk(dr[0], xr[0]);
  __syncthreads();
  for (; ch < n_chunks; ch += 2 * a.S) {
    load_chunk(ch + 2 * a.S, dr[0], xr[0]);             // in flight during two chunks' MFMAs
    mma_chunk();
    if (ch + a.S < n_chunks) {
      __syncthreads();
      store_chunk(dr[1], xr[1]);
      __syncthreads();
      load_chunk(ch + 3 * a.S, dr[1], xr[1]);
      mma_chunk();
    }
    if (ch + 2 * a.S < n_chunks) {
      __syncthreads();
      store_chunk(dr[0], xr[0]);
      __syncthreads();
    }
  }

  float* P = a.S > 1 ? a.partial + (size_t)split * a.slab : a.dw;
  float* PB = a.S > 1 ? a.partial + (size_t)split * a.slab + (size_t)a.K * a.Cout * a.Cin : a.db;
  const bool add = a.S == 1 && a.accumulate;
  if (do_db) {                                            // 4 row-quarters -> one sum per column, fixed order
    __syncthreads();
    float* red = reinterpret_cast<float*>(smem);
    red[tid] = db_acc;
    __syncthreads();
    if (tid < 64 && co0 + tid < a.Cout) {
      const float v = red[tid] + red[tid + 64] + red[tid + 128] + red[tid + 192];
      PB[co0 + tid] = add ? PB[co0 + tid] + v : v;
    }
  }
  const int ci = ci0 + wj * 32 + r;
  if (ci < a.Cin) {
#pragma unroll
    for (int k = 0; k < KT; ++k)
      if (k < ntap) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          const int co = co0 + wi * 32 + (i & 3) + 8 * (i >> 2) + 4 * h;
          if (co < a.Cout) {
            const size_t idx = ((size_t)(tap0 + k) * a.Cout + co) * a.Cin + ci;
            P[idx] = add ? P[idx] + acc[k][i] : acc[k][i];
          }
        }
      }
  }
}

// Taps per workgroup.  k = 7 and k = 11 in bf16 (the decoder's resblocks) take all taps in ONE pass when their halo fits the
// wide staging buffer: with 4 taps per group such a layer was 2 / 3 tap groups, each re-reading x and dy (time_shapes: the
// decoder's batch ran at 0.93 TB/s algorithmic; A/B on one box, whole step: 20.96-21.03 against 21.32-21.37 ms).  KT = 11 keeps
// 176 accumulator registers: one workgroup per CU.  The same for k = 5 (the WaveNet layers, 2 groups) measured neutral: not done.
int taps_per_group(const vits_wgrad_desc& d) {
  const int k = d.k;
  if ((k == 7 || k == 11) && d.dtype == VITS_DT_BF16 && TK + (k - 1) * d.dil <= xrows_of(k)) return k;
  return k <= 4 ? k : (k <= 8 ? (k + 1) / 2 : 4);
}

template <typename T, int KT>
int launch(const Table& tab, int blocks, hipStream_t s) {
  constexpr int PITCH = Pitch<T>::value;
  const size_t lds = (size_t)(TK + xrows_of(KT)) * PITCH;
  auto kern = wgrad_batch_kernel<T, KT>;
  { const hipError_t e = vits::ensure_max_dynamic_lds(reinterpret_cast<const void*>(kern)); if (e != hipSuccess) return vits::note_hip_error(e, "vits_conv1d_cl_wgrad_batch/attr"); }
  hipLaunchKernelGGL(kern, dim3(blocks), dim3(kThreads), lds, s, tab);
  return vits::check_launch("vits_conv1d_cl_wgrad_batch");
}

template <typename T>
int dispatch(const Table& tab, int kt, int blocks, hipStream_t s) {
  switch (kt) {
    case 1: return launch<T, 1>(tab, blocks, s);
    case 2: return launch<T, 2>(tab, blocks, s);
    case 3: return launch<T, 3>(tab, blocks, s);
    case 7: if constexpr (sizeof(T) == 2) return launch<T, 7>(tab, blocks, s); else return VITS_E_UNSUPPORTED;     // (bf16 only, see taps_per_group)
    case 11: if constexpr (sizeof(T) == 2) return launch<T, 11>(tab, blocks, s); else return VITS_E_UNSUPPORTED;
    default: return launch<T, 4>(tab, blocks, s);
  }
}

bool eligible(const vits_wgrad_desc& d) {
  if (!d.x || !d.dy || !d.dw || d.b <= 0 || d.t <= 0 || d.c_in <= 0 || d.c_out <= 0 || d.k <= 0 || d.dil <= 0 || d.pad < 0) return false;
  if ((d.stride > 1) || d.groups > 1 || (d.flags & VITS_CONV_FLAT)) return false;
  if (d.t + 2 * d.pad - d.dil * (d.k - 1) != d.t) return false;                   // "same" convolutions only: t_out == t
  { const int kt = taps_per_group(d); if (TK + (kt - 1) * d.dil > xrows_of(kt)) return false; }
  const int vec = d.dtype == VITS_DT_BF16 ? 8 : (d.dtype == VITS_DT_F32 ? 4 : 0);
  if (vec == 0) return false;
  const int ldx = d.ldx > 0 ? d.ldx : d.c_in, lddy = d.lddy > 0 ? d.lddy : d.c_out;
  if (d.c_in % vec || d.c_out % vec || ldx % vec || lddy % vec) return false;
  if (((d.flags & (VITS_CONV_MASK_IN | VITS_CONV_MASK_OUT)) != 0) && !d.lengths) return false;
  if ((size_t)d.b * d.t * (ldx > lddy ? ldx : lddy) * (d.dtype == VITS_DT_BF16 ? 2 : 4) >= ((size_t)1 << 31)) return false;   // 32-bit buffer offsets
  return true;
}

}  // namespace

namespace {
// Splits of the (b, t) reduction per entry of ONE launch group: every workgroup should walk about the same number of 128-row
// chunks, about 1.25 resident rounds of workgroups in all (2 workgroups per CU by registers and LDS).  A group whose tiles alone
// fill the chip runs unsplit (S = 1: nothing but dw is written); long reductions over few tiles (the decoder's last stages:
// 1 tile x 1024 chunks) are cut into up to 64 slabs.
void plan_group(const vits_wgrad_desc* g, int m, int kt, int* S) {
  double work = 0;
  for (int j = 0; j < m; ++j) {
    const double tiles = (double)vits::ceil_div(g[j].c_out, CT) * vits::ceil_div(g[j].c_in, CT) * vits::ceil_div(g[j].k, kt);
    work += tiles * g[j].b * vits::ceil_div(g[j].t, TK);
  }
  int per_wg = (int)(work / 640.0) + 1;                 // chunks one workgroup should walk
  if (per_wg < 8) per_wg = 8;
  for (int j = 0; j < m; ++j) {
    const int chunks = g[j].b * vits::ceil_div(g[j].t, TK);
    int s = vits::ceil_div(chunks, per_wg);
    if (s > 64) s = 64;
    if (s > chunks) s = chunks;
    S[j] = s < 1 ? 1 : s;
  }
}
}  // namespace

// splits_out[i] = the number of slabs entry i of a vits_conv1d_cl_wgrad_batch call over the same array will use (1 = none): the
// caller gives every entry with splits > 1 a workspace of splits * (k*c_out*c_in + c_out) floats.
extern "C" int vits_conv1d_cl_wgrad_batch_plan(const vits_wgrad_desc* descs, int count, int* splits_out) {
  if (!descs || count <= 0 || count > 1024 || !splits_out) return VITS_E_BADARG;
  for (int i = 0; i < count; ++i)
    if (!eligible(descs[i]) || descs[i].dtype != descs[0].dtype) return VITS_E_UNSUPPORTED;      // (as the launch itself would)
  bool done[1024] = {false};
  for (int i0 = 0; i0 < count; ++i0) {
    if (done[i0]) continue;
    const int kt = taps_per_group(descs[i0]);
    int sel[kMaxBatch], m = 0;
    for (int i = i0; i < count && m < kMaxBatch; ++i)
      if (!done[i] && taps_per_group(descs[i]) == kt) { sel[m++] = i; done[i] = true; }
    vits_wgrad_desc grp[kMaxBatch];
    int S[kMaxBatch];
    for (int j = 0; j < m; ++j) grp[j] = descs[sel[j]];
    plan_group(grp, m, kt, S);
    for (int j = 0; j < m; ++j) splits_out[sel[j]] = S[j];
  }
  return VITS_OK;
}

extern "C" int vits_conv1d_cl_wgrad_batch(const vits_wgrad_desc* descs, int count, void* stream, vits_wgrad_pending* pending) {
  if (!descs || count <= 0) return VITS_E_BADARG;
  for (int i = 0; i < count; ++i) {
    if (!eligible(descs[i])) return VITS_E_UNSUPPORTED;
    if (descs[i].dtype != descs[0].dtype) return VITS_E_UNSUPPORTED;
  }
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (pending) for (int i = 0; i < count; ++i) pending[i].splits = 0;
  // one launch per (taps-per-group, <= kMaxBatch entries) group, in the caller's order
  bool done[1024] = {false};
  if (count > 1024) return VITS_E_UNSUPPORTED;
  for (int i0 = 0; i0 < count; ++i0) {
    if (done[i0]) continue;
    const int kt = taps_per_group(descs[i0]);
    int sel[kMaxBatch], m = 0;
    for (int i = i0; i < count && m < kMaxBatch; ++i)
      if (!done[i] && taps_per_group(descs[i]) == kt) { sel[m++] = i; done[i] = true; }
    vits_wgrad_desc grp[kMaxBatch];
    for (int j = 0; j < m; ++j) grp[j] = descs[sel[j]];
    int S[kMaxBatch];
    plan_group(grp, m, kt, S);
    Table tab;
    tab.n = m;
    int blocks = 0;
    for (int j = 0; j < m; ++j) {
      const vits_wgrad_desc& d = grp[j];
      const size_t n = (size_t)d.k * d.c_out * d.c_in, nb = d.dbias ? (size_t)d.c_out : 0;
      int Sj = S[j];
      if (Sj > 1 && (!pending || !d.workspace || d.workspace_bytes < (size_t)Sj * (n + nb) * sizeof(float))) Sj = 1;   // no room for slabs
      Entry& e = tab.e[j];
      e.x = d.x; e.dy = d.dy; e.dw = d.dw; e.db = d.dbias; e.partial = static_cast<float*>(d.workspace); e.lengths = d.lengths;
      e.slab = Sj > 1 ? n + nb : 0;
      e.B = d.b; e.T = d.t; e.Cin = d.c_in; e.Cout = d.c_out; e.K = d.k; e.dil = d.dil; e.pad = d.pad;
      e.ldx = d.ldx > 0 ? d.ldx : d.c_in; e.lddy = d.lddy > 0 ? d.lddy : d.c_out; e.flags = d.flags;
      e.tiles_co = vits::ceil_div(d.c_out, CT); e.tiles_ci = vits::ceil_div(d.c_in, CT); e.tap_groups = vits::ceil_div(d.k, kt);
      e.S = Sj; e.first_block = blocks; e.accumulate = (d.flags & VITS_CONV_ACCUM) ? 1 : 0;
      e.in_slope = d.in_slope;
      blocks += e.tiles_co * e.tiles_ci * e.tap_groups * Sj;
      if (Sj > 1) pending[sel[j]] = vits_wgrad_pending{e.partial, d.dw, d.dbias, n, nb, (size_t)e.slab, Sj, e.accumulate};
    }
    const int rc = descs[0].dtype == VITS_DT_BF16 ? dispatch<__bf16>(tab, kt, blocks, s) : dispatch<float>(tab, kt, blocks, s);
    if (rc != VITS_OK) return rc;
  }
  return VITS_OK;
}
